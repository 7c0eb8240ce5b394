# Round-1 profile of the wavefront pipeline: one kernel-trace pass and separate PMC passes.
# usage (on the GPU box, from the repo root): bash tools/profile_r01.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r01
mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1 || exit 1
echo trace done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/fetch.log 2>&1 || exit 2
echo fetch done
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/write.log 2>&1 || exit 3
echo write done
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/l2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/l2.log 2>&1 || exit 4
echo l2 done
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/sq.log 2>&1 || exit 5
echo sq done
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ta -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/ta.log 2>&1 || exit 6
echo ta done
find $O -name "*.csv" | xargs ls -la
