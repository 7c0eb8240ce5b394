# Profile of the bench workload for one round: one kernel-trace pass and separate PMC passes (one counter block per run:
# rocprofv3 does not split counters over passes; no trace domain beside --pmc), condensed into profiles/<tag>_*.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag, e.g. r03> <commit id>   -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r03}
O=gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-strong-leg --no-loop-leg"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strong-leg --no-loop-leg > $O/trace.log 2>&1 || exit 1
echo trace done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/l2 -- $B > $O/l2.log 2>&1 || exit 4
echo hbm/l2 done
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1 || exit 6
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1 || exit 8
echo sq done
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ta -- $B > $O/ta.log 2>&1 || exit 7
echo ta done
python3 tools/summarize_round.py $O $T "${2:-unknown}" gpurun_out/profiles_$T
