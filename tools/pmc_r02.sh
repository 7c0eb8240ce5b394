# PMC passes over one bench step (128 spp) for a traversal variant; usage: bash tools/pmc_r02.sh <variant> <tag>
# One counter block per run (rocprofv3 does not split counters over passes); no trace domains beside --pmc.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PTX_TRAV_VARIANT=${1:-0}
O=gpurun_out/pmc_r02_${2:-v0}
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -- $B > $O/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/b -- $B > $O/b.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/c -- $B > $O/c.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM --output-format csv -d $O/d -- $B > $O/d.log 2>&1 || echo "pass d failed (non-fatal)"
echo done $1
