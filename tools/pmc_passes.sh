cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_a -- $B > gpurun_out/pmc_a.log 2>&1; echo a=$?
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmc_b -- $B > gpurun_out/pmc_b.log 2>&1; echo b=$?
timeout -s KILL 120 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM --output-format csv -d gpurun_out/pmc_c -- $B > gpurun_out/pmc_c.log 2>&1; echo c=$?
