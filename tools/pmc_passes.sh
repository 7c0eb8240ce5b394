# extra SQ passes for the traversal kernel (instruction mix and wait reasons)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_mix
mkdir -p $O
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -- $B > $O/a.log 2>&1; echo a=$?
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/b -- $B > $O/b.log 2>&1; echo b=$?
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d $O/c -- $B > $O/c.log 2>&1; echo c=$?
