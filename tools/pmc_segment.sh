# instruction mix / traffic of the segment kernel on a sphere scene (sky_map at 4K)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_seg; rm -rf $O; mkdir -p $O
B="python3 bench.py --scene sky_map --width 3840 --height 2160 --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -- $B > $O/a.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/b -- $B > $O/b.log 2>&1 || exit 2
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TA_TA_BUSY_sum --output-format csv -d $O/c -- $B > $O/c.log 2>&1 || exit 3
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- $B > $O/t.log 2>&1 || exit 4
echo done
