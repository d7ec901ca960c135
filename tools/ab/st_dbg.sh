for v in 0 1 2 3; do
export MI_DBG_ST=$v
tools/gprof.sh st$v $GRAFT_REPO_ROOT/tools/run_one_conv.py stem fwd 30 > /dev/null 2>&1
echo "dbg=$v $(python tools/stats_summary.py gpurun_out/st$v/out_kernel_stats.csv | head -1)"
done
