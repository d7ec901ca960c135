for v in 0 3 4 5; do
export MI_DBG_WG=$v
tools/gprof.sh l1w$v $GRAFT_REPO_ROOT/tools/run_one_conv.py l1 wgrad 30 > /dev/null 2>&1
echo "dbg=$v $(python tools/stats_summary.py gpurun_out/l1w$v/out_kernel_stats.csv | head -1)"
done
