cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/foldstat
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/pooled_breakdown.py > $O/log.txt 2>&1
grep -h "fold_moments\|panel_mfma_kernel<4, 0, 0\|panel_step_kernel<8, 64, 0" $O/*/*kernel_stats.csv | cut -c1-200
