cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ringprof
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/pooled_breakdown.py > $O/log.txt 2>&1
cat $O/*/*_kernel_stats.csv | head -12 | cut -c1-170
