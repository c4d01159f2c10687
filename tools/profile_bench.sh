cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-ess > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ess > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ess > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ess > $O/sq.log 2>&1
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-200
