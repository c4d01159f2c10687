# Round-2 profile of the bench command (run on the GPU box through gpurun): kernel stats, HBM traffic and SQ counters
# of the headline kernel in separate rocprofv3 passes, then the plain bench line.  usage: bash tools/profile_bench.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02}
mkdir -p $O
H="--no-cpu-baseline --no-ess --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 200 --warmup 8 $H > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 4 --warmup 1 $H > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 4 --warmup 1 $H > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --steps 6 --warmup 2 $H > $O/sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/extras -- python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-ess > $O/extras.log 2>&1
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
