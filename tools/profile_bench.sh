# Profile of the bench command (run on the GPU box through gpurun): kernel stats, HBM traffic and SQ counters
# of the headline kernel in separate rocprofv3 passes, then the plain bench line.  usage: bash tools/profile_bench.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03}
mkdir -p $O
H="--no-cpu-baseline --no-ess --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 200 --warmup 8 $H > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 4 --warmup 1 $H > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 4 --warmup 1 $H > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --steps 6 --warmup 2 $H > $O/sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/extras -- python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-ess > $O/extras.log 2>&1
# the per-chain mode's streams (SMCMC_MODE_PER_CHAIN, perchain_step_kernel): time, then HBM traffic in two passes
python3 $R/tools/perchain_time.py --json $O/perchain.json > $O/perchain.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pc_fetch -- python3 $R/tools/perchain_time.py --chains 65536 --steps 16 --launches 2 > $O/pc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pc_write -- python3 $R/tools/perchain_time.py --chains 65536 --steps 16 --launches 2 > $O/pc_write.log 2>&1
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
