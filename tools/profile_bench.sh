# Profile of the bench command (run on the GPU box through gpurun): kernel stats, HBM traffic and SQ counters
# of the headline kernel in separate rocprofv3 passes, then the plain bench line.  usage: bash tools/profile_bench.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04}
mkdir -p $O
H="--no-cpu-baseline --no-ess --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 200 --warmup 8 $H > $O/stats.log 2>&1
if [ "$2" = "stats-only" ]; then python3 $R/bench.py --no-extras > $O/bench_headline.json 2> $O/bench_headline.err; tail -1 $O/bench_headline.json | cut -c1-400; exit 0; fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 4 --warmup 1 $H > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 4 --warmup 1 $H > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --steps 6 --warmup 2 $H > $O/sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/extras -- python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-ess > $O/extras.log 2>&1
# the per-chain mode (SMCMC_MODE_PER_CHAIN): both kernels timed over the ensemble sizes; the streams of the one-chain-per-lane
# kernel (perchain_step_kernel) as HBM traffic in two passes
python3 $R/tools/perchain_time.py --kernel wave --chains 1 64 1024 4096 16384 65536 --steps 256 --launches 3 --json $O/perchain_wave.json > $O/perchain_wave.txt 2>&1
python3 $R/tools/perchain_time.py --kernel lane --chains 1 4096 16384 65536 --json $O/perchain.json > $O/perchain.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pc_fetch -- python3 $R/tools/perchain_time.py --kernel lane --chains 65536 --steps 16 --launches 2 > $O/pc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pc_write -- python3 $R/tools/perchain_time.py --kernel lane --chains 65536 --steps 16 --launches 2 > $O/pc_write.log 2>&1
# the large-dimension moment fold: old against new kernel (bitwise + time), then the new kernel's counters
$R/root-simple-mcmc_amd/build/micro/fold_bench 20 > $O/fold_bench.txt 2>&1
bash $R/tools/micro/fold_prof.sh $(basename $O)/fold > $O/fold_prof.txt 2>&1
$R/root-simple-mcmc_amd/build/micro/ordered_sum > $O/ordered_sum.txt 2>&1
# the reference-order large-dimension kernel: cycles per section (profiling build, tools/micro/build_panelprof.sh), its rates,
# and what the CU's LDS pipe charges for the broadcast reads its row loop lives on
python3 $R/tools/micro/panelprof.py 2>&1 | grep "cycles per step" | sort > $O/panel_sections.txt
python3 $R/tools/panel_time.py > $O/panel_time.txt 2>&1
$R/root-simple-mcmc_amd/build/micro/lds_bcast > $O/lds_bcast.txt 2>&1
# the per-chain wave kernel: cycles per section (tools/micro/build_pwprof.sh), the cost of the per-step record, the drop-in loop
python3 $R/tools/micro/pwprof.py 2>&1 | grep "cycles per step" > $O/perchain_wave_sections.txt
python3 $R/tools/record_cost.py 2>&1 | grep "us per step" > $O/record_cost.txt
python3 $R/tools/step_loop_time.py 2>&1 | grep steps_per_s > $O/step_loop.txt
# the flags the round's driver uses, on their own (short timed region)
python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-ess --no-cpu-baseline > $O/bench_driver_flags.json 2>> $O/bench.err
python3 $R/bench.py > $O/bench.json 2>> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
