# rocprofv3 passes over tools/micro/fold_bench (run on the GPU box through gpurun): kernel stats, then SQ / LDS / TA
# counters of fold_ring_kernel on the config-4 ring case.  usage: bash tools/micro/fold_prof.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-foldprof}; B=$R/root-simple-mcmc_amd/build/micro/fold_bench
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B 10 "ring of 8" > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/sq1 -- $B 2 "config 4 share, ring of 8" > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -- $B 2 "config 4 share, ring of 8" > $O/sq2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq3 -- $B 2 "config 4 share, ring of 8" > $O/sq3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B 2 "config 4 share, ring of 8" > $O/fetch.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/tcc -- $B 2 "config 4 share, ring of 8" > $O/tcc.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for d in ("sq1", "sq2", "sq3", "fetch", "tcc"):
    fs = glob.glob(f"{O}/{d}/*/*_counter_collection.csv")
    if not fs:
        print(d, "no counters:", open(f"{O}/{d}.log").read()[-400:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(sorted(fs)[-1])):
        if "fold_ring_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    for k, v in acc.items():
        vals = [v[i] for i in sorted(v)]
        print(d, k, "last dispatch", vals[-1], "n", len(vals))
fs = glob.glob(f"{O}/stats/*/*_kernel_stats.csv")
if fs: print(open(sorted(fs)[-1]).read())
PY
