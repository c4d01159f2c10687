cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/foldprof
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 $R/tools/pooled_breakdown.py > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/foldprof/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    if 'fold_moments' in k or 'panel_mfma_kernel<4, 0' in k:
        wc=v['SQ_WAVE_CYCLES']
        print(k, {n: round(x/wc,3) for n,x in v.items() if n.startswith('SQ_') and n!='SQ_WAVE_CYCLES'}, 'wave_cycles', wc)
PY
