#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
OUT=$R/gpurun_out/pmc_gemm2
mkdir -p $OUT
cd /tmp
for c in "NT 32640 12288 4096" "NT 32640 4096 11008" "NT 12288 4096 32640"; do
  tag=$(echo $c | tr ' ' '_')
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo $grp | cut -d' ' -f1)
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${tag}_$t -- python3 $R/tools/gemm_case.py $c 256 3 > $OUT/${tag}_$t.log 2>&1 || echo fail
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc_gemm2/*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        a={k: sum(v)/len(v) for k,v in agg.items()}
        line=os.path.basename(d.rstrip('/'))+' '+str({k: round(v/1e6,2) for k,v in a.items()})
        if 'GRBM_GUI_ACTIVE' in a:
            simd_cycles=a['GRBM_GUI_ACTIVE']/8*1024
            line+=f" | MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES']/simd_cycles:.3f}"
        print(line)
    for f in glob.glob(d+'/**/*kernel_trace.csv', recursive=True):
        ds=[(float(r['End_Timestamp'])-float(r['Start_Timestamp'])) for r in csv.DictReader(open(f)) if 'gemm' in r['Kernel_Name']]
        if ds: print('   durations us', [round(x/1e3,1) for x in ds])
PY
