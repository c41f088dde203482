#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
OUT=$R/gpurun_out/pmc_attn
mkdir -p $OUT
cd /tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  t=$(echo $grp | cut -d' ' -f1)
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$t -- python3 $R/tools/attn_bench.py 30 > $OUT/$t.log 2>&1 || echo fail
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc_attn/*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            for name in ('attn_fwd','attn_bwd_dq','attn_bwd_dkdv'):
                if name in k: agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
        for name,a in agg.items():
            m={k: sum(v)/len(v) for k,v in a.items()}
            line=name+' '+str({k: round(v/1e6,2) for k,v in m.items()})
            if 'GRBM_GUI_ACTIVE' in m:
                line+=f" | MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/(m['GRBM_GUI_ACTIVE']/8*1024):.3f}"
            print(line)
PY
