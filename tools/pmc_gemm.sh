#!/bin/bash
# PMC passes for representative GEMM cases (separate rocprofv3 runs per counter group, kernel-trace only).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp
i=0
for c in "NT 8704 4096 11008" "NN 8704 4096 12288" "NN 8704 4096 4096" "TN 4096 11008 8704" "NT 8704 12288 4096"; do
  i=$((i+1))
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/c${i}_$tag -- python3 $R/tools/gemm_case.py $c 256 3 > $OUT/c${i}_$tag.log 2>&1 || echo "fail $c $grp"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc_gemm/c*_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        print(os.path.basename(d.rstrip('/')), {k: sum(v)/len(v) for k,v in agg.items()})
PY
