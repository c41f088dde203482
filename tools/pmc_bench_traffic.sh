#!/bin/bash
# HBM traffic of the GEMM kernels of one bench step via rocprofv3 PMC (separate passes: FETCH_SIZE, WRITE_SIZE;
# kernel-trace only, no other trace domains).  Output: gpurun_out/pmc_bench/traffic.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
OUT=$R/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 1 --warmup 0 --grad-accum 1 --no-cpu-baseline --no-reference-batch > $OUT/$c.log 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv, glob, json, os
R=os.environ['GRAFT_REPO_ROOT']
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{R}/gpurun_out/pmc_bench/{c}/**/*counter_collection.csv", recursive=True)
    tot=0.0; n=0
    for r in csv.DictReader(open(f[0])):
        if 'gemm' in r['Kernel_Name'] and r['Counter_Name']==c:
            tot+=float(r['Counter_Value']); n+=1
    res[c]={"sum_kb":tot,"launches":n}
# gfx950: FETCH_SIZE counts 64 B per 128-B request for wide streaming reads -> x2 (MI355X_MICROARCH.md HBM section);
# WRITE_SIZE is exact for 8/16-B-per-lane streaming stores.  Units: KiB.
n=res["FETCH_SIZE"]["launches"]
import hashlib
out={"gemm_src_sha16": hashlib.sha256(open(f"{R}/multimodal_unlearning_amd/csrc/gemm.hip","rb").read()).hexdigest()[:16],
     "command":"rocprofv3 --kernel-trace --pmc <C> -- python3 bench.py --steps 1 --warmup 0 --grad-accum 1 --no-cpu-baseline --no-reference-batch",
     "gemm_launches":n,
     "fetch_bytes_per_launch": 2*res["FETCH_SIZE"]["sum_kb"]*1024/max(n,1),
     "write_bytes_per_launch": res["WRITE_SIZE"]["sum_kb"]*1024/max(res["WRITE_SIZE"]["launches"],1),
     "raw":res}
out["traffic_bytes_per_launch"]=out["fetch_bytes_per_launch"]+out["write_bytes_per_launch"]
json.dump(out, open(f"{R}/gpurun_out/pmc_bench/traffic.json","w"), indent=1)
print(json.dumps(out))
PY
