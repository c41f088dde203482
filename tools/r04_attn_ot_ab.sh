#!/bin/bash
# Same-box A/B: a^T (o-projection wgrad operand) written by the attention forward vs transposed in the backward.
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04ot
mkdir -p $O
cd $R
for i in 1 2; do
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-batch > $O/on_$i.json 2> $O/e_on_$i
MMU_NO_ATTN_OT=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-batch > $O/off_$i.json 2> $O/e_off_$i
done
python3 - <<PY
import json
for n in ("on_1", "off_1", "on_2", "off_2"):
    d = json.load(open("$O/%s.json" % n)); k = {x["name"]: x for x in d["roofline"]["kernels"]}
    print(n, d["value"], d["ms_per_step"], d["peak_hbm_gb"], k.get("attn_fwd_kernel<128,causal>", {}).get("avg_us"), k.get("transpose", {}).get("launches"), k.get("transpose", {}).get("total_ms"))
PY
