#!/bin/bash
# Same-box A/B of the pipelined optimiser (round 4): AdamW + W^T refresh on a side stream under the next step's forward vs the
# serial step, at the headline micro-batch and at the reference's (2 pairs x GA 4, merged).
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04ovl
mkdir -p $O
cd $R
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-batch > $O/head_overlap.json 2> $O/e1
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-batch --no-overlap-optimizer > $O/head_serial.json 2> $O/e2
python3 bench.py --micro-batch 2 --steps 8 --warmup 2 --no-cpu-baseline > $O/b2_overlap.json 2> $O/e3
python3 bench.py --micro-batch 2 --steps 8 --warmup 2 --no-cpu-baseline --no-overlap-optimizer > $O/b2_serial.json 2> $O/e4
python3 - <<PY
import json
for n in ("head_overlap", "head_serial", "b2_overlap", "b2_serial"):
    try:
        d = json.load(open("$O/%s.json" % n)); print(n, d["value"], d["ms_per_step"], d["step_frac_of_mfma_peak"], d["config"]["overlap_optimizer"], d["roofline"]["achieved"])
    except Exception as e:
        print(n, "failed", e); print(open("$O/e%d" % (("head_overlap", "head_serial", "b2_overlap", "b2_serial").index(n) + 1)).read()[-800:])
PY
