#!/bin/bash
# per-device micro-batch sweep {1, 2, 4, 8} pairs x GA 4 of the headline workload on the round-4 build (merged
# micro-batches where HBM allows).  Outputs: gpurun_out/r04sweep/b<B>.json
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04sweep
mkdir -p $O
cd $R
for B in 1 2 4 8; do
  timeout -k 10 300 python3 bench.py --micro-batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-reference-batch > $O/b$B.json 2> $O/b$B.err || echo "B=$B failed"
  python3 -c "import json; d=json.load(open('$O/b$B.json')); print('B=$B', d['value'], d['ms_per_step'], d['step_frac_of_mfma_peak'], d['config']['ga_groups'], d['peak_hbm_gb'], d['roofline']['achieved'])"
done
