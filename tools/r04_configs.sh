#!/bin/bash
# The other BASELINE configurations on the round-4 build, one box (outputs: gpurun_out/r04cfg/*.json)
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
O=$R/gpurun_out/r04cfg
mkdir -p $O
cd $R
run() { name=$1; shift; timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-batch "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return; }
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['unit'], d['step_frac_of_mfma_peak'], d['peak_hbm_gb'], d['roofline']['achieved'], d['config'].get('ga_groups'))"; }
run lora --lora-r 8 --loss grad_ascent --micro-batch 30 --grad-accum 1
run text --text-only --lora-r 8 --loss grad_ascent --micro-batch 64 --grad-accum 1
run npo --loss npo --grad-accum 1 --micro-batch 30
run kl13b --model 13b --loss KL --optim adamw8bit --micro-batch 12 --grad-accum 1
run scrub13b --model 13b --loss scrub --optim adamw8bit --micro-batch 10 --grad-accum 1
run kl13b_fp32_recompute --model 13b --loss KL --micro-batch 10 --grad-accum 1 --recompute
run npo_ref_batch --loss npo --micro-batch 2 --grad-accum 4
