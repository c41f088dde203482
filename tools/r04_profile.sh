#!/bin/bash
# Round-4 evidence run on one MI355X box (outputs under gpurun_out/r04/; summaries are copied to profiles/ by hand):
# default bench line, rocprofv3 kernel stats of the same command, PMC traffic of the GEMM launches, the reference-batch
# step merged vs looped and the SwiGLU-epilogue fusion on vs off on the SAME box.  Usage: bash tools/r04_profile.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python3 bench.py --steps 5 --warmup 2 --hbm-kernels > $O/bench.json 2> $O/bench.err && tail -c 400 $O/bench.json; echo
MMU_NO_SWIGLU_FUSION=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-batch > $O/bench_no_swiglu_fusion.json 2> $O/bench_nsf.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-batch > $O/bench_again.json 2> $O/bench_again.err
python3 bench.py --micro-batch 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_b2.json 2> $O/bench_b2.err
python3 bench.py --micro-batch 2 --steps 6 --warmup 2 --no-cpu-baseline --no-merge-ga > $O/bench_b2_looped.json 2> $O/bench_b2l.err
python3 - <<PY
import json
for n in ("bench", "bench_no_swiglu_fusion", "bench_again", "bench_b2", "bench_b2_looped"):
    try:
        d = json.load(open("$O/%s.json" % n)); print(n, d["value"], d["ms_per_step"], d["step_frac_of_mfma_peak"], d["config"].get("ga_groups"), d["roofline"]["achieved"])
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-reference-batch > $O/bench_under_rocprof.json 2> $O/prof.err || echo "rocprof stats failed"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats.csv && head -14 $O/rocprofv3_kernel_stats.csv | cut -c1-170
rm -rf $O/prof
bash $R/tools/pmc_bench_traffic.sh > $O/pmc_traffic.log 2>&1; cp $R/gpurun_out/pmc_bench/traffic.json $O/pmc_gemm_traffic.json 2>/dev/null; tail -2 $O/pmc_traffic.log | cut -c1-400
rm -rf $R/gpurun_out/pmc_bench/FETCH_SIZE $R/gpurun_out/pmc_bench/WRITE_SIZE
