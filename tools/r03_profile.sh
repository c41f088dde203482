#!/bin/bash
# Round-3 evidence run on one MI355X box (outputs under gpurun_out/r03/; summaries are copied to profiles/ by hand):
# default bench line, rocprofv3 kernel stats of the same command, PMC traffic of the GEMM launches, PMC MFMA-busy /
# wait breakdown of the attention and GEMM kernels.  Usage: bash tools/r03_profile.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python3 bench.py --steps 5 --warmup 2 --hbm-kernels > $O/bench.json 2> $O/bench.err && tail -c 600 $O/bench.json
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-reference-batch > $O/bench_under_rocprof.json 2> $O/prof.err || echo "rocprof stats failed"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats.csv && head -12 $O/rocprofv3_kernel_stats.csv | cut -c1-160
bash $R/tools/pmc_bench_traffic.sh > $O/pmc_traffic.log 2>&1; cp $R/gpurun_out/pmc_bench/traffic.json $O/pmc_gemm_traffic.json 2>/dev/null; tail -2 $O/pmc_traffic.log | cut -c1-400
bash $R/tools/pmc_attn.sh > $O/pmc_attn.txt 2>&1; tail -6 $O/pmc_attn.txt | cut -c1-300
bash $R/tools/pmc_gemm2.sh > $O/pmc_gemm.txt 2>&1; tail -12 $O/pmc_gemm.txt | cut -c1-300
