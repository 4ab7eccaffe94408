#!/bin/bash
# usage: tools/profile_configs.sh <tag>  (on the GPU box)
# rocprofv3 kernel trace + stats of bench.py for every single-GPU configuration;
# the kernel_stats rows of our kernels land in gpurun_out/<tag>_<config>_kernel_stats.csv
# next to the bench line of the same (profiled) run (the bare-stream decode of its
# decode_nogap record included: gap_*_kernel).
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "zipf 16 10" "zipf 1 50" "uniform 1 30" "geometric 4 10"; do
  set -- $cfg
  name=${1}_${2}gib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace_$name -- python3 bench.py --kind $1 --gib $2 --steps $3 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_${name}_bench_under_rocprof.json 2> gpurun_out/${tag}_${name}_trace.log || { tail -5 gpurun_out/${tag}_${name}_trace.log; exit 1; }
  f=$(find gpurun_out/${tag}_trace_$name -name "*kernel_stats.csv" | head -1)
  ( head -1 $f; grep -E "enc_pack|enc_fix|enc_bytes|enc_tile|enc_scan|enc_plan|dec_kernel|hist_kernel|hist_batches|gap_" $f ) > gpurun_out/${tag}_${name}_kernel_stats.csv
  cat gpurun_out/${tag}_${name}_kernel_stats.csv
  rm -rf gpurun_out/${tag}_trace_$name
done
