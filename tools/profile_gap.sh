#!/bin/bash
# usage: tools/profile_gap.sh <tag> [kinds...]  (on the GPU box)
# rocprofv3 kernel trace of one hf_build_gap_array call on a 1 GiB stream per kind
tag=${1:-gap}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for kind in "${@:-zipf}"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace_$kind -- python3 tools/prof_gap.py $kind > gpurun_out/${tag}_${kind}.log 2>&1 || { tail -5 gpurun_out/${tag}_${kind}.log; exit 1; }
  f=$(find gpurun_out/${tag}_trace_$kind -name "*kernel_stats.csv" | head -1)
  ( head -1 $f; grep -E "gap_|dec_kernel" $f ) > gpurun_out/${tag}_${kind}_kernel_stats.csv
  cat gpurun_out/${tag}_${kind}_kernel_stats.csv
  rm -rf gpurun_out/${tag}_trace_$kind
done
