#!/bin/bash
# usage: tools/profile_bench.sh <tag>  (on the GPU box)
# rocprofv3 kernel trace + stats of the default bench command, then the PMC
# passes; summaries land in gpurun_out/<tag>_* for copying into profiles/.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_trace.log || { tail -5 gpurun_out/${tag}_trace.log; exit 1; }
f=$(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
( head -1 $f; grep -E "enc_pack|enc_tile|enc_scan|dec_kernel|hist_kernel" $f ) > gpurun_out/${tag}_kernel_stats.csv
cat gpurun_out/${tag}_kernel_stats.csv
cat gpurun_out/${tag}_bench_under_rocprof.json
./tools/pmc.sh gpurun_out/${tag}_pmc zipf 1024 64 > gpurun_out/${tag}_pmc_summary.json || exit 1
cat gpurun_out/${tag}_pmc_summary.json
