#!/bin/bash
# usage: tools/profile_bench.sh <tag>  (on the GPU box)
# rocprofv3 kernel trace + stats of the default bench command, then the PMC
# passes; summaries land in gpurun_out/<tag>_* for copying into profiles/.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_trace.log || { tail -5 gpurun_out/${tag}_trace.log; exit 1; }
f=$(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
( head -1 $f; grep -E "enc_pack|enc_bytes|enc_tile|enc_scan|enc_plan|dec_kernel|hist_kernel|hist_batches" $f ) > gpurun_out/${tag}_kernel_stats.csv
cat gpurun_out/${tag}_kernel_stats.csv
cat gpurun_out/${tag}_bench_under_rocprof.json
./tools/pmc.sh gpurun_out/${tag}_pmc zipf 1024 64 > gpurun_out/${tag}_pmc_summary.json || exit 1
./tools/pmc.sh gpurun_out/${tag}_pmc_geo geometric 1024 64 > gpurun_out/${tag}_pmc_geometric_summary.json || exit 1
./tools/pmc.sh gpurun_out/${tag}_pmc_uni uniform 1024 64 > gpurun_out/${tag}_pmc_uniform_summary.json || exit 1
python3 -c "
import json
for f in ['', '_geometric', '_uniform']:
    d = json.load(open('gpurun_out/${tag}_pmc%s_summary.json' % f))
    for k, v in d.items():
        print(f or '_zipf', k, 'read %.3f GB write %.3f GB' % (v.get('hbm_read_bytes_corrected', 0) / 1e9, v.get('hbm_write_bytes', 0) / 1e9))
"
