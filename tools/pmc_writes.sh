#!/bin/bash
# usage: tools/pmc_writes.sh <out> <kind> <mib> -- one PMC pass (WRITE_SIZE) over the bare-stream decode kernels
out=$1; kind=${2:-zipf}; mib=${3:-1024}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gap_decode|gap_emit" --output-format csv -d $out/p5 -- python3 tools/prof_gap.py $kind $mib > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 tools/pmc_summary.py $out
