#!/bin/bash
# one SQ pass over the decode kernel of a library variant: pmc1.sh <variant> <kind> <mib> <debug>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
v=$1; kind=$2; mib=$3; dbg=$4
export HF_TOOLS_LIB=tools/_trace/lib_$v.so
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-include-regex "dec_kernel" --output-format csv -d gpurun_out/pmc1_$v -- python3 tools/prof.py $kind $mib 64 1 $dbg > gpurun_out/pmc1_$v.log 2>&1 || { echo "pass failed"; tail -5 gpurun_out/pmc1_$v.log; exit 1; }
tail -1 gpurun_out/pmc1_$v.log
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc1_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])]+=float(r["Counter_Value"])
for k in sorted(agg): print("$v", k, agg[k])
PY
