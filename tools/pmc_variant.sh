#!/bin/bash
# two PMC passes over the decode kernel for a variant library
out=$1; lib=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HF_TOOLS_LIB=$lib HF_TOOLS_NOCHECK=1
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --kernel-include-regex "dec_kernel" --output-format csv -d $out/p$i -- python3 tools/prof.py geometric 4096 64 1 > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out.p$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out
