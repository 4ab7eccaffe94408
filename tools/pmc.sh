#!/bin/bash
# usage: tools/pmc.sh <outdir> <kind> <mib> <cs> [lib-debug] -- PMC passes over our kernels, one rocprofv3 run each
# (PROF_SCRIPT=tools/prof_gap.py profiles the bare-stream decode instead of the three main kernels)
out=$1; kind=${2:-zipf}; mib=${3:-1024}; cs=${4:-64}; dbg=${5:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --kernel-include-regex "enc_pack|enc_fix|enc_bytes|enc_plan|enc_scan|dec_kernel|hist_kernel|hist_batches|gap_" --output-format csv -d $out/p$i -- python3 ${PROF_SCRIPT:-tools/prof.py} $kind $mib $cs 1 $dbg > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out.p$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out
