#!/bin/bash
# usage: tools/profile_round.sh <tag>  (on the GPU box)
# The round's profile: kernel trace + stats of bench.py for every single-GPU
# configuration (tools/profile_configs.sh), then the PMC passes (tools/pmc.sh)
# over the same workloads; summaries land in gpurun_out/<tag>_* for copying
# into profiles/.
tag=${1:-r02}
./tools/profile_configs.sh $tag || exit 1
# (2, 4 and 8 GiB: the shards of the multi-GPU runs, for bench.py's roofline.traffic there)
for cfg in "zipf 1024" "uniform 1024" "geometric 4096" "zipf 16384" "zipf 8192" "zipf 4096" "zipf 2048"; do
  set -- $cfg
  ./tools/pmc.sh gpurun_out/${tag}_pmc_${1}_${2} $1 $2 64 > gpurun_out/${tag}_pmc_${1}_${2}mib_summary.json || { echo "pmc $cfg failed"; exit 1; }
  rm -rf gpurun_out/${tag}_pmc_${1}_${2}
done
# the bare-stream decode (gap_sync_kernel, gap_decode_kernel, gap_emit_kernel)
for cfg in "zipf 1024" "geometric 4096"; do
  set -- $cfg
  PROF_SCRIPT=tools/prof_gap.py ./tools/pmc.sh gpurun_out/${tag}_pmc_gap_${1}_${2} $1 $2 64 > gpurun_out/${tag}_pmc_gap_${1}_${2}mib_summary.json || { echo "pmc gap $cfg failed"; exit 1; }
  rm -rf gpurun_out/${tag}_pmc_gap_${1}_${2}
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${tag}_pmc_*_summary.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if "hbm_read_bytes_corrected" in v:
            print(f.split("/")[-1], k, "read %.3f GB write %.3f GB" % (v["hbm_read_bytes_corrected"] / 1e9, v.get("hbm_write_bytes", 0) / 1e9),
                  "lds conflict/active %.2f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, v.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
