"""Records the HBM traffic per kernel launch measured by tools/pmc.sh (summary
JSON of tools/pmc_summary.py) in huffman_amd/pmc_traffic.json, keyed by the
workload, so that bench.py can report it next to the algorithmic bytes.
usage: python tools/make_traffic.py <pmc_summary.json> <kind> <bytes> <chunk_syms> <source label>"""
import json
import os
import sys

summary, kind, nbytes, cs, label = sys.argv[1:6]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "huffman_amd", "pmc_traffic.json")
try:
    table = json.load(open(path))
except (OSError, ValueError):
    table = {}
entry = {"source": label}
# the counting pass bench.py runs by default is hist_batches_kernel; the plain
# hist_kernel only stands in when the profile has no launch of the former
for name, v in sorted(json.load(open(summary)).items(), key=lambda kv: kv[0].startswith("hist_batches")):
    if "hbm_read_bytes_corrected" not in v or "hbm_write_bytes" not in v:
        continue
    key = ("histogram" if name.startswith("hist") else
           "encode" if name.startswith(("enc_pack", "enc_bytes")) else
           "decode" if name.startswith("dec_kernel") else
           "gap_sync" if name.startswith("gap_sync_kernel") else
           "gap_decode" if name.startswith("gap_decode_kernel") else None)
    if key:
        entry[key] = int(v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"])
        entry[key + "_kernel"] = name
# (a profile of the bare-stream decode only adds its own kernels to the workload's entry: the
# encoder and counting pass it runs on the way are not the ones bench.py times)
if any(k.startswith("gap_") for k in entry):
    gap = {k: v for k, v in entry.items() if k.startswith("gap_")}
    gap["gap_source"] = label
    entry = {**table.get(f"{kind}:{nbytes}:{cs}", {}), **gap}
table[f"{kind}:{nbytes}:{cs}"] = entry
json.dump(table, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(entry))
