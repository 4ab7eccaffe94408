"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel.

FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3.  On gfx950 FETCH_SIZE
counts a wide coalesced streaming read at half its bytes (64 B tallied per
128 B request), so the read side is doubled before it is compared with an
algorithmic byte count; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(enc_pack_kernel<[^>]*>|enc_pack_stream_kernel<[^>]*>|enc_fix_tiles_kernel|enc_bytes_kernel|enc_tile_bits_kernel|enc_scan_tiles_kernel|"
                      r"enc_plan_bits_kernel|dec_kernel_pair2<[^>]*>|gap_\w+_kernel(?:<[^>]*>)?|dec_kernel_reg<[^>]*>|dec_kernel<[^>]*>|hist_kernel<[^>]*>|"
                      r"hist_batches_kernel<[^>]*>)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
out = {}
for k, v in agg.items():
    per = {c: v[c] / max(1, calls[k][c]) for c in v}
    d = {c: per[c] for c in sorted(per)}
    if "FETCH_SIZE" in per:
        d["hbm_read_bytes_corrected"] = per["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in per:
        d["hbm_write_bytes"] = per["WRITE_SIZE"] * 1024
    out[k] = d
print(json.dumps(out, indent=1))
