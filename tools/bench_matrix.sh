#!/bin/bash
# usage: tools_bench_matrix.sh <tag>  -- runs a few bench configurations, one JSON line each
tag=$1
for cfg in "--chunk-syms 64" "--chunk-syms 64 --no-plan --enc-mode 1" "--chunk-syms 64 --no-plan --enc-mode 0" "--chunk-syms 128 --enc-mode 1" "--kind uniform" "--kind geometric" "--kind testfile"; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $cfg 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    r = json.loads(line)
    print('$cfg', '| value', r['value'], '| ms', r['kernel_ms'], '| frac', r['hbm_frac'], '| ratio', r['config']['compressed_ratio'], 'maxlen', r['config']['max_code_len'], 'ok', r['roundtrip_ok'])
" || exit 1
done
