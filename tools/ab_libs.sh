#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_libs.sh base.so new.so [kinds...]
base=$1; new=$2; shift 2
for k in "${@:-zipf}"; do
  for r in 1 2; do
    HF_TOOLS_LIB=$base timeout -k 10 120 python tools/time_kernels.py $k 1024 30 | sed 's/^/base /' || exit 1
    HF_TOOLS_LIB=$new timeout -k 10 120 python tools/time_kernels.py $k 1024 30 | sed 's/^/new  /' || exit 1
  done
done
