#!/bin/bash
# A/B timing of several library builds in ONE GPU session: bash tools/ab_bench_n.sh <rounds> lib1.so lib2.so ...
R=$1; shift
for i in $(seq 1 $R); do
  for lib in "$@"; do
    NINPOL_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
