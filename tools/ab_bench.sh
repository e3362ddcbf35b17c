#!/bin/bash
# A/B timing of library builds in ONE GPU session (devices differ by several per cent, so never compare across calls):
#   bash tools/ab_bench.sh tools/_bin/lib_A.so tools/_bin/lib_B.so [rounds]
R=${3:-3}
for i in $(seq 1 $R); do
  for lib in "$1" "$2"; do
    NINPOL_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
