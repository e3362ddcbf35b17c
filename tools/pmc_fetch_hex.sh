#!/bin/bash
# FETCH_SIZE (one counter per pass: FETCH_SIZE + WRITE_SIZE together exceed what the hardware collects at once) of the cube-node kernel under the environment's switches: bash tools/pmc_fetch_hex.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_fetch_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-meshes --no-e2e --no-extras > $OUT/bench.json 2> $OUT/err.txt || { echo failed; tail -3 $OUT/err.txt; exit 1; }
python3 - $OUT $1 <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nin_gls_hex8" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {c: round(sum(v) / len(v) / 1024 / 1024, 3) for c, v in acc.items()}, "GiB (raw counter x KiB)")
PY
