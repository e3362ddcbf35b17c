#!/bin/bash
# FETCH_SIZE calibration on the access shapes of our kernels (tools/micro_fetch.hip): build, run once plain (timings), once under
# rocprofv3 --pmc FETCH_SIZE (its own pass, kernel-trace only), print counter vs known bytes.  Output: gpurun_out/micro_fetch.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/tools/_bin $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/micro_fetch.hip -o $R/tools/_bin/micro_fetch || exit 1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/micro_fetch
rm -rf $OUT; mkdir -p $OUT
$R/tools/_bin/micro_fetch > $OUT/plain.txt 2>&1 || { cat $OUT/plain.txt; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- $R/tools/_bin/micro_fetch > $OUT/pmc.txt 2> $OUT/pmc.err || { tail -5 $OUT/pmc.err; exit 1; }
python3 - $OUT <<'PY' | tee $R/gpurun_out/micro_fetch.txt
import csv, glob, sys, re
out = sys.argv[1]
known = {}
for l in open(out + "/plain.txt"):
    m = re.match(r"(\S+)\s+known\s+([\d.]+) MiB\s+([\d.]+) ms\s+(\d+) GB/s", l)
    if m: known[m.group(1)] = (float(m.group(2)), float(m.group(3)), int(m.group(4)))
order = list(known)
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
print(f"{'access shape':16s} {'known MiB':>10s} {'FETCH_SIZE MiB':>15s} {'ratio':>7s} {'ms':>8s} {'GB/s':>6s}")
for name, r in zip(order, rows):
    kib = float(r["Counter_Value"])
    k, ms, gbs = known[name]
    print(f"{name:16s} {k:10.1f} {kib / 1024:15.1f} {kib / 1024 / k:7.3f} {ms:8.3f} {gbs:6d}")
PY
