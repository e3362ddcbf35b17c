#!/bin/bash
# per-kernel times of tools/time_methods.py on the given meshes: bash tools/prof_mesh.sh <tag> del40 [...]   -> gpurun_out/r04/kstats_<tag>.csv
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
NIN_METHODS=gls timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 tools/time_methods.py "$@" > $OUT/prof_$TAG.txt 2> $OUT/prof_$TAG.err || { echo "failed"; tail -5 $OUT/prof_$TAG.err; exit 1; }
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1:3]
st = glob.glob(out + f"/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(st)))
with open(out + f"/kstats_{tag}.csv", "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0].replace("void nin::(anonymous namespace)::", "").replace("nin::(anonymous namespace)::", "").split("(")[0]
        w.writerow(r)
for r in rows[:14]: print(",".join(r[:8])[:200])
PY
grep -h "Mnodes\|plan" $OUT/prof_$TAG.txt
rm -rf $OUT/prof_$TAG
