#!/bin/bash
# Regenerates the rocprofv3 evidence for profiles/<round>/ on the GPU box:
#   bash tools/profile_round.sh r01        (results under gpurun_out/prof_<round>/)
# Pass 1: --kernel-trace --stats of the default bench (no CPU-baseline leg).  Passes 2-4: separate --pmc runs
# (FETCH_SIZE; WRITE_SIZE; SQ counters), never combined with other trace domains.
set -u
ROUND=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-other-meshes --no-e2e > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { echo "stats pass failed"; tail -5 $OUT/stats.err; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-meshes --no-e2e > $OUT/pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.err; exit 1; }
done
# the block kernel by size class on the mixed and Kuhn-tet meshes (BASELINE config [3] at size: mixed10m)
NIN_GRID_BUILD=device timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mixed_tet -- python3 tools/time_methods.py mixed tet40 wedge60 mixed10m del54 delr40 delw100 > $OUT/methods_by_mesh.txt 2> $OUT/mixed_tet.err || { echo "mixed/tet pass failed"; tail -5 $OUT/mixed_tet.err; }
# issue-side counters of the one-wavefront multifrontal kernel on the Kuhn-tet mesh: the strip form and the row-lane form
bash tools/pmc_tet.sh $OUT/pmc_tet_strips > $OUT/pmc_tet_strips.txt 2>&1 || echo "tet pmc (strips) failed"
NIN_MFW_NO_STRIPS=1 bash tools/pmc_tet.sh $OUT/pmc_tet_rows > $OUT/pmc_tet_rows.txt 2>&1 || echo "tet pmc (rows) failed"
# ... and of the wide multifrontal kernel (all its size classes together) on a Delaunay mesh
bash tools/pmc_tet.sh $OUT/pmc_del_mfx del40 nin_gls_mfx > $OUT/pmc_del_mfx.txt 2>&1 || echo "delaunay pmc failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, collections, os
out = sys.argv[1]
def short(k):
    k = k.replace("void nin::(anonymous namespace)::", "").replace("nin::(anonymous namespace)::", "")
    return k.split("(")[0]
# kernel stats + head of the trace
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.reader(open(st))]
with open(out + "/kernel_stats.csv", "w", newline="") as f:
    csv.writer(f, quoting=csv.QUOTE_ALL).writerows(rows)
tr = glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True)[0]
with open(out + "/kernel_trace_head.csv", "w") as f:
    f.writelines(open(tr).readlines()[:40])
mt = glob.glob(out + "/mixed_tet/**/*kernel_stats.csv", recursive=True)
if mt:
    with open(out + "/kernel_stats_mixed_tet.csv", "w", newline="") as f:
        csv.writer(f, quoting=csv.QUOTE_ALL).writerows([r for r in csv.reader(open(mt[0]))])
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc[0-9]/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/pmc_tet_*/p[0-9]/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/pmc_del_*/p[0-9]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("nin_"): acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))   # (the apply form of the cube kernel, nin_gls_hex8w2_kernel<true>, is its own row)
with open(out + "/pmc_summary.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "counter", "dispatches", "average_per_dispatch"])
    for (k, c), v in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])): w.writerow([k, c, len(v), round(sum(v) / len(v), 1)])
avg = {kc: sum(v) / len(v) for kc, v in acc.items()}
def traffic(prefixes):   # bytes per launch: 2 x FETCH (gfx950 correction, MI355X_MICROARCH.md) + WRITE, KiB units
    t = 0.0
    for (k, c), v in avg.items():
        if any(k.startswith(p) for p in prefixes) and "<true>" not in k: t += v * 1024 * (2 if c == "FETCH_SIZE" else 1 if c == "WRITE_SIZE" else 0)   # (<true>: the apply leg, not the weights step)
    return int(t)
sys.path.insert(0, os.getcwd())
import bench
def traffic_lo(prefixes):   # the same with FETCH_SIZE taken as it reads: the lower bound when the reads are scattered gathers
    t = 0.0
    for (k, c), v in avg.items():
        if any(k.startswith(p) for p in prefixes) and "<true>" not in k: t += v * 1024 * (1 if c in ("FETCH_SIZE", "WRITE_SIZE") else 0)
    return int(t)
tj = {"kernel_source_sha16": bench.kernel_source_hash(),
      "note": "HBM bytes per launch from rocprofv3 PMC (separate --pmc passes for FETCH_SIZE and WRITE_SIZE, units KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section: gfx950 tallies 128-B requests at 64 B). That rule is calibrated for coalesced streams; tools/micro_fetch.hip (profiles/r03/micro_fetch.txt) confirms it for 4 / 8 / 16 B per lane and shows that scattered gathers read anywhere between 0.5 and 2.3 of the bytes they use, so for the GLS kernels (4-16 B gathers) the doubled figure is the guide's prescription and *_bounds = [FETCH as read + WRITE, 2 FETCH + WRITE] brackets it. pmc_summary.csv of the round holds the raw counters.",
      "gls_n216_bounds": [traffic_lo(["nin_gls_"]), traffic(["nin_gls_"])],
      "idw_n216_bounds": [traffic_lo(["nin_rows_kernel<0>"]), traffic(["nin_rows_kernel<0>"])],
      "gls_n216_bytes_per_launch": traffic(["nin_gls_"]),
      "idw_n216_bytes_per_launch": traffic(["nin_rows_kernel<0>"]),   # kernels_idw_ls.hip: METHOD 0 = IDW, 1 = LS
      "ls_n216_bytes_per_launch": traffic(["nin_rows_kernel<1>"])}
tj["raw_KiB"] = {f"{k}|{c}": round(v, 1) for (k, c), v in sorted(avg.items()) if c in ("FETCH_SIZE", "WRITE_SIZE")}
json.dump(tj, open(out + "/traffic.json", "w"), indent=1)
print(open(out + "/kernel_stats.csv").read()[:1500])
print(json.dumps(tj, indent=1)[:1500])
PY
