#!/bin/bash
# Issue-side counters of the one-wavefront multifrontal kernels: bash tools/pmc_tet.sh [out dir] [mesh of tools/time_methods.py] [kernel name part]
# (default: the Kuhn-tet mesh and nin_gls_mfw; "del40 nin_gls_mfx" = the wide kernel on a Delaunay mesh)
# (two --pmc passes, kernel-trace only; SQ counters are per XCD-sampled: ratios matter, not absolutes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/pmc_tet}
MESH=${2:-tet40}
KERN=${3:-nin_gls_mfw}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAIT_ANY"; do
  i=$((i+1))
  NIN_METHODS=gls timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/time_methods.py $MESH > $OUT/p$i.txt 2> $OUT/p$i.err || { echo "pass $i failed"; tail -5 $OUT/p$i.err; exit 1; }
done
python3 - $OUT $KERN <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1:3]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kern in k: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()): print(f"{c:34s} {len(v):3d} dispatches, average {sum(v) / len(v):16.1f}")
PY
