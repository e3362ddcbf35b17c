#!/usr/bin/env python3
"""Count the FP64 operations a kernel executes per pass, from its gfx950 ISA (straight-line kernels only: every
instruction of the node loop runs once per pass).  Used for bench.py's EXEC_FLOPS_PER_NODE_HEX.

    python tools/count_fp64.py [source.hip] [kernel-name-substring] [lanes per node]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ninpol_amd", "csrc", "kernels_gls_hex8mf.hip")
kern = sys.argv[2] if len(sys.argv) > 2 else "nin_gls_hex8w2_kernelILb0"
lanes_per_node = int(sys.argv[3]) if len(sys.argv) > 3 else 4
asm = "/tmp/_count_fp64.s"
sys.path.insert(0, ROOT)
from ninpol_amd.build import UNITS
extra = next((x for f, _, x in UNITS if f == os.path.basename(src)), [])
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + extra +
                      ["-I", os.path.dirname(src), src, "-o", asm], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kern + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ops = collections.Counter()
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    ops[t.split()[0]] += 1
total = sum(ops.values())
fma = sum(v for k, v in ops.items() if k.startswith(("v_fma_f64", "v_fmac_f64")))
oth = sum(v for k, v in ops.items() if k.startswith(("v_mul_f64", "v_add_f64")))
spc = sum(v for k, v in ops.items() if k.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_", "v_ldexp_f64", "v_frexp", "v_rndne_f64")))
print(f"{kern}: {total} instructions in the kernel body (one pass of the node loop + prologue)")
print(f"  FP64 fma {fma}, mul/add {oth}, special {spc}; per lane-pass {2 * fma + oth} flop; per node ({lanes_per_node} lanes) "
      f"{lanes_per_node * (2 * fma + oth)} flop")
for k, v in ops.most_common(25):
    print(f"  {v:6d} {k}")
