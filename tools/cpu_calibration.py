#!/usr/bin/env python3
"""Container-only: time the C restatement (oracle "port") against the compiled reference (oracle/_ref) on the same
samples, and write the ratio to profiles/cpu_calibration.json.

The reference never travels to the GPU box, so bench.py times the PORT there and carries this ratio in
its `cpu_baseline` block: reference-equivalent Mnodes/s = port Mnodes/s * t_port_over_t_ref.

    OPENBLAS_NUM_THREADS=1 python tools/cpu_calibration.py [--edges 64 88] [--repeats 2]
"""
import os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")   # before numpy / scipy load OpenBLAS (gls.pyx:63-70 asks for it)
import argparse
import json
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, nargs="+", default=[64, 88])
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--methods", nargs="+", default=["gls", "idw", "ls"])
    args = ap.parse_args()
    import ninpol_oracle as O
    from ninpol_amd import mesh as M
    assert O.have_reference(), "oracle/_ref is not built: python oracle/build_ref.py (dev container only)"
    threads = min(16, os.cpu_count() or 1)
    rows = []
    for n in args.edges:
        m = M.hex_mesh(n, jitter=0.15, seed=0)
        M.attach_fields(m, "u", perm="ALH")
        for meth in args.methods:
            t = {}
            for kind in ("port", "reference"):
                o = O.OracleInterpolator(kind, threads=threads)
                o.load_mesh(m)
                best = 1e30
                for _ in range(args.repeats):
                    t0 = time.perf_counter()
                    o.prepare(meth, "u")
                    best = min(best, time.perf_counter() - t0)
                t[kind] = best
                P = o.grid.n_points
                del o
            rows.append({"method": meth, "edge": n, "nodes": int(P), "threads": threads,
                         "t_port_s": round(t["port"], 4), "t_ref_s": round(t["reference"], 4),
                         "t_port_over_t_ref": round(t["port"] / t["reference"], 4)})
            print(rows[-1], flush=True)
    out = {"cpu_model": cpu_model(), "logical_cpus": os.cpu_count(), "threads": threads,
           "note": "min over repeats of prepare() (method kernel only); reference = oracle/_ref (the reference's own "
                   ".pyx compiled here, OpenMP as gls.pyx:87 / idw.pyx:55 ask, capped by the container's CPUs), "
                   "port = oracle/ninpol_oracle.c with the same thread count; OPENBLAS_NUM_THREADS=1",
           "samples": rows}
    for meth in args.methods:
        r = [x["t_port_over_t_ref"] for x in rows if x["method"] == meth]
        out[f"{meth}_t_port_over_t_ref"] = round(sum(r) / len(r), 4)
    path = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
