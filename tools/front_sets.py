"""The fronts of unstructured nodes: on the interior nodes of a Delaunay mesh, (1) the independent set of 3-face cells
mfx_desc.hpp computes (minimum-degree greedy) against mfw_desc.hpp's first-fit greedy and the exact maximum (branch and bound), (2) the size
class of the dense problem each node falls into.  CPU only:   python tools/front_sets.py [n] [lattice]"""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from ninpol_amd import mesh as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lattice = sys.argv[2] if len(sys.argv) > 2 else "bcc"
m = M.delaunay_tet_mesh(n, seed=0, lattice=lattice)
pts, tets = m.points, m.cells[0].data
interior = np.all((pts > 1e-12) & (pts < 1 - 1e-12), axis=1)
cells_of = collections.defaultdict(list)
for e, t in enumerate(tets):
    for v in t:
        cells_of[int(v)].append(e)
face = collections.defaultdict(list)
for e, t in enumerate(tets):
    for k in range(4):
        face[tuple(sorted(np.delete(t, k).tolist()))].append(e)


def graph(p):
    cs = cells_of[p]
    idx = {c: i for i, c in enumerate(cs)}
    adj = [0] * len(cs)
    for c in cs:
        t = tets[c]
        for k in range(4):
            if t[k] == p:
                continue
            for o in face[tuple(sorted(np.delete(t, k).tolist()))]:
                if o != c:
                    adj[idx[c]] |= 1 << idx[o]
    return adj


def greedy(adj):
    ne, best = len(adj), 0
    for s in range(ne):
        ch = 0
        for k in range(ne):
            c = (s + k) % ne
            if bin(adj[c]).count("1") == 3 and not (adj[c] & ch):
                ch |= 1 << c
        if bin(ch).count("1") > bin(best).count("1"):
            best = ch
    return bin(best).count("1")


def mindeg(adj):
    """what mfx_desc.hpp computes: minimum-residual-degree greedy from every fourth starting cell, the best kept"""
    ne, best = len(adj), 0
    elig = sum(1 << c for c in range(ne) if bin(adj[c]).count("1") == 3)
    for s in range(0, ne, 4):
        avail, n = elig, 0
        while avail and n < 16:
            pick, pd = -1, 99
            for k in range(ne):
                c = (s + k) % ne
                if (avail >> c) & 1:
                    d = bin(adj[c] & avail).count("1")
                    if d < pd:
                        pd, pick = d, c
            avail &= ~(adj[pick] | (1 << pick))
            n += 1
        best = max(best, n)
    return best


def exact(adj):
    best = [0]

    def rec(avail, size):
        if size + bin(avail).count("1") <= best[0]:
            return
        if not avail:
            best[0] = size
            return
        v, md = -1, 99
        a = avail
        while a:
            i = (a & -a).bit_length() - 1
            a &= a - 1
            d = bin(adj[i] & avail).count("1")
            if d < md:
                md, v = d, i
            if d <= 1:
                break
        rec(avail & ~(adj[v] | (1 << v)), size + 1)
        if md >= 1:
            rec(avail & ~(1 << v), size)
    rec((1 << len(adj)) - 1, 0)
    return best[0]


def size_class(F, D, nfree):
    rows, nc = 7 * F + D + 3 * nfree, 3 * D
    for k, (tq, tcb) in enumerate(((6, 10), (7, 11), (8, 13), (9, 15), (10, 16))):
        if rows <= 16 * tq and nc < 4 * tcb:
            return f"{tq}x{tcb}"
    return "block"


by_ne, classes = collections.defaultdict(list), collections.Counter()
for p in np.nonzero(interior)[0][:3000]:
    adj = graph(int(p))
    ne, nf = len(adj), sum(bin(a).count("1") for a in adj) // 2
    g = mindeg(adj)
    by_ne[ne].append((greedy(adj), g, exact(adj)))
    F = min(g, 16)
    D, nfree = ne - F, nf - 3 * F
    classes[size_class(F, D, nfree) if D <= 21 and nfree <= 16 else "block"] += 1
print(f"Delaunay {lattice} cloud n = {n}: {sum(len(v) for v in by_ne.values())} interior nodes")
print("cells  nodes  first-fit greedy (mfw_desc)  min-degree greedy (mfx_desc)  exact maximum   (mean fronts)")
for ne in sorted(by_ne):
    a = np.array(by_ne[ne])
    print(f"{ne:5d} {len(a):6d} {a[:, 0].mean():16.2f} {a[:, 1].mean():28.2f} {a[:, 2].mean():22.2f}")
a = np.concatenate([np.array(v) for v in by_ne.values()])
print(f"  all {len(a):6d} {a[:, 0].mean():16.3f} {a[:, 1].mean():28.3f} {a[:, 2].mean():22.3f}")
tot = sum(classes.values())
print("size classes of the dense problem:", {k: f"{100 * v / tot:.1f} %" for k, v in sorted(classes.items())})
