#!/usr/bin/env python3
"""Design aid (numpy, not product code): the multifrontal Householder QR that kernels_gls_hex8mf.hip implements,
written lane by lane for the 4 lanes of a node, checked against the oracle's GLS weights.

The 8 cells around an interior hexahedron node form the cube graph (cells = vertices, the 12 internal faces =
edges).  It is bipartite: 4 "even" cells E0..E3 that share no face, 4 "odd" cells O0..O3, E_l adjacent to every odd
cell but O_(3-l).  Ordering the unknowns [even cells | odd cells] makes the first 12 Householder steps four
INDEPENDENT 10 x (3 + 12) fronts (one per even cell: its cell row + the 9 rows of its 3 faces), one per lane; what
is left is a 32 x 12 dense problem whose rows already sit in the lanes that produced them (7 fill rows + one odd
cell row per lane), factored row-distributed with quad reductions.  ~6 k FMAs per node instead of ~20 k dense.

    python tools/proto_hex8_mf.py [edge]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def descriptor(cells, faces, face_cells):
    """Per node: lanes l = 0..3 -> (even cell local idx, odd cell local idx, 3 x (face local idx, neighbour local idx,
    even cell is side a)), neighbours in ascending odd-slot order, lane l not adjacent to odd slot 3 - l.
    Returns None when the cell graph is not the cube graph."""
    loc = {int(c): i for i, c in enumerate(cells)}
    adj = [[] for _ in range(8)]
    for fi, f in enumerate(faces):
        a, b = face_cells[f]
        if a not in loc or b not in loc:
            return None
        adj[loc[a]].append((loc[b], fi, True))    # (neighbour, face, this cell is side a)
        adj[loc[b]].append((loc[a], fi, False))
    if any(len(x) != 3 for x in adj):
        return None
    color = [-1] * 8
    color[0] = 0
    for _ in range(4):
        for i in range(8):
            if color[i] >= 0:
                for (j, _, _) in adj[i]:
                    if color[j] < 0:
                        color[j] = 1 - color[i]
                    elif color[j] == color[i]:
                        return None
    if sorted(color) != [0] * 4 + [1] * 4:
        return None
    even = [i for i in range(8) if color[i] == 0]
    odd_all = [i for i in range(8) if color[i] == 1]
    oslot = {}
    for l, e in enumerate(even):
        nb = {j for (j, _, _) in adj[e]}
        if len(nb) != 3:
            return None
        missing = [o for o in odd_all if o not in nb]
        if len(missing) != 1 or missing[0] in oslot:
            return None
        oslot[missing[0]] = 3 - l
    if sorted(oslot.values()) != [0, 1, 2, 3]:
        return None
    odd = [None] * 4
    for o, s in oslot.items():
        odd[s] = o
    lanes = []
    for l, e in enumerate(even):
        fs = sorted(adj[e], key=lambda t: oslot[t[0]])
        assert [oslot[t[0]] for t in fs] == [s for s in range(4) if s != 3 - l]
        lanes.append((e, odd[l], [(fi, j, sa) for (j, fi, sa) in fs]))
    return lanes


def house(alpha, ss):
    """beta, v_pivot, g for x = (alpha, rest), ss = |rest|^2 : H = I - g v v^T, v = (alpha - beta, rest)."""
    S = alpha * alpha + ss
    if ss == 0.0:
        return alpha, 0.0, 0.0
    sq = np.sqrt(S)
    beta = -np.copysign(sq, alpha)
    return beta, alpha - beta, 1.0 / (S + abs(alpha) * sq)


def node_weights(p, G, perm, dmag):
    eb, ee = G.esup_ptr[p], G.esup_ptr[p + 1]
    fb, fe = G.fsup_ptr[p], G.fsup_ptr[p + 1]
    cells, faces = G.esup[eb:ee], G.fsup[fb:fe]
    fc = {}
    for f in faces:
        a, b = G.esuf_ptr[f], G.esuf_ptr[f + 1]
        assert b - a == 2
        fc[f] = (int(G.esuf[a]), int(G.esuf[a + 1]))
    lanes = descriptor(cells, faces, fc)
    assert lanes is not None
    xv = G.point_coords[p]
    # ---- phase 1: per lane, front rows: 0 = even cell row, 1+3i+r = face i row r; cols: 0..2 own, 3+3t.. odd slot t, 15 = c
    R1, C = [], np.zeros((4, 8, 13))       # C[l]: rows 0..6 contribution, row 7 the odd cell row; cols 12 odd + c
    dsave = np.zeros((4, 2, 3))
    for l, (e, o, fl) in enumerate(lanes):
        F = np.zeros((10, 16))
        de = G.centroids[cells[e]] - xv
        F[0, 0:3] = de
        F[0, 15] = 1.0
        dsave[l, 0] = de
        dsave[l, 1] = G.centroids[cells[o]] - xv
        slots = [s for s in range(4) if s != 3 - l]
        for i, (fi, j, side_a) in enumerate(fl):
            f = faces[fi]
            N = G.normal_faces[f]
            T = xv - G.faces_centers[f]
            U = np.cross(N, T)
            eta = max(dmag[cells[e]], dmag[cells[j]], 0.0)
            tj = np.linalg.norm(U) ** (-eta)
            Ke, Kj = perm[cells[e]].reshape(3, 3), perm[cells[j]].reshape(3, 3)
            Bown = np.stack([Ke @ N, T, tj * U])
            Bnb = np.stack([Kj @ N, T, tj * U])
            sg = -1.0 if side_a else 1.0       # [-B_a | +B_b]
            F[1 + 3 * i:4 + 3 * i, 0:3] = sg * Bown
            t = slots[i]
            F[1 + 3 * i:4 + 3 * i, 3 + 3 * t:6 + 3 * t] = -sg * Bnb
        rinv = np.zeros(3)
        for k in range(3):
            beta, vp, g = house(F[k, k], float(F[k + 1:, k] @ F[k + 1:, k]))
            v = F[k:, k].copy()
            v[0] = vp
            w = g * (v @ F[k:, k + 1:])
            F[k:, k + 1:] -= np.outer(v, w)
            F[k, k] = beta
            F[k + 1:, k] = 0.0
            rinv[k] = 1.0 / beta
        R1.append((F[0:3].copy(), rinv))
        C[l, 0:7, :] = F[3:10, 3:16]
        C[l, 7, 3 * l:3 * l + 3] = dsave[l, 1]
        C[l, 7, 12] = 1.0
    # ---- phase 2: 32 x 12 (+ c), rows distributed: lane l holds C[l] (8 rows); step k pivots row q = k // 4 of lane k % 4
    rinv2 = np.zeros(12)
    for k in range(12):
        lam, q = k % 4, k // 4
        m = np.array([1.0 if l >= lam else 0.0 for l in range(4)])          # row q still active in lane l?
        X = C[:, q:, k].copy()
        X[:, 0] *= m
        alpha = C[lam, q, k]
        S = float((X * X).sum())                                          # includes alpha^2
        beta, vp, g = house(alpha, S - alpha * alpha)
        d = np.einsum("lr,lrj->j", X, C[:, q:, k + 1:])                   # partial dots with x, quad-reduced ...
        d -= beta * C[lam, q, k + 1:]                                     # ... pivot lane's correction: v = x - beta e_p
        w = g * d
        V = X.copy()
        V[lam, 0] = vp
        C[:, q:, k + 1:] -= V[:, :, None] * w[None, None, :]
        C[lam, q, k] = beta
        rinv2[k] = 1.0 / beta
    # ---- back substitution: odd unknowns (row k lives in lane k % 4, local row k // 4), then each lane's even cell
    y = np.zeros(12)
    for k in range(11, -1, -1):
        row = C[k % 4, k // 4]
        y[k] = (row[12] - row[k + 1:12] @ y[k + 1:]) * rinv2[k]
    wts = np.zeros(8)
    rr = float((C[:, 3:, 12] ** 2).sum())
    for l, (e, o, fl) in enumerate(lanes):
        Rr, rinv = R1[l]
        ye = np.zeros(3)
        for k in (2, 1, 0):
            ye[k] = (Rr[k, 15] - Rr[k, 3:15] @ y - Rr[k, k + 1:3] @ ye[k + 1:]) * rinv[k]
        wts[e] = (1.0 - dsave[l, 0] @ ye) / rr
        wts[o] = (1.0 - dsave[l, 1] @ y[3 * l:3 * l + 3]) / rr
    return wts


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    import ninpol_oracle as O
    from ninpol_amd import mesh as M
    m = M.hex_mesh(n, jitter=0.15, seed=0)
    M.attach_fields(m, "u", perm="ALH")
    o = O.OracleInterpolator("port", threads=8)
    o.load_mesh(m)
    W, _ = o.prepare("gls", "u")
    G = o.grid
    v2i = o.variable_to_index
    perm = o.cells_data[v2i["cells"]["permeability"]][:G.n_elems * 9].reshape(-1, 9)
    dmag = o.cells_data[v2i["cells"]["diff_mag"]][:G.n_elems]
    worst = 0.0
    cnt = 0
    for p in range(G.n_points):
        if G.boundary_points[p]:
            continue
        w = node_weights(p, G, perm, dmag)
        ref = W[p, :8]
        err = np.abs(w - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        cnt += 1
    print(f"{cnt} interior nodes, worst row-relative error vs oracle: {worst:.3e}")


if __name__ == "__main__":
    main()
