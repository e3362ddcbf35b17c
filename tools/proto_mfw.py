#!/usr/bin/env python3
"""Design aid (numpy, not product code): the arithmetic of kernels_gls_mfw.hip -- the one-wavefront multifrontal GLS
kernel for "two-coloured" nodes -- written the way the kernel computes it, checked against the oracle's GLS weights.

descriptor()    mfw_desc.hpp: two-colouring of the cell graph (cells = vertices, internal faces = edges), the colour
                class whose cells all have 3 faces at the node becomes the FRONTS (F <= 12), the others the DENSE cells
                (D <= 12); if the graph has odd cycles or no class qualifies, the general kind: the fronts are the
                largest greedy independent set of 3-face cells, the faces between two dense cells FREE faces whose rows
                join the dense problem as they stand (D <= 15); fronts and dense cells numbered in esup order.
phase 1         per front: the 10 x (3 own + 9 neighbour + c) front, three Householder steps on the own columns; the
                three rows of R folded into z = R_ee^-T d_e, u = z^T R_ed, s = z . b_e; 7 fill rows left.
dense problem   rows = the fronts' fill rows, then the dense cells' rows; columns = 3 per dense cell, then c.  Step t
                pivots on row t with the reflector v = (the pivot column's entries of the live rows, alpha - beta in the
                pivot row): w_j = g (v . C_j) -- one reduction per column -- and C_j -= w_j v for every row, the pivot
                row included (that makes it row t of R); retired rows are never touched again.
tail            R y = (Q^T c)(0:nc) by columns with the rows scaled as the kernel reads them, r_e = 1 - s + u . y for a
                front, r_o = 1 - d_o . y_o for a dense cell, weights r_i / (r . r).

    python tools/proto_mfw.py [tet|wedge|hex|mixed] [edge]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

MAX_FRONTS, MAX_DENSE, WIDE_DENSE, MAX_FREE, MAX_ROWS = 12, 12, 15, 14, 128


def descriptor(cells, faces, face_cells):
    """-> (kind, fronts, dense, front_faces, free_faces) or None.  kind 1 = two-coloured, 2 = general.  fronts / dense:
    positions in `cells` (esup order); front_faces[f] = 3 x (position of the face in `faces`, dense slot of the cell on
    the other side, front is the face's first cell); free_faces = (position of the face, dense slot of its first cell,
    dense slot of its second cell) for the faces between two dense cells."""
    ne, nf = len(cells), len(faces)
    if ne < 2 or ne > MAX_FRONTS + WIDE_DENSE or nf > 63 or 3 * nf < 2 * ne:
        return None
    loc = {int(c): i for i, c in enumerate(cells)}
    adj = [set() for _ in range(ne)]
    deg = [0] * ne
    ends = []
    for f in faces:
        a, b = face_cells[f]
        if b < 0 or a not in loc or b not in loc or a == b or loc[b] in adj[loc[a]]:
            return None                          # a boundary face, or two faces between the same pair of cells
        ia, ib = loc[a], loc[b]
        adj[ia].add(ib)
        adj[ib].add(ia)
        deg[ia] += 1
        deg[ib] += 1
        ends.append((ia, ib))
    colour = [-1] * ne
    colour[0] = 0
    frontier = [0]
    while frontier:
        nxt = []
        for i in frontier:
            for j in adj[i]:
                if colour[j] < 0:
                    colour[j] = 1 - colour[i]
                    nxt.append(j)
        frontier = nxt
    bipartite = min(colour) >= 0 and not any(colour[i] == colour[j] for i in range(ne) for j in adj[i])

    def qualifies(c):
        cls = [i for i in range(ne) if colour[i] == c]
        F, D = len(cls), ne - len(cls)
        return 1 <= F <= MAX_FRONTS and 1 <= D <= MAX_DENSE and nf == 3 * F and all(deg[i] == 3 for i in cls)

    kind, fronts = 0, None
    if bipartite:
        c = 0 if qualifies(0) else 1 if qualifies(1) else None
        if c is not None:
            kind, fronts = 1, [i for i in range(ne) if colour[i] == c]
    if fronts is None:                           # the largest greedy independent set of 3-face cells, one per starting cell
        best = []
        for start in range(ne):
            chosen = []
            for k in range(ne):
                c = (start + k) % ne
                if deg[c] == 3 and not (adj[c] & set(chosen)):
                    chosen.append(c)
            if len(chosen) > len(best):
                best = chosen
        F, D, nfree = len(best), ne - len(best), nf - 3 * len(best)
        if not (1 <= F <= MAX_FRONTS and 1 <= D <= WIDE_DENSE and 0 <= nfree <= MAX_FREE and 7 * MAX_FRONTS + D + 3 * nfree <= MAX_ROWS):
            return None
        kind, fronts = 2, sorted(best)
    dense = [i for i in range(ne) if i not in fronts]
    slot = {i: d for d, i in enumerate(dense)}
    rank = {i: f for f, i in enumerate(fronts)}
    front_faces = [[] for _ in fronts]
    free_faces = []
    for fi, (ia, ib) in enumerate(ends):
        if ia not in rank and ib not in rank:
            free_faces.append((fi, slot[ia], slot[ib]))
            continue
        a_front = ia in rank
        fc, oc = (ia, ib) if a_front else (ib, ia)
        front_faces[rank[fc]].append((fi, slot[oc], a_front))
    return kind, fronts, dense, front_faces, free_faces


def house(alpha, S):
    """beta, v_pivot, g for a column with pivot entry alpha and squared norm S (alpha^2 included)."""
    sq = np.sqrt(S)
    beta = -np.copysign(sq, alpha)
    return beta, alpha - beta, 1.0 / (S + abs(alpha) * sq)


BLOCKED = False      # the dense problem: one reflector at a time (kernel form "lane = row") or in panels of four (the MFMA form)


def dense_by_columns(C, nc):
    """step t pivots on row t; one reduction per column (rows_step of kernels_gls_mfw.hip); returns r . r"""
    live = np.ones(C.shape[0], dtype=bool)
    dk = float(C[:, 0] @ C[:, 0])                  # the first column's norm; the later ones ride with the reductions
    for t in range(nc):
        alpha = C[t, t]
        beta, vk, g = house(alpha, dk)
        v = np.where(live, C[:, t], 0.0)
        v[t] = vk
        w = g * (v @ C[:, t + 1:])
        C[:, t + 1:] -= np.outer(v, w)
        C[t, t] = beta
        live[t] = False
        if t + 1 < nc:
            dk = float(np.where(live, C[:, t + 1], 0.0) @ np.where(live, C[:, t + 1], 0.0))
    return float(np.where(live, C[:, nc], 0.0) @ np.where(live, C[:, nc], 0.0))


def dense_blocked(C, nc, nb=4):
    """The same Householder QR in panels of nb = 4 columns (compact WY), the way the strip form of the kernel computes it:
    panel p = columns 4 p .. 4 p + 3 (the last one may hold fewer pivot columns and then c, which sits at column nc),
    pivot rows 4 p + k.  In the panel: d_j = sum over the rows BELOW the pivot of a[r][k] a[r][j] (one reduction for all
    columns of the panel), the pivot row's entries separately, e_j = d_j + v_p a[p][j], w_j = -g e_j.  Then
    T (upper triangular, H_0 .. H_3 = I - V T V^T): T[k][k] = g_k, T[0:k, k] = -g_k T[0:k, 0:k] (V^T v_k);
    trailing columns: W = V^T C, W' = T^T W, C -= V W' -- the three products the matrix unit does.  Returns r . r."""
    m = C.shape[0]
    n_panels = (nc + nb - 1) // nb
    for p in range(n_panels):
        c0 = nb * p
        steps = min(nb, nc - c0)
        width = min(nb, nc + 1 - c0)               # columns of the panel block that exist (c may be one of them)
        P = C[:, c0:c0 + width]                    # view
        V = np.zeros((m, nb))
        gk = np.zeros(nb)
        for k in range(steps):
            rp = c0 + k
            below = np.arange(m) > rp
            xk = np.where(below, P[:, k], 0.0)
            d = xk @ P                             # d[j], j = 0 .. width - 1 (d[k] = |x|^2)
            ap = P[rp].copy()
            beta, vp, g = house(ap[k], ap[k] * ap[k] + d[k])
            e = d + vp * ap
            w = -g * e
            w[:k + 1] = 0.0
            vrow = xk.copy()
            vrow[rp] = vp
            P += np.outer(vrow, w)
            P[rp, k] = beta
            V[:, k] = vrow
            gk[k] = g
        G = V.T @ V
        T = np.zeros((nb, nb))
        for k in range(steps):
            T[k, k] = gk[k]
            if k:
                T[:k, k] = -gk[k] * (T[:k, :k] @ G[:k, k])
        if c0 + width < nc + 1:                    # trailing columns
            Ct = C[:, c0 + width:nc + 1]
            W = V.T @ Ct
            Ct -= V @ (T.T @ W)
        # (the kernel keeps v below the diagonal of the panel; R needs zeros there)
        for k in range(steps):
            C[c0 + k + 1:, c0 + k] = 0.0
    tail = C[nc:, nc]
    return float(tail @ tail)


def node_weights(p, G, perm, dmag):
    eb, ee = G.esup_ptr[p], G.esup_ptr[p + 1]
    fb, fe = G.fsup_ptr[p], G.fsup_ptr[p + 1]
    cells, faces = G.esup[eb:ee], G.fsup[fb:fe]
    fc = {}
    for f in faces:
        a, b = G.esuf_ptr[f], G.esuf_ptr[f + 1]
        fc[f] = (int(G.esuf[a]), int(G.esuf[a + 1]) if b - a == 2 else -1)
    desc = descriptor(cells, faces, fc)
    if desc is None:
        return None
    kind, fronts, dense, front_faces, free_faces = desc
    F, D = len(fronts), len(dense)
    nc = 3 * D
    xv = G.point_coords[p]
    # ---- phase 1: a front's rows 0 = cell row, 1 + 3 i + r = row r of face i; columns 0..2 own, 3 + 3 i .. neighbour i, 12 = c
    C = np.zeros((7 * F + D + 3 * len(free_faces), nc + 1))
    u = np.zeros((F, 9))
    s = np.zeros(F)
    d_front = np.zeros((F, 3))
    for f, e in enumerate(fronts):
        Fr = np.zeros((10, 13))
        de = G.centroids[cells[e]] - xv
        Fr[0, 0:3] = de
        Fr[0, 12] = 1.0
        d_front[f] = de
        for i, (fi, sl, a_front) in enumerate(front_faces[f]):
            face = faces[fi]
            j = dense[sl]
            N = G.normal_faces[face].astype(np.float64)
            T = xv - G.faces_centers[face]
            U = np.cross(N, T)
            eta = max(dmag[cells[e]], dmag[cells[j]], 0.0)
            tj = np.linalg.norm(U) ** (-eta)
            Ke, Kj = perm[cells[e]].reshape(3, 3), perm[cells[j]].reshape(3, 3)
            sg = -1.0 if a_front else 1.0          # row = [-B_a | +B_b]
            Fr[1 + 3 * i:4 + 3 * i, 0:3] = sg * np.stack([Ke @ N, T, tj * U])
            Fr[1 + 3 * i:4 + 3 * i, 3 + 3 * i:6 + 3 * i] = -sg * np.stack([Kj @ N, T, tj * U])
        rinv = np.zeros(3)
        for k in range(3):
            beta, vp, g = house(Fr[k, k], float(Fr[k:, k] @ Fr[k:, k]))
            v = Fr[k:, k].copy()
            v[0] = vp
            w = g * (v @ Fr[k:, k + 1:])
            Fr[k:, k + 1:] -= np.outer(v, w)
            Fr[k, k] = beta
            Fr[k + 1:, k] = 0.0
            rinv[k] = 1.0 / beta
        z = np.zeros(3)                            # z = R_ee^-T d_e
        z[0] = de[0] * rinv[0]
        z[1] = (de[1] - Fr[0, 1] * z[0]) * rinv[1]
        z[2] = (de[2] - Fr[0, 2] * z[0] - Fr[1, 2] * z[1]) * rinv[2]
        u[f] = z @ Fr[0:3, 3:12]
        s[f] = z @ Fr[0:3, 12]
        for i, (fi, sl, a_front) in enumerate(front_faces[f]):
            C[7 * f:7 * f + 7, 3 * sl:3 * sl + 3] = Fr[3:10, 3 + 3 * i:6 + 3 * i]
        C[7 * f:7 * f + 7, nc] = Fr[3:10, 12]
    d_dense = np.zeros((D, 3))
    for d, o in enumerate(dense):
        d_dense[d] = G.centroids[cells[o]] - xv
        C[7 * F + d, 3 * d:3 * d + 3] = d_dense[d]
        C[7 * F + d, nc] = 1.0
    for q, (fi, sa, sb) in enumerate(free_faces):   # a face between two dense cells: its rows [-B_a | +B_b] as they stand
        face = faces[fi]
        ja, jb = dense[sa], dense[sb]
        N = G.normal_faces[face].astype(np.float64)
        T = xv - G.faces_centers[face]
        U = np.cross(N, T)
        tj = np.linalg.norm(U) ** (-max(dmag[cells[ja]], dmag[cells[jb]], 0.0))
        Ka, Kb = perm[cells[ja]].reshape(3, 3), perm[cells[jb]].reshape(3, 3)
        r0 = 7 * F + D + 3 * q
        C[r0:r0 + 3, 3 * sa:3 * sa + 3] = -np.stack([Ka @ N, T, tj * U])
        C[r0:r0 + 3, 3 * sb:3 * sb + 3] = np.stack([Kb @ N, T, tj * U])
    if BLOCKED:
        rr = dense_blocked(C, nc)
    else:
        rr = dense_by_columns(C, nc)
    # ---- R y = (Q^T c)(0:nc): rows scaled by 1 / R(i, i), by columns
    R = C[:nc]
    ri = 1.0 / np.diag(R[:, :nc])
    ct = R[:, nc].copy()
    for k in range(nc - 1, -1, -1):
        yk = ct[k] * ri[k]
        ct[:k] -= yk * R[:k, k]
    y = ct * ri
    wts = np.zeros(len(cells))
    for f, e in enumerate(fronts):
        re = 1.0 - s[f]
        for i, (fi, sl, a_front) in enumerate(front_faces[f]):
            re += u[f, 3 * i:3 * i + 3] @ y[3 * sl:3 * sl + 3]
        wts[e] = re / rr
    for d, o in enumerate(dense):
        wts[o] = (1.0 - d_dense[d] @ y[3 * d:3 * d + 3]) / rr
    return wts


def main():
    global BLOCKED
    if "--blocked" in sys.argv:
        BLOCKED = True
        sys.argv.remove("--blocked")
    kind = sys.argv[1] if len(sys.argv) > 1 else "tet"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    import ninpol_oracle as O
    from ninpol_amd import mesh as M
    m = M.mixed_mesh(2 * n, n, n, jitter=0.1, seed=0) if kind == "mixed" else {"tet": M.tet_mesh, "wedge": M.wedge_mesh, "hex": M.hex_mesh}[kind](n, jitter=0.1, seed=0)
    M.attach_fields(m, "u", perm="ALH")
    o = O.OracleInterpolator("port", threads=8)
    o.load_mesh(m)
    W, _ = o.prepare("gls", "u")
    G = o.grid
    v2i = o.variable_to_index
    perm = o.cells_data[v2i["cells"]["permeability"]][:G.n_elems * 9].reshape(-1, 9)
    dmag = o.cells_data[v2i["cells"]["diff_mag"]][:G.n_elems]
    worst, cnt, skipped = 0.0, 0, 0
    for p in range(G.n_points):
        if G.boundary_points[p]:
            continue
        w = node_weights(p, G, perm, dmag)
        if w is None:
            skipped += 1
            continue
        ref = W[p, :len(w)]
        worst = max(worst, np.abs(w - ref).max() / np.abs(ref).max())
        cnt += 1
    print(f"{cnt} interior nodes the kernel takes ({skipped} others), worst row-relative error vs oracle: {worst:.3e}")


if __name__ == "__main__":
    main()
