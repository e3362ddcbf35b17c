#!/usr/bin/env python3
"""Generate tests/golden/pins/fan_exact.npz: on the FAN-tensor case (tests/utils/analytical.py:285-293 of the reference;
hex_mesh(n, jitter=0.15, seed=0), Neumann plane z = 0, n = 32 and 64) a sample of computed nodes with
  * `ref`   -- the weight rows the REFERENCE ITSELF returns (oracle/_ref: gls.pyx + SciPy's dgels), and
  * `exact` -- the same least-squares problem (the float64 matrix M_v of gls.pyx:252-416) solved in 80-bit extended
               precision (numpy longdouble Householder QR below; eps = 1.1e-19, cond(M_v) <= 1e6), rounded to float64.
The case is ill-conditioned (cond(M_v) = 3e5 .. 6e5): the reference itself is only accurate to ~1e-10 there, so element-wise
agreement of two correct codes is bounded by the sum of their distances to `exact`.  The GPU test holds the HIP path to
"no further from exact than the reference is" (tests/test_gpu_parity.py::test_gpu_gls_fan_exact_sample).
Dev container only (needs oracle/_ref); the .npz is data."""
import os
import sys

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.dirname(HERE)):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

import ninpol_oracle as O  # noqa: E402
from ninpol_amd import mesh as M  # noqa: E402

LD = np.longdouble


def system(p, G, perm, dmag, flag):
    """M_v as the reference assembles it (gls.pyx:252-416), float64 arithmetic in the reference's operation order;
    returns (M, n_elem) -- the weights are row n-1 of pinv(M) restricted to the cell rows."""
    cells = G.esup[G.esup_ptr[p]:G.esup_ptr[p + 1]]
    faces = G.fsup[G.fsup_ptr[p]:G.fsup_ptr[p + 1]]
    loc = {int(c): i for i, c in enumerate(cells)}
    ne, nf = len(cells), len(faces)
    bfaces = [f for f in faces if G.boundary_faces[f] == 1]
    m = ne + 3 * nf + len(bfaces)
    Mi = np.zeros((m, 3 * ne + 1))
    xv = G.point_coords[p]
    for e, c in enumerate(cells):
        Mi[e, 3 * e:3 * e + 3] = G.centroids[c] - xv
        Mi[e, -1] = 1.0
    row = ne
    for f in faces:
        a, b = G.esuf_ptr[f], G.esuf_ptr[f + 1]
        if b - a < 2:
            continue
        ca, cb = int(G.esuf[a]), int(G.esuf[a + 1])
        N = G.normal_faces[f]
        T = xv - G.faces_centers[f]
        U = np.array([N[1] * T[2] - N[2] * T[1], N[2] * T[0] - N[0] * T[2], N[0] * T[1] - N[1] * T[0]])
        eta = max(0.0, dmag[ca], dmag[cb])
        tau = np.sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]) ** (-eta)
        Ka, Kb = perm[ca].reshape(3, 3), perm[cb].reshape(3, 3)
        kna = np.array([Ka[r, 0] * N[0] + Ka[r, 1] * N[1] + Ka[r, 2] * N[2] for r in range(3)])
        knb = np.array([Kb[r, 0] * N[0] + Kb[r, 1] * N[1] + Kb[r, 2] * N[2] for r in range(3)])
        for k, (va, vb) in enumerate(((kna, knb), (T, T), (tau * U, tau * U))):
            Mi[row + k, 3 * loc[ca]:3 * loc[ca] + 3] = -va
            Mi[row + k, 3 * loc[cb]:3 * loc[cb] + 3] = vb
        row += 3
    if flag[p]:
        start = ne + 3 * nf
        for i, f in enumerate(bfaces):
            c0 = int(G.esuf[G.esuf_ptr[f]])
            K0 = perm[c0].reshape(3, 3)
            N = G.normal_faces[f]
            Mi[start + i, 3 * loc[c0]:3 * loc[c0] + 3] = [-(K0[r, 0] * N[0] + K0[r, 1] * N[1] + K0[r, 2] * N[2]) for r in range(3)]
    return Mi, ne


def last_row_of_pinv(Mi, ne):
    """Row n-1 of the least-squares solution of M X = [I_ne; 0] in extended precision (Householder QR, full column rank)."""
    A = Mi.astype(LD)
    m, n = A.shape
    B = np.zeros((m, ne), dtype=LD)
    B[:ne, :ne] = np.eye(ne, dtype=LD)
    for k in range(n):
        x = A[k:, k].copy()
        nx = np.sqrt((x * x).sum())
        if nx == 0:
            raise ZeroDivisionError("rank deficient")
        beta = -np.copysign(nx, x[0])
        v = x.copy()
        v[0] -= beta
        g = LD(1) / (v * v).sum() * 2
        A[k:, k:] -= np.outer(v, g * (v @ A[k:, k:]))
        B[k:, :] -= np.outer(v, g * (v @ B[k:, :]))
    return np.asarray(B[n - 1, :] / A[n - 1, n - 1], dtype=np.float64)   # back-substitution's first step IS row n-1


def main():
    assert O.have_reference(), "build oracle/_ref first"
    out = {}
    for n, n_int, n_neu in ((32, 384, 64), (64, 768, 128)):
        m = M.hex_mesh(n, jitter=0.15, seed=0)
        M.attach_fields(m, "u", perm="FAN", neumann_plane=(2, 0.0), seed=7)
        ref = O.OracleInterpolator("reference")
        ref.load_mesh(m)
        port = O.OracleInterpolator("port", threads=8)
        port.load_mesh(m)
        wr, nr = ref.prepare("gls", "u")
        wp, npo = port.prepare("gls", "u")
        G = port.grid
        v2i = port.variable_to_index
        perm = port.cells_data[v2i["cells"]["permeability"]][:G.n_elems * 9].reshape(-1, 9)
        dmag = port.cells_data[v2i["cells"]["diff_mag"]][:G.n_elems]
        flag = m.point_data["neumann_flag_u"].astype(bool)
        bp = G.boundary_points.astype(bool)
        deg = np.diff(G.esup_ptr)
        rng = np.random.default_rng(n)
        # interior nodes: the worst-disagreeing ones (port vs reference) + a random draw; Neumann-plane nodes with > 1 cell
        d = np.abs(wp - wr).max(axis=1) / np.abs(wr).max(axis=1).clip(1e-300)
        interior = np.nonzero(~bp)[0]
        worst = interior[np.argsort(d[interior])[-n_int // 2:]]
        rand = rng.choice(np.setdiff1d(interior, worst), n_int - len(worst), replace=False)
        neu = np.nonzero(flag & (deg == 4))[0]           # face-interior nodes of the plane (edges / corners: degenerate classes)
        neu = rng.choice(neu, min(n_neu, len(neu)), replace=False)
        nodes = np.sort(np.concatenate([worst, rand, neu])).astype(np.int64)
        exact = np.zeros((len(nodes), 8))
        for i, p in enumerate(nodes):
            Mi, ne = system(int(p), G, perm, dmag, flag)
            exact[i, :ne] = last_row_of_pinv(Mi, ne)
        sc = np.abs(exact).max(axis=1, keepdims=True)
        e_ref = (np.abs(wr[nodes] - exact) / sc).max()
        e_port = (np.abs(wp[nodes] - exact) / sc).max()
        print(f"n={n}: {len(nodes)} nodes ({len(neu)} Neumann); reference vs exact {e_ref:.3e}, port vs exact {e_port:.3e}, "
              f"port vs reference {(np.abs(wp[nodes] - wr[nodes]) / sc).max():.3e}")
        out[f"nodes_{n}"] = nodes
        out[f"exact_{n}"] = exact
        out[f"ref_{n}"] = wr[nodes]
        out[f"ref_err_{n}"] = np.array(e_ref)
    np.savez_compressed(os.path.join(HERE, "pins", "fan_exact.npz"), **out)


if __name__ == "__main__":
    main()
