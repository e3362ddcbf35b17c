#!/usr/bin/env python3
"""ALGORITHMIC FP64 flops per node of the multifrontal GLS formulation -- the useful arithmetic,
no role masks, no redundant panel factorisations, no padding -- next to the reference-equivalent dense dgels count of
SURVEY 8(d).  Counted on the block structure the kernels factor (tools/proto_hex8_mf.py, tools/proto_mfw.py):

  faces      per internal face: two K.N products (2 x 15), T (3), U = N x T (9), |U| (6), pow (1), tau.U (3)        = 52
  phase 1    per front (a cell with 3 faces whose 3 neighbours are dense cells): Householder QR of its 10 x 3 panel,
             applied to the front's 9 neighbour columns + c: step k works on 10 - k rows: 2 m for the norm, 4 m per
             trailing column (dot + update); then z = R^-T d (9), u = z^T R_ed (54), s = z.b (6)
  phase 2    the (7 F + D + 3 free) x (3 D + 1) dense problem (7 fill rows per front, one cell row per dense cell, three
             rows per free face): Householder steps on the 3 D columns, 2 m + 4 m per trailing column each
  tail       back substitution (n^2), residuals on the cell rows, r.r over the rows left, the divisions

A step's scalar chain (sqrt, reciprocal, sign) is O(1) and counted as 8.  fma = 2 flops.

    python tools/count_algorithmic_flops.py
"""


def householder(m, n_pivot, n_cols):
    """flops of n_pivot Householder steps on an m-row block with n_cols columns in all (pivot columns included)."""
    f = 0
    for k in range(n_pivot):
        rows = m - k
        f += 2 * rows + 8 + 4 * rows * (n_cols - k - 1)
    return f


def multifrontal(F, D, faces, free_faces=0):
    face = 52 * faces
    p1 = F * (householder(10, 3, 3 + 9 + 1) + 9 + 54 + 6)
    m2, n2 = 7 * F + D + 3 * free_faces, 3 * D
    p2 = householder(m2, n2, n2 + 1)
    tail = n2 * n2 + F * (2 * 9 + 2) + D * (2 * 3 + 1) + 2 * (m2 - n2) + (F + D) + 1
    return {"faces": face, "phase1": p1, "phase2": p2, "tail": tail, "total": face + p1 + p2 + tail}


def dense_reference(m, n, nrhs):
    """dgels on the dense m x n system with nrhs right-hand sides."""
    return 2 * m * n * n - 2 * n ** 3 / 3 + nrhs * (4 * m * n - 2 * n * n) + nrhs * n * n


CASES = {
    # name: (fronts, dense cells, internal faces, free faces)   reference: (m, n, nrhs)
    "cube node (hexahedra: 4 + 4 cells, 12 faces)": ((4, 4, 12, 0), (44, 25, 8)),
    "Kuhn node (tetrahedra: 12 + 12 cells, 36 faces)": ((12, 12, 36, 0), (132, 73, 24)),
    "wedge node (6 + 6 cells, 18 faces)": ((6, 6, 18, 0), (66, 37, 12)),
}


def main():
    print(f"{'node kind':52s} {'faces':>7s} {'phase 1':>8s} {'phase 2':>8s} {'tail':>6s} {'ALGORITHMIC':>12s} {'dense dgels':>12s}")
    for name, (mf, ref) in CASES.items():
        c = multifrontal(*mf)
        print(f"{name:52s} {c['faces']:7d} {c['phase1']:8d} {c['phase2']:8d} {c['tail']:6d} {c['total']:12d} {dense_reference(*ref):12.0f}")


if __name__ == "__main__":
    main()
