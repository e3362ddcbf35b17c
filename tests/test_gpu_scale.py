"""Parity at the BASELINE.json sizes (configs[1..3]) through size-independent properties, plus oracle
spot checks on random node samples.  -m gpu.

Properties used (they hold for the reference's formulation, so they are parity checks, not just sanity):
  * every non-empty row of W sums to 1: IDW by construction (idw.pyx:82-84), LS because
    sum_i (1 + lambda.d_i) = n + lambda.I (ls.pyx:126-135), GLS because the node value is reproduced for
    constant fields (the column of ones, gls.pyx:280);
  * linear exactness: for LS (any K) and for GLS with a CONSTANT permeability tensor, W . u(centroids) =
    u(nodes) for u = a + b.x on interior nodes (the reference publishes 1e-16 errors for its LIN case,
    accuracy.yaml LIN block; with a heterogeneous K a linear field violates GLS's flux-continuity rows, so
    it is not reproduced -- measured 3e-5 with the ALH tensor);
  * Dirichlet boundary rows are empty (idw.pyx:62-63).
"""
import numpy as np
import pytest

import util
from ninpol_amd import mesh as M

pytestmark = pytest.mark.gpu


def _interior(mesh):
    P = mesh.points
    lo, hi = P.min(axis=0), P.max(axis=0)
    return np.where(~np.any((np.abs(P - lo) < 1e-12) | (np.abs(P - hi) < 1e-12), axis=1))[0]


def _check_properties(I, mesh, method, linear_exact, tol_sum=1e-11, tol_lin=1e-10):
    W, nws = I.interpolate("u", method)
    P, E = I.grid.n_points, I.grid.n_elems
    assert W.shape == (P, E)
    inner = _interior(mesh)
    counts = np.diff(W.indptr)
    boundary = np.setdiff1d(np.arange(P), inner)
    assert counts[boundary].max() == 0                      # all-Dirichlet boundary: empty rows
    assert counts[inner].min() >= 1
    sums = np.asarray(W.sum(axis=1)).ravel()
    assert np.abs(sums[inner] - 1.0).max() <= tol_sum, (method, np.abs(sums[inner] - 1.0).max())
    assert np.all(np.isfinite(W.data))
    if linear_exact:
        cen = M.cell_centroids(mesh)
        coef = np.array([0.3, -1.1, 0.7])
        u_c, u_p = 2.0 + cen @ coef, 2.0 + mesh.points @ coef
        err = np.abs(W.dot(u_c)[inner] - u_p[inner]).max()
        assert err <= tol_lin, (method, err)
    return W


def test_1m_hex_idw_ls_gls():
    """configs[1] and [2]: IDW and GLS on the 1M-cell structured hexahedron mesh (100^3, jittered)."""
    import ninpol_amd
    mesh = M.hex_mesh(100, jitter=0.15, seed=0)
    M.attach_fields(mesh, "u", perm="LIN")     # the constant anisotropic tensor of the reference's LIN / QUAD cases
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=mesh)
    assert I.grid.n_elems == 1_000_000 and I.grid.n_points == 1_030_301 and I.grid.n_faces == 3_030_000
    W = _check_properties(I, mesh, "idw", linear_exact=False)
    assert W.nnz == 8 * 99 ** 3 and W.data.min() > 0.0
    _check_properties(I, mesh, "ls", linear_exact=True)
    _check_properties(I, mesh, "gls", linear_exact=True)


def test_mixed_mesh_sample_against_oracle(oracle_lib):
    """configs[3] in kind (irregular node degree, 8 .. ~30 cells per node): hex | pyramid+tet | tet mesh,
    every method against the oracle on a random sample of nodes."""
    import ninpol_amd
    mesh = M.mixed_mesh(40, 24, 24, jitter=0.1, seed=4)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=3)
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=mesh)
    o = oracle_lib.OracleInterpolator("port", threads=16)
    o.load_mesh(mesh)
    for k in util.GRID_ARRAYS:
        np.testing.assert_array_equal(getattr(I.grid, k), getattr(o.grid, k), err_msg=k)
    rng = np.random.default_rng(0)
    sample = np.sort(rng.choice(I.grid.n_points, 4000, replace=False)).astype(np.int64)
    full = np.arange(I.grid.n_points)
    for meth in ("idw", "ls", "gls"):
        w, nw = I.prepare_interpolator(meth, "u", full)
        wo, no = o.prepare(meth, "u", sample)
        tol = util.WEIGHT_RTOL if meth == "gls" else 1e-14
        assert util.rowscaled_err(w[sample], wo[sample]) <= tol, meth
        assert util.rowscaled_err(nw[sample], no[sample]) <= tol, meth


def test_10m_hex_gls_properties(oracle_lib):
    """The north-star workload itself: GLS on 216^3 = 10,077,696 hexahedra; size-independent properties."""
    import ninpol_amd
    mesh = M.hex_mesh(216, jitter=0.15, seed=0)
    M.attach_fields(mesh, "u", perm="ALH")
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=mesh)
    assert I.grid.n_elems == 10_077_696 and I.grid.n_points == 10_218_313
    W = _check_properties(I, mesh, "gls", linear_exact=False, tol_sum=1e-10)   # ALH tensor: heterogeneous K
    assert W.nnz == 8 * 215 ** 3
    worst, worst_ew = _oracle_on_node_runs(oracle_lib, mesh, W, runs=8, length=512, seed=11)
    print(f"216^3 GLS vs the oracle on 8 x 512 nodes: row-scaled {worst:.2e}, element-wise {worst_ew:.2e}")
    assert worst <= util.WEIGHT_RTOL and worst_ew <= util.ELEMENTWISE_RTOL_GLS, (worst, worst_ew)


def _oracle_on_node_runs(oracle_lib, mesh, W, runs, length, seed):
    """The oracle itself at size: `runs` runs of `length` consecutive nodes, each with the cells around it cut out of the big mesh
    (partition.extract_submesh: every cell and face of a sampled node is present, so its row is the row of the whole mesh) and
    mapped back to global column ids.  Returns the worst row-scaled and element-wise errors."""
    from ninpol_amd.partition import extract_submesh
    rng = np.random.default_rng(seed)
    P = mesh.points.shape[0]
    worst = worst_ew = 0.0
    for lo in rng.choice(P - length, runs, replace=False):
        lo = int(lo)
        sub, pid, cid, owned = extract_submesh(mesh, lo, lo + length)
        o = oracle_lib.OracleInterpolator("port", threads=16)
        o.load_mesh(sub)
        Wo, _ = o.interpolate("u", "gls")
        Wo = Wo.tocsr()[owned]
        Wg = W[lo:lo + length]
        np.testing.assert_array_equal(np.diff(Wg.indptr), np.diff(Wo.indptr))
        np.testing.assert_array_equal(Wg.indices, cid[Wo.indices])
        worst = max(worst, util.csr_rowscaled_err(Wg, Wg.indptr, Wg.indices, Wo.data))
        worst_ew = max(worst_ew, util.csr_elementwise_err(Wg, Wg.indptr, Wg.indices, Wo.data))
    return worst, worst_ew


def test_2m_unstructured_tets_gls_at_size(oracle_lib):
    """The unstructured mesh of the bench row at size: the Delaunay tetrahedrisation of a jittered 54^3 body-centred cloud
    (~2 M cells, 324 k nodes, 14 .. 42 cells around an interior node).  The launch plan keeps >= 95 % of the interior nodes off
    the block kernel; size-independent properties; the oracle on 8 runs of 256 nodes cut out of the big mesh."""
    import ninpol_amd
    mesh = M.delaunay_tet_mesh(54, seed=0)
    M.attach_fields(mesh, "u", perm="ALH")
    I = ninpol_amd.Interpolator(grid_build="device")
    I.load_mesh(mesh_obj=mesh)
    assert I.grid.n_elems > 1_900_000
    W = _check_properties(I, mesh, "gls", linear_exact=False, tol_sum=1e-10)
    plan = I.grid.gls_plan()
    n_int = len(_interior(mesh))
    one_wave = plan["mfw_large"] + plan["mfw_small"] + plan["mfw_general"] + plan["mfx"] + plan["hex8"]
    assert one_wave >= 0.95 * n_int, (plan, n_int)
    worst, worst_ew = _oracle_on_node_runs(oracle_lib, mesh, W, runs=8, length=256, seed=5)
    print(f"Delaunay 54^3 GLS vs the oracle on 8 x 256 nodes: row-scaled {worst:.2e}, element-wise {worst_ew:.2e}; plan {plan}")
    assert worst <= util.WEIGHT_RTOL and worst_ew <= util.ELEMENTWISE_RTOL_GLS, (worst, worst_ew)


def test_10m_mixed_gls_at_size(oracle_lib):
    """configs[3] AT SIZE: GLS on the 10,094,400-cell hex | pyramid | tet mesh (200 x 120 x 120 lattice cells, node
    degree 8 .. 26).  Closed-form counts, the size-independent properties, and the oracle itself on 4,096 nodes: eight
    runs of 512 consecutive nodes (x-fastest numbering: each run crosses the hexahedron, transition and tetrahedron
    regions), each with the cells around it cut out of the big mesh (partition.extract_submesh: every cell and face of
    a sampled node is present, so its row is the row of the whole mesh) and mapped back to global column ids."""
    import ninpol_amd
    from ninpol_amd.partition import extract_submesh
    mesh = M.mixed_mesh(200, 120, 120, jitter=0.1, seed=4)
    M.attach_fields(mesh, "u", perm="ALH")
    I = ninpol_amd.Interpolator(grid_build="device")
    I.load_mesh(mesh_obj=mesh)
    assert I.grid.n_elems == 10_094_400 and I.grid.n_points == 201 * 121 * 121 + 120 * 120   # + the pyramid apexes
    W = _check_properties(I, mesh, "gls", linear_exact=False, tol_sum=1e-10)
    ne = np.diff(np.asarray(I.grid.esup_ptr))
    assert ne.max() == 26 and set(np.unique(ne[_interior(mesh)])) >= {8, 16, 24, 26}
    rng = np.random.default_rng(7)
    P = I.grid.n_points
    worst, n_checked = 0.0, 0
    for lo in rng.choice(P - 512, 8, replace=False):
        lo = int(lo)
        sub, pid, cid, owned = extract_submesh(mesh, lo, lo + 512)
        o = oracle_lib.OracleInterpolator("port", threads=16)
        o.load_mesh(sub)
        Wo, _ = o.interpolate("u", "gls")
        Wo = Wo.tocsr()[owned]
        Wg = W[lo:lo + 512]
        np.testing.assert_array_equal(np.diff(Wg.indptr), np.diff(Wo.indptr))
        np.testing.assert_array_equal(Wg.indices, cid[Wo.indices])
        worst = max(worst, util.csr_rowscaled_err(Wg, Wg.indptr, Wg.indices, Wo.data))
        n_checked += 512
    assert n_checked == 4096 and worst <= util.WEIGHT_RTOL, worst


def test_80m_hex_gls_single_gpu_properties(oracle_lib):
    """BASELINE config [4]'s mesh -- 432^3 = 80,621,568 hexahedra, 81,182,737 nodes -- whole on ONE MI355X
    (36 GB of its 288 GB): device grid build, GLS, and the size-independent properties WITHOUT bringing 640 M entries to
    the host: row sums through the device-side apply on a constant field (1 on every computed row, exactly 0 on the
    empty Dirichlet rows), nnz and the row pattern through the device-side count / scan alone
    (nin_csr_compact_host with no entry buffers): nnz == 8 . 431^3, 8 entries on every interior node, none on the
    boundary; plus the oracle itself on two runs of 256 consecutive nodes cut out of the big mesh."""
    import ctypes
    import torch
    import ninpol_amd
    from ninpol_amd import _lib
    from ninpol_amd.partition import extract_submesh
    n = 432
    mesh = M.hex_mesh(n, jitter=0.15, seed=0)
    M.attach_fields(mesh, "u", perm="ALH")
    I = ninpol_amd.Interpolator(grid_build="device")
    I.load_mesh(mesh_obj=mesh)
    P, E = I.grid.n_points, I.grid.n_elems
    assert E == 80_621_568 and P == 81_182_737 and I.grid.n_faces == 3 * n * n * (n + 1)
    plan_counts = None
    # -- row sums: W . 1 on the device
    ones = np.ones(E)
    sums, nws = I.apply("u", "gls", values=ones)
    plan_counts = I.grid.gls_plan()
    assert plan_counts["hex8"] == (n - 1) ** 3, plan_counts          # every interior node is a cube node
    k = np.arange(P)
    i, j, l = k % (n + 1), (k // (n + 1)) % (n + 1), k // ((n + 1) * (n + 1))
    interior = (i > 0) & (i < n) & (j > 0) & (j < n) & (l > 0) & (l < n)
    del i, j, l, k
    assert np.all(sums[~interior] == 0.0) and not nws.any()
    assert np.abs(sums[interior] - 1.0).max() <= 1e-10
    del sums, ones
    # -- the pattern, counted on the device: weights into a device buffer, count + scan, only indptr comes back
    dp = I.device_plan("u", "gls")
    out = torch.empty(dp.nnz, dtype=torch.float64, device="cuda")
    dn = torch.empty(P, dtype=torch.float64, device="cuda")
    dp.launch(out.data_ptr(), dn.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    indptr = np.empty(P + 1, dtype=np.int32)
    nnz = ctypes.c_int64(0)
    _lib.check(_lib.load().nin_csr_compact_host(I.grid._h, ctypes.c_void_p(out.data_ptr()), indptr.ctypes.data_as(ctypes.c_void_p),
                                                None, None, ctypes.byref(nnz), None))
    assert nnz.value == 8 * (n - 1) ** 3 == 640_503_928
    cnt = np.diff(indptr)
    assert np.all(cnt[interior] == 8) and not cnt[~interior].any()
    # -- the oracle on two runs of nodes cut out of the big mesh (rows of the whole mesh, see test_10m_mixed_gls_at_size)
    esup_ptr = np.asarray(I.grid.esup_ptr)
    for lo in (40_000_000, 81_000_000):
        sub, pid, cid, owned = extract_submesh(mesh, lo, lo + 256)
        o = oracle_lib.OracleInterpolator("port", threads=8)
        o.load_mesh(sub)
        wo, _ = o.prepare("gls", "u")
        b, e = int(esup_ptr[lo]), int(esup_ptr[lo + 256])
        got = out[b:e].cpu().numpy()
        ref = np.concatenate([wo[p, :esup_ptr[lo + q + 1] - esup_ptr[lo + q]] for q, p in enumerate(owned)])
        rows = np.repeat(np.arange(256), np.diff(esup_ptr[lo:lo + 257]))
        scale = np.zeros(256)
        np.maximum.at(scale, rows, np.abs(ref))
        scale[scale == 0] = 1.0
        assert (np.abs(got - ref) / scale[rows]).max() <= util.WEIGHT_RTOL
