"""N > 1 path on CPU: world_size-2 gloo.  The partition (node blocks + replicated neighbour cells), the
local->global index maps, the slab loader and the exchange -- ShardedPlan: static columns / counts gathered
once, padded all-gather of values + Neumann array per step, two rotating buffer sets -- are the product's
(ninpol_amd/partition.py, the same code bench.py --gpus N runs); only the per-rank compute is the oracle
here, writing into host tensors, because the HIP kernels need a GPU -- this is a test, the product path
never routes through the oracle.  Expectation: the gathered matrix is bit-identical to the single-process
result (same rows, same order, same arithmetic)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import util  # noqa: F401  (path setup via conftest)
from ninpol_amd import mesh as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SLAB = (5, 4, 9)   # the slab case: hex_mesh(5, 4, 9), node planes dealt to the ranks


def _make_mesh(kind):
    if kind == "hexfree":     # all-Dirichlet boundary: neumann_ws is identically zero and is not gathered at all
        m = M.hex_mesh(5, 6, 6, jitter=0.15, seed=1)
        M.attach_fields(m, "u", perm="ALH", seed=5)
        return m
    if kind == "slab":
        m = M.hex_mesh(*SLAB, jitter=0.15, seed=1)
    elif kind == "mixed":
        m = M.mixed_mesh(8, 4, 4, jitter=0.1, seed=3)
    else:
        m = M.hex_mesh(6, 5, 7, jitter=0.15, seed=1)
    M.attach_fields(m, "u", perm="ALH", neumann_plane=(2, 0.0), seed=5)
    return m


class _OraclePlan:
    """What ShardedPlan needs of a device plan, on host tensors: CSR-position values with neumann_ws added to every
    entry of its row (interpolator.pyx:618), zero rows included."""

    def __init__(self, o, variable, method):
        self.o, self.variable, self.method = o, variable, method
        self.nnz = int(o.grid.esup_ptr[-1])
        self.n_points = int(o.grid.n_points)

    def launch_tensors(self, out, nws):
        import torch
        g = self.o.grid
        W, n = self.o.prepare(self.method, self.variable)
        counts = np.diff(g.esup_ptr)
        mask = np.arange(W.shape[1])[None, :] < counts[:, None]
        out[:self.nnz] = torch.from_numpy((W + n[:, None])[mask])
        nws[:self.n_points] = torch.from_numpy(n)


class _OracleCompute:
    def __init__(self):
        import ninpol_oracle
        self.o = ninpol_oracle.OracleInterpolator("port", threads=1)

    def load_mesh(self, filename="", mesh_obj=None):
        self.o.load_mesh(mesh_obj)
        self.grid = self.o.grid
        self.variable_to_index, self.points_data = self.o.variable_to_index, self.o.points_data   # (the Neumann flags)

    def device_plan(self, variable, method):
        return _OraclePlan(self.o, variable, method)


def _worker(rank, world, port, kind, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from ninpol_amd.partition import ShardedInterpolator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = ShardedInterpolator(device=None, make_interpolator=_OracleCompute)
        if kind == "slab":   # every rank generates only its own slab: no rank holds the whole mesh
            from ninpol_amd.partition import node_block
            plane_lo, plane_hi = node_block(SLAB[2] + 1, rank, world)
            sub, node_off, cell_off, own_lo, own_hi = M.hex_slab(*SLAB, plane_lo, plane_hi, jitter=0.15, seed=1)
            M.attach_fields(sub, "u", perm="ALH", neumann_plane=(2, 0.0), seed=5)
            S.load_shard(sub, node_off, cell_off, (own_lo, own_hi), (SLAB[0] + 1) * (SLAB[1] + 1) * (SLAB[2] + 1),
                         SLAB[0] * SLAB[1] * SLAB[2])
        else:
            S.load_mesh(_make_mesh(kind))
        for meth in ("idw", "ls", "gls"):
            W, nws = S.interpolate("u", meth)
            assert S.device_plan("u", meth).gather_neumann == (meth == "gls" and kind != "hexfree")
            W2, nws2 = S.interpolate("u", meth)    # second step: the other buffer set of the rotation
            assert np.array_equal(W.data, W2.data) and np.array_equal(nws, nws2, equal_nan=True)
            np.savez(os.path.join(out_dir, f"r{rank}_{meth}.npz"), indptr=W.indptr, indices=W.indices,
                     data=W.data, nws=nws)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["hex", "mixed", "slab", "hexfree"])
def test_two_rank_gather_matches_single(tmp_path, oracle_lib, kind):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, kind, str(tmp_path)), nprocs=world, join=True)
    mesh = _make_mesh(kind)
    o = oracle_lib.OracleInterpolator("port", threads=1)
    o.load_mesh(mesh)
    for meth in ("idw", "ls", "gls"):
        W, nws = o.interpolate("u", meth)
        for r in range(world):
            z = np.load(os.path.join(str(tmp_path), f"r{r}_{meth}.npz"))
            np.testing.assert_array_equal(z["indptr"], W.indptr)
            np.testing.assert_array_equal(z["indices"], W.indices)
            np.testing.assert_array_equal(z["data"], W.data)
            np.testing.assert_array_equal(z["nws"], nws)


def test_extract_submesh_invariants():
    from ninpol_amd.partition import extract_submesh, node_block
    mesh = _make_mesh("mixed")
    P = mesh.points.shape[0]
    seen = np.zeros(P, dtype=int)
    for r in range(3):
        lo, hi = node_block(P, r, 3)
        sub, pid, cid, owned = extract_submesh(mesh, lo, hi)
        assert np.all(np.diff(pid) > 0) and np.all(np.diff(cid) > 0)
        assert np.array_equal(pid[owned], np.arange(lo, hi))
        seen[lo:hi] += 1
        # every global cell touching an owned node is present
        goff = 0
        for b in mesh.cells:
            touch = ((b.data >= lo) & (b.data < hi)).any(axis=1)
            assert np.isin(goff + np.nonzero(touch)[0], cid).all()
            goff += len(b.data)
        np.testing.assert_array_equal(sub.points, mesh.points[pid])
    assert np.all(seen == 1)
