"""N > 1 with the REAL kernels: two ranks share the one GPU of the box (at most 6 processes may), each computes its
node block with the HIP path through ShardedPlan -- the code bench.py --gpus N runs: device-resident triplets,
static columns / counts gathered once, rotating buffer sets -- and the all-gather goes over gloo through host tensors
(comm_on_host; on the 8-GPU node it is RCCL on the device tensors).  The gathered matrix must be bit-identical to the
single-process GPU result -- the property the RCCL path relies on.  "slab": each rank generates only its own
slab (load_shard), as the bench does; no rank holds the whole mesh."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import util  # noqa: F401
from ninpol_amd import mesh as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SLAB = (9, 8, 10)


def _make_mesh(kind):
    m = M.mixed_mesh(9, 5, 5, jitter=0.1, seed=3) if kind == "mixed" else M.hex_mesh(*SLAB, jitter=0.15, seed=1)
    M.attach_fields(m, "u", perm="ALH", neumann_plane=(2, 0.0), seed=5)
    return m


def _worker(rank, world, port, kind, out_dir, exchange="allgather"):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ninpol_amd.partition import ShardedInterpolator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = ShardedInterpolator(device=0, comm_on_host=True, grid_build=("host", "device")[rank % 2], exchange=exchange)
        if kind == "slab":
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
            W2, nws2 = S.interpolate("u", meth)   # second step: the other buffer set of the rotation
            assert np.array_equal(W.data, W2.data, equal_nan=True) and np.array_equal(nws, nws2, equal_nan=True)
            if rank == 0:
                np.savez(os.path.join(out_dir, f"{kind}_{meth}.npz"), indptr=W.indptr, indices=W.indices, data=W.data, nws=nws)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,exchange", [("hex", "allgather"), ("mixed", "allgather"), ("slab", "allgather"),
                                           # the matrix exchange as direct peer-to-peer writes through the C ABI (nin_exchange_*): each
                                           # rank's unpadded block straight into the other's gathered buffer, opened over HIP IPC
                                           ("mixed", "p2p"), ("slab", "p2p")])
def test_two_ranks_on_one_gpu_match_single_process(kind, exchange, tmp_path):
    import ninpol_amd
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), exchange), nprocs=world, join=True)
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=_make_mesh(kind))
    for meth in ("idw", "ls", "gls"):
        W, nws = I.interpolate("u", meth)
        z = np.load(os.path.join(str(tmp_path), f"{kind}_{meth}.npz"))
        assert np.array_equal(z["indptr"], W.indptr) and np.array_equal(z["indices"], W.indices), meth
        assert np.array_equal(z["data"], W.data, equal_nan=True), meth
        assert np.array_equal(z["nws"], nws, equal_nan=True), meth


def _rccl_worker(rank, port, out_dir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ninpol_amd.partition import ShardedInterpolator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        S = ShardedInterpolator(device=0, grid_build="device")     # the collectives run on the device tensors (RCCL)
        S.load_mesh(_make_mesh("hex"))
        sp = S.device_plan("u", "gls")
        for _ in range(5):                                         # asynchronous steps over both buffer sets
            b = sp.step()
        sp.drain_all()
        torch.cuda.synchronize()
        W, nws = S.interpolate("u", "gls")
        np.savez(os.path.join(out_dir, "rccl.npz"), indptr=W.indptr, indices=W.indices, data=W.data, nws=nws,
                 vals=torch.cat(sp.pieces(sp.vals[b], 0)).cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_rccl_backend_one_rank_group(tmp_path):
    """The exchange exactly as the 8-GPU run issues it -- backend nccl (= RCCL), device tensors, async_op, both buffer
    sets -- on the one GPU this box has (a one-rank group): the call pattern and stream ordering, not the scaling."""
    import ninpol_amd
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=_make_mesh("hex"))
    W, nws = I.interpolate("u", "gls")
    z = np.load(os.path.join(str(tmp_path), "rccl.npz"))
    assert np.array_equal(z["indptr"], W.indptr) and np.array_equal(z["indices"], W.indices)
    assert np.array_equal(z["data"], W.data, equal_nan=True) and np.array_equal(z["nws"], nws, equal_nan=True)


def _twovars_mesh():
    """u: Neumann plane z = 0, v: all-Dirichlet, w: Neumann plane x = 1 (flags and values are per variable)."""
    m = M.mixed_mesh(9, 5, 5, jitter=0.1, seed=3)
    M.attach_fields(m, "u", perm="ALH", neumann_plane=(2, 0.0), seed=5)
    P = len(m.points)
    on = np.abs(m.points[:, 0] - 1.0) < 1e-12
    fw, vw = np.zeros(P), np.zeros(P)
    fw[on] = 1.0
    vw[on] = np.random.default_rng(9).uniform(0.0, 1.0, int(on.sum()))
    m.point_data.update({"neumann_flag_w": fw, "neumann_w": vw, "neumann_flag_v": np.zeros(P), "neumann_v": np.zeros(P)})
    m.cell_data["v"] = [np.sin(2.0 * np.asarray(a)) for a in m.cell_data["u"]]
    m.cell_data["w"] = [np.cos(3.0 * np.asarray(a)) for a in m.cell_data["u"]]
    return m


SEQ = ("u", "v", "u", "w", "v", "w")


def _edit_tables(I):
    """In place: v's flag row becomes w's, and every cell's permeability diagonal is scaled (diff_mag follows)."""
    pi = I.variable_to_index["points"]
    I.points_data[pi["neumann_flag_v"]][:] = I.points_data[pi["neumann_flag_w"]]
    ci = I.variable_to_index["cells"]
    E = I.grid.n_elems
    I.cells_data[ci["permeability"], :E * 9] *= 1.0 + 0.3 * np.tile(np.arange(9) % 4 == 0, E)
    I.cells_data[ci["diff_mag"], :E] = I.compute_diffusion_magnitude(I.cells_data[ci["permeability"], :E * 9].reshape(-1, 9))


def _alternate_worker(rank, world, port, out_dir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ninpol_amd.partition import ShardedInterpolator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = ShardedInterpolator(device=0, comm_on_host=True)
        mesh = _twovars_mesh()
        S.load_mesh(mesh)
        out = {}
        for i, var in enumerate(SEQ):                  # cached plans revisited; the device flags belong to the grid
            W, nws = S.interpolate(var, "gls")
            out[f"{i}_data"], out[f"{i}_nws"] = W.data, nws
        # bare plan steps, no refresh in between: the plan itself must notice that the grid holds another variable's flags
        pu, pw = S.device_plan("u", "gls"), S.device_plan("w", "gls")
        import torch
        for j, sp in enumerate((pu, pw, pu)):
            b = sp.step()
            sp.drain(b)
            torch.cuda.synchronize()
            out[f"bare{j}_vals"] = torch.cat(sp.pieces(sp.vals[b], 0)).cpu().numpy()
        u = np.concatenate(mesh.cell_data["u"])
        fields = np.stack([u, np.sin(3.0 * u)])
        for var in ("u", "w"):
            for meth in ("gls", "idw"):
                out[f"apply_{var}_{meth}"], out[f"apply_{var}_{meth}_nws"] = S.apply(var, meth, values=fields)
        _edit_tables(S.local)
        for var in ("v", "u"):
            W, nws = S.interpolate(var, "gls")
            out[f"edit_{var}_data"], out[f"edit_{var}_nws"] = W.data, nws
        if rank == 0:
            np.savez(os.path.join(out_dir, "alt.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_sharded_plans_follow_variable_switches_and_table_edits(tmp_path):
    """Round-2 advisor finding, on the real device state: ShardedInterpolator caches one plan per (variable, method) but
    the Neumann flags / permeability in HBM belong to the grid.  u, v, u, w, v, w in turn (three Neumann planes), bare
    plan.step() calls of two plans interleaved, the sharded apply, and in-place edits of a flag row and of the
    permeability: every result equals the single-process GPU result of a fresh Interpolator for that variable / table."""
    import ninpol_amd
    world = 2
    mp.spawn(_alternate_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "alt.npz"))
    mesh = _twovars_mesh()
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=mesh)
    ref = {v: I.interpolate(v, "gls") for v in ("u", "v", "w")}
    assert not np.array_equal(ref["u"][1], ref["w"][1], equal_nan=True) and not ref["v"][1].any()
    for i, var in enumerate(SEQ):
        assert np.array_equal(z[f"{i}_data"], ref[var][0].data, equal_nan=True), (i, var)
        assert np.array_equal(z[f"{i}_nws"], ref[var][1], equal_nan=True), (i, var)
    for j, var in enumerate(("u", "w", "u")):
        vals = z[f"bare{j}_vals"]
        assert np.array_equal(vals[vals != 0], ref[var][0].data, equal_nan=True), (j, var)
    u = np.concatenate(mesh.cell_data["u"])
    fields = np.stack([u, np.sin(3.0 * u)])
    for var in ("u", "w"):
        for meth in ("gls", "idw"):
            vals, nws = I.apply(var, meth, values=fields)
            assert np.array_equal(z[f"apply_{var}_{meth}"], vals, equal_nan=True), (var, meth)
            assert np.array_equal(z[f"apply_{var}_{meth}_nws"], nws, equal_nan=True), (var, meth)
    _edit_tables(I)
    for var in ("v", "u"):
        W, nws = I.interpolate(var, "gls")
        assert np.array_equal(z[f"edit_{var}_data"], W.data, equal_nan=True), var
        assert np.array_equal(z[f"edit_{var}_nws"], nws, equal_nan=True), var
    assert not np.array_equal(z["edit_u_data"], ref["u"][0].data, equal_nan=True)
