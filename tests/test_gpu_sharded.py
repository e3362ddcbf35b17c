"""N > 1 with the REAL kernels: two ranks share the one GPU of the box (at most 6 processes may), each computes its
node block with the HIP path, the all-gather runs over gloo through host tensors.  The gathered matrix must be
bit-identical to the single-process GPU result -- the property the RCCL path relies on."""
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


def _make_mesh(kind):
    m = M.mixed_mesh(9, 5, 5, jitter=0.1, seed=3) if kind == "mixed" else M.hex_mesh(9, 8, 10, jitter=0.15, seed=1)
    M.attach_fields(m, "u", perm="ALH", neumann_plane=(2, 0.0), seed=5)
    return m


def _worker(rank, world, port, kind, out_dir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ninpol_amd.partition import ShardedInterpolator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = ShardedInterpolator(device=0, comm_on_host=True, grid_build=("host", "device")[rank % 2])
        S.load_mesh(_make_mesh(kind))
        for meth in ("idw", "ls", "gls"):
            W, nws = S.interpolate("u", meth)
            if rank == 0:
                np.savez(os.path.join(out_dir, f"{kind}_{meth}.npz"), indptr=W.indptr, indices=W.indices, data=W.data, nws=nws)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["hex", "mixed"])
def test_two_ranks_on_one_gpu_match_single_process(kind, tmp_path):
    import ninpol_amd
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=_make_mesh(kind))
    for meth in ("idw", "ls", "gls"):
        W, nws = I.interpolate("u", meth)
        z = np.load(os.path.join(str(tmp_path), f"{kind}_{meth}.npz"))
        assert np.array_equal(z["indptr"], W.indptr) and np.array_equal(z["indices"], W.indices), meth
        assert np.array_equal(z["data"], W.data, equal_nan=True), meth
        assert np.array_equal(z["nws"], nws, equal_nan=True), meth
