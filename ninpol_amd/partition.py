"""Multi-GPU path: node-block sharding + one all-gather of the result (no other collective).

Every node's weight row depends only on its own esup / fsup rows and the cells / faces they reference
(the prange bodies idw.pyx:57-84, ls.pyx:56-135, gls.pyx:161-219 touch only row `point`), so the path
shards with no exchange during compute:

  * rank r owns the contiguous node block [P r / W, P (r+1) / W);
  * it keeps every cell that touches an owned node (the neighbouring cells are replicated), in the
    original relative order, and renumbers points / cells locally.  All cells and faces around an
    owned node are present, local ids are monotone in global ids, so each owned node sees exactly the
    rows it sees in the whole mesh, in the same order: results are bit-identical to a single-GPU run;
  * after the kernels, ONE all-gather with per-rank counts (allgatherv) reassembles the rows:
    per owned node its entry count, per entry (global column int32, value float64).  Over RCCL the
    uneven sizes are handled by exchanging the counts first and padding to the largest shard.  Everything
    stays on the device (ShardedPlan): columns and counts are gathered once per mesh, a step gathers the
    values and the Neumann array, asynchronously, under the next step's kernel.

torch.distributed is plumbing only (process group + the collective); backend "nccl" is RCCL over xGMI
on the GPU box and "gloo" in the CPU tests.
"""
import numpy as np
import scipy.sparse as sp

from . import topology as T
from .mesh import CellBlock, Mesh


def node_block(n_points, rank, world):
    """[lo, hi) of the contiguous node-index block of `rank`."""
    return (n_points * rank) // world, (n_points * (rank + 1)) // world


def extract_submesh(mesh, lo, hi):
    """Cells touching a node in [lo, hi) + the points they use, relative order preserved.

    Returns (submesh, point_ids, cell_ids, owned): point_ids / cell_ids map local -> global ids
    (ascending), `owned` are the LOCAL ids of the nodes in [lo, hi) (ascending)."""
    dim = T.mesh_dimension([b.type for b in mesh.cells])
    keep_blocks, cell_ids, used = [], [], []
    goff = 0
    for b in mesh.cells:
        if b.type not in T.TYPES_PER_DIMENSION[dim]:
            keep_blocks.append(None)
            continue
        d = np.asarray(b.data)
        m = ((d >= lo) & (d < hi)).any(axis=1)
        keep_blocks.append(m)
        cell_ids.append(goff + np.nonzero(m)[0])
        used.append(np.unique(d[m]))
        goff += len(d)
    point_ids = np.unique(np.concatenate(used)) if used else np.zeros(0, dtype=np.int64)
    cells, cell_data = [], {k: [] for k in mesh.cell_data}
    for bi, (b, m) in enumerate(zip(mesh.cells, keep_blocks)):
        if m is None or not m.any():
            continue
        cells.append(CellBlock(b.type, np.searchsorted(point_ids, np.asarray(b.data)[m])))
        for k in mesh.cell_data:
            cell_data[k].append(np.asarray(mesh.cell_data[k][bi])[m])
    point_data = {k: np.asarray(v)[point_ids] for k, v in mesh.point_data.items()}
    sub = Mesh(np.asarray(mesh.points)[point_ids], cells, point_data, cell_data)
    owned = np.nonzero((point_ids >= lo) & (point_ids < hi))[0]
    return sub, point_ids, np.concatenate(cell_ids) if cell_ids else np.zeros(0, dtype=np.int64), owned


def allgatherv(tensors, counts_hint=None, group=None):
    """All-gather 1-D tensors whose length differs per rank.  `tensors` is a list of same-length-role
    tensors of this rank (e.g. [values f64, columns i32]) sharing one length; returns, per input, the
    list of every rank's piece.  One count exchange + one padded all_gather per tensor."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = tensors[0].device
    n = torch.tensor([t.numel() for t in tensors], dtype=torch.int64, device=dev)
    all_n = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(all_n, n, group=group)
    all_n = torch.stack(all_n).cpu().numpy()          # (world, len(tensors))
    out = []
    for i, t in enumerate(tensors):
        mx = int(all_n[:, i].max())
        pad = torch.zeros(mx, dtype=t.dtype, device=dev)
        pad[:t.numel()] = t
        buf = torch.empty(world * mx, dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(buf, pad, group=group)
        out.append([buf[r * mx:r * mx + int(all_n[r, i])] for r in range(world)])
    return out


class P2PExchange:
    """The all-gather as direct peer-to-peer writes (libninpol_amd's nin_exchange_*, csrc/exchange.hip): every rank owns a gathered
    buffer of `world` slots; a push copies this rank's block straight into slot `rank` of every peer's buffer, one copy per peer on
    its own stream -- seven xGMI links in use at once where a ring all-gather uses one -- and nothing but the block itself travels (no
    padding to the longest shard).  The process group is used for the 64-byte handle exchange (once) and for the barrier between
    "all pushes complete" and "read": control only, no payload."""

    def __init__(self, device, rank, world, slot_bytes, group=None):
        import ctypes
        import torch
        import torch.distributed as dist
        from . import _lib
        self._lib, self._L = _lib, _lib.load()
        self.rank, self.world, self.group, self.device = rank, world, group, device
        h = ctypes.c_void_p()
        _lib.check(self._L.nin_exchange_create(int(device), int(rank), int(world), int(slot_bytes), ctypes.byref(h)))
        self._h = h
        mine = ctypes.create_string_buffer(64)
        _lib.check(self._L.nin_exchange_handle(self._h, mine))
        handles = [None] * world
        dist.all_gather_object(handles, mine.raw, group=group)
        allh = ctypes.create_string_buffer(b"".join(handles), 64 * world)
        _lib.check(self._L.nin_exchange_connect(self._h, allh))
        self.slot_bytes = int(self._L.nin_exchange_slot_bytes(self._h))
        self._ptr = int(self._L.nin_exchange_buffer(self._h))
        self._torch = torch

    def push(self, tensor, offset_bytes=0, stream=None):
        """tensor: this rank's block (contiguous, on this rank's device) -> slot `rank` of every rank's buffer."""
        import ctypes
        t = self._torch
        st = t.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        self._lib.check(self._L.nin_exchange_push(self._h, ctypes.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size(),
                                                  int(offset_bytes), ctypes.c_void_p(st)))

    def complete(self):
        """Every rank's pushes have landed in every rank's buffer when this returns (collective)."""
        import torch.distributed as dist
        self._lib.check(self._L.nin_exchange_wait_sent(self._h, None, 1))
        dist.barrier(group=self.group)

    def slot(self, r, dtype, count, offset_bytes=0):
        """Rank r's block in THIS rank's gathered buffer as a tensor view (no copy)."""
        t = self._torch
        es = t.empty(0, dtype=dtype).element_size()
        return _device_view(t, self._ptr + r * self.slot_bytes + offset_bytes, count, dtype, self.device, es)

    def close(self):
        if self._h:
            self._L.nin_exchange_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _device_view(torch, ptr, count, dtype, device, elem_size):
    """A torch tensor over `count` elements of device memory the library owns (via __cuda_array_interface__)."""
    typestr = {torch.float64: "<f8", torch.int32: "<i4", torch.int64: "<i8"}[dtype]

    class _Mem:
        __cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    if count == 0:
        return torch.empty(0, dtype=dtype, device=torch.device("cuda", device))
    return torch.as_tensor(_Mem(), device=torch.device("cuda", device))


class ShardedPlan:
    """Device-resident `interpolate(variable, method)` / `apply(...)` over the process group.

    Two exchanges, both ONE all-gather per step and nothing else:

      * `step()` -- the matrix: this rank's kernel writes its node block's CSR values, the all-gather reassembles the
        (count, column, value) triplets + the Neumann array on every rank.  Columns and counts do not depend on the step
        (they are the esup rows of the owned nodes, global cell ids): they are gathered once, on first use; a step moves
        8 B per ENTRY (+ 8 B per row for the Neumann array, only when it can be non-zero at all).
      * `apply_step(u)` -- what the reference's callers do with the matrix (`weights.dot(u)`,
        tests/utils/analytical.py:236): this rank applies its own row block on the device (nin_apply_device) and the
        all-gather moves 8 B per NODE and field -- 82 MB in total at 80 M cells against 4.5 GB per GPU for the matrix.

    Shards are padded to the longest (RCCL has no allgatherv); rank r's piece of a gathered tensor `t` is
    t[r * mx : r * mx + lens[r]].  Two output / gather buffer sets rotate: `step()` returns at once, the gather of step
    i runs (on the collective's stream) under the kernel of step i + 1, and the buffers of a step stay valid until the
    step after next starts.

    The Neumann flags / permeability on the device belong to the local grid, not to this plan: every step checks that
    they are still this plan's variable (DevicePlan.ensure_current), `refresh()` re-reads the caller's tables and
    re-derives `gather_neumann` (a collective: call it on every rank) -- ShardedInterpolator.interpolate() / apply() do
    that on every call, as the single-GPU interpolate() does; a bare step() loop (bench.py) pays neither."""

    def __init__(self, sharded, variable, method):
        import torch
        import torch.distributed as dist
        S = self.S = sharded
        self.variable, self.method = variable, method
        self.empty = S.local is None                  # this rank owns no node (more ranks than node planes, say)
        self.plan = None if self.empty else S.local.device_plan(variable, method)
        lo, hi = S.own_lo, S.own_hi
        if self.empty:
            self.eb = self.ee = 0
            self.local_points = self.local_elems = self.local_nnz = 0
        else:
            g = S.local.grid
            esup_ptr = np.asarray(g.esup_ptr)
            self.eb, self.ee = int(esup_ptr[lo]), int(esup_ptr[hi])
            self.local_points, self.local_elems, self.local_nnz = int(self.plan.n_points), int(g.n_elems), int(self.plan.nnz)
        self.n_owned = hi - lo
        dev, cdev = S.torch_device, S.comm_device
        self.dev = dev
        lens = torch.tensor([self.ee - self.eb, self.n_owned], dtype=torch.int64, device=cdev)
        all_lens = [torch.empty_like(lens) for _ in range(S.world)]
        dist.all_gather(all_lens, lens, group=S.group)
        self.lens = torch.stack(all_lens).cpu().numpy()          # (world, 2): entries, rows per rank
        self.mx_nnz, self.mx_rows = max(int(self.lens[:, 0].max()), 1), max(int(self.lens[:, 1].max()), 1)
        self.cols = self.counts = None        # static parts of the matrix exchange: gathered on the first step()
        self.out = self.vals = None
        n_nws = max(self.local_points, lo + self.mx_rows)        # the padded send window must stay inside the buffer
        self.nws = [torch.zeros(n_nws, dtype=torch.float64, device=dev) for _ in range(2)]
        self.neumann = None
        self.gather_neumann = None
        self._derive_gather_neumann()
        self.pending = [None, None]
        self.n_steps = 0
        self._apply = {}                      # n_fields -> buffers of apply_step
        # the matrix exchange as direct peer-to-peer writes (nin_exchange_*) instead of the all-gather: NIN_EXCHANGE=p2p, or
        # ShardedInterpolator(exchange="p2p").  The RCCL all-gather stays the default until a scaling curve exists.
        self.p2p = None
        self.exchange_mode = getattr(S, "exchange_mode", "allgather")

    # -- what depends on the caller's tables -------------------------------------------------------------------------
    def _derive_gather_neumann(self):
        """neumann_ws is identically zero unless the method is GLS and some node of the mesh carries the Neumann flag
        (idw.pyx / ls.pyx never write it; gls.pyx:470-472 only on flagged nodes): decided over all ranks, and then it
        is not gathered at all -- 11 % of a matrix step's bytes on an all-Dirichlet hexahedron mesh."""
        import torch
        import torch.distributed as dist
        S = self.S
        mine = 0 if self.empty else int(self.method == "gls" and S._any_neumann_flag(self.plan, self.variable))
        flagged = torch.tensor([mine], dtype=torch.int32, device=S.comm_device)
        dist.all_reduce(flagged, op=dist.ReduceOp.MAX, group=S.group)
        g = bool(int(flagged[0]))
        if g != self.gather_neumann:
            self.gather_neumann = g
            self.neumann = [(torch.empty if g else torch.zeros)(S.world * self.mx_rows, dtype=torch.float64, device=self.dev)
                            for _ in range(2)]

    def refresh(self):
        """Re-read the caller's field tables (flags, Neumann values, permeability, diff_mag) and re-derive what was
        derived from them.  Collective: every rank of the group calls it."""
        self.drain_all()
        if not self.empty:
            self.plan.refresh()
        self._derive_gather_neumann()

    # -- the matrix exchange --------------------------------------------------------------------------------------------
    def _matrix_buffers(self):
        import torch
        S, dev = self.S, self.dev
        cols_pad = torch.zeros(self.mx_nnz, dtype=torch.int32, device=dev)
        cnt_pad = torch.zeros(self.mx_rows, dtype=torch.int32, device=dev)
        if not self.empty:
            g = S.local.grid
            esup_ptr = np.asarray(g.esup_ptr)
            cols = S.global_cells(np.asarray(g.esup)[self.eb:self.ee]).astype(np.int32)
            cols_pad[:self.ee - self.eb] = torch.from_numpy(np.ascontiguousarray(cols)).to(dev)
            cnt_pad[:self.n_owned] = torch.from_numpy(np.diff(esup_ptr[S.own_lo:S.own_hi + 1]).astype(np.int32)).to(dev)
        self.cols = torch.empty(S.world * self.mx_nnz, dtype=torch.int32, device=dev)
        self.counts = torch.empty(S.world * self.mx_rows, dtype=torch.int32, device=dev)
        S._gather(self.cols, cols_pad)
        S._gather(self.counts, cnt_pad)
        n_out = max(self.local_nnz, self.eb + self.mx_nnz)
        self.out = [torch.zeros(n_out, dtype=torch.float64, device=dev) for _ in range(2)]
        self.vals = [torch.empty(S.world * self.mx_nnz, dtype=torch.float64, device=dev) for _ in range(2)]

    def drain(self, b):
        if self.pending[b] is not None:
            for w in self.pending[b]:
                w.wait()          # the current stream waits for the collective; the host does not block
            self.pending[b] = None

    def drain_all(self):
        self.drain(0)
        self.drain(1)

    def step(self, events=None, exchange=True):
        """One pass of the hot path over this rank's shard + the exchange; returns the buffer set it went to.
        events: optional (start, end) torch.cuda.Event pair recorded around the kernel launches (bench.py).
        exchange=False: the kernels alone (bench.py's compute-only leg)."""
        S = self.S
        if self.out is None:
            self._matrix_buffers()
        b = self.n_steps % 2
        self.n_steps += 1
        self.drain(b)             # the gather that last read this buffer set must have been delivered
        if events is not None:
            events[0].record()
        if not self.empty:
            S._launch(self.plan, self.out[b], self.nws[b])
        if events is not None:
            events[1].record()
        if exchange:
            self.exchange(b)
        return b

    def _p2p_setup(self):
        S = self.S
        slot = 8 * (self.mx_nnz + self.mx_rows)       # values, then the Neumann rows
        self.p2p = [P2PExchange(S.device, S.rank, S.world, slot, S.group) for _ in range(2)]   # one per buffer set
        self.vals = [None, None]
        self.neumann_p2p = [None, None]

    def exchange(self, b):
        """The all-gather of buffer set b alone (asynchronous; bench.py times it without the kernel too)."""
        S = self.S
        lo = S.own_lo
        if self.exchange_mode == "p2p":
            # each rank writes its UNPADDED value block (and Neumann rows) straight into every peer's gathered buffer
            if self.p2p is None:
                self._p2p_setup()
            x = self.p2p[b]
            x.push(self.out[b][self.eb:self.ee])
            if self.gather_neumann:
                x.push(self.nws[b][lo:lo + self.n_owned], offset_bytes=8 * self.mx_nnz)
            self.pending[b] = [_P2PWait(self, b)]
            return
        self.pending[b] = [S._gather(self.vals[b], self.out[b][self.eb:self.eb + self.mx_nnz], async_op=True)]
        if self.gather_neumann:
            self.pending[b].append(S._gather(self.neumann[b], self.nws[b][lo:lo + self.mx_rows], async_op=True))
        self.pending[b] = [w for w in self.pending[b] if w is not None] or None

    def pieces(self, t, which):
        """Per-rank pieces of a gathered tensor (which: 0 = per entry, 1 = per row)."""
        if isinstance(t, _P2PGathered):
            return t.pieces
        mx = self.mx_nnz if which == 0 else self.mx_rows
        return [t[r * mx:r * mx + int(self.lens[r, which])] for r in range(self.S.world)]

    # -- the apply exchange -----------------------------------------------------------------------------------------
    def _apply_buffers(self, k):
        import torch
        if k not in self._apply:
            dev, W = self.dev, self.S.world
            self._apply[k] = dict(
                node=[torch.zeros((k, max(self.local_points, 1)), dtype=torch.float64, device=dev) for _ in range(2)],
                send=[torch.zeros((k, self.mx_rows), dtype=torch.float64, device=dev) for _ in range(2)],
                recv=[torch.empty((W, k, self.mx_rows), dtype=torch.float64, device=dev) for _ in range(2)])
        return self._apply[k]

    def apply_step(self, u_local, events=None):
        """node values of this rank's block = W_block . u on the device, then ONE all-gather of 8 B per node and field.
        u_local: (k, n_local_cells) float64 tensor on this rank's device (values of the shard's cells, halo included).
        Returns the buffer set; `apply_result(k, b)` assembles (k, n_points) from it once drained."""
        S = self.S
        k = int(u_local.shape[0])
        B = self._apply_buffers(k)
        b = self.n_steps % 2
        self.n_steps += 1
        self.drain(b)
        if events is not None:
            events[0].record()
        if not self.empty:
            S._launch_apply(self.plan, u_local, B["node"][b], self.nws[b])
            B["send"][b][:, :self.n_owned].copy_(B["node"][b][:, S.own_lo:S.own_hi])
        if events is not None:
            events[1].record()
        self.pending[b] = [S._gather(B["recv"][b].view(-1), B["send"][b].view(-1), async_op=True)]
        if self.gather_neumann:
            lo = S.own_lo
            self.pending[b].append(S._gather(self.neumann[b], self.nws[b][lo:lo + self.mx_rows], async_op=True))
        self.pending[b] = [w for w in self.pending[b] if w is not None] or None
        return b

    def apply_result(self, k, b):
        """(k, n_points) node values on the device from buffer set b (drained first)."""
        import torch
        self.drain(b)
        recv = self._apply[k]["recv"][b]
        rows = self.lens[:, 1]
        if np.all(rows == self.mx_rows):
            return recv.permute(1, 0, 2).reshape(k, -1)
        return torch.cat([recv[r, :, :int(rows[r])] for r in range(self.S.world)], dim=1)


class _P2PGathered:
    """What `vals[b]` / `neumann[b]` are in p2p mode: the per-rank views into this rank's gathered buffer."""

    def __init__(self, pieces):
        self.pieces = pieces


class _P2PWait:
    """The handle ShardedPlan.drain() waits on in p2p mode: completes the pushes of buffer set b (a collective) and publishes
    the gathered views."""

    def __init__(self, plan, b):
        self.plan, self.b = plan, b

    def wait(self):
        import torch
        p, b = self.plan, self.b
        x = p.p2p[b]
        x.complete()
        W = p.S.world
        p.vals[b] = _P2PGathered([x.slot(r, torch.float64, int(p.lens[r, 0])) for r in range(W)])
        if p.gather_neumann:
            p.neumann[b] = _P2PGathered([x.slot(r, torch.float64, int(p.lens[r, 1]), offset_bytes=8 * p.mx_nnz) for r in range(W)])


class ShardedInterpolator:
    """`interpolate()` / `apply()` over a process group: rank r holds the node block [P r / W, P (r+1) / W) of the mesh
    plus the cells around it and computes those rows on its own GPU.  `interpolate()` delivers the whole
    (n_points x n_elems) matrix to every rank (the all-gather of north_star); `apply()` delivers W . u -- each rank
    applies its own row block and only node values travel.

    Two ways in: `load_mesh(mesh_obj)` -- every rank is handed the same whole mesh and cuts its block out of it -- and
    `load_shard(...)` -- every rank brings only its own shard (a slab generated or read per rank: no rank ever holds the
    whole mesh; this is what bench.py --gpus N uses).

    make_interpolator: factory for the per-rank compute object (default: ninpol_amd.Interpolator on this rank's GPU).
    The CPU tests inject an oracle-backed object here; the product path never does."""

    def __init__(self, group=None, device=None, make_interpolator=None, comm_on_host=False, grid_build="host",
                 num_threads=0, exchange=None):
        """device: this rank's GPU (None: the injected compute object works on host tensors -- CPU tests only).
        comm_on_host: stage the all-gather through host tensors (a gloo group next to GPU compute, e.g. several
        ranks rehearsing on one GPU); default: the collective runs on the compute device (backend nccl = RCCL)."""
        import torch
        import torch.distributed as dist
        self.comm_on_host = bool(comm_on_host)
        # how the matrix travels: "allgather" (one padded all-gather over the process group: RCCL on the GPU box) or "p2p"
        # (libninpol_amd's nin_exchange_*: direct writes into the peers' buffers; needs a real device)
        import os as _os
        self.exchange_mode = exchange or _os.environ.get("NIN_EXCHANGE", "allgather")
        if self.exchange_mode not in ("allgather", "p2p"):
            raise ValueError("exchange must be 'allgather' or 'p2p'")
        if self.exchange_mode == "p2p" and device is None:
            raise ValueError("exchange='p2p' writes device memory: it needs a GPU (device=...)")
        self.grid_build = grid_build
        self.num_threads = num_threads
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.torch_device = torch.device("cpu") if device is None else torch.device("cuda", device)
        self.comm_device = torch.device("cpu") if (device is None or self.comm_on_host) else self.torch_device
        self._make = make_interpolator
        self.local = None
        self._plans = {}

    # ---- loading ------------------------------------------------------------------------------------------------
    def load_mesh(self, mesh_obj, bounds=None):
        """bounds: optional world + 1 ascending node offsets (bounds[r] .. bounds[r + 1] = rank r's block) instead of the
        equal split -- e.g. blocks balanced by cost on a mesh whose nodes differ (a Kuhn-tetrahedra node costs ~20 cube
        nodes); a rank may own nothing."""
        n_points = int(np.asarray(mesh_obj.points).shape[0])
        if bounds is None:
            lo, hi = node_block(n_points, self.rank, self.world)
        else:
            bounds = [int(b) for b in bounds]
            if len(bounds) != self.world + 1 or bounds[0] != 0 or bounds[-1] != n_points or any(
                    a > b for a, b in zip(bounds, bounds[1:])):
                raise ValueError(f"bounds must be {self.world + 1} ascending node offsets from 0 to {n_points}")
            lo, hi = bounds[self.rank], bounds[self.rank + 1]
        dim = T.mesh_dimension([b.type for b in mesh_obj.cells])
        n_elems = int(sum(len(b.data) for b in mesh_obj.cells if b.type in T.TYPES_PER_DIMENSION[dim]))
        if hi == lo:
            self.load_shard(None, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), (0, 0), n_points, n_elems)
            return
        sub, point_ids, cell_ids, owned = extract_submesh(mesh_obj, lo, hi)
        assert len(owned) == hi - lo and np.array_equal(owned, np.arange(owned[0], owned[0] + len(owned)))
        self.load_shard(sub, point_ids, cell_ids, (int(owned[0]), int(owned[0]) + len(owned)), n_points, n_elems)

    def load_shard(self, shard_mesh, point_ids, cell_ids, owned, n_points, n_elems):
        """shard_mesh: this rank's cells and the points they use, local numbering monotone in the global one (None: this
        rank owns nothing).  point_ids / cell_ids: local -> global ids, as arrays or -- for a contiguous slab -- as
        integer offsets.  owned: the LOCAL node range (lo, hi) this rank owns (globally: a contiguous block; blocks of
        the ranks tile [0, n_points) in rank order).  n_points / n_elems: of the whole mesh."""
        self.n_points, self.n_elems = int(n_points), int(n_elems)
        self.point_ids, self.cell_ids = point_ids, cell_ids
        self.own_lo, self.own_hi = int(owned[0]), int(owned[1])
        self.lo = int(self.global_points(np.array([self.own_lo]))[0]) if self.own_hi > self.own_lo else 0
        self.hi = self.lo + (self.own_hi - self.own_lo)
        self._plans = {}
        if shard_mesh is None or self.own_hi == self.own_lo:
            self.local = None
            self.own_lo = self.own_hi = 0
            return
        if self._make is None:
            from .interpolator import Interpolator
            self.local = Interpolator(device=self.device if self.device is not None else 0, grid_build=self.grid_build,
                                      num_threads=self.num_threads)
        else:
            self.local = self._make()
        self.local.load_mesh(mesh_obj=shard_mesh)

    def _any_neumann_flag(self, plan, variable):
        """Does any node of this rank's shard carry neumann_flag_<variable>?  (True when the compute object does not
        expose its point table: then the Neumann array is always gathered.)"""
        try:
            if hasattr(plan, "any_neumann_flag"):
                return plan.any_neumann_flag()
            row = self.local.variable_to_index["points"]["neumann_flag_" + variable]
            return bool(np.any(np.asarray(self.local.points_data)[row].astype(np.int64) != 0))
        except (AttributeError, KeyError, IndexError, TypeError):
            return True

    def global_cells(self, local_ids):
        return local_ids + self.cell_ids if np.isscalar(self.cell_ids) else np.asarray(self.cell_ids)[local_ids]

    def global_points(self, local_ids):
        return local_ids + self.point_ids if np.isscalar(self.point_ids) else np.asarray(self.point_ids)[local_ids]

    def local_cell_ids(self):
        """Global ids of this rank's cells (local order)."""
        if self.local is None:
            return np.zeros(0, dtype=np.int64)
        n = int(self.local.grid.n_elems)
        return np.arange(n, dtype=np.int64) + self.cell_ids if np.isscalar(self.cell_ids) else np.asarray(self.cell_ids)

    # ---- the exchange ---------------------------------------------------------------------------------------------
    def _gather(self, out_t, in_t, async_op=False):
        import torch
        import torch.distributed as dist
        if self.comm_device != self.torch_device:      # rehearsal: gloo through host copies, synchronous
            o = torch.empty(out_t.shape, dtype=out_t.dtype)
            dist.all_gather_into_tensor(o, in_t.cpu().contiguous(), group=self.group)
            out_t.copy_(o)
            return None
        return dist.all_gather_into_tensor(out_t, in_t, group=self.group, async_op=async_op) if async_op else \
            dist.all_gather_into_tensor(out_t, in_t, group=self.group)

    def _launch(self, plan, out_t, nws_t):
        if hasattr(plan, "launch_tensors"):            # injected compute object (CPU tests)
            plan.launch_tensors(out_t, nws_t)
            return
        import torch
        plan.launch(out_t.data_ptr(), nws_t.data_ptr(), torch.cuda.current_stream(self.torch_device).cuda_stream,
                    add_neumann=True)

    def _launch_apply(self, plan, u_t, node_t, nws_t):
        if hasattr(plan, "apply_tensors"):             # injected compute object (CPU tests)
            plan.apply_tensors(u_t, node_t, nws_t)
            return
        import torch
        assert u_t.is_contiguous() and node_t.is_contiguous() and u_t.shape[1] == plan.n_elems
        plan.launch_apply(u_t.data_ptr(), int(u_t.shape[0]), node_t.data_ptr(), nws_t.data_ptr(),
                          torch.cuda.current_stream(self.torch_device).cuda_stream)

    def device_plan(self, variable, method):
        """The (variable, method) plan of this mesh; building one is a collective (every rank calls it)."""
        key = (variable, method)
        if key not in self._plans:
            self._plans[key] = ShardedPlan(self, variable, method)
        return self._plans[key]

    def interpolate_device(self, variable, method):
        """One step, waited for.  Returns (plan, b): the gathered triplets are plan.counts / plan.cols (static) and
        plan.vals[b] / plan.neumann[b], padded per rank -- see ShardedPlan.pieces()."""
        import torch
        fresh = (variable, method) not in self._plans
        sp_ = self.device_plan(variable, method)
        if not fresh:
            sp_.refresh()         # the caller's tables as they are now (another variable / an edit may have intervened)
        b = sp_.step()
        sp_.drain(b)
        if self.torch_device.type == "cuda":
            torch.cuda.synchronize(self.torch_device)
        return sp_, b

    def interpolate(self, variable, method):
        """The reference's result tuple, assembled from the gathered triplets: (csr_matrix (n_points x n_elems),
        neumann_ws), zeros eliminated as interpolator.pyx:622-624 does."""
        import torch
        sp_, b = self.interpolate_device(variable, method)
        cat = lambda t, which: torch.cat(sp_.pieces(t, which)).cpu().numpy()
        data, indices = cat(sp_.vals[b], 0), cat(sp_.cols, 0)
        counts, neumann = cat(sp_.counts, 1).astype(np.int64), cat(sp_.neumann[b], 1)
        indptr = np.concatenate([[0], np.cumsum(counts)])
        idx_t = np.int32 if max(len(data), self.n_elems, self.n_points) < np.iinfo(np.int32).max else np.int64
        full = sp.csr_matrix((data, indices.astype(idx_t), indptr.astype(idx_t)), shape=(self.n_points, self.n_elems))
        full.eliminate_zeros()
        return full, neumann

    def apply(self, variable, method, values=None, local_values=False):
        """`W.dot(u)` of the reference's callers (tests/utils/analytical.py:236) over the group: every rank applies its
        own row block on its GPU and ONE all-gather of node values (8 B per node and field) delivers the result to all.
        values: one cell field (n_elems,) or k of them (k, n_elems) in GLOBAL cell numbering (default: the cell
        variable `variable` of this rank's shard); local_values=True: `values` holds only this rank's cells, shard order
        (halo cells included) -- for callers that never hold the whole mesh.  Returns (node_values, neumann_ws) like
        Interpolator.apply: (n_points,) or (k, n_points), Dirichlet rows 0."""
        import torch
        fresh = (variable, method) not in self._plans
        sp_ = self.device_plan(variable, method)
        if not fresh:
            sp_.refresh()
        one = False
        err = None                                   # a bad argument must fail on EVERY rank, before any collective (advisor, round 3)
        k_known = True
        if sp_.empty:
            # (an empty rank owns no nodes: it learns the number of fields from the others unless it was given values itself)
            k_known = values is not None
            k = 1 if values is None or np.ndim(values) == 1 else int(np.shape(values)[0])
            one = values is None or np.ndim(values) == 1
            u = np.zeros((k, 0))
        else:
            L = self.local
            if values is None:
                u = np.asarray(L.cells_data[L.variable_to_index["cells"][variable]])[:sp_.local_elems]
            else:
                u = np.asarray(values, dtype=np.float64)
                n_expected = sp_.local_elems if local_values else self.n_elems
                if u.ndim not in (1, 2) or u.shape[-1] != n_expected:
                    err = f"values must have shape ({n_expected},) or (k, {n_expected}), not {u.shape}."
                    u = np.zeros((1, sp_.local_elems))
                elif not local_values:
                    u = u[..., self.local_cell_ids()]
            one = u.ndim == 1
            u = np.ascontiguousarray(u.reshape(1, -1) if one else u, dtype=np.float64)
            k = u.shape[0]
        # ONE agreement step before anything else: did any rank reject its argument, and do the ranks that know the number of
        # fields agree on it (it sizes the gather)?  max over (bad, k, -k): k_min = -max(-k)
        import torch.distributed as dist
        big = 1 << 40
        kk = torch.tensor([1 if err else 0, k if k_known else 0, -k if k_known else -big], dtype=torch.int64, device=self.comm_device)
        dist.all_reduce(kk, op=dist.ReduceOp.MAX, group=self.group)
        any_bad, k_max, k_min = int(kk[0]), int(kk[1]), -int(kk[2])
        if any_bad:
            raise ValueError(err or "values were rejected on another rank of the group (every rank raises; no collective was entered).")
        if k_min != big and k_min != k_max:
            raise ValueError(f"the ranks disagree about the number of fields: between {k_min} and {k_max} (every rank raises).")
        k = k_max if k_max > 0 else k
        ut = torch.from_numpy(u).to(self.torch_device) if not sp_.empty else torch.zeros((k, 0), dtype=torch.float64,
                                                                                          device=self.torch_device)
        b = sp_.apply_step(ut)
        vals = sp_.apply_result(k, b).cpu().numpy()
        neumann = torch.cat(sp_.pieces(sp_.neumann[b], 1)).cpu().numpy()
        return (vals[0] if one else vals), neumann
