"""The exact torch.distributed call pattern of bench.py's N > 1 path on a ONE-rank RCCL group (what a one-GPU box can
check: argument forms, async handles, stream ordering against a kernel launched through the C ABI).
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/nccl_selftest.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
import ninpol_amd
from ninpol_amd import mesh as M

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
dist.init_process_group("nccl", device_id=dev)
world = dist.get_world_size()
m = M.hex_mesh(24, jitter=0.15); M.attach_fields(m, "u", perm="ALH")
I = ninpol_amd.Interpolator(device=local_rank); I.load_mesh(mesh_obj=m)
plan = I.device_plan("u", "gls")
stream = torch.cuda.current_stream()
lens = torch.tensor([plan.nnz, plan.n_points], dtype=torch.int64, device=dev)
all_lens = [torch.empty_like(lens) for _ in range(world)]
dist.all_gather(all_lens, lens)
mx = int(torch.stack(all_lens).cpu().numpy()[:, 0].max())
cols = torch.from_numpy(I.grid.esup.astype(np.int32)).to(dev)
cnt = torch.from_numpy(np.diff(I.grid.esup_ptr).astype(np.int32)).to(dev)
g_vals = torch.empty(world * mx, dtype=torch.float64, device=dev)
g_cols = torch.empty(world * mx, dtype=torch.int32, device=dev)
g_cnt = torch.empty(world * plan.n_points, dtype=torch.int32, device=dev)
outs = [torch.empty(plan.nnz, dtype=torch.float64, device=dev) for _ in range(2)]
nws = torch.empty(plan.n_points, dtype=torch.float64, device=dev)
pending = [None, None]
for i in range(6):
    b = i % 2
    if pending[b]:
        for w in pending[b]: w.wait()
    plan.launch(outs[b].data_ptr(), nws.data_ptr(), stream.cuda_stream, add_neumann=True)
    pending[b] = [dist.all_gather_into_tensor(g_vals, outs[b][:mx], async_op=True),
                  dist.all_gather_into_tensor(g_cols, cols, async_op=True),
                  dist.all_gather_into_tensor(g_cnt, cnt, async_op=True)]
for p in pending:
    for w in p: w.wait()
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
ok = bool(torch.equal(g_vals[:plan.nnz], outs[1]) and torch.equal(g_cols[:plan.nnz], cols) and torch.equal(g_cnt, cnt))
print("nccl selftest:", "OK" if ok else "MISMATCH", "world", world)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
