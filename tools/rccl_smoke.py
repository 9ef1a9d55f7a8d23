"""RCCL on one device: the nccl process group of bench.py's multi-rank branch initialised with world_size 1, the two
collectives the branch issues (all_gather of the pose rows on device tensors, all_reduce MAX of the wall time) and the barrier.
A 1-GPU box cannot run two RCCL ranks (duplicate device); this proves the library, the torch binding and the call pattern."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
buf = torch.arange(5 * 17, dtype=torch.float64, device="cuda").reshape(5, 17)
out = [torch.zeros_like(buf)]
dist.all_gather(out, buf)
assert torch.equal(out[0], buf)
cnt = torch.tensor([5], dtype=torch.int64, device="cuda")
cnts = [torch.zeros_like(cnt)]
dist.all_gather(cnts, cnt)
assert int(cnts[0].item()) == 5
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
dist.barrier()
o = [None]
dist.all_gather_object(o, (0, 0, "x"))
assert o[0] == (0, 0, "x")
print("rccl smoke ok: backend", dist.get_backend(), "nccl version", torch.cuda.nccl.version())
dist.destroy_process_group()
