import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
P, r = dist.get_world_size(), dist.get_rank()
rows, d = 2912, 128          # scale-0.05 shard piece
x = torch.randn(rows, d)
def t(fn, n=30):
    fn(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dist.barrier()
    return (time.perf_counter() - t0) / n * 1e3
def ag_list():
    parts = [torch.empty_like(x) for _ in range(P)]
    dist.all_gather(parts, x); return torch.cat(parts, 0)
out = torch.empty(P * rows, d)
def ag_tensor(): dist.all_gather_into_tensor(out, x)
def bc():
    for i in range(P):
        b = x if r == i else torch.empty_like(x); dist.broadcast(b, src=i)
res = {"all_gather(list)+cat": t(ag_list), "P broadcasts": t(bc)}
try: res["all_gather_into_tensor"] = t(ag_tensor)
except Exception as e: res["all_gather_into_tensor"] = repr(e)[:80]
if r == 0: print(res)
dist.destroy_process_group()
