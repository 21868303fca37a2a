"""RCCL sanity/latency probe at world_size 1 (what can be checked on a 1-GPU box)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(22, dtype=torch.float64, device="cuda"); out = torch.zeros((1, 22), dtype=torch.float64, device="cuda")
for i in range(20): dist.all_gather_into_tensor(out, x)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(200): dist.all_gather_into_tensor(out, x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("all_gather_into_tensor world=1: host %.1f us/call, total %.1f us/call, sum %.1f" % ((t1 - t) / 200 * 1e6, (t2 - t) / 200 * 1e6, out.sum().item()))
dist.destroy_process_group()
