"""Times a pipelined batch of MSMs (development tool)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
ctx = zk.Context(0)
k, batch = 19, int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 1 << k
bases = ctx.g1_multiples(n); bid = ctx.register_bases(bases)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
cols = []
for _ in range(8):
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=gen); t[:, 3] &= 0x0FFFFFFFFFFFFFFF
    cols.append(t)
torch.cuda.synchronize()
ptrs = [cols[i % 8].data_ptr() for i in range(batch)]
ctx.msm_batch_dev(bid, ptrs[:4], n)
t0 = time.perf_counter()
ctx.msm_batch_dev(bid, ptrs, n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / batch
print(f"msm batch {batch}: {dt*1e3:.3f} ms per MSM  {n/dt/1e6:.1f} Mpts/s")
