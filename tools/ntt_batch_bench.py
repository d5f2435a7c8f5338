"""Times coeff_to_extended for a batch of columns (development tool)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
ctx = zk.Context(0)
k, batch = 19, int(sys.argv[1]) if len(sys.argv) > 1 else 64
j = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dom = zk.Domain(ctx, j, k)
n, en = 1 << k, dom.extended_len
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
t = torch.randint(-(1 << 63), (1 << 63) - 1, (batch * n, 4), dtype=torch.int64, device="cuda", generator=gen); t[:, 3] &= 0x0FFFFFFFFFFFFFFF
ext = torch.empty((batch * en, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
dom.coeff_to_extended_dev(t.data_ptr(), ext.data_ptr(), batch)
ctx.sync()
t0 = time.perf_counter()
for _ in range(3):
    dom.coeff_to_extended_dev(t.data_ptr(), ext.data_ptr(), batch)
ctx.sync()
dt = (time.perf_counter() - t0) / 3 / batch
print(f"coset ntt batch {batch}: {dt*1e3:.3f} ms per column  {32*(n+en)/dt/1e9:.1f} GB/s")
