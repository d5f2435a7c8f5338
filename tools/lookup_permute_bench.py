"""lookup_permute at 2^k rows, a few calls (for rocprofv3 --kernel-trace --stats): python tools/lookup_permute_bench.py [k]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
ctx = zk.Context(0)
dom = zk.Domain(ctx, 4, k)
n = 1 << k
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
table = torch.randint(0, 1 << 60, (n, 4), dtype=torch.int64, device="cuda", generator=gen)
idx = torch.randint(0, n - 8, (n,), device="cuda", generator=gen)
inp = table[idx].contiguous()
pa, ps = torch.zeros_like(table), torch.zeros_like(table)
torch.cuda.synchronize()
dom.lookup_permute_dev(inp.data_ptr(), table.data_ptr(), n - 6, pa.data_ptr(), ps.data_ptr())
t0 = time.perf_counter()
for _ in range(5):
    dom.lookup_permute_dev(inp.data_ptr(), table.data_ptr(), n - 6, pa.data_ptr(), ps.data_ptr())
print(f"lookup_permute 2^{k}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call")
