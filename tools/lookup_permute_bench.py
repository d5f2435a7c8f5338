"""lookup_permute at 2^k rows: single calls and one batched call over `count` lookups (for rocprofv3 --kernel-trace
--stats): python tools/lookup_permute_bench.py [k] [count]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = zk.Context(0)
dom = zk.Domain(ctx, 4, k)
n = 1 << k
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
tables, inputs, pas, pss = [], [], [], []
for l in range(count):
    if l % 3 == 2:      # every third lookup: table indices below 2^16, as the fixed-table lookups of the circuits are
        table = (torch.arange(n, device="cuda") % 65536).to(torch.int64)[:, None] * torch.tensor([[1, 0, 0, 0]], device="cuda")
    else:               # theta-compressed expressions: full-width values
        table = torch.randint(0, 1 << 60, (n, 4), dtype=torch.int64, device="cuda", generator=gen)
    idx = torch.randint(0, n - 8, (n,), device="cuda", generator=gen)
    tables.append(table.contiguous()); inputs.append(table[idx].contiguous())
    pas.append(torch.zeros_like(table)); pss.append(torch.zeros_like(table))
torch.cuda.synchronize()
ptr = lambda ts: [t.data_ptr() for t in ts]
dom.lookup_permute_dev(inputs[0].data_ptr(), tables[0].data_ptr(), n - 6, pas[0].data_ptr(), pss[0].data_ptr())
t0 = time.perf_counter()
for _ in range(5):
    dom.lookup_permute_dev(inputs[0].data_ptr(), tables[0].data_ptr(), n - 6, pas[0].data_ptr(), pss[0].data_ptr())
print(f"lookup_permute 2^{k}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per single call")
dom.lookup_permute_batch_dev(ptr(inputs), ptr(tables), n - 6, ptr(pas), ptr(pss))
t0 = time.perf_counter()
for _ in range(3):
    dom.lookup_permute_batch_dev(ptr(inputs), ptr(tables), n - 6, ptr(pas), ptr(pss))
dt = (time.perf_counter() - t0) / 3
print(f"lookup_permute 2^{k}, {count} lookups in one call: {dt * 1e3:.2f} ms = {dt / count * 1e3:.3f} ms per lookup")
