"""Which operation makes later page-locked uploads slow (development tool)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
ctx = zk.Context(0)
n = 1 << 19
bufs = [ctx.host_zeros((n - 34, 4)) for _ in range(64)]
d = ctx.alloc(64 * n * 32)
def rate(label):
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        for i, a in enumerate(bufs): ctx.upload_overlapped(a, d + i * n * 32)
        ctx.sync(); dt = time.perf_counter() - t0
    print(f"[rate] {label}: {sum(a.nbytes for a in bufs) / dt / 1e9:.1f} GB/s", flush=True)
rate("start")
x = ctx.download(d, (1, 4)); rate("after a small download to ordinary memory")
x = ctx.download(d, (n, 4)); rate("after a 16 MB download to ordinary memory")
p = ctx.upload(np.zeros((4, 4), dtype=np.uint64)); rate("after a small upload from ordinary memory")
p = ctx.upload(np.zeros((n, 4), dtype=np.uint64)); rate("after a 16 MB upload from ordinary memory")
p = ctx.upload(np.zeros((8 * n, 4), dtype=np.uint64)); rate("after a 128 MB upload from ordinary memory")
bid = ctx.register_bases(ctx.g1_multiples(n)); rate("after register_bases")
ctx.msm_batch_dev(bid, [d + i * n * 32 for i in range(16)], n); rate("after an MSM batch")
dom = zk.Domain(ctx, 4, 19)
import torch
t = torch.randint(0, 1 << 60, (n, 4), dtype=torch.int64, device="cuda"); pa = torch.zeros_like(t); ps = torch.zeros_like(t)
torch.cuda.synchronize()
dom.lookup_permute_dev(t.data_ptr(), t.data_ptr(), n - 6, pa.data_ptr(), ps.data_ptr()); rate("after lookup_permute")
h = ctx.host_zeros((40, 4)); ctx.host_free(h); rate("after freeing a page-locked buffer")
q = ctx.alloc(1 << 30); ctx.free(q); rate("after freeing 1 GiB of device memory")
