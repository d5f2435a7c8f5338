"""The prover's upload/commit pipeline alone, for a rocprofv3 --kernel-trace --memory-copy-trace timeline (development tool)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
ctx = zk.Context(0)
k, cols = 19, 128
n = 1 << k
usable = n - 40
many = [ctx.host_zeros((usable, 4)) for _ in range(cols)]
for h in many:
    h[:3000] = np.arange(12000, dtype=np.uint64).reshape(3000, 4)
bid = ctx.register_bases(ctx.g1_multiples(n))
d2 = ctx.alloc(cols * n * 32)
for rep in range(2):
    ctx.sync()
    t0 = time.perf_counter()
    pending = None
    times = []
    for g0 in range(0, cols, 32):
        for c in range(g0, g0 + 32):
            ctx.upload_overlapped(many[c], d2 + c * n * 32)
        if pending is not None:
            t1 = time.perf_counter()
            ctx.msm_batch_dev(bid, pending, n)
            times.append(round((time.perf_counter() - t1) * 1e3, 1))
        ctx.upload_fence()
        pending = [d2 + c * n * 32 for c in range(g0, g0 + 32)]
    t1 = time.perf_counter()
    ctx.msm_batch_dev(bid, pending, n)
    times.append(round((time.perf_counter() - t1) * 1e3, 1))
    ctx.sync()
    print(f"pipeline: {(time.perf_counter() - t0) * 1e3:.1f} ms; commitments per group {times}", flush=True)
