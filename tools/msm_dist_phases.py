import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import zkhip_pkg
zk = zkhip_pkg.load()
import bench
ctx = zk.Context(0)
n = 1 << 19
bases = ctx.g1_multiples(n); bid = ctx.register_bases(bases)
for which in ("W", "B"):
    d_sc = ctx.upload(bench.scalars_by_distribution(ctx, n, which, 20250523))
    ctx.msm_range_dev(bid, 0, d_sc, n)
    ctx.timing_enable(True); ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(10): ctx.msm_range_dev(bid, 0, d_sc, n)
    dt = (time.perf_counter() - t0) / 10
    print(which, f"{dt*1e3:.3f} ms", {f: round(ctx.timing_get(f)[0],1) for f in ("msm_sort","msm_accumulate","msm_merge","msm_reduce","msm_host_fold")})
    ctx.timing_enable(False)
