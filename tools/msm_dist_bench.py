"""Development tool: phase times of one 2^k MSM for the witness-like (W) and 16-bit (B) scalar distributions."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
import bench
ctx = zk.Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
n = 1 << k
bid = ctx.register_bases(ctx.g1_multiples(n))
for which in sys.argv[2:] or ["W", "B"]:
    d_sc = ctx.upload(bench.scalars_by_distribution(ctx, n, which, 20250523))
    ctx.msm_range_dev(bid, 0, d_sc, n)
    ctx.timing_enable(True); ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(5): ctx.msm_range_dev(bid, 0, d_sc, n)
    dt = (time.perf_counter() - t0) / 5
    print(f"msm 2^{k} {which}: {dt*1e3:.3f} ms", {f: round(ctx.timing_get(f)[0], 1) for f in ("msm_sort", "msm_accumulate", "msm_merge", "msm_reduce")})
    ctx.timing_enable(False)
    ctx.free(d_sc)
