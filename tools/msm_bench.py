import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
ctx = zk.Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << k
bases = ctx.g1_multiples(n); bid = ctx.register_bases(bases)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=gen); t[:, 3] &= 0x0FFFFFFFFFFFFFFF
torch.cuda.synchronize()
ctx.msm_range_dev(bid, 0, t.data_ptr(), n)
timing = os.environ.get('MSM_BENCH_TIMING', '1') != '0'
ctx.timing_enable(timing); ctx.timing_reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): ctx.msm_range_dev(bid, 0, t.data_ptr(), n)
dt = (time.perf_counter() - t0) / reps
print(f"msm 2^{k}: {dt*1e3:.3f} ms  {n/dt/1e6:.1f} Mpts/s", {f: round(ctx.timing_get(f)[0],1) for f in ("msm_sort","msm_accumulate","msm_merge","msm_reduce","msm_host_fold")})
