import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
import bench
ctx = zk.Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
dom = zk.Domain(ctx, 5, k)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
desc, out, keep, call, c_read, n_calc = bench.build_evalh(ctx, dom, gen, k)
torch.cuda.synchronize()
dom.evaluate_h(desc, out.data_ptr())
os.environ.pop("ZKHIP_DEBUG", None)
ctx.timing_enable(True); ctx.timing_reset()
t0 = time.perf_counter()
for _ in range(2): dom.evaluate_h(desc, out.data_ptr())
dt = (time.perf_counter() - t0) / 2
print(f"evaluate_h k={k}: {dt*1e3:.2f} ms  {32.0*dom.extended_len*(c_read+1)/dt/1e9:.1f} GB/s", {"kernel_ms": round(ctx.timing_get("evalh")[0]/1e3,2)})
