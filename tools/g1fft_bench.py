"""Development tool: time of g_to_lagrange for 2^k SRS points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zkhip_pkg
zk = zkhip_pkg.load()
from zkevm_circuits_amd import keygen
ctx = zk.Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
dom = zk.Domain(ctx, 3, k)
g = ctx.g1_multiples(1 << k)
t0 = time.perf_counter()
gl = keygen.g_to_lagrange(ctx, dom, g)
print(f"g_to_lagrange 2^{k}: {time.perf_counter() - t0:.3f} s")
