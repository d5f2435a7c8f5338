"""coeff_to_extended against coeff_to_cosets over all cosets (same values, compact layout): python tools/cosets_bench.py [k] [j]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
j = int(sys.argv[2]) if len(sys.argv) > 2 else 9
batch = 16
ctx = zk.Context(0)
dom = zk.Domain(ctx, j, k)
n, en = 1 << k, dom.extended_len
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
coeff = torch.randint(0, 1 << 60, (batch, n, 4), dtype=torch.int64, device="cuda", generator=gen)
ext = torch.empty((batch, en, 4), dtype=torch.int64, device="cuda")
cosets = list(range(en // n))
torch.cuda.synchronize()
for name, fn in (("coeff_to_extended", lambda: dom.coeff_to_extended_dev(coeff.data_ptr(), ext.data_ptr(), batch)),
                 ("coeff_to_cosets(all)", lambda: dom.coeff_to_cosets_dev(coeff.data_ptr(), cosets, ext.data_ptr(), batch))):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    ctx.sync()
    print(f"{name:24s} k={k} ext=2^{dom.extended_k} batch={batch}: {(time.perf_counter() - t0) / 5 / batch * 1e3:.3f} ms per column")
