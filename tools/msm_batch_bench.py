"""Times a pipelined batch of MSMs (development tool): python tools/msm_batch_bench.py [batch] [uniform|permuted]
permuted: the columns are the permuted input columns A' of lookup arguments (sorted by value), as the prover commits them."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
ctx = zk.Context(0)
if os.environ.get("TORCH_STREAM"):        # as bench.py and the prover run: the context on a stream of torch's
    _stream = torch.cuda.Stream()
    ctx.set_stream(_stream.cuda_stream)
k, batch = 19, int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 1 << k
bases = ctx.g1_multiples(n); bid = ctx.register_bases(bases)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
cols = []
for _ in range(8):
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=gen); t[:, 3] &= 0x0FFFFFFFFFFFFFFF
    cols.append(t)
mode = sys.argv[2] if len(sys.argv) > 2 else "uniform"
if mode in ("sparse", "fewbuckets", "bytes", "u64"):
    # witness-like columns: 3000 small values then zeros / a byte in every row / a 64-bit value in every row (canonical
    # value v is stored as v * R mod r; the device multiplies by R through a product with R^2)
    from zkevm_circuits_amd.prover import fr_limbs
    import numpy as np
    rng = np.random.default_rng(3)
    new = []
    for _ in range(8):
        if mode == "sparse":
            vals = [int(v) for v in rng.integers(0, 1 << 30, size=3000)] + [0] * (n - 3000)
        elif mode == "fewbuckets":       # 3000 values below 11: ten buckets of 300 entries
            vals = [int(v) for v in rng.integers(0, 11, size=3000)] + [0] * (n - 3000)
        elif mode == "bytes":
            vals = [int(v) for v in rng.integers(0, 256, size=n)]
        else:
            vals = [int(v) for v in rng.integers(0, 1 << 63, size=n)]
        new.append(torch.from_numpy(fr_limbs(vals).view(np.int64)).cuda())
    cols = new
if mode == "permuted":
    dom = zk.Domain(ctx, 4, k)
    out = []
    for t in cols:
        idx = torch.randint(0, n - 8, (n,), device="cuda", generator=gen)
        inp = t[idx].contiguous()
        pa, ps = torch.zeros_like(t), torch.zeros_like(t)
        torch.cuda.synchronize()          # the context runs on a stream of its own
        dom.lookup_permute_dev(inp.data_ptr(), t.data_ptr(), n - 6, pa.data_ptr(), ps.data_ptr())
        out += [pa, ps]
    cols = out[:8]
if mode == "distinct":      # as many different dense columns as the batch has members: they stream from HBM, as in a proof
    cols = []
    for _ in range(batch):
        t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=gen); t[:, 3] &= 0x0FFFFFFFFFFFFFFF
        cols.append(t)
torch.cuda.synchronize()
ptrs = [cols[i % len(cols)].data_ptr() for i in range(batch)]
ctx.msm_batch_dev(bid, ptrs[:16], n)   # warm-up: allocates every workspace of the pipeline
t0 = time.perf_counter()
ctx.msm_batch_dev(bid, ptrs, n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / batch
print(f"msm batch {batch} ({mode}): {dt*1e3:.3f} ms per MSM  {n/dt/1e6:.1f} Mpts/s")
