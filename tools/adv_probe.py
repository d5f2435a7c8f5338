import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import zkhip_pkg
zk = zkhip_pkg.load()
from zkevm_circuits_amd import prover, synthetic
ctx = zk.Context(0)
k = 19
_probe = {}
def rate_test(label, arrays=None):
    import numpy as np
    n_ = 1 << 19
    if "bufs" not in _probe:
        _probe["bufs"] = [ctx.host_zeros((n_ - 34, 4)) for _ in range(64)]
        _probe["d"] = ctx.alloc(64 * n_ * 32)
    bufs = arrays or _probe["bufs"]
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        for i, a in enumerate(bufs[:64]): ctx.upload_overlapped(a, _probe["d"] + i * n_ * 32)
        ctx.sync(); dt = time.perf_counter() - t0
    print(f"[rate] {label}: {sum(a.nbytes for a in bufs[:64]) / dt / 1e9:.1f} GB/s", flush=True)
rate_test("start")
cs = synthetic.standin_cs(units=52, lanes=8, wide_lookups=8, lookup_width=10, extra_gates=38)
dom = zk.Domain(ctx, cs.degree(), k)
bid = ctx.register_bases(ctx.g1_multiples(dom.n))
params = prover.ParamsKZG.from_registered(ctx, dom, bid, bid)
advice, instance, fixed, asm = synthetic.standin_witness(cs, dom.n, k, active_rows=3000, zeros=ctx.host_zeros)
rate_test("after witness generation")
rate_test("witness arrays, after witness generation", [a for a in advice(0, []) if ctx.is_pinned(a)])
pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
rate_test("after keygen")
prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite())
rate_test("after first proof")
rate_test("witness arrays, after first proof", [a for a in advice(0, []) if ctx.is_pinned(a)])
log = []
T0 = [0]
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **kw):
        t = time.perf_counter()
        r = f(*a, **kw)
        log.append((name, round((t - T0[0]) * 1e3, 2), round((time.perf_counter() - t) * 1e3, 2), len(a[0]) if a and hasattr(a[0], "__len__") else 0))
        return r
    setattr(obj, name, g)
wrap(params, "commit_lagrange"); wrap(ctx, "upload_fence"); wrap(ctx, "host_zeros"); wrap(ctx, "host_free"); wrap(ctx, "alloc")
T0[0] = time.perf_counter()
ph = {}
prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite(), ph)
print(ph["advice"])
for e in log[:40]: print(e)
# raw upload rate of the witness arrays, and the commitments alone with the columns resident
cols = [a for a in advice(0, []) ] if callable(advice) else advice
cols = [a for a in cols if ctx.is_pinned(a)]
d = ctx.alloc(len(cols) * dom.n * 32)
for rep in range(2):
    ctx.sync(); t0 = time.perf_counter()
    for i, a in enumerate(cols): ctx.upload_overlapped(a, d + i * dom.n * 32)
    t_submit = time.perf_counter() - t0
    ctx.sync(); dt = time.perf_counter() - t0
print(f"submission {t_submit * 1e3:.1f} ms; pinned buffers known to the context: {len(ctx._pinned)}")
print(f"{len(cols)} pinned columns: {sum(a.nbytes for a in cols) / dt / 1e9:.1f} GB/s, {dt * 1e3:.1f} ms")
ptrs = [d + i * dom.n * 32 for i in range(len(cols))]
for rep in range(2):
    ctx.sync(); t0 = time.perf_counter()
    for g in range(0, len(ptrs), 32): ctx.msm_batch_dev(bid, ptrs[g:g + 32], dom.n)
    dt = time.perf_counter() - t0
print(f"commitments in groups of 32: {dt * 1e3:.1f} ms = {dt / len(ptrs) * 1e3:.3f} ms each")
t0 = time.perf_counter(); ctx.msm_batch_dev(bid, ptrs, dom.n); dt = time.perf_counter() - t0
print(f"commitments in one batch: {dt * 1e3:.1f} ms = {dt / len(ptrs) * 1e3:.3f} ms each")
rates = []
for i, a in enumerate(cols):
    ctx.sync(); t0 = time.perf_counter(); ctx.upload_overlapped(a, d + i * dom.n * 32); ctx.sync()
    rates.append(round(a.nbytes / (time.perf_counter() - t0) / 1e9))
print("per-array GB/s:", rates)
fresh = [ctx.host_zeros(cols[0].shape) for _ in range(16)]
r2 = []
for i, a in enumerate(fresh):
    ctx.sync(); t0 = time.perf_counter(); ctx.upload_overlapped(a, d + i * dom.n * 32); ctx.sync()
    r2.append(round(a.nbytes / (time.perf_counter() - t0) / 1e9))
print("fresh arrays GB/s:", r2)
print("addresses:", [hex(a.ctypes.data) for a in cols[:3]], [hex(a.ctypes.data) for a in cols[-3:]], [hex(a.ctypes.data) for a in fresh[:3]])
