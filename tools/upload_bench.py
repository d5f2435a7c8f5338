"""Host-to-device rate of witness-column uploads from page-locked memory (development tool):
python tools/upload_bench.py [columns]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg
zk = zkhip_pkg.load()
ctx = zk.Context(0)
k, cols = 19, int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 1 << k
usable = n - 40
host = [ctx.host_zeros((usable, 4)) for _ in range(8)]
for h in host:
    h[:3000] = np.arange(12000, dtype=np.uint64).reshape(3000, 4)
blind = ctx.host_zeros((cols, 40, 4))
d = ctx.alloc(cols * n * 32)
gb = cols * usable * 32 / 1e9
for label, with_blind, overlapped in (("columns only, context stream", False, False), ("columns + blinding rows, context stream", True, False),
                                      ("columns only, transfer stream", False, True), ("columns + blinding rows, transfer stream", True, True)):
    for rep in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        for c in range(cols):
            up = ctx.upload_overlapped if overlapped else ctx.upload
            up(host[c % 8], d + c * n * 32)
            if with_blind:
                up(blind[c], d + c * n * 32 + usable * 32)
        ctx.sync()
        dt = time.perf_counter() - t0
    print(f"{label}: {gb / dt:.1f} GB/s ({dt * 1e3:.1f} ms for {cols} columns)")

# the same copies from as many distinct buffers as columns, and beside a running batch of commitments
import threading
import torch
many = [ctx.host_zeros((usable, 4)) for _ in range(cols)]
for h in many:
    h[:3000] = host[0][:3000]
bid = ctx.register_bases(ctx.g1_multiples(n))
ptrs = [d + c * n * 32 for c in range(cols)]
ctx.msm_batch_dev(bid, ptrs[:8], n)
for label, bufs, busy in (("distinct buffers, transfer stream", many, False), ("8 buffers, transfer stream, commitments running", host, True),
                          ("distinct buffers, transfer stream, commitments running", many, True)):
    d2 = ctx.alloc(cols * n * 32)
    for rep in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        for c in range(cols):
            ctx.upload_overlapped(bufs[c % len(bufs)], d2 + c * n * 32)
        t_msm = 0.0
        if busy:
            t1 = time.perf_counter()
            ctx.msm_batch_dev(bid, ptrs, n)          # reads the first allocation: no dependency on the copies
            t_msm = time.perf_counter() - t1
        ctx.sync()
        dt = time.perf_counter() - t0
    print(f"{label}: {gb / dt:.1f} GB/s ({dt * 1e3:.1f} ms for {cols} columns; commitments {t_msm * 1e3:.1f} ms)")
    ctx.free(d2)

# the prover's pipeline: groups of 32 columns, the copies of a group beside the commitments of the group before
for label, with_blind in (("pipeline, columns only", False), ("pipeline, columns + blinding rows", True)):
    d2 = ctx.alloc(cols * n * 32)
    for rep in range(2):
        ctx.sync()
        times = []
        t0 = time.perf_counter()
        pending = None
        for g0 in range(0, cols, 32):
            for c in range(g0, g0 + 32):
                ctx.upload_overlapped(many[c], d2 + c * n * 32)
                if with_blind:
                    ctx.upload_overlapped(blind[c], d2 + c * n * 32 + usable * 32)
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
        dt = time.perf_counter() - t0
    print(f"{label}: {dt * 1e3:.1f} ms for {cols} columns; commitments per group {times}")
    ctx.free(d2)
