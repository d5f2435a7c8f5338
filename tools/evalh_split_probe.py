"""Low-degree terms on half the extended domain: what the three programs cost (multiplications per row) and how long the
whole-domain call and the split evaluation take, for the benchmark's constraint system, the census-sized SuperCircuit
stand-in and the replay's shape.  python tools/evalh_split_probe.py [k]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkhip_pkg  # noqa: E402

zk = zkhip_pkg.load()
import bench  # noqa: E402
from zkevm_circuits_amd import evalh_host, ffi, plonk, synthetic  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
ctx = zk.Context(0)
gen = torch.Generator(device="cuda")
gen.manual_seed(3)
census = synthetic.supercircuit_shape("standard")
shapes = {
    "bench": bench.bench_cs(),
    "standin": synthetic.supercircuit_standin("standard")[0],
    "replay": synthetic.synthetic_cs(num_advice=census["advice"], num_fixed=census["fixed"], num_instance=1, num_gate_polys=3000,
                                     max_degree=9, rotations=(-5, -2, -1, 0, 1), num_lookups=census["lookups"], lookup_width=4,
                                     num_perm_cols=census["permutation_columns"]),
}
out = {}
for name, cs in shapes.items():
    dom = zk.Domain(ctx, cs.degree(), k)
    en = dom.extended_len
    ev = plonk.Evaluator(cs)
    low, high, whole = evalh_host.split_gain(dom, cs, ev)
    # degree histogram of the gate polynomials and the lookups' required degrees
    hist = {}
    for g in cs.gates:
        for p in g:
            d = plonk.expr_degree(p) if hasattr(plonk, "expr_degree") else None
            hist[d] = hist.get(d, 0) + 1
    ncols = (cs.num_fixed_columns + cs.num_advice_columns + cs.num_instance_columns + 3 + 3 * len(cs.lookups)
             + len(cs.permutation_columns) + cs.num_permutation_sets())
    cols = torch.randint(-(1 << 63), (1 << 63) - 1, (4, en, 4), dtype=torch.int64, device="cuda", generator=gen)
    cols[:, :, 3] &= 0x0FFFFFFFFFFFFFFF
    # timing only: every column pointer cycles over four random vectors (the kernels read them all the same)
    ptr = [cols[i % 4].data_ptr() for i in range(ncols)]
    it = iter(ptr)
    take = lambda c: [next(it) for _ in range(c)]                                          # noqa: E731
    call = evalh_host.EvalhCall(cs, ev)
    call.fixed, call.advice, call.instance = take(cs.num_fixed_columns), take(cs.num_advice_columns), take(cs.num_instance_columns)
    call.l0, call.l_last, call.l_active = take(3)
    call.lookups = [tuple(take(3)) for _ in cs.lookups]
    call.perm_sigma, call.perm_z = take(len(cs.permutation_columns)), take(cs.num_permutation_sets())
    sc = torch.randint(0, 1 << 62, (8, 4), dtype=torch.int64).numpy().view(np.uint64)
    call.challenges = sc[4:4 + cs.num_challenges] if cs.num_challenges <= 4 else np.tile(sc, (cs.num_challenges, 1))[:cs.num_challenges]
    call.y, call.beta, call.gamma, call.theta = sc[0], sc[1], sc[2], sc[3]
    d_h = ctx.alloc(en * 32)

    def timed(fn):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        return time.perf_counter() - t0
    t_whole = timed(lambda: dom.evaluate_h(call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), d_h))
    t_split = timed(lambda: evalh_host.evaluate_h_split(ctx, dom, call, d_h))
    out[name] = {"k": k, "extended_k": dom.extended_k, "multiplications_per_row": {"low": low, "high": high, "whole": whole},
                 "low_share": low / whole, "predicted_ratio": (low + 2 * high) / (2.0 * whole),
                 "whole_s": t_whole, "split_s": t_split, "measured_ratio": t_split / t_whole,
                 "gate_degree_histogram": {str(d): c for d, c in sorted(hist.items(), key=lambda x: str(x[0]))}}
    ctx.free(d_h)
    del cols
    dom.close()
print(json.dumps(out))
