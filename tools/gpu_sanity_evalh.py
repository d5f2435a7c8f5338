import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import zkhip_pkg
zk = zkhip_pkg.load()
from zkevm_circuits_amd import ffi, plonk, synthetic, evalh_host
import coracle as co
rng = np.random.default_rng(11)
def rand_fr(n):
    a = rng.integers(0, 2**64, size=(n,4), dtype=np.uint64); a[:,3] &= np.uint64(0x0fffffffffffffff); return a
ctx = zk.Context(0)
for (k, kw) in ((6, dict()), (8, dict(num_advice=12, num_gate_polys=40, num_lookups=3, num_perm_cols=7)),
                (10, dict(num_advice=20, num_fixed=6, num_gate_polys=120, num_lookups=4, num_perm_cols=9, rotations=(-5,-2,-1,0,1,3)))):
    cs = synthetic.synthetic_cs(**kw)
    ev = plonk.Evaluator(cs)
    j = cs.degree()
    dom = zk.Domain(ctx, j, k); od = co.Domain(j, k, threads=8)
    n, en = dom.n, dom.extended_len
    print("k", k, "degree", j, "ext_k", dom.extended_k, "gate calcs", len(ev.custom_gates.calculations), "blinding", cs.blinding_factors(), "sets", cs.num_permutation_sets(), flush=True)
    call = evalh_host.EvalhCall(cs, ev)
    call.fixed = [rand_fr(en) for _ in range(cs.num_fixed_columns)]
    adv_coeff = [rand_fr(n) for _ in range(cs.num_advice_columns)]
    ins_coeff = [rand_fr(n) for _ in range(cs.num_instance_columns)]
    call.challenges = rand_fr(cs.num_challenges)
    call.y, call.beta, call.gamma, call.theta = rand_fr(1)[0], rand_fr(1)[0], rand_fr(1)[0], rand_fr(1)[0]
    call.l0, call.l_last, call.l_active = rand_fr(en), rand_fr(en), rand_fr(en)
    lk_coeff = [(rand_fr(n), rand_fr(n), rand_fr(n)) for _ in cs.lookups]
    call.perm_sigma = [rand_fr(en) for _ in cs.permutation_columns]
    call.perm_z = [rand_fr(en) for _ in range(cs.num_permutation_sets())]
    # extended form via oracle
    call.advice = [od.coeff_to_extended(a) for a in adv_coeff]
    call.instance = [od.coeff_to_extended(a) for a in ins_coeff]
    call.lookups = [tuple(od.coeff_to_extended(p) for p in t) for t in lk_coeff]
    desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
    t=time.time(); o = od.evaluate_h(desc); to=time.time()-t
    t=time.time(); g = dom.evaluate_h(desc); tg=time.time()-t
    print("  extended/host:", "ok" if np.array_equal(g,o) else "MISMATCH", f"gpu {tg*1e3:.1f} ms cpu {to*1e3:.1f} ms", flush=True)
    if not np.array_equal(g,o):
        bad = np.nonzero(np.any(g!=o,axis=1))[0]; print("   bad rows", len(bad), bad[:10])
    # coefficient form
    call.advice, call.instance, call.lookups = adv_coeff, ins_coeff, lk_coeff
    desc2 = call.build(ffi.MEM_HOST, ffi.FORM_COEFF)
    g2 = dom.evaluate_h(desc2)
    print("  coeff/host:", "ok" if np.array_equal(g2,o) else "MISMATCH", flush=True)
    # device-resident extended
    call.advice = [ctx.upload(od.coeff_to_extended(a)) for a in adv_coeff]
    call.instance = [ctx.upload(od.coeff_to_extended(a)) for a in ins_coeff]
    call.lookups = [tuple(ctx.upload(od.coeff_to_extended(p)) for p in t) for t in lk_coeff]
    call.fixed = [ctx.upload(a) for a in call.fixed]
    call.l0, call.l_last, call.l_active = ctx.upload(call.l0), ctx.upload(call.l_last), ctx.upload(call.l_active)
    call.perm_sigma = [ctx.upload(a) for a in call.perm_sigma]; call.perm_z = [ctx.upload(a) for a in call.perm_z]
    d_out = ctx.alloc(en*32)
    desc3 = call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED)
    dom.evaluate_h(desc3, d_out)
    g3 = ctx.download(d_out, (en,4))
    print("  extended/device:", "ok" if np.array_equal(g3,o) else "MISMATCH", flush=True)
    # h pipeline: divide by vanishing, extended_to_coeff
    hq = od.extended_to_coeff(od.divide_by_vanishing_poly(o))
    gq = dom.extended_to_coeff(dom.divide_by_vanishing_poly(g))
    print("  divide+extended_to_coeff:", "ok" if np.array_equal(hq,gq) else "MISMATCH", flush=True)
    a = rand_fr(n)
    print("  l2c:", np.array_equal(dom.lagrange_to_coeff(a), od.lagrange_to_coeff(a)), "c2l:", np.array_equal(dom.coeff_to_lagrange(a), od.coeff_to_lagrange(a)), "c2e:", np.array_equal(dom.coeff_to_extended(a), od.coeff_to_extended(a)))
