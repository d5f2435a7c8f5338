"""cProfile of one stand-in create_proof (host side): python tools/proof_profile.py"""
import cProfile, faulthandler, os, pstats, sys, time
faulthandler.dump_traceback_later(int(os.environ.get("WATCHDOG", "150")), exit=True)
T0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import zkhip_pkg
zk = zkhip_pkg.load()
from zkevm_circuits_amd import prover, synthetic
print("ctx", time.perf_counter() - T0, flush=True)
ctx = zk.Context(0)
print("ctx ok", flush=True)
k = 19
cs, _shape = synthetic.supercircuit_standin("standard")
dom = zk.Domain(ctx, cs.degree(), k)
bid = ctx.register_bases(ctx.g1_multiples(dom.n))
print("bases ok", flush=True)
params = prover.ParamsKZG.from_registered(ctx, dom, bid, bid)
advice, instance, fixed, asm = synthetic.standin_witness(cs, dom.n, k, active_rows=3000, zeros=ctx.host_zeros)
print("witness ok", time.perf_counter() - T0, flush=True)
pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
if os.environ.get("TRACE_SYNC"):
    def wrap(obj, name):
        f = getattr(obj, name)
        def g(*a, **kw):
            r = f(*a, **kw)
            ctx.sync()
            print("  done", name, round(time.perf_counter() - T0, 3), flush=True)
            return r
        setattr(obj, name, g)
    for name in ("evaluate_h", "divide_by_vanishing_poly_dev", "extended_to_coeff_dev", "coeff_to_extended_dev"):
        wrap(dom, name)
    wrap(ctx, "msm_batch_dev")
print("keygen ok", time.perf_counter() - T0, flush=True)
ws = prover.ProofWorkspace(ctx)
prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite(), workspace=ws)
print("first proof ok", time.perf_counter() - T0, flush=True)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
ph = {}
prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite(), ph, workspace=ws)
pr.disable()
print("create_proof", time.perf_counter() - t0, ph)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
