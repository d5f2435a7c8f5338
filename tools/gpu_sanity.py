import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import zkhip_pkg
zk = zkhip_pkg.load()
import coracle as co

rng = np.random.default_rng(1)
def rand_fe(n, field):
    p = [0x43e1f593f0000001,0x2833e84879b97091,0xb85045b68181585d,0x30644e72e131a029] if field==0 else [0x3c208c16d87cfd47,0x97816a916871ca8d,0xb85045b68181585d,0x30644e72e131a029]
    a = rng.integers(0, 2**64, size=(n,4), dtype=np.uint64)
    a[:,3] &= np.uint64(0x0fffffffffffffff)   # < 2^252 < p
    return a

ctx = zk.Context(0)
for field in (0,1):
    a, b = rand_fe(5000, field), rand_fe(5000, field)
    for op in (0,1,2):
        g = ctx.field_binop(field, op, a, b); o = co.field_binop(field, op, a, b)
        print("field", field, "op", op, "ok" if np.array_equal(g,o) else "MISMATCH", flush=True)
    g = ctx.field_binop(field, 3, a, b); o = co.field_binop(field, 2, co.field_binop(field,2,a,a), b)
    print("field", field, "sqr ok" if np.array_equal(g,o) else "sqr MISMATCH")
print("probe", ctx.probe_mad_rate(1024))
for log_n in (1,2,3,5,8,9,11,12,14,17):
    d = co.Domain(3, log_n)
    omega = d.constant(0)
    a = rand_fe(1<<log_n, 0)
    t=time.time(); g = ctx.ntt(a, log_n, omega); tg=time.time()-t
    o = co.best_fft(a, omega, log_n, threads=8)
    print("ntt", log_n, "ok" if np.array_equal(g,o) else "MISMATCH", f"{tg*1e3:.1f} ms", flush=True)
