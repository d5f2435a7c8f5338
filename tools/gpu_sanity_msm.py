import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import zkhip_pkg
zk = zkhip_pkg.load()
import coracle as co
rng = np.random.default_rng(7)
def rand_fr(n):
    a = rng.integers(0, 2**64, size=(n,4), dtype=np.uint64); a[:,3] &= np.uint64(0x0fffffffffffffff); return a
def from_ints(vals):
    c = np.array([[(v>>(64*i))&(2**64-1) for i in range(4)] for v in vals], dtype=np.uint64)
    return co.from_canonical(0, c)
ctx = zk.Context(0)
# group law incl. special cases
bases = co.g1_bases(64)
p = np.concatenate([bases[:8], bases[:4], np.zeros((2,8),dtype=np.uint64), bases[5:6]])
neg = bases[:4].copy()
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
q = np.concatenate([bases[8:16], bases[:4], bases[:2], np.zeros((1,8),dtype=np.uint64)])
g = ctx.g1_add_affine(p, q); o = co.g1_add_affine(p, q)
print("g1 add", "ok" if np.array_equal(g,o) else "MISMATCH", flush=True)
for n in (1, 2, 31, 32, 1000, 4096, 1<<14, 1<<16):
    t=time.time(); bases = co.g1_bases(n); tb=time.time()-t
    t=time.time(); bid = ctx.register_bases(bases); treg=time.time()-t
    for name, sc in (("uniform", rand_fr(n)), ("zeros", np.zeros((n,4),dtype=np.uint64)), ("ones", from_ints([1]*n)),
                     ("small", from_ints([int(x) for x in rng.integers(0, 1<<16, size=n)])),
                     ("pm", from_ints([0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001 - 1 - int(x) for x in rng.integers(0, 3, size=n)]))):
        if n > 4096 and name in ("zeros","pm"): continue
        t=time.time(); gj = ctx.msm(bid, sc); tg=time.time()-t
        t=time.time(); oj = co.best_multiexp(sc, bases, threads=8); to=time.time()-t
        ga, oa = ctx.g1_to_affine(gj), co.g1_to_affine(oj)
        print(f"msm n={n} {name}: {'ok' if np.array_equal(ga,oa) else 'MISMATCH'} gpu {tg*1e3:.1f} ms cpu {to*1e3:.1f} ms (reg {treg*1e3:.0f} ms)", flush=True)
    ctx.release_bases(bid)
