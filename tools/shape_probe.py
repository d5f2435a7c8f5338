import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import zkhip_pkg
zk = zkhip_pkg.load()
from zkevm_circuits_amd import evalh_host, ffi, plonk, synthetic
ctx = zk.Context(0)
cs = synthetic.supercircuit_standin("standard")[0]
dom = zk.Domain(ctx, cs.degree(), 12)
ev = plonk.Evaluator(cs)
print(evalh_host.prepare(dom, cs, ev, (ffi.TERMS_HIGH, 4)))
