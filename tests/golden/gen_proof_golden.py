#!/usr/bin/env python3
"""Writes tests/golden/proof_k7.json: one proof of the test circuit of tests/test_gpu_prover.py made on the GPU (seeded
RNG), with the verifying-key commitments and the public inputs the verifier model needs.  Needs an MI355X:
    python tests/golden/gen_proof_golden.py [output path]
The CPU suite replays the file through the verifier model; the GPU suite checks that the prover still produces these
bytes."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402,F401
import zkhip_pkg  # noqa: E402

zk = zkhip_pkg.load()
import coracle  # noqa: E402
import prover_cases as tp  # noqa: E402
from zkevm_circuits_amd import prover  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "proof_k7.json")
    k, seed = 7, 1
    ctx = zk.Context(0)
    cs = tp.circuit()
    dom = zk.Domain(ctx, cs.degree(), k)
    params = prover.ParamsKZG(ctx, dom, tp.srs(coracle, dom.n))
    advice, instance, fixed, asm = tp.witness(cs, dom.n, 7)
    pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
    proof = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(seed), prover.Blake2bWrite())
    pt = lambda p: [hex(p[0]), hex(p[1])]
    with open(out, "w") as f:
        json.dump({"k": k, "rng_seed": seed, "witness_seed": 7, "tau": hex(tp.TAU),
                   "instance": [[hex(v) for v in col] for col in instance],
                   "fixed_commitments": [pt(p) for p in pk.fixed_commitments],
                   "permutation_commitments": [pt(p) for p in pk.permutation_commitments],
                   "transcript_repr": hex(pk.transcript_repr), "proof": proof.hex()}, f, indent=1)
    print("wrote", out, len(proof), "bytes")


if __name__ == "__main__":
    main()
