"""CPU suite: the proof fixture made on the GPU (tests/golden/proof_k7.json) replayed through the big-int verifier
model against a verifying key that is computed here without the device — the fixed and permutation commitments follow
from the circuit's columns and the SRS trapdoor as [f(tau)] G."""
import pytest

import bn254 as bn
import halo2_model as hm
import prover_cases as pc
import verifier_model as vm
from util import golden

P = bn.FR_MODULUS


def _lagrange_commitment(values, omega, n):
    """[f(tau)] G for the polynomial with f(omega^i) = values[i]"""
    tau = pc.TAU
    num = (pow(tau, n, P) - 1) * pow(n, P - 2, P) % P
    acc = 0
    for i, v in enumerate(values):
        if v:
            w = pow(omega, i, P)
            acc = (acc + v * num % P * w % P * pow((tau - w) % P, P - 2, P)) % P
    return bn.g1_mul(bn.G1_GENERATOR, acc)


@pytest.fixture(scope="module")
def case():
    gold = golden("proof_k7.json")
    k = gold["k"]
    n = 1 << k
    cs = pc.circuit()
    _, instance, fixed, asm = pc.witness(cs, n, gold["witness_seed"])
    omega = hm.EvaluationDomain(cs.degree(), k).omega
    mapping = [list(zip((int(x) for x in asm.map_col[c]), (int(x) for x in asm.map_row[c])))
               for c in range(asm.num_columns)]
    vk = {"fixed_commitments": [_lagrange_commitment(col, omega, n) for col in fixed],
          "permutation_commitments": [_lagrange_commitment(col, omega, n) for col in hm.permutation_sigmas(omega, mapping)],
          "transcript_repr": int(gold["transcript_repr"], 16)}
    return gold, k, cs, vk, instance


def test_fixture_key_matches_the_circuit(case):
    gold, k, cs, vk, instance = case
    assert int(gold["tau"], 16) == pc.TAU
    assert [[hex(x) for x in p] for p in vk["fixed_commitments"]] == gold["fixed_commitments"]
    assert [[hex(x) for x in p] for p in vk["permutation_commitments"]] == gold["permutation_commitments"]
    assert [[hex(v) for v in col] for col in instance] == gold["instance"]


def test_golden_proof_verifies(case):
    gold, k, cs, vk, instance = case
    vm.verify_proof(cs, k, vk, instance, bytes.fromhex(gold["proof"]), pc.TAU)


def test_golden_proof_tampered_is_rejected(case):
    gold, k, cs, vk, instance = case
    proof = bytes.fromhex(gold["proof"])
    for pos in (0, 32 * 9 + 3, len(proof) - 32 * 12, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 2
        with pytest.raises(vm.VerifyError):
            vm.verify_proof(cs, k, vk, instance, bytes(bad), pc.TAU)
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, [[instance[0][1], instance[0][0]]], proof, pc.TAU)
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, instance, proof, pc.TAU + 1)
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, instance, proof + b"\0", pc.TAU)


def test_transcript_and_point_encoding_round_trip():
    import numpy as np
    import zkevm_circuits_amd  # noqa: F401
    from zkevm_circuits_amd import prover
    w = prover.Blake2bWrite()
    pts = [bn.g1_mul(bn.G1_GENERATOR, s) for s in (1, 2, 12345, P - 1)]
    w.common_scalar(5)
    for p in pts:
        w.write_point(np.array(bn.g1_affine_to_limbs(p), dtype=np.uint64))
    c1 = w.squeeze_challenge()
    w.write_scalar(P - 1)
    c2 = w.squeeze_challenge()
    assert c1 != c2 and prover.compress_point(None)[31] == 0x80 and vm.decompress_point(prover.compress_point(None)) is None
    r = vm.Blake2bRead(w.finalize())
    r.common_scalar(5)
    assert [r.read_point() for _ in pts] == pts
    assert r.squeeze_challenge() == c1 and r.read_scalar() == P - 1 and r.squeeze_challenge() == c2
    with pytest.raises(vm.VerifyError):
        r.read_scalar()
    with pytest.raises(ValueError):
        w.write_point(np.zeros(8, dtype=np.uint64))
    # a challenge is the 64-byte digest reduced mod r (Challenge255 / from_uniform_bytes), after the prefix byte 0
    import hashlib
    h = hashlib.blake2b(b"\x00", digest_size=64, person=b"Halo2-Transcript")
    assert prover.Blake2bWrite().squeeze_challenge() == int.from_bytes(h.digest(), "little") % P
    assert prover.collect_queries(pc.circuit()) == vm.collect_queries(pc.circuit())
