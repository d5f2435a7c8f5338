"""End to end on the GPU: key generation and `create_proof` (every polynomial operation on the device, through the C
ABI) for a small circuit with a satisfied witness — gates with rotations and a challenge, a two-column lookup, copy
constraints over two permutation sets, one instance column — and the proof checked by the big-int verifier model, which
evaluates the identities from the opened values and checks the SHPLONK opening with the test SRS's trapdoor."""
import numpy as np
import pytest

import bn254 as bn
import verifier_model as vm
from util import affine_arr

from zkevm_circuits_amd import ffi, keygen, plonk, prover
import zkevm_circuits_amd as zk

pytestmark = pytest.mark.gpu
P = bn.FR_MODULUS
TAU = 0x1F3A9C0D5E6B7788990A1B2C3D4E5F60718293A4B5C6D7E8F9  # trapdoor of the test SRS


def _circuit():
    from zkevm_circuits_amd.plonk import Advice, Challenge, Fixed, Product, Sub
    a, b, c = Advice(0), Advice(1), Advice(2)
    q, q_acc = Fixed(0), Fixed(1)
    mul = Sub(Product(a, b), c)
    cs = plonk.ConstraintSystem(num_fixed_columns=4, num_advice_columns=4, num_instance_columns=1, num_challenges=1)
    cs.gates = [[Product(q, mul), Product(q, Product(Challenge(0), mul))],
                [Product(q_acc, Sub(Sub(Advice(3, 1), Advice(3)), c))]]
    cs.lookups = [plonk.Lookup([Product(q, a), Product(q, b)], [Fixed(2), Fixed(3)])]
    cs.permutation_columns = [(plonk.ANY_ADVICE, 3), (plonk.ANY_INSTANCE, 0), (plonk.ANY_ADVICE, 0), (plonk.ANY_ADVICE, 2)]
    return cs


def _witness(cs, n, seed):
    """rows < m are active: b = a^2 (lookup into the table of squares), c = a b, s accumulates c; the copy constraints
    tie s[0] and s[m] to the instance column and a few cells of a and c to each other."""
    usable = n - (cs.blinding_factors() + 1)
    m = 20
    rng = np.random.default_rng(seed)
    a = [int(v) for v in rng.integers(0, 11, size=m)] + [0] * (usable - m)
    a[0], a[3] = 2, 8
    a[4] = a[1]
    b = [v * v for v in a]
    c = [v * w for v, w in zip(a, b)]
    s = [5]
    for i in range(m):
        s.append(s[-1] + c[i])
    s += [0] * (usable - len(s))
    fixed = [[1] * m + [0] * (n - m), [1] * m + [0] * (n - m),
             [i if i < usable else 0 for i in range(n)], [i * i if i < usable else 0 for i in range(n)]]
    asm = keygen.Assembly(n, len(cs.permutation_columns))       # argument columns: s, instance, a, c
    asm.copy(0, 0, 1, 0)
    asm.copy(0, m, 1, 1)
    asm.copy(2, 1, 2, 4)
    asm.copy(3, 0, 2, 3)
    return [a, b, c, s], [[5, s[m]]], fixed, asm


@pytest.fixture(scope="module")
def setup(ctx):
    k = 7
    cs = _circuit()
    assert cs.degree() == 5 and cs.num_permutation_sets() == 2
    dom = zk.Domain(ctx, cs.degree(), k)
    g, acc = [], bn.G1_GENERATOR
    for _ in range(dom.n):
        g.append(acc)
        acc = bn.g1_mul(acc, TAU)
    params = prover.ParamsKZG(ctx, dom, affine_arr(g))
    advice, instance, fixed, asm = _witness(cs, dom.n, 7)
    pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
    vk = {"fixed_commitments": pk.fixed_commitments, "permutation_commitments": pk.permutation_commitments,
          "transcript_repr": pk.transcript_repr}
    yield k, cs, dom, params, pk, vk, advice, instance
    for p in pk.pointers():
        ctx.free(p)
    params.close()
    dom.close()


def _prove(ctx, setup, advice, instance, seed=1):
    k, cs, dom, params, pk, vk, _, _ = setup
    return prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(seed), prover.Blake2bWrite())


def test_proof_of_a_satisfied_circuit_verifies(ctx, setup):
    k, cs, dom, params, pk, vk, advice, instance = setup
    proof = _prove(ctx, setup, advice, instance)
    adv_q, fix_q, _ = prover.collect_queries(cs)
    points = cs.num_advice_columns + 3 * len(cs.lookups) + cs.num_permutation_sets() + 1 + (cs.degree() - 1) + 2
    scalars = len(adv_q) + len(fix_q) + 1 + len(cs.permutation_columns) + (3 * cs.num_permutation_sets() - 1) + 5 * len(cs.lookups)
    assert len(proof) == 32 * (points + scalars)
    vm.verify_proof(cs, k, vk, instance, proof, TAU)
    # the same witness and RNG seed give the same bytes; another seed another proof of the same statement
    assert _prove(ctx, setup, advice, instance) == proof
    other = _prove(ctx, setup, advice, instance, seed=2)
    assert other != proof
    vm.verify_proof(cs, k, vk, instance, other, TAU)


def test_verifier_rejects_what_it_must(ctx, setup):
    k, cs, dom, params, pk, vk, advice, instance = setup
    proof = _prove(ctx, setup, advice, instance)
    # another public input
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, [[instance[0][0], instance[0][1] + 1]], proof, TAU)
    # a changed evaluation and a changed commitment
    for pos in (len(proof) - 32 * 10, 5):
        bad = bytearray(proof)
        bad[pos] ^= 1
        with pytest.raises(vm.VerifyError):
            vm.verify_proof(cs, k, vk, instance, bytes(bad), TAU)
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, instance, proof[:-32], TAU)


@pytest.mark.parametrize("what", ["gate", "accumulator", "copy", "instance"])
def test_proof_of_a_broken_witness_does_not_verify(ctx, setup, what):
    """the prover runs to the end on a witness that violates a constraint (as upstream's does): the quotient is then no
    polynomial of the committed degree and the verifier's identity fails"""
    k, cs, dom, params, pk, vk, advice, instance = setup
    advice = [list(col) for col in advice]
    instance = [list(col) for col in instance]
    if what == "gate":
        advice[2][2] += 1                    # c != a b on an active row
    elif what == "accumulator":
        advice[3][7] += 1                    # s[7] != s[6] + c[6]
    elif what == "copy":
        v = 10 if advice[0][1] == 9 else 9   # a[4] != a[1]; gates, lookup and accumulator stay satisfied
        advice[0][4], advice[1][4], advice[2][4] = v, v * v, v * v * v
        for i in range(4, 20):
            advice[3][i + 1] = advice[3][i] + advice[2][i]
        instance[0][1] = advice[3][20]
    else:
        instance[0][0] += 1                  # s[0] no longer equals the public input
    proof = _prove(ctx, setup, advice, instance)
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, instance, proof, TAU)


def test_lookup_input_outside_the_table_is_reported(ctx, setup):
    k, cs, dom, params, pk, vk, advice, instance = setup
    advice = [list(col) for col in advice]
    advice[1][2] += 1                        # (a, b) is no row of the table of squares
    with pytest.raises(ffi.ZkhipError):
        _prove(ctx, setup, advice, instance)
    # the context stays usable
    vm.verify_proof(cs, k, vk, instance, _prove(ctx, setup, setup[6], instance), TAU)


def test_create_proof_argument_checks(ctx, setup):
    k, cs, dom, params, pk, vk, advice, instance = setup
    with pytest.raises(ValueError):
        _prove(ctx, setup, advice[:3], instance)
    with pytest.raises(ValueError):
        _prove(ctx, setup, advice, [[0] * dom.n])
    with pytest.raises(ValueError):
        prover.keygen(ctx, dom, params, cs, [[0] * dom.n], None)
