"""End to end on the GPU: key generation and `create_proof` (every polynomial operation on the device, through the C
ABI) for a small circuit with a satisfied witness — gates with rotations and a challenge, a two-column lookup, copy
constraints over two permutation sets, one instance column — and the proof checked by the big-int verifier model, which
evaluates the identities from the opened values and checks the SHPLONK opening with the test SRS's trapdoor."""
import pytest

import bn254 as bn
import verifier_model as vm
from prover_cases import TAU, circuit as _circuit, srs as _srs, wide_circuit, wide_witness, witness as _witness
from util import golden

from zkevm_circuits_amd import ffi, prover
import zkevm_circuits_amd as zk

pytestmark = pytest.mark.gpu
P = bn.FR_MODULUS


@pytest.fixture(scope="module", params=[7, 13])
def setup(ctx, co, request):
    k = request.param
    cs = _circuit()
    assert cs.degree() == 5 and cs.num_permutation_sets() == 2
    dom = zk.Domain(ctx, cs.degree(), k)
    params = prover.ParamsKZG(ctx, dom, _srs(co, dom.n))
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


def test_prover_reproduces_the_golden_proof(ctx, setup):
    """tests/golden/proof_k7.json (made by tests/golden/gen_proof_golden.py; the CPU suite verifies it against a
    verifying key computed without the device): same circuit, witness and RNG seed give the same bytes"""
    k, cs, dom, params, pk, vk, advice, instance = setup
    gold = golden("proof_k7.json")
    if k != gold["k"]:
        pytest.skip("the fixture is for k = 7")
    assert [[hex(x) for x in p] for p in pk.fixed_commitments] == gold["fixed_commitments"]
    assert [[hex(x) for x in p] for p in pk.permutation_commitments] == gold["permutation_commitments"]
    assert _prove(ctx, setup, advice, instance, seed=gold["rng_seed"]).hex() == gold["proof"]


def test_proving_key_file_round_trip(ctx, setup, tmp_path):
    """the key written as `ProvingKey::write(RawBytes)` and read back (both header layouts) proves to the same bytes, and
    its file has the size the layout predicts"""
    import os
    k, cs, dom, params, pk, vk, advice, instance = setup
    proof = _prove(ctx, setup, advice, instance)
    F, Pc, n, en = cs.num_fixed_columns, len(cs.permutation_columns), dom.n, dom.extended_len
    for layout, head in (("v3", 7), ("legacy", 8)):
        path = str(tmp_path / f"k{k}_{layout}.pk")
        prover.save_proving_key(ctx, dom, pk, path, layout)
        assert os.path.getsize(path) == (head + (F + Pc) * 64 + 3 * (4 + en * 32) + 6 * 4
                                         + (F + Pc) * (2 * (4 + n * 32) + 4 + en * 32))
        pk2 = prover.load_proving_key(ctx, dom, cs, path)
        assert pk2.fixed_commitments == pk.fixed_commitments and pk2.transcript_repr == pk.transcript_repr
        got = prover.create_proof(ctx, dom, params, pk2, advice, instance, prover.FieldRng(1), prover.Blake2bWrite())
        assert got == proof
        for p in pk2.pointers():
            ctx.free(p)
    with pytest.raises(ValueError):
        other = zk.Domain(ctx, cs.degree(), k + 1)
        try:
            prover.load_proving_key(ctx, other, cs, path)
        finally:
            other.close()


def test_bounded_footprint_gives_the_same_proof(ctx, setup):
    """the quotient numerator evaluated coset group by coset group (what a GPU with less free memory than all extended
    cosets need does) gives the same bytes as all cosets at once; the OS-entropy RNG gives a verifying proof"""
    k, cs, dom, params, pk, vk, advice, instance = setup
    proof = _prove(ctx, setup, advice, instance)
    # coset groups need a key whose extended vectors are in extended order; a key with the vectors split by row parity
    # (low-degree terms on half the domain) holds every coset at once and says so.  The bytes are the same either way.
    _, _, fixed, asm = _witness(cs, dom.n, 7)
    keys = {flag: prover.keygen(ctx, dom, params, cs, fixed, asm, low_degree_split=flag) for flag in (False, True)}
    assert keys[True].polys.split and not keys[False].polys.split
    assert prover.create_proof(ctx, dom, params, keys[True], advice, instance, prover.FieldRng(1), prover.Blake2bWrite()) == proof
    assert prover.create_proof(ctx, dom, params, keys[False], advice, instance, prover.FieldRng(1), prover.Blake2bWrite()) == proof
    for groups in (2, 4):
        got = prover.create_proof(ctx, dom, params, keys[False], advice, instance, prover.FieldRng(1), prover.Blake2bWrite(),
                                  coset_groups=groups)
        assert got == proof, groups
    with pytest.raises(ValueError):
        prover.create_proof(ctx, dom, params, keys[False], advice, instance, prover.FieldRng(1), prover.Blake2bWrite(), coset_groups=3)
    with pytest.raises(ValueError):
        prover.create_proof(ctx, dom, params, keys[True], advice, instance, prover.FieldRng(1), prover.Blake2bWrite(), coset_groups=2)
    for key in keys.values():
        for p in key.pointers():
            ctx.free(p)
    fresh = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.OsFieldRng(), prover.Blake2bWrite())
    assert fresh != proof
    vm.verify_proof(cs, k, vk, instance, fresh, TAU)


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
        m = min(len(advice[0]) - 2, 3000)    # active rows, as _witness chose them
        for i in range(4, m):
            advice[3][i + 1] = advice[3][i] + advice[2][i]
        instance[0][1] = advice[3][m]
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


def test_full_size_proof_verifies(ctx, co):
    """k = 19, the reference's circuit size (/root/reference/zkevm-circuits/src/constant.rs:10): SRS of 2^19 points
    with its Lagrange form from the device's G1 transform, key generation, create_proof over 2^19 rows and the 2^21
    extended domain, and the verifier model on the resulting bytes"""
    import time
    k = 19
    cs = _circuit()
    dom = zk.Domain(ctx, cs.degree(), k)
    t0 = time.perf_counter()
    g = _srs(co, dom.n)
    t1 = time.perf_counter()
    params = prover.ParamsKZG(ctx, dom, g)
    t2 = time.perf_counter()
    advice, instance, fixed, asm = _witness(cs, dom.n, 19)
    pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
    t3 = time.perf_counter()
    proof = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(19), prover.Blake2bWrite())
    t4 = time.perf_counter()
    vk = {"fixed_commitments": pk.fixed_commitments, "permutation_commitments": pk.permutation_commitments,
          "transcript_repr": pk.transcript_repr}
    vm.verify_proof(cs, k, vk, instance, proof, TAU)
    print(f"k=19: srs {t1 - t0:.1f} s (CPU), params {t2 - t1:.1f} s, keygen {t3 - t2:.1f} s, create_proof {t4 - t3:.1f} s "
          f"(host conversions included), verify {time.perf_counter() - t4:.1f} s")
    bad = bytearray(proof)
    bad[40] ^= 1
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, instance, bytes(bad), TAU)
    for p in pk.pointers():
        ctx.free(p)
    params.close()
    dom.close()


@pytest.mark.parametrize("k,units,degree9", [(9, 3, False), (10, 16, True)])
def test_wide_circuit_proof_verifies(ctx, co, k, units, degree9):
    """many columns (80 advice, 16 lookups, 49 permutation columns in seven sets at 16 units), a degree-nine gate (extended domain 8n), rotations
    -1 / 0 / +1: the fused evaluate_h program against the verifier's expression-by-expression evaluation"""
    cs = wide_circuit(units, degree9)
    assert cs.degree() == (9 if degree9 else 5)
    dom = zk.Domain(ctx, cs.degree(), k)
    assert dom.extended_k == k + (3 if degree9 else 2)
    params = prover.ParamsKZG(ctx, dom, _srs(co, dom.n))
    advice, instance, fixed, asm = wide_witness(cs, dom.n, k, units)
    pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
    vk = {"fixed_commitments": pk.fixed_commitments, "permutation_commitments": pk.permutation_commitments,
          "transcript_repr": pk.transcript_repr}
    proof = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite())
    vm.verify_proof(cs, k, vk, instance, proof, TAU)
    # the witness handed over as Montgomery limbs in page-locked memory: the columns reach the device on the transfer
    # stream, group after group beside the commitments of the group before (80 columns = three groups at 16 units), and
    # the proof has the same bytes
    pinned = []
    for col in advice:
        a = ctx.host_zeros((len(col), 4))
        a[:] = prover.fr_limbs(col)
        pinned.append(a)
    assert prover.create_proof(ctx, dom, params, pk, pinned, instance, prover.FieldRng(k), prover.Blake2bWrite()) == proof
    # the witness already in device memory (uploaded by the caller while it was produced): the same bytes again
    resident = [prover.DeviceColumn(ctx.upload(a), len(a)) for a in pinned]
    assert prover.create_proof(ctx, dom, params, pk, resident, instance, prover.FieldRng(k), prover.Blake2bWrite()) == proof
    with pytest.raises(ValueError):
        prover.create_proof(ctx, dom, params, pk, [prover.DeviceColumn(resident[0].ptr, 5)] + resident[1:], instance,
                            prover.FieldRng(k), prover.Blake2bWrite())
    for col in resident:
        ctx.free(col.ptr)
    for a in pinned:
        ctx.host_free(a)
    # one wrong cell in the last unit: d != a(-1) + b on one row
    advice = [list(col) for col in advice]
    advice[5 * (units - 1) + 4][9] += 1
    proof = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite())
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, instance, proof, TAU)
    for p in pk.pointers():
        ctx.free(p)
    params.close()
    dom.close()


def test_columns_that_only_low_degree_terms_read_are_extended_to_half_the_domain(ctx, co):
    """degree nine, eight cosets: the lookup terms have degree <= 4, so no term above the bound names a lookup polynomial
    and the split key extends those onto the even-indexed rows only; the bytes equal those of a key without the split"""
    from zkevm_circuits_amd import synthetic
    k, units = 9, 4
    cs = synthetic.satisfied_cs(units, degree9=True, plain_lookups=True)
    dom = zk.Domain(ctx, cs.degree(), k)
    params = prover.ParamsKZG(ctx, dom, _srs(co, dom.n))
    advice, instance, fixed, asm = wide_witness(cs, dom.n, k, units)
    keys = {flag: prover.keygen(ctx, dom, params, cs, fixed, asm, low_degree_split=flag) for flag in (False, True)}
    used = keys[True].high_degree_columns
    assert keys[False].high_degree_columns is None
    assert len(used["advice"]) == cs.num_advice_columns and len(used["lookups"]) == len(cs.lookups)
    assert not any(f for t in used["lookups"] for f in t)        # every lookup polynomial: even-indexed rows only
    assert any(used["advice"]) and any(used["perm_z"])           # the degree-nine gate and the permutation products: all rows
    proofs = {flag: prover.create_proof(ctx, dom, params, key, advice, instance, prover.FieldRng(3), prover.Blake2bWrite())
              for flag, key in keys.items()}
    assert proofs[True] == proofs[False]
    vk = {"fixed_commitments": keys[True].fixed_commitments, "permutation_commitments": keys[True].permutation_commitments,
          "transcript_repr": keys[True].transcript_repr}
    vm.verify_proof(cs, k, vk, instance, proofs[True], TAU)
    for key in keys.values():
        for p in key.pointers():
            ctx.free(p)
    params.close()
    dom.close()


@pytest.mark.parametrize("units,extra_gates", [(16, 0), (76, 0), (76, 40)])
def test_full_size_wide_proof_verifies(ctx, co, units, extra_gates):
    """k = 19 with 16 units: 80 advice columns, 16 lookups, 49 permutation columns, a degree-nine gate, i.e. 146
    coset extensions to 2^22, evaluate_h over 2^22 rows, eight quotient pieces.  With 76 units the column counts of the
    reference's SuperCircuit estimate: 380 advice columns, 76 lookups, 229 permutation columns in 33 sets; with 40 extra
    gate polynomials per unit also its estimated 3000 gate polynomials of degree up to nine."""
    import time
    k = 19
    cs = wide_circuit(units, True, extra_gates)
    assert cs.blinding_factors() == 5
    dom = zk.Domain(ctx, cs.degree(), k)
    assert dom.extended_k == 22
    params = prover.ParamsKZG(ctx, dom, _srs(co, dom.n))
    advice, instance, fixed, asm = wide_witness(cs, dom.n, k, units, active_rows=3000, as_arrays=True)
    t0 = time.perf_counter()
    pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
    t1 = time.perf_counter()
    phases = {}
    proof = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite(), phases)
    t2 = time.perf_counter()
    vk = {"fixed_commitments": pk.fixed_commitments, "permutation_commitments": pk.permutation_commitments,
          "transcript_repr": pk.transcript_repr}
    vm.verify_proof(cs, k, vk, instance, proof, TAU)
    print(f"k=19 wide, {units} units, {sum(len(g) for g in cs.gates)} gate polynomials: keygen {t1 - t0:.1f} s, create_proof {t2 - t1:.1f} s (host conversions included), "
          f"verify {time.perf_counter() - t2:.1f} s, proof {len(proof)} bytes; phases "
          + ", ".join(f"{name} {sec:.2f}" for name, sec in phases.items()))
    assert set(phases) == {"advice", "lookup_permuted", "permutation_product", "lookup_product", "random_poly", "quotient",
                           "quotient_evaluate_h", "quotient_pieces", "evaluations", "multiopen"}
    for p in pk.pointers():
        ctx.free(p)
    params.close()
    dom.close()


@pytest.mark.parametrize("with_lookup,with_copy", [(False, False), (True, False), (False, True)])
def test_minimal_circuits(ctx, co, with_lookup, with_copy):
    """no lookups, no permutation argument, no instance column, no challenge: every optional part of the proof absent in
    turn (quotient degree 2 without a lookup: extended domain 2n)"""
    from zkevm_circuits_amd import keygen, plonk
    from zkevm_circuits_amd.plonk import Advice, Fixed, Product, Sub
    k = 6
    cs = plonk.ConstraintSystem(num_fixed_columns=2, num_advice_columns=2)
    cs.gates = [[Product(Fixed(0), Sub(Product(Advice(0), Advice(0)), Advice(1)))]]
    if with_lookup:
        cs.lookups = [plonk.Lookup([Advice(0)], [Fixed(1)])]
    if with_copy:
        cs.permutation_columns = [(plonk.ANY_ADVICE, 0), (plonk.ANY_ADVICE, 1)]
    dom = zk.Domain(ctx, cs.degree(), k)
    assert dom.extended_k == k + (2 if with_lookup else 1)
    n = dom.n
    usable = n - (cs.blinding_factors() + 1)
    a = [(3 * i) % 7 for i in range(usable)]
    a[5], a[9] = 4, 16                                   # b[5] = a[5]^2 = a[9]
    b = [v * v for v in a]
    fixed = [[1] * usable + [0] * (n - usable), [i if i < usable else 0 for i in range(n)]]
    asm = None
    if with_copy:
        asm = keygen.Assembly(n, 2)
        asm.copy(1, 5, 0, 9)
    params = prover.ParamsKZG(ctx, dom, _srs(co, n))
    pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
    vk = {"fixed_commitments": pk.fixed_commitments, "permutation_commitments": pk.permutation_commitments,
          "transcript_repr": pk.transcript_repr}
    proof = prover.create_proof(ctx, dom, params, pk, [a, b], [], prover.FieldRng(3), prover.Blake2bWrite())
    vm.verify_proof(cs, k, vk, [], proof, TAU)
    b[11] += 1
    proof = prover.create_proof(ctx, dom, params, pk, [a, b], [], prover.FieldRng(3), prover.Blake2bWrite())
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, [], proof, TAU)
    for p in pk.pointers():
        ctx.free(p)
    params.close()
    dom.close()


def test_two_phase_circuit(ctx, co):
    """a second-phase advice column that depends on a first-phase challenge (r = c0 a, the shape of the reference's
    challenge-compressed columns, /root/reference/zkevm-circuits/src/core_circuit.rs:153-154 and src/util.rs:63-67): the
    witness callable runs once per phase, the commitments and challenges interleave per phase in the transcript"""
    from zkevm_circuits_amd import keygen, plonk
    from zkevm_circuits_amd.plonk import Advice, Challenge, Fixed, Product, Sub
    k = 6
    cs = plonk.ConstraintSystem(num_fixed_columns=1, num_advice_columns=3, num_challenges=2,
                                advice_column_phase=[0, 1, 0], challenge_phase=[0, 1])
    rel = Sub(Advice(1), Product(Challenge(0), Advice(0)))
    cs.gates = [[Product(Fixed(0), rel), Product(Fixed(0), Product(Challenge(1), Product(rel, Advice(2, 1))))]]
    cs.permutation_columns = [(plonk.ANY_ADVICE, 1), (plonk.ANY_ADVICE, 0)]
    assert [ph for ph, _, _ in cs.phases()] == [0, 1]
    dom = zk.Domain(ctx, cs.degree(), k)
    n = dom.n
    usable = n - (cs.blinding_factors() + 1)
    a = [(5 * i + 1) % 11 for i in range(usable)]
    a[3] = a[2]
    other = [i * i for i in range(usable)]
    calls = []

    def witness(phase, challenges):
        calls.append((phase, list(challenges)))
        r = [challenges[0] * v % P for v in a] if phase == 1 else [0] * usable      # unknown in the first phase
        return [a, r, other]
    asm = keygen.Assembly(n, 2)
    asm.copy(0, 2, 0, 3)                                 # r[2] = r[3], true because a[2] = a[3]
    params = prover.ParamsKZG(ctx, dom, _srs(co, n))
    pk = prover.keygen(ctx, dom, params, cs, [[1] * usable + [0] * (n - usable)], asm)
    vk = {"fixed_commitments": pk.fixed_commitments, "permutation_commitments": pk.permutation_commitments,
          "transcript_repr": pk.transcript_repr}
    proof = prover.create_proof(ctx, dom, params, pk, witness, [], prover.FieldRng(3), prover.Blake2bWrite())
    assert [c[0] for c in calls] == [0, 1] and calls[0][1] == [0, 0] and calls[1][1][0] != 0 and calls[1][1][1] == 0
    vm.verify_proof(cs, k, vk, [], proof, TAU)
    # a second-phase column computed from anything but the real challenge fails
    proof = prover.create_proof(ctx, dom, params, pk, lambda ph, ch: [a, [7 * v for v in a], other], [], prover.FieldRng(3),
                                prover.Blake2bWrite())
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk, [], proof, TAU)
    for p in pk.pointers():
        ctx.free(p)
    params.close()
    dom.close()


def test_upstream_random_polynomial_and_query_order_modes(ctx, co):
    """the restated byte-level rules: the random polynomial drawn from per-thread ChaCha20 streams (it depends on the thread
    count, and the proof verifies for each), and a circuit's own query order (the evaluations move inside the proof, the
    verifying key's digest changes with it, and the verifier reading the same lists accepts)"""
    k = 7
    cs = _circuit()
    dom = zk.Domain(ctx, cs.degree(), k)
    params = prover.ParamsKZG(ctx, dom, _srs(co, dom.n))
    advice, instance, fixed, asm = _witness(cs, dom.n, 7)
    pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
    vk = {"fixed_commitments": pk.fixed_commitments, "permutation_commitments": pk.permutation_commitments,
          "transcript_repr": pk.transcript_repr}
    proofs = {}
    for threads in (None, 1, 3, 16):
        proofs[threads] = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(1), prover.Blake2bWrite(),
                                              random_poly_threads=threads)
        vm.verify_proof(cs, k, vk, instance, proofs[threads], TAU)
    assert len(set(proofs.values())) == 4
    # the circuit's own order of first queries: the derived order reversed
    adv_q, fix_q, inst_q = prover.collect_queries(cs)
    cs2 = _circuit()
    cs2.advice_queries, cs2.fixed_queries = list(reversed(adv_q)), list(reversed(fix_q))
    pk2 = prover.keygen(ctx, dom, params, cs2, fixed, asm)
    assert pk2.fixed_commitments == pk.fixed_commitments and pk2.transcript_repr != pk.transcript_repr
    vk2 = dict(vk, transcript_repr=pk2.transcript_repr)
    proof2 = prover.create_proof(ctx, dom, params, pk2, advice, instance, prover.FieldRng(1), prover.Blake2bWrite())
    assert proof2 != proofs[None] and len(proof2) == len(proofs[None])
    vm.verify_proof(cs2, k, vk2, instance, proof2, TAU)
    with pytest.raises(vm.VerifyError):
        vm.verify_proof(cs, k, vk2, instance, proof2, TAU)          # read in the derived order, the evaluations do not fit
    for key in (pk, pk2):
        for p in key.pointers():
            ctx.free(p)
    params.close()
    dom.close()


def test_workspace_kept_between_proofs_gives_the_same_bytes(ctx, setup):
    """`ProofWorkspace`: the large blocks outlive the call and serve the next proof (also one with another witness);
    the bytes equal those of a call that allocates and frees"""
    k, cs, dom, params, pk, vk, advice, instance = setup
    proof = _prove(ctx, setup, advice, instance)
    ws = prover.ProofWorkspace(ctx)
    first = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(1), prover.Blake2bWrite(), workspace=ws)
    held = dict(ws.blocks)
    assert held and first == proof
    other = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(2), prover.Blake2bWrite(), workspace=ws)
    assert other != proof and ws.blocks == held
    vm.verify_proof(cs, k, vk, instance, other, TAU)
    again = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(1), prover.Blake2bWrite(), workspace=ws)
    assert again == proof
    ws.close()
    assert not ws.blocks
