"""CPU suite: the on-disk formats around the proving path (parameter file, polynomial records, proof JSON).

The proof JSON and instance encoding are checked against literal bytes derived from the reference's own description of
them (prover/src/proof.rs); the parameter / polynomial layouts are round-trip and byte-layout checks only (unpinned: the
halo2 fork that defines them is not part of the reference tree)."""
import base64
import io
import json
import random
import struct

import numpy as np
import pytest

import bn254 as bn
from zkevm_circuits_amd import formats

P = bn.FR_MODULUS


def _mont_rows(vals):
    return np.array([[((v << 256) % P >> (64 * i)) & (2**64 - 1) for i in range(4)] for v in vals], dtype="<u8")


def test_instances_big_endian_canonical():
    rng = random.Random(5)
    vals = [0, 1, P - 1, 0x0102030405060708090A0B0C0D0E0F10] + [rng.randrange(P) for _ in range(6)]
    data = formats.serialize_instances([_mont_rows(vals)])
    assert len(data) == 32 * len(vals)
    # value 1 -> 31 zero bytes then 0x01; the fourth value ends ... 0f 10
    assert data[32:64] == b"\x00" * 31 + b"\x01"
    assert data[96:128][-2:] == b"\x0f\x10"
    for i, v in enumerate(vals):
        assert int.from_bytes(data[32 * i:32 * i + 32], "big") == v
    back = formats.deserialize_instances(data)
    assert len(back) == 1 and np.array_equal(back[0], _mont_rows(vals))
    with pytest.raises(formats.FormatError):
        formats.deserialize_instances(P.to_bytes(32, "big"))
    with pytest.raises(formats.FormatError):
        formats.deserialize_instances(b"\x00" * 33)
    with pytest.raises(formats.FormatError):
        formats.serialize_instances([_mont_rows([1]), _mont_rows([2])])


def test_proof_json_fields(tmp_path):
    inst = [_mont_rows([7, P - 2])]
    proof = bytes(range(200))
    path = formats.dump_proof_json(str(tmp_path), "super", proof, inst, vk=b"\x01\x02")
    assert path.endswith("/full_proof_super.json")
    doc = json.load(open(path))
    assert sorted(doc) == ["instances", "proof", "vk"]
    assert base64.b64decode(doc["proof"]) == proof
    assert base64.b64decode(doc["instances"]) == (7).to_bytes(32, "big") + (P - 2).to_bytes(32, "big")
    p2, i2, vk2 = formats.load_proof_json(str(tmp_path), "super")
    assert p2 == proof and vk2 == b"\x01\x02" and np.array_equal(i2[0], inst[0])
    with pytest.raises(formats.FormatError):
        formats.load_proof_json(str(tmp_path), "absent")
    # chunk proof: the protocol bytes lead, the proof fields sit flattened beside them
    formats.dump_proof_json(str(tmp_path), "chunk", proof, inst, vk=b"", protocol=b'{"domain":{}}')
    raw = open(formats.proof_path(str(tmp_path), "chunk")).read()
    assert raw.startswith('{"protocol":"') and list(json.loads(raw)) == ["protocol", "proof", "instances", "vk"]
    pr, p3, i3, vk3 = formats.load_proof_json(str(tmp_path), "chunk", with_protocol=True)
    assert pr == b'{"domain":{}}' and p3 == proof and vk3 == b"" and np.array_equal(i3[0], inst[0])
    with pytest.raises(formats.FormatError):
        formats.load_proof_json(str(tmp_path), "super", with_protocol=True)     # a plain proof has no protocol


def test_params_raw_layout_and_round_trip(tmp_path):
    k = 5
    rng = np.random.default_rng(3)
    g = rng.integers(0, 2**63, size=(1 << k, 8), dtype=np.uint64)
    gl = rng.integers(0, 2**63, size=(1 << k, 8), dtype=np.uint64)
    g2, s_g2 = bytes(range(128)), bytes(reversed(range(128)))
    path = str(tmp_path / "k5.params")
    formats.write_params_raw(path, formats.ParamsRaw(k, g, gl, g2, s_g2))
    raw = open(path, "rb").read()
    assert len(raw) == 4 + 2 * 32 * 64 + 256
    assert raw[:4] == struct.pack("<I", k)
    assert raw[4:12] == int(g[0, 0]).to_bytes(8, "little")                 # x limb 0 of g[0]
    assert raw[4 + 32 * 64:][:8] == int(gl[0, 0]).to_bytes(8, "little")    # g_lagrange follows g
    assert raw[-256:-128] == g2 and raw[-128:] == s_g2
    p = formats.read_params_raw(path)
    assert p.k == k and p.n == 32 and p.nbytes() == len(raw)
    assert np.array_equal(p.g, g) and np.array_equal(p.g_lagrange, gl) and p.g2 == g2 and p.s_g2 == s_g2
    open(path, "ab").write(b"\x00")
    with pytest.raises(formats.FormatError):
        formats.read_params_raw(path)
    with pytest.raises(formats.FormatError):
        formats.ParamsRaw(k, g[:3], gl, g2, s_g2)


def test_polynomial_raw_records():
    rng = random.Random(9)
    a = _mont_rows([rng.randrange(P) for _ in range(17)])
    b = _mont_rows([])
    f = io.BytesIO()
    formats.write_polynomial_raw(f, a)
    formats.write_polynomial_raw(f, b)
    raw = f.getvalue()
    assert raw[:4] == struct.pack(">I", 17) and len(raw) == 4 + 17 * 32 + 4
    assert raw[4:36] == a[0].tobytes()
    f.seek(0)
    assert np.array_equal(formats.read_polynomial_raw(f), a)
    assert formats.read_polynomial_raw(f).shape == (0, 4)
    with pytest.raises(formats.FormatError):
        formats.read_polynomial_raw(f)                         # end of file
    with pytest.raises(formats.FormatError):
        formats.read_polynomial_raw(io.BytesIO(struct.pack(">I", 2) + b"\x00" * 32))
    bad = np.array([[(P >> (64 * i)) & (2**64 - 1) for i in range(4)]], dtype="<u8")   # the modulus itself
    g = io.BytesIO()
    formats.write_polynomial_raw(g, bad)
    g.seek(0)
    with pytest.raises(formats.FormatError):
        formats.read_polynomial_raw(g)


def _points(rng, count):
    """affine points as raw Montgomery limbs (multiples of the generator by the big-int model)"""
    q = bn.FQ_MODULUS
    pts = [bn.g1_mul(bn.G1_GENERATOR, rng.randrange(1, 1 << 40)) for _ in range(count)]
    return np.array([[((c << 256) % q >> (64 * i)) & (2**64 - 1) for c in pt for i in range(4)] for pt in pts],
                    dtype="<u8").reshape(count, 8)


@pytest.mark.parametrize("layout", ["v3", "legacy"])
def test_verifying_and_proving_key_raw_bytes(tmp_path, layout):
    """`VerifyingKey::write` / `ProvingKey::write` in RawBytes (unpinned: restated layouts of the absent halo2 fork): byte
    layout of the header, packed selector bits, round trip, the reader telling the two header layouts apart, and the
    errors (truncation, trailing bytes, wrong column count, a coordinate that is no residue)"""
    rng = random.Random(11)
    k, ek = 4, 6
    n, en = 1 << k, 1 << ek
    sel = [np.array([rng.random() < 0.5 for _ in range(n)]) for _ in range(2)]
    vk = formats.VerifyingKeyRaw(k, _points(rng, 3), _points(rng, 2), sel, True)
    f = io.BytesIO()
    formats.write_verifying_key_raw(f, vk, layout)
    raw = f.getvalue()
    head = 7 if layout == "v3" else 8
    assert raw[:head] == (bytes([3, k, 1]) + struct.pack("<I", 3) if layout == "v3" else struct.pack(">II", k, 3))
    assert len(raw) == head + 5 * 64 + 2 * (n // 8)
    assert raw[head:head + 64] == vk.fixed_commitments[0].tobytes()
    assert raw[-2] == sum(int(b) << i for i, b in enumerate(sel[1][:8]))          # least significant bit first
    back = formats.read_verifying_key_raw(io.BytesIO(raw), 2, 2)
    assert back.k == k and np.array_equal(back.fixed_commitments, vk.fixed_commitments)
    assert np.array_equal(back.permutation_commitments, vk.permutation_commitments)
    assert all(np.array_equal(a, b) for a, b in zip(back.selectors, sel)) and back.compress_selectors
    with pytest.raises(formats.FormatError):
        formats.read_verifying_key_raw(io.BytesIO(raw[:-1]), 2, 2)
    with pytest.raises(formats.FormatError):
        formats.read_verifying_key_raw(io.BytesIO(raw), 9, 2)                    # more permutation columns than stored
    with pytest.raises(formats.FormatError):
        formats.read_verifying_key_raw(io.BytesIO(b"\x07" + raw[1:]), 2, 2)      # unknown version byte
    bad = bytearray(raw)
    bad[head:head + 32] = b"\xff" * 32                                            # x coordinate beyond the modulus
    with pytest.raises(formats.FormatError):
        formats.read_verifying_key_raw(io.BytesIO(bytes(bad)), 2, 2)

    poly = lambda length: _mont_rows([rng.randrange(P) for _ in range(length)])
    pk = formats.ProvingKeyRaw(vk, l0=poly(en), l_last=poly(en), l_active_row=poly(en),
                               fixed_values=[poly(n) for _ in range(3)], fixed_polys=[poly(n) for _ in range(3)],
                               fixed_cosets=[poly(en) for _ in range(3)], permutations=[poly(n) for _ in range(2)],
                               permutation_polys=[poly(n) for _ in range(2)], permutation_cosets=[poly(en) for _ in range(2)])
    pk.check(ek)
    path = str(tmp_path / "k4.pk")
    formats.write_proving_key_raw(path, pk, layout)
    size = len(raw) + 3 * (4 + en * 32) + 4 * 6 + 3 * 2 * (4 + n * 32) + 3 * (4 + en * 32) + 2 * 2 * (4 + n * 32) + 2 * (4 + en * 32)
    assert (tmp_path / "k4.pk").stat().st_size == size
    got = formats.read_proving_key_raw(path, 2, 2)
    got.check(ek)
    for name in formats.ProvingKeyRaw.SINGLE:
        assert np.array_equal(getattr(got, name), getattr(pk, name))
    for name in formats.ProvingKeyRaw.SLICES:
        assert all(np.array_equal(a, b) for a, b in zip(getattr(got, name), getattr(pk, name)))
    with pytest.raises(formats.FormatError):
        got.check(ek + 1)
    with open(path, "ab") as g:
        g.write(b"\x00")
    with pytest.raises(formats.FormatError):
        formats.read_proving_key_raw(path, 2, 2)


def _scrape(lines, tag):
    """What scripts/report_system_data/report_bench.py:57-75 does with an End line: collapse runs of equal characters,
    keep what follows the first dot."""
    import itertools
    line = [i for i in lines if "End" in i and tag in i][0]
    return "".join(g[0] for g in itertools.groupby(line)).split(".", 1)[-1].strip()


def test_reference_log_lines_are_scrapable():
    lines = formats.reference_log_lines("replay_k19", 280000, 19, {"create_proof": 2.7203, "verify_proof": 0.0125,
                                                                     "generate_witness": 1.5})
    # the summary line as report_bench.py:29-34 splits it
    fields = [i for i in lines if formats.CIRCUIT_SUMMARY in i][0].split(",")
    got = {}
    for spl in fields:
        for key in ("max_num_row", "degree"):
            if key in spl:
                got[key] = spl.strip().removeprefix(key + ":")
    assert got == {"max_num_row": "280000", "degree": "19"}
    assert [i for i in lines if "running" in i and "test" in i][0].split()[1] == "1"
    assert [i.split(":")[1].split(".")[0].replace(" ", "") for i in lines if "test result" in i][0] == "ok"
    # 2.7203 s prints as 2.720s; the scraper's run-collapsing leaves 2.720s intact, and 12.500ms becomes 12.50ms
    assert _scrape(lines, "[Create_proof]") == "2.720s"
    assert _scrape(lines, "[Verify proof]") == "12.50ms"
    assert [i for i in lines if "End" in i and "[Generate witness]" in i][0].split(" ")[-1].endswith("1.500s")
    order = [i.split("]")[0] for i in lines if i.startswith("Start:")]
    assert [o.split("[")[1] for o in order] == ["Generate witness", "Create_proof", "Verify proof"]
    assert formats._ark_time(3.2e-6) == "3.200µs" and formats._ark_time(5e-9) == "5ns"
    with pytest.raises(formats.FormatError):
        formats.reference_log_lines("x", 1, 1, {"setup": 1.0})
