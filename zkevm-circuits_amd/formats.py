"""On-disk formats either side of the proving path: structured-reference-string files, polynomial files and the proof JSON.

Host-side I/O only.  In `SerdeFormat::RawBytes` a field element is its four Montgomery limbs, little-endian, and an affine
G1 point is x then y (64 bytes) -- the very layout the device kernels read, so a parameter file is memory-mapped and its
point tables go to `Context.register_bases` without a conversion pass.

Provenance:
  * the proof JSON and the instance encoding follow the reference's own files (prover/src/proof.rs:24-44 for the three
    base64 fields, :129-150 for big-endian canonical instances, :94-96 for the file name);
  * the parameter and polynomial layouts belong to the halo2 fork the reference depends on (`ParamsKZG::write_custom`,
    `Polynomial::write`; call sites benches/benchmark_list/main.rs:441-461) whose sources are not part of the reference
    tree: they are restated from the published halo2 v0.3.0 layout and are *unpinned* -- no file produced by the
    reference is available to check them against.
"""
from __future__ import annotations

import base64
import json
import os
import struct

import numpy as np

from .plonk import FR_MODULUS

FQ_MODULUS = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
_R_INV = pow(1 << 256, -1, FR_MODULUS)
G1_RAW = 64     # x | y, four u64 limbs each
G2_RAW = 128    # x.c0 | x.c1 | y.c0 | y.c1
FR_RAW = 32


class FormatError(ValueError):
    pass


# ---- ParamsKZG, SerdeFormat::RawBytes ---------------------------------------------------------------------------------
class ParamsRaw:
    """k, g[0..2^k], g_lagrange[0..2^k] as (2^k, 8) uint64 arrays of Montgomery limbs, g2 and s_g2 as 128 raw bytes each."""

    def __init__(self, k: int, g, g_lagrange, g2: bytes, s_g2: bytes):
        n = 1 << k
        g = np.asarray(g, dtype="<u8").reshape(-1, 8)
        g_lagrange = np.asarray(g_lagrange, dtype="<u8").reshape(-1, 8)
        if g.shape[0] != n or g_lagrange.shape[0] != n:
            raise FormatError(f"params: 2^{k} points expected, got {g.shape[0]} and {g_lagrange.shape[0]}")
        if len(g2) != G2_RAW or len(s_g2) != G2_RAW:
            raise FormatError("params: g2 / s_g2 must be 128 raw bytes")
        self.k, self.g, self.g_lagrange, self.g2, self.s_g2 = k, g, g_lagrange, bytes(g2), bytes(s_g2)

    @property
    def n(self) -> int:
        return 1 << self.k

    def nbytes(self) -> int:
        return 4 + 2 * self.n * G1_RAW + 2 * G2_RAW


def write_params_raw(path: str, p: ParamsRaw) -> None:
    """k as u32 little-endian, g, g_lagrange, g2, s_g2 (benches/benchmark_list/main.rs:441-446 writes this file)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<I", p.k))
        f.write(np.ascontiguousarray(p.g, dtype="<u8").tobytes())
        f.write(np.ascontiguousarray(p.g_lagrange, dtype="<u8").tobytes())
        f.write(p.g2)
        f.write(p.s_g2)


def read_params_raw(path: str, max_k: int = 28) -> ParamsRaw:
    """Memory-maps the two point tables (a k = 19 file is 64 MiB, a k = 26 one 8 GiB): nothing is copied until uploaded."""
    size = os.path.getsize(path)
    if size < 4:
        raise FormatError("params: file shorter than its header")
    with open(path, "rb") as f:
        (k,) = struct.unpack("<I", f.read(4))
    if k > max_k:
        raise FormatError(f"params: k = {k} is beyond {max_k}")
    n = 1 << k
    want = 4 + 2 * n * G1_RAW + 2 * G2_RAW
    if size != want:
        raise FormatError(f"params: k = {k} needs {want} bytes, file has {size}")
    raw = np.memmap(path, dtype=np.uint8, mode="r", offset=4, shape=(2 * n * G1_RAW,))
    g = _limb_view(raw[: n * G1_RAW], 8)
    gl = _limb_view(raw[n * G1_RAW:], 8)
    with open(path, "rb") as f:
        f.seek(4 + 2 * n * G1_RAW)
        g2 = f.read(G2_RAW)
        s_g2 = f.read(G2_RAW)
    return ParamsRaw(k, g, gl, g2, s_g2)


def _limb_view(raw_u8: np.ndarray, limbs: int) -> np.ndarray:
    """(rows, limbs) little-endian u64 view of a byte array, never a copy.  The tables of a parameter file start at byte 4,
    so the view is not 8-byte aligned: numpy reads such arrays correctly, and the library only ever copies them to the
    device."""
    return raw_u8.view("<u8").reshape(-1, limbs)


# ---- Polynomial<F, B>, SerdeFormat::RawBytes --------------------------------------------------------------------------
def write_polynomial_raw(f, values) -> None:
    """u32 big-endian length, then the elements as raw Montgomery limbs (the pieces of a proving-key file)."""
    a = np.ascontiguousarray(values, dtype="<u8").reshape(-1, 4)
    if a.shape[0] >= 1 << 32:
        raise FormatError("polynomial: length does not fit u32")
    f.write(struct.pack(">I", a.shape[0]))
    f.write(a.tobytes())


def read_polynomial_raw(f, max_len: int = 1 << 30) -> np.ndarray:
    head = f.read(4)
    if len(head) != 4:
        raise FormatError("polynomial: truncated length")
    (n,) = struct.unpack(">I", head)
    if n > max_len:
        raise FormatError(f"polynomial: length {n} is beyond {max_len}")
    body = f.read(n * FR_RAW)
    if len(body) != n * FR_RAW:
        raise FormatError("polynomial: truncated body")
    a = np.frombuffer(body, dtype="<u8").reshape(n, 4)
    # a raw element must still be a reduced residue: the device arithmetic assumes limbs below the modulus
    if n and not _all_below_modulus(a):
        raise FormatError("polynomial: element not below the field modulus")
    return a


def _all_below_modulus(a: np.ndarray, modulus: int = FR_MODULUS) -> bool:
    p = [(modulus >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        limb = a[:, i]
        lt |= eq & (limb < np.uint64(p[i]))
        eq &= limb == np.uint64(p[i])
    return bool(lt.all())


# ---- VerifyingKey / ProvingKey, SerdeFormat::RawBytes -----------------------------------------------------------------
# The key files the reference's harness caches next to the parameters (benches/benchmark_list/main.rs:448-461 write them,
# :463-501 read them back; prover/src/util.rs:48-57 for the verifying key alone).  The byte layouts belong to the halo2 fork,
# whose source is absent: both layouts that releases of halo2_proofs 0.3.0 have used are implemented, the reader tells them
# apart by their first byte, and both are *unpinned* (no key file written by the reference exists to check them against):
#   "v3"     version byte 0x03, k as one byte, compress_selectors as one byte, number of fixed commitments u32 LE
#   "legacy" k u32 BE, number of fixed commitments u32 BE
# then, in both: the fixed commitments, the permutation commitments (their number comes from the constraint system, not
# from the file), and every selector's 2^k assignment bits packed eight to a byte, least significant bit first (none
# when selectors were not compressed into fixed columns).  As upstream, the constraint system itself is not stored: the
# reader is told how many permutation columns and selectors the circuit's `configure` produces.
# A proving key is its verifying key followed by l0, l_last, l_active_row (extended domain), fixed_values (Lagrange),
# fixed_polys (coefficients), fixed_cosets (extended), and the permutation argument's permutations, polys, cosets; the
# three single polynomials as polynomial records, the six lists as u32 BE count + records.
VK_VERSION = 0x03


class VerifyingKeyRaw:
    def __init__(self, k, fixed_commitments, permutation_commitments, selectors=(), compress_selectors=True):
        self.k = int(k)
        self.fixed_commitments = np.asarray(fixed_commitments, dtype="<u8").reshape(-1, 8)
        self.permutation_commitments = np.asarray(permutation_commitments, dtype="<u8").reshape(-1, 8)
        self.selectors = [np.asarray(sel, dtype=bool) for sel in selectors]
        self.compress_selectors = bool(compress_selectors)
        for sel in self.selectors:
            if sel.shape != (1 << self.k,):
                raise FormatError("verifying key: a selector holds one bit per row")
        if self.selectors and not self.compress_selectors:
            raise FormatError("verifying key: selectors are stored only when they were compressed")


def write_verifying_key_raw(f, vk: VerifyingKeyRaw, layout: str = "v3") -> None:
    if layout == "v3":
        if vk.k > 255:
            raise FormatError("verifying key: k does not fit one byte")
        f.write(bytes([VK_VERSION, vk.k, 1 if vk.compress_selectors else 0]))
        f.write(struct.pack("<I", len(vk.fixed_commitments)))
    elif layout == "legacy":
        f.write(struct.pack(">I", vk.k))
        f.write(struct.pack(">I", len(vk.fixed_commitments)))
    else:
        raise FormatError(f"verifying key: unknown layout {layout!r}")
    f.write(np.ascontiguousarray(vk.fixed_commitments, dtype="<u8").tobytes())
    f.write(np.ascontiguousarray(vk.permutation_commitments, dtype="<u8").tobytes())
    for sel in vk.selectors:
        f.write(np.packbits(sel, bitorder="little").tobytes())


def _read_exact(f, n: int, what: str) -> bytes:
    b = f.read(n)
    if len(b) != n:
        raise FormatError(f"{what}: truncated")
    return b


def _read_points(f, count: int, what: str) -> np.ndarray:
    pts = np.frombuffer(_read_exact(f, count * G1_RAW, what), dtype="<u8").reshape(count, 8)
    for half in (pts[:, :4], pts[:, 4:]):
        if count and not _all_below_modulus(half, FQ_MODULUS):
            raise FormatError(f"{what}: coordinate not below the base-field modulus")
    return pts


def read_verifying_key_raw(f, num_permutation_columns: int, num_selectors: int = 0, max_k: int = 28) -> VerifyingKeyRaw:
    """num_permutation_columns / num_selectors: what the circuit's constraint system has (upstream re-runs `configure`)."""
    first = _read_exact(f, 1, "verifying key")
    if first[0] == VK_VERSION:
        k, compress = _read_exact(f, 2, "verifying key")
        if compress > 1:
            raise FormatError("verifying key: compress_selectors is not a boolean")
        (nfixed,) = struct.unpack("<I", _read_exact(f, 4, "verifying key"))
    elif first[0] == 0:
        (k,) = struct.unpack(">I", first + _read_exact(f, 3, "verifying key"))
        (nfixed,) = struct.unpack(">I", _read_exact(f, 4, "verifying key"))
        compress = 1
    else:
        raise FormatError(f"verifying key: unknown version byte {first[0]:#x}")
    if k > max_k:
        raise FormatError(f"verifying key: k = {k} is beyond {max_k}")
    if nfixed > 1 << 20:
        raise FormatError("verifying key: implausible number of fixed commitments")
    fixed = _read_points(f, nfixed, "verifying key")
    perm = _read_points(f, num_permutation_columns, "verifying key")
    selectors = []
    if compress:
        nbytes = ((1 << k) + 7) // 8
        for _ in range(num_selectors):
            bits = np.unpackbits(np.frombuffer(_read_exact(f, nbytes, "verifying key"), dtype=np.uint8), bitorder="little")
            selectors.append(bits[: 1 << k].astype(bool))
    return VerifyingKeyRaw(k, fixed, perm, selectors, bool(compress))


def write_polynomial_slice_raw(f, polys) -> None:
    f.write(struct.pack(">I", len(polys)))
    for poly in polys:
        write_polynomial_raw(f, poly)


def read_polynomial_slice_raw(f, max_polys: int = 1 << 16, max_len: int = 1 << 30):
    (count,) = struct.unpack(">I", _read_exact(f, 4, "polynomial slice"))
    if count > max_polys:
        raise FormatError(f"polynomial slice: {count} polynomials is beyond {max_polys}")
    return [read_polynomial_raw(f, max_len) for _ in range(count)]


class ProvingKeyRaw:
    """Host arrays ((len, 4) uint64 Montgomery limbs) of everything `ProvingKey::write` stores."""
    SINGLE = ("l0", "l_last", "l_active_row")
    SLICES = ("fixed_values", "fixed_polys", "fixed_cosets", "permutations", "permutation_polys", "permutation_cosets")

    def __init__(self, vk: VerifyingKeyRaw, **polys):
        self.vk = vk
        for name in self.SINGLE:
            setattr(self, name, np.asarray(polys[name], dtype="<u8").reshape(-1, 4))
        for name in self.SLICES:
            setattr(self, name, [np.asarray(a, dtype="<u8").reshape(-1, 4) for a in polys[name]])

    def check(self, extended_k: int) -> None:
        n, en = 1 << self.vk.k, 1 << extended_k
        want = {"l0": en, "l_last": en, "l_active_row": en, "fixed_values": n, "fixed_polys": n, "fixed_cosets": en,
                "permutations": n, "permutation_polys": n, "permutation_cosets": en}
        for name in self.SINGLE:
            if len(getattr(self, name)) != want[name]:
                raise FormatError(f"proving key: {name} has {len(getattr(self, name))} elements, the domain needs {want[name]}")
        for name in self.SLICES:
            for a in getattr(self, name):
                if len(a) != want[name]:
                    raise FormatError(f"proving key: a polynomial of {name} has {len(a)} elements, the domain needs {want[name]}")
        if not (len(self.fixed_values) == len(self.fixed_polys) == len(self.fixed_cosets) == len(self.vk.fixed_commitments)):
            raise FormatError("proving key: fixed column lists differ in length")
        if not (len(self.permutations) == len(self.permutation_polys) == len(self.permutation_cosets)
                == len(self.vk.permutation_commitments)):
            raise FormatError("proving key: permutation lists differ in length")


def write_proving_key_raw(path: str, pk: ProvingKeyRaw, layout: str = "v3") -> None:
    with open(path, "wb") as f:
        write_verifying_key_raw(f, pk.vk, layout)
        for name in ProvingKeyRaw.SINGLE:
            write_polynomial_raw(f, getattr(pk, name))
        for name in ProvingKeyRaw.SLICES:
            write_polynomial_slice_raw(f, getattr(pk, name))


def read_proving_key_raw(path: str, num_permutation_columns: int, num_selectors: int = 0) -> ProvingKeyRaw:
    with open(path, "rb") as f:
        vk = read_verifying_key_raw(f, num_permutation_columns, num_selectors)
        polys = {name: read_polynomial_raw(f) for name in ProvingKeyRaw.SINGLE}
        for name in ProvingKeyRaw.SLICES:
            polys[name] = read_polynomial_slice_raw(f)
        if f.read(1):
            raise FormatError("proving key: trailing bytes")
    return ProvingKeyRaw(vk, **polys)


# ---- instances and the proof JSON -------------------------------------------------------------------------------------
def _mont_to_int(limbs) -> int:
    v = 0
    for i, w in enumerate(limbs):
        v |= int(w) << (64 * i)
    return v * _R_INV % FR_MODULUS


def _int_to_mont(v: int) -> np.ndarray:
    m = (v << 256) % FR_MODULUS
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype="<u8")


def serialize_instances(instances) -> bytes:
    """One instance column; each value as its 32 canonical bytes, big-endian (prover/src/proof.rs:129-142)."""
    if len(instances) != 1:
        raise FormatError("instances: exactly one instance column is serialised")
    col = np.asarray(instances[0], dtype="<u8").reshape(-1, 4)
    return b"".join(_mont_to_int(row).to_bytes(32, "big") for row in col)


def deserialize_instances(data: bytes):
    """Inverse of serialize_instances (prover/src/proof.rs:55-63); a non-canonical value is an error as in from_repr."""
    if len(data) % 32:
        raise FormatError("instances: length is not a multiple of 32")
    out = np.empty((len(data) // 32, 4), dtype="<u8")
    for i in range(len(data) // 32):
        v = int.from_bytes(data[32 * i: 32 * i + 32], "big")
        if v >= FR_MODULUS:
            raise FormatError("instances: value not below the field modulus")
        out[i] = _int_to_mont(v)
    return [out]


def proof_path(directory: str, name: str) -> str:
    return f"{directory}/full_proof_{name}.json"       # prover/src/proof.rs:94-96


def dump_proof_json(directory: str, name: str, proof: bytes, instances, vk: bytes = b"", protocol: bytes | None = None) -> str:
    """{"proof", "instances", "vk"}, each standard base64 (prover/src/proof.rs:24-48, 78-84); with `protocol` (the
    serialised PlonkProtocol of a chunk proof) that field comes first and the others are flattened beside it
    (prover/src/proof/chunk.rs:22-28).  Returns the path."""
    doc = {}
    if protocol is not None:
        doc["protocol"] = base64.b64encode(bytes(protocol)).decode("ascii")
    doc["proof"] = base64.b64encode(bytes(proof)).decode("ascii")
    doc["instances"] = base64.b64encode(serialize_instances(instances)).decode("ascii")
    doc["vk"] = base64.b64encode(bytes(vk)).decode("ascii")
    path = proof_path(directory, name)
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    return path


def load_proof_json(directory: str, name: str, with_protocol: bool = False):
    """(proof, instances, vk), or (protocol, proof, instances, vk) for a chunk proof."""
    path = proof_path(directory, name)
    if not os.path.exists(path):
        raise FormatError(f"File {path} doesn't exist")
    with open(path) as f:
        doc = json.load(f)
    try:
        proof = base64.b64decode(doc["proof"], validate=True)
        inst = base64.b64decode(doc["instances"], validate=True)
        vk = base64.b64decode(doc["vk"], validate=True)
        protocol = base64.b64decode(doc["protocol"], validate=True) if with_protocol else None
    except (KeyError, ValueError) as e:
        raise FormatError(f"proof JSON: {e}") from None
    if with_protocol:
        return protocol, proof, deserialize_instances(inst), vk
    return proof, deserialize_instances(inst), vk


# ---- benchmark log lines ----------------------------------------------------------------------------------------------
# The reference's benchmark prints a summary line and brackets every phase with ark_std's start_timer! / end_timer!
# (zkevm-circuits/benches/benchmark_list/main.rs:44-52 for the tags, :68-71 for the summary, :210-245 for the messages);
# scripts/report_system_data/report_bench.py:10-110 scrapes those lines into its report.  Unpinned: the lines are checked
# against a restatement of that script's parsing steps, not against the script itself, and ark_std's timer format is
# restated from its published source.
CIRCUIT_SUMMARY = "[Circuit summary]"
PHASE_TAGS = {
    "generate_witness": ("[Generate witness]", "Generate witness of one transaction's trace."),
    "create_circuit": ("[Create circuit]", "Create a new SubCircuit from witness."),
    "create_proof": ("[Create_proof]", "Create proof/Round 1/1"),
    "verify_proof": ("[Verify proof]", "Verify proof/Round 1/1"),
}


def _ark_time(seconds: float) -> str:
    """Elapsed time as ark_std's end_timer! prints it: 12.345s, 45.678ms, 9.012µs or 7ns."""
    ns = int(round(seconds * 1e9))
    secs, millis, micros, nanos = ns // 10**9, ns // 10**6 % 1000, ns // 1000 % 1000, ns % 1000
    if secs:
        return f"{secs}.{millis:03d}s"
    if millis:
        return f"{millis}.{micros:03d}ms"
    if micros:
        return f"{micros}.{nanos:03d}µs"
    return f"{nanos}ns"


def reference_log_lines(test_id: str, max_num_row: int, degree: int, phases: dict, passed: bool = True):
    """The lines a run of the reference's benchmark would leave for `report_bench.py`: the circuit summary, cargo's
    `running 1 test`, a Start / End pair per phase in `phases` (name -> seconds, names as in PHASE_TAGS) and the test
    result.  End lines pad the message with dots to a fixed width, as ark_std does."""
    unknown = set(phases) - set(PHASE_TAGS)
    if unknown:
        raise FormatError(f"log: unknown phases {sorted(unknown)}")
    lines = [f"{CIRCUIT_SUMMARY}/id:{test_id}, max_num_row:{max_num_row}, degree:{degree}, round:1, use params file:false",
             "running 1 test"]
    for name in PHASE_TAGS:                      # the order the benchmark runs them in
        if name not in phases:
            continue
        tag, text = PHASE_TAGS[name]
        msg = f"{tag}/{test_id}/{text}"
        lines.append(f"{'Start:':8} {msg}")
        lines.append(f"{'End:':8} {msg} {'.' * max(3, 75 - len(msg))}{_ark_time(phases[name])}")
    lines.append("test result: ok. 1 passed; 0 failed; 0 ignored; 0 measured; 0 filtered out" if passed
                 else "test result: FAILED. 0 passed; 1 failed; 0 ignored; 0 measured; 0 filtered out")
    return lines
