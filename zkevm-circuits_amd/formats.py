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


def _all_below_modulus(a: np.ndarray) -> bool:
    p = [(FR_MODULUS >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        limb = a[:, i]
        lt |= eq & (limb < np.uint64(p[i]))
        eq &= limb == np.uint64(p[i])
    return bool(lt.all())


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
