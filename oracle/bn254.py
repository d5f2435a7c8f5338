"""BN254 big-int model: Fr, Fq, G1, Poseidon (TEST INFRASTRUCTURE, not a product path).

Pure-Python restatement of the arithmetic contract of `halo2curves 0.6.1`
(`bn256::{Fr,Fq,G1,G1Affine}`, pinned at /root/reference/Cargo.lock:2213-2217), which the
reference reaches through `halo2_proofs::halo2curves::bn256::*`
(/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:20).  The crate's source is not in
the reference tree, so the constants below are re-derived numerically by `self_check()` rather
than trusted, and Fr arithmetic is pinned by the reference's own Poseidon known-answer tests
(/root/reference/poseidon/src/hash.rs:903-950) and Fr byte-convention tests
(/root/reference/zkevm-circuits/src/util.rs:832-882).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

# ---------------------------------------------------------------------------------------------
# Field constants
# ---------------------------------------------------------------------------------------------
FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FQ_MODULUS = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_BITS = 256
MASK64 = (1 << 64) - 1

FR_S = 28                       # 2-adicity of r-1
FR_GENERATOR = 7                # multiplicative generator used by halo2curves
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (FR_MODULUS - 1) >> FR_S, FR_MODULUS)
FR_ZETA = 0x0000000000000000B3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD  # primitive cube root
FR_DELTA = pow(FR_GENERATOR, 1 << FR_S, FR_MODULUS)

G1_B = 3
G1_GENERATOR = (1, 2)


class Field:
    """Prime field with halo2curves' in-memory convention: 4 x u64 LE limbs, Montgomery form."""

    def __init__(self, p: int):
        self.p = p
        self.R = (1 << R_BITS) % p
        self.R2 = self.R * self.R % p
        self.R3 = self.R2 * self.R % p
        self.Rinv = pow(self.R, -1, p)
        self.inv64 = (-pow(p, -1, 1 << 64)) % (1 << 64)
        self.inv32 = self.inv64 & 0xFFFFFFFF

    # canonical int <-> Montgomery limbs -------------------------------------------------------
    def to_mont(self, a: int) -> int:
        return a * self.R % self.p

    def from_mont(self, m: int) -> int:
        return m * self.Rinv % self.p

    def to_limbs(self, a: int):
        """canonical value -> [u64;4] Montgomery limbs (what sits in an `Fr([u64;4])`)."""
        m = self.to_mont(a % self.p)
        return [(m >> (64 * i)) & MASK64 for i in range(4)]

    def from_limbs(self, limbs) -> int:
        m = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
        assert m < self.p, "non-canonical Montgomery residue"
        return self.from_mont(m)

    # `PrimeField::to_repr` / `from_repr`: canonical little-endian 32 bytes -----------------------
    def to_repr(self, a: int) -> bytes:
        return (a % self.p).to_bytes(32, "little")

    def from_repr(self, b: bytes):
        v = int.from_bytes(b, "little")
        return v if v < self.p else None

    # `FromUniformBytes<64>`: 512-bit little-endian integer reduced mod p ------------------------
    def from_uniform_bytes(self, b: bytes) -> int:
        assert len(b) == 64
        return int.from_bytes(b, "little") % self.p

    def inv(self, a: int) -> int:
        return pow(a, -1, self.p) if a % self.p else 0


Fr = Field(FR_MODULUS)
Fq = Field(FQ_MODULUS)


def limbs_of_int(v: int):
    return [(v >> (64 * i)) & MASK64 for i in range(4)]


# ---------------------------------------------------------------------------------------------
# G1: y^2 = x^3 + 3 over Fq. Affine points are (x, y) tuples, identity is None.
# ---------------------------------------------------------------------------------------------
def g1_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - G1_B) % FQ_MODULUS == 0


def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % FQ_MODULUS)


def g1_add(P, Q):
    p = FQ_MODULUS
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def g1_mul(P, k: int):
    k %= FR_MODULUS
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, P)
        P = g1_add(P, P)
        k >>= 1
    return acc


def g1_msm_naive(scalars, points):
    acc = None
    for s, P in zip(scalars, points):
        acc = g1_add(acc, g1_mul(P, s))
    return acc


def g1_affine_to_limbs(P):
    """G1Affine{x,y} as 8 u64: Montgomery x limbs then y limbs; identity is (0,0)."""
    if P is None:
        return [0] * 8
    return Fq.to_limbs(P[0]) + Fq.to_limbs(P[1])


def g1_affine_from_limbs(l):
    if all(int(v) == 0 for v in l):
        return None
    return (Fq.from_limbs(l[0:4]), Fq.from_limbs(l[4:8]))


def g1_jacobian_from_limbs(l):
    """G1{x,y,z} Jacobian (X/Z^2, Y/Z^3) as 12 u64 Montgomery limbs -> affine tuple / None."""
    X, Y, Z = Fq.from_limbs(l[0:4]), Fq.from_limbs(l[4:8]), Fq.from_limbs(l[8:12])
    if Z == 0:
        return None
    zi = Fq.inv(Z)
    zi2 = zi * zi % FQ_MODULUS
    return (X * zi2 % FQ_MODULUS, Y * zi2 * zi % FQ_MODULUS)


def g1_bases(n: int):
    """[(i+1)]G for i < n by running addition from G = (1, 2)."""
    out, acc = [], None
    for _ in range(n):
        acc = g1_add(acc, G1_GENERATOR)
        out.append(acc)
    return out


# ---------------------------------------------------------------------------------------------
# Poseidon (x^5, t=3, R_F=8, R_P=57 over Fr) with constants regenerated from the Grain LFSR of
# the Poseidon reference parameter script; the reference holds the same numbers as decimal
# strings (/root/reference/poseidon/src/poseidon/primitives/bn256/fp.rs:60-280) and uses them at
# /root/reference/poseidon/src/poseidon/primitives.rs:91-141 (permute) and :143-173 (sponge).
# ---------------------------------------------------------------------------------------------
def _grain_bits(field_tag, sbox, n, t, r_f, r_p):
    bits = []
    for val, width in ((field_tag, 2), (sbox, 4), (n, 12), (t, 12), (r_f, 10), (r_p, 10)):
        bits += [int(c) for c in bin(val)[2:].zfill(width)]
    bits += [1] * 30
    assert len(bits) == 80

    def step():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0)
        bits.append(nb)
        return nb

    for _ in range(160):
        step()
    while True:
        nb = step()
        while nb == 0:
            step()
            nb = step()
        yield step()


def poseidon_params(t=3, r_f=8, r_p=57, n=254, p=FR_MODULUS):
    gen = _grain_bits(1, 0, n, t, r_f, r_p)

    def rand_int():
        v = 0
        for _ in range(n):
            v = (v << 1) | next(gen)
        return v

    rc = []
    while len(rc) < (r_f + r_p) * t:
        v = rand_int()
        if v < p:
            rc.append(v)
    while True:
        xs_ys = [rand_int() % p for _ in range(2 * t)]
        if len(set(xs_ys)) == 2 * t:
            break
    xs, ys = xs_ys[:t], xs_ys[t:]
    mds = [[pow(xs[i] + ys[j], -1, p) for j in range(t)] for i in range(t)]
    rounds = [rc[i * t:(i + 1) * t] for i in range(r_f + r_p)]
    return rounds, mds


_POSEIDON = None


def poseidon_permute(state):
    global _POSEIDON
    if _POSEIDON is None:
        _POSEIDON = poseidon_params()
    rounds, mds = _POSEIDON
    p = FR_MODULUS
    t = 3
    half_f, r_p = 4, 57
    st = list(state)

    def mix(s):
        return [sum(mds[i][j] * s[j] for j in range(t)) % p for i in range(t)]

    for rnd, rcs in enumerate(rounds):
        full = rnd < half_f or rnd >= half_f + r_p
        if full:
            st = [pow((st[i] + rcs[i]) % p, 5, p) for i in range(t)]
        else:
            st = [(st[i] + rcs[i]) % p for i in range(t)]
            st[0] = pow(st[0], 5, p)
        st = mix(st)
    return st


def poseidon_hash_with_domain(inp, domain: int) -> int:
    """`Fr::hash_with_domain([a,b], domain)` (/root/reference/poseidon/src/hash.rs:106-108):
    iden3 layout — capacity word is state[0], inputs right-aligned, output state[0]."""
    st = [domain % FR_MODULUS, inp[0] % FR_MODULUS, inp[1] % FR_MODULUS]
    return poseidon_permute(st)[0]


def poseidon_hash_msg(msg, cap: int) -> int:
    """`Fr::hash_msg(msg, Some(cap))` (/root/reference/poseidon/src/hash.rs:114-117):
    variable-length iden3 sponge, rate 2, zero padding to a multiple of the rate."""
    st = [cap % FR_MODULUS, 0, 0]
    m = list(msg)
    if len(m) % 2:
        m.append(0)
    for i in range(0, len(m), 2):
        st[1] = (st[1] + m[i]) % FR_MODULUS
        st[2] = (st[2] + m[i + 1]) % FR_MODULUS
        st = poseidon_permute(st)
    return st[0]


# ---------------------------------------------------------------------------------------------
def self_check():
    """Numerically re-derive every constant used downstream."""
    for F in (Fr, Fq):
        assert F.inv64 * F.p % (1 << 64) == (1 << 64) - 1
    assert Fr.to_limbs(1) == [0xAC96341C4FFFFFFB, 0x36FC76959F60CD29, 0x666EA36F7879462E, 0x0E0A77C19A07DF2F]
    assert Fq.to_limbs(1) == [0xD35D438DC58F0D9D, 0x0A78EB28F5C70B3D, 0x666EA36F7879462C, 0x0E0A77C19A07DF2F]
    assert Fr.inv64 == 0xC2E1F593EFFFFFFF and Fq.inv64 == 0x87D20782E4866389
    assert (FR_MODULUS - 1) % (1 << FR_S) == 0 and ((FR_MODULUS - 1) >> FR_S) % 2 == 1
    assert FR_ROOT_OF_UNITY == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert pow(FR_ROOT_OF_UNITY, 1 << FR_S, FR_MODULUS) == 1
    assert pow(FR_ROOT_OF_UNITY, 1 << (FR_S - 1), FR_MODULUS) == FR_MODULUS - 1
    assert FR_ZETA == pow(FR_GENERATOR, (FR_MODULUS - 1) // 3, FR_MODULUS)
    assert FR_ZETA != 1 and pow(FR_ZETA, 3, FR_MODULUS) == 1
    assert FR_DELTA == 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2
    assert g1_is_on_curve(G1_GENERATOR)
    assert g1_mul(G1_GENERATOR, FR_MODULUS - 1) == g1_neg(G1_GENERATOR)


if __name__ == "__main__":
    self_check()
    print("bn254 constants ok")
    print(hex(poseidon_hash_with_domain([1, 2], 0)))
