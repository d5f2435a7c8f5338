"""Shared helpers of the test-suite: hex fixtures <-> Montgomery limb arrays, seeded inputs."""
import json
import os

import numpy as np

import bn254 as bn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def fr_arr(vals):
    """canonical ints -> (n, 4) uint64 Montgomery limbs"""
    return np.array([bn.Fr.to_limbs(int(v)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fq_arr(vals):
    return np.array([bn.Fq.to_limbs(int(v)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def hex_fr(hexes):
    return fr_arr([int(h, 16) for h in hexes])


def fr_ints(arr):
    return [bn.Fr.from_limbs(row) for row in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


def fq_ints(arr):
    return [bn.Fq.from_limbs(row) for row in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


def affine_arr(points):
    """[(x, y) | None] -> (n, 8) uint64"""
    return np.array([bn.g1_affine_to_limbs(p) for p in points], dtype=np.uint64).reshape(-1, 8)


def affine_pt(row):
    return bn.g1_affine_from_limbs([int(v) for v in row])


def hex_pt(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def rand_fr(rng, n):
    """uniform-ish canonical Montgomery residues (< 2^252 < r), numpy Generator"""
    a = rng.integers(0, 2 ** 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    return a


def witness_like(rng, n, to_mont):
    """60 % zero, 20 % one, 10 % < 2^8, 5 % < 2^16, 5 % uniform (canonical values, then Montgomery form)"""
    u = rng.random(n)
    canon = np.zeros((n, 4), dtype=np.uint64)
    canon[(u >= 0.6) & (u < 0.8), 0] = 1
    m = (u >= 0.8) & (u < 0.9)
    canon[m, 0] = rng.integers(0, 1 << 8, size=int(m.sum()), dtype=np.uint64)
    m = (u >= 0.9) & (u < 0.95)
    canon[m, 0] = rng.integers(0, 1 << 16, size=int(m.sum()), dtype=np.uint64)
    m = u >= 0.95
    canon[m] = rand_fr(rng, int(m.sum()))
    return to_mont(canon)
