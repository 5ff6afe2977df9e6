"""The integer claims the encode / split code of csrc/kg_device.hpp rests on, restated in numpy and brute-forced
(CPU only; the kernels themselves are checked against the oracle in the GPU parity tests).

  * half codes: residues travel as dwords, "not an amino acid" = 2^20; c0*8000 + c1*400 + c2*20 + c3 (mod 2^32)
    is < 2^20 iff all four residues are valid;
  * split_fast: quotient and remainder of hi*160000 + lo by n with 24-bit and 32-bit multiplies, exact for
    64 <= n < 2^31;
  * tag_qs: the 8-bit fingerprint of (q, slot) never collides for keys that share a home slot and is close to
    uniform for neighbours.
"""
import itertools

import numpy as np

BAD = 1 << 20
M32 = (1 << 32) - 1


def half_code(c0, c1, c2, c3):
    mul24 = lambda a, b: ((a & 0xFFFFFF) * (b & 0xFFFFFF)) & M32          # v_mul_u32_u24
    return (mul24(c0, 8000) + mul24(c1, 400) + mul24(c2, 20) + c3) & M32


def test_half_code_validity_needs_no_per_residue_compare():
    # exhaustive over which residues are bad, with the extreme valid codes in the other places
    for bad in itertools.product((False, True), repeat=4):
        for fill in (0, 19):
            c = [BAD if b else fill for b in bad]
            h = half_code(*c)
            assert (h < BAD) == (not any(bad)), (bad, fill, h)
    # and every valid half code is the base-20 number
    rng = np.random.default_rng(1)
    for c in rng.integers(0, 20, (1000, 4)):
        assert half_code(*map(int, c)) == int(c[0]) * 8000 + int(c[1]) * 400 + int(c[2]) * 20 + int(c[3]) < 160000
    # the subset sums of the weights of the bad places are never a multiple of 4096 (= 2^32 / 2^20)
    for k in range(1, 5):
        for sub in itertools.combinations((8000, 400, 20, 1), k):
            assert sum(sub) % 4096 != 0


def split_fast(hi, lo, n):
    """kg_device.hpp split_fast with Python integers in numpy object arrays (exact), 32-bit wrap made explicit."""
    m35 = (1 << 35) // n
    vh = hi * 20000 + (lo >> 3)
    assert (vh < (1 << 32)).all()                       # value >> 3 fits 32 bits
    v32 = (hi * 160000 + lo) & M32
    q = (vh * m35) >> 32                                # v_mul_hi_u32
    r = (v32 - ((q * n) & M32)) & M32                   # v_mul_lo_u32, 32-bit subtract
    fix = r >= n
    return q + fix, np.where(fix, r - n, r)


def test_split_fast_is_exact_for_64_le_n_lt_2_31():
    rng = np.random.default_rng(2)
    ns = [64, 65, 100, 101, 1009, 50021, 1_000_003, 12_345_678, 1_400_303_159, (1 << 30), (1 << 31) - 19, (1 << 31) - 1]
    ns += [int(x) for x in rng.integers(64, 1 << 31, 20)]
    for n in ns:
        hi = np.concatenate([rng.integers(0, 160000, 200_000), [0, 159999, 159999, 0]]).astype(object)
        lo = np.concatenate([rng.integers(0, 160000, 200_000), [0, 159999, 0, 159999]]).astype(object)
        # values just below / at / above multiples of n, where a quotient estimate is most likely to be off by one
        k = rng.integers(0, 20 ** 8 // n + 1, 100_000).astype(object)
        d = rng.integers(-2, 3, 100_000).astype(object)
        v2 = np.clip(k * n + d, 0, 20 ** 8 - 1)
        hi, lo = np.concatenate([hi, v2 // 160000]), np.concatenate([lo, v2 % 160000])
        v = hi * 160000 + lo
        q, r = split_fast(hi, lo, n)
        assert (q == v // n).all() and (r == v % n).all(), n


def tag_qs(q, slot):
    q = q.astype(np.uint64); slot = slot.astype(np.uint64)
    mul24 = lambda a, b: ((a & np.uint64(0xFFFFFF)) * np.uint64(b)) & np.uint64(M32)
    q32 = (q ^ (q >> np.uint64(32))) & np.uint64(M32)
    h = mul24(slot, 0x9E3779) ^ mul24(slot >> np.uint64(24), 0x85EBCB) ^ mul24(q32 ^ (q32 >> np.uint64(19)), 0xC2B2AF)
    t = (h >> np.uint64(16)) & np.uint64(0xFF)
    return np.where(t == 0xFF, 0xFE, t)


def test_fingerprint_of_q_and_slot():
    rng = np.random.default_rng(3)
    n = 1_400_303_159
    s = rng.integers(0, n, 1_000_000)
    q1 = rng.integers(0, 19, s.size)
    q2 = (q1 + rng.integers(1, 19, s.size)) % 19
    assert not (tag_qs(q1, s) == tag_qs(q2, s)).any()                     # same home slot, different key: never equal
    for d in (1, 2, 3, 7, 15):                                            # neighbours: about 1 / 255
        rate = float((tag_qs(q1, s) == tag_qs(q2, s + d)).mean())
        assert rate < 2.0 / 255, (d, rate)
    hist = np.bincount(tag_qs(q1, s).astype(int), minlength=256)
    assert hist[0xFF] == 0 and hist[:0xFE].min() > 0.8 * s.size / 256 and hist[:0xFE].max() < 1.2 * s.size / 256
