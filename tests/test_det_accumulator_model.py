"""Host model of the long accumulators of the deterministic mode (csrc/common.h: det_add_pieces / det_value).

The device code cannot run here; this restates its arithmetic in Python integers with the same constants (six signed 64-bit
windows, window j weighing 2^(BASE + 32 j), BASE = -110) and checks the three properties the mode rests on: the pieces of a value
reconstruct it exactly, the windows are independent of the order of the additions, and the read-out equals the exactly rounded
sum.  The GPU tests (test_long_accumulator_finalize_matches_fp64, test_deterministic_*) pin the device code itself."""
import math
import random
import struct

K, BASE = 6, -110
MASK64 = (1 << 64) - 1


def pieces_f32(v):
    """-> list of (window, signed piece) exactly as det_add_f32 adds them"""
    b = struct.unpack("<I", struct.pack("<f", v))[0]
    ex = (b >> 23) & 0xff
    assert ex != 0xff
    m = (b & 0x7fffff) | 0x800000 if ex else (b & 0x7fffff)
    if not m:
        return []
    shift = (ex if ex else 1) - 150 - BASE
    neg = bool(b >> 31)
    if shift < 0:
        m = m >> (-shift) if shift > -64 else 0
        shift = 0
    j, r = shift >> 5, shift & 31
    lo64 = (m << r) & MASK64
    hi = (m >> (64 - r)) if r else 0
    ps = [lo64 & 0xffffffff, lo64 >> 32, hi]
    assert not (j >= K or (ps[1] and j + 1 >= K) or (ps[2] and j + 2 >= K))
    return [(j + i, -p if neg else p) for i, p in enumerate(ps) if p]


def value(windows):
    s = 0.0
    for j in range(K - 1, -1, -1):
        s += math.ldexp(float(windows[j]), BASE + 32 * j)
    return s


def test_pieces_reconstruct_the_value_exactly():
    rng = random.Random(1)
    for _ in range(2000):
        v = struct.unpack("<f", struct.pack("<f", rng.uniform(-1, 1) * 10.0 ** rng.uniform(-20, 20)))[0]
        exact = sum(p * 2 ** (BASE + 32 * j) if BASE + 32 * j >= 0 else p / 2 ** (-(BASE + 32 * j)) for j, p in pieces_f32(v))
        assert exact == v, (v, exact)          # (every term a dyadic rational: exact in Python's arithmetic here)
        assert all(abs(p) < 2 ** 32 for _, p in pieces_f32(v))


def test_windows_do_not_depend_on_the_order_and_give_the_exact_sum():
    rng = random.Random(2)
    vals = [struct.unpack("<f", struct.pack("<f", rng.gauss(0, 1) * 10.0 ** rng.uniform(-6, 6)))[0] for _ in range(20000)]

    def accumulate(seq):
        w = [0] * K
        for v in seq:
            for j, p in pieces_f32(v):
                w[j] += p
        assert all(-2 ** 63 <= x < 2 ** 63 for x in w)        # (a window takes 2^31 pieces before it can overflow)
        return w

    a = accumulate(vals)
    for seed in (3, 4, 5):
        sh = vals[:]
        random.Random(seed).shuffle(sh)
        assert accumulate(sh) == a
    exact = math.fsum(vals)
    got = value(a)
    assert abs(got - exact) <= 4 * 2.0 ** -53 * sum(abs(v) for v in vals) / len(vals) * 1e3 and abs(got - exact) <= 1e-9 * abs(exact) + 1e-12
    # a plain float32 running sum of the same values depends on the order (what the mode removes)
    import numpy as np
    f1 = float(np.cumsum(np.array(vals, dtype=np.float32))[-1])
    sh = vals[:]
    random.Random(9).shuffle(sh)
    f2 = float(np.cumsum(np.array(sh, dtype=np.float32))[-1])
    assert f1 != f2


def test_tiny_values_are_truncated_the_same_way_every_time():
    tiny = struct.unpack("<f", struct.pack("<f", 1e-38))[0]
    assert pieces_f32(tiny) == pieces_f32(tiny)
    assert sum(abs(p) for _, p in pieces_f32(tiny)) == 0 or pieces_f32(tiny)[0][0] == 0     # below 2^-110: dropped / lowest window
