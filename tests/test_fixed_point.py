"""drt_amd/csrc/drt_fixed.h -- the 128-bit fixed-point accumulator of the deterministic mode (include/drt_hip.h drt_deterministic;
SURVEY.md section 5 "race detection / sanitizers"; the sum it makes order-independent is the one reference optim.py:155-171 clamps) --
compiled for the host and held against Python's exact integers: conversion truncates the magnitude at 2^-80, the sum is exact, the
conversion back rounds once to nearest-even, non-finite and huge contributions become sticky flags."""
import ctypes
import math
import random
from fractions import Fraction

import numpy as np


def _from(hs, x):
    hi, lo, fl = ctypes.c_int64(), ctypes.c_uint64(), ctypes.c_uint32()
    hs.hs_fx_from(x, ctypes.byref(hi), ctypes.byref(lo), ctypes.byref(fl))
    return hi.value * 2 ** 64 + lo.value, fl.value


def _to(hs, v, flags=0):
    v %= 2 ** 128
    hi, lo = v >> 64, v & (2 ** 64 - 1)
    return hs.hs_fx_to(hi - 2 ** 64 if hi >= 2 ** 63 else hi, lo, flags)


def test_conversion_truncates_at_the_unit_and_is_exact_above_it(hostsim):
    rnd = random.Random(1)
    for _ in range(50000):
        x = rnd.choice([-1, 1]) * rnd.random() * 2.0 ** rnd.uniform(-120, 50)
        v, fl = _from(hostsim, x)
        if abs(x) >= 2.0 ** 46:
            assert v == 0 and fl == (4 if x < 0 else 2)
            continue
        assert fl == 0 and v == int(Fraction(x) * 2 ** 80)           # int(): toward zero
        if abs(x) >= 2.0 ** -28:
            assert _to(hostsim, v) == x                                 # every mantissa bit above the unit: exact round trip
    for x, fl in ((math.inf, 2), (-math.inf, 4), (math.nan, 1), (0.0, 0), (-0.0, 0), (5e-324, 0)):
        assert _from(hostsim, x) == (0, fl)


def test_the_sum_is_rounded_once_to_nearest_even(hostsim):
    rnd = random.Random(2)
    for _ in range(50000):
        v = rnd.getrandbits(rnd.randint(1, 126)) * rnd.choice([-1, 1])
        if rnd.random() < 0.25:                                         # exact ties
            p = rnd.randint(54, 120)
            v = (((rnd.getrandbits(53) | (1 << 52)) << (p - 52)) | (1 << (p - 53))) * rnd.choice([-1, 1])
        assert _to(hostsim, v) == float(Fraction(v, 2 ** 80))           # (Fraction -> float rounds to nearest-even)
    assert math.isnan(_to(hostsim, 7, 1)) and math.isnan(_to(hostsim, 7, 6))
    assert _to(hostsim, 7, 2) == math.inf and _to(hostsim, -7, 4) == -math.inf


def test_any_order_gives_the_same_bits_and_the_correctly_rounded_sum(hostsim):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(20000) * 10.0 ** rng.uniform(-6, 2, 20000)).astype(np.float64)
    ref = hostsim.hs_fx_sum(x.ctypes.data, len(x))
    for _ in range(5):
        y = np.ascontiguousarray(rng.permutation(x))
        assert hostsim.hs_fx_sum(y.ctypes.data, len(y)) == ref          # bit for bit
    exact = sum((Fraction(int(Fraction(float(v)) * 2 ** 80), 2 ** 80) for v in x), Fraction(0))
    assert ref == float(exact)
    assert abs(ref - math.fsum(x)) <= 20000 * 2.0 ** -80 + abs(ref) * 2.0 ** -52
