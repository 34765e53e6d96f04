// drt_fixed.h -- order-independent accumulation of float64 gradients: 128-bit fixed point.
//
// The vertex gradient of a step is a sum of millions of per-path / per-edge contributions scattered with atomics (reference
// optim.py:155-171 clamps that sum; SURVEY.md section 5, "race detection / sanitizers": float64 atomics make it order-nondeterministic at
// 1e-16 relative).  DRT_DETERMINISTIC mode accumulates in two's-complement 128-bit FIXED point instead -- (hi: int64, lo: uint64), one
// unit = 2^-kFxFrac -- with integer atomics: integer addition is associative and commutative, so the sum is the same bit pattern whatever
// order the contributions arrive in, whichever wave, workgroup, stream or graph replay produced them.  No sort, no segmented pass.
//   * a contribution |x| >= 2^-(kFxFrac - 52) = 3.7e-9 is represented EXACTLY (all 53 mantissa bits above the unit); a smaller one is
//     truncated toward zero at 2^-80 = 8.3e-25 absolute -- far below the rounding of the float64 sum it replaces;
//   * |x| >= 2^kFxHuge = 7e13, infinities and NaN do not enter the integer sum: they set sticky flags (FxFlag*) in a side word per
//     value and the converted result is +-inf / NaN accordingly (limit_hook turns those into +-1 / 0 as it does for the float64 sum);
//   * the sum itself holds |sum| < 2^47 with > 2^33 such contributions before it could wrap;
//   * fx_to_double rounds the exact integer sum ONCE, to nearest-even: the result is the correctly rounded value of the exact sum of the
//     (truncated) contributions -- usually closer to the true sum than any float64 summation order.
// Everything here is plain C++ (host + device) except the atomics: tests/hostsim checks conversion and rounding against Python integers.
#pragma once
#include "drt_common.h"

namespace drt {

constexpr int kFxFrac = 80;                 // fractional bits
constexpr int kFxHuge = 46;                 // |x| >= 2^46 is treated as infinite
enum : uint32_t { kFxNaN = 1u, kFxPosInf = 2u, kFxNegInf = 4u };

struct Fx128 {
    int64_t hi;
    uint64_t lo;
};

DRT_HD Fx128 fx_add(Fx128 a, Fx128 b) {
    Fx128 r;
    r.lo = a.lo + b.lo;
    r.hi = (int64_t)((uint64_t)a.hi + (uint64_t)b.hi + (r.lo < a.lo ? 1u : 0u));
    return r;
}
DRT_HD Fx128 fx_neg(Fx128 a) {
    Fx128 r;
    r.lo = ~a.lo + 1u;
    r.hi = (int64_t)(~(uint64_t)a.hi + (r.lo == 0 ? 1u : 0u));
    return r;
}

// x -> fixed point (truncation of the magnitude toward zero); returns the sticky flags of a value that does not enter the sum (then r = 0).
DRT_HD uint32_t fx_from_double(double x, Fx128& r) {
    r.hi = 0; r.lo = 0;
    uint64_t u;
    memcpy(&u, &x, 8);
    const bool neg = (u >> 63) != 0;
    const int be = (int)((u >> 52) & 0x7FFu);
    uint64_t m = u & 0xFFFFFFFFFFFFFull;
    if (be == 0x7FF) return m ? kFxNaN : (neg ? kFxNegInf : kFxPosInf);
    if (be == 0) return 0u;                                 // zero and float64 subnormals (< 2^-1022): below the unit by far
    m |= 1ull << 52;                                        // value = m * 2^(be - 1075)
    const int sh = be - 1075 + kFxFrac;                      // magnitude in units = m << sh  (sh may be negative)
    if (be - 1023 >= kFxHuge) return neg ? kFxNegInf : kFxPosInf;
    Fx128 mag{0, 0};
    if (sh >= 64) { mag.hi = (int64_t)(m << (sh - 64)); }     // (sh <= kFxHuge - 1 - 52 + kFxFrac = 73: m << 9 fits 62 bits)
    else if (sh > 0) { mag.lo = m << sh; mag.hi = (int64_t)(m >> (64 - sh)); }
    else if (sh > -53) { mag.lo = m >> (-sh); }
    r = neg ? fx_neg(mag) : mag;
    return 0u;
}

DRT_HD int fx_clz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return v ? __clzll((long long)v) : 64;
#else
    return v ? __builtin_clzll(v) : 64;
#endif
}

// The exact integer sum, rounded once (nearest, ties to even) to float64; `flags` = OR of the sticky flags seen.
DRT_HD double fx_to_double(Fx128 a, uint32_t flags) {
    if (flags) {
        const bool pinf = (flags & kFxPosInf) != 0, ninf = (flags & kFxNegInf) != 0;
        if ((flags & kFxNaN) || (pinf && ninf)) { uint64_t q = 0x7FF8000000000000ull; double d; memcpy(&d, &q, 8); return d; }
        uint64_t q = pinf ? 0x7FF0000000000000ull : 0xFFF0000000000000ull; double d; memcpy(&d, &q, 8); return d;
    }
    const bool neg = a.hi < 0;
    const Fx128 mag = neg ? fx_neg(a) : a;
    const uint64_t mh = (uint64_t)mag.hi, ml = mag.lo;
    if (mh == 0 && ml == 0) return 0.0;
    const int p = mh ? 127 - fx_clz64(mh) : 63 - fx_clz64(ml);          // position of the leading one
    uint64_t top;                                                       // the 53 leading bits (fewer when p < 52: then exact)
    bool round_bit = false, sticky = false;
    if (p <= 52) {
        top = ml;                                                       // exact
    } else {
        const int s = p - 52;                                           // bits dropped
        // (mh:ml) >> s
        top = s >= 64 ? (mh >> (s - 64)) : ((ml >> s) | (s ? (mh << (64 - s)) : 0));
        const int rb = s - 1;                                           // position of the round bit
        round_bit = rb >= 64 ? ((mh >> (rb - 64)) & 1u) != 0 : ((ml >> rb) & 1u) != 0;
        if (rb >= 64) sticky = ml != 0 || (rb > 64 && (mh & ((1ull << (rb - 64)) - 1ull)) != 0);
        else sticky = rb > 0 && (ml & ((1ull << rb) - 1ull)) != 0;
        top &= (1ull << 53) - 1ull;
        if (round_bit && (sticky || (top & 1u))) ++top;                 // may carry into bit 53: still an exact double below
    }
    double d = (double)top;                                             // exact: top <= 2^53
    const int e = (p <= 52 ? 0 : p - 52) - kFxFrac;
    d = ldexp(d, e);
    return neg ? -d : d;
}

// A thread's own running sum (loss terms): exact, so the order of the items a thread happens to get does not matter.
struct FxAcc {
    Fx128 v{0, 0};
    uint32_t flags = 0;
    DRT_HD void add(double x) { Fx128 t; flags |= fx_from_double(x, t); v = fx_add(v, t); }
};

// One accumulator cell per float64 value, as the caller allocates it (DRT_FX_BYTES_PER_VALUE = 24 bytes each, zero-filled): the 128-bit sum
// and the sticky flags.  A kernel that is handed `double* grad` in deterministic mode indexes the SAME element numbers in cells.
struct FxCell {
    int64_t hi;
    uint64_t lo;
    uint64_t flags;
};
constexpr int kFxBytesPerValue = 24;
static_assert(sizeof(FxCell) == kFxBytesPerValue, "DRT_FX_BYTES_PER_VALUE of include/drt_hip.h");

#if defined(__HIPCC__)
__device__ __forceinline__ void fx_atomic_add(FxCell* c, Fx128 t, uint32_t f) {
    if (f) { atomicOr(reinterpret_cast<unsigned long long*>(&c->flags), (unsigned long long)f); return; }
    if (t.hi == 0 && t.lo == 0) return;
    unsigned long long* lo = reinterpret_cast<unsigned long long*>(&c->lo);
    unsigned long long* hi = reinterpret_cast<unsigned long long*>(&c->hi);
    unsigned long long carry = 0;
    if (t.lo) {
        const unsigned long long old = atomicAdd(lo, (unsigned long long)t.lo);
        carry = old + (unsigned long long)t.lo < old ? 1ull : 0ull;      // this addition is the one that wrapped the low word: it owns the carry
    }
    const unsigned long long h = (unsigned long long)t.hi + carry;
    if (h) atomicAdd(hi, h);
}
__device__ __forceinline__ void fx_atomic_add(FxCell* c, double x) {
    Fx128 t;
    const uint32_t f = fx_from_double(x, t);
    fx_atomic_add(c, t, f);
}
// the 64 lanes' sums, exactly, in every lane
__device__ __forceinline__ FxAcc fx_wave_sum(FxAcc a) {
    for (int off = 32; off >= 1; off >>= 1) {
        Fx128 o;
        o.hi = __shfl_xor((long long)a.v.hi, off);
        o.lo = (uint64_t)__shfl_xor((long long)a.v.lo, off);
        a.v = fx_add(a.v, o);
        a.flags |= (uint32_t)__shfl_xor((int)a.flags, off);
    }
    return a;
}
#endif

}  // namespace drt
