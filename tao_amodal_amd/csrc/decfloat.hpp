// Decimal text -> binary64, correctly rounded, for host and device code: the
// value Python's float() / strtod give, bit for bit (what json.load hands the
// reference: lvis_amodal/results.py:29-30, tools/eval_on_tao_amodal.py:127-128).
//
//   * Clinger's fast path: at most 15 significant digits and |exponent| <= 22 --
//     the digits and the power of ten are exact doubles, ONE IEEE multiplication
//     or division rounds correctly;
//   * the Eisel-Lemire algorithm for everything else up to 19 significant digits
//     (D. Lemire, "Number parsing at a gigabyte per second", SPE 2021): a 64 x 128
//     bit product with a truncated power of five (pow5_128.inc, tools/
//     gen_pow5_table.py);
//   * dec_to_double() answers false where neither decides (more than 19 digits,
//     the algorithm's rare undecided products): the caller hands the text to
//     the host's std::from_chars (csrc/ingest.cpp).
//
// tests/test_decfloat.py drives the host build against Python's float() on
// millions of strings (repr of random doubles, halfway cases, subnormals, the
// edges of both paths).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define DECF_HD __host__ __device__ __forceinline__
#else
#define DECF_HD inline
#endif

namespace decf {

#define DECF_Q_MIN (-342)
#define DECF_Q_MAX 308

#if defined(__HIP_DEVICE_COMPILE__)
__device__ const uint64_t pow5_dev[2 * (DECF_Q_MAX - DECF_Q_MIN + 1)] = {
#include "pow5_128.inc"
};
#define DECF_POW5(i) pow5_dev[i]
#else
static const uint64_t pow5_host[2 * (DECF_Q_MAX - DECF_Q_MIN + 1)] = {
#include "pow5_128.inc"
};
#define DECF_POW5(i) pow5_host[i]
#endif

struct U128 {
    uint64_t lo, hi;
};

DECF_HD U128 mul64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return U128{a * b, __umul64hi(a, b)};
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    return U128{(uint64_t)p, (uint64_t)(p >> 64)};
#endif
}

DECF_HD int clz64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

DECF_HD double from_bits(uint64_t b)
{
    union {
        uint64_t u;
        double d;
    } v;
    v.u = b;
    return v.d;
}

// w * 10^q, w != 0 an integer of up to 64 bits: false = undecided
DECF_HD bool eisel_lemire(uint64_t w, int64_t q, uint64_t &bits)
{
    if (q < DECF_Q_MIN) {          // below the smallest subnormal's half: 0
        bits = 0;
        return true;
    }
    if (q > DECF_Q_MAX) {          // above the largest double: inf
        bits = 0x7ffull << 52;
        return true;
    }
    const int lz = clz64(w);
    w <<= lz;
    const int idx = 2 * (int)(q - DECF_Q_MIN);
    U128 p = mul64(w, DECF_POW5(idx));
    const uint64_t precision_mask = 0xFFFFFFFFFFFFFFFFull >> 55;     // mantissa + 3 bits
    if ((p.hi & precision_mask) == precision_mask) {
        const U128 p2 = mul64(w, DECF_POW5(idx + 1));
        p.lo += p2.hi;
        if (p2.hi > p.lo) p.hi++;
    }
    // the truncated product may hide a carry where the low word is all ones
    // (outside -27 <= q <= 55, where the power of five is exact): undecided
    if (p.lo == 0xFFFFFFFFFFFFFFFFull && !(q >= -27 && q <= 55)) return false;
    const int upperbit = (int)(p.hi >> 63);
    uint64_t mantissa = p.hi >> (upperbit + 64 - 52 - 3);
    // floor(log2(10^q)) + 63: 217706 / 2^16 = log2(10) to the precision needed
    const int64_t power = ((217706 * q) >> 16) + 63;
    int64_t power2 = power + upperbit - lz + 1023;
    if (power2 <= 0) {             // subnormal, or zero
        if (-power2 + 1 >= 64) {
            bits = 0;
            return true;
        }
        mantissa >>= -power2 + 1;
        mantissa += mantissa & 1;
        mantissa >>= 1;
        power2 = mantissa < (1ull << 52) ? 0 : 1;
        bits = mantissa + ((uint64_t)power2 << 52) - (power2 ? (1ull << 52) : 0);
        // (mantissa == 2^52 is the smallest normal: exponent 1, fraction 0)
        return true;
    }
    // halfway between two doubles: round to even needs the exact product
    if (p.lo <= 1 && q >= -4 && q <= 23 && (mantissa & 3) == 1) {
        if ((mantissa << (upperbit + 64 - 52 - 3)) == p.hi) mantissa &= ~1ull;
    }
    mantissa += mantissa & 1;
    mantissa >>= 1;
    if (mantissa >= (2ull << 52)) {
        mantissa = 1ull << 52;
        power2++;
    }
    mantissa &= ~(1ull << 52);
    if (power2 >= 0x7ff) {
        bits = 0x7ffull << 52;
        return true;
    }
    bits = mantissa | ((uint64_t)power2 << 52);
    return true;
}

// (-1)^neg * m * 10^e10, m the number's first n_digits significant digits (no
// digit dropped): false = undecided (the caller's slow path)
DECF_HD bool dec_to_double(bool neg, uint64_t m, int n_digits, int64_t e10, double &out)
{
    uint64_t bits;
    if (m == 0) {
        bits = 0;
    } else if (n_digits <= 15 && e10 >= -22 && e10 <= 22) {
        // exact operands, one correctly rounded IEEE operation
        const double P10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,
                                1e8,  1e9,  1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        const double x = e10 < 0 ? (double)m / P10[-e10] : (double)m * P10[e10];
        out = neg ? -x : x;
        return true;
    } else {
        if (n_digits > 19) return false;
        if (!eisel_lemire(m, e10, bits)) return false;
    }
    out = from_bits(bits | (neg ? 1ull << 63 : 0));
    return true;
}

// A JSON number at s[0 .. n): -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)? and
// nothing else.  Returns the number of bytes used (0: not such a number, or one
// this code does not decide -- more than 19 significant digits, an exponent
// beyond four digits, an undecided product).  GET(i) = byte i of the text.
template <class Get>
DECF_HD int parse_json_number(Get get, int n, double &out, bool *is_int = nullptr,
                              int64_t *int_val = nullptr)
{
    int i = 0;
    const bool neg = n > 0 && get(0) == '-';
    if (neg) i++;
    if (i >= n) return 0;
    uint64_t m = 0;
    int nd = 0;                 // significant digits in m
    int64_t e10 = 0;
    bool lead = true;           // only zeros so far
    const int i_int = i;
    unsigned c = (unsigned)get(i) - '0';
    if (c > 9u) return 0;
    if (c == 0) {
        i++;
        if (i < n && (unsigned)get(i) - '0' <= 9u) return 0;      // 01: not JSON
    } else {
        while (i < n && (c = (unsigned)get(i) - '0') <= 9u) {
            if (nd >= 19) return 0;
            m = m * 10 + c;
            nd++;
            lead = false;
            i++;
        }
    }
    bool integer = true;
    (void)i_int;
    if (i < n && get(i) == '.') {
        integer = false;
        i++;
        const int f0 = i;
        while (i < n && (c = (unsigned)get(i) - '0') <= 9u) {
            if (c != 0 || !lead) {
                if (nd >= 19) return 0;
                m = m * 10 + c;
                nd++;
                lead = false;
            }
            e10--;
            i++;
        }
        if (i == f0) return 0;                                    // "1." is not JSON
    }
    if (i < n && (get(i) == 'e' || get(i) == 'E')) {
        integer = false;
        i++;
        bool eneg = false;
        if (i < n && (get(i) == '+' || get(i) == '-')) {
            eneg = get(i) == '-';
            i++;
        }
        const int x0 = i;
        int64_t ex = 0;
        while (i < n && (c = (unsigned)get(i) - '0') <= 9u) {
            if (i - x0 >= 4) return 0;                            // (left to the slow path)
            ex = ex * 10 + c;
            i++;
        }
        if (i == x0) return 0;
        e10 += eneg ? -ex : ex;
    }
    if (is_int) {
        // an integer of at most 18 digits: exact in int64 (ids)
        *is_int = integer && nd <= 18;
        if (*is_int && int_val) *int_val = neg ? -(int64_t)m : (int64_t)m;
    }
    if (!dec_to_double(neg, m, nd, e10, out)) return 0;
    return i;
}

}  // namespace decf
