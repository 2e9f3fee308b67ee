// BabyBear / EF4 arithmetic shared by host code and HIP kernels.
//
// Field constants follow reference basic/src/field/mod.rs:45 (MOD = 0x78000001) and SURVEY.md
// App. A.1/A.2 (generator 31, two-adic generator 0x1a427a41, EF4 = F[x]/(x^4 - 11)).
//
// Representation policy (DESIGN.md "Data layout"): everything that lives in HBM or crosses the
// C ABI is CANONICAL u32 (reference as_u32_vec, basic/src/field/mod.rs:48-63).  Montgomery form
// (R = 2^32) is used only for constants (twiddles, challenge powers) and inside kernels:
// mont_mul(canonical, montgomery) == canonical product, so streaming data never needs converting.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TS_HD __host__ __device__ __forceinline__
#else
#define TS_HD inline
#endif

namespace ts {

constexpr uint32_t P = 0x78000001u;
constexpr uint32_t P_INV = 0x88000001u;      // p^-1 mod 2^32
constexpr uint32_t P_NEG_INV = 0x77ffffffu;  // -p^-1 mod 2^32
constexpr uint32_t R_MOD_P = 0x0ffffffeu;    // 2^32 mod p  (Montgomery form of 1)
constexpr uint32_t R2_MOD_P = 0x45dddde3u;   // 2^64 mod p  (checked by tests/test_host_field)
constexpr uint32_t GENERATOR = 31u;
constexpr uint32_t TWO_ADIC_GEN_27 = 0x1a427a41u;
constexpr uint32_t EF_W = 11u;

TS_HD uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
// a, b in [0, p): if the true result is out of range, the wrapped value is the larger one, so an
// unsigned min picks the canonical representative (3 full-rate VALU ops, no compare/select pair)
TS_HD uint32_t add(uint32_t a, uint32_t b) {
    uint32_t s = a + b;
    return umin32(s, s - P);
}
TS_HD uint32_t sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;
    return umin32(d, d + P);
}
TS_HD uint32_t neg(uint32_t a) { return a ? P - a : 0u; }

TS_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// Montgomery reduction of t < p * 2^32: returns t * 2^-32 mod p in [0, p).
// (Measured on gfx950 with the SQ counters, profiles/r02_alu_loops_sq.txt: in register-resident loops of
// this very code SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU quad-cycles, i.e. EVERY integer VALU
// instruction -- VOP2 add/sub/min and VOP3 v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 alike -- holds
// its SIMD for one 4-cycle issue slot.  The cost of a formulation is therefore its instruction COUNT;
// an earlier note here priced VOP3 ops at ~2.5 slots from wall-clock microbenchmarks, which the
// counters refuted.)
// Additive form: m = -t p^-1 mod 2^32 makes t + m p divisible by 2^32; the quotient is < 2p, so one
// sub + min finishes.  On the device that is v_mul_lo_u32, v_mad_u64_u32 (product and 64-bit add in
// one instruction), v_sub, v_min: one instruction fewer than the subtractive form (mul_lo, mul_hi, sub,
// add, min).  t + m p < p 2^32 + 2^32 p < 2^64.
TS_HD uint32_t mont_reduce(uint64_t t) {
    uint32_t m = (uint32_t)t * P_NEG_INV;
    uint32_t r = (uint32_t)((t + (uint64_t)m * P) >> 32);
    return umin32(r, r - P);
}
// [0, 2p) -> [0, p)
TS_HD uint32_t red2p(uint32_t x) { return umin32(x, x - P); }
// Montgomery product without the final range correction: a * b * 2^-32 mod p in [0, 2p), for
// a * b < p * 2^32 (e.g. a < 2p, b < p).  The NTT kernels keep their butterflies in [0, 2p).
TS_HD uint32_t mont_mul_lazy(uint32_t a, uint32_t b) {
    const uint64_t t = (uint64_t)a * b;
    const uint32_t m = (uint32_t)t * P_NEG_INV;
    return (uint32_t)((t + (uint64_t)m * P) >> 32);
}
// a * b * 2^-32 mod p.  Needs a*b < p*2^32 (true if either operand is < p).
TS_HD uint32_t mont_mul(uint32_t a, uint32_t b) { return mont_reduce((uint64_t)a * b); }
TS_HD uint32_t to_mont(uint32_t a) { return mont_mul(a, R2_MOD_P); }
TS_HD uint32_t from_mont(uint32_t a) { return mont_reduce((uint64_t)a); }

// plain (canonical x canonical -> canonical) product; host-side convenience, slow path on device
TS_HD uint32_t mul(uint32_t a, uint32_t b) { return mont_mul(to_mont(a), b); }

TS_HD uint32_t pow_canon(uint32_t a, uint64_t e) {
    uint32_t r = R_MOD_P, am = to_mont(a);
    while (e) {
        if (e & 1) r = mont_mul(r, am);
        am = mont_mul(am, am);
        e >>= 1;
    }
    return from_mont(r);
}
// Montgomery-domain exponentiation / inverse (input and output in Montgomery form)
TS_HD uint32_t mont_pow(uint32_t am, uint64_t e) {
    uint32_t r = R_MOD_P;
    while (e) {
        if (e & 1) r = mont_mul(r, am);
        am = mont_mul(am, am);
        e >>= 1;
    }
    return r;
}
// a^(p-2), p-2 = 0x77ffffff: 4-bit windows 7,7,f,f,f,f,f,f  (31 squarings + 10 products)
TS_HD uint32_t mont_inv(uint32_t a) {
    uint32_t a2 = mont_mul(a, a), a3 = mont_mul(a2, a), a6 = mont_mul(a3, a3);
    uint32_t a7 = mont_mul(a6, a), a14 = mont_mul(a7, a7), a15 = mont_mul(a14, a);
    uint32_t r = a7;
    for (int k = 0; k < 4; k++) r = mont_mul(r, r);
    r = mont_mul(r, a7);
    for (int w = 0; w < 6; w++) {
        for (int k = 0; k < 4; k++) r = mont_mul(r, r);
        r = mont_mul(r, a15);
    }
    return r;
}
TS_HD uint32_t inv_canon(uint32_t a) { return pow_canon(a, P - 2); }
TS_HD uint32_t two_adic_generator(unsigned bits) {
    return pow_canon(TWO_ADIC_GEN_27, 1ull << (27 - bits));
}

// ---------------------------------------------------------------------------------- EF4
// Coefficient order [c0,c1,c2,c3] (reference basic/src/field/mod.rs:58-63).  The struct does not
// record whether the coefficients are canonical or Montgomery: each call site says which.
struct alignas(16) Ef {
    uint32_t c[4];
};

TS_HD Ef ef_zero() { return Ef{{0, 0, 0, 0}}; }
TS_HD Ef ef_add(Ef a, Ef b) {
    return Ef{{add(a.c[0], b.c[0]), add(a.c[1], b.c[1]), add(a.c[2], b.c[2]), add(a.c[3], b.c[3])}};
}
TS_HD Ef ef_sub(Ef a, Ef b) {
    return Ef{{sub(a.c[0], b.c[0]), sub(a.c[1], b.c[1]), sub(a.c[2], b.c[2]), sub(a.c[3], b.c[3])}};
}
TS_HD Ef ef_neg(Ef a) { return Ef{{neg(a.c[0]), neg(a.c[1]), neg(a.c[2]), neg(a.c[3])}}; }
TS_HD bool ef_eq(Ef a, Ef b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
// (a in form X) * (b Montgomery base) -> form X
TS_HD Ef ef_mul_base(Ef a, uint32_t bm) {
    return Ef{{mont_mul(a.c[0], bm), mont_mul(a.c[1], bm), mont_mul(a.c[2], bm), mont_mul(a.c[3], bm)}};
}
// ---- lazy 64-bit accumulation of products of values < p ------------------------------------------
// Invariant between steps: acc < p*2^32 (2^62.91).  A product is < p^2 < 2^61.82, so after two more
// products acc < 2p*2^32 still: its high word is < 2p and one conditional subtraction of p there
// (sub + min) restores the invariant without changing acc mod p.
TS_HD uint64_t lazy_mac(uint64_t acc, uint32_t a, uint32_t b) { return acc + (uint64_t)a * b; }
TS_HD uint64_t lazy_fix(uint64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    // as a two-lane vector: the high word is corrected in place (add, min) and the pair goes straight
    // back into v_mad_u64_u32; written with shifts and an OR the compiler rebuilt the 64-bit value
    // with a zero move and a 64-bit add per call (5 instead of 2 VALU instructions)
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    v2u u = __builtin_bit_cast(v2u, acc);
    u.y = umin32(u.y, u.y - P);
    return __builtin_bit_cast(uint64_t, u);
#else
    uint32_t hi = (uint32_t)(acc >> 32);
    hi = umin32(hi, hi - P);
    return ((uint64_t)hi << 32) | (uint32_t)acc;
#endif
}
// acc < 2p*2^32 -> acc * 2^-32 mod p, canonical
TS_HD uint32_t lazy_finish(uint64_t acc) { return mont_reduce(lazy_fix(acc)); }

// Montgomery product of two EF4 elements: if both are Montgomery the result is Montgomery; if one
// is canonical and the other Montgomery the result is canonical.
// x^4 = 11: with w_k = 11 b_k every output coefficient is ONE sum of four products,
//   r0 = a0 b0 + a1 w3 + a2 w2 + a3 w1,  r1 = a0 b1 + a1 b0 + a2 w3 + a3 w2,  ...
// and four products of values < p stay below 4 p^2 < 2p*2^32 < 2^64: a single lazy reduction per
// coefficient (16 multiply-adds, 3 products for the w_k, 4 reductions: ~55 VALU instructions; the
// previous form reduced ten partial sums and folded by 11 afterwards, ~85).
TS_HD Ef ef_mul(Ef a, Ef b) {
    constexpr uint32_t W_M = (uint32_t)(((uint64_t)EF_W << 32) % P);  // 11 in Montgomery form
    const uint32_t w1 = mont_mul(b.c[1], W_M), w2 = mont_mul(b.c[2], W_M), w3 = mont_mul(b.c[3], W_M);
    const uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
    const uint64_t r0 = lazy_mac(lazy_mac(lazy_mac(lazy_mac(0, a0, b.c[0]), a1, w3), a2, w2), a3, w1);
    const uint64_t r1 = lazy_mac(lazy_mac(lazy_mac(lazy_mac(0, a0, b.c[1]), a1, b.c[0]), a2, w3), a3, w2);
    const uint64_t r2 = lazy_mac(lazy_mac(lazy_mac(lazy_mac(0, a0, b.c[2]), a1, b.c[1]), a2, b.c[0]), a3, w3);
    const uint64_t r3 = lazy_mac(lazy_mac(lazy_mac(lazy_mac(0, a0, b.c[3]), a1, b.c[2]), a2, b.c[1]), a3, b.c[0]);
    return Ef{{lazy_finish(r0), lazy_finish(r1), lazy_finish(r2), lazy_finish(r3)}};
}
TS_HD Ef ef_to_mont(Ef a) { return Ef{{to_mont(a.c[0]), to_mont(a.c[1]), to_mont(a.c[2]), to_mont(a.c[3])}}; }
TS_HD Ef ef_from_mont(Ef a) {
    return Ef{{from_mont(a.c[0]), from_mont(a.c[1]), from_mont(a.c[2]), from_mont(a.c[3])}};
}
TS_HD Ef ef_one_mont() { return Ef{{R_MOD_P, 0, 0, 0}}; }
TS_HD Ef ef_from_base(uint32_t a) { return Ef{{a, 0, 0, 0}}; }

// Inverse split in two halves so that callers can batch the base-field inversion:
//   a^-1 = num * nrm^-1,   nrm in F (Montgomery in -> Montgomery out)
// Tower F < F[y]/(y^2-11) < F[x]/(x^2-y): a = A + xB;  a(A - xB) = A^2 - yB^2 = c0 + c1 y;
// (c0 + c1 y)(c0 - c1 y) = c0^2 - 11 c1^2.
TS_HD void ef_inv_parts(Ef a, Ef& num, uint32_t& nrm) {
    constexpr uint32_t W_M = (uint32_t)(((uint64_t)EF_W << 32) % P);
    uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
    uint32_t A2_0 = add(mont_mul(a0, a0), mont_mul(W_M, mont_mul(a2, a2)));
    uint32_t A2_1 = mont_mul(add(a0, a0), a2);
    uint32_t B2_0 = add(mont_mul(a1, a1), mont_mul(W_M, mont_mul(a3, a3)));
    uint32_t B2_1 = mont_mul(add(a1, a1), a3);
    uint32_t c0 = sub(A2_0, mont_mul(W_M, B2_1));
    uint32_t c1 = sub(A2_1, B2_0);
    nrm = sub(mont_mul(c0, c0), mont_mul(W_M, mont_mul(c1, c1)));
    Ef conj = Ef{{a0, neg(a1), a2, neg(a3)}};
    Ef cc = Ef{{c0, 0, neg(c1), 0}};
    num = ef_mul(conj, cc);
}
TS_HD Ef ef_inv(Ef a) {  // Montgomery in, Montgomery out
    Ef num;
    uint32_t nrm;
    ef_inv_parts(a, num, nrm);
    return ef_mul_base(num, mont_inv(nrm));
}
TS_HD Ef ef_pow(Ef am, uint64_t e) {  // Montgomery
    Ef r = ef_one_mont();
    while (e) {
        if (e & 1) r = ef_mul(r, am);
        am = ef_mul(am, am);
        e >>= 1;
    }
    return r;
}

TS_HD uint32_t bitrev32(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0u;
#else
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

}  // namespace ts
