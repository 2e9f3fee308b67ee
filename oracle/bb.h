/*
 * oracle/bb.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * BabyBear (p = 15*2^27 + 1) and its degree-4 binomial extension F[x]/(x^4 - 11),
 * in plain canonical representation (every value in [0, p), products via 64-bit `%`).
 * Deliberately NOT Montgomery, so that it is independent of the HIP kernels.
 *
 * Restates (semantics only; Plonky3 @72b2fc16 is not on disk):
 *   reference basic/src/field/mod.rs:45      MOD = 0x78000001
 *   reference basic/src/field/mod.rs:48-63   as_u32_vec: canonical u32, EF4 = [c0,c1,c2,c3]
 *   reference common/src/lib.rs:5-35         AsU32Vec
 *   reference uni-stark/src/scripts/bf_unistark.rs:42-43   31^-1 = 64944062 => generator 31
 *   two_adic_generator(27) = 31^15 = 0x1a427a41 (arithmetic; App. A.1)
 */
#ifndef TS_ORACLE_BB_H
#define TS_ORACLE_BB_H

#include <stdint.h>
#include <stddef.h>

#define BB_P 2013265921u /* 0x78000001 */
#define BB_GENERATOR 31u
#define BB_TWO_ADICITY 27
#define BB_TWO_ADIC_GEN_27 0x1a427a41u
#define EF4_W 11u /* x^4 = 11 */

static inline uint32_t bb_add(uint32_t a, uint32_t b) {
    uint32_t s = a + b; /* < 2^32 since a,b < 2^31 */
    return s >= BB_P ? s - BB_P : s;
}
static inline uint32_t bb_sub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + BB_P - b; }
static inline uint32_t bb_neg(uint32_t a) { return a ? BB_P - a : 0; }
static inline uint32_t bb_mul(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) % BB_P);
}
static inline uint32_t bb_pow(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = bb_mul(r, a);
        a = bb_mul(a, a);
        e >>= 1;
    }
    return r;
}
static inline uint32_t bb_inv(uint32_t a) { return bb_pow(a, BB_P - 2); }
/* generator of the order-2^bits subgroup (App. A.1) */
static inline uint32_t bb_two_adic_generator(unsigned bits) {
    return bb_pow(BB_TWO_ADIC_GEN_27, 1ull << (BB_TWO_ADICITY - bits));
}

typedef struct {
    uint32_t c[4];
} ef4;

static inline ef4 ef4_zero(void) { ef4 r = {{0, 0, 0, 0}}; return r; }
static inline ef4 ef4_one(void) { ef4 r = {{1, 0, 0, 0}}; return r; }
static inline ef4 ef4_from_base(uint32_t a) { ef4 r = {{a, 0, 0, 0}}; return r; }
static inline int ef4_eq(ef4 a, ef4 b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
static inline int ef4_is_zero(ef4 a) { return !(a.c[0] | a.c[1] | a.c[2] | a.c[3]); }
static inline ef4 ef4_add(ef4 a, ef4 b) {
    ef4 r;
    for (int i = 0; i < 4; i++) r.c[i] = bb_add(a.c[i], b.c[i]);
    return r;
}
static inline ef4 ef4_sub(ef4 a, ef4 b) {
    ef4 r;
    for (int i = 0; i < 4; i++) r.c[i] = bb_sub(a.c[i], b.c[i]);
    return r;
}
static inline ef4 ef4_neg(ef4 a) {
    ef4 r;
    for (int i = 0; i < 4; i++) r.c[i] = bb_neg(a.c[i]);
    return r;
}
static inline ef4 ef4_mul_base(ef4 a, uint32_t b) {
    ef4 r;
    for (int i = 0; i < 4; i++) r.c[i] = bb_mul(a.c[i], b);
    return r;
}
static inline ef4 ef4_add_base(ef4 a, uint32_t b) { a.c[0] = bb_add(a.c[0], b); return a; }
static inline ef4 ef4_sub_base(ef4 a, uint32_t b) { a.c[0] = bb_sub(a.c[0], b); return a; }
/* schoolbook product reduced by x^4 = 11 */
static inline ef4 ef4_mul(ef4 a, ef4 b) {
    uint32_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) t[i + j] = bb_add(t[i + j], bb_mul(a.c[i], b.c[j]));
    ef4 r;
    r.c[0] = bb_add(t[0], bb_mul(EF4_W, t[4]));
    r.c[1] = bb_add(t[1], bb_mul(EF4_W, t[5]));
    r.c[2] = bb_add(t[2], bb_mul(EF4_W, t[6]));
    r.c[3] = t[3];
    return r;
}
static inline ef4 ef4_pow(ef4 a, uint64_t e) {
    ef4 r = ef4_one();
    while (e) {
        if (e & 1) r = ef4_mul(r, a);
        a = ef4_mul(a, a);
        e >>= 1;
    }
    return r;
}
/* Inverse through the tower F < F[y]/(y^2-11) < F[x]/(x^2-y):
 * a = A + xB, A = a0 + a2 y, B = a1 + a3 y;  a*(A - xB) = A^2 - y B^2 = C in F[y];
 * C*(c0 - c1 y) = c0^2 - 11 c1^2 in F. */
static inline ef4 ef4_inv(ef4 a) {
    uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
    /* A^2 = (a0^2 + 11 a2^2) + (2 a0 a2) y */
    uint32_t A2_0 = bb_add(bb_mul(a0, a0), bb_mul(EF4_W, bb_mul(a2, a2)));
    uint32_t A2_1 = bb_mul(2, bb_mul(a0, a2));
    /* B^2 = (a1^2 + 11 a3^2) + (2 a1 a3) y ;  y*B^2 = 11*(2 a1 a3) + (a1^2 + 11 a3^2) y */
    uint32_t B2_0 = bb_add(bb_mul(a1, a1), bb_mul(EF4_W, bb_mul(a3, a3)));
    uint32_t B2_1 = bb_mul(2, bb_mul(a1, a3));
    uint32_t c0 = bb_sub(A2_0, bb_mul(EF4_W, B2_1));
    uint32_t c1 = bb_sub(A2_1, B2_0);
    uint32_t nrm = bb_sub(bb_mul(c0, c0), bb_mul(EF4_W, bb_mul(c1, c1)));
    uint32_t ni = bb_inv(nrm);
    /* (A - xB) * (c0 - c1 y) * ni ; with y = x^2 */
    ef4 conj = {{a0, bb_neg(a1), a2, bb_neg(a3)}};
    ef4 cc = {{c0, 0, bb_neg(c1), 0}};
    return ef4_mul_base(ef4_mul(conj, cc), ni);
}
static inline ef4 ef4_div(ef4 a, ef4 b) { return ef4_mul(a, ef4_inv(b)); }

static inline unsigned ts_log2_strict(size_t n) {
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    return k;
}
static inline size_t ts_bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

#endif
