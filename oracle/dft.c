/*
 * oracle/dft.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restates the Plonky3 p3-dft / p3-matrix semantics the reference calls
 * (the crate is not on disk):
 *   fri/src/two_adic_pcs.rs:233-241  commit: coset_lde_batch(evals, log_blowup, 31/shift)
 *                                    .bit_reverse_rows().to_row_major_matrix()
 * Two independent implementations: a naive O(n^2) DFT (definition) and a radix-2 NTT; the
 * tests check one against the other.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* out[k][c] = sum_j in[j][c] * w^(jk), w = omega_n (or omega_n^-1, scaled by 1/n, if inverse) */
void ts_or_naive_dft(const uint32_t* in, uint32_t* out, size_t n, size_t w, int inverse) {
    unsigned log_n = ts_log2_strict(n);
    uint32_t g = bb_two_adic_generator(log_n);
    if (inverse) g = bb_inv(g);
    uint32_t ninv = bb_inv((uint32_t)(n % BB_P));
    for (size_t k = 0; k < n; k++) {
        uint32_t gk = bb_pow(g, k);
        for (size_t c = 0; c < w; c++) {
            uint32_t acc = 0, x = 1;
            for (size_t j = 0; j < n; j++) {
                acc = bb_add(acc, bb_mul(in[j * w + c], x));
                x = bb_mul(x, gk);
            }
            out[k * w + c] = inverse ? bb_mul(acc, ninv) : acc;
        }
    }
}

void ts_or_bit_reverse_rows(uint32_t* m, size_t h, size_t w) {
    unsigned bits = ts_log2_strict(h);
    uint32_t* tmp = (uint32_t*)malloc(w * sizeof(uint32_t));
    for (size_t i = 0; i < h; i++) {
        size_t j = ts_bitrev(i, bits);
        if (i < j) {
            memcpy(tmp, m + i * w, w * 4);
            memcpy(m + i * w, m + j * w, w * 4);
            memcpy(m + j * w, tmp, w * 4);
        }
    }
    free(tmp);
}

/* in-place radix-2 decimation-in-time on whole rows: natural in, natural out */
void ts_or_dft_batch(uint32_t* m, size_t n, size_t w, int inverse) {
    if (n == 1) return;
    unsigned log_n = ts_log2_strict(n);
    uint32_t g = bb_two_adic_generator(log_n);
    if (inverse) g = bb_inv(g);
    uint32_t* tw = (uint32_t*)malloc((n / 2) * sizeof(uint32_t));
    tw[0] = 1;
    for (size_t i = 1; i < n / 2; i++) tw[i] = bb_mul(tw[i - 1], g);
    ts_or_bit_reverse_rows(m, n, w);
    for (unsigned s = 0; s < log_n; s++) {
        size_t half = (size_t)1 << s;
        size_t step = n >> (s + 1);
#pragma omp parallel for schedule(static) if (n * w > 65536)
        for (size_t blk = 0; blk < n / (2 * half); blk++) {
            size_t base = blk * 2 * half;
            for (size_t j = 0; j < half; j++) {
                uint32_t t = tw[j * step];
                uint32_t* a = m + (base + j) * w;
                uint32_t* b = m + (base + j + half) * w;
                for (size_t c = 0; c < w; c++) {
                    uint32_t u = a[c], v = bb_mul(b[c], t);
                    a[c] = bb_add(u, v);
                    b[c] = bb_sub(u, v);
                }
            }
        }
    }
    if (inverse) {
        uint32_t ninv = bb_inv((uint32_t)(n % BB_P));
#pragma omp parallel for schedule(static) if (n * w > 65536)
        for (size_t i = 0; i < n * w; i++) m[i] = bb_mul(m[i], ninv);
    }
    free(tw);
}

/* App. A.5: per column c = iDFT_n(col); out row j = sum_k c_k (shift * omega_N^j)^k, natural j */
void ts_or_coset_lde_batch(const uint32_t* evals, size_t n, size_t w, unsigned added_bits,
                           uint32_t shift, uint32_t* out) {
    size_t N = n << added_bits;
    memset(out, 0, N * w * sizeof(uint32_t));
    memcpy(out, evals, n * w * sizeof(uint32_t));
    ts_or_dft_batch(out, n, w, 1);
    uint32_t sk = 1;
    for (size_t k = 0; k < n; k++) {
        for (size_t c = 0; c < w; c++) out[k * w + c] = bb_mul(out[k * w + c], sk);
        sk = bb_mul(sk, shift);
    }
    ts_or_dft_batch(out, N, w, 0);
}

/* fri/src/two_adic_pcs.rs:233-241 */
void ts_or_commit_lde(const uint32_t* evals, unsigned log_n, size_t w, uint32_t domain_shift,
                      unsigned log_blowup, uint32_t* out) {
    size_t n = (size_t)1 << log_n;
    uint32_t shift = bb_mul(BB_GENERATOR, bb_inv(domain_shift)); /* :235 */
    ts_or_coset_lde_batch(evals, n, w, log_blowup, shift, out);
    ts_or_bit_reverse_rows(out, n << log_blowup, w);
}

/* Lagrange evaluation straight from the definition: the unique degree<n polynomial with
 * p(domain_shift * omega_n^i) = col[i], evaluated at base point x (x not in the domain). */
uint32_t ts_or_eval_interpolant_naive(const uint32_t* col, size_t n, uint32_t domain_shift,
                                      uint32_t x) {
    unsigned log_n = ts_log2_strict(n);
    uint32_t g = bb_two_adic_generator(log_n);
    uint32_t acc = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t xi = bb_mul(domain_shift, bb_pow(g, i));
        uint32_t num = 1, den = 1;
        for (size_t j = 0; j < n; j++) {
            if (j == i) continue;
            uint32_t xj = bb_mul(domain_shift, bb_pow(g, j));
            num = bb_mul(num, bb_sub(x, xj));
            den = bb_mul(den, bb_sub(xi, xj));
        }
        acc = bb_add(acc, bb_mul(col[i], bb_mul(num, bb_inv(den))));
    }
    return acc;
}
