// Coset LDE on column-major BabyBear columns: inverse NTT -> coset scaling -> forward NTT, with
// the result in bit-reversed row order (reference fri/src/two_adic_pcs.rs:233-241:
// `dft.coset_lde_batch(evals, log_blowup, shift).bit_reverse_rows()`).
//
// Formulation (DESIGN.md "NTT"): block-twiddle radix-2 transform.  Forward stage s (m = 2^s
// blocks, distance t = n/2^(s+1)) applies (a, b) -> (a + w b, a - w b) with ONE twiddle per block,
// w = W[m + blk] = omega_{2m}^bitrev(blk); natural-order input, bit-reversed output, no separate
// twist between passes.  The inverse runs the stages backwards with (A, B) -> (A + B, (A - B)/w).
// Stages whose distance is >= 2^LOG_M are done by the "strided" kernels on LDS tiles of
// [n / 2^LOG_M][T] elements (T consecutive rows => coalesced segments); the remaining LOG_M stages
// by the "contiguous" kernels on 2^LOG_M-element chunks.  Data in HBM is canonical; twiddles are
// Montgomery, so mont_mul(data, twiddle) is canonical.
#include "kernels.hpp"

namespace ts {

constexpr int LOG_M = 12;          // contiguous chunk = 4096 elements = 16 KiB of LDS
constexpr int CHUNK = 1 << LOG_M;
constexpr int NT = 256;            // threads per workgroup
constexpr int TILE_ELEMS = 8192;   // strided tile = 32 KiB of LDS
constexpr int SHIFT_LO_BITS = 10;  // coset scale s^k = hi[k >> 10] * lo[k & 1023]

// ------------------------------------------------------------------ tables
__global__ void k_build_twiddles(uint32_t* __restrict__ W, uint32_t* __restrict__ Winv,
                                 unsigned log_size) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1u << log_size)) return;
    if (idx == 0) {
        W[0] = R_MOD_P;
        Winv[0] = R_MOD_P;
        return;
    }
    unsigned logm = 31 - __clz(idx);
    uint32_t i = idx - (1u << logm);
    // omega_{2m}: generator of the subgroup of order 2^(logm+1)
    uint32_t g27 = to_mont(TWO_ADIC_GEN_27);
    uint32_t g = mont_pow(g27, 1ull << (27 - (logm + 1)));
    uint32_t e = bitrev32(i, logm);
    uint32_t w = mont_pow(g, e);
    W[idx] = w;
    Winv[idx] = mont_inv(w);
}

void launch_build_twiddles(Context& ctx, uint32_t* W, uint32_t* Winv, unsigned log_size) {
    uint32_t n = 1u << log_size;
    TS_LAUNCH(ctx, k_build_twiddles, dim3((n + 255) / 256), dim3(256), 0, W, Winv,
                       log_size);
    TS_HIP(hipGetLastError());
}

// lo[beta][j] = s_beta^j * scale,  hi[beta][j] = s_beta^(j << 10)   (Montgomery)
__global__ void k_build_shift_tables(uint32_t* __restrict__ lo, uint32_t* __restrict__ hi,
                                     uint32_t n_hi, uint32_t shift_mont, unsigned log_N,
                                     unsigned log_blowup, uint32_t scale_mont) {
    uint32_t beta = blockIdx.y;
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_lo = 1u << SHIFT_LO_BITS;
    if (j >= n_lo + n_hi) return;
    // s_beta = shift * omega_N^bitrev_b(beta)
    uint32_t g27 = to_mont(TWO_ADIC_GEN_27);
    uint32_t gN = mont_pow(g27, 1ull << (27 - log_N));
    uint32_t s = mont_mul(shift_mont, mont_pow(gN, bitrev32(beta, log_blowup)));
    if (j < n_lo) {
        lo[beta * n_lo + j] = mont_mul(mont_pow(s, j), scale_mont);
    } else {
        uint32_t jj = j - n_lo;
        hi[beta * n_hi + jj] = mont_pow(s, (uint64_t)jj << SHIFT_LO_BITS);
    }
}

// ------------------------------------------------------------------ transposes
// src row-major [n][w] natural  ->  dst[c][p] = src[bitrev(p)][c]
__global__ void k_transpose_bitrev(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                   unsigned log_n, uint32_t w, uint64_t dst_col_stride) {
    __shared__ uint32_t tile[64][65];
    const unsigned tr = log_n < 6 ? log_n : 6;  // log2 of tile rows
    const uint32_t rows = 1u << tr;
    const uint32_t p0 = blockIdx.x << tr;
    const uint32_t c0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (uint32_t i = ty; i < rows; i += 4) {
        uint32_t r = bitrev32(p0 + i, log_n);
        uint32_t c = c0 + tx;
        if (c < w) tile[i][tx] = src[(uint64_t)r * w + c];
    }
    __syncthreads();
    for (uint32_t cc = ty; cc < 64; cc += 4) {
        uint32_t c = c0 + cc;
        if (c < w && tx < rows) dst[(uint64_t)c * dst_col_stride + p0 + tx] = tile[tx][cc];
    }
}

void launch_transpose_bitrev(Context& ctx, const uint32_t* src, uint32_t* dst, unsigned log_n,
                             uint32_t w, uint64_t dst_col_stride) {
    unsigned tr = log_n < 6 ? log_n : 6;
    dim3 grid(1u << (log_n - tr), (w + 63) / 64);
    TS_LAUNCH(ctx, k_transpose_bitrev, grid, dim3(256), 0, src, dst, log_n, w,
                       dst_col_stride);
    TS_HIP(hipGetLastError());
}

// dst row-major [h][w]  <-  src column-major
__global__ void k_transpose_to_row_major(const uint32_t* __restrict__ src, uint64_t col_stride,
                                         uint32_t* __restrict__ dst, uint64_t h, uint32_t w) {
    __shared__ uint32_t tile[64][65];
    const uint64_t r0 = (uint64_t)blockIdx.x * 64;
    const uint32_t c0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (uint32_t cc = ty; cc < 64; cc += 4) {
        uint32_t c = c0 + cc;
        if (c < w && r0 + tx < h) tile[tx][cc] = src[(uint64_t)c * col_stride + r0 + tx];
    }
    __syncthreads();
    for (uint32_t i = ty; i < 64; i += 4) {
        uint32_t c = c0 + tx;
        if (c < w && r0 + i < h) dst[(r0 + i) * w + c] = tile[i][tx];
    }
}

void launch_transpose_to_row_major(Context& ctx, const uint32_t* src, uint64_t col_stride,
                                   uint32_t* dst, uint64_t h, uint32_t w) {
    dim3 grid((unsigned)((h + 63) / 64), (w + 63) / 64);
    TS_LAUNCH(ctx, k_transpose_to_row_major, grid, dim3(256), 0, src, col_stride,
                       dst, h, w);
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ LDS stage loops
// `s` holds 2^log_len logical elements for each of T = 2^log_T side-by-side transforms, element
// (e, jj) at s[(e << log_T) + jj].  Local stage u corresponds to global stage s_base + u; the
// twiddle of local block blk is W[2^(s_base+u) + (c << u) + blk].
__device__ __forceinline__ void tile_forward(uint32_t* s, unsigned log_len, unsigned log_T,
                                             unsigned s_base, uint32_t c,
                                             const uint32_t* __restrict__ W) {
    if (log_len == 0) return;
    const uint32_t total = 1u << (log_len - 1 + log_T);  // butterflies per stage
    for (unsigned u = 0; u < log_len; u++) {
        const unsigned log_t = log_len - 1 - u;
        const uint32_t wbase = (1u << (s_base + u)) + (c << u);
        for (uint32_t b = threadIdx.x; b < total; b += NT) {
            uint32_t jj = b & ((1u << log_T) - 1);
            uint32_t bb = b >> log_T;
            uint32_t blk = bb >> log_t;
            uint32_t q = bb & ((1u << log_t) - 1);
            uint32_t i0 = ((((blk << 1) << log_t) + q) << log_T) + jj;
            uint32_t i1 = i0 + (1u << (log_t + log_T));
            uint32_t w = W[wbase + blk];
            uint32_t a = s[i0];
            uint32_t v = mont_mul(s[i1], w);
            s[i0] = add(a, v);
            s[i1] = sub(a, v);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void tile_inverse(uint32_t* s, unsigned log_len, unsigned log_T,
                                             unsigned s_base, uint32_t c,
                                             const uint32_t* __restrict__ Winv) {
    if (log_len == 0) return;
    const uint32_t total = 1u << (log_len - 1 + log_T);
    for (int u = (int)log_len - 1; u >= 0; u--) {
        const unsigned log_t = log_len - 1 - u;
        const uint32_t wbase = (1u << (s_base + u)) + (c << u);
        for (uint32_t b = threadIdx.x; b < total; b += NT) {
            uint32_t jj = b & ((1u << log_T) - 1);
            uint32_t bb = b >> log_T;
            uint32_t blk = bb >> log_t;
            uint32_t q = bb & ((1u << log_t) - 1);
            uint32_t i0 = ((((blk << 1) << log_t) + q) << log_T) + jj;
            uint32_t i1 = i0 + (1u << (log_t + log_T));
            uint32_t w = Winv[wbase + blk];
            uint32_t a = s[i0], v = s[i1];
            s[i0] = add(a, v);
            s[i1] = mont_mul(sub(a, v), w);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ inverse NTT
// Contiguous pass: chunk `c` of column `col` (chunk length = 2^log_len, log_len = min(log_n, LOG_M)),
// global stages log_n-1 .. log_n-log_len.  If it is the only pass, also scales by 1/n.
__global__ void __launch_bounds__(NT)
k_intt_contig(uint32_t* __restrict__ data, uint64_t col_stride, unsigned log_n, unsigned log_len,
              const uint32_t* __restrict__ Winv, uint32_t scale_mont, int apply_scale) {
    __shared__ uint32_t s[CHUNK];
    const uint32_t c = blockIdx.x;
    uint32_t* g = data + (uint64_t)blockIdx.y * col_stride + ((uint64_t)c << log_len);
    const uint32_t len = 1u << log_len;
    for (uint32_t i = threadIdx.x; i < len; i += NT) s[i] = g[i];
    __syncthreads();
    tile_inverse(s, log_len, 0, log_n - log_len, c, Winv);
    if (apply_scale) {
        for (uint32_t i = threadIdx.x; i < len; i += NT) g[i] = mont_mul(s[i], scale_mont);
    } else {
        for (uint32_t i = threadIdx.x; i < len; i += NT) g[i] = s[i];
    }
}

// Strided pass: global stages sA-1 .. 0 (sA = log_n - LOG_M) on the elements
// {j1 * 2^LOG_M + j2 : j1 < 2^sA} for T consecutive j2; ends with the 1/n scaling.
__global__ void __launch_bounds__(NT)
k_intt_strided(uint32_t* __restrict__ data, uint64_t col_stride, unsigned sA, unsigned log_T,
               const uint32_t* __restrict__ Winv, uint32_t scale_mont) {
    __shared__ uint32_t s[TILE_ELEMS];
    const uint32_t j2_0 = blockIdx.x << log_T;
    uint32_t* g = data + (uint64_t)blockIdx.y * col_stride + j2_0;
    const uint32_t total = 1u << (sA + log_T);
    const uint32_t tmask = (1u << log_T) - 1;
    for (uint32_t i = threadIdx.x; i < total; i += NT)
        s[i] = g[((uint64_t)(i >> log_T) << LOG_M) + (i & tmask)];
    __syncthreads();
    tile_inverse(s, sA, log_T, 0, 0, Winv);
    for (uint32_t i = threadIdx.x; i < total; i += NT)
        g[((uint64_t)(i >> log_T) << LOG_M) + (i & tmask)] = mont_mul(s[i], scale_mont);
}

// ------------------------------------------------------------------ forward coset NTT
// Strided pass of coset beta: coefficient k = j1 * 2^LOG_M + j2 is scaled by s_beta^k, then global
// stages 0 .. sA-1; result goes to block beta of `out`.
__global__ void __launch_bounds__(NT)
k_lde_fwd_strided(const uint32_t* __restrict__ coef, uint64_t in_col_stride,
                  uint32_t* __restrict__ out, uint64_t out_col_stride, unsigned log_n, unsigned sA,
                  unsigned log_T, const uint32_t* __restrict__ W, const uint32_t* __restrict__ lo,
                  const uint32_t* __restrict__ hi, uint32_t n_hi) {
    __shared__ uint32_t s[TILE_ELEMS];
    const uint32_t beta = blockIdx.z;
    const uint32_t j2_0 = blockIdx.x << log_T;
    const uint32_t* g = coef + (uint64_t)blockIdx.y * in_col_stride + j2_0;
    uint32_t* o = out + (uint64_t)blockIdx.y * out_col_stride + ((uint64_t)beta << log_n) + j2_0;
    const uint32_t* lo_b = lo + ((uint64_t)beta << SHIFT_LO_BITS);
    const uint32_t* hi_b = hi + (uint64_t)beta * n_hi;
    const uint32_t total = 1u << (sA + log_T);
    const uint32_t tmask = (1u << log_T) - 1;
    for (uint32_t i = threadIdx.x; i < total; i += NT) {
        uint32_t off = ((i >> log_T) << LOG_M) + (i & tmask);
        uint32_t k = off + j2_0;
        uint32_t v = g[off];
        v = mont_mul(v, lo_b[k & ((1u << SHIFT_LO_BITS) - 1)]);
        v = mont_mul(v, hi_b[k >> SHIFT_LO_BITS]);
        s[i] = v;
    }
    __syncthreads();
    tile_forward(s, sA, log_T, 0, 0, W);
    for (uint32_t i = threadIdx.x; i < total; i += NT)
        o[((uint64_t)(i >> log_T) << LOG_M) + (i & tmask)] = s[i];
}

// Contiguous pass.  Two uses:
//  (a) log_n <= LOG_M: src = coefficients (scaled here), all log_n stages, one chunk per column;
//  (b) log_n  > LOG_M: in place on `out` after the strided pass (scale = 0): chunk c of coset block
//      beta, global stages sA .. log_n-1.
__global__ void __launch_bounds__(NT)
k_lde_fwd_contig(const uint32_t* __restrict__ src, uint64_t src_col_stride,
                 uint32_t* __restrict__ out, uint64_t out_col_stride, unsigned log_n,
                 unsigned log_len, int scale, const uint32_t* __restrict__ W,
                 const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t n_hi) {
    __shared__ uint32_t s[CHUNK];
    const uint32_t c = blockIdx.x;
    const uint32_t beta = blockIdx.z;
    const uint32_t len = 1u << log_len;
    uint32_t* o = out + (uint64_t)blockIdx.y * out_col_stride + ((uint64_t)beta << log_n) +
                  ((uint64_t)c << log_len);
    if (scale) {
        const uint32_t* g = src + (uint64_t)blockIdx.y * src_col_stride;
        const uint32_t* lo_b = lo + ((uint64_t)beta << SHIFT_LO_BITS);
        const uint32_t* hi_b = hi + (uint64_t)beta * n_hi;
        for (uint32_t i = threadIdx.x; i < len; i += NT) {
            uint32_t v = g[i];
            v = mont_mul(v, lo_b[i & ((1u << SHIFT_LO_BITS) - 1)]);
            v = mont_mul(v, hi_b[i >> SHIFT_LO_BITS]);
            s[i] = v;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < len; i += NT) s[i] = o[i];
    }
    __syncthreads();
    tile_forward(s, log_len, 0, log_n - log_len, c, W);
    for (uint32_t i = threadIdx.x; i < len; i += NT) o[i] = s[i];
}

// ------------------------------------------------------------------ host driver
void coset_lde(Context& ctx, uint32_t* evals, uint64_t in_col_stride, uint32_t ncols, unsigned log_n,
               unsigned log_blowup, uint32_t shift, uint32_t* out, uint64_t out_col_stride) {
    TS_REQUIRE(log_n + log_blowup <= 27, TS_ERR_INVALID, "coset_lde: log_n + log_blowup > 27");
    TS_REQUIRE(ncols >= 1 && ncols <= 65535, TS_ERR_INVALID, "coset_lde: bad column count");
    const unsigned log_len = log_n < (unsigned)LOG_M ? log_n : (unsigned)LOG_M;
    const unsigned sA = log_n - log_len;
    TS_REQUIRE((1u << sA) <= (unsigned)TILE_ELEMS, TS_ERR_UNSUPPORTED,
               "coset_lde: trace longer than 2^25 rows is not supported yet");
    ctx.ensure_twiddles(log_n == 0 ? 1 : log_n);
    const uint32_t* W = ctx.d_twiddle_fwd;
    const uint32_t* Winv = ctx.d_twiddle_inv;
    const uint64_t n = 1ull << log_n;
    const uint32_t n_inv_mont = to_mont(inv_canon((uint32_t)(n % P)));
    unsigned log_T = 0;
    if (sA) {
        while ((1u << (sA + log_T + 1)) <= (unsigned)TILE_ELEMS && log_T < 6) log_T++;
    }

    // inverse transform in place -> natural-order coefficients, scaled by 1/n
    TS_LAUNCH(ctx, k_intt_contig, dim3(1u << sA, ncols), dim3(NT), 0, evals,
                       in_col_stride, log_n, log_len, Winv, n_inv_mont, sA == 0 ? 1 : 0);
    if (sA)
        TS_LAUNCH(ctx, k_intt_strided, dim3(1u << (LOG_M - log_T), ncols), dim3(NT), 0,
                           evals, in_col_stride, sA, log_T, Winv, n_inv_mont);

    // per-coset scale tables
    const uint32_t n_cosets = 1u << log_blowup;
    const uint32_t n_lo = 1u << SHIFT_LO_BITS;
    const uint32_t n_hi = log_n > (unsigned)SHIFT_LO_BITS ? 1u << (log_n - SHIFT_LO_BITS) : 1u;
    DevBuf<uint32_t> lo(&ctx, (size_t)n_cosets * n_lo), hi(&ctx, (size_t)n_cosets * n_hi);
    TS_LAUNCH(ctx, k_build_shift_tables, dim3((n_lo + n_hi + 255) / 256, n_cosets), dim3(256), 0,
                       lo.p, hi.p, n_hi, to_mont(shift), log_n + log_blowup, log_blowup,
                       R_MOD_P);

    if (sA) {
        TS_LAUNCH(ctx, k_lde_fwd_strided, dim3(1u << (LOG_M - log_T), ncols, n_cosets), dim3(NT),
                           0, evals, in_col_stride, out, out_col_stride, log_n, sA,
                           log_T, W, lo.p, hi.p, n_hi);
        TS_LAUNCH(ctx, k_lde_fwd_contig, dim3(1u << sA, ncols, n_cosets), dim3(NT), 0,
                           (const uint32_t*)nullptr, (uint64_t)0, out, out_col_stride,
                           log_n, log_len, 0, W, lo.p, hi.p, n_hi);
    } else {
        TS_LAUNCH(ctx, k_lde_fwd_contig, dim3(1, ncols, n_cosets), dim3(NT), 0, evals,
                           in_col_stride, out, out_col_stride, log_n, log_len, 1, W, lo.p, hi.p, n_hi);
    }
    TS_HIP(hipGetLastError());
}

}  // namespace ts
