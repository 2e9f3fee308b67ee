// Coset LDE on column-major BabyBear columns: inverse NTT -> coset scaling -> forward NTT, with
// the result in bit-reversed row order (reference fri/src/two_adic_pcs.rs:233-241:
// `dft.coset_lde_batch(evals, log_blowup, shift).bit_reverse_rows()`).
//
// Formulation (DESIGN.md "NTT"): block-twiddle radix-2 transform.  Forward stage s (m = 2^s
// blocks, distance t = n/2^(s+1)) applies (a, b) -> (a + w b, a - w b) with ONE twiddle per block,
// w = W[m + blk] = omega_{2m}^bitrev(blk); natural-order input, bit-reversed output, no separate
// twist between passes.  The inverse runs the stages backwards with (A, B) -> (A + B, (A - B)/w).
//
// Execution: up to 4 stages at a time are done in registers (a thread owns the 16 elements of a
// radix-16 group), with one LDS exchange between such rounds; the LDS image is padded by one word
// per 16 so that every round's access pattern is bank-conflict free.  Three kernels:
//   k_intt_contig    stages log_n-1 .. sA of the inverse on 4096-element chunks   (only n > 4096)
//   k_lde_mid        the strided stages of the inverse (sA-1 .. 0), then for every coset: scale
//                    coefficient k by s_beta^k / n and run the strided stages of the forward
//                    transform, writing coset block beta -- the coefficients never touch HBM
//   k_lde_fwd_contig stages sA .. log_n-1 of the forward transform, in place on 4096-element chunks
// For n <= 4096 k_lde_mid alone does everything.  Data in HBM is canonical; twiddles are
// Montgomery, so mont_mul(data, twiddle) is canonical.
#include "kernels.hpp"

namespace ts {

constexpr int SHIFT_LO_BITS = 10;  // coset scale s^k = hi[k >> 10] * lo[k & 1023]

__device__ __forceinline__ uint32_t pad(uint32_t i) { return i + (i >> 4); }
constexpr int padded(int n) { return n + (n >> 4); }

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
    TS_LAUNCH(ctx, k_build_twiddles, dim3((n + 255) / 256), dim3(256), 0, W, Winv, log_size);
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

void launch_build_shift_tables(Context& ctx, uint32_t* lo, uint32_t* hi, uint32_t n_hi,
                               uint32_t n_cosets, uint32_t shift_mont, unsigned log_N,
                               unsigned log_blowup, uint32_t scale_mont) {
    const uint32_t n_lo = 1u << SHIFT_LO_BITS;
    TS_LAUNCH(ctx, k_build_shift_tables, dim3((n_lo + n_hi + 255) / 256, n_cosets), dim3(256), 0, lo,
              hi, n_hi, shift_mont, log_N, log_blowup, scale_mont);
    TS_HIP(hipGetLastError());
}

// T[beta][k] = lo[beta][k & 1023] * hi[beta][k >> 10] = s_beta^k * scale   (Montgomery), k < n
__global__ void __launch_bounds__(256)
k_build_scale_table(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t n_hi,
                    unsigned log_n, uint32_t* __restrict__ T) {
    const uint32_t beta = blockIdx.y;
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1ull << log_n)) return;
    const uint32_t n_lo = 1u << SHIFT_LO_BITS;
    T[((uint64_t)beta << log_n) + k] =
        mont_mul(lo[beta * n_lo + (k & (n_lo - 1))], hi[(uint64_t)beta * n_hi + (k >> SHIFT_LO_BITS)]);
}

const uint32_t* coset_scale_table(Context& ctx, unsigned log_n, unsigned log_blowup, uint32_t shift) {
    for (auto& t : ctx.scale_tables)
        if (t.log_n == log_n && t.log_blowup == log_blowup && t.shift == shift) {
            t.last_use = ++ctx.scale_clock;
            return t.d;
        }
    const uint32_t n_cosets = 1u << log_blowup;
    const size_t words = (size_t)n_cosets << log_n;
    // keep at most 8 tables / 2 GiB; evicting needs the stream idle (a launch may still read one)
    size_t total = words * 4;
    for (auto& t : ctx.scale_tables) total += t.words * 4;
    while (!ctx.scale_tables.empty() && (ctx.scale_tables.size() >= 8 || total > (2ull << 30))) {
        size_t victim = 0;
        for (size_t i = 1; i < ctx.scale_tables.size(); i++)
            if (ctx.scale_tables[i].last_use < ctx.scale_tables[victim].last_use) victim = i;
        ctx.sync();
        total -= ctx.scale_tables[victim].words * 4;
        (void)hipFree(ctx.scale_tables[victim].d);
        ctx.scale_tables.erase(ctx.scale_tables.begin() + victim);
    }
    uint32_t* d = nullptr;
    TS_HIP(hipMalloc((void**)&d, words * 4));
    const uint64_t n = 1ull << log_n;
    const uint32_t n_lo = 1u << SHIFT_LO_BITS;
    const uint32_t n_hi = log_n > (unsigned)SHIFT_LO_BITS ? 1u << (log_n - SHIFT_LO_BITS) : 1u;
    DevBuf<uint32_t> lo(&ctx, (size_t)n_cosets * n_lo), hi(&ctx, (size_t)n_cosets * n_hi);
    const uint32_t n_inv_mont = to_mont(inv_canon((uint32_t)(n % P)));
    launch_build_shift_tables(ctx, lo.p, hi.p, n_hi, n_cosets, to_mont(shift), log_n + log_blowup, log_blowup,
                              n_inv_mont);
    TS_LAUNCH(ctx, k_build_scale_table, dim3((unsigned)((n + 255) / 256), n_cosets), dim3(256), 0,
              (const uint32_t*)lo.p, (const uint32_t*)hi.p, n_hi, log_n, d);
    TS_HIP(hipGetLastError());
    ctx.scale_tables.push_back(Context::ScaleTable{log_n, log_blowup, shift, d, words, ++ctx.scale_clock});
    return d;
}

// ------------------------------------------------------------------ transposes
// src row-major [n][w] natural  ->  dst[c][p] = src[bitrev(p)][c]
__global__ void k_transpose_bitrev(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                   unsigned log_n, uint32_t w, uint64_t dst_col_stride,
                                   uint32_t src_width) {
    __shared__ uint32_t tile[64][65];
    const unsigned tr = log_n < 6 ? log_n : 6;  // log2 of tile rows
    const uint32_t rows = 1u << tr;
    const uint32_t p0 = blockIdx.x << tr;
    const uint32_t c0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (rows == 64 && c0 + 64 <= w) {
        // full tile: 16 loads in flight, then 16 LDS writes; as run-time loops every iteration waited
        // for its own load.  bitrev(p0 + i) for i < 64 = bitrev(p0) + (bitrev6(i) << (log_n - 6))
        // (p0 is a multiple of 64: its six low bits are free), so the source rows of a thread are a
        // base plus constants.
        const uint32_t rb = bitrev32(p0, log_n);
        const uint32_t* sp = src + (uint64_t)rb * src_width + c0 + tx;
        const uint64_t row_step = (uint64_t)src_width << (log_n - 6);
        uint32_t t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t i = ty + 4 * (uint32_t)k;  // tile row
            t[k] = sp[(uint64_t)(__brev(i) >> 26) * row_step];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) tile[ty + 4 * k][tx] = t[k];
        __syncthreads();
        uint32_t* dp = dst + (uint64_t)(c0 + ty) * dst_col_stride + p0 + tx;
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = tile[tx][ty + 4 * k];
#pragma unroll
        for (int k = 0; k < 16; k++) dp[(uint64_t)(4 * k) * dst_col_stride] = t[k];
        return;
    }
    for (uint32_t i = ty; i < rows; i += 4) {
        uint32_t r = bitrev32(p0 + i, log_n);
        uint32_t c = c0 + tx;
        if (c < w) tile[i][tx] = src[(uint64_t)r * src_width + c];
    }
    __syncthreads();
    for (uint32_t cc = ty; cc < 64; cc += 4) {
        uint32_t c = c0 + cc;
        if (c < w && tx < rows) dst[(uint64_t)c * dst_col_stride + p0 + tx] = tile[tx][cc];
    }
}

void launch_transpose_bitrev(Context& ctx, const uint32_t* src, uint32_t* dst, unsigned log_n,
                             uint32_t w, uint64_t dst_col_stride, uint32_t src_width) {
    if (w == 0) return;
    unsigned tr = log_n < 6 ? log_n : 6;
    dim3 grid(1u << (log_n - tr), (w + 63) / 64);
    TS_LAUNCH(ctx, k_transpose_bitrev, grid, dim3(256), 0, src, dst, log_n, w, dst_col_stride,
              src_width ? src_width : w);
    TS_HIP(hipGetLastError());
}

// src row-major [n][w]  ->  dst[c][r] = src[r][c], any n
__global__ void k_transpose_plain(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                  uint64_t n, uint32_t w, uint64_t dst_col_stride) {
    __shared__ uint32_t tile[64][65];
    const uint64_t r0 = (uint64_t)blockIdx.x * 64;
    const uint32_t c0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (uint32_t i = ty; i < 64; i += 4) {
        const uint32_t c = c0 + tx;
        if (c < w && r0 + i < n) tile[i][tx] = src[(r0 + i) * w + c];
    }
    __syncthreads();
    for (uint32_t cc = ty; cc < 64; cc += 4) {
        const uint32_t c = c0 + cc;
        if (c < w && r0 + tx < n) dst[(uint64_t)c * dst_col_stride + r0 + tx] = tile[tx][cc];
    }
}

void launch_transpose_plain(Context& ctx, const uint32_t* src, uint32_t* dst, uint64_t n, uint32_t w,
                            uint64_t dst_col_stride) {
    dim3 grid((unsigned)((n + 63) / 64), (w + 63) / 64);
    TS_LAUNCH(ctx, k_transpose_plain, grid, dim3(256), 0, src, dst, n, w, dst_col_stride);
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
    TS_LAUNCH(ctx, k_transpose_to_row_major, grid, dim3(256), 0, src, col_stride, dst, h, w);
    TS_HIP(hipGetLastError());
}

}  // namespace ts
