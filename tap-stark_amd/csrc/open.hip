// Opening kernels (reference fri/src/two_adic_pcs.rs:260-389 `open`):
//   :358-369  ys = interpolate_coset(BitRev(first n rows), 31, z)     -> bary_weights + bary_dots
//   :678-720  inv_denoms[X] = 1/(x_X - z), x_X = 31 * w_N^bitrev(X)   -> computed on the fly
//   :371-381  ro[X] += alpha^offset * (sum_i alpha^i p_i[X] - reduced_ys) * inv_denom[X]  -> reduce
// All matrices are column-major with bit-reversed rows, so "row X" is a coalesced read per column.
// Dot products are accumulated lazily: 64-bit multiply-adds (one v_mad_u64_u32 each) with a cheap
// range fix every two terms and a single Montgomery reduction at the end.
#include "kernels.hpp"

namespace ts {

// omega_{2^L}^bitrev_L(r) in Montgomery form from the block-twiddle table
__device__ __forceinline__ uint32_t root_bitrev(const uint32_t* __restrict__ W, unsigned L, uint64_t r) {
    if (L == 0) return R_MOD_P;
    uint32_t w = W[((uint64_t)1 << (L - 1)) + (r >> 1)];
    return (r & 1) ? neg(w) : w;
}

// ------------------------------------------------------------------ barycentric weights
// out[p][t] = x_t / (z_p - x_t), x_t = 31 * omega_n^bitrev(t)   (Montgomery EF4)
constexpr int BW_ROWS = 4;

__global__ void __launch_bounds__(256)
k_bary_weights(unsigned log_n, const uint32_t* __restrict__ W, uint32_t gen_mont, Ef z0, Ef z1,
               uint32_t n_points, Ef* __restrict__ out) {
    const uint64_t n = 1ull << log_n;
    const uint64_t t0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * BW_ROWS;
    if (t0 >= n) return;
    Ef num[2 * BW_ROWS];
    uint32_t nrm[2 * BW_ROWS], pre[2 * BW_ROWS], xs[BW_ROWS];
    uint32_t run = R_MOD_P;
#pragma unroll
    for (int k = 0; k < BW_ROWS; k++) {
        const bool ok = t0 + k < n;
        uint32_t x = ok ? mont_mul(gen_mont, root_bitrev(W, log_n, t0 + k)) : R_MOD_P;
        xs[k] = x;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const Ef z = p == 0 ? z0 : z1;
            Ef u = Ef{{sub(z.c[0], x), z.c[1], z.c[2], z.c[3]}};  // z - x
            if (p < (int)n_points && ok) {
                ef_inv_parts(u, num[2 * k + p], nrm[2 * k + p]);
            } else {
                num[2 * k + p] = ef_zero();
                nrm[2 * k + p] = R_MOD_P;
            }
            pre[2 * k + p] = run;
            run = mont_mul(run, nrm[2 * k + p]);
        }
    }
    uint32_t inv = mont_inv(run);
#pragma unroll
    for (int j = 2 * BW_ROWS - 1; j >= 0; j--) {
        uint32_t ninv = mont_mul(inv, pre[j]);
        inv = mont_mul(inv, nrm[j]);
        const int k = j >> 1, p = j & 1;
        if (p < (int)n_points && t0 + k < n)
            out[(uint64_t)p * n + t0 + k] = ef_mul_base(num[j], mont_mul(ninv, xs[k]));
    }
}

void launch_bary_weights(Context& ctx, unsigned log_n, const Ef* points_mont, uint32_t n_points,
                         Ef* out, uint32_t coset_gen) {
    TS_REQUIRE(n_points >= 1 && n_points <= 2, TS_ERR_INVALID, "bary_weights: 1 or 2 points");
    ctx.ensure_twiddles(log_n == 0 ? 1 : log_n);
    const uint64_t threads = (((uint64_t)1 << log_n) + BW_ROWS - 1) / BW_ROWS;
    Ef z0 = points_mont[0], z1 = n_points > 1 ? points_mont[1] : points_mont[0];
    TS_LAUNCH(ctx, k_bary_weights, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, log_n,
              ctx.d_twiddle_fwd, to_mont(coset_gen ? coset_gen : GENERATOR), z0, z1, n_points, out);
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ barycentric dot products
// acc[col][p][k] += sum_t m[col][t] * weights[p][t].c[k]  over the first n rows (a skinny product
// [w x n].[n x 4 NP]).  A workgroup streams tiles of COLS columns x TR rows through LDS (loads
// coalesced along rows, all in flight at once); then each lane owns one column and one row subset
// and reads the tile's weights as LDS broadcasts.  COLS = 64 for wide matrices, 8 for the width-4
// quotient chunks (so that all 64 lanes stay busy).  Per-workgroup sums go to a partial buffer
// [block][col][NP*4] that k_bary_finish adds up.
template <int NP, int COLS>
__global__ void __launch_bounds__(256)
k_bary_dots(const uint32_t* __restrict__ m, uint64_t col_stride, uint32_t width, unsigned log_n,
            const Ef* __restrict__ weights, uint32_t* __restrict__ partial, uint32_t rows_per_block) {
    constexpr int RS = 64 / COLS;        // row subsets per wave
    constexpr int TR = 16 * 4 * RS;      // tile rows: 16 per (wave, row subset)
    constexpr int PER_THREAD = COLS * TR / 256;  // = 16
    // tile element (column c, row t = 16*sub + rr) lives at rr*257 + sub*COLS + c: the compute
    // phase reads it conflict-free (lanes = (sub, c)), the store phase with small conflicts only
    __shared__ uint32_t tile[16 * 257];
    __shared__ uint32_t wts[TR][NP * 4];
    __shared__ uint32_t red[4 * RS][COLS][NP * 4];
    const uint64_t n = 1ull << log_n;
    const uint32_t c0 = blockIdx.y * COLS;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t cl = lane % COLS, rs = lane / COLS;
    const uint32_t sub = wv * RS + rs;  // which 16-row slice of the tile this thread reduces
    const uint64_t row_begin = (uint64_t)blockIdx.x * rows_per_block;
    const uint64_t row_end = row_begin + rows_per_block < n ? row_begin + rows_per_block : n;
    uint64_t acc[NP * 4];
#pragma unroll
    for (int j = 0; j < NP * 4; j++) acc[j] = 0;
    for (uint64_t row0 = row_begin; row0 < row_end; row0 += TR) {
        uint32_t v[PER_THREAD];
#pragma unroll
        for (int k = 0; k < PER_THREAD; k++) {
            const uint32_t e = threadIdx.x + (uint32_t)k * 256;
            const uint32_t c = c0 + e / TR;
            const uint64_t t = row0 + e % TR;
            v[k] = (c < width && t < row_end) ? m[(uint64_t)c * col_stride + t] : 0u;
        }
        for (uint32_t e = threadIdx.x; e < (uint32_t)(TR * NP); e += 256) {
            const uint32_t p = e / TR, tt = e % TR;
            const uint64_t t = row0 + tt;
            Ef w = ef_zero();
            if (t < row_end) w = weights[(uint64_t)p * n + t];
            wts[tt][p * 4 + 0] = w.c[0];
            wts[tt][p * 4 + 1] = w.c[1];
            wts[tt][p * 4 + 2] = w.c[2];
            wts[tt][p * 4 + 3] = w.c[3];
        }
#pragma unroll
        for (int k = 0; k < PER_THREAD; k++) {
            const uint32_t e = threadIdx.x + (uint32_t)k * 256;
            const uint32_t t = e % TR;
            tile[(t & 15) * 257 + (t >> 4) * COLS + e / TR] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
            const uint32_t tt = sub * 16 + rr;
            const uint32_t x = tile[rr * 257 + sub * COLS + cl];
#pragma unroll
            for (int j = 0; j < NP * 4; j++) acc[j] = lazy_mac(acc[j], x, wts[tt][j]);
            if (rr & 1) {
#pragma unroll
                for (int j = 0; j < NP * 4; j++) acc[j] = lazy_fix(acc[j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NP * 4; j++) red[sub][cl][j] = lazy_finish(acc[j]);
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < (uint32_t)(COLS * NP * 4); idx += 256) {
        const uint32_t c = idx / (NP * 4), j = idx % (NP * 4);
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < 4 * RS; q++) sum = add(sum, red[q][c][j]);
        if (c0 + c < width)
            partial[((uint64_t)blockIdx.x * width + c0 + c) * (NP * 4) + j] = sum;
    }
}

// out[j] = sum over blocks of partial[block][j]  (mod p); one workgroup per 4 output words.  Up to two
// jobs in one launch (the trace's sums and the quotient chunks'): workgroups 0 .. g0-1 take job 0.
struct BaryFinishJob {
    const uint32_t* partial;
    uint32_t* out;
    uint32_t n_blocks, n_words;
};
__global__ void __launch_bounds__(256)
k_bary_finish(BaryFinishJob j0, BaryFinishJob j1, uint32_t g0) {
    __shared__ uint32_t red[4][4];
    const bool first = blockIdx.x < g0;
    const uint32_t* __restrict__ partial = first ? j0.partial : j1.partial;
    uint32_t* __restrict__ out = first ? j0.out : j1.out;
    const uint32_t n_blocks = first ? j0.n_blocks : j1.n_blocks, n_words = first ? j0.n_words : j1.n_words;
    const uint32_t j = (first ? blockIdx.x : blockIdx.x - g0) * 4 + (threadIdx.x & 3);
    uint32_t v = 0;
    if (j < n_words) {
        // eight loads in flight (as one load and one add per trip every trip waited for its own load: with
        // 2048 partial blocks that was 32 round trips to L2, 10-14 us for a kernel that moves kilobytes)
        uint32_t b = threadIdx.x >> 2;
        for (; b + 7 * 64 < n_blocks; b += 8 * 64) {
            uint32_t x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = partial[(uint64_t)(b + 64 * k) * n_words + j];
#pragma unroll
            for (int k = 0; k < 8; k++) v = add(v, x[k]);
        }
        for (; b < n_blocks; b += 64) v = add(v, partial[(uint64_t)b * n_words + j]);
    }
    // reduce over the 64 threads that share (threadIdx.x & 3): lanes 4 apart within a wave, 4 waves
#pragma unroll
    for (int off = 32; off >= 4; off >>= 1) v = add(v, __shfl_down(v, off, 64));
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < 4) red[wv][lane] = v;
    __syncthreads();
    if (threadIdx.x < 4 && j < n_words)
        out[j] = add(add(red[0][threadIdx.x], red[1][threadIdx.x]),
                     add(red[2][threadIdx.x], red[3][threadIdx.x]));
}

// the finishing pass of one or two launch_bary_dots(..., pending) calls
void launch_bary_finish(Context& ctx, BaryPending& pend) {
    if (pend.n == 0) return;
    BaryFinishJob j[2] = {{nullptr, nullptr, 0, 0}, {nullptr, nullptr, 0, 0}};
    uint32_t g[2] = {0, 0};
    for (uint32_t i = 0; i < pend.n; i++) {
        j[i] = BaryFinishJob{pend.partial[i].p, pend.out[i], pend.n_blocks[i], pend.n_words[i]};
        g[i] = (pend.n_words[i] + 3) / 4;
    }
    TS_LAUNCH(ctx, k_bary_finish, dim3(g[0] + g[1]), dim3(256), 0, j[0], j[1], g[0]);
    TS_HIP(hipGetLastError());
    for (uint32_t i = 0; i < pend.n; i++) pend.partial[i].reset();
    pend.n = 0;
}

void launch_bary_dots(Context& ctx, const ColMat& m, unsigned log_n, const Ef* weights,
                      uint32_t n_points, Ef* out, BaryPending* pending) {
    const uint64_t n = 1ull << log_n;
    const bool narrow = m.width <= 8;
    const uint32_t tr = narrow ? 512 : 64;
    uint32_t n_blocks = (uint32_t)((n + tr - 1) / tr);
    if (n_blocks > 2048) n_blocks = 2048;
    uint32_t rows_per_block = (uint32_t)(((n + n_blocks - 1) / n_blocks + tr - 1) / tr * tr);
    n_blocks = (uint32_t)((n + rows_per_block - 1) / rows_per_block);
    const uint32_t n_words = m.width * n_points * 4;
    DevBuf<uint32_t> partial(&ctx, (size_t)n_blocks * n_words);
    const uint32_t* md = m.d;
    if (narrow) {
        dim3 grid(n_blocks, (m.width + 7) / 8);
        if (n_points == 2)
            TS_LAUNCH(ctx, (k_bary_dots<2, 8>), grid, dim3(256), 0, md, m.col_stride, m.width, log_n,
                      weights, partial.p, rows_per_block);
        else
            TS_LAUNCH(ctx, (k_bary_dots<1, 8>), grid, dim3(256), 0, md, m.col_stride, m.width, log_n,
                      weights, partial.p, rows_per_block);
    } else {
        dim3 grid(n_blocks, (m.width + 63) / 64);
        if (n_points == 2)
            TS_LAUNCH(ctx, (k_bary_dots<2, 64>), grid, dim3(256), 0, md, m.col_stride, m.width, log_n,
                      weights, partial.p, rows_per_block);
        else
            TS_LAUNCH(ctx, (k_bary_dots<1, 64>), grid, dim3(256), 0, md, m.col_stride, m.width, log_n,
                      weights, partial.p, rows_per_block);
    }
    TS_HIP(hipGetLastError());
    BaryPending own;
    BaryPending& pend = pending ? *pending : own;
    if (pend.n == 2) launch_bary_finish(ctx, pend);
    pend.partial[pend.n] = std::move(partial);
    pend.out[pend.n] = reinterpret_cast<uint32_t*>(out);
    pend.n_blocks[pend.n] = n_blocks;
    pend.n_words[pend.n] = n_words;
    pend.n++;
    if (!pending) launch_bary_finish(ctx, pend);
}

// ------------------------------------------------------------------ reduce (generic)
// S(X) = sum_i alpha^i * p_i[X]  (dot_ext_powers, :375), canonical; alpha powers are wave-uniform
// acc[0..3] += sum_i w_i * p_i[X] over `width` columns (lazy: acc stays below p*2^32 between calls)
__device__ __forceinline__ void row_dot_acc(uint64_t (&acc)[4], const uint32_t* __restrict__ m,
                                            uint64_t col_stride, uint32_t width, uint64_t X,
                                            const uint32_t* __restrict__ weights) {
    uint64_t a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    // batches of 8 columns: the 8 loads are issued back to back (one load per iteration with a
    // wait behind it left the kernel latency-bound), then 32 MACs with a range fix every 2 columns
    constexpr int B = 8;
    const uint32_t* col = m + X;
    uint32_t i = 0;
    for (; i + B <= width; i += B) {
        uint32_t v[B];
#pragma unroll
        for (int k = 0; k < B; k++) v[k] = col[(uint64_t)(i + k) * col_stride];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const uint32_t* ap = weights + 4 * (i + k);
            a0 = lazy_mac(a0, v[k], ap[0]);
            a1 = lazy_mac(a1, v[k], ap[1]);
            a2 = lazy_mac(a2, v[k], ap[2]);
            a3 = lazy_mac(a3, v[k], ap[3]);
            if (k & 1) {
                a0 = lazy_fix(a0);
                a1 = lazy_fix(a1);
                a2 = lazy_fix(a2);
                a3 = lazy_fix(a3);
            }
        }
    }
    // tail (< 8 columns): same pattern, predicated loads
    if (i < width) {
        uint32_t v[B];
#pragma unroll
        for (int k = 0; k < B; k++) v[k] = i + k < width ? col[(uint64_t)(i + k) * col_stride] : 0u;
#pragma unroll
        for (int k = 0; k < B; k++) {
            if (i + k < width) {
                const uint32_t* ap = weights + 4 * (i + k);
                a0 = lazy_mac(a0, v[k], ap[0]);
                a1 = lazy_mac(a1, v[k], ap[1]);
                a2 = lazy_mac(a2, v[k], ap[2]);
                a3 = lazy_mac(a3, v[k], ap[3]);
            }
            if (k & 1) {
                a0 = lazy_fix(a0);
                a1 = lazy_fix(a1);
                a2 = lazy_fix(a2);
                a3 = lazy_fix(a3);
            }
        }
    }
    acc[0] = a0; acc[1] = a1; acc[2] = a2; acc[3] = a3;
}
// S(X) = sum_i alpha^i * p_i[X]  (dot_ext_powers, :375), canonical; alpha powers are wave-uniform
__device__ __forceinline__ Ef row_dot_alpha(const uint32_t* __restrict__ m, uint64_t col_stride,
                                            uint32_t width, uint64_t X,
                                            const uint32_t* __restrict__ alpha_pows) {
    uint64_t acc[4] = {0, 0, 0, 0};
    row_dot_acc(acc, m, col_stride, width, X, alpha_pows);
    return Ef{{lazy_finish(acc[0]), lazy_finish(acc[1]), lazy_finish(acc[2]), lazy_finish(acc[3])}};
}

// 1/(x - z_p) for NP points with one shared base-field inversion (Montgomery)
template <int NP>
__device__ __forceinline__ void inv_denoms(uint32_t x_mont, const Ef* z_mont, Ef* out) {
    Ef num[NP];
    uint32_t nrm[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const Ef z = z_mont[p];
        Ef u = Ef{{sub(x_mont, z.c[0]), neg(z.c[1]), neg(z.c[2]), neg(z.c[3])}};  // x - z
        ef_inv_parts(u, num[p], nrm[p]);
    }
    if (NP == 2) {
        uint32_t inv = mont_inv(mont_mul(nrm[0], nrm[NP - 1]));
        out[0] = ef_mul_base(num[0], mont_mul(inv, nrm[NP - 1]));
        out[NP - 1] = ef_mul_base(num[NP - 1], mont_mul(inv, nrm[0]));
    } else {
        out[0] = ef_mul_base(num[0], mont_inv(nrm[0]));
    }
}

template <int NP>
__global__ void __launch_bounds__(256)
k_reduce(const uint32_t* __restrict__ m, uint64_t col_stride, uint32_t width, unsigned log_h,
         const uint32_t* __restrict__ W, uint32_t gen_mont, const uint32_t* __restrict__ alpha_pows,
         ReduceArgs args, Ef* __restrict__ ro) {
    const uint64_t h = 1ull << log_h;
    const uint64_t X = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (X >= h) return;
    const Ef S = row_dot_alpha(m, col_stride, width, X, alpha_pows);
    // x_X = 31 * omega_h^bitrev(X)  (:698-705), Montgomery
    const uint32_t x = mont_mul(gen_mont, root_bitrev(W, log_h, X));
    Ef inv_d[NP];
    inv_denoms<NP>(x, args.z_mont, inv_d);
    Ef acc = args.accumulate ? ro[X] : ef_zero();
#pragma unroll
    for (int p = 0; p < NP; p++) {
        Ef t = ef_mul(ef_sub(S, args.rys[p]), args.off_mont[p]);  // canonical
        acc = ef_add(acc, ef_mul(t, inv_d[p]));                   // canonical
    }
    ro[X] = acc;
}

void launch_reduce(Context& ctx, const ColMat& m, unsigned log_h, const uint32_t* d_alpha_pows_mont,
                   const ReduceArgs& args, Ef* ro) {
    TS_REQUIRE(args.n_points >= 1 && args.n_points <= 2, TS_ERR_INVALID, "reduce: 1 or 2 points");
    ctx.ensure_twiddles(log_h == 0 ? 1 : log_h);
    const uint64_t h = 1ull << log_h;
    dim3 grid((unsigned)((h + 255) / 256));
    if (args.n_points == 2)
        TS_LAUNCH(ctx, k_reduce<2>, grid, dim3(256), 0, (const uint32_t*)m.d, m.col_stride, m.width,
                  log_h, (const uint32_t*)ctx.d_twiddle_fwd, to_mont(GENERATOR), d_alpha_pows_mont, args,
                  ro);
    else
        TS_LAUNCH(ctx, k_reduce<1>, grid, dim3(256), 0, (const uint32_t*)m.d, m.col_stride, m.width,
                  log_h, (const uint32_t*)ctx.d_twiddle_fwd, to_mont(GENERATOR), d_alpha_pows_mont, args,
                  ro);
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ reduce, prove() shape, one pass
// ro[X] = inv(x - zeta)      * [ off_t0 (S_t - rys_t0) + sum_c off_c (S_c - rys_c) ]
//       + inv(x - zeta omega) *   off_t1 (S_t - rys_t1)
// where S_t is shared by the two trace openings (two_adic_pcs.rs:344-387 visits the trace twice).
// Evaluated as  g0 = off_t0 S_t + D - k0,  g1 = off_t1 S_t - k1,  D = one dot product over every
// chunk column with the folded weights alpha^k off_c (FusedReduceArgs): four EF4 products per row
// instead of 5 + n_chunks -- this kernel, too, is bound by VALU issue, not by its 4 TB/s.
__global__ void __launch_bounds__(256)
k_reduce_fused(const uint32_t* __restrict__ trace, uint64_t trace_stride, uint32_t width,
               unsigned log_h, const uint32_t* __restrict__ W, uint32_t gen_mont,
               const uint32_t* __restrict__ alpha_pows, FusedReduceArgs a, Ef* __restrict__ ro) {
    const uint64_t X = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // local row
    if (X >= a.rows) return;
    const Ef St = row_dot_alpha(trace, trace_stride, width, X, alpha_pows);
    const uint32_t x = mont_mul(gen_mont, root_bitrev(W, log_h, a.row0 + X));
    Ef inv_d[2];
    inv_denoms<2>(x, a.z_mont, inv_d);
    uint64_t acc[4] = {0, 0, 0, 0};
    uint32_t c = 0;
    for (; c + 2 <= a.n_chunks; c += 2) {
        // two width-4 chunks = one full batch of eight columns (their weights are consecutive): eight
        // loads in flight, 32 multiply-adds, 16 range fixes -- as two 4-column tails it cost twice that
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = a.chunk[c + (k >> 2)][(uint64_t)(k & 3) * a.chunk_stride + X];
        const uint32_t* ap = a.chunk_w + 16 * c;
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = lazy_mac(acc[j], v[k], ap[4 * k + j]);
            if (k & 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = lazy_fix(acc[j]);
            }
        }
    }
    for (; c < a.n_chunks; c++) row_dot_acc(acc, a.chunk[c], a.chunk_stride, 4, X, a.chunk_w + 16 * c);
    const Ef D{{lazy_finish(acc[0]), lazy_finish(acc[1]), lazy_finish(acc[2]), lazy_finish(acc[3])}};
    const Ef g0 = ef_sub(ef_add(ef_mul(St, a.off_t[0]), D), a.k0);
    const Ef g1 = ef_sub(ef_mul(St, a.off_t[1]), a.k1);
    const Ef r = ef_add(ef_mul(g0, inv_d[0]), ef_mul(g1, inv_d[1]));
    *reinterpret_cast<uint4*>(ro + X) = make_uint4(r.c[0], r.c[1], r.c[2], r.c[3]);
}

void launch_reduce_fused(Context& ctx, const ColMat& trace, unsigned log_h,
                         const uint32_t* d_alpha_pows_mont, const FusedReduceArgs& args, Ef* ro) {
    TS_REQUIRE(args.n_chunks <= (uint32_t)MAX_QUOTIENT_CHUNKS, TS_ERR_INVALID, "reduce_fused: too many chunks");
    ctx.ensure_twiddles(log_h == 0 ? 1 : log_h);
    FusedReduceArgs a = args;
    if (a.rows == 0) {
        a.row0 = 0;
        a.rows = 1ull << log_h;
    }
    TS_REQUIRE(a.row0 + a.rows <= (1ull << log_h), TS_ERR_INVALID, "reduce_fused: row range");
    TS_LAUNCH(ctx, k_reduce_fused, dim3((unsigned)((a.rows + 255) / 256)), dim3(256), 0,
              (const uint32_t*)trace.d, trace.col_stride, trace.width, log_h,
              (const uint32_t*)ctx.d_twiddle_fwd, to_mont(GENERATOR), d_alpha_pows_mont, a, ro);
    TS_HIP(hipGetLastError());
}

}  // namespace ts
