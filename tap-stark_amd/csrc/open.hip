// Opening kernels (reference fri/src/two_adic_pcs.rs:260-389 `open`):
//   :358-369  ys = interpolate_coset(BitRev(first n rows), 31, z)     -> bary_weights + bary_sums
//   :678-720  inv_denoms[X] = 1/(x_X - z), x_X = 31 * w_N^bitrev(X)   -> computed on the fly
//   :371-381  ro[X] += alpha^offset * (sum_i alpha^i p_i[X] - reduced_ys) * inv_denom[X]  -> reduce
// All matrices are column-major with bit-reversed rows, so "row X" is a coalesced read per column.
// A matrix opened at two points (trace at zeta and zeta*w_n) shares one pass over its columns.
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
                         Ef* out) {
    TS_REQUIRE(n_points >= 1 && n_points <= 2, TS_ERR_INVALID, "bary_weights: 1 or 2 points");
    ctx.ensure_twiddles(log_n == 0 ? 1 : log_n);
    const uint64_t threads = (((uint64_t)1 << log_n) + BW_ROWS - 1) / BW_ROWS;
    Ef z0 = points_mont[0], z1 = n_points > 1 ? points_mont[1] : points_mont[0];
    TS_LAUNCH(ctx, k_bary_weights, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       log_n, ctx.d_twiddle_fwd, to_mont(GENERATOR), z0, z1, n_points, out);
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ barycentric sums
// partial[chunk][col][p] = sum over the chunk's rows of m[col][t] * weights[p][t]
constexpr int BS_COLS = 8;           // columns per workgroup
constexpr int BS_ROWS_PER_THREAD = 16;
constexpr int BS_CHUNK = 256 * BS_ROWS_PER_THREAD;

__device__ __forceinline__ uint32_t wave_sum_modp(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = add(v, __shfl_down(v, off, 64));
    return v;
}

template <int NP>
__global__ void __launch_bounds__(256)
k_bary_sums(const uint32_t* __restrict__ m, uint64_t col_stride, uint32_t width, unsigned log_n,
            const Ef* __restrict__ weights, Ef* __restrict__ partial) {
    const uint64_t n = 1ull << log_n;
    const uint32_t chunk = blockIdx.x;
    const uint32_t c0 = blockIdx.y * BS_COLS;
    uint32_t acc[BS_COLS][NP][4];
#pragma unroll
    for (int c = 0; c < BS_COLS; c++)
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc[c][p][k] = 0;
    for (int it = 0; it < BS_ROWS_PER_THREAD; it++) {
        const uint64_t t = (uint64_t)chunk * BS_CHUNK + (uint64_t)it * 256 + threadIdx.x;
        if (t >= n) break;
        Ef wgt[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) wgt[p] = weights[(uint64_t)p * n + t];
#pragma unroll
        for (int c = 0; c < BS_COLS; c++) {
            if (c0 + c < width) {
                const uint32_t v = m[(uint64_t)(c0 + c) * col_stride + t];
#pragma unroll
                for (int p = 0; p < NP; p++)
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        acc[c][p][k] = add(acc[c][p][k], mont_mul(v, wgt[p].c[k]));
            }
        }
    }
    // workgroup reduction: wave shuffles, then 4 wave leaders through LDS
    __shared__ uint32_t red[4][BS_COLS * NP * 4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < BS_COLS; c++)
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t v = wave_sum_modp(acc[c][p][k]);
                if (lane == 0) red[wave][(c * NP + p) * 4 + k] = v;
            }
    __syncthreads();
    if (threadIdx.x < BS_COLS * NP * 4) {
        const uint32_t j = threadIdx.x;
        uint32_t v = add(add(red[0][j], red[1][j]), add(red[2][j], red[3][j]));
        const uint32_t c = j / (NP * 4), rem = j % (NP * 4);
        if (c0 + c < width) {
            uint32_t* o = reinterpret_cast<uint32_t*>(partial + ((uint64_t)chunk * width + c0 + c) * NP);
            o[rem] = v;
        }
    }
}

// out[col][p] = sum over chunks of partial[chunk][col][p]
__global__ void __launch_bounds__(256)
k_bary_finish(const uint32_t* __restrict__ partial, uint32_t n_chunks, uint32_t n_words,
              uint32_t* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_words) return;
    uint32_t v = 0;
    for (uint32_t ch = 0; ch < n_chunks; ch++) v = add(v, partial[(uint64_t)ch * n_words + j]);
    out[j] = v;
}

void launch_bary_sums(Context& ctx, const ColMat& m, unsigned log_n, const Ef* weights,
                      uint32_t n_points, Ef* out) {
    const uint64_t n = 1ull << log_n;
    const uint32_t n_chunks = (uint32_t)((n + BS_CHUNK - 1) / BS_CHUNK);
    DevBuf<Ef> partial(&ctx, (size_t)n_chunks * m.width * n_points);
    dim3 grid(n_chunks, (m.width + BS_COLS - 1) / BS_COLS);
    if (n_points == 2)
        TS_LAUNCH(ctx, k_bary_sums<2>, grid, dim3(256), 0, m.d, m.col_stride, m.width,
                           log_n, weights, partial.p);
    else
        TS_LAUNCH(ctx, k_bary_sums<1>, grid, dim3(256), 0, m.d, m.col_stride, m.width,
                           log_n, weights, partial.p);
    const uint32_t n_words = m.width * n_points * 4;
    TS_LAUNCH(ctx, k_bary_finish, dim3((n_words + 255) / 256), dim3(256), 0, reinterpret_cast<const uint32_t*>(partial.p), n_chunks, n_words,
                       reinterpret_cast<uint32_t*>(out));
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ reduce
template <int NP>
__global__ void __launch_bounds__(256)
k_reduce(const uint32_t* __restrict__ m, uint64_t col_stride, uint32_t width, unsigned log_h,
         const uint32_t* __restrict__ W, uint32_t gen_mont, const uint32_t* __restrict__ alpha_pows,
         ReduceArgs args, Ef* __restrict__ ro) {
    const uint64_t h = 1ull << log_h;
    const uint64_t X = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (X >= h) return;
    // S(X) = sum_i alpha^i * p_i[X]  (dot_ext_powers, :375) -- canonical
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (uint32_t i = 0; i < width; i++) {
        const uint32_t v = m[(uint64_t)i * col_stride + X];
        const uint32_t* ap = alpha_pows + 4 * i;  // wave-uniform
        s0 = add(s0, mont_mul(v, ap[0]));
        s1 = add(s1, mont_mul(v, ap[1]));
        s2 = add(s2, mont_mul(v, ap[2]));
        s3 = add(s3, mont_mul(v, ap[3]));
    }
    const Ef S = Ef{{s0, s1, s2, s3}};
    // x_X = 31 * omega_h^bitrev(X)  (:698-705), Montgomery
    const uint32_t x = mont_mul(gen_mont, root_bitrev(W, log_h, X));
    Ef num[NP];
    uint32_t nrm[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const Ef z = args.z_mont[p];
        Ef u = Ef{{sub(x, z.c[0]), neg(z.c[1]), neg(z.c[2]), neg(z.c[3])}};  // x - z
        ef_inv_parts(u, num[p], nrm[p]);
    }
    uint32_t ninv[NP];
    if (NP == 2) {
        uint32_t inv = mont_inv(mont_mul(nrm[0], nrm[1]));
        ninv[0] = mont_mul(inv, nrm[NP - 1]);
        ninv[NP - 1] = mont_mul(inv, nrm[0]);
    } else {
        ninv[0] = mont_inv(nrm[0]);
    }
    Ef acc = args.accumulate ? ro[X] : ef_zero();
#pragma unroll
    for (int p = 0; p < NP; p++) {
        Ef inv_denom = ef_mul_base(num[p], ninv[p]);            // Montgomery
        Ef t = ef_mul(ef_sub(S, args.rys[p]), args.off_mont[p]);  // canonical
        acc = ef_add(acc, ef_mul(t, inv_denom));                 // canonical
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
        TS_LAUNCH(ctx, k_reduce<2>, grid, dim3(256), 0, m.d, m.col_stride, m.width,
                           log_h, ctx.d_twiddle_fwd, to_mont(GENERATOR), d_alpha_pows_mont, args, ro);
    else
        TS_LAUNCH(ctx, k_reduce<1>, grid, dim3(256), 0, m.d, m.col_stride, m.width,
                           log_h, ctx.d_twiddle_fwd, to_mont(GENERATOR), d_alpha_pows_mont, args, ro);
    TS_HIP(hipGetLastError());
}

}  // namespace ts
