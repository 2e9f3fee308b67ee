// FRI commit-phase folding and query-phase gathers.
//   fold:    reference fri/src/two_adic_pcs.rs:116-147 `fold_matrix` (same math as
//            fri/src/fold_even_odd.rs:20-52): out[i] = (1/2 + b/2 g^-bitrev(i)) lo + (1/2 - b/2 g^-bitrev(i)) hi
//            with (lo, hi) = (f[2i], f[2i+1]) and g = two_adic_generator(log2(h) + 1).
//   gathers: reference fri/src/prover.rs:69-90 `bf_answer_query` and
//            fri/src/two_adic_pcs.rs:399-414 (open_batch at `index >> bits_reduced`).
// The folded vector is an array of EF4 (16 B); a fold thread produces two adjacent outputs, i.e.
// exactly one leaf of the NEXT round's commit-phase matrix, and hashes it in the same pass.
#include "blake3.hpp"
#include "kernels.hpp"

namespace ts {

__device__ __forceinline__ Ef load_ef(const Ef* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Ef{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void store_ef(Ef* p, Ef e) {
    *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]);
}

// out = (lo + hi)/2 + (lo - hi) * w * (beta/2);  w = g^-bitrev(i) (Montgomery base)
__device__ __forceinline__ Ef fold_one(Ef lo, Ef hi, uint32_t w_mont, Ef half_beta_mont,
                                       uint32_t half_mont) {
    Ef s = ef_mul_base(ef_add(lo, hi), half_mont);
    Ef d = ef_mul_base(ef_sub(lo, hi), w_mont);
    return ef_add(s, ef_mul(d, half_beta_mont));
}

__global__ void __launch_bounds__(256)
k_fri_fold_pairs(const Ef* __restrict__ in, uint64_t h, const uint32_t* __restrict__ Winv,
                 Ef half_beta_mont, uint32_t half_mont, Ef* __restrict__ out,
                 uint32_t* __restrict__ next_digests) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // output pair index
    if (2 * j >= h) return;
    const uint64_t i0 = 2 * j, i1 = 2 * j + 1;
    Ef a = fold_one(load_ef(in + 2 * i0), load_ef(in + 2 * i0 + 1), Winv[h + i0], half_beta_mont,
                    half_mont);
    Ef b = fold_one(load_ef(in + 2 * i1), load_ef(in + 2 * i1 + 1), Winv[h + i1], half_beta_mont,
                    half_mont);
    store_ef(out + i0, a);
    store_ef(out + i1, b);
    if (next_digests) {
        uint32_t m[16] = {a.c[0], a.c[1], a.c[2], a.c[3], b.c[0], b.c[1], b.c[2], b.c[3],
                          0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t cv[8];
        b3::iv(cv);
        b3::compress(cv, m, 32, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
        uint4* o = reinterpret_cast<uint4*>(next_digests + 8 * j);
        o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
    }
}

__global__ void k_fri_fold_single(const Ef* __restrict__ in, uint64_t h,
                                  const uint32_t* __restrict__ Winv, Ef half_beta_mont,
                                  uint32_t half_mont, Ef* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h) return;
    store_ef(out + i, fold_one(load_ef(in + 2 * i), load_ef(in + 2 * i + 1), Winv[h + i],
                               half_beta_mont, half_mont));
}

void launch_fri_fold(Context& ctx, const Ef* in, uint64_t h, Ef beta, Ef* out,
                     uint32_t* next_digests) {
    unsigned log_h = 0;
    while ((1ull << log_h) < h) log_h++;
    TS_REQUIRE((1ull << log_h) == h, TS_ERR_INVALID, "fri_fold: length not a power of two");
    ctx.ensure_twiddles(log_h + 1);
    const uint32_t half_mont = to_mont(inv_canon(2));
    const Ef half_beta_mont = ef_mul_base(ef_to_mont(beta), half_mont);
    if (h >= 2) {
        const uint64_t pairs = h / 2;
        TS_LAUNCH(ctx, k_fri_fold_pairs, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0,
                           in, h, ctx.d_twiddle_inv, half_beta_mont, half_mont, out,
                           next_digests);
    } else {
        TS_REQUIRE(next_digests == nullptr, TS_ERR_INVALID, "fri_fold: no next round at h = 1");
        TS_LAUNCH(ctx, k_fri_fold_single, dim3(1), dim3(64), 0, in, h,
                           ctx.d_twiddle_inv, half_beta_mont, half_mont, out);
    }
    TS_HIP(hipGetLastError());
}

// acc[i] += other[i]   (reference fri/src/prover.rs:124-126)
__global__ void k_vec_add(Ef* __restrict__ acc, const Ef* __restrict__ other, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_ef(acc + i, ef_add(load_ef(acc + i), load_ef(other + i)));
}
void launch_vec_add(Context& ctx, Ef* acc, const Ef* other, uint64_t n) {
    TS_LAUNCH(ctx, k_vec_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, acc,
                       other, n);
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ gathers
// out[q][0..total_width) = row (indices[q] >> shift) of every matrix, concatenated
__global__ void k_gather_rows(LeafMats mats, const uint32_t* __restrict__ indices, uint32_t n_idx,
                              unsigned shift, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = mats.total_width;
    if (t >= n_idx * total) return;
    const uint32_t q = t / total;
    uint32_t c = t % total;
    const uint64_t row = indices[q] >> shift;
    uint32_t mi = 0;
    while (c >= mats.width[mi]) {
        c -= mats.width[mi];
        mi++;
    }
    out[t] = mats.d[mi][(uint64_t)c * mats.col_stride[mi] + row];
}
void launch_gather_rows(Context& ctx, const LeafMats& mats, const uint32_t* d_indices,
                        uint32_t n_idx, unsigned index_shift, uint32_t* out) {
    const uint32_t total = n_idx * mats.total_width;
    if (!total) return;
    TS_LAUNCH(ctx, k_gather_rows, dim3((total + 255) / 256), dim3(256), 0, mats,
                       d_indices, n_idx, index_shift, out);
    TS_HIP(hipGetLastError());
}

// out[q][l][0..8) = sibling digest at level l of leaf (indices[q] >> shift)
__global__ void k_gather_paths(const uint32_t* __restrict__ tree, unsigned log_leaves,
                               const uint32_t* __restrict__ indices, uint32_t n_idx, unsigned shift,
                               uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * log_leaves * 8) return;
    const uint32_t word = t & 7;
    const uint32_t l = (t >> 3) % log_leaves;
    const uint32_t q = (t >> 3) / log_leaves;
    const uint64_t leaf = indices[q] >> shift;
    uint64_t off = 0;
    for (unsigned k = 0; k < l; k++) off += (uint64_t)1 << (log_leaves - k);
    const uint64_t node = off + ((leaf >> l) ^ 1);
    out[t] = tree[8 * node + word];
}
void launch_gather_paths(Context& ctx, const uint32_t* tree, unsigned log_leaves,
                         const uint32_t* d_indices, uint32_t n_idx, unsigned index_shift,
                         uint32_t* out) {
    const uint32_t total = n_idx * log_leaves * 8;
    if (!total) return;
    TS_LAUNCH(ctx, k_gather_paths, dim3((total + 255) / 256), dim3(256), 0, tree,
                       log_leaves, d_indices, n_idx, index_shift, out);
    TS_HIP(hipGetLastError());
}

// out[q][0..8) = (vec[2r], vec[2r+1]), r = indices[q] >> shift
__global__ void k_gather_ef_pairs(const uint32_t* __restrict__ vec, const uint32_t* __restrict__ indices,
                                  uint32_t n_idx, unsigned shift, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * 8) return;
    const uint64_t r = indices[t >> 3] >> shift;
    out[t] = vec[8 * r + (t & 7)];
}
void launch_gather_ef_pairs(Context& ctx, const Ef* vec, const uint32_t* d_indices, uint32_t n_idx,
                            unsigned index_shift, uint32_t* out) {
    if (!n_idx) return;
    TS_LAUNCH(ctx, k_gather_ef_pairs, dim3((n_idx * 8 + 255) / 256), dim3(256), 0, reinterpret_cast<const uint32_t*>(vec), d_indices, n_idx, index_shift, out);
    TS_HIP(hipGetLastError());
}

}  // namespace ts
