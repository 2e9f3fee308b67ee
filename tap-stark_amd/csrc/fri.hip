// FRI commit-phase folding and query-phase gathers.
//   fold:    reference fri/src/two_adic_pcs.rs:116-147 `fold_matrix` (same math as
//            fri/src/fold_even_odd.rs:20-52): out[i] = (1/2 + b/2 g^-bitrev(i)) lo + (1/2 - b/2 g^-bitrev(i)) hi
//            with (lo, hi) = (f[2i], f[2i+1]) and g = two_adic_generator(log2(h) + 1).
//   gathers: reference fri/src/prover.rs:69-90 `bf_answer_query` and
//            fri/src/two_adic_pcs.rs:399-414 (open_batch at `index >> bits_reduced`).
// The folded vector is an array of EF4 (16 B); a fold thread produces two adjacent outputs, i.e.
// exactly one leaf of the NEXT round's commit-phase matrix, and hashes it in the same pass.
#include "blake3.hpp"
#include "blake3_quad.hpp"
#include "chal_dev.hpp"
#include "kernels.hpp"
#include "merkle_tree.hpp"
#include "leaf_tree.hpp"

namespace ts {

__device__ __forceinline__ Ef load_ef(const Ef* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Ef{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void store_ef(Ef* p, Ef e) {
    *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]);
}

constexpr uint32_t HALF_MONT = 0x07ffffffu;  // to_mont(2^-1)

// out = (lo + hi)/2 + (lo - hi) * w * (beta/2);  w = g^-bitrev(i) (Montgomery base)
__device__ __forceinline__ Ef fold_one(Ef lo, Ef hi, uint32_t w_mont, Ef half_beta_mont,
                                       uint32_t half_mont) {
    Ef s = ef_mul_base(ef_add(lo, hi), half_mont);
    Ef d = ef_mul_base(ef_sub(lo, hi), w_mont);
    return ef_add(s, ef_mul(d, half_beta_mont));
}

__global__ void __launch_bounds__(256)
k_fri_fold_pairs(const Ef* __restrict__ in, uint64_t h, const uint32_t* __restrict__ Winv,
                 const Ef* __restrict__ beta_ptr, Ef* __restrict__ out,
                 uint32_t* __restrict__ next_digests) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // output pair index
    if (2 * j >= h) return;
    const uint32_t half_mont = HALF_MONT;
    const Ef half_beta_mont = ef_mul_base(ef_to_mont(load_ef(beta_ptr)), half_mont);
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
                                  const uint32_t* __restrict__ Winv,
                                  const Ef* __restrict__ beta_ptr, Ef* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h) return;
    const uint32_t half_mont = HALF_MONT;
    const Ef half_beta_mont = ef_mul_base(ef_to_mont(load_ef(beta_ptr)), half_mont);
    store_ef(out + i, fold_one(load_ef(in + 2 * i), load_ef(in + 2 * i + 1), Winv[h + i],
                               half_beta_mont, half_mont));
}

void launch_fri_fold_dev(Context& ctx, const Ef* in, uint64_t h, const Ef* d_beta, Ef* out,
                         uint32_t* next_digests, uint64_t h_global, uint64_t row0) {
    if (h_global == 0) h_global = h;
    unsigned log_h = 0;
    while ((1ull << log_h) < h_global) log_h++;
    TS_REQUIRE((1ull << log_h) == h_global && row0 + h <= h_global, TS_ERR_INVALID,
               "fri_fold: length not a power of two");
    ctx.ensure_twiddles(log_h + 1);
    // the kernels read Winv[h + i] for local i: shift the table so that this is the twiddle of
    // global row row0 + i, Winv[h_global + row0 + i]
    const uint32_t* tw = ctx.d_twiddle_inv + (h_global - h) + row0;
    if (h >= 2) {
        const uint64_t pairs = h / 2;
        TS_LAUNCH(ctx, k_fri_fold_pairs, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, in, h,
                  tw, d_beta, out, next_digests);
    } else {
        TS_REQUIRE(next_digests == nullptr, TS_ERR_INVALID, "fri_fold: no next round at h = 1");
        TS_LAUNCH(ctx, k_fri_fold_single, dim3(1), dim3(64), 0, in, h, tw, d_beta, out);
    }
    TS_HIP(hipGetLastError());
}

void launch_fri_fold(Context& ctx, const Ef* in, uint64_t h, Ef beta, Ef* out,
                     uint32_t* next_digests) {
    DevBuf<Ef> d_beta(&ctx, 1);
    TS_HIP(hipMemcpyAsync(d_beta.p, &beta, sizeof(Ef), hipMemcpyHostToDevice, ctx.stream));
    ctx.sync();  // `beta` is a stack temporary
    launch_fri_fold_dev(ctx, in, h, d_beta.p, out, next_digests);
}

// ---- one commit-phase round in one launch (merkle_tree.hpp) ------------------------------------
// leaf i of the round's matrix = (cur[2i], cur[2i+1]).  FOLD: cur is not in memory yet: it is the
// fold of the previous round's vector with the challenge that round's kernel left in device memory
// (cur[k] = fold(prev[2k], prev[2k+1])), computed, stored and hashed by the thread that owns the leaf.
static_assert(FRI_ROUND_MAX_LOG == mt::MAX_LOG_TREE, "kernels.hpp and merkle_tree.hpp disagree");

template <bool FOLD>
struct FriLeaves {
    const Ef* prev;
    const uint32_t* tw;   // tw[k] = g^-bitrev(k), the twiddle of output k of the fold
    Ef half_beta_mont;
    Ef* cur;
    uint32_t* level0;     // the tree's leaf digests
    __device__ __forceinline__ void fill(uint32_t* in, uint64_t node0, uint32_t count) {
        for (uint32_t n = threadIdx.x; n < count; n += mt::NTH) {
            const uint64_t i = node0 + n;
            Ef a, b;
            if (FOLD) {
                a = fold_one(load_ef(prev + 4 * i), load_ef(prev + 4 * i + 1), tw[2 * i], half_beta_mont, HALF_MONT);
                b = fold_one(load_ef(prev + 4 * i + 2), load_ef(prev + 4 * i + 3), tw[2 * i + 1], half_beta_mont,
                             HALF_MONT);
                store_ef(cur + 2 * i, a);
                store_ef(cur + 2 * i + 1, b);
            } else {
                a = load_ef(cur + 2 * i);
                b = load_ef(cur + 2 * i + 1);
            }
            uint32_t m[16] = {a.c[0], a.c[1], a.c[2], a.c[3], b.c[0], b.c[1], b.c[2], b.c[3],
                              0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t cv[8];
            b3::iv(cv);
            b3::compress(cv, m, 32, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
            uint4* o = reinterpret_cast<uint4*>(level0 + 8 * i);
            o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
            o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
            for (int k = 0; k < 8; k++) in[k * mt::CH + n] = cv[k];
        }
        b3::lds_barrier();
    }
};

template <bool FOLD>
__global__ void __launch_bounds__(mt::NTH)
k_fri_round(const Ef* __restrict__ prev, const uint32_t* __restrict__ tw, const Ef* __restrict__ beta_prev,
            Ef* __restrict__ cur, uint32_t* __restrict__ tree, unsigned log_leaves,
            uint32_t* __restrict__ ticket, DevChallenger* __restrict__ ch, uint32_t* __restrict__ root_out,
            Ef* __restrict__ beta_out) {
    __shared__ mt::Lds lds;
    __shared__ uint32_t s_last;
    const mt::Levels lv{tree, 0, (uint64_t)1 << log_leaves};
    FriLeaves<FOLD> prod{prev, tw, ef_zero(), cur, tree};
    if (FOLD) prod.half_beta_mont = ef_mul_base(ef_to_mont(load_ef(beta_prev)), HALF_MONT);
    mt::T9::tree_body(lds, s_last, prod, lv, log_leaves, ticket, ch, root_out, beta_out);
}

// The same round for a TALL vector, through the leaf-tree kernel (leaf_tree.hpp): a lane folds and
// hashes R leaves, the first log2(R) levels stay in its registers.  Before round 5 a round above
// 2^17 leaves was a fold launch, one launch per level and the tree launch.
template <bool FOLD>
struct FriLeaf {
    const Ef* prev;
    const uint32_t* tw;
    const Ef* beta_prev;
    Ef* cur;
    static const char* name(int lr) {
        static const char* const N[2][4] = {{"k_leaf_tree<0,fri_leaf>", "k_leaf_tree<1,fri_leaf>",
                                             "k_leaf_tree<2,fri_leaf>", "k_leaf_tree<3,fri_leaf>"},
                                            {"k_leaf_tree<0,fri_fold>", "k_leaf_tree<1,fri_fold>",
                                             "k_leaf_tree<2,fri_fold>", "k_leaf_tree<3,fri_fold>"}};
        return N[FOLD ? 1 : 0][lr];
    }
    __device__ __forceinline__ void digest(uint64_t i, uint32_t cv[8]) const {
        Ef a, b;
        if (FOLD) {
            const Ef half_beta_mont = ef_mul_base(ef_to_mont(load_ef(beta_prev)), HALF_MONT);
            a = fold_one(load_ef(prev + 4 * i), load_ef(prev + 4 * i + 1), tw[2 * i], half_beta_mont, HALF_MONT);
            b = fold_one(load_ef(prev + 4 * i + 2), load_ef(prev + 4 * i + 3), tw[2 * i + 1], half_beta_mont,
                         HALF_MONT);
            store_ef(cur + 2 * i, a);
            store_ef(cur + 2 * i + 1, b);
        } else {
            a = load_ef(cur + 2 * i);
            b = load_ef(cur + 2 * i + 1);
        }
        const uint32_t m[16] = {a.c[0], a.c[1], a.c[2], a.c[3], b.c[0], b.c[1], b.c[2], b.c[3],
                                0, 0, 0, 0, 0, 0, 0, 0};
        b3::iv(cv);
        b3::compress(cv, m, 32, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
    }
};

bool launch_fri_round_tall(Context& ctx, const Ef* prev, const Ef* d_beta_prev, Ef* cur, uint64_t h,
                           uint32_t* tree, DevChallenger* ch, uint32_t* root_out, Ef* beta_out,
                           uint64_t h_global, uint64_t row0) {
    if (h_global == 0) h_global = h;
    unsigned log_h = 0, log_hg = 0;
    while ((1ull << log_h) < h) log_h++;
    while ((1ull << log_hg) < h_global) log_hg++;
    TS_REQUIRE((1ull << log_h) == h && (1ull << log_hg) == h_global && row0 + h <= h_global, TS_ERR_INVALID,
               "fri_round_tall: leaf counts must be powers of two");
    if (!leaf_tree_enabled(log_h)) {  // the round-4 path: fold (+ leaf digests), levels, tree
        if (prev != nullptr)
            launch_fri_fold_dev(ctx, prev, 2 * h, d_beta_prev, cur, tree, 2 * h_global, 2 * row0);
        else
            launch_leaf_hash_ef_pairs(ctx, reinterpret_cast<const uint32_t*>(cur), h, tree);
        return launch_merkle_levels(ctx, tree, log_h, ch, root_out, beta_out);
    }
    if (prev != nullptr) {
        // the fold's output has 2 h_global elements (twiddles of order 4 h_global); this slab's first
        // output is global element 2 row0
        ctx.ensure_twiddles(log_hg + 2);
        launch_leaf_tree(ctx, FriLeaf<true>{prev, ctx.d_twiddle_inv + 2 * h_global + 2 * row0, d_beta_prev, cur},
                         tree, log_h, ch, root_out, beta_out);
    } else {
        launch_leaf_tree(ctx, FriLeaf<false>{nullptr, nullptr, nullptr, cur}, tree, log_h, ch, root_out, beta_out);
    }
    return ch != nullptr;
}

unsigned fri_round_max_log() {
    static const unsigned v = [] {
        const char* e = getenv("TS_FRI_ROUND_LOG");  // 0: never; up to 22
        const int x = e ? atoi(e) : 17;
        return (unsigned)(x < 0 ? 0 : x > (int)FRI_ROUND_MAX_LOG ? (int)FRI_ROUND_MAX_LOG : x);
    }();
    return v;
}

void launch_fri_round(Context& ctx, const Ef* prev, const Ef* d_beta_prev, Ef* cur, uint64_t h,
                      uint32_t* tree, DevChallenger* ch, uint32_t* root_out, Ef* beta_out) {
    unsigned log_h = 0;
    while ((1ull << log_h) < h) log_h++;
    TS_REQUIRE((1ull << log_h) == h && log_h <= FRI_ROUND_MAX_LOG, TS_ERR_INVALID,
               "fri_round: leaf count not a power of two <= 2^22");
    const dim3 grid(1u << (log_h - mt::block_log(log_h)));
    if (prev != nullptr) {
        ctx.ensure_twiddles(log_h + 2);  // the fold's output has 2h elements: twiddles of order 4h
        TS_LAUNCH(ctx, k_fri_round<true>, grid, dim3(mt::NTH), 0, prev, ctx.d_twiddle_inv + 2 * h, d_beta_prev,
                  cur, tree, log_h, ctx.ticket(), ch, root_out, beta_out);
    } else {
        TS_LAUNCH(ctx, k_fri_round<false>, grid, dim3(mt::NTH), 0, prev, (const uint32_t*)nullptr,
                  (const Ef*)nullptr, cur, tree, log_h, ctx.ticket(), ch, root_out, beta_out);
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
    uint32_t mi = 0;
    while (c >= mats.width[mi]) {
        c -= mats.width[mi];
        mi++;
    }
    const uint64_t row = ((uint64_t)indices[q] >> shift) >> mats.row_shift[mi];
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

// ------------------------------------------------------------------ device-side transcript
// One round of fri/src/prover.rs:113-116 without a host round trip: observe the root that the
// Merkle kernels just wrote, sample beta, leave both where the host will collect them later.
__global__ void k_chal_round(DevChallenger* __restrict__ ch, const uint32_t* __restrict__ root,
                             uint32_t* __restrict__ root_out, Ef* __restrict__ beta_out) {
    __shared__ DevChallenger lc;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t r[8];
    for (int i = 0; i < 8; i++) {
        r[i] = root[i];
        root_out[i] = r[i];
    }
    dc_copy(&lc, ch);
    Ef beta = dc_observe_root_and_sample(&lc, r);
    dc_copy(ch, &lc);
    store_ef(beta_out, beta);
}
void launch_chal_round(Context& ctx, DevChallenger* ch, const uint32_t* root, uint32_t* root_out,
                       Ef* beta_out) {
    TS_LAUNCH(ctx, k_chal_round, dim3(1), dim3(64), 0, ch, root, root_out, beta_out);
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ FRI tail
// Every remaining round of bf_commit_phase (fri/src/prover.rs:111-127) once the folded vector has
// at most 2^FRI_TAIL_LOG elements, in ONE workgroup: commit (leaf hashes + tree in LDS, also
// written to the tail arenas for the query phase), observe/sample on the device challenger, fold.
// Arena layout for tail round t (vector length L_t = L0 >> t): vector at sum_{j<t} L_j (in Ef),
// tree (L_t - 1 digests, leaves first) at sum_{j<t} (L_j - 1) (in digests).
constexpr int TAIL_NT = 512;
constexpr int TAIL_MAX = 1 << FRI_TAIL_LOG;
constexpr int TAIL_STRIDE = TAIL_MAX / 2;  // digest images: [word][node], one stride for both

// Every compression here is shared by four lanes (blake3_quad.hpp): the kernel is one long chain of
// dependent compressions (leaf, log2(h) levels, the sponge, per round), i.e. pure latency -- and with one
// wave per SIMD every instruction costs its four cycles whether it is on the chain or not.  So (round 5,
// in-kernel time stamps: 0.95 -> 0.6 us per level, 3.5 -> 0.8 us per sponge step):
//  * a lane's 28 message words are fetched in ONE batch (one LDS latency per compression instead of four),
//    at byte offsets it derives once per kernel: its parent index is folded in, and the ping / pong image is
//    a compile-time constant that goes into the instruction's offset field (the levels are unrolled in
//    pairs), so a level issues no address arithmetic at all;
//  * the sponge step is one more tree level on four lanes (dc_round_quad, chal_dev.hpp).
#ifdef TS_TAIL_STAMPS
__device__ unsigned long long g_stamps[512];
#define STAMP(i) do { if (threadIdx.x == 0) { g_stamps[2*(i)] = __builtin_amdgcn_s_memrealtime(); g_stamps[2*(i)+1] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" __attribute__((visibility("default"))) void ts_debug_stamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 512);
}
// the last k_fri_round launch (merkle_tree.hpp: TS_TREE_STAMP)
extern "C" __attribute__((visibility("default"))) void ts_debug_tree_stamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mt::g_tree_stamps), sizeof(unsigned long long) * 64);
}
#else
#define STAMP(i) do { } while (0)
#endif

// One tree level of the tail: n_par parents from image PAR to image PAR ^ 1 (and to the tree arena at `out`).
template <int PAR>
__device__ __forceinline__ void tail_level(uint32_t (*dig)[8 * TAIL_STRIDE], const uint32_t pm[28],
                                           const b3::QuadIv& node_iv, uint32_t n_par, uint32_t* __restrict__ out) {
    const uint32_t j = threadIdx.x & 3;
    const char* src = reinterpret_cast<const char*>(dig[PAR]);
    uint32_t* dst = dig[PAR ^ 1];
    // pass p takes parents 128 p + (lane >> 2): 1 KiB further on in every image row
    for (uint32_t q = threadIdx.x, off = 0; q < 4 * n_par; q += TAIL_NT, off += 4 * 2 * (TAIL_NT / 4)) {
        const uint32_t i = q >> 2;
        uint32_t m[28];
        if (off == 0) {
#pragma unroll
            for (int k = 0; k < 28; k++) m[k] = *reinterpret_cast<const uint32_t*>(src + pm[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 28; k++) m[k] = *reinterpret_cast<const uint32_t*>(src + pm[k] + off);
        }
        uint32_t lo, hi;
        b3::compress_quad(node_iv, [&](int k) { return m[k]; }, lo, hi);
        uint32_t* o = out + 8 * (uint64_t)i;
        o[j] = lo;
        o[4 + j] = hi;
        dst[j * TAIL_STRIDE + i] = lo;
        dst[(4 + j) * TAIL_STRIDE + i] = hi;
    }
    b3::lds_barrier();
}

__global__ void __launch_bounds__(TAIL_NT)
k_fri_tail(const Ef* __restrict__ in, uint32_t L0, uint32_t blowup, DevChallenger* __restrict__ ch,
           const uint32_t* __restrict__ Winv, Ef* __restrict__ tail_vecs,
           uint32_t* __restrict__ tail_trees, uint32_t* __restrict__ roots_out,
           Ef* __restrict__ betas_out, Ef* __restrict__ final_out, uint32_t pow_bits,
           uint32_t* __restrict__ pow_out, const Ef* __restrict__ beta_in) {
    __shared__ Ef bufA[TAIL_MAX];
    __shared__ Ef bufB[TAIL_MAX / 2];
    __shared__ uint32_t dig[2][8 * TAIL_STRIDE];  // digest images [word][node], used in turn
    __shared__ Ef s_beta;
    __shared__ DevChallenger s_ch;  // the sponge's working copy (chal_dev.hpp); wave 0 only
    if (threadIdx.x == 0) dc_copy(&s_ch, ch);
    const uint32_t j = threadIdx.x & 3;
    // pm: byte offsets in an image of the 28 message words of parent lane >> 2 (children 2i, 2i + 1);
    // lidx / lmask: leaf word indices and their validity (a leaf is 8 words, the rest of the block zero)
    uint32_t pm[28], lidx[28], lmask = 0;
    {
        uint32_t idx[28];
        b3::quad_schedule(j, idx);
#pragma unroll
        for (int k = 0; k < 28; k++) {
            pm[k] = 4 * ((idx[k] & 7) * TAIL_STRIDE + (idx[k] >> 3) + 2 * (threadIdx.x >> 2));
            lidx[k] = idx[k] & 7;
            lmask |= (idx[k] < 8 ? 1u : 0u) << k;
        }
    }
    const b3::QuadIv node_iv = b3::quad_iv(j, 64, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
    const b3::QuadIv leaf_iv = b3::quad_iv(j, 32, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
    Ef* cur = bufA;
    Ef* nxt = bufB;
    STAMP(0);
    if (beta_in != nullptr) {
        // `in` is the previous round's vector (2 L0 elements): its fold with that round's challenge is this
        // kernel's first vector (one launch less on the chain than a fold kernel in front)
        const Ef hb = ef_mul_base(ef_to_mont(load_ef(beta_in)), HALF_MONT);
        for (uint32_t i = threadIdx.x; i < L0; i += TAIL_NT)
            cur[i] = fold_one(load_ef(in + 2 * i), load_ef(in + 2 * i + 1), Winv[L0 + i], hb, HALF_MONT);
    } else {
        for (uint32_t i = threadIdx.x; i < L0; i += TAIL_NT) cur[i] = load_ef(in + i);
    }
    __syncthreads();
    STAMP(1);
    uint32_t L = L0, voff = 0, toff = 0, t = 0;
    while (L > blowup) {
        const uint32_t h = L >> 1;
        STAMP(2 + 5 * t);
        for (uint32_t i = threadIdx.x; i < L; i += TAIL_NT) store_ef(tail_vecs + voff + i, cur[i]);
        // leaves: rows (cur[2i], cur[2i+1]) = 8 words, one short block
        for (uint32_t q = threadIdx.x; q < 4 * h; q += TAIL_NT) {
            const uint32_t i = q >> 2;
            const uint32_t* row = reinterpret_cast<const uint32_t*>(cur + 2 * i);
            uint32_t lo, hi;
            uint32_t m[28];
#pragma unroll
            for (int k = 0; k < 28; k++) m[k] = ((lmask >> k) & 1u) ? row[lidx[k]] : 0u;
            b3::compress_quad(leaf_iv, [&](int k) { return m[k]; }, lo, hi);
            uint32_t* o = tail_trees + 8 * (uint64_t)(toff + i);
            o[j] = lo;
            o[4 + j] = hi;
            dig[0][j * TAIL_STRIDE + i] = lo;
            dig[0][(4 + j) * TAIL_STRIDE + i] = hi;
        }
        b3::lds_barrier();
        STAMP(3 + 5 * t);
        uint32_t n = h, lvl_off = toff, par = 0;
        while (n > 1) {
            tail_level<0>(dig, pm, node_iv, n >> 1, tail_trees + 8 * (uint64_t)(lvl_off + n));
            lvl_off += n;
            n >>= 1;
            par = 1;
            if (n <= 1) break;
            tail_level<1>(dig, pm, node_iv, n >> 1, tail_trees + 8 * (uint64_t)(lvl_off + n));
            lvl_off += n;
            n >>= 1;
            par = 0;
        }
        STAMP(4 + 5 * t);
        // the root is node 0 of image `par`: observe it, sample beta (fri/src/prover.rs:113-116)
        if (threadIdx.x < 64) dc_round_quad<TAIL_STRIDE>(&s_ch, dig[par], pm, node_iv, roots_out + 8 * t, &s_beta, betas_out + t);
        b3::lds_barrier();
        STAMP(5 + 5 * t);
        const Ef half_beta_mont = ef_mul_base(ef_to_mont(s_beta), HALF_MONT);
        for (uint32_t i = threadIdx.x; i < h; i += TAIL_NT)
            nxt[i] = fold_one(cur[2 * i], cur[2 * i + 1], Winv[h + i], half_beta_mont, HALF_MONT);
        b3::lds_barrier();
        STAMP(6 + 5 * t);
        Ef* tmp = cur;
        cur = nxt;
        nxt = tmp;
        voff += L;
        toff += L - 1;
        L = h;
        t++;
    }
    for (uint32_t i = threadIdx.x; i < L; i += TAIL_NT) store_ef(final_out + i, cur[i]);
    STAMP(100);
    if (threadIdx.x == 0) dc_copy(ch, &s_ch);  // wave 0 made every change to the copy
    // The proof-of-work witness (fri/src/prover.rs:43, basic/src/challenger/mod.rs:95-114): the smallest w
    // < 4096 for which a CLONE of the transcript, after observing (w, 0 x 7), samples `pow_bits` zero bits.
    // With an empty input buffer those eight words fill it exactly: one permutation of (w, 0 x 7, capacity
    // half), and the sample is the digest's last word mod p.  On the host that is ~256 sponge steps one
    // after another (30-40 us of every proof); here every lane tries one candidate.  The transcript itself
    // does not move: the host checks the hint with one step of its own and grinds itself if it must.
    if (pow_out != nullptr) {
        __shared__ uint32_t s_pow;
        const bool can = s_ch.n_in == 0 && s_ch.permutation == 0;  // uniform
        if (threadIdx.x == 0) s_pow = FRI_POW_NONE;
        b3::lds_barrier();
        if (can) {
            uint32_t m[16], d[8];
#pragma unroll
            for (int i = 1; i < 8; i++) m[i] = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) m[8 + i] = s_ch.state[8 + i];
            // 256 candidates a pass -- one wave per SIMD, 1.4 us; a pass finds one 19 times in 20 at 8 bits
            constexpr uint32_t PER_PASS = 256;
            for (uint32_t w0 = 0; w0 < (1u << 12); w0 += PER_PASS) {
                if (threadIdx.x < PER_PASS) {
                    m[0] = w0 + threadIdx.x;
                    b3::hash64(m, d);
                    if (pow_bits == 0 || ((d[7] % P) >> (32 - pow_bits)) == 0) atomicMin(&s_pow, w0 + threadIdx.x);
                }
                b3::lds_barrier();
                const bool found = s_pow != FRI_POW_NONE;  // uniform: every add is behind the barrier
                b3::lds_barrier();                          // ... and nobody adds again before all have read
                if (found) break;
            }
        }
        if (threadIdx.x == 0) *pow_out = s_pow;
    }
    STAMP(101);
}

void launch_fri_tail(Context& ctx, const Ef* in, uint32_t L0, uint32_t blowup, DevChallenger* ch,
                     Ef* tail_vecs, uint32_t* tail_trees, uint32_t* roots_out, Ef* betas_out,
                     Ef* final_out, uint32_t pow_bits, uint32_t* pow_out, const Ef* beta_in) {
    TS_REQUIRE(pow_bits <= 31, TS_ERR_INVALID, "fri_tail: proof-of-work bits > 31");
    TS_REQUIRE(L0 <= (uint32_t)TAIL_MAX && L0 >= 1, TS_ERR_INVALID, "fri_tail: vector too long");
    unsigned log_l = 0;
    while ((1u << log_l) < L0) log_l++;
    ctx.ensure_twiddles(log_l + (beta_in ? 1 : 0) == 0 ? 1 : log_l + (beta_in ? 1 : 0));
    TS_LAUNCH(ctx, k_fri_tail, dim3(1), dim3(TAIL_NT), 0, in, L0, blowup, ch,
              (const uint32_t*)ctx.d_twiddle_inv, tail_vecs, tail_trees, roots_out, betas_out,
              final_out, pow_bits, pow_out, beta_in);
    TS_HIP(hipGetLastError());
}

// one descriptor's share: per query 8 value words (the row of two EF4; skipped when vec == nullptr) then
// 8*log_leaves path words
__device__ __forceinline__ void gather_desc(const FriGatherDesc d, const uint32_t* __restrict__ indices,
                                            uint32_t n_idx, uint32_t* __restrict__ out) {
    const uint32_t per_q = 8 + 8 * d.log_leaves;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * per_q) return;
    const uint32_t q = t / per_q, e = t % per_q;
    const uint64_t row = indices[q] >> d.shift;
    if (e < 8) {
        if (d.vec != nullptr) out[d.out_vals + (uint64_t)q * 8 + e] = d.vec[8 * row + e];
    } else {
        const uint32_t l = (e - 8) >> 3, word = (e - 8) & 7;
        uint64_t off = 0;
        for (unsigned k = 0; k < l; k++) off += (uint64_t)1 << (d.log_leaves - k);
        const uint64_t node = off + ((row >> l) ^ 1);
        out[d.out_path + ((uint64_t)q * d.log_leaves + l) * 8 + word] = d.tree[8 * node + word];
    }
}
// every commit-phase opening of every query in one launch (bf_answer_query, fri/src/prover.rs:69-90):
// blockIdx.y = round
__global__ void k_gather_fri(const FriGatherDesc* __restrict__ descs, const uint32_t* __restrict__ indices,
                             uint32_t n_idx, uint32_t* __restrict__ out) {
    gather_desc(descs[blockIdx.y], indices, n_idx, out);
}

// The query phase of one proof in one launch (kernels.hpp): blockIdx.y < n_rows: the opened rows of
// committed batch blockIdx.y (k_gather_rows' work, the matrix table read from device memory); then one
// descriptor each.
__global__ void k_gather_queries(const RowGatherJob* __restrict__ rows, uint32_t n_rows,
                                 const FriGatherDesc* __restrict__ descs, const uint32_t* __restrict__ indices,
                                 uint32_t n_idx, uint32_t* __restrict__ out) {
    if (blockIdx.y >= n_rows) {
        gather_desc(descs[blockIdx.y - n_rows], indices, n_idx, out);
        return;
    }
    const RowGatherJob& jb = rows[blockIdx.y];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = jb.mats.total_width;
    if (t >= n_idx * total) return;
    const uint32_t q = t / total;
    uint32_t c = t % total;
    uint32_t mi = 0;
    while (c >= jb.mats.width[mi]) {
        c -= jb.mats.width[mi];
        mi++;
    }
    const uint64_t row = ((uint64_t)indices[q] >> jb.shift) >> jb.mats.row_shift[mi];
    out[jb.out + t] = jb.mats.d[mi][(uint64_t)c * jb.mats.col_stride[mi] + row];
}
void launch_gather_queries(Context& ctx, const RowGatherJob* d_rows, uint32_t n_rows, uint32_t max_row_width,
                           const FriGatherDesc* d_descs, uint32_t n_descs, uint32_t max_log_leaves,
                           const uint32_t* d_indices, uint32_t n_idx, uint32_t* out) {
    if (!n_idx || n_rows + n_descs == 0) return;
    const uint32_t per_q = std::max(n_descs ? 8 + 8 * max_log_leaves : 0u, n_rows ? max_row_width : 0u);
    if (!per_q) return;
    TS_LAUNCH(ctx, k_gather_queries, dim3((n_idx * per_q + 255) / 256, n_rows + n_descs), dim3(256), 0, d_rows,
              n_rows, d_descs, d_indices, n_idx, out);
    TS_HIP(hipGetLastError());
}

void launch_gather_fri(Context& ctx, const FriGatherDesc* d_descs, uint32_t n_rounds,
                       uint32_t max_log_leaves, const uint32_t* d_indices, uint32_t n_idx,
                       uint32_t* out) {
    if (!n_rounds || !n_idx) return;
    const uint32_t per_q = 8 + 8 * max_log_leaves;
    TS_LAUNCH(ctx, k_gather_fri, dim3((n_idx * per_q + 255) / 256, n_rounds), dim3(256), 0, d_descs,
              d_indices, n_idx, out);
    TS_HIP(hipGetLastError());
}

}  // namespace ts
