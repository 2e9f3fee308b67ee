// Blake3 Merkle MMCS kernels (build-defined spec; there is no Merkle tree
// in the reference, whose BFMmcs is a Bitcoin taptree: basic/src/mmcs/taptree_mmcs.rs:101-114).
//   leaf  = Blake3(row of matrix 0 || row of matrix 1 || ...), elements as canonical u32 LE
//   node  = Blake3(left || right)
// Leaves: one thread per row; column-major matrices make every column read a coalesced 256 B
// per wavefront.  Digests are stored as 8 consecutive words per node, levels back to back.
#include <stdlib.h>

#include "blake3.hpp"
#include "blake3_quad.hpp"
#include "merkle_tree.hpp"
#include "leaf_tree.hpp"
#include "chal_dev.hpp"
#include "kernels.hpp"

namespace ts {

// ROWS rows per thread (r, r + height/ROWS, ...): independent Blake3 chains in one thread give the
// scheduler something to overlap with each chain's long dependency path
template <int ROWS>
__global__ void __launch_bounds__(256)
k_leaf_hash(const uint32_t* const* __restrict__ cols, uint32_t total, uint64_t height,
            uint32_t* __restrict__ digests) {
    const uint64_t per = height / ROWS;  // host guarantees divisibility for ROWS > 1
    const uint64_t r0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r0 >= per) return;
    uint32_t cv[ROWS][8];
#pragma unroll
    for (int k = 0; k < ROWS; k++) b3::iv(cv[k]);
    const uint32_t n_blocks = total == 0 ? 1 : (total + 15) / 16;
    // walk the concatenated row 16 words (one Blake3 block) at a time; the 16 loads of a block are
    // independent and issue back to back
    for (uint32_t blk = 0; blk < n_blocks; blk++) {
        uint32_t m[ROWS][16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t c = blk * 16 + j;
            const uint32_t* col = c < total ? cols[c] : nullptr;
#pragma unroll
            for (int k = 0; k < ROWS; k++) m[k][j] = col ? col[r0 + (uint64_t)k * per] : 0u;
        }
        uint32_t words = total - blk * 16 < 16 ? total - blk * 16 : 16;
        uint32_t flags = (blk == 0 ? b3::CHUNK_START : 0u) |
                         (blk + 1 == n_blocks ? (b3::CHUNK_END | b3::ROOT) : 0u);
#pragma unroll
        for (int k = 0; k < ROWS; k++) b3::compress(cv[k], m[k], words * 4, flags);
    }
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        uint4* o = reinterpret_cast<uint4*>(digests + 8 * (r0 + (uint64_t)k * per));
        o[0] = make_uint4(cv[k][0], cv[k][1], cv[k][2], cv[k][3]);
        o[1] = make_uint4(cv[k][4], cv[k][5], cv[k][6], cv[k][7]);
    }
}

// one matrix (width <= 256): column c of the row is base[c * stride + r] -- no pointer table, no
// per-word bounds checks; n_full whole 64-byte blocks, then (rem != 0) one short block of rem words.
// (Round 3: widths that are not multiples of 16 took the pointer-table kernel before -- the 163-column
// trace of config 5 hashed at 0.74 of the Blake3 rate against 0.97 here.)
__global__ void __launch_bounds__(256)
k_leaf_hash_strided(const uint32_t* __restrict__ base, uint64_t stride, uint32_t n_full, uint32_t rem,
                    uint64_t height, uint32_t* __restrict__ digests) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= height) return;
    uint32_t cv[8];
    b3::iv(cv);
    const uint32_t* p = base + r;
    for (uint32_t blk = 0; blk < n_full; blk++) {
        uint32_t m[16];
#pragma unroll
        for (int j = 0; j < 16; j++) m[j] = p[(uint64_t)j * stride];
        p += 16 * stride;
        const uint32_t flags = (blk == 0 ? b3::CHUNK_START : 0u) |
                               (blk + 1 == n_full && rem == 0 ? (b3::CHUNK_END | b3::ROOT) : 0u);
        b3::compress(cv, m, 64, flags);
    }
    if (rem != 0) {
        uint32_t m[16];
#pragma unroll
        for (int j = 0; j < 16; j++) m[j] = (uint32_t)j < rem ? p[(uint64_t)j * stride] : 0u;
        b3::compress(cv, m, rem * 4, (n_full == 0 ? b3::CHUNK_START : 0u) | b3::CHUNK_END | b3::ROOT);
    }
    uint4* o = reinterpret_cast<uint4*>(digests + 8 * r);
    o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// rows wider than one Blake3 chunk (256 elements): chunk chaining + parent tree per row
// (b3::hash_stream); the subtree stack is indexed by a wave-uniform depth and lives in scratch.
// Rare shape (bf_mmcs.rs:17-68 allows any width), kept simple: one thread per row.
__global__ void __launch_bounds__(256)
k_leaf_hash_wide(const uint32_t* const* __restrict__ cols, uint32_t total, uint64_t height,
                 uint32_t* __restrict__ digests) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= height) return;
    uint32_t cv[8];
    b3::hash_stream([cols, r](uint64_t c) { return cols[c][r]; }, total, cv);
    uint4* o = reinterpret_cast<uint4*>(digests + 8 * r);
    o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

void launch_leaf_hash(Context& ctx, const LeafMats& mats, uint64_t height, uint32_t* digests) {
    TS_REQUIRE(mats.total_width <= (1u << 20), TS_ERR_UNSUPPORTED,
               "leaf rows wider than 2^20 field elements are not supported");
    TS_REQUIRE(mats.cols != nullptr, TS_ERR_INVALID, "leaf_hash: column pointer table missing");
    if (mats.total_width > 256) {
        TS_LAUNCH(ctx, k_leaf_hash_wide, dim3((unsigned)((height + 255) / 256)), dim3(256), 0, mats.cols,
                  mats.total_width, height, digests);
        TS_HIP(hipGetLastError());
        return;
    }
    static const int rows_per_thread = [] {
        const char* e = getenv("TS_LEAF_ROWS");
        return e ? atoi(e) : 1;
    }();
    static const int strided = [] {
        const char* e = getenv("TS_LEAF_STRIDED");
        return e ? atoi(e) : 1;
    }();
    if (strided && mats.n_mats == 1 && mats.d[0] != nullptr && mats.total_width >= 1) {
        TS_LAUNCH(ctx, k_leaf_hash_strided, dim3((unsigned)((height + 255) / 256)), dim3(256), 0, mats.d[0],
                  mats.col_stride[0], mats.total_width / 16, mats.total_width % 16, height, digests);
    } else if (rows_per_thread == 2 && height % 2 == 0 && height >= (1u << 16)) {
        TS_LAUNCH(ctx, k_leaf_hash<2>, dim3((unsigned)((height / 2 + 255) / 256)), dim3(256), 0, mats.cols,
                  mats.total_width, height, digests);
    } else {
        TS_LAUNCH(ctx, k_leaf_hash<1>, dim3((unsigned)((height + 255) / 256)), dim3(256), 0, mats.cols,
                  mats.total_width, height, digests);
    }
    TS_HIP(hipGetLastError());
}

// rows of two EF4 (32 bytes) -> one short block each
__global__ void __launch_bounds__(256)
k_leaf_hash_ef_pairs(const uint4* __restrict__ vec, uint64_t n_rows, uint32_t* __restrict__ digests) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    uint4 a = vec[2 * r], b = vec[2 * r + 1];
    uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t cv[8];
    b3::iv(cv);
    b3::compress(cv, m, 32, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
    uint4* o = reinterpret_cast<uint4*>(digests + 8 * r);
    o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

void launch_leaf_hash_ef_pairs(Context& ctx, const uint32_t* vec, uint64_t n_rows, uint32_t* digests) {
    TS_LAUNCH(ctx, k_leaf_hash_ef_pairs, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<const uint4*>(vec), n_rows, digests);
    TS_HIP(hipGetLastError());
}

// one level: parents[i] = Blake3(children[2i] || children[2i+1]); PAR parents per thread
// (i, i + n/PAR, ...: independent chains, as in k_leaf_hash)
template <int PAR>
__global__ void __launch_bounds__(256)
k_merkle_level(const uint4* __restrict__ children, uint4* __restrict__ parents, uint64_t n_parents) {
    const uint64_t per = n_parents / PAR;
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= per) return;
    uint32_t m[PAR][16], cv[PAR][8];
#pragma unroll
    for (int k = 0; k < PAR; k++) {
        const uint64_t i = i0 + (uint64_t)k * per;
        const uint4 a = children[4 * i], b = children[4 * i + 1], c = children[4 * i + 2], d = children[4 * i + 3];
        m[k][0] = a.x; m[k][1] = a.y; m[k][2] = a.z; m[k][3] = a.w;
        m[k][4] = b.x; m[k][5] = b.y; m[k][6] = b.z; m[k][7] = b.w;
        m[k][8] = c.x; m[k][9] = c.y; m[k][10] = c.z; m[k][11] = c.w;
        m[k][12] = d.x; m[k][13] = d.y; m[k][14] = d.z; m[k][15] = d.w;
    }
#pragma unroll
    for (int k = 0; k < PAR; k++) b3::hash64(m[k], cv[k]);
#pragma unroll
    for (int k = 0; k < PAR; k++) {
        const uint64_t i = i0 + (uint64_t)k * per;
        parents[2 * i] = make_uint4(cv[k][0], cv[k][1], cv[k][2], cv[k][3]);
        parents[2 * i + 1] = make_uint4(cv[k][4], cv[k][5], cv[k][6], cv[k][7]);
    }
}

// (A variant with fully coalesced traffic -- 16-byte loads at consecutive addresses into LDS, each
// thread then picking up its 64 bytes, digests leaving the same way -- measured 0.421 against 0.410 ms
// per C3 proof: the access pattern is not what holds these launches at 2.8 TB/s; most of the 27 per
// proof are short levels of 2^16..2^18 parents whose time is launch and tail latency.)
static void launch_level(Context& ctx, const uint32_t* children, uint32_t* parents, uint64_t n_parents) {
    static const int par = [] {
        // two parents per thread measured no better inside whole proofs (3.32 vs 3.27-3.33 ms/step);
        // like TS_LEAF_ROWS this stays a knob for experiments
        const char* e = getenv("TS_LEVEL_PAR");
        return e ? atoi(e) : 1;
    }();
    if (par == 2 && n_parents >= (1u << 18) && n_parents % 2 == 0)
        TS_LAUNCH(ctx, k_merkle_level<2>, dim3((unsigned)((n_parents / 2 + 255) / 256)), dim3(256), 0,
                  reinterpret_cast<const uint4*>(children), reinterpret_cast<uint4*>(parents), n_parents);
    else
        TS_LAUNCH(ctx, k_merkle_level<1>, dim3((unsigned)((n_parents + 255) / 256)), dim3(256), 0,
                  reinterpret_cast<const uint4*>(children), reinterpret_cast<uint4*>(parents), n_parents);
}

void launch_merkle_one_level(Context& ctx, const uint32_t* children, uint32_t* parents,
                             uint64_t n_parents) {
    launch_level(ctx, children, parents, n_parents);
    TS_HIP(hipGetLastError());
}

// mixed-height injection: nodes[i] = Blake3(nodes[i] || inj[i])
__global__ void __launch_bounds__(256)
k_merkle_inject(uint4* __restrict__ nodes, const uint4* __restrict__ inj, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 a = nodes[2 * i], b = nodes[2 * i + 1], c = inj[2 * i], d = inj[2 * i + 1];
    uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t cv[8];
    b3::hash64(m, cv);
    nodes[2 * i] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    nodes[2 * i + 1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}
void launch_merkle_inject(Context& ctx, uint32_t* nodes, const uint32_t* inj, uint64_t n) {
    TS_LAUNCH(ctx, k_merkle_inject, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
              reinterpret_cast<uint4*>(nodes), reinterpret_cast<const uint4*>(inj), n);
    TS_HIP(hipGetLastError());
}

// Top of a sharded tree: level 0 = the G gathered sub-tree roots, then log2(G) levels; G is small
// (<= 64), so one wavefront does a level at a time.
__global__ void __launch_bounds__(64)
k_shard_top(const uint32_t* __restrict__ subroots, uint32_t G, uint32_t* __restrict__ top,
            DevChallenger* __restrict__ ch, uint32_t* __restrict__ root_out, Ef* __restrict__ beta_out) {
    for (uint32_t i = threadIdx.x; i < 8 * G; i += 64) top[i] = subroots[i];
    __threadfence_block();
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t cnt = G; cnt > 1; cnt >>= 1) {
        const uint32_t i = threadIdx.x;
        if (i < cnt / 2) {
            uint32_t m[16], cv[8];
            for (int k = 0; k < 16; k++) m[k] = top[8 * (off + 2 * i) + k];
            b3::hash64(m, cv);
            for (int k = 0; k < 8; k++) top[8 * (off + cnt + i) + k] = cv[k];
        }
        off += cnt;
        __threadfence_block();
        __syncthreads();
    }
    if (ch != nullptr && threadIdx.x == 0) {
        uint32_t root[8];
        for (int k = 0; k < 8; k++) {
            root[k] = top[8 * off + k];
            root_out[k] = root[k];
        }
        __shared__ DevChallenger lc;  // the sponge runs on a copy in LDS (chal_dev.hpp)
        dc_copy(&lc, ch);
        const Ef beta = dc_observe_root_and_sample(&lc, root);
        dc_copy(ch, &lc);
        *reinterpret_cast<uint4*>(beta_out) = make_uint4(beta.c[0], beta.c[1], beta.c[2], beta.c[3]);
    }
}
void launch_shard_top(Context& ctx, const uint32_t* subroots, uint32_t G, uint32_t* top,
                      DevChallenger* ch, uint32_t* root_out, Ef* beta_out) {
    TS_REQUIRE(G >= 1 && G <= 64 && (G & (G - 1)) == 0, TS_ERR_INVALID, "shard_top: G must be 2^k <= 64");
    TS_LAUNCH(ctx, k_shard_top, dim3(1), dim3(64), 0, subroots, G, top, ch, root_out, beta_out);
    TS_HIP(hipGetLastError());
}

// Up to 2^22 first-level nodes: the whole tree in ONE launch (merkle_tree.hpp).  Taller trees run
// their first levels one launch per level (bandwidth-bound there, ~3.7 TB/s of digest traffic).
__global__ void __launch_bounds__(mt::NTH)
k_merkle_tree(uint32_t* __restrict__ tree, unsigned log_leaves, unsigned first_level,
              uint32_t* __restrict__ ticket, DevChallenger* __restrict__ ch,
              uint32_t* __restrict__ root_out, Ef* __restrict__ beta_out) {
    __shared__ mt::Lds lds;
    __shared__ uint32_t s_last;
    uint64_t off = 0;
    for (unsigned l = 0; l < first_level; l++) off += (uint64_t)1 << (log_leaves - l);
    const unsigned remaining = log_leaves - first_level;
    const mt::Levels lv{tree, off, (uint64_t)1 << remaining};
    mt::T9::StagedNodes<false> prod{lv.at(0, 0)};
    mt::T9::tree_body(lds, s_last, prod, lv, remaining, ticket, ch, root_out, beta_out);
}

void launch_merkle_tree_from(Context& ctx, uint32_t* tree, unsigned log_leaves, unsigned first_level,
                             DevChallenger* ch, uint32_t* root_out, Ef* beta_out) {
    TS_REQUIRE(first_level < log_leaves && log_leaves - first_level <= mt::MAX_LOG_TREE, TS_ERR_INVALID,
               "merkle_tree_from: between 2 and 2^22 nodes in the first level");
    const unsigned remaining = log_leaves - first_level;
    TS_LAUNCH(ctx, k_merkle_tree, dim3(1u << (remaining - mt::block_log(remaining))), dim3(mt::NTH), 0, tree,
              log_leaves, first_level, ctx.ticket(), ch, root_out, beta_out);
    TS_HIP(hipGetLastError());
}

// ---- leaves + tree in one launch (leaf_tree.hpp) ---------------------------------------------------
// the leaf hashes of k_leaf_hash_strided / k_leaf_hash<1> / k_leaf_hash_ef_pairs as per-row functors
struct StridedLeaf {
    const uint32_t* base;
    uint64_t stride;
    uint32_t n_full, rem;
    static const char* name(int lr) {
        static const char* const N[4] = {"k_leaf_tree<0,strided>", "k_leaf_tree<1,strided>", "k_leaf_tree<2,strided>",
                                         "k_leaf_tree<3,strided>"};
        return N[lr];
    }
    __device__ __forceinline__ void digest(uint64_t r, uint32_t cv[8]) const {
        b3::iv(cv);
        const uint32_t* p = base + r;
        for (uint32_t blk = 0; blk < n_full; blk++) {
            uint32_t m[16];
#pragma unroll
            for (int j = 0; j < 16; j++) m[j] = p[(uint64_t)j * stride];
            p += 16 * stride;
            const uint32_t flags = (blk == 0 ? b3::CHUNK_START : 0u) |
                                   (blk + 1 == n_full && rem == 0 ? (b3::CHUNK_END | b3::ROOT) : 0u);
            b3::compress(cv, m, 64, flags);
        }
        if (rem != 0) {
            uint32_t m[16];
#pragma unroll
            for (int j = 0; j < 16; j++) m[j] = (uint32_t)j < rem ? p[(uint64_t)j * stride] : 0u;
            b3::compress(cv, m, rem * 4, (n_full == 0 ? b3::CHUNK_START : 0u) | b3::CHUNK_END | b3::ROOT);
        }
    }
};

struct TableLeaf {
    const uint32_t* const* cols;
    uint32_t total;
    static const char* name(int lr) {
        static const char* const N[4] = {"k_leaf_tree<0,table>", "k_leaf_tree<1,table>", "k_leaf_tree<2,table>",
                                         "k_leaf_tree<3,table>"};
        return N[lr];
    }
    __device__ __forceinline__ void digest(uint64_t r, uint32_t cv[8]) const {
        b3::iv(cv);
        const uint32_t n_blocks = total == 0 ? 1 : (total + 15) / 16;
        for (uint32_t blk = 0; blk < n_blocks; blk++) {
            uint32_t m[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t c = blk * 16 + j;
                // a pointer read from memory is a generic one to the compiler (flat_load: an aperture
                // check per access); these are device allocations: say so
                typedef const uint32_t __attribute__((address_space(1))) * gptr;
                m[j] = c < total ? ((gptr)cols[c])[r] : 0u;
            }
            const uint32_t words = total - blk * 16 < 16 ? total - blk * 16 : 16;
            const uint32_t flags = (blk == 0 ? b3::CHUNK_START : 0u) |
                                   (blk + 1 == n_blocks ? (b3::CHUNK_END | b3::ROOT) : 0u);
            b3::compress(cv, m, words * 4, flags);
        }
    }
};

struct EfPairLeaf {
    const uint4* vec;
    static const char* name(int lr) {
        static const char* const N[4] = {"k_leaf_tree<0,ef_pairs>", "k_leaf_tree<1,ef_pairs>",
                                         "k_leaf_tree<2,ef_pairs>", "k_leaf_tree<3,ef_pairs>"};
        return N[lr];
    }
    __device__ __forceinline__ void digest(uint64_t r, uint32_t cv[8]) const {
        const uint4 a = vec[2 * r], b = vec[2 * r + 1];
        const uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, 0, 0, 0, 0, 0, 0, 0, 0};
        b3::iv(cv);
        b3::compress(cv, m, 32, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
    }
};

bool leaf_tree_enabled(unsigned log_leaves) {
    static const int on = [] {
        const char* e = getenv("TS_LEAF_TREE");  // 0: the round-4 path (leaf launch, level launches, tree launch)
        return e ? atoi(e) : 1;
    }();
    return on != 0 && log_leaves >= mt::LEAF_TREE_MIN_LOG;
}

bool launch_commit_tree(Context& ctx, const LeafMats& mats, unsigned log_leaves, uint32_t* tree,
                        DevChallenger* ch, uint32_t* root_out, Ef* beta_out) {
    const uint64_t height = (uint64_t)1 << log_leaves;
    if (!leaf_tree_enabled(log_leaves) || mats.total_width > 256) {
        launch_leaf_hash(ctx, mats, height, tree);
        return launch_merkle_levels(ctx, tree, log_leaves, ch, root_out, beta_out);
    }
    TS_REQUIRE(mats.cols != nullptr, TS_ERR_INVALID, "commit_tree: column pointer table missing");
    if (mats.n_mats == 1 && mats.d[0] != nullptr && mats.total_width >= 1)
        launch_leaf_tree(ctx, StridedLeaf{mats.d[0], mats.col_stride[0], mats.total_width / 16, mats.total_width % 16},
                         tree, log_leaves, ch, root_out, beta_out);
    else
        launch_leaf_tree(ctx, TableLeaf{mats.cols, mats.total_width}, tree, log_leaves, ch, root_out, beta_out);
    return ch != nullptr;
}

bool launch_commit_tree_ef_pairs(Context& ctx, const uint32_t* vec, unsigned log_leaves, uint32_t* tree,
                                 DevChallenger* ch, uint32_t* root_out, Ef* beta_out) {
    if (!leaf_tree_enabled(log_leaves)) {
        launch_leaf_hash_ef_pairs(ctx, vec, (uint64_t)1 << log_leaves, tree);
        return launch_merkle_levels(ctx, tree, log_leaves, ch, root_out, beta_out);
    }
    launch_leaf_tree(ctx, EfPairLeaf{reinterpret_cast<const uint4*>(vec)}, tree, log_leaves, ch, root_out, beta_out);
    return ch != nullptr;
}

// Measured (tools/time_tree.py, us per tree above the leaves, one launch / per-level launches down to
// 2^16): 2^17 34 / 33, 2^18 60 / 44, 2^20 89 / 71, 2^22 216 / 152.  In one launch the bulk levels run
// as four-lane compressions out of LDS between workgroup barriers at 3 workgroups per CU, ~19 G
// compressions/s; k_merkle_level streams them at ~34 G/s.  So the single launch takes over at 2^17.
unsigned merkle_tree_max_log() {
    static const unsigned v = [] {
        const char* e = getenv("TS_TREE_MAX_LOG");  // up to 22, for A/B runs
        const int x = e ? atoi(e) : 17;
        return (unsigned)(x < 1 ? 1 : x > (int)mt::MAX_LOG_TREE ? (int)mt::MAX_LOG_TREE : x);
    }();
    return v;
}

bool launch_merkle_levels(Context& ctx, uint32_t* tree, unsigned log_leaves, DevChallenger* ch,
                          uint32_t* root_out, Ef* beta_out) {
    unsigned level = 0;
    uint64_t off = 0;
    while (log_leaves - level > merkle_tree_max_log()) {
        const uint64_t n_children = (uint64_t)1 << (log_leaves - level);
        const uint64_t n_parents = n_children / 2;
        launch_level(ctx, tree + 8 * off, tree + 8 * (off + n_children), n_parents);
        off += n_children;
        level++;
    }
    const unsigned remaining = log_leaves - level;
    if (remaining == 0) return false;
    TS_LAUNCH(ctx, k_merkle_tree, dim3(1u << (remaining - mt::block_log(remaining))), dim3(mt::NTH), 0, tree,
              log_leaves, level, ctx.ticket(), ch, root_out, beta_out);
    TS_HIP(hipGetLastError());
    return ch != nullptr;
}

}  // namespace ts
