// A whole Merkle tree in ONE launch (device only).
//
// Workgroup g takes a block of B = 2^log_b consecutive nodes of the first level and reduces it to
// its sub-root; the workgroup that finishes LAST (ticket counter in device memory, sub-roots handed
// over with sc1 stores / loads behind one acquire in the finisher: see the hand-off note in
// finish_tree) reduces the n_sub sub-roots to the root and, if asked, runs the device challenger
// step.  Every level is stored in the tree (levels back to back, as the gathers of the query phase
// expect).
//
// Two block bodies share the finisher:
//
//  * tree_body (Tree<9>; k_merkle_tree, k_fri_round): up to 2^22 first-level nodes that lie in the
//    tree already or are produced chunk by chunk (`Producer::fill`).  A block passes through LDS in
//    chunks of CH = 512 nodes, image layout [word][node] (conflict-free for the coalesced staging
//    and for the compressions).  Four lanes share one compression (blake3_quad.hpp), so a level
//    keeps all 256 lanes busy down to 64 parents; a chunk is therefore reduced 512 -> 32 nodes only
//    (four levels), the 32 are parked in a third image, and the parked nodes of all chunks (<= 256)
//    are reduced together: the latency-bound narrow levels run once per block, not once per chunk.
//    Level barriers wait on LDS only, a level's digests go to the tree with plain stores that stay
//    in flight.
//
//  * leaf_tree_body (Tree<8>; k_leaf_tree in leaf_tree.hpp): LEAVES AND TREE in one launch, for trees
//    of 2^8 leaves and more (round 5; before it a tall tree was a leaf launch, one launch per level
//    down to 2^17 nodes -- every one re-reading from HBM what the previous one wrote -- and the
//    tree launch).  A lane hashes R = 2^LOG_R leaves (rows base + 256 k + lane, k < R: coalesced
//    columns, adjacent lanes = adjacent rows) and keeps the R digests in registers.  The first
//    LOG_R levels never leave the registers: in step s the two lanes that differ in bit s hold
//    sibling nodes for every k, so they swap half of their digests (DPP quad_perm for s = 0, 1,
//    ds_swizzle for s = 2: no LDS memory, no barrier) and each compresses half of the pairs --
//    every lane busy with whole compressions at the ALU rate of the leaf hashes themselves.  The
//    256 nodes a workgroup is left with go through LDS: 128 and 64 parents one compression per
//    lane, the narrow rest four lanes per compression.  Up to 256 sub-roots: hand-off and finisher
//    as above, one launch for everything; more: the launch ends with its sub-roots in the tree and
//    the whole-tree kernel takes that level over (a finisher would work through them chunk after
//    chunk, alone on the chip: measured 55 us for 2048 sub-roots against 21 for a second launch).
#pragma once
#include "blake3_quad.hpp"
#include "chal_dev.hpp"

namespace ts {
namespace mt {

constexpr int NTH = 256;
constexpr uint32_t KEEP = 32;  // nodes a chunk is reduced to when its block has several chunks
constexpr uint32_t LOG_KEEP = 5;

// host + device: the constants of the whole-tree kernels (Tree<9>)
constexpr uint32_t CH = 512;            // nodes per LDS chunk
constexpr uint32_t LOG_CH = 9;
constexpr unsigned MAX_LOG_BLOCK = 12;  // 8 chunks -> 256 parked nodes
constexpr unsigned MAX_LOG_SUB = 10;    // sub-roots the last workgroup reduces (2 chunks)
constexpr unsigned MAX_LOG_TREE = MAX_LOG_BLOCK + MAX_LOG_SUB;

// host + device: log2 of the block a workgroup takes, for a tree of 2^remaining first-level nodes
TS_HD unsigned block_log(unsigned remaining) {
    if (remaining <= 8) return remaining;
    return remaining - 8 > MAX_LOG_SUB ? remaining - MAX_LOG_SUB : 8;
}

// leaf_tree_body: the smallest tree it takes, the most sub-roots its in-kernel finisher takes, and the
// leaves per lane (log2) for a tree of 2^log_leaves leaves.  Measured on 2^22 leaves
// (tools/time_leaf_tree.py, us, rows of 2 / 64 elements): the finisher reduces its sub-roots chunk
// after chunk, alone on the chip -- 2048 of them cost 55 us against 21 us for a second launch of the
// whole-tree kernel -- so it takes one chunk at most; four leaves per lane (4096 workgroups) 221 /
// 505, eight 219 / 528 (issue fill 0.95 against 0.87: eight unrolled copies of the row hash), two 231 /
// 508, one 265 / 527; the round-4 path (leaf launch, five level launches, tree launch) 250 / 523.
constexpr unsigned LEAF_TREE_MIN_LOG = 8;
constexpr unsigned LEAF_TREE_MAX_LOG_SUB = 8;
TS_HD unsigned leaf_tree_log_r(unsigned log_leaves) {
    return log_leaves >= 20 ? 2 : log_leaves >= 18 ? log_leaves - 18 : 0;
}

#if defined(__HIPCC__)

// diagnostic build only (python -m tapstark_amd.build -DTS_TAIL_STAMPS, tools/tail_stamps.py): time stamps of the
// last whole-tree launch of a translation unit -- workgroup 0 for its block, the finisher for the rest
#ifdef TS_TAIL_STAMPS
__device__ static unsigned long long g_tree_stamps[64];
#define TS_TREE_STAMP(i, who) do { if (threadIdx.x == 0 && (who)) { g_tree_stamps[i] = __builtin_amdgcn_s_memrealtime(); g_tree_stamps[32 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TS_TREE_STAMP(i, who) do { } while (0)
#endif

// where the tree keeps its levels, relative to the first level this launch works on
struct Levels {
    uint32_t* tree;    // base of the whole tree
    uint64_t off0;     // digests before the first level
    uint64_t n0;       // nodes in the first level (a power of two)
    __device__ __forceinline__ uint32_t* at(unsigned level, uint64_t node) const {
        return tree + 8 * (off0 + 2 * (n0 - (n0 >> level)) + node);
    }
};

template <unsigned LC>
struct Tree {
    static constexpr uint32_t CH = 1u << LC;

    struct Lds {
        uint32_t in[8 * CH];    // a chunk as staged / produced
        uint32_t ab[8 * CH];    // ping (nodes 0..CH/2-1) and pong (nodes CH/2..3CH/4-1) of the levels
        uint32_t keep[8 * CH];  // parked chunk results
        DevChallenger chal;     // the transcript's working copy (prefetch_challenger, finish_tree)
    };

    // per-lane constants of the levels that run four lanes per compression
    struct Quad {
        uint32_t pm[28];  // byte offsets inside an image of the 28 message words of parent (lane >> 2)
        b3::QuadIv iv;    // initial state column of a node = hash64(left || right)
    };
    __device__ static __forceinline__ void quad_setup(Quad& q) {
        uint32_t idx[28];
        b3::quad_schedule(threadIdx.x & 3, idx);
#pragma unroll
        for (int k = 0; k < 28; k++) q.pm[k] = 4 * ((idx[k] & 7) * CH + (idx[k] >> 3) + 2 * (threadIdx.x >> 2));
        q.iv = b3::quad_iv(threadIdx.x & 3, 64, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);
    }

    // One level: n_par parents of the nodes in image `src` go to image `dst` and to the tree at `out`.
    // wide: one compression per LANE (b3::hash64 on two ds_read_b64 streams) instead of four lanes per
    // compression: 10.5 against 14.4 issued instructions per compression, at three times the chain
    // length -- right where other workgroups fill the SIMDs (leaf_tree_body's blocks), wrong for a
    // finisher that runs alone.  publish_one: the level's single node is written through for another
    // workgroup to read.
    // (Run-time image pointers: one copy of the code, one add per message word.  This is the level of the
    // wide levels, of the chunked blocks and of the leaf-tree kernel's finisher; the chains of the
    // whole-tree kernels and the narrow levels of a leaf-tree block take level_fixed below.)
    __device__ static __forceinline__ void level_step(const Quad& q, const uint32_t* src, uint32_t* dst, uint32_t n_par,
                                                      uint32_t* out, bool publish_one, bool wide) {
        if (wide) {
            for (uint32_t i = threadIdx.x; i < n_par; i += NTH) {
                uint32_t m[16], cv[8];
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    const uint2 v = *reinterpret_cast<const uint2*>(src + w * CH + 2 * i);
                    m[w] = v.x;
                    m[8 + w] = v.y;
                }
                b3::hash64(m, cv);
#pragma unroll
                for (int w = 0; w < 8; w++) dst[w * CH + i] = cv[w];
                uint4* o = reinterpret_cast<uint4*>(out + 8 * i);
                o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
            }
        } else {
            const uint32_t j = threadIdx.x & 3;
            const char* sb = reinterpret_cast<const char*>(src);
            // pass p takes parents (NTH / 4) p + (lane >> 2): 2 NTH bytes further on in every image row
            for (uint32_t t = threadIdx.x, off = 0; t < 4 * n_par; t += NTH, off += 2 * NTH) {
                const uint32_t i = t >> 2;
                uint32_t m[28];  // all 28 words requested before the first round needs one
#pragma unroll
                for (int k = 0; k < 28; k++) m[k] = *reinterpret_cast<const uint32_t*>(sb + q.pm[k] + off);
                uint32_t lo, hi;
                b3::compress_quad(q.iv, [&](int k) { return m[k]; }, lo, hi);
                dst[j * CH + i] = lo;
                dst[(4 + j) * CH + i] = hi;
                uint32_t* o = out + 8 * i;
                if (publish_one) {
                    __hip_atomic_store(o + j, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(o + 4 + j, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    o[j] = lo;
                    o[4 + j] = hi;
                }
            }
        }
        b3::lds_barrier();
    }

    // The same level for the whole-tree kernels' chains (Tree<9>), between the two images `in` (PAR = 0)
    // and `ab` (PAR = 1) taken in turn: image addresses are constants, so a lane's 28 message addresses
    // are two register sets it derives once, and a level that fits one pass (<= NTH / 4 parents: every
    // level of a block of 256 or fewer nodes but the first) is straight-line code -- no pass loop, no
    // address arithmetic, no register copies.  With ONE wave on a SIMD every instruction -- scalar ones,
    // waits and hazard nops included -- costs the wave a four-cycle issue slot, so a level's time is its
    // instruction count: ~350 with the run-time pointers above, ~270 here; the eight levels of a 256-node
    // block 8.5 -> 7.3 us (tools/tail_stamps.py).  (A first attempt -- three image pairs in / ping / pong,
    // the single pass peeled inside the pass loop -- measured 8.8: the compiler folded the peeled pass
    // back into the loop, address adds and all, and there were three copies of the code.)
    template <int PAR, bool SINGLE_PASS_ONLY = false>
    __device__ static __forceinline__ void level_fixed(Lds& lds, const Quad& q, uint32_t n_par, uint32_t* out,
                                                       bool publish_one) {
        const uint32_t j = threadIdx.x & 3;
        const char* sb = reinterpret_cast<const char*>(PAR ? lds.ab : lds.in);
        uint32_t* dst = PAR ? lds.in : lds.ab;
        auto finish = [&](uint32_t i, uint32_t lo, uint32_t hi) {
            dst[j * CH + i] = lo;
            dst[(4 + j) * CH + i] = hi;
            uint32_t* o = out + 8 * i;
            if (publish_one) {
                __hip_atomic_store(o + j, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 4 + j, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                o[j] = lo;
                o[4 + j] = hi;
            }
        };
        if (SINGLE_PASS_ONLY || 4 * n_par <= NTH) {
            if (threadIdx.x < 4 * n_par) {
                uint32_t m[28], lo, hi;
#pragma unroll
                for (int k = 0; k < 28; k++) m[k] = *reinterpret_cast<const uint32_t*>(sb + q.pm[k]);
                b3::compress_quad(q.iv, [&](int k) { return m[k]; }, lo, hi);
                finish(threadIdx.x >> 2, lo, hi);
            }
        } else {
            for (uint32_t t = threadIdx.x, off = 0; t < 4 * n_par; t += NTH, off += 2 * NTH) {
                uint32_t m[28], lo, hi;
#pragma unroll
                for (int k = 0; k < 28; k++) m[k] = *reinterpret_cast<const uint32_t*>(sb + q.pm[k] + off);
                b3::compress_quad(q.iv, [&](int k) { return m[k]; }, lo, hi);
                finish(t >> 2, lo, hi);
            }
        }
        b3::lds_barrier();
    }

    // Reduces `count` nodes of image `src` (nodes 0..count-1, both powers of two, count <= CH) to `stop`
    // nodes.  src holds nodes [node0, node0 + count) of relative level `level`; every level produced is
    // stored in the tree.  Returns the image holding the result.  With publish, the single node of the
    // last level (stop == 1) is written through for another workgroup to read.
    // wide_from: levels with at least this many parents run one compression per lane (0 = never).
    // FROM_IN (Tree<9> only): src is lds.in and the levels alternate between `in` and `ab` (level_fixed);
    // the producer's nodes are overwritten.
    template <bool FROM_IN = false>
    __device__ static __forceinline__ const uint32_t* reduce_levels(Lds& lds, const uint32_t* src, uint32_t count,
                                                                    uint32_t stop, const Levels& lv, unsigned level,
                                                                    uint64_t node0, const Quad& q, bool publish,
                                                                    uint32_t wide_from = 0) {
        if constexpr (FROM_IN && LC == 9) {
            uint32_t n_par = count >> 1;
            for (;;) {
                if (n_par < stop) return lds.in;
                level++;
                node0 >>= 1;
                level_fixed<0>(lds, q, n_par, lv.at(level, node0), publish && n_par == 1);
                n_par >>= 1;
                if (n_par < stop) return lds.ab;
                level++;
                node0 >>= 1;
                level_fixed<1>(lds, q, n_par, lv.at(level, node0), publish && n_par == 1);
                n_par >>= 1;
            }
        } else {
            unsigned l = 0;
            for (uint32_t n_par = count >> 1; n_par >= stop; n_par >>= 1, l++) {
                uint32_t* dst = lds.ab + ((l & 1) ? CH / 2 : 0);
                level++;
                node0 >>= 1;
                level_step(q, src, dst, n_par, lv.at(level, node0), publish && n_par == 1,
                           wide_from != 0 && n_par >= wide_from);
                src = dst;
            }
            return src;
        }
    }

    // One block: `count` (a power of two <= 8 * CH) nodes [node0, node0 + count) of relative
    // level `level`, produced chunk by chunk by prod.fill(lds.in, first node, how many), down to one node.
    // Returns the image whose node 0 is the block's root.
    template <class Producer>
    __device__ static __forceinline__ const uint32_t* reduce_block(Lds& lds, Producer& prod, uint32_t count,
                                                                   const Levels& lv, unsigned level, uint64_t node0,
                                                                   const Quad& q, bool publish) {
        if (count <= CH) {
            prod.fill(lds.in, node0, count);
            TS_TREE_STAMP(1, blockIdx.x == 0 && level == 0);
            if (count == 1) return lds.in;
            return reduce_levels<true>(lds, lds.in, count, 1, lv, level, node0, q, publish);
        }
        const uint32_t n_chunks = count >> LC;
        for (uint32_t c = 0; c < n_chunks; c++) {
            prod.fill(lds.in, node0 + (uint64_t)c * CH, CH);
            const uint32_t* x = reduce_levels<true>(lds, lds.in, CH, KEEP, lv, level, node0 + (uint64_t)c * CH, q, false);
            {
                const uint32_t w = threadIdx.x >> LOG_KEEP, n = threadIdx.x & (KEEP - 1);  // 8 x 32 = 256 lanes
                lds.keep[w * CH + c * KEEP + n] = x[w * CH + n];
            }
            b3::lds_barrier();
        }
        return reduce_levels(lds, lds.keep, n_chunks * KEEP, 1, lv, level + (LC - LOG_KEEP), node0 >> (LC - LOG_KEEP), q,
                             publish);
    }

    // first-level nodes that already lie in the tree; SC1: they were written by other workgroups of
    // this launch (loads must bypass this CU's L1)
    template <bool SC1>
    struct StagedNodes {
        const uint32_t* nodes;  // the level's node 0
        __device__ __forceinline__ void fill(uint32_t* in, uint64_t node0, uint32_t count) {
            const uint32_t* p = nodes + 8 * node0;
            if (SC1) {
                for (uint32_t e = threadIdx.x; e < 8 * count; e += NTH)
                    in[(e & 7) * CH + (e >> 3)] = __hip_atomic_load(p + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const uint4* p4 = reinterpret_cast<const uint4*>(p);
                for (uint32_t e = threadIdx.x; e < 2 * count; e += NTH) {  // 16 bytes = half a node per lane
                    const uint4 v = p4[e];
                    uint32_t* o = in + (e & 1) * 4 * CH + (e >> 1);
                    o[0] = v.x;
                    o[CH] = v.y;
                    o[2 * CH] = v.z;
                    o[3 * CH] = v.w;
                }
            }
            __syncthreads();
        }
    };

    // Every workgroup fetches the transcript's 36 words at its start (one coalesced load that nobody
    // waits for): the workgroup that turns out to be the finisher has them in LDS when it gets there
    // instead of paying a global round trip at the very end of the dependency chain.  The previous
    // launch on the stream wrote them; nothing in this launch does before the finisher.
    __device__ static __forceinline__ void prefetch_challenger(Lds& lds, const DevChallenger* ch) {
        if (ch != nullptr && threadIdx.x < DC_WORDS)
            reinterpret_cast<uint32_t*>(&lds.chal)[threadIdx.x] = reinterpret_cast<const uint32_t*>(ch)[threadIdx.x];
    }

    // What every workgroup does once its block is down to its sub-root `top` (node blockIdx.x of
    // relative level log_b; n_sub = gridDim.x of them): hand-off, the last one reduces the sub-roots
    // to the root, then the device challenger step.  (prefetch_challenger ran at the workgroup's start.)
    __device__ static __forceinline__ void finish_tree(Lds& lds, uint32_t& s_last, const uint32_t* top,
                                                       const Levels& lv, unsigned log_b, uint32_t n_sub,
                                                       const Quad& q, uint32_t* ticket,
                                                       DevChallenger* ch, uint32_t* root_out, Ef* beta_out) {
        bool finisher = n_sub == 1;
        if (n_sub > 1) {
            // Hand-off between workgroups.  Per-XCD L2s are not coherent with each other and a CU's L1
            // is never refreshed by another CU's stores, so (MI355X_MICROARCH.md, "Valid forms" and its
            // table of measured hand-offs): every handed-off byte -- the 32-byte sub-root -- is stored
            // sc1 (written through, dropped from the XCD's L2) and loaded sc1 (bypassing L1); the
            // storing wave drains its stores, the workgroup meets, ONE lane adds to the ticket counter
            // at agent scope; the workgroup whose add came last loads after a barrier behind that add.
            // No RELEASE fence on the producing side (sc1 stores are written through; 2-6 us saved per
            // workgroup).  The consuming side keeps ONE agent-scope acquire in the finishing workgroup:
            // the guide's fence-free consumer form is measured at one workgroup per CU only, and this
            // kernel runs two or three per CU with several proofs in flight.  It costs the
            // finisher ~2.4 us per tree (30.1 against 27.7 us for a 2^16-leaf top) -- once per launch, not
            // per workgroup -- and the sub-root loads stay sc1 on top of it.  tests/test_build_isa.py
            // checks the disassembly: sc1 on these stores and loads, buffer_inv sc1, no flat_ access.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            TS_TREE_STAMP(3, blockIdx.x == 0);
            if (threadIdx.x == 0) {
                const uint32_t tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = tk == n_sub - 1 ? 1u : 0u;
            }
            __syncthreads();
            TS_TREE_STAMP(4, blockIdx.x == 0);
            if (s_last) {
                TS_TREE_STAMP(5, true);
                if (threadIdx.x == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                TS_TREE_STAMP(6, true);
                StagedNodes<true> sub{lv.at(log_b, 0)};
                top = reduce_block(lds, sub, n_sub, lv, log_b, 0, q, false);
                TS_TREE_STAMP(7, true);
                if (threadIdx.x == 0)  // ready for the next launch on this stream
                    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                finisher = true;
            }
        }
        // device-resident transcript (fri/src/prover.rs:114-116): the workgroup that produced the root
        // observes it and samples the next challenge -- as one more tree level on four lanes
        // (dc_round_quad, chal_dev.hpp).  root_out may be page-locked host memory: the commitment goes
        // straight to the caller's mailbox.
        if (finisher && threadIdx.x < 64) {
            TS_TREE_STAMP(8, true);
            uint32_t* img = const_cast<uint32_t*>(top);  // node 1 of the final image is free
            if (ch != nullptr) {
                dc_round_quad<CH>(&lds.chal, img, q.pm, q.iv, root_out, nullptr, beta_out);
                if (threadIdx.x < DC_WORDS)
                    reinterpret_cast<uint32_t*>(ch)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&lds.chal)[threadIdx.x];
            } else if (root_out != nullptr && threadIdx.x < 8) {
                root_out[threadIdx.x] = img[threadIdx.x * CH];
            }
            TS_TREE_STAMP(9, true);
        }
    }

    // The body of a whole-tree kernel over staged / chunk-produced first-level nodes.  Grid = n_sub =
    // 2^(remaining - log_b) workgroups of NTH lanes, remaining = log2(lv.n0) <= MAX_LOG_TREE, log_b =
    // block_log(remaining).
    template <class Producer>
    __device__ static __forceinline__ void tree_body(Lds& lds, uint32_t& s_last, Producer& prod, const Levels& lv,
                                                     unsigned remaining, uint32_t* ticket, DevChallenger* ch,
                                                     uint32_t* root_out, Ef* beta_out) {
        TS_TREE_STAMP(0, blockIdx.x == 0);
        prefetch_challenger(lds, ch);
        Quad q;
        quad_setup(q);
        const unsigned log_b = block_log(remaining);
        const uint32_t B = 1u << log_b;
        const uint32_t n_sub = 1u << (remaining - log_b);
        const uint32_t* top = reduce_block(lds, prod, B, lv, 0, (uint64_t)blockIdx.x * B, q, n_sub > 1);
        TS_TREE_STAMP(2, blockIdx.x == 0);
        finish_tree(lds, s_last, top, lv, log_b, n_sub, q, ticket, ch, root_out, beta_out);
    }
};

using T9 = Tree<9>;
using Lds = T9::Lds;

// ---- leaves and tree in one launch -------------------------------------------------------------
// the lane whose id differs in bit S (S = 0, 1: DPP quad_perm, a modifier-class move; S = 2: ds_swizzle
// in bit-mask mode, and = 0x1f, xor = 4 -- the LDS crossbar, no LDS memory)
template <int S>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    static_assert(S >= 0 && S <= 2, "lane_xor: in-register levels reach three lane bits");
    if constexpr (S == 0) return b3::quad_perm<0xB1>(v);
    else if constexpr (S == 1) return b3::quad_perm<0x4E>(v);
    else return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);
}

// One in-register level.  Before: D[0..C) with D[m] the digest of node (base + 256 k + lane) >> S of
// relative level S, k = (m << S) + (lane & (2^S - 1)).  After: D[0..C/2), the same with S + 1.  The
// partner lane (bit S flipped) holds the sibling of every one of this lane's nodes; the lane with
// bit S clear takes the even m of each pair of digests, the other the odd m.
template <int S, int C>
__device__ __forceinline__ void register_level(uint32_t (*D)[8], const Levels& lv, uint64_t base) {
    const uint32_t tid = threadIdx.x;
    const bool hi = (tid >> S) & 1u;
#pragma unroll
    for (int m = 0; m < C / 2; m++) {
        uint32_t msg[16], cv[8];
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const uint32_t send = hi ? D[2 * m][w] : D[2 * m + 1][w];
            const uint32_t keep = hi ? D[2 * m + 1][w] : D[2 * m][w];
            const uint32_t recv = lane_xor<S>(send);
            msg[w] = hi ? recv : keep;      // left child: the lane with bit S clear
            msg[8 + w] = hi ? keep : recv;  // right child
        }
        b3::hash64(msg, cv);
#pragma unroll
        for (int w = 0; w < 8; w++) D[m][w] = cv[w];
        const uint32_t k = ((uint32_t)(2 * m + (hi ? 1 : 0)) << S) + (tid & ((1u << S) - 1));
        uint4* o = reinterpret_cast<uint4*>(lv.at(S + 1, (base + 256ull * k + tid) >> (S + 1)));
        o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
    }
}

using T8 = Tree<8>;

// D[k] = digest of leaf row0 + 256 k, k = K .. R-1, stored as level 0.  (A recursion, not a loop: the
// compiler left `#pragma unroll` over a leaf hash that loops itself partly rolled and put D in scratch.)
template <int K, int R, class Leaf>
__device__ __forceinline__ void leaf_rows(Leaf& leaf, uint32_t (*D)[8], const Levels& lv, uint64_t row0) {
    if constexpr (K < R) {
        const uint64_t row = row0 + 256u * K;
        leaf.digest(row, D[K]);
        uint4* o = reinterpret_cast<uint4*>(lv.at(0, row));
        o[0] = make_uint4(D[K][0], D[K][1], D[K][2], D[K][3]);
        o[1] = make_uint4(D[K][4], D[K][5], D[K][6], D[K][7]);
        leaf_rows<K + 1, R>(leaf, D, lv, row0);
    }
}

// Grid = 2^(log_leaves - 8 - LOG_R) workgroups of NTH lanes; leaf.digest(row, cv) hashes leaf `row`
// (any side effect -- the FRI fold's store -- included).  All lanes of every workgroup are active
// (the lane exchanges need them).  finish = false (n_sub > 2^LEAF_TREE_MAX_LOG_SUB): the launch ends
// with the sub-roots in the tree and a whole-tree kernel takes the level over.
template <int LOG_R, class Leaf>
__device__ __forceinline__ void leaf_tree_body(T8::Lds& lds, uint32_t& s_last, Leaf& leaf, const Levels& lv,
                                               unsigned log_leaves, bool finish, uint32_t* ticket,
                                               DevChallenger* ch, uint32_t* root_out, Ef* beta_out) {
    constexpr int R = 1 << LOG_R;
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * (256u * R);
    if (finish) T8::prefetch_challenger(lds, ch);
    uint32_t D[R][8];
    leaf_rows<0, R>(leaf, D, lv, base + tid);
    if constexpr (LOG_R >= 1) register_level<0, R>(D, lv, base);
    if constexpr (LOG_R >= 2) register_level<1, R / 2>(D, lv, base);
    if constexpr (LOG_R >= 3) register_level<2, R / 4>(D, lv, base);
    // D[0]: node (base + 256 k + tid) >> LOG_R of level LOG_R, k = tid & (R - 1): slot among the
    // workgroup's 256 nodes of that level
    {
        const uint32_t slot = (256u >> LOG_R) * (tid & (R - 1)) + (tid >> LOG_R);
#pragma unroll
        for (int w = 0; w < 8; w++) lds.in[w * T8::CH + slot] = D[0][w];
    }
    b3::lds_barrier();
    T8::Quad q;
    T8::quad_setup(q);
    const unsigned log_b = 8 + LOG_R;
    const uint32_t n_sub = 1u << (log_leaves - log_b);
    // the block's eight levels between the images `in` and `ab` in turn: 128 and 64 parents one compression
    // per lane, then 32 .. 1 four lanes per compression as straight-line levels (level_fixed: while a
    // workgroup is in these narrow levels most of its waves wait at barriers, so their length is what the
    // launch's issue fill pays: 0.68 for the 16-column chunk tree against 0.95 for the 64-column trace tree)
    const bool pub = finish && n_sub > 1;
    uint64_t node = base >> LOG_R;
    unsigned level = LOG_R;
    auto out = [&]() {
        level++;
        node >>= 1;
        return lv.at(level, node);
    };
    T8::level_step(q, lds.in, lds.ab, 128, out(), false, true);
    T8::level_step(q, lds.ab, lds.in, 64, out(), false, true);
    for (uint32_t n_par = 32;;) {
        T8::template level_fixed<0, true>(lds, q, n_par, out(), false);
        n_par >>= 1;
        T8::template level_fixed<1, true>(lds, q, n_par, out(), pub && n_par == 1);
        n_par >>= 1;
        if (n_par == 0) break;
    }
    const uint32_t* top = lds.in;
    if (finish) T8::finish_tree(lds, s_last, top, lv, log_b, n_sub, q, ticket, ch, root_out, beta_out);
}

#endif

}  // namespace mt
}  // namespace ts
