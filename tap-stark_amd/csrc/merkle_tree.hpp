// A whole Merkle tree (up to 2^22 nodes in its first level) in ONE launch (device only).
//
// Workgroup g takes a block of B = 2^log_b consecutive nodes of the first level and reduces it to
// its sub-root; the workgroup that finishes LAST (ticket counter in device memory, sub-roots handed
// over with sc1 stores / loads behind one acquire in the finisher: see the hand-off note in tree_body) reduces the n_sub
// sub-roots to the root and, if asked, runs the device challenger step.  Every level is stored in
// the tree (levels back to back, as the gathers of the query phase expect).
//
// A block passes through LDS in chunks of CH = 512 nodes, image layout [word][node] (conflict-free
// for the coalesced staging and for the compressions).  Four lanes share one compression
// (blake3_quad.hpp), so a level keeps all 256 lanes busy down to 64 parents; a chunk is therefore
// reduced 512 -> 32 nodes only (four levels), the 32 are parked in a third image, and the parked
// nodes of all chunks (<= 256) are reduced together: the latency-bound narrow levels run once per
// block, not once per chunk.  Level barriers wait on LDS only, a level's digests go to the tree with
// plain stores that stay in flight.
//
// Where the first-level nodes come from is a template parameter (`Producer::fill`): digests already
// in the tree (leaf kernels wrote them), or computed on the spot -- the FRI commit phase folds the
// previous round's vector and hashes the pairs right here (fri.hip), which makes a commit round one
// launch instead of fold + levels + top.
#pragma once
#include "blake3_quad.hpp"
#include "chal_dev.hpp"

namespace ts {
namespace mt {

constexpr int NTH = 256;
constexpr uint32_t CH = 512;          // nodes per LDS chunk
constexpr uint32_t LOG_CH = 9;
constexpr uint32_t KEEP = 32;         // nodes a chunk is reduced to when its block has several chunks
constexpr uint32_t LOG_KEEP = 5;
constexpr unsigned MAX_LOG_BLOCK = 12;  // 8 chunks -> 256 parked nodes
constexpr unsigned MAX_LOG_SUB = 10;    // sub-roots the last workgroup reduces (2 chunks)
constexpr unsigned MAX_LOG_TREE = MAX_LOG_BLOCK + MAX_LOG_SUB;

struct Lds {
    uint32_t in[8 * CH];    // a chunk as staged / produced
    uint32_t ab[8 * CH];    // ping (nodes 0..255) and pong (nodes 256..383) of the levels
    uint32_t keep[8 * CH];  // parked chunk results
};

// host + device: log2 of the block a workgroup takes, for a tree of 2^remaining first-level nodes
TS_HD unsigned block_log(unsigned remaining) {
    if (remaining <= 8) return remaining;
    return remaining - 8 > MAX_LOG_SUB ? remaining - MAX_LOG_SUB : 8;
}

#if defined(__HIPCC__)

// where the tree keeps its levels, relative to the first level this launch works on
struct Levels {
    uint32_t* tree;    // base of the whole tree
    uint64_t off0;     // digests before the first level
    uint64_t n0;       // nodes in the first level (a power of two)
    __device__ __forceinline__ uint32_t* at(unsigned level, uint64_t node) const {
        return tree + 8 * (off0 + 2 * (n0 - (n0 >> level)) + node);
    }
};

// Reduces `count` nodes of image `src` (nodes 0..count-1, both powers of two, count <= CH) to `stop`
// nodes.  src holds nodes [node0, node0 + count) of relative level `level`; every level produced is
// stored in the tree.  Returns the image holding the result.  With publish, the single node of the
// last level (stop == 1) is written through for another workgroup to read.
__device__ __forceinline__ const uint32_t* reduce_levels(Lds& lds, const uint32_t* src, uint32_t count,
                                                         uint32_t stop, const Levels& lv, unsigned level,
                                                         uint64_t node0, const uint32_t moff[28],
                                                         bool publish) {
    const uint32_t j = threadIdx.x & 3;
    unsigned l = 0;
    for (uint32_t n_par = count >> 1; n_par >= stop; n_par >>= 1, l++) {
        uint32_t* dst = lds.ab + ((l & 1) ? 256 : 0);
        level++;
        node0 >>= 1;
        uint32_t* out = lv.at(level, node0);
        for (uint32_t t = threadIdx.x; t < 4 * n_par; t += NTH) {
            const uint32_t i = t >> 2;
            const uint32_t* base = src + 2 * i;
            uint32_t lo, hi;
            b3::compress_quad(j, b3::iv_word(j), b3::iv_word(4 + j),
                              [&](int k) { return base[moff[k]]; }, 64,
                              b3::CHUNK_START | b3::CHUNK_END | b3::ROOT, lo, hi);
            dst[j * CH + i] = lo;
            dst[(4 + j) * CH + i] = hi;
            uint32_t* o = out + 8 * i;
            if (publish && n_par == 1) {
                __hip_atomic_store(o + j, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 4 + j, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                o[j] = lo;
                o[4 + j] = hi;
            }
        }
        b3::lds_barrier();
        src = dst;
    }
    return src;
}

// One block: `count` (a power of two <= 2^MAX_LOG_BLOCK) nodes [node0, node0 + count) of relative
// level `level`, produced chunk by chunk by prod.fill(lds.in, first node, how many), down to one node.
// Returns the image whose node 0 is the block's root.
template <class Producer>
__device__ __forceinline__ const uint32_t* reduce_block(Lds& lds, Producer& prod, uint32_t count,
                                                        const Levels& lv, unsigned level, uint64_t node0,
                                                        const uint32_t moff[28], bool publish) {
    if (count <= CH) {
        prod.fill(lds.in, node0, count);
        if (count == 1) return lds.in;
        return reduce_levels(lds, lds.in, count, 1, lv, level, node0, moff, publish);
    }
    const uint32_t n_chunks = count >> LOG_CH;
    for (uint32_t c = 0; c < n_chunks; c++) {
        prod.fill(lds.in, node0 + (uint64_t)c * CH, CH);
        const uint32_t* x = reduce_levels(lds, lds.in, CH, KEEP, lv, level, node0 + (uint64_t)c * CH, moff, false);
        {
            const uint32_t w = threadIdx.x >> LOG_KEEP, n = threadIdx.x & (KEEP - 1);  // 8 x 32 = 256 lanes
            lds.keep[w * CH + c * KEEP + n] = x[w * CH + n];
        }
        b3::lds_barrier();
    }
    return reduce_levels(lds, lds.keep, n_chunks * KEEP, 1, lv, level + (LOG_CH - LOG_KEEP),
                         node0 >> (LOG_CH - LOG_KEEP), moff, publish);
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

__device__ __forceinline__ void quad_offsets(uint32_t moff[28]) {
    uint32_t idx[28];
    b3::quad_schedule(threadIdx.x & 3, idx);
#pragma unroll
    for (int k = 0; k < 28; k++) moff[k] = (idx[k] & 7) * CH + (idx[k] >> 3);
}

// The body of a tree kernel.  Grid = n_sub = 2^(remaining - log_b) workgroups of NTH lanes,
// remaining = log2(lv.n0) <= MAX_LOG_TREE, log_b = block_log(remaining).
template <class Producer>
__device__ __forceinline__ void tree_body(Lds& lds, uint32_t& s_last, Producer& prod, const Levels& lv,
                                          unsigned remaining, uint32_t* ticket, DevChallenger* ch,
                                          uint32_t* root_out, Ef* beta_out) {
    uint32_t moff[28];
    quad_offsets(moff);
    const unsigned log_b = block_log(remaining);
    const uint32_t B = 1u << log_b;
    const uint32_t n_sub = 1u << (remaining - log_b);
    const uint32_t* top = reduce_block(lds, prod, B, lv, 0, (uint64_t)blockIdx.x * B, moff, n_sub > 1);
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
        if (threadIdx.x == 0) {
            const uint32_t tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = tk == n_sub - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            StagedNodes<true> sub{lv.at(log_b, 0)};
            top = reduce_block(lds, sub, n_sub, lv, log_b, 0, moff, false);
            if (threadIdx.x == 0)  // ready for the next launch on this stream
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            finisher = true;
        }
    }
    // device-resident transcript (fri/src/prover.rs:114-116): the workgroup that produced the root
    // observes it and samples the next challenge
    if (finisher && ch != nullptr && threadIdx.x == 0) {
        uint32_t root[8];
        for (int k = 0; k < 8; k++) {
            root[k] = top[k * CH];
            root_out[k] = root[k];
        }
        // (the parked-node image is free by now: the sponge runs on a copy there, see chal_dev.hpp)
        DevChallenger* lc = reinterpret_cast<DevChallenger*>(lds.keep);
        dc_copy(lc, ch);
        const Ef beta = dc_observe_root_and_sample(lc, root);
        dc_copy(ch, lc);
        *reinterpret_cast<uint4*>(beta_out) = make_uint4(beta.c[0], beta.c[1], beta.c[2], beta.c[3]);
    }
}

#endif

}  // namespace mt
}  // namespace ts
