// Device-resident copy of the Fiat-Shamir challenger, used while the FRI commit phase runs without
// host round trips (root -> observe -> beta happens in a one-thread kernel or inside the FRI tail
// kernel).  Bit-for-bit the same sponge as the host BfChallenger (challenger.cpp), i.e. reference
// basic/src/challenger/mod.rs:151-174 (duplexing), :183-194 (observe), :261-313 (sample).
#pragma once
#include "blake3.hpp"
#include "blake3_quad.hpp"

namespace ts {

// layout shared with BfChallenger::export_state / import_state (34 words) + 2 config words
struct DevChallenger {
    uint32_t state[16];
    uint32_t n_in;
    uint32_t in_buf[8];
    uint32_t n_out;
    uint32_t out_buf[8];
    uint32_t permutation;  // 0 = Blake3Permutation, 1 = reverse (fri/tests/fri.rs:35-48)
    uint32_t sample_ext;   // 1: sample() yields EF4, 0: BabyBear
};

#if defined(__HIPCC__)
__device__ inline void dc_duplexing(DevChallenger* c) {
    for (uint32_t i = 0; i < c->n_in; i++) c->state[i] = c->in_buf[i];
    c->n_in = 0;
    if (c->permutation == 0) {
        uint32_t m[16], d[8];
        for (int i = 0; i < 16; i++) m[i] = c->state[i];
        b3::hash64(m, d);
        for (int i = 0; i < 8; i++) {
            c->state[i] = 0;
            c->state[8 + i] = d[i];
        }
    } else {
        for (int i = 0; i < 8; i++) {
            uint32_t t = c->state[i];
            c->state[i] = c->state[15 - i];
            c->state[15 - i] = t;
        }
    }
    c->n_out = 8;
    for (int i = 0; i < 8; i++) c->out_buf[i] = c->state[8 + i];
}
__device__ inline void dc_observe(DevChallenger* c, uint32_t word) {
    c->n_out = 0;
    c->in_buf[c->n_in++] = word;
    if (c->n_in == 8) dc_duplexing(c);
}
__device__ inline uint32_t dc_pop(DevChallenger* c) {
    if (c->n_in != 0 || c->n_out == 0) dc_duplexing(c);
    uint32_t v = c->out_buf[--c->n_out];
    return v % P;
}
// The sponge is a chain of dependent, dynamically indexed word accesses (in_buf[n_in++], out_buf[--n_out],
// state[i]); run straight on the copy in device memory every one of them is a global round trip.  The
// kernels therefore work on a copy in LDS (one lane) and write it back once.
constexpr int DC_WORDS = sizeof(DevChallenger) / 4;  // 36
__device__ inline void dc_copy(DevChallenger* dst, const DevChallenger* src) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    uint32_t t[DC_WORDS];
#pragma unroll
    for (int i = 0; i < DC_WORDS; i++) t[i] = s[i];  // independent loads, all in flight
#pragma unroll
    for (int i = 0; i < DC_WORDS; i++) d[i] = t[i];
}
// observe a commitment, then sample one challenge (fri/src/prover.rs:114-116)
__device__ inline Ef dc_observe_root_and_sample(DevChallenger* c, const uint32_t* root) {
    for (int i = 0; i < 8; i++) dc_observe(c, root[i]);
    Ef r = ef_zero();
    if (c->sample_ext) {
        for (int i = 0; i < 4; i++) r.c[i] = dc_pop(c);
    } else {
        r.c[0] = dc_pop(c);
    }
    return r;
}

// Orders one wave's LDS stores before its later LDS loads by other lanes (lock-step lanes of one wave
// need no s_barrier; the fences keep the compiler from moving the accesses across).
__device__ __forceinline__ void dc_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The same step on FOUR lanes, for the kernels whose last workgroup holds the root in an LDS digest
// image (word w of node n at img[w * STRIDE + n]; node 0 = the root).  With an empty input buffer and
// Blake3's permutation, observing the 8 words of a root fills the buffer exactly: the 8th observe
// duplexes, state = root || capacity, ONE hash64 -- the compression of a tree node whose children are
// (root, capacity half) --, state' = 0 || digest, and the samples are the digest's last words popped
// from the end.  So the capacity half is parked as node 1 of the image and the sponge runs as one more
// tree level on lanes 0..3 (blake3_quad.hpp: a quarter of the dependent chain), every index static.
// Measured in k_fri_tail: 3.5 us -> 0.8 us per round.  Anything else (pending input, the reverse test
// permutation) takes the one-lane form above.
//   lc: the challenger's working copy in LDS; node_iv: quad_iv(lane, 64, one-block flags); pm[k]: byte offsets into the image of this lane's 28 message
//   words for the node pair (0, 1) (quad_schedule; only lanes 0..3 use them); root_out / beta_out: global
//   (or page-locked host) destinations, may be null; beta_lds: where the workgroup reads beta afterwards.
// Call with the whole of wave 0 (threadIdx.x < 64 at least), uniformly; the caller's barrier follows.
template <uint32_t STRIDE>
__device__ __forceinline__ void dc_round_quad(DevChallenger* lc, uint32_t* img, const uint32_t pm[28],
                                              const b3::QuadIv& node_iv, uint32_t* root_out, Ef* beta_lds,
                                              Ef* beta_out) {
    const uint32_t tid = threadIdx.x;
    const bool fast = lc->n_in == 0 && lc->permutation == 0;
    if (!fast) {
        if (tid == 0) {
            uint32_t root[8];
            for (int k = 0; k < 8; k++) {
                root[k] = img[k * STRIDE];
                if (root_out != nullptr) root_out[k] = root[k];
            }
            const Ef beta = dc_observe_root_and_sample(lc, root);
            if (beta_lds != nullptr) *beta_lds = beta;
            if (beta_out != nullptr) *reinterpret_cast<uint4*>(beta_out) = make_uint4(beta.c[0], beta.c[1], beta.c[2], beta.c[3]);
        }
        dc_wave_sync();
        return;
    }
    if (tid < 8) {
        const uint32_t r = img[tid * STRIDE];
        if (root_out != nullptr) root_out[tid] = r;
        lc->in_buf[tid] = r;                       // what the eight observes leave behind
        img[tid * STRIDE + 1] = lc->state[8 + tid];  // the capacity half: node 1
        lc->state[tid] = 0;
    }
    // lanes 0..3 read what lanes 0..7 just stored: same wave, so no s_barrier, but the LDS stores must
    // be ordered before the loads for the compiler too (wave-scope release/acquire: no instruction
    // beyond the lgkmcnt wait the loads need anyway)
    dc_wave_sync();
    if (tid < 4) {
        uint32_t m[28];
#pragma unroll
        for (int k = 0; k < 28; k++)
            m[k] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(img) + pm[k]);
        uint32_t lo, hi;
        b3::compress_quad(node_iv, [&](int k) { return m[k]; }, lo, hi);
        lc->state[8 + tid] = lo;
        lc->state[12 + tid] = hi;
        lc->out_buf[tid] = lo;
        lc->out_buf[4 + tid] = hi;
        // pops take out_buf[7], [6], ...: challenge coefficient i is word 7 - i of the digest, lane 3 - i's `hi`
        const uint32_t n_pop = lc->sample_ext ? 4u : 1u;
        const uint32_t i = 3 - tid;
        const uint32_t c = i < n_pop ? hi % P : 0u;
        if (tid == 0) lc->n_out = 8 - n_pop;
        if (beta_lds != nullptr) beta_lds->c[i] = c;
        if (beta_out != nullptr) beta_out->c[i] = c;
    }
    dc_wave_sync();  // the caller's lanes read lc->state / out_buf next (copy-out of the challenger)
}
#endif

}  // namespace ts
