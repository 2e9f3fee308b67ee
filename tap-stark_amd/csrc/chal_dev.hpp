// Device-resident copy of the Fiat-Shamir challenger, used while the FRI commit phase runs without
// host round trips (root -> observe -> beta happens in a one-thread kernel or inside the FRI tail
// kernel).  Bit-for-bit the same sponge as the host BfChallenger (challenger.cpp), i.e. reference
// basic/src/challenger/mod.rs:151-174 (duplexing), :183-194 (observe), :261-313 (sample).
#pragma once
#include "blake3.hpp"

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
#endif

}  // namespace ts
