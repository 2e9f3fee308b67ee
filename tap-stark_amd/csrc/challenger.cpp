// Fiat-Shamir challenger on the host: strictly serial and consumes only 32-byte roots
// (SURVEY.md section 3.4).  Mirrors reference basic/src/challenger/mod.rs; see host.hpp for the
// per-method line references.
#include <string.h>

#include "blake3.hpp"
#include "host.hpp"

namespace ts {

void BfChallenger::duplexing() {
    // mod.rs:154-157: buffered inputs overwrite the rate part of the state
    for (int i = 0; i < n_in_; i++) state_[i] = in_[i];
    n_in_ = 0;
    if (permutation_ == Blake3Permutation) {
        // mod.rs:34-48: Blake3 over the 64 state bytes; state[0..8] = 0, state[8..16] = digest
        uint32_t digest[8];
        b3::hash64(state_, digest);
        for (int i = 0; i < 8; i++) {
            state_[i] = 0;
            state_[8 + i] = digest[i];
        }
    } else {
        // fri/tests/fri.rs:43-45
        for (int i = 0; i < 8; i++) {
            uint32_t t = state_[i];
            state_[i] = state_[15 - i];
            state_[15 - i] = t;
        }
    }
    // mod.rs:169-172
    n_out_ = 8;
    for (int i = 0; i < 8; i++) out_[i] = state_[8 + i];
}

void BfChallenger::observe(uint32_t word) {
    n_out_ = 0;  // any buffered output is now invalid
    in_[n_in_++] = word;
    if (n_in_ == 8) duplexing();
}

void BfChallenger::observe_commitment(const uint32_t d[8]) {
    for (int i = 0; i < 8; i++) observe(d[i]);
}

uint32_t BfChallenger::pop() {
    if (n_in_ != 0 || n_out_ == 0) duplexing();
    uint32_t v = out_[--n_out_];  // Vec::pop
    return v % P;                 // chan_field.rs:12-18
}

uint32_t BfChallenger::sample_base() { return pop(); }

Ef BfChallenger::sample_ext() {
    Ef r;
    for (int i = 0; i < 4; i++) r.c[i] = pop();
    return r;
}

Ef BfChallenger::sample() {
    if (sample_ext_) return sample_ext();
    return Ef{{pop(), 0, 0, 0}};
}

uint64_t BfChallenger::sample_bits(unsigned bits) {
    Ef s = sample();
    return bits == 0 ? 0 : ((uint64_t)s.c[0] >> (32 - bits));
}

bool BfChallenger::check_witness(unsigned bits, uint32_t witness) {
    observe(witness);
    for (int i = 0; i < 7; i++) observe(0);
    return sample_bits(bits) == 0;
}

uint32_t BfChallenger::grind(unsigned bits) {
    // chan_field.rs:35-42: mod_p() = 1 << (U8_NUM * 3) = 4096 candidates; serial find => smallest
    for (uint32_t w = 0; w < (1u << 12); w++) {
        BfChallenger clone = *this;
        if (clone.check_witness(bits, w)) {
            bool ok = check_witness(bits, w);
            (void)ok;
            return w;
        }
    }
    throw Error(TS_ERR_INVARIANT, "failed to find witness");
}

void BfChallenger::export_state(uint32_t out[34]) const {
    memcpy(out, state_, 64);
    out[16] = (uint32_t)n_in_;
    for (int i = 0; i < 8; i++) out[17 + i] = i < n_in_ ? in_[i] : 0;
    out[25] = (uint32_t)n_out_;
    for (int i = 0; i < 8; i++) out[26 + i] = i < n_out_ ? out_[i] : 0;
}

void BfChallenger::export_dev(DevChallenger& d) const {
    memset(&d, 0, sizeof d);
    memcpy(d.state, state_, 64);
    d.n_in = (uint32_t)n_in_;
    memcpy(d.in_buf, in_, 32);
    d.n_out = (uint32_t)n_out_;
    memcpy(d.out_buf, out_, 32);
    d.permutation = (uint32_t)permutation_;
    d.sample_ext = sample_ext_ ? 1u : 0u;
}

void BfChallenger::import_dev(const DevChallenger& d) {
    memcpy(state_, d.state, 64);
    n_in_ = (int)d.n_in;
    memcpy(in_, d.in_buf, 32);
    n_out_ = (int)d.n_out;
    memcpy(out_, d.out_buf, 32);
}

}  // namespace ts
