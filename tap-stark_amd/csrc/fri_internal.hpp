// Pieces of bf_prove (fri/src/prover.rs:19-141) shared by the single-GPU prover (prover.cpp) and the
// sharded one (sharded.cpp).
#pragma once
#include "host.hpp"

namespace ts {

struct FriRound {
    const Ef* vec = nullptr;        // committed vector (rows of two): length 2 * 2^log_leaves
    const uint32_t* tree = nullptr;
    unsigned log_leaves = 0;
    uint32_t root[8];
};

// Device-side state of one commit phase.  The transcript lives on the device from begin to finish:
// per round the kernel that makes the root observes it and samples beta (d_betas[r]).
struct FriCommit {
    std::vector<FriRound> rounds;
    std::vector<DevBuf<Ef>> keep_vecs;
    std::vector<DevBuf<uint32_t>> keep_trees;
    // challenger | round roots | final values in ONE block: one D2H brings all three back at the end
    // (three copies were three launches on the stream); the pointers below look into it
    DevBuf<uint32_t> d_block;
    struct { uint32_t* p = nullptr; } d_chal, d_roots;
    struct { Ef* p = nullptr; } d_final;
    DevBuf<Ef> d_betas;
    uint32_t R_total = 0;
    uint64_t final_len = 0;
    // the tail kernel's proof-of-work hint (word FRI_POW_WORD of the challenger's 64-word slot): the
    // witness it found, or FRI_POW_NONE
    uint32_t pow_hint = 0xffffffffu;
    DevChallenger* dch() { return reinterpret_cast<DevChallenger*>(d_chal.p); }
};
constexpr size_t FRI_POW_WORD = 40;

// moves the transcript to the device and sizes the per-round buffers
void fri_commit_begin(Context& ctx, const FriConfig& fri, unsigned log_max_height,
                      const BfChallenger& challenger, FriCommit& st);
// prover.rs:111-127 on a vector every rank holds whole: rounds until `blowup` values are left,
// adding inputs[next_in..] when the folded length reaches theirs (:124-126)
void fri_commit_rounds(Context& ctx, const FriConfig& fri, DevBuf<Ef> folded, uint64_t len,
                       std::vector<DevBuf<Ef>>& inputs, const std::vector<unsigned>& log_lens,
                       size_t next_in, FriCommit& st);
// brings roots, final values and the transcript back; checks prover.rs:129-134; returns final_poly
Ef fri_commit_finish(Context& ctx, const FriConfig& fri, BfChallenger& challenger, FriCommit& st);
// prover.rs:43 challenger.grind(bits): takes the device's hint if one step of the host transcript
// confirms it, grinds on the host otherwise
uint32_t fri_pow_witness(Context& ctx, BfChallenger& challenger, unsigned bits, const FriCommit& st);

// small host helpers
void h2d(Context& ctx, void* dst, const void* src, size_t bytes);
void d2h_sync(Context& ctx, void* dst, const void* src, size_t bytes);
unsigned log2_strict(uint64_t n);
Ef efc_mul(Ef a, Ef b);
Ef efc_mul_base(Ef a, uint32_t b);
Ef efc_pow(Ef a, uint64_t e);
constexpr uint32_t TSPF_MAGIC = 0x46505354u;

}  // namespace ts
