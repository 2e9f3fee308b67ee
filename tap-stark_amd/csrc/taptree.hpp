// Taptree-compatible commitment: shared declarations of the
// kernels (taptree.hip) and the host side (taptree.cpp).  Reference: basic/src/tcs/{mod,builder,
// complete_taptree}.rs, basic/src/mmcs/taptree_mmcs.rs.
#pragma once
#include <stdint.h>

#include <memory>
#include <vector>

#include "bb.hpp"
#include "context.hpp"

namespace ts {

// SHA-256 states after the tag blocks of the BIP-341 tagged hashes
struct TapMid {
    uint32_t leaf[8];    // "TapLeaf"
    uint32_t branch[8];  // "TapBranch"
};

// bytes of the minimal script-number encoding of v > 0 (little-endian sign-magnitude)
TS_HD uint32_t tap_scriptnum_len(uint32_t v) {
    uint32_t n = 1;
    while (n < 5 && (v >> (8 * n - 1)) != 0) n++;
    return n;
}
// bytes rust-bitcoin's Builder::push_int(v) appends, 0 <= v < 2^32
TS_HD uint32_t tap_push_int_len(uint32_t v) { return v <= 16 ? 1u : 1u + tap_scriptnum_len(v); }

// What the device needs to assemble the leaf scripts of Q trees (tcs/mod.rs:197-225)
struct TapTemplate {
    const uint32_t* seg_words;         // every lock script as big-endian words, 4-byte aligned starts
    const uint64_t* seg_word_off;      // [Q][1 + n_evals] first word of a segment
    const uint32_t* seg_len;           // [Q][1 + n_evals] its length in bytes
    const uint64_t* const_len;         // [Q] sum of the segment lengths of a tree
    const uint32_t* const* cols;       // [n_evals * u32_size] column base pointers of the padded row
    const uint8_t* shift;              // [n_evals * u32_size] row = leaf index >> shift
    uint32_t n_evals, u32_size;
    uint32_t elem_stride;              // words between consecutive rows of a column (1: column-major;
                                       // 8: an array of EF4 pairs, the FRI commit-phase matrices)
    uint64_t tree_stride;              // digests per tree (2 N - 1)
    // optional (nullptr: none): the stream state after the leaf header and the first lock script,
    // [Q][n_len][TAP_PREFIX_WORDS], entry d = script length const_len + 2 (n_values + 1) + 1 + d
    const uint32_t* prefix;
    uint32_t n_len;
};
// h[8], the 16 words of the open block, words in it, partial word, its bytes
constexpr uint32_t TAP_PREFIX_WORDS = 32;

void launch_tapleaf_blob(Context& ctx, const uint32_t* words, const uint64_t* word_off,
                         const uint64_t* byte_len, uint64_t n_leaves, const TapMid& mid,
                         uint32_t* digests);
void launch_tapleaf_template(Context& ctx, const TapTemplate& t, uint64_t n_leaves, uint32_t n_trees,
                             const TapMid& mid, uint32_t* digests);
// fills t.prefix (n_trees x t.n_len entries) for a template whose other fields are set
void launch_tap_prefix(Context& ctx, const TapTemplate& t, uint32_t n_trees, const TapMid& mid, uint32_t* table);
// every upper level of n_trees trees stored `tree_stride` digests apart, levels back to back
void launch_tapbranch_levels(Context& ctx, uint32_t* trees, uint64_t tree_stride, unsigned log_leaves,
                             uint32_t n_trees, const TapMid& mid);
void launch_tap_gather_paths(Context& ctx, const uint32_t* trees, uint64_t tree_stride,
                             unsigned log_leaves, const uint32_t* tree_of, const uint64_t* index,
                             uint32_t n, uint32_t* out);

// host (taptree.cpp): Q taptrees over N = 2^log_height leaves whose padded row is described by
// `cols` / `shifts` / `elem_stride`; the lock scripts of tree q are scripts
// [first_lock + q (1 + n_evals), ...) of the table.  Returns the trees ([Q][2N-1][8] state words);
// roots_words receives Q x 8 words, each digest as its bytes read little-endian.
struct TapLocks;
DevBuf<uint32_t> tap_build_trees(Context& ctx, const std::vector<const uint32_t*>& cols,
                                 const std::vector<uint8_t>& shifts, uint32_t elem_stride,
                                 unsigned log_height, uint32_t u32_size, uint32_t num_queries,
                                 const TapLocks& locks, size_t first_lock, std::vector<uint32_t>& roots_words);
const TapMid& tap_mid();

}  // namespace ts
