// BLAKE3 compression for host code and HIP kernels (written from the published BLAKE3 spec).
//
// Uses: (1) the Fiat-Shamir permutation, reference basic/src/challenger/mod.rs:34-48
//           (one 64-byte block, CHUNK_START|CHUNK_END|ROOT);
//       (2) the build-defined Merkle MMCS: leaf = Blake3(row bytes),
//           node = Blake3(left || right).
// Rows up to one chunk (1024 bytes = 256 field elements) take the straight-line single-chunk path;
// wider rows go through `hash_stream` below: chunk chaining (counter = chunk index) and the BLAKE3
// parent tree (left subtree = the largest power of two of chunks), pinned by the official vectors
// (tests/golden/blake3_official.json).
#pragma once
#include <stdint.h>
#include "bb.hpp"

namespace ts {
namespace b3 {

constexpr uint32_t CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8;

#define TS_B3_IV0 0x6A09E667u
#define TS_B3_IV1 0xBB67AE85u
#define TS_B3_IV2 0x3C6EF372u
#define TS_B3_IV3 0xA54FF53Au
#define TS_B3_IV4 0x510E527Fu
#define TS_B3_IV5 0x9B05688Cu
#define TS_B3_IV6 0x1F83D9ABu
#define TS_B3_IV7 0x5BE0CD19u

TS_HD uint32_t rotr(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(x, x, n);
#else
    return (x >> n) | (x << (32 - n));
#endif
}

// rotr(x ^ y, 16): on the device two v_xor_b32_sdwa, each writing one 16-bit half of the result
// from the opposite halves of the operands, instead of v_xor + v_alignbit (measured on gfx950,
// tools/microbench4.hip: 56.4 against 51.5 G compressions/s; the compiler's v_add3_u32 for a + b + m
// beats two v_add_u32: 51.5 against 46.9)
TS_HD uint32_t xor_rotr16(uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0"
        : "=v"(r)
        : "v"(x), "v"(y));
    asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"
        : "+v"(r)
        : "v"(x), "v"(y));
    return r;
#else
    return rotr(x ^ y, 16);
#endif
}

#define TS_B3_G(a, b, c, d, mx, my) \
    a = a + b + (mx);               \
    d = xor_rotr16(d, a);           \
    c = c + d;                      \
    b = rotr(b ^ c, 12);            \
    a = a + b + (my);               \
    d = rotr(d ^ a, 8);             \
    c = c + d;                      \
    b = rotr(b ^ c, 7);

// One round with the message words addressed through a compile-time schedule.
#define TS_B3_ROUND(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15) \
    TS_B3_G(s0, s4, s8, s12, m0, m1)                                                      \
    TS_B3_G(s1, s5, s9, s13, m2, m3)                                                      \
    TS_B3_G(s2, s6, s10, s14, m4, m5)                                                     \
    TS_B3_G(s3, s7, s11, s15, m6, m7)                                                     \
    TS_B3_G(s0, s5, s10, s15, m8, m9)                                                     \
    TS_B3_G(s1, s6, s11, s12, m10, m11)                                                   \
    TS_B3_G(s2, s7, s8, s13, m12, m13)                                                    \
    TS_B3_G(s3, s4, s9, s14, m14, m15)

// cv (8 words, updated in place) <- compress(cv, m[16], counter = 0, block_len, flags).
// The 7 rounds use the spec's message permutation unrolled into fixed schedules, so `m` stays in
// registers and is never moved.
TS_HD void compress_ctr(uint32_t cv[8], const uint32_t m[16], uint32_t counter, uint32_t block_len,
                        uint32_t flags) {
    uint32_t s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6],
             s7 = cv[7];
    uint32_t s8 = TS_B3_IV0, s9 = TS_B3_IV1, s10 = TS_B3_IV2, s11 = TS_B3_IV3;
    uint32_t s12 = counter, s13 = 0, s14 = block_len, s15 = flags;
    TS_B3_ROUND(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15])
    TS_B3_ROUND(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8])
    TS_B3_ROUND(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1])
    TS_B3_ROUND(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6])
    TS_B3_ROUND(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4])
    TS_B3_ROUND(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7])
    TS_B3_ROUND(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13])
    cv[0] = s0 ^ s8;
    cv[1] = s1 ^ s9;
    cv[2] = s2 ^ s10;
    cv[3] = s3 ^ s11;
    cv[4] = s4 ^ s12;
    cv[5] = s5 ^ s13;
    cv[6] = s6 ^ s14;
    cv[7] = s7 ^ s15;
}

// counter = 0: every single-chunk input and every parent node
TS_HD void compress(uint32_t cv[8], const uint32_t m[16], uint32_t block_len, uint32_t flags) {
    compress_ctr(cv, m, 0u, block_len, flags);
}

TS_HD void iv(uint32_t cv[8]) {
    cv[0] = TS_B3_IV0; cv[1] = TS_B3_IV1; cv[2] = TS_B3_IV2; cv[3] = TS_B3_IV3;
    cv[4] = TS_B3_IV4; cv[5] = TS_B3_IV5; cv[6] = TS_B3_IV6; cv[7] = TS_B3_IV7;
}

// Blake3 of exactly 64 bytes (16 words): Merkle node = hash(left || right); sponge permutation.
TS_HD void hash64(const uint32_t m[16], uint32_t out[8]) {
    iv(out);
    compress(out, m, 64, CHUNK_START | CHUNK_END | ROOT);
}

// parent node of the BLAKE3 tree: compress(IV, left cv || right cv, PARENT [| ROOT])
TS_HD void parent_cv(const uint32_t l[8], const uint32_t r[8], bool root, uint32_t out[8]) {
    uint32_t m[16];
    for (int k = 0; k < 8; k++) {
        m[k] = l[k];
        m[8 + k] = r[k];
    }
    iv(out);
    compress(out, m, 64, PARENT | (root ? ROOT : 0u));
}

constexpr int MAX_TREE_DEPTH = 22;  // 2^22 chunks = 2^30 words: more than any row

// BLAKE3 (hash mode, 32-byte output) of `total_words` little-endian u32 words delivered by
// load(i), any length: the incremental algorithm of the BLAKE3 paper (section 5.1.2) -- a stack of
// subtree chaining values, merged whenever the number of finished chunks gains a factor of two.
template <class Load>
TS_HD void hash_stream(Load load, uint64_t total_words, uint32_t out[8]) {
    const uint64_t n_chunks = total_words == 0 ? 1 : (total_words + 255) / 256;
    uint32_t stack[MAX_TREE_DEPTH][8];
    int depth = 0;
    for (uint64_t chunk = 0; chunk < n_chunks; chunk++) {
        const uint64_t w0 = chunk * 256;
        const uint32_t cw = (uint32_t)(total_words - w0 < 256 ? total_words - w0 : 256);
        const uint32_t n_blocks = cw == 0 ? 1 : (cw + 15) / 16;
        const bool last_chunk = chunk + 1 == n_chunks;
        uint32_t cv[8];
        iv(cv);
        for (uint32_t blk = 0; blk < n_blocks; blk++) {
            uint32_t m[16];
            const uint32_t words = cw - blk * 16 < 16 ? cw - blk * 16 : 16;
            for (uint32_t j = 0; j < 16; j++) m[j] = j < words ? load(w0 + blk * 16 + j) : 0u;
            const uint32_t flags = (blk == 0 ? CHUNK_START : 0u) |
                                   (blk + 1 == n_blocks ? CHUNK_END : 0u) |
                                   (blk + 1 == n_blocks && n_chunks == 1 ? ROOT : 0u);
            compress_ctr(cv, m, (uint32_t)chunk, words * 4, flags);
        }
        if (!last_chunk) {
            // push, then merge completed subtrees: one merge per trailing zero of the chunk count
            uint64_t done = chunk + 1;
            while ((done & 1) == 0) {
                uint32_t p[8];
                parent_cv(stack[depth - 1], cv, false, p);
                for (int k = 0; k < 8; k++) cv[k] = p[k];
                depth--;
                done >>= 1;
            }
            for (int k = 0; k < 8; k++) stack[depth][k] = cv[k];
            depth++;
        } else {
            // the last chunk closes every open subtree, top of the stack first; the last merge is the root
            while (depth > 0) {
                uint32_t p[8];
                parent_cv(stack[depth - 1], cv, depth == 1, p);
                for (int k = 0; k < 8; k++) cv[k] = p[k];
                depth--;
            }
            for (int k = 0; k < 8; k++) out[k] = cv[k];
        }
    }
}

}  // namespace b3
}  // namespace ts
