// SHA-256 (FIPS 180-4) for host code and HIP kernels, and the BIP-340 tagged hash built on it.
//
// Used by the taptree-compatible MMCS: the reference's real BFMmcs
// commits to a Bitcoin taptree whose nodes are tagged SHA-256 hashes computed inside the external
// `bitcoin` crate (basic/src/tcs/builder.rs:24-29,64 NodeInfo::new_leaf_with_ver / combine_with_order;
// basic/src/tcs/complete_taptree.rs:51-64 TapNodeHash::from_node_hashes).  BIP-341:
//   TapLeaf   = tagged_hash("TapLeaf",   leaf_version || compact_size(len(script)) || script)
//   TapBranch = tagged_hash("TapBranch", min(a, b) || max(a, b))        (lexicographic order)
//   tagged_hash(tag, m) = SHA256(SHA256(tag) || SHA256(tag) || m)
// Digests are kept as the 8 big-endian state words H0..H7 (so lexicographic byte order = numeric
// order word by word); bytes are produced only at the ABI.
#pragma once
#include <stdint.h>
#include <string.h>

#include "bb.hpp"

namespace ts {
namespace sha {

#define TS_SHA_K_LIST                                                                                  \
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u,       \
        0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu,   \
        0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu,   \
        0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u,   \
        0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu,   \
        0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu,   \
        0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u,   \
        0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,   \
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u,   \
        0xc67178f2u

TS_HD uint32_t rotr(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(x, x, n);
#else
    return (x >> n) | (x << (32 - n));
#endif
}

// three-input bit functions: one v_bitop3_b32 each on gfx950 (left to the compiler, sigma / Sigma
// became two xors and Ch / Maj and-xor chains: ~1700 VALU instructions per compression instead of
// ~1400, in kernels that are bound by VALU issue)
TS_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
TS_HD uint32_t ch3(uint32_t e, uint32_t f, uint32_t g) {  // e ? f : g, bitwise
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(e, f, g, 0xca);
#else
    return (e & f) ^ (~e & g);
#endif
}
TS_HD uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8);
#else
    return (a & b) ^ (a & c) ^ (b & c);
#endif
}

// h (8 words, updated in place) <- compression of one 64-byte block given as 16 big-endian words.
// The message schedule runs in a 16-word ring so that the whole state stays in registers.
TS_HD void compress(uint32_t h[8], const uint32_t m[16]) {
#if defined(__HIP_DEVICE_COMPILE__)
    static const __device__ uint32_t K[64] = {TS_SHA_K_LIST};
#else
    static const uint32_t K[64] = {TS_SHA_K_LIST};
#endif
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = m[i];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
            const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        const uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
        const uint32_t ch = ch3(e, f, g);
        const uint32_t t1 = hh + S1 + ch + K[i] + w[i & 15];
        const uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
        const uint32_t maj = maj3(a, b, c);
        const uint32_t t2 = S0 + maj;
        hh = g; g = f; f = e; e = d + t1;
        d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

TS_HD void iv(uint32_t h[8]) {
    h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
    h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
}

// lexicographic comparison of two digests held as big-endian state words
TS_HD bool digest_less(const uint32_t a[8], const uint32_t b[8]) {
    for (int k = 0; k < 8; k++) {
        if (a[k] != b[k]) return a[k] < b[k];
    }
    return false;
}

// TapBranch (BIP-341): tagged hash of the two children in lexicographic order.  `mid` = the
// SHA-256 state after the 64-byte block SHA256("TapBranch") || SHA256("TapBranch").
TS_HD void tapbranch(const uint32_t mid[8], const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    const bool a_first = !digest_less(b, a);
    uint32_t m[16];
    for (int k = 0; k < 8; k++) {
        m[k] = a_first ? a[k] : b[k];
        m[8 + k] = a_first ? b[k] : a[k];
    }
    for (int k = 0; k < 8; k++) out[k] = mid[k];
    compress(out, m);
    // padding block: 0x80, zeros, bit length of 64 + 64 bytes
    uint32_t p[16] = {0x80000000u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1024u};
    compress(out, p);
}

// ------------------------------------------------------------------ host-side conveniences
struct Hasher {  // streaming SHA-256 over bytes
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len = 0;
    Hasher() { iv(h); }
    void update(const uint8_t* p, size_t n) {
        for (size_t i = 0; i < n; i++) {
            buf[len++ & 63] = p[i];
            if ((len & 63) == 0) block();
        }
    }
    void finish(uint32_t out[8]) {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while ((len & 63) != 56) update(&zero, 1);
        uint8_t lb[8];
        for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(lb, 8);
        for (int k = 0; k < 8; k++) out[k] = h[k];
    }

private:
    void block() {
        uint32_t m[16];
        for (int i = 0; i < 16; i++)
            m[i] = (uint32_t)buf[4 * i] << 24 | (uint32_t)buf[4 * i + 1] << 16 |
                   (uint32_t)buf[4 * i + 2] << 8 | buf[4 * i + 3];
        compress(h, m);
    }
};
inline void words_to_bytes(const uint32_t w[8], uint8_t out[32]) {
    for (int k = 0; k < 8; k++)
        for (int j = 0; j < 4; j++) out[4 * k + j] = (uint8_t)(w[k] >> (24 - 8 * j));
}
inline void bytes_to_words(const uint8_t in[32], uint32_t w[8]) {
    for (int k = 0; k < 8; k++)
        w[k] = (uint32_t)in[4 * k] << 24 | (uint32_t)in[4 * k + 1] << 16 | (uint32_t)in[4 * k + 2] << 8 |
               in[4 * k + 3];
}
// state after SHA256(tag) || SHA256(tag)
inline void tag_midstate(const char* tag, uint32_t mid[8]) {
    Hasher t;
    t.update(reinterpret_cast<const uint8_t*>(tag), strlen(tag));
    uint32_t th[8];
    t.finish(th);
    uint32_t m[16];
    for (int k = 0; k < 8; k++) m[k] = m[8 + k] = th[k];
    iv(mid);
    compress(mid, m);
}

}  // namespace sha
}  // namespace ts
