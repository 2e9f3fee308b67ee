/*
 * oracle/blake3_ref.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * BLAKE3 (default hash mode, 32-byte output), written from the published BLAKE3
 * specification (the `blake3 = "1.5"` crate the reference depends on is not on disk:
 * reference basic/Cargo.toml, call site basic/src/challenger/mod.rs:35-39).
 * Handles any input length (single chunk and the multi-chunk tree mode).
 *
 * Pinned by the reference's own known answers (tests/test_oracle_kats.py):
 *   reference scripts/src/hashes/blake3.rs:538  Blake3(16 x LE u32 1)
 *   reference scripts/src/hashes/blake3.rs:555  Blake3(15 x LE u32 1)
 */
#include "oracle.h"
#include <string.h>

static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                               0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t MSG_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};

enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void g(uint32_t* s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx;
    s[d] = rotr32(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];
    s[b] = rotr32(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my;
    s[d] = rotr32(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];
    s[b] = rotr32(s[b] ^ s[c], 7);
}

/* full 16-word compression output */
static void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter,
                     uint32_t block_len, uint32_t flags, uint32_t out[16]) {
    uint32_t s[16] = {cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7],
                      IV[0], IV[1], IV[2], IV[3], (uint32_t)counter, (uint32_t)(counter >> 32),
                      block_len, flags};
    uint32_t m[16];
    memcpy(m, block, sizeof m);
    for (int r = 0; r < 7; r++) {
        g(s, 0, 4, 8, 12, m[0], m[1]);
        g(s, 1, 5, 9, 13, m[2], m[3]);
        g(s, 2, 6, 10, 14, m[4], m[5]);
        g(s, 3, 7, 11, 15, m[6], m[7]);
        g(s, 0, 5, 10, 15, m[8], m[9]);
        g(s, 1, 6, 11, 12, m[10], m[11]);
        g(s, 2, 7, 8, 13, m[12], m[13]);
        g(s, 3, 4, 9, 14, m[14], m[15]);
        if (r < 6) {
            uint32_t t[16];
            for (int i = 0; i < 16; i++) t[i] = m[MSG_PERM[i]];
            memcpy(m, t, sizeof m);
        }
    }
    for (int i = 0; i < 8; i++) {
        out[i] = s[i] ^ s[i + 8];
        out[i + 8] = s[i + 8] ^ cv[i];
    }
}

static void load_block(const uint8_t* p, size_t len, uint32_t w[16]) {
    uint8_t buf[64];
    memset(buf, 0, 64);
    memcpy(buf, p, len);
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) |
               ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
}

/* An "output" that can be finalised either as a chaining value or as the root. */
typedef struct {
    uint32_t cv[8];
    uint32_t block[16];
    uint64_t counter;
    uint32_t block_len;
    uint32_t flags;
} b3_output;

static void output_cv(const b3_output* o, uint32_t cv[8]) {
    uint32_t out[16];
    compress(o->cv, o->block, o->counter, o->block_len, o->flags, out);
    memcpy(cv, out, 32);
}

/* process one chunk (<= 1024 bytes) into an output for its last block */
static b3_output chunk_output(const uint8_t* p, size_t len, uint64_t chunk_counter) {
    uint32_t cv[8];
    memcpy(cv, IV, 32);
    size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    b3_output o;
    for (size_t b = 0; b < nblocks; b++) {
        size_t off = b * 64;
        size_t bl = len - off < 64 ? len - off : 64;
        uint32_t w[16];
        load_block(p + off, bl, w);
        uint32_t flags = (b == 0 ? CHUNK_START : 0);
        if (b + 1 == nblocks) {
            memcpy(o.cv, cv, 32);
            memcpy(o.block, w, 64);
            o.counter = chunk_counter;
            o.block_len = (uint32_t)bl;
            o.flags = flags | CHUNK_END;
        } else {
            uint32_t out[16];
            compress(cv, w, chunk_counter, 64, flags, out);
            memcpy(cv, out, 32);
        }
    }
    return o;
}

static b3_output parent_output(const uint32_t l[8], const uint32_t r[8]) {
    b3_output o;
    memcpy(o.cv, IV, 32);
    memcpy(o.block, l, 32);
    memcpy(o.block + 8, r, 32);
    o.counter = 0;
    o.block_len = 64;
    o.flags = PARENT;
    return o;
}

/* recursive tree hash of `len` bytes starting at chunk index `chunk0`; len > 0 unless the
 * whole input is empty */
static b3_output subtree(const uint8_t* p, size_t len, uint64_t chunk0) {
    if (len <= 1024) return chunk_output(p, len, chunk0);
    /* left subtree = largest power-of-two number of chunks strictly less than total */
    size_t chunks = (len + 1023) / 1024;
    size_t left_chunks = 1;
    while (left_chunks * 2 < chunks) left_chunks *= 2;
    size_t left_len = left_chunks * 1024;
    b3_output lo = subtree(p, left_len, chunk0);
    b3_output ro = subtree(p + left_len, len - left_len, chunk0 + left_chunks);
    uint32_t lcv[8], rcv[8];
    output_cv(&lo, lcv);
    output_cv(&ro, rcv);
    return parent_output(lcv, rcv);
}

void ts_or_blake3(const uint8_t* in, size_t len, uint8_t out[32]) {
    b3_output o = subtree(in, len, 0);
    o.flags |= ROOT;
    uint32_t cv[8];
    output_cv(&o, cv); /* root output block 0 (counter = 0 for the root of a tree; chunk 0 for 1 chunk) */
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)cv[i];
        out[4 * i + 1] = (uint8_t)(cv[i] >> 8);
        out[4 * i + 2] = (uint8_t)(cv[i] >> 16);
        out[4 * i + 3] = (uint8_t)(cv[i] >> 24);
    }
}

/* hash a sequence of u32 words serialised little-endian (leaf/row hashing) */
void ts_or_blake3_words(const uint32_t* w, size_t n, uint32_t out[8]) {
    uint8_t digest[32];
    /* little-endian host assumed (x86-64): the word array IS the LE byte string */
    ts_or_blake3((const uint8_t*)w, 4 * n, digest);
    memcpy(out, digest, 32);
}
