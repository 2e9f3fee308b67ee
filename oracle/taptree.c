/*
 * oracle/taptree.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's taptree commitment:
 *   basic/src/tcs/builder.rs:24-93          TreeBuilder::add_leaf / build_tree
 *   basic/src/tcs/complete_taptree.rs:53-64 verify_inclusion, :90-133 combine
 *   basic/src/tcs/mod.rs:197-225            CommitedLeaf::generate_script
 *   basic/src/tcs/mod.rs:339-378            padding_matrix
 *   basic/src/tcs/mod.rs:238-292            commit_polys / commit_poly_with_query_times
 * The hashing itself lives in the un-vendored `bitcoin` fork (branch bitvm); what is restated is
 * its published algorithm, BIP-340/341: tagged_hash(tag, m) = SHA256(SHA256(tag) || SHA256(tag) || m),
 * TapLeaf = tagged("TapLeaf", version || compact_size(len) || script), TapBranch =
 * tagged("TapBranch", min(a, b) || max(a, b)).
 *
 * PARITY STATUS: SHA-256 is pinned by the NIST vectors, TapLeaf/TapBranch by the BIP-341 wallet
 * test vectors (tests/golden/kats.json, tests/test_oracle_taptree.py), the row -> leaf layout by the
 * known answer written in basic/src/tcs/mod.rs:594-602, the tree shape by the properties the
 * reference's own tests assert (complete_taptree.rs:163-369).  The lock-script BYTES are
 * unpinned: they come from un-vendored crates (bitcomm / primitives) and are an input here.
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

typedef struct {
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len;
} sha256_ctx;

static uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_block(sha256_ctx* c) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)c->buf[4 * i] << 24 | (uint32_t)c->buf[4 * i + 1] << 16 |
               (uint32_t)c->buf[4 * i + 2] << 8 | c->buf[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t v[8];
    memcpy(v, c->h, 32);
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ror32(v[4], 6) ^ ror32(v[4], 11) ^ ror32(v[4], 25);
        uint32_t ch = (v[4] & v[5]) ^ (~v[4] & v[6]);
        uint32_t t1 = v[7] + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ror32(v[0], 2) ^ ror32(v[0], 13) ^ ror32(v[0], 22);
        uint32_t mj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);
        uint32_t t2 = S0 + mj;
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1;
        v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) c->h[i] += v[i];
}

static void sha256_init(sha256_ctx* c) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                   0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(c->h, iv, 32);
    c->len = 0;
}
static void sha256_update(sha256_ctx* c, const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->buf[c->len++ & 63] = p[i];
        if ((c->len & 63) == 0) sha256_block(c);
    }
}
static void sha256_final(sha256_ctx* c, uint8_t out[32]) {
    uint64_t bits = c->len * 8;
    uint8_t b = 0x80;
    sha256_update(c, &b, 1);
    b = 0;
    while ((c->len & 63) != 56) sha256_update(c, &b, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha256_update(c, lb, 8);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(c->h[i] >> (24 - 8 * j));
}

void ts_or_sha256(const uint8_t* in, size_t len, uint8_t out[32]) {
    sha256_ctx c;
    sha256_init(&c);
    sha256_update(&c, in, len);
    sha256_final(&c, out);
}

/* BIP-340 tagged hash */
void ts_or_tagged_hash(const char* tag, const uint8_t* msg, size_t len, uint8_t out[32]) {
    uint8_t th[32];
    ts_or_sha256((const uint8_t*)tag, strlen(tag), th);
    sha256_ctx c;
    sha256_init(&c);
    sha256_update(&c, th, 32);
    sha256_update(&c, th, 32);
    sha256_update(&c, msg, len);
    sha256_final(&c, out);
}

/* NodeInfo::new_leaf_with_ver(script, ver).node_hash -- builder.rs:24-29 (ver = TapScript = 0xc0) */
void ts_or_tapleaf_hash(const uint8_t* script, size_t len, unsigned version, uint8_t out[32]) {
    uint8_t* m = (uint8_t*)malloc(len + 8);
    size_t k = 0;
    m[k++] = (uint8_t)version;
    if (len < 0xfd) {
        m[k++] = (uint8_t)len;
    } else if (len <= 0xffff) {
        m[k++] = 0xfd; m[k++] = (uint8_t)len; m[k++] = (uint8_t)(len >> 8);
    } else {
        m[k++] = 0xfe;
        for (int j = 0; j < 4; j++) m[k++] = (uint8_t)(len >> (8 * j));
    }
    memcpy(m + k, script, len);
    ts_or_tagged_hash("TapLeaf", m, k + len, out);
    free(m);
}

/* TapNodeHash::from_node_hashes(a, b) -- complete_taptree.rs:57-60: children in lexicographic order */
void ts_or_tapbranch(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    uint8_t m[64];
    if (memcmp(a, b, 32) <= 0) {
        memcpy(m, a, 32); memcpy(m + 32, b, 32);
    } else {
        memcpy(m, b, 32); memcpy(m + 32, a, 32);
    }
    ts_or_tagged_hash("TapBranch", m, 64, out);
}

/* ------------------------------------------------------------------ script pieces */
/* rust-bitcoin Builder::push_int(v), v >= 0: OP_0, OP_1..OP_16, or a minimal script-number push */
size_t ts_or_script_push_int(uint64_t v, uint8_t* out) {
    if (v == 0) { out[0] = 0x00; return 1; }
    if (v <= 16) { out[0] = (uint8_t)(0x50 + v); return 1; }
    uint8_t b[9];
    size_t n = 0;
    while (v) { b[n++] = (uint8_t)v; v >>= 8; }
    if (b[n - 1] & 0x80) b[n++] = 0; /* sign bit would read as negative */
    out[0] = (uint8_t)n;
    memcpy(out + 1, b, n);
    return 1 + n;
}

/* CommitedLeaf::generate_script -- tcs/mod.rs:197-225.
 * locks: 1 + n_evals byte strings (index lock first); values: n_evals * u32_size limbs
 * (as_u32_vec order, field/mod.rs:48-63); pushed limb u32_size-1 first (:214-217).
 * Returns the script length (writes at most `cap` bytes; call with cap = 0 to size). */
size_t ts_or_tap_leaf_script(const uint8_t* const* locks, const size_t* lock_lens, uint64_t index,
                             const uint32_t* values, uint32_t n_evals, uint32_t u32_size, uint8_t* out,
                             size_t cap) {
    size_t k = 0;
    uint8_t tmp[16];
#define PUT(p, n)                                        \
    do {                                                 \
        if (k + (n) <= cap) memcpy(out + k, (p), (n));   \
        k += (n);                                        \
    } while (0)
    PUT(locks[0], lock_lens[0]);
    size_t n = ts_or_script_push_int(index, tmp);
    tmp[n++] = 0x88; /* OP_EQUALVERIFY */
    PUT(tmp, n);
    for (uint32_t j = 0; j < n_evals; j++) {
        PUT(locks[1 + j], lock_lens[1 + j]);
        for (uint32_t l = u32_size; l-- > 0;) {
            n = ts_or_script_push_int(values[(size_t)j * u32_size + l], tmp);
            tmp[n++] = 0x88;
            PUT(tmp, n);
        }
    }
    tmp[0] = 0x51; /* OP_1 */
    PUT(tmp, 1);
#undef PUT
    return k;
}

/* ------------------------------------------------------------------ padding_matrix */
/* tcs/mod.rs:339-378.  Matrices sorted by height, tallest first (stable: sorted_by_key(Reverse));
 * a matrix of height h contributes row (leaf >> (log_max - log_h)) to every leaf, one element at a
 * time in column order.  out: max_height x total_width, row-major.  Returns total_width. */
size_t ts_or_padding_matrix(int n_mats, const uint32_t* const* mats, const size_t* heights,
                            const size_t* widths, uint32_t* out) {
    int* order = (int*)malloc(sizeof(int) * (size_t)n_mats);
    for (int i = 0; i < n_mats; i++) order[i] = i;
    for (int i = 1; i < n_mats; i++) { /* stable insertion sort, descending height */
        int v = order[i], j = i;
        while (j > 0 && heights[order[j - 1]] < heights[v]) { order[j] = order[j - 1]; j--; }
        order[j] = v;
    }
    size_t max_h = heights[order[0]], total = 0;
    for (int i = 0; i < n_mats; i++) total += widths[i];
    unsigned log_max = ts_log2_strict(max_h);
    size_t col = 0;
    for (int oi = 0; oi < n_mats; oi++) {
        int m = order[oi];
        unsigned lh = ts_log2_strict(heights[m]);
        for (size_t index = 0; index < heights[m]; index++) {
            size_t curr = index << (log_max - lh), next = (index + 1) << (log_max - lh);
            for (size_t i = 0; i < widths[m]; i++)
                for (size_t leaf = curr; leaf < next; leaf++)
                    out[leaf * total + col + i] = mats[m][index * widths[m] + i];
        }
        col += widths[m];
    }
    free(order);
    return total;
}

/* ------------------------------------------------------------------ build_tree */
/* builder.rs:38-93 on the leaf hashes: level by level, adjacent pairs (a power-of-two leaf count is
 * asserted at :40, so the `reminder_node` branch never fires).  Returns every level
 * (levels[0] = leaves) in one buffer of (2n - 1) x 32 bytes and the index dictionary: a NodeInfo
 * keeps its leaves in depth-first order, and combine_with_order(a, b) puts b's leaves first when
 * b's hash sorts before a's (!left_first, builder.rs:69-82; the fork's source is not vendored --
 * this reading is the one under which the assertions of complete_taptree.rs:163-209 hold).
 * leaf_indices[m] = position of merkle leaf m in that depth-first order (reverse_idx_dict, :95-101). */
void ts_or_taptree_build(size_t n, const uint8_t* leaf_hashes, uint8_t* nodes, size_t* leaf_indices) {
    memcpy(nodes, leaf_hashes, n * 32);
    size_t* t2m = (size_t*)malloc(sizeof(size_t) * n); /* t_idx_to_m_idx */
    size_t* tmp = (size_t*)malloc(sizeof(size_t) * n);
    for (size_t i = 0; i < n; i++) t2m[i] = i;
    uint8_t* cur = nodes;
    for (size_t cnt = n, span = 1; cnt > 1; cnt >>= 1, span <<= 1) {
        uint8_t* nxt = cur + cnt * 32;
        for (size_t i = 0; i < cnt / 2; i++) {
            const uint8_t *a = cur + 64 * i, *b = a + 32;
            ts_or_tapbranch(a, b, nxt + 32 * i);
            int left_first = memcmp(a, b, 32) <= 0;
            if (!left_first) { /* :69-82 swap the two index ranges */
                size_t a0 = 2 * i * span;
                memcpy(tmp, t2m + a0 + span, span * sizeof(size_t));
                memcpy(tmp + span, t2m + a0, span * sizeof(size_t));
                memcpy(t2m + a0, tmp, 2 * span * sizeof(size_t));
            }
        }
        cur = nxt;
    }
    for (size_t t = 0; t < n; t++) leaf_indices[t2m[t]] = t;
    free(t2m);
    free(tmp);
}

/* sibling path of merkle leaf `index` (leaf-most first) from the level buffer of ts_or_taptree_build */
void ts_or_taptree_path(size_t n, const uint8_t* nodes, size_t index, uint8_t* path) {
    const uint8_t* cur = nodes;
    size_t l = 0;
    for (size_t cnt = n; cnt > 1; cnt >>= 1, l++) {
        memcpy(path + 32 * l, cur + 32 * ((index >> l) ^ 1), 32);
        cur += cnt * 32;
    }
}

/* verify_inclusion -- complete_taptree.rs:53-64 */
int ts_or_taptree_verify_inclusion(const uint8_t root[32], const uint8_t leaf[32], const uint8_t* path,
                                   size_t depth) {
    uint8_t cur[32], nxt[32];
    memcpy(cur, leaf, 32);
    for (size_t l = 0; l < depth; l++) {
        ts_or_tapbranch(cur, path + 32 * l, nxt);
        memcpy(cur, nxt, 32);
    }
    return memcmp(cur, root, 32) == 0;
}

/* commit_polys -- tcs/mod.rs:238-282 for ONE tree: leaf_ys from padding_matrix, one script per
 * leaf, CompleteTaptree::new_with_scripts.  locks: 1 + n_evals lock scripts of this tree.
 * nodes: (2 max_h - 1) x 32 bytes out.  Returns 0, or -1 if the total width is not a multiple
 * of u32_size. */
int ts_or_tap_commit_polys(int n_mats, const uint32_t* const* mats, const size_t* heights,
                           const size_t* widths, uint32_t u32_size, const uint8_t* const* locks,
                           const size_t* lock_lens, uint8_t* nodes) {
    size_t max_h = 0, total = 0;
    for (int i = 0; i < n_mats; i++) {
        if (heights[i] > max_h) max_h = heights[i];
        total += widths[i];
    }
    if (total % u32_size) return -1;
    uint32_t n_evals = (uint32_t)(total / u32_size);
    uint32_t* ys = (uint32_t*)malloc(max_h * total * 4 + 4);
    ts_or_padding_matrix(n_mats, mats, heights, widths, ys);
    uint8_t* leaves = (uint8_t*)malloc(max_h * 32);
    for (size_t idx = 0; idx < max_h; idx++) {
        size_t len = ts_or_tap_leaf_script(locks, lock_lens, idx, ys + idx * total, n_evals, u32_size, NULL, 0);
        uint8_t* s = (uint8_t*)malloc(len + 1);
        ts_or_tap_leaf_script(locks, lock_lens, idx, ys + idx * total, n_evals, u32_size, s, len);
        ts_or_tapleaf_hash(s, len, 0xc0, leaves + 32 * idx);
        free(s);
    }
    size_t* li = (size_t*)malloc(sizeof(size_t) * max_h);
    ts_or_taptree_build(max_h, leaves, nodes, li);
    free(li);
    free(leaves);
    free(ys);
    return 0;
}
