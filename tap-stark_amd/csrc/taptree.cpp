// Taptree-compatible MMCS, host side.  Reference:
//   basic/src/tcs/builder.rs:24-93            TreeBuilder::add_leaf / build_tree
//   basic/src/tcs/complete_taptree.rs:5-161   CompleteTaptree: new_with_scripts, combine,
//                                             get_leaf_proof, verify_inclusion
//   basic/src/tcs/mod.rs:197-225              CommitedLeaf::generate_script (the leaf script)
//   basic/src/tcs/mod.rs:238-292,339-378      commit_polys, commit_poly_with_query_times, padding_matrix
//   basic/src/mmcs/taptree_mmcs.rs:46-114     TapTreeMmcs: commit / open_batch / verify_batch
//
// What is and is not the reference's.  The tree (tagged SHA-256 TapLeaf / TapBranch with sorted
// children, complete binary shape, one tree per query), the leaf layout (padding_matrix) and the
// leaf-script skeleton (lock script, pushed value, OP_EQUALVERIFY ... OP_1) are the reference's.
// The bytes of a lock script are NOT available: `locking_script_with_type(CompressType::U32)` lives
// in the un-vendored crates bitcomm / primitives (basic/Cargo.toml:7-11), and so does the secret
// generator.  The caller therefore supplies the lock scripts as bytes (one for the index, one per
// evaluation, per tree); `winternitz_lock_script` below is a stand-in written from the LOCAL copy of
// the same construction (scripts/src/bit_comm/winternitz.rs:58-297 checksig_verify,
// scripts/src/bit_comm/bit_comm_u32.rs:80-85 recover_message_at_stack, scripts/src/u32/u32_std.rs:
// 122-173 u32_compress), which lets tests and benchmarks run with scripts of realistic size.
// Executing the leaf script against a Winternitz witness (tcs/mod.rs:143-147 verify_proof) needs
// a Bitcoin script interpreter and is out of scope: verify_batch checks that the leaf rebuilt from
// the opened values and the lock scripts is in the tree.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "abi_types.hpp"
#include "fri_internal.hpp"
#include "sha256.hpp"
#include "taptree.hpp"

namespace ts {

// ------------------------------------------------------------------ hashes (host)
const TapMid& tap_mid() {
    static const TapMid m = [] {
        TapMid t;
        sha::tag_midstate("TapLeaf", t.leaf);
        sha::tag_midstate("TapBranch", t.branch);
        return t;
    }();
    return m;
}

static void put_compact_size(std::vector<uint8_t>& b, uint64_t n) {
    if (n < 0xfd) {
        b.push_back((uint8_t)n);
    } else if (n <= 0xffff) {
        b.push_back(0xfd);
        b.push_back((uint8_t)n);
        b.push_back((uint8_t)(n >> 8));
    } else {
        b.push_back(0xfe);
        for (int j = 0; j < 4; j++) b.push_back((uint8_t)(n >> (8 * j)));
    }
}

// NodeInfo::new_leaf_with_ver(script, TapScript).node_hash (builder.rs:24-29)
void tapleaf_hash(const uint8_t* script, size_t len, uint32_t out[8]) {
    TS_REQUIRE(len <= 0xffffffffull, TS_ERR_INVALID, "tapleaf: script too long");
    sha::Hasher h;
    memcpy(h.h, tap_mid().leaf, 32);
    h.len = 64;
    std::vector<uint8_t> hdr{0xc0};
    put_compact_size(hdr, len);
    h.update(hdr.data(), hdr.size());
    h.update(script, len);
    h.finish(out);
}

// ------------------------------------------------------------------ script assembly (host)
void script_push_int(std::vector<uint8_t>& s, uint32_t v) {
    if (v == 0) {
        s.push_back(0x00);
    } else if (v <= 16) {
        s.push_back((uint8_t)(0x50 + v));
    } else {
        const uint32_t n = tap_scriptnum_len(v);
        s.push_back((uint8_t)n);
        for (uint32_t j = 0; j < n; j++) s.push_back(j < 4 ? (uint8_t)(v >> (8 * j)) : 0);
    }
}

// tcs/mod.rs:197-225: locks[0] = index lock, locks[1 + j] = lock of evaluation j;
// values = n_evals * u32_size canonical limbs (as_u32_vec order)
std::vector<uint8_t> tap_leaf_script(const std::vector<std::pair<const uint8_t*, size_t>>& locks,
                                     uint64_t index, const uint32_t* values, uint32_t n_evals,
                                     uint32_t u32_size) {
    TS_REQUIRE(locks.size() == 1 + (size_t)n_evals, TS_ERR_INVALID, "leaf script: one lock per value + index");
    std::vector<uint8_t> s;
    s.insert(s.end(), locks[0].first, locks[0].first + locks[0].second);
    script_push_int(s, (uint32_t)index);
    s.push_back(0x88);
    for (uint32_t j = 0; j < n_evals; j++) {
        s.insert(s.end(), locks[1 + j].first, locks[1 + j].first + locks[1 + j].second);
        for (uint32_t l = u32_size; l-- > 0;) {
            script_push_int(s, values[(size_t)j * u32_size + l]);
            s.push_back(0x88);
        }
    }
    s.push_back(0x51);
    return s;
}

// ------------------------------------------------------------------ RIPEMD-160 / hash160 (host)
namespace {

uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

void ripemd160(const uint8_t* msg, size_t len, uint8_t out[20]) {
    static const uint8_t RL[80] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 7, 4, 13, 1, 10, 6, 15, 3, 12, 0, 9, 5,
                                   2, 14, 11, 8, 3, 10, 14, 4, 9, 15, 8, 1, 2, 7, 0, 6, 13, 11, 5, 12, 1, 9, 11, 10, 0, 8, 12, 4,
                                   13, 3, 7, 15, 14, 5, 6, 2, 4, 0, 5, 9, 7, 12, 2, 10, 14, 1, 3, 8, 11, 6, 15, 13};
    static const uint8_t RR[80] = {5, 14, 7, 0, 9, 2, 11, 4, 13, 6, 15, 8, 1, 10, 3, 12, 6, 11, 3, 7, 0, 13, 5, 10, 14, 15, 8, 12,
                                   4, 9, 1, 2, 15, 5, 1, 3, 7, 14, 6, 9, 11, 8, 12, 2, 10, 0, 4, 13, 8, 6, 4, 1, 3, 11, 15, 0,
                                   5, 12, 2, 13, 9, 7, 10, 14, 12, 15, 10, 4, 1, 5, 8, 7, 6, 2, 13, 14, 0, 3, 9, 11};
    static const uint8_t SL[80] = {11, 14, 15, 12, 5, 8, 7, 9, 11, 13, 14, 15, 6, 7, 9, 8, 7, 6, 8, 13, 11, 9, 7, 15, 7, 12, 15, 9,
                                   11, 7, 13, 12, 11, 13, 6, 7, 14, 9, 13, 15, 14, 8, 13, 6, 5, 12, 7, 5, 11, 12, 14, 15, 14, 15, 9, 8,
                                   9, 14, 5, 6, 8, 6, 5, 12, 9, 15, 5, 11, 6, 8, 13, 12, 5, 12, 13, 14, 11, 8, 5, 6};
    static const uint8_t SR[80] = {8, 9, 9, 11, 13, 15, 15, 5, 7, 7, 8, 11, 14, 14, 12, 6, 9, 13, 15, 7, 12, 8, 9, 11, 7, 7, 12, 7,
                                   6, 15, 13, 11, 9, 7, 15, 11, 8, 6, 6, 14, 12, 13, 5, 14, 13, 13, 7, 5, 15, 5, 8, 11, 14, 14, 6, 14,
                                   6, 9, 12, 9, 12, 5, 15, 8, 8, 5, 12, 9, 12, 5, 14, 6, 8, 13, 6, 5, 15, 13, 11, 11};
    static const uint32_t KL[5] = {0x00000000u, 0x5A827999u, 0x6ED9EBA1u, 0x8F1BBCDCu, 0xA953FD4Eu};
    static const uint32_t KR[5] = {0x50A28BE6u, 0x5C4DD124u, 0x6D703EF3u, 0x7A6D76E9u, 0x00000000u};
    uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    std::vector<uint8_t> m(msg, msg + len);
    m.push_back(0x80);
    while (m.size() % 64 != 56) m.push_back(0);
    const uint64_t bits = (uint64_t)len * 8;
    for (int j = 0; j < 8; j++) m.push_back((uint8_t)(bits >> (8 * j)));
    auto f = [](int j, uint32_t x, uint32_t y, uint32_t z) -> uint32_t {
        switch (j / 16) {
            case 0: return x ^ y ^ z;
            case 1: return (x & y) | (~x & z);
            case 2: return (x | ~y) ^ z;
            case 3: return (x & z) | (y & ~z);
            default: return x ^ (y | ~z);
        }
    };
    for (size_t off = 0; off < m.size(); off += 64) {
        uint32_t X[16];
        for (int i = 0; i < 16; i++)
            X[i] = (uint32_t)m[off + 4 * i] | (uint32_t)m[off + 4 * i + 1] << 8 |
                   (uint32_t)m[off + 4 * i + 2] << 16 | (uint32_t)m[off + 4 * i + 3] << 24;
        uint32_t al = h[0], bl = h[1], cl = h[2], dl = h[3], el = h[4];
        uint32_t ar = h[0], br = h[1], cr = h[2], dr = h[3], er = h[4];
        for (int j = 0; j < 80; j++) {
            uint32_t t = rol(al + f(j, bl, cl, dl) + X[RL[j]] + KL[j / 16], SL[j]) + el;
            al = el; el = dl; dl = rol(cl, 10); cl = bl; bl = t;
            t = rol(ar + f(79 - j, br, cr, dr) + X[RR[j]] + KR[j / 16], SR[j]) + er;
            ar = er; er = dr; dr = rol(cr, 10); cr = br; br = t;
        }
        const uint32_t t = h[1] + cl + dr;
        h[1] = h[2] + dl + er;
        h[2] = h[3] + el + ar;
        h[3] = h[4] + al + br;
        h[4] = h[0] + bl + cr;
        h[0] = t;
    }
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(h[i] >> (8 * j));
}

void hash160(const uint8_t* msg, size_t len, uint8_t out[20]) {
    sha::Hasher s;
    s.update(msg, len);
    uint32_t w[8];
    s.finish(w);
    uint8_t d[32];
    sha::words_to_bytes(w, d);
    ripemd160(d, 32, out);
}

// scripts/src/bit_comm/winternitz.rs:282-297 generate_public_key: hash160 applied 1 + DIGITS times
// to secret || digit_index
constexpr int W_DIGITS = 15, W_N0 = 8, W_N1 = 2, W_N = 10, W_LOG_D = 4;

void winternitz_pubkey(const uint8_t* secret, size_t n, uint32_t digit_index, uint8_t out[20]) {
    std::vector<uint8_t> s(secret, secret + n);
    s.push_back((uint8_t)digit_index);
    hash160(s.data(), s.size(), out);
    for (int i = 0; i < W_DIGITS; i++) {
        uint8_t t[20];
        hash160(out, 20, t);
        memcpy(out, t, 20);
    }
}

enum : uint8_t {
    OP_0 = 0x00, OP_1 = 0x51, OP_3 = 0x53, OP_15 = 0x5f, OP_IF = 0x63, OP_ELSE = 0x67, OP_ENDIF = 0x68,
    OP_TOALTSTACK = 0x6b, OP_FROMALTSTACK = 0x6c, OP_2DROP = 0x6d, OP_DUP = 0x76, OP_PICK = 0x79,
    OP_ROLL = 0x7a, OP_ROT = 0x7b, OP_SWAP = 0x7c, OP_TUCK = 0x7d, OP_EQUALVERIFY = 0x88,
    OP_NEGATE = 0x8f, OP_ADD = 0x93, OP_SUB = 0x94, OP_MIN = 0xa3, OP_GREATERTHAN = 0xa0,
    OP_HASH160 = 0xa9,
};

// winternitz.rs:171-274 checksig_verify(pub_key)
void checksig_verify(std::vector<uint8_t>& s, const uint8_t pub[W_N][20]) {
    for (int digit_index = 0; digit_index < W_N; digit_index++) {
        s.push_back(OP_15);  // { DIGITS }
        s.push_back(OP_MIN);
        s.push_back(OP_DUP);
        s.push_back(OP_TOALTSTACK);
        s.push_back(OP_TOALTSTACK);
        for (int i = 0; i < W_DIGITS; i++) {
            s.push_back(OP_DUP);
            s.push_back(OP_HASH160);
        }
        s.push_back(OP_FROMALTSTACK);
        s.push_back(OP_PICK);
        s.push_back(20);  // push of the 20-byte public key
        s.insert(s.end(), pub[W_N - 1 - digit_index], pub[W_N - 1 - digit_index] + 20);
        s.push_back(OP_EQUALVERIFY);
        for (int i = 0; i < (W_DIGITS + 1) / 2; i++) s.push_back(OP_2DROP);
    }
    // 1. checksum of the message digits
    s.push_back(OP_FROMALTSTACK);
    s.push_back(OP_DUP);
    s.push_back(OP_NEGATE);
    for (int i = 1; i < W_N0; i++) {
        s.push_back(OP_FROMALTSTACK);
        s.push_back(OP_TUCK);
        s.push_back(OP_SUB);
    }
    script_push_int(s, W_DIGITS * W_N0);  // 120
    s.push_back(OP_ADD);
    // 2. the signed checksum digits
    s.push_back(OP_FROMALTSTACK);
    for (int i = 0; i < W_N1 - 1; i++) {
        for (int k = 0; k < W_LOG_D; k++) {
            s.push_back(OP_DUP);
            s.push_back(OP_ADD);
        }
        s.push_back(OP_FROMALTSTACK);
        s.push_back(OP_ADD);
    }
    // 3. equal
    s.push_back(OP_EQUALVERIFY);
    // digits -> bytes
    for (int i = 0; i < W_N0 / 2; i++) {
        s.push_back(OP_SWAP);
        for (int k = 0; k < W_LOG_D; k++) {
            s.push_back(OP_DUP);
            s.push_back(OP_ADD);
        }
        s.push_back(OP_ADD);
        if (i != W_N0 / 2 - 1) s.push_back(OP_TOALTSTACK);
    }
    for (int i = 0; i < W_N0 / 2 - 1; i++) s.push_back(OP_FROMALTSTACK);
}

// scripts/src/u32/u32_std.rs:122-173 u32_compress
void u32_compress(std::vector<uint8_t>& s) {
    s.push_back(OP_SWAP);
    s.push_back(OP_ROT);
    s.push_back(OP_3);
    s.push_back(OP_ROLL);
    s.push_back(OP_DUP);
    script_push_int(s, 127);
    s.push_back(OP_GREATERTHAN);
    s.push_back(OP_IF);
    script_push_int(s, 128);
    s.push_back(OP_SUB);
    s.push_back(OP_1);
    s.push_back(OP_ELSE);
    s.push_back(OP_0);
    s.push_back(OP_ENDIF);
    s.push_back(OP_TOALTSTACK);
    for (int i = 0; i < 3; i++) {  // OP_256MUL = eight doublings (scripts/src/pseudo.rs)
        for (int k = 0; k < 8; k++) {
            s.push_back(OP_DUP);
            s.push_back(OP_ADD);
        }
        s.push_back(OP_ADD);
    }
    s.push_back(OP_FROMALTSTACK);
    s.push_back(OP_IF);
    s.push_back(OP_NEGATE);
    s.push_back(OP_ENDIF);
}

}  // namespace

// Stand-in for `locking_script_with_type(CompressType::U32)` of a bit commitment over `u32_count`
// u32 limbs (1: BabyBear, 4: EF4), from the local copy of the construction:
//   per limb: checksig_verify(pubkey of secret_i) || u32_compress  (bit_comm_u32.rs:80-85)
//   several limbs: each followed by OP_TOALTSTACK, then as many OP_FROMALTSTACK
//   (scripts/src/bit_comm/bit_comm.rs:90-96,128-150).  Limb i uses secret || i.
std::vector<uint8_t> winternitz_lock_script(const uint8_t* secret, size_t n, uint32_t u32_count) {
    TS_REQUIRE(u32_count == 1 || u32_count == 4, TS_ERR_INVALID, "lock script: 1 or 4 limbs");
    std::vector<uint8_t> s;
    for (uint32_t limb = 0; limb < u32_count; limb++) {
        std::vector<uint8_t> sec(secret, secret + n);
        if (u32_count > 1) sec.push_back((uint8_t)limb);
        uint8_t pub[W_N][20];
        for (int d = 0; d < W_N; d++) winternitz_pubkey(sec.data(), sec.size(), (uint32_t)d, pub[d]);
        checksig_verify(s, pub);
        u32_compress(s);
        if (u32_count > 1) s.push_back(OP_TOALTSTACK);
    }
    if (u32_count > 1)
        for (uint32_t limb = 0; limb < u32_count; limb++) s.push_back(OP_FROMALTSTACK);
    return s;
}

// ------------------------------------------------------------------ CompleteTaptree
// complete_taptree.rs:5-10.  A power-of-two tree built on the device keeps every level in HBM
// (leaves first); a combined tree (complete_taptree.rs:90-133) keeps its two halves.
struct TapTree {
    // device part (a complete tree of n_leaves = 2^log_leaves leaves), empty for a combined tree
    Context* ctx = nullptr;
    DevBuf<uint32_t> digests;  // (2 n - 1) x 8 state words
    unsigned log_leaves = 0;
    // combined tree
    std::shared_ptr<TapTree> left, right;
    bool left_first = true;  // NodeInfo::combine_with_order's flag: left.root <= right.root
    uint64_t n_leaves = 0;
    uint32_t root[8] = {0};
};

static void pack_words(const uint8_t* bytes, size_t len, std::vector<uint32_t>& words) {
    for (size_t i = 0; i < len; i += 4) {
        uint32_t w = 0;
        for (size_t j = 0; j < 4; j++) w |= (uint32_t)(i + j < len ? bytes[i + j] : 0) << (24 - 8 * j);
        words.push_back(w);
    }
}

// new_with_scripts (complete_taptree.rs:67-75) + build_tree (builder.rs:38-93)
std::shared_ptr<TapTree> taptree_from_scripts(Context& ctx, const uint8_t* scripts, const uint64_t* offsets,
                                              uint64_t n_leaves) {
    TS_REQUIRE(n_leaves >= 1 && (n_leaves & (n_leaves - 1)) == 0, TS_ERR_INVARIANT,
               "build_tree: the leaf count must be a power of two (builder.rs:40)");
    TS_REQUIRE(n_leaves <= (1ull << 27), TS_ERR_INVALID, "taptree: too many leaves");
    std::vector<uint32_t> words;
    std::vector<uint64_t> word_off(n_leaves), byte_len(n_leaves);
    for (uint64_t i = 0; i < n_leaves; i++) {
        TS_REQUIRE(offsets[i + 1] >= offsets[i] && offsets[i + 1] - offsets[i] <= 0xffffffffull,
                   TS_ERR_INVALID, "taptree: bad script offsets");
        word_off[i] = words.size();
        byte_len[i] = offsets[i + 1] - offsets[i];
        pack_words(scripts + offsets[i], byte_len[i], words);
    }
    words.push_back(0);  // a zero-length last script still has a valid address
    auto t = std::make_shared<TapTree>();
    t->ctx = &ctx;
    t->n_leaves = n_leaves;
    t->log_leaves = log2_strict(n_leaves);
    t->digests = DevBuf<uint32_t>(&ctx, (2 * n_leaves - 1) * 8);
    DevBuf<uint32_t> d_words(&ctx, words.size());
    DevBuf<uint64_t> d_off(&ctx, n_leaves), d_len(&ctx, n_leaves);
    h2d(ctx, d_words.p, words.data(), words.size() * 4);
    h2d(ctx, d_off.p, word_off.data(), n_leaves * 8);
    h2d(ctx, d_len.p, byte_len.data(), n_leaves * 8);
    launch_tapleaf_blob(ctx, d_words.p, d_off.p, d_len.p, n_leaves, tap_mid(), t->digests.p);
    launch_tapbranch_levels(ctx, t->digests.p, 2 * n_leaves - 1, t->log_leaves, 1, tap_mid());
    d2h_sync(ctx, t->root, t->digests.p + (2 * n_leaves - 2) * 8, 32);
    return t;
}

// combine (complete_taptree.rs:90-133): leaves of `a` keep indices [0, a.n), those of `b` follow
std::shared_ptr<TapTree> taptree_combine(std::shared_ptr<TapTree> a, std::shared_ptr<TapTree> b) {
    auto t = std::make_shared<TapTree>();
    t->left = a;
    t->right = b;
    t->n_leaves = a->n_leaves + b->n_leaves;
    t->left_first = !sha::digest_less(b->root, a->root);
    sha::tapbranch(tap_mid().branch, a->root, b->root, t->root);
    return t;
}

// get_leaf_proof (complete_taptree.rs:155-159): the leaf hash and the sibling path, leaf-most first
void taptree_leaf_proof(const TapTree& t, uint64_t index, uint32_t leaf[8], std::vector<uint32_t>& path) {
    TS_REQUIRE(index < t.n_leaves, TS_ERR_INVALID, "taptree: leaf index out of range");
    if (t.left) {
        const bool in_left = index < t.left->n_leaves;
        taptree_leaf_proof(in_left ? *t.left : *t.right, in_left ? index : index - t.left->n_leaves, leaf, path);
        const uint32_t* sib = in_left ? t.right->root : t.left->root;
        path.insert(path.end(), sib, sib + 8);
        return;
    }
    Context& ctx = *t.ctx;
    TS_HIP(hipSetDevice(ctx.device));
    const uint32_t zero = 0;
    DevBuf<uint32_t> d_tree(&ctx, 1);
    DevBuf<uint64_t> d_idx(&ctx, 1);
    DevBuf<uint32_t> d_out(&ctx, 8 * (size_t)std::max(1u, t.log_leaves) + 8);
    h2d(ctx, d_tree.p, &zero, 4);
    h2d(ctx, d_idx.p, &index, 8);
    launch_tap_gather_paths(ctx, t.digests.p, 2 * t.n_leaves - 1, t.log_leaves, d_tree.p, d_idx.p, 1, d_out.p);
    TS_HIP(hipMemcpyAsync(d_out.p + 8 * (size_t)t.log_leaves, t.digests.p + 8 * index, 32,
                          hipMemcpyDeviceToDevice, ctx.stream));
    std::vector<uint32_t> g(8 * (size_t)t.log_leaves + 8);
    d2h_sync(ctx, g.data(), d_out.p, g.size() * 4);
    memcpy(leaf, &g[8 * (size_t)t.log_leaves], 32);
    path.insert(path.end(), g.begin(), g.begin() + 8 * (size_t)t.log_leaves);
}

// verify_inclusion (complete_taptree.rs:53-64)
bool taptree_verify_inclusion(const uint32_t root[8], const uint32_t leaf[8], const uint32_t* path,
                              size_t depth) {
    uint32_t cur[8];
    memcpy(cur, leaf, 32);
    for (size_t l = 0; l < depth; l++) {
        uint32_t nxt[8];
        sha::tapbranch(tap_mid().branch, cur, path + 8 * l, nxt);
        memcpy(cur, nxt, 32);
    }
    return memcmp(cur, root, 32) == 0;
}

// ------------------------------------------------------------------ TapTreeMmcs
// taptree_mmcs.rs:24-28: ProverData = Vec<CommitedData>, one per query; here the matrices are held
// once and the num_queries trees side by side.
struct TapMmcsData {
    Context* ctx = nullptr;
    std::vector<DevBuf<uint32_t>> storage;  // column-major matrices
    std::vector<ColMat> mats;               // commit order (non-increasing heights)
    unsigned log_height = 0;
    uint32_t u32_size = 1, n_evals = 0, num_queries = 0;
    DevBuf<uint32_t> trees;  // [Q][2N-1][8]
    DevBuf<const uint32_t*> cols;
    std::vector<uint8_t> lock_bytes;     // every lock script of every tree
    std::vector<uint64_t> lock_offsets;  // [Q (1 + n_evals) + 1]
    std::vector<uint32_t> roots;         // [Q][8] state words
};

std::unique_ptr<TapMmcsData> tap_mmcs_commit(Context& ctx, std::vector<DeviceMatrix>& inputs,
                                             uint32_t u32_size, uint32_t num_queries,
                                             const uint8_t* lock_scripts, const uint64_t* lock_offsets) {
    TS_REQUIRE(!inputs.empty() && inputs.size() <= (size_t)MAX_BATCH_MATS, TS_ERR_INVALID,
               "tap mmcs: between 1 and 64 matrices");
    TS_REQUIRE(u32_size == 1 || u32_size == 4, TS_ERR_INVALID, "tap mmcs: F::U32_SIZE is 1 or 4 (tcs/mod.rs:239-246)");
    TS_REQUIRE(num_queries >= 1 && num_queries <= 4096, TS_ERR_INVALID, "tap mmcs: num_queries in [1, 4096]");
    auto d = std::make_unique<TapMmcsData>();
    d->ctx = &ctx;
    d->u32_size = u32_size;
    d->num_queries = num_queries;
    uint64_t max_h = inputs[0].height;
    uint32_t total_w = 0;
    for (size_t i = 0; i < inputs.size(); i++) {
        DeviceMatrix& m = inputs[i];
        TS_REQUIRE(m.buf.p && m.layout == DeviceMatrix::ROW_MAJOR && m.width >= 1, TS_ERR_INVALID,
                   "tap mmcs: uploaded row-major matrices expected");
        log2_strict(m.height);
        // padding_matrix sorts by height but open_batch returns rows in the given order and asserts
        // both agree (taptree_mmcs.rs:68-72): only a non-increasing order ever passes that assert
        TS_REQUIRE(i == 0 || m.height <= inputs[i - 1].height, TS_ERR_INVARIANT,
                   "tap mmcs: matrices must come tallest first (taptree_mmcs.rs:68-72)");
        TS_REQUIRE(m.width % u32_size == 0, TS_ERR_INVALID, "tap mmcs: width not a multiple of U32_SIZE");
        total_w += m.width;
    }
    d->log_height = log2_strict(max_h);
    d->n_evals = total_w / u32_size;
    const size_t n_seg = 1 + (size_t)d->n_evals;
    TS_REQUIRE(lock_scripts && lock_offsets, TS_ERR_INVALID, "tap mmcs: lock scripts missing");
    d->lock_offsets.assign(lock_offsets, lock_offsets + (size_t)num_queries * n_seg + 1);
    for (size_t i = 0; i + 1 < d->lock_offsets.size(); i++)
        TS_REQUIRE(d->lock_offsets[i + 1] >= d->lock_offsets[i] &&
                       d->lock_offsets[i + 1] - d->lock_offsets[i] < (1ull << 24),
                   TS_ERR_INVALID, "tap mmcs: bad lock script offsets");
    d->lock_bytes.assign(lock_scripts + d->lock_offsets.front(), lock_scripts + d->lock_offsets.back());
    const uint64_t base = d->lock_offsets.front();
    for (auto& o : d->lock_offsets) o -= base;

    std::vector<const uint32_t*> cols;
    std::vector<uint8_t> shifts;
    for (auto& m : inputs) {
        DevBuf<uint32_t> cmaj(&ctx, (size_t)m.height * m.width);
        launch_transpose_plain(ctx, m.buf.p, cmaj.p, m.height, m.width, m.height);
        ColMat cm;
        cm.d = cmaj.p;
        cm.height = m.height;
        cm.width = m.width;
        cm.col_stride = m.height;
        for (uint32_t c = 0; c < m.width; c++) {
            cols.push_back(cm.d + (uint64_t)c * cm.col_stride);
            shifts.push_back((uint8_t)(d->log_height - log2_strict(m.height)));
        }
        d->mats.push_back(cm);
        d->storage.push_back(std::move(cmaj));
        m.buf.reset();
    }
    d->cols = DevBuf<const uint32_t*>(&ctx, cols.size());  // (kept for callers that walk the columns)
    h2d(ctx, d->cols.p, cols.data(), cols.size() * sizeof(const uint32_t*));
    TapLocks tl;
    tl.bytes = d->lock_bytes.data();
    tl.offsets = d->lock_offsets.data();
    tl.n_scripts = d->lock_offsets.size() - 1;
    d->trees = tap_build_trees(ctx, cols, shifts, 1, d->log_height, u32_size, num_queries, tl, 0, d->roots);
    // the ABI of the stand-alone MMCS hands out digests as bytes: keep the state words here
    for (auto& w : d->roots) w = __builtin_bswap32(w);
    return d;
}

DevBuf<uint32_t> tap_build_trees(Context& ctx, const std::vector<const uint32_t*>& cols,
                                 const std::vector<uint8_t>& shifts, uint32_t elem_stride,
                                 unsigned log_height, uint32_t u32_size, uint32_t num_queries,
                                 const TapLocks& locks, size_t first_lock, std::vector<uint32_t>& roots_words) {
    const uint64_t N = 1ull << log_height;
    const uint32_t n_evals = (uint32_t)(cols.size() / u32_size);
    const size_t n_seg = 1 + (size_t)n_evals;
    TS_REQUIRE(cols.size() % u32_size == 0 && cols.size() == shifts.size(), TS_ERR_INVALID, "tap trees: columns");
    TS_REQUIRE(locks.offsets && first_lock + (size_t)num_queries * n_seg <= locks.n_scripts, TS_ERR_INVALID,
               "tap trees: the lock-script table is too short for this commitment");
    // segments as big-endian words
    std::vector<uint32_t> seg_words;
    std::vector<uint64_t> seg_word_off((size_t)num_queries * n_seg), const_len(num_queries, 0);
    std::vector<uint32_t> seg_len((size_t)num_queries * n_seg);
    for (size_t s = 0; s < (size_t)num_queries * n_seg; s++) {
        const uint64_t o0 = locks.offsets[first_lock + s], o1 = locks.offsets[first_lock + s + 1];
        TS_REQUIRE(o1 >= o0 && o1 - o0 < (1ull << 24), TS_ERR_INVALID, "tap trees: bad lock script offsets");
        seg_word_off[s] = seg_words.size();
        seg_len[s] = (uint32_t)(o1 - o0);
        const_len[s / n_seg] += seg_len[s];
        pack_words(locks.bytes + o0, seg_len[s], seg_words);
    }
    seg_words.push_back(0);
    DevBuf<const uint32_t*> d_cols(&ctx, cols.size());
    DevBuf<uint8_t> d_shift(&ctx, shifts.size());
    DevBuf<uint32_t> d_seg_words(&ctx, seg_words.size()), d_seg_len(&ctx, seg_len.size());
    DevBuf<uint64_t> d_seg_off(&ctx, seg_word_off.size()), d_const(&ctx, const_len.size());
    h2d(ctx, d_cols.p, cols.data(), cols.size() * sizeof(const uint32_t*));
    h2d(ctx, d_shift.p, shifts.data(), shifts.size());
    h2d(ctx, d_seg_words.p, seg_words.data(), seg_words.size() * 4);
    h2d(ctx, d_seg_len.p, seg_len.data(), seg_len.size() * 4);
    h2d(ctx, d_seg_off.p, seg_word_off.data(), seg_word_off.size() * 8);
    h2d(ctx, d_const.p, const_len.data(), const_len.size() * 8);
    const uint64_t stride = 2 * N - 1;
    DevBuf<uint32_t> trees(&ctx, (size_t)num_queries * stride * 8);
    TapTemplate t;
    t.seg_words = d_seg_words.p;
    t.seg_word_off = d_seg_off.p;
    t.seg_len = d_seg_len.p;
    t.const_len = d_const.p;
    t.cols = d_cols.p;
    t.shift = d_shift.p;
    t.n_evals = n_evals;
    t.u32_size = u32_size;
    t.elem_stride = elem_stride;
    t.tree_stride = stride;
    t.prefix = nullptr;
    t.n_len = 5 * ((uint32_t)cols.size() + 1) + 1;  // a push takes 1..6 bytes
    // the state after the first lock script, tabulated per script length, when that is less work
    // than hashing that script for every leaf (TS_TAP_PREFIX=0 switches it off for A/B runs)
    static const bool use_prefix = [] {
        const char* e = getenv("TS_TAP_PREFIX");
        return !(e && e[0] == '0');
    }();
    DevBuf<uint32_t> d_prefix;
    if (use_prefix && N >= 4ull * t.n_len && (uint64_t)num_queries * t.n_len * TAP_PREFIX_WORDS * 4 <= (64u << 20)) {
        d_prefix = DevBuf<uint32_t>(&ctx, (size_t)num_queries * t.n_len * TAP_PREFIX_WORDS);
        launch_tap_prefix(ctx, t, num_queries, tap_mid(), d_prefix.p);
        t.prefix = d_prefix.p;
    }
    {
        StageTimer tm(&ctx, "taptree leaves");
        launch_tapleaf_template(ctx, t, N, num_queries, tap_mid(), trees.p);
    }
    {
        StageTimer tm(&ctx, "taptree branches");
        launch_tapbranch_levels(ctx, trees.p, stride, log_height, num_queries, tap_mid());
    }
    // roots: the last digest of every tree
    DevBuf<uint32_t> d_roots(&ctx, (size_t)num_queries * 8);
    for (uint32_t q = 0; q < num_queries; q++)
        TS_HIP(hipMemcpyAsync(d_roots.p + 8 * (size_t)q, trees.p + ((size_t)q * stride + stride - 1) * 8, 32,
                              hipMemcpyDeviceToDevice, ctx.stream));
    roots_words.resize((size_t)num_queries * 8);
    d2h_sync(ctx, roots_words.data(), d_roots.p, roots_words.size() * 4);  // also covers the pageable tables
    for (auto& w : roots_words) w = __builtin_bswap32(w);  // state word -> its bytes read little-endian
    return trees;
}

// verify_batch on words: leaf rebuilt from the lock scripts, the index and the opened values
bool tap_verify_words(const TapLocks& locks, size_t first, uint32_t n_evals, uint32_t u32_size,
                      uint64_t index, const uint32_t* values, const uint32_t* path_words, size_t depth,
                      const uint32_t root_words[8]) {
    if (!locks.offsets || first + n_evals + 1 > locks.n_scripts || depth > 128) return false;
    for (uint32_t c = 0; c < n_evals * u32_size; c++)
        if (values[c] >= P) return false;
    std::vector<std::pair<const uint8_t*, size_t>> ls;
    for (uint32_t s = 0; s <= n_evals; s++) {
        const uint64_t o0 = locks.offsets[first + s], o1 = locks.offsets[first + s + 1];
        if (o1 < o0) return false;
        ls.push_back({locks.bytes + o0, (size_t)(o1 - o0)});
    }
    const std::vector<uint8_t> script = tap_leaf_script(ls, index, values, n_evals, u32_size);
    uint32_t leaf[8], root[8];
    tapleaf_hash(script.data(), script.size(), leaf);
    std::vector<uint32_t> pw(8 * depth);
    for (size_t k = 0; k < 8 * depth; k++) pw[k] = __builtin_bswap32(path_words[k]);
    for (int k = 0; k < 8; k++) root[k] = __builtin_bswap32(root_words[k]);
    return taptree_verify_inclusion(root, leaf, pw.data(), depth);
}

// open_batch (taptree_mmcs.rs:46-75): the rows at index >> bits_reduced, the leaf's sibling path in
// tree `query_times_index`
void tap_mmcs_open_batch(const TapMmcsData& d, uint32_t query_times_index, uint64_t index,
                         std::vector<uint32_t>& rows, std::vector<uint32_t>& path) {
    Context& ctx = *d.ctx;
    TS_REQUIRE(query_times_index < d.num_queries, TS_ERR_INVALID, "tap mmcs: query_times_index out of range");
    TS_REQUIRE(index < (1ull << d.log_height), TS_ERR_INVALID, "tap mmcs: index out of range");
    rows.clear();
    uint32_t total = 0;
    for (auto& m : d.mats) total += m.width;
    DevBuf<uint32_t> d_rows(&ctx, total);
    uint32_t off = 0;
    for (auto& m : d.mats) {
        const uint64_t r = index >> (d.log_height - log2_strict(m.height));
        // a strided 2-D copy: one element per column
        TS_HIP(hipMemcpy2DAsync(d_rows.p + off, 4, m.d + r, m.col_stride * 4, 4, m.width,
                                hipMemcpyDeviceToDevice, ctx.stream));
        off += m.width;
    }
    rows.resize(total);
    DevBuf<uint32_t> d_tree(&ctx, 1);
    DevBuf<uint64_t> d_idx(&ctx, 1);
    DevBuf<uint32_t> d_path(&ctx, 8 * (size_t)std::max(1u, d.log_height));
    h2d(ctx, d_tree.p, &query_times_index, 4);
    h2d(ctx, d_idx.p, &index, 8);
    launch_tap_gather_paths(ctx, d.trees.p, (2ull << d.log_height) - 1, d.log_height, d_tree.p, d_idx.p, 1,
                            d_path.p);
    path.resize(8 * (size_t)d.log_height);
    TS_HIP(hipMemcpyAsync(rows.data(), d_rows.p, total * 4, hipMemcpyDeviceToHost, ctx.stream));
    d2h_sync(ctx, path.data(), d_path.p, path.size() * 4);
}

}  // namespace ts

// ------------------------------------------------------------------ C ABI
struct ts_taptree {
    std::shared_ptr<ts::TapTree> t;
};
struct ts_tap_mmcs_data {
    std::unique_ptr<ts::TapMmcsData> d;
};

namespace {

template <class F>
ts_status tap_guard(ts::Context* ctx, F&& f) {
    try {
        if (ctx) TS_HIP(hipSetDevice(ctx->device));
        f();
        return TS_OK;
    } catch (const ts::Error& e) {
        if (ctx) ctx->last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        return TS_ERR_OOM;
    } catch (const std::exception& e) {
        if (ctx) ctx->last_error = e.what();
        return TS_ERR_INVALID;
    }
}
ts::Context* ctx_of(ts_ctx* c) { return c ? &c->ctx : nullptr; }

void export_digest(const uint32_t w[8], uint8_t out[32]) { ts::sha::words_to_bytes(w, out); }

}  // namespace

extern "C" {

ts_status ts_tapleaf_hash(const uint8_t* script, size_t len, uint8_t out[32]) {
    if ((!script && len) || !out) return TS_ERR_INVALID;
    return tap_guard(nullptr, [&] {
        uint32_t w[8];
        ts::tapleaf_hash(script, len, w);
        export_digest(w, out);
    });
}

ts_status ts_tapbranch_hash(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    if (!a || !b || !out) return TS_ERR_INVALID;
    uint32_t wa[8], wb[8], wo[8];
    ts::sha::bytes_to_words(a, wa);
    ts::sha::bytes_to_words(b, wb);
    ts::sha::tapbranch(ts::tap_mid().branch, wa, wb, wo);
    export_digest(wo, out);
    return TS_OK;
}

ts_status ts_tap_winternitz_lock_script(const uint8_t* secret, size_t secret_len, uint32_t u32_count,
                                        uint8_t* out, size_t cap, size_t* len_out) {
    if ((!secret && secret_len) || !len_out) return TS_ERR_INVALID;
    return tap_guard(nullptr, [&] {
        std::vector<uint8_t> s = ts::winternitz_lock_script(secret, secret_len, u32_count);
        *len_out = s.size();
        TS_REQUIRE(out && s.size() <= cap, ts::TS_ERR_BUFFER, "lock script buffer too small");
        memcpy(out, s.data(), s.size());
    });
}

ts_status ts_tap_leaf_script(const uint8_t* lock_scripts, const uint64_t* lock_offsets, uint32_t n_evals,
                             uint32_t u32_size, uint64_t index, const uint32_t* values, uint8_t* out,
                             size_t cap, size_t* len_out) {
    if (!lock_scripts || !lock_offsets || !len_out || (!values && n_evals)) return TS_ERR_INVALID;
    return tap_guard(nullptr, [&] {
        TS_REQUIRE(u32_size == 1 || u32_size == 4, ts::TS_ERR_INVALID, "U32_SIZE is 1 or 4");
        std::vector<std::pair<const uint8_t*, size_t>> locks;
        for (uint32_t s = 0; s <= n_evals; s++) {
            TS_REQUIRE(lock_offsets[s + 1] >= lock_offsets[s], ts::TS_ERR_INVALID, "bad lock offsets");
            locks.push_back({lock_scripts + lock_offsets[s], (size_t)(lock_offsets[s + 1] - lock_offsets[s])});
        }
        std::vector<uint8_t> s = ts::tap_leaf_script(locks, index, values, n_evals, u32_size);
        *len_out = s.size();
        TS_REQUIRE(out && s.size() <= cap, ts::TS_ERR_BUFFER, "leaf script buffer too small");
        memcpy(out, s.data(), s.size());
    });
}

ts_status ts_taptree_from_scripts(ts_ctx* ctx, const uint8_t* scripts, const uint64_t* offsets,
                                  uint64_t n_leaves, ts_taptree** out) {
    if (!ctx || !scripts || !offsets || !out) return TS_ERR_INVALID;
    *out = nullptr;
    return tap_guard(ctx_of(ctx), [&] {
        auto t = std::make_unique<ts_taptree>();
        t->t = ts::taptree_from_scripts(*ctx_of(ctx), scripts, offsets, n_leaves);
        *out = t.release();
    });
}

ts_status ts_taptree_combine(const ts_taptree* a, const ts_taptree* b, ts_taptree** out) {
    if (!a || !b || !out || !a->t || !b->t) return TS_ERR_INVALID;
    *out = nullptr;
    return tap_guard(nullptr, [&] {
        auto t = std::make_unique<ts_taptree>();
        t->t = ts::taptree_combine(a->t, b->t);
        *out = t.release();
    });
}

ts_status ts_taptree_info(const ts_taptree* t, uint64_t* leaf_count, uint8_t root[32]) {
    if (!t || !t->t) return TS_ERR_INVALID;
    if (leaf_count) *leaf_count = t->t->n_leaves;
    if (root) export_digest(t->t->root, root);
    return TS_OK;
}

ts_status ts_taptree_leaf_proof(const ts_taptree* t, uint64_t index, uint8_t leaf_hash[32],
                                uint8_t* path, uint32_t cap_nodes, uint32_t* depth) {
    if (!t || !t->t || !leaf_hash || !depth) return TS_ERR_INVALID;
    ts::Context* ctx = nullptr;
    for (const ts::TapTree* p = t->t.get(); p; p = p->left.get())
        if (p->ctx) ctx = p->ctx;
    return tap_guard(ctx, [&] {
        uint32_t leaf[8];
        std::vector<uint32_t> pw;
        ts::taptree_leaf_proof(*t->t, index, leaf, pw);
        export_digest(leaf, leaf_hash);
        *depth = (uint32_t)(pw.size() / 8);
        TS_REQUIRE(pw.empty() || (path && *depth <= cap_nodes), ts::TS_ERR_BUFFER, "path buffer too small");
        for (uint32_t l = 0; l < *depth; l++) export_digest(&pw[8 * (size_t)l], path + 32 * (size_t)l);
    });
}

int ts_taptree_verify_inclusion(const uint8_t root[32], const uint8_t leaf_hash[32], const uint8_t* path,
                                uint32_t depth) {
    if (!root || !leaf_hash || (!path && depth) || depth > 128) return 0;  // TAPROOT_CONTROL_MAX_NODE_COUNT
    uint32_t r[8], l[8];
    ts::sha::bytes_to_words(root, r);
    ts::sha::bytes_to_words(leaf_hash, l);
    std::vector<uint32_t> pw(8 * (size_t)depth);
    for (uint32_t k = 0; k < depth; k++) ts::sha::bytes_to_words(path + 32 * (size_t)k, &pw[8 * (size_t)k]);
    return ts::taptree_verify_inclusion(r, l, pw.data(), depth) ? 1 : 0;
}

void ts_taptree_free(ts_taptree* t) { delete t; }

ts_status ts_tap_mmcs_commit(ts_ctx* ctx, uint32_t n_mats, ts_matrix* const* mats, uint32_t u32_size,
                             uint32_t num_queries, const uint8_t* lock_scripts,
                             const uint64_t* lock_offsets, uint8_t* roots_out, ts_tap_mmcs_data** out) {
    if (!ctx || !mats || !out || !lock_scripts || !lock_offsets) return TS_ERR_INVALID;
    *out = nullptr;
    return tap_guard(ctx_of(ctx), [&] {
        TS_REQUIRE(n_mats >= 1 && n_mats <= (uint32_t)ts::MAX_BATCH_MATS, ts::TS_ERR_INVALID,
                   "tap mmcs: between 1 and 64 matrices");
        std::vector<ts::DeviceMatrix> ms;
        for (uint32_t i = 0; i < n_mats; i++) {
            TS_REQUIRE(mats[i] && mats[i]->m.buf.p, ts::TS_ERR_INVALID,
                       "tap mmcs: null or consumed matrix");
            ms.push_back(std::move(mats[i]->m));
        }
        auto d = std::make_unique<ts_tap_mmcs_data>();
        try {
            d->d = ts::tap_mmcs_commit(*ctx_of(ctx), ms, u32_size, num_queries, lock_scripts, lock_offsets);
        } catch (...) {
            // nothing was consumed if the arguments were refused: give the matrices back
            for (uint32_t i = 0; i < n_mats; i++)
                if (ms[i].buf.p) mats[i]->m = std::move(ms[i]);
            throw;
        }
        if (roots_out)
            for (uint32_t q = 0; q < num_queries; q++) export_digest(&d->d->roots[8 * (size_t)q], roots_out + 32 * (size_t)q);
        *out = d.release();
    });
}

ts_status ts_tap_mmcs_info(const ts_tap_mmcs_data* d, uint32_t* n_mats, uint32_t* log_max_height,
                           uint32_t* n_evals, uint32_t* num_queries) {
    if (!d || !d->d) return TS_ERR_INVALID;
    if (n_mats) *n_mats = (uint32_t)d->d->mats.size();
    if (log_max_height) *log_max_height = d->d->log_height;
    if (n_evals) *n_evals = d->d->n_evals;
    if (num_queries) *num_queries = d->d->num_queries;
    return TS_OK;
}

ts_status ts_tap_mmcs_open_batch(const ts_tap_mmcs_data* d, uint32_t query_times_index, uint64_t index,
                                 uint32_t* rows_out, uint8_t* path_out, uint8_t* script_out,
                                 size_t script_cap, size_t* script_len) {
    if (!d || !d->d || !rows_out || !path_out) return TS_ERR_INVALID;
    return tap_guard(d->d->ctx, [&] {
        std::vector<uint32_t> rows, path;
        ts::tap_mmcs_open_batch(*d->d, query_times_index, index, rows, path);
        memcpy(rows_out, rows.data(), rows.size() * 4);
        for (unsigned l = 0; l < d->d->log_height; l++) export_digest(&path[8 * (size_t)l], path_out + 32 * (size_t)l);
        if (script_len) {  // CommitedProof.leaf (tcs/mod.rs:103-108): the opened leaf's script
            const size_t n_seg = 1 + (size_t)d->d->n_evals;
            std::vector<std::pair<const uint8_t*, size_t>> locks;
            for (size_t s = 0; s < n_seg; s++) {
                const size_t k = (size_t)query_times_index * n_seg + s;
                locks.push_back({d->d->lock_bytes.data() + d->d->lock_offsets[k],
                                 (size_t)(d->d->lock_offsets[k + 1] - d->d->lock_offsets[k])});
            }
            std::vector<uint8_t> s = ts::tap_leaf_script(locks, index, rows.data(), d->d->n_evals, d->d->u32_size);
            *script_len = s.size();
            TS_REQUIRE(script_out && s.size() <= script_cap, ts::TS_ERR_BUFFER, "leaf script buffer too small");
            memcpy(script_out, s.data(), s.size());
        }
    });
}

ts_status ts_tap_mmcs_verify_batch(const uint8_t* lock_scripts, const uint64_t* lock_offsets,
                                   uint32_t n_evals, uint32_t u32_size, uint64_t index,
                                   const uint32_t* opened_values, const uint8_t* path, uint32_t depth,
                                   const uint8_t root[32], int* ok) {
    if (!lock_scripts || !lock_offsets || !opened_values || !root || !ok || (!path && depth)) return TS_ERR_INVALID;
    *ok = 0;
    return tap_guard(nullptr, [&] {
        TS_REQUIRE(u32_size == 1 || u32_size == 4, ts::TS_ERR_INVALID, "U32_SIZE is 1 or 4");
        TS_REQUIRE(depth <= 128, ts::TS_ERR_INVALID, "taproot paths have at most 128 nodes");
        for (uint32_t c = 0; c < n_evals * u32_size; c++)
            TS_REQUIRE(opened_values[c] < ts::P, ts::TS_ERR_INVALID, "non-canonical opened value");
        std::vector<std::pair<const uint8_t*, size_t>> locks;
        for (uint32_t s = 0; s <= n_evals; s++)
            locks.push_back({lock_scripts + lock_offsets[s], (size_t)(lock_offsets[s + 1] - lock_offsets[s])});
        std::vector<uint8_t> script = ts::tap_leaf_script(locks, index, opened_values, n_evals, u32_size);
        uint32_t leaf[8];
        ts::tapleaf_hash(script.data(), script.size(), leaf);
        uint8_t lb[32];
        export_digest(leaf, lb);
        *ok = ts_taptree_verify_inclusion(root, lb, path, depth);
    });
}

void ts_tap_mmcs_free(ts_tap_mmcs_data* d) { delete d; }

}  // extern "C"
