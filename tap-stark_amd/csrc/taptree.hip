// Taptree kernels: the reference's real MMCS commits to a Bitcoin
// taptree (basic/src/mmcs/taptree_mmcs.rs:101-114 -> basic/src/tcs/mod.rs:238-292 commit_polys ->
// basic/src/tcs/complete_taptree.rs:67-75 new_with_scripts -> basic/src/tcs/builder.rs:38-93).
//
//   k_tapleaf_blob       TapLeaf hashes of host-supplied leaf scripts (any bytes)
//   k_tapleaf_template   TapLeaf hashes of the MMCS leaf scripts, ASSEMBLED ON THE DEVICE from the
//                        per-tree lock-script segments and the committed matrices
//                        (tcs/mod.rs:197-225 generate_script, tcs/mod.rs:339-378 padding_matrix):
//                        the scripts (kilobytes per leaf, Q trees) never exist in memory
//   k_tapbranch_level    one tree level: parent = TapBranch(sorted(left, right)), all Q trees at once
//   k_tap_gather_paths   sibling paths of Q (tree, leaf) pairs
//
// One thread hashes one leaf.  SHA-256 is a serial chain per leaf and the script is variable-length
// (script-number pushes are minimally encoded), so every thread keeps its own streaming state: the
// 8 state words and a partial word in registers, the 16-word block in LDS ([word][thread]: the word
// index differs between lanes but the bank does not, so accesses are conflict-free).  Constant
// segments are stored as big-endian words and enter the stream with two shifts per word, whatever
// the byte alignment.  These kernels are VALU-bound (SHA-256 rounds), not HBM-bound: a leaf reads
// 4 bytes per committed element and runs ~12 compressions per lock script.
#include "kernels.hpp"
#include "sha256.hpp"
#include "taptree.hpp"

namespace ts {

namespace {

constexpr int TPB = 128;

// Compressions happen at points every lane of a wavefront reaches together.  Lanes drift apart by a
// few bytes (a value below 2^23 is pushed with fewer bytes: 3 % of the leaves of an EF4-pair matrix,
// so most wavefronts hold one), and a stream that compresses "when my 16th word arrives" then runs
// every block twice for such a wavefront, once per group of lanes.  Here a lane's words go to a
// 32-word ring; inside a constant segment every 16 source words add exactly 16 stream words to every
// lane, whatever its alignment, so the compression after them is unconditional (each lane has
// 16..31 words pending); only the few words around the pushes are drained under a per-lane test.
constexpr uint32_t RING = 32;

struct Stream {
    uint32_t h[8];
    uint32_t acc;    // partial word: `a` leading bytes valid (big-endian), the rest zero
    uint32_t a;      // 0..3
    uint32_t wpos;   // words written
    uint32_t rpos;   // words compressed (a multiple of 16)
    uint32_t* ring;  // this thread's column of the [RING][TPB] image

    __device__ __forceinline__ void init(const uint32_t* iv, uint32_t* lds_column) {
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = iv[k];
        acc = 0; a = 0; wpos = 0; rpos = 0;
        ring = lds_column;
    }
    __device__ __forceinline__ uint32_t pending() const { return wpos - rpos; }
    __device__ __forceinline__ void compress_block() {  // needs pending() >= 16
        uint32_t m[16];
#pragma unroll
        for (int i = 0; i < 16; i++) m[i] = ring[((rpos + i) & (RING - 1)) * TPB];
        sha::compress(h, m);
        rpos += 16;
    }
    // brings pending() below 16; every put_* below leaves at most RING words pending provided it
    // started below 16, and the callers drain between them
    __device__ __forceinline__ void drain() {
        if (pending() >= 16) compress_block();
    }
    __device__ __forceinline__ void emit_word(uint32_t x) {
        ring[(wpos & (RING - 1)) * TPB] = x;
        wpos++;
    }
    __device__ __forceinline__ void put_byte(uint32_t b) {
        acc |= b << (24 - 8 * a);
        if (++a == 4) {
            emit_word(acc);
            acc = 0;
            a = 0;
        }
    }
    // one big-endian source word = 4 stream bytes = exactly one stream word
    __device__ __forceinline__ void put_word(uint32_t x) {
        if (a == 0) {
            emit_word(x);
        } else {
            emit_word(acc | (x >> (8 * a)));
            acc = x << (32 - 8 * a);
        }
    }
    // bytes [0, len) of a segment stored as big-endian words (zero padded); pending() < 16 on entry
    // and on return
    __device__ __forceinline__ void put_segment(const uint32_t* __restrict__ words, uint32_t len) {
        const uint32_t full = len >> 2;
        uint32_t k = 0;
        for (; k + 16 <= full; k += 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) put_word(words[k + i]);
            compress_block();  // every lane alike
        }
        for (; k < full; k++) put_word(words[k]);
        const uint32_t tail = len & 3;
        if (tail) {
            const uint32_t x = words[full];
            for (uint32_t j = 0; j < tail; j++) put_byte((x >> (24 - 8 * j)) & 0xff);
        }
        drain();
    }
    // rust-bitcoin Builder::push_int for 0 <= v < 2^32 (script numbers: minimal little-endian
    // sign-magnitude; 0 -> OP_0, 1..16 -> OP_1..OP_16), followed by one opcode
    __device__ __forceinline__ void put_push_int_op(uint32_t v, uint32_t opcode) {
        if (v == 0) {
            put_byte(0x00);
        } else if (v <= 16) {
            put_byte(0x50 + v);
        } else {
            const uint32_t n = tap_scriptnum_len(v);
            put_byte(n);
            for (uint32_t j = 0; j < n; j++) put_byte(j < 4 ? (v >> (8 * j)) & 0xff : 0);
        }
        put_byte(opcode);
        drain();
    }
    // padding + length; total = bytes hashed since the IV (tag block included)
    __device__ __forceinline__ void finish(uint64_t total_bytes) {
        put_byte(0x80);
        if (a) {
            emit_word(acc);
            acc = 0;
            a = 0;
        }
        while ((pending() & 15) != 14) emit_word(0);
        const uint64_t bits = total_bytes * 8;
        emit_word((uint32_t)(bits >> 32));
        emit_word((uint32_t)bits);
        while (pending() != 0) compress_block();
    }
};

__device__ __forceinline__ void put_leaf_header(Stream& s, uint64_t script_len) {
    s.put_byte(0xc0);  // LeafVersion::TapScript (builder.rs:26)
    if (script_len < 0xfd) {
        s.put_byte((uint32_t)script_len);
    } else if (script_len <= 0xffff) {
        s.put_byte(0xfd);
        s.put_byte(script_len & 0xff);
        s.put_byte((script_len >> 8) & 0xff);
    } else {
        s.put_byte(0xfe);
        for (int j = 0; j < 4; j++) s.put_byte((script_len >> (8 * j)) & 0xff);
    }
}

__device__ __forceinline__ void store_digest(uint32_t* out, const uint32_t h[8]) {
    uint4* o = reinterpret_cast<uint4*>(out);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// leaf i: script bytes [byte_off[i], byte_off[i+1]) of the blob; script i starts at word
// word_off[i] of `words` (each script packed from a 4-byte boundary as big-endian words)
__global__ void __launch_bounds__(TPB)
k_tapleaf_blob(const uint32_t* __restrict__ words, const uint64_t* __restrict__ word_off,
               const uint64_t* __restrict__ byte_len, uint64_t n_leaves, TapMid mid,
               uint32_t* __restrict__ digests) {
    __shared__ uint32_t lds[RING * TPB];
    const uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n_leaves) return;
    Stream s;
    s.init(mid.leaf, lds + threadIdx.x);
    const uint64_t len = byte_len[i];
    put_leaf_header(s, len);
    s.put_segment(words + word_off[i], (uint32_t)len);
    const uint64_t hdr = 1 + (len < 0xfd ? 1 : len <= 0xffff ? 3 : 5);
    s.finish(64 + hdr + len);
    store_digest(digests + 8 * i, s.h);
}

// The script length is hashed first (compact size in the leaf header), so the hash state after
// header + LOCK[q][0] depends on (q, length) only -- and a tree's leaves have few distinct lengths
// (they differ by the sizes of the pushes).  k_tap_prefix tabulates that state for every length a
// leaf of tree q can have; k_tapleaf_template then starts from the table instead of hashing the
// first lock script again for each of the N leaves: one of the 1 + n_evals lock scripts less per
// leaf (a third of the work for the FRI matrices and for a two-column trace).
__global__ void __launch_bounds__(TPB)
k_tap_prefix(TapTemplate t, TapMid mid, uint32_t* __restrict__ table) {
    __shared__ uint32_t lds[RING * TPB];
    const uint32_t d = blockIdx.x * TPB + threadIdx.x;
    if (d >= t.n_len) return;
    const uint32_t q = blockIdx.y;
    const uint32_t n_seg = 1 + t.n_evals;
    const uint64_t len = t.const_len[q] + 2 * (uint64_t)(t.n_evals * t.u32_size + 1) + 1 + d;
    Stream s;
    s.init(mid.leaf, lds + threadIdx.x);
    put_leaf_header(s, len);
    s.put_segment(t.seg_words + t.seg_word_off[(uint64_t)q * n_seg], t.seg_len[(uint64_t)q * n_seg]);
    uint32_t* e = table + ((uint64_t)q * t.n_len + d) * TAP_PREFIX_WORDS;
#pragma unroll
    for (int k = 0; k < 8; k++) e[k] = s.h[k];
    const uint32_t pend = s.pending();  // < 16 after a segment
    for (uint32_t k = 0; k < 16; k++) e[8 + k] = k < pend ? s.ring[((s.rpos + k) & (RING - 1)) * TPB] : 0u;
    e[24] = pend;
    e[25] = s.acc;
    e[26] = s.a;
}

// MMCS leaf `idx` of tree q = blockIdx.y (tcs/mod.rs:197-225):
//   LOCK[q][0] push(idx) OP_EQUALVERIFY
//   for every evaluation j: LOCK[q][1+j] { push(limb) OP_EQUALVERIFY } for limbs U-1 .. 0
//   OP_1
// evaluation j = columns [j U, (j+1) U) of the padded row (tcs/mod.rs:339-378): column c of the
// padded row is cols[c][idx >> shift[c]]
__global__ void __launch_bounds__(TPB)
k_tapleaf_template(TapTemplate t, uint64_t n_leaves, TapMid mid, uint32_t* __restrict__ digests) {
    __shared__ uint32_t lds[RING * TPB];
    const uint64_t idx = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= n_leaves) return;
    const uint32_t q = blockIdx.y;
    const uint32_t n_seg = 1 + t.n_evals;
    const uint64_t* seg_off = t.seg_word_off + (uint64_t)q * n_seg;
    const uint32_t* seg_len = t.seg_len + (uint64_t)q * n_seg;
    // pass 1: the script length (it is hashed first, as a compact size)
    uint64_t pushes = tap_push_int_len((uint32_t)idx);
    for (uint32_t c = 0; c < t.n_evals * t.u32_size; c++)
        pushes += tap_push_int_len(t.cols[c][(idx >> t.shift[c]) * t.elem_stride]);
    const uint32_t n_push = t.n_evals * t.u32_size + 1;
    const uint64_t len = t.const_len[q] + pushes + n_push + 1;  // + one OP_EQUALVERIFY per push + OP_1
    Stream s;
    if (t.prefix != nullptr) {
        const uint32_t* e = t.prefix + ((uint64_t)q * t.n_len + (pushes - n_push)) * TAP_PREFIX_WORDS;
        s.init(e, lds + threadIdx.x);
        s.wpos = e[24];
        s.acc = e[25];
        s.a = e[26];
        for (uint32_t k = 0; k < s.wpos; k++) s.ring[k * TPB] = e[8 + k];
    } else {
        s.init(mid.leaf, lds + threadIdx.x);
        put_leaf_header(s, len);
        s.put_segment(t.seg_words + seg_off[0], seg_len[0]);
    }
    s.put_push_int_op((uint32_t)idx, 0x88);  // OP_EQUALVERIFY
    for (uint32_t j = 0; j < t.n_evals; j++) {
        s.put_segment(t.seg_words + seg_off[1 + j], seg_len[1 + j]);
        for (uint32_t l = t.u32_size; l-- > 0;) {
            const uint32_t c = j * t.u32_size + l;
            s.put_push_int_op(t.cols[c][(idx >> t.shift[c]) * t.elem_stride], 0x88);
        }
    }
    s.put_byte(0x51);  // OP_1
    const uint64_t hdr = 1 + (len < 0xfd ? 1 : len <= 0xffff ? 3 : 5);
    s.finish(64 + hdr + len);
    store_digest(digests + ((uint64_t)q * t.tree_stride + idx) * 8, s.h);
}

// parents[q][i] = TapBranch(children[q][2i], children[q][2i+1]), q = blockIdx.y
__global__ void __launch_bounds__(256)
k_tapbranch_level(const uint32_t* __restrict__ trees, uint64_t tree_stride, uint64_t child_off,
                  uint64_t n_parents, TapMid mid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents) return;
    uint32_t* tree = const_cast<uint32_t*>(trees) + (uint64_t)blockIdx.y * tree_stride * 8;
    const uint4* ch = reinterpret_cast<const uint4*>(tree + 8 * (child_off + 2 * i));
    const uint4 a0 = ch[0], a1 = ch[1], b0 = ch[2], b1 = ch[3];
    const uint32_t a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const uint32_t b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    uint32_t out[8];
    sha::tapbranch(mid.branch, a, b, out);
    store_digest(tree + 8 * (child_off + 2 * n_parents + i), out);
}

// out[k][l] = sibling of leaf index[k] at level l in tree tree_of[k]
__global__ void k_tap_gather_paths(const uint32_t* __restrict__ trees, uint64_t tree_stride,
                                   unsigned log_leaves, const uint32_t* __restrict__ tree_of,
                                   const uint64_t* __restrict__ index, uint32_t n,
                                   uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * log_leaves * 8) return;
    const uint32_t word = t & 7, l = (t >> 3) % log_leaves, k = (t >> 3) / log_leaves;
    uint64_t off = 0;
    for (unsigned j = 0; j < l; j++) off += (uint64_t)1 << (log_leaves - j);
    const uint64_t node = (index[k] >> l) ^ 1;
    out[t] = trees[((uint64_t)tree_of[k] * tree_stride + off + node) * 8 + word];
}

}  // namespace

void launch_tapleaf_blob(Context& ctx, const uint32_t* words, const uint64_t* word_off,
                         const uint64_t* byte_len, uint64_t n_leaves, const TapMid& mid,
                         uint32_t* digests) {
    TS_LAUNCH(ctx, k_tapleaf_blob, dim3((unsigned)((n_leaves + TPB - 1) / TPB)), dim3(TPB), 0, words,
              word_off, byte_len, n_leaves, mid, digests);
    TS_HIP(hipGetLastError());
}

void launch_tapleaf_template(Context& ctx, const TapTemplate& t, uint64_t n_leaves, uint32_t n_trees,
                             const TapMid& mid, uint32_t* digests) {
    TS_REQUIRE(n_trees >= 1 && n_trees <= 65535, TS_ERR_INVALID, "taptree: 1..65535 trees");
    TS_LAUNCH(ctx, k_tapleaf_template, dim3((unsigned)((n_leaves + TPB - 1) / TPB), n_trees), dim3(TPB), 0,
              t, n_leaves, mid, digests);
    TS_HIP(hipGetLastError());
}

void launch_tap_prefix(Context& ctx, const TapTemplate& t, uint32_t n_trees, const TapMid& mid, uint32_t* table) {
    TS_REQUIRE(n_trees >= 1 && n_trees <= 65535 && t.n_len >= 1, TS_ERR_INVALID, "taptree: prefix table shape");
    TS_LAUNCH(ctx, k_tap_prefix, dim3((t.n_len + TPB - 1) / TPB, n_trees), dim3(TPB), 0, t, mid, table);
    TS_HIP(hipGetLastError());
}

void launch_tapbranch_levels(Context& ctx, uint32_t* trees, uint64_t tree_stride, unsigned log_leaves,
                             uint32_t n_trees, const TapMid& mid) {
    uint64_t off = 0;
    for (unsigned l = 0; l < log_leaves; l++) {
        const uint64_t n_children = (uint64_t)1 << (log_leaves - l);
        const uint64_t n_parents = n_children / 2;
        TS_LAUNCH(ctx, k_tapbranch_level, dim3((unsigned)((n_parents + 255) / 256), n_trees), dim3(256), 0,
                  trees, tree_stride, off, n_parents, mid);
        off += n_children;
    }
    TS_HIP(hipGetLastError());
}

void launch_tap_gather_paths(Context& ctx, const uint32_t* trees, uint64_t tree_stride,
                             unsigned log_leaves, const uint32_t* tree_of, const uint64_t* index,
                             uint32_t n, uint32_t* out) {
    if (n == 0 || log_leaves == 0) return;
    const uint32_t total = n * log_leaves * 8;
    TS_LAUNCH(ctx, k_tap_gather_paths, dim3((total + 255) / 256), dim3(256), 0, trees, tree_stride,
              log_leaves, tree_of, index, n, out);
    TS_HIP(hipGetLastError());
}

}  // namespace ts
