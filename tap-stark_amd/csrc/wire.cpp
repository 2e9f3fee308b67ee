// Proof wire format: TSPF v1 words <-> the postcard encoding of the
// reference's serde types, host only.
//
// The reference derives serde on its proof types and names postcard as the carrier
// (uni-stark/Cargo.toml:44; the round trip in uni-stark/tests/mul_air.rs:133-137):
//   Proof        { commitments, opened_values, opening_proof, degree_bits }   uni-stark/src/proof.rs:19-25
//   Commitments  { trace, quotient_chunks }                                    proof.rs:28-31
//   OpenedValues { trace_local, trace_next, quotient_chunks: Vec<Vec<_>> }     proof.rs:34-38
//   FriProof     { commit_phase_commits, query_proofs, final_poly, pow_witness }  fri/src/proof.rs:13-21
//   BfQueryProof { input_proof, commit_phase_openings: Vec<(Vec<Vec<F>>, MmcsProof)> }  fri/src/proof.rs:28-33
//   BatchOpening { opened_values: Vec<Vec<Val>>, opening_proof }                fri/src/two_adic_pcs.rs:63-68
// postcard 1.0: a struct or tuple is its fields in order; Vec<T> = varint(len) then the elements;
// u32/usize = LEB128 varint; u8 = one byte; [T; N] = N elements without a length.  A BabyBear
// element serialises as its canonical u32, an EF4 element as its 4 coefficients.
// Two types are this build's own, because the reference's are taptree-specific:
//   Commitment = Vec<[[u8; 4]; 8]> with ONE root (survey row M)  -> varint(1) + 32 bytes,
//   MMCS proof = Vec<[u8; 32]>, the sibling path, leaf level first -> varint(len) + 32 len bytes
// (the shape of upstream Plonky3's FieldMerkleTreeMmcs proof).  The grinding witness is NOT a field
// element: `type Witness = PF` with PF = [u8; 4] (basic/src/challenger/mod.rs:91,
// chan_field.rs:61), which postcard writes as 4 raw bytes, little-endian word, no range check.
#include <string.h>

#include <vector>

#include "fri_internal.hpp"

namespace ts {

namespace {

struct WordReader {
    const uint32_t* w;
    size_t len, pos = 0;
    bool bad = false;
    uint32_t get() {
        if (pos >= len) { bad = true; return 0; }
        return w[pos++];
    }
    const uint32_t* take(size_t n) {
        if (pos > len || n > len - pos) { bad = true; return nullptr; }
        const uint32_t* p = w + pos;
        pos += n;
        return p;
    }
};

struct ByteWriter {
    std::vector<uint8_t>& b;
    void varint(uint64_t v) {
        while (v >= 0x80) {
            b.push_back((uint8_t)(v | 0x80));
            v >>= 7;
        }
        b.push_back((uint8_t)v);
    }
    bool bad = false;
    void felts(const uint32_t* p, size_t n) {  // n canonical field elements, no length
        for (size_t i = 0; i < n; i++) {
            if (p[i] >= P) bad = true;
            varint(p[i]);
        }
    }
    void raw_u32(uint32_t v) {  // [u8; 4]
        for (int k = 0; k < 4; k++) b.push_back((uint8_t)(v >> (8 * k)));
    }
    void digest(const uint32_t* d) {  // [[u8; 4]; 8]: the words' little-endian bytes
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 4; k++) b.push_back((uint8_t)(d[i] >> (8 * k)));
    }
    void commitment(const uint32_t* d, uint32_t n_roots = 1) {  // Vec<[[u8; 4]; 8]>
        varint(n_roots);
        for (uint32_t k = 0; k < n_roots; k++) digest(d + 8 * (size_t)k);
    }
};

struct ByteReader {
    const uint8_t* b;
    size_t len, pos = 0;
    bool bad = false;
    uint64_t varint() {
        uint64_t v = 0;
        for (unsigned shift = 0; shift < 64; shift += 7) {
            if (pos >= len) { bad = true; return 0; }
            const uint8_t x = b[pos++];
            v |= (uint64_t)(x & 0x7f) << shift;
            if (!(x & 0x80)) return v;
        }
        bad = true;
        return 0;
    }
    uint32_t felt() {
        const uint64_t v = varint();
        if (v >= P) bad = true;  // a canonical BabyBear element
        return (uint32_t)v;
    }
    uint32_t raw_u32() {  // [u8; 4]
        if (pos > len || len - pos < 4) { bad = true; return 0; }
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) w |= (uint32_t)b[pos++] << (8 * k);
        return w;
    }
    // a count of elements that take at least `min_bytes` each cannot exceed what is left
    bool fits(uint64_t count, uint64_t min_bytes) {
        if (bad || pos > len || count > (len - pos) / min_bytes) { bad = true; return false; }
        return true;
    }
    void digest(std::vector<uint32_t>& out) {
        if (pos > len || len - pos < 32) { bad = true; return; }
        for (int i = 0; i < 8; i++) {
            uint32_t w = 0;
            for (int k = 0; k < 4; k++) w |= (uint32_t)b[pos++] << (8 * k);
            out.push_back(w);
        }
    }
    // a commitment with exactly `n_roots` roots (0: read the count and return it)
    uint64_t commitment(std::vector<uint32_t>& out, uint64_t n_roots = 1) {
        const uint64_t n = varint();
        if (n_roots ? n != n_roots : (n == 0 || n > 4096)) {
            bad = true;
            return 0;
        }
        for (uint64_t k = 0; k < n && !bad; k++) digest(out);
        return n;
    }
};

// sanity bound for counts read from untrusted bytes (a proof never has this many of anything)
constexpr uint64_t MAX_COUNT = 1u << 24;

}  // namespace

// TSPF v1 -> postcard.  Returns false if the words are not a well-formed TSPF v1 proof.
bool tspf_to_postcard(const uint32_t* words, size_t n_words, std::vector<uint8_t>& out) {
    WordReader r{words, n_words};
    ByteWriter w{out};
    if (r.get() != TSPF_MAGIC) return false;
    const uint32_t version = r.get();
    if (version != 1 && version != 2) return false;
    const uint32_t degree_bits = r.get(), width = r.get(), qd = r.get();
    // TSPF v2 (proofs over the taptree MMCS): num_queries roots per commitment, as the reference's
    // `Commitment = Vec<TreeRoot>` (basic/src/mmcs/taptree_mmcs.rs:43)
    const uint32_t n_roots = version == 2 ? r.get() : 1;
    if (r.bad || qd > 64 || n_roots == 0 || n_roots > 4096) return false;
    // commitments
    const uint32_t* tc = r.take(8 * (size_t)n_roots);
    const uint32_t* qc = r.take(8 * (size_t)n_roots);
    if (r.bad) return false;
    w.commitment(tc, n_roots);
    w.commitment(qc, n_roots);
    // opened_values
    for (int k = 0; k < 2; k++) {  // trace_local, trace_next
        const uint32_t* v = r.take(4 * (size_t)width);
        if (r.bad) return false;
        w.varint(width);
        w.felts(v, 4 * (size_t)width);
    }
    w.varint(qd);
    for (uint32_t c = 0; c < qd; c++) {
        const uint32_t* v = r.take(16);
        if (r.bad) return false;
        w.varint(4);
        w.felts(v, 16);
    }
    // opening_proof: FriProof
    const uint32_t R = r.get();
    if (r.bad || R > 64) return false;
    w.varint(R);
    for (uint32_t i = 0; i < R; i++) {
        const uint32_t* d = r.take(8 * (size_t)n_roots);
        if (r.bad) return false;
        w.commitment(d, n_roots);
    }
    const uint32_t Q = r.get();
    if (r.bad || Q > MAX_COUNT) return false;
    w.varint(Q);
    for (uint32_t q = 0; q < Q; q++) {
        const uint32_t n_batches = r.get();  // input_proof: Vec<BatchOpening>
        if (r.bad || n_batches > 64) return false;
        w.varint(n_batches);
        for (uint32_t k = 0; k < n_batches; k++) {
            const uint32_t n_mats = r.get();
            if (r.bad || n_mats > 64) return false;
            w.varint(n_mats);
            for (uint32_t m = 0; m < n_mats; m++) {
                const uint32_t mw = r.get();
                const uint32_t* v = r.take(mw);
                if (r.bad) return false;
                w.varint(mw);
                w.felts(v, mw);
            }
            const uint32_t pl = r.get();
            const uint32_t* path = r.take(8 * (size_t)pl);
            if (r.bad || pl > 64) return false;
            w.varint(pl);
            for (uint32_t l = 0; l < pl; l++) w.digest(path + 8 * (size_t)l);
        }
        w.varint(R);  // commit_phase_openings: Vec<(Vec<Vec<F>>, proof)>
        for (uint32_t i = 0; i < R; i++) {
            const uint32_t* v = r.take(8);
            const uint32_t pl = r.get();
            const uint32_t* path = r.take(8 * (size_t)pl);
            if (r.bad || pl > 64) return false;
            w.varint(1);  // one matrix ...
            w.varint(2);  // ... whose opened row has two EF4 elements
            w.felts(v, 8);
            w.varint(pl);
            for (uint32_t l = 0; l < pl; l++) w.digest(path + 8 * (size_t)l);
        }
    }
    const uint32_t* fp = r.take(4);
    const uint32_t pow = r.get();
    if (r.bad || r.pos != n_words) return false;
    w.felts(fp, 4);
    w.raw_u32(pow);
    w.varint(degree_bits);
    return !w.bad;
}

// postcard -> TSPF v1.  Returns false on malformed input (truncated, non-canonical field element,
// a shape TSPF cannot hold, trailing bytes).
// want_version: 0 = infer (one root per commitment -> v1, several -> v2); 1 / 2 = that framing.  The
// postcard bytes do not say which MMCS made them: a proof over taptrees with num_queries = 1 has one
// root per commitment too, and only the caller knows it must come back as v2.
bool postcard_to_tspf(const uint8_t* bytes, size_t n_bytes, std::vector<uint32_t>& out, int want_version) {
    if (want_version < 0 || want_version > 2) return false;
    ByteReader r{bytes, n_bytes};
    std::vector<uint32_t> body;  // everything after the header words
    const uint64_t n_roots = r.commitment(body, 0);  // 1: TSPF v1 (or a v2 proof of one query); more: v2
    if (r.bad) return false;
    if (want_version == 1 && n_roots != 1) return false;
    const bool v2 = want_version == 2 || (want_version == 0 && n_roots != 1);
    r.commitment(body, n_roots);
    uint64_t width = 0;
    for (int k = 0; k < 2; k++) {
        const uint64_t wd = r.varint();
        if (r.bad || wd > MAX_COUNT || (k == 1 && wd != width) || !r.fits(4 * wd, 1)) return false;
        width = wd;
        for (uint64_t i = 0; i < 4 * wd && !r.bad; i++) body.push_back(r.felt());
        if (r.bad) return false;
    }
    const uint64_t qd = r.varint();
    if (r.bad || qd > 64) return false;
    for (uint64_t c = 0; c < qd; c++) {
        if (r.varint() != 4) return false;
        for (int i = 0; i < 16; i++) body.push_back(r.felt());
    }
    const uint64_t R = r.varint();
    if (r.bad || R > 64) return false;
    body.push_back((uint32_t)R);
    for (uint64_t i = 0; i < R && !r.bad; i++) r.commitment(body, n_roots);
    const uint64_t Q = r.varint();
    if (r.bad || Q > MAX_COUNT || !r.fits(Q, 1)) return false;
    body.push_back((uint32_t)Q);
    for (uint64_t q = 0; q < Q && !r.bad; q++) {
        const uint64_t n_batches = r.varint();
        if (r.bad || n_batches > 64) return false;
        body.push_back((uint32_t)n_batches);
        for (uint64_t k = 0; k < n_batches; k++) {
            const uint64_t n_mats = r.varint();
            if (r.bad || n_mats > 64) return false;
            body.push_back((uint32_t)n_mats);
            for (uint64_t m = 0; m < n_mats; m++) {
                const uint64_t mw = r.varint();
                if (r.bad || mw > MAX_COUNT || !r.fits(mw, 1)) return false;
                body.push_back((uint32_t)mw);
                for (uint64_t i = 0; i < mw && !r.bad; i++) body.push_back(r.felt());
                if (r.bad) return false;
            }
            const uint64_t pl = r.varint();
            if (r.bad || pl > 64) return false;
            body.push_back((uint32_t)pl);
            for (uint64_t l = 0; l < pl && !r.bad; l++) r.digest(body);
            if (r.bad) return false;
        }
        if (r.varint() != R) return false;
        for (uint64_t i = 0; i < R; i++) {
            if (r.varint() != 1 || r.varint() != 2) return false;
            for (int j = 0; j < 8; j++) body.push_back(r.felt());
            const uint64_t pl = r.varint();
            if (r.bad || pl > 64) return false;
            body.push_back((uint32_t)pl);
            for (uint64_t l = 0; l < pl && !r.bad; l++) r.digest(body);
            if (r.bad) return false;
        }
    }
    for (int i = 0; i < 4; i++) body.push_back(r.felt());
    const uint32_t pow = r.raw_u32();
    body.push_back(pow);
    const uint64_t degree_bits = r.varint();
    if (r.bad || r.pos != n_bytes || degree_bits > 64) return false;
    out.clear();
    out.reserve(5 + body.size());
    out.push_back(TSPF_MAGIC);
    out.push_back(v2 ? 2 : 1);
    out.push_back((uint32_t)degree_bits);
    out.push_back((uint32_t)width);
    out.push_back((uint32_t)qd);
    if (v2) out.push_back((uint32_t)n_roots);
    out.insert(out.end(), body.begin(), body.end());
    return true;
}

}  // namespace ts
