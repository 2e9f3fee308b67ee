// Native verifier (host only): accepts exactly the proofs the
// reference's verifier accepts, for the Blake3-Merkle MMCS of this build.
//   uni-stark/src/verifier.rs:19-161   verify
//   fri/src/two_adic_pcs.rs:421-534    Pcs::verify (reduced openings per query)
//   fri/src/verifier.rs:20-60          verify_shape_and_sample_challenges
//   fri/src/verifier.rs:62-98          verify_challenges
//   fri/src/verifier.rs:100-165        verify_query
//   fri/src/two_adic_pcs.rs:87-114     fold_row
//   uni-stark/src/folder.rs:101-105    VerifierConstraintFolder::assert_zero
// Proofs are read in the TSPF v1 wire format.
#include <string.h>

#include <vector>

#include "blake3.hpp"
#include "host.hpp"

namespace ts {

namespace {

// canonical-domain EF helpers (host scalars only)
Ef c_mul(Ef a, Ef b) { return ef_mul(a, ef_to_mont(b)); }
Ef c_mul_base(Ef a, uint32_t b) { return ef_mul_base(a, to_mont(b)); }
Ef c_inv(Ef a) { return ef_from_mont(ef_inv(ef_to_mont(a))); }
Ef c_pow(Ef a, uint64_t e) { return ef_from_mont(ef_pow(ef_to_mont(a), e)); }
Ef c_one() { return Ef{{1, 0, 0, 0}}; }
Ef c_sub_base(Ef a, uint32_t b) { a.c[0] = sub(a.c[0], b); return a; }
Ef c_add_base(Ef a, uint32_t b) { a.c[0] = add(a.c[0], b); return a; }

struct Reader {
    const uint32_t* w;
    size_t len, pos = 0;
    bool bad = false;
    uint32_t get() {
        if (pos >= len) { bad = true; return 0; }
        return w[pos++];
    }
    const uint32_t* take(size_t n) {
        if (n > len - pos || pos > len) { bad = true; return nullptr; }
        const uint32_t* p = w + pos;
        pos += n;
        return p;
    }
};

void hash_words(const uint32_t* w, size_t n, uint32_t out[8]) {
    // Blake3 over n words, any length (multi-chunk rows included)
    b3::hash_stream([w](uint64_t i) { return w[i]; }, n, out);
}
void compress2(const uint32_t* l, const uint32_t* r, uint32_t out[8]) {
    uint32_t m[16];
    memcpy(m, l, 32);
    memcpy(m + 8, r, 32);
    b3::hash64(m, out);
}

// BFMmcs::verify_batch for the Blake3 Merkle MMCS (mixed heights: rows of a shorter matrix are
// injected at the layer of its height).  rows = concatenated opened rows in matrix order.
bool mmcs_verify(const std::vector<uint64_t>& heights, const std::vector<uint32_t>& widths,
                 uint64_t index, const uint32_t* rows, const uint32_t* path, size_t path_len,
                 const uint32_t root[8]) {
    uint64_t max_h = 0;
    for (size_t i = 0; i < heights.size(); i++) max_h = std::max(max_h, heights[i]);
    unsigned log_max = 0;
    while ((1ull << log_max) < max_h) log_max++;
    if (path_len != log_max || (index >> log_max) != 0) return false;
    auto hash_height = [&](uint64_t h, uint32_t out[8]) -> bool {
        std::vector<uint32_t> buf;
        size_t off = 0;
        for (size_t i = 0; i < heights.size(); i++) {
            if (heights[i] == h) buf.insert(buf.end(), rows + off, rows + off + widths[i]);
            off += widths[i];
        }
        if (buf.empty()) return false;
        hash_words(buf.data(), buf.size(), out);
        return true;
    };
    uint32_t cur[8];
    if (!hash_height(max_h, cur)) return false;
    uint64_t idx = index;
    for (unsigned l = 0; l < log_max; l++) {
        uint32_t nxt[8];
        if (idx & 1) compress2(path + 8 * l, cur, nxt);
        else compress2(cur, path + 8 * l, nxt);
        idx >>= 1;
        uint32_t inj[8];
        if (hash_height(max_h >> (l + 1), inj)) compress2(nxt, inj, cur);
        else memcpy(cur, nxt, 32);
    }
    return memcmp(cur, root, 32) == 0;
}

// two_adic_pcs.rs:87-114
Ef fold_row(uint64_t index, unsigned log_height, Ef beta, Ef e0, Ef e1) {
    const uint32_t s = pow_canon(two_adic_generator(log_height + 1), bitrev32((uint32_t)index, log_height));
    const uint32_t x0 = s, x1 = neg(s);
    Ef num = c_mul(c_sub_base(beta, x0), ef_sub(e1, e0));
    return ef_add(e0, c_mul_base(num, inv_canon(sub(x1, x0))));
}

// tape evaluation over EF4 (verifier side: opened values are extension elements)
void eval_tape_ext(const AirProgram& air, const Ef* local, const Ef* next,
                   const std::vector<uint32_t>& pis, Ef is_first, Ef is_last, Ef is_trans,
                   std::vector<Ef>& v) {
    const uint32_t* tape = air.tape.data();
    const uint32_t n_nodes = tape[4];
    const uint32_t* nodes = tape + 6;
    v.resize(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        const uint32_t op = nodes[3 * i], a = nodes[3 * i + 1], b = nodes[3 * i + 2];
        switch (op) {
            case T_CONST: v[i] = ef_from_base(a); break;
            case T_MAIN: v[i] = a ? next[b] : local[b]; break;
            case T_PUBLIC: v[i] = ef_from_base(pis[a]); break;
            case T_IS_FIRST: v[i] = is_first; break;
            case T_IS_LAST: v[i] = is_last; break;
            case T_IS_TRANSITION: v[i] = is_trans; break;
            case T_ADD: v[i] = ef_add(v[a], v[b]); break;
            case T_SUB: v[i] = ef_sub(v[a], v[b]); break;
            case T_NEG: v[i] = ef_neg(v[a]); break;
            default: v[i] = c_mul(v[a], v[b]); break;
        }
    }
}

}  // namespace

// Pcs::verify, fri/src/two_adic_pcs.rs:421-534, with verify_shape_and_sample_challenges and
// verify_challenges / verify_query of fri/src/verifier.rs:20-165, for any rounds x matrices x points.
// `words` = the FriProof (TSPF v1 order, from the commit-phase round count on).  Returns 0 or the
// error code of include/tapstark.h ts_verify.
// `tap` != nullptr: every commitment is num_queries taptree roots (8 words each, `root` points at
// the first) and the MMCS checks are TapTreeMmcs::verify_batch in tree q (taptree_mmcs.rs:77-99)
static int fri_verify_impl(const FriConfig& fri, BfChallenger& challenger,
                           const std::vector<PcsRoundClaim>& rounds, bool pass_through,
                           const uint32_t* words, size_t n_words, const TapLocks* tap = nullptr);

int pcs_verify(const FriConfig& fri, BfChallenger& challenger, const std::vector<PcsRoundClaim>& rounds,
               const uint32_t* words, size_t n_words) {
    return fri_verify_impl(fri, challenger, rounds, false, words, n_words);
}
int fri_verify_pass_through(const FriConfig& fri, BfChallenger& challenger, const uint32_t* words,
                            size_t n_words) {
    return fri_verify_impl(fri, challenger, {}, true, words, n_words);
}

static int fri_verify_impl(const FriConfig& fri, BfChallenger& challenger,
                           const std::vector<PcsRoundClaim>& rounds, bool pass_through,
                           const uint32_t* words, size_t n_words, const TapLocks* tap) {
    const size_t n_roots = tap ? fri.num_queries : 1, cw = 8 * n_roots;
    Reader rb{words, n_words};
    unsigned log_global_max_height = 0;  // two_adic_pcs.rs:447-455
    for (auto& r : rounds)
        for (auto& m : r.mats) {
            if (m.log_height > 27 || m.log_height < fri.log_blowup || m.points.size() != m.values.size())
                return 1;
            for (auto& v : m.values)
                if (v.size() != m.width) return 1;
            log_global_max_height = std::max(log_global_max_height, m.log_height);
        }
    const Ef batch_alpha = pass_through ? c_one() : challenger.sample();  // :443
    const uint32_t R = rb.get();
    if (rb.bad || R > 31) return 9;
    const uint32_t* commits = rb.take(cw * (size_t)R);
    if (rb.bad) return 9;
    std::vector<Ef> betas(R);
    for (uint32_t r = 0; r < R; r++) {  // fri/src/verifier.rs:32-39
        for (size_t k = 0; k < n_roots; k++) challenger.observe_commitment(commits + cw * r + 8 * k);
        betas[r] = challenger.sample();
    }
    // taptree: first lock script of every commitment (commit order: input rounds, then FRI rounds)
    std::vector<size_t> lock_base(rounds.size());
    size_t fri_lock_base = 0;
    for (size_t r = 0; r < rounds.size(); r++) {
        size_t tw = 0;
        for (auto& m : rounds[r].mats) tw += m.width;
        lock_base[r] = fri_lock_base;
        fri_lock_base += n_roots * (1 + tw);
    }
    if (tap && tap->n_scripts < fri_lock_base + (size_t)R * n_roots * 3) return 1;
    const uint32_t Q = rb.get();
    if (rb.bad) return 9;
    if (Q != fri.num_queries) return 2;  // :39-41 InvalidProofShape
    const unsigned log_max_height = R + fri.log_blowup;
    // BabyBear's two-adicity is 27: a taller domain does not exist, and `have`/`ro` below,
    // sample_bits and two_adic_generator all assume it (both paths; R comes from untrusted input).
    if (log_max_height > 27) return pass_through ? 9 : 1;
    if (!pass_through && log_max_height != log_global_max_height) return 1;
    if (pass_through) log_global_max_height = log_max_height;
    // the PoW witness follows the queries in the buffer: locate it with a dry parse
    const size_t save = rb.pos;
    for (uint32_t q = 0; q < Q && !rb.bad; q++) {
        const uint32_t nb = rb.get();
        for (uint32_t k = 0; k < nb && !rb.bad; k++) {
            if (pass_through) {  // (log_height, value)
                rb.take(5);
                continue;
            }
            const uint32_t nm = rb.get();
            for (uint32_t i = 0; i < nm && !rb.bad; i++) rb.take(rb.get());
            rb.take(8 * (size_t)rb.get());
        }
        for (uint32_t r = 0; r < R && !rb.bad; r++) {
            rb.take(8);
            rb.take(8 * (size_t)rb.get());
        }
    }
    const uint32_t* tail = rb.take(5);
    if (rb.bad) return 9;
    if (rb.pos != rb.len) return 9;
    for (int k = 0; k < 4; k++)
        if (tail[k] >= P) return 9;
    const Ef final_poly = Ef{{tail[0], tail[1], tail[2], tail[3]}};
    const uint32_t pow_witness = tail[4];
    rb.pos = save;
    if (!challenger.check_witness(fri.proof_of_work_bits, pow_witness)) return 3;  // :44-46
    std::vector<uint64_t> indices(Q);
    for (uint32_t q = 0; q < Q; q++) indices[q] = challenger.sample_bits(log_max_height);  // :50-52

    for (uint32_t q = 0; q < Q; q++) {  // verify_challenges, fri/src/verifier.rs:62-98
        const uint64_t index = indices[q];
        // ---- open_input, two_adic_pcs.rs:457-528: the reduced opening of every height
        Ef ro[32], alpha_pow[32];
        bool have[32] = {false};
        for (int i = 0; i < 32; i++) {
            ro[i] = ef_zero();
            alpha_pow[i] = c_one();
        }
        if (pass_through) {  // fri.rs:126-140: the input proof IS the reduced openings
            const uint32_t n_in = rb.get();
            if (rb.bad || n_in > 32) return 9;
            unsigned prev = 32;
            for (uint32_t k = 0; k < n_in; k++) {
                const uint32_t* e = rb.take(5);
                if (rb.bad) return 9;
                const unsigned lh = e[0];
                if (lh >= prev || lh > log_max_height) return 1;  // sorted by descending height
                prev = lh;
                for (int j = 1; j < 5; j++)
                    if (e[j] >= P) return 9;
                ro[lh] = Ef{{e[1], e[2], e[3], e[4]}};
                have[lh] = true;
            }
        } else if (rb.get() != rounds.size()) {
            return 1;  // one BatchOpening per commit round
        }
        for (auto& round : rounds) {
            const uint32_t nm = rb.get();
            if (rb.bad || nm != round.mats.size()) return 1;
            std::vector<uint32_t> rows;
            std::vector<uint64_t> heights;
            std::vector<uint32_t> widths;
            unsigned log_batch_max = 0;
            for (uint32_t i = 0; i < nm; i++) {
                const uint32_t wd = rb.get();
                if (rb.bad || wd != round.mats[i].width) return 1;
                const uint32_t* vals = rb.take(wd);
                if (rb.bad) return 9;
                for (uint32_t c = 0; c < wd; c++)
                    if (vals[c] >= P) return 9;
                rows.insert(rows.end(), vals, vals + wd);
                heights.push_back(1ull << round.mats[i].log_height);
                widths.push_back(wd);
                log_batch_max = std::max(log_batch_max, round.mats[i].log_height);
            }
            const uint32_t plen = rb.get();
            const uint32_t* path = rb.take(8 * (size_t)plen);
            if (rb.bad) return 9;
            // :470-486 reduced_index = index >> (log_global_max_height - log_batch_max_height)
            const uint64_t reduced_index = index >> (log_global_max_height - log_batch_max);
            if (tap) {
                // taptree_mmcs.rs:68-72: the opened rows must come tallest matrix first
                for (size_t i = 1; i < heights.size(); i++)
                    if (heights[i] > heights[i - 1]) return 4;
                const size_t ri = (size_t)(&round - rounds.data());
                const uint32_t n_evals = (uint32_t)rows.size();
                if (plen != log_batch_max ||
                    !tap_verify_words(*tap, lock_base[ri] + (size_t)q * (1 + n_evals), n_evals, 1, reduced_index,
                                      rows.data(), path, plen, round.root + 8 * (size_t)q))
                    return 4;
            } else if (!mmcs_verify(heights, widths, reduced_index, rows.data(), path, plen, round.root)) {
                return 4;  // InputError
            }
            size_t off = 0;
            for (uint32_t i = 0; i < nm; i++) {  // :490-523
                const PcsMatClaim& m = round.mats[i];
                const unsigned lh = m.log_height;
                const uint64_t rev = bitrev32((uint32_t)(index >> (log_global_max_height - lh)), lh);
                const uint32_t x = mul(GENERATOR, pow_canon(two_adic_generator(lh), rev));
                have[lh] = true;
                for (size_t p = 0; p < m.points.size(); p++) {
                    Ef acc = ef_zero();
                    for (uint32_t c = 0; c < m.width; c++) {
                        Ef diff = c_add_base(ef_neg(m.values[p][c]), rows[off + c]);
                        acc = ef_add(acc, c_mul(alpha_pow[lh], diff));
                        alpha_pow[lh] = c_mul(alpha_pow[lh], batch_alpha);
                    }
                    const Ef den = c_add_base(ef_neg(m.points[p]), x);
                    ro[lh] = ef_add(ro[lh], c_mul(acc, c_inv(den)));
                }
                off += m.width;
            }
        }
        // ---- verify_query, fri/src/verifier.rs:100-165
        Ef folded_eval = ef_zero();
        uint64_t query_index = index;
        for (uint32_t r = 0; r < R; r++) {
            const unsigned log_folded_height = log_max_height - 1 - r;
            const uint64_t point_index = query_index & 1;
            const uint64_t index_pair = query_index >> 1;
            if (have[log_folded_height + 1]) folded_eval = ef_add(folded_eval, ro[log_folded_height + 1]);  // :127-130
            const uint32_t* vals = rb.take(8);
            const uint32_t plen = rb.get();
            const uint32_t* path = rb.take(8 * (size_t)plen);
            if (rb.bad) return 9;
            for (int k = 0; k < 8; k++)
                if (vals[k] >= P) return 9;
            const Ef e0 = Ef{{vals[0], vals[1], vals[2], vals[3]}}, e1 = Ef{{vals[4], vals[5], vals[6], vals[7]}};
            const Ef committed = point_index ? e1 : e0;
            // :139-141 asserts this from the second round on; checking the first round as well is a
            // strict superset (the reduced opening must be what was committed)
            if (!ef_eq(folded_eval, committed)) return 8;
            std::vector<uint64_t> hh{1ull << log_folded_height};
            std::vector<uint32_t> ww{8};
            if (tap) {
                if (plen != log_folded_height ||
                    !tap_verify_words(*tap, fri_lock_base + ((size_t)r * n_roots + q) * 3, 2, 4, index_pair, vals,
                                      path, plen, commits + cw * r + 8 * (size_t)q))
                    return 5;
            } else if (!mmcs_verify(hh, ww, index_pair, vals, path, plen, commits + 8 * r)) {
                return 5;  // :143-146
            }
            query_index = index_pair;
            folded_eval = fold_row(query_index, log_folded_height, betas[r], e0, e1);  // :149-154
        }
        if (!ef_eq(folded_eval, final_poly)) return 6;  // :92-94 FinalPolyMismatch
    }
    return 0;
}

// 0 = accept; otherwise the reference's error (see include/tapstark.h ts_verify)
static int verify_impl(const FriConfig& fri, const AirProgram& air, BfChallenger& challenger,
                       const uint32_t* proof, size_t n_words, const std::vector<uint32_t>& pis,
                       const TapLocks* tap);

int verify(const FriConfig& fri, const AirProgram& air, BfChallenger& challenger,
           const uint32_t* proof, size_t n_words, const std::vector<uint32_t>& pis) {
    return verify_impl(fri, air, challenger, proof, n_words, pis, nullptr);
}
int verify_tap(const FriConfig& fri, const AirProgram& air, BfChallenger& challenger,
               const uint32_t* proof, size_t n_words, const std::vector<uint32_t>& pis,
               const TapLocks& locks) {
    return verify_impl(fri, air, challenger, proof, n_words, pis, &locks);
}

static int verify_impl(const FriConfig& fri, const AirProgram& air, BfChallenger& challenger,
                       const uint32_t* proof, size_t n_words, const std::vector<uint32_t>& pis,
                       const TapLocks* tap) {
    if (pis.size() != air.n_public) return 1;
    Reader rb{proof, n_words};
    if (rb.get() != 0x46505354u || rb.get() != (tap ? 2u : 1u)) return 9;
    const unsigned degree_bits = rb.get();
    const uint32_t pw = rb.get(), pqd = rb.get();
    if (tap && rb.get() != fri.num_queries) return 1;  // TSPF v2: roots per commitment
    if (rb.bad || degree_bits > 27) return 9;
    const size_t n_roots = tap ? fri.num_queries : 1, cw = 8 * n_roots;
    const unsigned lqd = air.log_quotient_degree;
    const uint32_t qd = 1u << lqd, w = air.width;
    // verifier.rs:49-59 valid_shape
    if (pw != w || pqd != qd) return 1;
    const uint32_t* trace_root = rb.take(cw);
    const uint32_t* quot_root = rb.take(cw);
    const uint32_t* trace_local = rb.take(4 * (size_t)w);
    const uint32_t* trace_next = rb.take(4 * (size_t)w);
    const uint32_t* qchunks = rb.take(16 * (size_t)qd);
    if (rb.bad) return 9;
    // (the proof is a word stream: Ef is 16-byte aligned, the words are not)
    std::vector<Ef> tl(w), tn(w), qc(4 * (size_t)qd);
    memcpy((void*)tl.data(), trace_local, 16 * (size_t)w);
    memcpy((void*)tn.data(), trace_next, 16 * (size_t)w);
    memcpy((void*)qc.data(), qchunks, 64 * (size_t)qd);
    for (auto* vec : {&tl, &tn, &qc})
        for (auto& e : *vec)
            for (int k = 0; k < 4; k++)
                if (e.c[k] >= P) return 9;

    // verifier.rs:69-75
    for (size_t k = 0; k < n_roots; k++) challenger.observe_commitment(trace_root + 8 * k);
    const Ef alpha = challenger.sample();
    for (size_t k = 0; k < n_roots; k++) challenger.observe_commitment(quot_root + 8 * k);
    const Ef zeta = challenger.sample();
    const uint32_t gn = two_adic_generator(degree_bits);
    const Ef zeta_next = c_mul_base(zeta, gn);

    // ---- pcs.verify, two_adic_pcs.rs:421-534 (verifier.rs:77-101 builds these claims)
    PcsRoundClaim r0, r1;
    r0.root = trace_root;
    r0.mats.push_back(PcsMatClaim{degree_bits + fri.log_blowup, w, {zeta, zeta_next}, {tl, tn}});
    r1.root = quot_root;
    for (uint32_t c = 0; c < qd; c++) {
        std::vector<Ef> vals(qc.begin() + 4 * (size_t)c, qc.begin() + 4 * (size_t)c + 4);
        r1.mats.push_back(PcsMatClaim{degree_bits + fri.log_blowup, 4, {zeta}, {vals}});
    }
    Reader fr{proof + rb.pos, n_words - rb.pos};
    const int rc = fri_verify_impl(fri, challenger, {r0, r1}, false, fr.w, fr.len, tap);
    if (rc) return rc;

    // ---- verifier.rs:103-132 quotient recombination
    std::vector<uint32_t> shifts(qd);
    const uint32_t gq = two_adic_generator(degree_bits + lqd);
    for (uint32_t c = 0; c < qd; c++) shifts[c] = mul(GENERATOR, pow_canon(gq, c));
    Ef quotient = ef_zero();
    for (uint32_t i = 0; i < qd; i++) {
        Ef zp = c_one();
        for (uint32_t j = 0; j < qd; j++) {
            if (j == i) continue;
            const uint32_t sj_inv = inv_canon(shifts[j]);
            // zp_at_point(z) = (z/shift)^(2^log_n) - 1
            Ef a = c_sub_base(c_pow(c_mul_base(zeta, sj_inv), 1ull << degree_bits), 1);
            const uint32_t bden = sub(pow_canon(mul(shifts[i], sj_inv), 1ull << degree_bits), 1);
            zp = c_mul(zp, c_mul_base(a, inv_canon(bden)));
        }
        for (int e = 0; e < 4; e++) {
            Ef mono = ef_zero();
            mono.c[e] = 1;
            quotient = ef_add(quotient, c_mul(c_mul(zp, mono), qc[4 * (size_t)i + e]));
        }
    }
    // :136 selectors_at_point (trace domain shift 1)
    const Ef zh = c_sub_base(c_pow(zeta, 1ull << degree_bits), 1);
    const uint32_t gn_inv = inv_canon(gn);
    const Ef is_first = c_mul(zh, c_inv(c_sub_base(zeta, 1)));
    const Ef is_last = c_mul(zh, c_inv(c_sub_base(zeta, gn_inv)));
    const Ef is_trans = c_sub_base(zeta, gn_inv);
    const Ef inv_zeroifier = c_inv(zh);
    // :138-153 fold the constraints at zeta
    std::vector<Ef> v;
    eval_tape_ext(air, tl.data(), tn.data(), pis, is_first, is_last, is_trans, v);
    const uint32_t* tape = air.tape.data();
    const uint32_t* cons = tape + 6 + 3 * (size_t)tape[4];
    Ef acc = ef_zero();
    for (uint32_t c = 0; c < air.n_constraints; c++) acc = ef_add(c_mul(acc, alpha), v[cons[c]]);
    if (!ef_eq(c_mul(acc, inv_zeroifier), quotient)) return 7;  // :157 OodEvaluationMismatch
    return 0;
}

}  // namespace ts
