// prove() over the reference's own MMCS: `TapTreeMmcs` (basic/src/mmcs/taptree_mmcs.rs:24-119) as the
// input MMCS of `TwoAdicFriPcs` and as the FRI MMCS, the configuration of uni-stark/tests/fib_air.rs:
// 117-131.  Same numeric pipeline as prover.cpp (LDE, quotient, opened values, reduce, fold); what
// changes is the commitment:
//   - a commitment is num_queries taptrees (tcs/mod.rs:284-292), built on the device from the LDE
//     columns and the caller's lock scripts (taptree.hip k_tapleaf_template);
//   - `challenger.observe(commit)` observes all num_queries roots (challenger/mod.rs:211-223);
//   - query q opens every commitment in tree q (fri/src/prover.rs:50-56, two_adic_pcs.rs:399-414:
//     open_batch(query_times_index = q, ...)).
// The transcript runs on the host here (one root download per commitment): hashing a level of
// kilobyte-sized script leaves dwarfs the round trip.  Proof: TSPF v2 = v1 with a sixth header word
// (num_queries) and num_queries x 8 words per commitment; a digest is 8 words = its 32 bytes read
// little-endian (chan_field.rs:87-95 u256_to_u32).
#include <string.h>

#include <algorithm>

#include "fri_internal.hpp"
#include "taptree.hpp"

namespace ts {

namespace {

struct TapCommit {
    DevBuf<uint32_t> trees;        // [Q][2N-1][8] state words
    std::vector<uint32_t> roots;   // Q x 8 words (bytes read little-endian)
    unsigned log_height = 0;
};

void observe_roots(BfChallenger& ch, const std::vector<uint32_t>& roots) {
    for (size_t q = 0; q < roots.size() / 8; q++) ch.observe_commitment(&roots[8 * q]);
}

TapCommit commit_columns(Context& ctx, const PcsData& data, uint32_t Q, const TapLocks& locks, size_t& cursor) {
    std::vector<const uint32_t*> cols;
    std::vector<uint8_t> shifts;
    for (auto& cm : data.ldes) {
        TS_REQUIRE(cm.height == (1ull << data.log_height), TS_ERR_UNSUPPORTED,
                   "prove over taptrees: matrices of one height per commitment");
        for (uint32_t c = 0; c < cm.width; c++) {
            cols.push_back(cm.d + (uint64_t)c * cm.col_stride);
            shifts.push_back(0);
        }
    }
    TapCommit tc;
    tc.log_height = data.log_height;
    tc.trees = tap_build_trees(ctx, cols, shifts, 1, data.log_height, 1, Q, locks, cursor, tc.roots);
    cursor += (size_t)Q * (1 + cols.size());
    return tc;
}

}  // namespace

std::vector<uint32_t> prove_tap(TwoAdicFriPcs& pcs, const AirProgram& air, BfChallenger& challenger,
                                DeviceMatrix trace, const std::vector<uint32_t>& public_values,
                                const TapLocks& locks) {
    Context& ctx = pcs.ctx();
    const FriConfig& fri = pcs.fri();
    TS_REQUIRE(trace.width == air.width, TS_ERR_INVALID, "prove: trace width != AIR width");
    TS_REQUIRE(public_values.size() == air.n_public, TS_ERR_INVALID, "prove: wrong number of public values");
    TS_REQUIRE(locks.bytes && locks.offsets, TS_ERR_INVALID, "prove over taptrees: no lock-script table");
    const uint32_t w = air.width, Q = fri.num_queries;
    const uint64_t n = trace.height;
    const unsigned log_n = log2_strict(n), lqd = air.log_quotient_degree;
    const uint32_t qd = 1u << lqd;
    const unsigned log_N = log_n + fri.log_blowup;
    const uint64_t N = 1ull << log_N;
    const uint32_t R = log_N - fri.log_blowup;
    TS_REQUIRE(lqd <= fri.log_blowup, TS_ERR_INVARIANT,
               "quotient domain larger than the committed LDE (log_quotient_degree > log_blowup)");
    TS_REQUIRE(locks.n_scripts >= (size_t)Q * ((1 + w) + (1 + 4 * (size_t)qd) + 3 * (size_t)R), TS_ERR_INVALID,
               "prove over taptrees: the lock-script table is shorter than Q ((1+w) + (1+4 qd) + 3 log2(n))");
    size_t cursor = 0;

    // ---- prover.rs:50-63 commit to the trace, alpha
    std::vector<DeviceMatrix> tv;
    tv.push_back(std::move(trace));
    std::unique_ptr<PcsData> trace_data = pcs.commit(tv, {1u}, /*build_tree=*/false);
    TapCommit trace_commit = commit_columns(ctx, *trace_data, Q, locks, cursor);
    observe_roots(challenger, trace_commit.roots);
    const Ef alpha = challenger.sample();

    // ---- :65-84 quotient chunks, their commitment, zeta
    std::vector<DeviceMatrix> chunks = pcs.quotient_chunks(*trace_data, air, public_values, alpha);
    std::vector<uint32_t> qshifts(qd);
    const uint32_t gq = two_adic_generator(log_n + lqd);
    for (uint32_t c = 0; c < qd; c++) qshifts[c] = mul(GENERATOR, pow_canon(gq, c));
    std::unique_ptr<PcsData> quotient_data = pcs.commit(chunks, qshifts, false);
    TapCommit quotient_commit = commit_columns(ctx, *quotient_data, Q, locks, cursor);
    observe_roots(challenger, quotient_commit.roots);
    const Ef zeta = challenger.sample();

    // ---- :94-104 open: opened values + reduced openings (needs the LDEs only)
    const Ef batch_alpha = challenger.sample();
    std::vector<Ef> opened;
    DevBuf<Ef> folded = pcs.open_reduce(*trace_data, *quotient_data, zeta, batch_alpha, opened);

    // ---- bf_commit_phase, fri/src/prover.rs:93-141 (host transcript)
    struct Round {
        DevBuf<Ef> vec;
        TapCommit commit;
    };
    std::vector<Round> rounds;
    uint64_t len = N;
    while (len > fri.blowup()) {
        const uint64_t h = len / 2;
        Round r;
        // RowMajorMatrix::new(folded, 2): row i = (f[2i], f[2i+1]) = 8 consecutive words
        std::vector<const uint32_t*> cols(8);
        for (int c = 0; c < 8; c++) cols[c] = reinterpret_cast<const uint32_t*>(folded.p) + c;
        TapCommit tc;
        tc.log_height = log2_strict(h);
        tc.trees = tap_build_trees(ctx, cols, std::vector<uint8_t>(8, 0), 8, tc.log_height, 4, Q, locks, cursor,
                                   tc.roots);
        cursor += (size_t)Q * 3;
        observe_roots(challenger, tc.roots);       // :114
        const Ef beta = challenger.sample();       // :116
        DevBuf<Ef> out(&ctx, h);
        launch_fri_fold(ctx, folded.p, h, beta, out.p, nullptr);  // :119
        r.vec = std::move(folded);
        r.commit = std::move(tc);
        rounds.push_back(std::move(r));
        folded = std::move(out);
        len = h;
    }
    std::vector<Ef> finals(fri.blowup());
    d2h_sync(ctx, finals.data(), folded.p, finals.size() * sizeof(Ef));
    for (auto& e : finals)  // :130-134
        TS_REQUIRE(memcmp(e.c, finals[0].c, 16) == 0, TS_ERR_INVARIANT, "FRI: final polynomial is not constant");
    const Ef final_poly = finals[0];
    const uint32_t pow_witness = challenger.grind(fri.proof_of_work_bits);  // :43

    // ---- query phase :45-59
    std::vector<uint32_t> indices(Q);
    for (uint32_t q = 0; q < Q; q++) indices[q] = (uint32_t)challenger.sample_bits(log_N);
    std::vector<uint32_t> tree_of(Q);
    for (uint32_t q = 0; q < Q; q++) tree_of[q] = q;
    DevBuf<uint32_t> d_idx(&ctx, Q), d_tree(&ctx, Q);
    h2d(ctx, d_idx.p, indices.data(), Q * 4);
    h2d(ctx, d_tree.p, tree_of.data(), Q * 4);
    const PcsData* in_data[2] = {trace_data.get(), quotient_data.get()};
    const TapCommit* in_commit[2] = {&trace_commit, &quotient_commit};
    std::vector<std::vector<uint32_t>> in_rows(2), in_paths(2);
    for (int k = 0; k < 2; k++) {
        LeafMats lm = in_data[k]->leaf_mats();
        // leaf_mats() points at the Blake3 column table, which this flow never built
        std::vector<const uint32_t*> cols;
        for (auto& cm : in_data[k]->ldes)
            for (uint32_t c = 0; c < cm.width; c++) cols.push_back(cm.d + (uint64_t)c * cm.col_stride);
        DevBuf<const uint32_t*> d_cols(&ctx, cols.size());
        h2d(ctx, d_cols.p, cols.data(), cols.size() * sizeof(const uint32_t*));
        lm.cols = d_cols.p;
        DevBuf<uint32_t> d_rows(&ctx, (size_t)Q * lm.total_width), d_path(&ctx, (size_t)Q * 8 * log_N);
        DevBuf<uint64_t> d_idx64(&ctx, Q);
        std::vector<uint64_t> i64(indices.begin(), indices.end());
        h2d(ctx, d_idx64.p, i64.data(), Q * 8);
        launch_gather_rows(ctx, lm, d_idx.p, Q, 0, d_rows.p);
        launch_tap_gather_paths(ctx, in_commit[k]->trees.p, 2 * N - 1, log_N, d_tree.p, d_idx64.p, Q, d_path.p);
        in_rows[k].resize((size_t)Q * lm.total_width);
        in_paths[k].resize((size_t)Q * 8 * log_N);
        TS_HIP(hipMemcpyAsync(in_rows[k].data(), d_rows.p, in_rows[k].size() * 4, hipMemcpyDeviceToHost, ctx.stream));
        d2h_sync(ctx, in_paths[k].data(), d_path.p, in_paths[k].size() * 4);
    }
    // bf_answer_query :69-90: round i opens row index >> i >> 1 of its h x 2 matrix, in tree q
    std::vector<std::vector<uint32_t>> f_vals(R), f_paths(R);
    for (uint32_t r = 0; r < R; r++) {
        const unsigned ll = rounds[r].commit.log_height;
        std::vector<uint64_t> ri(Q);
        for (uint32_t q = 0; q < Q; q++) ri[q] = indices[q] >> (r + 1);
        std::vector<uint32_t> ri32(ri.begin(), ri.end());
        DevBuf<uint64_t> d_ri(&ctx, Q);
        DevBuf<uint32_t> d_ri32(&ctx, Q), d_vals(&ctx, (size_t)Q * 8), d_path(&ctx, std::max<size_t>((size_t)Q * 8 * ll, 8));
        h2d(ctx, d_ri.p, ri.data(), Q * 8);
        h2d(ctx, d_ri32.p, ri32.data(), Q * 4);
        launch_gather_ef_pairs(ctx, rounds[r].vec.p, d_ri32.p, Q, 0, d_vals.p);
        launch_tap_gather_paths(ctx, rounds[r].commit.trees.p, (2ull << ll) - 1, ll, d_tree.p, d_ri.p, Q, d_path.p);
        f_vals[r].resize((size_t)Q * 8);
        f_paths[r].resize((size_t)Q * 8 * ll);
        TS_HIP(hipMemcpyAsync(f_vals[r].data(), d_vals.p, f_vals[r].size() * 4, hipMemcpyDeviceToHost, ctx.stream));
        if (ll) TS_HIP(hipMemcpyAsync(f_paths[r].data(), d_path.p, f_paths[r].size() * 4, hipMemcpyDeviceToHost, ctx.stream));
        ctx.sync();
    }

    // ---- Proof (uni-stark/src/prover.rs:105-118), TSPF v2
    std::vector<uint32_t> pf;
    auto push = [&](uint32_t v) { pf.push_back(v); };
    auto push_n = [&](const uint32_t* p, size_t k) { pf.insert(pf.end(), p, p + k); };
    auto push_path = [&](const uint32_t* state_words, size_t depth) {  // state words -> bytes read LE
        for (size_t k = 0; k < 8 * depth; k++) pf.push_back(__builtin_bswap32(state_words[k]));
    };
    push(TSPF_MAGIC);
    push(2);
    push(log_n);
    push(w);
    push(qd);
    push(Q);
    push_n(trace_commit.roots.data(), trace_commit.roots.size());
    push_n(quotient_commit.roots.data(), quotient_commit.roots.size());
    for (auto& e : opened) push_n(e.c, 4);
    push(R);
    for (uint32_t r = 0; r < R; r++) push_n(rounds[r].commit.roots.data(), rounds[r].commit.roots.size());
    push(Q);
    for (uint32_t q = 0; q < Q; q++) {
        push(2);  // input_proof: one BatchOpening per commit round (two_adic_pcs.rs:399-414)
        for (int k = 0; k < 2; k++) {
            const auto& ldes = in_data[k]->ldes;
            size_t tw = 0;
            for (auto& cm : ldes) tw += cm.width;
            push((uint32_t)ldes.size());
            size_t c = (size_t)q * tw;
            for (auto& cm : ldes) {
                push(cm.width);
                push_n(&in_rows[k][c], cm.width);
                c += cm.width;
            }
            push(log_N);
            push_path(&in_paths[k][(size_t)q * 8 * log_N], log_N);
        }
        for (uint32_t r = 0; r < R; r++) {  // commit_phase_openings
            const unsigned ll = rounds[r].commit.log_height;
            push_n(&f_vals[r][(size_t)q * 8], 8);
            push(ll);
            push_path(ll ? &f_paths[r][(size_t)q * 8 * ll] : nullptr, ll);
        }
    }
    push_n(final_poly.c, 4);
    push(pow_witness);
    return pf;
}

}  // namespace ts
