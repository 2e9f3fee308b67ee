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

// which of the Q trees of every commitment this rank builds: [q0, q0 + cnt), `per` = trees per rank
// (the last ranks may own fewer, or none when there are more ranks than queries)
struct TreeShare {
    uint32_t Q = 0, per = 0, q0 = 0, cnt = 0;
    const Comm* comm = nullptr;
    TreeShare(uint32_t num_queries, const Comm* c) : Q(num_queries), comm(c) {
        const uint32_t G = c ? (uint32_t)c->world : 1u, g = c ? (uint32_t)c->rank : 0u;
        per = (Q + G - 1) / G;
        q0 = std::min(Q, g * per);
        cnt = std::min(Q, q0 + per) - q0;
    }
    bool owns(uint32_t q) const { return q >= q0 && q < q0 + cnt; }
};

struct TapCommit {
    DevBuf<uint32_t> trees;        // [cnt][2N-1][8] state words: the trees this rank owns
    std::vector<uint32_t> roots;   // Q x 8 words (bytes read little-endian), every tree's
    unsigned log_height = 0;
};

void observe_roots(BfChallenger& ch, const std::vector<uint32_t>& roots) {
    for (size_t q = 0; q < roots.size() / 8; q++) ch.observe_commitment(&roots[8 * q]);
}

// the rank's trees over the padded row described by cols / shifts / elem_stride, then the roots of
// all Q trees (exchange: per x 32 bytes from every rank, rank order = tree order)
TapCommit commit_trees(Context& ctx, const TreeShare& sh, const std::vector<const uint32_t*>& cols,
                       const std::vector<uint8_t>& shifts, uint32_t elem_stride, unsigned log_height,
                       uint32_t u32_size, const TapLocks& locks, size_t& cursor) {
    const size_t n_seg = 1 + cols.size() / u32_size;
    TapCommit tc;
    tc.log_height = log_height;
    std::vector<uint32_t> mine;
    if (sh.cnt)
        tc.trees = tap_build_trees(ctx, cols, shifts, elem_stride, log_height, u32_size, sh.cnt, locks,
                                   cursor + (size_t)sh.q0 * n_seg, mine);
    cursor += (size_t)sh.Q * n_seg;
    if (!sh.comm) {
        tc.roots = std::move(mine);
        return tc;
    }
    const size_t seg = (size_t)sh.per * 8, G = (size_t)sh.comm->world;
    mine.resize(seg, 0);
    DevBuf<uint32_t> d_send(&ctx, seg), d_recv(&ctx, seg * G);
    h2d(ctx, d_send.p, mine.data(), seg * 4);
    sh.comm->all_gather(d_send.p, d_recv.p, seg * 4, ctx.stream);
    std::vector<uint32_t> all(seg * G);
    d2h_sync(ctx, all.data(), d_recv.p, all.size() * 4);
    tc.roots.assign(all.begin(), all.begin() + (size_t)sh.Q * 8);  // tree q sits at q: ranges are contiguous
    return tc;
}

TapCommit commit_columns(Context& ctx, const TreeShare& sh, const PcsData& data, const TapLocks& locks,
                         size_t& cursor) {
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
    return commit_trees(ctx, sh, cols, shifts, 1, data.log_height, 1, locks, cursor);
}

}  // namespace

// comm == nullptr: one GPU.  Otherwise every rank holds the WHOLE trace and repeats the numeric
// pipeline (milliseconds), while the commitments -- the seconds: kilobytes of SHA-256 per leaf, times
// num_queries trees -- are split by TREE: rank g builds trees [g per, (g+1) per) of every commitment,
// the roots are all-gathered (32 bytes per tree) so that every transcript observes all of them, and
// query q is answered by the rank that owns tree q.  Trees are independent, so nothing else moves.
std::vector<uint32_t> prove_tap(TwoAdicFriPcs& pcs, const AirProgram& air, BfChallenger& challenger,
                                DeviceMatrix trace, const std::vector<uint32_t>& public_values,
                                const TapLocks& locks, const Comm* comm) {
    Context& ctx = pcs.ctx();
    const FriConfig& fri = pcs.fri();
    TS_REQUIRE(trace.width == air.width, TS_ERR_INVALID, "prove: trace width != AIR width");
    TS_REQUIRE(public_values.size() == air.n_public, TS_ERR_INVALID, "prove: wrong number of public values");
    TS_REQUIRE(locks.bytes && locks.offsets, TS_ERR_INVALID, "prove over taptrees: no lock-script table");
    TS_REQUIRE(!comm || (comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world), TS_ERR_INVALID,
               "prove over taptrees: bad communicator");
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
    const TreeShare sh(Q, comm && comm->world > 1 ? comm : nullptr);
    size_t cursor = 0;

    // ---- prover.rs:50-63 commit to the trace, alpha
    std::vector<DeviceMatrix> tv;
    tv.push_back(std::move(trace));
    std::unique_ptr<PcsData> trace_data = pcs.commit(tv, {1u}, /*build_tree=*/false);
    TapCommit trace_commit = commit_columns(ctx, sh, *trace_data, locks, cursor);
    observe_roots(challenger, trace_commit.roots);
    const Ef alpha = challenger.sample();

    // ---- :65-84 quotient chunks, their commitment, zeta
    std::vector<DeviceMatrix> chunks = pcs.quotient_chunks(*trace_data, air, public_values, alpha);
    std::vector<uint32_t> qshifts(qd);
    const uint32_t gq = two_adic_generator(log_n + lqd);
    for (uint32_t c = 0; c < qd; c++) qshifts[c] = mul(GENERATOR, pow_canon(gq, c));
    std::unique_ptr<PcsData> quotient_data = pcs.commit(chunks, qshifts, false);
    TapCommit quotient_commit = commit_columns(ctx, sh, *quotient_data, locks, cursor);
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
    DevBuf<Ef> d_beta(&ctx, 1);
    while (len > fri.blowup()) {
        const uint64_t h = len / 2;
        Round r;
        // RowMajorMatrix::new(folded, 2): row i = (f[2i], f[2i+1]) = 8 consecutive words
        std::vector<const uint32_t*> cols(8);
        for (int c = 0; c < 8; c++) cols[c] = reinterpret_cast<const uint32_t*>(folded.p) + c;
        r.commit = commit_trees(ctx, sh, cols, std::vector<uint8_t>(8, 0), 8, log2_strict(h), 4, locks, cursor);
        observe_roots(challenger, r.commit.roots);  // :114
        const Ef beta = challenger.sample();        // :116
        DevBuf<Ef> out(&ctx, h);
        h2d(ctx, d_beta.p, &beta, sizeof(Ef));
        launch_fri_fold_dev(ctx, folded.p, h, d_beta.p, out.p, nullptr);  // :119
        r.vec = std::move(folded);
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

    // ---- query phase :45-59: query q opens every commitment in tree q, i.e. on the rank that owns it
    std::vector<uint32_t> indices(Q);
    for (uint32_t q = 0; q < Q; q++) indices[q] = (uint32_t)challenger.sample_bits(log_N);
    const PcsData* in_data[2] = {trace_data.get(), quotient_data.get()};
    const TapCommit* in_commit[2] = {&trace_commit, &quotient_commit};
    // words of one answered query (the same for every query)
    size_t wpq = 1;
    for (auto* d : in_data) {
        wpq += 1 + d->ldes.size() + 1 + 8 * (size_t)log_N;
        for (auto& cm : d->ldes) wpq += cm.width;
    }
    for (uint32_t r = 0; r < R; r++) wpq += 8 + 1 + 8 * (size_t)rounds[r].commit.log_height;
    const uint32_t nq = sh.cnt;  // queries answered here: q0 .. q0 + nq - 1, in local trees 0 .. nq - 1
    std::vector<uint32_t> answers((size_t)(sh.comm ? sh.per : Q) * wpq, 0);
    if (nq) {
        std::vector<uint32_t> tree_of(nq), idx32(nq);
        std::vector<uint64_t> idx64(nq);
        for (uint32_t j = 0; j < nq; j++) {
            tree_of[j] = j;
            idx32[j] = indices[sh.q0 + j];
            idx64[j] = indices[sh.q0 + j];
        }
        DevBuf<uint32_t> d_idx(&ctx, nq), d_tree(&ctx, nq);
        DevBuf<uint64_t> d_idx64(&ctx, nq);
        h2d(ctx, d_idx.p, idx32.data(), nq * 4);
        h2d(ctx, d_tree.p, tree_of.data(), nq * 4);
        h2d(ctx, d_idx64.p, idx64.data(), nq * 8);
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
            DevBuf<uint32_t> d_rows(&ctx, (size_t)nq * lm.total_width), d_path(&ctx, (size_t)nq * 8 * log_N);
            launch_gather_rows(ctx, lm, d_idx.p, nq, 0, d_rows.p);
            launch_tap_gather_paths(ctx, in_commit[k]->trees.p, 2 * N - 1, log_N, d_tree.p, d_idx64.p, nq, d_path.p);
            in_rows[k].resize((size_t)nq * lm.total_width);
            in_paths[k].resize((size_t)nq * 8 * log_N);
            TS_HIP(hipMemcpyAsync(in_rows[k].data(), d_rows.p, in_rows[k].size() * 4, hipMemcpyDeviceToHost, ctx.stream));
            d2h_sync(ctx, in_paths[k].data(), d_path.p, in_paths[k].size() * 4);
        }
        // bf_answer_query :69-90: round i opens row index >> i >> 1 of its h x 2 matrix, in tree q
        std::vector<std::vector<uint32_t>> f_vals(R), f_paths(R);
        for (uint32_t r = 0; r < R; r++) {
            const unsigned ll = rounds[r].commit.log_height;
            std::vector<uint64_t> ri(nq);
            for (uint32_t j = 0; j < nq; j++) ri[j] = idx64[j] >> (r + 1);
            std::vector<uint32_t> ri32(ri.begin(), ri.end());
            DevBuf<uint64_t> d_ri(&ctx, nq);
            DevBuf<uint32_t> d_ri32(&ctx, nq), d_vals(&ctx, (size_t)nq * 8),
                d_path(&ctx, std::max<size_t>((size_t)nq * 8 * ll, 8));
            h2d(ctx, d_ri.p, ri.data(), nq * 8);
            h2d(ctx, d_ri32.p, ri32.data(), nq * 4);
            launch_gather_ef_pairs(ctx, rounds[r].vec.p, d_ri32.p, nq, 0, d_vals.p);
            launch_tap_gather_paths(ctx, rounds[r].commit.trees.p, (2ull << ll) - 1, ll, d_tree.p, d_ri.p, nq, d_path.p);
            f_vals[r].resize((size_t)nq * 8);
            f_paths[r].resize((size_t)nq * 8 * ll);
            TS_HIP(hipMemcpyAsync(f_vals[r].data(), d_vals.p, f_vals[r].size() * 4, hipMemcpyDeviceToHost, ctx.stream));
            if (ll) TS_HIP(hipMemcpyAsync(f_paths[r].data(), d_path.p, f_paths[r].size() * 4, hipMemcpyDeviceToHost, ctx.stream));
            ctx.sync();
        }
        std::vector<uint32_t> one;
        for (uint32_t j = 0; j < nq; j++) {
            one.clear();
            auto push = [&](uint32_t v) { one.push_back(v); };
            auto push_n = [&](const uint32_t* p, size_t k) { one.insert(one.end(), p, p + k); };
            auto push_path = [&](const uint32_t* state_words, size_t depth) {  // state words -> bytes read LE
                for (size_t k = 0; k < 8 * depth; k++) one.push_back(__builtin_bswap32(state_words[k]));
            };
            push(2);  // input_proof: one BatchOpening per commit round (two_adic_pcs.rs:399-414)
            for (int k = 0; k < 2; k++) {
                const auto& ldes = in_data[k]->ldes;
                size_t tw = 0;
                for (auto& cm : ldes) tw += cm.width;
                push((uint32_t)ldes.size());
                size_t c = (size_t)j * tw;
                for (auto& cm : ldes) {
                    push(cm.width);
                    push_n(&in_rows[k][c], cm.width);
                    c += cm.width;
                }
                push(log_N);
                push_path(&in_paths[k][(size_t)j * 8 * log_N], log_N);
            }
            for (uint32_t r = 0; r < R; r++) {  // commit_phase_openings
                const unsigned ll = rounds[r].commit.log_height;
                push_n(&f_vals[r][(size_t)j * 8], 8);
                push(ll);
                push_path(ll ? &f_paths[r][(size_t)j * 8 * ll] : nullptr, ll);
            }
            TS_REQUIRE(one.size() == wpq, TS_ERR_INVARIANT, "taptree query: segment size");
            memcpy(&answers[(size_t)j * wpq], one.data(), wpq * 4);
        }
    }
    if (sh.comm) {  // collect the answers: per x wpq words from every rank, rank order = query order
        const size_t seg = (size_t)sh.per * wpq, G = (size_t)sh.comm->world;
        DevBuf<uint32_t> d_send(&ctx, seg), d_recv(&ctx, seg * G);
        h2d(ctx, d_send.p, answers.data(), seg * 4);
        sh.comm->all_gather(d_send.p, d_recv.p, seg * 4, ctx.stream);
        answers.resize(seg * G);
        d2h_sync(ctx, answers.data(), d_recv.p, answers.size() * 4);
    }

    // ---- Proof (uni-stark/src/prover.rs:105-118), TSPF v2
    std::vector<uint32_t> pf;
    auto push = [&](uint32_t v) { pf.push_back(v); };
    auto push_n = [&](const uint32_t* p, size_t k) { pf.insert(pf.end(), p, p + k); };
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
    push_n(answers.data(), (size_t)Q * wpq);  // query q sits at q: the ranks' ranges are contiguous
    push_n(final_poly.c, 4);
    push(pow_witness);
    return pf;
}

}  // namespace ts
