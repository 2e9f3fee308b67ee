// One proof over G GPUs (BASELINE config 4 "FRI/Merkle sharded over 8 GPUs").
//
// Partition: rank g owns the bit-reversed LDE rows [g N/G, (g+1) N/G) of every committed matrix.
// With G <= 2^log_blowup these are whole cosets of H_n, so that
//   - the coset NTTs of a rank need only the (replicated) trace: no inter-GPU butterflies,
//   - a slab is a Merkle sub-tree: only the G sub-roots (32 B each) are exchanged per tree,
//   - the quotient's `next` row (natural index + qd) stays inside a coset,
//   - the reduce step is row-local, the opened values are interpolated on an owned coset,
//   - FRI folds adjacent rows (2i, 2i+1): local while a slab has >= 2 values,
//   - a query index and all its shifted indices index >> k belong to one rank.
// Exchange steps (collectives supplied by the host, RCCL over xGMI in production):
//   1. all-gather of the trace row slices (n w 4 / G bytes per rank) -- the one bulk transfer;
//      skipped when every rank already holds the whole trace (ShardOptions::trace_replicated),
//   2. broadcast of each quotient chunk from the rank that owns its coset (16 n bytes),
//   3. all-gather of G sub-roots per tree (2 commits + one per sharded FRI round),
//   4. all-gather of the FRI vector when the slabs get short (the rest runs replicated),
//   5. all-gather of the answered queries.
// The transcript is replicated: every rank observes the same roots and samples the same challenges.
// Proofs are bit-identical to prove()'s (tests/test_gpu_sharded.py).
#include <stdio.h>
#include <string.h>

#include "fri_internal.hpp"

namespace ts {

namespace {

struct Shard {
    Context& ctx;
    const FriConfig& fri;
    const Comm& comm;
    uint32_t G, rank, log_G;
    uint32_t cosets, beta0;  // cosets per rank, first owned coset
};

// what one rank keeps of a committed batch
struct ShardedData {
    PcsData local;              // slab LDEs (height N/G) and the slab's Merkle tree
    unsigned log_global = 0;    // log2 N
    std::vector<uint32_t> top;  // (2G-1) x 8 words: the G sub-roots, ..., the root
    uint32_t root[8];
};

// Every collective of a proof goes through these two.  With the stage timers on (ts_ctx_set_timing)
// each call leaves one entry "collective: <kind> <bytes> B (<site>)" = ms between two events on the
// context's stream around the call (enqueue -> complete, the wait for the slowest peer included), so
// that a run on real GPUs says per collective where the time went (bench.py: `collectives`).
void coll_all_gather(Context& ctx, const Comm& comm, const char* site, const void* send, void* recv,
                     size_t bytes_per_rank) {
    char name[160];
    snprintf(name, sizeof name, "collective: all_gather %zu B/rank (%s)", bytes_per_rank, site);
    StageTimer t(&ctx, name);
    comm.all_gather(send, recv, bytes_per_rank, ctx.stream);
}
void coll_broadcast(Context& ctx, const Comm& comm, const char* site, void* buf, size_t bytes, int root) {
    char name[160];
    snprintf(name, sizeof name, "collective: broadcast %zu B (%s)", bytes, site);
    StageTimer t(&ctx, name);
    comm.broadcast(buf, bytes, root, ctx.stream);
}

// sub-root of this rank -> gathered sub-roots -> top levels (+ challenger step) on the device
void gather_top(Shard& sh, const char* site, const uint32_t* d_subroot, uint32_t* d_top, DevChallenger* dch,
                uint32_t* d_root_out, Ef* d_beta_out) {
    DevBuf<uint32_t> all(&sh.ctx, 8 * (size_t)sh.G);
    coll_all_gather(sh.ctx, sh.comm, site, d_subroot, all.p, 32);
    launch_shard_top(sh.ctx, all.p, sh.G, d_top, dch, d_root_out, d_beta_out);
}

// two_adic_pcs.rs:227-245 for the owned cosets + the slab's sub-tree; `evals` are whole matrices
// mix (local quotient, below): a qd x qd canonical matrix applied across the batch's matrices (the
// chunk LDEs, 4 columns each) between the LDE and the leaf hashes; mix_local: `evals` are this rank's
// own matrices (they differ from rank to rank), so the column-sharded inverse does not apply
std::unique_ptr<ShardedData> commit_sharded(Shard& sh, std::vector<DeviceMatrix>& evals,
                                            const std::vector<uint32_t>& domain_shifts,
                                            const std::vector<uint32_t>* mix = nullptr, bool mix_local = false) {
    Context& ctx = sh.ctx;
    TS_REQUIRE(!evals.empty() && evals.size() <= (size_t)MAX_BATCH_MATS, TS_ERR_INVALID,
               "commit: between 1 and 16 matrices per batch");
    const uint64_t n = evals[0].height;
    const unsigned log_n = log2_strict(n);
    const unsigned log_N = log_n + sh.fri.log_blowup;
    TS_REQUIRE(log_N <= 27, TS_ERR_INVALID, "commit: LDE larger than the two-adic subgroup");
    const uint64_t rows = (uint64_t)sh.cosets << log_n;  // slab height
    const unsigned log_rows = log2_strict(rows);
    ctx.ensure_twiddles(std::max(1u, log_N));

    auto data = std::make_unique<ShardedData>();
    data->log_global = log_N;
    PcsData& loc = data->local;
    loc.log_height = log_rows;
    {
        StageTimer t(&ctx, "coset_lde");
        // a batch (the quotient chunks) in one allocation, matrix after matrix, as TwoAdicFriPcs::commit
        // does: the slab's leaf hash then takes it as one strided matrix
        size_t total_w = 0;
        for (auto& m : evals) total_w += m.width;
        DevBuf<uint32_t> batch;
        if (evals.size() > 1) batch = DevBuf<uint32_t>(&ctx, total_w * rows);
        size_t batch_col = 0;
        for (size_t i = 0; i < evals.size(); i++) {
            DeviceMatrix& m = evals[i];
            TS_REQUIRE(m.height == n && m.width >= 1 && m.buf.p, TS_ERR_INVALID,
                       "sharded commit: matrices of one height expected");
            DevBuf<uint32_t> colmajor;
            uint32_t* ev = m.buf.p;
            struct { uint32_t* p; } lde{nullptr};
            DevBuf<uint32_t> own;
            if (batch.p) {
                lde.p = batch.p + batch_col * rows;
                batch_col += m.width;
            } else {
                own = DevBuf<uint32_t>(&ctx, (size_t)m.width * rows);
                lde.p = own.p;
            }
            const uint32_t shift = mul(GENERATOR, inv_canon(domain_shifts[i]));  // two_adic_pcs.rs:235
            // The inverse transform is replicated: every rank transposes and inverts every column, then
            // runs the forward transforms of its own cosets.  (Rounds 2-4 carried an option that did the
            // per-column part of the inverse for w/G columns per rank and all-gathered the
            // half-transformed columns, SURVEY.md section 8(e) steps 1-2: it took 0.5 / 0.35 ms off the
            // compute path of configs 4 / 5 at G = 8 and added 1.5 / 0.8 ms of bulk all-gathers by the link
            // model -- removed in round 5, numbers in HISTORY.md.)
            {
                bool r16 = false;
                if (m.layout == DeviceMatrix::ROW_MAJOR) {
                    StageTimer t(&ctx, "lde: transpose (every column on every rank)");
                    colmajor = DevBuf<uint32_t>(&ctx, (size_t)m.width * n);
                    r16 = launch_transpose_bitrev_r16(ctx, m.buf.p, colmajor.p, log_n, m.width, n);
                    if (!r16) launch_transpose_bitrev(ctx, m.buf.p, colmajor.p, log_n, m.width, n);
                    ev = colmajor.p;
                }
                coset_lde(ctx, ev, n, m.width, log_n, sh.fri.log_blowup, shift, lde.p, rows, sh.beta0, sh.cosets,
                          r16);
            }
            ColMat cm;
            cm.d = lde.p;
            cm.height = rows;
            cm.width = m.width;
            cm.col_stride = rows;
            loc.ldes.push_back(cm);
            if (own.p) loc.lde_storage.push_back(std::move(own));
            m.buf.reset();  // consumed
        }
        if (batch.p) loc.lde_storage.push_back(std::move(batch));
    }
    if (mix) {
        StageTimer t(&ctx, "mix chunk LDEs (local quotient)");
        const uint32_t qd = (uint32_t)evals.size();
        TS_REQUIRE(mix->size() == (size_t)qd * qd, TS_ERR_INVARIANT, "chunk mix: matrix size");
        std::vector<uint32_t*> ptrs;
        for (auto& cm : loc.ldes) {
            TS_REQUIRE(cm.width == 4 && cm.col_stride == rows, TS_ERR_INVARIANT, "chunk mix: chunk LDE shape");
            ptrs.push_back(cm.d);
        }
        std::vector<uint32_t> mm(mix->size());
        for (size_t i = 0; i < mm.size(); i++) mm[i] = to_mont((*mix)[i]);
        DevBuf<uint32_t*> d_ptrs(&ctx, ptrs.size());
        DevBuf<uint32_t> d_mix(&ctx, mm.size());
        h2d(ctx, d_ptrs.p, ptrs.data(), ptrs.size() * sizeof(uint32_t*));
        h2d(ctx, d_mix.p, mm.data(), mm.size() * 4);
        launch_chunk_mix(ctx, d_ptrs.p, qd, rows, rows, d_mix.p);
    }
    {
        StageTimer t(&ctx, "merkle_commit");
        loc.tree = DevBuf<uint32_t>(&ctx, merkle_total_digests(log_rows) * 8);
        std::vector<const uint32_t*> cols;
        for (auto& cm : loc.ldes)
            for (uint32_t c = 0; c < cm.width; c++) cols.push_back(cm.d + (uint64_t)c * cm.col_stride);
        loc.col_table = DevBuf<const uint32_t*>(&ctx, cols.size());
        h2d(ctx, loc.col_table.p, cols.data(), cols.size() * sizeof(const uint32_t*));
        loc.col_table_uploaded = true;
        LeafMats lm = loc.leaf_mats();
        if (loc.ldes.size() > 1) {  // laid back to back above: one matrix of the summed width for the leaf hash
            bool contiguous = true;
            uint32_t wsum = 0;
            for (auto& cm : loc.ldes) {
                contiguous = contiguous && cm.col_stride == rows && cm.d == loc.ldes[0].d + (uint64_t)wsum * rows;
                wsum += cm.width;
            }
            if (contiguous && wsum <= 256) {
                lm.n_mats = 1;
                lm.width[0] = wsum;
            }
        }
        launch_commit_tree(ctx, lm, log_rows, loc.tree.p);  // leaves + the slab's sub-tree
        DevBuf<uint32_t> d_top(&ctx, 8 * (size_t)(2 * sh.G - 1));
        gather_top(sh, "commit sub-roots", loc.tree.p + 8 * (merkle_total_digests(log_rows) - 1), d_top.p,
                   nullptr, nullptr, nullptr);
        data->top.resize(8 * (size_t)(2 * sh.G - 1));
        d2h_sync(ctx, data->top.data(), d_top.p, data->top.size() * 4);  // also covers `cols`
        memcpy(data->root, &data->top[8 * (size_t)(2 * sh.G - 2)], 32);
        memcpy(loc.root, &data->top[8 * (size_t)sh.rank], 32);
    }
    return data;
}

// ---- local quotient ------------------------------------------------------------------------------
// The reference evaluates the quotient q = constraints / Z_H on the quotient domain 31 H_{n qd}
// (prover.rs:65-77) -- in the sharded layout the first qd cosets, all on the low ranks -- splits the
// values into qd chunks (chunk c = the values on D_c = 31 w^c H_n, w = omega_{n qd}; :78-80) and
// commits to the LDE of each chunk's interpolant q_c (:82-83).  Writing q = sum_k X^(kn) Q_k with
// deg Q_k < n, that interpolant is q mod (X^n - a_c) = sum_k a_c^k Q_k, a_c = (31 w^c)^n = 31^n w_qd^c.
// A rank that owns cosets beta0 .. beta0 + qd - 1 (aligned) holds the trace LDE on s_g H_{n qd},
// s_g = 31 omega_N^bitrev(beta0), in the same bit-reversed layout, so the SAME quotient kernel run on
// its slab with shift s_g gives q there: the values on s_g w^c' H_n, whose interpolants are
// r_c' = q mod (X^n - b_c') = sum_k b_c'^k Q_k, b_c' = s_g^n w_qd^c'.  Both families are the images of
// (Q_k) under Vandermonde matrices, so q_c = sum_c' M[c][c'] r_c' with M = V(a) V(b)^-1, a qd x qd
// matrix of base-field constants: the rank extends the r_c' to its cosets like any committed matrix
// (coset_lde with the domain shift s_g w^c') and mixes the qd results row by row.  No rank waits for
// the owner of the quotient domain and nothing is broadcast.  Exact field arithmetic: the chunk LDEs
// are the reference's whenever q is a polynomial of degree < n qd, i.e. whenever the trace satisfies
// its constraints (ShardOptions::local_quotient).
std::vector<uint32_t> chunk_mix_matrix(uint32_t qd, unsigned log_n, uint32_t s_g) {
    const unsigned lqd = log2_strict(qd);
    const uint32_t gqd = two_adic_generator(lqd);
    const uint32_t an = pow_canon(GENERATOR, 1ull << log_n), bn = pow_canon(s_g, 1ull << log_n);
    std::vector<uint32_t> A((size_t)qd * qd), B((size_t)qd * qd), Binv((size_t)qd * qd, 0);
    for (uint32_t c = 0; c < qd; c++) {
        const uint32_t a = mul(an, pow_canon(gqd, c)), b = mul(bn, pow_canon(gqd, c));
        uint32_t pa = 1, pb = 1;
        for (uint32_t k = 0; k < qd; k++) {
            A[(size_t)c * qd + k] = pa;
            B[(size_t)c * qd + k] = pb;
            pa = mul(pa, a);
            pb = mul(pb, b);
        }
        Binv[(size_t)c * qd + c] = 1;
    }
    // Gauss-Jordan on [B | I] over the field (the b_c' are distinct: B is invertible)
    for (uint32_t col = 0; col < qd; col++) {
        uint32_t piv = col;
        while (piv < qd && B[(size_t)piv * qd + col] == 0) piv++;
        TS_REQUIRE(piv < qd, TS_ERR_INVARIANT, "chunk mix: singular Vandermonde");
        if (piv != col)
            for (uint32_t k = 0; k < qd; k++) {
                std::swap(B[(size_t)piv * qd + k], B[(size_t)col * qd + k]);
                std::swap(Binv[(size_t)piv * qd + k], Binv[(size_t)col * qd + k]);
            }
        const uint32_t inv = inv_canon(B[(size_t)col * qd + col]);
        for (uint32_t k = 0; k < qd; k++) {
            B[(size_t)col * qd + k] = mul(B[(size_t)col * qd + k], inv);
            Binv[(size_t)col * qd + k] = mul(Binv[(size_t)col * qd + k], inv);
        }
        for (uint32_t r = 0; r < qd; r++) {
            const uint32_t f = B[(size_t)r * qd + col];
            if (r == col || f == 0) continue;
            for (uint32_t k = 0; k < qd; k++) {
                B[(size_t)r * qd + k] = sub(B[(size_t)r * qd + k], mul(f, B[(size_t)col * qd + k]));
                Binv[(size_t)r * qd + k] = sub(Binv[(size_t)r * qd + k], mul(f, Binv[(size_t)col * qd + k]));
            }
        }
    }
    std::vector<uint32_t> M((size_t)qd * qd, 0);
    for (uint32_t c = 0; c < qd; c++)
        for (uint32_t cp = 0; cp < qd; cp++) {
            uint32_t acc = 0;
            for (uint32_t k = 0; k < qd; k++) acc = add(acc, mul(A[(size_t)c * qd + k], Binv[(size_t)k * qd + cp]));
            M[(size_t)c * qd + cp] = acc;
        }
    return M;
}

// siblings of sub-tree `rank` in the top levels, leaf-most first
void push_top_path(std::vector<uint32_t>& out, const uint32_t* top, uint32_t G, uint32_t rank) {
    uint32_t off = 0;
    for (uint32_t cnt = G, l = 0; cnt > 1; cnt >>= 1, l++) {
        const uint32_t* node = top + 8 * (size_t)(off + ((rank >> l) ^ 1));
        out.insert(out.end(), node, node + 8);
        off += cnt;
    }
}

}  // namespace

std::vector<uint32_t> prove_sharded(TwoAdicFriPcs& pcs, const Comm& comm, const AirProgram& air,
                                    BfChallenger& challenger, DeviceMatrix trace_rows,
                                    const std::vector<uint32_t>& public_values, const ShardOptions& opt) {
    Context& ctx = pcs.ctx();
    const FriConfig& fri = pcs.fri();
    TS_REQUIRE(comm.world >= 1 && comm.rank >= 0 && comm.rank < comm.world && comm.all_gather &&
                   comm.broadcast,
               TS_ERR_INVALID, "prove_sharded: bad communicator");
    const uint32_t G = (uint32_t)comm.world;
    TS_REQUIRE((G & (G - 1)) == 0 && G <= fri.blowup(), TS_ERR_UNSUPPORTED,
               "prove_sharded: the number of ranks must be a power of two <= 2^log_blowup (whole "
               "cosets per rank); run independent proofs per GPU otherwise");
    Shard sh{ctx, fri, comm, G, (uint32_t)comm.rank, log2_strict(G), fri.blowup() / G,
             (uint32_t)comm.rank * (fri.blowup() / G)};
    TS_REQUIRE(trace_rows.width == air.width, TS_ERR_INVALID, "prove: trace width != AIR width");
    TS_REQUIRE(trace_rows.layout == DeviceMatrix::ROW_MAJOR && trace_rows.buf.p, TS_ERR_INVALID,
               "prove_sharded: the trace slice must be an uploaded row-major matrix");
    TS_REQUIRE(public_values.size() == air.n_public, TS_ERR_INVALID,
               "prove: wrong number of public values");
    const uint32_t w = air.width;
    const uint64_t degree = opt.trace_replicated ? trace_rows.height : trace_rows.height * G;  // prover.rs:43-44
    const unsigned log_degree = log2_strict(degree);
    const unsigned lqd = air.log_quotient_degree;
    const uint32_t qd = 1u << lqd;
    const unsigned log_N = log_degree + fri.log_blowup;
    const uint64_t N = 1ull << log_N, n = degree;
    const uint64_t loc0 = N / G;  // slab height
    TS_REQUIRE(lqd <= fri.log_blowup, TS_ERR_INVARIANT,
               "quotient domain larger than the committed LDE (log_quotient_degree > log_blowup)");
    ctx.ensure_twiddles(std::max(1u, log_N));
    TwoAdicFriPcs::Slab slab;
    slab.row0 = (uint64_t)sh.rank * loc0;
    slab.rows = loc0;
    slab.beta0 = sh.beta0;

    // ---- exchange 1: every rank gets the whole trace (row slices in rank order = natural order)
    DeviceMatrix trace;
    if (opt.trace_replicated) {
        trace = std::move(trace_rows);
    } else {
        StageTimer t(&ctx, "all-gather trace");
        trace.buf = DevBuf<uint32_t>(&ctx, (size_t)n * w);
        trace.height = n;
        trace.width = w;
        trace.layout = DeviceMatrix::ROW_MAJOR;
        coll_all_gather(ctx, comm, "trace rows", trace_rows.buf.p, trace.buf.p, (size_t)trace_rows.height * w * 4);
        trace_rows.buf.reset();
    }

    // prover.rs:50-63
    std::vector<DeviceMatrix> tv;
    tv.push_back(std::move(trace));
    std::unique_ptr<ShardedData> trace_data = commit_sharded(sh, tv, {1u});
    challenger.observe_commitment(trace_data->root);
    const Ef alpha = challenger.sample();

    // Everything after the trace commitment, with the quotient made locally or broadcast.  The local
    // path is exact whenever constraints / Z_H is a polynomial (every valid trace); for an invalid
    // trace its mixed chunk LDEs are not low-degree and FRI's final polynomial is not constant -- on
    // every rank alike, the final vector being replicated -- where ts_prove and the broadcast path
    // (and a release build of the reference, uni-stark/src/prover.rs:40-41) still hand out a proof
    // for the verifier to reject.  So that the call stays a drop-in for EVERY trace, that one failure
    // sends all ranks back to the state after alpha and through the broadcast path.
    const BfChallenger after_alpha = challenger;
    auto rest = [&](bool local_quotient) -> std::vector<uint32_t> {
    // :65-80 the quotient domain is the first qd cosets: chunk c = coset bitrev(c), computed by the
    // rank that owns that coset, then broadcast (exchange 2)
    std::vector<uint32_t> qshifts(qd);
    const uint32_t gq = two_adic_generator(log_degree + lqd);
    std::unique_ptr<ShardedData> quotient_data;
    if (local_quotient) {
        // every rank on its own cosets ("local quotient" above); with qd = 1 the values on the rank's
        // first coset determine q itself and the mix is the identity
        const uint32_t s_g =
            mul(GENERATOR, pow_canon(two_adic_generator(log_N), bitrev32(sh.beta0, fri.log_blowup)));
        std::vector<DeviceMatrix> chunks = pcs.quotient_chunks_slab(trace_data->local.ldes[0], log_degree,
                                                                    TwoAdicFriPcs::Slab{}, air, public_values,
                                                                    alpha, s_g);
        for (uint32_t c = 0; c < qd; c++) qshifts[c] = mul(s_g, pow_canon(gq, c));
        std::vector<uint32_t> mix;
        if (qd > 1) mix = chunk_mix_matrix(qd, log_degree, s_g);
        quotient_data = commit_sharded(sh, chunks, qshifts, qd > 1 ? &mix : nullptr, true);  // :82-83
    } else {
        std::vector<DeviceMatrix> chunks =
            pcs.quotient_chunks_slab(trace_data->local.ldes[0], log_degree, slab, air, public_values, alpha);
        {
            StageTimer t(&ctx, "broadcast quotient chunks");
            for (uint32_t c = 0; c < qd; c++) {
                const uint32_t owner = bitrev32(c, lqd) / sh.cosets;
                coll_broadcast(ctx, comm, "quotient chunk", chunks[c].buf.p, (size_t)n * 16, (int)owner);
            }
        }
        for (uint32_t c = 0; c < qd; c++) qshifts[c] = mul(GENERATOR, pow_canon(gq, c));
        quotient_data = commit_sharded(sh, chunks, qshifts);  // :82-83
    }
    challenger.observe_commitment(quotient_data->root);                                 // :84
    const Ef zeta = challenger.sample();                                                // :91

    // :94-104 open: values from an owned coset, reduce on the slab
    const Ef batch_alpha = challenger.sample();
    std::vector<Ef> opened;
    DevBuf<Ef> folded =
        pcs.open_reduce_slab(trace_data->local, quotient_data->local, log_N, slab, zeta, batch_alpha, opened);

    // ---- bf_commit_phase (fri/src/prover.rs:93-141): sharded rounds, then replicated rounds
    FriCommit st;
    Ef final_poly;
    uint32_t R_sh = 0;                // sharded rounds: st.rounds[0..R_sh) describe slabs
    DevBuf<uint32_t> d_tops;          // [R_sh][2G-1][8]
    std::vector<uint32_t> tops;
    const size_t top_words = 8 * (size_t)(2 * G - 1);
    {
        StageTimer t(&ctx, "FRI commit phase");
        fri_commit_begin(ctx, fri, log_N, challenger, st);
        d_tops = DevBuf<uint32_t>(&ctx, std::max<size_t>(top_words * st.R_total, 8));
        uint64_t len = N, loc = loc0;
        const uint64_t min_loc = std::max<uint64_t>(2, 1ull << opt.min_local_log);
        auto sharded_round = [&](uint64_t l_glob, uint64_t l_loc) {
            return l_glob > fri.blowup() && l_loc >= min_loc;
        };
        // != nullptr: `folded` is not in memory yet -- it is the fold of this slab (the last round's)
        // with the last round's challenge, and the next round's launch computes it while hashing
        const Ef* prev = nullptr;
        while (sharded_round(len, loc)) {
            FriRound r;
            const uint64_t h_loc = loc / 2, h_glob = len / 2;
            r.log_leaves = log2_strict(h_loc);
            const size_t ri = st.rounds.size();
            DevBuf<uint32_t> tree(&ctx, merkle_total_digests(r.log_leaves) * 8);
            if (prev) folded = DevBuf<Ef>(&ctx, loc);
            // the slab's sub-tree: (fold +) leaves + levels in one launch (leaf_tree.hpp)
            launch_fri_round_tall(ctx, prev, prev ? st.d_betas.p + ri - 1 : nullptr, folded.p, h_loc, tree.p, nullptr,
                                  nullptr, nullptr, h_glob, (uint64_t)sh.rank * h_loc);
            // exchange 3; the top kernel observes the root and samples beta on every rank alike
            gather_top(sh, "FRI round sub-roots", tree.p + 8 * (merkle_total_digests(r.log_leaves) - 1),
                       d_tops.p + top_words * ri,
                       st.dch(), st.d_roots.p + 8 * ri, st.d_betas.p + ri);
            r.vec = folded.p;
            r.tree = tree.p;
            if (sharded_round(h_glob, h_loc)) {
                prev = folded.p;  // fri/src/prover.rs:119 happens inside the next round's launch
                st.keep_vecs.push_back(std::move(folded));
            } else {
                DevBuf<Ef> out(&ctx, h_loc);
                launch_fri_fold_dev(ctx, folded.p, h_loc, st.d_betas.p + ri, out.p, nullptr, h_glob,
                                    (uint64_t)sh.rank * h_loc);
                st.keep_vecs.push_back(std::move(folded));
                folded = std::move(out);
                prev = nullptr;
            }
            st.keep_trees.push_back(std::move(tree));
            st.rounds.push_back(r);
            len = h_glob;
            loc = h_loc;
            R_sh++;
        }
        // exchange 4: the rest is short; every rank folds the whole vector
        DevBuf<Ef> full(&ctx, len);
        coll_all_gather(ctx, comm, "FRI vector", folded.p, full.p, (size_t)loc * sizeof(Ef));
        st.keep_vecs.push_back(std::move(folded));
        std::vector<DevBuf<Ef>> no_inputs;
        fri_commit_rounds(ctx, fri, std::move(full), len, no_inputs, {}, 0, st);
        tops.resize(std::max<size_t>(top_words * R_sh, 8));
        if (R_sh)
            TS_HIP(hipMemcpyAsync(tops.data(), d_tops.p, top_words * R_sh * 4, hipMemcpyDeviceToHost,
                                  ctx.stream));
        final_poly = fri_commit_finish(ctx, fri, challenger, st);
    }
    const uint32_t R = (uint32_t)st.rounds.size();

    uint32_t pow_witness;
    {
        StageTimer t(&ctx, "grind for proof-of-work witness");
        pow_witness = fri_pow_witness(ctx, challenger, fri.proof_of_work_bits, st);  // prover.rs:43
    }

    // ---- query phase (prover.rs:45-59): a rank answers the queries that fall into its slab
    StageTimer tq(&ctx, "query phase");
    const uint32_t Q = fri.num_queries;
    std::vector<uint32_t> indices(Q);
    for (uint32_t q = 0; q < Q; q++) indices[q] = (uint32_t)challenger.sample_bits(log_N);
    std::vector<uint32_t> own, li, gi;
    for (uint32_t q = 0; q < Q; q++)
        if (indices[q] / loc0 == sh.rank) {
            own.push_back(q);
            li.push_back((uint32_t)(indices[q] - slab.row0));
            gi.push_back(indices[q]);
        }
    const uint32_t n_own = (uint32_t)own.size();
    const ShardedData* in_rounds[2] = {trace_data.get(), quotient_data.get()};
    const unsigned log_loc0 = log2_strict(loc0);
    // words of one answered query (the same for every query)
    size_t wpq = 1;
    for (auto* d : in_rounds) {
        wpq += 1 + d->local.ldes.size() + 1 + 8 * (size_t)log_N;
        for (auto& m : d->local.ldes) wpq += m.width;
    }
    for (uint32_t r = 0; r < R; r++) wpq += 8 + 1 + 8 * (size_t)(log_N - 1 - r);
    std::vector<uint32_t> seg((size_t)Q * wpq, 0);
    if (n_own) {
        DevBuf<uint32_t> d_li(&ctx, n_own), d_gi(&ctx, n_own);
        h2d(ctx, d_li.p, li.data(), n_own * 4);
        h2d(ctx, d_gi.p, gi.data(), n_own * 4);
        LeafMats lms[2];
        size_t o_rows[2], o_path[2], off = 0;
        for (int k = 0; k < 2; k++) {
            lms[k] = in_rounds[k]->local.leaf_mats();
            o_rows[k] = off; off += (size_t)n_own * lms[k].total_width;
            o_path[k] = off; off += (size_t)n_own * 8 * log_loc0;
        }
        std::vector<size_t> o_fvals(R), o_fpath(R);
        for (uint32_t r = 0; r < R; r++) {
            o_fvals[r] = off; off += (size_t)n_own * 8;
            o_fpath[r] = off; off += (size_t)n_own * 8 * st.rounds[r].log_leaves;
        }
        DevBuf<uint32_t> d_out(&ctx, std::max<size_t>(off, 1));
        for (int k = 0; k < 2; k++) {
            launch_gather_rows(ctx, lms[k], d_li.p, n_own, 0, d_out.p + o_rows[k]);
            launch_gather_paths(ctx, in_rounds[k]->local.tree.p, log_loc0, d_li.p, n_own, 0,
                                d_out.p + o_path[k]);
        }
        // bf_answer_query :69-90: sharded rounds by local index, replicated rounds by global index
        std::vector<FriGatherDesc> descs(std::max(R, 1u));
        uint32_t max_sh = 0, max_rep = 0;
        for (uint32_t r = 0; r < R; r++) {
            descs[r].vec = reinterpret_cast<const uint32_t*>(st.rounds[r].vec);
            descs[r].tree = st.rounds[r].tree;
            descs[r].log_leaves = st.rounds[r].log_leaves;
            descs[r].shift = r + 1;
            descs[r].out_vals = o_fvals[r];
            descs[r].out_path = o_fpath[r];
            (r < R_sh ? max_sh : max_rep) = std::max(r < R_sh ? max_sh : max_rep, st.rounds[r].log_leaves);
        }
        DevBuf<FriGatherDesc> d_descs(&ctx, descs.size());
        h2d(ctx, d_descs.p, descs.data(), descs.size() * sizeof(FriGatherDesc));
        launch_gather_fri(ctx, d_descs.p, R_sh, max_sh, d_li.p, n_own, d_out.p);
        launch_gather_fri(ctx, d_descs.p + R_sh, R - R_sh, max_rep, d_gi.p, n_own, d_out.p);
        std::vector<uint32_t> g(std::max<size_t>(off, 1));
        d2h_sync(ctx, g.data(), d_out.p, off * 4);

        std::vector<uint32_t> one;
        for (uint32_t j = 0; j < n_own; j++) {
            one.clear();
            auto push = [&](uint32_t v) { one.push_back(v); };
            auto push_n = [&](const uint32_t* p, size_t k) { one.insert(one.end(), p, p + k); };
            push(2);  // input_proof: one BatchOpening per commit round (two_adic_pcs.rs:399-414)
            for (int k = 0; k < 2; k++) {
                push(lms[k].n_mats);
                size_t c = o_rows[k] + (size_t)j * lms[k].total_width;
                for (uint32_t i = 0; i < lms[k].n_mats; i++) {
                    push(lms[k].width[i]);
                    push_n(&g[c], lms[k].width[i]);
                    c += lms[k].width[i];
                }
                push(log_N);
                push_n(&g[o_path[k] + (size_t)j * 8 * log_loc0], 8 * (size_t)log_loc0);
                push_top_path(one, in_rounds[k]->top.data(), G, sh.rank);
            }
            for (uint32_t r = 0; r < R; r++) {  // commit_phase_openings
                const unsigned ll = st.rounds[r].log_leaves;
                push_n(&g[o_fvals[r] + (size_t)j * 8], 8);
                push(log_N - 1 - r);
                push_n(&g[o_fpath[r] + (size_t)j * 8 * ll], 8 * (size_t)ll);
                if (r < R_sh) push_top_path(one, &tops[top_words * r], G, sh.rank);
            }
            TS_REQUIRE(one.size() == wpq, TS_ERR_INVARIANT, "sharded query: segment size");
            memcpy(&seg[(size_t)own[j] * wpq], one.data(), wpq * 4);
        }
    }
    // ---- exchange 5: collect the answers
    std::vector<uint32_t> all_seg((size_t)G * Q * wpq);
    if (Q) {
        DevBuf<uint32_t> d_seg(&ctx, seg.size()), d_all(&ctx, all_seg.size());
        h2d(ctx, d_seg.p, seg.data(), seg.size() * 4);
        coll_all_gather(ctx, comm, "query answers", d_seg.p, d_all.p, seg.size() * 4);
        d2h_sync(ctx, all_seg.data(), d_all.p, all_seg.size() * 4);
    }

    // ---- Proof (uni-stark/src/prover.rs:105-118) in TSPF v1 order
    std::vector<uint32_t> pf;
    pf.reserve(64 + opened.size() * 4 + 8 * (size_t)R + (size_t)Q * wpq);
    pf.push_back(TSPF_MAGIC);
    pf.push_back(1);
    pf.push_back(log_degree);
    pf.push_back(w);
    pf.push_back(qd);
    pf.insert(pf.end(), trace_data->root, trace_data->root + 8);
    pf.insert(pf.end(), quotient_data->root, quotient_data->root + 8);
    for (auto& e : opened) pf.insert(pf.end(), e.c, e.c + 4);
    pf.push_back(R);
    for (uint32_t r = 0; r < R; r++) pf.insert(pf.end(), st.rounds[r].root, st.rounds[r].root + 8);
    pf.push_back(Q);
    for (uint32_t q = 0; q < Q; q++) {
        const size_t owner = indices[q] / loc0;
        const uint32_t* s = &all_seg[(owner * Q + q) * wpq];
        pf.insert(pf.end(), s, s + wpq);
    }
    pf.insert(pf.end(), final_poly.c, final_poly.c + 4);
    pf.push_back(pow_witness);
    return pf;
    };  // rest

    if (opt.local_quotient && sh.cosets >= qd) {
        try {
            return rest(true);
        } catch (const FinalPolyNotConstant&) {
            ctx.sync();  // the failed attempt's launches are done before its buffers go back to the pool
            ctx.local_quotient_fallbacks++;
            challenger = after_alpha;
        }
    }
    return rest(false);
}

}  // namespace ts
