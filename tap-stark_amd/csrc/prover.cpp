// prove() and TwoAdicFriPcs on the device.  Transcript order, proof structure and every index
// convention follow the reference:
//   uni-stark/src/prover.rs:25-119      prove
//   fri/src/two_adic_pcs.rs:227-245     commit
//   fri/src/two_adic_pcs.rs:247-258     get_evaluations_on_domain (fused into the quotient kernel)
//   fri/src/two_adic_pcs.rs:260-419     open
//   fri/src/prover.rs:19-141            bf_prove / bf_commit_phase / bf_answer_query
// The GPU owns the data from the uploaded trace to the opened rows; the host owns the transcript
// (one 32-byte root down, one challenge up per commitment).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fri_internal.hpp"

namespace ts {

// small uploads go through the context's page-locked arena: truly asynchronous, and the caller's
// buffer (a stack temporary, a vector about to die) is free as soon as this returns
// (uploads above 1 MiB -- lock-script tables, script blobs -- go straight from the caller's buffer,
// which the caller keeps alive until its next blocking call)
void h2d(Context& ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    const void* from = bytes <= (1u << 20) ? ctx.stage(src, bytes) : src;
    TS_HIP(hipMemcpyAsync(dst, from, bytes, hipMemcpyHostToDevice, ctx.stream));
}
void d2h_sync(Context& ctx, void* dst, const void* src, size_t bytes) {
    if (bytes) TS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx.stream));
    ctx.sync();
}

unsigned log2_strict(uint64_t n) {
    unsigned k = 0;
    while ((1ull << k) < n) k++;
    TS_REQUIRE((1ull << k) == n, TS_ERR_INVALID, "height must be a power of two");
    return k;
}

// canonical-domain EF helpers for the handful of host-side scalars
Ef efc_mul(Ef a, Ef b) { return ef_mul(a, ef_to_mont(b)); }
Ef efc_mul_base(Ef a, uint32_t b) { return ef_mul_base(a, to_mont(b)); }
Ef efc_pow(Ef a, uint64_t e) { return ef_from_mont(ef_pow(ef_to_mont(a), e)); }
Ef efc_one() { return Ef{{1, 0, 0, 0}}; }

LeafMats PcsData::leaf_mats() const {
    LeafMats lm;
    memset(&lm, 0, sizeof lm);
    lm.n_mats = (uint32_t)ldes.size();
    for (size_t i = 0; i < ldes.size(); i++) {
        lm.d[i] = ldes[i].d;
        lm.col_stride[i] = ldes[i].col_stride;
        lm.width[i] = ldes[i].width;
        lm.row_shift[i] = (uint8_t)(log_height - log2_strict(ldes[i].height));
        lm.total_width += ldes[i].width;
    }
    // the table is only uploaded when a leaf kernel addressed columns through it (mmcs_commit); readers
    // of a strided-committed batch use d[] / col_stride and must not be handed uninitialised pointers
    lm.cols = col_table_uploaded ? col_table.p : nullptr;
    return lm;
}

// ------------------------------------------------------------------ BFMmcs::commit
// basic/src/mmcs/bf_mmcs.rs:22-35 on matrices already resident (column-major): builds the Blake3
// Merkle tree over data.ldes (data.log_height = log2 of the tallest) and leaves the root in data.root.
void mmcs_commit(Context& ctx, PcsData& data) {
    const unsigned log_H = data.log_height;
    const uint64_t N = 1ull << log_H;
    bool uniform = true;
    for (auto& cm : data.ldes) uniform = uniform && cm.height == N;
    {
        StageTimer t(&ctx, "merkle_commit");
        data.tree = DevBuf<uint32_t>(&ctx, merkle_total_digests(log_H) * 8);
        // column pointers grouped by height (tallest first), commit order inside a group
        std::vector<const uint32_t*> cols;
        struct Group { uint64_t height; size_t first; uint32_t total; uint32_t n_mats; const ColMat* only; };
        std::vector<Group> groups;
        for (unsigned lh = log_H + 1; lh-- > 0;) {
            Group g{1ull << lh, cols.size(), 0, 0, nullptr};
            for (auto& cm : data.ldes)
                if (cm.height == g.height) {
                    for (uint32_t c = 0; c < cm.width; c++) cols.push_back(cm.d + (uint64_t)c * cm.col_stride);
                    g.total += cm.width;
                    g.n_mats++;
                    g.only = &cm;
                }
            if (g.total) groups.push_back(g);
        }
        data.col_table = DevBuf<const uint32_t*>(&ctx, cols.size());
        bool table_uploaded = false;
        auto upload_table = [&] {  // only the leaf kernels that address columns through the table read it
            if (!table_uploaded) h2d(ctx, data.col_table.p, cols.data(), cols.size() * sizeof(const uint32_t*));
            table_uploaded = true;
            data.col_table_uploaded = true;
        };
        auto group_mats = [&](const Group& g) {
            LeafMats lm;
            memset(&lm, 0, sizeof lm);
            lm.cols = data.col_table.p + g.first;
            lm.total_width = g.total;
            // one matrix -- or several lying back to back with one stride (commit() stores a batch of
            // equal-height matrices that way) -- lets the leaf kernel address the columns by stride
            const ColMat* first = nullptr;
            bool contiguous = true;
            uint32_t wsum = 0;
            for (auto& cm : data.ldes)
                if (cm.height == g.height) {
                    if (!first) first = &cm;
                    contiguous = contiguous && cm.col_stride == first->col_stride &&
                                 cm.d == first->d + (uint64_t)wsum * first->col_stride;
                    wsum += cm.width;
                }
            static const bool strided_knob = [] { const char* e = getenv("TS_LEAF_STRIDED"); return !e || atoi(e) != 0; }();
            if (!strided_knob) upload_table();  // TS_LEAF_STRIDED=0 sends even one matrix through the table kernels
            if (first && contiguous && wsum >= 1 && wsum <= 256) {
                lm.n_mats = 1;
                lm.d[0] = first->d;
                lm.col_stride[0] = first->col_stride;
                lm.width[0] = wsum;
            } else {
                upload_table();  // pointer-table leaf kernels (several scattered matrices, rows wider than 256)
            }
            return lm;
        };
        auto group_leaves = [&](const Group& g, uint32_t* digests) {
            launch_leaf_hash(ctx, group_mats(g), g.height, digests);
        };
        uint32_t* mail = nullptr;
        if (uniform) {  // leaves and every level in one launch (leaf_tree.hpp)
            // the kernel that makes the root writes it into the context's mailbox (host memory) as well:
            // no copy kernel between the tree and the host's next transcript step
            static const bool no_mail = [] { const char* e = getenv("TS_NO_MAILBOX"); return e && atoi(e) != 0; }();
            if (log_H >= 1 && !no_mail) mail = ctx.mailbox(8);
            launch_commit_tree(ctx, group_mats(groups[0]), log_H, data.tree.p, nullptr, mail, nullptr);
        } else {
            group_leaves(groups[0], data.tree.p);
            DevBuf<uint32_t> inj(&ctx, 8 * (N / 2));
            size_t gi = 1;
            for (unsigned l = 1; l <= log_H; l++) {
                uint32_t* children = data.tree.p + 8 * merkle_level_offset(log_H, l - 1);
                uint32_t* parents = data.tree.p + 8 * merkle_level_offset(log_H, l);
                const uint64_t n_par = N >> l;
                launch_merkle_one_level(ctx, children, parents, n_par);
                if (gi < groups.size() && groups[gi].height == n_par) {
                    group_leaves(groups[gi], inj.p);
                    launch_merkle_inject(ctx, parents, inj.p, n_par);
                    gi++;
                }
            }
        }
        if (mail) {
            ctx.sync_point(mail, 32);
            memcpy(data.root, mail, 32);
        } else {
            d2h_sync(ctx, data.root, data.tree.p + 8 * (merkle_total_digests(log_H) - 1), 32);
        }
    }
}

// ------------------------------------------------------------------ commit
// Matrices of different heights share one tree (basic/src/mmcs/bf_mmcs.rs:22-35 commits a mixed batch;
// the tree itself is build-defined, see merkle.hip): leaves hash the rows of the tallest matrices,
// and the rows of the matrices of height h are compressed into the level that has h nodes.
std::unique_ptr<PcsData> TwoAdicFriPcs::commit(std::vector<DeviceMatrix>& evals,
                                               const std::vector<uint32_t>& domain_shifts, bool build_tree) {
    TS_REQUIRE(!evals.empty() && evals.size() <= (size_t)MAX_BATCH_MATS, TS_ERR_INVALID,
               "commit: between 1 and MAX_BATCH_MATS (64) matrices per batch");
    TS_REQUIRE(evals.size() == domain_shifts.size(), TS_ERR_INVALID, "commit: one domain per matrix");
    uint64_t max_n = 0;
    for (auto& m : evals) {
        TS_REQUIRE(m.width >= 1 && m.buf.p, TS_ERR_INVALID, "commit: empty matrix");
        log2_strict(m.height);
        max_n = std::max(max_n, m.height);
    }
    const unsigned log_N = log2_strict(max_n) + fri_.log_blowup;
    TS_REQUIRE(log_N <= 27, TS_ERR_INVALID, "commit: LDE larger than the two-adic subgroup");
    ctx_.ensure_twiddles(std::max(1u, log_N));

    auto data = std::make_unique<PcsData>();
    data->log_height = log_N;
    {
        StageTimer t(&ctx_, "coset_lde");
        // A batch of equal-height matrices (the quotient chunks) gets ONE allocation, matrix after
        // matrix: to the leaf hash and to the opening's dot products it is then a single matrix of
        // the summed width (strided addressing, one launch) instead of a pointer table / a launch each.
        bool same_height = evals.size() > 1;
        size_t total_w = 0;
        for (auto& m : evals) {
            same_height = same_height && m.height == evals[0].height;
            total_w += m.width;
        }
        DevBuf<uint32_t> batch;
        if (same_height) batch = DevBuf<uint32_t>(&ctx_, total_w * (evals[0].height << fri_.log_blowup));
        size_t batch_col = 0;
        // exactly two column-major matrices of one shape (the two quotient chunks of a degree-3 AIR):
        // ONE set of LDE launches for both (coset_lde: evals2) -- 8 columns instead of 4 twice
        static const bool pair_knob = [] { const char* e = getenv("TS_LDE_PAIR"); return !e || atoi(e) != 0; }();
        const bool pair = pair_knob && same_height && evals.size() == 2 && evals[0].width == evals[1].width &&
                          evals[0].layout == DeviceMatrix::COL_MAJOR_BITREV &&
                          evals[1].layout == DeviceMatrix::COL_MAJOR_BITREV;
        if (pair) {
            for (size_t i = 0; i < 2; i++)
                TS_REQUIRE(domain_shifts[i] != 0 && domain_shifts[i] < P, TS_ERR_INVALID, "bad domain shift");
            const uint64_t n = evals[0].height;
            const unsigned log_n = log2_strict(n);
            const uint64_t Ni = n << fri_.log_blowup;
            const uint32_t w = evals[0].width;
            // two_adic_pcs.rs:235: shift = Val::generator() / domain.shift
            coset_lde(ctx_, evals[0].buf.p, n, 2 * w, log_n, fri_.log_blowup, mul(GENERATOR, inv_canon(domain_shifts[0])),
                      batch.p, Ni, 0, 0, false, evals[1].buf.p, mul(GENERATOR, inv_canon(domain_shifts[1])), w);
            for (size_t i = 0; i < 2; i++) {
                ColMat cm;
                cm.d = batch.p + i * (size_t)w * Ni;
                cm.height = Ni;
                cm.width = w;
                cm.col_stride = Ni;
                data->ldes.push_back(cm);
            }
            evals[0].buf.reset();  // consumed, both
            evals[1].buf.reset();
        }
        for (size_t i = 0; i < (pair ? 0 : evals.size()); i++) {
            DeviceMatrix& m = evals[i];
            TS_REQUIRE(domain_shifts[i] != 0 && domain_shifts[i] < P, TS_ERR_INVALID, "bad domain shift");
            const uint64_t n = m.height;
            const unsigned log_n = log2_strict(n);
            const uint64_t Ni = n << fri_.log_blowup;
            DevBuf<uint32_t> colmajor;
            uint32_t* ev = m.buf.p;
            bool r16 = false;  // the transpose already ran the first round of the inverse transform
            if (m.layout == DeviceMatrix::ROW_MAJOR) {
                colmajor = DevBuf<uint32_t>(&ctx_, (size_t)m.width * n);
                r16 = launch_transpose_bitrev_r16(ctx_, m.buf.p, colmajor.p, log_n, m.width, n);
                if (!r16) launch_transpose_bitrev(ctx_, m.buf.p, colmajor.p, log_n, m.width, n);
                ev = colmajor.p;
            }
            DevBuf<uint32_t> lde;
            uint32_t* lde_p;
            if (same_height) {
                lde_p = batch.p + batch_col * Ni;
                batch_col += m.width;
            } else {
                lde = DevBuf<uint32_t>(&ctx_, (size_t)m.width * Ni);
                lde_p = lde.p;
            }
            // two_adic_pcs.rs:235: shift = Val::generator() / domain.shift
            const uint32_t shift = mul(GENERATOR, inv_canon(domain_shifts[i]));
            coset_lde(ctx_, ev, n, m.width, log_n, fri_.log_blowup, shift, lde_p, Ni, 0, 0, r16);
            ColMat cm;
            cm.d = lde_p;
            cm.height = Ni;
            cm.width = m.width;
            cm.col_stride = Ni;
            data->ldes.push_back(cm);
            if (!same_height) data->lde_storage.push_back(std::move(lde));
            m.buf.reset();  // consumed
        }
        if (same_height) data->lde_storage.push_back(std::move(batch));
    }
    if (build_tree) mmcs_commit(ctx_, *data);
    return data;
}

// ------------------------------------------------------------------ quotient
std::vector<DeviceMatrix> TwoAdicFriPcs::quotient_chunks(const PcsData& trace_data,
                                                         const AirProgram& air,
                                                         const std::vector<uint32_t>& pis, Ef alpha) {
    TS_REQUIRE(trace_data.ldes.size() >= 1, TS_ERR_INVALID, "quotient: no trace matrix");
    TS_REQUIRE(trace_data.log_height >= fri_.log_blowup, TS_ERR_INVALID, "quotient: bad trace data");
    return quotient_chunks_slab(trace_data.ldes[0], trace_data.log_height - fri_.log_blowup, Slab{}, air,
                                pis, alpha);
}

std::vector<DeviceMatrix> TwoAdicFriPcs::quotient_chunks_slab(const ColMat& lde_slab, unsigned log_n,
                                                              const Slab& slab, const AirProgram& air,
                                                              const std::vector<uint32_t>& pis, Ef alpha,
                                                              uint32_t domain_shift) {
    StageTimer t(&ctx_, "compute quotient polynomial");
    TS_REQUIRE(lde_slab.width == air.width, TS_ERR_INVALID, "quotient: trace width != AIR width");
    TS_REQUIRE(pis.size() == air.n_public, TS_ERR_INVALID, "quotient: wrong number of public values");
    const unsigned lqd = air.log_quotient_degree;
    // two_adic_pcs.rs:256: assert!(lde.height() >= domain.size())
    TS_REQUIRE(lqd <= fri_.log_blowup, TS_ERR_INVARIANT,
               "quotient domain larger than the committed LDE (log_quotient_degree > log_blowup)");
    const uint64_t n = 1ull << log_n, qn = n << lqd;
    const uint32_t qd = 1u << lqd;
    // rows of the quotient domain (the first qn bit-reversed rows of the LDE) inside the slab
    const uint64_t slab_rows = slab.rows ? slab.rows : lde_slab.height;
    const uint64_t row_begin = std::min<uint64_t>(slab.row0, qn);
    const uint64_t row_end = std::min<uint64_t>(slab.row0 + slab_rows, qn);
    // whole cosets only: `next` (natural index + qd) stays inside a coset of H_n
    TS_REQUIRE(row_begin % n == 0 && row_end % n == 0, TS_ERR_INVALID, "quotient: slab must hold whole cosets");

    // selectors depend on the shape only: kept in the context between proofs (two tables, LRU)
    Context::SelTable* st = nullptr;
    for (auto& t : ctx_.sel_tables)
        if (t.d && t.log_n == log_n && t.log_qd == lqd && t.shift == domain_shift) st = &t;
    if (!st) {
        st = ctx_.sel_tables[0].last_use <= ctx_.sel_tables[1].last_use ? &ctx_.sel_tables[0] : &ctx_.sel_tables[1];
        if (st->d) {
            ctx_.sync();  // an earlier launch may still read the old table
            (void)hipFree(st->d);
            st->d = nullptr;
        }
        TS_HIP(hipMalloc((void**)&st->d, 3 * qn * sizeof(uint32_t)));
        launch_selectors(ctx_, log_n, lqd, st->d, st->d + qn, st->d + 2 * qn, domain_shift);
        st->log_n = log_n;
        st->log_qd = lqd;
        st->shift = domain_shift;
    }
    st->last_use = ++ctx_.sel_clock;
    struct { uint32_t* p; } sel{st->d};

    // constants / public values in Montgomery form
    std::vector<uint32_t> consts(std::max<size_t>(air.const_canonical.size(), 1), 0);
    for (size_t k = 0; k < air.const_canonical.size(); k++) {
        uint32_t v = air.const_public_idx[k] != ~0u ? pis[air.const_public_idx[k]] : air.const_canonical[k];
        TS_REQUIRE(v < P, TS_ERR_INVALID, "non-canonical public value");
        consts[k] = to_mont(v);
    }
    // alpha^(K-1-i): folder.rs:60-64 unrolled (acc = acc*alpha + c_i)
    const uint32_t K = air.n_constraints;
    std::vector<uint32_t> apow(std::max<size_t>(4 * (size_t)K, 4), 0);
    {
        Ef am = ef_to_mont(alpha), cur = ef_one_mont();
        for (uint32_t i = 0; i < K; i++) {
            memcpy(&apow[4 * (size_t)(K - 1 - i)], cur.c, 16);
            cur = ef_mul(cur, am);
        }
    }
    // one upload for both (a small host-to-device copy is a launch of its own on the stream)
    consts.resize((consts.size() + 3) & ~(size_t)3, 0);  // the powers stay 16-byte aligned
    const size_t n_consts = consts.size();
    consts.insert(consts.end(), apow.begin(), apow.end());
    DevBuf<uint32_t> d_consts(&ctx_, consts.size());
    h2d(ctx_, d_consts.p, consts.data(), consts.size() * 4);
    struct { uint32_t* p; } d_apow{d_consts.p + n_consts};

    std::vector<DeviceMatrix> chunks(qd);
    QuotOut qo;
    memset(&qo, 0, sizeof qo);
    for (uint32_t c = 0; c < qd; c++) {
        chunks[c].buf = DevBuf<uint32_t>(&ctx_, 4 * n);
        chunks[c].height = n;
        chunks[c].width = 4;
        chunks[c].layout = DeviceMatrix::COL_MAJOR_BITREV;
        qo.chunk[c] = chunks[c].buf.p;
    }
    if (row_begin < row_end) {
        ColMat lde = lde_slab;
        lde.d = lde_slab.d - slab.row0;  // global row r of the slab's range lives at d[r]
        launch_quotient(ctx_, air, lde, log_n, lqd, d_consts.p, d_apow.p, sel.p, sel.p + qn, sel.p + 2 * qn,
                        qo, row_begin, row_end, domain_shift);
    }
    return chunks;  // (the staged uploads live in the context's pinned arena: no sync needed)
}

// ------------------------------------------------------------------ open
DevBuf<Ef> TwoAdicFriPcs::open_reduce(const PcsData& trace_data, const PcsData& quotient_data, Ef zeta,
                                      Ef alpha, std::vector<Ef>& opened_values) {
    TS_REQUIRE(trace_data.log_height == quotient_data.log_height, TS_ERR_INVALID,
               "open: trace and quotient LDE heights differ");
    return open_reduce_slab(trace_data, quotient_data, trace_data.log_height, Slab{}, zeta, alpha,
                            opened_values);
}

// `slab` (rows != 0): the two PcsData hold only global rows [row0, row0 + rows) of the LDEs, which
// start with the whole coset beta0 (sharded prover).  The opened values are interpolated on that
// coset -- any coset of the LDE determines the polynomials, so every rank gets the same values.
DevBuf<Ef> TwoAdicFriPcs::open_reduce_slab(const PcsData& trace_data, const PcsData& quotient_data,
                                           unsigned log_N, const Slab& slab, Ef zeta, Ef alpha,
                                           std::vector<Ef>& opened_values) {
    TS_REQUIRE(trace_data.ldes.size() == 1, TS_ERR_UNSUPPORTED, "open: one trace matrix expected");
    const unsigned log_n = log_N - fri_.log_blowup;
    const uint64_t n = 1ull << log_n;
    const uint64_t N = slab.rows ? slab.rows : 1ull << log_N;  // rows held here
    TS_REQUIRE(N >= n && trace_data.ldes[0].height == N, TS_ERR_INVALID, "open: slab shape");
    // x of local row t < n: coset_gen * omega_n^bitrev(t), coset_gen = 31 * omega_N^bitrev_b(beta0)
    const uint32_t coset_gen =
        mul(GENERATOR, pow_canon(two_adic_generator(log_N), bitrev32(slab.beta0, fri_.log_blowup)));
    const ColMat& tr = trace_data.ldes[0];
    const uint32_t w = tr.width;
    const uint32_t qd = (uint32_t)quotient_data.ldes.size();
    for (auto& m : quotient_data.ldes)
        TS_REQUIRE(m.width == 4, TS_ERR_INVALID, "open: quotient chunks must have width 4");
    ctx_.ensure_twiddles(std::max(1u, log_N));

    // prover.rs:92 zeta_next = trace_domain.next_point(zeta) = zeta * omega_n
    const Ef zeta_next = efc_mul_base(zeta, two_adic_generator(log_n));
    const Ef pts_mont[2] = {ef_to_mont(zeta), ef_to_mont(zeta_next)};

    // ---- opened values: barycentric interpolation on the low coset (two_adic_pcs.rs:358-369)
    std::vector<Ef> raw(2 * (size_t)w + 4 * (size_t)qd);
    {
        StageTimer t(&ctx_, "compute opened values with Lagrange interpolation");
        DevBuf<Ef> weights(&ctx_, 2 * n);
        launch_bary_weights(ctx_, log_n, pts_mont, 2, weights.p, coset_gen);
        // the last kernels of the stage write the sums straight into the context's mailbox (host memory)
        struct { Ef* p; } sums{reinterpret_cast<Ef*>(ctx_.mailbox(4 * raw.size()))};
        BaryPending pend;  // the trace's partial sums and the chunks' are added up in one launch
        launch_bary_dots(ctx_, tr, log_n, weights.p, 2, sums.p, &pend);  // [col][point]
        bool chunks_contiguous = true;  // commit() lays the chunk LDEs back to back
        for (uint32_t c = 0; c < qd; c++)
            chunks_contiguous = chunks_contiguous && quotient_data.ldes[c].col_stride == quotient_data.ldes[0].col_stride &&
                                quotient_data.ldes[c].d == quotient_data.ldes[0].d + (uint64_t)4 * c * quotient_data.ldes[0].col_stride;
        if (chunks_contiguous && qd > 1) {
            ColMat all = quotient_data.ldes[0];
            all.width = 4 * qd;
            launch_bary_dots(ctx_, all, log_n, weights.p, 1, sums.p + 2 * w, &pend);
        } else {
            for (uint32_t c = 0; c < qd; c++)
                launch_bary_dots(ctx_, quotient_data.ldes[c], log_n, weights.p, 1, sums.p + 2 * w + 4 * c, &pend);
        }
        launch_bary_finish(ctx_, pend);
        ctx_.sync_point(sums.p, raw.size() * sizeof(Ef));
        memcpy(raw.data(), sums.p, raw.size() * sizeof(Ef));
    }
    // p(z) = ((z/s)^n - 1)/n * sum_i p_i x_i/(z - x_i) on the coset s*H_n (s = 31 unless sharded)
    const uint32_t gen_inv = inv_canon(coset_gen);
    const uint32_t n_inv = inv_canon((uint32_t)(n % P));
    Ef scale[2];
    for (int p = 0; p < 2; p++) {
        Ef u = efc_mul_base(p == 0 ? zeta : zeta_next, gen_inv);
        Ef un = efc_pow(u, n);
        un.c[0] = sub(un.c[0], 1);
        scale[p] = efc_mul_base(un, n_inv);
    }
    opened_values.assign(2 * (size_t)w + 4 * (size_t)qd, ef_zero());
    for (uint32_t c = 0; c < w; c++) {
        opened_values[c] = efc_mul(raw[2 * c], scale[0]);          // trace_local
        opened_values[w + c] = efc_mul(raw[2 * c + 1], scale[1]);  // trace_next
    }
    for (uint32_t k = 0; k < 4 * qd; k++) opened_values[2 * w + k] = efc_mul(raw[2 * w + k], scale[0]);

    // ---- reduce (two_adic_pcs.rs:371-383)
    StageTimer t(&ctx_, "reduce rows");
    const uint32_t max_w = std::max(w, 4u);
    std::vector<uint32_t> apow(4 * (size_t)max_w);
    const Ef am = ef_to_mont(alpha);
    {
        Ef cur = ef_one_mont();
        for (uint32_t i = 0; i < max_w; i++) {
            memcpy(&apow[4 * (size_t)i], cur.c, 16);
            cur = ef_mul(cur, am);
        }
    }
    // (uploaded below, together with the chunk weights: one copy on the stream instead of two)
    auto reduced_ys = [&](const Ef* ys, uint32_t width) {  // dot_product(alpha.powers(), ys), :372
        Ef acc = ef_zero();
        for (uint32_t i = 0; i < width; i++) {
            Ef ap;
            memcpy(ap.c, &apow[4 * (size_t)i], 16);
            acc = ef_add(acc, ef_mul(ys[i], ap));  // canonical x Montgomery -> canonical
        }
        return acc;
    };
    DevBuf<Ef> ro(&ctx_, N);
    {
        // offsets follow two_adic_pcs.rs:371,383: num_reduced grows by the width after every
        // (matrix, point); all matrices share log_height here
        TS_REQUIRE(qd <= (uint32_t)MAX_QUOTIENT_CHUNKS, TS_ERR_UNSUPPORTED, "open: more than 64 quotient chunks");
        FusedReduceArgs a;
        memset(&a, 0, sizeof a);
        a.z_mont[0] = pts_mont[0];
        a.z_mont[1] = pts_mont[1];
        uint64_t num_reduced = 0;
        a.off_t[0] = ef_pow(am, num_reduced);
        a.k0 = ef_mul(reduced_ys(&opened_values[0], w), a.off_t[0]);  // canonical x Montgomery -> canonical
        num_reduced += w;
        a.off_t[1] = ef_pow(am, num_reduced);
        a.k1 = ef_mul(reduced_ys(&opened_values[w], w), a.off_t[1]);
        num_reduced += w;
        a.n_chunks = qd;
        a.chunk_stride = N;
        a.row0 = slab.row0;
        a.rows = N;
        std::vector<uint32_t> cw(16 * (size_t)qd);
        for (uint32_t c = 0; c < qd; c++) {
            TS_REQUIRE(quotient_data.ldes[c].col_stride == N, TS_ERR_INVALID, "open: chunk stride");
            a.chunk[c] = quotient_data.ldes[c].d;
            const Ef off_c = ef_pow(am, num_reduced);
            a.k0 = ef_add(a.k0, ef_mul(reduced_ys(&opened_values[2 * w + 4 * c], 4), off_c));
            for (int k = 0; k < 4; k++) {  // alpha^k off_c: the weight of column k of chunk c
                Ef ap;
                memcpy(ap.c, &apow[4 * (size_t)k], 16);
                const Ef wk = ef_mul(ap, off_c);
                memcpy(&cw[16 * (size_t)c + 4 * k], wk.c, 16);
            }
            num_reduced += 4;
        }
        const size_t n_apow = apow.size();
        apow.insert(apow.end(), cw.begin(), cw.end());
        DevBuf<uint32_t> d_apow(&ctx_, apow.size());
        h2d(ctx_, d_apow.p, apow.data(), apow.size() * 4);
        a.chunk_w = d_apow.p + n_apow;
        launch_reduce_fused(ctx_, tr, log_N, d_apow.p, a, ro.p);
    }
    return ro;
}

// ------------------------------------------------------------------ open_batch
void TwoAdicFriPcs::open_batch(const PcsData& d, uint64_t index, std::vector<uint32_t>& rows,
                               std::vector<uint32_t>& path) {
    TS_REQUIRE(index < (1ull << d.log_height), TS_ERR_INVALID, "open_batch: index out of range");
    LeafMats lm = d.leaf_mats();
    DevBuf<uint32_t> d_idx(&ctx_, 1), d_rows(&ctx_, std::max(lm.total_width, 1u)),
        d_path(&ctx_, std::max(8u * d.log_height, 8u));
    uint32_t idx32 = (uint32_t)index;
    h2d(ctx_, d_idx.p, &idx32, 4);
    launch_gather_rows(ctx_, lm, d_idx.p, 1, 0, d_rows.p);
    launch_gather_paths(ctx_, d.tree.p, d.log_height, d_idx.p, 1, 0, d_path.p);
    rows.resize(lm.total_width);
    path.resize(8 * (size_t)d.log_height);
    if (!rows.empty()) TS_HIP(hipMemcpyAsync(rows.data(), d_rows.p, rows.size() * 4, hipMemcpyDeviceToHost, ctx_.stream));
    if (!path.empty()) TS_HIP(hipMemcpyAsync(path.data(), d_path.p, path.size() * 4, hipMemcpyDeviceToHost, ctx_.stream));
    ctx_.sync();
}

// ------------------------------------------------------------------ bf_commit_phase
// fri/src/prover.rs:93-141.  The transcript moves to the device for the whole phase: per round the
// kernel that makes the root observes it and samples beta, the fold reads beta from device memory,
// and once the vector is short (and no further input is waiting to be added) the remaining rounds
// run inside one workgroup (launch_fri_tail).  One D2H at the end brings back roots, the final
// values and the challenger state.
void fri_commit_begin(Context& ctx, const FriConfig& fri, unsigned log_max_height,
                      const BfChallenger& challenger, FriCommit& st) {
    TS_REQUIRE(log_max_height >= fri.log_blowup, TS_ERR_INVALID, "FRI: vector shorter than the blowup");
    st.R_total = log_max_height - fri.log_blowup;
    DevChallenger hc;
    challenger.export_dev(hc);
    static_assert(sizeof(DevChallenger) <= 64 * 4, "the challenger's slot in the block");
    const size_t n_roots = std::max<size_t>(8 * (size_t)st.R_total, 8);
    st.d_block = DevBuf<uint32_t>(&ctx, 64 + n_roots + 4 * (size_t)fri.blowup());
    st.d_chal.p = st.d_block.p;
    st.d_roots.p = st.d_block.p + 64;
    st.d_final.p = reinterpret_cast<Ef*>(st.d_block.p + 64 + n_roots);  // 16-byte aligned: 64 + 8 R words
    uint32_t slot[64] = {0};
    static_assert(sizeof(DevChallenger) <= FRI_POW_WORD * 4, "the hint word lies behind the challenger");
    memcpy(slot, &hc, sizeof hc);
    slot[FRI_POW_WORD] = FRI_POW_NONE;
    h2d(ctx, st.d_chal.p, slot, sizeof slot);
    st.d_betas = DevBuf<Ef>(&ctx, std::max<size_t>(st.R_total, 1));
}

void fri_commit_rounds(Context& ctx, const FriConfig& fri, DevBuf<Ef> folded, uint64_t len,
                       std::vector<DevBuf<Ef>>& inputs, const std::vector<unsigned>& log_lens,
                       size_t next_in, FriCommit& st) {
    DevChallenger* dch = st.dch();
    // != nullptr: `folded` is not in memory yet -- it is the fold of this vector (the last round's)
    // with the last round's challenge, and the next round's kernel computes it while hashing
    const Ef* prev = nullptr;
    // rounds done with one launch each; the rest goes to the tail kernel
    auto big = [&](uint64_t l) {
        return l > fri.blowup() && (l > (1ull << FRI_TAIL_LOG) || next_in < inputs.size());
    };
    while (big(len)) {  // :111
        FriRound r;
        const uint64_t h = len / 2;
        r.log_leaves = log2_strict(h);
        const size_t ri = st.rounds.size();
        DevBuf<uint32_t> tree(&ctx, merkle_total_digests(r.log_leaves) * 8);
        if (prev) folded = DevBuf<Ef>(&ctx, len);
        // :113 commit_matrix, :114-116 observe + sample (in the kernel that makes the root)
        if (fri_round_max_log() != 0 && r.log_leaves <= fri_round_max_log()) {
            launch_fri_round(ctx, prev, prev ? st.d_betas.p + ri - 1 : nullptr, folded.p, h, tree.p, dch,
                             st.d_roots.p + 8 * ri, st.d_betas.p + ri);
        } else {  // tall rounds: fold + leaves + tree in one launch (leaf_tree.hpp)
            if (!launch_fri_round_tall(ctx, prev, prev ? st.d_betas.p + ri - 1 : nullptr, folded.p, h, tree.p, dch,
                                       st.d_roots.p + 8 * ri, st.d_betas.p + ri))
                launch_chal_round(ctx, dch, tree.p + 8 * (merkle_total_digests(r.log_leaves) - 1),
                                  st.d_roots.p + 8 * ri, st.d_betas.p + ri);
        }
        prev = nullptr;
        r.vec = folded.p;
        r.tree = tree.p;
        const bool add_pending = next_in < inputs.size() && (1ull << log_lens[next_in]) == h;
        // the tail kernel folds its own first vector (one launch less), unless an input joins it first
        const bool tail_folds = !add_pending && !big(h) && h > fri.blowup() && next_in >= inputs.size();
        if (tail_folds) {
            prev = folded.p;
            st.keep_vecs.push_back(std::move(folded));
        } else if (add_pending || !big(h)) {
            // the next vector is needed in memory now: an input is added to it, or the tail takes it
            DevBuf<Ef> out(&ctx, h);
            launch_fri_fold_dev(ctx, folded.p, h, st.d_betas.p + ri, out.p, nullptr);  // :119 fold_matrix
            if (add_pending) {  // :124-126 izip!(&mut folded, v).for_each(|(c, x)| *c += x)
                launch_vec_add(ctx, out.p, inputs[next_in].p, h);
                st.keep_vecs.push_back(std::move(inputs[next_in]));
                next_in++;
            }
            st.keep_vecs.push_back(std::move(folded));
            folded = std::move(out);
        } else {
            prev = folded.p;  // :119 happens inside the next round's launch
            st.keep_vecs.push_back(std::move(folded));
        }
        st.keep_trees.push_back(std::move(tree));
        st.rounds.push_back(r);
        len = h;
    }
    TS_REQUIRE(next_in == inputs.size(), TS_ERR_INVARIANT, "FRI: an input was never folded in");
    if (len > fri.blowup()) {  // tail rounds in one workgroup
        const uint32_t L0 = (uint32_t)len;
        DevBuf<Ef> tail_vecs(&ctx, 2 * (size_t)L0);
        DevBuf<uint32_t> tail_trees(&ctx, 8 * 2 * (size_t)L0);
        const size_t ri = st.rounds.size();
        static const bool host_grind = [] { const char* e = getenv("TS_HOST_GRIND"); return e && atoi(e) != 0; }();
        launch_fri_tail(ctx, prev ? prev : folded.p, L0, fri.blowup(), dch, tail_vecs.p, tail_trees.p,
                        st.d_roots.p + 8 * ri, st.d_betas.p + ri, st.d_final.p, fri.proof_of_work_bits,
                        host_grind ? nullptr : st.d_chal.p + FRI_POW_WORD, prev ? st.d_betas.p + ri - 1 : nullptr);
        uint32_t L = L0;
        size_t voff = 0, toff = 0;
        while (L > fri.blowup()) {
            FriRound r;
            r.log_leaves = log2_strict(L / 2);
            r.vec = tail_vecs.p + voff;
            r.tree = tail_trees.p + 8 * toff;
            st.rounds.push_back(r);
            voff += L;
            toff += L - 1;
            L >>= 1;
        }
        len = L;
        st.keep_vecs.push_back(std::move(tail_vecs));
        st.keep_trees.push_back(std::move(tail_trees));
    } else {
        TS_HIP(hipMemcpyAsync(st.d_final.p, folded.p, len * sizeof(Ef), hipMemcpyDeviceToDevice,
                              ctx.stream));
    }
    st.keep_vecs.push_back(std::move(folded));
    st.final_len = len;
}

Ef fri_commit_finish(Context& ctx, const FriConfig& fri, BfChallenger& challenger, FriCommit& st) {
    // :129-134 `blowup` evaluations of a constant polynomial
    TS_REQUIRE(st.final_len == fri.blowup(), TS_ERR_INVARIANT, "FRI: folded length != blowup");
    TS_REQUIRE(st.rounds.size() == st.R_total, TS_ERR_INVARIANT, "FRI: round count");
    const uint32_t R_total = st.R_total;
    std::vector<Ef> fin(st.final_len);
    std::vector<uint32_t> roots(std::max<size_t>(8 * (size_t)R_total, 8));
    DevChallenger hc;
    std::vector<uint32_t> block(st.d_block.n);
    ctx.d2h_point(block.data(), st.d_block.p, block.size() * 4);
    memcpy(&hc, block.data(), sizeof hc);
    st.pow_hint = block[FRI_POW_WORD];
    memcpy(roots.data(), block.data() + 64, roots.size() * 4);
    memcpy(fin.data(), block.data() + 64 + roots.size(), fin.size() * sizeof(Ef));
    challenger.import_dev(hc);
    for (uint32_t r = 0; r < R_total; r++) memcpy(st.rounds[r].root, &roots[8 * (size_t)r], 32);
    const Ef final_poly = fin[0];
    for (auto& x : fin)
        if (!ef_eq(x, final_poly))
            throw FinalPolyNotConstant("FRI: final polynomial is not constant (assert_eq!(x, final_poly))");
    return final_poly;
}

uint32_t fri_pow_witness(Context& ctx, BfChallenger& challenger, unsigned bits, const FriCommit& st) {
    if (st.pow_hint < (1u << 12)) {
        BfChallenger clone = challenger;
        if (clone.check_witness(bits, st.pow_hint)) {
            challenger = clone;
            ctx.pow_hints_accepted++;
            return st.pow_hint;
        }
        ctx.pow_hints_rejected++;  // never expected: the device search and the host sponge disagree
    }
    ctx.pow_host_grinds++;
    return challenger.grind(bits);
}

// ------------------------------------------------------------------ bf_prove
// fri/src/prover.rs:19-141 (bf_prove, bf_commit_phase, bf_answer_query) with the open_input closure
// of two_adic_pcs.rs:399-414.  `inputs` are the reduced openings by strictly descending height.
// Appends the FriProof to `pf`.
void TwoAdicFriPcs::fri_prove(std::vector<DevBuf<Ef>>& inputs, const std::vector<unsigned>& log_lens,
                              BfChallenger& challenger,
                              const std::vector<const PcsData*>& input_rounds,
                              std::vector<uint32_t>& pf, bool pass_through) {
    Context& ctx = ctx_;
    const FriConfig& fri = fri_;
    TS_REQUIRE(!inputs.empty() && inputs.size() == log_lens.size(), TS_ERR_INVALID, "FRI: no input");
    // pass-through input proof (fri/tests/fri.rs:109-118): the literal reduced openings; the input
    // vectors stay alive (and unmodified) in the commit state until the queries are answered
    std::vector<const Ef*> in_ptr;
    for (auto& v : inputs) in_ptr.push_back(v.p);
    if (pass_through)
        TS_REQUIRE(input_rounds.empty() && log_lens.back() >= 1, TS_ERR_INVALID,
                   "FRI: pass-through input proof takes no committed batches");
    for (size_t k = 1; k < log_lens.size(); k++)
        TS_REQUIRE(log_lens[k] < log_lens[k - 1], TS_ERR_INVALID, "FRI: inputs must descend in height");
    const unsigned log_max_height = log_lens[0];  // prover.rs:30
    TS_REQUIRE(log_lens.back() >= fri.log_blowup, TS_ERR_INVALID, "FRI: vector shorter than the blowup");
    for (const PcsData* d : input_rounds)
        TS_REQUIRE(d->log_height <= log_max_height, TS_ERR_INVALID,
                   "FRI: a committed batch is taller than every opened matrix");

    FriCommit st;
    Ef final_poly;
    {
        StageTimer t(&ctx, "FRI commit phase");
        fri_commit_begin(ctx, fri, log_max_height, challenger, st);
        DevBuf<Ef> first = std::move(inputs[0]);
        // TS_FRI_GRAPH=1 (measurement knob, DESIGN.md "hipGraph"): the commit phase -- the launch-bound
        // loop of the path, ~20 dependent launches with no host interaction -- is captured into a
        // hipGraph and replayed; the instantiated graph is kept per context and updated in place
        // (hipGraphExecUpdate) when a proof of the same shape comes with other buffer addresses.
        // A capture cannot hipMalloc, so: the first proof of a SHAPE (log_blowup + every input height:
        // what the block sizes depend on) runs eagerly and records the sizes the phase allocates;
        // before a capture the pool is made to hold all of them at once (Context::reserve); and an
        // allocation that still misses inside the capture (Context::CaptureMiss: no HIP call was made)
        // ends the capture and re-runs the phase eagerly from the untouched inputs.
        // TS_FRI_GRAPH=2 skips the reservation: the test hook that exercises that fall-back; =3 treats the
        // replay as failed after a good capture (instantiate / launch failure: the same fall-back).
        const char* genv = getenv("TS_FRI_GRAPH");
        const int want_graph = (genv && !ctx.timing && !ctx.kernel_timing) ? atoi(genv) : 0;
        const uint64_t len0 = 1ull << log_max_height;
        bool done = false;
        if (want_graph) {
            std::vector<uint32_t> key{fri.log_blowup};
            for (unsigned l : log_lens) key.push_back(l);
            auto known = ctx.fri_graph_sizes.find(key);
            if (known == ctx.fri_graph_sizes.end()) {  // first proof of this shape: eager, recording
                std::vector<size_t> sizes;
                ctx.alloc_log = &sizes;
                try {
                    fri_commit_rounds(ctx, fri, std::move(first), len0, inputs, log_lens, 1, st);
                } catch (...) {
                    ctx.alloc_log = nullptr;
                    throw;
                }
                ctx.alloc_log = nullptr;
                ctx.fri_graph_sizes[key] = std::move(sizes);
                done = true;
            } else if (want_graph != 2 && !ctx.reserve(known->second)) {
                ctx.fri_graph_reserve_failures++;  // hipMalloc refused the reservation: eager, and on the record
            } else {
                hipGraph_t g = nullptr;
                bool miss = false;
                TS_HIP(hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeThreadLocal));
                ctx.capturing = true;
                try {
                    fri_commit_rounds(ctx, fri, std::move(first), len0, inputs, log_lens, 1, st);
                } catch (Context::CaptureMiss&) {
                    miss = true;
                } catch (...) {
                    (void)hipStreamEndCapture(ctx.stream, &g);
                    if (g) (void)hipGraphDestroy(g);
                    ctx.capturing = false;
                    ctx.flush_deferred({});
                    throw;
                }
                const hipError_t ec = hipStreamEndCapture(ctx.stream, &g);
                // Nothing captured has run yet.  A capture miss, a failed end-of-capture, and a graph
                // that cannot be instantiated / updated / launched (e.g. out of memory on a new shape)
                // all take the same way out: drop the graph and the half-built round state while frees
                // are still parked, take the input vectors back (the rounds had moved some of them into
                // the state), return every other parked block to the pool, then run eagerly.
                bool replayed = false;
                if (!miss && ec == hipSuccess) {
                    bool ready = false;
                    if (ctx.fri_graph_exec) {
                        hipGraphNode_t err_node = nullptr;
                        hipGraphExecUpdateResult res;
                        ready = hipGraphExecUpdate(ctx.fri_graph_exec, g, &err_node, &res) == hipSuccess &&
                                res == hipGraphExecUpdateSuccess;
                        if (!ready) {
                            (void)hipGetLastError();
                            (void)hipGraphExecDestroy(ctx.fri_graph_exec);
                            ctx.fri_graph_exec = nullptr;
                        }
                    }
                    if (!ready) {
                        ready = hipGraphInstantiate(&ctx.fri_graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
                        if (!ready) ctx.fri_graph_exec = nullptr;
                    }
                    // TS_FRI_GRAPH=3: the test hook for this branch (pretend the launch failed)
                    if (ready && want_graph != 3)
                        replayed = hipGraphLaunch(ctx.fri_graph_exec, ctx.stream) == hipSuccess;
                }
                if (g) (void)hipGraphDestroy(g);
                if (replayed) {
                    ctx.capturing = false;
                    ctx.flush_deferred({});
                    ctx.fri_graph_replays++;
                    done = true;
                } else {
                    (void)hipGetLastError();
                    st.rounds.clear();
                    st.keep_vecs.clear();
                    st.keep_trees.clear();
                    std::vector<void*> mine;
                    for (const Ef* q : in_ptr) mine.push_back(const_cast<Ef*>(q));
                    ctx.capturing = false;
                    const std::vector<void*> back = ctx.flush_deferred(mine);
                    for (void* q : back)
                        for (size_t k = 0; k < in_ptr.size(); k++)
                            if (q == (const void*)in_ptr[k]) {
                                DevBuf<Ef> b = DevBuf<Ef>::adopt(&ctx, static_cast<Ef*>(q), 1ull << log_lens[k]);
                                if (k == 0) first = std::move(b);
                                else inputs[k] = std::move(b);
                            }
                    ctx.fri_graph_fallbacks++;
                }
            }
        }
        if (!done) fri_commit_rounds(ctx, fri, std::move(first), len0, inputs, log_lens, 1, st);
        final_poly = fri_commit_finish(ctx, fri, challenger, st);
    }
    std::vector<FriRound>& rounds = st.rounds;
    const uint32_t R = (uint32_t)rounds.size();

    // :43 proof of work
    uint32_t pow_witness;
    {
        StageTimer t(&ctx, "grind for proof-of-work witness");
        pow_witness = fri_pow_witness(ctx, challenger, fri.proof_of_work_bits, st);
    }

    // ---- query phase :45-59
    StageTimer tq(&ctx, "query phase");
    const uint32_t Q = fri.num_queries;
    std::vector<uint32_t> indices(std::max(Q, 1u));
    for (uint32_t q = 0; q < Q; q++) indices[q] = (uint32_t)challenger.sample_bits(log_max_height);
    // (uploaded below together with the FRI gather descriptors: one copy on the stream)

    // gather everything into one buffer, one D2H
    const size_t n_in_rounds = input_rounds.size();
    std::vector<LeafMats> lms(n_in_rounds);
    std::vector<size_t> o_rows(n_in_rounds), o_path(n_in_rounds);
    size_t off = 0;
    for (size_t k = 0; k < n_in_rounds; k++) {
        lms[k] = input_rounds[k]->leaf_mats();
        o_rows[k] = off; off += (size_t)Q * lms[k].total_width;
        o_path[k] = off; off += (size_t)Q * 8 * input_rounds[k]->log_height;
    }
    std::vector<size_t> o_fvals(R), o_fpath(R);
    for (uint32_t r = 0; r < R; r++) {
        o_fvals[r] = off; off += (size_t)Q * 8;
        o_fpath[r] = off; off += (size_t)Q * 8 * rounds[r].log_leaves;
    }
    std::vector<size_t> o_pass(pass_through ? in_ptr.size() : 0);
    for (size_t k = 0; k < o_pass.size(); k++) {
        o_pass[k] = off; off += (size_t)Q * 8;
    }
    // The whole query phase is ONE launch (launch_gather_queries): the opened rows of every committed batch
    // (two_adic_pcs.rs:403-409: bits_reduced = log_global_max_height - log_max_height(batch)), and as
    // descriptors the batches' Merkle paths (no values), bf_answer_query :69-90 for every commit round
    // (index_i = index >> i >> 1) and the pass-through inputs (values only: the pair holding element
    // index >> shift).
    std::vector<RowGatherJob> rjobs(n_in_rounds);
    uint32_t max_row_w = 0;
    for (size_t k = 0; k < n_in_rounds; k++) {
        rjobs[k].mats = lms[k];
        rjobs[k].shift = log_max_height - input_rounds[k]->log_height;
        rjobs[k].pad = 0;
        rjobs[k].out = o_rows[k];
        max_row_w = std::max(max_row_w, lms[k].total_width);
    }
    std::vector<FriGatherDesc> descs;
    uint32_t max_ll = 0;
    for (size_t k = 0; k < n_in_rounds; k++) {
        FriGatherDesc d{};
        d.vec = nullptr;
        d.tree = input_rounds[k]->tree.p;
        d.log_leaves = input_rounds[k]->log_height;
        d.shift = log_max_height - input_rounds[k]->log_height;
        d.out_path = o_path[k];
        descs.push_back(d);
        max_ll = std::max(max_ll, d.log_leaves);
    }
    for (uint32_t r = 0; r < R; r++) {
        FriGatherDesc d{};
        d.vec = reinterpret_cast<const uint32_t*>(rounds[r].vec);
        d.tree = rounds[r].tree;
        d.log_leaves = rounds[r].log_leaves;
        d.shift = r + 1;
        d.out_vals = o_fvals[r];
        d.out_path = o_fpath[r];
        descs.push_back(d);
        max_ll = std::max(max_ll, d.log_leaves);
    }
    for (size_t k = 0; k < o_pass.size(); k++) {
        FriGatherDesc d{};
        d.vec = reinterpret_cast<const uint32_t*>(in_ptr[k]);
        d.log_leaves = 0;
        d.shift = log_max_height - log_lens[k] + 1;
        d.out_vals = o_pass[k];
        descs.push_back(d);
    }
    // row jobs, descriptors, then the indices, in ONE upload
    const size_t b_rows = rjobs.size() * sizeof(RowGatherJob), b_descs = descs.size() * sizeof(FriGatherDesc);
    static_assert(sizeof(RowGatherJob) % 8 == 0 && sizeof(FriGatherDesc) % 8 == 0, "tables stay 8-byte aligned");
    std::vector<unsigned char> up(b_rows + b_descs + indices.size() * 4);
    if (b_rows) memcpy(up.data(), rjobs.data(), b_rows);
    if (b_descs) memcpy(up.data() + b_rows, descs.data(), b_descs);
    memcpy(up.data() + b_rows + b_descs, indices.data(), indices.size() * 4);
    DevBuf<unsigned char> d_up(&ctx, up.size());
    h2d(ctx, d_up.p, up.data(), up.size());
    DevBuf<uint32_t> d_out(&ctx, std::max<size_t>(off, 1));
    launch_gather_queries(ctx, reinterpret_cast<const RowGatherJob*>(d_up.p), (uint32_t)rjobs.size(), max_row_w,
                          reinterpret_cast<const FriGatherDesc*>(d_up.p + b_rows), (uint32_t)descs.size(), max_ll,
                          reinterpret_cast<const uint32_t*>(d_up.p + b_rows + b_descs), Q, d_out.p);
    std::vector<uint32_t> g(std::max<size_t>(off, 1));
    d2h_sync(ctx, g.data(), d_out.p, off * 4);

    // ---- FriProof (fri/src/proof.rs) in TSPF v1 order
    pf.reserve(pf.size() + 16 + off + (size_t)Q * (8 + 2 * R + 4 * n_in_rounds));
    auto push = [&](uint32_t v) { pf.push_back(v); };
    auto push_n = [&](const uint32_t* p, size_t k) { pf.insert(pf.end(), p, p + k); };
    push(R);
    for (uint32_t r = 0; r < R; r++) push_n(rounds[r].root, 8);
    push(Q);
    for (uint32_t q = 0; q < Q; q++) {
        if (pass_through) {  // fri.rs:109-118: [(log_height, value)] by descending height
            push((uint32_t)in_ptr.size());
            for (size_t k = 0; k < in_ptr.size(); k++) {
                push(log_lens[k]);
                const uint32_t half = (indices[q] >> (log_max_height - log_lens[k])) & 1;
                push_n(&g[o_pass[k] + (size_t)q * 8 + 4 * half], 4);
            }
        } else {
            push((uint32_t)n_in_rounds);  // input_proof: one BatchOpening per commit round
        }
        for (size_t k = 0; k < n_in_rounds; k++) {
            const LeafMats& lm = lms[k];
            push(lm.n_mats);
            size_t c = o_rows[k] + (size_t)q * lm.total_width;
            for (uint32_t i = 0; i < lm.n_mats; i++) {
                push(lm.width[i]);
                push_n(&g[c], lm.width[i]);
                c += lm.width[i];
            }
            const unsigned lh = input_rounds[k]->log_height;
            push(lh);
            push_n(&g[o_path[k] + (size_t)q * 8 * lh], 8 * (size_t)lh);
        }
        for (uint32_t r = 0; r < R; r++) {  // commit_phase_openings
            push_n(&g[o_fvals[r] + (size_t)q * 8], 8);
            push(rounds[r].log_leaves);
            push_n(&g[o_fpath[r] + (size_t)q * 8 * rounds[r].log_leaves], 8 * (size_t)rounds[r].log_leaves);
        }
    }
    push_n(final_poly.c, 4);
    push(pow_witness);
}

// ------------------------------------------------------------------ Pcs::open, any shape
// two_adic_pcs.rs:260-419.  Opened values come back in (round, matrix, point, column) order.
std::vector<uint32_t> TwoAdicFriPcs::open(const std::vector<OpenRound>& rounds, BfChallenger& challenger,
                                          std::vector<Ef>& opened_values) {
    TS_REQUIRE(!rounds.empty(), TS_ERR_INVALID, "open: no rounds");
    const Ef alpha = challenger.sample();  // :312
    const Ef am = ef_to_mont(alpha);
    uint32_t max_w = 1;
    unsigned log_global_max = 0;
    for (auto& r : rounds) {
        TS_REQUIRE(r.data && r.points.size() == r.data->ldes.size(), TS_ERR_INVALID,
                   "open: one point list per committed matrix");
        for (auto& m : r.data->ldes) max_w = std::max(max_w, m.width);
        log_global_max = std::max(log_global_max, r.data->log_height);  // :319-326
    }
    ctx_.ensure_twiddles(std::max(1u, log_global_max));
    std::vector<uint32_t> apow(4 * (size_t)max_w);
    {
        Ef cur = ef_one_mont();
        for (uint32_t i = 0; i < max_w; i++) {
            memcpy(&apow[4 * (size_t)i], cur.c, 16);
            cur = ef_mul(cur, am);
        }
    }
    DevBuf<uint32_t> d_apow(&ctx_, apow.size());
    h2d(ctx_, d_apow.p, apow.data(), apow.size() * 4);

    opened_values.clear();
    DevBuf<Ef> ro[32];            // :331 reduced_openings by log_height
    uint64_t num_reduced[32] = {0};  // :332
    const uint32_t gen_inv = inv_canon(GENERATOR);
    for (auto& r : rounds) {
        for (size_t mi = 0; mi < r.data->ldes.size(); mi++) {
            const ColMat& m = r.data->ldes[mi];
            const unsigned log_h = log2_strict(m.height);
            TS_REQUIRE(log_h >= fri_.log_blowup, TS_ERR_INVALID, "open: matrix shorter than the blowup");
            const unsigned log_n = log_h - fri_.log_blowup;
            const uint64_t n = 1ull << log_n;
            const uint32_t w = m.width;
            const uint32_t n_inv = inv_canon((uint32_t)(n % P));
            const auto& pts = r.points[mi];
            for (size_t p0 = 0; p0 < pts.size(); p0 += 2) {
                const uint32_t np = (uint32_t)std::min<size_t>(2, pts.size() - p0);
                Ef pts_mont[2] = {ef_to_mont(pts[p0]), ef_to_mont(pts[p0 + np - 1])};
                // :358-369 interpolate_coset on the low coset (first n bit-reversed rows)
                std::vector<Ef> raw((size_t)w * np);
                {
                    StageTimer t(&ctx_, "compute opened values with Lagrange interpolation");
                    DevBuf<Ef> weights(&ctx_, (size_t)np * n);
                    launch_bary_weights(ctx_, log_n, pts_mont, np, weights.p);
                    DevBuf<Ef> sums(&ctx_, raw.size());
                    launch_bary_dots(ctx_, m, log_n, weights.p, np, sums.p);  // [col][point]
                    d2h_sync(ctx_, raw.data(), sums.p, raw.size() * sizeof(Ef));
                }
                StageTimer t(&ctx_, "reduce rows");
                ReduceArgs a;
                memset(&a, 0, sizeof a);
                a.n_points = np;
                for (uint32_t p = 0; p < np; p++) {
                    // p(z) = ((z/31)^n - 1)/n * sum_i p_i x_i/(z - x_i)
                    Ef un = efc_pow(efc_mul_base(pts[p0 + p], gen_inv), n);
                    un.c[0] = sub(un.c[0], 1);
                    const Ef scale = efc_mul_base(un, n_inv);
                    Ef rys = ef_zero();  // :372 dot_product(alpha.powers(), ys)
                    for (uint32_t c = 0; c < w; c++) {
                        const Ef y = efc_mul(raw[(size_t)c * np + p], scale);
                        opened_values.push_back(y);
                        Ef ap;
                        memcpy(ap.c, &apow[4 * (size_t)c], 16);
                        rys = ef_add(rys, ef_mul(y, ap));
                    }
                    a.z_mont[p] = pts_mont[p];
                    a.off_mont[p] = ef_pow(am, num_reduced[log_h]);  // :371
                    a.rys[p] = rys;
                    num_reduced[log_h] += w;  // :383
                }
                if (!ro[log_h].p) {
                    ro[log_h] = DevBuf<Ef>(&ctx_, m.height);
                    a.accumulate = 0;
                } else {
                    a.accumulate = 1;
                }
                launch_reduce(ctx_, m, log_h, d_apow.p, a, ro[log_h].p);  // :375-381
            }
        }
    }
    // :389-393 fri_input: the reduced openings by descending height
    std::vector<DevBuf<Ef>> inputs;
    std::vector<unsigned> log_lens;
    for (int lh = 31; lh >= 0; lh--)
        if (ro[lh].p) {
            inputs.push_back(std::move(ro[lh]));
            log_lens.push_back((unsigned)lh);
        }
    TS_REQUIRE(!inputs.empty(), TS_ERR_INVALID, "open: nothing to open");
    TS_REQUIRE(log_lens[0] == log_global_max, TS_ERR_UNSUPPORTED,
               "open: the tallest committed matrix must be opened at one point at least");
    std::vector<const PcsData*> datas;
    for (auto& r : rounds) datas.push_back(r.data);
    std::vector<uint32_t> pf;
    fri_prove(inputs, log_lens, challenger, datas, pf);
    return pf;
}

// ------------------------------------------------------------------ prove
std::vector<uint32_t> prove(TwoAdicFriPcs& pcs, const AirProgram& air, BfChallenger& challenger,
                            DeviceMatrix trace, const std::vector<uint32_t>& public_values) {
    Context& ctx = pcs.ctx();
    const FriConfig& fri = pcs.fri();
    TS_REQUIRE(trace.width == air.width, TS_ERR_INVALID, "prove: trace width != AIR width");
    TS_REQUIRE(public_values.size() == air.n_public, TS_ERR_INVALID,
               "prove: wrong number of public values");
    const uint64_t degree = trace.height;  // prover.rs:43-44
    const unsigned log_degree = log2_strict(degree);
    const unsigned lqd = air.log_quotient_degree;  // :46
    const uint32_t qd = 1u << lqd;
    const unsigned log_N = log_degree + fri.log_blowup;
    const uint32_t w = air.width;
    TS_REQUIRE(lqd <= fri.log_blowup, TS_ERR_INVARIANT,
               "quotient domain larger than the committed LDE (log_quotient_degree > log_blowup)");
    ctx.ensure_twiddles(std::max(1u, log_N));

    // :50-53 commit to trace data (natural domain: shift 1)
    std::vector<DeviceMatrix> tv;
    tv.push_back(std::move(trace));
    std::unique_ptr<PcsData> trace_data = pcs.commit(tv, {1u});
    challenger.observe_commitment(trace_data->root);  // :60
    const Ef alpha = challenger.sample();              // :63

    // :65-80 quotient on the disjoint domain, flattened and split into qd chunks
    std::vector<DeviceMatrix> chunks = pcs.quotient_chunks(*trace_data, air, public_values, alpha);
    // split_domains (:80): chunk c lives on {log_n, shift = 31 * omega_{n*qd}^c}
    std::vector<uint32_t> qshifts(qd);
    const uint32_t gq = two_adic_generator(log_degree + lqd);
    for (uint32_t c = 0; c < qd; c++) qshifts[c] = mul(GENERATOR, pow_canon(gq, c));
    std::unique_ptr<PcsData> quotient_data = pcs.commit(chunks, qshifts);  // :82-83
    challenger.observe_commitment(quotient_data->root);                    // :84
    const Ef zeta = challenger.sample();                                   // :91

    // :94-104 open; two_adic_pcs.rs:312 batch-combination challenge first.  Same result as
    // pcs.open({trace: [zeta, zeta_next]}, {chunks: [zeta]}), through the one-pass reduce kernel.
    const Ef batch_alpha = challenger.sample();
    std::vector<Ef> opened;
    std::vector<DevBuf<Ef>> inputs;
    inputs.push_back(pcs.open_reduce(*trace_data, *quotient_data, zeta, batch_alpha, opened));

    // ---- Proof (prover.rs:105-118) in TSPF v1 order
    std::vector<uint32_t> pf;
    pf.reserve(64 + opened.size() * 4);
    pf.push_back(TSPF_MAGIC);
    pf.push_back(1);
    pf.push_back(log_degree);
    pf.push_back(w);
    pf.push_back(qd);
    pf.insert(pf.end(), trace_data->root, trace_data->root + 8);
    pf.insert(pf.end(), quotient_data->root, quotient_data->root + 8);
    for (auto& e : opened) pf.insert(pf.end(), e.c, e.c + 4);
    pcs.fri_prove(inputs, {log_N}, challenger, {trace_data.get(), quotient_data.get()}, pf);
    return pf;
}

}  // namespace ts
