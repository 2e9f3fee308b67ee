/*
 * oracle/stark.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of the reference's prover hot path and of the verifier that accepts it:
 *   uni-stark/src/prover.rs:25-119     prove
 *   uni-stark/src/prover.rs:122-194    quotient_values
 *   uni-stark/src/folder.rs:60-64      ProverConstraintFolder::assert_zero (acc = acc*alpha + c)
 *   uni-stark/src/verifier.rs:19-161   verify
 *   fri/src/two_adic_pcs.rs:227-245    Pcs::commit
 *   fri/src/two_adic_pcs.rs:247-258    get_evaluations_on_domain
 *   fri/src/two_adic_pcs.rs:260-419    Pcs::open  (+ :678-720 compute_inverse_denominators)
 *   fri/src/two_adic_pcs.rs:421-534    Pcs::verify
 *   fri/src/two_adic_pcs.rs:87-147     fold_row / fold_matrix
 *   fri/src/prover.rs:19-141           bf_prove / bf_answer_query / bf_commit_phase
 *   fri/src/verifier.rs:20-165         verify_shape_and_sample_challenges / verify_challenges /
 *                                      verify_query
 * Plonky3 helper semantics (selectors_on_coset, split_evals, interpolate_coset, ...) follow
 * SURVEY.md Appendix A (upstream source is not on disk).
 */
#include "oracle_internal.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers */
static void wb_init(ts_or_wbuf* b, uint32_t* w, size_t cap) {
    b->w = w;
    b->len = 0;
    b->cap = cap;
    b->overflow = 0;
}
static void wb_push(ts_or_wbuf* b, uint32_t v) {
    if (b->len < b->cap) b->w[b->len] = v; else b->overflow = 1;
    b->len++;
}
static void wb_push_n(ts_or_wbuf* b, const uint32_t* v, size_t n) {
    for (size_t i = 0; i < n; i++) wb_push(b, v[i]);
}
static void wb_push_ef(ts_or_wbuf* b, ef4 e) { wb_push_n(b, e.c, 4); }

static ef4 ef4_load(const uint32_t* p) { ef4 r = {{p[0], p[1], p[2], p[3]}}; return r; }

/* roots per commitment: 1 (Blake3 Merkle MMCS) or num_queries (taptree mode, mmcs.c) */
static uint32_t n_roots(void) {
    uint32_t q = ts_or_mmcs_tap_queries();
    return q ? q : 1;
}
/* CanObserve<Vec<[PF; N]>> (basic/src/challenger/mod.rs:211-223): every root of the commitment */
static void observe_roots(ts_or_challenger* c, const uint32_t* roots) {
    for (uint32_t q = 0; q < n_roots(); q++) ts_or_chal_observe_digest(c, roots + 8 * (size_t)q);
}
static void copy_roots(const ts_or_mmcs_data* d, uint32_t* out) {
    for (uint32_t q = 0; q < n_roots(); q++) memcpy(out + 8 * (size_t)q, ts_or_mmcs_root(d, q), 32);
}

static ef4 chal_sample_ef(ts_or_challenger* c) {
    ef4 r;
    ts_or_chal_sample(c, r.c);
    return r;
}

typedef struct {
    unsigned log_n;
    uint32_t shift;
} dom_t;

/* last transcript (stage tests) */
static __thread uint32_t g_tr[4 * 3 + 1 + 4 * 40 + 2 + 256];
static __thread size_t g_tr_len;

/* ------------------------------------------------------- quotient (prover) */
/* selectors_on_coset (App. A.4) at the natural index i of the quotient domain */
typedef struct {
    uint32_t is_first, is_last, is_transition, inv_zeroifier;
} sel_t;

void ts_or_quotient_values(const uint32_t* tape, size_t n_words, const uint32_t* lde,
                           unsigned log_n, unsigned log_blowup, const uint32_t* pis,
                           const uint32_t alpha_w[4], uint32_t* out) {
    ts_or_tape t;
    if (ts_or_tape_parse(tape, n_words, &t)) return;
    (void)log_blowup;
    unsigned lqd = (unsigned)ts_or_air_log_quotient_degree(tape, n_words);
    size_t w = t.width;
    size_t n = (size_t)1 << log_n, qn = n << lqd;
    unsigned log_qn = log_n + lqd;
    size_t next_step = (size_t)1 << lqd; /* prover.rs:139-140 */
    ef4 alpha = ef4_load(alpha_w);
    uint32_t gq = bb_two_adic_generator(log_qn);
    uint32_t gn_inv = bb_inv(bb_two_adic_generator(log_n)); /* omega_n^-1 */
    uint32_t s_pow_n = bb_pow(BB_GENERATOR, n);
    /* Z_H takes only qd distinct values: 31^n * omega_qd^(i mod qd) - 1 */
    uint32_t zh[64], zh_inv[64];
    uint32_t gqd = bb_two_adic_generator(lqd);
    for (size_t k = 0; k < next_step; k++) {
        zh[k] = bb_sub(bb_mul(s_pow_n, bb_pow(gqd, k)), 1);
        zh_inv[k] = bb_inv(zh[k]);
    }
#pragma omp parallel
    {
        uint32_t* v = (uint32_t*)malloc((t.n_nodes + 1) * sizeof(uint32_t));
#pragma omp for schedule(static)
        for (size_t i = 0; i < qn; i++) {
            uint32_t x = bb_mul(BB_GENERATOR, bb_pow(gq, i));
            uint32_t z = zh[i & (next_step - 1)];
            sel_t s;
            s.is_first = bb_mul(z, bb_inv(bb_sub(x, 1)));
            s.is_last = bb_mul(z, bb_inv(bb_sub(x, gn_inv)));
            s.is_transition = bb_sub(x, gn_inv);
            s.inv_zeroifier = zh_inv[i & (next_step - 1)];
            /* get_evaluations_on_domain (two_adic_pcs.rs:247-258): natural row i of the
             * quotient domain = stored (bit-reversed) row bitrev_{log_qn}(i) */
            size_t r0 = ts_bitrev(i, log_qn);
            size_t r1 = ts_bitrev((i + next_step) & (qn - 1), log_qn);
            ts_or_tape_eval_base(&t, lde + r0 * w, lde + r1 * w, pis, s.is_first, s.is_last,
                                 s.is_transition, v);
            ef4 acc = ef4_zero();
            for (uint32_t c = 0; c < t.n_constraints; c++) /* folder.rs:60-64 */
                acc = ef4_add_base(ef4_mul(acc, alpha), v[t.constraints[c]]);
            acc = ef4_mul_base(acc, s.inv_zeroifier); /* prover.rs:183 */
            memcpy(out + 4 * i, acc.c, 16);
        }
        free(v);
    }
}

/* prover.rs:78-80 */
void ts_or_split_quotient(const uint32_t* qvals, unsigned log_n, unsigned log_qd, uint32_t* out) {
    size_t n = (size_t)1 << log_n, qd = (size_t)1 << log_qd;
    for (size_t r = 0; r < n * qd; r++) {
        size_t c = r % qd, pos = r / qd;
        memcpy(out + (c * n + pos) * 4, qvals + 4 * r, 16);
    }
}

/* ---------------------------------------------------------------- FRI folding */
/* two_adic_pcs.rs:116-147 */
void ts_or_fold_matrix(const uint32_t* in, size_t h, const uint32_t beta_w[4], uint32_t* out) {
    unsigned log_h = ts_log2_strict(h);
    uint32_t g_inv = bb_inv(bb_two_adic_generator(log_h + 1));
    uint32_t one_half = bb_inv(2);
    ef4 half_beta = ef4_mul_base(ef4_load(beta_w), one_half);
#pragma omp parallel for schedule(static) if (h > 4096)
    for (size_t i = 0; i < h; i++) {
        /* powers[j] = half_beta * g_inv^j, then bit-reversed: row i uses j = bitrev(i) */
        ef4 power = ef4_mul_base(half_beta, bb_pow(g_inv, ts_bitrev(i, log_h)));
        ef4 lo = ef4_load(in + 8 * i), hi = ef4_load(in + 8 * i + 4);
        ef4 a = ef4_mul(ef4_add_base(power, one_half), lo);
        ef4 b = ef4_mul(ef4_sub(ef4_from_base(one_half), power), hi);
        ef4 r = ef4_add(a, b);
        memcpy(out + 4 * i, r.c, 16);
    }
}

/* two_adic_pcs.rs:87-114 */
static ef4 fold_row(size_t index, unsigned log_height, ef4 beta, ef4 e0, ef4 e1) {
    uint32_t s = bb_pow(bb_two_adic_generator(log_height + 1), ts_bitrev(index, log_height));
    uint32_t x0 = s, x1 = bb_neg(s); /* two_adic_generator(1) = -1; 2-element bitrev = id */
    ef4 num = ef4_mul(ef4_sub_base(beta, x0), ef4_sub(e1, e0));
    return ef4_add(e0, ef4_mul_base(num, bb_inv(bb_sub(x1, x0))));
}
void ts_or_fold_row(size_t index, unsigned log_height, const uint32_t beta[4],
                    const uint32_t e0[4], const uint32_t e1[4], uint32_t out[4]) {
    ef4 r = fold_row(index, log_height, ef4_load(beta), ef4_load(e0), ef4_load(e1));
    memcpy(out, r.c, 16);
}

/* ------------------------------------------------------------------ PCS open */
typedef struct {
    const ts_or_mmcs_data* data;
    const int* n_points;      /* per matrix */
    const ef4* const* points; /* per matrix */
} open_round;

/* interpolate_coset (App. A.6) on the first h rows (bit-reversed storage) of `m`, coset shift 31 */
static void interpolate_low_coset(const uint32_t* m, size_t h, size_t w, ef4 z, ef4* ys) {
    unsigned log_h = ts_log2_strict(h);
    uint32_t g = bb_two_adic_generator(log_h);
    ef4* d = (ef4*)malloc(h * sizeof(ef4));
#pragma omp parallel for schedule(static) if (h > 1024)
    for (size_t i = 0; i < h; i++) {
        uint32_t x = bb_mul(BB_GENERATOR, bb_pow(g, i));
        /* x_i / (z - x_i) */
        d[i] = ef4_mul_base(ef4_inv(ef4_sub_base(z, x)), x);
    }
#pragma omp parallel for schedule(static) if (w > 8)
    for (size_t c = 0; c < w; c++) {
        ef4 acc = ef4_zero();
        for (size_t i = 0; i < h; i++) {
            size_t r = ts_bitrev(i, log_h); /* BitReversalPerm view (two_adic_pcs.rs:365) */
            acc = ef4_add(acc, ef4_mul_base(d[i], m[r * w + c]));
        }
        ys[c] = acc;
    }
    /* ((z/31)^h - 1) / h */
    ef4 u = ef4_mul_base(z, bb_inv(BB_GENERATOR));
    ef4 scale = ef4_mul_base(ef4_sub_base(ef4_pow(u, h), 1), bb_inv((uint32_t)(h % BB_P)));
    for (size_t c = 0; c < w; c++) ys[c] = ef4_mul(ys[c], scale);
    free(d);
}

/* two_adic_pcs.rs:312-389 given the batch challenge alpha.
 * opened: appended in (round, matrix, point, column) order.
 * ro[log_height]: malloc'ed vector of 2^log_height EF or NULL. */
static void pcs_open_compute(const ts_or_fri_config* cfg, int n_rounds, const open_round* rounds,
                             ef4 alpha, ef4* opened, size_t* n_opened, ef4* ro[32]) {
    size_t num_reduced[32];
    memset(num_reduced, 0, sizeof num_reduced);
    for (int i = 0; i < 32; i++) ro[i] = NULL;
    size_t no = 0;
    for (int r = 0; r < n_rounds; r++) {
        const ts_or_mmcs_data* d = rounds[r].data;
        for (int mi = 0; mi < ts_or_mmcs_n_mats(d); mi++) {
            const uint32_t* m = ts_or_mmcs_matrix(d, mi);
            size_t h = ts_or_mmcs_height(d, mi), w = ts_or_mmcs_width(d, mi);
            unsigned lh = ts_log2_strict(h);
            if (!ro[lh]) ro[lh] = (ef4*)calloc(h, sizeof(ef4));
            uint32_t gh = bb_two_adic_generator(lh);
            /* alpha^i for i < w */
            ef4* apow = (ef4*)malloc((w + 1) * sizeof(ef4));
            apow[0] = ef4_one();
            for (size_t i = 1; i <= w; i++) apow[i] = ef4_mul(apow[i - 1], alpha);
            for (int pi = 0; pi < rounds[r].n_points[mi]; pi++) {
                ef4 z = rounds[r].points[mi][pi];
                ef4* ys = opened + no;
                interpolate_low_coset(m, h >> cfg->log_blowup, w, z, ys); /* :358-369 */
                no += w;
                ef4 off = ef4_pow(alpha, num_reduced[lh]); /* :371 */
                ef4 rys = ef4_zero();                      /* :372 */
                for (size_t i = 0; i < w; i++) rys = ef4_add(rys, ef4_mul(apow[i], ys[i]));
                ef4* rov = ro[lh];
#pragma omp parallel for schedule(static) if (h > 1024)
                for (size_t X = 0; X < h; X++) {
                    /* :698-717: x = 31 * omega_h^bitrev(X); inv_denom = 1/(x - z) */
                    uint32_t x = bb_mul(BB_GENERATOR, bb_pow(gh, ts_bitrev(X, lh)));
                    ef4 inv_denom = ef4_inv(ef4_sub(ef4_from_base(x), z));
                    ef4 row = ef4_zero(); /* dot_ext_powers :375 */
                    for (size_t i = 0; i < w; i++)
                        row = ef4_add(row, ef4_mul_base(apow[i], m[X * w + i]));
                    ef4 t = ef4_mul(ef4_mul(off, ef4_sub(row, rys)), inv_denom); /* :380 */
                    rov[X] = ef4_add(rov[X], t);
                }
                num_reduced[lh] += w; /* :383 */
            }
            free(apow);
        }
    }
    *n_opened = no;
}

void ts_or_open_reduce(const uint32_t* trace_lde, size_t w, const uint32_t* const* chunk_ldes,
                       unsigned log_qd, unsigned log_n, unsigned log_blowup,
                       const uint32_t zeta_w[4], const uint32_t alpha_w[4], uint32_t* opened_out,
                       uint32_t* ro_out) {
    ts_or_fri_config cfg = {log_blowup, 0, 0};
    size_t N = (size_t)1 << (log_n + log_blowup);
    size_t qd = (size_t)1 << log_qd;
    uint32_t root[8];
    const uint32_t* tm[1] = {trace_lde};
    size_t th[1] = {N}, tw[1] = {w};
    ts_or_mmcs_data* td = ts_or_mmcs_commit(1, tm, th, tw, root);
    size_t qh[64], qw[64];
    for (size_t c = 0; c < qd; c++) { qh[c] = N; qw[c] = 4; }
    ts_or_mmcs_data* qdta = ts_or_mmcs_commit((int)qd, chunk_ldes, qh, qw, root);
    ef4 zeta = ef4_load(zeta_w);
    ef4 zeta_next = ef4_mul_base(zeta, bb_two_adic_generator(log_n));
    ef4 tpts[2] = {zeta, zeta_next};
    const ef4* tpp[1] = {tpts};
    int tnp[1] = {2};
    const ef4* qpp[64];
    int qnp[64];
    for (size_t c = 0; c < qd; c++) { qpp[c] = &zeta; qnp[c] = 1; }
    open_round rounds[2] = {{td, tnp, tpp}, {qdta, qnp, qpp}};
    ef4* opened = (ef4*)malloc((2 * w + 4 * qd) * sizeof(ef4));
    size_t no;
    ef4* ro[32];
    pcs_open_compute(&cfg, 2, rounds, ef4_load(alpha_w), opened, &no, ro);
    memcpy(opened_out, opened, no * 16);
    memcpy(ro_out, ro[log_n + log_blowup], N * 16);
    for (int i = 0; i < 32; i++) free(ro[i]);
    free(opened);
    ts_or_mmcs_free(td);
    ts_or_mmcs_free(qdta);
}

/* ------------------------------------------------------------------- bf_prove */
typedef struct {
    int n_rounds;
    const ts_or_mmcs_data* const* data; /* per commit round (input) */
    unsigned log_global_max_height;
} input_opener;

/* two_adic_pcs.rs:399-414 open_input closure -> appended to the proof buffer */
static void write_input_proof(ts_or_wbuf* b, const input_opener* op, size_t query_index) {
    wb_push(b, (uint32_t)op->n_rounds);
    for (int r = 0; r < op->n_rounds; r++) {
        const ts_or_mmcs_data* d = op->data[r];
        unsigned lmh = ts_or_mmcs_log_max_height(d);
        unsigned bits_reduced = op->log_global_max_height - lmh;
        size_t reduced_index = query_index >> bits_reduced;
        int nm = ts_or_mmcs_n_mats(d);
        size_t tot = 0;
        for (int i = 0; i < nm; i++) tot += ts_or_mmcs_width(d, i);
        uint32_t* rows = (uint32_t*)malloc(tot * 4 + 4);
        uint32_t* path = (uint32_t*)malloc((lmh + 1) * 32);
        ts_or_mmcs_open(d, reduced_index, rows, path);
        wb_push(b, (uint32_t)nm);
        size_t off = 0;
        for (int i = 0; i < nm; i++) {
            size_t w = ts_or_mmcs_width(d, i);
            wb_push(b, (uint32_t)w);
            wb_push_n(b, rows + off, w);
            off += w;
        }
        wb_push(b, lmh);
        wb_push_n(b, path, (size_t)lmh * 8);
        free(rows);
        free(path);
    }
}

/* fri/src/prover.rs:19-141.  inputs[k] has length 2^log_lens[k] (descending).
 * pass_through != 0: fri/tests/fri.rs:109-118 style input proof (the literal reduced openings) */
static int bf_prove(const ts_or_fri_config* cfg, int n_inputs, ef4* const* inputs,
                    const unsigned* log_lens, ts_or_challenger* chal, const input_opener* op,
                    int pass_through, ts_or_wbuf* b) {
    unsigned log_max_height = log_lens[0];
    size_t blowup = (size_t)1 << cfg->log_blowup;
    size_t len = (size_t)1 << log_max_height;
    ef4* folded = (ef4*)malloc(len * sizeof(ef4));
    memcpy(folded, inputs[0], len * sizeof(ef4));
    int next_in = 1;
    int max_rounds = (int)log_max_height + 1;
    ts_or_mmcs_data** data = (ts_or_mmcs_data**)calloc(max_rounds, sizeof(void*));
    const size_t cw = 8 * (size_t)n_roots(); /* words per commitment */
    uint32_t* commits = (uint32_t*)malloc(max_rounds * cw * 4);
    ef4 betas[40];
    int R = 0;
    /* commit phase :111-127 */
    ts_or_mmcs_tap_u32(4); /* the commit-phase matrices hold EF4 elements (ChallengeMmcs) */
    while (len > blowup) {
        const uint32_t* leaves = (const uint32_t*)folded; /* RowMajorMatrix(folded, 2) as h x 8 base */
        size_t h = len / 2, w8 = 8;
        uint32_t first_root[8];
        data[R] = ts_or_mmcs_commit(1, &leaves, &h, &w8, first_root);
        copy_roots(data[R], commits + cw * R);
        observe_roots(chal, commits + cw * R);
        ef4 beta = chal_sample_ef(chal);
        betas[R] = beta;
        ef4* nf = (ef4*)malloc(h * sizeof(ef4));
        ts_or_fold_matrix((const uint32_t*)folded, h, beta.c, (uint32_t*)nf);
        free(folded);
        folded = nf;
        len = h;
        R++;
        if (next_in < n_inputs && ((size_t)1 << log_lens[next_in]) == len) { /* :124-126 */
            for (size_t i = 0; i < len; i++) folded[i] = ef4_add(folded[i], inputs[next_in][i]);
            next_in++;
        }
    }
    ts_or_mmcs_tap_u32(1);
    /* :129-134 */
    int rc = 0;
    if (len != blowup) rc = -5;
    ef4 final_poly = folded[0];
    for (size_t i = 0; i < len; i++)
        if (!ef4_eq(folded[i], final_poly)) rc = -5;
    free(folded);
    uint32_t pow_witness = 0;
    if (rc == 0 && ts_or_chal_grind(chal, cfg->proof_of_work_bits, &pow_witness)) rc = -4;

    wb_push(b, (uint32_t)R);
    for (int r = 0; r < R; r++) wb_push_n(b, commits + cw * r, cw);
    wb_push(b, cfg->num_queries);
    g_tr_len = 12;
    g_tr[g_tr_len++] = (uint32_t)R;
    for (int r = 0; r < R; r++) { memcpy(g_tr + g_tr_len, betas[r].c, 16); g_tr_len += 4; }
    g_tr[g_tr_len++] = pow_witness;
    g_tr[g_tr_len++] = cfg->num_queries;
    if (rc == 0) {
        for (uint32_t q = 0; q < cfg->num_queries; q++) { /* :45-59 */
            size_t index = (size_t)ts_or_chal_sample_bits(chal, log_max_height);
            if (g_tr_len < sizeof g_tr / 4) g_tr[g_tr_len++] = (uint32_t)index;
            ts_or_mmcs_tap_select(q); /* open_batch(query_times_index = q, ...), fri/src/prover.rs:50-56 */
            if (pass_through) {
                wb_push(b, (uint32_t)n_inputs);
                for (int k = 0; k < n_inputs; k++) {
                    wb_push(b, log_lens[k]);
                    wb_push_ef(b, inputs[k][index >> (log_max_height - log_lens[k])]);
                }
            } else {
                write_input_proof(b, op, index);
            }
            for (int i = 0; i < R; i++) { /* bf_answer_query :69-90 */
                size_t index_i = index >> i >> 1;
                uint32_t row[8];
                unsigned lh = ts_or_mmcs_log_max_height(data[i]);
                uint32_t* path = (uint32_t*)malloc((lh + 1) * 32);
                ts_or_mmcs_open(data[i], index_i, row, path);
                wb_push_n(b, row, 8);
                wb_push(b, lh);
                wb_push_n(b, path, (size_t)lh * 8);
                free(path);
            }
        }
    }
    wb_push_ef(b, final_poly);
    wb_push(b, pow_witness);
    for (int r = 0; r < R; r++) ts_or_mmcs_free(data[r]);
    free(data);
    free(commits);
    return rc;
}

/* --------------------------------------------------------------------- prove */
static ts_or_mmcs_data* pcs_commit(const ts_or_fri_config* cfg, int n_mats, const dom_t* doms,
                                   const uint32_t* const* evals, const size_t* widths,
                                   uint32_t root[8]) {
    uint32_t** ldes = (uint32_t**)malloc(n_mats * sizeof(uint32_t*));
    size_t* hs = (size_t*)malloc(n_mats * sizeof(size_t));
    for (int i = 0; i < n_mats; i++) {
        size_t N = (size_t)1 << (doms[i].log_n + cfg->log_blowup);
        ldes[i] = (uint32_t*)malloc(N * widths[i] * 4);
        ts_or_commit_lde(evals[i], doms[i].log_n, widths[i], doms[i].shift, cfg->log_blowup,
                         ldes[i]);
        hs[i] = N;
    }
    ts_or_mmcs_data* d =
        ts_or_mmcs_commit(n_mats, (const uint32_t* const*)ldes, hs, widths, root);
    for (int i = 0; i < n_mats; i++) free(ldes[i]);
    free(ldes);
    free(hs);
    return d;
}

/* sort log-heights descending and run FRI: two_adic_pcs.rs:389-416 */
static int pcs_open(const ts_or_fri_config* cfg, int n_rounds, const open_round* rounds,
                    ts_or_challenger* chal, ef4* opened, size_t* n_opened, ts_or_wbuf* b) {
    ef4 alpha = chal_sample_ef(chal); /* :312 */
    memcpy(g_tr + 8, alpha.c, 16);
    ef4* ro[32];
    pcs_open_compute(cfg, n_rounds, rounds, alpha, opened, n_opened, ro);
    ef4* inputs[32];
    unsigned log_lens[32];
    int n_inputs = 0;
    for (int lh = 31; lh >= 0; lh--)
        if (ro[lh]) {
            inputs[n_inputs] = ro[lh];
            log_lens[n_inputs++] = (unsigned)lh;
        }
    const ts_or_mmcs_data* datas[64];
    for (int r = 0; r < n_rounds; r++) datas[r] = rounds[r].data;
    input_opener op = {n_rounds, datas, log_lens[0]};
    int rc = bf_prove(cfg, n_inputs, inputs, log_lens, chal, &op, 0, b);
    for (int i = 0; i < 32; i++) free(ro[i]);
    return rc;
}

#define TSPF_MAGIC 0x46505354u

/* prover.rs:40-41: `#[cfg(debug_assertions)] check_constraints(...)`.  On by default (a debug build of
 * the reference); off = a release build, which proves whatever trace it is given. */
static int g_debug_assertions = 1;
void ts_or_set_debug_assertions(int on) { g_debug_assertions = on; }

int64_t ts_or_prove(const ts_or_fri_config* cfg, const uint32_t* tape, size_t n_tape,
                    ts_or_challenger* chal, const uint32_t* trace, unsigned log_n,
                    const uint32_t* pis, uint32_t* proof_out, size_t cap_words) {
    ts_or_tape t;
    if (ts_or_tape_parse(tape, n_tape, &t)) return -2;
    size_t n = (size_t)1 << log_n, w = t.width;
    /* prover.rs:40-41 (debug_assertions) */
    if (g_debug_assertions && ts_or_check_constraints(tape, n_tape, trace, n, pis) >= 0) return -3;
    unsigned lqd = (unsigned)ts_or_air_log_quotient_degree(tape, n_tape); /* :46 */
    size_t qd = (size_t)1 << lqd;
    if (lqd > cfg->log_blowup) return -2; /* two_adic_pcs.rs:256 */
    ts_or_wbuf b;
    wb_init(&b, proof_out, cap_words);

    /* :50-53 commit to trace */
    dom_t td = {log_n, 1};
    const size_t cw = 8 * (size_t)n_roots();
    uint32_t first_root[8];
    uint32_t* trace_root = (uint32_t*)malloc(cw * 4);
    uint32_t* quot_root = (uint32_t*)malloc(cw * 4);
    ts_or_mmcs_data* tdata = pcs_commit(cfg, 1, &td, &trace, &w, first_root);
    copy_roots(tdata, trace_root);
    observe_roots(chal, trace_root); /* :60 */
    ef4 alpha = chal_sample_ef(chal);            /* :63 */
    memcpy(g_tr, alpha.c, 16);

    /* :65-77 quotient */
    uint32_t* qvals = (uint32_t*)malloc(n * qd * 16);
    ts_or_quotient_values(tape, n_tape, ts_or_mmcs_matrix(tdata, 0), log_n, cfg->log_blowup, pis,
                          alpha.c, qvals);
    uint32_t* chunks = (uint32_t*)malloc(n * qd * 16);
    ts_or_split_quotient(qvals, log_n, lqd, chunks); /* :78-79 */
    free(qvals);
    dom_t qdoms[64];
    const uint32_t* qev[64];
    size_t qw[64];
    uint32_t gq = bb_two_adic_generator(log_n + lqd);
    for (size_t c = 0; c < qd; c++) { /* split_domains :80 (App. A.3) */
        qdoms[c].log_n = log_n;
        qdoms[c].shift = bb_mul(BB_GENERATOR, bb_pow(gq, c));
        qev[c] = chunks + c * n * 4;
        qw[c] = 4;
    }
    ts_or_mmcs_data* qdata = pcs_commit(cfg, (int)qd, qdoms, qev, qw, first_root); /* :82-83 */
    free(chunks);
    copy_roots(qdata, quot_root);
    observe_roots(chal, quot_root); /* :84 */
    ef4 zeta = chal_sample_ef(chal);            /* :91 */
    memcpy(g_tr + 4, zeta.c, 16);
    ef4 zeta_next = ef4_mul_base(zeta, bb_two_adic_generator(log_n)); /* :92 */

    /* header + commitments */
    wb_push(&b, TSPF_MAGIC);
    if (ts_or_mmcs_tap_queries()) { /* TSPF v2: taptree commitments, num_queries roots each */
        wb_push(&b, 2);
        wb_push(&b, log_n);
        wb_push(&b, (uint32_t)w);
        wb_push(&b, (uint32_t)qd);
        wb_push(&b, ts_or_mmcs_tap_queries());
    } else {
        wb_push(&b, 1);
        wb_push(&b, log_n);
        wb_push(&b, (uint32_t)w);
        wb_push(&b, (uint32_t)qd);
    }
    wb_push_n(&b, trace_root, cw);
    wb_push_n(&b, quot_root, cw);
    free(trace_root);
    free(quot_root);

    /* :94-104 open */
    ef4 tpts[2] = {zeta, zeta_next};
    const ef4* tpp[1] = {tpts};
    int tnp[1] = {2};
    const ef4* qpp[64];
    int qnp[64];
    for (size_t c = 0; c < qd; c++) { qpp[c] = &zeta; qnp[c] = 1; }
    open_round rounds[2] = {{tdata, tnp, tpp}, {qdata, qnp, qpp}};
    ef4* opened = (ef4*)malloc((2 * w + 4 * qd) * sizeof(ef4));
    size_t n_opened = 0;
    /* opened values must precede the opening proof in the buffer: run open into a side buffer */
    size_t side_cap = cap_words;
    uint32_t* side = (uint32_t*)malloc(side_cap * 4 + 4);
    ts_or_wbuf sb;
    wb_init(&sb, side, side_cap);
    int rc = pcs_open(cfg, 2, rounds, chal, opened, &n_opened, &sb);
    /* :105-112 trace_local, trace_next, quotient_chunks */
    wb_push_n(&b, (const uint32_t*)opened, n_opened * 4);
    if (sb.overflow) b.overflow = 1;
    else wb_push_n(&b, side, sb.len);
    free(side);
    free(opened);
    ts_or_mmcs_free(tdata);
    ts_or_mmcs_free(qdata);
    if (rc) return rc;
    if (b.overflow) return -1;
    return (int64_t)b.len;
}

size_t ts_or_last_transcript(uint32_t* out, size_t cap) {
    size_t n = g_tr_len < cap ? g_tr_len : cap;
    memcpy(out, g_tr, n * 4);
    return g_tr_len;
}

/* -------------------------------------------------------------------- verify */
typedef struct {
    const uint32_t* w;
    size_t len, pos;
    int bad;
} rbuf;
static uint32_t rb_get(rbuf* r) {
    if (r->pos >= r->len) { r->bad = 1; return 0; }
    return r->w[r->pos++];
}
static const uint32_t* rb_take(rbuf* r, size_t n) {
    if (r->pos + n > r->len || r->pos + n < r->pos) { r->bad = 1; return NULL; }
    const uint32_t* p = r->w + r->pos;
    r->pos += n;
    return p;
}

/* claims for Pcs::verify: per round, per matrix: domain + (point, values) list */
typedef struct {
    dom_t dom;
    size_t width;
    int n_points;
    const ef4* points;
    const ef4* const* values; /* per point: width values */
} mat_claim;
typedef struct {
    const uint32_t* commit;
    int n_mats;
    const mat_claim* mats;
} round_claim;

static int strict_first_round = 1;
void ts_or_set_strict(int s) { strict_first_round = s; }

/* Pcs::verify, two_adic_pcs.rs:421-534 + fri/src/verifier.rs.  `pass_through`: the input proof is
 * the literal list of (log_height, reduced opening) (fri/tests/fri.rs:109-118,136). */
static int fri_verify(const ts_or_fri_config* cfg, int n_rounds, const round_claim* rounds,
                      ef4 alpha, ts_or_challenger* chal, rbuf* rb, int pass_through) {
    /* verify_shape_and_sample_challenges, verifier.rs:20-60 */
    uint32_t R = rb_get(rb);
    if (rb->bad || R > 31) return 9;
    const size_t cw = 8 * (size_t)n_roots();
    const uint32_t* commits = rb_take(rb, (size_t)R * cw);
    if (rb->bad) return 9;
    ef4 betas[32];
    for (uint32_t r = 0; r < R; r++) {
        observe_roots(chal, commits + cw * r);
        betas[r] = chal_sample_ef(chal);
    }
    /* taptree mode: where each commitment's lock scripts start in the table (commit order: the
     * input rounds, then the FRI rounds with 1 + 2 scripts per tree) */
    size_t lock_base[64], fri_lock_base = 0;
    for (int r = 0; r < n_rounds && r < 64; r++) {
        size_t tw = 0;
        for (int i = 0; i < rounds[r].n_mats; i++) tw += rounds[r].mats[i].width;
        lock_base[r] = fri_lock_base;
        fri_lock_base += (size_t)n_roots() * (1 + tw);
    }
    uint32_t Q = rb_get(rb);
    if (rb->bad) return 9;
    if (Q != cfg->num_queries) return 2; /* InvalidProofShape :39-41 */
    /* The PoW witness sits after the queries in the buffer: find it by a dry parse. */
    size_t save = rb->pos;
    unsigned log_max_height = R + cfg->log_blowup;
    for (uint32_t q = 0; q < Q && !rb->bad; q++) {
        uint32_t nb = rb_get(rb);
        for (uint32_t k = 0; k < nb && !rb->bad; k++) {
            if (pass_through) { rb_take(rb, 5); continue; }
            uint32_t nm = rb_get(rb);
            for (uint32_t i = 0; i < nm && !rb->bad; i++) rb_take(rb, rb_get(rb));
            rb_take(rb, (size_t)rb_get(rb) * 8);
        }
        for (uint32_t r = 0; r < R && !rb->bad; r++) {
            rb_take(rb, 8);
            rb_take(rb, (size_t)rb_get(rb) * 8);
        }
    }
    const uint32_t* tail = rb_take(rb, 5);
    if (rb->bad) return 9;
    ef4 final_poly = ef4_load(tail);
    uint32_t pow_witness = tail[4];
    size_t end_pos = rb->pos;
    rb->pos = save;
    if (!ts_or_chal_check_witness(chal, cfg->proof_of_work_bits, pow_witness)) return 3; /* :44-46 */
    size_t* indices = (size_t*)malloc((Q + 1) * sizeof(size_t));
    for (uint32_t q = 0; q < Q; q++) indices[q] = (size_t)ts_or_chal_sample_bits(chal, log_max_height);

    int rc = 0;
    for (uint32_t q = 0; q < Q && rc == 0; q++) { /* verify_challenges :62-98 */
        size_t index = indices[q];
        ts_or_mmcs_tap_select(q); /* verify_batch(query_times_index = q, ...) */
        ef4 ro_by_lh[32];
        int has_ro[32];
        memset(has_ro, 0, sizeof has_ro);
        uint32_t nb = rb_get(rb);
        if (pass_through) {
            for (uint32_t k = 0; k < nb; k++) {
                const uint32_t* p = rb_take(rb, 5);
                if (p[0] > 31) { rc = 9; break; }
                ro_by_lh[p[0]] = ef4_load(p + 1);
                has_ro[p[0]] = 1;
            }
        } else {
            if ((int)nb != n_rounds) { rc = 1; break; }
            ef4 alpha_pow[32];
            for (int i = 0; i < 32; i++) { alpha_pow[i] = ef4_one(); ro_by_lh[i] = ef4_zero(); }
            for (int r = 0; r < n_rounds && rc == 0; r++) { /* two_adic_pcs.rs:463-524 */
                uint32_t nm = rb_get(rb);
                if ((int)nm != rounds[r].n_mats) { rc = 1; break; }
                size_t heights[64], widths[64];
                size_t tot = 0;
                const uint32_t* rows_start = NULL;
                /* opened rows are stored as (width, values...) per matrix: gather */
                uint32_t* rows = NULL;
                size_t save2 = rb->pos;
                for (uint32_t i = 0; i < nm; i++) {
                    uint32_t wd = rb_get(rb);
                    rb_take(rb, wd);
                    if (rb->bad) break;
                    if (wd != rounds[r].mats[i].width) rc = 1;
                    widths[i] = wd;
                    heights[i] = (size_t)1 << (rounds[r].mats[i].dom.log_n + cfg->log_blowup);
                    tot += wd;
                }
                if (rb->bad || rc) { if (!rc) rc = 9; break; }
                rows = (uint32_t*)malloc(tot * 4 + 4);
                rb->pos = save2;
                size_t off = 0;
                for (uint32_t i = 0; i < nm; i++) {
                    uint32_t wd = rb_get(rb);
                    memcpy(rows + off, rb_take(rb, wd), wd * 4);
                    off += wd;
                }
                (void)rows_start;
                uint32_t plen = rb_get(rb);
                const uint32_t* path = rb_take(rb, (size_t)plen * 8);
                if (rb->bad) { free(rows); rc = 9; break; }
                size_t max_h = 0;
                for (uint32_t i = 0; i < nm; i++) if (heights[i] > max_h) max_h = heights[i];
                unsigned log_bmh = ts_log2_strict(max_h);
                unsigned bits_reduced = log_max_height - log_bmh;
                size_t reduced_index = index >> bits_reduced;
                ts_or_mmcs_tap_u32(1);
                ts_or_mmcs_tap_verify_base(lock_base[r]);
                if (!ts_or_mmcs_verify((int)nm, heights, widths, reduced_index, rows, path, plen,
                                       rounds[r].commit)) { free(rows); rc = 4; break; }
                off = 0;
                for (uint32_t i = 0; i < nm; i++) {
                    const mat_claim* mc = &rounds[r].mats[i];
                    unsigned lh = mc->dom.log_n + cfg->log_blowup;
                    unsigned br = log_max_height - lh;
                    size_t rev = ts_bitrev(index >> br, lh);
                    uint32_t x = bb_mul(BB_GENERATOR, bb_pow(bb_two_adic_generator(lh), rev));
                    has_ro[lh] = 1;
                    for (int p = 0; p < mc->n_points; p++) {
                        ef4 z = mc->points[p];
                        ef4 acc = ef4_zero();
                        for (size_t c = 0; c < mc->width; c++) {
                            ef4 diff = ef4_add_base(ef4_neg(mc->values[p][c]), rows[off + c]);
                            acc = ef4_add(acc, ef4_mul(alpha_pow[lh], diff));
                            alpha_pow[lh] = ef4_mul(alpha_pow[lh], alpha);
                        }
                        ef4 den = ef4_add_base(ef4_neg(z), x);
                        ro_by_lh[lh] = ef4_add(ro_by_lh[lh], ef4_div(acc, den));
                    }
                    off += mc->width;
                }
                free(rows);
            }
            if (rc) break;
        }
        /* verify_query, verifier.rs:100-165 */
        ef4 folded_eval = ef4_zero();
        size_t query_index = index;
        for (uint32_t r = 0; r < R; r++) {
            unsigned log_folded_height = log_max_height - 1 - r;
            size_t point_index = query_index & 1;
            size_t index_pair = query_index >> 1;
            if (has_ro[log_folded_height + 1])
                folded_eval = ef4_add(folded_eval, ro_by_lh[log_folded_height + 1]);
            const uint32_t* vals = rb_take(rb, 8);
            uint32_t plen = rb_get(rb);
            const uint32_t* path = rb_take(rb, (size_t)plen * 8);
            if (rb->bad) { rc = 9; break; }
            ef4 committed = ef4_load(vals + 4 * point_index);
            if (log_folded_height < log_max_height - 1 || strict_first_round) {
                if (!ef4_eq(folded_eval, committed)) { rc = 8; break; } /* :139-141 */
            }
            size_t h = (size_t)1 << log_folded_height, w8 = 8;
            ts_or_mmcs_tap_u32(4);
            ts_or_mmcs_tap_verify_base(fri_lock_base + (size_t)r * n_roots() * 3);
            int mm_ok = ts_or_mmcs_verify(1, &h, &w8, index_pair, vals, path, plen, commits + cw * r);
            ts_or_mmcs_tap_u32(1);
            if (!mm_ok) {
                rc = 5;
                break;
            }
            query_index = index_pair;
            folded_eval = fold_row(query_index, log_folded_height, betas[r], ef4_load(vals),
                                   ef4_load(vals + 4));
        }
        if (rc) break;
        if (!ef4_eq(folded_eval, final_poly)) rc = 6; /* :92-94 */
    }
    free(indices);
    if (rc == 0) rb->pos = end_pos;
    return rc;
}

int ts_or_verify(const ts_or_fri_config* cfg, const uint32_t* tape, size_t n_tape,
                 ts_or_challenger* chal, const uint32_t* proof, size_t n_words,
                 const uint32_t* pis) {
    ts_or_tape t;
    if (ts_or_tape_parse(tape, n_tape, &t)) return 9;
    rbuf rb = {proof, n_words, 0, 0};
    if (rb_get(&rb) != TSPF_MAGIC) return 9;
    const uint32_t version = rb_get(&rb);
    if (version != (ts_or_mmcs_tap_queries() ? 2u : 1u)) return 9;
    unsigned degree_bits = rb_get(&rb);
    uint32_t pw = rb_get(&rb), pqd = rb_get(&rb);
    if (version == 2 && rb_get(&rb) != ts_or_mmcs_tap_queries()) return 1;
    if (rb.bad || degree_bits > 27) return 9;
    const size_t cw = 8 * (size_t)n_roots();
    unsigned lqd = (unsigned)ts_or_air_log_quotient_degree(tape, n_tape);
    size_t qd = (size_t)1 << lqd, w = t.width;
    if (pw != w || pqd != qd) return 1; /* verifier.rs:49-59 valid_shape */
    const uint32_t* trace_root = rb_take(&rb, cw);
    const uint32_t* quot_root = rb_take(&rb, cw);
    const ef4* trace_local = (const ef4*)rb_take(&rb, w * 4);
    const ef4* trace_next = (const ef4*)rb_take(&rb, w * 4);
    const ef4* qchunks = (const ef4*)rb_take(&rb, qd * 16);
    if (rb.bad) return 9;

    /* :69-75 */
    observe_roots(chal, trace_root);
    ef4 alpha = chal_sample_ef(chal);
    observe_roots(chal, quot_root);
    ef4 zeta = chal_sample_ef(chal);
    uint32_t gn = bb_two_adic_generator(degree_bits);
    ef4 zeta_next = ef4_mul_base(zeta, gn);

    /* :77-101 */
    ef4 tpts[2] = {zeta, zeta_next};
    const ef4* tvals[2] = {trace_local, trace_next};
    mat_claim tclaim = {{degree_bits, 1}, w, 2, tpts, tvals};
    mat_claim qclaims[64];
    const ef4* qvals[64][1];
    uint32_t gq = bb_two_adic_generator(degree_bits + lqd);
    for (size_t c = 0; c < qd; c++) {
        qvals[c][0] = qchunks + 4 * c;
        qclaims[c].dom.log_n = degree_bits;
        qclaims[c].dom.shift = bb_mul(BB_GENERATOR, bb_pow(gq, c));
        qclaims[c].width = 4;
        qclaims[c].n_points = 1;
        qclaims[c].points = &zeta;
        qclaims[c].values = qvals[c];
    }
    round_claim rounds[2] = {{trace_root, 1, &tclaim}, {quot_root, (int)qd, qclaims}};
    ef4 batch_alpha = chal_sample_ef(chal); /* two_adic_pcs.rs:443 */
    int rc = fri_verify(cfg, 2, rounds, batch_alpha, chal, &rb, 0);
    if (rc) return rc;
    if (rb.pos != rb.len) return 9;

    /* :103-120 zps */
    ef4 zps[64];
    for (size_t i = 0; i < qd; i++) {
        ef4 prod = ef4_one();
        for (size_t j = 0; j < qd; j++) {
            if (j == i) continue;
            /* zp_at_point(z) = (z/shift)^(2^log_n) - 1 (App. A.3) */
            uint32_t sj_inv = bb_inv(qclaims[j].dom.shift);
            ef4 a = ef4_sub_base(ef4_pow(ef4_mul_base(zeta, sj_inv), (uint64_t)1 << degree_bits), 1);
            uint32_t fp = qclaims[i].dom.shift; /* first_point */
            uint32_t bden = bb_sub(bb_pow(bb_mul(fp, sj_inv), (uint64_t)1 << degree_bits), 1);
            prod = ef4_mul(prod, ef4_mul_base(a, bb_inv(bden)));
        }
        zps[i] = prod;
    }
    /* :122-132 quotient = sum_i sum_e zps[i] * x^e * chunk_i[e] */
    ef4 quotient = ef4_zero();
    for (size_t i = 0; i < qd; i++)
        for (int e = 0; e < 4; e++) {
            ef4 mono = ef4_zero();
            mono.c[e] = 1;
            quotient = ef4_add(quotient, ef4_mul(ef4_mul(zps[i], mono), qchunks[4 * i + e]));
        }
    /* :136 selectors_at_point (App. A.4), trace domain shift = 1 */
    ef4 zh = ef4_sub_base(ef4_pow(zeta, (uint64_t)1 << degree_bits), 1);
    uint32_t gn_inv = bb_inv(gn);
    ef4 is_first = ef4_div(zh, ef4_sub_base(zeta, 1));
    ef4 is_last = ef4_div(zh, ef4_sub_base(zeta, gn_inv));
    ef4 is_trans = ef4_sub_base(zeta, gn_inv);
    ef4 inv_zeroifier = ef4_inv(zh);
    /* :138-153 */
    ef4* v = (ef4*)malloc((t.n_nodes + 1) * sizeof(ef4));
    ts_or_tape_eval_ext(&t, trace_local, trace_next, pis, is_first, is_last, is_trans, v);
    ef4 acc = ef4_zero();
    for (uint32_t c = 0; c < t.n_constraints; c++) /* folder.rs:101-105 */
        acc = ef4_add(ef4_mul(acc, alpha), v[t.constraints[c]]);
    free(v);
    if (!ef4_eq(ef4_mul(acc, inv_zeroifier), quotient)) return 7; /* :157 */
    return 0;
}

/* ---------------------------------------------------- fri/tests/pcs.rs shape */
int ts_or_pcs_roundtrip(const ts_or_fri_config* cfg, int n_rounds, const int* mats_per_round,
                        const unsigned* log_degrees, const size_t* widths,
                        const uint32_t* const* evals, int tamper) {
    ts_or_challenger p_chal, v_chal;
    ts_or_chal_init(&p_chal, 0, 1);
    ts_or_chal_init(&v_chal, 0, 1);
    ts_or_mmcs_data* datas[16];
    uint32_t roots[16][8];
    int k = 0;
    for (int r = 0; r < n_rounds; r++) { /* pcs.rs:62-66 */
        dom_t doms[16];
        for (int i = 0; i < mats_per_round[r]; i++) { doms[i].log_n = log_degrees[k + i]; doms[i].shift = 1; }
        datas[r] = pcs_commit(cfg, mats_per_round[r], doms, evals + k, widths + k, roots[r]);
        k += mats_per_round[r];
    }
    for (int r = 0; r < n_rounds; r++) ts_or_chal_observe_digest(&p_chal, roots[r]); /* :69 */
    ef4 zeta = chal_sample_ef(&p_chal);                                                /* :72 */
    open_round rounds[16];
    int npts[16][16];
    const ef4* ppts[16][16];
    size_t total_w = 0;
    for (int r = 0; r < n_rounds; r++) {
        for (int i = 0; i < mats_per_round[r]; i++) { npts[r][i] = 1; ppts[r][i] = &zeta; }
        rounds[r].data = datas[r];
        rounds[r].n_points = npts[r];
        rounds[r].points = ppts[r];
    }
    for (int i = 0; i < k; i++) total_w += widths[i];
    ef4* opened = (ef4*)malloc((total_w + 1) * sizeof(ef4));
    size_t n_opened;
    size_t cap = 1 << 22;
    uint32_t* buf = (uint32_t*)malloc(cap * 4);
    ts_or_wbuf b;
    wb_init(&b, buf, cap);
    int rc = pcs_open(cfg, n_rounds, rounds, &p_chal, opened, &n_opened, &b); /* :80 */
    if (rc == 0 && b.overflow) rc = -1;
    if (rc == 0) {
        if (tamper == 1) opened[0].c[0] = bb_add(opened[0].c[0], 1);
        if (tamper == 2) buf[b.len - 5] = bb_add(buf[b.len - 5], 1); /* final_poly */
        for (int r = 0; r < n_rounds; r++) ts_or_chal_observe_digest(&v_chal, roots[r]);
        ef4 vzeta = chal_sample_ef(&v_chal);
        if (!ef4_eq(vzeta, zeta)) rc = 100;
        round_claim rcl[16];
        mat_claim mcl[16][16];
        const ef4* vals[16][16][1];
        size_t off = 0;
        int kk = 0;
        for (int r = 0; r < n_rounds; r++) {
            for (int i = 0; i < mats_per_round[r]; i++) {
                vals[r][i][0] = opened + off;
                off += widths[kk];
                mcl[r][i].dom.log_n = log_degrees[kk];
                mcl[r][i].dom.shift = 1;
                mcl[r][i].width = widths[kk];
                mcl[r][i].n_points = 1;
                mcl[r][i].points = &vzeta;
                mcl[r][i].values = vals[r][i];
                kk++;
            }
            rcl[r].commit = roots[r];
            rcl[r].n_mats = mats_per_round[r];
            rcl[r].mats = mcl[r];
        }
        if (rc == 0) {
            ef4 alpha = chal_sample_ef(&v_chal);
            rbuf rb = {buf, b.len, 0, 0};
            rc = fri_verify(cfg, n_rounds, rcl, alpha, &v_chal, &rb, 0);
            if (rc == 0 && rb.pos != rb.len) rc = 9;
        }
        if (rc == 0 && ts_or_chal_sample_bits(&p_chal, 8) != ts_or_chal_sample_bits(&v_chal, 8))
            rc = 101;
    }
    free(buf);
    free(opened);
    for (int r = 0; r < n_rounds; r++) ts_or_mmcs_free(datas[r]);
    return rc;
}

/* ------------------------------------------------------ fri/tests/fri.rs shape */
int ts_or_fri_roundtrip(const ts_or_fri_config* cfg, int n_inputs, const unsigned* log_lens,
                        const uint32_t* const* inputs, int sample_ext, int perm_kind) {
    ts_or_challenger p_chal, v_chal;
    ts_or_chal_init(&p_chal, perm_kind, sample_ext);
    ts_or_chal_init(&v_chal, perm_kind, sample_ext);
    size_t cap = 1 << 22;
    uint32_t* buf = (uint32_t*)malloc(cap * 4);
    ts_or_wbuf b;
    wb_init(&b, buf, cap);
    int rc = bf_prove(cfg, n_inputs, (ef4* const*)inputs, log_lens, &p_chal, NULL, 1, &b);
    if (rc == 0 && b.overflow) rc = -1;
    if (rc == 0) {
        rbuf rb = {buf, b.len, 0, 0};
        rc = fri_verify(cfg, 0, NULL, ef4_one(), &v_chal, &rb, 1);
        if (rc == 0 && rb.pos != rb.len) rc = 9;
        /* fri.rs:142-146 */
        if (rc == 0 && ts_or_chal_sample_bits(&p_chal, 8) != ts_or_chal_sample_bits(&v_chal, 8))
            rc = 101;
    }
    free(buf);
    return rc;
}

/* ------------------------------------------------------ Pcs::commit + Pcs::open, exported
 * The pcs.rs flow (fri/tests/pcs.rs:62-90) with everything returned, so that the GPU path can be
 * compared bit for bit: commit every round, observe the commitments, sample zeta, open every matrix
 * at zeta.  roots_out: n_rounds x 8 words; opened_out: sum(widths) x 4 words in (round, matrix,
 * column) order; proof_out: the FriProof part of the TSPF format.  Returns proof words or < 0. */
static int64_t pcs_commit_open_impl(const ts_or_fri_config* cfg, ts_or_challenger* chal, int n_rounds,
                                    const int* mats_per_round, const unsigned* log_degrees,
                                    const size_t* widths, const uint32_t* const* evals, int multi,
                                    uint32_t* roots_out, uint32_t* zeta_out, uint32_t* opened_out,
                                    uint32_t* proof_out, size_t cap_words);
int64_t ts_or_pcs_commit_open(const ts_or_fri_config* cfg, ts_or_challenger* chal, int n_rounds,
                              const int* mats_per_round, const unsigned* log_degrees,
                              const size_t* widths, const uint32_t* const* evals,
                              uint32_t* roots_out, uint32_t* zeta_out, uint32_t* opened_out,
                              uint32_t* proof_out, size_t cap_words) {
    return pcs_commit_open_impl(cfg, chal, n_rounds, mats_per_round, log_degrees, widths, evals, 0,
                                roots_out, zeta_out, opened_out, proof_out, cap_words);
}
/* the same with several points per matrix: matrix k (counted over all rounds) is opened at
 * zeta * 7^j for j < 1 + k % 3 (two_adic_pcs.rs:344-387 loops over any list of points) */
int64_t ts_or_pcs_commit_open_multi(const ts_or_fri_config* cfg, ts_or_challenger* chal, int n_rounds,
                                    const int* mats_per_round, const unsigned* log_degrees,
                                    const size_t* widths, const uint32_t* const* evals,
                                    uint32_t* roots_out, uint32_t* zeta_out, uint32_t* opened_out,
                                    uint32_t* proof_out, size_t cap_words) {
    return pcs_commit_open_impl(cfg, chal, n_rounds, mats_per_round, log_degrees, widths, evals, 1,
                                roots_out, zeta_out, opened_out, proof_out, cap_words);
}
static int64_t pcs_commit_open_impl(const ts_or_fri_config* cfg, ts_or_challenger* chal, int n_rounds,
                                    const int* mats_per_round, const unsigned* log_degrees,
                                    const size_t* widths, const uint32_t* const* evals, int multi,
                                    uint32_t* roots_out, uint32_t* zeta_out, uint32_t* opened_out,
                                    uint32_t* proof_out, size_t cap_words) {
    ts_or_mmcs_data* datas[16];
    int k = 0;
    for (int r = 0; r < n_rounds; r++) {
        dom_t doms[16];
        for (int i = 0; i < mats_per_round[r]; i++) {
            doms[i].log_n = log_degrees[k + i];
            doms[i].shift = 1;
        }
        datas[r] = pcs_commit(cfg, mats_per_round[r], doms, evals + k, widths + k, roots_out + 8 * r);
        k += mats_per_round[r];
    }
    for (int r = 0; r < n_rounds; r++) ts_or_chal_observe_digest(chal, roots_out + 8 * r);
    ef4 zeta = chal_sample_ef(chal);
    memcpy(zeta_out, zeta.c, 16);
    open_round rounds[16];
    int npts[16][16];
    const ef4* ppts[16][16];
    ef4 zpow[3] = {zeta, ef4_mul_base(zeta, 7), ef4_mul_base(zeta, 49)};
    size_t total_w = 0;
    int kk = 0;
    for (int r = 0; r < n_rounds; r++) {
        for (int i = 0; i < mats_per_round[r]; i++, kk++) {
            npts[r][i] = multi ? 1 + kk % 3 : 1;
            ppts[r][i] = zpow;
        }
        rounds[r].data = datas[r];
        rounds[r].n_points = npts[r];
        rounds[r].points = ppts[r];
    }
    for (int i = 0; i < k; i++) total_w += widths[i] * (multi ? 1 + i % 3 : 1);
    ef4* opened = (ef4*)malloc((total_w + 1) * sizeof(ef4));
    size_t n_opened = 0;
    ts_or_wbuf b;
    wb_init(&b, proof_out, cap_words);
    int rc = pcs_open(cfg, n_rounds, rounds, chal, opened, &n_opened, &b);
    memcpy(opened_out, opened, n_opened * 16);
    free(opened);
    for (int r = 0; r < n_rounds; r++) ts_or_mmcs_free(datas[r]);
    if (rc) return rc;
    if (b.overflow) return -1;
    return (int64_t)b.len;
}

/* ------------------------------------------------------ bf_prove / FRI verify alone, exported
 * fri/tests/fri.rs:51-147: FRI over given input vectors (EF4, strictly descending lengths) with the
 * literal reduced openings as the "input opening proof" (:109-118).  The challenger may be of either
 * kind (Blake3 or the test's reverse permutation; F = EF4 or BabyBear embedded). */
int64_t ts_or_fri_prove(const ts_or_fri_config* cfg, ts_or_challenger* chal, int n_inputs,
                        const unsigned* log_lens, const uint32_t* const* inputs,
                        uint32_t* proof_out, size_t cap_words) {
    ts_or_wbuf b;
    wb_init(&b, proof_out, cap_words);
    int rc = bf_prove(cfg, n_inputs, (ef4* const*)inputs, log_lens, chal, NULL, 1, &b);
    if (rc) return rc;
    if (b.overflow) return -1;
    return (int64_t)b.len;
}
int ts_or_fri_verify(const ts_or_fri_config* cfg, ts_or_challenger* chal, const uint32_t* proof,
                     size_t n_words) {
    rbuf rb = {proof, n_words, 0, 0};
    int rc = fri_verify(cfg, 0, NULL, ef4_one(), chal, &rb, 1);
    if (rc == 0 && rb.pos != rb.len) rc = 9;
    return rc;
}
