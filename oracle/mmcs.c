/*
 * oracle/mmcs.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Blake3 Merkle "mixed matrix commitment scheme" behind the reference's BFMmcs shape
 *   basic/src/mmcs/bf_mmcs.rs:17-68        trait: commit / open_batch / verify_batch
 *   basic/src/mmcs/taptree_mmcs.rs:46-75   open_batch index semantics
 *     (row `index >> (log_max_height - log_height)` of every matrix)
 *
 * PARITY UNPINNED: the reference's own MMCS is a Bitcoin taptree with no
 * Blake3 Merkle tree anywhere; this spec is build-defined (mirroring
 * upstream Plonky3 FieldMerkleTreeMmcs + SerializingHasher32<Blake3> +
 * CompressionFunctionFromHasher<Blake3,2,32>):
 *   leaf digest  = Blake3(row of matrix 0 || row of matrix 1 || ...), the tallest matrices in
 *                  commit order, each element as canonical u32 LE (EF4 = its 4 coefficients,
 *                  basic/src/field/mod.rs:48-63)
 *   node digest  = Blake3(left || right)
 *   shorter matrices are injected at the layer whose size equals their height:
 *                  node = Blake3( Blake3(left||right) || Blake3(rows of those matrices) )
 *   commitment   = one root [[u8;4];8]; proof = sibling digests, leaf level first.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

struct ts_or_mmcs_data {
    int n_mats;
    size_t* heights;
    size_t* widths;
    uint32_t** mats;
    unsigned log_max_h;
    uint32_t** layers; /* layers[l] has (max_h >> l) digests of 8 words */
    /* taptree mode (below): Q trees of (2 max_h - 1) x 32 bytes each, and the Q roots as words */
    uint32_t tap_q;
    uint8_t* tap_nodes;
    uint32_t* tap_roots;
};

/* ------------------------------------------------------------------ taptree mode
 * The reference's real BFMmcs (basic/src/mmcs/taptree_mmcs.rs:24-119): num_queries taptrees per
 * commitment (taptree.c).  While the mode is on, ts_or_mmcs_commit / _open / _verify speak taptree:
 *   - commit k (in call order) takes lock scripts [cursor, cursor + Q (1 + n_evals)) of the table,
 *     n_evals = total width / u32_size, u32_size as last set by ts_or_mmcs_tap_u32 (1 for BabyBear
 *     matrices, 4 for the EF4 matrices of the FRI commit phase: tcs/mod.rs:239-246);
 *   - open/verify use tree `cur_q` (taptree_mmcs.rs:46-75: open_batch(query_times_index, ...));
 *   - a digest is 8 words = its 32 bytes read as little-endian u32 (chan_field.rs:87-95 u256_to_u32). */
static __thread struct {
    uint32_t Q;
    const uint8_t* bytes;
    const uint64_t* offsets;
    size_t cursor;       /* next lock script of the table (prover side) */
    uint32_t u32_size;
    uint32_t cur_q;
    size_t verify_base;  /* first lock script of the commitment being verified */
} g_tap;

void ts_or_mmcs_set_taptree(uint32_t Q, const uint8_t* lock_bytes, const uint64_t* lock_offsets) {
    g_tap.Q = Q;
    g_tap.bytes = lock_bytes;
    g_tap.offsets = lock_offsets;
    g_tap.cursor = 0;
    g_tap.u32_size = 1;
    g_tap.cur_q = 0;
    g_tap.verify_base = 0;
}
uint32_t ts_or_mmcs_tap_queries(void) { return g_tap.Q; }
void ts_or_mmcs_tap_u32(uint32_t u32_size) { g_tap.u32_size = u32_size; }
void ts_or_mmcs_tap_select(uint32_t q) { g_tap.cur_q = q; }
void ts_or_mmcs_tap_verify_base(size_t first_lock) { g_tap.verify_base = first_lock; }
uint32_t ts_or_mmcs_n_roots(const ts_or_mmcs_data* d) { return d->tap_q ? d->tap_q : 1; }
const uint32_t* ts_or_mmcs_root(const ts_or_mmcs_data* d, uint32_t q) {
    return d->tap_q ? d->tap_roots + 8 * (size_t)q : d->layers[d->log_max_h];
}
static void bytes_to_words(const uint8_t* b, uint32_t* w) {
    for (int k = 0; k < 8; k++)
        w[k] = (uint32_t)b[4 * k] | (uint32_t)b[4 * k + 1] << 8 | (uint32_t)b[4 * k + 2] << 16 |
               (uint32_t)b[4 * k + 3] << 24;
}
static void words_to_bytes(const uint32_t* w, uint8_t* b) {
    for (int k = 0; k < 8; k++)
        for (int j = 0; j < 4; j++) b[4 * k + j] = (uint8_t)(w[k] >> (8 * j));
}
static void tap_locks(size_t first, uint32_t n, const uint8_t** ptrs, size_t* lens) {
    for (uint32_t s = 0; s < n; s++) {
        ptrs[s] = g_tap.bytes + g_tap.offsets[first + s];
        lens[s] = (size_t)(g_tap.offsets[first + s + 1] - g_tap.offsets[first + s]);
    }
}

static void compress2(const uint32_t* l, const uint32_t* r, uint32_t out[8]) {
    uint32_t buf[16];
    memcpy(buf, l, 32);
    memcpy(buf + 8, r, 32);
    ts_or_blake3_words(buf, 16, out);
}

/* hash the concatenation of row `r` of every matrix whose height == h (commit order) */
static int hash_rows_of_height(int n_mats, const uint32_t* const* mats, const size_t* heights,
                               const size_t* widths, size_t h, size_t r, uint32_t out[8]) {
    size_t total = 0;
    for (int i = 0; i < n_mats; i++)
        if (heights[i] == h) total += widths[i];
    if (total == 0) return 0;
    uint32_t* buf = (uint32_t*)malloc(total * 4);
    size_t off = 0;
    for (int i = 0; i < n_mats; i++)
        if (heights[i] == h) {
            memcpy(buf + off, mats[i] + r * widths[i], widths[i] * 4);
            off += widths[i];
        }
    ts_or_blake3_words(buf, total, out);
    free(buf);
    return 1;
}

ts_or_mmcs_data* ts_or_mmcs_commit(int n_mats, const uint32_t* const* mats, const size_t* heights,
                                   const size_t* widths, uint32_t root[8]) {
    ts_or_mmcs_data* d = (ts_or_mmcs_data*)calloc(1, sizeof *d);
    d->n_mats = n_mats;
    d->heights = (size_t*)malloc(n_mats * sizeof(size_t));
    d->widths = (size_t*)malloc(n_mats * sizeof(size_t));
    d->mats = (uint32_t**)malloc(n_mats * sizeof(uint32_t*));
    size_t max_h = 0;
    for (int i = 0; i < n_mats; i++) {
        d->heights[i] = heights[i];
        d->widths[i] = widths[i];
        size_t bytes = heights[i] * widths[i] * 4;
        d->mats[i] = (uint32_t*)malloc(bytes ? bytes : 4);
        memcpy(d->mats[i], mats[i], bytes);
        if (heights[i] > max_h) max_h = heights[i];
    }
    d->log_max_h = ts_log2_strict(max_h);
    if (g_tap.Q) { /* tcs/mod.rs:284-292 commit_poly_with_query_times: Q x commit_polys */
        size_t total = 0;
        for (int i = 0; i < n_mats; i++) total += widths[i];
        uint32_t n_evals = (uint32_t)(total / g_tap.u32_size);
        size_t nodes_per = (2 * max_h - 1) * 32;
        d->tap_q = g_tap.Q;
        d->tap_nodes = (uint8_t*)malloc(nodes_per * g_tap.Q);
        d->tap_roots = (uint32_t*)malloc(32 * (size_t)g_tap.Q);
        const uint8_t** lp = (const uint8_t**)malloc((n_evals + 1) * sizeof(void*));
        size_t* ll = (size_t*)malloc((n_evals + 1) * sizeof(size_t));
        for (uint32_t q = 0; q < g_tap.Q; q++) {
            tap_locks(g_tap.cursor, n_evals + 1, lp, ll);
            g_tap.cursor += n_evals + 1;
            ts_or_tap_commit_polys(n_mats, (const uint32_t* const*)d->mats, d->heights, d->widths,
                                   g_tap.u32_size, lp, ll, d->tap_nodes + nodes_per * q);
            bytes_to_words(d->tap_nodes + nodes_per * q + nodes_per - 32, d->tap_roots + 8 * (size_t)q);
        }
        free(lp);
        free(ll);
        d->layers = (uint32_t**)calloc(d->log_max_h + 1, sizeof(uint32_t*));
        memcpy(root, d->tap_roots, 32);
        return d;
    }
    d->layers = (uint32_t**)malloc((d->log_max_h + 1) * sizeof(uint32_t*));
    d->layers[0] = (uint32_t*)malloc(max_h * 32);
#pragma omp parallel for schedule(static) if (max_h > 4096)
    for (size_t r = 0; r < max_h; r++)
        hash_rows_of_height(n_mats, (const uint32_t* const*)d->mats, d->heights, d->widths, max_h,
                            r, d->layers[0] + 8 * r);
    for (unsigned l = 1; l <= d->log_max_h; l++) {
        size_t sz = max_h >> l;
        d->layers[l] = (uint32_t*)malloc(sz * 32);
#pragma omp parallel for schedule(static) if (sz > 4096)
        for (size_t i = 0; i < sz; i++) {
            uint32_t node[8], inj[8];
            compress2(d->layers[l - 1] + 16 * i, d->layers[l - 1] + 16 * i + 8, node);
            if (hash_rows_of_height(n_mats, (const uint32_t* const*)d->mats, d->heights,
                                    d->widths, sz, i, inj))
                compress2(node, inj, d->layers[l] + 8 * i);
            else
                memcpy(d->layers[l] + 8 * i, node, 32);
        }
    }
    memcpy(root, d->layers[d->log_max_h], 32);
    return d;
}

void ts_or_mmcs_free(ts_or_mmcs_data* d) {
    if (!d) return;
    for (int i = 0; i < d->n_mats; i++) free(d->mats[i]);
    for (unsigned l = 0; l <= d->log_max_h; l++) free(d->layers[l]);
    free(d->tap_nodes);
    free(d->tap_roots);
    free(d->mats);
    free(d->layers);
    free(d->heights);
    free(d->widths);
    free(d);
}

unsigned ts_or_mmcs_log_max_height(const ts_or_mmcs_data* d) { return d->log_max_h; }
const uint32_t* ts_or_mmcs_layer(const ts_or_mmcs_data* d, unsigned level) {
    return d->layers[level];
}
const uint32_t* ts_or_mmcs_matrix(const ts_or_mmcs_data* d, int i) { return d->mats[i]; }
size_t ts_or_mmcs_height(const ts_or_mmcs_data* d, int i) { return d->heights[i]; }
size_t ts_or_mmcs_width(const ts_or_mmcs_data* d, int i) { return d->widths[i]; }
int ts_or_mmcs_n_mats(const ts_or_mmcs_data* d) { return d->n_mats; }

/* bf_mmcs.rs:37-42 + taptree_mmcs.rs:46-63: row index >> (log_max_h - log_h) per matrix */
void ts_or_mmcs_open(const ts_or_mmcs_data* d, size_t index, uint32_t* rows_out,
                     uint32_t* path_out) {
    size_t off = 0;
    for (int i = 0; i < d->n_mats; i++) {
        unsigned lh = ts_log2_strict(d->heights[i]);
        size_t r = index >> (d->log_max_h - lh);
        memcpy(rows_out + off, d->mats[i] + r * d->widths[i], d->widths[i] * 4);
        off += d->widths[i];
    }
    if (d->tap_q) {
        size_t max_h = (size_t)1 << d->log_max_h;
        uint8_t* pb = (uint8_t*)malloc(32 * (size_t)d->log_max_h + 32);
        ts_or_taptree_path(max_h, d->tap_nodes + (2 * max_h - 1) * 32 * (size_t)g_tap.cur_q, index, pb);
        for (unsigned l = 0; l < d->log_max_h; l++) bytes_to_words(pb + 32 * l, path_out + 8 * l);
        free(pb);
        return;
    }
    for (unsigned l = 0; l < d->log_max_h; l++)
        memcpy(path_out + 8 * l, d->layers[l] + 8 * ((index >> l) ^ 1), 32);
}

int ts_or_mmcs_verify(int n_mats, const size_t* heights, const size_t* widths, size_t index,
                      const uint32_t* rows, const uint32_t* path, size_t path_len,
                      const uint32_t root[8]) {
    size_t max_h = 0;
    for (int i = 0; i < n_mats; i++)
        if (heights[i] > max_h) max_h = heights[i];
    unsigned log_max_h = ts_log2_strict(max_h);
    if (path_len != log_max_h) return 0;
    if (index >> log_max_h) return 0;
    if (g_tap.Q) {
        /* taptree_mmcs.rs:77-99 verify_batch -> tcs/mod.rs:425-436: rebuild the leaf from the opened
         * values (they must come tallest matrix first, taptree_mmcs.rs:68-72), check its inclusion
         * under root `cur_q` of the commitment (`root` points at the Q roots).  Script execution
         * against a witness (tcs/mod.rs:143-147) is not restated. */
        size_t total = 0;
        for (int i = 0; i < n_mats; i++) {
            if (i && heights[i] > heights[i - 1]) return 0;
            total += widths[i];
        }
        uint32_t n_evals = (uint32_t)(total / g_tap.u32_size);
        const uint8_t** lp = (const uint8_t**)malloc((n_evals + 1) * sizeof(void*));
        size_t* ll = (size_t*)malloc((n_evals + 1) * sizeof(size_t));
        tap_locks(g_tap.verify_base + (size_t)g_tap.cur_q * (n_evals + 1), n_evals + 1, lp, ll);
        size_t len = ts_or_tap_leaf_script(lp, ll, index, rows, n_evals, g_tap.u32_size, NULL, 0);
        uint8_t* sc = (uint8_t*)malloc(len + 1);
        ts_or_tap_leaf_script(lp, ll, index, rows, n_evals, g_tap.u32_size, sc, len);
        uint8_t leaf[32], rb[32];
        ts_or_tapleaf_hash(sc, len, 0xc0, leaf);
        uint8_t* pb = (uint8_t*)malloc(32 * path_len + 32);
        for (size_t l = 0; l < path_len; l++) words_to_bytes(path + 8 * l, pb + 32 * l);
        words_to_bytes(root + 8 * (size_t)g_tap.cur_q, rb);
        int ok = ts_or_taptree_verify_inclusion(rb, leaf, pb, path_len);
        free(pb); free(sc); free(lp); free(ll);
        return ok;
    }
    /* row pointers into the concatenated opened rows (each "matrix" has one row here) */
    const uint32_t** rp = (const uint32_t**)malloc(n_mats * sizeof(uint32_t*));
    size_t* one_h = (size_t*)malloc(n_mats * sizeof(size_t));
    size_t off = 0;
    for (int i = 0; i < n_mats; i++) {
        rp[i] = rows + off;
        off += widths[i];
        one_h[i] = heights[i];
    }
    uint32_t cur[8];
    /* treat each opened row as a height-tagged single-row matrix: row index 0 */
    {
        size_t total = 0;
        for (int i = 0; i < n_mats; i++)
            if (heights[i] == max_h) total += widths[i];
        uint32_t* buf = (uint32_t*)malloc(total * 4 + 4);
        size_t o = 0;
        for (int i = 0; i < n_mats; i++)
            if (heights[i] == max_h) {
                memcpy(buf + o, rp[i], widths[i] * 4);
                o += widths[i];
            }
        ts_or_blake3_words(buf, total, cur);
        free(buf);
    }
    size_t idx = index;
    for (unsigned l = 0; l < log_max_h; l++) {
        uint32_t nxt[8];
        if (idx & 1)
            compress2(path + 8 * l, cur, nxt);
        else
            compress2(cur, path + 8 * l, nxt);
        idx >>= 1;
        size_t sz = max_h >> (l + 1);
        size_t total = 0;
        for (int i = 0; i < n_mats; i++)
            if (heights[i] == sz) total += widths[i];
        if (total) {
            uint32_t* buf = (uint32_t*)malloc(total * 4);
            size_t o = 0;
            for (int i = 0; i < n_mats; i++)
                if (heights[i] == sz) {
                    memcpy(buf + o, rp[i], widths[i] * 4);
                    o += widths[i];
                }
            uint32_t inj[8];
            ts_or_blake3_words(buf, total, inj);
            free(buf);
            compress2(nxt, inj, cur);
        } else {
            memcpy(cur, nxt, 32);
        }
    }
    free(rp);
    free(one_h);
    return memcmp(cur, root, 32) == 0;
}
