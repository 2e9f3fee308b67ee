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
};

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
