/*
 * oracle/challenger.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of the reference's Fiat-Shamir duplex sponge
 *   basic/src/challenger/mod.rs:23-49   Blake3Permutation
 *   basic/src/challenger/mod.rs:67-84   BfChallenger state
 *   basic/src/challenger/mod.rs:95-114  grind / check_witness
 *   basic/src/challenger/mod.rs:151-174 duplexing
 *   basic/src/challenger/mod.rs:183-194 observe
 *   basic/src/challenger/mod.rs:261-313 sample (base / EF4)
 *   basic/src/challenger/mod.rs:341-348 sample_bits
 *   basic/src/challenger/chan_field.rs:12-18 from_pf ; :35-42 mod_p (= 1 << 12 for U32)
 * Pinned by script_expr/src/challenger_expr.rs:279-296 (second sample == 1103171332).
 */
#include "oracle.h"
#include <string.h>

void ts_or_chal_init(ts_or_challenger* c, int perm_kind, int sample_ext) {
    memset(c, 0, sizeof *c);
    c->perm_kind = perm_kind;
    c->sample_ext = sample_ext;
}

/* mod.rs:34-48: hash the 16 words as 64 LE bytes; state[0..8] = 0; state[8..16] = digest.
 * fri/tests/fri.rs:43-45: TestPermutation reverses the 16 words. */
static void permute(ts_or_challenger* c) {
    if (c->perm_kind == 0) {
        uint8_t digest[32];
        ts_or_blake3((const uint8_t*)c->state, 64, digest);
        memset(c->state, 0, 32);
        memcpy(&c->state[8], digest, 32);
    } else {
        for (int i = 0; i < 8; i++) {
            uint32_t t = c->state[i];
            c->state[i] = c->state[15 - i];
            c->state[15 - i] = t;
        }
    }
    c->n_perms++;
}

/* mod.rs:151-174 */
static void duplexing(ts_or_challenger* c) {
    for (int i = 0; i < c->n_in; i++) c->state[i] = c->in_buf[i];
    c->n_in = 0;
    permute(c);
    c->n_out = 8;
    for (int i = 0; i < 8; i++) c->out_buf[i] = c->state[8 + i];
}

/* mod.rs:183-194 */
void ts_or_chal_observe(ts_or_challenger* c, uint32_t word) {
    c->n_out = 0; /* any buffered output is now invalid */
    c->in_buf[c->n_in++] = word;
    if (c->n_in == 8) duplexing(c);
}

/* mod.rs:197-223: a commitment [[u8;4];8] is observed word by word */
void ts_or_chal_observe_digest(ts_or_challenger* c, const uint32_t d[8]) {
    for (int i = 0; i < 8; i++) ts_or_chal_observe(c, d[i]);
}

/* one pop: mod.rs:269-279 / 289-300 ; chan_field.rs:12-18 (u32 LE % p) */
static uint32_t pop_base(ts_or_challenger* c) {
    if (c->n_in != 0 || c->n_out == 0) duplexing(c);
    uint32_t v = c->out_buf[--c->n_out]; /* Vec::pop => from the end */
    return v % BB_P;
}

uint32_t ts_or_chal_sample_base(ts_or_challenger* c) { return pop_base(c); }

void ts_or_chal_sample_ext(ts_or_challenger* c, uint32_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = pop_base(c);
}

void ts_or_chal_sample(ts_or_challenger* c, uint32_t out[4]) {
    if (c->sample_ext) {
        ts_or_chal_sample_ext(c, out);
    } else {
        out[0] = pop_base(c);
        out[1] = out[2] = out[3] = 0;
    }
}

/* mod.rs:341-348: sample a full challenge, take c0 canonical, >> (32 - bits) */
uint64_t ts_or_chal_sample_bits(ts_or_challenger* c, unsigned bits) {
    uint32_t s[4];
    ts_or_chal_sample(c, s);
    return bits == 0 ? 0 : ((uint64_t)s[0] >> (32 - bits));
}

/* mod.rs:108-114 */
int ts_or_chal_check_witness(ts_or_challenger* c, unsigned bits, uint32_t witness) {
    ts_or_chal_observe(c, witness);
    for (int i = 0; i < 7; i++) ts_or_chal_observe(c, 0);
    return ts_or_chal_sample_bits(c, bits) == 0;
}

/* mod.rs:95-105: serial maybe-rayon => find_any == find => smallest witness (App. A.8) */
int ts_or_chal_grind(ts_or_challenger* c, unsigned bits, uint32_t* witness) {
    for (uint32_t w = 0; w < (1u << 12); w++) {
        ts_or_challenger clone = *c;
        if (ts_or_chal_check_witness(&clone, bits, w)) {
            int ok = ts_or_chal_check_witness(c, bits, w);
            (void)ok;
            *witness = w;
            return 0;
        }
    }
    return -1;
}
