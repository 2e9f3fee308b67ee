/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the tap-stark uni-stark/fri prover hot path and of the
 * verifier that accepts its proofs.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library; the product (tap-stark_amd/) never does.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - Blake3 and the Fiat-Shamir challenger are PINNED by the reference's own known answers
 *     (scripts/src/hashes/blake3.rs:538,555; script_expr/src/challenger_expr.rs:279-296).
 *   - Every numeric prover stage is "parity unpinned" by the reference (it holds no golden
 *     vector for them and cannot be built here: Rust, un-vendored Plonky3); the stages are exact
 *     field arithmetic, pinned by mathematical uniqueness (naive O(n^2) DFT vs fast NTT,
 *     direct polynomial evaluation vs barycentric) and by the restated verifier accepting.
 *   - The Merkle MMCS (Blake3 leaves/nodes) has NO reference counterpart (the reference uses a
 *     Bitcoin taptree): build-defined spec, pinned only by Blake3 KATs and
 *     commit/open/verify round trips.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef TS_ORACLE_H
#define TS_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "bb.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ Blake3 */
void ts_or_blake3(const uint8_t* in, size_t len, uint8_t out[32]);
void ts_or_blake3_words(const uint32_t* w, size_t n, uint32_t out[8]);

/* --------------------------------------------------------------- challenger */
/* reference basic/src/challenger/mod.rs:67-84 (BfChallenger<F, U32, P, 16>) */
typedef struct {
    uint32_t state[16]; /* sponge_state: 16 x [u8;4], kept as LE u32 */
    uint32_t in_buf[8];
    int n_in;
    uint32_t out_buf[8];
    int n_out;
    int perm_kind;  /* 0 = Blake3Permutation (mod.rs:34-48); 1 = TestPermutation = reverse
                       (fri/tests/fri.rs:35-48) */
    int sample_ext; /* 1: F = EF4 (4 pops per sample); 0: F = BabyBear */
    uint64_t n_perms;
} ts_or_challenger;

void ts_or_chal_init(ts_or_challenger* c, int perm_kind, int sample_ext);
void ts_or_chal_observe(ts_or_challenger* c, uint32_t word);
void ts_or_chal_observe_digest(ts_or_challenger* c, const uint32_t d[8]);
uint32_t ts_or_chal_sample_base(ts_or_challenger* c);
void ts_or_chal_sample_ext(ts_or_challenger* c, uint32_t out[4]);
/* sample(): EF4 or base according to sample_ext; always writes 4 words (c1..c3 = 0 for base) */
void ts_or_chal_sample(ts_or_challenger* c, uint32_t out[4]);
uint64_t ts_or_chal_sample_bits(ts_or_challenger* c, unsigned bits);
int ts_or_chal_check_witness(ts_or_challenger* c, unsigned bits, uint32_t witness);
/* returns 0 and the smallest witness in [0,4096), or -1 ("failed to find witness") */
int ts_or_chal_grind(ts_or_challenger* c, unsigned bits, uint32_t* witness);

/* --------------------------------------------------------------------- DFT */
/* all matrices row-major, canonical u32 */
void ts_or_naive_dft(const uint32_t* in, uint32_t* out, size_t n, size_t w, int inverse);
void ts_or_dft_batch(uint32_t* m, size_t n, size_t w, int inverse);
void ts_or_coset_lde_batch(const uint32_t* evals, size_t n, size_t w, unsigned added_bits,
                           uint32_t shift, uint32_t* out);
void ts_or_bit_reverse_rows(uint32_t* m, size_t h, size_t w);
/* Pcs::commit's LDE step (fri/src/two_adic_pcs.rs:233-241): out = N x w, bit-reversed rows */
void ts_or_commit_lde(const uint32_t* evals, unsigned log_n, size_t w, uint32_t domain_shift,
                      unsigned log_blowup, uint32_t* out);
/* direct evaluation of the interpolant of a column at one base point (pinning helper) */
uint32_t ts_or_eval_interpolant_naive(const uint32_t* col_evals, size_t n, uint32_t domain_shift,
                                      uint32_t x);

/* -------------------------------------------------------------------- MMCS */
typedef struct ts_or_mmcs_data ts_or_mmcs_data;
ts_or_mmcs_data* ts_or_mmcs_commit(int n_mats, const uint32_t* const* mats, const size_t* heights,
                                   const size_t* widths, uint32_t root[8]);
void ts_or_mmcs_free(ts_or_mmcs_data* d);
unsigned ts_or_mmcs_log_max_height(const ts_or_mmcs_data* d);
/* rows_out: concatenated opened rows in matrix order; path_out: log_max_height x 8 words */
void ts_or_mmcs_open(const ts_or_mmcs_data* d, size_t index, uint32_t* rows_out,
                     uint32_t* path_out);
int ts_or_mmcs_verify(int n_mats, const size_t* heights, const size_t* widths, size_t index,
                      const uint32_t* rows, const uint32_t* path, size_t path_len,
                      const uint32_t root[8]);
const uint32_t* ts_or_mmcs_layer(const ts_or_mmcs_data* d, unsigned level);

/* ---------------------------------------------------------------- AIR tape */
/* Serialised SymbolicExpression DAG (uni-stark/src/symbolic_expression.rs:12-37,
 * symbolic_variable.rs:9-15).  Words:
 *   [0]=TS_TAPE_MAGIC [1]=version(1) [2]=width [3]=n_public [4]=n_nodes [5]=n_constraints
 *   then n_nodes x {op, a, b}, then n_constraints node ids (assert_zero call order). */
#define TS_TAPE_MAGIC 0x54415354u /* 'TSAT' */
enum {
    TS_OP_CONST = 0,      /* a = canonical value */
    TS_OP_MAIN = 1,       /* a = offset (0 local, 1 next), b = column */
    TS_OP_PUBLIC = 2,     /* a = index */
    TS_OP_IS_FIRST = 3,
    TS_OP_IS_LAST = 4,
    TS_OP_IS_TRANSITION = 5,
    TS_OP_ADD = 6, /* a, b = node ids (< own id) */
    TS_OP_SUB = 7,
    TS_OP_NEG = 8, /* a */
    TS_OP_MUL = 9
};
int ts_or_tape_validate(const uint32_t* tape, size_t n_words);
int ts_or_air_max_constraint_degree(const uint32_t* tape, size_t n_words);
int ts_or_air_log_quotient_degree(const uint32_t* tape, size_t n_words);
/* uni-stark/src/check_constraints.rs:11-39; returns -1 if all hold, else row*2^16+constraint */
int ts_or_tape_constraint_values(const uint32_t* tape, size_t n_words, const uint32_t* local,
                                 const uint32_t* next, size_t m, const uint32_t* pis,
                                 const uint32_t* sels, uint32_t* out);
void ts_or_set_debug_assertions(int on); /* prover.rs:40-41; default on */
int64_t ts_or_check_constraints(const uint32_t* tape, size_t n_words, const uint32_t* trace,
                                size_t n, const uint32_t* pis);

/* -------------------------------------------------------------- STARK stages */
typedef struct {
    uint32_t log_blowup;
    uint32_t num_queries;
    uint32_t proof_of_work_bits;
} ts_or_fri_config;

/* uni-stark/src/prover.rs:122-194: quotient values in natural order over the quotient domain.
 * lde = committed trace LDE (N x w, bit-reversed rows); out = n*qd x 4 words */
void ts_or_quotient_values(const uint32_t* tape, size_t n_words, const uint32_t* lde,
                           unsigned log_n, unsigned log_blowup, const uint32_t* pis,
                           const uint32_t alpha[4], uint32_t* out);
/* prover.rs:78-80: flatten_to_base + split_evals: out[c] = n x 4 (chunk c = rows r%qd==c) */
void ts_or_split_quotient(const uint32_t* qvals, unsigned log_n, unsigned log_qd, uint32_t* out);
/* fri/src/two_adic_pcs.rs:116-147 fold_matrix on a vector of EF (len = 2h -> h) */
void ts_or_fold_matrix(const uint32_t* in, size_t h, const uint32_t beta[4], uint32_t* out);
/* two_adic_pcs.rs:87-114 fold_row */
void ts_or_fold_row(size_t index, unsigned log_height, const uint32_t beta[4],
                    const uint32_t e0[4], const uint32_t e1[4], uint32_t out[4]);
/* two_adic_pcs.rs:260-389 for the prove() shape: trace LDE (N x w) opened at zeta and
 * zeta*omega_n, qd chunk LDEs (N x 4 each) at zeta.
 * opened_out = (2w + 4qd) x 4 words; ro_out = N x 4 words. */
void ts_or_open_reduce(const uint32_t* trace_lde, size_t w, const uint32_t* const* chunk_ldes,
                       unsigned log_qd, unsigned log_n, unsigned log_blowup,
                       const uint32_t zeta[4], const uint32_t alpha[4], uint32_t* opened_out,
                       uint32_t* ro_out);

/* ------------------------------------------------------------- whole proofs */
/* Proof wire format "TSPF v1": see DESIGN.md.  Returns the number of u32 words written, or a
 * negative error: -1 buffer too small, -2 bad tape, -3 constraints violated, -4 grind failed,
 * -5 FRI final-poly assertion (fri/src/prover.rs:130-134). */
int64_t ts_or_prove(const ts_or_fri_config* cfg, const uint32_t* tape, size_t n_tape,
                    ts_or_challenger* chal, const uint32_t* trace, unsigned log_n,
                    const uint32_t* pis, uint32_t* proof_out, size_t cap_words);

/* uni-stark/src/verifier.rs:19-161.  0 = accept; otherwise an error code:
 * 1 InvalidProofShape, 2 InvalidOpeningArgument(FRI shape), 3 InvalidPowWitness,
 * 4 MMCS verify failed (input), 5 MMCS verify failed (commit phase), 6 FinalPolyMismatch,
 * 7 OodEvaluationMismatch, 8 folded-eval mismatch (fri/src/verifier.rs:139-141 assert),
 * 9 malformed buffer. */
int ts_or_verify(const ts_or_fri_config* cfg, const uint32_t* tape, size_t n_tape,
                 ts_or_challenger* chal, const uint32_t* proof, size_t n_words,
                 const uint32_t* pis);

/* stage dump of the most recent ts_or_prove in this thread (challenges, for stage tests):
 * words: alpha[4] zeta[4] batch_alpha[4] n_betas betas[4*n] pow_witness n_q indices[n_q] */
size_t ts_or_last_transcript(uint32_t* out, size_t cap);

/* ----------------------------------------------- generic PCS (fri/tests/pcs.rs) */
/* One "round" = one commit() batch.  mats[i]: log_n, width, evals (natural order, on the
 * natural domain shift=1).  All matrices of all rounds are opened at the single point zeta
 * sampled after observing the commitments (pcs.rs:79-90).  Returns 0 iff the restated
 * Pcs::verify accepts the proof produced by the restated Pcs::open and the prover's and
 * verifier's transcripts agree afterwards. */
int ts_or_pcs_roundtrip(const ts_or_fri_config* cfg, int n_rounds, const int* mats_per_round,
                        const unsigned* log_degrees, const size_t* widths,
                        const uint32_t* const* evals, int tamper);

/* fri/tests/fri.rs:52-147: FRI over plain reduced-opening vectors with a pass-through input
 * proof and TestPermutation.  inputs: descending power-of-two lengths, EF elements
 * (base_field!=0: F = BabyBear vectors stored as 1 word each).  Returns 0 iff verify accepts and
 * transcripts agree. */
int ts_or_fri_roundtrip(const ts_or_fri_config* cfg, int n_inputs, const unsigned* log_lens,
                        const uint32_t* const* inputs, int base_field, int perm_kind);

/* fri/tests/pcs.rs:62-90 with everything returned (commit every round, observe, sample zeta, open
 * every matrix at zeta); see stark.c */
int64_t ts_or_pcs_commit_open(const ts_or_fri_config* cfg, ts_or_challenger* chal, int n_rounds,
                              const int* mats_per_round, const unsigned* log_degrees,
                              const size_t* widths, const uint32_t* const* evals,
                              uint32_t* roots_out, uint32_t* zeta_out, uint32_t* opened_out,
                              uint32_t* proof_out, size_t cap_words);

/* the same with 1 + k % 3 points zeta * 7^j for matrix k (counted over all rounds) */
int64_t ts_or_pcs_commit_open_multi(const ts_or_fri_config* cfg, ts_or_challenger* chal, int n_rounds,
                                    const int* mats_per_round, const unsigned* log_degrees,
                                    const size_t* widths, const uint32_t* const* evals,
                                    uint32_t* roots_out, uint32_t* zeta_out, uint32_t* opened_out,
                                    uint32_t* proof_out, size_t cap_words);

/* fri/tests/fri.rs:51-147: bf_prove over given EF4 vectors with pass-through input openings, and
 * the matching verifier; see stark.c */
int64_t ts_or_fri_prove(const ts_or_fri_config* cfg, ts_or_challenger* chal, int n_inputs,
                        const unsigned* log_lens, const uint32_t* const* inputs,
                        uint32_t* proof_out, size_t cap_words);
int ts_or_fri_verify(const ts_or_fri_config* cfg, ts_or_challenger* chal, const uint32_t* proof,
                     size_t n_words);

/* taptree mode of the MMCS (mmcs.c): see there.  Q = 0 switches it off. */
void ts_or_mmcs_set_taptree(uint32_t Q, const uint8_t* lock_bytes, const uint64_t* lock_offsets);
uint32_t ts_or_mmcs_tap_queries(void);
void ts_or_mmcs_tap_u32(uint32_t u32_size);
void ts_or_mmcs_tap_select(uint32_t q);
void ts_or_mmcs_tap_verify_base(size_t first_lock);
uint32_t ts_or_mmcs_n_roots(const ts_or_mmcs_data* d);
const uint32_t* ts_or_mmcs_root(const ts_or_mmcs_data* d, uint32_t q);

/* ------------------------------------------------------------------ taptree commitment (taptree.c)
 * reference basic/src/tcs/{mod,builder,complete_taptree}.rs; hashing per BIP-340/341 */
void ts_or_sha256(const uint8_t* in, size_t len, uint8_t out[32]);
void ts_or_tagged_hash(const char* tag, const uint8_t* msg, size_t len, uint8_t out[32]);
void ts_or_tapleaf_hash(const uint8_t* script, size_t len, unsigned version, uint8_t out[32]);
void ts_or_tapbranch(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
size_t ts_or_script_push_int(uint64_t v, uint8_t* out);
size_t ts_or_tap_leaf_script(const uint8_t* const* locks, const size_t* lock_lens, uint64_t index,
                             const uint32_t* values, uint32_t n_evals, uint32_t u32_size, uint8_t* out,
                             size_t cap);
size_t ts_or_padding_matrix(int n_mats, const uint32_t* const* mats, const size_t* heights,
                            const size_t* widths, uint32_t* out);
void ts_or_taptree_build(size_t n, const uint8_t* leaf_hashes, uint8_t* nodes, size_t* leaf_indices);
void ts_or_taptree_path(size_t n, const uint8_t* nodes, size_t index, uint8_t* path);
int ts_or_taptree_verify_inclusion(const uint8_t root[32], const uint8_t leaf[32], const uint8_t* path,
                                   size_t depth);
int ts_or_tap_commit_polys(int n_mats, const uint32_t* const* mats, const size_t* heights,
                           const size_t* widths, uint32_t u32_size, const uint8_t* const* locks,
                           const size_t* lock_lens, uint8_t* nodes);

#ifdef __cplusplus
}
#endif
#endif
