// Host-callable launchers of the HIP kernels (one translation unit per kernel family).
#pragma once
#include <stdint.h>

#include "bb.hpp"
#include "context.hpp"

namespace ts {

// Column-major device matrix: element (row r, column c) at d[c * col_stride + r].
struct ColMat {
    uint32_t* d = nullptr;
    uint64_t height = 0;
    uint32_t width = 0;
    uint64_t col_stride = 0;
};

// ---- ntt.hip ---------------------------------------------------------------------------------
void launch_build_twiddles(Context& ctx, uint32_t* W, uint32_t* Winv, unsigned log_size);
// per-coset scale tables: lo[beta][j] = s_beta^j * scale (j < 1024), hi[beta][j] = s_beta^(1024 j)
void launch_build_shift_tables(Context& ctx, uint32_t* lo, uint32_t* hi, uint32_t n_hi,
                               uint32_t n_cosets, uint32_t shift_mont, unsigned log_N,
                               unsigned log_blowup, uint32_t scale_mont);
// T[beta][k] = s_beta^k / n (Montgomery), s_beta = shift * w_N^bitrev_b(beta), k < n: the factor the
// LDE applies to coefficient k before the forward transform of coset beta.  Cached per context.
const uint32_t* coset_scale_table(Context& ctx, unsigned log_n, unsigned log_blowup, uint32_t shift);
// src: row-major n x w (natural rows)  ->  dst: column-major, rows in bit-reversed order
// src_width (0 = w): the row length of `src` when only w of its columns (starting at `src`) are taken
void launch_transpose_bitrev(Context& ctx, const uint32_t* src, uint32_t* dst, unsigned log_n,
                             uint32_t w, uint64_t dst_col_stride, uint32_t src_width = 0);
// The same with the first round (four stages) of the inverse transform folded in, for a matrix that
// goes to coset_lde next (ntt_lde.hip).  Returns false -- and does nothing -- where coset_lde has no
// contiguous inverse pass to shorten (small n): call launch_transpose_bitrev then; if true, pass
// first_round_done = true to coset_lde.
bool launch_transpose_bitrev_r16(Context& ctx, const uint32_t* src, uint32_t* dst, unsigned log_n, uint32_t w,
                                 uint64_t dst_col_stride, uint32_t src_width = 0);
// same without the bit reversal (rows stay where they are): BFMmcs::commit on given matrices
void launch_transpose_plain(Context& ctx, const uint32_t* src, uint32_t* dst, uint64_t n, uint32_t w,
                            uint64_t dst_col_stride);
// dst row-major (h x w)  <-  src column-major; used by debug downloads and row gathers
void launch_transpose_to_row_major(Context& ctx, const uint32_t* src, uint64_t col_stride,
                                   uint32_t* dst, uint64_t h, uint32_t w);
// Coset low-degree extension of `ncols` columns (reference fri/src/two_adic_pcs.rs:233-241):
//   in : evals[c][p] = column value at subgroup index bitrev(p)   (n per column, DESTROYED)
//   out: out[c][beta*n + t] = p_c(shift * w_N^bitrev(beta*n+t)),  N = n << log_blowup
// With a coset range (beta0, n_beta > 0) only the row blocks beta0 .. beta0+n_beta-1 are produced,
// at out[c][(beta - beta0)*n + t]: the slab of a rank that owns those cosets (sharded prover).
// Two matrices of one height in ONE set of launches (the two quotient chunks: each is four columns, and a
// launch set of its own left the chip a quarter full three times over): evals2 != nullptr holds columns
// gw .. ncols-1 (same column stride), extended on its own coset shift2; `out` takes all ncols columns.
void coset_lde(Context& ctx, uint32_t* evals, uint64_t in_col_stride, uint32_t ncols, unsigned log_n,
               unsigned log_blowup, uint32_t shift, uint32_t* out, uint64_t out_col_stride,
               uint32_t beta0 = 0, uint32_t n_beta = 0, bool first_round_done = false,
               uint32_t* evals2 = nullptr, uint32_t shift2 = 0, uint32_t gw = 0);

// ---- merkle.hip ------------------------------------------------------------------------------
constexpr int MAX_BATCH_MATS = 64;
struct LeafMats {
    const uint32_t* d[MAX_BATCH_MATS];
    uint64_t col_stride[MAX_BATCH_MATS];
    uint32_t width[MAX_BATCH_MATS];
    // row of matrix i opened for leaf index r: r >> row_shift[i] (log_max_height - log_height_i;
    // basic/src/mmcs/bf_mmcs.rs:37-42)
    uint8_t row_shift[MAX_BATCH_MATS];
    uint32_t n_mats;
    uint32_t total_width;
    // device array of total_width column base pointers (column c of the concatenated row), so
    // that the leaf kernel addresses a row element with one uniform pointer load
    const uint32_t* const* cols;
};
// leaf digests (8 words each) of `height` rows: Blake3(row of mat 0 || row of mat 1 || ...)
void launch_leaf_hash(Context& ctx, const LeafMats& mats, uint64_t height, uint32_t* digests);
// leaf digests of an array-of-EF4 vector taken as rows of two elements (FRI commit-phase matrix)
void launch_leaf_hash_ef_pairs(Context& ctx, const uint32_t* vec, uint64_t n_rows, uint32_t* digests);
// builds every upper level of the tree; `tree` holds level l at offset level_off(l) (in digests)
// If `ch` is given, the kernel that produces the root also observes it on the device challenger
// and writes the root and the sampled challenge (returns false if no kernel could do it, i.e. the
// tree is a single leaf).
struct DevChallenger;
bool launch_merkle_levels(Context& ctx, uint32_t* tree, unsigned log_leaves,
                          DevChallenger* ch = nullptr, uint32_t* root_out = nullptr,
                          Ef* beta_out = nullptr);
// Leaf digests AND every level in one launch where the shape allows (leaf_tree.hpp; 2^8 leaves and
// up, rows of at most 256 elements; TS_LEAF_TREE=0 or another shape: launch_leaf_hash +
// launch_merkle_levels).  Return value as launch_merkle_levels.
bool leaf_tree_enabled(unsigned log_leaves);
bool launch_commit_tree(Context& ctx, const LeafMats& mats, unsigned log_leaves, uint32_t* tree,
                        DevChallenger* ch = nullptr, uint32_t* root_out = nullptr, Ef* beta_out = nullptr);
bool launch_commit_tree_ef_pairs(Context& ctx, const uint32_t* vec, unsigned log_leaves, uint32_t* tree,
                                 DevChallenger* ch = nullptr, uint32_t* root_out = nullptr,
                                 Ef* beta_out = nullptr);
// mixed-height batches: one level at a time, with the digests of the rows of the matrices whose
// height equals the level's node count compressed into the nodes (node = Blake3(node || inj))
// sharded trees: the G sub-tree roots (gathered, rank order) -> the log2(G) top levels.  `top`
// receives 2G-1 digests (the G roots first, the tree root last).  With `ch` the root is observed
// and the next challenge sampled, as in launch_merkle_levels.
void launch_shard_top(Context& ctx, const uint32_t* subroots, uint32_t G, uint32_t* top,
                      DevChallenger* ch, uint32_t* root_out, Ef* beta_out);
void launch_merkle_one_level(Context& ctx, const uint32_t* children, uint32_t* parents,
                             uint64_t n_parents);
void launch_merkle_inject(Context& ctx, uint32_t* nodes, const uint32_t* inj, uint64_t n);
inline uint64_t merkle_level_offset(unsigned log_leaves, unsigned level) {
    // levels are stored back to back: leaves first
    uint64_t off = 0;
    for (unsigned l = 0; l < level; l++) off += (uint64_t)1 << (log_leaves - l);
    return off;
}
inline uint64_t merkle_total_digests(unsigned log_leaves) { return ((uint64_t)2 << log_leaves) - 1; }

// ---- quotient.hip ----------------------------------------------------------------------------
struct AirProgram;  // air.hpp
void launch_selectors(Context& ctx, unsigned log_n, unsigned log_qd, uint32_t* is_first,
                      uint32_t* is_last, uint32_t* is_transition, uint32_t shift = GENERATOR);
// quotient chunks, each written column-major (4 columns x n) with bit-reversed rows:
// coefficient k of chunk c at chunk[c][k * n + pos]
constexpr int MAX_QUOTIENT_CHUNKS = 64;  // quotient_degree <= 64 (log_quotient_degree <= log_blowup <= 8)
struct QuotOut {
    uint32_t* chunk[MAX_QUOTIENT_CHUNKS];
};
// Rows [row_begin, row_end) of the quotient domain only (row_end = 0: all); trace_lde.d must then
// be such that d[c * col_stride + r] is valid for those rows r and their `next` rows (a sharded
// prover passes its slab pointer minus the slab's first row; whole cosets keep `next` local).
void launch_quotient(Context& ctx, const AirProgram& air, const ColMat& trace_lde, unsigned log_n,
                     unsigned log_qd, const uint32_t* d_consts_mont, const uint32_t* d_alpha_pows_mont,
                     const uint32_t* is_first, const uint32_t* is_last, const uint32_t* is_transition,
                     const QuotOut& out, uint64_t row_begin = 0, uint64_t row_end = 0,
                     uint32_t shift = GENERATOR);
// sharded.cpp "local quotient": in place on the slab LDEs of the qd chunk matrices (4 columns each),
// out[c] = sum_c' mix[c * qd + c'] * in[c'] per row and per column (mix: Montgomery form, device)
void launch_chunk_mix(Context& ctx, uint32_t* const* d_chunk_ptrs, uint32_t qd, uint64_t rows, uint64_t col_stride,
                      const uint32_t* d_mix_mont);

// check_constraints.rs:11-39 on the row-major trace; *d_violation (preset to ~0) receives
// row * 2^16 + constraint index of the first violated constraint
void launch_check_constraints(Context& ctx, const AirProgram& air, const uint32_t* trace_row_major,
                              uint64_t n, const uint32_t* d_consts_mont,
                              unsigned long long* d_violation);

// ---- open.hip --------------------------------------------------------------------------------
// d[p][i] = x_i / (z_p - x_i) (Montgomery EF4) for the low coset 31*H_n in bit-reversed order,
// for up to 2 points; out layout [point][n] of Ef
// coset_gen (canonical, 0 = 31): the weights of the coset coset_gen * H_n instead (any coset of
// the LDE determines the polynomial, so a sharded prover interpolates on one it owns)
void launch_bary_weights(Context& ctx, unsigned log_n, const Ef* points_mont, uint32_t n_points,
                         Ef* out, uint32_t coset_gen = 0);
// out[col][p] = sum_i m[col][i] * d[p][i]   (canonical EF4), i over the first n rows
// `pending` != nullptr: the finishing pass (partial sums -> out) is left to launch_bary_finish, which takes
// up to two pending products in one launch
struct BaryPending {
    DevBuf<uint32_t> partial[2];
    uint32_t* out[2] = {nullptr, nullptr};
    uint32_t n_blocks[2] = {0, 0}, n_words[2] = {0, 0};
    uint32_t n = 0;
};
void launch_bary_dots(Context& ctx, const ColMat& m, unsigned log_n, const Ef* weights,
                      uint32_t n_points, Ef* out, BaryPending* pending = nullptr);
void launch_bary_finish(Context& ctx, BaryPending& pending);
// ro[X] (+)= sum_p off_p * (S(X) - rys_p) / (x_X - z_p),  S(X) = sum_i alpha^i m[i][X]
struct ReduceArgs {
    Ef z_mont[2];
    Ef off_mont[2];  // alpha^num_reduced (Montgomery)
    Ef rys[2];       // reduced opened values (canonical)
    uint32_t n_points;
    uint32_t accumulate;  // 0: ro = ..., 1: ro += ...
};
void launch_reduce(Context& ctx, const ColMat& m, unsigned log_h, const uint32_t* d_alpha_pows_mont,
                   const ReduceArgs& args, Ef* ro);
// the prove() shape in one pass: trace opened at (zeta, zeta*omega), n_chunks width-4 matrices at zeta
struct FusedReduceArgs {
    Ef z_mont[2];
    Ef off_t[2];   // alpha^0, alpha^w            (Montgomery)
    // The chunk terms sum_c off_c (S_c - rys_c), off_c = alpha^(2w + 4c), are folded on the host:
    // column k of chunk c is weighted with alpha^k off_c (chunk_w, device memory, Montgomery, 4 EF4
    // per chunk) so that ONE dot product over all chunk columns gives sum_c off_c S_c, and every
    // constant goes into k0 = off_t0 rys_t0 + sum_c off_c rys_c, k1 = off_t1 rys_t1 (canonical).
    Ef k0, k1;
    const uint32_t* chunk_w;
    const uint32_t* chunk[MAX_QUOTIENT_CHUNKS];
    uint64_t chunk_stride;
    uint32_t n_chunks;
    // slab of a sharded prover: global rows [row0, row0 + rows) live at local rows [0, rows) of
    // every matrix and of `ro` (rows = 0: the whole domain)
    uint64_t row0, rows;
};
void launch_reduce_fused(Context& ctx, const ColMat& trace, unsigned log_h,
                         const uint32_t* d_alpha_pows_mont, const FusedReduceArgs& args, Ef* ro);

// ---- fri.hip ---------------------------------------------------------------------------------
// out[i] = fold(in[2i], in[2i+1]; beta) (reference two_adic_pcs.rs:116-147); h = output length.
// If next_digests != nullptr (h >= 2) also writes the h/2 leaf digests of the next round.
void launch_fri_fold(Context& ctx, const Ef* in, uint64_t h, Ef beta_canonical, Ef* out,
                     uint32_t* next_digests);
// same with beta read from device memory (written by launch_chal_round)
// slab form: `in`/`out` hold global outputs [row0, row0 + h) of a fold whose output has h_global
// rows (h_global = 0: the whole vector, h_global = h, row0 = 0)
void launch_fri_fold_dev(Context& ctx, const Ef* in, uint64_t h, const Ef* d_beta, Ef* out,
                         uint32_t* next_digests, uint64_t h_global = 0, uint64_t row0 = 0);
// One commit-phase round in one launch: leaves (cur[2i], cur[2i+1]) hashed, the whole tree built,
// the root observed and the next challenge sampled.  prev != nullptr: cur (2h elements) is first
// computed as the fold of prev (4h elements) with the challenge at d_beta_prev, and stored.
constexpr unsigned FRI_ROUND_MAX_LOG = 22;  // = mt::MAX_LOG_TREE (merkle_tree.hpp)
void launch_fri_round(Context& ctx, const Ef* prev, const Ef* d_beta_prev, Ef* cur, uint64_t h,
                      uint32_t* tree, DevChallenger* ch, uint32_t* root_out, Ef* beta_out);
// The same for a round of any height, through the leaf-tree kernel (leaf_tree.hpp: one launch from 2^8
// to 2^22 leaves; TS_LEAF_TREE=0 or a smaller round: fold launch, level launches, tree launch).  Slab
// form as launch_fri_fold_dev, in LEAVES: `cur`/`tree` hold leaves [row0, row0 + h) of a round of
// h_global leaves (the sub-tree of a sharded prover's slab).  Returns whether the challenger step ran
// (as launch_merkle_levels).
bool launch_fri_round_tall(Context& ctx, const Ef* prev, const Ef* d_beta_prev, Ef* cur, uint64_t h,
                           uint32_t* tree, DevChallenger* ch, uint32_t* root_out, Ef* beta_out,
                           uint64_t h_global = 0, uint64_t row0 = 0);
unsigned merkle_tree_max_log();  // trees up to this many levels are one launch (TS_TREE_MAX_LOG)
unsigned fri_round_max_log();    // commit rounds up to this many levels are one launch (TS_FRI_ROUND_LOG)
// device-resident transcript (chal_dev.hpp): observe the root at `root`, sample beta
struct DevChallenger;
void launch_chal_round(Context& ctx, DevChallenger* ch, const uint32_t* root, uint32_t* root_out,
                       Ef* beta_out);
// all remaining commit-phase rounds once the vector has <= 2^FRI_TAIL_LOG elements, one workgroup
constexpr int FRI_TAIL_LOG = 10;
// pow_out != nullptr: the kernel also grinds (fri/src/prover.rs:43) -- *pow_out = the smallest witness
// below 4096 that passes pow_bits on the transcript as the last round leaves it, FRI_POW_NONE if it
// cannot tell (pending transcript input, the test permutation) or none passes
constexpr uint32_t FRI_POW_NONE = 0xffffffffu;
void launch_fri_tail(Context& ctx, const Ef* in, uint32_t L0, uint32_t blowup, DevChallenger* ch,
                     Ef* tail_vecs, uint32_t* tail_trees, uint32_t* roots_out, Ef* betas_out,
                     Ef* final_out, uint32_t pow_bits = 0, uint32_t* pow_out = nullptr,
                     const Ef* beta_in = nullptr);
// (beta_in != nullptr: `in` holds 2 L0 elements, the previous round's vector, and the kernel starts by
// folding it with *beta_in)
void launch_vec_add(Context& ctx, Ef* acc, const Ef* other, uint64_t n);
// gathers: rows of column-major matrices and Merkle paths at given indices
void launch_gather_rows(Context& ctx, const LeafMats& mats, const uint32_t* d_indices,
                        uint32_t n_idx, unsigned index_shift, uint32_t* out);
void launch_gather_paths(Context& ctx, const uint32_t* tree, unsigned log_leaves,
                         const uint32_t* d_indices, uint32_t n_idx, unsigned index_shift,
                         uint32_t* out);
void launch_gather_ef_pairs(Context& ctx, const Ef* vec, const uint32_t* d_indices, uint32_t n_idx,
                            unsigned index_shift, uint32_t* out);
// all commit-phase openings in one launch: one descriptor per FRI round (device array)
struct FriGatherDesc {
    const uint32_t* vec;   // committed vector as words (8 per row)
    const uint32_t* tree;  // its Merkle tree
    uint32_t log_leaves;
    uint32_t shift;        // row = index >> shift
    uint64_t out_vals;     // word offset of [query][8] in `out`
    uint64_t out_path;     // word offset of [query][log_leaves][8] in `out`
};
void launch_gather_fri(Context& ctx, const FriGatherDesc* d_descs, uint32_t n_rounds,
                       uint32_t max_log_leaves, const uint32_t* d_indices, uint32_t n_idx,
                       uint32_t* out);
// The whole query phase of one proof in ONE launch (after the host's sync the stream is empty, and
// every launch on an empty stream costs the ~4 us it takes to reach the GPU: five launches were 16 us
// of a 3.4 ms proof).  Row jobs: the opened rows of a committed batch (k_gather_rows' work); descriptor
// jobs as above, where vec == nullptr means "path only" (the batch's Merkle path) and log_leaves == 0
// "values only" (a pass-through input).  Both tables are device arrays.
struct RowGatherJob {
    LeafMats mats;
    uint32_t shift;  // row = index >> shift (>> row_shift[i] per matrix)
    uint32_t pad;
    uint64_t out;    // word offset of [query][total_width] in `out`
};
void launch_gather_queries(Context& ctx, const RowGatherJob* d_rows, uint32_t n_rows, uint32_t max_row_width,
                           const FriGatherDesc* d_descs, uint32_t n_descs, uint32_t max_log_leaves,
                           const uint32_t* d_indices, uint32_t n_idx, uint32_t* out);

// ---- alubench.hip ---------------------------------------------------------------------------
// whole-chip rate of NTT butterflies (kind 0) or Blake3 compressions (kind 1), no memory traffic
double alu_ceiling(Context& ctx, int kind);

// ---- tracegen.hip ----------------------------------------------------------------------------
// row-major traces generated in place (no H2D): Fibonacci (uni-stark/tests/fib_air.rs:59-78) and the
// build-defined SynthMulAir-w trace (airs.py generate_synth_mul_trace)
void launch_trace_fibonacci(Context& ctx, uint32_t* out, uint32_t a, uint32_t b, uint64_t n);
void launch_trace_synth_mul(Context& ctx, uint32_t* out, uint64_t n, uint32_t width, uint64_t seed);
// build-defined SynthExt-w trace (airs.py generate_synth_ext_trace; config 5's stand-in)
void launch_trace_synth_ext(Context& ctx, uint32_t* out, uint64_t n, uint32_t width, uint64_t seed);

}  // namespace ts
