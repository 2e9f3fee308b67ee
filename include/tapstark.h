/*
 * tapstark.h -- C ABI of the MI355X-native tap-stark prover hot path (libtapstark_hip.so).
 *
 * Drop-in boundary for the reference's uni-stark/fri prover.  Every entry
 * point names the reference interface it replaces (paths relative to the tap-stark repository).
 * Conventions:
 *   - field elements cross the boundary as CANONICAL u32 (reference basic/src/field/mod.rs:48-63
 *     `as_u32_vec`); an EF4 element is its 4 coefficients [c0,c1,c2,c3];
 *   - digests/commitments are 8 u32 words = the reference's [[u8;4];8] (little-endian words);
 *   - host buffers are caller-owned and only read during the call; device objects are
 *     library-owned handles, freed explicitly;
 *   - every function returns a ts_status (0 = ok) and never throws or aborts; where the reference
 *     panics (assert!/expect) the status is TS_ERR_INVARIANT and ts_last_error() holds the text;
 *   - a context is driven by one host thread; different contexts are independent;
 *   - there is NO CPU fallback: without a HIP device ts_ctx_create fails.
 */
#ifndef TAPSTARK_H
#define TAPSTARK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int ts_status;
enum {
    TS_OK = 0,
    TS_ERR_INVALID = 1,
    TS_ERR_HIP = 2,
    TS_ERR_OOM = 3,
    TS_ERR_UNSUPPORTED = 4,
    TS_ERR_INVARIANT = 5,
    TS_ERR_BUFFER = 6,
    TS_ERR_COMM = 7 /* a ts_comm callback reported failure */
};

typedef struct ts_ctx ts_ctx;               /* one per GPU */
typedef struct ts_matrix ts_matrix;         /* device-resident RowMajorMatrix<Val> */
typedef struct ts_pcs_data ts_pcs_data;     /* Pcs::ProverData / BFMmcs::ProverData */
typedef struct ts_air ts_air;               /* compiled constraint tape */
typedef struct ts_challenger ts_challenger; /* BfChallenger (host side) */

/* reference fri/src/config.rs:11-16 FriConfig (the `mmcs` field is the built-in Blake3 Merkle
 * MMCS) */
typedef struct {
    uint32_t log_blowup;
    uint32_t num_queries;
    uint32_t proof_of_work_bits;
} ts_fri_config;

/* Number of HIP devices this process sees (0 without a GPU or a driver): lets a multi-rank caller
 * check "one device per rank" before it starts any rank (examples/prove_sharded.cpp). */
int ts_device_count(void);
/* ------------------------------------------------------------------ context */
ts_status ts_ctx_create(int device, ts_ctx** out);
void ts_ctx_destroy(ts_ctx* ctx);
const char* ts_last_error(const ts_ctx* ctx);
ts_status ts_ctx_synchronize(ts_ctx* ctx);
/* HIP stream every kernel of this context is launched on (as a void*), for event timing */
void* ts_ctx_stream(ts_ctx* ctx);
/* per-stage timers (HIP events on the context's stream); names follow the reference's tracing
 * spans (uni-stark/src/prover.rs:53,82,121; fri/src/two_adic_pcs.rs:355-374; fri/src/prover.rs:18,45,92) */
ts_status ts_ctx_set_timing(ts_ctx* ctx, int enabled);
/* writes "name=ms;name=ms;..." of the stages recorded since the last call */
ts_status ts_ctx_take_timings(ts_ctx* ctx, char* buf, size_t cap);

/* MEASUREMENT AID: the ceiling of a "zero-host-sync" proof.  mode 1: the next ts_prove records the bytes the
 * host reads at each of its mid-proof synchronisations (two commitment roots, the opened-value sums, the FRI
 * block); mode 2: a ts_prove of the SAME trace, AIR and configuration skips those synchronisations and is
 * handed the recorded bytes at once -- as if the challenges cost nothing -- and must return the same proof;
 * mode 0: off.  Call it before EVERY proof it should apply to (it rewinds the log).  Not for production: with a
 * different trace a mode-2 proof is garbage. */
ts_status ts_ctx_set_replay(ts_ctx* ctx, int mode);
/* per-kernel timers: HIP events recorded on the context's stream around EVERY kernel launch
 * (resolved lazily, no sync per kernel); take writes "kernel=launches:total_ms;..." */
ts_status ts_ctx_set_kernel_timing(ts_ctx* ctx, int enabled);
ts_status ts_ctx_take_kernel_timings(ts_ctx* ctx, char* buf, size_t cap);
/* TS_FRI_GRAPH knob (the FRI commit phase, fri/src/prover.rs:93-141, replayed as a hipGraph):
 * out[0] = commit phases replayed from a graph, out[1] = captures abandoned for the eager path
 * (an allocation the pool could not serve inside the capture), out[2] = shapes whose block sizes
 * are known, out[3] = bytes the context's device pool holds.  Diagnostics only. */
ts_status ts_ctx_graph_stats(ts_ctx* ctx, uint64_t out[4]);
/* One diagnostic counter of the context by index: 0-3 as ts_ctx_graph_stats; 4 = graph attempts the
 * pool could not reserve for (the phase ran eagerly); 5 = ts_prove_sharded calls whose local quotient
 * (ts_shard_options.local_quotient) was redone through the broadcast path because FRI's final
 * polynomial was not constant, i.e. the trace was invalid (fri/src/prover.rs:129-134).
 * 6 / 7 / 8 = proof-of-work witnesses (fri/src/prover.rs:43) taken from the device search after the
 * host's one-step check_witness / device candidates the host refused (never expected) / searches the
 * host ran itself (no device candidate below 2^12, or TS_HOST_GRIND=1).
 * TS_ERR_INVALID for an unknown index. */
ts_status ts_ctx_stat(ts_ctx* ctx, int which, uint64_t* out);

/* ------------------------------------------------------------------ matrices */
/* RowMajorMatrix<Val>::new(values, width): host row-major canonical values -> device */
ts_status ts_matrix_upload(ts_ctx* ctx, const uint32_t* host_row_major, uint64_t height,
                           uint32_t width, ts_matrix** out);
/* same from a device pointer (e.g. a torch tensor's data_ptr); the data is copied */
/* The same without waiting for the copy: `host_pinned` must come from ts_host_alloc (page-locked:
 * the copy then runs at the full PCIe rate, asynchronously on the context's stream) and must stay
 * untouched until the next ts_ctx_synchronize / blocking call on this context. */
ts_status ts_matrix_upload_async(ts_ctx* ctx, const uint32_t* host_pinned, uint64_t height,
                                 uint32_t width, ts_matrix** out);
ts_status ts_host_alloc(size_t bytes, void** out);
void ts_host_free(void* p);
ts_status ts_matrix_from_device(ts_ctx* ctx, const uint32_t* dev_row_major, uint64_t height,
                                uint32_t width, ts_matrix** out);
/* Traces generated on the device (no H2D).  ts_trace_fibonacci = generate_trace_rows(a, b, n) of
 * uni-stark/tests/fib_air.rs:59-78 (n x 2: row 0 = (a, b), then (l, r) -> (r, l + r));
 * ts_trace_synth_mul = the build-defined SynthMulAir-`width` trace of BASELINE configs 3/4
 * (SplitMix64 stream `seed`, tap-stark_amd/airs.py generate_synth_mul_trace);
 * ts_trace_synth_ext = the SynthExt-`width` trace of config 5's stand-in (generate_synth_ext_trace). */
ts_status ts_trace_fibonacci(ts_ctx* ctx, uint32_t a, uint32_t b, uint64_t n, ts_matrix** out);
ts_status ts_trace_synth_mul(ts_ctx* ctx, uint64_t n, uint32_t width, uint64_t seed, ts_matrix** out);
ts_status ts_trace_synth_ext(ts_ctx* ctx, uint64_t n, uint32_t width, uint64_t seed, ts_matrix** out);
ts_status ts_matrix_dims(const ts_matrix* m, uint64_t* height, uint32_t* width);
/* row-major, natural row order */
ts_status ts_matrix_download(ts_ctx* ctx, const ts_matrix* m, uint32_t* host_row_major);
void ts_matrix_free(ts_ctx* ctx, ts_matrix* m);

/* ------------------------------------------------------------------ AIR */
/* Tape = serialised result of get_symbolic_constraints (uni-stark/src/symbolic_builder.rs:52-64):
 *   [0]=0x54415354 [1]=1 [2]=width [3]=n_public [4]=n_nodes [5]=n_constraints,
 *   n_nodes x {op,a,b}, n_constraints node ids in assert_zero order.
 * ops (symbolic_expression.rs:12-37): 0 CONST(a=value) 1 MAIN(a=offset 0|1,b=column)
 *   2 PUBLIC(a=index) 3 IS_FIRST_ROW 4 IS_LAST_ROW 5 IS_TRANSITION 6 ADD(a,b) 7 SUB(a,b) 8 NEG(a)
 *   9 MUL(a,b) */
/* ctx == NULL builds a host-only AIR (degree rules + verifier use; no kernels) */
ts_status ts_air_compile(ts_ctx* ctx, const uint32_t* tape, size_t n_words, ts_air** out);
/* get_log_quotient_degree, uni-stark/src/symbolic_builder.rs:15-32 */
ts_status ts_air_info(const ts_air* air, uint32_t* width, uint32_t* n_public,
                      uint32_t* max_constraint_degree, uint32_t* log_quotient_degree);
/* 1 if the quotient kernel was specialised for this AIR with hiprtc, 0 if the generic on-device
 * interpreter is used (hiprtc missing, TS_NO_JIT set in the environment, the program is above the
 * compile budget, or a background compilation has not finished yet).
 * Compile budget (hiprtc's time grows faster than the program): up to TS_JIT_SYNC_INSTR (default 2048)
 * lowered instructions the kernel is compiled inside ts_air_compile; up to TS_JIT_MAX_INSTR (default
 * 32768) by a child process (the helper `ts_jitc` beside the library; TS_JITC_PATH overrides; without it
 * such programs stay on the interpreter) while proofs already run on the interpreter -- the first use after
 * it finishes switches over, the proof words are the same either way; larger programs stay on the
 * interpreter, which has no limit on program size or live values.  TS_JIT_CACHE_DIR (an existing
 * directory) keeps the code objects across processes: a later ts_air_compile of the same AIR loads
 * instead of compiling. */
int ts_air_is_jit(const ts_air* air);
/* joins a background compilation; state: 0 none, 3 specialised kernel loaded, 4 compilation failed */
ts_status ts_air_jit_wait(ts_ctx* ctx, ts_air* air, int* state, double* compile_seconds);
void ts_air_free(ts_ctx* ctx, ts_air* air);
/* Inspection of what the tape was lowered to (none of these needs a GPU; `air` may be host-only).
 * The reference's counterpart is the monomorphised `Air::eval` inside quotient_values
 * (uni-stark/src/prover.rs:170-181): user code compiled into the prover.  Here the tape is lowered
 * to a register program (csrc/air.cpp) that the on-device interpreter runs, and that program to
 * straight-line HIP source compiled with hiprtc (csrc/jit.cpp); tests interpret the former and
 * compile the latter against the oracle's direct evaluation of the tape.
 * ts_air_program: out = [n_regs, n_instr, n_consts, n_instr x {op,dst,a,b}, n_consts x canonical
 *   value, n_consts x (public-value index or 0xffffffff)]; ops: 0 LOAD(a=row offset,b=column)
 *   1 CONST(a=const index) 2 SEL(a=0 first|1 last|2 transition) 3 ADD 4 SUB 5 NEG 6 MUL
 *   7 ASSERT(a=register, b=constraint index).  TS_ERR_BUFFER (with *n_words set) if cap is short.
 * ts_air_jit_source: the HIP source (not NUL-terminated; *n_bytes set even on TS_ERR_BUFFER).
 * ts_air_jit_compile: that source through hiprtc for `arch` ("gfx950"); the code object and the
 *   compile time.  TS_ERR_UNSUPPORTED if hiprtc is missing or the compilation fails. */
ts_status ts_air_program(const ts_air* air, uint32_t* out, size_t cap_words, size_t* n_words);
ts_status ts_air_jit_source(const ts_air* air, char* buf, size_t cap, size_t* n_bytes);
ts_status ts_air_jit_compile(const ts_air* air, const char* arch, void* code_out, size_t cap,
                             size_t* n_bytes, double* seconds);

/* ------------------------------------------------------------------ PCS */
/* Pcs::commit, fri/src/two_adic_pcs.rs:227-245: for each (domain, evals): coset LDE with shift
 * 31/domain.shift, bit-reversed rows, then mmcs.commit.  The matrices may have different
 * (power-of-two) heights: a shorter one's row digests are injected at the tree layer of its
 * height and open_batch reduces the index (bf_mmcs.rs:10-15, tcs/mod.rs:339-378).  The matrices are consumed (like the moved
 * `RowMajorMatrix` arguments).  root_out = commitment. */
ts_status ts_pcs_commit(ts_ctx* ctx, const ts_fri_config* cfg, uint32_t n_mats,
                        ts_matrix* const* evals, const uint32_t* domain_shifts,
                        uint32_t root_out[8], ts_pcs_data** out);
/* BFMmcs::commit, basic/src/mmcs/bf_mmcs.rs:22-35 (reference impl taptree_mmcs.rs:101-114): commit
 * to the given matrices as they are (no LDE, rows in the given order), any widths -- rows wider
 * than 256 elements hash as multi-chunk Blake3 -- and power-of-two heights.  The result is a
 * ts_pcs_data: ts_pcs_open_batch, ts_pcs_data_digests, ts_pcs_data_lde (= get_matrices) apply.
 * The matrices are consumed. */
ts_status ts_mmcs_commit(ts_ctx* ctx, uint32_t n_mats, ts_matrix* const* mats, uint32_t root_out[8],
                         ts_pcs_data** out);
/* BFMmcs::get_matrices (basic/src/mmcs/bf_mmcs.rs:52): committed LDE `idx`, row-major,
 * bit-reversed row order, N x width */
ts_status ts_pcs_data_lde(ts_ctx* ctx, const ts_pcs_data* d, uint32_t idx, uint32_t* host_row_major);
ts_status ts_pcs_data_info(const ts_pcs_data* d, uint32_t* n_mats, uint32_t* log_height);
/* height (LDE rows) and width of committed matrix `idx`; log_height above is the tallest one's */
ts_status ts_pcs_data_matrix_info(const ts_pcs_data* d, uint32_t idx, uint64_t* height, uint32_t* width);
/* Merkle digest layer `level` (0 = leaves), (N >> level) x 8 words */
ts_status ts_pcs_data_digests(ts_ctx* ctx, const ts_pcs_data* d, uint32_t level, uint32_t* host_out);
/* BFMmcs::open_batch (bf_mmcs.rs:37-42; taptree_mmcs.rs:46-63): rows of every matrix at `index`
 * (concatenated) and the sibling path (log_height x 8 words, leaf level first) */
ts_status ts_pcs_open_batch(ts_ctx* ctx, const ts_pcs_data* d, uint64_t index, uint32_t* rows_out,
                            uint32_t* path_out);
void ts_pcs_data_free(ts_ctx* ctx, ts_pcs_data* d);

/* get_evaluations_on_domain + quotient_values + flatten_to_base + split_evals
 * (two_adic_pcs.rs:247-258; uni-stark/src/prover.rs:68-80,122-194): quotient_degree matrices of
 * n x 4, ready for ts_pcs_commit with domain shifts 31 * w_{n*qd}^c.
 * chunks_out must have room for 2^log_quotient_degree handles. */
ts_status ts_quotient_chunks(ts_ctx* ctx, const ts_pcs_data* trace_data, uint32_t log_blowup,
                             const ts_air* air, const uint32_t* public_values, uint32_t n_public,
                             const uint32_t alpha[4], ts_matrix** chunks_out);

/* Pcs::open up to the FRI input (two_adic_pcs.rs:312-389) for the prove() shape: round 0 = trace
 * data opened at {zeta, zeta*w_n}, round 1 = quotient data (qd matrices) opened at {zeta}.
 * opened_out: (2*w + 4*qd) EF4 in proof order (trace_local, trace_next, chunks);
 * reduced_out: N EF4 (the single FRI input vector), or NULL. */
ts_status ts_pcs_open_reduce(ts_ctx* ctx, const ts_fri_config* cfg, const ts_pcs_data* trace_data,
                             const ts_pcs_data* quotient_data, const uint32_t zeta[4],
                             const uint32_t batch_alpha[4], uint32_t* opened_out,
                             uint32_t* reduced_out);

/* Pcs::open, fri/src/two_adic_pcs.rs:260-419, for any rounds x matrices x points (the shapes of
 * fri/tests/pcs.rs:62-90 and uni-stark/src/prover.rs:94-104 alike; matrices of different heights in
 * one batch and several batches are allowed).  Samples the batch challenge from `chal`, computes the
 * opened values and runs bf_prove (fri/src/prover.rs:19-63) on the reduced openings.
 *   rounds[r]   committed batches (Pcs::ProverData), in the order the verifier will list them
 *   n_points[k] number of opening points of matrix k, k running over (round, matrix)
 *   points      4 canonical words per point, in the same order
 *   opened_out  EF4 values in (round, matrix, point, column) order
 *   proof_out   the FriProof in TSPF v1 words (commit-phase round count first; DESIGN.md "Proof
 *               format"); per query the input proof holds one BatchOpening per round
 * TS_ERR_BUFFER if a buffer is too small (the needed sizes are still written). */
ts_status ts_pcs_open(ts_ctx* ctx, const ts_fri_config* cfg, ts_challenger* chal, uint32_t n_rounds,
                      const ts_pcs_data* const* rounds, const uint32_t* n_points,
                      const uint32_t* points, uint32_t* opened_out, size_t opened_cap_words,
                      size_t* n_opened_words, uint32_t* proof_out, size_t proof_cap_words,
                      size_t* n_proof_words);

/* bf_prove (fri/src/prover.rs:19-63) on its own, as fri/tests/fri.rs:51-147 drives it: `n_inputs`
 * host vectors of EF4 (4 words per element; BabyBear values are embedded as (v, 0, 0, 0)) of
 * strictly descending power-of-two lengths 2^log_lens[k]; the "input opening proof" of a query is
 * the literal reduced openings [(log_height, value)] (fri.rs:109-118).  Any challenger kind
 * (ts_chal_new: Blake3 or the test permutation, EF4 or BabyBear samples).  The proof is the FriProof
 * in TSPF v1 words with that input-proof shape.  ts_fri_verify = verify_shape_and_sample_challenges
 * + verify_challenges (fri/src/verifier.rs:20-98) for such a proof; verdict codes as ts_verify. */
ts_status ts_fri_prove(ts_ctx* ctx, const ts_fri_config* cfg, ts_challenger* chal, uint32_t n_inputs,
                       const uint32_t* log_lens, const uint32_t* const* inputs, uint32_t* proof_out,
                       size_t cap_words, size_t* n_words_out);
ts_status ts_fri_verify(const ts_fri_config* cfg, ts_challenger* chal, const uint32_t* proof,
                        size_t n_words, int* verdict);

/* FriGenericConfig::fold_matrix, two_adic_pcs.rs:116-147 (host in, host out; 2h EF4 -> h EF4) */
ts_status ts_fri_fold(ts_ctx* ctx, const uint32_t* in, uint64_t h, const uint32_t beta[4],
                      uint32_t* out);
/* The same on DEVICE vectors (16-byte aligned), enqueued on the context's stream: what a caller that
 * keeps its FRI vectors in HBM binds, and what `bench.py --workload fold` times against the reference's only
 * benchmark (fri/benches/fold_even_odd.rs:14-46). */
ts_status ts_fri_fold_device(ts_ctx* ctx, const uint32_t* in_dev, uint64_t h, const uint32_t beta[4],
                             uint32_t* out_dev);

/* ------------------------------------------------------------------ challenger */
/* BfChallenger::new, basic/src/challenger/mod.rs:122-137.  permutation: 0 = Blake3Permutation
 * (mod.rs:23-49), 1 = reverse-the-state test permutation (fri/tests/fri.rs:35-48).
 * sample_ext: 1 = F is EF4 (as in uni-stark/tests/fib_air.rs:108), 0 = BabyBear. */
ts_status ts_chal_new(int permutation, int sample_ext, ts_challenger** out);
ts_status ts_chal_clone(const ts_challenger* c, ts_challenger** out);
void ts_chal_free(ts_challenger* c);
void ts_chal_observe(ts_challenger* c, uint32_t word);                   /* mod.rs:183-194 */
void ts_chal_observe_commitment(ts_challenger* c, const uint32_t d[8]);  /* mod.rs:197-223 */
void ts_chal_sample(ts_challenger* c, uint32_t out[4]);                  /* mod.rs:261-313 */
uint64_t ts_chal_sample_bits(ts_challenger* c, uint32_t bits);           /* mod.rs:341-348 */
int ts_chal_check_witness(ts_challenger* c, uint32_t bits, uint32_t witness); /* mod.rs:108-114 */
/* mod.rs:95-105; TS_ERR_INVARIANT when no witness in [0, 4096) ("failed to find witness") */
ts_status ts_chal_grind(ts_challenger* c, uint32_t bits, uint32_t* witness);
/* state export for tests: state[16], n_in, in[8], n_out, out[8] (34 words) */
void ts_chal_state(const ts_challenger* c, uint32_t out[34]);

/* ------------------------------------------------------------------ prove */
/* uni_stark::prove(config, air, challenger, trace, public_values) -> Proof
 * (uni-stark/src/prover.rs:25-119).  `trace` is consumed.  The proof is written in the TSPF v1
 * wire format: header, commitments, opened_values, opening_proof.  Like a release
 * build of the reference, it does not run check_constraints (prover.rs:40-41 is debug-only). */
ts_status ts_prove(ts_ctx* ctx, const ts_fri_config* cfg, const ts_air* air, ts_challenger* chal,
                   ts_matrix* trace, const uint32_t* public_values, uint32_t n_public,
                   uint32_t* proof_out, size_t cap_words, size_t* n_words_out);

/* ---- one proof over the GPUs of a node (BASELINE config 4) ----
 * The library does no networking of its own: the host hands it the collectives (RCCL through
 * torch.distributed in tap-stark_amd/dist.py; any MPI-like layer can stand in).  Buffers are DEVICE
 * pointers of the calling rank's GPU.  A callback is ordered on `hip_stream` (the context's
 * stream): it either enqueues its work there or synchronises the stream first, and when it returns
 * later work on that stream sees the result.  Return 0 on success. */
typedef struct {
    int rank, world;
    void* user;
    /* recv_dev receives world * bytes_per_rank bytes, the contributions in rank order */
    int (*all_gather)(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_rank,
                      void* hip_stream);
    int (*broadcast)(void* user, void* buf_dev, size_t bytes, int root, void* hip_stream);
    /* optional (may be NULL): called on a rank whose ts_prove_sharded fails, so that its peers'
     * pending collectives fail too instead of waiting for ever */
    void (*abort)(void* user);
} ts_comm;

/* Native communicators (csrc/comm.cpp) for hosts without Python.
 * RCCL over xGMI, one process or thread per GPU: rank 0 calls ts_rccl_unique_id and passes the 128
 * bytes to its peers by any channel; every rank then calls ts_comm_rccl_create with its context
 * (the communicator binds to the context's device; collectives are enqueued on the context's
 * stream, no host synchronisation).  librccl is bound at run time; TS_ERR_UNSUPPORTED if absent. */
typedef struct ts_rccl_comm ts_rccl_comm;
int ts_rccl_available(void);
ts_status ts_rccl_unique_id(uint8_t out[128]);
ts_status ts_comm_rccl_create(ts_ctx* ctx, const uint8_t unique_id[128], int rank, int world,
                              ts_comm* out, ts_rccl_comm** handle);
void ts_comm_rccl_destroy(ts_rccl_comm* handle);
/* What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice /
 * ncclGetVersion; -1 where the symbol is missing or the communicator was aborted) beside what it was
 * created with.  ts_comm_rccl_create already refuses (TS_ERR_COMM) a communicator whose count or rank
 * differ from the arguments; this is for launchers that want to print them. */
typedef struct {
    int rank, world;                 /* as passed to ts_comm_rccl_create */
    int comm_count, comm_user_rank;  /* as RCCL reports them */
    int comm_device, rccl_version, aborted;
    int checked;  /* 1: ts_comm_rccl_create confirmed rank / world with ncclCommUserRank / ncclCommCount;
                   * 0: this librccl lacks them, the check could not run */
} ts_rccl_info;
ts_status ts_comm_rccl_info(const ts_rccl_comm* handle, ts_rccl_info* out);
/* In-process group: `world` ranks = `world` host threads of one process, each with its own
 * context (same device or one device each: copies go device to device, peer to peer over xGMI).
 * Every rank's thread takes its ts_comm with ts_comm_local_get. */
typedef struct ts_comm_group ts_comm_group;
ts_status ts_comm_local_group_create(int world, ts_comm_group** out);
ts_status ts_comm_local_get(ts_comm_group* group, int rank, ts_comm* out);
void ts_comm_local_group_destroy(ts_comm_group* group);
/* A group is ONE-SHOT with respect to failure: after ts_comm.abort on any rank, or after a rendezvous
 * timed out (a peer died silently; 600 s unless TS_COMM_TIMEOUT_S or ..._set_timeout says otherwise),
 * every collective on it returns an error.  ts_comm_local_group_reset makes it usable again; call it
 * only once EVERY rank's ts_prove_sharded has returned (no thread may still be inside a collective). */
ts_status ts_comm_local_group_reset(ts_comm_group* group);
ts_status ts_comm_local_group_set_timeout(ts_comm_group* group, int seconds);

/* Throughput mode in one call: n_proofs independent proofs (uni-stark/src/prover.rs:25-35, one trace each, a
 * fresh BfChallenger each) of one AIR on n_lanes contexts of ONE device, one host thread per lane inside
 * the call; proof i runs on lane lane_of[i] (< n_lanes) with traces[i] (consumed; it must have been created
 * on that lane's context) and airs[lane_of[i]] (compiled on that context).  No two proofs start within
 * gate_ms of each other (0 = no gate; a quarter of one proof's solo time keeps the lanes in complementary
 * phases).  last_proof_out receives the proof of the highest index; start_ms_out / wall_ms_out (n_proofs each,
 * may be NULL): when each ts_prove-equivalent started, relative to the call's start, and how long it took.
 * What a host with cheap threads does itself (examples/prove_stream.cpp); a host behind an interpreter
 * lock gets the loop without paying its lock per proof. */
ts_status ts_prove_stream(ts_ctx* const* ctxs, const ts_air* const* airs, uint32_t n_lanes,
                          const ts_fri_config* cfg, ts_matrix* const* traces, const uint32_t* lane_of,
                          uint32_t n_proofs, const uint32_t* public_values, uint32_t n_public, double gate_ms,
                          uint32_t* last_proof_out, size_t cap_words, size_t* n_words_out,
                          double* start_ms_out, double* wall_ms_out);

/* prove() with the work of ONE proof split over comm->world ranks, one GPU each: rank g owns the
 * bit-reversed LDE rows [g N/G, (g+1) N/G) -- whole cosets, so world must be a power of two
 * <= 2^log_blowup (TS_ERR_UNSUPPORTED otherwise) -- with their Merkle sub-trees, FRI slabs and
 * queries.  `trace_rows` holds natural rows [g n/G, (g+1) n/G) of the trace and is consumed.
 * Every rank must pass a challenger in the same state; every rank receives the whole proof, which
 * is bit-identical to ts_prove's on the whole trace.  FRI rounds stay sharded while a rank holds
 * >= 2^min_local_log values. */
typedef struct {
    uint32_t struct_size;      /* = sizeof(ts_shard_options): a caller built against another layout of this
                                  struct (ABI 4 had a field here that did nothing) is refused with
                                  TS_ERR_INVALID instead of having its fields read as something else */
    uint32_t min_local_log;    /* 0 = default (12) */
    uint32_t trace_replicated; /* 1: `trace_rows` is the WHOLE trace on every rank (e.g. made by
                                  ts_trace_* on each device); the trace all-gather is skipped */
    uint32_t local_quotient;   /* 1: every rank evaluates the quotient (uni-stark/src/prover.rs:65-80) on
                                  its OWN cosets and derives its slab of the chunk LDEs from that: no rank
                                  waits for the owner of the quotient domain, no chunk broadcast.  Needs
                                  2^log_blowup / world >= quotient degree (else ignored).  The same proof as
                                  ts_prove for EVERY trace: for one that violates its constraints (ts_prove,
                                  like a release build of the reference, still hands out a proof, which the
                                  verifier rejects) constraints / Z_H is no polynomial, the mixed chunks are
                                  not low-degree, FRI's final polynomial is not constant
                                  (fri/src/prover.rs:129-134) -- on every rank alike -- and all ranks redo the
                                  quotient through the broadcast path (ts_ctx_stat 5 counts it) */
} ts_shard_options;
ts_status ts_prove_sharded(ts_ctx* ctx, const ts_fri_config* cfg, const ts_comm* comm,
                           const ts_air* air, ts_challenger* chal, ts_matrix* trace_rows,
                           const uint32_t* public_values, uint32_t n_public,
                           const ts_shard_options* options /* NULL = defaults */,
                           uint32_t* proof_out, size_t cap_words, size_t* n_words_out);

/* check_constraints (uni-stark/src/check_constraints.rs:11-39; what a debug build of prove() runs
 * first, prover.rs:40-41) on the GPU.  `trace` is NOT consumed.  *first_violation = -1 if every
 * constraint holds on every row, else row * 65536 + constraint index of the first failure. */
ts_status ts_check_constraints(ts_ctx* ctx, const ts_air* air, const ts_matrix* trace,
                               const uint32_t* public_values, uint32_t n_public,
                               int64_t* first_violation);

/* ------------------------------------------------------------------ verify (host only) */
/* uni_stark::verify (uni-stark/src/verifier.rs:19-25; pcs.verify fri/src/two_adic_pcs.rs:421-534;
 * fri/src/verifier.rs:20-165).  Needs no GPU: `air` may come from ts_air_compile(NULL, ...).
 * *verdict: 0 accept, 1 InvalidProofShape, 2 InvalidOpeningArgument (FRI shape), 3 InvalidPowWitness,
 * 4 input-MMCS error, 5 commit-phase MMCS error, 6 FinalPolyMismatch, 7 OodEvaluationMismatch,
 * 8 folded-evaluation mismatch (the reference asserts), 9 malformed proof buffer. */
ts_status ts_verify(const ts_fri_config* cfg, const ts_air* air, ts_challenger* chal,
                    const uint32_t* proof, size_t n_words, const uint32_t* public_values,
                    uint32_t n_public, int* verdict);

/* Pcs::verify (fri/src/two_adic_pcs.rs:421-534 with fri/src/verifier.rs:20-165) for any rounds x
 * matrices x points: the counterpart of ts_pcs_open, host only.  Arguments follow ts_pcs_open:
 *   commitments   n_rounds x 8 words
 *   mats_per_round[r], then per matrix k in (round, matrix) order: log_degrees[k] (log2 of the
 *   committed matrix's height), widths[k], n_points[k]; points: 4 words each; opened_values: EF4 in
 *   (round, matrix, point, column) order; fri_proof: what ts_pcs_open wrote.
 * The challenger must be in the state the prover's had when ts_pcs_open was called.
 * *verdict as for ts_verify (0 accept, 1 shape, 2 FRI shape, 3 PoW, 4 input MMCS, 5 commit-phase
 * MMCS, 6 final poly, 8 folded evaluation, 9 malformed). */
ts_status ts_pcs_verify(const ts_fri_config* cfg, ts_challenger* chal, uint32_t n_rounds,
                        const uint32_t* commitments, const uint32_t* mats_per_round,
                        const uint32_t* log_degrees, const uint32_t* widths, const uint32_t* n_points,
                        const uint32_t* points, const uint32_t* opened_values,
                        const uint32_t* fri_proof, size_t n_words, int* verdict);

/* ------------------------------------------------------------------ proof wire format (host only) */
/* TSPF v1 words <-> the postcard encoding of the reference's serde proof types
 * (uni-stark/src/proof.rs:17-38, fri/src/proof.rs:8-33, fri/src/two_adic_pcs.rs:63-68; postcard is
 * the carrier the reference names: uni-stark/Cargo.toml:44, uni-stark/tests/mul_air.rs:133-137).
 * Field order and element encodings are the reference's; the commitment (one 32-byte root) and the
 * MMCS opening proof (sibling path) are this build's Merkle types.
 * TS_ERR_INVALID for a malformed input, TS_ERR_BUFFER if the output does not fit (size still set). */
ts_status ts_proof_to_postcard(const uint32_t* proof, size_t n_words, uint8_t* out, size_t cap_bytes,
                               size_t* n_bytes_out);
ts_status ts_proof_from_postcard(const uint8_t* bytes, size_t n_bytes, uint32_t* proof_out,
                                 size_t cap_words, size_t* n_words_out);
/* The postcard bytes do not record which MMCS produced them.  ts_proof_from_postcard infers the TSPF
 * version from the number of roots per commitment (one -> v1, several -> v2), which is wrong for the
 * one case of a proof over taptrees (v2) made with num_queries = 1.  tspf_version = 1 or 2 asks for
 * that framing explicitly (1 refuses several roots; 2 always writes the num_queries header word);
 * 0 infers as above. */
ts_status ts_proof_from_postcard_v(const uint8_t* bytes, size_t n_bytes, int tspf_version,
                                   uint32_t* proof_out, size_t cap_words, size_t* n_words_out);

/* ------------------------------------------------------------------ taptree-compatible MMCS
 * The reference's real BFMmcs (basic/src/mmcs/taptree_mmcs.rs:24-119) commits to Bitcoin taptrees:
 * tagged SHA-256 (BIP-341 TapLeaf / TapBranch with lexicographically sorted children), a complete
 * binary tree over one script leaf per row (basic/src/tcs/builder.rs:38-93), one tree per query.
 * Digests cross the ABI as 32 bytes (the byte string Bitcoin hashes), not as words.
 *
 * The lock scripts inside a leaf script come from un-vendored crates (bitcomm / primitives) and are
 * therefore SUPPLIED BY THE CALLER as bytes; ts_tap_winternitz_lock_script is a stand-in built from
 * the reference's local copy of the construction (scripts/src/bit_comm/winternitz.rs:171-274,
 * bit_comm_u32.rs:80-85, u32/u32_std.rs:122-173).  Script execution against a witness
 * (tcs/mod.rs:143-147) is out of scope. */
typedef struct ts_taptree ts_taptree;
typedef struct ts_tap_mmcs_data ts_tap_mmcs_data;
/* TapLeaf hash of a script (NodeInfo::new_leaf_with_ver(script, TapScript), builder.rs:24-29), host */
ts_status ts_tapleaf_hash(const uint8_t* script, size_t len, uint8_t out[32]);
/* TapNodeHash::from_node_hashes(a, b) (complete_taptree.rs:57-60), host */
ts_status ts_tapbranch_hash(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
/* lock script of a bit commitment over u32_count limbs (1 = BabyBear, 4 = EF4), host */
ts_status ts_tap_winternitz_lock_script(const uint8_t* secret, size_t secret_len, uint32_t u32_count,
                                        uint8_t* out, size_t cap, size_t* len_out);
/* CommitedLeaf::generate_script (tcs/mod.rs:197-225), host: lock_offsets has n_evals + 2 entries
 * (index lock, then one lock per evaluation); values = n_evals * u32_size canonical limbs */
ts_status ts_tap_leaf_script(const uint8_t* lock_scripts, const uint64_t* lock_offsets, uint32_t n_evals,
                             uint32_t u32_size, uint64_t index, const uint32_t* values, uint8_t* out,
                             size_t cap, size_t* len_out);
/* CompleteTaptree::new_with_scripts (complete_taptree.rs:67-75): script i = bytes
 * [offsets[i], offsets[i+1]); n_leaves must be a power of two (builder.rs:40).  Hashed on the device. */
ts_status ts_taptree_from_scripts(ts_ctx* ctx, const uint8_t* scripts, const uint64_t* offsets,
                                  uint64_t n_leaves, ts_taptree** out);
/* CompleteTaptree::combine (complete_taptree.rs:90-133): leaves of `a` keep their indices, those of
 * `b` follow; the inputs stay valid */
ts_status ts_taptree_combine(const ts_taptree* a, const ts_taptree* b, ts_taptree** out);
ts_status ts_taptree_info(const ts_taptree* t, uint64_t* leaf_count, uint8_t root[32]);
/* get_leaf_proof (complete_taptree.rs:155-159): leaf hash + sibling path, leaf-most first */
ts_status ts_taptree_leaf_proof(const ts_taptree* t, uint64_t index, uint8_t leaf_hash[32],
                                uint8_t* path, uint32_t cap_nodes, uint32_t* depth);
/* verify_inclusion (complete_taptree.rs:53-64), host; 1 = included */
int ts_taptree_verify_inclusion(const uint8_t root[32], const uint8_t leaf_hash[32], const uint8_t* path,
                                uint32_t depth);
void ts_taptree_free(ts_taptree* t);
/* TapTreeMmcs::commit (taptree_mmcs.rs:101-114 -> tcs/mod.rs:284-292,238-282,339-378): num_queries
 * trees over the rows of the matrices (tallest first, power-of-two heights, consumed).  u32_size =
 * F::U32_SIZE (1 or 4): an evaluation is u32_size consecutive columns.  Tree q uses lock scripts
 * [q (1 + n_evals), (q + 1)(1 + n_evals)) of `lock_offsets` (num_queries (1 + n_evals) + 1 entries,
 * n_evals = total width / u32_size).  The leaf scripts are assembled and hashed on the device.
 * roots_out: num_queries x 32 bytes (Commitment = Vec<TreeRoot>). */
ts_status ts_tap_mmcs_commit(ts_ctx* ctx, uint32_t n_mats, ts_matrix* const* mats, uint32_t u32_size,
                             uint32_t num_queries, const uint8_t* lock_scripts,
                             const uint64_t* lock_offsets, uint8_t* roots_out, ts_tap_mmcs_data** out);
ts_status ts_tap_mmcs_info(const ts_tap_mmcs_data* d, uint32_t* n_mats, uint32_t* log_max_height,
                           uint32_t* n_evals, uint32_t* num_queries);
/* open_batch (taptree_mmcs.rs:46-75): rows of every matrix at index >> bits_reduced (concatenated),
 * the sibling path in tree query_times_index (log_max_height x 32 bytes) and, if script_len is not
 * NULL, the opened leaf's script (CommitedProof.leaf, tcs/mod.rs:103-108) */
ts_status ts_tap_mmcs_open_batch(const ts_tap_mmcs_data* d, uint32_t query_times_index, uint64_t index,
                                 uint32_t* rows_out, uint8_t* path_out, uint8_t* script_out,
                                 size_t script_cap, size_t* script_len);
/* verify_batch (taptree_mmcs.rs:77-99), host: rebuilds the leaf from the tree's lock scripts
 * (n_evals + 2 offsets), the index and the opened values, and checks its inclusion under `root` */
ts_status ts_tap_mmcs_verify_batch(const uint8_t* lock_scripts, const uint64_t* lock_offsets,
                                   uint32_t n_evals, uint32_t u32_size, uint64_t index,
                                   const uint32_t* opened_values, const uint8_t* path, uint32_t depth,
                                   const uint8_t root[32], int* ok);
void ts_tap_mmcs_free(ts_tap_mmcs_data* d);

/* uni_stark::prove / verify with TapTreeMmcs as the MMCS of the PCS and of FRI -- the reference's
 * own configuration (uni-stark/tests/fib_air.rs:117-131): every commitment is cfg->num_queries
 * taptrees, the challenger observes all their roots (basic/src/challenger/mod.rs:211-223), query q
 * opens in tree q (fri/src/prover.rs:50-56).  The lock scripts are one flat table in the order the
 * reference's bit-commitment manager hands them out (tcs/mod.rs:251-260), i.e. commit order:
 *   trace: Q (1 + width); quotient chunks: Q (1 + 4 quotient_degree); each of the log2(n) FRI rounds:
 *   Q (1 + 2)  -- per tree: the index lock first, then one lock per evaluation
 * (n_scripts entries, n_scripts + 1 offsets).  Proof = TSPF v2: v1 with a sixth header word
 * (num_queries) and num_queries x 8 words per commitment, a digest being its 32 bytes read as
 * little-endian words.  ts_verify_tap is host only; verdict codes as ts_verify. */
ts_status ts_prove_tap(ts_ctx* ctx, const ts_fri_config* cfg, const ts_air* air, ts_challenger* chal,
                       ts_matrix* trace, const uint32_t* public_values, uint32_t n_public,
                       const uint8_t* lock_scripts, const uint64_t* lock_offsets, size_t n_scripts,
                       uint32_t* proof_out, size_t cap_words, size_t* n_words_out);
/* ts_prove_tap with the commitments of ONE proof split over comm->world GPUs BY TREE (the
 * reference clones the tree per query, tcs/mod.rs:284-292: the trees of a commitment are independent).
 * Every rank passes the WHOLE trace and a challenger in the same state and repeats the numeric
 * pipeline (milliseconds); rank g builds trees [g per, (g+1) per), per = ceil(num_queries / world), of
 * every commitment -- the SHA-256 of kilobyte leaf scripts, which is where the seconds go -- the roots
 * are all-gathered (32 bytes per tree), and query q is answered by the rank that owns tree q.  Every
 * rank receives the whole proof, identical to ts_prove_tap's.  Only comm->all_gather is used. */
ts_status ts_prove_tap_sharded(ts_ctx* ctx, const ts_fri_config* cfg, const ts_comm* comm, const ts_air* air,
                               ts_challenger* chal, ts_matrix* trace, const uint32_t* public_values,
                               uint32_t n_public, const uint8_t* lock_scripts, const uint64_t* lock_offsets,
                               size_t n_scripts, uint32_t* proof_out, size_t cap_words, size_t* n_words_out);
ts_status ts_verify_tap(const ts_fri_config* cfg, const ts_air* air, ts_challenger* chal,
                        const uint32_t* proof, size_t n_words, const uint32_t* public_values,
                        uint32_t n_public, const uint8_t* lock_scripts, const uint64_t* lock_offsets,
                        size_t n_scripts, int* verdict);

/* Measurement aid: the whole-chip rate of NTT butterflies (kind 0), Blake3 compressions (kind 1) or
 * SHA-256 compressions (kind 2) with no memory traffic -- kinds 3 / 4: the butterflies with the LDS traffic
 * of a contiguous NTT pass added (4-byte / 16-byte accesses; measurement of what LDS costs in power) --, using the library's own arithmetic -- the integer-ALU ceiling bench.py
 * reports beside the achieved rates. */
ts_status ts_bench_alu(ts_ctx* ctx, int kind, double* units_per_second);

/* Measurement aid: one stage of the path in a sustained loop on resident, arbitrary data, mean
 * milliseconds per repetition (HIP events on the context's stream).  stage 0: the coset LDE of a
 * 2^log_n x width matrix (fri/src/two_adic_pcs.rs:233-241: all NTT passes, every coset); stage 1: the
 * hashing of BFMmcs::commit (basic/src/mmcs/bf_mmcs.rs:22-35) over a 2^(log_n + log_blowup) x width
 * matrix; stages 2, 3, 4: one pass of that LDE alone (inverse contiguous stages / strided middle /
 * forward contiguous stages; log_n > 12).  What tools/power_per_stage.py samples clock and power over, per kernel family and per
 * working-set size. */
ts_status ts_bench_stage(ts_ctx* ctx, int stage, unsigned log_n, uint32_t width, unsigned log_blowup,
                         uint32_t reps, double* ms_per_rep);

/* library/ABI version (bumped on any incompatible change) */
uint32_t ts_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
