//! `extern "C"` declarations of include/tapstark.h (ABI version 5).  One line per entry point the
//! Rust side uses; the header is the authority for argument meaning.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_uint, c_void};

macro_rules! opaque { ($($n:ident),*) => { $( #[repr(C)] pub struct $n { _p: [u8; 0] } )* } }
opaque!(ts_ctx, ts_matrix, ts_air, ts_pcs_data, ts_challenger, ts_rccl_comm, ts_comm_group, ts_taptree,
        ts_tap_mmcs_data);

pub type ts_status = c_int;
pub const TS_OK: ts_status = 0;
pub const TS_ERR_INVARIANT: ts_status = 5; // where the reference would have panicked
pub const TS_ERR_BUFFER: ts_status = 6;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ts_fri_config {
    pub log_blowup: u32,
    pub num_queries: u32,
    pub proof_of_work_bits: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct ts_comm {
    pub rank: c_int,
    pub world: c_int,
    pub user: *mut c_void,
    pub all_gather: Option<unsafe extern "C" fn(*mut c_void, *const c_void, *mut c_void, usize, *mut c_void) -> c_int>,
    pub broadcast: Option<unsafe extern "C" fn(*mut c_void, *mut c_void, usize, c_int, *mut c_void) -> c_int>,
    pub abort: Option<unsafe extern "C" fn(*mut c_void)>,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct ts_shard_options {
    pub struct_size: u32, // = size_of::<ts_shard_options>() (the library refuses any other layout)
    pub min_local_log: u32,
    pub trace_replicated: u32,
    pub local_quotient: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct ts_rccl_info {
    pub rank: c_int,
    pub world: c_int,
    pub comm_count: c_int,
    pub comm_user_rank: c_int,
    pub comm_device: c_int,
    pub rccl_version: c_int,
    pub aborted: c_int,
    pub checked: c_int,
}

extern "C" {
    pub fn ts_abi_version() -> u32;
    pub fn ts_ctx_graph_stats(ctx: *mut ts_ctx, out: *mut u64) -> ts_status;
    pub fn ts_ctx_stat(ctx: *mut ts_ctx, which: c_int, out: *mut u64) -> ts_status;
    pub fn ts_device_count() -> c_int;
    pub fn ts_bench_stage(ctx: *mut ts_ctx, stage: c_int, log_n: c_uint, width: u32, log_blowup: c_uint,
                          reps: u32, ms_per_rep: *mut f64) -> ts_status;
    pub fn ts_ctx_create(device: c_int, out: *mut *mut ts_ctx) -> ts_status;
    pub fn ts_ctx_destroy(ctx: *mut ts_ctx);
    pub fn ts_last_error(ctx: *const ts_ctx) -> *const c_char;
    pub fn ts_ctx_synchronize(ctx: *mut ts_ctx) -> ts_status;

    pub fn ts_matrix_upload(ctx: *mut ts_ctx, host_row_major: *const u32, height: u64, width: u32,
                            out: *mut *mut ts_matrix) -> ts_status;
    /// page-locked host memory + asynchronous upload on the context's stream (the PCIe-rate path)
    pub fn ts_host_alloc(bytes: usize, out: *mut *mut c_void) -> ts_status;
    pub fn ts_host_free(p: *mut c_void);
    pub fn ts_matrix_upload_async(ctx: *mut ts_ctx, host_pinned: *const u32, height: u64, width: u32,
                                  out: *mut *mut ts_matrix) -> ts_status;
    pub fn ts_matrix_download(ctx: *mut ts_ctx, m: *const ts_matrix, host_row_major: *mut u32) -> ts_status;
    pub fn ts_matrix_dims(m: *const ts_matrix, height: *mut u64, width: *mut u32) -> ts_status;
    pub fn ts_matrix_free(ctx: *mut ts_ctx, m: *mut ts_matrix);

    pub fn ts_air_compile(ctx: *mut ts_ctx, tape: *const u32, n_words: usize, out: *mut *mut ts_air) -> ts_status;
    pub fn ts_air_info(air: *const ts_air, width: *mut u32, n_public: *mut u32, max_degree: *mut u32,
                       log_quotient_degree: *mut u32) -> ts_status;
    pub fn ts_air_free(ctx: *mut ts_ctx, air: *mut ts_air);
    pub fn ts_air_is_jit(air: *const ts_air) -> c_int;
    pub fn ts_air_jit_wait(ctx: *mut ts_ctx, air: *mut ts_air, state: *mut c_int, compile_seconds: *mut f64) -> ts_status;
    pub fn ts_air_program(air: *const ts_air, out: *mut u32, cap_words: usize, n_words: *mut usize) -> ts_status;

    pub fn ts_pcs_commit(ctx: *mut ts_ctx, cfg: *const ts_fri_config, n_mats: u32,
                         evals: *const *mut ts_matrix, domain_shifts: *const u32, root_out: *mut u32,
                         out: *mut *mut ts_pcs_data) -> ts_status;
    pub fn ts_pcs_data_info(d: *const ts_pcs_data, n_mats: *mut u32, log_height: *mut u32) -> ts_status;
    pub fn ts_pcs_data_matrix_info(d: *const ts_pcs_data, idx: u32, height: *mut u64, width: *mut u32) -> ts_status;
    pub fn ts_pcs_data_lde(ctx: *mut ts_ctx, d: *const ts_pcs_data, idx: u32, host_row_major: *mut u32) -> ts_status;
    pub fn ts_pcs_data_free(ctx: *mut ts_ctx, d: *mut ts_pcs_data);
    pub fn ts_quotient_chunks(ctx: *mut ts_ctx, trace_data: *const ts_pcs_data, log_blowup: u32,
                              air: *const ts_air, public_values: *const u32, n_public: u32,
                              alpha: *const u32, chunks_out: *mut *mut ts_matrix) -> ts_status;
    pub fn ts_pcs_open(ctx: *mut ts_ctx, cfg: *const ts_fri_config, chal: *mut ts_challenger, n_rounds: u32,
                       rounds: *const *const ts_pcs_data, n_points: *const u32, points: *const u32,
                       opened_out: *mut u32, opened_cap_words: usize, n_opened_words: *mut usize,
                       proof_out: *mut u32, proof_cap_words: usize, n_proof_words: *mut usize) -> ts_status;
    pub fn ts_pcs_verify(cfg: *const ts_fri_config, chal: *mut ts_challenger, n_rounds: u32,
                         commitments: *const u32, mats_per_round: *const u32, log_degrees: *const u32,
                         widths: *const u32, n_points: *const u32, points: *const u32,
                         opened_values: *const u32, fri_proof: *const u32, n_words: usize,
                         verdict: *mut c_int) -> ts_status;

    pub fn ts_chal_new(permutation: c_int, sample_ext: c_int, out: *mut *mut ts_challenger) -> ts_status;
    pub fn ts_chal_free(c: *mut ts_challenger);
    pub fn ts_chal_observe(c: *mut ts_challenger, word: u32);
    pub fn ts_chal_observe_commitment(c: *mut ts_challenger, d: *const u32);
    pub fn ts_chal_sample(c: *mut ts_challenger, out: *mut u32);
    pub fn ts_chal_state(c: *const ts_challenger, out: *mut u32);

    pub fn ts_prove(ctx: *mut ts_ctx, cfg: *const ts_fri_config, air: *const ts_air, chal: *mut ts_challenger,
                    trace: *mut ts_matrix, public_values: *const u32, n_public: u32, proof_out: *mut u32,
                    cap_words: usize, n_words_out: *mut usize) -> ts_status;
    pub fn ts_prove_sharded(ctx: *mut ts_ctx, cfg: *const ts_fri_config, comm: *const ts_comm,
                            air: *const ts_air, chal: *mut ts_challenger, trace_rows: *mut ts_matrix,
                            public_values: *const u32, n_public: u32, options: *const ts_shard_options,
                            proof_out: *mut u32, cap_words: usize, n_words_out: *mut usize) -> ts_status;
    pub fn ts_verify(cfg: *const ts_fri_config, air: *const ts_air, chal: *mut ts_challenger,
                     proof: *const u32, n_words: usize, public_values: *const u32, n_public: u32,
                     verdict: *mut c_int) -> ts_status;
    pub fn ts_proof_to_postcard(proof: *const u32, n_words: usize, out: *mut u8, cap_bytes: usize,
                                n_bytes_out: *mut usize) -> ts_status;
    /// tspf_version: 0 infer, 1 / 2 explicit (a taptree proof with one query needs 2)
    pub fn ts_proof_from_postcard_v(bytes: *const u8, n_bytes: usize, tspf_version: c_int, proof_out: *mut u32,
                                    cap_words: usize, n_words_out: *mut usize) -> ts_status;
    pub fn ts_fri_fold_device(ctx: *mut ts_ctx, in_dev: *const u32, h: u64, beta: *const u32,
                              out_dev: *mut u32) -> ts_status;

    // the reference's own MMCS: TapTreeMmcs (csrc/taptree.cpp, tap_prover.cpp)
    pub fn ts_tapleaf_hash(script: *const u8, len: usize, out: *mut u8) -> ts_status;
    pub fn ts_tapbranch_hash(a: *const u8, b: *const u8, out: *mut u8) -> ts_status;
    pub fn ts_tap_mmcs_commit(ctx: *mut ts_ctx, n_mats: u32, mats: *const *mut ts_matrix, u32_size: u32,
                              num_queries: u32, lock_scripts: *const u8, lock_offsets: *const u64,
                              roots_out: *mut u8, out: *mut *mut ts_tap_mmcs_data) -> ts_status;
    pub fn ts_tap_mmcs_info(d: *const ts_tap_mmcs_data, n_mats: *mut u32, log_max_height: *mut u32,
                            n_evals: *mut u32, num_queries: *mut u32) -> ts_status;
    pub fn ts_tap_mmcs_open_batch(d: *const ts_tap_mmcs_data, query_times_index: u32, index: u64,
                                  rows_out: *mut u32, path_out: *mut u8, script_out: *mut u8, script_cap: usize,
                                  script_len: *mut usize) -> ts_status;
    pub fn ts_tap_mmcs_verify_batch(lock_scripts: *const u8, lock_offsets: *const u64, n_evals: u32, u32_size: u32,
                                    index: u64, opened_values: *const u32, path: *const u8, depth: u32,
                                    root: *const u8, ok: *mut c_int) -> ts_status;
    pub fn ts_tap_mmcs_free(d: *mut ts_tap_mmcs_data);
    pub fn ts_prove_tap(ctx: *mut ts_ctx, cfg: *const ts_fri_config, air: *const ts_air, chal: *mut ts_challenger,
                        trace: *mut ts_matrix, public_values: *const u32, n_public: u32, lock_scripts: *const u8,
                        lock_offsets: *const u64, n_scripts: usize, proof_out: *mut u32, cap_words: usize,
                        n_words_out: *mut usize) -> ts_status;
    pub fn ts_prove_tap_sharded(ctx: *mut ts_ctx, cfg: *const ts_fri_config, comm: *const ts_comm,
                                air: *const ts_air, chal: *mut ts_challenger, trace: *mut ts_matrix,
                                public_values: *const u32, n_public: u32, lock_scripts: *const u8,
                                lock_offsets: *const u64, n_scripts: usize, proof_out: *mut u32, cap_words: usize,
                                n_words_out: *mut usize) -> ts_status;
    pub fn ts_verify_tap(cfg: *const ts_fri_config, air: *const ts_air, chal: *mut ts_challenger,
                         proof: *const u32, n_words: usize, public_values: *const u32, n_public: u32,
                         lock_scripts: *const u8, lock_offsets: *const u64, n_scripts: usize,
                         verdict: *mut c_int) -> ts_status;

    // native communicators (csrc/comm.cpp)
    pub fn ts_rccl_available() -> c_int;
    pub fn ts_rccl_unique_id(out: *mut u8) -> ts_status;
    pub fn ts_comm_rccl_create(ctx: *mut ts_ctx, unique_id: *const u8, rank: c_int, world: c_int,
                               out: *mut ts_comm, handle: *mut *mut ts_rccl_comm) -> ts_status;
    pub fn ts_comm_rccl_destroy(handle: *mut ts_rccl_comm);
    pub fn ts_comm_rccl_info(handle: *const ts_rccl_comm, out: *mut ts_rccl_info) -> ts_status;
    pub fn ts_comm_local_group_reset(group: *mut ts_comm_group) -> ts_status;
    pub fn ts_comm_local_group_set_timeout(group: *mut ts_comm_group, seconds: c_int) -> ts_status;
    pub fn ts_comm_local_group_create(world: c_int, out: *mut *mut ts_comm_group) -> ts_status;
    pub fn ts_comm_local_get(group: *mut ts_comm_group, rank: c_int, out: *mut ts_comm) -> ts_status;
    pub fn ts_comm_local_group_destroy(group: *mut ts_comm_group);
}
