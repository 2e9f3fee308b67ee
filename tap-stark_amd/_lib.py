"""ctypes loader of the C-ABI HIP library (include/tapstark.h).

There is no CPU fallback: if the library is missing or no HIP device is present, every product
entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# TS_LIB_PATH points the binding at another build of the same library (same-box A/B of kernels)
LIB_PATH = os.environ.get("TS_LIB_PATH") or os.path.join(PKG, "lib", "libtapstark_hip.so")

u32p = C.POINTER(C.c_uint32)
voidpp = C.POINTER(C.c_void_p)

# every symbol include/tapstark.h declares (tests check the built library exports all of them)
ABI_SYMBOLS = [
    "ts_abi_version", "ts_device_count", "ts_ctx_create", "ts_ctx_destroy", "ts_last_error", "ts_ctx_synchronize",
    "ts_ctx_stream", "ts_ctx_set_timing", "ts_ctx_take_timings", "ts_ctx_set_replay", "ts_ctx_set_kernel_timing",
    "ts_ctx_take_kernel_timings", "ts_ctx_graph_stats", "ts_ctx_stat", "ts_matrix_upload",
    "ts_matrix_from_device", "ts_trace_fibonacci", "ts_trace_synth_mul", "ts_trace_synth_ext", "ts_matrix_dims", "ts_matrix_download", "ts_matrix_free",
    "ts_air_compile", "ts_air_info", "ts_air_is_jit", "ts_air_jit_wait", "ts_air_free", "ts_air_program", "ts_air_jit_source", "ts_air_jit_compile", "ts_pcs_commit", "ts_mmcs_commit", "ts_pcs_data_lde",
    "ts_pcs_data_info", "ts_pcs_data_matrix_info", "ts_pcs_data_digests", "ts_pcs_open_batch", "ts_pcs_data_free",
    "ts_quotient_chunks", "ts_pcs_open_reduce", "ts_pcs_open", "ts_pcs_verify", "ts_fri_prove", "ts_fri_verify", "ts_fri_fold", "ts_fri_fold_device", "ts_chal_new", "ts_chal_clone",
    "ts_chal_free", "ts_chal_observe", "ts_chal_observe_commitment", "ts_chal_sample",
    "ts_chal_sample_bits", "ts_chal_check_witness", "ts_chal_grind", "ts_chal_state", "ts_prove", "ts_prove_stream", "ts_prove_sharded", "ts_verify", "ts_check_constraints",
    "ts_proof_to_postcard", "ts_proof_from_postcard", "ts_proof_from_postcard_v",
    "ts_rccl_available", "ts_rccl_unique_id", "ts_comm_rccl_create", "ts_comm_rccl_destroy",
    "ts_comm_rccl_info",
    "ts_comm_local_group_create", "ts_comm_local_get", "ts_comm_local_group_destroy",
    "ts_comm_local_group_reset", "ts_comm_local_group_set_timeout",
    "ts_bench_alu", "ts_bench_stage", "ts_host_alloc", "ts_host_free", "ts_matrix_upload_async",
    "ts_tapleaf_hash", "ts_tapbranch_hash", "ts_tap_winternitz_lock_script", "ts_tap_leaf_script",
    "ts_taptree_from_scripts", "ts_taptree_combine", "ts_taptree_info", "ts_taptree_leaf_proof",
    "ts_taptree_verify_inclusion", "ts_taptree_free", "ts_tap_mmcs_commit", "ts_tap_mmcs_info",
    "ts_tap_mmcs_open_batch", "ts_tap_mmcs_verify_batch", "ts_tap_mmcs_free",
    "ts_prove_tap", "ts_prove_tap_sharded", "ts_verify_tap",
]

STATUS = {0: "TS_OK", 1: "TS_ERR_INVALID", 2: "TS_ERR_HIP", 3: "TS_ERR_OOM",
          4: "TS_ERR_UNSUPPORTED", 5: "TS_ERR_INVARIANT", 6: "TS_ERR_BUFFER", 7: "TS_ERR_COMM"}


class TsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class FriConfigC(C.Structure):
    _fields_ = [("log_blowup", C.c_uint32), ("num_queries", C.c_uint32),
                ("proof_of_work_bits", C.c_uint32)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
BROADCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
ABORT_FN = C.CFUNCTYPE(None, C.c_void_p)


class CommC(C.Structure):
    """``ts_comm`` (include/tapstark.h)."""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("user", C.c_void_p),
                ("all_gather", ALL_GATHER_FN), ("broadcast", BROADCAST_FN), ("abort", ABORT_FN)]


class RcclInfoC(C.Structure):
    """``ts_rccl_info`` (include/tapstark.h)."""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("comm_count", C.c_int),
                ("comm_user_rank", C.c_int), ("comm_device", C.c_int), ("rccl_version", C.c_int),
                ("aborted", C.c_int), ("checked", C.c_int)]


class ShardOptionsC(C.Structure):
    """``ts_shard_options`` (include/tapstark.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("min_local_log", C.c_uint32), ("trace_replicated", C.c_uint32),
                ("local_quotient", C.c_uint32)]


_lib = None


def lib() -> C.CDLL:
    """Loads the library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TsError(2, f"{LIB_PATH} is missing: run __graft_entry__.build() "
                             "(python -m tapstark_amd.build); there is no CPU fallback")
        # PyTorch-ROCm wheels carry their own libamdhip64; a process that loaded the system copy
        # first cannot initialise torch's afterwards ("No HIP GPUs are available").  Loading torch's
        # first makes both sides share one runtime (same SONAME), which the sharded prover needs
        # (tap-stark_amd/dist.py).  TS_PRELOAD_TORCH=0 skips it for torch-free hosts.
        if os.environ.get("TS_PRELOAD_TORCH", "1") != "0":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        l = C.CDLL(LIB_PATH)
        l.ts_last_error.restype = C.c_char_p
        l.ts_last_error.argtypes = [C.c_void_p]
        l.ts_ctx_stream.restype = C.c_void_p
        l.ts_ctx_stream.argtypes = [C.c_void_p]
        l.ts_chal_sample_bits.restype = C.c_uint64
        l.ts_abi_version.restype = C.c_uint32
        for name in ("ts_ctx_destroy", "ts_matrix_free", "ts_air_free", "ts_pcs_data_free",
                     "ts_chal_free", "ts_chal_observe", "ts_chal_observe_commitment",
                     "ts_chal_sample", "ts_chal_state"):
            getattr(l, name).restype = None
        l.ts_ctx_destroy.argtypes = [C.c_void_p]
        l.ts_matrix_free.argtypes = [C.c_void_p, C.c_void_p]
        l.ts_air_free.argtypes = [C.c_void_p, C.c_void_p]
        l.ts_pcs_data_free.argtypes = [C.c_void_p, C.c_void_p]
        l.ts_chal_free.argtypes = [C.c_void_p]
        l.ts_chal_observe.argtypes = [C.c_void_p, C.c_uint32]
        l.ts_chal_observe_commitment.argtypes = [C.c_void_p, u32p]
        l.ts_chal_sample.argtypes = [C.c_void_p, u32p]
        l.ts_chal_state.argtypes = [C.c_void_p, u32p]
        l.ts_chal_sample_bits.argtypes = [C.c_void_p, C.c_uint32]
        l.ts_chal_check_witness.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        l.ts_chal_grind.argtypes = [C.c_void_p, C.c_uint32, u32p]
        l.ts_chal_new.argtypes = [C.c_int, C.c_int, voidpp]
        l.ts_chal_clone.argtypes = [C.c_void_p, voidpp]
        l.ts_ctx_create.argtypes = [C.c_int, voidpp]
        l.ts_ctx_synchronize.argtypes = [C.c_void_p]
        l.ts_ctx_set_timing.argtypes = [C.c_void_p, C.c_int]
        l.ts_ctx_take_timings.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        l.ts_ctx_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        l.ts_ctx_set_replay.argtypes = [C.c_void_p, C.c_int]
        l.ts_ctx_take_kernel_timings.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        l.ts_matrix_upload.argtypes = [C.c_void_p, u32p, C.c_uint64, C.c_uint32, voidpp]
        l.ts_matrix_from_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, voidpp]
        l.ts_trace_fibonacci.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, voidpp]
        l.ts_trace_synth_mul.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, voidpp]
        l.ts_trace_synth_ext.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, voidpp]
        l.ts_matrix_dims.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), u32p]
        l.ts_matrix_download.argtypes = [C.c_void_p, C.c_void_p, u32p]
        l.ts_air_compile.argtypes = [C.c_void_p, u32p, C.c_size_t, voidpp]
        l.ts_air_info.argtypes = [C.c_void_p, u32p, u32p, u32p, u32p]
        l.ts_air_is_jit.argtypes = [C.c_void_p]
        l.ts_air_jit_wait.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        l.ts_air_program.argtypes = [C.c_void_p, u32p, C.c_size_t, C.POINTER(C.c_size_t)]
        l.ts_air_jit_source.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        l.ts_air_jit_compile.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_double)]
        l.ts_pcs_commit.argtypes = [C.c_void_p, C.POINTER(FriConfigC), C.c_uint32, voidpp, u32p,
                                    u32p, voidpp]
        l.ts_mmcs_commit.argtypes = [C.c_void_p, C.c_uint32, voidpp, u32p, voidpp]
        l.ts_pcs_data_lde.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, u32p]
        l.ts_pcs_data_info.argtypes = [C.c_void_p, u32p, u32p]
        l.ts_pcs_data_digests.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, u32p]
        l.ts_pcs_open_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, u32p, u32p]
        l.ts_quotient_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, u32p,
                                         C.c_uint32, u32p, voidpp]
        l.ts_pcs_open_reduce.argtypes = [C.c_void_p, C.POINTER(FriConfigC), C.c_void_p, C.c_void_p,
                                         u32p, u32p, u32p, u32p]
        l.ts_pcs_data_matrix_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), u32p]
        l.ts_pcs_open.argtypes = [C.c_void_p, C.POINTER(FriConfigC), C.c_void_p, C.c_uint32, voidpp,
                                  u32p, u32p, u32p, C.c_size_t, C.POINTER(C.c_size_t), u32p,
                                  C.c_size_t, C.POINTER(C.c_size_t)]
        l.ts_pcs_verify.argtypes = [C.POINTER(FriConfigC), C.c_void_p, C.c_uint32, u32p, u32p, u32p, u32p,
                                    u32p, u32p, u32p, u32p, C.c_size_t, C.POINTER(C.c_int)]
        l.ts_fri_prove.argtypes = [C.c_void_p, C.POINTER(FriConfigC), C.c_void_p, C.c_uint32, u32p,
                                   C.POINTER(u32p), u32p, C.c_size_t, C.POINTER(C.c_size_t)]
        l.ts_fri_verify.argtypes = [C.POINTER(FriConfigC), C.c_void_p, u32p, C.c_size_t,
                                    C.POINTER(C.c_int)]
        l.ts_fri_fold.argtypes = [C.c_void_p, u32p, C.c_uint64, u32p, u32p]
        l.ts_fri_fold_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, u32p, C.c_void_p]
        l.ts_prove.argtypes = [C.c_void_p, C.POINTER(FriConfigC), C.c_void_p, C.c_void_p, C.c_void_p,
                               u32p, C.c_uint32, u32p, C.c_size_t, C.POINTER(C.c_size_t)]
        l.ts_prove_stream.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(FriConfigC),
                                      C.POINTER(C.c_void_p), u32p, C.c_uint32, u32p, C.c_uint32, C.c_double,
                                      u32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)]
        l.ts_prove_sharded.argtypes = [C.c_void_p, C.POINTER(FriConfigC), C.POINTER(CommC), C.c_void_p,
                                       C.c_void_p, C.c_void_p, u32p, C.c_uint32,
                                       C.POINTER(ShardOptionsC), u32p, C.c_size_t,
                                       C.POINTER(C.c_size_t)]
        u8p = C.POINTER(C.c_uint8)
        l.ts_rccl_unique_id.argtypes = [u8p]
        l.ts_comm_rccl_create.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.POINTER(CommC), voidpp]
        l.ts_comm_rccl_destroy.argtypes = [C.c_void_p]
        l.ts_comm_rccl_destroy.restype = None
        l.ts_comm_rccl_info.argtypes = [C.c_void_p, C.POINTER(RcclInfoC)]
        l.ts_comm_local_group_reset.argtypes = [C.c_void_p]
        l.ts_comm_local_group_set_timeout.argtypes = [C.c_void_p, C.c_int]
        l.ts_comm_local_group_create.argtypes = [C.c_int, voidpp]
        l.ts_comm_local_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(CommC)]
        l.ts_comm_local_group_destroy.argtypes = [C.c_void_p]
        l.ts_comm_local_group_destroy.restype = None
        u64p = C.POINTER(C.c_uint64)
        szp = C.POINTER(C.c_size_t)
        l.ts_bench_alu.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        l.ts_bench_stage.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint32, C.c_uint, C.c_uint32,
                                     C.POINTER(C.c_double)]
        l.ts_ctx_graph_stats.argtypes = [C.c_void_p, u64p]
        l.ts_ctx_stat.argtypes = [C.c_void_p, C.c_int, u64p]
        l.ts_host_alloc.argtypes = [C.c_size_t, voidpp]
        l.ts_host_free.argtypes = [C.c_void_p]
        l.ts_host_free.restype = None
        l.ts_matrix_upload_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, voidpp]
        l.ts_tapleaf_hash.argtypes = [C.c_char_p, C.c_size_t, u8p]
        l.ts_tapbranch_hash.argtypes = [u8p, u8p, u8p]
        l.ts_tap_winternitz_lock_script.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, u8p, C.c_size_t, szp]
        l.ts_tap_leaf_script.argtypes = [C.c_char_p, u64p, C.c_uint32, C.c_uint32, C.c_uint64, u32p, u8p,
                                         C.c_size_t, szp]
        l.ts_taptree_from_scripts.argtypes = [C.c_void_p, C.c_char_p, u64p, C.c_uint64, voidpp]
        l.ts_taptree_combine.argtypes = [C.c_void_p, C.c_void_p, voidpp]
        l.ts_taptree_info.argtypes = [C.c_void_p, u64p, u8p]
        l.ts_taptree_leaf_proof.argtypes = [C.c_void_p, C.c_uint64, u8p, u8p, C.c_uint32, u32p]
        l.ts_taptree_verify_inclusion.argtypes = [u8p, u8p, u8p, C.c_uint32]
        l.ts_taptree_free.argtypes = [C.c_void_p]
        l.ts_taptree_free.restype = None
        l.ts_tap_mmcs_commit.argtypes = [C.c_void_p, C.c_uint32, voidpp, C.c_uint32, C.c_uint32, C.c_char_p,
                                         u64p, u8p, voidpp]
        l.ts_tap_mmcs_info.argtypes = [C.c_void_p, u32p, u32p, u32p, u32p]
        l.ts_tap_mmcs_open_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, u32p, u8p, u8p, C.c_size_t, szp]
        l.ts_tap_mmcs_verify_batch.argtypes = [C.c_char_p, u64p, C.c_uint32, C.c_uint32, C.c_uint64, u32p, u8p,
                                               C.c_uint32, u8p, C.POINTER(C.c_int)]
        l.ts_prove_tap.argtypes = [C.c_void_p, C.POINTER(FriConfigC), C.c_void_p, C.c_void_p, C.c_void_p, u32p,
                                   C.c_uint32, C.c_char_p, u64p, C.c_size_t, u32p, C.c_size_t, szp]
        l.ts_prove_tap_sharded.argtypes = [C.c_void_p, C.POINTER(FriConfigC), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, u32p, C.c_uint32, C.c_char_p, u64p, C.c_size_t, u32p,
                                           C.c_size_t, szp]
        l.ts_verify_tap.argtypes = [C.POINTER(FriConfigC), C.c_void_p, C.c_void_p, u32p, C.c_size_t, u32p,
                                    C.c_uint32, C.c_char_p, u64p, C.c_size_t, C.POINTER(C.c_int)]
        l.ts_tap_mmcs_free.argtypes = [C.c_void_p]
        l.ts_tap_mmcs_free.restype = None
        l.ts_proof_to_postcard.argtypes = [u32p, C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        l.ts_proof_from_postcard.argtypes = [u8p, C.c_size_t, u32p, C.c_size_t, C.POINTER(C.c_size_t)]
        l.ts_proof_from_postcard_v.argtypes = [u8p, C.c_size_t, C.c_int, u32p, C.c_size_t, C.POINTER(C.c_size_t)]
        l.ts_check_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u32p, C.c_uint32,
                                           C.POINTER(C.c_int64)]
        l.ts_verify.argtypes = [C.POINTER(FriConfigC), C.c_void_p, C.c_void_p, u32p, C.c_size_t, u32p,
                                C.c_uint32, C.POINTER(C.c_int)]
        _lib = l
    return _lib
