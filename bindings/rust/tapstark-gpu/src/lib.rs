//! tapstark-gpu: the reference's `prove` / `Pcs` surface on `libtapstark_hip.so` (MI355X).
//!
//! **Uncompiled in this repository's pipeline** (no Rust toolchain in the build image; see
//! Cargo.toml).  Written against tap-stark at the commit SURVEY.md names and include/tapstark.h
//! ABI version 5.  The tested callers of the same C ABI are the ctypes binding
//! (tap-stark_amd/stark.py) and examples/fib_air.cpp.
//!
//! ```ignore
//! // uni-stark/tests/fib_air.rs:117-149 with the GPU prover
//! let ctx = GpuContext::new(0);
//! let pcs = GpuFriPcs { ctx: &ctx, fri: FriConfig { log_blowup: 2, num_queries: 28, proof_of_work_bits: 8 } };
//! let trace = generate_trace_rows::<Val>(0, 1, 1 << 3);
//! let pis = vec![BabyBear::from_canonical_u64(0), BabyBear::from_canonical_u64(1), BabyBear::from_canonical_u64(21)];
//! let proof = prove_gpu(&pcs, &FibonacciAir {}, &mut GpuChallenger::new(), trace, &pis);
//! let bytes = postcard::to_allocvec(&proof).unwrap();      // == ts_proof_to_postcard
//! ```
pub mod air;
pub mod challenger;
pub mod comm;
pub mod context;
pub mod ffi;
pub mod pcs;
pub mod proof;
pub mod prove;
pub mod tap;

pub use air::serialize_constraints;
pub use comm::{prove_gpu_sharded, rccl_unique_id, RcclComm};
pub use context::{DeviceMatrix, GpuChallenger, GpuContext};
pub use pcs::{FriConfig, GpuFriPcs, GpuPcsError, GpuProverData};
pub use proof::Proof;
pub use prove::{prove_gpu, prove_gpu_stepwise, CompiledAir};
pub use tap::{prove_gpu_tap, verify_gpu_tap, GpuTapProof, GpuTapProverData, GpuTapTreeMmcs, LockTable};
