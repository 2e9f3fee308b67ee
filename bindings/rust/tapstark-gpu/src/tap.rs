//! `prove` / `verify` with the reference's own MMCS, `TapTreeMmcs`, as the input MMCS of the PCS and as
//! the FRI MMCS (uni-stark/tests/fib_air.rs:117-131) -- `ts_prove_tap`, `ts_prove_tap_sharded`,
//! `ts_verify_tap`.  A commitment is `num_queries` taptrees; the leaf scripts are assembled and hashed
//! on the device from the committed values and the trees' LOCK SCRIPTS, which are an input: they come
//! from the bit-commitment manager of the reference (un-vendored `bitcomm` / `primitives` crates).
//!
//! `LockTable::from_manager` replays, in commit order, the assignments the reference's `commit_polys`
//! makes (basic/src/tcs/mod.rs:251-260: one `assign_bc(CommitType::U32)` for the index, then one per
//! evaluation), once per tree (`commit_poly_with_query_times`, :284-292), for the three kinds of
//! commitment of a proof: trace, quotient chunks, FRI rounds.
use std::ptr;

use primitives::{BCManager, BCommitOperator, BCommitWithSecret, CommitType, CompressType, SecretGen};
use p3_air::Air;
use p3_field::PrimeField32;
use p3_matrix::dense::RowMajorMatrix;
use p3_matrix::Matrix;
use uni_stark::SymbolicAirBuilder;

use crate::comm::RcclComm;
use crate::context::{DeviceMatrix, GpuChallenger};
use crate::ffi::*;
use crate::pcs::GpuFriPcs;
use crate::proof::{Proof, Val};
use crate::prove::CompiledAir;

/// Every lock script of a proof, flat: `bytes[offsets[i] .. offsets[i + 1]]` is script i.
pub struct LockTable {
    pub bytes: Vec<u8>,
    pub offsets: Vec<u64>,
}
impl LockTable {
    pub fn n_scripts(&self) -> usize {
        self.offsets.len() - 1
    }
    fn push(&mut self, script: &[u8]) {
        self.bytes.extend_from_slice(script);
        self.offsets.push(self.bytes.len() as u64);
    }
    /// trace: Q x (1 + width) U32 locks; quotient chunks: Q x (1 + 4 qd) U32 locks; each of the
    /// log2(degree) FRI rounds: Q x (1 index + 2 U128 locks) -- include/tapstark.h, `ts_prove_tap`.
    /// Takes the bit-commitment manager itself: `SyncBcManager::assign_bc` (tcs/mod.rs:59) is private
    /// to `basic`; a maintainer either passes the inner manager (a fresh `BM::new(SG::new())`, what
    /// `SyncBcManager::new` wraps, :63-68) or makes that method `pub`.  The verifier side must build
    /// its table from a manager in the same state, exactly as it must for the reference's own MMCS.
    pub fn from_manager<BM, SG, BC, B>(manager: &mut BM, num_queries: usize, width: usize,
                                       quotient_degree: usize, log_degree: usize) -> Self
    where
        BM: BCManager<SG, B, BC>,
        SG: SecretGen,
        BC: BCommitOperator<B>,
        B: BCommitWithSecret,
    {
        let mut t = LockTable { bytes: Vec::new(), offsets: vec![0] };
        let mut commitment = |t: &mut LockTable, n_evals: usize, ty: CommitType| {
            for _ in 0..num_queries {
                let index_bc = manager.assign_bc(CommitType::U32);
                t.push(index_bc.locking_script_with_type(CompressType::U32).compile().as_bytes());
                for _ in 0..n_evals {
                    let bc = manager.assign_bc(ty.clone());
                    t.push(bc.locking_script_with_type(CompressType::U32).compile().as_bytes());
                }
            }
        };
        commitment(&mut t, width, CommitType::U32);
        commitment(&mut t, 4 * quotient_degree, CommitType::U32);
        for _ in 0..log_degree {
            commitment(&mut t, 2, CommitType::U128);
        }
        t
    }
}

fn capacity(pcs: &GpuFriPcs<'_>, degree: usize, width: usize, qd: usize) -> usize {
    let log_degree = degree.trailing_zeros() as usize;
    let log_n = log_degree + pcs.fri.log_blowup;
    let q = pcs.fri.num_queries;
    64 + 8 * width + 16 * qd + 16 * q + 8 * q * log_degree
        + q * (16 + width + 5 * qd + 16 * log_n + log_degree * (9 + 8 * log_n))
}

/// The reference's `prove` over `TapTreeMmcs`; with `comm`, the commitments of this ONE proof are
/// split by tree over the ranks (every rank passes the whole trace and gets the whole proof).
pub fn prove_gpu_tap<A>(pcs: &GpuFriPcs<'_>, air: &A, challenger: &mut GpuChallenger, trace: RowMajorMatrix<Val>,
                        public_values: &Vec<Val>, locks: &LockTable, comm: Option<&RcclComm>) -> Proof
where
    A: Air<SymbolicAirBuilder<Val>>,
{
    let ctx = pcs.ctx;
    let cair = CompiledAir::new(ctx, air, public_values.len());
    let words: Vec<u32> = trace.values.iter().map(|v| v.as_canonical_u32()).collect();
    let pis: Vec<u32> = public_values.iter().map(|v| v.as_canonical_u32()).collect();
    let m = DeviceMatrix::upload(ctx, &words, trace.height(), trace.width()).into_raw();
    let cfg = pcs.fri.raw();
    let mut out = vec![0u32; capacity(pcs, trace.height(), trace.width(), 1 << cair.log_quotient_degree)];
    let mut n = 0usize;
    let pis_p = if pis.is_empty() { ptr::null() } else { pis.as_ptr() };
    let rc = unsafe {
        match comm {
            None => ts_prove_tap(ctx.raw, &cfg, cair.raw, challenger.raw, m, pis_p, pis.len() as u32,
                                 locks.bytes.as_ptr(), locks.offsets.as_ptr(), locks.n_scripts(), out.as_mut_ptr(),
                                 out.len(), &mut n),
            Some(c) => ts_prove_tap_sharded(ctx.raw, &cfg, &c.comm, cair.raw, challenger.raw, m, pis_p,
                                            pis.len() as u32, locks.bytes.as_ptr(), locks.offsets.as_ptr(),
                                            locks.n_scripts(), out.as_mut_ptr(), out.len(), &mut n),
        }
    };
    ctx.check(rc, "ts_prove_tap");
    unsafe { ts_matrix_free(ctx.raw, m) };
    Proof::from_tspf(&out[..n]) // TSPF v2: `Commitment` = the num_queries roots, as in the reference
}

/// `verify` for a proof made by `prove_gpu_tap` (host only); the verdict codes are `ts_verify`'s,
/// 0 = accept.  `proof_words` = the TSPF v2 words (`Proof` keeps no copy of them: a caller that wants
/// to verify natively passes what the prover wrote, or re-encodes with `ts_proof_from_postcard`).
pub fn verify_gpu_tap<A>(pcs: &GpuFriPcs<'_>, air: &A, challenger: &mut GpuChallenger, proof_words: &[u32],
                         public_values: &Vec<Val>, locks: &LockTable) -> i32
where
    A: Air<SymbolicAirBuilder<Val>>,
{
    let cair = CompiledAir::new(pcs.ctx, air, public_values.len());
    let pis: Vec<u32> = public_values.iter().map(|v| v.as_canonical_u32()).collect();
    let cfg = pcs.fri.raw();
    let mut verdict = -1;
    pcs.ctx.check(
        unsafe {
            ts_verify_tap(&cfg, cair.raw, challenger.raw, proof_words.as_ptr(), proof_words.len(),
                          if pis.is_empty() { ptr::null() } else { pis.as_ptr() }, pis.len() as u32,
                          locks.bytes.as_ptr(), locks.offsets.as_ptr(), locks.n_scripts(), &mut verdict)
        },
        "ts_verify_tap",
    );
    verdict
}
