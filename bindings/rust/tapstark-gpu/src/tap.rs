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

// ------------------------------------------------------------------ BFMmcs surface of TapTreeMmcs
/// `TapTreeMmcs<Val>` (basic/src/mmcs/taptree_mmcs.rs:24-119) with the trees built on the device:
/// `commit` / `open_batch` / `verify_batch` with the trait's argument meaning.  `Commitment` is the
/// reference's `Vec<TreeRoot>` (num_queries roots, each `[[u8; 4]; 8]`), `Proof` here is the sibling
/// path plus the opened leaf's script (`CommitedProof.leaf`, tcs/mod.rs:103-108) as bytes: turning
/// those into the reference's `CommitedProof<BO, B>` (which also carries the bit-commitment
/// operators) is the caller's, since `BO` / `B` live in the un-vendored crates.
pub struct GpuTapTreeMmcs<'c> {
    pub ctx: &'c crate::context::GpuContext,
    pub num_queries: usize,
    /// lock scripts of ONE commitment in tree order: num_queries x (1 + n_evals) scripts
    pub locks: LockTable,
}
pub struct GpuTapProverData<'c> {
    ctx: &'c crate::context::GpuContext,
    raw: *mut ts_tap_mmcs_data,
    widths: Vec<usize>,
    log_max_height: usize,
}
impl Drop for GpuTapProverData<'_> {
    fn drop(&mut self) {
        let _ = self.ctx;
        unsafe { ts_tap_mmcs_free(self.raw) }
    }
}
pub struct GpuTapProof {
    pub path: Vec<[u8; 32]>,
    pub leaf_script: Vec<u8>,
}

impl<'c> GpuTapTreeMmcs<'c> {
    /// `BFMmcs::commit` (taptree_mmcs.rs:101-114): base-field matrices (U32_SIZE = 1)
    pub fn commit(&self, inputs: Vec<RowMajorMatrix<Val>>) -> (Vec<[[u8; 4]; 8]>, GpuTapProverData<'c>) {
        let widths: Vec<usize> = inputs.iter().map(|m| m.width()).collect();
        let log_max_height = inputs.iter().map(|m| m.height()).max().unwrap().trailing_zeros() as usize;
        let mats: Vec<*mut ts_matrix> = inputs
            .iter()
            .map(|m| {
                let words: Vec<u32> = m.values.iter().map(|v| v.as_canonical_u32()).collect();
                DeviceMatrix::upload(self.ctx, &words, m.height(), m.width()).into_raw()
            })
            .collect();
        let mut roots = vec![0u8; 32 * self.num_queries];
        let mut raw = ptr::null_mut();
        self.ctx.check(
            unsafe {
                ts_tap_mmcs_commit(self.ctx.raw, mats.len() as u32, mats.as_ptr(), 1, self.num_queries as u32,
                                   self.locks.bytes.as_ptr(), self.locks.offsets.as_ptr(), roots.as_mut_ptr(),
                                   &mut raw)
            },
            "ts_tap_mmcs_commit",
        );
        for m in mats {
            unsafe { ts_matrix_free(self.ctx.raw, m) };
        }
        // TreeRoot = u256_to_u32(root bytes) (chan_field.rs:87-95): 8 groups of 4 bytes
        let commitment = roots
            .chunks(32)
            .map(|r| {
                let mut t = [[0u8; 4]; 8];
                for k in 0..8 {
                    t[k].copy_from_slice(&r[4 * k..4 * k + 4]);
                }
                t
            })
            .collect();
        (commitment, GpuTapProverData { ctx: self.ctx, raw, widths, log_max_height })
    }

    /// `BFMmcs::open_batch` (taptree_mmcs.rs:46-75)
    pub fn open_batch(&self, query_times_index: usize, query_index: usize, data: &GpuTapProverData<'c>)
                      -> (Vec<Vec<Val>>, GpuTapProof) {
        use p3_field::AbstractField;
        let total: usize = data.widths.iter().sum();
        let mut rows = vec![0u32; total];
        let mut path = vec![0u8; 32 * data.log_max_height];
        let mut script = vec![0u8; 1 << 20];
        let mut script_len = 0usize;
        self.ctx.check(
            unsafe {
                ts_tap_mmcs_open_batch(data.raw, query_times_index as u32, query_index as u64, rows.as_mut_ptr(),
                                       path.as_mut_ptr(), script.as_mut_ptr(), script.len(), &mut script_len)
            },
            "ts_tap_mmcs_open_batch",
        );
        script.truncate(script_len);
        let mut at = 0;
        let opened = data
            .widths
            .iter()
            .map(|&w| {
                let r = rows[at..at + w].iter().map(|&v| Val::from_canonical_u32(v)).collect();
                at += w;
                r
            })
            .collect();
        let path = path.chunks(32).map(|c| c.try_into().unwrap()).collect();
        (opened, GpuTapProof { path, leaf_script: script })
    }

    /// `BFMmcs::verify_batch` (taptree_mmcs.rs:77-99), host only: the leaf is rebuilt from the tree's
    /// lock scripts, `query_index` and the opened values, and checked against root `query_times_index`
    pub fn verify_batch(&self, query_times_index: usize, query_index: usize, opened_values: &Vec<Vec<Val>>,
                        proof: &GpuTapProof, roots: &Vec<[[u8; 4]; 8]>) -> Result<(), ()> {
        let vals: Vec<u32> = opened_values.iter().flatten().map(|v| v.as_canonical_u32()).collect();
        let n_evals = vals.len();
        let first = query_times_index * (1 + n_evals);
        let base = self.locks.offsets[first];
        let offs: Vec<u64> = self.locks.offsets[first..first + n_evals + 2].iter().map(|o| o - base).collect();
        let root: Vec<u8> = roots[query_times_index].iter().flatten().copied().collect();
        let path: Vec<u8> = proof.path.iter().flatten().copied().collect();
        let mut ok = 0;
        let rc = unsafe {
            ts_tap_mmcs_verify_batch(self.locks.bytes.as_ptr().add(base as usize), offs.as_ptr(), n_evals as u32, 1,
                                     query_index as u64, vals.as_ptr(), path.as_ptr(), proof.path.len() as u32,
                                     root.as_ptr(), &mut ok)
        };
        if rc == TS_OK && ok == 1 { Ok(()) } else { Err(()) }
    }
}
