//! `impl Pcs<Challenge, GpuChallenger> for GpuFriPcs` -- the reference's PCS trait
//! (basic/src/bf_pcs.rs:19-88) over the library: `ProverData` is a device handle, the LDE never
//! leaves HBM.  Mirrors `TwoAdicFriPcs` (fri/src/two_adic_pcs.rs:203-535).
//!
//! `StarkGenericConfig::Pcs` additionally demands `PcsExpr` (uni-stark/src/config.rs:35-41: the
//! Bitcoin-script form of the verifier, out of scope here), so `GpuFriPcs` cannot be the `Pcs` of
//! the reference's `StarkConfig` as it stands; `prove_gpu` / `prove_gpu_stepwise` (prove.rs) take
//! it directly and return the same `Proof` fields.
use std::ptr;

use basic::bf_pcs::{OpenedValues as PcsOpenedValues, Pcs};
use p3_commit::{PolynomialSpace, TwoAdicMultiplicativeCoset};
use p3_field::{AbstractExtensionField, AbstractField, PrimeField32};
use p3_matrix::bitrev::BitReversableMatrix;
use p3_matrix::dense::RowMajorMatrix;
use p3_matrix::Matrix;

use crate::context::{DeviceMatrix, GpuChallenger, GpuContext};
use crate::ffi::*;
use crate::proof::{Challenge, Commitment, FriProof, Val, Words};

/// fri/src/config.rs:11-16 (the MMCS is the library's Blake3 Merkle tree: no `mmcs` field)
#[derive(Clone, Copy, Debug)]
pub struct FriConfig {
    pub log_blowup: usize,
    pub num_queries: usize,
    pub proof_of_work_bits: usize,
}
impl FriConfig {
    pub(crate) fn raw(&self) -> ts_fri_config {
        ts_fri_config {
            log_blowup: self.log_blowup as u32,
            num_queries: self.num_queries as u32,
            proof_of_work_bits: self.proof_of_work_bits as u32,
        }
    }
}

pub struct GpuFriPcs<'c> {
    pub ctx: &'c GpuContext,
    pub fri: FriConfig,
}

/// `Pcs::ProverData`: committed LDEs + Merkle tree, resident in HBM
pub struct GpuProverData<'c> {
    pub(crate) ctx: &'c GpuContext,
    pub(crate) raw: *mut ts_pcs_data,
    pub(crate) dims: Vec<(usize, usize)>, // (LDE height, width) per committed matrix
}
impl Drop for GpuProverData<'_> {
    fn drop(&mut self) {
        unsafe { ts_pcs_data_free(self.ctx.raw, self.raw) }
    }
}

#[derive(Debug)]
pub enum GpuPcsError {
    /// the verdict codes of `ts_verify` / `ts_pcs_verify` (include/tapstark.h): 1 InvalidProofShape,
    /// 2 FRI shape, 3 InvalidPowWitness, 4 input MMCS, 5 commit-phase MMCS, 6 FinalPolyMismatch, ...
    Rejected(i32),
}

fn ef_words(e: &Challenge) -> [u32; 4] {
    let c: &[Val] = e.as_base_slice();
    [c[0].as_canonical_u32(), c[1].as_canonical_u32(), c[2].as_canonical_u32(), c[3].as_canonical_u32()]
}

impl<'c> GpuFriPcs<'c> {
    /// `get_evaluations_on_domain` + `quotient_values` + `flatten_to_base` + `split_evals`
    /// (two_adic_pcs.rs:247-258; uni-stark/src/prover.rs:68-80,122-194) in one device call: the
    /// trait's `get_evaluations_on_domain` has to return a HOST matrix, which would drag the LDE over
    /// PCIe, so the fused step is what `prove_gpu` uses.
    pub fn quotient_chunks(&self, trace_data: &GpuProverData<'c>, air: *const ts_air, qd: usize,
                           public_values: &[u32], alpha: &Challenge) -> Vec<DeviceMatrix<'c>> {
        let mut out = vec![ptr::null_mut(); qd];
        let a = ef_words(alpha);
        self.ctx.check(
            unsafe {
                ts_quotient_chunks(self.ctx.raw, trace_data.raw, self.fri.log_blowup as u32, air,
                                   if public_values.is_empty() { ptr::null() } else { public_values.as_ptr() },
                                   public_values.len() as u32, a.as_ptr(), out.as_mut_ptr())
            },
            "ts_quotient_chunks",
        );
        out.into_iter().map(|raw| DeviceMatrix { ctx: self.ctx, raw }).collect()
    }
}

impl<'c> Pcs<Challenge, GpuChallenger> for GpuFriPcs<'c> {
    type Domain = TwoAdicMultiplicativeCoset<Val>;
    type Commitment = Commitment;
    type ProverData = GpuProverData<'c>;
    type Proof = FriProof;
    type Error = GpuPcsError;

    /// two_adic_pcs.rs:220-226
    fn natural_domain_for_degree(&self, degree: usize) -> Self::Domain {
        assert!(degree.is_power_of_two());
        TwoAdicMultiplicativeCoset { log_n: degree.trailing_zeros() as usize, shift: Val::one() }
    }

    /// two_adic_pcs.rs:227-245
    fn commit(&self, evaluations: Vec<(Self::Domain, RowMajorMatrix<Val>)>) -> (Self::Commitment, Self::ProverData) {
        let mut mats = Vec::with_capacity(evaluations.len());
        let mut shifts = Vec::with_capacity(evaluations.len());
        for (domain, evals) in evaluations {
            assert_eq!(domain.size(), evals.height()); // :234
            let words: Vec<u32> = evals.values.iter().map(|v| v.as_canonical_u32()).collect();
            mats.push(DeviceMatrix::upload(self.ctx, &words, evals.height(), evals.width()).into_raw());
            shifts.push(domain.shift.as_canonical_u32());
        }
        let cfg = self.fri.raw();
        let mut root = [0u32; 8];
        let mut raw = ptr::null_mut();
        self.ctx.check(
            unsafe {
                ts_pcs_commit(self.ctx.raw, &cfg, mats.len() as u32, mats.as_ptr(), shifts.as_ptr(),
                              root.as_mut_ptr(), &mut raw)
            },
            "ts_pcs_commit",
        );
        for m in mats {
            unsafe { ts_matrix_free(self.ctx.raw, m) } // consumed: frees the (now empty) handle
        }
        let mut n = 0u32;
        unsafe { ts_pcs_data_info(raw, &mut n, ptr::null_mut()) };
        let dims = (0..n)
            .map(|i| {
                let (mut h, mut w) = (0u64, 0u32);
                unsafe { ts_pcs_data_matrix_info(raw, i, &mut h, &mut w) };
                (h as usize, w as usize)
            })
            .collect();
        (vec![root.map(u32::to_le_bytes)], GpuProverData { ctx: self.ctx, raw, dims })
    }

    /// two_adic_pcs.rs:247-258 -- the slow, host-side form the trait demands (used by callers that
    /// really want the evaluations; `prove_gpu` uses `quotient_chunks` instead)
    fn get_evaluations_on_domain<'a>(&self, prover_data: &'a Self::ProverData, idx: usize,
                                     domain: Self::Domain) -> impl Matrix<Val> + 'a {
        assert_eq!(domain.shift, Val::generator()); // :254
        let (height, width) = prover_data.dims[idx];
        assert!(height >= domain.size()); // :256
        let mut words = vec![0u32; height * width];
        self.ctx.check(
            unsafe { ts_pcs_data_lde(self.ctx.raw, prover_data.raw, idx as u32, words.as_mut_ptr()) },
            "ts_pcs_data_lde",
        );
        words.truncate(domain.size() * width); // split_rows(domain.size()).0
        let vals: Vec<Val> = words.into_iter().map(Val::from_canonical_u32).collect();
        RowMajorMatrix::new(vals, width).bit_reverse_rows()
    }

    /// two_adic_pcs.rs:260-419
    fn open(&self, rounds: Vec<(&Self::ProverData, Vec<Vec<Challenge>>)>,
            challenger: &mut GpuChallenger) -> (PcsOpenedValues<Challenge>, Self::Proof) {
        let cfg = self.fri.raw();
        let datas: Vec<*const ts_pcs_data> = rounds.iter().map(|(d, _)| d.raw as *const _).collect();
        let (mut n_points, mut points, mut n_opened, mut log_max) = (Vec::new(), Vec::new(), 0usize, 0usize);
        for (data, per_mat) in &rounds {
            assert_eq!(per_mat.len(), data.dims.len());
            for (m, pts) in per_mat.iter().enumerate() {
                n_points.push(pts.len() as u32);
                for p in pts {
                    points.extend_from_slice(&ef_words(p));
                }
                n_opened += 4 * pts.len() * data.dims[m].1;
                log_max = log_max.max(data.dims[m].0.trailing_zeros() as usize);
            }
        }
        let mut opened = vec![0u32; n_opened.max(1)];
        // FriProof: R commits, Q queries x (input openings with paths + R commit-phase openings)
        let total_w: usize = rounds.iter().map(|(d, _)| d.dims.iter().map(|x| x.1 + 2).sum::<usize>() + 2).sum();
        let r = log_max - self.fri.log_blowup;
        let cap = 16 + 8 * r + self.fri.num_queries * (2 + total_w + rounds.len() * 8 * log_max + r * (9 + 8 * log_max));
        let mut proof = vec![0u32; cap];
        let (mut n_o, mut n_p) = (0usize, 0usize);
        self.ctx.check(
            unsafe {
                ts_pcs_open(self.ctx.raw, &cfg, challenger.raw, rounds.len() as u32, datas.as_ptr(),
                            n_points.as_ptr(), points.as_ptr(), opened.as_mut_ptr(), opened.len(), &mut n_o,
                            proof.as_mut_ptr(), proof.len(), &mut n_p)
            },
            "ts_pcs_open",
        );
        // opened values come back in (round, matrix, point, column) order
        let mut rd = Words::new(&opened[..n_o]);
        let values = rounds
            .iter()
            .map(|(data, per_mat)| {
                per_mat
                    .iter()
                    .enumerate()
                    .map(|(m, pts)| pts.iter().map(|_| (0..data.dims[m].1).map(|_| rd.ef()).collect()).collect())
                    .collect()
            })
            .collect();
        (values, Words::new(&proof[..n_p]).fri_proof())
    }

    /// two_adic_pcs.rs:421-534 (host only)
    fn verify(&self, rounds: Vec<(Self::Commitment, Vec<(Self::Domain, Vec<(Challenge, Vec<Challenge>)>)>)>,
              proof: &Self::Proof, challenger: &mut GpuChallenger) -> Result<(), Self::Error> {
        let cfg = self.fri.raw();
        let (mut commits, mut per_round, mut logs, mut widths, mut n_points) =
            (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
        let (mut points, mut opened) = (Vec::new(), Vec::new());
        for (com, mats) in &rounds {
            assert_eq!(com.len(), 1, "one root per commitment");
            commits.extend(com[0].iter().map(|b| u32::from_le_bytes(*b)));
            per_round.push(mats.len() as u32);
            for (domain, pts) in mats {
                logs.push(domain.log_n as u32);
                widths.push(pts.first().map_or(0, |p| p.1.len()) as u32);
                n_points.push(pts.len() as u32);
                for (z, ys) in pts {
                    points.extend_from_slice(&ef_words(z));
                    for y in ys {
                        opened.extend_from_slice(&ef_words(y));
                    }
                }
            }
        }
        let words = proof.to_tspf_words();
        let mut verdict = -1;
        let rc = unsafe {
            ts_pcs_verify(&cfg, challenger.raw, rounds.len() as u32, commits.as_ptr(), per_round.as_ptr(),
                          logs.as_ptr(), widths.as_ptr(), n_points.as_ptr(), points.as_ptr(), opened.as_ptr(),
                          words.as_ptr(), words.len(), &mut verdict)
        };
        assert_eq!(rc, TS_OK, "ts_pcs_verify: bad arguments");
        if verdict == 0 { Ok(()) } else { Err(GpuPcsError::Rejected(verdict)) }
    }
}

impl FriProof {
    /// back to the TSPF v1 FriProof words (the inverse of `Words::fri_proof`)
    pub fn to_tspf_words(&self) -> Vec<u32> {
        let mut w = Vec::new();
        let digest = |w: &mut Vec<u32>, d: &[[u8; 4]; 8]| w.extend(d.iter().map(|b| u32::from_le_bytes(*b)));
        let path = |w: &mut Vec<u32>, p: &Vec<[u8; 32]>| {
            w.push(p.len() as u32);
            for node in p {
                w.extend(node.chunks(4).map(|c| u32::from_le_bytes([c[0], c[1], c[2], c[3]])));
            }
        };
        w.push(self.commit_phase_commits.len() as u32);
        for c in &self.commit_phase_commits {
            digest(&mut w, &c[0]);
        }
        w.push(self.query_proofs.len() as u32);
        for q in &self.query_proofs {
            w.push(q.input_proof.len() as u32);
            for b in &q.input_proof {
                w.push(b.opened_values.len() as u32);
                for row in &b.opened_values {
                    w.push(row.len() as u32);
                    w.extend(row.iter().map(|v| v.as_canonical_u32()));
                }
                path(&mut w, &b.opening_proof);
            }
            for (vals, p) in &q.commit_phase_openings {
                for e in &vals[0] {
                    w.extend_from_slice(&ef_words(e));
                }
                path(&mut w, p);
            }
        }
        w.extend_from_slice(&ef_words(&self.final_poly));
        w.push(u32::from_le_bytes(self.pow_witness));
        w
    }
}
