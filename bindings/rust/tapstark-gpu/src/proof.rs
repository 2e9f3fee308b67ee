//! The proof as Rust values: TSPF v1 / v2 words -> structs with the reference's field
//! names and field order (uni-stark/src/proof.rs:17-37, fri/src/proof.rs:13-33,
//! fri/src/two_adic_pcs.rs:63-68), deriving serde so that `postcard::to_allocvec(&proof)` yields the
//! bytes `ts_proof_to_postcard` yields.
//!
//! Two types are this build's, because the reference's are tied to its taptree MMCS
//! (`CommitedProof<BO, B>`, `Vec<TreeRoot>` of num_queries roots): `Commitment` = a Vec holding ONE
//! root, `MmcsProof` = the sibling path.  `Proof<SC>` of the reference cannot be instantiated for a
//! Merkle MMCS without editing the `CommitedProof` bounds.
use p3_baby_bear::BabyBear;
use p3_field::extension::BinomialExtensionField;
use p3_field::{AbstractExtensionField, AbstractField};
use serde::{Deserialize, Serialize};

pub type Val = BabyBear;
pub type Challenge = BinomialExtensionField<BabyBear, 4>;
pub type Commitment = Vec<[[u8; 4]; 8]>; // like Vec<TreeRoot> (basic/src/mmcs/taptree_mmcs.rs:16,43)
pub type MmcsProof = Vec<[u8; 32]>;

#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct Commitments {
    pub trace: Commitment,
    pub quotient_chunks: Commitment,
}
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct OpenedValues {
    pub trace_local: Vec<Challenge>,
    pub trace_next: Vec<Challenge>,
    pub quotient_chunks: Vec<Vec<Challenge>>,
}
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct BatchOpening {
    pub opened_values: Vec<Vec<Val>>,
    pub opening_proof: MmcsProof,
}
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct BfQueryProof {
    pub input_proof: Vec<BatchOpening>,
    pub commit_phase_openings: Vec<(Vec<Vec<Challenge>>, MmcsProof)>,
}
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct FriProof {
    pub commit_phase_commits: Vec<Commitment>,
    pub query_proofs: Vec<BfQueryProof>,
    pub final_poly: Challenge,
    pub pow_witness: [u8; 4], // `type Witness = PF`, PF = [u8; 4] (basic/src/challenger/mod.rs:91)
}
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct Proof {
    pub commitments: Commitments,
    pub opened_values: OpenedValues,
    pub opening_proof: FriProof,
    pub degree_bits: usize,
}

pub(crate) struct Words<'a> {
    w: &'a [u32],
    pos: usize,
    n_roots: usize, // roots per commitment: 1 (TSPF v1, Blake3 Merkle MMCS), num_queries (v2, taptrees)
}
impl<'a> Words<'a> {
    pub(crate) fn new(w: &'a [u32]) -> Self {
        Self { w, pos: 0, n_roots: 1 }
    }
    fn commitment(&mut self) -> Commitment {
        (0..self.n_roots).map(|_| self.digest()).collect()
    }
    pub(crate) fn get(&mut self) -> u32 {
        let v = self.w[self.pos];
        self.pos += 1;
        v
    }
    pub(crate) fn take(&mut self, n: usize) -> &'a [u32] {
        let s = &self.w[self.pos..self.pos + n];
        self.pos += n;
        s
    }
    fn digest(&mut self) -> [[u8; 4]; 8] {
        let mut d = [[0u8; 4]; 8];
        for (k, w) in self.take(8).iter().enumerate() {
            d[k] = w.to_le_bytes();
        }
        d
    }
    fn path(&mut self) -> MmcsProof {
        let n = self.get() as usize;
        (0..n)
            .map(|_| {
                let mut b = [0u8; 32];
                for (k, w) in self.take(8).iter().enumerate() {
                    b[4 * k..4 * k + 4].copy_from_slice(&w.to_le_bytes());
                }
                b
            })
            .collect()
    }
    pub(crate) fn ef(&mut self) -> Challenge {
        let c = self.take(4);
        Challenge::from_base_slice(&[
            Val::from_canonical_u32(c[0]),
            Val::from_canonical_u32(c[1]),
            Val::from_canonical_u32(c[2]),
            Val::from_canonical_u32(c[3]),
        ])
    }
    pub(crate) fn fri_proof(&mut self) -> FriProof {
        let r = self.get() as usize;
        let commit_phase_commits = (0..r).map(|_| self.commitment()).collect();
        let q = self.get() as usize;
        let mut query_proofs = Vec::with_capacity(q);
        for _ in 0..q {
            let n_batches = self.get() as usize;
            let mut input_proof = Vec::with_capacity(n_batches);
            for _ in 0..n_batches {
                let n_mats = self.get() as usize;
                let opened_values = (0..n_mats)
                    .map(|_| {
                        let w = self.get() as usize;
                        self.take(w).iter().map(|&v| Val::from_canonical_u32(v)).collect()
                    })
                    .collect();
                input_proof.push(BatchOpening { opened_values, opening_proof: self.path() });
            }
            let commit_phase_openings = (0..r)
                .map(|_| {
                    let (a, b) = (self.ef(), self.ef());
                    (vec![vec![a, b]], self.path())
                })
                .collect();
            query_proofs.push(BfQueryProof { input_proof, commit_phase_openings });
        }
        let final_poly = self.ef();
        let pow_witness = self.get().to_le_bytes();
        FriProof { commit_phase_commits, query_proofs, final_poly, pow_witness }
    }
}

impl Proof {
    /// TSPF v1: `[magic, 1, degree_bits, width, quotient_degree]`, commitments, opened values, FriProof.
    /// TSPF v2 (proofs over taptrees, `ts_prove_tap`): a sixth header word `num_queries`, and every
    /// commitment is that many roots -- `Vec<TreeRoot>` exactly as the reference's `Commitment`.
    pub fn from_tspf(words: &[u32]) -> Self {
        let mut r = Words::new(words);
        assert_eq!(r.get(), 0x4650_5354, "TSPF magic");
        let version = r.get();
        assert!(version == 1 || version == 2, "TSPF version");
        let degree_bits = r.get() as usize;
        let width = r.get() as usize;
        let qd = r.get() as usize;
        if version == 2 {
            r.n_roots = r.get() as usize;
        }
        let commitments = Commitments { trace: r.commitment(), quotient_chunks: r.commitment() };
        let trace_local = (0..width).map(|_| r.ef()).collect();
        let trace_next = (0..width).map(|_| r.ef()).collect();
        let quotient_chunks = (0..qd).map(|_| (0..4).map(|_| r.ef()).collect()).collect();
        let opening_proof = r.fri_proof();
        assert_eq!(r.pos, words.len(), "trailing words in the proof");
        Proof {
            commitments,
            opened_values: OpenedValues { trace_local, trace_next, quotient_chunks },
            opening_proof,
            degree_bits,
        }
    }
}
