//! `prove_gpu`: the reference's `prove` (uni-stark/src/prover.rs:25-119) with the same arguments,
//! in two forms: (1) one call, `ts_prove` -- the whole transcript runs inside the library;
//! (2) step by step through `impl Pcs for GpuFriPcs`, line for line the reference's body, for
//! callers that interleave their own logic.  Both give the same bytes.
use std::ptr;

use p3_air::Air;
use p3_challenger::{CanObserve, CanSample};
use p3_commit::PolynomialSpace;
use p3_field::{AbstractField, PrimeField32};
use p3_matrix::dense::RowMajorMatrix;
use p3_matrix::Matrix;
use uni_stark::SymbolicAirBuilder;

use basic::bf_pcs::Pcs;

use crate::air::serialize_constraints;
use crate::context::{DeviceMatrix, GpuChallenger, GpuContext, PinnedTrace};
use crate::ffi::*;
use crate::pcs::{FriConfig, GpuFriPcs};
use crate::proof::{Challenge, Commitments, OpenedValues, Proof, Val};

/// An AIR registered with the library (tape uploaded, quotient kernel specialised with hiprtc)
pub struct CompiledAir<'c> {
    ctx: &'c GpuContext,
    pub(crate) raw: *mut ts_air,
    pub log_quotient_degree: usize,
}
impl<'c> CompiledAir<'c> {
    pub fn new<A: Air<SymbolicAirBuilder<Val>>>(ctx: &'c GpuContext, air: &A, num_public_values: usize) -> Self {
        let tape = serialize_constraints::<Val, A>(air, num_public_values);
        let mut raw = ptr::null_mut();
        ctx.check(unsafe { ts_air_compile(ctx.raw, tape.as_ptr(), tape.len(), &mut raw) }, "ts_air_compile");
        let mut lqd = 0u32;
        unsafe { ts_air_info(raw, ptr::null_mut(), ptr::null_mut(), ptr::null_mut(), &mut lqd) };
        Self { ctx, raw, log_quotient_degree: lqd as usize }
    }
}
impl Drop for CompiledAir<'_> {
    fn drop(&mut self) {
        unsafe { ts_air_free(self.ctx.raw, self.raw) }
    }
}

fn proof_capacity(fri: &FriConfig, degree: usize, width: usize, qd: usize) -> usize {
    let log_n = degree.trailing_zeros() as usize + fri.log_blowup;
    let r = log_n - fri.log_blowup;
    64 + 8 * width + 16 * qd + 8 * r + fri.num_queries * (16 + width + 5 * qd + 16 * log_n + r * (9 + 8 * log_n))
}

/// Form 1.  `challenger` is left in the state the reference's `&mut Challenger` would be in.
pub fn prove_gpu<A>(pcs: &GpuFriPcs<'_>, air: &A, challenger: &mut GpuChallenger, trace: RowMajorMatrix<Val>,
                    public_values: &Vec<Val>) -> Proof
where
    A: Air<SymbolicAirBuilder<Val>>,
{
    let ctx = pcs.ctx;
    let cair = CompiledAir::new(ctx, air, public_values.len());
    let words: Vec<u32> = trace.values.iter().map(|v| v.as_canonical_u32()).collect();
    let pis: Vec<u32> = public_values.iter().map(|v| v.as_canonical_u32()).collect();
    let m = DeviceMatrix::upload(ctx, &words, trace.height(), trace.width());
    let cfg = pcs.fri.raw();
    let mut out = vec![0u32; proof_capacity(&pcs.fri, trace.height(), trace.width(), 1 << cair.log_quotient_degree)];
    let mut n = 0usize;
    let raw_m = m.into_raw();
    ctx.check(
        unsafe {
            ts_prove(ctx.raw, &cfg, cair.raw, challenger.raw, raw_m,
                     if pis.is_empty() { ptr::null() } else { pis.as_ptr() }, pis.len() as u32, out.as_mut_ptr(),
                     out.len(), &mut n)
        },
        "ts_prove",
    );
    unsafe { ts_matrix_free(ctx.raw, raw_m) };
    Proof::from_tspf(&out[..n])
}

/// A stream of proofs from HOST traces at the PCIe-inclusive rate (INTEGRATION.md "Three rates").
///
/// `prove_gpu` above uploads synchronously from pageable memory: the GPU idles during the copy and
/// the copy runs at about half the link rate.  Here `lanes` contexts (2 is enough to hide the
/// upload; 4 also hides the latency-bound phases of a proof, as bench.py does) each own a pinned
/// buffer: `fill(i, buf)` writes trace `i` into the lane's buffer (canonical u32, row-major), the
/// upload is enqueued asynchronously on the lane's stream and `ts_prove` follows on the same
/// stream, so the upload of one lane overlaps the proofs of the others.  Mirrors bench.py's
/// `h2d_inclusive` leg (6.1-6.4 ms per 2^20 x 64 proof = the 43 GB/s the link gives; 2.8 ms when
/// the trace is born on the device).  One host thread per lane, as the library requires.
pub fn prove_gpu_stream<A, F>(device: i32, fri: FriConfig, air: &A, num_public_values: usize, height: usize,
                              width: usize, n_proofs: usize, lanes: usize, fill: F,
                              public_values: &(dyn Fn(usize) -> Vec<Val> + Sync)) -> Vec<Proof>
where
    A: Air<SymbolicAirBuilder<Val>> + Sync,
    F: Fn(usize, &mut [u32]) + Sync,
{
    use std::sync::atomic::{AtomicUsize, Ordering};
    use std::sync::Mutex;

    let next = AtomicUsize::new(0);
    let proofs: Mutex<Vec<Option<Proof>>> = Mutex::new((0..n_proofs).map(|_| None).collect());
    std::thread::scope(|s| {
        for _ in 0..lanes.max(1) {
            s.spawn(|| {
                let ctx = GpuContext::new(device);
                let pcs = GpuFriPcs { ctx: &ctx, fri };
                let cair = CompiledAir::new(&ctx, air, num_public_values);
                let mut buf = PinnedTrace::new(height, width);
                let cfg = fri.raw();
                let mut out = vec![0u32; proof_capacity(&pcs.fri, height, width, 1 << cair.log_quotient_degree)];
                loop {
                    let i = next.fetch_add(1, Ordering::Relaxed);
                    if i >= n_proofs {
                        break;
                    }
                    // the previous ts_prove on this context has returned (it blocks until the proof is on
                    // the host), so the lane's buffer is free to overwrite
                    fill(i, buf.as_mut_slice());
                    let pis: Vec<u32> = public_values(i).iter().map(|v| v.as_canonical_u32()).collect();
                    let raw_m = DeviceMatrix::upload_async(&ctx, &buf).into_raw();
                    let mut challenger = GpuChallenger::new();
                    let mut n = 0usize;
                    ctx.check(
                        unsafe {
                            ts_prove(ctx.raw, &cfg, cair.raw, challenger.raw, raw_m,
                                     if pis.is_empty() { ptr::null() } else { pis.as_ptr() }, pis.len() as u32,
                                     out.as_mut_ptr(), out.len(), &mut n)
                        },
                        "ts_prove",
                    );
                    unsafe { ts_matrix_free(ctx.raw, raw_m) };
                    proofs.lock().unwrap()[i] = Some(Proof::from_tspf(&out[..n]));
                    let _ = &mut challenger;
                }
            });
        }
    });
    proofs.into_inner().unwrap().into_iter().map(|p| p.expect("every proof index was taken")).collect()
}

/// Form 2: the body of uni-stark/src/prover.rs:40-118 over the `Pcs` trait.
pub fn prove_gpu_stepwise<A>(pcs: &GpuFriPcs<'_>, air: &A, challenger: &mut GpuChallenger,
                             trace: RowMajorMatrix<Val>, public_values: &Vec<Val>) -> Proof
where
    A: Air<SymbolicAirBuilder<Val>>,
{
    let degree = trace.height(); // :43
    let log_degree = degree.trailing_zeros() as usize;
    let cair = CompiledAir::new(pcs.ctx, air, public_values.len());
    let log_quotient_degree = cair.log_quotient_degree; // :46
    let quotient_degree = 1 << log_quotient_degree;
    let pis: Vec<u32> = public_values.iter().map(|v| v.as_canonical_u32()).collect();

    let trace_domain = pcs.natural_domain_for_degree(degree); // :50
    let (trace_commit, trace_data) = pcs.commit(vec![(trace_domain, trace)]); // :52-53
    challenger.observe(trace_commit.clone()); // :60
    let alpha: Challenge = challenger.sample(); // :63

    // :65-80 quotient domain, evaluations on it, quotient_values, flatten_to_base, split_evals:
    // one fused device call; chunk c lives on the domain {log_n, 31 * w_{n qd}^c} (split_domains)
    let chunks = pcs.quotient_chunks(&trace_data, cair.raw, quotient_degree, &pis, &alpha);
    let quotient_domain = trace_domain.create_disjoint_domain(1 << (log_degree + log_quotient_degree));
    let qc_domains = quotient_domain.split_domains(quotient_degree);
    // :82-83 commit to the chunks (already resident: no upload)
    let cfg = pcs.fri.raw();
    let shifts: Vec<u32> = qc_domains.iter().map(|d| d.shift.as_canonical_u32()).collect();
    let raws: Vec<*mut ts_matrix> = chunks.into_iter().map(DeviceMatrix::into_raw).collect();
    let (mut root, mut qraw) = ([0u32; 8], ptr::null_mut());
    pcs.ctx.check(
        unsafe {
            ts_pcs_commit(pcs.ctx.raw, &cfg, raws.len() as u32, raws.as_ptr(), shifts.as_ptr(), root.as_mut_ptr(),
                          &mut qraw)
        },
        "ts_pcs_commit (quotient chunks)",
    );
    for m in raws {
        unsafe { ts_matrix_free(pcs.ctx.raw, m) };
    }
    let quotient_commit = vec![root.map(u32::to_le_bytes)];
    let quotient_data = crate::pcs::GpuProverData {
        ctx: pcs.ctx,
        raw: qraw,
        dims: vec![(degree << pcs.fri.log_blowup, 4); quotient_degree],
    };
    challenger.observe(quotient_commit.clone()); // :84

    let zeta: Challenge = challenger.sample(); // :91
    let zeta_next = trace_domain.next_point(zeta).unwrap(); // :92
    let (opened_values, opening_proof) = pcs.open(
        vec![
            (&trace_data, vec![vec![zeta, zeta_next]]),
            (&quotient_data, (0..quotient_degree).map(|_| vec![zeta]).collect()),
        ],
        challenger,
    ); // :94-104
    let trace_local = opened_values[0][0][0].clone(); // :105-110
    let trace_next = opened_values[0][0][1].clone();
    let quotient_chunks = opened_values[1].iter().map(|v| v[0].clone()).collect();
    Proof {
        commitments: Commitments { trace: trace_commit, quotient_chunks: quotient_commit },
        opened_values: OpenedValues { trace_local, trace_next, quotient_chunks },
        opening_proof,
        degree_bits: log_degree,
    }
}
