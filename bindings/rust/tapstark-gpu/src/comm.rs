//! One proof over the GPUs of a node (`ts_prove_sharded`): the collectives are the library's own
//! RCCL communicator (csrc/comm.cpp).  One process (or thread) per GPU; rank 0 makes the 128-byte
//! unique id and hands it to its peers by any channel it has.
use std::ptr;

use p3_air::Air;
use p3_field::PrimeField32;
use p3_matrix::dense::RowMajorMatrix;
use p3_matrix::Matrix;
use uni_stark::SymbolicAirBuilder;

use crate::context::{DeviceMatrix, GpuChallenger, GpuContext};
use crate::ffi::*;
use crate::pcs::GpuFriPcs;
use crate::proof::{Proof, Val};
use crate::prove::CompiledAir;

pub fn rccl_unique_id() -> [u8; 128] {
    let mut id = [0u8; 128];
    assert_eq!(unsafe { ts_rccl_unique_id(id.as_mut_ptr()) }, TS_OK, "librccl not available");
    id
}

pub struct RcclComm {
    pub(crate) comm: ts_comm,
    handle: *mut ts_rccl_comm,
}
impl RcclComm {
    pub fn new(ctx: &GpuContext, unique_id: &[u8; 128], rank: i32, world: i32) -> Self {
        let mut comm: ts_comm = unsafe { core::mem::zeroed() };
        let mut handle = ptr::null_mut();
        ctx.check(
            unsafe { ts_comm_rccl_create(ctx.raw, unique_id.as_ptr(), rank, world, &mut comm, &mut handle) },
            "ts_comm_rccl_create",
        );
        Self { comm, handle }
    }
}
impl Drop for RcclComm {
    fn drop(&mut self) {
        unsafe { ts_comm_rccl_destroy(self.handle) }
    }
}

/// Rank g passes natural rows [g n/G, (g+1) n/G) of the trace; every rank gets the whole proof,
/// bit-identical to `prove_gpu` on the whole trace.  G must be a power of two <= 2^log_blowup.
pub fn prove_gpu_sharded<A>(pcs: &GpuFriPcs<'_>, comm: &RcclComm, air: &A, challenger: &mut GpuChallenger,
                            trace_rows: RowMajorMatrix<Val>, public_values: &Vec<Val>) -> Proof
where
    A: Air<SymbolicAirBuilder<Val>>,
{
    let ctx = pcs.ctx;
    let cair = CompiledAir::new(ctx, air, public_values.len());
    let words: Vec<u32> = trace_rows.values.iter().map(|v| v.as_canonical_u32()).collect();
    let pis: Vec<u32> = public_values.iter().map(|v| v.as_canonical_u32()).collect();
    let m = DeviceMatrix::upload(ctx, &words, trace_rows.height(), trace_rows.width()).into_raw();
    let cfg = pcs.fri.raw();
    let degree = trace_rows.height() * comm.comm.world as usize;
    let log_n = degree.trailing_zeros() as usize + pcs.fri.log_blowup;
    let (w, qd, r) = (trace_rows.width(), 1usize << cair.log_quotient_degree, log_n - pcs.fri.log_blowup);
    let cap = 64 + 8 * w + 16 * qd + 8 * r + pcs.fri.num_queries * (16 + w + 5 * qd + 16 * log_n + r * (9 + 8 * log_n));
    let mut out = vec![0u32; cap];
    let mut n = 0usize;
    ctx.check(
        unsafe {
            ts_prove_sharded(ctx.raw, &cfg, &comm.comm, cair.raw, challenger.raw, m,
                             if pis.is_empty() { ptr::null() } else { pis.as_ptr() }, pis.len() as u32, ptr::null(),
                             out.as_mut_ptr(), out.len(), &mut n)
        },
        "ts_prove_sharded",
    );
    unsafe { ts_matrix_free(ctx.raw, m) };
    Proof::from_tspf(&out[..n])
}
