//! RAII wrappers of the opaque handles.
use core::ffi::CStr;
use std::ptr;

use crate::ffi::*;

/// One per GPU, driven by one thread (SURVEY.md section 8(b) "Threading").
pub struct GpuContext {
    pub(crate) raw: *mut ts_ctx,
}
unsafe impl Send for GpuContext {}

impl GpuContext {
    pub fn new(device: i32) -> Self {
        assert_eq!(unsafe { ts_abi_version() }, 5, "libtapstark_hip ABI version");
        let mut raw = ptr::null_mut();
        let rc = unsafe { ts_ctx_create(device, &mut raw) };
        assert_eq!(rc, TS_OK, "ts_ctx_create({device}) failed: {}", last_error(ptr::null()));
        Self { raw }
    }

    /// The reference's prover path panics instead of returning `Result` (uni-stark/src/prover.rs:92,
    /// fri/src/two_adic_pcs.rs:234,254-256, fri/src/prover.rs:33-36,130-134): a non-zero status is
    /// turned back into a panic carrying the library's message.
    pub(crate) fn check(&self, rc: ts_status, what: &str) {
        if rc != TS_OK {
            panic!("{what}: status {rc}: {}", last_error(self.raw));
        }
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { ts_ctx_destroy(self.raw) }
    }
}

pub(crate) fn last_error(ctx: *const ts_ctx) -> String {
    let p = unsafe { ts_last_error(ctx) };
    if p.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

/// `RowMajorMatrix<Val>` resident in HBM; consumed by `commit` / `prove` like the moved argument.
pub struct DeviceMatrix<'c> {
    pub(crate) ctx: &'c GpuContext,
    pub(crate) raw: *mut ts_matrix,
}

impl<'c> DeviceMatrix<'c> {
    pub fn upload(ctx: &'c GpuContext, canonical_row_major: &[u32], height: usize, width: usize) -> Self {
        assert_eq!(canonical_row_major.len(), height * width);
        let mut raw = ptr::null_mut();
        ctx.check(
            unsafe { ts_matrix_upload(ctx.raw, canonical_row_major.as_ptr(), height as u64, width as u32, &mut raw) },
            "ts_matrix_upload",
        );
        Self { ctx, raw }
    }
    /// The PCIe-rate path: the copy is enqueued on the context's stream and returns at once; `src`
    /// must stay untouched until the context's next blocking call (e.g. the `ts_prove` that consumes
    /// the matrix), which the borrow of `src` for `'c` does not express -- see `prove_gpu_stream`.
    pub fn upload_async(ctx: &'c GpuContext, src: &PinnedTrace) -> Self {
        let mut raw = ptr::null_mut();
        ctx.check(
            unsafe { ts_matrix_upload_async(ctx.raw, src.ptr, src.height as u64, src.width as u32, &mut raw) },
            "ts_matrix_upload_async",
        );
        Self { ctx, raw }
    }
    pub(crate) fn into_raw(mut self) -> *mut ts_matrix {
        core::mem::replace(&mut self.raw, ptr::null_mut())
    }
}

impl Drop for DeviceMatrix<'_> {
    fn drop(&mut self) {
        if !self.raw.is_null() {
            unsafe { ts_matrix_free(self.ctx.raw, self.raw) }
        }
    }
}

/// A row-major trace in page-locked host memory (`ts_host_alloc`): what a trace generator should
/// write into if the trace is born on the host.  Pageable memory (`Vec<u32>`) uploads through a
/// staging copy at roughly half the rate.
pub struct PinnedTrace {
    pub(crate) ptr: *mut u32,
    pub height: usize,
    pub width: usize,
}
unsafe impl Send for PinnedTrace {}

impl PinnedTrace {
    pub fn new(height: usize, width: usize) -> Self {
        let mut p: *mut core::ffi::c_void = ptr::null_mut();
        let rc = unsafe { ts_host_alloc(height * width * 4, &mut p) };
        assert_eq!(rc, TS_OK, "ts_host_alloc({} bytes)", height * width * 4);
        Self { ptr: p as *mut u32, height, width }
    }
    /// canonical u32 values, row-major
    pub fn as_mut_slice(&mut self) -> &mut [u32] {
        unsafe { core::slice::from_raw_parts_mut(self.ptr, self.height * self.width) }
    }
}
impl Drop for PinnedTrace {
    fn drop(&mut self) {
        unsafe { ts_host_free(self.ptr as *mut core::ffi::c_void) }
    }
}

/// `BfChallenger<F, U32, Blake3Permutation, 16>` inside the library.  The Rust challenger of the
/// reference produces the same transcript; `GpuChallenger` exists so that `prove_gpu` can hand the
/// library a challenger in the caller's state and give the state back.
pub struct GpuChallenger {
    pub(crate) raw: *mut ts_challenger,
}

impl GpuChallenger {
    /// `BfChallenger::new(Blake3Permutation)`, challenges in EF4 (uni-stark/tests/fib_air.rs:108)
    pub fn new() -> Self {
        let mut raw = ptr::null_mut();
        assert_eq!(unsafe { ts_chal_new(0, 1, &mut raw) }, TS_OK);
        Self { raw }
    }
    pub fn observe_commitment(&mut self, root: &[u32; 8]) {
        unsafe { ts_chal_observe_commitment(self.raw, root.as_ptr()) }
    }
    pub fn sample_ext(&mut self) -> [u32; 4] {
        let mut out = [0u32; 4];
        unsafe { ts_chal_sample(self.raw, out.as_mut_ptr()) };
        out
    }
    /// 16 sponge words, input buffer (count + 8), output buffer (count + 8): 34 words
    pub fn state(&self) -> [u32; 34] {
        let mut out = [0u32; 34];
        unsafe { ts_chal_state(self.raw, out.as_mut_ptr()) };
        out
    }
}

impl Drop for GpuChallenger {
    fn drop(&mut self) {
        unsafe { ts_chal_free(self.raw) }
    }
}
