//! The reference's challenger traits on the library's challenger, so that `GpuChallenger` can be the
//! `Challenger` type parameter of `StarkConfig` / `Pcs` (uni-stark/src/config.rs:51-53:
//! `BfGrindingChallenger + CanObserve<Commitment> + CanSample<Challenge>`).  Same sponge as
//! `BfChallenger<_, U32, Blake3Permutation, 16>` (basic/src/challenger/mod.rs:67-348).
use basic::challenger::BfGrindingChallenger;
use p3_challenger::{CanObserve, CanSample, CanSampleBits};
use p3_field::{AbstractExtensionField, AbstractField};

use crate::context::GpuChallenger;
use crate::ffi::*;
use crate::proof::{Challenge, Commitment, Val};

unsafe impl Send for GpuChallenger {}
unsafe impl Sync for GpuChallenger {} // shared only by `&`: every mutation takes `&mut self`

impl Clone for GpuChallenger {
    fn clone(&self) -> Self {
        let mut raw = core::ptr::null_mut();
        assert_eq!(unsafe { ts_chal_clone(self.raw, &mut raw) }, TS_OK);
        Self { raw }
    }
}

/// basic/src/challenger/mod.rs:183-194: one `[u8; 4]` permutation-field element
impl CanObserve<[u8; 4]> for GpuChallenger {
    fn observe(&mut self, value: [u8; 4]) {
        unsafe { ts_chal_observe(self.raw, u32::from_le_bytes(value)) }
    }
}

/// mod.rs:211-223: a commitment is a Vec of roots, each 8 words
impl CanObserve<Commitment> for GpuChallenger {
    fn observe(&mut self, value: Commitment) {
        for root in value {
            let words: [u32; 8] = core::array::from_fn(|k| u32::from_le_bytes(root[k]));
            unsafe { ts_chal_observe_commitment(self.raw, words.as_ptr()) }
        }
    }
}

/// mod.rs:282-304
impl CanSample<Challenge> for GpuChallenger {
    fn sample(&mut self) -> Challenge {
        let c = self.sample_ext();
        Challenge::from_base_slice(&c.map(Val::from_canonical_u32))
    }
}

/// mod.rs:341-348
impl CanSampleBits<usize> for GpuChallenger {
    fn sample_bits(&mut self, bits: usize) -> usize {
        unsafe { ts_chal_sample_bits(self.raw, bits as u32) as usize }
    }
}

/// mod.rs:86-115
impl BfGrindingChallenger for GpuChallenger {
    type Witness = [u8; 4];

    fn grind(&mut self, bits: usize) -> Self::Witness {
        let mut w = 0u32;
        let rc = unsafe { ts_chal_grind(self.raw, bits as u32, &mut w) };
        assert_eq!(rc, TS_OK, "failed to find witness"); // mod.rs:100 .expect(...)
        w.to_le_bytes()
    }

    fn check_witness(&mut self, bits: usize, witness: Self::Witness) -> bool {
        unsafe { ts_chal_check_witness(self.raw, bits as u32, u32::from_le_bytes(witness)) != 0 }
    }
}
