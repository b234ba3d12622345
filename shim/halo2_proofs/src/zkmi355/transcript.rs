//! Thunks behind `zk_transcript_vtable`: the library calls back into the caller's transcript, so
//! `create_proof` stays generic over `T: TranscriptWrite<G1Affine, E>` exactly like upstream —
//! Blake2bWrite for the benches [REF circuit-benchmarks/src/super_circuit.rs:112], snark-verifier's
//! PoseidonTranscript for `gen_snark_shplonk` [REF prover/src/common/prover/utils.rs:31] and its
//! EvmTranscript for `gen_evm_proof_shplonk` [REF prover/src/common/prover/evm.rs:67] all work
//! unchanged, and the proof bytes end up in the caller's writer.
//! (For the three stock transcripts the library also has built-in implementations —
//! `zk_proof_set_transcript_kind` — which save the FFI round trips.)
use std::os::raw::{c_int, c_void};

use halo2curves::bn256::{Fr, G1Affine};

use super::ffi::zk_transcript_vtable;
use crate::transcript::{EncodedChallenge, TranscriptWrite};

/// What `user` points at for the duration of one proving session.
pub(super) struct Hook<'a, E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>> {
    pub transcript: &'a mut T,
    pub failed: bool,
    pub _marker: std::marker::PhantomData<E>,
}

// `G1Affine` is { x: Fq, y: Fq } in Montgomery limbs and the identity is (0, 0): the 64 bytes the
// library hands over ARE a `G1Affine` (SURVEY §8b data layout); same for the 32 bytes of an `Fr`.
unsafe fn point(p: *const c_void) -> G1Affine {
    std::ptr::read_unaligned(p as *const G1Affine)
}
unsafe fn scalar(p: *const c_void) -> Fr {
    std::ptr::read_unaligned(p as *const Fr)
}

unsafe extern "C" fn common_point<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>(user: *mut c_void, p: *const c_void) -> c_int {
    let h = &mut *(user as *mut Hook<E, T>);
    match h.transcript.common_point(point(p)) {
        Ok(()) => 0,
        Err(_) => { h.failed = true; 1 }
    }
}
unsafe extern "C" fn common_scalar<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>(user: *mut c_void, s: *const c_void) -> c_int {
    let h = &mut *(user as *mut Hook<E, T>);
    match h.transcript.common_scalar(scalar(s)) {
        Ok(()) => 0,
        Err(_) => { h.failed = true; 1 }
    }
}
unsafe extern "C" fn write_point<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>(user: *mut c_void, p: *const c_void) -> c_int {
    let h = &mut *(user as *mut Hook<E, T>);
    match h.transcript.write_point(point(p)) {
        Ok(()) => 0,
        Err(_) => { h.failed = true; 1 }
    }
}
unsafe extern "C" fn write_scalar<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>(user: *mut c_void, s: *const c_void) -> c_int {
    let h = &mut *(user as *mut Hook<E, T>);
    match h.transcript.write_scalar(scalar(s)) {
        Ok(()) => 0,
        Err(_) => { h.failed = true; 1 }
    }
}
unsafe extern "C" fn squeeze<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>(user: *mut c_void, out: *mut c_void) -> c_int {
    let h = &mut *(user as *mut Hook<E, T>);
    // halo2: `squeeze_challenge_scalar::<()>()` = squeeze_challenge().get_scalar()
    let c: Fr = h.transcript.squeeze_challenge().get_scalar();
    std::ptr::write_unaligned(out as *mut Fr, c);
    0
}

pub(super) fn vtable<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>() -> zk_transcript_vtable {
    zk_transcript_vtable {
        common_point: common_point::<E, T>,
        common_scalar: common_scalar::<E, T>,
        write_point: write_point::<E, T>,
        write_scalar: write_scalar::<E, T>,
        squeeze_challenge: squeeze::<E, T>,
    }
}
