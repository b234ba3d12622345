//! `extern "C"` transcription of `include/zkmi355.h` — the subset the shim calls.
//! Kept in sync by `tests/test_abi.py::test_rust_ffi_block_matches_the_header` (every function
//! declared here must exist in the header with the same number of parameters).
//!
//! Layout contract: `Fr` / `Fq` cross the boundary as their in-memory form in
//! halo2curves — 4 × u64 little-endian limbs, Montgomery — so `&[Fr]` is passed as `*const c_void`
//! without conversion; `G1Affine` is `{x, y}` = 64 bytes, identity = all zero.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct zk_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zk_srs {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zk_pk {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zk_proof {
    _private: [u8; 0],
}

pub const ZK_OK: c_int = 0;
pub const ZK_ERR_INVALID_ARG: c_int = -1;
pub const ZK_ERR_HIP: c_int = -2;
pub const ZK_ERR_OOM: c_int = -3;
pub const ZK_ERR_NO_DEVICE: c_int = -4;
pub const ZK_ERR_UNSUPPORTED: c_int = -5;

pub const ZK_MULTIOPEN_GWC: c_int = 0;
pub const ZK_MULTIOPEN_SHPLONK: c_int = 1;

pub const ZK_TRANSCRIPT_BLAKE2B: c_int = 0;
pub const ZK_TRANSCRIPT_POSEIDON: c_int = 1;
pub const ZK_TRANSCRIPT_EVM: c_int = 2;

/// `zk_transcript_vtable`: every transcript operation of a proving session is forwarded to the
/// caller's `T: TranscriptWrite<G1Affine, _>` (see `transcript.rs`).
#[repr(C)]
pub struct zk_transcript_vtable {
    pub common_point: unsafe extern "C" fn(user: *mut c_void, affine64: *const c_void) -> c_int,
    pub common_scalar: unsafe extern "C" fn(user: *mut c_void, fr32: *const c_void) -> c_int,
    pub write_point: unsafe extern "C" fn(user: *mut c_void, affine64: *const c_void) -> c_int,
    pub write_scalar: unsafe extern "C" fn(user: *mut c_void, fr32: *const c_void) -> c_int,
    pub squeeze_challenge: unsafe extern "C" fn(user: *mut c_void, fr32_out: *mut c_void) -> c_int,
}

/// `zk_mock_failure`: one record of `zk_mock_verify` (kind 1 gate polynomial / 2 lookup / 3 permutation; index; sub; row)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct zk_mock_failure {
    pub kind: u32,
    pub index: u32,
    pub sub: u32,
    pub row: u32,
}
pub const ZK_MOCK_GATE: u32 = 1;
pub const ZK_MOCK_LOOKUP: u32 = 2;
pub const ZK_MOCK_PERMUTATION: u32 = 3;

pub type zk_allgather_fn = unsafe extern "C" fn(user: *mut c_void, send: *const c_void, bytes: usize, recv: *mut c_void) -> c_int;

extern "C" {
    // ---- context
    pub fn zk_ctx_create(device: c_int, out: *mut *mut zk_ctx) -> c_int;
    pub fn zk_ctx_destroy(ctx: *mut zk_ctx);
    /// joins every stream of the library (main, copy, auxiliary, MSM side streams)
    pub fn zk_ctx_sync(ctx: *mut zk_ctx) -> c_int;
    pub fn zk_ctx_streams_busy(ctx: *mut zk_ctx, busy: *mut c_int) -> c_int;
    pub fn zk_last_error(ctx: *const zk_ctx) -> *const c_char;

    // ---- ParamsKZG
    pub fn zk_srs_create(ctx: *mut zk_ctx, k: u32, h_g: *const c_void, h_g_lagrange: *const c_void, out: *mut *mut zk_srs) -> c_int;
    pub fn zk_srs_destroy(ctx: *mut zk_ctx, srs: *mut zk_srs);
    pub fn zk_commit(ctx: *mut zk_ctx, srs: *const zk_srs, basis: c_int, d_scalars: *const c_void, n: usize, h_out_affine: *mut c_void) -> c_int;
    pub fn zk_msm_g1_host(ctx: *mut zk_ctx, h_scalars: *const c_void, h_bases: *const c_void, n: usize, h_out_affine: *mut c_void) -> c_int;

    // ---- keygen_pk / create_proof
    pub fn zk_pk_create(ctx: *mut zk_ctx, srs: *const zk_srs, h_blob: *const c_void, blob_len: usize, out: *mut *mut zk_pk) -> c_int;
    pub fn zk_pk_destroy(ctx: *mut zk_ctx, pk: *mut zk_pk);
    pub fn zk_pk_vk(ctx: *mut zk_ctx, pk: *const zk_pk, h_commitments: *mut c_void, h_vk_repr: *mut c_void) -> c_int;
    pub fn zk_pk_set_transcript_repr(ctx: *mut zk_ctx, pk: *mut zk_pk, h_repr_fr32: *const c_void) -> c_int;
    pub fn zk_pk_shape(ctx: *mut zk_ctx, pk: *const zk_pk, out16: *mut u32) -> c_int;

    /// dev::MockProver::verify_par / verify_at_rows_par on the device (NULL row lists = every usable row)
    pub fn zk_mock_verify(ctx: *mut zk_ctx, pk: *const zk_pk, h_advice: *const *const c_void, h_instance: *const *const c_void, h_challenges: *const c_void,
                          gate_rows: *const u32, num_gate_rows: usize, lookup_rows: *const u32, num_lookup_rows: usize,
                          out: *mut zk_mock_failure, cap: usize, count: *mut usize) -> c_int;
    pub fn zk_proof_mock_verify(ctx: *mut zk_ctx, proof: *mut zk_proof, gate_rows: *const u32, num_gate_rows: usize, lookup_rows: *const u32, num_lookup_rows: usize,
                                out: *mut zk_mock_failure, cap: usize, count: *mut usize) -> c_int;
    pub fn zk_host_mock_challenges(count: u32, out_fr32: *mut c_void) -> c_int;

    pub fn zk_proof_begin_instances(ctx: *mut zk_ctx, pk: *const zk_pk, h_instance: *const *const c_void, h_instance_len: *const u32, seed16: *const u8, out: *mut *mut zk_proof) -> c_int;
    pub fn zk_proof_set_multiopen(ctx: *mut zk_ctx, proof: *mut zk_proof, kind: c_int) -> c_int;
    pub fn zk_proof_set_transcript(ctx: *mut zk_ctx, proof: *mut zk_proof, vtable: *const zk_transcript_vtable, user: *mut c_void) -> c_int;
    pub fn zk_proof_set_transcript_kind(ctx: *mut zk_ctx, proof: *mut zk_proof, kind: c_int) -> c_int;
    pub fn zk_proof_set_sharding(ctx: *mut zk_ctx, proof: *mut zk_proof, rank: u32, world: u32, gather: zk_allgather_fn, user: *mut c_void) -> c_int;
    pub fn zk_proof_advice_phase(ctx: *mut zk_ctx, proof: *mut zk_proof, col_index: *const u32, h_cols: *const *const c_void, ncols: u32, h_challenges: *mut c_void, num_challenges: *mut u32) -> c_int;
    pub fn zk_proof_finish(ctx: *mut zk_ctx, proof: *mut zk_proof, h_proof: *mut c_void, proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn zk_proof_abort(ctx: *mut zk_ctx, proof: *mut zk_proof);

    // ---- pinned host memory for witness columns (optional: uploads at link rate)
    pub fn zk_host_register(ctx: *mut zk_ctx, ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn zk_host_unregister(ctx: *mut zk_ctx, ptr: *mut c_void) -> c_int;
}
