//! zkmi355 backend of `halo2_proofs::plonk::{create_proof, keygen_pk, keygen_pk2}` (feature `zkmi355`).
//!
//! Division of work (generics cannot cross a C ABI, so orchestration stays here):
//!   Rust  : `Circuit::synthesize` once per phase (witness generation, unchanged upstream code),
//!           `batch_invert_assigned`, the caller's transcript `T` and RNG `R`, error mapping
//!   device: every commitment (MSM), NTT / coset NTT, quotient evaluation, permutation and lookup
//!           arguments, evaluations, SHPLONK / GWC multi-open — `libzkmi355.so`
//!
//! Call sites served, unchanged: `create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<_>, Challenge255<_>, _, Blake2bWrite<..>, _>`
//! [REF circuit-benchmarks/src/super_circuit.rs:117-132], `gen_snark_shplonk` / `gen_evm_proof_shplonk`
//! through snark-verifier-sdk [REF prover/src/common/prover/utils.rs:31, evm.rs:67], `keygen_pk2`
//! [REF prover/src/common/prover/utils.rs:55].
//!
//! NOT compiled in the build image (no Rust toolchain there); see shim/README.md.
mod export;
pub mod ffi;
mod transcript;

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::sync::{Mutex, OnceLock};

use ff::Field;
use halo2curves::bn256::{Bn256, Fr, G1Affine};
use rand_core::RngCore;

use crate::circuit::Value;
use crate::plonk::{
    self, Advice, Any, Assigned, Assignment, Challenge, Circuit, Column, ConstraintSystem, Error, Fixed, FloorPlanner, Instance, Selector, VerifyingKey,
};
use crate::poly::commitment::{CommitmentScheme, Params, Prover};
use crate::poly::kzg::commitment::ParamsKZG;
use crate::poly::{batch_invert_assigned, LagrangeCoeff, Polynomial};
use crate::transcript::{EncodedChallenge, TranscriptWrite};

/// The upstream CPU implementations, for A/B runs next to the routed entry points (`plonk::create_proof` etc. resolve to
/// this module's functions when the feature is on; `plonk::prover` / `plonk::keygen` are private modules).
pub mod cpu {
    pub use crate::plonk::keygen::{keygen_pk, keygen_pk2};
    pub use crate::plonk::prover::create_proof;
}

/// One `zk_ctx` per process and device (the reference drives one proof at a time per `Prover`
/// [REF prover/src/test/chunk.rs:19-25]); `ZKMI355_DEVICE` selects the GPU (default 0; under
/// `torch.distributed`-style launchers set it from LOCAL_RANK).
struct Gpu {
    ctx: *mut ffi::zk_ctx,
    /// SRS handles by (`k`, the raw bytes of `g[1] = s·G`): `ParamsKZG::downsize` yields a new `params` value, uploaded on
    /// first use; two setups of the same size with different secrets (tests do that) are different entries
    srs: HashMap<(u32, [u8; 64]), *mut ffi::zk_srs>,
    /// device-side keys by `vk.transcript_repr()` — a hash of the pinned verifying key, so it survives every move of the
    /// `ProvingKey` (upstream callers return it by value and park it in a map [REF prover/src/common/prover/utils.rs:55-59]);
    /// the address of a `VerifyingKey` does not, and a freed address may be reused by another circuit's key.
    /// Two `ProvingKey`s with the same repr (same circuit, same params) share one device key, which is what they describe.
    keys: HashMap<[u8; 32], *mut ffi::zk_pk>,
}
unsafe impl Send for Gpu {}

fn gpu() -> &'static Mutex<Gpu> {
    static GPU: OnceLock<Mutex<Gpu>> = OnceLock::new();
    GPU.get_or_init(|| {
        let dev: i32 = std::env::var("ZKMI355_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { ffi::zk_ctx_create(dev, &mut ctx) };
        assert!(rc == ffi::ZK_OK, "zkmi355: no gfx950 device ({rc}); there is no CPU fallback — build without the `zkmi355` feature for the CPU prover");
        Mutex::new(Gpu { ctx, srs: HashMap::new(), keys: HashMap::new() })
    })
}

fn check(g: &Gpu, rc: i32, what: &str) -> Result<(), Error> {
    if rc == ffi::ZK_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::zk_last_error(g.ctx)) }.to_string_lossy().into_owned();
    log::error!("zkmi355 {what}: status {rc}: {msg}");
    // never unwind across the FFI; map to the closest halo2 error (SURVEY §8b "Errors")
    Err(match rc {
        // plonk::Error has no variant for a failing backend: a rejected witness / blob is a constraint-system failure,
        // everything else (device, memory) surfaces as Synthesis; the log line above carries the library's message
        ffi::ZK_ERR_INVALID_ARG => Error::ConstraintSystemFailure,
        _ => Error::Synthesis,
    })
}

impl Gpu {
    fn srs(&mut self, params: &ParamsKZG<Bn256>) -> Result<*mut ffi::zk_srs, Error> {
        let k = params.k();
        let (g, gl) = (&params.g, &params.g_lagrange); // pub(crate) fields of ParamsKZG: Vec<G1Affine>, n each, in-memory form = ABI form
        let mut tag = [0u8; 64];
        // G1Affine is two 32-byte Montgomery-form coordinates in memory (the ABI's form): g[1] identifies the secret
        tag.copy_from_slice(unsafe { std::slice::from_raw_parts(&g[1.min(g.len() - 1)] as *const G1Affine as *const u8, 64) });
        let key = (k, tag);
        if let Some(s) = self.srs.get(&key) {
            return Ok(*s);
        }
        let mut out = std::ptr::null_mut();
        let rc = unsafe { ffi::zk_srs_create(self.ctx, k, g.as_ptr() as *const c_void, gl.as_ptr() as *const c_void, &mut out) };
        check(self, rc, "zk_srs_create")?;
        self.srs.insert(key, out);
        Ok(out)
    }
}

// ------------------------------------------------------------------------------------------ keygen
/// `keygen_pk2(params, circuit)` [REF prover/src/common/prover/utils.rs:55]: upstream keygen produces
/// the `ProvingKey` the rest of the Rust world expects (vk, cs, fixed / permutation data); the same
/// fixed and sigma columns go to the device as key blob v3, and the device-side key is remembered
/// under `vk.transcript_repr()` (move-stable: the `ProvingKey` is returned by value and moved again by the caller).  `vk.transcript_repr()` — the value pinned at
/// [REF zkevm-circuits/src/super_circuit/test.rs:70-85] — is installed so that proofs absorb exactly
/// what upstream `verify_proof` absorbs.
pub fn keygen_pk2<ConcreteCircuit>(params: &ParamsKZG<Bn256>, circuit: &ConcreteCircuit) -> Result<plonk::ProvingKey<G1Affine>, Error>
where
    ConcreteCircuit: Circuit<Fr>,
{
    let pk = plonk::keygen::keygen_pk2(params, circuit)?; // upstream, CPU: vk + Rust-side pk
    register_key(params, &pk)?;
    Ok(pk)
}
pub fn keygen_pk<ConcreteCircuit>(params: &ParamsKZG<Bn256>, vk: VerifyingKey<G1Affine>, circuit: &ConcreteCircuit) -> Result<plonk::ProvingKey<G1Affine>, Error>
where
    ConcreteCircuit: Circuit<Fr>,
{
    let pk = plonk::keygen::keygen_pk(params, vk, circuit)?;
    register_key(params, &pk)?;
    Ok(pk)
}

fn register_key(params: &ParamsKZG<Bn256>, pk: &plonk::ProvingKey<G1Affine>) -> Result<(), Error> {
    let vk = pk.get_vk();
    let cs = vk.cs();
    // Lagrange forms kept by upstream's ProvingKey: `fixed_values` (after selector compression) and
    // `permutation.permutations` (sigma columns).  Both are private to `plonk` / `plonk::permutation` upstream: the fork
    // needs `pub(crate)` on `ProvingKey::{fixed_values, permutation}` and on `permutation::ProvingKey::permutations`
    // (shim/README.md, "visibility patch")
    let fixed: Vec<Vec<Fr>> = pk.fixed_values.iter().map(|p| p.to_vec()).collect();
    let sigma: Vec<Vec<Fr>> = pk.permutation.permutations.iter().map(|p| p.to_vec()).collect();
    let blob = export::key_blob(cs, params.k(), &fixed, &sigma);
    let mut g = gpu().lock().unwrap();
    let srs = g.srs(params)?;
    let mut dpk = std::ptr::null_mut();
    let rc = unsafe { ffi::zk_pk_create(g.ctx, srs, blob.as_ptr() as *const c_void, blob.len(), &mut dpk) };
    check(&g, rc, "zk_pk_create")?;
    let repr: Fr = vk.transcript_repr();
    let rc = unsafe { ffi::zk_pk_set_transcript_repr(g.ctx, dpk, &repr as *const Fr as *const c_void) };
    check(&g, rc, "zk_pk_set_transcript_repr")?;
    // sanity: the device's commitments must be upstream's (same SRS, same columns)
    #[cfg(debug_assertions)]
    {
        let n_com = vk.fixed_commitments().len() + vk.permutation().commitments().len();
        let mut coms = vec![G1Affine::default(); n_com];
        let rc = unsafe { ffi::zk_pk_vk(g.ctx, dpk, coms.as_mut_ptr() as *mut c_void, std::ptr::null_mut()) };
        check(&g, rc, "zk_pk_vk")?;
        let want: Vec<G1Affine> = vk.fixed_commitments().iter().chain(vk.permutation().commitments().iter()).cloned().collect();
        assert_eq!(coms, want, "zkmi355: device keygen disagrees with upstream keygen_vk");
    }
    if let Some(old) = g.keys.insert(key_of(vk), dpk) {
        unsafe { ffi::zk_pk_destroy(g.ctx, old) }; // the same circuit keyed again (e.g. after `clear_pks`): the newer device key wins
    }
    Ok(())
}

/// Registry key of a verifying key: the canonical bytes of `vk.transcript_repr()`.
fn key_of(vk: &VerifyingKey<G1Affine>) -> [u8; 32] {
    use ff::PrimeField;
    let repr = vk.transcript_repr().to_repr();
    let mut out = [0u8; 32];
    out.copy_from_slice(repr.as_ref());
    out
}

/// Releases the device-side key of `pk` (its fixed / sigma columns and coset cache — tens of GiB at k = 20).  Upstream's
/// `ProvingKey` has no hook for this; callers that drop keys (`Prover::clear_pks` [REF prover/src/common/prover/utils.rs:49-60])
/// call it next to the drop.  Dropping without it only leaks device memory until `release_all_keys`.
pub fn release_key(pk: &plonk::ProvingKey<G1Affine>) {
    let mut g = gpu().lock().unwrap();
    if let Some(dpk) = g.keys.remove(&key_of(pk.get_vk())) {
        unsafe { ffi::zk_pk_destroy(g.ctx, dpk) };
    }
}
/// Releases every device-side key (what `clear_pks` means for the device).
pub fn release_all_keys() {
    let mut g = gpu().lock().unwrap();
    let ctx = g.ctx;
    for (_, dpk) in g.keys.drain() {
        unsafe { ffi::zk_pk_destroy(ctx, dpk) };
    }
}

// ---------------------------------------------------------------------------------------- prover
/// Which multi-open argument `P: Prover` stands for (upstream types: `ProverSHPLONK`, `ProverGWC`).
pub trait MultiOpenKind {
    const KIND: i32;
}
impl<'p> MultiOpenKind for crate::poly::kzg::multiopen::ProverSHPLONK<'p, Bn256> {
    const KIND: i32 = ffi::ZK_MULTIOPEN_SHPLONK;
}
impl<'p> MultiOpenKind for crate::poly::kzg::multiopen::ProverGWC<'p, Bn256> {
    const KIND: i32 = ffi::ZK_MULTIOPEN_GWC;
}

/// upstream's `WitnessCollection` (plonk/prover.rs), restated: records the advice cells of the
/// current phase, serves the challenges of earlier phases, ignores everything else.
struct WitnessCollection<'a> {
    k: u32,
    current_phase: u8,
    advice_phase: &'a [u8],
    advice: Vec<Polynomial<Assigned<Fr>, LagrangeCoeff>>,
    challenges: &'a HashMap<usize, Fr>,
    instances: &'a [&'a [Fr]],
    usable_rows: std::ops::RangeTo<usize>,
}

impl<'a> Assignment<Fr> for WitnessCollection<'a> {
    fn enter_region<NR, N>(&mut self, _: N) where NR: Into<String>, N: FnOnce() -> NR {}
    fn exit_region(&mut self) {}
    fn enable_selector<A, AR>(&mut self, _: A, _: &Selector, _: usize) -> Result<(), Error> where A: FnOnce() -> AR, AR: Into<String> { Ok(()) }
    fn annotate_column<A, AR>(&mut self, _: A, _: Column<Any>) where A: FnOnce() -> AR, AR: Into<String> {}
    fn query_instance(&self, column: Column<Instance>, row: usize) -> Result<Value<Fr>, Error> {
        if !self.usable_rows.contains(&row) {
            return Err(Error::not_enough_rows_available(self.k));
        }
        self.instances.get(column.index()).and_then(|c| c.get(row)).map(|v| Value::known(*v)).ok_or(Error::BoundsFailure)
    }
    fn assign_advice<V, VR, A, AR>(&mut self, _: A, column: Column<Advice>, row: usize, to: V) -> Result<(), Error>
    where V: FnOnce() -> Value<VR>, VR: Into<Assigned<Fr>>, A: FnOnce() -> AR, AR: Into<String> {
        if self.advice_phase[column.index()] != self.current_phase {
            return Ok(()); // a column of another phase: not collected in this pass
        }
        if !self.usable_rows.contains(&row) {
            return Err(Error::not_enough_rows_available(self.k));
        }
        *self.advice.get_mut(column.index()).and_then(|v| v.get_mut(row)).ok_or(Error::BoundsFailure)? = to().into_field().assign()?;
        Ok(())
    }
    fn assign_fixed<V, VR, A, AR>(&mut self, _: A, _: Column<Fixed>, _: usize, _: V) -> Result<(), Error>
    where V: FnOnce() -> Value<VR>, VR: Into<Assigned<Fr>>, A: FnOnce() -> AR, AR: Into<String> { Ok(()) }
    fn copy(&mut self, _: Column<Any>, _: usize, _: Column<Any>, _: usize) -> Result<(), Error> { Ok(()) }
    fn fill_from_row(&mut self, _: Column<Fixed>, _: usize, _: Value<Assigned<Fr>>) -> Result<(), Error> { Ok(()) }
    fn get_challenge(&self, challenge: Challenge) -> Value<Fr> {
        self.challenges.get(&challenge.index()).cloned().map(Value::known).unwrap_or_else(Value::unknown)
    }
    fn push_namespace<NR, N>(&mut self, _: N) where NR: Into<String>, N: FnOnce() -> NR {}
    fn pop_namespace(&mut self, _: Option<String>) {}
}

/// Drop-in for `halo2_proofs::plonk::create_proof` (KZG over Bn256; one circuit per call, which is
/// what every reference call site passes: `&[circuit]`, `&[&instances]`).
///
/// Same generic parameter list as upstream, so the call sites' turbofish
/// `create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, ChaChaRng, Blake2bWrite<..>, SuperCircuit<..>>`
/// [REF circuit-benchmarks/src/super_circuit.rs:117-124] compiles unchanged; the associated-type equalities on `Scheme` pin it
/// to KZG over Bn256 (the only scheme the reference instantiates), which makes `params`, `pk` and `instances` concrete types here.
pub fn create_proof<'params, Scheme, P, E, R, T, ConcreteCircuit>(
    params: &'params Scheme::ParamsProver,
    pk: &plonk::ProvingKey<Scheme::Curve>,
    circuits: &[ConcreteCircuit],
    instances: &[&[&[Scheme::Scalar]]],
    mut rng: R,
    transcript: &mut T,
) -> Result<(), Error>
where
    Scheme: CommitmentScheme<Scalar = Fr, Curve = G1Affine, ParamsProver = ParamsKZG<Bn256>>,
    P: Prover<'params, Scheme> + MultiOpenKind,
    E: EncodedChallenge<G1Affine>,
    R: RngCore + Send,
    T: TranscriptWrite<G1Affine, E>,
    ConcreteCircuit: Circuit<Fr>,
{
    if circuits.len() != 1 || instances.len() != 1 {
        // halo2 batches several circuit instances into one transcript; no reference call site does.
        return plonk::prover::create_proof::<Scheme, P, E, R, T, ConcreteCircuit>(params, pk, circuits, instances, rng, transcript);
    }
    let (circuit, instance) = (&circuits[0], instances[0]);
    let vk = pk.get_vk();
    let cs: &ConstraintSystem<Fr> = vk.cs();
    if instance.len() != cs.num_instance_columns() {
        return Err(Error::InvalidInstances);
    }
    let n = params.n() as usize;
    let usable = n - (cs.blinding_factors() + 1);
    let mut g = gpu().lock().unwrap();
    let dpk = *g.keys.get(&key_of(vk)).ok_or_else(|| {
        log::error!("zkmi355: this ProvingKey was not produced by zkmi355::keygen_pk / keygen_pk2");
        Error::Synthesis
    })?;

    // ---- session: vk.transcript_repr and the instance values are absorbed by the library, in upstream's order
    let mut seed = [0u8; 16];
    rng.fill_bytes(&mut seed); // blinding rows / blinding polynomial: drawn on the library side from this seed (validity-neutral)
    let inst_ptrs: Vec<*const c_void> = instance.iter().map(|c| c.as_ptr() as *const c_void).collect();
    let inst_lens: Vec<u32> = instance.iter().map(|c| c.len() as u32).collect();
    if inst_lens.iter().any(|l| *l as usize > usable) {
        return Err(Error::InstanceTooLarge);
    }
    let mut sess = std::ptr::null_mut();
    let rc = unsafe { ffi::zk_proof_begin_instances(g.ctx, dpk, inst_ptrs.as_ptr(), inst_lens.as_ptr(), seed.as_ptr(), &mut sess) };
    check(&g, rc, "zk_proof_begin_instances")?;
    let abort = |g: &Gpu, sess| unsafe { ffi::zk_proof_abort(g.ctx, sess) };

    let mut hook = transcript::Hook::<E, T> { transcript, failed: false, _marker: std::marker::PhantomData };
    let vt = transcript::vtable::<E, T>();
    let rc = unsafe { ffi::zk_proof_set_transcript(g.ctx, sess, &vt, &mut hook as *mut _ as *mut c_void) };
    if let Err(e) = check(&g, rc, "zk_proof_set_transcript") { abort(&g, sess); return Err(e); }
    let rc = unsafe { ffi::zk_proof_set_multiopen(g.ctx, sess, P::KIND) };
    if let Err(e) = check(&g, rc, "zk_proof_set_multiopen") { abort(&g, sess); return Err(e); }

    // ---- one synthesis pass per phase (SuperCircuit: three [REF zkevm-circuits/src/util.rs:120-133])
    let config = {
        let mut meta = ConstraintSystem::default();
        #[cfg(feature = "circuit-params")]
        let config = ConcreteCircuit::configure_with_params(&mut meta, circuit.params());
        #[cfg(not(feature = "circuit-params"))]
        let config = ConcreteCircuit::configure(&mut meta);
        config
    };
    let advice_phase: Vec<u8> = cs.advice_column_phase();
    let challenge_phase: Vec<u8> = cs.challenge_phase();
    let mut challenges: HashMap<usize, Fr> = HashMap::new();
    let mut shape = [0u32; 16];
    unsafe { ffi::zk_pk_shape(g.ctx, dpk, shape.as_mut_ptr()) };
    let mut challenge_buf = vec![Fr::ZERO; (shape[10] as usize).max(1)];
    // `cs.phases()` yields sealed::Phase(0) ..= the highest advice phase, in order: the position IS the phase number
    // (the inner u8 is private to plonk::circuit)
    for (phase, _) in cs.phases().enumerate() {
        let phase = phase as u8;
        let mut witness = WitnessCollection {
            k: params.k(),
            current_phase: phase,
            advice_phase: &advice_phase,
            advice: vec![Polynomial { values: vec![Assigned::Zero; n], _marker: std::marker::PhantomData }; cs.num_advice_columns()],
            challenges: &challenges,
            instances: instance,
            usable_rows: ..usable,
        };
        if let Err(e) = ConcreteCircuit::FloorPlanner::synthesize(&mut witness, circuit, config.clone(), cs.constants().clone()) {
            abort(&g, sess);
            return Err(e);
        }
        let cols: Vec<u32> = (0..cs.num_advice_columns() as u32).filter(|i| advice_phase[*i as usize] == phase).collect();
        let mut picked: Vec<Polynomial<Assigned<Fr>, LagrangeCoeff>> = Vec::with_capacity(cols.len());
        for i in &cols {
            picked.push(std::mem::replace(&mut witness.advice[*i as usize], Polynomial { values: Vec::new(), _marker: std::marker::PhantomData }));
        }
        drop(witness);
        let values: Vec<Polynomial<Fr, LagrangeCoeff>> = batch_invert_assigned(picked); // rational -> field, as upstream
        let ptrs: Vec<*const c_void> = values.iter().map(|c| c.as_ptr() as *const c_void).collect();
        // Page-locked witness memory uploads at 0.60 ms per 32 MiB column, pageable memory at 0.80 ms (tools/h2d_rate.py) -- but
        // registering a column costs 0.66 ms by itself, so pinning right here only pays when it happens off the critical path.
        // Off by default (ZKMI355_PIN_WITNESS=1 turns it on); the real fix is a synthesis that writes into a `zk_host_alloc` arena
        // kept across proofs.
        let pin = std::env::var("ZKMI355_PIN_WITNESS").map(|v| v == "1").unwrap_or(false);
        let mut pinned: Vec<*mut c_void> = Vec::new();
        if pin {
            for c in &values {
                let p = c.as_ptr() as *mut c_void;
                if unsafe { ffi::zk_host_register(g.ctx, p, c.len() * 32) } == ffi::ZK_OK {
                    pinned.push(p);
                }
            }
        }
        let mut count = challenge_buf.len() as u32;
        let rc = unsafe { ffi::zk_proof_advice_phase(g.ctx, sess, cols.as_ptr(), ptrs.as_ptr(), cols.len() as u32, challenge_buf.as_mut_ptr() as *mut c_void, &mut count) };
        for p in pinned {
            unsafe { ffi::zk_host_unregister(g.ctx, p) }; // the phase has consumed the columns when it returns
        }
        if let Err(e) = check(&g, rc, "zk_proof_advice_phase") { abort(&g, sess); return Err(e); }
        // the challenges that became usable after this phase, in challenge-index order
        let mut next = 0usize;
        for (idx, p) in challenge_phase.iter().enumerate() {
            if *p == phase {
                challenges.insert(idx, challenge_buf[next]);
                next += 1;
            }
        }
        debug_assert_eq!(next, count as usize);
    }

    // ---- everything else runs on the device; the proof bytes went through `transcript`
    let mut len = 0usize;
    let mut dummy = [0u8; 8];
    let rc = unsafe { ffi::zk_proof_finish(g.ctx, sess, dummy.as_mut_ptr() as *mut c_void, dummy.len(), &mut len) };
    check(&g, rc, "zk_proof_finish")?; // finish frees the session also on failure
    if hook.failed {
        return Err(Error::Transcript(std::io::Error::new(std::io::ErrorKind::Other, "transcript write failed")));
    }
    Ok(())
}
