//! Tier T1: a proof made by the zkmi355 backend is accepted by the UPSTREAM verifier —
//! the reference's only acceptance criterion for the hot path
//! [REF circuit-benchmarks/src/super_circuit.rs:141-154].  Needs cargo + an MI355X:
//!   ZKMI355_LIB_DIR=… cargo test --features zkmi355 --test t1_verify -- --nocapture
#![cfg(feature = "zkmi355")]

use halo2_proofs::circuit::{Layouter, SimpleFloorPlanner, Value};
use halo2_proofs::halo2curves::bn256::{Bn256, Fr, G1Affine};
use halo2_proofs::plonk::{
    self, verify_proof, Advice, Circuit, Column, ConstraintSystem, Error, Fixed, Instance, Selector, TableColumn,
};
use halo2_proofs::poly::kzg::commitment::{KZGCommitmentScheme, ParamsKZG};
use halo2_proofs::poly::kzg::multiopen::{ProverGWC, ProverSHPLONK, VerifierGWC, VerifierSHPLONK};
use halo2_proofs::poly::kzg::strategy::SingleStrategy;
use halo2_proofs::poly::Rotation;
use halo2_proofs::transcript::{Blake2bRead, Blake2bWrite, Challenge255, TranscriptReadBuffer, TranscriptWriterBuffer};
use halo2_proofs::zkmi355;
use rand_core::SeedableRng;
use rand_xorshift::XorShiftRng;

/// gates with a rotation, a degree-5 gate, copy constraints into the instance column, two lookups into
/// the same table (merged by chunk_lookups into one argument with two input tuples), a second phase.
#[derive(Clone, Default)]
struct TestCircuit {
    rows: usize,
}
#[derive(Clone)]
struct TestConfig {
    a: Column<Advice>,
    b: Column<Advice>,
    c: Column<Advice>,
    rlc: Column<Advice>,
    q_mul: Selector,
    q_pow: Selector,
    q_lk: Selector,
    table: TableColumn,
    constant: Column<Fixed>,
    instance: Column<Instance>,
    challenge: plonk::Challenge,
}

impl Circuit<Fr> for TestCircuit {
    type Config = TestConfig;
    type FloorPlanner = SimpleFloorPlanner;
    #[cfg(feature = "circuit-params")]
    type Params = ();

    fn without_witnesses(&self) -> Self {
        self.clone()
    }

    fn configure(meta: &mut ConstraintSystem<Fr>) -> TestConfig {
        let (a, b, c) = (meta.advice_column(), meta.advice_column(), meta.advice_column());
        let challenge = meta.challenge_usable_after(plonk::FirstPhase);
        let rlc = meta.advice_column_in(plonk::SecondPhase);
        let (q_mul, q_pow, q_lk) = (meta.selector(), meta.selector(), meta.complex_selector());
        let table = meta.lookup_table_column();
        let constant = meta.fixed_column();
        let instance = meta.instance_column();
        for col in [a, b, c] {
            meta.enable_equality(col);
        }
        meta.enable_equality(instance);
        meta.enable_constant(constant);
        meta.create_gate("mul", |m| {
            let (q, a, b, c_next) = (m.query_selector(q_mul), m.query_advice(a, Rotation::cur()), m.query_advice(b, Rotation::cur()), m.query_advice(c, Rotation::next()));
            vec![q * (a * b - c_next)]
        });
        meta.create_gate("degree 5", |m| {
            // vanishes wherever the mul gate does; raises the circuit degree so that the quotient has several pieces
            let (q, a, b, c_next) = (m.query_selector(q_pow), m.query_advice(a, Rotation::cur()), m.query_advice(b, Rotation::cur()), m.query_advice(c, Rotation::next()));
            let one = halo2_proofs::plonk::Expression::Constant(Fr::one());
            vec![q * (a.clone() * b.clone() - c_next) * (a + one.clone()) * (b + one.clone() + one)]
        });
        meta.create_gate("rlc", |m| {
            let (q, a, b, r) = (m.query_selector(q_mul), m.query_advice(a, Rotation::cur()), m.query_advice(b, Rotation::cur()), m.query_advice(rlc, Rotation::cur()));
            let ch = m.query_challenge(challenge);
            vec![q * (a + ch * b - r)]
        });
        // two lookups into the same table: merged into one mv-lookup argument by chunk_lookups()
        meta.lookup("a in table", |m| {
            let (q, a) = (m.query_selector(q_lk), m.query_advice(a, Rotation::cur()));
            vec![(q * a, table)]
        });
        meta.lookup("b in table", |m| {
            let (q, b) = (m.query_selector(q_lk), m.query_advice(b, Rotation::cur()));
            vec![(q * b, table)]
        });
        // keygen (in this fork) applies `chunk_lookups()` to the configured system, as for every circuit of the reference
        // [REF zkevm-circuits/src/super_circuit/test.rs:59 does the same by hand to read the degree]
        TestConfig { a, b, c, rlc, q_mul, q_pow, q_lk, table, constant, instance, challenge }
    }

    fn synthesize(&self, cfg: TestConfig, mut layouter: impl Layouter<Fr>) -> Result<(), Error> {
        layouter.assign_table(|| "table", |mut t| {
            for i in 0..64usize {
                t.assign_cell(|| "t", cfg.table, i, || Value::known(Fr::from((i * i) as u64)))?;
            }
            Ok(())
        })?;
        let ch = layouter.get_challenge(cfg.challenge);
        let out = layouter.assign_region(|| "main", |mut region| {
            let mut last = None;
            for i in 0..self.rows {
                let row = 3 * i + 1;
                let (x, y) = (Fr::from((i % 8) as u64).square(), Fr::from(((i + 3) % 8) as u64).square());
                cfg.q_mul.enable(&mut region, row)?;
                cfg.q_lk.enable(&mut region, row)?;
                region.assign_advice(|| "a", cfg.a, row, || Value::known(x))?;
                region.assign_advice(|| "b", cfg.b, row, || Value::known(y))?;
                last = Some(region.assign_advice(|| "c", cfg.c, row + 1, || Value::known(x * y))?);
                region.assign_advice(|| "rlc", cfg.rlc, row, || ch.map(|r| x + r * y))?;
                cfg.q_pow.enable(&mut region, row)?;
            }
            Ok(last.unwrap())
        });
        let out = out?;
        layouter.constrain_instance(out.cell(), cfg.instance, 0)
    }
}

fn prove_and_verify(shplonk: bool) {
    let k = 10;
    let circuit = TestCircuit { rows: 100 };
    let public = {
        let i = circuit.rows - 1;
        Fr::from((i % 8) as u64).square() * Fr::from(((i + 3) % 8) as u64).square()
    };
    // upstream MockProver first: the witness must satisfy the circuit
    halo2_proofs::dev::MockProver::run(k, &circuit, vec![vec![public]]).unwrap().assert_satisfied_par();
    let params = ParamsKZG::<Bn256>::unsafe_setup_with_s(k, Fr::from(1234u64)); // as [REF zkevm-circuits/src/super_circuit/test.rs:74]
    let pk = zkmi355::keygen_pk2(&params, &circuit).expect("keygen");
    let rng = XorShiftRng::from_seed([0u8; 16]); // the reference prover's gen_rng [REF prover/src/utils.rs:192-195]
    let mut transcript = Blake2bWrite::<_, G1Affine, Challenge255<_>>::init(vec![]);
    if shplonk {
        zkmi355::create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _, _>(&params, &pk, &[circuit.clone()], &[&[&[public]]], rng, &mut transcript).expect("proof");
    } else {
        zkmi355::create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC<'_, Bn256>, Challenge255<G1Affine>, _, _, _>(&params, &pk, &[circuit.clone()], &[&[&[public]]], rng, &mut transcript).expect("proof");
    }
    let proof = transcript.finalize();
    // ---- the UPSTREAM verifier, unmodified [REF circuit-benchmarks/src/super_circuit.rs:141-154]
    let verifier_params = params.verifier_params();
    let strategy = SingleStrategy::new(&params);
    let mut read = Blake2bRead::<_, G1Affine, Challenge255<_>>::init(&proof[..]);
    let ok = if shplonk {
        verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _>(verifier_params, pk.get_vk(), strategy, &[&[&[public]]], &mut read)
    } else {
        verify_proof::<KZGCommitmentScheme<Bn256>, VerifierGWC<'_, Bn256>, Challenge255<G1Affine>, _, _>(verifier_params, pk.get_vk(), strategy, &[&[&[public]]], &mut read)
    };
    ok.expect("upstream verify_proof must accept the zkmi355 proof (T1)");
    // and a tampered proof must not pass
    let mut bad = proof.clone();
    let mid = bad.len() / 2;
    bad[mid] ^= 1;
    let mut read = Blake2bRead::<_, G1Affine, Challenge255<_>>::init(&bad[..]);
    let strategy = SingleStrategy::new(&params);
    assert!(verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _>(verifier_params, pk.get_vk(), strategy, &[&[&[public]]], &mut read).is_err() || !shplonk);
}

#[test]
fn t1_shplonk_blake2b() {
    prove_and_verify(true);
}

#[test]
fn t1_gwc_blake2b() {
    prove_and_verify(false);
}

/// A/B: the upstream CPU prover and the zkmi355 prover are both accepted for the same key and
/// witness (their bytes differ: the blinding values come from different generators).
#[test]
fn cpu_and_gpu_provers_are_interchangeable() {
    let k = 10;
    let circuit = TestCircuit { rows: 100 };
    let public = {
        let i = circuit.rows - 1;
        Fr::from((i % 8) as u64).square() * Fr::from(((i + 3) % 8) as u64).square()
    };
    let params = ParamsKZG::<Bn256>::unsafe_setup_with_s(k, Fr::from(1234u64));
    let pk = zkmi355::keygen_pk2(&params, &circuit).unwrap();
    for gpu in [false, true] {
        let rng = XorShiftRng::from_seed([7u8; 16]);
        let mut transcript = Blake2bWrite::<_, G1Affine, Challenge255<_>>::init(vec![]);
        if gpu {
            zkmi355::create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _, _>(&params, &pk, &[circuit.clone()], &[&[&[public]]], rng, &mut transcript).unwrap();
        } else {
            zkmi355::cpu::create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _, _>(&params, &pk, &[circuit.clone()], &[&[&[public]]], rng, &mut transcript).unwrap();
        }
        let proof = transcript.finalize();
        let mut read = Blake2bRead::<_, G1Affine, Challenge255<_>>::init(&proof[..]);
        verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _>(params.verifier_params(), pk.get_vk(), SingleStrategy::new(&params), &[&[&[public]]], &mut read)
            .expect("accepted");
    }
}

/// The reference returns the `ProvingKey` by value and parks it in a map before proving
/// [REF prover/src/common/prover/utils.rs:49-60]: the device-side key must be found after any number of moves
/// (the registry is keyed by `vk.transcript_repr()`, not by an address), and two circuits must never see each other's key.
#[test]
fn proving_key_survives_moves_between_keygen_and_create_proof() {
    use std::collections::BTreeMap;
    let k = 10;
    let params = ParamsKZG::<Bn256>::unsafe_setup_with_s(k, Fr::from(1234u64));
    let mut pk_map: BTreeMap<String, halo2_proofs::plonk::ProvingKey<G1Affine>> = BTreeMap::new();
    for rows in [100usize, 60] {
        let circuit = TestCircuit { rows };
        let pk = zkmi355::keygen_pk2(&params, &circuit).unwrap(); // returned by value ...
        let boxed = Box::new(pk); // ... moved to the heap ...
        pk_map.insert(format!("layer{rows}"), *boxed); // ... and into the map, as `Prover::pk_map` does
    }
    for rows in [60usize, 100] {
        let circuit = TestCircuit { rows };
        let public = {
            let i = circuit.rows - 1;
            Fr::from((i % 8) as u64).square() * Fr::from(((i + 3) % 8) as u64).square()
        };
        let pk = &pk_map[&format!("layer{rows}")];
        let mut transcript = Blake2bWrite::<_, G1Affine, Challenge255<_>>::init(vec![]);
        zkmi355::create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _, _>(&params, pk, &[circuit.clone()], &[&[&[public]]], XorShiftRng::from_seed([1u8; 16]), &mut transcript)
            .expect("the device key is found after the moves");
        let proof = transcript.finalize();
        let mut read = Blake2bRead::<_, G1Affine, Challenge255<_>>::init(&proof[..]);
        verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _>(params.verifier_params(), pk.get_vk(), SingleStrategy::new(&params), &[&[&[public]]], &mut read)
            .expect("accepted");
    }
    for (_, pk) in pk_map.iter() {
        zkmi355::release_key(pk); // what `clear_pks` needs next to its `pk_map.clear()`
    }
}
