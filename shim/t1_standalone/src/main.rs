//! T1: proofs made by zkmi355 in front of the reference's own verifier.
//!
//!   t1_standalone repr   <kit dir>    per case: read desc.txt + params.bin, `keygen_vk` with upstream halo2, write
//!                                     vk_repr.hex (vk.transcript_repr(), canonical big-endian hex) and vk_commitments.hex,
//!                                     compare the commitments with expect_vk_commitments.hex (zkmi355's keygen)
//!   t1_standalone verify <kit dir>    per case: upstream `verify_proof` on proof_shplonk.bin / proof_gwc.bin, with the call
//!                                     of the reference's benchmark [REF circuit-benchmarks/src/super_circuit.rs:137-154]
//!
//! The circuits are DATA (desc.txt, written by tools/t1_kit.py from the fixtures of the GPU test suite): `configure` replays the
//! configure-time calls in the recorded order, `synthesize` assigns the fixed cells and issues the copy constraints in the
//! recorded order.  Selectors are plain fixed columns (no selector compression), lookups go through `lookup_any` (the mv-lookup
//! fork; `keygen_vk` chunks them itself).  Nothing of zkmi355 is linked into this program.
use std::collections::HashMap;
use std::fs;
use std::io::BufReader;
use std::path::{Path, PathBuf};
use std::sync::OnceLock;

use halo2_proofs::circuit::{Cell, Layouter, SimpleFloorPlanner, Value};
use halo2_proofs::halo2curves::bn256::{Bn256, Fr, G1Affine};
use halo2_proofs::halo2curves::ff::PrimeField;
use halo2_proofs::halo2curves::group::GroupEncoding;
use halo2_proofs::plonk::{
    keygen_vk, verify_proof, Advice, Challenge, Circuit, Column, ConstraintSystem, Error, Expression, FirstPhase, Fixed, Instance,
    SecondPhase, ThirdPhase, VirtualCells,
};
use halo2_proofs::poly::commitment::{Params, ParamsProver};
use halo2_proofs::poly::kzg::commitment::{KZGCommitmentScheme, ParamsKZG};
use halo2_proofs::poly::kzg::multiopen::{VerifierGWC, VerifierSHPLONK};
use halo2_proofs::poly::kzg::strategy::SingleStrategy;
use halo2_proofs::poly::Rotation;
use halo2_proofs::transcript::{Blake2bRead, Challenge255, TranscriptReadBuffer};
use halo2_proofs::SerdeFormat;

// ------------------------------------------------------------------------------------------- the description
#[derive(Clone, Debug)]
enum Op {
    Gate(Vec<String>),
    /// name, input expressions, table expressions (postfix token lists)
    Lookup(String, Vec<Vec<String>>, Vec<Vec<String>>),
    /// column kind ("fixed" | "advice" | "instance"), index
    EnableEquality(String, usize),
}

type CellRef = (String, usize, usize); // kind, column, row

#[derive(Clone, Debug, Default)]
struct Desc {
    k: u32,
    blinding_factors: usize,
    degree: usize,
    num_fixed: usize,
    advice_phase: Vec<u8>,
    num_instance: usize,
    challenge_phase: Vec<u8>,
    ops: Vec<Op>,
    fixed_cells: Vec<(usize, usize, Fr)>,
    copies: Vec<(CellRef, CellRef)>,
}

fn fr_from_hex(h: &str) -> Fr {
    // canonical value, big-endian hex, any length up to 64 digits
    let h = h.trim_start_matches("0x");
    let padded = format!("{:0>64}", h);
    let mut le = [0u8; 32];
    for i in 0..32 {
        le[31 - i] = u8::from_str_radix(&padded[2 * i..2 * i + 2], 16).expect("hex digit");
    }
    Option::<Fr>::from(Fr::from_repr(le)).expect("canonical field element")
}

fn fr_to_hex(v: &Fr) -> String {
    let repr = v.to_repr();
    let le: &[u8] = repr.as_ref();
    le.iter().rev().map(|b| format!("{:02x}", b)).collect()
}

fn parse_desc(text: &str) -> Desc {
    let mut d = Desc::default();
    let mut lines = text.lines().filter(|l| !l.trim().is_empty());
    assert_eq!(lines.next().unwrap().trim(), "zkmi355-t1-kit 1", "not a kit description");
    for line in lines {
        let tok: Vec<&str> = line.split_whitespace().collect();
        match tok[0] {
            "k" => d.k = tok[1].parse().unwrap(),
            "blinding_factors" => d.blinding_factors = tok[1].parse().unwrap(),
            "degree" => d.degree = tok[1].parse().unwrap(),
            "fixed" => d.num_fixed = tok[1].parse().unwrap(),
            "advice" => d.advice_phase = tok[1..].iter().map(|p| p.parse().unwrap()).collect(),
            "instance" => d.num_instance = tok[1].parse().unwrap(),
            "challenges" => d.challenge_phase = tok[1..].iter().map(|p| p.parse().unwrap()).collect(),
            "gate" => d.ops.push(Op::Gate(tok[1..].iter().map(|s| s.to_string()).collect())),
            "lookup" => {
                let n_in: usize = tok[2].parse().unwrap();
                let mut groups: Vec<Vec<String>> = vec![vec![]];
                for t in &tok[3..] {
                    if *t == "|" {
                        groups.push(vec![]);
                    } else {
                        groups.last_mut().unwrap().push(t.to_string());
                    }
                }
                assert_eq!(groups.len(), 2 * n_in, "a lookup has as many table expressions as input expressions");
                let tables = groups.split_off(n_in);
                d.ops.push(Op::Lookup(tok[1].to_string(), groups, tables));
            }
            "enable_equality" => d.ops.push(Op::EnableEquality(tok[1].to_string(), tok[2].parse().unwrap())),
            "fixed_cell" => d.fixed_cells.push((tok[1].parse().unwrap(), tok[2].parse().unwrap(), fr_from_hex(tok[3]))),
            "copy" => d.copies.push((
                (tok[1].to_string(), tok[2].parse().unwrap(), tok[3].parse().unwrap()),
                (tok[4].to_string(), tok[5].parse().unwrap(), tok[6].parse().unwrap()),
            )),
            "end" => break,
            other => panic!("unknown line kind {other}"),
        }
    }
    d
}

/// `Circuit::configure` has no `self`: the description of the case being processed lives here.
static DESC: OnceLock<std::sync::Mutex<Desc>> = OnceLock::new();
fn current_desc() -> Desc {
    DESC.get().expect("description set").lock().unwrap().clone()
}
fn set_desc(d: Desc) {
    let m = DESC.get_or_init(|| std::sync::Mutex::new(Desc::default()));
    *m.lock().unwrap() = d;
}

// ------------------------------------------------------------------------------------------- the circuit
#[derive(Clone, Debug)]
struct KitConfig {
    fixed: Vec<Column<Fixed>>,
    advice: Vec<Column<Advice>>,
    instance: Vec<Column<Instance>>,
}

#[derive(Clone, Debug, Default)]
struct KitCircuit {
    desc: Desc,
}

/// postfix tokens -> Expression, querying the columns in token order (= the order zkmi355's key blob lists the queries in)
fn build_expr(meta: &mut VirtualCells<'_, Fr>, cfg: &KitConfig, challenges: &[Challenge], tokens: &[String]) -> Expression<Fr> {
    let mut st: Vec<Expression<Fr>> = vec![];
    for t in tokens {
        match t.as_str() {
            "+" => {
                let b = st.pop().unwrap();
                let a = st.pop().unwrap();
                st.push(a + b);
            }
            "-" => {
                let b = st.pop().unwrap();
                let a = st.pop().unwrap();
                st.push(a - b);
            }
            "*" => {
                let b = st.pop().unwrap();
                let a = st.pop().unwrap();
                st.push(a * b);
            }
            "n" => {
                let a = st.pop().unwrap();
                st.push(-a);
            }
            leaf => {
                let parts: Vec<&str> = leaf.split(':').collect();
                match parts[0] {
                    "a" => st.push(meta.query_advice(cfg.advice[parts[1].parse::<usize>().unwrap()], Rotation(parts[2].parse().unwrap()))),
                    "f" => st.push(meta.query_fixed(cfg.fixed[parts[1].parse::<usize>().unwrap()], Rotation(parts[2].parse().unwrap()))),
                    "i" => st.push(meta.query_instance(cfg.instance[parts[1].parse::<usize>().unwrap()], Rotation(parts[2].parse().unwrap()))),
                    "k" => st.push(Expression::Constant(fr_from_hex(parts[1]))),
                    "c" => st.push(meta.query_challenge(challenges[parts[1].parse::<usize>().unwrap()])),
                    other => panic!("unknown token kind {other}"),
                }
            }
        }
    }
    assert_eq!(st.len(), 1, "malformed expression");
    st.pop().unwrap()
}

impl Circuit<Fr> for KitCircuit {
    type Config = KitConfig;
    type FloorPlanner = SimpleFloorPlanner;
    #[cfg(feature = "circuit-params")]
    type Params = ();

    fn without_witnesses(&self) -> Self {
        self.clone()
    }

    fn configure(meta: &mut ConstraintSystem<Fr>) -> KitConfig {
        let d = current_desc();
        let fixed: Vec<Column<Fixed>> = (0..d.num_fixed).map(|_| meta.fixed_column()).collect();
        let advice: Vec<Column<Advice>> = d
            .advice_phase
            .iter()
            .map(|p| match *p {
                0 => meta.advice_column_in(FirstPhase),
                1 => meta.advice_column_in(SecondPhase),
                2 => meta.advice_column_in(ThirdPhase),
                _ => panic!("halo2 has three phases"),
            })
            .collect();
        let instance: Vec<Column<Instance>> = (0..d.num_instance).map(|_| meta.instance_column()).collect();
        let challenges: Vec<Challenge> = d
            .challenge_phase
            .iter()
            .map(|p| match *p {
                0 => meta.challenge_usable_after(FirstPhase),
                1 => meta.challenge_usable_after(SecondPhase),
                2 => meta.challenge_usable_after(ThirdPhase),
                _ => panic!("halo2 has three phases"),
            })
            .collect();
        let cfg = KitConfig { fixed, advice, instance };
        for op in &d.ops {
            match op {
                Op::Gate(tokens) => {
                    meta.create_gate("kit gate", |meta| vec![build_expr(meta, &cfg, &challenges, tokens)]);
                }
                Op::Lookup(name, inputs, tables) => {
                    let name: &'static str = Box::leak(name.clone().into_boxed_str()); // as [REF zkevm-circuits/src/evm_circuit/execution.rs:981]
                    meta.lookup_any(name, |meta| {
                        // all input expressions first, then all table expressions: the order their columns were registered in
                        let ins: Vec<Expression<Fr>> = inputs.iter().map(|t| build_expr(meta, &cfg, &challenges, t)).collect();
                        let tabs: Vec<Expression<Fr>> = tables.iter().map(|t| build_expr(meta, &cfg, &challenges, t)).collect();
                        ins.into_iter().zip(tabs.into_iter()).collect::<Vec<_>>()
                    });
                }
                Op::EnableEquality(kind, idx) => match kind.as_str() {
                    "fixed" => meta.enable_equality(cfg.fixed[*idx]),
                    "advice" => meta.enable_equality(cfg.advice[*idx]),
                    "instance" => meta.enable_equality(cfg.instance[*idx]),
                    other => panic!("unknown column kind {other}"),
                },
            }
        }
        cfg
    }

    fn synthesize(&self, cfg: KitConfig, mut layouter: impl Layouter<Fr>) -> Result<(), Error> {
        let d = &self.desc;
        // ONE region at offset 0: rows of the description are absolute rows.  Fixed cells first, then the copy constraints
        // between advice / fixed cells in the recorded order (the order decides the cycles of the permutation).
        let pending_instance: Vec<(Cell, usize, usize)> = layouter.assign_region(
            || "kit",
            |mut region| {
                let mut cells: HashMap<CellRef, Cell> = HashMap::new();
                for (col, row, v) in &d.fixed_cells {
                    let c = region.assign_fixed(|| "fixed", cfg.fixed[*col], *row, || Value::known(*v))?;
                    cells.insert(("fixed".to_string(), *col, *row), c.cell());
                }
                let mut cell_of = |region: &mut halo2_proofs::circuit::Region<'_, Fr>, r: &CellRef| -> Result<Cell, Error> {
                    if let Some(c) = cells.get(r) {
                        return Ok(*c);
                    }
                    let c = match r.0.as_str() {
                        // keygen never looks at advice values: the cell only has to exist
                        "advice" => region.assign_advice(|| "advice", cfg.advice[r.1], r.2, || Value::<Fr>::unknown())?.cell(),
                        "fixed" => region.assign_fixed(|| "fixed zero", cfg.fixed[r.1], r.2, || Value::known(Fr::zero()))?.cell(),
                        other => panic!("no region cell for a column of kind {other}"),
                    };
                    cells.insert(r.clone(), c);
                    Ok(c)
                };
                let mut pending = vec![];
                for (a, b) in &d.copies {
                    if a.0 == "instance" || b.0 == "instance" {
                        let (cell_side, inst_side) = if a.0 == "instance" { (b, a) } else { (a, b) };
                        pending.push((cell_of(&mut region, cell_side)?, inst_side.1, inst_side.2));
                    } else {
                        assert!(pending.is_empty(), "copies with instance cells come last in a kit description");
                        let ca = cell_of(&mut region, a)?;
                        let cb = cell_of(&mut region, b)?;
                        region.constrain_equal(ca, cb)?;
                    }
                }
                Ok(pending)
            },
        )?;
        for (cell, col, row) in pending_instance {
            layouter.constrain_instance(cell, cfg.instance[col], row)?;
        }
        Ok(())
    }
}

// ------------------------------------------------------------------------------------------- the two commands
fn case_dirs(kit: &Path) -> Vec<PathBuf> {
    let mut v: Vec<PathBuf> = fs::read_dir(kit)
        .expect("kit directory")
        .filter_map(|e| e.ok().map(|e| e.path()))
        .filter(|p| p.join("desc.txt").is_file())
        .collect();
    v.sort();
    v
}

fn load_case(dir: &Path) -> (Desc, ParamsKZG<Bn256>) {
    let d = parse_desc(&fs::read_to_string(dir.join("desc.txt")).unwrap());
    let f = fs::File::open(dir.join("params.bin")).expect("params.bin");
    // the format the reference's prover reads its SRS files in [REF prover/src/utils.rs:33,77]; RawBytes also checks the curve equation
    let params = ParamsKZG::<Bn256>::read_custom(&mut BufReader::new(f), SerdeFormat::RawBytes).expect("params.bin: ParamsKZG::read_custom");
    assert_eq!(params.k(), d.k, "params.bin is for another k");
    (d, params)
}

fn hex(bytes: &[u8]) -> String {
    bytes.iter().map(|b| format!("{:02x}", b)).collect()
}

fn keygen(d: &Desc, params: &ParamsKZG<Bn256>) -> halo2_proofs::plonk::VerifyingKey<G1Affine> {
    set_desc(d.clone());
    let circuit = KitCircuit { desc: d.clone() };
    let vk = keygen_vk(params, &circuit).expect("keygen_vk");
    let cs = vk.cs();
    assert_eq!(cs.degree(), d.degree, "cs.degree() differs from the description: the constraint systems are not the same");
    assert_eq!(cs.blinding_factors(), d.blinding_factors, "cs.blinding_factors() differs from the description");
    vk
}

fn cmd_repr(kit: &Path) -> bool {
    let mut all_equal = true;
    for dir in case_dirs(kit) {
        let name = dir.file_name().unwrap().to_string_lossy().to_string();
        let (d, params) = load_case(&dir);
        let vk = keygen(&d, &params);
        let repr: Fr = vk.transcript_repr();
        fs::write(dir.join("vk_repr.hex"), format!("{}\n", fr_to_hex(&repr))).unwrap();
        let coms: Vec<String> = vk
            .fixed_commitments()
            .iter()
            .chain(vk.permutation().commitments().iter())
            .map(|p| hex(p.to_bytes().as_ref()))
            .collect();
        fs::write(dir.join("vk_commitments.hex"), coms.join("\n") + "\n").unwrap();
        let expect: Vec<String> = fs::read_to_string(dir.join("expect_vk_commitments.hex")).unwrap().split_whitespace().map(|s| s.to_string()).collect();
        let same = expect == coms;
        all_equal &= same;
        println!(
            "{name}: k = {}, degree {}, vk.transcript_repr = 0x{}, fixed + permutation commitments {} zkmi355's keygen",
            d.k,
            d.degree,
            fr_to_hex(&repr),
            if same { "EQUAL" } else { "DIFFER from" }
        );
        if !same {
            for (i, (a, b)) in coms.iter().zip(expect.iter()).enumerate() {
                if a != b {
                    println!("    commitment {i}: upstream {a}  zkmi355 {b}");
                }
            }
            if coms.len() != expect.len() {
                println!("    {} commitments upstream, {} from zkmi355", coms.len(), expect.len());
            }
        }
    }
    all_equal
}

fn cmd_verify(kit: &Path) -> bool {
    let mut all_ok = true;
    let mut seen = 0;
    for dir in case_dirs(kit) {
        let name = dir.file_name().unwrap().to_string_lossy().to_string();
        let (d, params) = load_case(&dir);
        let vk = keygen(&d, &params);
        let instances: Vec<Vec<Fr>> = fs::read_to_string(dir.join("instances.txt"))
            .unwrap_or_default()
            .lines()
            .map(|l| l.split_whitespace().map(fr_from_hex).collect())
            .collect();
        assert_eq!(instances.len(), d.num_instance, "instances.txt: one line per instance column");
        let inst_refs: Vec<&[Fr]> = instances.iter().map(|c| &c[..]).collect();
        let verifier_params = params.verifier_params();
        for (file, shplonk) in [("proof_shplonk.bin", true), ("proof_gwc.bin", false)] {
            let Ok(proof) = fs::read(dir.join(file)) else { continue };
            seen += 1;
            let mut transcript = Blake2bRead::<_, G1Affine, Challenge255<_>>::init(&proof[..]);
            let strategy = SingleStrategy::new(&params);
            // the call of [REF circuit-benchmarks/src/super_circuit.rs:141-154]
            let res = if shplonk {
                verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, Challenge255<G1Affine>, Blake2bRead<&[u8], G1Affine, Challenge255<G1Affine>>, SingleStrategy<'_, Bn256>>(
                    verifier_params,
                    &vk,
                    strategy,
                    &[&inst_refs[..]],
                    &mut transcript,
                )
            } else {
                verify_proof::<KZGCommitmentScheme<Bn256>, VerifierGWC<'_, Bn256>, Challenge255<G1Affine>, Blake2bRead<&[u8], G1Affine, Challenge255<G1Affine>>, SingleStrategy<'_, Bn256>>(
                    verifier_params,
                    &vk,
                    strategy,
                    &[&inst_refs[..]],
                    &mut transcript,
                )
            };
            match res {
                Ok(_) => println!("{name}: {file} ({} bytes) ACCEPTED by upstream verify_proof", proof.len()),
                Err(e) => {
                    all_ok = false;
                    println!("{name}: {file} ({} bytes) REJECTED by upstream verify_proof: {e:?}", proof.len());
                }
            }
            // and the verifier is not vacuous: one flipped bit must be refused
            let mut bad = proof.clone();
            let mid = bad.len() / 2;
            bad[mid] ^= 1;
            let mut transcript = Blake2bRead::<_, G1Affine, Challenge255<_>>::init(&bad[..]);
            let strategy = SingleStrategy::new(&params);
            let tampered_ok = if shplonk {
                verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, Challenge255<G1Affine>, Blake2bRead<&[u8], G1Affine, Challenge255<G1Affine>>, SingleStrategy<'_, Bn256>>(verifier_params, &vk, strategy, &[&inst_refs[..]], &mut transcript).is_ok()
            } else {
                verify_proof::<KZGCommitmentScheme<Bn256>, VerifierGWC<'_, Bn256>, Challenge255<G1Affine>, Blake2bRead<&[u8], G1Affine, Challenge255<G1Affine>>, SingleStrategy<'_, Bn256>>(verifier_params, &vk, strategy, &[&inst_refs[..]], &mut transcript).is_ok()
            };
            if tampered_ok {
                all_ok = false;
                println!("{name}: {file} with one bit flipped was ACCEPTED: the check is vacuous");
            }
        }
    }
    if seen == 0 {
        println!("no proof_*.bin found: run `python tools/t1_kit.py prove <kit dir>` after `repr`");
        return false;
    }
    println!("{}", if all_ok { "T1: every zkmi355 proof was accepted by upstream verify_proof" } else { "T1: FAILED (see above)" });
    all_ok
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if args.len() != 3 || !(args[1] == "repr" || args[1] == "verify") {
        eprintln!("usage: t1_standalone repr|verify <kit dir>");
        std::process::exit(2);
    }
    let kit = PathBuf::from(&args[2]);
    let ok = if args[1] == "repr" { cmd_repr(&kit) } else { cmd_verify(&kit) };
    std::process::exit(if ok { 0 } else { 1 });
}
