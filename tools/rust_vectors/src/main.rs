//! Emits tests/golden/ref_{msm,ntt,proof,proof_sparse,params}.json from the reference's own dependencies (arkworks 0.3,
//! jf-plonk @ bcd92b2) so that this repository's oracle and HIP path can be pinned against them.
//!
//! UNBUILT SOURCE: there is no Rust toolchain where this repository is developed.  API names that could not be checked
//! against the crates' sources are marked [DEP-RECALLED]; if one of them does not compile, adjust the call - the JSON
//! layout (consumed by tests/test_ref_vectors.py) is what matters.
//!
//! Inputs are re-derived exactly as tests/golden/make_golden.py derives them:
//!   SplitMix64(seed) -> 4 x u64 little-endian -> 256-bit integer -> mod r.
//! Field elements are written as 64 hex digits, big-endian, canonical (not Montgomery); G1 points as [x, y] or null.
use ark_bn254::{Bn254, Fq, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::{msm::VariableBaseMSM, AffineCurve, PairingEngine, ProjectiveCurve};
use ark_ff::{BigInteger, FftField, Field, One, PrimeField, Zero};
use ark_poly::{univariate::DensePolynomial, EvaluationDomain, Radix2EvaluationDomain};
use ark_serialize::CanonicalSerialize;
use jf_plonk::{
    proof_system::{structs::UniversalSrs, PlonkKzgSnark, UniversalSNARK},
    transcript::SolidityTranscript,
};
use jf_relation::{Arithmetization, Circuit, PlonkCircuit};
use serde_json::{json, Value};
use std::collections::BTreeMap;

// ---- portable PRNG shared with oracle/bn254.py and cap_amd/bench_utils.py -----------------------------------------
struct SplitMix64(u64);
impl SplitMix64 {
    fn next(&mut self) -> u64 {
        self.0 = self.0.wrapping_add(0x9E3779B97F4A7C15);
        let mut z = self.0;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
        z ^ (z >> 31)
    }
    fn field(&mut self) -> Fr {
        let mut bytes = [0u8; 32];
        for i in 0..4 {
            bytes[8 * i..8 * i + 8].copy_from_slice(&self.next().to_le_bytes());
        }
        Fr::from_le_bytes_mod_order(&bytes)
    }
}

fn hx<F: PrimeField>(v: &F) -> String {
    hex::encode(v.into_repr().to_bytes_be())
}
fn pt(p: &G1Affine) -> Value {
    if p.is_zero() {
        Value::Null
    } else {
        json!([hx(&p.x), hx(&p.y)])
    }
}
fn ser<T: CanonicalSerialize>(t: &T) -> String {
    let mut buf = Vec::new();
    t.serialize(&mut buf).unwrap();
    hex::encode(buf)
}

// ---- MSM: ark_ec::msm::VariableBaseMSM on the seeds of tests/golden/msm.json ------------------------------------
fn msm_vectors() -> Value {
    let g = G1Affine::prime_subgroup_generator();
    let mut out = vec![];
    for &n in &[1usize, 2, 3, 31, 32, 33, 100, 1000, 4099] {
        let mut ks = SplitMix64(1000 + n as u64);
        let bases: Vec<G1Affine> = (0..n).map(|_| g.mul(ks.field().into_repr()).into_affine()).collect();
        let mut sc = SplitMix64(2000 + n as u64);
        let scalars: Vec<_> = (0..n).map(|_| sc.field().into_repr()).collect();
        let r = VariableBaseMSM::multi_scalar_mul(&bases, &scalars).into_affine();
        out.push(json!({"n": n, "base_seed": 1000 + n, "scalar_seed": 2000 + n, "edge": false, "result": pt(&r)}));
    }
    Value::Array(out)
}

// ---- NTT: ark_poly::Radix2EvaluationDomain on the seeds of tests/golden/ntt.json --------------------------------
fn ntt_vectors() -> Value {
    let mut out = vec![];
    for log_n in 0..=12u32 {
        let n = 1usize << log_n;
        let mut rng = SplitMix64(3000 + log_n as u64);
        let a: Vec<Fr> = (0..n).map(|_| rng.field()).collect();
        let d = Radix2EvaluationDomain::<Fr>::new(n).unwrap();
        let l = |v: Vec<Fr>| -> Vec<String> { v.iter().map(hx).collect() };
        out.push(json!({"log_n": log_n, "seed": 3000 + log_n,
            "ntt": l(d.fft(&a)), "intt": l(d.ifft(&a)), "coset_ntt": l(d.coset_fft(&a)), "coset_intt": l(d.coset_ifft(&a))}));
    }
    Value::Array(out)
}

// ---- an RNG that hands the prover a scripted sequence of field elements ------------------------------------------
// ark-ff 0.3 `Fp256::rand` draws 4 u64, masks the top bits and keeps the value if it is below the modulus, reading it
// as the Montgomery representation.  Feeding the Montgomery limbs of b therefore makes `Fr::rand` return b.
struct ScriptedRng {
    words: Vec<u64>,
    pos: usize,
    draws: usize,
}
impl ScriptedRng {
    fn from_fields(vals: &[Fr]) -> Self {
        let mut words = vec![];
        for v in vals {
            words.extend_from_slice(&v.0 .0); // Fp256(BigInteger256([u64; 4])): the Montgomery limbs
        }
        ScriptedRng { words, pos: 0, draws: 0 }
    }
}
impl rand_core::RngCore for ScriptedRng {
    fn next_u32(&mut self) -> u32 {
        self.next_u64() as u32
    }
    fn next_u64(&mut self) -> u64 {
        self.draws += 1;
        let w = if self.pos < self.words.len() { self.words[self.pos] } else { 0 };
        self.pos += 1;
        w
    }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        for chunk in dest.chunks_mut(8) {
            let w = self.next_u64().to_le_bytes();
            chunk.copy_from_slice(&w[..chunk.len()]);
        }
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand_core::Error> {
        self.fill_bytes(dest);
        Ok(())
    }
}
impl rand_core::CryptoRng for ScriptedRng {}

// SRS = powers of a known tau, built field by field (UniversalParams' fields are public in ark-poly-commit @ cafc05e)
fn known_tau_srs(tau: Fr, max_degree: usize) -> UniversalSrs<Bn254> {
    let g = G1Affine::prime_subgroup_generator();
    let h = G2Affine::prime_subgroup_generator();
    let mut powers = Vec::with_capacity(max_degree + 1);
    let mut x = Fr::one();
    for _ in 0..=max_degree {
        powers.push(g.mul(x.into_repr()).into_affine());
        x *= tau;
    }
    let beta_h = h.mul(tau.into_repr()).into_affine();
    // hiding powers [gamma tau^i] G for degrees 0 ..= max_degree + 1, as KZG10::setup fills the map: jf-plonk's
    // preprocess trims them into CommitKey::powers_of_gamma_g by degree (an empty map would panic there), and the
    // consumer checks that capgpu_plonk_key_serialize emits exactly that vector for a key made under the loaded SRS
    let gamma = SplitMix64(0x6A).field();
    let mut powers_of_gamma_g = BTreeMap::new();
    let mut y = gamma;
    for i in 0..=(max_degree + 1) {
        powers_of_gamma_g.insert(i, g.mul(y.into_repr()).into_affine());
        y *= tau;
    }
    // [DEP-RECALLED] field names of ark_poly_commit::kzg10::UniversalParams
    UniversalSrs::<Bn254> {
        powers_of_g: powers,
        powers_of_gamma_g,
        h,
        beta_h,
        neg_powers_of_h: BTreeMap::new(),
        prepared_h: h.into(),
        prepared_beta_h: beta_h.into(),
    }
}

// A small TurboPlonk circuit through jf-relation's public builder: every selector kind of the CAP circuits appears
// (linear combination, multiplication, the x^5 of Rescue, constants, public inputs, the boolean / ecc style products).
fn build_circuit(seed: u64) -> (PlonkCircuit<Fr>, Vec<Fr>) {
    let mut rng = SplitMix64(seed);
    let mut c = PlonkCircuit::<Fr>::new_turbo_plonk();
    let pubs: Vec<Fr> = (0..3).map(|_| rng.field()).collect();
    let pv: Vec<_> = pubs.iter().map(|p| c.create_public_variable(*p).unwrap()).collect();
    let mut acc = pv[0];
    for i in 0..6 {
        let w = c.create_variable(rng.field()).unwrap();
        let s = c.add(acc, w).unwrap();
        let m = c.mul(s, pv[i % 3]).unwrap();
        // [DEP-RECALLED] x^5 gadget used by the Rescue permutation; drop these two lines if the name differs
        let p5 = c.power_5_gen(m).unwrap();
        acc = c.sub(p5, w).unwrap();
        let b = c.create_bool_variable(i % 2 == 0).unwrap(); // [DEP-RECALLED] returns BoolVar
        let _ = c.mul(acc, b.into()).unwrap();
    }
    let k = c.create_constant_variable(Fr::from(7u64)).unwrap();
    let _ = c.mul_add(&[acc, k, pv[1], pv[2]], &[Fr::from(3u64), Fr::from(5u64)]).unwrap(); // q_mul0 / q_mul1 gate
    c.finalize_for_arithmetization().unwrap();
    (c, pubs)
}

// A CAP-SHAPED instance (round 6): what the witness of a real note looks like - mostly zeros (padding up to the domain),
// booleans (bit decompositions of amounts and scalars), a few 64-bit values and only some full-width cells
// (src/circuit/transfer.rs:53-193).  The device prover commits to such wire columns through their VALUES on the
// Lagrange-form key (capgpu_plonk_set_wire_commit): this vector pins that path against jellyfish's coefficient-form
// commitments.  64 range-check bits + their recomposition, a boolean-selected sum, many padding gates.
fn build_sparse_circuit(seed: u64) -> (PlonkCircuit<Fr>, Vec<Fr>) {
    let mut rng = SplitMix64(seed);
    let mut c = PlonkCircuit::<Fr>::new_turbo_plonk();
    let amount: u64 = rng.next() >> 3;
    let pubs: Vec<Fr> = vec![Fr::from(amount), rng.field()];
    let pv: Vec<_> = pubs.iter().map(|p| c.create_public_variable(*p).unwrap()).collect();
    // amount = sum 2^i b_i, every b_i a boolean variable (boolean gates: q_mul0 = 1, q_o = 1 on (b, b, 0, 0, b))
    let mut acc = c.zero();
    let mut pow = Fr::from(1u64);
    for i in 0..61 {
        let b = c.create_bool_variable((amount >> i) & 1 == 1).unwrap(); // [DEP-RECALLED]
        let t = c.mul_constant(b.into(), &pow).unwrap(); // [DEP-RECALLED] q_lc with a constant coefficient
        acc = c.add(acc, t).unwrap();
        pow = pow + pow;
    }
    c.equal_gate(acc, pv[0]).unwrap(); // [DEP-RECALLED] enforce_equal in later versions
    // one full-width corner: x^5 of the second public input, selected by a bit
    let sel = c.create_bool_variable(true).unwrap();
    let p5 = c.power_5_gen(pv[1]).unwrap();
    let _ = c.mul(p5, sel.into()).unwrap();
    // zero padding well past the gates above: 300+ rows of the 512-row domain hold nothing but zeros
    for _ in 0..180 {
        let z = c.create_variable(Fr::from(0u64)).unwrap();
        let _ = c.add(z, c.zero()).unwrap();
    }
    c.finalize_for_arithmetization().unwrap();
    (c, pubs)
}

fn proof_and_params_vectors() -> (Value, Value, Value) {
    let tau = SplitMix64(0xCA9).field();
    let (dense, _) = proof_vector(tau, build_circuit(77), 200, true);
    let (sparse, _) = proof_vector(tau, build_sparse_circuit(78), 201, false);
    let (_, params) = proof_vector(tau, build_circuit(77), 200, true);
    (dense, sparse, params.unwrap())
}

fn proof_vector(tau: Fr, built: (PlonkCircuit<Fr>, Vec<Fr>), blinder_seed: u64, with_params: bool) -> (Value, Option<Value>) {
    let (circuit, pubs) = built;
    let n = circuit.eval_domain_size().unwrap();
    let domain = Radix2EvaluationDomain::<Fr>::new(n).unwrap();
    let srs = known_tau_srs(tau, circuit.srs_size().unwrap());
    let (pk, vk) = PlonkKzgSnark::<Bn254>::preprocess(&srs, &circuit).unwrap();

    // the circuit as tables, in the order the prover consumes them: this is the INPUT the other implementations get
    let evals = |p: &DensePolynomial<Fr>| -> Vec<String> { domain.fft(&p.coeffs).iter().map(hx).collect() };
    let selectors: Vec<Vec<String>> = circuit.compute_selector_polynomials().unwrap().iter().map(evals).collect();
    let sigmas: Vec<Vec<String>> =
        circuit.compute_extended_permutation_polynomials().unwrap().iter().map(evals).collect();
    let wires: Vec<Vec<String>> = circuit.compute_wire_polynomials().unwrap().iter().map(evals).collect();

    // blinders in draw order: 2 per wire polynomial, then 3 for the permutation product
    let mut b = SplitMix64(blinder_seed);
    let blinders: Vec<Fr> = (0..13).map(|_| b.field()).collect();
    let mut rng = ScriptedRng::from_fields(&blinders);
    let ext_msg = b"memo-key".to_vec();
    let proof =
        PlonkKzgSnark::<Bn254>::prove::<_, _, SolidityTranscript>(&mut rng, &circuit, &pk, Some(ext_msg.clone())).unwrap();
    PlonkKzgSnark::<Bn254>::verify::<SolidityTranscript>(&vk, &pubs, &proof, Some(ext_msg.clone())).unwrap();

    // [DEP-RECALLED] field names of jf_plonk::proof_system::structs::{Proof, ProofEvaluations, VerifyingKey}
    let c2v = |v: &[ark_poly_commit::kzg10::Commitment<Bn254>]| -> Vec<Value> { v.iter().map(|c| pt(&c.0)).collect() };
    let proof_json = json!({
        "source": "jf-plonk @ bcd92b2 / arkworks 0.3 (tools/rust_vectors)",
        "log_n": n.trailing_zeros(), "num_inputs": pubs.len(), "tau": hx(&tau), "ext_msg": hex::encode(&ext_msg),
        "blinders": blinders.iter().map(hx).collect::<Vec<_>>(), "rng_u64_draws": rng.draws,
        "selectors": selectors, "sigma": sigmas, "wires": wires, "pub_inputs": pubs.iter().map(hx).collect::<Vec<_>>(),
        "k": vk.k.iter().map(hx).collect::<Vec<_>>(),
        "selector_comms": c2v(&vk.selector_comms), "sigma_comms": c2v(&vk.sigma_comms),
        "wires_poly_comms": c2v(&proof.wires_poly_comms), "prod_perm_poly_comm": pt(&proof.prod_perm_poly_comm.0),
        "split_quot_poly_comms": c2v(&proof.split_quot_poly_comms),
        "opening_proof": pt(&proof.opening_proof.0), "shifted_opening_proof": pt(&proof.shifted_opening_proof.0),
        "wires_evals": proof.poly_evals.wires_evals.iter().map(hx).collect::<Vec<_>>(),
        "wire_sigma_evals": proof.poly_evals.wire_sigma_evals.iter().map(hx).collect::<Vec<_>>(),
        "perm_next_eval": hx(&proof.poly_evals.perm_next_eval),
        "proof_bytes": ser(&proof),
    });
    if !with_params {
        return (proof_json, None);
    }
    let params_json = json!({
        "source": "jf-plonk @ bcd92b2 / ark-serialize 0.3 (tools/rust_vectors)",
        "log_n": n.trailing_zeros(), "num_inputs": pubs.len(), "tau": hx(&tau),
        "srs": ser(&srs), "vk": ser(&vk), "proving_key": ser(&pk),
        "g1_generator_compressed": ser(&G1Affine::prime_subgroup_generator()),
        "g2_generator_compressed": ser(&G2Affine::prime_subgroup_generator()),
    });
    let _ = (G1Projective::zero(), G2Projective::zero(), Fq::zero(), Fr::multiplicative_generator(), <Bn254 as PairingEngine>::Fqk::one());
    (proof_json, Some(params_json))
}

fn main() {
    let dir = std::env::args().nth(1).unwrap_or_else(|| "../../tests/golden".to_string());
    let write = |name: &str, v: &Value| {
        let path = format!("{}/{}", dir, name);
        std::fs::write(&path, serde_json::to_string(v).unwrap()).unwrap();
        println!("wrote {}", path);
    };
    write("ref_msm.json", &msm_vectors());
    write("ref_ntt.json", &ntt_vectors());
    let (proof, sparse, params) = proof_and_params_vectors();
    write("ref_proof.json", &proof);
    write("ref_proof_sparse.json", &sparse);
    write("ref_params.json", &params);
}
