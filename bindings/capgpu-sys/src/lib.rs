//! Raw bindings of `include/capgpu.h` (libcapgpu.so, the MI355X PLONK prover) and the thin safe layer above them.
//!
//! UNBUILT in this repository (no Rust toolchain there); see Cargo.toml.  The `extern "C"` block mirrors the header
//! symbol for symbol - integers little-endian, `Fr`/`Fq` = `[u64; 4]` (arkworks `BigInteger256` limb order, Montgomery
//! form where the header says so), G1 affine = 8 words, G1 Jacobian = 12 words, G2 affine = 16 words.  Every function
//! returns `CAPGPU_OK` (0) or a negative `CAPGPU_ERR_*`; nothing unwinds across the boundary.
//!
//! Replaces, in the reference (EspressoSystems/cap):
//!   `PlonkKzgSnark::preprocess` at src/proof/transfer.rs:133, mint.rs:76, freeze.rs:102   -> `ProvingKey::preprocess`
//!   `PlonkKzgSnark::prove`      at src/proof/transfer.rs:181-186, mint.rs:113, freeze.rs:151 -> `ProvingKey::prove`
//!   `VariableBaseMSM::multi_scalar_mul` (ark-ec 0.3.0, via KZG10::commit)                  -> `Srs::msm`
//!   `Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place` (ark-poly 0.3.0)    -> `ntt_in_place`
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const CAPGPU_OK: c_int = 0;
pub const CAPGPU_ERR_INVALID_ARG: c_int = -1;
pub const CAPGPU_ERR_NO_DEVICE: c_int = -2;
pub const CAPGPU_ERR_HIP: c_int = -3;
pub const CAPGPU_ERR_BAD_HANDLE: c_int = -4;
pub const CAPGPU_ERR_OOM: c_int = -5;
pub const CAPGPU_ERR_NOT_INITIALISED: c_int = -6;
pub const CAPGPU_ERR_PROOF: c_int = -7;
pub const CAPGPU_ERR_SERIALIZATION: c_int = -8;
pub const CAPGPU_ERR_COMM: c_int = -9;

pub const NUM_WIRE_TYPES: usize = 5;
pub const NUM_SELECTORS: usize = 13;

/// `input_form` of the `_ex` PLONK entry points: values on the domain, or polynomials in coefficient form - what
/// jf-relation's `Arithmetization` trait returns, passed through without a CPU transform.
pub const CAPGPU_INPUT_EVALS: c_int = 0;
pub const CAPGPU_INPUT_COEFFS: c_int = 1;

/// `capgpu_proof`: the fields of `jf_plonk::proof_system::structs::Proof`, in order (plookup_proof = None).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct capgpu_proof {
    pub wires_poly_comms: [[u64; 8]; NUM_WIRE_TYPES],
    pub prod_perm_poly_comm: [u64; 8],
    pub split_quot_poly_comms: [[u64; 8]; NUM_WIRE_TYPES],
    pub opening_proof: [u64; 8],
    pub shifted_opening_proof: [u64; 8],
    pub wires_evals: [[u64; 4]; NUM_WIRE_TYPES],
    pub wire_sigma_evals: [[u64; 4]; NUM_WIRE_TYPES - 1],
    pub perm_next_eval: [u64; 4],
}

/// `capgpu_verifying_key`: `jf_plonk::proof_system::structs::VerifyingKey` without the open key.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct capgpu_verifying_key {
    pub domain_size: u64,
    pub num_inputs: u64,
    pub k: [[u64; 4]; NUM_WIRE_TYPES],
    pub selector_comms: [[u64; 8]; NUM_SELECTORS],
    pub sigma_comms: [[u64; 8]; NUM_WIRE_TYPES],
}

#[link(name = "capgpu")]
extern "C" {
    // ---- lifecycle, devices
    pub fn capgpu_init(device_ids: *const c_int, n_devices: c_int) -> c_int;
    pub fn capgpu_shutdown();
    pub fn capgpu_last_error() -> *const c_char;
    pub fn capgpu_version() -> *const c_char;
    pub fn capgpu_device_count(count_out: *mut c_int) -> c_int;
    pub fn capgpu_set_device(slot: c_int) -> c_int;
    pub fn capgpu_get_device(slot_out: *mut c_int, hip_device_out: *mut c_int) -> c_int;
    pub fn capgpu_device_info(name_out: *mut c_char, cu_count_out: *mut c_int, hbm_bytes_out: *mut u64) -> c_int;
    pub fn capgpu_device_peer_info(slot_a: c_int, slot_b: c_int, access_out: *mut c_int) -> c_int;
    pub fn capgpu_mem_info(free_bytes_out: *mut u64, total_bytes_out: *mut u64) -> c_int;
    // ---- footprint control and diagnostics
    pub fn capgpu_trim(bytes_released_out: *mut u64, contexts_busy_out: *mut c_int) -> c_int;
    pub fn capgpu_set_memory_limit(scratch_bytes_per_device: u64) -> c_int;
    pub fn capgpu_scratch_info(scratch_bytes_out: *mut u64, limit_out: *mut u64) -> c_int;
    pub fn capgpu_trace_enable(on: c_int) -> c_int;
    pub fn capgpu_trace_dump(path: *const c_char, events_out: *mut u64) -> c_int;
    // ---- device memory / stream
    pub fn capgpu_malloc(dev_ptr_out: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn capgpu_free(dev_ptr: *mut c_void) -> c_int;
    pub fn capgpu_memcpy_h2d(dev_dst: *mut c_void, host_src: *const c_void, bytes: usize) -> c_int;
    pub fn capgpu_memcpy_d2h(host_dst: *mut c_void, dev_src: *const c_void, bytes: usize) -> c_int;
    pub fn capgpu_sync() -> c_int;
    pub fn capgpu_sync_all() -> c_int;
    pub fn capgpu_runtime_info(hip_runtime_version_out: *mut c_int, hip_driver_version_out: *mut c_int) -> c_int;
    pub fn capgpu_timer_begin() -> c_int;
    pub fn capgpu_timer_end(ms_out: *mut f64) -> c_int;
    pub fn capgpu_context_count(count_out: *mut c_int) -> c_int;
    pub fn capgpu_physical_device_count(count_out: *mut c_int) -> c_int;
    pub fn capgpu_set_stream(hip_stream: *mut c_void) -> c_int;
    // ---- SRS
    pub fn capgpu_srs_upload(bases: *const c_void, n: usize, stride_bytes: usize, coords_montgomery: c_int,
                             handle_out: *mut u64) -> c_int;
    pub fn capgpu_srs_generate(tau: *const u64, n: usize, handle_out: *mut u64) -> c_int;
    pub fn capgpu_srs_generate_hiding(tau: *const u64, gamma: *const u64, n: usize, handle_out: *mut u64) -> c_int;
    pub fn capgpu_srs_generate_affine_seq(a: *const u64, b: *const u64, n: usize, handle_out: *mut u64) -> c_int;
    pub fn capgpu_srs_size(handle: u64, n_out: *mut usize) -> c_int;
    pub fn capgpu_srs_shards(handle: u64, shards_out: *mut c_int) -> c_int;
    pub fn capgpu_srs_download(handle: u64, offset: usize, n: usize, out: *mut c_void) -> c_int;
    pub fn capgpu_srs_free(handle: u64) -> c_int;
    // ---- MSM
    pub fn capgpu_msm_g1(srs_handle: u64, offset: usize, scalars: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn capgpu_msm_g1_lagrange(
        srs_handle: u64,
        log_n: u32,
        scalars: *const u64,
        count: usize,
        scalars_montgomery: c_int,
        out_xyz: *mut u64,
    ) -> c_int;
    pub fn capgpu_msm_g1_batch(srs_handle: u64, offsets: *const usize, scalars: *const *const u64, ns: *const usize,
                               count: c_int, out_xyz: *mut u64) -> c_int;
    pub fn capgpu_msm_g1_dev(srs_handle: u64, offset: usize, d_scalars: *const c_void, scalar_stride: usize, n: usize,
                             count: c_int, scalars_montgomery: c_int, d_out_xyz: *mut c_void) -> c_int;
    pub fn capgpu_msm_scalars_upload(srs_handle: u64, offset: usize, scalars: *const u64, scalar_stride: usize, n: usize,
                                     count: c_int, scalars_handle_out: *mut u64) -> c_int;
    pub fn capgpu_msm_scalars_scatter_dev(srs_handle: u64, offset: usize, d_scalars: *const c_void, scalar_stride: usize,
                                          n: usize, count: c_int, scalars_handle_out: *mut u64) -> c_int;
    pub fn capgpu_msm_scalars_free(scalars_handle: u64) -> c_int;
    pub fn capgpu_msm_g1_resident(srs_handle: u64, scalars_handle: u64, scalars_montgomery: c_int,
                                  d_out_xyz: *mut c_void) -> c_int;
    pub fn capgpu_msm_shard_stats(scalar_bytes_out: *mut u64, partial_bytes_out: *mut u64, calls_out: *mut u64,
                                  replications_out: *mut u64) -> c_int;
    pub fn capgpu_msm_plan(srs_handle: u64, n: usize, count: c_int, buf: *mut c_char, cap: usize) -> c_int;
    pub fn capgpu_g1_sum(points_xyz: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    // ---- one process per GPU: the RCCL exchange
    pub fn capgpu_comm_unique_id(id_out: *mut u8) -> c_int;
    pub fn capgpu_comm_init(rank: c_int, world: c_int, id: *const u8) -> c_int;
    pub fn capgpu_comm_init_loopback(world: c_int) -> c_int;
    pub fn capgpu_comm_loopback_set_rank(rank: c_int) -> c_int;
    pub fn capgpu_comm_destroy() -> c_int;
    pub fn capgpu_comm_info(rank_out: *mut c_int, world_out: *mut c_int) -> c_int;
    pub fn capgpu_msm_g1_sharded_dev(srs_handle: u64, offset: usize, d_scalars: *const c_void, scalar_stride: usize,
                                     n_local: usize, count: c_int, scalars_montgomery: c_int,
                                     d_out_xyz: *mut c_void) -> c_int;
    pub fn capgpu_msm_g1_sharded(srs_handle: u64, offset: usize, scalars: *const u64, n_local: usize,
                                 out_xyz: *mut u64) -> c_int;
    pub fn capgpu_plonk_shard_msm(on: c_int) -> c_int;
    // ---- NTT
    pub fn capgpu_ntt_fr(data: *mut u64, log_n: u32, dir: c_int, coset: c_int) -> c_int;
    pub fn capgpu_ntt_fr_batch(data: *const *mut u64, count: c_int, log_n: u32, dir: c_int, coset: c_int) -> c_int;
    pub fn capgpu_ntt_fr_dev(d_data: *mut c_void, stride_elems: usize, count: c_int, log_n: u32, dir: c_int,
                             coset: c_int) -> c_int;
    // ---- PLONK
    pub fn capgpu_plonk_preprocess(srs_handle: u64, n: usize, num_inputs: usize, selectors: *const u64,
                                   sigma_evals: *const u64, pk_handle_out: *mut u64,
                                   vk_out: *mut capgpu_verifying_key) -> c_int;
    pub fn capgpu_plonk_preprocess_ex(srs_handle: u64, n: usize, num_inputs: usize, selectors: *const u64,
                                      sigmas: *const u64, input_form: c_int, pk_handle_out: *mut u64,
                                      vk_out: *mut capgpu_verifying_key) -> c_int;
    pub fn capgpu_plonk_free_key(pk_handle: u64) -> c_int;
    pub fn capgpu_plonk_key_info(pk_handle: u64, domain_size_out: *mut usize, num_inputs_out: *mut usize,
                                 srs_handle_out: *mut u64) -> c_int;
    pub fn capgpu_plonk_prove(pk_handle: u64, wires: *const u64, pub_inputs: *const u64, num_inputs: usize,
                              ext_msg: *const u8, ext_msg_len: usize, blinders: *const u64,
                              proof_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_set_coalescing(window_us: u32, max_batch: u32) -> c_int;
    pub fn capgpu_plonk_set_wire_commit(mode: c_int) -> c_int;
    pub fn capgpu_plonk_graph_stats(segments_captured_out: *mut u64, segments_replayed_out: *mut u64) -> c_int;
    pub fn capgpu_plonk_coalescing_stats(batches_out: *mut u64, proofs_out: *mut u64) -> c_int;
    pub fn capgpu_plonk_prove_batch(pk_handle: u64, count: c_int, wires: *const u64, pub_inputs: *const u64,
                                    num_inputs: usize, ext_msg: *const u8, ext_msg_len: usize, blinders: *const u64,
                                    proofs_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_prove_multi(pk_handles: *const u64, count: c_int, wires: *const u64, pub_inputs: *const u64,
                                    num_inputs: usize, ext_msgs: *const *const u8, ext_msg_lens: *const usize,
                                    blinders: *const u64, proofs_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_prove_multi_dev(pk_handles: *const u64, count: c_int, d_wires: *const c_void,
                                        pub_inputs: *const u64, num_inputs: usize, ext_msgs: *const *const u8,
                                        ext_msg_lens: *const usize, blinders: *const u64,
                                        proofs_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_prove_batch_dev(pk_handle: u64, count: c_int, d_wires: *const c_void, pub_inputs: *const u64,
                                        num_inputs: usize, ext_msg: *const u8, ext_msg_len: usize,
                                        blinders: *const u64, proofs_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_prove_ex(pk_handle: u64, wires: *const u64, pub_inputs: *const u64, num_inputs: usize,
                                 ext_msg: *const u8, ext_msg_len: usize, blinders: *const u64, input_form: c_int,
                                 proof_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_prove_batch_ex(pk_handle: u64, count: c_int, wires: *const u64, pub_inputs: *const u64,
                                       num_inputs: usize, ext_msg: *const u8, ext_msg_len: usize,
                                       blinders: *const u64, input_form: c_int, proofs_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_prove_multi_ex(pk_handles: *const u64, count: c_int, wires: *const u64,
                                       pub_inputs: *const u64, num_inputs: usize, ext_msgs: *const *const u8,
                                       ext_msg_lens: *const usize, blinders: *const u64, input_form: c_int,
                                       proofs_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_prove_multi_dev_ex(pk_handles: *const u64, count: c_int, d_wires: *const c_void,
                                           pub_inputs: *const u64, num_inputs: usize, ext_msgs: *const *const u8,
                                           ext_msg_lens: *const usize, blinders: *const u64, input_form: c_int,
                                           proofs_out: *mut capgpu_proof) -> c_int;
    pub fn capgpu_plonk_prove_batch_dev_ex(pk_handle: u64, count: c_int, d_wires: *const c_void,
                                           pub_inputs: *const u64, num_inputs: usize, ext_msg: *const u8,
                                           ext_msg_len: usize, blinders: *const u64, input_form: c_int,
                                           proofs_out: *mut capgpu_proof) -> c_int;
    // ---- verification (host only)
    pub fn capgpu_g2_generator(out: *mut u64) -> c_int;
    pub fn capgpu_g2_mul(q: *const u64, scalar: *const u64, out: *mut u64) -> c_int;
    pub fn capgpu_pairing_check(g1_points: *const u64, g2_points: *const u64, n: usize, ok_out: *mut c_int) -> c_int;
    pub fn capgpu_plonk_verify(vk: *const capgpu_verifying_key, g2_h: *const u64, g2_beta_h: *const u64,
                               pub_inputs: *const u64, num_inputs: usize, proof: *const capgpu_proof,
                               ext_msg: *const u8, ext_msg_len: usize, ok_out: *mut c_int) -> c_int;
    pub fn capgpu_plonk_batch_verify(vks: *const *const capgpu_verifying_key, g2_h: *const u64, g2_beta_h: *const u64,
                                     pub_inputs: *const *const u64, num_inputs: *const usize,
                                     proofs: *const *const capgpu_proof, ext_msgs: *const *const u8,
                                     ext_msg_lens: *const usize, count: usize, ok_out: *mut c_int) -> c_int;
    pub fn capgpu_plonk_batch_verify_dev(vks: *const *const capgpu_verifying_key, g2_h: *const u64,
                                         g2_beta_h: *const u64, pub_inputs: *const *const u64,
                                         num_inputs: *const usize, proofs: *const *const capgpu_proof,
                                         ext_msgs: *const *const u8, ext_msg_lens: *const usize, count: usize,
                                         ok_out: *mut c_int) -> c_int;
    pub fn capgpu_proof_serialize(proof: *const capgpu_proof, out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn capgpu_proof_deserialize(bytes: *const u8, len: usize, proof_out: *mut capgpu_proof,
                                    consumed_out: *mut usize) -> c_int;
    // ---- on-disk parameter formats
    pub fn capgpu_g1_decompress(input: *const u8, n: usize, out_xy: *mut u64) -> c_int;
    pub fn capgpu_g1_compress(xy: *const u64, n: usize, out: *mut u8) -> c_int;
    pub fn capgpu_srs_deserialize(bytes: *const u8, len: usize, max_degree: usize, handle_out: *mut u64,
                                  h_out: *mut u64, beta_h_out: *mut u64, consumed_out: *mut usize) -> c_int;
    pub fn capgpu_srs_serialize(handle: u64, h: *const u64, beta_h: *const u64, out: *mut u8, cap: usize,
                                len_out: *mut usize) -> c_int;
    pub fn capgpu_plonk_vk_serialize(vk: *const capgpu_verifying_key, g: *const u64, gamma_g: *const u64,
                                     h: *const u64, beta_h: *const u64, out: *mut u8, cap: usize,
                                     len_out: *mut usize) -> c_int;
    pub fn capgpu_plonk_vk_deserialize(bytes: *const u8, len: usize, vk_out: *mut capgpu_verifying_key,
                                       g_out: *mut u64, gamma_g_out: *mut u64, h_out: *mut u64, beta_h_out: *mut u64,
                                       consumed_out: *mut usize) -> c_int;
    pub fn capgpu_plonk_key_serialize(pk_handle: u64, gamma_g: *const u64, h: *const u64, beta_h: *const u64,
                                      out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn capgpu_plonk_key_deserialize(bytes: *const u8, len: usize, srs_handle_out: *mut u64,
                                        pk_handle_out: *mut u64, vk_out: *mut capgpu_verifying_key, h_out: *mut u64,
                                        beta_h_out: *mut u64, consumed_out: *mut usize) -> c_int;
    // ---- instrumentation
    pub fn capgpu_ubench_mad_rate(lane_ops_per_s_out: *mut f64) -> c_int;
    pub fn capgpu_ubench_issue_rates(rates_out: *mut f64, count: c_int) -> c_int;
    pub fn capgpu_profile_enable(on: c_int) -> c_int;
    pub fn capgpu_profile_reset() -> c_int;
    pub fn capgpu_profile_get(name: *const c_char, total_ms_out: *mut f64, launches_out: *mut u64) -> c_int;
    pub fn capgpu_profile_dump(buf: *mut c_char, cap: usize) -> c_int;
}

// ---------------------------------------------------------------------------------------------------------------
// Safe layer.  Error = (code, thread-local message); the caller maps it to `PlonkError`, which the reference's
// prove() already maps to `TxnApiError::FailedSnark(String)` (src/proof/transfer.rs:187), and
// `CAPGPU_ERR_SERIALIZATION` to `TxnApiError::DeserializationError` (src/errors.rs:81-90).
// ---------------------------------------------------------------------------------------------------------------
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "capgpu error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}
pub type Result<T> = std::result::Result<T, Error>;

pub fn check(rc: c_int) -> Result<()> {
    if rc == CAPGPU_OK {
        return Ok(());
    }
    let message = unsafe { std::ffi::CStr::from_ptr(capgpu_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

/// Binds the process to the listed HIP devices (idempotent).  With more than one, the library deals host-buffer
/// batches and coalesced `prove` calls over them by itself: the rayon loop of
/// `TxnsParams::generate_txns` (src/utils/params_builder.rs:194-226) needs no change to use every GPU of the node.
pub fn init(devices: &[i32]) -> Result<()> {
    check(unsafe { capgpu_init(if devices.is_empty() { std::ptr::null() } else { devices.as_ptr() }, devices.len() as c_int) })
}

/// Gather concurrent single-proof calls into device batches (what a rayon `par_iter` over notes produces).
pub fn set_coalescing(window_us: u32, max_batch: u32) -> Result<()> {
    check(unsafe { capgpu_plonk_set_coalescing(window_us, max_batch) })
}

/// How the wire commitments are computed: `true` (the default) from the witness VALUES on the Lagrange-form key - round 1's
/// time then depends on the witness's sparsity -, `false` from coefficients, as jf-plonk does (witness-independent work;
/// the mode for deployments where an adversary can time proofs: include/capgpu.h, "TIMING AND THE SECRET WITNESS").
pub fn set_wire_commit_from_evals(on: bool) -> Result<()> {
    check(unsafe { capgpu_plonk_set_wire_commit(if on { 1 } else { 0 }) })
}

/// Releases the workspace of every idle device context (scratch, pinned result areas, captured launch graphs): tables -
/// SRS, proving keys, NTT domains - stay.  Returns the device bytes given back.  A process that shares the GPU calls this
/// when it goes idle; the next proof allocates again.
pub fn trim() -> Result<u64> {
    let mut released = 0u64;
    let mut busy: c_int = 0;
    check(unsafe { capgpu_trim(&mut released, &mut busy) })?;
    Ok(released)
}

/// Caps the workspace the library holds per device (0 = no cap).  A `prove` that would grow past it fails with
/// `CAPGPU_ERR_OOM` (after trimming the device's idle contexts): prove in smaller batches, or raise the cap.
pub fn set_memory_limit(scratch_bytes_per_device: u64) -> Result<()> {
    check(unsafe { capgpu_set_memory_limit(scratch_bytes_per_device) })
}

/// A device-resident commit key (`UniversalSrs::powers_of_g` / `CommitKey::powers_of_g`).
pub struct Srs {
    handle: u64,
}
impl Srs {
    /// `bases`: arkworks `GroupAffine` values as they sit in memory.  `stride` = `size_of::<G1Affine>()` (72 with the
    /// `infinity` flag byte at offset 64; `GroupAffine` is `repr(Rust)` - assert the offsets in the caller's tests) or
    /// 64 for packed (x, y) pairs.
    ///
    /// # Safety
    /// `bases` must point to `n * stride` readable bytes.
    pub unsafe fn upload_raw(bases: *const u8, n: usize, stride: usize) -> Result<Srs> {
        let mut handle = 0u64;
        check(capgpu_srs_upload(bases as *const c_void, n, stride, 1, &mut handle))?;
        Ok(Srs { handle })
    }
    /// The bytes of an ark-serialized `UniversalSrs` (`load_srs`, src/proof/mod.rs:106; `load_universal_parameter`,
    /// src/parameters.rs:97-109).  Returns the SRS and the open key's (h, beta_h).
    pub fn deserialize(bytes: &[u8], max_degree: usize) -> Result<(Srs, [u64; 16], [u64; 16])> {
        let (mut handle, mut used) = (0u64, 0usize);
        let (mut h, mut beta_h) = ([0u64; 16], [0u64; 16]);
        check(unsafe {
            capgpu_srs_deserialize(bytes.as_ptr(), bytes.len(), max_degree, &mut handle, h.as_mut_ptr(),
                                   beta_h.as_mut_ptr(), &mut used)
        })?;
        Ok((Srs { handle }, h, beta_h))
    }
    pub fn handle(&self) -> u64 {
        self.handle
    }
    pub fn len(&self) -> Result<usize> {
        let mut n = 0usize;
        check(unsafe { capgpu_srs_size(self.handle, &mut n) })?;
        Ok(n)
    }
    /// `VariableBaseMSM::multi_scalar_mul(&bases[offset..offset + scalars.len()], scalars)`: `scalars` are canonical
    /// integers (`into_repr()`), 4 words each; the result is (X, Y, Z) of a `GroupProjective`, Montgomery words.
    pub fn msm(&self, offset: usize, scalars: &[[u64; 4]]) -> Result<[u64; 12]> {
        let mut out = [0u64; 12];
        check(unsafe { capgpu_msm_g1(self.handle, offset, scalars.as_ptr() as *const u64, scalars.len(), out.as_mut_ptr()) })?;
        Ok(out)
    }
}
impl Drop for Srs {
    fn drop(&mut self) {
        unsafe { capgpu_srs_free(self.handle) };
    }
}

/// `Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place` on a `Vec<Fr>` viewed as words.
pub fn ntt_in_place(data: &mut [[u64; 4]], inverse: bool, coset: bool) -> Result<()> {
    assert!(data.len().is_power_of_two());
    check(unsafe {
        capgpu_ntt_fr(data.as_mut_ptr() as *mut u64, data.len().trailing_zeros(), inverse as c_int, coset as c_int)
    })
}

/// A device-resident proving key (`jf_plonk::proof_system::structs::ProvingKey`).
pub struct ProvingKey {
    handle: u64,
    pub vk: capgpu_verifying_key,
    pub domain_size: usize,
    pub num_inputs: usize,
}
impl ProvingKey {
    /// `PlonkKzgSnark::preprocess(srs, circuit)` from the circuit's tables of VALUES on the domain
    /// (`CAPGPU_INPUT_EVALS`): `selectors` = 13 columns (q_lc x4, q_mul x2, q_hash x4, q_o, q_c, q_ecc), `sigma` = the 5
    /// extended-permutation columns, n values each, column-major, Montgomery words.  This name has meant EVALUATIONS
    /// since round 1 and keeps meaning it: a caller holding jf-relation's polynomials uses `preprocess_coeffs` - the two
    /// forms cannot be told apart from the data, so the form is in the function name, not in a default.
    pub fn preprocess(srs: &Srs, n: usize, num_inputs: usize, selectors: &[[u64; 4]], sigma: &[[u64; 4]]) -> Result<ProvingKey> {
        Self::preprocess_form(srs, n, num_inputs, selectors, sigma, CAPGPU_INPUT_EVALS)
    }
    /// The same from POLYNOMIALS in coefficient form (`CAPGPU_INPUT_COEFFS`):
    /// `arkworks::poly_columns(&circuit.compute_selector_polynomials()?, n)` and
    /// `..compute_extended_permutation_polynomials()`, passed through as jf-relation computed them (no transform on the
    /// CPU; the device evaluates sigma where round 2 needs its values).  Same key bytes as `preprocess` on the values.
    pub fn preprocess_coeffs(srs: &Srs, n: usize, num_inputs: usize, selector_polys: &[[u64; 4]], sigma_polys: &[[u64; 4]]) -> Result<ProvingKey> {
        Self::preprocess_form(srs, n, num_inputs, selector_polys, sigma_polys, CAPGPU_INPUT_COEFFS)
    }
    /// Either form, stated explicitly (`CAPGPU_INPUT_EVALS` / `CAPGPU_INPUT_COEFFS`).
    pub fn preprocess_form(srs: &Srs, n: usize, num_inputs: usize, selectors: &[[u64; 4]], sigma: &[[u64; 4]],
                           input_form: c_int) -> Result<ProvingKey> {
        assert_eq!(selectors.len(), NUM_SELECTORS * n);
        assert_eq!(sigma.len(), NUM_WIRE_TYPES * n);
        let mut handle = 0u64;
        let mut vk: capgpu_verifying_key = unsafe { std::mem::zeroed() };
        check(unsafe {
            capgpu_plonk_preprocess_ex(srs.handle, n, num_inputs, selectors.as_ptr() as *const u64,
                                       sigma.as_ptr() as *const u64, input_form, &mut handle, &mut vk)
        })?;
        Ok(ProvingKey { handle, vk, domain_size: n, num_inputs })
    }
    /// `PlonkKzgSnark::prove::<_, _, SolidityTranscript>(rng, circuit, pk, Some(ext_msg))` for one note: callable from
    /// any thread; with coalescing on, concurrent calls share device batches.  `wires`: the 5 finalised wire COLUMNS, n
    /// VALUES each (`CAPGPU_INPUT_EVALS` - what this name has always taken); `blinders`: 13 `Fr::rand(rng)` draws in
    /// jf-plonk's order (2 per wire polynomial, then 3).
    pub fn prove(&self, wires: &[[u64; 4]], pub_inputs: &[[u64; 4]], ext_msg: &[u8], blinders: &[[u64; 4]; 13]) -> Result<capgpu_proof> {
        self.prove_form(wires, pub_inputs, ext_msg, blinders, CAPGPU_INPUT_EVALS)
    }
    /// The same from the 5 UNBLINDED wire POLYNOMIALS, n coefficients each (`CAPGPU_INPUT_COEFFS`) -
    /// `arkworks::poly_columns(&circuit.compute_wire_polynomials()?, n)`: what a jf-relation caller holds.  Same proof
    /// bytes as `prove` on the values.
    pub fn prove_coeffs(&self, wire_polys: &[[u64; 4]], pub_inputs: &[[u64; 4]], ext_msg: &[u8], blinders: &[[u64; 4]; 13]) -> Result<capgpu_proof> {
        self.prove_form(wire_polys, pub_inputs, ext_msg, blinders, CAPGPU_INPUT_COEFFS)
    }
    /// Either form, stated explicitly.
    pub fn prove_form(&self, wires: &[[u64; 4]], pub_inputs: &[[u64; 4]], ext_msg: &[u8], blinders: &[[u64; 4]; 13],
                      input_form: c_int) -> Result<capgpu_proof> {
        assert_eq!(wires.len(), NUM_WIRE_TYPES * self.domain_size);
        assert_eq!(pub_inputs.len(), self.num_inputs);
        let mut proof: capgpu_proof = unsafe { std::mem::zeroed() };
        check(unsafe {
            capgpu_plonk_prove_ex(self.handle, wires.as_ptr() as *const u64, pub_inputs.as_ptr() as *const u64,
                                  pub_inputs.len(), ext_msg.as_ptr(), ext_msg.len(), blinders.as_ptr() as *const u64,
                                  input_form, &mut proof)
        })?;
        Ok(proof)
    }
    pub fn handle(&self) -> u64 {
        self.handle
    }
}
impl Drop for ProvingKey {
    fn drop(&mut self) {
        unsafe { capgpu_plonk_free_key(self.handle) };
    }
}

/// The 769 ark-serialize bytes of a proof as it sits inside a `TransferNote` (src/transfer.rs:54-66).
pub fn proof_bytes(proof: &capgpu_proof) -> Result<Vec<u8>> {
    let mut out = vec![0u8; 1024];
    let mut len = 0usize;
    check(unsafe { capgpu_proof_serialize(proof, out.as_mut_ptr(), out.len(), &mut len) })?;
    out.truncate(len);
    Ok(out)
}

/// Conversions between arkworks 0.3 values and the ABI's words, and the circuit columns from jf-relation's PUBLIC API.
/// [DEP-RECALLED: written from the crates' APIs at the pinned revisions; this module has never been compiled.]
#[cfg(feature = "arkworks")]
pub mod arkworks {
    use ark_bn254::{Fq, Fr, G1Projective};
    use ark_ff::{BigInteger256, PrimeField};
    use ark_poly::univariate::DensePolynomial;

    /// `Fp256` keeps its Montgomery limbs in `.0.0`: the ABI's "Montgomery words".
    pub fn fr_words(x: &Fr) -> [u64; 4] {
        (x.0).0
    }
    pub fn fr_from_words(w: [u64; 4]) -> Fr {
        Fr::new(BigInteger256(w))
    }
    /// canonical integer of a scalar (`into_repr()`): what the MSM takes
    pub fn fr_canonical(x: &Fr) -> [u64; 4] {
        x.into_repr().0
    }
    /// (X, Y, Z) Montgomery words -> `GroupProjective` (Z = 0: the identity)
    pub fn g1_from_xyz(w: &[u64; 12]) -> G1Projective {
        let f = |i: usize| Fq::new(BigInteger256([w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]]));
        G1Projective::new(f(0), f(1), f(2))
    }
    /// A family of polynomials as the ABI's coefficient-form columns (`CAPGPU_INPUT_COEFFS`): n coefficients each,
    /// column-major, a `DensePolynomial` shorter than n (arkworks strips trailing zeros) zero-padded.  jf-relation's
    /// `Arithmetization` trait hands the prover exactly these polynomials, so nothing is transformed on the CPU:
    ///   let n = circuit.eval_domain_size()?;
    ///   let selectors = poly_columns(&circuit.compute_selector_polynomials()?, n);              // 13 x n
    ///   let sigma     = poly_columns(&circuit.compute_extended_permutation_polynomials()?, n);  // 5 x n
    ///   let wires     = poly_columns(&circuit.compute_wire_polynomials()?, n);                  // 5 x n (per proof)
    /// A memcpy of 5 n field elements per proof - against the 10 n-point FFTs per proof the evaluation form used to
    /// cost this shim (5 here to undo jf-relation's interpolation, 5 on the device to redo it).
    pub fn poly_columns(polys: &[DensePolynomial<Fr>], n: usize) -> Vec<[u64; 4]> {
        let mut out = vec![[0u64; 4]; polys.len() * n];
        for (i, p) in polys.iter().enumerate() {
            assert!(p.coeffs.len() <= n, "polynomial longer than the evaluation domain");
            for (j, c) in p.coeffs.iter().enumerate() {
                out[i * n + j] = fr_words(c);
            }
        }
        out
    }
    /// The evaluation-form columns (`CAPGPU_INPUT_EVALS`) of a family of polynomials: one n-point FFT each on the CPU.
    /// Kept for callers written against rounds 1-3 of this crate; new code passes `poly_columns` to the `_coeffs` entry
    /// points and lets the device do the transform.
    #[deprecated(note = "pass poly_columns(..) to preprocess_coeffs / prove_coeffs instead: no CPU transform")]
    pub fn circuit_columns(polys: &[DensePolynomial<Fr>], n: usize) -> Vec<[u64; 4]> {
        use ark_poly::{EvaluationDomain, Radix2EvaluationDomain};
        let domain = Radix2EvaluationDomain::<Fr>::new(n).expect("power-of-two domain");
        let mut out = vec![[0u64; 4]; polys.len() * n];
        for (i, p) in polys.iter().enumerate() {
            for (j, v) in domain.fft(&p.coeffs).iter().enumerate() {
                out[i * n + j] = fr_words(v);
            }
        }
        out
    }
}
