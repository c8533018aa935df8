/* capgpu.h - C ABI of libcapgpu.so: the MI355X (gfx950) replacement for the two
 * primitives that dominate CAP's PLONK prove() path, and for the prover that
 * schedules them.
 *
 * This is the drop-in boundary.  The reference (jf-cap, Rust)
 * has no FFI of its own; the seam is cut where its prover hands flat arrays to
 * arkworks.  Every entry point names the reference interface it replaces:
 *
 *   capgpu_msm_g1*        ark_ec::msm::VariableBaseMSM::multi_scalar_mul      (ark-ec 0.3.0, Cargo.lock:103-105)
 *                         via ark_poly_commit::kzg10::KZG10::commit           (Cargo.lock:208-210)
 *   capgpu_ntt_fr*        ark_poly::Radix2EvaluationDomain::{fft,ifft,
 *                         coset_fft,coset_ifft}_in_place                      (ark-poly 0.3.0, Cargo.lock:194-196)
 *   capgpu_srs_*          jf_plonk UniversalSrs / CommitKey powers_of_g       (src/proof/mod.rs:59-69, 74-109)
 *   capgpu_plonk_preprocess   PlonkKzgSnark::preprocess   (call sites src/proof/transfer.rs:133, mint.rs:76, freeze.rs:102)
 *   capgpu_plonk_prove        PlonkKzgSnark::prove::<_,_,SolidityTranscript>
 *                                                         (call sites src/proof/transfer.rs:181-186, mint.rs:113, freeze.rs:151)
 *   capgpu_plonk_verify       PlonkKzgSnark::verify::<SolidityTranscript>
 *                                                         (call sites src/proof/transfer.rs:202-207, mint.rs:132, freeze.rs:170)
 *
 * Conventions
 *   - All integers little-endian.  Field element Fr / Fq = uint64_t[4] (arkworks
 *     BigInteger256 limb order).  "Montgomery" = arkworks' in-memory Fp256 form
 *     (value * 2^256 mod p).  Scalars handed to the MSM are canonical integers
 *     (what `into_repr()` yields), exactly as arkworks' MSM takes them.
 *   - G1 affine = x, y (Montgomery), 64 bytes; the point at infinity is (0, 0)
 *     when no flag byte is present (see capgpu_srs_upload for arkworks' 72-byte
 *     struct).  G1 Jacobian = X, Y, Z (Montgomery), 96 bytes, infinity Z = 0
 *     (arkworks GroupProjective field order).
 *   - Every function returns CAPGPU_OK (0) or a negative CAPGPU_ERR_* code; it
 *     never aborts and never unwinds.  capgpu_last_error() gives a thread-local
 *     message.  The Rust shim maps non-zero to PlonkError, which prove()
 *     already maps to TxnApiError::FailedSnark (src/proof/transfer.rs:187).
 *   - Host-pointer entry points copy in/out; the caller owns its buffers for the
 *     duration of the call.  *_dev entry points take device pointers obtained
 *     from capgpu_malloc and enqueue on the library stream (capgpu_sync waits).
 *   - Thread safety: entry points may be called from any thread (rayon workers
 *     in the reference, src/utils/params_builder.rs:194-226); calls serialise
 *     on an internal lock per DEVICE CONTEXT.  One process drives as many GPUs
 *     as capgpu_init binds (see "devices" below): handles are process-wide,
 *     resident tables are replicated to a device on first use there, and
 *     host-buffer calls of threads that did not bind themselves to a device
 *     are dealt over the devices by the library.  Concurrent
 *     capgpu_plonk_prove calls can be gathered into device batches instead of
 *     queueing up: capgpu_plonk_set_coalescing.
 *   - There is no CPU fallback: without a usable gfx950 device capgpu_init
 *     fails with CAPGPU_ERR_NO_DEVICE and every other call fails with
 *     CAPGPU_ERR_NOT_INITIALISED.
 */
#ifndef CAPGPU_H
#define CAPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPGPU_OK 0
#define CAPGPU_ERR_INVALID_ARG (-1)
#define CAPGPU_ERR_NO_DEVICE (-2)
#define CAPGPU_ERR_HIP (-3)
#define CAPGPU_ERR_BAD_HANDLE (-4)
#define CAPGPU_ERR_OOM (-5)
#define CAPGPU_ERR_NOT_INITIALISED (-6)
#define CAPGPU_ERR_PROOF (-7) /* prover-side failure: wrong quotient degree (unsatisfied circuit), bad sizes */
#define CAPGPU_ERR_SERIALIZATION (-8) /* malformed parameter blob: ark_serialize::SerializationError, which the
                                         reference maps to TxnApiError::DeserializationError (src/errors.rs:81-85) */
#define CAPGPU_ERR_COMM (-9) /* multi-process exchange: a peer rank failed its part, or did not arrive in time */

#define CAPGPU_NUM_WIRE_TYPES 5
#define CAPGPU_NUM_SELECTORS 13

/* Form in which the *_ex PLONK entry points take a circuit's columns (wires, selectors, sigmas):
 *   CAPGPU_INPUT_EVALS   values on the evaluation domain - n per column, row j = the value at omega^j: the finalised
 *                        circuit's tables (what the entry points without _ex take);
 *   CAPGPU_INPUT_COEFFS  polynomials in coefficient form - n coefficients per column, zero-padded: exactly what
 *                        jf-relation's `Arithmetization` trait hands the prover (`compute_wire_polynomials`,
 *                        `compute_selector_polynomials`, `compute_extended_permutation_polynomials`; the reference
 *                        holds that circuit object at src/proof/transfer.rs:181-186 and :124-155, mint.rs:76/113,
 *                        freeze.rs:102/151).  The device then skips its own interpolation and runs one forward
 *                        transform where the permutation product needs the values; a Rust binding passes the trait's
 *                        output straight through instead of undoing its interpolation on the CPU.
 * Both forms give the same proof and the same keys, byte for byte. */
#define CAPGPU_INPUT_EVALS 0
#define CAPGPU_INPUT_COEFFS 1

/* ---- lifecycle ------------------------------------------------------------------------- */
/* Binds this process to the n_devices GPUs listed in device_ids (NULL / 0 selects HIP device 0): one device context
 * - stream, resident tables, scratch, lock - per id, numbered 0 .. n_devices - 1 in the order given ("slots").  The
 * reference is ONE process whose rayon threads each call prove() (src/utils/params_builder.rs:194-226); with several
 * devices bound, the library itself spreads such calls:
 *   - handles (SRS, proving keys) are process-wide; the tables behind them are copied to a device the first time it
 *     needs them (peer copy over xGMI);
 *   - capgpu_plonk_prove_batch / _prove_multi cut a host-resident batch into one part per device, proved concurrently;
 *     coalesced capgpu_plonk_prove calls (capgpu_plonk_set_coalescing) form one batch per free device; single
 *     host-buffer MSM / NTT calls go to a free device;
 *   - an SRS of >= 2^20 points (CAPGPU_SHARD_MIN_POINTS) is SHARDED by point range over the devices when it is
 *     uploaded or generated: every capgpu_msm_g1* call on it runs on all devices at once, each on the points it holds,
 *     and one exchange of the 96-byte partials (peer copies) plus n_devices - 1 additions gives the result;
 *   - *_dev entry points and capgpu_malloc work on the device of the calling thread: slot 0, or the slot the thread
 *     chose with capgpu_set_device.
 * A device may be listed once (CAPGPU_ERR_INVALID_ARG otherwise; CAPGPU_ALLOW_DUPLICATE_DEVICES=1 lifts this for tests
 * that drive the multi-device paths on one GPU).  CAPGPU_CONTEXTS_PER_DEVICE=k gives every listed device k contexts,
 * whose batches overlap on that device; the default is 4 when ONE device is bound (capgpu_device_count then reports 4:
 * a host-buffer batch is cut in two, gathered batches of coalesced calls take any free one, two parts at a time) and 1 per device otherwise.  One process per GPU (torchrun) keeps working: each process binds one device
 * and the ranks meet through capgpu_comm_* ("multi-GPU" below).  Idempotent: a second call is a no-op. */
int capgpu_init(const int* device_ids, int n_devices);
/* number of device CONTEXTS bound by capgpu_init (0 before it): the range of capgpu_set_device's slots.  NOT a GPU
 * count - one bound GPU has four contexts by default; capgpu_context_count is the same number under its proper name,
 * capgpu_physical_device_count the number of distinct HIP devices behind them. */
int capgpu_device_count(int* count_out);
int capgpu_context_count(int* count_out);
int capgpu_physical_device_count(int* count_out);
/* Binds the CALLING THREAD to context `slot` (0 .. count - 1): its *_dev calls, capgpu_malloc / memcpy / sync and its
 * host-buffer calls then all run there.  slot = -1 (the default of every thread) unbinds: device-pointer calls use slot
 * 0, host-buffer calls are dealt by the library. */
int capgpu_set_device(int slot);
/* the calling thread's binding (-1: none) and the HIP device id its device-pointer calls use */
int capgpu_get_device(int* slot_out, int* hip_device_out);
void capgpu_shutdown(void);
const char* capgpu_last_error(void);
const char* capgpu_version(void);
/* name (<= 255 chars + NUL), compute units, HBM bytes of the bound device */
int capgpu_device_info(char* name_out, int* cu_count_out, uint64_t* hbm_bytes_out);
/* Memory path between the devices of two contexts, settled by capgpu_init (hipDeviceCanAccessPeer +
 * hipDeviceEnablePeerAccess for every bound pair): *access_out = 1 direct peer access (xGMI / PCIe P2P: replication of
 * keys and SRS tables, scalar slices and partials of sharded MSMs travel device to device), 0 none (the runtime stages
 * such copies through host memory; everything still works), 2 the two contexts share one device. */
int capgpu_device_peer_info(int slot_a, int slot_b, int* access_out);
/* free / total device memory of the calling thread's device, in bytes (hipMemGetInfo) */
int capgpu_mem_info(uint64_t* free_bytes_out, uint64_t* total_bytes_out);
/* Footprint control (SURVEY 8b ownership rules: the library owns its workspace, the caller decides how much of the GPU
 * that may be).  A context's scratch - prover workspace, MSM workspace, NTT scratch, staging - grows to the largest call
 * it has served and stays (~25 GB per context after a 256-proof batch at n = 2^15; four contexts per bound device):
 *  - capgpu_trim releases the scratch, the pinned result area and the captured launch graphs of every context no call is
 *    running on; tables the caller created (SRS, proving keys, NTT domains) stay.  *bytes_released_out: device bytes
 *    given back; *contexts_busy_out: contexts skipped because a call was running on them.  The next call on a trimmed
 *    context allocates again (a hipMalloc per buffer, ~ms).
 *  - capgpu_set_memory_limit caps the scratch the library holds PER DEVICE (sum over that device's contexts; 0 = no
 *    cap, the default).  A call that would grow past it first takes the growth slack off, then trims the device's idle
 *    contexts, and then fails with CAPGPU_ERR_OOM naming the bytes it needed - the caller proves in smaller batches
 *    (scratch is proportional to the batch; what the failed call had already grown is released at its context's next
 *    entry) or raises the cap.  Scratch already held above a new cap is trimmed from
 *    idle contexts at once.  Tables are not counted.
 *  - capgpu_scratch_info: scratch bytes currently held on the calling thread's device, and the cap. */
int capgpu_trim(uint64_t* bytes_released_out, int* contexts_busy_out);
int capgpu_set_memory_limit(uint64_t scratch_bytes_per_device);
int capgpu_scratch_info(uint64_t* scratch_bytes_out, uint64_t* limit_out);
/* Host-side phase trace (diagnostics; no reference counterpart): while on, the library timestamps the phases the kernel
 * profiler cannot see - a coalesced call's queueing, window and context wait, the host-to-device copies of a batch's
 * witnesses, the host steps between the prover's rounds, the release of the callers - into a ring of 2^20 events.
 * capgpu_trace_enable(1) starts a fresh trace, (0) stops it; capgpu_trace_dump writes one line per event
 * ("t_us thread tag a b"; tools/gpu_phase_trace.py reads it) and returns the count.  Off: one relaxed load per site. */
int capgpu_trace_enable(int on);
int capgpu_trace_dump(const char* path, uint64_t* events_out);

/* ---- device memory / stream (plumbing for callers that keep data resident) -------------- */
int capgpu_malloc(void** dev_ptr_out, size_t bytes);
int capgpu_free(void* dev_ptr);
int capgpu_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes);
int capgpu_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes);
int capgpu_sync(void);
/* waits for everything enqueued on every bound device (hipDeviceSynchronize per device): what a caller without a HIP
 * binding of its own uses where a torch program would call torch.cuda.synchronize() */
int capgpu_sync_all(void);
/* HIP runtime / driver version the PROCESS runs on (hipRuntimeGetVersion: e.g. 70226015).  The library is built against
 * /opt/rocm; a process that loaded another libamdhip64.so.7 first - PyTorch's wheel bundles its own - runs the library on
 * THAT runtime (same SONAME: the loader keeps the first).  bench.py records it: the host-witness legs differ by 3 % between
 * the two runtimes on this image. */
int capgpu_runtime_info(int* hip_runtime_version_out, int* hip_driver_version_out);
/* Device time, in milliseconds, of the work the calling thread's context executes between the two calls (HIP events on
 * its stream; _end waits for that work).  SURVEY 8d's "hipEvent around device section": bench.py times its MSM / NTT
 * legs with it.  ONE measurement per context at a time, owned by the thread that opened it: that thread may restart it
 * with a second _begin; _begin from another thread while it is open, and _end without an open measurement of the
 * calling thread, return CAPGPU_ERR_INVALID_ARG (threads that time concurrently bind different contexts:
 * capgpu_set_device). */
int capgpu_timer_begin(void);
int capgpu_timer_end(double* ms_out);
/* Run all subsequent work on the caller's hipStream_t (e.g. torch's current stream); NULL
 * restores the library's own stream. */
int capgpu_set_stream(void* hip_stream);

/* ---- SRS / commit key: stays device-resident across proofs -------------------------------- */
/* bases: n affine G1 points, stride_bytes apart (64 = packed x,y; 72 = arkworks GroupAffine with
 * a trailing `infinity: bool` byte at offset 64).  coords_montgomery: 1 for arkworks memory.
 * Expands every base into its window multiples on device (one-time cost). */
int capgpu_srs_upload(const void* bases, size_t n, size_t stride_bytes, int coords_montgomery,
                      uint64_t* handle_out);
/* Synthetic SRS [tau^i] G, i < n, generated on device - the counterpart of
 * universal_setup(max_degree, rng) (src/proof/mod.rs:59-69) for benches without the Aztec file.
 * tau: canonical Fr integer. */
int capgpu_srs_generate(const uint64_t tau[4], size_t n, uint64_t* handle_out);
/* The same with the hiding powers a KZG10 setup also produces: powers_of_gamma_g = { i: [gamma tau^i] G } for degrees
 * 0 .. n (max_degree + 1), kept with the handle, written by capgpu_srs_serialize and - by degree - into the commit key of
 * every proving key preprocessed under it (capgpu_plonk_key_serialize).  The prover's commitments stay non-hiding, as
 * jf-plonk's are; the powers exist so that the stored files are what a reference-side consumer expects. */
int capgpu_srs_generate_hiding(const uint64_t tau[4], const uint64_t gamma[4], size_t n, uint64_t* handle_out);
/* bases[i] = [a + i*b] G (canonical Fr integers) - synthetic bases for the 2^24 scaling config */
int capgpu_srs_generate_affine_seq(const uint64_t a[4], const uint64_t b[4], size_t n, uint64_t* handle_out);
int capgpu_srs_size(uint64_t handle, size_t* n_out);
/* point-range shards the SRS is held in: 1, or the device count for a sharded SRS (capgpu_init) */
int capgpu_srs_shards(uint64_t handle, int* shards_out);
/* copies bases [offset, offset+n) back as packed 64-byte Montgomery affine points */
int capgpu_srs_download(uint64_t handle, size_t offset, size_t n, void* out);
int capgpu_srs_free(uint64_t handle);

/* ---- MSM: replaces VariableBaseMSM::multi_scalar_mul ------------------------------------------ */
/* out = sum_i scalars[i] * bases[offset + i];  scalars canonical 4 x u64; out Jacobian 96 B (X, Y, Z Montgomery; Z = 0:
 * infinity).  The triple is A representative of the point - like ark-ec's G1Projective it depends on the order the
 * additions were made in, which on the device varies from run to run (bucket lists are filled with atomics): compare
 * results in affine form (X / Z^2, Y / Z^3), as every caller of the reference does through into_affine(). */
int capgpu_msm_g1(uint64_t srs_handle, size_t offset, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]);
/* The same commitment from a polynomial's VALUES: out = sum_{j < n} s_j [L_j(tau)] G + sum_{e < 3} s_(n+e) [tau^(n+e) -
 * tau^e] G, n = 2^log_n, L_j the Lagrange basis of the n-th roots of unity; `count` <= n + 3 scalars (the last three
 * slots are the blinders of jf-plonk's polynomials: (b0 + b1 X)(X^n - 1) for a wire, (b0 + b1 X + b2 X^2)(X^n - 1) for the
 * permutation product); scalars_montgomery != 0: arkworks' Fr memory form.  Equals capgpu_msm_g1 on the coefficients
 * ark-poly's ifft makes of the values - the form rounds 1 and 2 of the prover commit in (capgpu_plonk_set_wire_commit).
 * The Lagrange-form commit key is derived from the SRS on first use (the SRS must hold n + 3 points) and kept with it. */
int capgpu_msm_g1_lagrange(uint64_t srs_handle, uint32_t log_n, const uint64_t* scalars, size_t count,
                           int scalars_montgomery, uint64_t out_xyz[12]);
int capgpu_msm_g1_batch(uint64_t srs_handle, const size_t* offsets, const uint64_t* const* scalars,
                        const size_t* ns, int count, uint64_t* out_xyz /* count*12 */);
/* Device-resident form: d_scalars = count arrays of n scalars, scalar_stride elements apart;
 * scalars_montgomery != 0 converts from Montgomery first (what a polynomial's coefficients are);
 * d_out_xyz = count * 96 bytes on device. */
int capgpu_msm_g1_dev(uint64_t srs_handle, size_t offset, const void* d_scalars, size_t scalar_stride, size_t n,
                      int count, int scalars_montgomery, void* d_out_xyz);

/* ---- scalars resident with their points ("GPU g holds its bases resident and receives the matching scalar
 * slice") ------------------------------------------------------------------------------------------------------------
 * capgpu_msm_g1_dev on a sharded SRS has to scatter the caller's scalars over the devices on EVERY call (32 B x n leaving
 * one GPU: 448 MB of a 2^24-point MSM).  A caller that runs more than one MSM on the same scalars - or that can place
 * them once, ahead of time - makes them resident instead: `count` arrays over points [offset, offset + n) are cut by
 * the SRS's point ranges, each slice stored on the device that holds its points (host memory goes to each device
 * directly; device memory of the calling thread's context by one peer copy per slice).  capgpu_msm_g1_resident then
 * exchanges nothing but the 96-byte partials.  Works on an unsharded SRS too (one slice, on the calling thread's
 * context).  The set is immutable; free it with capgpu_msm_scalars_free (capgpu_shutdown frees what is left). */
int capgpu_msm_scalars_upload(uint64_t srs_handle, size_t offset, const uint64_t* scalars, size_t scalar_stride,
                              size_t n, int count, uint64_t* scalars_handle_out);
int capgpu_msm_scalars_scatter_dev(uint64_t srs_handle, size_t offset, const void* d_scalars, size_t scalar_stride,
                                   size_t n, int count, uint64_t* scalars_handle_out);
int capgpu_msm_scalars_free(uint64_t scalars_handle);
/* d_out_xyz: count * 96 bytes on the calling thread's device, as capgpu_msm_g1_dev */
int capgpu_msm_g1_resident(uint64_t srs_handle, uint64_t scalars_handle, int scalars_montgomery, void* d_out_xyz);
/* Bytes moved between device contexts (or from the host) by sharded MSMs since capgpu_init - scalar slices and
 * 96-byte partials -, the number of sharded MSM calls, and how many times an SRS or proving key was replicated onto
 * another context.  Any out pointer may be NULL. */
int capgpu_msm_shard_stats(uint64_t* scalar_bytes_out, uint64_t* partial_bytes_out, uint64_t* calls_out,
                           uint64_t* replications_out);

/* Diagnostic: which window table, sort and split `count` MSMs of n points on this SRS would take, as text
 * ("c=15 windows=18 sort=two-level parts=256 n_sub=65536 slice=1").  Tests pin the plan of the BASELINE sizes with
 * it, so that a size limit can never silently move a configuration to a slower path. */
int capgpu_msm_plan(uint64_t srs_handle, size_t n, int count, char* buf, size_t cap);

/* out = sum of n Jacobian points (96 B each, host memory): the combine step after the all-gather of a
 * point-range-sharded MSM (replaces the G-1 `GroupProjective::add_assign` a multi-GPU caller would do). */
int capgpu_g1_sum(const uint64_t* points_xyz, size_t n, uint64_t out_xyz[12]);

/* ---- multi-GPU: one process per GPU, MSM sharded by point range -------------------------------
 * The reference parallelises inside one process (rayon, src/utils/params_builder.rs:194-226); a multi-GPU
 * deployment starts one worker process per GPU.  Rank g uploads (or generates) the bases of ITS point range as its
 * SRS and passes the matching scalar slice; every rank runs the whole Pippenger locally down to one point, then ONE
 * exchange step - an RCCL all-gather of the 96-byte partial per MSM on the library stream, device to device over xGMI -
 * and G - 1 group additions on the device give every rank the full result.  (RCCL has no elliptic-curve reduction
 * operator, hence no all-reduce; bucket arrays are never exchanged.)
 *
 * capgpu_comm_unique_id: called by ONE rank; the 128 bytes (an ncclUniqueId) travel to the other ranks by whatever
 * channel the job has (MPI, a socket, torch.distributed - bench.py broadcasts them).  capgpu_comm_init is collective:
 * it returns once all `world` ranks have called it - or fails with CAPGPU_ERR_COMM when they have not arrived within
 * CAPGPU_COMM_TIMEOUT_MS (default 60000).  RCCL is loaded at that moment (dlopen of librccl.so.1; a copy
 * already in the process is reused), so single-GPU users need no RCCL at all.
 * Failures are agreed on: a rank whose local MSM fails still enters the exchange (its status travels with the
 * partials) and every rank returns an error; a rank that never enters it is caught by the same deadline, the
 * communicator is aborted and the call returns CAPGPU_ERR_COMM instead of hanging. */
int capgpu_comm_unique_id(uint8_t id_out[128]);
int capgpu_comm_init(int rank, int world, const uint8_t id[128]);
int capgpu_comm_destroy(void);
/* rank / world of the communicator; world == 0 when there is none */
int capgpu_comm_info(int* rank_out, int* world_out);
/* Test communicator: a world of `world` ranks that THIS process plays one after the other on its device, through the
 * same payload layout, gather buffer and summation kernel as the RCCL path (the all-gather itself becomes a copy into
 * the rank's slot).  capgpu_msm_g1_sharded_dev is then called once per rank (capgpu_comm_loopback_set_rank before each;
 * the call of the last rank leaves the sum), and with capgpu_plonk_shard_msm(1) the prover plays all ranks of every
 * commitment MSM itself.  Lets a 1-GPU box execute the N > 1 code paths. */
int capgpu_comm_init_loopback(int world);
int capgpu_comm_loopback_set_rank(int rank);
/* out (on every rank) = sum over ranks of sum_i scalars_r[i] * bases_r[offset + i]: `count` MSMs in one launch, their
 * partials exchanged in ONE all-gather of count * 96 bytes per rank.  Arguments as capgpu_msm_g1_dev (n_local = this
 * rank's points; may differ between ranks, count may not). */
int capgpu_msm_g1_sharded_dev(uint64_t srs_handle, size_t offset, const void* d_scalars, size_t scalar_stride,
                              size_t n_local, int count, int scalars_montgomery, void* d_out_xyz);
int capgpu_msm_g1_sharded(uint64_t srs_handle, size_t offset, const uint64_t* scalars, size_t n_local,
                          uint64_t out_xyz[12]);
/* BASELINE config 4, mode A: with on != 0 every commitment MSM of capgpu_plonk_preprocess / capgpu_plonk_prove* is cut
 * by point range over the ranks of the communicator (each rank holds the whole commit key, uses its range) and all
 * ranks must then make the same calls with the same inputs; they all return the same proofs.  Mode B - the default,
 * and the faster one for throughput - is replicas: independent proofs on independent ranks, no communicator needed. */
int capgpu_plonk_shard_msm(int on);

/* ---- NTT: replaces Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place -------- */
/* in place, natural order in/out, Montgomery Fr; dir: 0 forward, 1 inverse (includes n^-1);
 * coset: 0/1 (generator 5: scale by 5^i before the forward transform / by 5^-i after the inverse). */
int capgpu_ntt_fr(uint64_t* data, uint32_t log_n, int dir, int coset);
int capgpu_ntt_fr_batch(uint64_t* const* data, int count, uint32_t log_n, int dir, int coset);
int capgpu_ntt_fr_dev(void* d_data, size_t stride_elems, int count, uint32_t log_n, int dir, int coset);

/* ---- PLONK (TurboPlonk, 5 wires, 13 selectors) ------------------------------------------------ */
typedef struct capgpu_proof {
  uint64_t wires_poly_comms[CAPGPU_NUM_WIRE_TYPES][8];      /* affine, Montgomery */
  uint64_t prod_perm_poly_comm[8];
  uint64_t split_quot_poly_comms[CAPGPU_NUM_WIRE_TYPES][8];
  uint64_t opening_proof[8];
  uint64_t shifted_opening_proof[8];
  uint64_t wires_evals[CAPGPU_NUM_WIRE_TYPES][4];           /* Fr, Montgomery */
  uint64_t wire_sigma_evals[CAPGPU_NUM_WIRE_TYPES - 1][4];
  uint64_t perm_next_eval[4];
} capgpu_proof;

typedef struct capgpu_verifying_key {
  uint64_t domain_size;
  uint64_t num_inputs;
  uint64_t k[CAPGPU_NUM_WIRE_TYPES][4];                     /* coset representatives, Montgomery */
  uint64_t selector_comms[CAPGPU_NUM_SELECTORS][8];         /* q_lc x4, q_mul x2, q_hash x4, q_o, q_c, q_ecc */
  uint64_t sigma_comms[CAPGPU_NUM_WIRE_TYPES][8];
} capgpu_verifying_key;

/* selectors: 13 columns of n Fr (Montgomery), column-major, gate order above; sigma_evals: 5 columns
 * of n Fr = sigma_i(omega^j) (the extended permutation as field elements k_i' * omega^j').
 * n must be a power of two, 16 <= n (the quotient is interpolated on 6n points, which must hold its 5n + 8
 * coefficients and the 5 (n + 2) the split-quotient commitments read), n + 3 <= SRS size. */
int capgpu_plonk_preprocess(uint64_t srs_handle, size_t n, size_t num_inputs, const uint64_t* selectors,
                            const uint64_t* sigma_evals, uint64_t* pk_handle_out, capgpu_verifying_key* vk_out);
/* The same with the columns in `input_form` (above): CAPGPU_INPUT_COEFFS takes the 13 selector polynomials and the 5
 * extended-permutation polynomials, n coefficients each (a DensePolynomial shorter than n is zero-padded by the
 * caller), in the same column order. */
int capgpu_plonk_preprocess_ex(uint64_t srs_handle, size_t n, size_t num_inputs, const uint64_t* selectors,
                               const uint64_t* sigmas, int input_form, uint64_t* pk_handle_out,
                               capgpu_verifying_key* vk_out);
int capgpu_plonk_free_key(uint64_t pk_handle);
/* Shape of a resident proving key: the sizes every prove call's arrays must have (wires: count * 5 * domain_size
 * field elements, pub_inputs: count * num_inputs, blinders: count * 13) and the SRS it commits with.  Any out
 * pointer may be NULL. */
int capgpu_plonk_key_info(uint64_t pk_handle, size_t* domain_size_out, size_t* num_inputs_out,
                          uint64_t* srs_handle_out);

/* wires: 5 columns of n Fr (Montgomery), column-major (the finalised circuit's wire assignment);
 * pub_inputs: num_inputs Fr (Montgomery); ext_msg: the caller's transcript init message
 * (src/proof/transfer.rs:178-180) or NULL; blinders: 13 Fr (Montgomery) drawn by the caller's RNG in
 * the order jf-plonk draws them: 2 per wire polynomial (constant, linear), then 3 for the
 * permutation product polynomial. */
int capgpu_plonk_prove(uint64_t pk_handle, const uint64_t* wires, const uint64_t* pub_inputs, size_t num_inputs,
                       const uint8_t* ext_msg, size_t ext_msg_len, const uint64_t* blinders, capgpu_proof* proof_out);
/* Coalescing of concurrent capgpu_plonk_prove calls (off by default).  The reference proves notes from many rayon
 * worker threads, one prove() per note (src/utils/params_builder.rs:194-226); behind one device those calls would run
 * one after the other at single-proof latency.  With window_us > 0, calls for proving keys of one domain size under one
 * SRS (the notes of different kinds the reference proves side by side share batches) that arrive within
 * window_us microseconds of each other (the window restarts with every arrival, 16 windows at most) - or while the
 * device is busy with a previous batch - are gathered (up to max_batch; 0 = 256) and proved as ONE device batch; each caller receives its own proof and its own return code
 * (an unsatisfied witness fails only its owner).  window_us = 0 switches it off. */
int capgpu_plonk_set_coalescing(uint32_t window_us, uint32_t max_batch);
/* How round 1 computes the five wire commitments.  jf-plonk commits to each blinded wire polynomial through its n + 2
 * COEFFICIENTS (KZG10::commit under src/proof/transfer.rs:181-186).  The same group element is
 *     sum_j w_j [L_j(tau)] G + b0 [tau^n - 1] G + b1 [tau^(n+1) - tau] G
 * - an MSM of the column's n VALUES and its two blinders on the Lagrange-form commit key of the domain, which the
 * library derives from the SRS once per (SRS, domain size) by a group inverse transform (cap_amd/csrc/lagrange.hip;
 * 2 x 64 B x (n + 3) x 18-20 window rows of device memory, built by capgpu_plonk_preprocess or by the first proof; the
 * permutation product's commitment is taken the same way, with its three blinders).  The
 * witness values of a CAP circuit are mostly zeros, booleans and range-check limbs (src/circuit/transfer.rs:53-193): as
 * MSM scalars they have at most one non-zero digit where a coefficient has seventeen.  Proof bytes are identical.
 * mode: 1 = from evaluations (the default), 0 = from coefficients, -1 = back to the default (CAPGPU_WIRE_COMMIT=coeffs
 * makes 0 the process default).  Process-wide; takes effect with the next prove call.
 * When the Lagrange-form key cannot be built (device memory) the call commits from coefficients instead of failing.
 *
 * TIMING AND THE SECRET WITNESS - what this library does and does not promise.  Like the arkworks prover it replaces
 * (ark-ec's multi_scalar_mul skips zero scalars and treats ones apart; its field inversions are variable-time) this
 * library is NOT hardened against timing side channels: kernel durations and memory traffic may depend on secret data,
 * and an observer who can time proofs precisely must be kept away by the deployment, not by this code.  What depends on
 * what:
 *  - mode 1 (default): the wire MSMs' scalars are the secret witness values themselves; the bucket sort skips zero digits,
 *    so round 1's duration grows with the number of non-zero 15-bit digits of the witness (zeros, booleans and small
 *    limbs are cheap: the same property the speed-up comes from).  It reveals an aggregate of the witness's sparsity per
 *    proof (per batch, in a batch), not individual values.
 *  - mode 0 (CAPGPU_WIRE_COMMIT=coeffs): the scalars are the blinded polynomials' coefficients - full-width values
 *    whatever the witness holds; round 1's work is then independent of the witness to the degree jf-plonk's own is.  This
 *    is the mode to choose where proof timing is observable by an adversary.
 *  - the permutation product's commitment (either mode) and everything from round 3 on work on challenge-randomised,
 *    full-width data; round 2's one shared inversion runs a fixed-length chain in both modes (it costs nothing).
 * Power, electromagnetic and co-tenant cache channels on a shared GPU are out of scope in both modes. */
int capgpu_plonk_set_wire_commit(int mode);
/* device batches run and proofs made through the coalescer so far */
int capgpu_plonk_coalescing_stats(uint64_t* batches_out, uint64_t* proofs_out);
/* Small batches (count <= CAPGPU_GRAPH_MAX_BATCH, default 16; 0 switches it off) replay their kernel schedule as
 * hipGraphs: the ~100 launches of a proof fall into seven segments between the host's transcript steps; the second call
 * with the same key, batch size and buffers captures them, later calls launch seven graphs instead.  Proofs are the same
 * bytes either way.  Counters since process start: segments captured (instantiated) and segments replayed. */
int capgpu_plonk_graph_stats(uint64_t* segments_captured_out, uint64_t* segments_replayed_out);
/* Same, `count` independent proofs under one key pipelined on the device; per-proof arrays are
 * consecutive (wires: count * 5 * n, pub_inputs: count * num_inputs, blinders: count * 13).  With several device
 * contexts bound (capgpu_init) and a calling thread that did not bind itself to one, the batch is cut into contiguous
 * parts of at least CAPGPU_DEAL_MIN (default 8) proofs, one per context, proved concurrently; the proofs are those of the
 * undivided call.  On failure the first failing part's code and message are returned and proofs_out is unspecified. */
int capgpu_plonk_prove_batch(uint64_t pk_handle, int count, const uint64_t* wires, const uint64_t* pub_inputs,
                             size_t num_inputs, const uint8_t* ext_msg, size_t ext_msg_len,
                             const uint64_t* blinders, capgpu_proof* proofs_out);
/* Proofs of SEVERAL proving keys in one device batch: pk_handles[i] is the key of proof i.  The reference proves its
 * transfer, mint and freeze notes side by side (TxnsParams::generate_txns, src/utils/params_builder.rs:194-226); on the
 * device, proofs of different circuits over the same evaluation domain share every MSM and NTT launch.  All keys of a
 * call must have the same domain size and come from the same SRS (CAPGPU_ERR_INVALID_ARG otherwise).  wires: count * 5
 * * n field elements; pub_inputs: count rows of num_inputs elements, num_inputs = the largest public-input count among
 * the keys - a key with fewer inputs uses the first of its row, the rest is ignored; ext_msgs / ext_msg_lens: one
 * transcript init message per proof, or NULL; blinders: count * 13.  Every proof is bit-identical to the one
 * capgpu_plonk_prove makes for the same inputs. */
int capgpu_plonk_prove_multi(const uint64_t* pk_handles, int count, const uint64_t* wires, const uint64_t* pub_inputs,
                             size_t num_inputs, const uint8_t* const* ext_msgs, const size_t* ext_msg_lens,
                             const uint64_t* blinders, capgpu_proof* proofs_out);
int capgpu_plonk_prove_multi_dev(const uint64_t* pk_handles, int count, const void* d_wires, const uint64_t* pub_inputs,
                                 size_t num_inputs, const uint8_t* const* ext_msgs, const size_t* ext_msg_lens,
                                 const uint64_t* blinders, capgpu_proof* proofs_out);
/* Device-resident witness form used by the benchmark (inputs already in HBM). */
int capgpu_plonk_prove_batch_dev(uint64_t pk_handle, int count, const void* d_wires, const uint64_t* pub_inputs,
                                 size_t num_inputs, const uint8_t* ext_msg, size_t ext_msg_len,
                                 const uint64_t* blinders, capgpu_proof* proofs_out);

/* The prove entry points with the wire columns in `input_form` (above).  CAPGPU_INPUT_COEFFS: `wires` holds, per proof,
 * the 5 UNBLINDED wire polynomials of n coefficients (jf-relation's compute_wire_polynomials; the blinders are added on
 * the device as before).  Everything else - layouts, batching over contexts, coalescing (calls of different forms are
 * gathered separately), errors - is that of the entry point without _ex, which is the _ex one with CAPGPU_INPUT_EVALS. */
int capgpu_plonk_prove_ex(uint64_t pk_handle, const uint64_t* wires, const uint64_t* pub_inputs, size_t num_inputs,
                          const uint8_t* ext_msg, size_t ext_msg_len, const uint64_t* blinders, int input_form,
                          capgpu_proof* proof_out);
int capgpu_plonk_prove_batch_ex(uint64_t pk_handle, int count, const uint64_t* wires, const uint64_t* pub_inputs,
                                size_t num_inputs, const uint8_t* ext_msg, size_t ext_msg_len,
                                const uint64_t* blinders, int input_form, capgpu_proof* proofs_out);
int capgpu_plonk_prove_multi_ex(const uint64_t* pk_handles, int count, const uint64_t* wires,
                                const uint64_t* pub_inputs, size_t num_inputs, const uint8_t* const* ext_msgs,
                                const size_t* ext_msg_lens, const uint64_t* blinders, int input_form,
                                capgpu_proof* proofs_out);
int capgpu_plonk_prove_multi_dev_ex(const uint64_t* pk_handles, int count, const void* d_wires,
                                    const uint64_t* pub_inputs, size_t num_inputs, const uint8_t* const* ext_msgs,
                                    const size_t* ext_msg_lens, const uint64_t* blinders, int input_form,
                                    capgpu_proof* proofs_out);
int capgpu_plonk_prove_batch_dev_ex(uint64_t pk_handle, int count, const void* d_wires, const uint64_t* pub_inputs,
                                    size_t num_inputs, const uint8_t* ext_msg, size_t ext_msg_len,
                                    const uint64_t* blinders, int input_form, capgpu_proof* proofs_out);

/* ---- verification (host only: needs neither a GPU nor capgpu_init) ---------------------------------------- */
/* G2 elements: x.c0, x.c1, y.c0, y.c1 of the twist point (Fq2 = Fq[u]/(u^2+1)), Montgomery, 16 words;
 * all-zero = infinity.  They are the `h` / `beta_h` of jf-plonk's VerifyingKey.open_key. */
int capgpu_g2_generator(uint64_t out[16]);
/* out = scalar * q (canonical 4 x u64 scalar): builds [tau]H for a synthetic SRS (src/proof/mod.rs:59-69) */
int capgpu_g2_mul(const uint64_t q[16], const uint64_t scalar[4], uint64_t out[16]);
/* *ok_out = (prod_i e(P_i, Q_i) == 1);  P_i: n affine G1 points (8 words each), Q_i: n G2 points (16 words each) */
int capgpu_pairing_check(const uint64_t* g1_points, const uint64_t* g2_points, size_t n, int* ok_out);
/* Replaces PlonkKzgSnark::verify::<SolidityTranscript> (src/proof/transfer.rs:192-212, mint.rs:124-140,
 * freeze.rs:162-178).  Returns CAPGPU_OK with *ok_out = 1 (accept) / 0 (reject); a negative code only for
 * malformed arguments (wrong number of public inputs, G2 elements off the curve). */
int capgpu_plonk_verify(const capgpu_verifying_key* vk, const uint64_t g2_h[16], const uint64_t g2_beta_h[16],
                        const uint64_t* pub_inputs, size_t num_inputs, const capgpu_proof* proof,
                        const uint8_t* ext_msg, size_t ext_msg_len, int* ok_out);

/* Replaces PlonkKzgSnark::batch_verify as used by txn_batch_verify (src/lib.rs:455-529): one pairing product for
 * `count` proofs (possibly of different circuits / keys under one SRS).  ext_msgs / ext_msg_lens may be NULL. */
int capgpu_plonk_batch_verify(const capgpu_verifying_key* const* vks, const uint64_t g2_h[16],
                              const uint64_t g2_beta_h[16], const uint64_t* const* pub_inputs,
                              const size_t* num_inputs, const capgpu_proof* const* proofs,
                              const uint8_t* const* ext_msgs, const size_t* ext_msg_lens, size_t count, int* ok_out);
/* The same predicate with its group arithmetic on the device: the ~35 (point, scalar) terms of every
 * proof, weights folded in, are two multi-scalar multiplications on the prover's MSM kernels (the bases are uploaded
 * like an SRS and their window tables built on the device); the transcripts and the final pairing product stay on the
 * host.  Accepts and rejects exactly what capgpu_plonk_batch_verify does.  Needs capgpu_init
 * (CAPGPU_ERR_NOT_INITIALISED otherwise: no host path hides behind this entry point). */
int capgpu_plonk_batch_verify_dev(const capgpu_verifying_key* const* vks, const uint64_t g2_h[16],
                                  const uint64_t g2_beta_h[16], const uint64_t* const* pub_inputs,
                                  const size_t* num_inputs, const capgpu_proof* const* proofs,
                                  const uint8_t* const* ext_msgs, const size_t* ext_msg_lens, size_t count, int* ok_out);
/* ark-serialize 0.3 CanonicalSerialize bytes of the Proof as it sits inside a TransferNote / MintNote / FreezeNote
 * (src/transfer.rs:60): compressed G1 (32 B), Fr little-endian, Vec = u64 length prefix, plookup_proof = None.
 * 769 bytes; *len_out receives the size. */
int capgpu_proof_serialize(const capgpu_proof* proof, uint8_t* out, size_t cap, size_t* len_out);
/* The inverse (`Proof::deserialize`, what reading a note from bytes does): CAPGPU_ERR_SERIALIZATION on every encoding
 * ark-serialize rejects (vector lengths, non-canonical x or scalar, x off the curve, both flag bits, a plookup
 * proof).  Host only.  *consumed_out receives the bytes read. */
int capgpu_proof_deserialize(const uint8_t* bytes, size_t len, capgpu_proof* proof_out, size_t* consumed_out);

/* ---- on-disk parameter formats ------------------------------------------------------
 * The reference stores and loads its parameters as ark-serialize 0.3 `CanonicalSerialize` bytes
 * (store_data / load_data, src/parameters.rs:560-577; load_srs, src/proof/mod.rs:74-109) and notes that
 * "deserializing these parameter files takes longer than reproducing them" (src/lib.rs:81-86): the cost is a
 * square root in Fq per compressed point.  Here the bulk G1 decompression is one kernel launch.
 * Encodings: usize = u64 LE; Vec / BTreeMap = u64 length + items; Fr = 32 B LE canonical; G1 compressed = x (32 B
 * LE) with 0x80 of the last byte = "y is the larger root" and 0x40 = infinity; G2 compressed = x.c0, x.c1 (64 B)
 * with the same flags in the last byte.  The field order inside each struct is restated from the crates'
 * definitions (ark-poly-commit @ cafc05e, jf-plonk @ bcd92b2), which are not in the reference tree: parity with a
 * blob written by the reference is unpinned. */

/* n compressed G1 points (32 B each) <-> affine (x, y), 8 Montgomery words each, (0, 0) = infinity.
 * Decompression fails with CAPGPU_ERR_SERIALIZATION on x >= p, x not on the curve, or both flag bits set. */
int capgpu_g1_decompress(const uint8_t* in, size_t n, uint64_t* out_xy);
int capgpu_g1_compress(const uint64_t* xy, size_t n, uint8_t* out);

/* UniversalSrs blob (parameters::load_universal_parameter, src/parameters.rs:97-109; load_srs): validates every
 * point, keeps the first max_degree + 1 powers of g resident (0 = all) and returns the handle plus the open key
 * (h, beta_h: G2 affine, x.c0 x.c1 y.c0 y.c1 Montgomery).  *consumed_out = bytes read. */
int capgpu_srs_deserialize(const uint8_t* bytes, size_t len, size_t max_degree, uint64_t* handle_out,
                           uint64_t h_out[16], uint64_t beta_h_out[16], size_t* consumed_out);
/* parameters::store_universal_parameter_for_demo (src/parameters.rs:47-65).  out == NULL queries the size. */
int capgpu_srs_serialize(uint64_t handle, const uint64_t h[16], const uint64_t beta_h[16], uint8_t* out, size_t cap,
                         size_t* len_out);

/* jf-plonk VerifyingKey blob (store_/load_*_verifying_key, src/parameters.rs:190-241, 314-362, 438-478) without
 * the note-shape trailer the Transfer/Mint/Freeze wrappers append (src/proof/transfer.rs:83-94); host only.
 * g, gamma_g: the G1 part of the open key (gamma_g may be NULL on serialize = infinity: commitments here are
 * non-hiding and neither prover nor verifier reads it).  capgpu_plonk_key_serialize with gamma_g == NULL writes the
 * gamma_g of the blob the key - or the UniversalSrs it was preprocessed under (degree 0 of its hiding powers) - was
 * loaded from, so that load -> store gives the file back; infinity for a synthetic SRS. */
int capgpu_plonk_vk_serialize(const capgpu_verifying_key* vk, const uint64_t g[8], const uint64_t gamma_g[8],
                              const uint64_t h[16], const uint64_t beta_h[16], uint8_t* out, size_t cap,
                              size_t* len_out);
int capgpu_plonk_vk_deserialize(const uint8_t* bytes, size_t len, capgpu_verifying_key* vk_out, uint64_t g_out[8],
                                uint64_t gamma_g_out[8], uint64_t h_out[16], uint64_t beta_h_out[16],
                                size_t* consumed_out);

/* jf-plonk ProvingKey blob (store_/load_*_proving_key, src/parameters.rs:113-188, 244-312, 364-436), again without
 * the wrapper's trailer: sigma and selector polynomials, the commit key and the verifying key.  Deserialising
 * decompresses the commit key on the device, registers it as a new SRS (*srs_handle_out, owned by the caller) and
 * rebuilds the resident tables of the prover without redoing the 18 interpolations and commitments of
 * preprocess.  out == NULL on serialize queries the size. */
int capgpu_plonk_key_serialize(uint64_t pk_handle, const uint64_t gamma_g[8], const uint64_t h[16],
                               const uint64_t beta_h[16], uint8_t* out, size_t cap, size_t* len_out);
int capgpu_plonk_key_deserialize(const uint8_t* bytes, size_t len, uint64_t* srs_handle_out, uint64_t* pk_handle_out,
                                 capgpu_verifying_key* vk_out, uint64_t h_out[16], uint64_t beta_h_out[16],
                                 size_t* consumed_out);

/* ---- instrumentation ------------------------------------------------------------------------------ */
/* Measured issue rate of v_mad_u64_u32 on the bound device, in lane-operations per second (8 independent chains per
 * lane, 8 waves per SIMD, ~50 ms).  A lazy Montgomery multiplication is 171 of them (81 + 81 + 9), so rate / 171 is the
 * chip's multiplication ceiling - what bench.py prices the ALU-bound kernels against. */
int capgpu_ubench_mad_rate(double* lane_ops_per_s_out);
/* The same measurement for the instruction classes the hot kernels are made of, all at that occupancy and chain count,
 * in lane-operations per second: rates_out[0..count) = v_mad_u64_u32, v_add_u32, v_and_b32, v_mov_b32, v_lshl_add_u64,
 * v_lshrrev_b64, v_alignbit_b32, v_mul_lo_u32, then [8] a mixed stream - three multiply-adds, one plain instruction, the
 * shape of a column-wise Montgomery product - at the same occupancy and [9] the same stream held to three waves per SIMD,
 * msm_accumulate's occupancy (count <= 10; ~0.4 s).  bench.py prices a kernel's instruction mix
 * (profiles/isa_mix_r03.json) against them: issue_frac. */
int capgpu_ubench_issue_rates(double* rates_out, int count);
/* When enabled, every kernel launch is bracketed by HIP events on the launch stream and accumulated
 * per kernel name (costs a few microseconds per launch; leave off for throughput runs). */
int capgpu_profile_enable(int on);
int capgpu_profile_reset(void);
/* name == kernel name (e.g. "msm_accumulate"); total milliseconds and launch count since reset */
int capgpu_profile_get(const char* name, double* total_ms_out, uint64_t* launches_out);
/* writes up to cap bytes of "name total_ms launches\n" lines */
int capgpu_profile_dump(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* CAPGPU_H */
