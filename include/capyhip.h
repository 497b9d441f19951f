/*
 * capyhip.h — C ABI of libcapyhip.so: the MI355X (gfx950) batched crypto core that stands in for
 * capyCRYPT's sponge and Ed448 hot path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference (Rust, /root/reference) has no FFI today: its seams are the Rust functions and
 * traits cited on each entry point below.  A batch of 1 equals the reference's scalar call.
 * The Rust-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   d            security parameter 224 | 256 | 384 | 512        (SecParam, src/lib.rs:111-135)
 *   msgs/offsets n messages packed in one buffer; message i is msgs[offsets[i] .. offsets[i+1]).
 *                offsets has n+1 entries, non-decreasing.  Host batches are re-packed to aligned starts as needed.
 *   keys / pws   n per-item keys or passwords.  key_offsets (pw_offsets) == NULL: n byte strings of key_len (pw_len)
 *                bytes back to back; otherwise n+1 non-decreasing byte offsets into the buffer, key i =
 *                keys[key_offsets[i] .. key_offsets[i+1]) -- one length per item, as the reference takes any &[u8]
 *                per message (src/ecc/signable.rs:40-43, src/ecc/keypair.rs:41, src/sha3/hashable.rs:33-35) -- and
 *                key_len is ignored.  A key is at most 1 MiB long.
 *   scalars      56-byte BIG-endian, unreduced                   (src/sha3/aux_functions.rs:102-110)
 *   points       affine (x, y), 2 x 56-byte little-endian canonical field elements (x first)
 *   return       0 = ok, <0 = CAPY_ERR_*; capy_last_error() gives the text (thread local)
 *   host forms   (no _dev suffix) take and return HOST buffers and block until the results are there.  Buffers of up to
 *                1 MiB go through a per-thread arena of pinned, device-mapped host memory (no copy calls: the kernels
 *                read and write it directly; CAPY_DEBUG=host_arena=0 switches it off), larger ones through cached staging
 *                blocks on the device.
 *   *_dev        same operation on buffers already resident in device memory, enqueued on `stream`
 *                (a hipStream_t passed as void*; NULL = the CALLING THREAD's default stream: the library is built with
 *                -fgpu-default-stream=per-thread since r03, so that host threads -- the library's own per-device workers
 *                among them -- do not serialise on one legacy stream.  NOTE for callers that hold a LEGACY default
 *                stream, e.g. torch.cuda.current_stream().cuda_stream == 0 for torch's default stream: handle 0 passed
 *                here means hipStreamPerThread, not the legacy stream.  Work on the two is still ordered by the
 *                legacy stream's implicit synchronisation with every blocking stream, but not under stream capture
 *                and not against work the caller put on a NON-blocking stream: pass that stream's handle then),
 *                no host synchronisation and no device
 *                copy from host memory: internal scratch is pooled per (host thread, device, stream), grows by
 *                allocating (never by freeing) and is returned by capy_release_workspace(); secret intermediates
 *                in it (z||pw, ke||ka, s, k, the ECDH point) are zeroed on the stream at the end of the call.
 *                One exception: cSHAKE/KMAC at d = 224 with a customisation string longer than 162 bytes stages its
 *                prefix with a synchronous copy.
 *                Message starts that are 8-byte aligned take the coalesced fast path (any alignment is correct).
 *                The kernels read whole aligned 8-byte words: up to 7 bytes past the end of an 8-byte aligned
 *                message are read (never written, never past the aligned word that holds its last byte).
 * All entry points are thread safe; no pointer is retained after return.  Randomness (nonces) is
 * always an input so results are reproducible.
 */
#ifndef CAPYHIP_H
#define CAPYHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPY_OK 0
#define CAPY_ERR_UNSUPPORTED_SECPARAM (-1) /* OperationError::UnsupportedSecurityParameter, src/lib.rs:10 */
#define CAPY_ERR_ARG (-2)
#define CAPY_ERR_HIP (-3)
#define CAPY_ERR_UNSUPPORTED (-4)

/* per-item status codes written by the *_decrypt / verify entry points */
#define CAPY_ITEM_OK 0
#define CAPY_ITEM_FAIL 1 /* SHA3DecryptionFailure / KeyDecryptionError / SignatureVerificationFailure */

/* Constant-address table lookups for Ed448 scalar multiplications (see capy_ed448_set_hardened).  The values are fixed:
 * 1 meant "every multiplication" in the r02 library and keeps that meaning; 2 and 3 (r03 only) are refused. */
#define CAPY_HARDEN_OFF 0      /* indexed kernels everywhere (benchmarks; single-tenant devices) */
#define CAPY_HARDEN_ALL 1      /* every scalar multiplication, the raw scalarmul / basemul calls included */
#define CAPY_HARDEN_PROTOCOL 4 /* DEFAULT: the multiplications by secret scalars inside the protocol calls */
#define CAPY_OPT_DEFAULT (-1)

/* Per-call options (r04): what the process-wide setters capy_ed448_set_hardened / _set_scalar_star / _set_generator fix
 * for everybody, a single call can choose for itself through the *_ex entry points -- two host threads can then hold
 * different modes at the same time.  Fields left at CAPY_OPT_DEFAULT / 0 / NULL take the process-wide setting.
 * Initialise with CAPY_CALL_OPTIONS_INIT (struct_size lets the struct grow). */
typedef struct capy_call_options {
    uint32_t struct_size; /* sizeof(capy_call_options) */
    int32_t hardened;     /* CAPY_HARDEN_* or CAPY_OPT_DEFAULT */
    int32_t scalar_star;  /* 0, 1, 2 (capy_ed448_set_scalar_star) or CAPY_OPT_DEFAULT */
    int32_t generator;    /* handle from capy_ed448_generator_create; 0 = the process generator */
    void *stream;         /* *_dev_ex forms: the hipStream_t to enqueue on (NULL = the calling thread's default stream);
                             ignored by the host-buffer forms */
} capy_call_options;
#define CAPY_CALL_OPTIONS_INIT {(uint32_t)sizeof(capy_call_options), CAPY_OPT_DEFAULT, CAPY_OPT_DEFAULT, 0, NULL}

/* ABI identity.  CAPY_ABI_VERSION is a monotonic integer, raised whenever an entry point is added or the meaning of an
 * argument changes; a binding compares capy_abi_version() of the library it loaded with the CAPY_ABI_VERSION it was
 * written against (refuse an older library, accept a newer one) -- capy_version() is for
 * humans.  History:
 *   1  r01: sponge + Ed448 batch entry points               2  r02: per-item key lengths (key_offsets), capy_set_devices,
 *                                                               capy_ed448_validate_batch, one-item-per-wave families
 *   3  r03: capy_ed448_set_hardened / _set_scalar_star / _set_generator, KEM sponge half, key_encrypt / key_decrypt
 *   4  r04: capy_call_options and the *_ex entry points, generator handles, five further *_dev forms;
 *           capy_ed448_set_hardened values 2 and 3 refused (1 = every multiplication again, 4 = the protocol default)
 *   5  r05: capy_abi_version, capy_set_min_items_per_device, capy_debug_last_sponge_kernel, capy_debug_affinity_plan
 *   6  r06: capy_device_topology */
#define CAPY_ABI_VERSION 6
int capy_abi_version(void);
const char *capy_last_error(void);
const char *capy_version(void);
int capy_device_count(void);
int capy_set_device(int device); /* device used by the calling thread's subsequent calls */
/* Multi-GPU ("batches shard trivially across the 8 GPUs of one node", BASELINE north_star).
 * After capy_set_devices(ids, n) every HOST-buffer entry point below cuts its batch into n contiguous shards --
 * balanced by message bytes where the call carries messages, by count otherwise --, runs shard k on device ids[k]
 * (one PERSISTENT worker thread per list position: it keeps its device, its scratch pools and its cache of staging
 * buffers from call to call and is pinned to the CPUs the device is attached to, /sys/bus/pci/devices/<bdf>/local_cpulist;
 * sharded calls from several host threads take turns) and has it write its slice of the caller's output arrays.
 * Items are independent:
 * no collective and no peer-to-peer traffic; results, output order and in-place effects are identical to the
 * single-device call for every device list (an id may repeat).  n = 0 returns to the calling thread's current
 * device.  Process-wide; the *_dev entry points are unaffected (their buffers live on one device).
 * capy_get_devices writes at most `capacity` ids and returns the length of the configured list. */
int capy_set_devices(const int *ids, int n);
int capy_get_devices(int *ids, int capacity);
/* The minimum-shard rule: a sharded call uses only as many of the listed devices (the first ones) as leave each at least
 * `n` items; default 1.  Small batches are latency-bound -- one device runs 1024 messages of 5 MiB through sha3_encrypt in
 * 1.6 x the time it needs for 128, and 32 768 Ed448 multiplications in 0.19 x the time of 262 144 -- so cutting them finer
 * buys nothing and costs a PCIe hop per device; INTEGRATION.md section 5 lists the batch sizes per operation from which a
 * second device pays.  Process-wide; n = 0 restores the default. */
int capy_set_min_items_per_device(size_t n);
/* The cut capy_set_devices uses: bounds[r] .. bounds[r+1] is the item range of device r of n_devices (bounds has
 * n_devices + 1 entries).  byte_offsets = the n+1 message offsets of the call (balance by bytes: an item goes to the
 * shard its midpoint falls in), or NULL (balance by count).  Pure host arithmetic. */
int capy_shard_plan(size_t n, int n_devices, const uint64_t *byte_offsets, uint64_t *bounds);
int capy_device_synchronize(void); /* every configured device, else the current one */
/* Where `device` sits: its PCI bus id ("0000:c1:00.0", empty if unknown) into pci_bus_id, its NUMA node (-1 if the system does
 * not say) into *numa_node, and the CPUs a worker thread of capy_set_devices pins itself to for it -- the device's
 * /sys/bus/pci/devices/<bdf>/local_cpulist intersected with the calling thread's affinity mask -- into cpus[0 .. capacity).
 * Returns the number of such CPUs (it may exceed capacity; 0 = no pinning: sysfs hides the device, or the intersection is empty),
 * < 0 on a bad argument.  Any pointer may be NULL.  No reference counterpart (the reference runs on the CPU); bench.py prints it
 * per rank so that a multi-GPU record documents its own topology. */
int capy_device_topology(int device, char *pci_bus_id, size_t pci_capacity, int *numa_node, int *cpus, int capacity);
/* Free the calling thread's pooled device scratch on every device (synchronises); also done when the thread ends. */
int capy_release_workspace(void);
/* Test hook: how many bytes of the scratch ranges that the calling thread's LAST protocol call declared secret (secret
 * scalars, nonces, shared points, derived keys, z || pw, and the keyed sponge states that the phase / time-slice schedules
 * carry between launches) are non-zero now.  Synchronises `stream`.  Must be 0 once the call has returned and the stream is
 * idle. */
int capy_debug_secret_scratch_nonzero(void *stream, uint64_t *nonzero_bytes);
/* Test hook: which kernel family the calling thread's last variable-base / fixed-base launch took: 1 indexed lookups,
 * 2 constant-address lookups, + 16 for the one-item-per-wave kernels of small batches; 0 = none yet. */
int capy_debug_last_curve_kernel(int *variable_base, int *fixed_base);
/* Test hook: which sponge kernel / schedule the calling thread's last digest or encrypt / decrypt launch took, and in how many
 * launches of the data pass (phases, time slices).  kind: 1 one lane per sponge (latency-tuned), 2 two lanes per sponge,
 * 3 rotating one-/two-lane schedule, 4 one lane per sponge (issue-tuned), 5 wave-quantisation split, (6 two items per wave:
 * r02-r04, replaced by 10),
 * 7 uniform-framing kernel, 8 rotating-occupancy schedule, 9 uniform-framing kernel in time slices, 10 one wave per item with
 * bit-interleaved Keccak lanes; sha3_encrypt / decrypt and
 * the other symmetric halves: 20 four lanes per item, (21 one wave per item: r02-r04, replaced by 27), 22 four lanes per item in time slices, 23 one lane
 * per sponge (two lanes per item), 24 the same in time slices, 25 the same on the rotating-occupancy schedule, 26 two passes,
 * 27 two waves per item with bit-interleaved Keccak lanes;
 * 0 = none yet.  Lets the tests assert that the path they mean to cover is the one that ran. */
int capy_debug_last_sponge_kernel(int *kind, int *launches);
/* Test hook for the CPU pinning of the per-device workers (pure host arithmetic, no GPU needed): parses `local_cpulist`
 * (the text of /sys/bus/pci/devices/<bdf>/local_cpulist, e.g. "0-15,128-143\n"), intersects it with the n_allowed CPU
 * numbers in allowed_cpus (the process's affinity mask) and writes the result, ascending, to out_cpus (at most `capacity`
 * of them).  Returns the size of the intersection; 0 = the worker would leave its affinity alone (unparsable list, or no
 * common CPU).  <0 = CAPY_ERR_ARG. */
int capy_debug_affinity_plan(const char *local_cpulist, const int *allowed_cpus, int n_allowed, int *out_cpus, int capacity);

/* ------------------------------------------------------------------ sponge (src/sha3) */

/* SHA3-d of n messages.  digests: n * d/8 bytes.
 * Replaces shake(), src/sha3/shake_functions.rs:24-32, as called by
 * SpongeHashable::compute_sha3_hash, src/sha3/hashable.rs:19-21 (bit-exact incl. its suffix rule). */
int capy_sha3_batch(int d, size_t n, const uint8_t *msgs, const uint64_t *offsets, uint8_t *digests);
int capy_sha3_batch_dev(int d, size_t n, const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len,
                        uint64_t msg_stride, uint8_t *digests, void *stream);

/* cSHAKE: out_i = cshake(x_i, l_bits, N, S, d).  outs: n * l_bits/8 bytes.
 * Replaces cshake(), src/sha3/shake_functions.rs:49-64 (capacity = d), including its N = S = "" corner (:59-61: the
 * dropped shake() call whose buffer mutation is kept), which both forms reproduce bit for bit (the device-buffer form
 * through a scratch copy of the messages with their trailers, synchronously: crate-internal, unreachable through
 * kmac_xof). */
int capy_cshake_batch(int d, size_t n, const uint8_t *xs, const uint64_t *offsets, size_t l_bits,
                      const uint8_t *fn_name, size_t fn_len, const uint8_t *custom, size_t custom_len,
                      uint8_t *outs);
/* Same on device buffers (layout as capy_sha3_batch_dev); output i at outs + i*out_stride. */
int capy_cshake_batch_dev(int d, size_t n, const uint8_t *xs, const uint64_t *offsets, uint64_t uniform_len,
                          uint64_t msg_stride, size_t l_bits, const uint8_t *fn_name, size_t fn_len,
                          const uint8_t *custom, size_t custom_len, uint8_t *outs, uint64_t out_stride, void *stream);

/* KMACXOF: out_i = kmac_xof(key_i, x_i, l_bits, S, d); keys as in Conventions (fixed key_len or key_offsets).
 * Replaces kmac_xof(), src/sha3/shake_functions.rs:79-89 (pub), and
 * SpongeHashable::compute_tagged_hash, src/sha3/hashable.rs:33-35 (l_bits = d). */
int capy_kmac_xof_batch(int d, size_t n, const uint8_t *keys, size_t key_len, const uint64_t *key_offsets,
                        const uint8_t *xs, const uint64_t *offsets, size_t l_bits, const uint8_t *custom,
                        size_t custom_len, uint8_t *outs);
/* device form: keys at keys + i*key_stride (key_len bytes each), or via key_offsets (n+1 DEVICE offsets, key_len and
 * key_stride ignored); x_i via offsets or (uniform_len, msg_stride); x may be NULL with uniform_len = 0; outs at
 * outs + i*out_stride (out_stride multiple of 8). */
int capy_kmac_xof_batch_dev(int d, size_t n, const uint8_t *keys, size_t key_len, uint64_t key_stride,
                            const uint64_t *key_offsets, const uint8_t *xs, const uint64_t *offsets,
                            uint64_t uniform_len, uint64_t msg_stride, size_t l_bits, const uint8_t *custom,
                            size_t custom_len, uint8_t *outs, uint64_t out_stride, void *stream);

/* SpongeEncryptable::sha3_encrypt, src/sha3/encryptable.rs:29-45.
 * pws: n passwords (fixed pw_len or pw_offsets, see Conventions); zs: n caller-supplied 512-byte nonces (the reference
 * draws them from thread_rng, :31); msgs transformed in place to ciphertext; tags: n * 64 bytes. */
int capy_sha3_encrypt_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                            const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets, uint8_t *tags);
/* SpongeEncryptable::sha3_decrypt, src/sha3/encryptable.rs:58-83.  status[i] = CAPY_ITEM_OK and msg i
 * holds the plaintext, or CAPY_ITEM_FAIL and msg i is restored to the ciphertext (:77-82). */
int capy_sha3_decrypt_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                            const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets, const uint8_t *tags,
                            int32_t *status);
/* device forms: pw_offsets (if not NULL) are n+1 DEVICE offsets; pws_bytes is then the total number of password bytes
 * (pw_offsets[n] - pw_offsets[0]; it sizes the z||pw scratch without reading device memory), ignored otherwise. */
int capy_sha3_encrypt_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                uint64_t pws_bytes, const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets,
                                uint64_t uniform_len, uint64_t msg_stride, uint8_t *tags, void *stream);
int capy_sha3_decrypt_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                uint64_t pws_bytes, const uint8_t *zs, uint8_t *msgs, const uint64_t *offsets,
                                uint64_t uniform_len, uint64_t msg_stride, const uint8_t *tags, int32_t *status,
                                void *stream);

/* The sponge half of KEMEncryptable::kem_encrypt / kem_decrypt, src/kem/encryptable.rs:47-59, 84-104
 * (SURVEY.md §8f rank 1): identical flow to sha3_encrypt with the ML-KEM shared secret in place of the
 * password and the customisation strings "KEMKE" / "KEMKA".  ML-KEM itself (capy_kem) stays on the host. */
int capy_kem_sponge_encrypt_batch(int d, size_t n, const uint8_t *secrets, size_t secret_len, const uint8_t *zs,
                                  uint8_t *msgs, const uint64_t *offsets, uint8_t *tags);
int capy_kem_sponge_decrypt_batch(int d, size_t n, const uint8_t *secrets, size_t secret_len, const uint8_t *zs,
                                  uint8_t *msgs, const uint64_t *offsets, const uint8_t *tags, int32_t *status);
/* device forms (r04): secrets at secrets + i*secret_len; messages as in capy_sha3_encrypt_batch_dev */
int capy_kem_sponge_encrypt_batch_dev(int d, size_t n, const uint8_t *secrets, size_t secret_len, const uint8_t *zs,
                                      uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride,
                                      uint8_t *tags, void *stream);
int capy_kem_sponge_decrypt_batch_dev(int d, size_t n, const uint8_t *secrets, size_t secret_len, const uint8_t *zs,
                                      uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride,
                                      const uint8_t *tags, int32_t *status, void *stream);

/* ------------------------------------------------------------------ Ed448 (tiny_ed448_goldilocks boundary) */

/* out_i = [scalar_i] P_i  — `ExtendedPoint * Scalar` followed by to_affine()
 * (call sites src/ecc/encryptable.rs:37,78; src/ecc/signable.rs:77). */
int capy_ed448_scalarmul_batch(size_t n, const uint8_t *scalars_be, const uint8_t *points_xy, uint8_t *out_xy);
int capy_ed448_scalarmul_batch_dev(size_t n, const uint8_t *scalars_be, const uint8_t *points_xy,
                                   uint8_t *out_xy, void *stream);
/* The point `ExtendedPoint::generator()` stands for.  Default: the RFC 8032 Ed448 base point -- an ASSUMPTION about the
 * absent curve crate; capy_ed448_set_generator(xy) replaces it (xy must pass the validation below
 * and must have the prime order r -- [r] xy = (0, 1), checked on the host; NULL restores the default) for every later fixed-base
 * multiplication, key pair, signature and ECDHIES call of the process; the per-device fixed-base tables are rebuilt on
 * next use.  Calls already in flight finish on the old tables (retired, not freed); meant to be called once at start-up. */
int capy_ed448_set_generator(const uint8_t *xy);
int capy_ed448_get_generator(uint8_t *xy);
/* Register a further generator (same checks) for calls that select it through capy_call_options::generator; its
 * fixed-base tables are built per device on first use.  The same point registered twice returns the same handle; handles
 * stay valid for the life of the process (at most 63 of them). */
int capy_ed448_generator_create(const uint8_t *xy, int *handle);
/* The second hedge of that kind: Signable::sign computes its nonce as `bytes_to_scalar(k_bytes) * Scalar::from(4)`
 * (src/ecc/signable.rs:46) -- the crate's `*` operator on a value that is NOT reduced mod r, where every other call
 * site spells out mul_mod -- and then `k - h.mul_mod(&s)` (:54).  What `*` and `-` do to an unreduced Scalar cannot
 * be read off /root/reference (assumption (iii)).  mode 0 (default): `*` is the product mod r.
 * mode 1: `*` wraps at 2^448 and `-` is crypto-bigint's sub_mod on the unreduced value (z = k - hs, + r on borrow).
 * mode 2: `*` wraps at 2^448, `-` reduces (z = (k - hs) mod r).  In modes 1 and 2 U = [k]G takes the unreduced k.
 * Signatures of all three verify; their bytes differ.  Process-wide; affects capy_schnorr_sign_* only. */
int capy_ed448_set_scalar_star(int mode);
/* out_i = [scalar_i] G — `ExtendedPoint::generator() * Scalar`
 * (src/ecc/keypair.rs:44, src/ecc/signable.rs:48,77, src/ecc/encryptable.rs:38). */
int capy_ed448_basemul_batch(size_t n, const uint8_t *scalars_be, uint8_t *out_xy);
int capy_ed448_basemul_batch_dev(size_t n, const uint8_t *scalars_be, uint8_t *out_xy, void *stream);
/* out_i = P_i + Q_i — `ExtendedPoint + ExtendedPoint` (src/ecc/signable.rs:77) */
int capy_ed448_add_batch(size_t n, const uint8_t *p_xy, const uint8_t *q_xy, uint8_t *out_xy);
/* out_i = [a_i] G + [b_i] P_i in one pass (the shape of verify, src/ecc/signable.rs:77) */
int capy_ed448_double_scalarmul_batch(size_t n, const uint8_t *a_be, const uint8_t *b_be,
                                      const uint8_t *points_xy, uint8_t *out_xy);
/* device forms (r04) */
int capy_ed448_add_batch_dev(size_t n, const uint8_t *p_xy, const uint8_t *q_xy, uint8_t *out_xy, void *stream);
int capy_ed448_double_scalarmul_batch_dev(size_t n, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *points_xy,
                                          uint8_t *out_xy, void *stream);

/* Side-channel hardening: constant-address table lookups (process-wide default; per call: capy_call_options::hardened).
 * An indexed window table makes the ADDRESS stream of a scalar multiplication depend on the scalar's digits (control flow
 * is uniform either way): a cache-timing channel on a GPU shared with untrusted tenants; the reference's curve crate
 * advertises fixed-time lookups (tests/integration_tests.rs:131-134).  In the constant-address form every row of the
 * window table is read per window and the wanted one kept by masking (variable base: the 17-row per-item table, 4-bit
 * windows in the batched kernels; fixed base: a second shared table with 5-bit windows, 17 entries read per window by
 * scalar loads, or 7-bit windows picked by a one-hot product on the matrix cores); the sign is applied by masks; no
 * address and no branch depends on a scalar (evidence: profiles/r04_constant_address_counters.txt,
 * tests/test_constant_address.py).  Results are bit-identical with the indexed kernels.
 *   CAPY_HARDEN_PROTOCOL (4, DEFAULT): the multiplications by SECRET scalars inside the protocol calls -- capy_keypair_*,
 *          capy_schnorr_sign_* (the nonce k), capy_key_encrypt_* (the ephemeral k, both multiplications),
 *          capy_key_decrypt_* (the private scalar) -- at every batch size (the one-item-per-wave kernels have
 *          constant-address forms too).  Verification and the raw capy_ed448_scalarmul / basemul calls, whose scalars
 *          the library takes to be public, keep the indexed kernels.
 *   CAPY_HARDEN_ALL (1): also the raw scalarmul / basemul calls (for callers that pass secrets through them) -- the
 *          meaning 1 had in the r02 library.
 *   CAPY_HARDEN_OFF (0): indexed kernels everywhere.
 *   2 and 3 (r03's "raw calls only" / "everything") are refused with CAPY_ERR_ARG: r03 had silently narrowed 1 to the
 *   protocol calls, so a caller written against either release gets an error rather than a changed meaning.
 * Cost against the indexed kernels (profiles/r03_ed448_hardened.txt, r03_ed448_fb7_mfma.txt): variable base 1.2-1.3x;
 * fixed base 1.7x (65 additions instead of 39: the batched kernel picks its 7-bit-window table entries with a one-hot
 * byte matrix product on the matrix cores, csrc/ed448_fb7.h), 1.5x for the one-item-per-wave kernels of small batches;
 * key pair 1.6x, sign 1.4x. */
int capy_ed448_set_hardened(int mode);

/* Tuning / A-B switch (process-wide): batches of up to max_items scalar multiplications take the one-item-per-wave
 * kernels (csrc/ed448_wave.h: a field element spread over 16 lanes, the four coordinates of a point in the four rows of
 * a wave; 6x / 2.5x lower latency than one item per lane for variable / fixed base, less throughput; fixed-base
 * batches switch at 5/16 of the value, with constant-address lookups at 7/16).  0 = never, negative = the built-in
 * default (8192).  Results are bit-identical either way; the constant-address (hardened) forms exist for both. */
int capy_ed448_set_wave_max(long max_items);
/* Tuning / A-B switch (process-wide): batches of min_items < n <= max_items public-scalar multiplications (the raw
 * scalarmul / double_scalarmul calls, verification) take the four-lanes-per-item kernels (csrc/ed448_quad.h: X, Y, Z, T of
 * the accumulator in the four lanes of a quad, the field multiplications of a formula level side by side; 2.3x lower
 * latency than one item per lane, less throughput).  Defaults 4096 / 32768 (negative restores them; max = 0: never); the
 * two-lane range below takes 16384 < n <= 32768 out of it.  An explicit max also bounds the constant-address quad kernel
 * (vb_quad_ct_kernel: secret scalars, window table in LDS; default 4096 < n <= 32768).  Results are bit-identical either way. */
int capy_ed448_set_quad_range(long min_items, long max_items);
/* The same for the two-lanes-per-item kernels (csrc/ed448_duo.h: (Y, Z) and (X, T) of the accumulator in the two lanes of
 * a pair; one wave of 32 items per SIMD at 32 768 items, where the four-lane form needs two).  Checked before the quad
 * range.  Defaults 16384 / 32768 (negative restores them; max = 0: never).  Results are bit-identical either way. */
int capy_ed448_set_duo_range(long min_items, long max_items);

/* status[i] = CAPY_ITEM_OK iff point i has canonical coordinates (both < p) and lies on the curve.  The multiplication
 * and protocol entry points do NOT validate their point inputs (results for off-curve or non-canonical input are
 * unspecified, as with the reference's ExtendedPoint built from raw coordinates); callers that take points from
 * untrusted sources (a public key or asym_nonce read from a file, src/lib.rs:101-108) run this first. */
int capy_ed448_validate_batch(size_t n, const uint8_t *points_xy, int32_t *status);
int capy_ed448_validate_batch_dev(size_t n, const uint8_t *points_xy, int32_t *status, void *stream);

/* ------------------------------------------------------------------ src/ecc protocols */

/* KeyPair::new, src/ecc/keypair.rs:41-51: pub_i = [4 * KMAC(pw_i,"",448,"SK",d) mod r] G */
int capy_keypair_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets, uint8_t *pub_xy);
/* Signable::sign, src/ecc/signable.rs:40-57.  h: n*56 bytes, z_be: n*56 bytes (big-endian scalar). */
int capy_schnorr_sign_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                            const uint8_t *msgs, const uint64_t *offsets, uint8_t *h, uint8_t *z_be);
/* Signable::verify, src/ecc/signable.rs:72-86.  status[i] = CAPY_ITEM_OK | CAPY_ITEM_FAIL */
int capy_schnorr_verify_batch(int d, size_t n, const uint8_t *pub_xy, const uint8_t *msgs,
                              const uint64_t *offsets, const uint8_t *h, const uint8_t *z_be, int32_t *status);
/* KeyEncryptable::key_encrypt, src/ecc/encryptable.rs:34-50.  k_rand: n caller-supplied 56-byte
 * nonces (:36).  msgs -> ciphertext in place; z_xy: n*112 (asym_nonce); tags: n*56. */
int capy_key_encrypt_batch(int d, size_t n, const uint8_t *pub_xy, const uint8_t *k_rand, uint8_t *msgs,
                           const uint64_t *offsets, uint8_t *z_xy, uint8_t *tags);
/* KeyEncryptable::key_decrypt, src/ecc/encryptable.rs:72-94 (restore-on-failure, :88-93). */
int capy_key_decrypt_batch(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                           const uint8_t *z_xy, uint8_t *msgs, const uint64_t *offsets, const uint8_t *tags,
                           int32_t *status);

/* Device-buffer forms of the src/ecc protocols (same semantics; messages via offsets or uniform_len/msg_stride,
 * enqueued on `stream`, no host synchronisation). */
int capy_keypair_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                           uint8_t *pub_xy, void *stream);
int capy_schnorr_sign_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride,
                                uint8_t *h, uint8_t *z_be, void *stream);
int capy_schnorr_verify_batch_dev(int d, size_t n, const uint8_t *pub_xy, const uint8_t *msgs, const uint64_t *offsets,
                                  uint64_t uniform_len, uint64_t msg_stride, const uint8_t *h, const uint8_t *z_be,
                                  int32_t *status, void *stream);
int capy_key_encrypt_batch_dev(int d, size_t n, const uint8_t *pub_xy, const uint8_t *k_rand, uint8_t *msgs,
                               const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride, uint8_t *z_xy,
                               uint8_t *tags, void *stream);
int capy_key_decrypt_batch_dev(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                               const uint8_t *z_xy, uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len,
                               uint64_t msg_stride, const uint8_t *tags, int32_t *status, void *stream);

/* The same calls with per-call options (r04; opt may be NULL = no options).  Host-buffer forms: same arguments + opt.
 * Device-buffer forms: the stream comes from opt->stream. */
int capy_ed448_scalarmul_batch_ex(size_t n, const uint8_t *scalars_be, const uint8_t *points_xy, uint8_t *out_xy,
                                  const capy_call_options *opt);
int capy_ed448_scalarmul_batch_dev_ex(size_t n, const uint8_t *scalars_be, const uint8_t *points_xy, uint8_t *out_xy,
                                      const capy_call_options *opt);
int capy_ed448_basemul_batch_ex(size_t n, const uint8_t *scalars_be, uint8_t *out_xy, const capy_call_options *opt);
int capy_ed448_basemul_batch_dev_ex(size_t n, const uint8_t *scalars_be, uint8_t *out_xy, const capy_call_options *opt);
int capy_keypair_batch_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets, uint8_t *pub_xy,
                          const capy_call_options *opt);
int capy_schnorr_sign_batch_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                               const uint8_t *msgs, const uint64_t *offsets, uint8_t *h, uint8_t *z_be,
                               const capy_call_options *opt);
int capy_schnorr_verify_batch_ex(int d, size_t n, const uint8_t *pub_xy, const uint8_t *msgs, const uint64_t *offsets,
                                 const uint8_t *h, const uint8_t *z_be, int32_t *status, const capy_call_options *opt);
int capy_key_encrypt_batch_ex(int d, size_t n, const uint8_t *pub_xy, const uint8_t *k_rand, uint8_t *msgs,
                              const uint64_t *offsets, uint8_t *z_xy, uint8_t *tags, const capy_call_options *opt);
int capy_key_decrypt_batch_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                              const uint8_t *z_xy, uint8_t *msgs, const uint64_t *offsets, const uint8_t *tags,
                              int32_t *status, const capy_call_options *opt);
int capy_keypair_batch_dev_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                              uint8_t *pub_xy, const capy_call_options *opt);
int capy_schnorr_sign_batch_dev_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                   const uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride,
                                   uint8_t *h, uint8_t *z_be, const capy_call_options *opt);
int capy_schnorr_verify_batch_dev_ex(int d, size_t n, const uint8_t *pub_xy, const uint8_t *msgs, const uint64_t *offsets,
                                     uint64_t uniform_len, uint64_t msg_stride, const uint8_t *h, const uint8_t *z_be,
                                     int32_t *status, const capy_call_options *opt);
int capy_key_encrypt_batch_dev_ex(int d, size_t n, const uint8_t *pub_xy, const uint8_t *k_rand, uint8_t *msgs,
                                  const uint64_t *offsets, uint64_t uniform_len, uint64_t msg_stride, uint8_t *z_xy,
                                  uint8_t *tags, const capy_call_options *opt);
int capy_key_decrypt_batch_dev_ex(int d, size_t n, const uint8_t *pws, size_t pw_len, const uint64_t *pw_offsets,
                                  const uint8_t *z_xy, uint8_t *msgs, const uint64_t *offsets, uint64_t uniform_len,
                                  uint64_t msg_stride, const uint8_t *tags, int32_t *status, const capy_call_options *opt);

/* ------------------------------------------------------------------ measurement helpers */

/* Tuning / test knob: GPU lanes per sponge. 0 = automatic (a wave per item for digest batches of at most two items
 * per SIMD, 2 lanes for batches of at most 32 items per SIMD, the rotating one-/two-lane schedule for uniform
 * digest batches between 32 and 64 items per SIMD, else 1), 1 or 2 = forced, 3 = the rotating schedule wherever it is
 * eligible. Results are identical either way. Process-wide and not synchronised with calls in flight: set it before
 * the threads that use the library start, not while they run. */
int capy_set_sponge_lanes(int lanes);

/* Which kernel a uniform, 8-byte aligned capy_sha3_batch_dev() launch of this shape takes on the current device
 * (so that a profile can be read against the right kernel name): *kind = 1 sponge_kernel<RW,false,0>,
 * 2 sponge_kernel_k2<RW,0>, 3 sponge_mixed_kernel<RW> launched *phases times, 4 sponge_kernel<RW,true,0>,
 * 5 a full-chip head on sponge_kernel<RW,false,0> plus a remainder on kind 2 or 3 (*phases = launches in all),
 * 7 sponge_uniform_kernel<RW> (more than 128 items per SIMD, wave-uniform framing: csrc/sponge_uniform.h; long messages just
 *   above a whole number of waves per SIMD run as a sequence of time-sliced launches of it, still reported as 7),
 * 8 sponge_rot_kernel<RW> launched *phases times (between 64 and 128 items per SIMD, long messages: csrc/sponge_rot.h),
 * 10 sponge_il_digest_kernel<RW> (one item per wave, bit-interleaved Keccak lanes: batches of up to two items per SIMD, any
 *   message length; csrc/sponge_wide_il.h).  (6 was the two-items-per-wave kernel of r02-r04.) */
int capy_sha3_launch_plan(int d, size_t n, uint64_t uniform_len, uint64_t msg_stride, int *kind, int *phases);
/* Fill a device buffer with the harness PRNG (SplitMix64 counter mode, seed + 8-byte word index). */
int capy_fill_random_dev(uint8_t *dst, uint64_t nbytes, uint64_t seed, void *stream);
/* Run `iters` back-to-back keccak-f[1600] on n lane-resident states (VALU ceiling probe); returns the
 * XOR of all state words through *checksum_dev (8 bytes, device) so the work is not elided. */
int capy_keccak_valu_probe_dev(uint64_t n_states, uint32_t iters, uint64_t *checksum_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif
