/*
 * bppp.h -- C ABI of the MI355X-native Bulletproofs++ u64 range-proof engine (libbppp_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of distributed-lab/bp-pp 0.1.1: batch verification (and, next,
 * batch proving) of independent u64 range proofs that share one generator set.  The reference has no FFI of its
 * own (it is a pure-Rust crate); each entry point below names the reference interface it replaces
 * (file:line under /root/reference/src) and INTEGRATION.md shows the Rust `extern "C"` facade a maintainer
 * would add.  Plain pointers and sizes only; no torch / HIP types in the signatures.
 *
 * Encodings (identical to the byte strings k256 0.13.3 produces):
 *   point   64 B  affine big-endian x || y; the identity is 64 zero bytes
 *                 (ProjectivePoint::to_affine() + to_encoded_point(false) without the 0x04 tag)
 *   scalar  32 B  big-endian canonical (Scalar::to_bytes / from_repr)
 *   u64 proof 928 B = 13 points + 3 scalars, in this order (reciprocal.rs:30-33, circuit.rs:24-33):
 *                 c_l, c_r, c_o, c_s, r[0..3], x[0..3], reciprocal.r, l[0], l[1], n[0]
 *
 * Threading: a context is bound to one GPU and one HIP stream.  Every call on a context holds the context's lock, so calls from
 * several host threads are safe and run one after the other (the reference's types are Send + Sync; SURVEY 8b "Threading"); for
 * concurrency use one context per thread -- bppp_ctx_create_shared gives each its own workspace over one table set.  Different
 * contexts (e.g. one per GPU, one process per GPU) are independent.  The library never retains caller pointers.
 *
 * There is NO CPU fallback: without a usable gfx950 device every compute entry point returns BPPP_ERR_NO_DEVICE.
 */
#ifndef BPPP_H
#define BPPP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPPP_POINT_BYTES 64
#define BPPP_SCALAR_BYTES 32
#define BPPP_U64_PROOF_BYTES 928
#define BPPP_U64_TRACE_BYTES 704
#define BPPP_G_VEC_FULL_SZ 16 /* u64_proof.rs:12 */
#define BPPP_H_VEC_FULL_SZ 32 /* u64_proof.rs:14 */

/* return codes */
#define BPPP_OK 0
#define BPPP_ERR_NO_DEVICE (-1)   /* no HIP device / not gfx950 / HIP runtime failed to initialise */
#define BPPP_ERR_INVALID_ARG (-2) /* null pointer, bad size, unsupported window width */
#define BPPP_ERR_HIP (-3)         /* a HIP call failed; see bppp_last_error() */
#define BPPP_ERR_ENCODING (-4)    /* a generator is not a valid curve point */
#define BPPP_ERR_NOMEM (-5)
/* BPPP_ERR_RCCL (-6) is defined with the device groups below */
#define BPPP_ERR_CLOSED (-7)      /* a single-proof call (bppp_u64_*_one) arrived while its context was being destroyed */

/* per-proof status written by the verify kernels (0 = fine) */
#define BPPP_ST_BAD_ENCODING 1 /* off-curve point / coordinate >= p / scalar >= n: k256 deserialisation would fail */
#define BPPP_ST_DEGENERATE 2   /* the reference would panic here: challenge >= n (transcript.rs:13) or zero inverse
                                  (circuit.rs:192,196, reciprocal.rs:181, util.rs:119) */

#if defined(__GNUC__)
#define BPPP_API __attribute__((visibility("default")))
#else
#define BPPP_API
#endif

typedef struct bppp_ctx bppp_ctx;

/* Concurrency: a context owns its workspace and streams; calls on it are serialized by its lock (the *_device variants hold it while
 * they enqueue: their kernels are ordered on the context's stream, and the workspace they share is reused in that order).  Calls
 * are asynchronous on the context's stream only where the entry point says so (*_device variants); the host-buffer variants return
 * after the results have been copied back. */

/* U64RangeProofProtocol { g, g_vec[16], h_vec[32] } (u64_proof.rs:19-28) bound to GPU `device`.
 * Builds the fixed-base tables for the 49 generators on the GPU.  fb_window_bits: 0 = the library's choice from the HBM that is FREE on
 * the device when the context is created (hipMemGetInfo): the FEWEST windows (= table additions) per scalar whose tables fit.  The
 * windows of a scalar come in two widths sized to the bit -- a signed recoding needs 258 bits of windows, so n windows are
 * floor(258 / n)-bit windows with 258 mod n of them one bit wider -- and a layout is named by its WINDOW CODE Wb + 100 ka (ka windows of
 * Wb + 1 bits at the low end, then windows of Wb bits): 523 = 11 windows (5 x 24 + 6 x 23 bits), 4.3 GB per generator; 621 = 12 windows,
 * 1.2 GB; 1119 = 13, 403 MB; 618 = 14, 168 MB; 317 = 15, 75 MB; 216 = 16, 38 MB; ... 208 = 32 windows, 278 KB.  A table fits when it
 * takes at most 76 % of the free memory, leaves 50 GB of it (or half, if less is free) and its build scratch fits beside it; should
 * the allocation fail anyway, the next count is tried.  On an otherwise empty MI355X the 49 generators get 11 windows (210 GB: 726 table
 * additions per proof); when that does not fit, this generator shape takes a table in TWO regions -- g and g_vec, the 17 generators that
 * both fixed-base sums of a verify run over, at 11 windows, h_vec at 12: 112 GB, 758 additions ("fb_window_bits_hi" = 523, "fb_hi_bases"
 * = 17, "fb_window_bits" = 621) -- and below that the general rule (BPPP_NO_WIDE_TABLES=1 / BPPP_NO_MIXED_WINDOWS=1 in the environment
 * skip the first / the first two).  bppp_ctx_get_option "fb_window_bits" tells which layout was taken.
 * An explicit fb_window_bits is a window code as above (any that tiles 258 bits with windows of 8 .. 24 bits) or one of the uniform
 * widths 22, 20, 19, 18, 10 (signed digits: ceil(257 / W) windows of 2^(W-1) entries), 16, 8, 4 (unsigned digits, 256 / W windows of
 * 2^W - 1 entries; 4 is the layout of the "ct_prover" tables).  For bppp_wnla_ctx_create the general rule over 1 + ng + nh generators
 * (769 generators: 14 windows, 129 GB).  bppp_ctx_save_tables / bppp_ctx_create_from_tables keep a built table set of ONE region (a
 * uniform width, or a two-width window code) as a file. */
BPPP_API int bppp_ctx_create(bppp_ctx** out, const uint8_t g[64], const uint8_t* g_vec /* 16 x 64 */,
                    const uint8_t* h_vec /* 32 x 64 */, int device, int fb_window_bits);
/* Destroys the context.  Calls that are INSIDE the library when it is called are waited for (the single-proof entry points return
 * BPPP_ERR_CLOSED to callers that arrive while it runs); once it has RETURNED the handle is dead: starting a new call on it is the
 * caller's error, as with any freed object -- the caller orders its last call before the destroy. */
BPPP_API void bppp_ctx_destroy(bppp_ctx* ctx);

/* Run the context's kernels on a caller-owned hipStream_t (passed as void*); NULL restores the context's own stream.  The
 * context's own streams are created non-blocking: they are NOT ordered against the NULL (legacy default) stream, and the NULL
 * stream itself cannot be selected here (its handle is the NULL pointer).  A caller that needs its own work ordered with the
 * *_device entry points -- e.g. the RCCL all-reduce of reject_count -- passes the stream that work runs on. */
BPPP_API int bppp_ctx_set_stream(bppp_ctx* ctx, void* hip_stream);
/* Tunables.  "rlc_superchunk": proofs per superchunk of the bucket (Pippenger) stage of the RLC mode below -- 0 switches the stage
 * off (chunks of 8 only), otherwise a multiple of 8 in [64, 8192]; default 4096.  "host_chunk": the host-buffer verify entry points
 * (bppp_u64_verify_batch, bppp_u64_verify_batch_rlc) run a batch of more than 1.5 x host_chunk proofs in parts and upload part k + 1 on a
 * second stream while part k is being verified (proofs are independent: the results are those of one call).  The first part is
 * host_chunk proofs -- the one upload nothing hides -- and every next one 7 times its predecessor, what PCIe moves while a part is
 * verified (2^20 proofs = 2^17 + 7 * 2^17: the second part runs at the rate of a resident batch) --
 * a multiple of 64, >= 1024; default 131072 (one full grid of the per-proof kernels); 0 = upload the whole batch first.  "inject_alloc_fault" = k
 * (testing aid): the k-th device allocation this context makes from now on fails, so the call that makes it returns BPPP_ERR_NOMEM and
 * the context stays usable; 0 clears it.  "max_batch": the u64 verify entry
 * points run a batch of more than max_batch proofs as consecutive parts of max_batch on the same stream, which bounds the per-proof
 * workspace (about 30 KB per proof) whatever n is -- a multiple of 64, >= 1024; default: the largest power of two up to 2097152 (63 GB of
 * workspace) that takes at most 70 % of the HBM the tables left free at context creation (still 2097152 beside the 210 GB of tables
 * an empty MI355X gets: 288 GiB are 309 GB).  A first guess only: when the workspace of a part cannot be allocated after all (memory
 * taken since by another context or process), the call releases what it holds, HALVES max_batch -- down to 4096 -- and runs the part
 * again; BPPP_ERR_NOMEM comes back only when even that does not fit.  Contexts from bppp_ctx_create_shared start from the parent's value.
 * "generic_parts" = 0 (default: by size -- today always one) | 1 .. 4: bppp_reciprocal_verify_batch_device runs a call as that many
 * contiguous parts on as many streams (an A/B switch: round 6 measured it and found one part best up to 2^17 instances of configs[4]'s
 * shape, two parts 2 % ahead at 2^18 -- what two 2^17 calls cost); "generic_stagger" = 0 | 1 (default) | 2 | 3: the parts' chains start
 * together, or each behind the one before's phase 1 / C0 stage / rounds.
 * "rlc_chunk" = 8 | 32 | 0 (default): in the RLC modes of the u64 verifier, the proofs per chunk of the stage behind the bucket stage; 0 = per call,
 * from what the previous RLC call on this context rejected -- chunks of 32 while at most one proof in 256 was bad, and the bucket
 * stage's superchunks halved (or the stage skipped) when most of them would hold a bad proof and fail ("rlc_superchunk" set explicitly
 * is taken as it is); "rlc_history" = 0 forgets that rate.  bppp_ctx_get_option reads "last_rlc_superchunk" / "last_rlc_chunk" (what
 * the last call used) and "rlc_has_history" (0 / 1) and "rlc_reject_ppm" (the rate the next one will plan with, parts per million).  Accept bits never
 * depend on these choices.
 * "ct_prover" = 1: the provers' and the committer's sums over SECRET scalars -- bppp_u64_prove_* and bppp_u64_commit_value_batch (x, s,
 * the reciprocals and every blinding draw, i.e. V, r_com, c_o, c_l, c_r, c_s); bppp_reciprocal_prove_batch* (the commitment to the
 * reciprocals and the circuit stage below it); bppp_circuit_prove_batch* (c_l, c_r, c_o, c_s and the prover-side commitment of the
 * blinded vectors); bppp_wnla_prove_batch* called on its own (X and R of every round: there l and n are the caller's secrets) --
 * run in a form with no secret-dependent address, branch or instruction count.  NOT covered: bppp_msm_batch and bppp_wnla_commit_batch
 * (plain sums with no notion of what is secret), and the WNLA stage INSIDE the u64 / reciprocal / circuit provers, whose vectors the
 * argument folds and reveals by design.  The form:
 * (4-bit windows over a 3 MB table, every entry of every window read and selected by mask, complete addition law), as k256 does for
 * the reference (reciprocal.rs:88-95,118); the default (0) gathers one table entry per window at an address the digit selects, which
 * is a memory-access side channel towards whoever shares the GPU.  The proofs are byte-identical either way; the cost is reported in
 * bench.py's prove_2pow14.  INTEGRATION.md section 7 lists what is secret, what is public, and what the mode covers.
 * "coalesce_max" (1 .. 65536, default 1024), "coalesce_us" (0 .. 1000000, default 100), "coalesce_lanes" (1 .. 8, default 2): the
 * single-proof front end below (bppp_u64_verify_one / bppp_u64_prove_one); changing one drains the running front end. */
BPPP_API int bppp_ctx_set_option(bppp_ctx* ctx, const char* name, long value);
/* Reads a tunable back, or one of the read-only facts "fb_window_bits" (the width in use: the library's choice when the context was
 * created with 0), "device", "n_generators", "last_verify_plan" / "last_prove_plan" (the plan code -- see bppp_u64_plan -- of the
 * context's last u64 verify / prove call, or of the last part of a call that ran in parts; 0 before the first).  Negative =
 * BPPP_ERR_INVALID_ARG (unknown name). */
BPPP_API long bppp_ctx_get_option(bppp_ctx* ctx, const char* name);
/* Which kernels a u64 verify (prove = 0) or prove (prove = 1) call of n proofs runs on a device of n_simds SIMDs (CUs x 4; MI355X: 1024):
 * the size decides among seven (six) launch sequences, from a wavefront per sum for a handful of proofs to one lane per proof from 2^17
 * on; where a call fills the chip's wavefront slots only once or twice (2^17 .. 2.25 x 2^17 proofs on an MI355X, except sizes whose last
 * generation of wavefronts would be 30 .. 70 % full) it runs as TWO half-batch chains of kernels on two streams (twin=2: the code then
 * describes what each chain runs), and up to 2^17 proofs the one-lane sums pace their wave priority (pace=1); beyond, how many proofs
 * share one field inversion in the verifier's table build and rounds (shared_inv = 8, 16 from 2^20; csrc/plan_core.h lists the regimes
 * with their thresholds and the measurements behind them).  A pure function -- no device, no context; flags: bit 0 = RLC mode (verify) /
 * "ct_prover" (prove), bit 1 = per-kernel timing on.  Returns the plan as a non-negative code whose fields bppp_plan_describe spells out
 * ("phase1=wg4 tables=beside/1 fb=l8 c0var=small round=small tail_beside=1 small=1 split=0 twin=1 pace=0 shared_inv=0"), or BPPP_ERR_INVALID_ARG.  A context's
 * diagnostic environment switches (BPPP_NO_SMALL_KERNELS etc.) are not visible here; "last_verify_plan" reports what really ran. */
BPPP_API long bppp_u64_plan(int prove, size_t n, int n_simds, int flags);
/* Text form of a plan code into buf (NUL-terminated, at most cap bytes); returns the length the full text needs, as snprintf does. */
BPPP_API int bppp_plan_describe(long code, int prove, char* buf, size_t cap);
/* Block the calling host thread until everything queued by this context (current stream + its helper stream) has finished. */
BPPP_API int bppp_ctx_synchronize(bppp_ctx* ctx);

/* U64RangeProofProtocol::verify (u64_proof.rs:42-54) for n independent proofs, fresh
 * `merlin::Transcript::new(label)` per proof (benches/range_proof.rs:47).
 * accept[i] = 1 iff the reference's verify returns true for (commitments[i], proofs[i]); status[i] (optional, may be
 * NULL) holds BPPP_ST_* flags; a proof with a non-zero status is never accepted.  Host pointers; copies in and out. */
BPPP_API int bppp_u64_verify_batch(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n,
                          const uint8_t* commitments /* n x 64 */, const uint8_t* proofs /* n x 928 */,
                          uint8_t* accept /* n */, int32_t* status /* n or NULL */);

/* Same, but every buffer is DEVICE memory on the context's GPU (inputs already resident in HBM).  Asynchronous on
 * the context's stream; d_trace (optional, n x 704) receives per-proof intermediates: 10 challenges
 * (e, rho, lambda, beta, delta, tau, y1..y4; 32 B each) then 6 points (V+r, C0..C4; 64 B each).
 * reject_count (optional, device int32[1]) receives the number of proofs with accept == 0 -- the value a multi-GPU
 * caller all-reduces over RCCL. */
BPPP_API int bppp_u64_verify_batch_device(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n,
                                 const void* d_commitments, const void* d_proofs, void* d_accept, void* d_status,
                                 void* d_trace, void* d_reject_count);

/* Optional batch mode: the same pipeline, but the per-proof final check (the 49-base MSM of wnla.rs:80-82, ~26 % of the work) is
 * replaced by checks of random linear combinations with secret 128-bit weights derived from `seed` (Keccak PRF of seed || proof
 * index), in two stages: (1) superchunks of 4096 proofs (bppp_ctx_set_option "rlc_superchunk"): sum_j w_j C4_j by bucket
 * accumulation -- `util::vector_mul` over ProjectivePoint (util.rs:46-60) as a Pippenger MSM with LDS-staged buckets -- against ONE
 * 49-base MSM of the combined scalars; (2) the proofs of a superchunk that fails are re-checked in chunks of 8 (one short scalar
 * multiplication per proof, one 49-base MSM per chunk), and the chunks that fail THAT are checked exactly, proof by proof.  So
 * accept[] is still per proof; it equals exact mode's except that a chunk holding an invalid proof passes with probability <=
 * 2^-128.  The seed must be unpredictable to whoever produced the proofs and chosen after they are fixed (e.g. 32 bytes of OS
 * randomness per call).  Everything up to and including the four WNLA rounds -- every transcript challenge and hashed commitment
 * -- is computed exactly as in exact mode. */
BPPP_API int bppp_u64_verify_batch_rlc_device(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n,
                                              const void* d_commitments, const void* d_proofs, void* d_accept, void* d_status,
                                              void* d_reject_count, const uint8_t seed[32]);
/* the same with host buffers (as bppp_u64_verify_batch) */
BPPP_API int bppp_u64_verify_batch_rlc(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                                       const uint8_t* proofs, uint8_t* accept, int32_t* status, const uint8_t seed[32]);

/* The same verify over the reference's WIRE content: what `reciprocal::SerializableProof` / `circuit::SerializableProof`
 * (reciprocal.rs:37-41, circuit.rs:37-46) carry -- k256 `AffinePoint`s, as the 33 bytes of `GroupEncoding::to_bytes`
 * (SEC1 compressed 02|03 || x; the identity is 33 zero bytes -- the same bytes transcript.rs:7 hashes), and 32-byte
 * big-endian scalars.  (serde writes the identity as the ONE byte 0x00 instead -- JSON "00"; a caller holding serde output
 * widens that to 33 zero bytes, as bp_pp_amd/wire.py and the Rust facade do.)  proofs: n x 525 bytes (13 x 33 in the order
 * c_l, c_r, c_o, c_s, r[0..3], x[0..3], reciprocal.r, then l[0], l[1], n[0]); commitments: n x 33 bytes.  Points are
 * decompressed on the device (one square root in Fp per point); a point k256's from_bytes would reject (bad tag, x >= p,
 * not on the curve -- including 02 || 0, whose x^3 + 7 = 7 is a non-residue) yields BPPP_ST_BAD_ENCODING for that proof.  Host pointers / device pointers as above. */
#define BPPP_U64_PROOF_SEC1_BYTES 525
#define BPPP_POINT_SEC1_BYTES 33
BPPP_API int bppp_u64_verify_batch_sec1(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n,
                                        const uint8_t* commitments /* n x 33 */, const uint8_t* proofs /* n x 525 */,
                                        uint8_t* accept /* n */, int32_t* status /* n or NULL */);
BPPP_API int bppp_u64_verify_batch_sec1_device(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n,
                                               const void* d_commitments, const void* d_proofs, void* d_accept, void* d_status,
                                               void* d_trace, void* d_reject_count);

/* U64RangeProofProtocol::prove (u64_proof.rs:57-82) for n independent values, fresh `Transcript::new(label)` per proof.
 * rnd holds, per proof, the 52 scalars the reference draws with `Scalar::generate_biased(rng)` in its draw order
 * (reciprocal.rs:121 r_blind | circuit.rs:264-298 ro x7, rl x6, rr x5 | circuit.rs:371 ls x17 | circuit.rs:372 ns x16), so
 * the output is byte-identical to the reference prover driven by the same RNG stream.  Also returns the commitments
 * x*g + s*h_vec[0].  status (optional): BPPP_ST_BAD_ENCODING for a non-canonical input scalar, BPPP_ST_DEGENERATE where the
 * reference would panic.  Host pointers. */
BPPP_API int bppp_u64_prove_batch(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x,
                                  const uint8_t* s /* n x 32 */, const uint8_t* rnd /* n x 52 x 32 */,
                                  uint8_t* proofs /* n x 928 */, uint8_t* commitments /* n x 64 */, int32_t* status /* n or NULL */);
/* Same with DEVICE buffers, asynchronous on the context's stream. */
BPPP_API int bppp_u64_prove_batch_device(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, const void* d_x,
                                         const void* d_s, const void* d_rnd, void* d_proofs, void* d_commitments, void* d_status);

/* The same with the output in the crate's wire format -- what serde gives for SerializableProof (wnla.rs:33-61, circuit.rs:36-76,
 * reciprocal.rs:37-59) and what bppp_u64_verify_batch_sec1 takes: 13 SEC1-compressed points + 3 scalars = 525 bytes per proof,
 * 33-byte commitments (the identity as 33 zero bytes).  Compressed on the device. */
BPPP_API int bppp_u64_prove_batch_sec1(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x,
                                       const uint8_t* s /* n x 32 */, const uint8_t* rnd /* n x 52 x 32 */,
                                       uint8_t* proofs525 /* n x 525 */, uint8_t* commitments33 /* n x 33 */, int32_t* status /* n or NULL */);
BPPP_API int bppp_u64_prove_batch_sec1_device(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, const void* d_x,
                                              const void* d_s, const void* d_rnd, void* d_proofs525, void* d_commitments33, void* d_status);

/* ---- the reference's own calling pattern: ONE proof per call, from as many host threads as the caller likes ----
 * `U64RangeProofProtocol::verify(&self, v, proof, t) -> bool` (u64_proof.rs:42-54) and `::prove(&self, x, s, t, rng) -> Proof`
 * (u64_proof.rs:57-82) take one proof; the crate's types are Send + Sync, so a service calls them from N threads at once
 * (benches/range_proof.rs:47-50 is the single-threaded form).  A GPU call of one proof costs a full dependent chain (2.3 ms) however
 * empty the chip is, so these entry points do not launch anything themselves: the request joins the context's open batch and the
 * calling thread sleeps; dispatcher threads of the context run what has gathered -- everything that arrived within "coalesce_us"
 * microseconds of the batch's first request, or "coalesce_max" requests, whichever comes first -- as ONE call of the batched
 * verifier / prover on pinned staging, up to "coalesce_lanes" batches overlapping on the GPU, and wake each caller with its own row.
 * The answer is exactly the batched entry point's for that row (same accept bit, same status, same bytes), whoever shared the batch:
 * every row has its own transcript, so callers may use different labels or transcripts, and a malformed proof flags only itself.
 * Only a failure of the batched call as a whole (BPPP_ERR_NOMEM, BPPP_ERR_HIP) is shared: it is every caller's return code.
 * Blocking, callable from any number of threads on the same context (the context lock is NOT held while waiting).  The front end
 * (dispatcher threads, `coalesce_lanes` contexts over this context's tables, staging for coalesce_max rows) is created by the first
 * call.  bppp_ctx_destroy drains it: calls already inside complete normally, calls arriving later return BPPP_ERR_CLOSED.
 *   label form:       the proof's transcript starts as Transcript::new(label) (what every call site of the reference does);
 *   transcript form:  `state` is the caller's merlin transcript (203 bytes, see bppp_u64_verify_batch_transcript below), advanced in
 *                     place exactly as the reference's `t: &mut Transcript` is (left untouched when the call fails or the proof is
 *                     flagged BPPP_ST_BAD_ENCODING); a state merlin cannot be in is refused with BPPP_ERR_INVALID_ARG.
 * *accept = 1 iff the reference's verify returns true; *status (optional) = BPPP_ST_* flags. */
BPPP_API int bppp_u64_verify_one(bppp_ctx* ctx, const uint8_t* label, size_t label_len, const uint8_t commitment[64],
                                 const uint8_t proof[928], uint8_t* accept, int32_t* status /* or NULL */);
BPPP_API int bppp_u64_verify_one_transcript(bppp_ctx* ctx, uint8_t state[203], const uint8_t commitment[64], const uint8_t proof[928],
                                            uint8_t* accept, int32_t* status /* or NULL */);
/* rnd: the 52 scalars of bppp_u64_prove_batch (52 x 32 bytes, the reference's draw order); proof and commitment = x*g + s*h_vec[0] out. */
BPPP_API int bppp_u64_prove_one(bppp_ctx* ctx, const uint8_t* label, size_t label_len, uint64_t x, const uint8_t s[32],
                                const uint8_t* rnd /* 52 x 32 */, uint8_t proof[928], uint8_t commitment[64], int32_t* status /* or NULL */);
BPPP_API int bppp_u64_prove_one_transcript(bppp_ctx* ctx, uint8_t state[203], uint64_t x, const uint8_t s[32], const uint8_t* rnd /* 52 x 32 */,
                                           uint8_t proof[928], uint8_t commitment[64], int32_t* status /* or NULL */);
/* ReciprocalRangeProofProtocol::verify(&self, commitment, proof, t) (reciprocal.rs:98-107) the same way: ONE instance per call at the
 * runtime dimensions of bppp_reciprocal_verify_batch (context from bppp_wnla_ctx_create; proof layout as there), gathered with the other
 * threads' calls of the same shape into one batched call.  A context serves at most four different shapes this way. */
BPPP_API int bppp_reciprocal_verify_one(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t dim_nd, size_t dim_np,
                                        const uint8_t commitment[64], const uint8_t* proof, size_t rounds, size_t nl, size_t nn,
                                        uint8_t* accept, int32_t* status /* or NULL */);
BPPP_API int bppp_reciprocal_verify_one_transcript(bppp_ctx* ctx, uint8_t state[203], size_t dim_nd, size_t dim_np, const uint8_t commitment[64],
                                                   const uint8_t* proof, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                                                   int32_t* status /* or NULL */);
/* Counters of the front end since it was created: out = {requests, batches, largest batch, batches sealed full, batches sealed by the
 * deadline, microseconds its dispatchers spent inside batched calls, microseconds they waited for callers' row copies, 0}; which = 0
 * verify, 1 prove.  All zero before the first call. */
BPPP_API int bppp_ctx_get_coalesce_stats(bppp_ctx* ctx, int which, uint64_t out[8]);

/* U64RangeProofProtocol::commit_value (u64_proof.rs:37-39): out[i] = x[i]*g + s[i]*h_vec[0], host pointers. */
BPPP_API int bppp_u64_commit_value_batch(bppp_ctx* ctx, size_t n, const uint64_t* x, const uint8_t* s /* n x 32 */,
                                uint8_t* out /* n x 64 */);

/* ---- the crate's `wnla` API surface (wnla.rs:12-19, 66-121), generic sizes ----
 * WeightNormLinearArgument { g, g_vec[ng], h_vec[nh], c, rho, mu }: the generators are the context (shared by the batch),
 * c (nh scalars, zero-padded as circuit.rs:237-239 does), rho, mu are per instance.  A context created here serves the wnla
 * entry points only (the u64 entry points require ng = 16, nh = 32, which bppp_ctx_create builds). */
BPPP_API int bppp_wnla_ctx_create(bppp_ctx** out, const uint8_t g[64], const uint8_t* g_vec /* ng x 64 */, size_t ng,
                                  const uint8_t* h_vec /* nh x 64 */, size_t nh, int device, int fb_window_bits);
/* The same with a TABLE BUDGET: fb_table_budget_bytes > 0 bounds what the fixed-base tables may take, whichever way their layout is
 * chosen (any generator shape, the u64 one included: ng = 16, nh = 32).  With fb_window_bits = 0 the library takes the fewest windows
 * whose tables fit both the budget and the free-memory rule above; an explicit fb_window_bits whose tables exceed the budget is
 * BPPP_ERR_INVALID_ARG.  0 = no budget (bppp_ctx_create / bppp_wnla_ctx_create: on an otherwise empty MI355X the u64 shape then takes
 * 210 GB for its last 2.7 % of throughput -- INTEGRATION.md 5 has the cost curve).  What a context took and why can be read back:
 * bppp_ctx_get_option "fb_table_bytes", "fb_table_budget_bytes", "fb_windows" (table additions per scalar), "fb_window_bits_widest".
 * Contexts from bppp_ctx_create_shared inherit the parent's budget, max_batch and host_chunk. */
BPPP_API int bppp_wnla_ctx_create_budget(bppp_ctx** out, const uint8_t g[64], const uint8_t* g_vec /* ng x 64 */, size_t ng,
                                         const uint8_t* h_vec /* nh x 64 */, size_t nh, int device, int fb_window_bits,
                                         uint64_t fb_table_budget_bytes);
/* WeightNormLinearArgument::commit (wnla.rs:66-72): out[i] = v*g + <h_vec, l_i> + <g_vec, n_i>, v = <c_i, l_i> + |n_i|^2_mu. */
BPPP_API int bppp_wnla_commit_batch(bppp_ctx* ctx, size_t n, const uint8_t* c /* n x nh x 32 */, const uint8_t* mu /* n x 32 */,
                                    const uint8_t* l /* n x nl x 32 */, size_t nl, const uint8_t* nvec /* n x nn x 32 */, size_t nn,
                                    uint8_t* out /* n x 64 */, int32_t* status /* n or NULL */);
/* WeightNormLinearArgument::verify (wnla.rs:75-121), fresh Transcript::new(label) per instance.  rounds = proof.x.len() =
 * proof.r.len() (a proof with different lengths is rejected by the reference at wnla.rs:76-78 before anything else);
 * proof_r / proof_x hold each instance's vectors in the reference's order (index rounds-1 is consumed first). */
BPPP_API int bppp_wnla_verify_batch(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n,
                                    const uint8_t* commitments /* n x 64 */, const uint8_t* c /* n x nh x 32 */,
                                    const uint8_t* rho /* n x 32 */, const uint8_t* mu /* n x 32 */, size_t rounds,
                                    const uint8_t* proof_r /* n x rounds x 64 */, const uint8_t* proof_x /* n x rounds x 64 */,
                                    const uint8_t* proof_l /* n x nl x 32 */, size_t nl, const uint8_t* proof_n /* n x nn x 32 */,
                                    size_t nn, uint8_t* accept /* n */, int32_t* status /* n or NULL */);
/* The same over DEVICE buffers (every pointer is device memory of the context's GPU, layouts as above; d_status may be NULL), asynchronous
 * on the context's stream -- the resident form the throughput of `wnla::verify` is measured on (bench.py --workload wnla). */
BPPP_API int bppp_wnla_verify_batch_device(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                                          const void* d_c, const void* d_rho, const void* d_mu, size_t rounds, const void* d_proof_r,
                                          const void* d_proof_x, const void* d_proof_l, size_t nl, const void* d_proof_n, size_t nn,
                                          void* d_accept, void* d_status);

/* out[i] = sum_j scalars[i][j] * B[base_index[j]] for n independent rows, B = the context's generators in table order
 * (0 = g, 1 .. NG = g_vec || g_vec_, NG + 1 .. NG + NH = h_vec || h_vec_), base_index strictly increasing.  The crate's commit
 * functions are instances: ArithmeticCircuit::commit(v, s) (circuit.rs:146-151) = {0: v[0], NG+1: s, NG+10 ..: v[1..]},
 * ReciprocalRangeProofProtocol::commit_value (reciprocal.rs:88-90) = {0: x, NG+1: s}, ::commit_poles (reciprocal.rs:93-95)
 * = {NG+1: s, NG+10 ..: r}.  A non-canonical scalar flags its row (status BPPP_ST_BAD_ENCODING, output = identity). */
BPPP_API int bppp_msm_batch(bppp_ctx* ctx, size_t n, size_t nterms, const int32_t* base_index /* nterms */,
                            const uint8_t* scalars /* n x nterms x 32 */, uint8_t* out /* n x 64 */, int32_t* status /* n or NULL */);

/* ReciprocalRangeProofProtocol::prove(commitment, witness, t, rng) (reciprocal.rs:110-146) for runtime dim_nd / dim_np, context
 * as for bppp_reciprocal_verify_batch.  Per instance: the value commitment (commit_value(x, s), e.g. from bppp_msm_batch), the
 * witness x, s (32-byte scalars), digits (dim_nd scalars: the base-dim_np digits of x), m (dim_np multiplicities), and the
 * prover's random scalars rnd in the reference's draw order: r_blind, then the circuit prover's 18 + (dim_nd + 1) + dim_nd
 * (20 + 2 dim_nd in all).  proofs: n x (64 (5 + 2 rounds) + 32 (nl + nn)), (rounds, nl, nn) = bppp_wnla_proof_shape(NH, NG),
 * in the layout bppp_reciprocal_verify_batch takes (for dim_nd = dim_np = 16 the 928-byte u64 proof). */
BPPP_API int bppp_reciprocal_prove_batch(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                         const uint8_t* commitments, const uint8_t* x, const uint8_t* s, const uint8_t* digits,
                                         const uint8_t* m, const uint8_t* rnd, uint8_t* proofs, int32_t* status /* n or NULL */);

/* ArithmeticCircuit (circuit.rs:95-139) shared by a batch, and ArithmeticCircuit::verify (circuit.rs:154-256) for n
 * independent (commitments, proof) instances of it.  dims = {dim_nm, dim_no, k, dim_nl, dim_nv, dim_nw} with the reference's
 * own relations dim_nl = dim_nv k, dim_nw = 2 dim_nm + dim_no; W_m (dim_nm x dim_nw), W_l (dim_nl x dim_nw), a_m, a_l are
 * row-major 32-byte big-endian scalars; the `partition` closure is given as four index tables (value = index into w_o, or -1
 * for None): part_lo / part_ll / part_lr (dim_nv entries each, PartitionType::{LO, LL, LR}) and part_no (dim_nm entries,
 * PartitionType::NO).  The context comes from bppp_wnla_ctx_create(g, g_vec || g_vec_, NG, h_vec || h_vec_, NH) with
 * NG >= dim_nm and NH >= dim_nv + 9 (both powers of two, as the WNLA stage needs).
 * Instance inputs: commitments n x k x 64 (the `v` slice of verify), proofs n x (64 (4 + 2 rounds) + 32 (nl + nn)) laid out
 *   c_l, c_r, c_o, c_s | r[rounds] | x[rounds] | l[nl] | n[nn];   the transcript starts as Transcript::new(label). */
typedef struct bppp_circuit bppp_circuit;
BPPP_API int bppp_circuit_create(bppp_ctx* ctx, bppp_circuit** out, const size_t dims[6], int f_l, int f_m, const uint8_t* W_m,
                                 const uint8_t* W_l, const uint8_t* a_m, const uint8_t* a_l, const int32_t* part_lo,
                                 const int32_t* part_ll, const int32_t* part_lr, const int32_t* part_no);
BPPP_API void bppp_circuit_destroy(bppp_circuit* circuit);
BPPP_API int bppp_circuit_verify_batch(bppp_ctx* ctx, const bppp_circuit* circuit, const uint8_t* label, size_t label_len, size_t n,
                                       const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn,
                                       uint8_t* accept /* n */, int32_t* status /* n or NULL */);
/* ... over DEVICE buffers (commitments n x k x 64, proofs, accept n, status n or NULL), asynchronous on the context's stream
 * (bench.py --workload circuit). */
BPPP_API int bppp_circuit_verify_batch_device(bppp_ctx* ctx, const bppp_circuit* circuit, const uint8_t* label, size_t label_len, size_t n,
                                             const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl, size_t nn,
                                             void* d_accept, void* d_status);

/* ArithmeticCircuit::prove(v, witness, t, rng) (circuit.rs:260-556) for n instances of a shared circuit (bppp_circuit_create).
 * Per instance: v_commitments k x 64 (the reference's `v` argument: circuit.commit of each witness.v[i], e.g. from
 * bppp_msm_batch), witness v (k x dim_nv scalars), s_v (k), w_l, w_r (dim_nm), w_o (dim_no), and the prover's random scalars
 * rnd in the reference's draw order -- 18 + dim_nv + dim_nm of them: r_o (7), r_l (6), r_r (5) (circuit.rs:264-298), then
 * l_s (dim_nv), n_s (dim_nm) (circuit.rs:371-372) -- so that the proof equals the CPU prover's for the same RNG stream.
 * proofs: n x (64 (4 + 2 rounds) + 32 (nl + nn)) with (rounds, nl, nn) = bppp_wnla_proof_shape(NH, NG), laid out as the
 * verifier takes them.  Non-canonical inputs flag the instance (status, zeroed proof). */
BPPP_API int bppp_circuit_prove_batch(bppp_ctx* ctx, const bppp_circuit* circuit, const uint8_t* label, size_t label_len, size_t n,
                                      const uint8_t* v_commitments, const uint8_t* v, const uint8_t* s_v, const uint8_t* w_l,
                                      const uint8_t* w_r, const uint8_t* w_o, const uint8_t* rnd, uint8_t* proofs,
                                      int32_t* status /* n or NULL */);

/* WeightNormLinearArgument::prove(commitment, t, l, n) (wnla.rs:125-190) for n instances sharing the context's generators
 * (bppp_wnla_ctx_create); c (|c| = nh), rho, mu, the commitment and the witness vectors l (nl entries) and n (nn entries) are
 * per instance, the transcript is Transcript::new(label).  The proof shape follows from nl and nn alone
 * (bppp_wnla_proof_shape: the recursion stops when |l| + |n| < 6): proof_r / proof_x are n x rounds x 64 in the reference's
 * vector order (last round first), proof_l n x nl_out x 32, proof_n n x nn_out x 32.  A non-canonical input scalar or an
 * undecodable commitment flags the instance (status BPPP_ST_BAD_ENCODING, zeroed proof); rho = 0 at some level, where the
 * reference unwrap()s, gives BPPP_ST_DEGENERATE. */
BPPP_API void bppp_wnla_proof_shape(size_t nl, size_t nn, size_t* rounds, size_t* nl_out, size_t* nn_out);
BPPP_API int bppp_wnla_prove_batch(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                                   const uint8_t* c, const uint8_t* rho, const uint8_t* mu, const uint8_t* l, size_t nl,
                                   const uint8_t* n_vec, size_t nn, uint8_t* proof_r, uint8_t* proof_x, uint8_t* proof_l,
                                   uint8_t* proof_n, int32_t* status /* n or NULL */);

/* ReciprocalRangeProofProtocol::verify (reciprocal.rs:98-107) for runtime dim_nd / dim_np (dim_np <= dim_nd + 1): e.g.
 * dim_nd = 256, dim_np = 16 -> |g_vec| = 256, |h_vec| + |h_vec_| = 512, 8 WNLA rounds (BASELINE configs[4]).  The context
 * comes from bppp_wnla_ctx_create(g, g_vec || g_vec_, NG, h_vec || h_vec_, NH) with NG >= dim_nd, NH >= dim_nd + 10.
 * Proof layout per instance, 64 (5 + 2 rounds) + 32 (nl + nn) bytes:
 *   c_l, c_r, c_o, c_s | r[rounds] | x[rounds] | reciprocal r | l[nl] | n[nn]       (for dim_nd = 16 this is the 928-byte u64 form)
 * dim_nd = dim_np = 16 over a context of 16 + 32 generators with the standard proof shape (rounds 4, nl 2, nn 1) IS the u64 protocol
 * (u64_proof.rs:42-54): such calls run on the u64 entry points' specialised kernels -- same verdicts and statuses, ten times the rate. */
BPPP_API int bppp_reciprocal_verify_batch(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                          const uint8_t* commitments /* n x 64 */, const uint8_t* proofs, size_t rounds, size_t nl,
                                          size_t nn, uint8_t* accept /* n */, int32_t* status /* n or NULL */);

/* ---- setup: the step before the path (benches/range_proof.rs:18-20, u64_proof.rs:37) ----
 * bppp_derive_generators: n reproducible generators with unknown discrete logarithms, indices first_index .. first_index + n - 1 of
 * the stream defined by `seed` (try-and-increment: x = SHAKE256(seed || "bppp-gen" || u32le(index) || u32le(counter)) as a
 * big-endian integer, first counter with x < p and x^3 + 7 a square; y = the even root).  64 bytes each, host only (no GPU needed).
 * The reference draws random points instead (`ProjectivePoint::random`); any 49 valid points work with every entry point here. */
BPPP_API int bppp_derive_generators(const uint8_t* seed, size_t seed_len, size_t first_index, size_t n, uint8_t* out /* n x 64 */);
/* The fixed-base tables of a context as a file, and a context created from such a file instead of from the generators (any of the
 * bppp_ctx_create / bppp_wnla_ctx_create shapes).  NOTE: the tables are BUILT on the GPU in 0.5 s (79 GB, 22-bit) to 3 s -- faster than any
 * disk or PCIe can deliver them -- so the file is for reproducibility and inspection, not for start-up time; what saves memory and
 * time is bppp_ctx_create_shared.  The file (magic "BPPPTAB3" for a uniform width, "BPPPTAB4" for a two-width window code; a table in two regions is not saveable) carries a checksum over its header (generator counts, window width), the generators and the table body, and the
 * stored generators are validated as bppp_ctx_create validates them: a truncated, damaged or foreign file makes
 * bppp_ctx_create_from_tables fail (BPPP_ERR_INVALID_ARG / BPPP_ERR_ENCODING) instead of yielding a verifier over wrong bases.  The
 * checksum is not a MAC: whoever can rewrite the file can rewrite it too -- rebuild the tables when the storage is not trusted. */
BPPP_API int bppp_ctx_save_tables(bppp_ctx* ctx, const char* path);
BPPP_API int bppp_ctx_create_from_tables(bppp_ctx** out, const char* path, int device);
/* A further context on the same GPU that shares `parent`'s generators and tables (read-only) and owns its streams and workspaces:
 * concurrent callers on one GPU, one copy of the tables.  `parent` must be destroyed after every context created from it. */
BPPP_API int bppp_ctx_create_shared(bppp_ctx** out, bppp_ctx* parent);

/* ---- the reference's `t: &mut Transcript` (u64_proof.rs:42, wnla.rs:75, circuit.rs:154; SURVEY 8b "Ownership") ----
 * Every entry point above takes a label and starts each proof from `Transcript::new(label)`, which is what all of the reference's
 * own call sites do.  A caller that has already appended to its transcript (binding a context, a transaction id, ...) passes the
 * transcript itself instead, as merlin's STROBE-128 state serialized to 203 bytes:
 *     bytes 0..199  the Keccak-f[1600] state (strobe.rs `state`), byte order as in memory
 *     byte  200     pos          byte 201  pos_begin          byte 202  cur_flags
 * states holds n_states = 1 transcript (shared by every proof of the batch) or n_states = n (one per proof).  states_out
 * (optional, n x 203) receives each proof's transcript as the reference's verify leaves it for the caller -- advanced through the
 * last `wnla_challenge` -- so the caller can keep using it; a proof flagged BPPP_ST_BAD_ENCODING (inputs k256 would not have
 * deserialized, so the reference's verify is never entered) gets its input state back unchanged.  A state merlin cannot be in
 * (pos >= 166) is refused: BPPP_ERR_INVALID_ARG from the host entry point, BPPP_ST_BAD_ENCODING per proof from the device one. */
#define BPPP_TRANSCRIPT_STATE_BYTES 203
BPPP_API int bppp_u64_verify_batch_transcript(bppp_ctx* ctx, size_t n, const uint8_t* states /* n_states x 203 */, size_t n_states,
                                              const uint8_t* commitments /* n x 64 */, const uint8_t* proofs /* n x 928 */,
                                              uint8_t* accept /* n */, int32_t* status /* n or NULL */,
                                              uint8_t* states_out /* n x 203 or NULL */);
/* the same with DEVICE buffers, asynchronous on the context's stream */
BPPP_API int bppp_u64_verify_batch_transcript_device(bppp_ctx* ctx, size_t n, const void* d_states, size_t n_states,
                                                     const void* d_commitments, const void* d_proofs, void* d_accept, void* d_status,
                                                     void* d_reject_count, void* d_states_out);
/* U64RangeProofProtocol::prove(x, s, t, rng) (u64_proof.rs:57-82) with the caller's transcripts, as for verify: states in (1 or n),
 * each proof's advanced state out (optional); rnd, proofs, commitments, status as in bppp_u64_prove_batch. */
BPPP_API int bppp_u64_prove_batch_transcript(bppp_ctx* ctx, size_t n, const uint8_t* states /* n_states x 203 */, size_t n_states,
                                             const uint64_t* x, const uint8_t* s /* n x 32 */, const uint8_t* rnd /* n x 52 x 32 */,
                                             uint8_t* proofs /* n x 928 */, uint8_t* commitments /* n x 64 */,
                                             int32_t* status /* n or NULL */, uint8_t* states_out /* n x 203 or NULL */);
BPPP_API int bppp_u64_prove_batch_transcript_device(bppp_ctx* ctx, size_t n, const void* d_states, size_t n_states, const void* d_x,
                                                    const void* d_s, const void* d_rnd, void* d_proofs, void* d_commitments,
                                                    void* d_status, void* d_states_out);
/* The generic verifiers with the caller's transcripts -- `t: &mut Transcript` of WeightNormLinearArgument::verify (wnla.rs:75),
 * ReciprocalRangeProofProtocol::verify (reciprocal.rs:98) and ArithmeticCircuit::verify (circuit.rs:154): states in (1 or n), each
 * instance's advanced state out (optional); every other argument as in the label forms above. */
BPPP_API int bppp_wnla_verify_batch_transcript(bppp_ctx* ctx, size_t n, const uint8_t* states, size_t n_states, const uint8_t* commitments,
                                               const uint8_t* c, const uint8_t* rho, const uint8_t* mu, size_t rounds, const uint8_t* proof_r,
                                               const uint8_t* proof_x, const uint8_t* proof_l, size_t nl, const uint8_t* proof_n, size_t nn,
                                               uint8_t* accept, int32_t* status, uint8_t* states_out);
BPPP_API int bppp_reciprocal_verify_batch_transcript(bppp_ctx* ctx, size_t n, const uint8_t* states, size_t n_states, size_t dim_nd,
                                                     size_t dim_np, const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl,
                                                     size_t nn, uint8_t* accept, int32_t* status, uint8_t* states_out);
BPPP_API int bppp_circuit_verify_batch_transcript(bppp_ctx* ctx, const bppp_circuit* circuit, size_t n, const uint8_t* states,
                                                  size_t n_states, const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl,
                                                  size_t nn, uint8_t* accept, int32_t* status, uint8_t* states_out);
/* The generic provers with the caller's transcripts -- `t: &mut Transcript` of WeightNormLinearArgument::prove (wnla.rs:125),
 * ReciprocalRangeProofProtocol::prove (reciprocal.rs:109) and ArithmeticCircuit::prove (circuit.rs:260): states in (1 or n), each
 * instance's transcript as the reference's prove leaves it out (optional; for a WNLA shape with no rounds, |l| + |n| < 6, that is
 * the input state untouched); every other argument as in the label forms above, the proofs byte-identical to the reference
 * prover's on the same transcript and draws. */
BPPP_API int bppp_wnla_prove_batch_transcript(bppp_ctx* ctx, size_t n, const uint8_t* states, size_t n_states, const uint8_t* commitments,
                                              const uint8_t* c, const uint8_t* rho, const uint8_t* mu, const uint8_t* l, size_t nl,
                                              const uint8_t* nvec, size_t nn, uint8_t* proof_r, uint8_t* proof_x, uint8_t* proof_l,
                                              uint8_t* proof_n, int32_t* status, uint8_t* states_out);
BPPP_API int bppp_reciprocal_prove_batch_transcript(bppp_ctx* ctx, size_t n, const uint8_t* states, size_t n_states, size_t dim_nd,
                                                    size_t dim_np, const uint8_t* commitments, const uint8_t* x, const uint8_t* s,
                                                    const uint8_t* digits, const uint8_t* m, const uint8_t* rnd, uint8_t* proofs,
                                                    int32_t* status, uint8_t* states_out);
BPPP_API int bppp_circuit_prove_batch_transcript(bppp_ctx* ctx, const bppp_circuit* circuit, size_t n, const uint8_t* states,
                                                 size_t n_states, const uint8_t* v_commitments, const uint8_t* v, const uint8_t* s_v,
                                                 const uint8_t* w_l, const uint8_t* w_r, const uint8_t* w_o, const uint8_t* rnd,
                                                 uint8_t* proofs, int32_t* status, uint8_t* states_out);
/* merlin::Transcript on serialized states, host only (no GPU needed): Transcript::new(label), append_message(label, msg)
 * (transcript.rs:7 uses it for points, wnla.rs:91-92 for u64s) and challenge_bytes(label, out) (transcript.rs:12).  A Rust caller
 * holding a real merlin::Transcript does not need these; a C caller builds its pre-loaded states with them. */
BPPP_API int bppp_transcript_new(const uint8_t* label, size_t label_len, uint8_t state_out[203]);
BPPP_API int bppp_transcript_append_message(uint8_t state[203], const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len);
BPPP_API int bppp_transcript_challenge_bytes(uint8_t state[203], const uint8_t* label, size_t label_len, uint8_t* out, size_t n);

/* ---- one batch over the GPUs of a node (BASELINE configs[2]; SURVEY 8b `device_mask`, 8e) ----
 * Proofs are independent (fresh transcript per proof, benches/range_proof.rs:47), so a batch shards by proof index with no
 * data-path collective: device r of a group of G verifies proofs [n r / G, n (r + 1) / G) (bppp_shard_range) against its own
 * replica of the fixed-base tables, and the single exchange is the number of rejected proofs: one 4-byte ncclAllReduce(sum,
 * int32) over RCCL/xGMI, after which every device holds the batch's global reject count (0 <=> the batch is accepted).
 * A group owns one context, one stream and (for G > 1) one RCCL communicator per device, all inside this process; a call runs
 * one host thread per device.  RCCL is loaded at group creation with dlopen("librccl.so.1") -- the library has no link-time
 * dependency on it, and a group of ONE device never needs it (set BPPP_FORCE_RCCL=1 to route a one-device group through a
 * one-rank communicator anyway, which is how the single-GPU test tier exercises this path). */
typedef struct bppp_group bppp_group;
#define BPPP_ERR_RCCL (-6) /* librccl could not be loaded, or an RCCL call failed; see bppp_last_error() */
BPPP_API void bppp_shard_range(size_t n_total, int rank, int world, size_t* lo, size_t* hi);
/* devices: n_devices distinct HIP device ordinals.  Every device gets the same generators / window width (see bppp_ctx_create). */
BPPP_API int bppp_group_create(bppp_group** out, const uint8_t g[64], const uint8_t* g_vec, const uint8_t* h_vec, const int* devices,
                               int n_devices, int fb_window_bits);
/* The same over any generator set (bppp_wnla_ctx_create on every device): the group the generic reciprocal verifier shards over. */
BPPP_API int bppp_wnla_group_create(bppp_group** out, const uint8_t g[64], const uint8_t* g_vec, size_t ng, const uint8_t* h_vec, size_t nh,
                                    const int* devices, int n_devices, int fb_window_bits);
BPPP_API void bppp_group_destroy(bppp_group* grp);
BPPP_API int bppp_group_size(const bppp_group* grp);
BPPP_API bppp_ctx* bppp_group_ctx(bppp_group* grp, int rank); /* the rank-th device's context (owned by the group) */
/* Options: every name bppp_ctx_set_option takes (applied to each rank's context), and "inject_fault_rank" = r (testing aid: rank r's
 * part of the NEXT sharded call fails with BPPP_ERR_NOMEM before it enqueues anything; -1 clears it). */
BPPP_API int bppp_group_set_option(bppp_group* grp, const char* name, long value);
/* Failure semantics of every *_sharded* call (csrc/group_core.h).  A group serves ONE sharded call at a time (calls from several
 * host threads are serialized by the group's lock).  The ranks agree on their return codes through a host-side vote BEFORE any of
 * them enqueues the all-reduce: if one rank fails (BPPP_ERR_NOMEM for its workspace, a HIP error, an invalid argument), no rank
 * enters the collective, the ranks that had already enqueued work wait for it, and the call returns the failing rank's code with
 * the group intact.  If an RCCL call itself fails, every communicator of the group is aborted (ncclCommAbort) so that no rank stays
 * blocked, the call returns BPPP_ERR_RCCL, and so does every later sharded call on that group (destroy it and create a new one). */

/* U64RangeProofProtocol::verify for ONE batch of n proofs in HOST memory, sharded over the group.  accept / status as in
 * bppp_u64_verify_batch; *reject_count (optional) receives the all-reduced number of rejected proofs. */
BPPP_API int bppp_u64_verify_batch_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                                           const uint8_t* proofs, uint8_t* accept, int32_t* status, int32_t* reject_count);
/* The same with every shard already resident on its device: rank r's arrays hold its n_r = hi - lo proofs of bppp_shard_range(n,
 * r, G) in DEVICE memory of device r; d_reject_count[r] (device int32[1], required) receives the GLOBAL reject count on every
 * device.  Returns when all devices have finished. */
BPPP_API int bppp_u64_verify_batch_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n,
                                                  const void* const* d_commitments, const void* const* d_proofs, void* const* d_accept,
                                                  void* const* d_status /* entries may be NULL */, void* const* d_reject_count);
/* The optional random-linear-combination mode (bppp_u64_verify_batch_rlc), sharded: each device checks its own shard's
 * combinations (sound per shard: the weights are a PRF of seed and the proof's index within its shard) and contributes its
 * reject count.  accept / status stay per proof. */
BPPP_API int bppp_u64_verify_batch_rlc_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                                               const uint8_t* proofs, uint8_t* accept, int32_t* status, int32_t* reject_count,
                                               const uint8_t seed[32]);
BPPP_API int bppp_u64_verify_batch_rlc_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n,
                                                      const void* const* d_commitments, const void* const* d_proofs, void* const* d_accept,
                                                      void* const* d_status, void* const* d_reject_count, const uint8_t seed[32]);
/* SEC1-compressed inputs (bppp_u64_verify_batch_sec1: 33-byte commitments, 525-byte proofs), sharded; expanded on each device. */
BPPP_API int bppp_u64_verify_batch_sec1_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments33,
                                                const uint8_t* proofs525, uint8_t* accept, int32_t* status, int32_t* reject_count);
BPPP_API int bppp_u64_verify_batch_sec1_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n,
                                                       const void* const* d_commitments33, const void* const* d_proofs525,
                                                       void* const* d_accept, void* const* d_status, void* const* d_reject_count);
/* The caller's transcripts (`t: &mut Transcript`, u64_proof.rs:42; bppp_u64_verify_batch_transcript), sharded: states = one
 * 203-byte state shared by the batch (n_states = 1, every rank receives it) or one per proof (n_states = n, split like the
 * proofs); states_out (optional) = each proof's transcript after verify.  Device form: d_states[r] is rank r's share (or the one
 * shared state, resident on device r), d_states_out entries may be NULL. */
BPPP_API int bppp_u64_verify_batch_transcript_sharded(bppp_group* grp, size_t n, const uint8_t* states, size_t n_states,
                                                      const uint8_t* commitments, const uint8_t* proofs, uint8_t* accept, int32_t* status,
                                                      uint8_t* states_out, int32_t* reject_count);
BPPP_API int bppp_u64_verify_batch_transcript_sharded_device(bppp_group* grp, size_t n, const void* const* d_states, size_t n_states,
                                                             const void* const* d_commitments, const void* const* d_proofs,
                                                             void* const* d_accept, void* const* d_status, void* const* d_reject_count,
                                                             void* const* d_states_out);
/* U64RangeProofProtocol::prove (u64_proof.rs:57-82; bppp_u64_prove_batch[_device]) for ONE batch of n values sharded over the
 * group: the same contiguous split, every proof byte-identical to the single-device call's.  Proofs are independent, so there is
 * no exchange step and no collective; the ranks only agree on their return codes (failure semantics above).  Host form: x, s, rnd
 * in, proofs, commitments and (optional) status out for the whole batch.  Device form: per-device arrays of pointers to each
 * rank's resident shard (d_status optional as a whole or per rank). */
BPPP_API int bppp_u64_prove_batch_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x,
                                          const uint8_t* s /* n x 32 */, const uint8_t* rnd /* n x 52 x 32 */,
                                          uint8_t* proofs /* n x 928 */, uint8_t* commitments /* n x 64 */, int32_t* status /* n or NULL */);
BPPP_API int bppp_u64_prove_batch_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n,
                                                 const void* const* d_x, const void* const* d_s, const void* const* d_rnd,
                                                 void* const* d_proofs, void* const* d_commitments, void* const* d_status);

/* ReciprocalRangeProofProtocol::verify (reciprocal.rs:98-107) for ONE batch of n instances sharded over a group made by
 * bppp_wnla_group_create -- BASELINE configs[4]: 2^18 instances of the (dim_nd 256, dim_np 16) shape over 8 GPUs.  Arguments as
 * bppp_reciprocal_verify_batch[_rlc][_device]; same contiguous split, same 4-byte all-reduce of the reject count.  In the device
 * form d_status[r] is required wherever rank r's shard is not empty. */
BPPP_API int bppp_reciprocal_verify_batch_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd,
                                                  size_t dim_np, const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl,
                                                  size_t nn, uint8_t* accept, int32_t* status, int32_t* reject_count);
BPPP_API int bppp_reciprocal_verify_batch_rlc_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd,
                                                      size_t dim_np, const uint8_t* commitments, const uint8_t* proofs, size_t rounds,
                                                      size_t nl, size_t nn, uint8_t* accept, int32_t* status, int32_t* reject_count,
                                                      const uint8_t seed[32]);
BPPP_API int bppp_reciprocal_verify_batch_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd,
                                                         size_t dim_np, const void* const* d_commitments, const void* const* d_proofs,
                                                         size_t rounds, size_t nl, size_t nn, void* const* d_accept, void* const* d_status,
                                                         void* const* d_reject_count);
BPPP_API int bppp_reciprocal_verify_batch_rlc_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd,
                                                             size_t dim_np, const void* const* d_commitments, const void* const* d_proofs,
                                                             size_t rounds, size_t nl, size_t nn, void* const* d_accept,
                                                             void* const* d_status, void* const* d_reject_count, const uint8_t seed[32]);

/* the same with DEVICE buffers (d_status required), asynchronous on the context's stream; the workspace lives in the context */
BPPP_API int bppp_reciprocal_verify_batch_device(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd,
                                                 size_t dim_np, const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl,
                                                 size_t nn, void* d_accept, void* d_status);
/* The optional random-linear-combination mode (as bppp_u64_verify_batch_rlc above) for the reciprocal verifier at any dimensions:
 * the final MSM over all 1 + |g_vec| + |h_vec| generators -- 769 bases for BASELINE configs[4]'s shape, about half of that
 * verifier's time -- is done once per chunk of 8 instances on secretly weighted sums; chunks that do not pass are re-checked
 * exactly, so accept / status stay per instance.  seed: 32 unpredictable bytes chosen after the proofs are fixed. */
BPPP_API int bppp_reciprocal_verify_batch_rlc(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                              const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn,
                                              uint8_t* accept, int32_t* status, const uint8_t seed[32]);
BPPP_API int bppp_reciprocal_verify_batch_rlc_device(bppp_ctx* ctx, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd,
                                                     size_t dim_np, const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl,
                                                     size_t nn, void* d_accept, void* d_status, const uint8_t seed[32]);

/* Profiling aid for bench.py: when enabled, every kernel launch of the verify pipeline is bracketed by HIP events on
 * the context's stream; bppp_ctx_get_timings returns accumulated milliseconds and launch counts per kernel since the
 * last reset.  names[i] points to a static string. */
BPPP_API int bppp_ctx_enable_timing(bppp_ctx* ctx, int enable);
BPPP_API int bppp_ctx_get_timings(bppp_ctx* ctx, int max_entries, const char** names, double* total_ms, int64_t* launches,
                         int reset);

/* Workspace the context currently holds on the GPU, in bytes (tables + per-proof workspace). */
BPPP_API size_t bppp_ctx_device_bytes(const bppp_ctx* ctx);

BPPP_API const char* bppp_strerror(int code);
BPPP_API const char* bppp_last_error(void); /* thread-local detail of the last BPPP_ERR_HIP / BPPP_ERR_NOMEM */

#ifdef __cplusplus
}
#endif
#endif /* BPPP_H */
