/*
 * plonkit_amd — C ABI of the MI355X-native PLONK prover hot path (libplonkit_amd.so).
 *
 * This is the drop-in boundary for the arithmetic that fluidex/plonkit reaches in bellman_ce
 * (reference call sites in parentheses; file:line under /root/reference).  plonkit has no FFI of
 * its own — it links bellman_ce statically — so each entry point below names the Rust call it
 * replaces; INTEGRATION.md shows the `extern "C"` block a plonkit maintainer would add.
 *
 * Conventions
 *   - plk_fr          = 4 x u64 little-endian limbs, MONTGOMERY form (R = 2^256): byte-identical
 *                       to ff_ce's in-memory `Fr`, so a Rust `&[Fr]` can be passed as is.
 *   - plk_g1_affine   = x[4] || y[4], Montgomery Fq; the point at infinity is x = y = 0
 *                       (pairing_ce's G1Affine carries a separate `infinity` flag: repack once).
 *   - plk_g1_jacobian = X || Y || Z, infinity is Z = 0.
 *   - every function returns int32_t: PLK_OK or an error code; text via plk_last_error().
 *     Nothing throws across the boundary.  Calls are blocking unless a stream is passed.
 *   - one plk_ctx drives ONE GPU (one process per GPU; multi-GPU = ranks joined by plk_comm_init: the
 *     96-byte partial sums of every commitment are all-gathered over RCCL and added, inside the library).
 *   - `*_dev` entry points take HIP device pointers (e.g. torch tensors' data_ptr()) and a
 *     hipStream_t passed as void* (NULL = the context's own stream); `host` ones take host memory.
 *     The context's stream is NOT ordered against the caller's streams: data a caller's kernel is still writing (a torch op
 *     on torch's stream, say) must be complete before a `_dev` call that passes NULL reads it — synchronise first, or pass the
 *     producing stream, which the call then runs on (the commitments instead wait for an event recorded on it, see below).
 *   - There is NO CPU fallback: without a gfx950 device plk_create fails with PLK_ERR_HIP.
 */
#ifndef PLONKIT_AMD_H
#define PLONKIT_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef struct { uint64_t l[4]; } plk_fr;
typedef struct { uint64_t x[4], y[4]; } plk_g1_affine;
typedef struct { uint64_t x[4], y[4], z[4]; } plk_g1_jacobian;
typedef struct plk_ctx plk_ctx;

enum {
    PLK_OK = 0,
    PLK_ERR_ARG = 1,        /* null pointer, bad flag                                         */
    PLK_ERR_SIZE = 2,       /* size not a power of two / log_n > 28 (2-adicity of Fr)         */
    PLK_ERR_SRS = 3,        /* no SRS uploaded or SRS shorter than the request                */
    PLK_ERR_HIP = 4,        /* HIP runtime error, or no gfx950 device                         */
    PLK_ERR_UNSAT = 5,      /* witness does not satisfy the circuit ("must satisfy")          */
    PLK_ERR_FORMAT = 6,     /* malformed r1cs / wtns / key / proof bytes                      */
    PLK_ERR_IO = 7
};

const char *plk_last_error(void);
const char *plk_version(void);
int32_t plk_device_count(void);

/* bellman_ce::worker::Worker::new() (src/plonk.rs:41,47,183): the execution resource handle. */
int32_t plk_create(int32_t device, plk_ctx **out);
void plk_destroy(plk_ctx *ctx);
int32_t plk_synchronize(plk_ctx *ctx);

/* ---- SRS: Crs<Bn256, CrsForMonomialForm> kept resident in HBM (src/plonk.rs:53; src/reader.rs:74-77) */
int32_t plk_srs_upload(plk_ctx *ctx, const plk_g1_affine *bases_host, uint64_t n);
int32_t plk_srs_set_dev(plk_ctx *ctx, const void *bases_dev, uint64_t n);       /* borrowed, not copied */
uint64_t plk_srs_size(const plk_ctx *ctx);
/* Crs::crs_42(size, &Worker) generalised (src/plonk.rs:30-48, `plonkit setup`): fills the resident SRS
 * with tau^(start+i)*G, i < n, computed on the GPU; tau = 42 reproduces the reference's local keys. */
int32_t plk_srs_generate(plk_ctx *ctx, uint64_t n, uint64_t start, uint32_t tau);
int32_t plk_srs_generate_fr(plk_ctx *ctx, uint64_t n, uint64_t start, const plk_fr *tau);     /* any tau (test keys only: tau is public) */
int32_t plk_srs_download(plk_ctx *ctx, uint64_t offset, uint64_t n, plk_g1_affine *out_host);
/* optional: builds now what the first commitment against the resident key(s) would build — the fixed-base table of the
 * MSM (15 shifted copies of the points, ~40 ms at 2^20 points).  A host program calls it on a second thread while it is
 * still parsing the circuit (the `plonkit` binary does); without it the first commitment pays for the table.           */
int32_t plk_srs_precompute(plk_ctx *ctx);
/* SEVERAL PROOFS IN FLIGHT ON ONE GPU.  SetupForProver::prove takes &self (src/plonk.rs:132-176): nothing stops a Rust host from
 * proving from two threads against one setup, and on this hardware it pays — the strict challenge chain of one proof leaves
 * latency-bound stretches (bucket-reduction tails, host round trips, small point-wise launches: ~3.5 ms of a 2^20 proof) that the
 * kernels of a second proof fill.  The unit of concurrency is the context: one plk_ctx per host thread (a context is NOT
 * thread-safe), any number of contexts per device, ONE plk_setup shared by all of them (its lazily cached extensions are
 * filled under a lock), one plk_circuit per witness.  plk_ctx_share_srs makes `dst` borrow `src`'s resident key(s) and MSM
 * fixed-base table(s) instead of building its own (0.94 GiB and ~40 ms per key at 2^20 points): `src` builds what is missing,
 * keeps ownership and refuses to replace a key that a borrower holds (PLK_ERR_ARG; a Lagrange-form key it did not have when the
 * loans were made may still be installed — the borrowers do not see it, share again for that).  Destroying `src` before its
 * borrowers is safe: the key and the tables outlive it until the last borrower returns its loan (plk_destroy / a key of its own).
 * A borrower that is given a key of its own (upload / generate / set_dev) simply stops borrowing.
 * plk_ctx_share_srs MUST NOT RUN CONCURRENTLY WITH ANY CALL ON `src` (nor on `dst`): `src` swaps its monomial / Lagrange key slots for the
 * duration of a Lagrange-form commitment, and a loan taken at that moment would lend the wrong key.  Share first, then start the threads.  */
int32_t plk_ctx_share_srs(plk_ctx *dst, plk_ctx *src);

/* ---- Polynomial::{fft,ifft,coset_fft,icoset_fft} over Fr (bellman_ce::plonk::polynomials; driven
 *      from setup() src/plonk.rs:104 and prove_by_steps src/plonk.rs:152-159).
 *      Natural order in and out.  inverse=0: evaluate on coset*<omega_n>; inverse=1: interpolate
 *      (scaled by 1/n, then by coset^-i).  coset == NULL means 1.                                  */
int32_t plk_ntt(plk_ctx *ctx, plk_fr *data_host, uint32_t log_n, int32_t inverse, const plk_fr *coset);
int32_t plk_ntt_dev(plk_ctx *ctx, void *data_dev, uint32_t log_n, int32_t inverse, const plk_fr *coset, void *stream);
/* Polynomial::coset_lde(4): n coefficients -> 4n evaluations on 7*<omega_4n> */
int32_t plk_lde4(plk_ctx *ctx, const plk_fr *coeffs_host, uint32_t log_n, plk_fr *out_4n_host);
int32_t plk_lde4_dev(plk_ctx *ctx, const void *coeffs_dev, uint32_t log_n, void *out_4n_dev, void *stream);
/* The same 4n evaluations of `count` (<= 16) polynomials in the COSET-MAJOR order the prover's round 3 works in:
 * out[p][k * n + r] = f_p(7 * omega_4n^(4 r + k)), k = 0..3 — four n-point coset transforms per polynomial, all of them in
 * one launch per pass (prove_by_steps' ~18 coset_lde(4) calls, src/plonk.rs:152-159).  A permutation of plk_lde4_dev's output;
 * exposed so that the layout is testable on its own.                                                                      */
int32_t plk_lde4_coset_major_dev(plk_ctx *ctx, const void *const *coeffs_dev, uint32_t count, uint32_t log_n, void *const *out_4n_dev, void *stream);
/* The way back (round 3 of prove_by_steps: Polynomial::icoset_fft at 4n, src/plonk.rs:152-159): 4n values in that coset-major order
 * -> the 4n coefficients in natural order, in place (four inverse n-point coset transforms in one launch per pass + a 4-point combine).
 * Test hook like the function above.                                                                                              */
int32_t plk_icoset4_coset_major_dev(plk_ctx *ctx, void *data_4n_dev, uint32_t log_n, void *stream);

/* ---- kate_commitment::commit_using_monomials -> multiexp::dense_multiexp (src/plonk.rs:122-124 and
 *      the 11 commitments of prove): sum_i scalars[i] * srs[base_offset + i], scalars Montgomery Fr.
 *      One pass of the kernels takes up to 2^24 terms; plk_msm_g1, _dev and _partial_dev (and plk_prove) cut longer
 *      vectors into successive pieces themselves, the _batch_dev and _enqueue_dev entry points return PLK_ERR_SIZE. */
int32_t plk_msm_g1(plk_ctx *ctx, const plk_fr *scalars_host, uint64_t n, uint64_t base_offset, plk_g1_affine *out);
int32_t plk_msm_g1_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, plk_g1_affine *out, void *stream);
/* `count` commitments of equal length against the same bases in one pass of the kernels (the 4 wire /
 * 4 quotient / 2 opening commitments of a proof, the 11 of a verification key)                       */
int32_t plk_msm_g1_batch_dev(plk_ctx *ctx, const void *const *scalars_dev, uint32_t count, uint64_t n, uint64_t base_offset, plk_g1_affine *out, void *stream);
/* the same sum left in Jacobian form, for cross-rank combination (multi-GPU shards) */
int32_t plk_msm_g1_partial_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, plk_g1_jacobian *out, void *stream);
/* enqueue only (no host sync); finish with _finish.  The pair is a FIFO of depth THREE: the context owns three sets of
 * MSM scratch, result buffers and streams, so two more commitments can be enqueued before the first is finished — the
 * accumulation of commitment k then shares the GPU with the latency-bound bucket reduction of k-1 and the digit /
 * partition kernels of k+1 (2^20 terms: 1.39 ms per commitment back to back, 1.46-1.50 with two in flight, 1.66 one
 * at a time).  A slot is taken lowest index first: a caller that keeps two in flight never allocates the third set.
 * `stream` is where the scalars were produced: the kernels run on the slot's own stream after an event recorded there,
 * and the scalars must stay untouched until the matching _finish.  A fourth enqueue, or a finish with nothing in
 * flight, returns PLK_ERR_ARG.                                                                                      */
int32_t plk_msm_g1_enqueue_dev(plk_ctx *ctx, const void *scalars_dev, uint64_t n, uint64_t base_offset, void *stream);
int32_t plk_msm_g1_finish(plk_ctx *ctx, plk_g1_jacobian *out);
/* the same FIFO with a BATCH of `count` (<= 8) commitments of equal length against the same bases per slot — the shape of the
 * prover's rounds (4 wire, 4 quotient, 2 opening commitments): one pass of the kernels serves the whole batch.
 * _finish_batch returns the count Jacobian sums in order; _finish_batch_sharded additionally runs the context's combiner
 * (plk_comm_init / plk_set_commit_shard: one exchange for the batch) and returns affine points.                           */
int32_t plk_msm_g1_enqueue_batch_dev(plk_ctx *ctx, const void *const *scalars_dev, uint32_t count, uint64_t n, uint64_t base_offset, void *stream);
int32_t plk_msm_g1_finish_batch(plk_ctx *ctx, plk_g1_jacobian *out, uint32_t count);
int32_t plk_msm_g1_finish_batch_sharded(plk_ctx *ctx, plk_g1_affine *out, uint32_t count);
/* tracing hook: HIP events around the bucket-accumulation kernel of the last MSM (bench roofline) */
int32_t plk_set_kernel_timing(plk_ctx *ctx, int32_t on);
int32_t plk_msm_last_kernel_ms(plk_ctx *ctx, float *accumulate_ms);

/* ---- Crs::<Lagrange>::from_powers (src/plonk.rs:179-185): inverse NTT over G1 (dump-lagrange)  */
int32_t plk_g1_intt(plk_ctx *ctx, const plk_g1_affine *in_host, uint32_t log_n, plk_g1_affine *out_host);
/* same, from the first 2^log_n points of the resident SRS into a device buffer (64 B per point) */
int32_t plk_g1_intt_srs_dev(plk_ctx *ctx, uint32_t log_n, void *out_dev, void *stream);

/* ---- multi-GPU commitments inside plk_prove / plk_setup_write_vk (the MSM shards, everything else
 *      is replicated).  One process per GPU; rank r keeps only the SRS points [first_index, first_index + plk_srs_size)
 *      resident (plk_srs_upload of its slice, or plk_srs_generate(ctx, n, first_index, tau)), computes every
 *      commitment over that index range and hands the `count` (<= 8) Jacobian partial sums to `combine`, which must
 *      replace them in place by the sums over all ranks — an all_gather of 96 bytes per commitment over RCCL and a
 *      host EC sum (plonkit_amd/sharded.py: ShardedProver); EC addition is not an RCCL reduction op.  All ranks then
 *      hold the same commitments, derive the same challenges and produce the same proof bytes as a single GPU.
 *      combine == NULL switches back to single-GPU commitments.  A Lagrange-form key, if used, is sliced the same way. */
typedef int32_t (*plk_combine_fn)(void *user, plk_g1_jacobian *sums, uint32_t count);
int32_t plk_set_commit_shard(plk_ctx *ctx, uint64_t first_index, plk_combine_fn combine, void *user);

/* ---- the same, with the exchange built in (comm.cpp): the counterpart of `Worker::new()` (src/plonk.rs:41,47,183) for a node
 *      of GPUs is ONE PROCESS PER GPU, each with its own plk_ctx, joined by an RCCL communicator.  A caller in any language
 *      (the Rust host of INTEGRATION.md, the `plonkit` binary of this package) needs no collective library of its own:
 *        rank 0:  plk_comm_unique_id(&id), hand the 128 bytes to the other ranks (file, pipe, env — the caller's choice)
 *        all:     plk_create(device_of_rank) ; plk_comm_init(ctx, rank, world, &id, first_index)
 *                 plk_srs_upload(ctx, key + first_index, n_local)      (or plk_srs_generate(ctx, n_local, first_index, tau))
 *                 plk_setup_prepare / plk_setup_write_vk / plk_prove as on one GPU: identical bytes on every rank
 *      plk_comm_init creates the communicator on the context's device (ncclCommInitRank — collective: every rank must
 *      call it) and installs the built-in combiner: one ncclAllGather of count x 96 bytes on a stream of its own per
 *      batch of commitments, then world-1 host EC additions each (EC addition is not an RCCL reduction op).
 *      RCCL is bound at run time (librccl.so.1); without it these calls return PLK_ERR_HIP and everything else works.
 *      plk_comm_init_tcp is the same combiner over a TCP hub on 127.0.0.1:port (rank 0 listens) for the one case RCCL
 *      refuses — several ranks sharing ONE device, as on a single-GPU test box.                                        */
typedef struct { char bytes[128]; } plk_comm_id;                       /* an ncclUniqueId */
int32_t plk_comm_unique_id(plk_comm_id *out);
int32_t plk_comm_init(plk_ctx *ctx, int32_t rank, int32_t world, const plk_comm_id *id, uint64_t first_index);
int32_t plk_comm_init_tcp(plk_ctx *ctx, int32_t rank, int32_t world, uint16_t port, uint64_t first_index);
int32_t plk_comm_set_shard(plk_ctx *ctx, uint64_t first_index);         /* same communicator, another slice of the key */
/* Two ways to use the ranks of a communicator (bellman's Worker, src/plonk.rs:41,47,183, has one: threads of one address space):
 *   PLK_SHARD_REPLICATE (default)  every rank runs the same plk_prove on the same circuit; only the commitments are split.  The
 *                                  transforms, the quotient and the openings are done G times over (Amdahl: <= 2.5x at 2^20).
 *   PLK_SHARD_SCATTER              OWNER COMPUTES: rank 0 alone runs plk_prove / plk_setup_write_vk.  For every batch of commitments
 *                                  it sends each other rank that rank's slice of the scalar vectors (N/G x 32 B per vector: 4 MiB per
 *                                  xGMI link at 2^20, 64 MiB at 2^24; grouped ncclSend / ncclRecv) and gets 96 bytes back; the other
 *                                  ranks only hold their slice of the key and sit in plk_comm_serve.  Same proof bytes.  Rank r must
 *                                  hold the key points [r * L, (r + 1) * L), L = the owner's key size (plk_comm_init's first_index).
 * Every rank of a communicator must be in the same mode: PLK_SHARD_MODE=scatter in the environment of all of them, or
 * plk_comm_set_mode on all of them right after plk_comm_init.
 * PLK_SHARD_SCATTER IS EXPERIMENTAL BETWEEN GPUS: its RCCL transport (header broadcast + grouped ncclSend / ncclRecv) has run on one GPU only
 * (plk_comm_selftest, plk_comm_scatter_selftest) and over the TCP tier of the tests — no multi-GPU node has run it.  On an RCCL communicator of
 * more than one rank plk_comm_set_mode(SCATTER) therefore runs plk_comm_selftest first (a collective, like the call itself) and refuses the mode
 * if that fails.  If the owner's plk_prove / plk_setup_write_vk returns an error after a batch went out (PLK_ERR_UNSAT, ...), the batch's exchange
 * is still run, so that the workers stay in step and the communicator stays usable.                                                       */
#define PLK_SHARD_REPLICATE 0
#define PLK_SHARD_SCATTER 1
int32_t plk_comm_set_mode(plk_ctx *ctx, int32_t mode);
/* worker ranks (rank > 0) of a scatter-mode communicator: commit whatever the owner sends against this context's key slice until the
 * owner calls plk_comm_stop_workers (returns PLK_OK) or goes away (PLK_ERR_HIP / PLK_ERR_IO).  *batches = batches served.            */
int32_t plk_comm_serve(plk_ctx *ctx, uint64_t *batches);
int32_t plk_comm_stop_workers(plk_ctx *ctx);                            /* owner: ends every worker's plk_comm_serve */
/* every rank: one header broadcast + one grouped ring step (send to rank + 1, receive from rank - 1) over the RCCL communicator, checked
 * byte by byte — the transport of owner-computes mode, exercised before the mode is trusted on a new node (works with one rank too).       */
int32_t plk_comm_selftest(plk_ctx *ctx);
/* one GPU, RCCL communicator of ONE rank, key of >= 2^log_n points: the scatter step of owner-computes mode through the real RCCL branch
 * (comm_send_work / comm_recv_work, rank 0 receiving its own share), `iterations` times, with the scalars STILL BEING WRITTEN on the context's
 * stream when the step is called — the ordering the TCP tier cannot test.  The commitment of what arrived must equal the commitment of the
 * vector itself; *mismatches = iterations where it does not (0 is the pass).  No reference counterpart (bellman's Worker shares memory).    */
int32_t plk_comm_scatter_selftest(plk_ctx *ctx, uint32_t log_n, uint32_t iterations, uint32_t *mismatches);
/* ranks RCCL itself counts in the context's communicator (ncclCommCount); 0 without an RCCL communicator — the multi-GPU bench line prints it */
int32_t plk_comm_nccl_count(const plk_ctx *ctx, int32_t *count);
int32_t plk_comm_destroy(plk_ctx *ctx);                                 /* back to single-GPU commitments */
/* plk_msm_g1_finish + the combiner: the commitment over all ranks' shards (each rank enqueued its own slice), affine */
int32_t plk_msm_g1_finish_sharded(plk_ctx *ctx, plk_g1_affine *out);
/* the combiner on its own (no context, no GPU): the TCP transport opened directly, and the plk_combine_fn it serves —
 * plk_set_commit_shard(ctx, first, plk_comm_combine, comm) is what plk_comm_init_tcp does.  Used by the CPU tests. */
int32_t plk_comm_open_tcp(int32_t rank, int32_t world, uint16_t port, void **comm_out);
int32_t plk_comm_combine(void *comm, plk_g1_jacobian *sums, uint32_t count);
void plk_comm_close(void *comm);
/* test tier: the scatter step of owner-computes mode on HOST buffers over a plk_comm_open_tcp communicator (same header, shares and sequence
 * numbers as the device path).  Rank 0 sends vecs[0..count) (n elements of 32 B; count = 0: the stop message); rank r > 0 blocks for the next
 * message and receives count_out x len_out x 32 bytes, len_out = its share [r * slice, min((r + 1) * slice, n)).                        */
int32_t plk_comm_scatter_host(void *comm, const void *const *vecs, uint32_t count, uint64_t n, uint64_t slice,
                              void *mine, uint64_t mine_cap, uint32_t *count_out, uint64_t *len_out);
int32_t plk_comm_info(const plk_ctx *ctx, int32_t *rank, int32_t *world, uint64_t *exchanges);

/* ---- Lagrange-form key: Crs<E, CrsForLagrangeForm> (L_i(tau)*G, i < N), the optional `-l` key of `plonkit prove`
 *      (src/bin/main.rs:384-391; src/plonk.rs:138-146: with it, prove() commits the witness and grand-product
 *      polynomials from their evaluations — commit_using_values — instead of their coefficients).  A second resident
 *      SRS with its own fixed-base table; while one is set, plk_prove uses it for those 5 commitments and requires
 *      its size to equal the circuit's domain.  The proof bytes are the same either way.
 *      set_dev: points already on the device, e.g. the output of plk_g1_intt_srs_dev (not copied, must stay alive). */
int32_t plk_srs_lagrange_upload(plk_ctx *ctx, const plk_g1_affine *bases, uint64_t n);
int32_t plk_srs_lagrange_set_dev(plk_ctx *ctx, const void *bases_dev, uint64_t n);
int32_t plk_srs_lagrange_clear(plk_ctx *ctx);
uint64_t plk_srs_lagrange_size(const plk_ctx *ctx);

/* ---- host-side G1 helpers (pure CPU, usable without a GPU) ---------------------------------- */
int32_t plk_g1_sum_jacobian(const plk_g1_jacobian *parts, uint64_t n, plk_g1_affine *out);
int32_t plk_g1_on_curve(const plk_g1_affine *p);                       /* 1 / 0                   */
/* big-endian canonical 64-byte encoding of Proof/VerificationKey/Crs::write     */
int32_t plk_g1_to_bytes(const plk_g1_affine *p, uint8_t out[64]);
int32_t plk_g1_from_bytes(const uint8_t in[64], plk_g1_affine *out);
int32_t plk_fr_to_bytes(const plk_fr *a, uint8_t out[32]);
int32_t plk_fr_from_bytes(const uint8_t in[32], plk_fr *out);

/* ---- SRS key files: Crs::read / Crs::write (src/reader.rs:67-89; src/bin/main.rs:341,379), monomial and
 *      Lagrange containers alike.  parse: points == NULL only reports n and the G2 bytes;
 *      every point is range- and curve-checked like Crs::read.  serialize: out == NULL only reports len.  */
int32_t plk_key_parse(const uint8_t *data, uint64_t len, plk_g1_affine *points, uint64_t cap, uint64_t *n_out, uint8_t g2_out[256]);
int32_t plk_key_serialize(const plk_g1_affine *points, uint64_t n, const uint8_t g2[256], uint8_t *out, uint64_t cap, uint64_t *len);
void plk_crs42_g2_bytes(uint8_t out[256]);        /* G2 section of Crs::crs_42: {G2, 42*G2} */

/* ---- RollingKeccakTranscript (src/plonk.rs:10,140,152; spec contrib/template.sol:267-307) ---- */
typedef struct { uint8_t state0[32], state1[32]; uint32_t counter; } plk_transcript;
void plk_transcript_init(plk_transcript *t);
void plk_transcript_absorb_fr(plk_transcript *t, const plk_fr *v);
void plk_transcript_absorb_g1(plk_transcript *t, const plk_g1_affine *p);
void plk_transcript_challenge(plk_transcript *t, plk_fr *out);
void plk_keccak256(const uint8_t *in, uint64_t len, uint8_t out[32]);

/* ---- verifier (pure CPU, as in the reference): verifier::verify::<E,P,RollingKeccakTranscript>(&proof,&vk,None)
 *      behind plonk::verify (src/plonk.rs:189-210; CLI src/bin/main.rs:425-437).  Takes the bytes of vk.bin and
 *      proof.bin; *valid = 1/0.  Malformed files (short, point off the curve, scalar >= r)
 *      return PLK_ERR_ARG — the reference panics in its readers at that point.  The final check
 *      e(A, g2[0]) * e(B, g2[1]) == 1 is a real BN254 optimal-ate pairing (pairing.cpp), no trapdoor.          */
int32_t plk_verify(const uint8_t *vk, uint64_t vk_len, const uint8_t *proof, uint64_t proof_len, int32_t *valid);
/* plonk::verify with options.  PLK_VERIFY_STRICT_INPUTS: refuse keys with num_inputs = 0, as the Solidity verifier the reference
 * generates does (contrib/template.sol:697 `require(vk.num_inputs >= 1)`).  bellman's Rust verifier behind `plonkit verify`
 * (src/plonk.rs:189-210) has no such requirement as far as this package can tell (UNPINNED: the reference holds no zero-input
 * fixture), so plk_verify accepts them by default; it applies the strict rule when the environment variable
 * PLK_VERIFY_STRICT_INPUTS is set (to anything but 0).  Unknown flag bits: PLK_ERR_ARG.                                          */
#define PLK_VERIFY_STRICT_INPUTS 1u
int32_t plk_verify_ex(const uint8_t *vk, uint64_t vk_len, const uint8_t *proof, uint64_t proof_len, uint32_t flags, int32_t *valid);
/* e(a, g2_a) * e(b, g2_b) == 1 ?   G2 as 128 bytes x.c1|x.c0|y.c1|y.c0 big-endian (the key/vk file encoding) */
int32_t plk_pairing_check(const plk_g1_affine *a, const uint8_t *g2_a, const plk_g1_affine *b, const uint8_t *g2_b, int32_t *is_one);

/* ---- circuit pipeline: circom loaders + transpile + setup + prove ----------------------------
 * plk_circuit mirrors CircomCircuit{r1cs, witness, wire_mapping: None, aux_offset: 1}
 * (src/circom_circuit.rs:41-47).  Loaders follow src/reader.rs:178-241, src/r1cs_file.rs:100-154
 * (r1cs), src/reader.rs:92-175 (witness).  `is_json` selects the parser as the reference does by
 * file suffix (src/reader.rs:93,179).                                                          */
typedef struct plk_circuit plk_circuit;
int32_t plk_circuit_load(const uint8_t *r1cs, uint64_t r1cs_len, int32_t r1cs_is_json,
                         const uint8_t *witness, uint64_t witness_len, int32_t witness_is_json,
                         plk_circuit **out);                            /* witness may be NULL    */
/* synthetic chain circuit of exactly `target_gates` PLONK gates + 1 public input, with witness
 * (xoshiro256** seed, pinned constraint shapes); bench / test input.    */
int32_t plk_circuit_synthetic(uint64_t target_gates, uint64_t seed, plk_circuit **out);
/* the same generator with two more knobs (bench / test input).  witness_seed != 0: the SAME R1CS (it depends on `seed` only) with
 * another satisfying witness — what a prover serving many requests for one circuit sees (CircomCircuit{r1cs, witness},
 * src/circom_circuit.rs:41-47: one r1cs, a witness per proof).  lc_terms = 5..64: "dense" body — every constraint's A side is a
 * linear combination of lc_terms earlier wires plus a constant, the shape of a circom Poseidon round, which the transpiler folds
 * through the d column with q_d_next = -1 (src/circom_circuit.rs:114-131): d, q_d_next and the fourth quotient chunk are live, so a
 * proof does all 11 commitments (the pinned-subset circuit leaves two of them empty).  That chaining rule is unpinned (SURVEY.md
 * A.3).  lc_terms = 0: exactly plk_circuit_synthetic.                                                                            */
int32_t plk_circuit_synthetic_ex(uint64_t target_gates, uint64_t seed, uint64_t witness_seed, uint32_t lc_terms, plk_circuit **out);
/* what = 0: iden3 .r1cs v1 bytes, 1: .wtns v2 bytes (the reference's own input formats,
 * src/r1cs_file.rs:100-154, src/reader.rs:124-175); out == NULL only reports the length.            */
int32_t plk_circuit_export(const plk_circuit *c, int32_t what, uint8_t *out, uint64_t cap, uint64_t *len);
void plk_circuit_free(plk_circuit *c);
/* plonk::analyse (src/plonk.rs:72-93): JSON as serde_json::to_string prints it (src/tests.rs:14) */
int32_t plk_circuit_analyse(const plk_circuit *c, char *out_json, uint64_t cap);

/* SetupForProver (src/plonk.rs:50-55): setup polynomials resident on the GPU */
typedef struct plk_setup plk_setup;
/* SetupForProver::prepare_setup_for_prover (src/plonk.rs:97-119): transpile + setup() = 11 iNTT(N) */
int32_t plk_setup_prepare(plk_ctx *ctx, const plk_circuit *c, plk_setup **out);
/* the same in two calls: the host phase (transpile + columns, pure CPU, no context) may run while another thread is still
 * bringing the GPU up (plk_create, key upload, plk_srs_precompute); plk_setup_upload then moves it to the device
 * (11 uploads, 11 iNTT, the permutation).  plk_setup_prepare == plk_setup_prepare_host + plk_setup_upload.              */
int32_t plk_setup_prepare_host(const plk_circuit *c, plk_setup **out);
int32_t plk_setup_upload(plk_ctx *ctx, plk_setup *s);
void plk_setup_free(plk_setup *s);
uint64_t plk_setup_domain_size(const plk_setup *s);                    /* N = n + 1              */
/* the same N straight from the circuit: transpile only (pure CPU), no columns, no device work — all that `dump-lagrange` needs of
 * prepare_setup_for_prover (src/bin/main.rs:360-381 builds the whole setup to read setup.n) */
int32_t plk_circuit_domain_size(const plk_circuit *c, uint64_t *n_out);
/* make_verification_key + VerificationKey::write (src/plonk.rs:122-124; src/bin/main.rs:501-502).
 * g2_bytes = the 2 x 128 bytes of the key file's G2 section, copied through.                    */
int32_t plk_setup_write_vk(plk_ctx *ctx, const plk_setup *s, const uint8_t g2_bytes[256], uint8_t *out, uint64_t cap, uint64_t *len);
/* SetupForProver::prove(circuit, "keccak") with the monomial key (src/plonk.rs:132-159) followed by
 * Proof::write (src/bin/main.rs:407-408).  Fails with PLK_ERR_UNSAT like the reference's
 * `expect("must satisfy")` (src/plonk.rs:137).                                                  */
int32_t plk_prove(plk_ctx *ctx, const plk_setup *s, const plk_circuit *c, uint8_t *proof_out, uint64_t cap, uint64_t *len);
/* per-phase wall-clock of the last plk_prove on this ctx, milliseconds (tracing hook) */
int32_t plk_prove_timings(const plk_ctx *ctx, double *out_ms, uint32_t cap, uint32_t *count);
/* the vectors between the rounds of the last plk_prove on this ctx (tracing hook: a proof that differs from the reference's
 * is localised to a round).  which: 0..3 wire polynomials a, b, c, d (N coefficients), 4 grand product z (N), 5 quotient t
 * (4N coefficients), 6 linearisation r (N), 7 / 8 the opening quotients behind W_z / W_z_omega (N).  out_host == NULL only
 * reports the length.                                                                                                    */
int32_t plk_prove_trace(plk_ctx *ctx, uint32_t which, plk_fr *out_host, uint64_t cap, uint64_t *n);

/* ---- the polynomial helpers of rounds 2, 4 and 5 on their own (bellman_ce::plonk::polynomials / better_cs::prover, reached
 *      from prove_by_steps src/plonk.rs:152-159): Polynomial::evaluate_at, the division by (x - z) behind the two opening
 *      proofs ((p(x) - p(z)) / (x - z), n coefficients, the top one zero), and the permutation grand product
 *      z_0 = 1, z_{i+1} = z_i * prod_j (w_j + beta k_j omega^i + gamma) / (w_j + beta sigma_j + gamma)  (SURVEY.md A.4 round 2;
 *      k = 1, 5, 7, 10; no per-element inversion: prefix x suffix products and one host inversion).  Device vectors, N = 2^log_n. */
int32_t plk_poly_evaluate_at_dev(plk_ctx *ctx, const void *coeffs_dev, uint64_t n, const plk_fr *z, plk_fr *out, void *stream);
int32_t plk_poly_divide_by_linear_dev(plk_ctx *ctx, const void *coeffs_dev, uint64_t n, const plk_fr *z, void *quotient_dev, void *stream);
int32_t plk_permutation_grand_product_dev(plk_ctx *ctx, const void *const wires_dev[4], const void *const sigmas_dev[4], const plk_fr *beta, const plk_fr *gamma,
                                          uint32_t log_n, void *z_values_dev, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
