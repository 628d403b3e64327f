/* za_amd -- C ABI of the MI355X-native Groth16/BN254 prover hot path for adria0/za.
 *
 * This is the drop-in boundary a Rust (or cgo / ctypes) caller binds instead of the bellman_ce
 * calls za makes today.  Every entry point cites the reference interface it replaces; paths are
 * relative to the za repository.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - Fr / Fq element, "Montgomery LE": x*2^256 mod p as 4 x u64 little-endian limbs (32 bytes) --
 *     the in-memory form of pairing_ce::bn256::{Fr,Fq}.
 *   - scalar, "canonical LE": the integer itself, 32 bytes little-endian (FrRepr, i.e. what
 *     `into_repr()` yields and what bellman's multiexp consumes).
 *   - G1 affine base: x||y Montgomery LE, 64 bytes; G2 affine base: x.c0||x.c1||y.c0||y.c1, 128 bytes;
 *     the identity is all-zero bytes.
 *   - Jacobian result: X||Y||Z Montgomery LE (G1 96 bytes, G2 192 bytes); Z = 0 is the identity,
 *     otherwise the library returns Z = 1 (normalised).
 *   - proof: 256 bytes, canonical BIG-endian field elements A.x A.y B.x.c0 B.x.c1 B.y.c0 B.y.c1 C.x C.y
 *     (the numbers proof.json prints, prover/src/groth16/format.rs:46-52,69-78).
 *   - All functions return 0 on success or a ZG_E_* code; no exception crosses the boundary.
 *   - One zg_ctx per process and GPU; calls on one ctx must be serialised by the caller (the
 *     reference is synchronous too).  Handles are freed by their *_free function.
 *   - `d_` pointers are device (HBM) addresses on the ctx's GPU; others are host addresses.
 */
#ifndef ZA_AMD_H
#define ZA_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes.  1..8 mirror bellman::SynthesisError as wrapped by za's Error::Synthesis
 * (prover/src/groth16/error.rs:4,20-24). */
#define ZG_OK 0
#define ZG_E_ASSIGNMENT_MISSING 1
#define ZG_E_DIVISION_BY_ZERO 2
#define ZG_E_UNSATISFIABLE 3
#define ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE 4
#define ZG_E_UNEXPECTED_IDENTITY 5
#define ZG_E_IO 6
#define ZG_E_MALFORMED_VERIFYING_KEY 7
#define ZG_E_UNCONSTRAINED_VARIABLE 8
#define ZG_E_BAD_FORMAT 20      /* za Error::BadFormat (error.rs) */
#define ZG_E_QUERY_SIZE 21      /* multiexp's `query_size == density total` assertion */
#define ZG_E_INVALID_ARG 22
#define ZG_E_BUFFER_TOO_SMALL 23
#define ZG_E_HIP 30
#define ZG_E_NO_DEVICE 31

typedef struct zg_ctx zg_ctx;
typedef struct zg_params zg_params;
typedef struct zg_r1cs zg_r1cs;

/* ---- context ------------------------------------------------------------------------------
 * Replaces bellman's `Worker::new()` thread pool (implicit in create_random_proof,
 * prover/src/groth16/prover.rs:173).  `device` is the HIP device ordinal.  Fails with
 * ZG_E_NO_DEVICE when no GPU is visible: there is no CPU fallback.
 * The context's eight side streams (the multiexps' queues) take 70-80 ms to create: zg_init starts that on a helper thread
 * of its own and returns; the first multiexp batch (or zg_free) waits for it. */
int zg_init(int device, zg_ctx** out);
void zg_free(zg_ctx* ctx);
const char* zg_strerror(int code);
/* Last detailed message recorded on this ctx (never NULL). */
const char* zg_last_error(const zg_ctx* ctx);
/* Use an existing HIP stream (hipStream_t cast to void*) for every launch; NULL = the ctx's own. */
int zg_set_stream(zg_ctx* ctx, void* hip_stream);
/* Options of a context.  ZG_OPT_ONE_SHOT (value != 0): this context proves once per key (za's `prove` command: read the
 * key, prove, exit), so the per-key precomputation that only pays over several proofs is skipped -- the h query's
 * evaluation-form bases (2 s at 2^22) and the window / subset-sum tables (0.3 s) -- and the proof takes bellman's own
 * schedule of seven transforms.  Same proof bytes either way. */
#define ZG_OPT_ONE_SHOT 1
/* What a deployed host may want to change about the precomputation and the witness hand-over (none of it changes a
 * proof's bytes):
 *   ZG_OPT_PACK           zg_prove / zg_witness_to_device hand a witness that is mostly zeros and ones over in packed
 *                         form (host threads build it, 3 % of the bytes cross PCIe): 1 = whenever the witness is of
 *                         that kind (default), 0 = never (the 32-byte array is copied as it is), 2 = as 1, but while the
 *                         pass measures slower than the plain copy on this host the next 32 witnesses go over unpacked.
 *   ZG_OPT_PACK_THREADS   host threads of that pass, the caller's included (default min(16, half the hardware threads);
 *                         0 = the default).
 *   ZG_OPT_PACK_MIN_LOG_N witnesses of fewer than 2^value values are never packed (default 18).
 *   ZG_OPT_PACK_PIN       1 (default): the pass's worker threads are bound to the CPUs of the GPU's NUMA node, where
 *                         zg_host_alloc's pinned memory lives; 0 = left to the scheduler.
 *   ZG_OPT_H_EVAL         1 (default): keys get their h query in evaluation form (one transform fewer per proof, paid
 *                         once per key); 0 = bellman's coefficient form.  Read when a key is loaded / first used.
 *   ZG_OPT_TABLES         bit mask over (h, l, a, b_g1, b_g2): the queries that may get a window / subset-sum table
 *                         (default 31; 0 = none).  Read when a key's tables are first built.
 *   ZG_OPT_TABLES_MAX_MB  cap on the memory of one table in MB (default 0 = 60 % of the free HBM). */
#define ZG_OPT_PACK 2
#define ZG_OPT_PACK_THREADS 3
#define ZG_OPT_PACK_MIN_LOG_N 4
#define ZG_OPT_PACK_PIN 5
#define ZG_OPT_H_EVAL 6
#define ZG_OPT_TABLES 7
#define ZG_OPT_TABLES_MAX_MB 8
#define ZG_OPT_PACK_BUDGET_PERMILLE 9   /* ZG_OPT_PACK = 2: the share of the plain copy's time the pass may take (default 900) */
#define ZG_OPT_COMBINE_EARLY 11        /* 1 (default): the witness part of the combination runs beside the h multiexp; 0: all of it at the end */
#define ZG_OPT_PACK_RAW_PERMILLE 10     /* largest share of a pinned witness the copy engine may take as it is while the threads pack the rest, in 1/1000 (default 400; 0 = none) */
/* Schedule switches: what the tests, the fuzzer and the A/B tools force (a deployment leaves the defaults; none of it
 * changes a proof's bytes).  The library reads no environment variable for any of this -- ZA_AMD_TRACE (host time stamps
 * to stderr) is the only one it reads at all.
 *   ZG_OPT_SMALL_LOG_N     multiexps of up to 2^value points run in small-job mode (queued whole, always a window table,
 *                          digits planned for depth); default 16, 0 = mode off.  Read per proof and when tables are built.
 *   ZG_OPT_SMALL_W_LOG_N   the same bound for a witness query whose "real" scalars number at most 2^SMALL_LOG_N;
 *                          default -1 = 600 000 positions (a 1/8 shard of a 2^22 query), 0 = off.
 *   ZG_OPT_SORT_STAGED     digit sort: -1 (default) chosen per job, 0 = the two-level scatter, 2 = the LDS-staged
 *                          three-level partition for every large job, 3 = for every job with at least 2^7 buckets.
 *   ZG_OPT_PART_STAGE_MIN  last-level sort segments of at least this many entries go through the LDS stage
 *                          (default -1 = 16384; 0 = all of them).
 *   ZG_OPT_SUBSET_W        positions per subset-sum group, 4 or 8 (default 8).  Read when a key's tables are built.
 *   ZG_OPT_COMPACT_TABLE   1 (default): compact window table behind the subset sums.  Read when tables are built.
 *   ZG_OPT_H_PARTS         parts the h multiexp is split into (0 = default: 2 for a lone context from 2^22 points, else 1).
 *   ZG_OPT_H_PIPE          the last transform kernel is launched range by range and every h part is gated on its range alone:
 *                          -1 (default) where the h multiexp runs in two parts anyway (a lone context, from 2^22 rows: one
 *                          per cent of a 2^22 proof, measured four times out of four), 1 from 2^20 rows, 0 never. */
#define ZG_OPT_SMALL_LOG_N 12
#define ZG_OPT_SMALL_W_LOG_N 13
#define ZG_OPT_SORT_STAGED 14
#define ZG_OPT_PART_STAGE_MIN 15
#define ZG_OPT_SUBSET_W 16
#define ZG_OPT_COMPACT_TABLE 17
#define ZG_OPT_H_PARTS 18
#define ZG_OPT_H_PIPE 19
/*   ZG_OPT_SHARE_SORT      the witness queries multiply the same scalars: 1 (default) the b_g1 multiexp runs over the b_g2
 *                          multiexp's sorted (position, digit) pairs, and the queries of full density whose window tables
 *                          have one plan (the library gives them one where it can) share ONE digit sort over the witness
 *                          positions; 2 = that one sort for ALL witness queries with such tables, each reading it through
 *                          its position map (measured: slower for ragged densities; the tests use it); 0 = a sort per
 *                          query.  Read when a key's tables are built (the common plan) and per proof. */
#define ZG_OPT_SHARE_SORT 20
/*   ZG_OPT_NTT_BATCH_LOG_M domains of up to 2^value rows: every pass of the H pipeline takes the three vectors a, b, c in ONE
 *                          launch (default 21: a bit-heavy 2^18 / 2^19 proof -17 %, 2^20 -6 %, 2^21 -4 %, no difference at 2^22; a dense
 *                          witness +-1 % at 2^18 - 2^20; 0 = always one vector per launch). */
#define ZG_OPT_NTT_BATCH_LOG_M 21
/*   ZG_OPT_C_FOLD          the quotient (a b - c) / Z is linear in c, and c's values on the domain are the C sides of the R1CS
 *                          rows as evaluated -- for a circuit of bit signals, bits and small integers.  1 (default): for a key
 *                          with the evaluation form whose first witness gives rows of few non-zero 16-bit digits (at most 2
 *                          per row on average, rows equal to 2 not counted; decided once per key, with its tables), zg_prove*
 *                          transform a and b only -- FOUR transforms instead of six -- and the C term enters as 2 m more
 *                          points of the l multiexp (a second transformed basis and its doubles, built once per key), for
 *                          domains of 2^ZG_OPT_C_FOLD_MIN_LOG_M (default 20) to 2^26 rows -- and EVERY proof of such a key
 *                          is checked on the device: the evaluator counts the digits of this witness's rows and a proof
 *                          whose rows hold field elements runs six transforms, its rows struck from the l multiexp (no host
 *                          round trip: 2^22, a dense witness through a key folded by a bit witness 34.5 -> 29.8 ms, six
 *                          transforms 29.6; the guard costs a bit witness about 0.1 ms); 2 = for every such key whatever its
 *                          witness and size, unguarded (tests); 0 = six transforms.  The proof is the same group element
 *                          either way; of the phase API only zg_prove_h_eval_enqueue_fold_dev folds.  Measured (config-4 shape, one box, alternated):
 *                          2^22 10.3 -> 9.4 ms, 2^21 5.0 -> 4.7, 2^20 2.97 -> 2.92, 2^19 and below slower (hence the limit). */
#define ZG_OPT_C_FOLD 22
#define ZG_OPT_C_FOLD_MIN_LOG_M 23
/*   ZG_OPT_UPLOAD_GATES    value = log2 of the smallest witness it applies to (default 23; 0 = never): zg_prove hands a witness of
 *                          2^value values or more that does not pack (field elements) over in three chunks, each classified
 *                          behind its copy, and the l multiexp runs in three parts gated on them: its first parts sort and
 *                          accumulate while the rest of the witness is on the bus -- the one stretch of such a proof in which
 *                          the GPU idled.  A lone context only.  Measured, values alternated in one process: 2^24 uniform witness
 *                          124.5 -> 119.1 ms; 2^22 33.33 vs 33.34 (three jobs' tails cost what the 2.3 ms window buys). */
#define ZG_OPT_UPLOAD_GATES 24
int zg_set_option(zg_ctx* ctx, int option, int64_t value);
int zg_sync(zg_ctx* ctx);

/* ---- device memory plumbing ---------------------------------------------------------------- */
int zg_dev_alloc(zg_ctx* ctx, size_t bytes, void** d_ptr);
int zg_dev_free(zg_ctx* ctx, void* d_ptr);
int zg_h2d(zg_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int zg_d2h(zg_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* ---- multi-scalar multiplication -----------------------------------------------------------
 * Replaces bellman_ce::multiexp::multiexp(pool, (bases, skip), density_map, exponents) for
 * G = G1 and G = G2 (8 calls inside create_proof; call site prover/src/groth16/prover.rs:173).
 *
 * Host form (bellman semantics): `scalars` has n_scalars entries; `density_bits` (nullable = full
 * density) is an LSB-first bitmap over the scalars; `bases` holds exactly popcount(density)
 * points, consumed sequentially for set bits.  Mismatch -> ZG_E_QUERY_SIZE. */
int zg_msm_g1(zg_ctx* ctx, const uint8_t* bases64, size_t n_bases, const uint8_t* scalars32,
              const uint64_t* density_bits, size_t n_scalars, uint8_t out_jac[96]);
int zg_msm_g2(zg_ctx* ctx, const uint8_t* bases128, size_t n_bases, const uint8_t* scalars32,
              const uint64_t* density_bits, size_t n_scalars, uint8_t out_jac[192]);
/* Device form: n bases resident in HBM; base i is multiplied by d_scalars[d_scalar_index[i]]
 * (or d_scalars[i] when d_scalar_index is NULL). */
int zg_msm_g1_dev(zg_ctx* ctx, const void* d_bases, const void* d_scalars,
                  const uint32_t* d_scalar_index, size_t n, uint8_t out_jac[96]);
int zg_msm_g2_dev(zg_ctx* ctx, const void* d_bases, const void* d_scalars,
                  const uint32_t* d_scalar_index, size_t n, uint8_t out_jac[192]);

/* Resident bases: bellman's multiexp takes its bases as a shared `Arc<Vec<G::Affine>>` that outlives the call
 * (SourceBuilder); the analogue here is a handle that keeps n affine bases in HBM (converted to the device form once)
 * and, if table_budget_gb > 0 and it pays, a fixed-base window table 2^(c w) P_i over them (see zg_params_prepare);
 * table_budget_gb < 0: a table within |table_budget_gb| is built whenever one fits (the cost model is not asked).
 * zg_msm_resident takes host scalars (n <= the handle's n, canonical LE; pinned memory from zg_host_alloc gives PCIe
 * speed) and overlaps their chunked upload with the multiexp, which is BASELINE.json's "G1 MSM points/s" measurement:
 * bases resident, scalars uploaded.  zg_msm_resident_dev takes scalars already in HBM.  dense_hint != 0 promises that
 * essentially no scalar is 0 or 1 (uniform scalars: skips a classification pass); results are exact either way.
 * info: n, table bytes, c, digits per scalar. */
typedef struct zg_msm_bases zg_msm_bases;
int zg_msm_bases_g1(zg_ctx* ctx, const uint8_t* bases64, size_t n, double table_budget_gb, zg_msm_bases** out);
int zg_msm_bases_g1_dev(zg_ctx* ctx, const void* d_bases, size_t n, double table_budget_gb, zg_msm_bases** out);
int zg_msm_bases_g2(zg_ctx* ctx, const uint8_t* bases128, size_t n, double table_budget_gb, zg_msm_bases** out);
void zg_msm_bases_free(zg_msm_bases* b);
int zg_msm_bases_info(const zg_msm_bases* b, uint64_t info[4]);
int zg_msm_resident(zg_ctx* ctx, zg_msm_bases* b, const uint8_t* scalars32, size_t n, int dense_hint, uint8_t* out_jac);
int zg_msm_resident_dev(zg_ctx* ctx, const zg_msm_bases* b, const void* d_scalars, size_t n, int dense_hint,
                        uint8_t* out_jac);
/* Sum of `count` Jacobian G1 points (96 bytes each, as returned above): combines the partial results of a multiexp
 * sharded over several GPUs once they have been gathered.  Host, no ctx. */
int zg_g1_sum(const uint8_t* jac96, size_t count, uint8_t out_jac[96]);
/* Pinned (page-locked) host memory for witnesses and scalars handed to zg_prove / zg_msm_resident. */
int zg_host_alloc(zg_ctx* ctx, size_t bytes, void** out);
int zg_host_free(zg_ctx* ctx, void* p);

/* ---- Fr number-theoretic transform ---------------------------------------------------------
 * Replaces bellman_ce::domain::EvaluationDomain::{fft, ifft, coset_fft, icoset_fft} (inside
 * create_proof).  m = 2^log_m Montgomery-LE elements, natural order in and out, in place.
 * inverse != 0 applies omega^-1 and the 1/m factor; coset != 0 applies the g^i (forward) or
 * g^-i (inverse) distribution with g = Fr::multiplicative_generator() = 7.
 * log_m >= 28 -> ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE (Fr::S = 28). */
int zg_ntt_fr(zg_ctx* ctx, uint8_t* inout, uint32_t log_m, int inverse, int coset);
int zg_ntt_fr_dev(zg_ctx* ctx, void* d_inout, uint32_t log_m, int inverse, int coset);

/* ---- H polynomial ----------------------------------------------------------------------------
 * Replaces the block of create_proof that turns the a, b, c evaluation vectors into the
 * coefficients of H = (A*B - C)/Z: from_coeffs, ifft x3, coset_fft x3, mul_assign, sub_assign,
 * divide_by_z_on_coset, icoset_fft, into_coeffs, truncate, into_repr.
 * a, b, c: n_rows Montgomery-LE values (n_rows <= m, zero-padded internally);
 * h_out: m-1 canonical-LE scalars. */
int zg_h_scalars(zg_ctx* ctx, const uint8_t* a, const uint8_t* b, const uint8_t* c, size_t n_rows,
                 uint32_t log_m, uint8_t* h_out);

/* ---- R1CS ------------------------------------------------------------------------------------
 * Replaces what bellman's ProvingAssignment receives through Circuit::synthesize /
 * ConstraintSystem::{alloc_input, alloc, enforce} (za side: prover/src/groth16/prover.rs:45-103,
 * LC conversion prover/src/groth16/format.rs:207-221).  Variables are flat: index 0 is ONE, then
 * the other public inputs, then the aux variables (ordering rule: prover.rs:55-88).  The three
 * matrices are CSR: row_ptr[k] has n_constraints+1 u64 entries, col[k] u32 variable indices,
 * coeff[k] 32-byte canonical-LE Fr coefficients; k = 0,1,2 for A, B, C with <A,w>*<B,w> = <C,w>
 * (za's QEQ a*b+c=0 is passed as a*b = -c, prover.rs:98).  The `input_i * 0 = 0` rows bellman
 * appends are added internally.  Density maps are derived here by MENTION, as bellman does. */
int zg_r1cs_load(zg_ctx* ctx, uint32_t n_inputs, uint32_t n_aux, uint64_t n_constraints,
                 const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                 const uint8_t* const coeff[3], zg_r1cs** out);
void zg_r1cs_free(zg_r1cs* r);
/* n_inputs, n_aux, n_constraints, log_m, a_aux_density_total, b_input_density_total,
 * b_aux_density_total */
int zg_r1cs_info(const zg_r1cs* r, uint64_t info[7]);
/* Constraints::satisfies_with_signals (prover/src/groth16/prover.rs:155, compiler/src/types/constraint.rs:29-67), on the
 * GPU: *first_bad = index of the first constraint with <A,w><B,w> != <C,w>, or UINT64_MAX if all hold.  witness32 is a
 * host pointer (canonical LE, as for zg_prove). */
int zg_r1cs_check(zg_ctx* ctx, const zg_r1cs* r, const uint8_t* witness32, uint64_t* first_bad);
/* Evaluate a = A*w, b = B*w, c = C*w on the GPU (n_constraints + n_inputs rows each, Montgomery LE). */
int zg_r1cs_eval(zg_ctx* ctx, const zg_r1cs* r, const uint8_t* witness32, uint8_t* a_out,
                 uint8_t* b_out, uint8_t* c_out);

/* ---- proving key -----------------------------------------------------------------------------
 * zg_params_load replaces bellman_ce::groth16::Parameters::<Bn256>::read(reader, checked)
 * (prover/src/groth16/format.rs:285): vk (alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1,
 * delta_g2, u32be |ic|, ic[]), then u32be-length-prefixed h, l, a, b_g1, b_g2, points uncompressed
 * big-endian.  `checked` enables the tests Parameters::read(.., true) makes per point -- on the curve, and for G2
 * in the order-r subgroup (pairing_ce's into_affine); identity inside h/l/a/b_* is rejected either way.  The five queries
 * are decoded and validated ON THE DEVICE (csrc/keyload.hip: the raw bytes are uploaded, one kernel per query writes the
 * resident arrays and reports the first bad point), the verifying key on the host; the errors are bellman's, worded as
 * the host-only parser words them.  rank/world shard every query into `world` contiguous chunks and keep chunk `rank`
 * in HBM (world = 1: everything); a checked load validates the whole key on every rank.  *consumed receives the number of
 * bytes read. */
int zg_params_load(zg_ctx* ctx, const uint8_t* bellman_params, size_t len, int checked, int rank,
                   int world, zg_params** out, size_t* consumed);
/* Same, from Montgomery-LE arrays already in memory (what a Rust caller holding a
 * Parameters<Bn256> has).  vk5 = alpha_g1(64) beta_g1(64) beta_g2(128) delta_g1(64) delta_g2(128). */
int zg_params_from_mont(zg_ctx* ctx, const uint8_t vk5[448], const uint8_t* h, size_t n_h,
                        const uint8_t* l, size_t n_l, const uint8_t* a, size_t n_a,
                        const uint8_t* b_g1, size_t n_b_g1, const uint8_t* b_g2, size_t n_b_g2,
                        int rank, int world, zg_params** out);
void zg_params_free(zg_params* p);
/* Parameters::write (prover/src/groth16/format.rs:250) of a resident key that carries its verifying key (one loaded by
 * zg_params_load / zg_pk_load or made by zg_setup*; world = 1).  *written always receives the size; ZG_E_BUFFER_TOO_SMALL
 * if cap is insufficient (or out is NULL). */
int zg_params_write(zg_ctx* ctx, const zg_params* p, uint8_t* out, size_t cap, size_t* written);
/* Only the verifying key (bellman VerifyingKey::write; host data, no GPU traffic): what zg_verify / zg_vk_json take. */
int zg_params_write_vk(const zg_params* p, uint8_t* out, size_t cap, size_t* written);
/* Skip za's proving-key container header (u32be-len AST blob, constraints, ignore_signals;
 * prover/src/groth16/format.rs:223-247) and return the offset of the bellman Parameters body. */
int zg_pk_params_offset(const uint8_t* pk, size_t len, size_t* offset);

/* ---- circuits shipped inside a za proving key --------------------------------
 * zg_pk_load reads a whole `proving.key` as `za setup` writes it (write_pk, prover/src/groth16/format.rs:223-252;
 * read_pk :254-293): the bincode'd constraints become the CSR R1CS in the variable numbering of
 * CircomCircuit::synthesize (prover.rs:45-103: ONE, main public signals as inputs, ignore_signals skipped, the rest
 * aux; a*b + c = 0 becomes A = a, B = b, C = -c), the bellman Parameters body becomes the resident key.
 * zg_witness_from_binwitness turns the evaluator's `.binwitness` dump (compiler/src/tester/embeeded.rs:47-57,
 * FS::write_256_w32 at compiler/src/algebra/fs.rs:116-126) into the witness vector zg_prove expects. */
int zg_pk_load(zg_ctx* ctx, const uint8_t* pk, size_t len, int checked, int rank, int world, zg_r1cs** r1cs_out,
               zg_params** params_out);
int zg_witness_from_binwitness(zg_ctx* ctx, const zg_r1cs* r, const uint8_t* binwitness, size_t len,
                               uint8_t* witness32_out, size_t cap);

/* ---- set-up -------------------------------------------------------------------
 * zg_setup replaces bellman_ce::groth16::generate_parameters(circuit, g1, g2, alpha, beta, gamma, delta, tau), the
 * deterministic core of generate_random_parameters (prover/src/groth16/prover.rs:122), with the standard generators.
 * toxic = alpha, beta, gamma, delta, tau as five canonical-LE scalars.  The key stays resident in *out; if
 * params_out != NULL the bellman Parameters::write bytes (format.rs:250) are stored there (ZG_E_BUFFER_TOO_SMALL if
 * cap is insufficient; *written always receives the needed size).  Errors: ZG_E_UNCONSTRAINED_VARIABLE (an aux
 * variable's l query is the identity), ZG_E_DIVISION_BY_ZERO (gamma or delta = 0), ZG_E_BAD_FORMAT (a query polynomial
 * vanishes at tau although the variable is mentioned: the key would not match the prover's density maps). */
int zg_setup(zg_ctx* ctx, const zg_r1cs* r1cs, const uint8_t toxic[160], zg_params** out, uint8_t* params_out,
             size_t cap, size_t* written);
int zg_setup_random(zg_ctx* ctx, const zg_r1cs* r1cs, zg_params** out, uint8_t* params_out, size_t cap, size_t* written);

/* ---- prove -----------------------------------------------------------------------------------
 * Replaces bellman_ce::groth16::create_proof(circuit, params, r, s) -- the deterministic core of
 * create_random_proof (prover/src/groth16/prover.rs:173).  witness32 = (n_inputs + n_aux)
 * canonical-LE scalars, witness[0] must be 1.  r, s canonical LE.
 * Errors: ZG_E_UNEXPECTED_IDENTITY (delta = 0 or an identity proof element),
 * ZG_E_QUERY_SIZE (key does not match the circuit's densities), ZG_E_ASSIGNMENT_MISSING (NULL witness),
 * ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE. */
int zg_prove(zg_ctx* ctx, const zg_params* params, const zg_r1cs* r1cs, const uint8_t* witness32,
             const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[256]);
/* Same with the witness already resident in HBM (the timed form in bench.py). */
int zg_prove_dev(zg_ctx* ctx, const zg_params* params, const zg_r1cs* r1cs, const void* d_witness,
                 const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[256]);
/* A witness in PACKED form, for circuits whose signals are mostly bits (a SHA-256 circuit: > 99 %; the upload of the
 * 32-byte-per-value form is 2.3 ms of a 14 ms proof at 2^22, the packed form of the same witness is 1-6 MB).  Not a
 * bellman interface: the glue that turns za's signal values (`FS`, prover/src/groth16/format.rs:203-218) into the
 * array zg_prove takes can emit this form instead.
 *   classes      2 bits per value, four values per byte, value i in bits 2*(i%4).. of byte i/4:
 *                0 = the value 0, 1 = the value 1, 2 = any other value, 3 = invalid
 *   other_index  the positions of the class-2 values, strictly increasing
 *   others32     their values, canonical LE, in that order
 * zg_witness_pack (host only) converts a witness32 array; it returns ZG_E_BUFFER_TOO_SMALL (and the needed count in
 * *n_others) when more than `cap` values are of class 2.  zg_prove_packed = zg_prove on such a witness;
 * ZG_E_BAD_FORMAT if the three arrays do not describe exactly one value per variable. */
int zg_witness_pack(const uint8_t* witness32, size_t n_values, uint8_t* classes, uint32_t* other_index,
                    uint8_t* others32, size_t cap, size_t* n_others);
int zg_prove_packed(zg_ctx* ctx, const zg_params* params, const zg_r1cs* r1cs, const uint8_t* classes,
                    const uint32_t* other_index, const uint8_t* others32, size_t n_others,
                    const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[256]);
/* create_random_proof: r, s from the OS entropy source. */
int zg_prove_random(zg_ctx* ctx, const zg_params* params, const zg_r1cs* r1cs,
                    const uint8_t* witness32, uint8_t proof_out[256]);

/* Phases of zg_prove_dev for one-process-per-GPU sharding:
 *   rank 0:     zg_prove_h_dev         -> m canonical-LE H scalars in HBM (last one is zero)
 *   every rank: zg_prove_partials_dev  -> its partial sums over its shard of h, l, a, b_g1, b_g2
 *   rank 0:     zg_prove_combine       -> adds the gathered partials, applies r, s, normalises.
 * partial layout (ZG_PARTIALS_BYTES): h(96) l(96) a(96) b_g1(96) b_g2(192), Jacobian Montgomery LE.
 * d_h points at the H scalars of THIS rank's h-shard (zg_params_shard_range(.., 0, ..)). */
#define ZG_PARTIALS_BYTES 576
int zg_prove_h_dev(zg_ctx* ctx, const zg_r1cs* r1cs, const void* d_witness, void* d_h_out);
int zg_prove_partials_dev(zg_ctx* ctx, const zg_params* params, const zg_r1cs* r1cs,
                          const void* d_witness, const void* d_h_shard,
                          uint8_t partials[ZG_PARTIALS_BYTES]);
/* The same with a subset of the five sums (mask bit 0 h, 1 l, 2 a, 3 b_g1, 4 b_g2; skipped ones are returned as the
 * identity), and a non-blocking form of zg_prove_h_dev: lets every rank run its witness MSMs (bits 1-4, no H needed)
 * while rank 0's NTT pipeline is still in flight; zg_sync before reading d_h_out. */
int zg_prove_partials_ex(zg_ctx* ctx, const zg_params* params, const zg_r1cs* r1cs, const void* d_witness,
                         const void* d_h_shard, unsigned mask, uint8_t partials[ZG_PARTIALS_BYTES]);
int zg_prove_h_enqueue_dev(zg_ctx* ctx, const zg_r1cs* r1cs, const void* d_witness, void* d_h_out);
/* H in EVALUATION form: d_e_out receives m canonical-LE scalars e[p] = ((A B - C) / Z)(g omega^bitrev(p)) -- the values
 * bellman's create_proof holds just before its last transform (icoset_fft).  zg_prove* multiplies them into the key's
 * TRANSFORMED h bases (built once per key: the same linear map applied to the points), so a proof runs six transforms
 * instead of seven and gives the same group element; zg_set_option(ZG_OPT_H_EVAL, 0) keeps the coefficient form.  Blocking. */
int zg_prove_h_eval_dev(zg_ctx* ctx, const zg_r1cs* r1cs, const void* d_witness, void* d_e_out);
/* The sharded form of the same: zg_prove_h_eval_enqueue_dev is zg_prove_h_enqueue_dev for evaluation-form scalars
 * (non-blocking; zg_sync before reading d_e_out), and zg_prove_partials_ex with ZG_MASK_H_EVAL or-ed into its mask takes
 * d_h_shard as this rank's range of them -- positions [lo, hi) = chunk `rank` of ceil(m / world) -- and multiplies them
 * into the key's slice of the transformed h bases.  zg_params_h_eval_info: {1 if the key has such bases, log2(m), lo, hi};
 * a sharded key gets them in zg_params_load (when |h| = 2^k - 1, as bellman writes it), an unsharded one on first use.
 * Every rank of a proof must use the same form: agree on min(info[0]) over the ranks first (za_amd/dist.py does). */
#define ZG_MASK_H_EVAL 0x20u
int zg_prove_h_eval_enqueue_dev(zg_ctx* ctx, const zg_r1cs* r1cs, const void* d_witness, void* d_e_out);
/* The same for a caller that names the key, so that the rank that runs the transforms of a sharded proof (rank 0) can drop
 * C's two as well (ZG_OPT_C_FOLD): where `params` has the bases of the folded C rows -- rank 0's shard of a sharded key gets
 * them in zg_params_load, an unsharded key on first use; decided from the rows' values as in zg_prove -- the pipeline runs
 * FOUR transforms, d_e_out receives a b / Z instead of (a b - c) / Z, and the next zg_prove_partials_ex on this context with
 * this key and the l query in its mask adds the multiexp of the rows' values to its l partial (*folded = 1: the rows are
 * ARMED -- with ZG_OPT_C_FOLD = 1 the device decides per proof, from the digits this witness's rows evaluate to, whether the
 * C term travels in the l partial (four transforms) or stays in the scattered scalars (six; the rows are then struck from the
 * l multiexp); zg_last_stats[3] after that partials call says which.  Either way the sum of the partials is the same).  The other
 * ranks do what they did: their h shards multiply whatever scalars they are sent.  Any other key: *folded = 0 and
 * exactly zg_prove_h_eval_enqueue_dev.  The sum of all ranks' partials is the same group element either way.
 * (What it costs rank 0 when its shard is loaded: a second group transform of the h query and 128 m bytes, whether or not
 * the circuit's rows turn out to fold; zg_set_option(ZG_OPT_C_FOLD, 0) before zg_params_load leaves both out.) */
int zg_prove_h_eval_enqueue_fold_dev(zg_ctx* ctx, const zg_params* params, const zg_r1cs* r1cs, const void* d_witness,
                                     void* d_e_out, int* folded);
int zg_params_h_eval_info(const zg_params* params, uint64_t info[4]);
/* zg_prove's hand-over of a HOST witness on its own, for callers that drive the phases themselves (every rank of a sharded
 * proof needs the witness): n = n_inputs + n_aux canonical-LE scalars, copied -- in packed form by a few host threads
 * when they are mostly zeros and ones -- into the context's own buffer; *d_witness receives its device
 * address, valid until the context's next hand-over (this call or zg_prove) or zg_free; *bytes_sent (may be NULL) what
 * crossed PCIe.  Queued on the context's stream, not waited for: the phase entry points order themselves behind it. */
int zg_witness_to_device(zg_ctx* ctx, const uint8_t* witness32, size_t n, void** d_witness, size_t* bytes_sent);
int zg_prove_combine(zg_ctx* ctx, const zg_params* params, const uint8_t* partials_all, int world,
                     const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[256]);
/* [lo, hi) of query `which` (0 h, 1 l, 2 a, 3 b_g1, 4 b_g2) held by this params handle. */
int zg_params_shard_range(const zg_params* p, int which, uint64_t* lo, uint64_t* hi);
/* Full (unsharded) length of query `which`. */
int zg_params_query_total(const zg_params* p, int which, uint64_t* total);
/* Optional, once per (key, circuit): builds the fixed-base window tables now instead of inside the first
 * zg_prove* call with this key (bellman has no equivalent: its multiexp takes the bases as they are,
 * prover.rs:173).  A table holds 2^(c*w) * P_i for every digit position w, so a multiexp needs one bucket set
 * instead of one per window; it costs nwin (about 13 at 2^22) times the query's HBM footprint.  d_witness (device,
 * as for zg_prove_dev) sizes the l / a / b tables from its share of 0/1 values; NULL builds the h table only.
 * Call it before sharing `params` between host threads.  zg_set_option(ZG_OPT_TABLES, mask) says which queries may have
 * a table (bit mask: 1 h, 2 l, 4 a, 8 b_g1, 16 b_g2; 0 = none), ZG_OPT_TABLES_MAX_MB caps one table (default: 60 % of the
 * free HBM); the library reads no environment variable for either.  Without room the plain per-window schedule is
 * used.  Results are identical either way. */
int zg_params_prepare(zg_ctx* ctx, const zg_params* params, const zg_r1cs* r1cs, const void* d_witness);
/* (A key whose C rows are folded builds the folded query's table, 5, and NOT the plain l query's: zg_prove* never reads the
 * latter for such a key -- a proof the per-proof guard runs with six transforms takes the folded query with its rows struck
 * out.  The plain l table is built on demand only if the plain phase API, zg_prove_partials_ex without an armed fold, is used
 * on the key; then both stay resident.)
 * Which table query `which` (0 h, 1 l, 2 a, 3 b_g1, 4 b_g2; 5 = the l query with the C rows folded in, ZG_OPT_C_FOLD --
 * a key that has it builds no table for the plain l query) currently has: kind 0 none, 1 window table (dense
 * assignment), 2 subset-sum table (sums over groups of four positions, for an assignment that is at least 90 % zeros
 * and ones: a group of four 0/1 scalars costs one mixed add instead of one per 1; 3.75x the query's footprint), and
 * its size in bytes. */
int zg_params_table_info(const zg_params* params, int which, int* kind, uint64_t* bytes);

/* ---- verify (host CPU) ---------------------------------------------------------
 * zg_verify replaces bellman_ce::groth16::{prepare_verifying_key, verify_proof} (prover/src/groth16/prover.rs:191-200,
 * helper.rs:149-159; za2c `verify`, binding/c/native/src/lib.rs:101-117).  vk_bytes: bellman VerifyingKey::write
 * bytes -- which is also the prefix of Parameters::write, so a whole key may be passed.  public_inputs32: n_public
 * canonical-LE scalars (ONE excluded).  *ok = 1 if e(A,B) = e(alpha,beta) e(sum ic_i x_i, gamma) e(C,delta), else 0.
 * ZG_E_MALFORMED_VERIFYING_KEY if n_public + 1 != |ic| (as bellman); ZG_E_BAD_FORMAT for off-curve proof points.
 * Needs no GPU and no ctx. */
int zg_verify(const uint8_t* vk_bytes, size_t vk_len, const uint8_t proof[256], const uint8_t* public_inputs32,
              size_t n_public, int* ok);
/* vk JSON exactly as JsonVerifyingKey::to_json emits it (prover/src/groth16/format.rs:130-167): fields alpha_g1,
 * beta_g1, beta_g2, delta_g1, delta_g2, gamma_g2, ic, input_names; names = n_names NUL-terminated strings back to back. */
int zg_vk_json(const uint8_t* vk_bytes, size_t vk_len, const char* names, size_t n_names, char* out, size_t cap);

/* The readers: JsonVerifyingKey::{from_json, to_bellman} (prover/src/groth16/format.rs:169-194) and
 * JsonProofAndInput::json_to_bellman (format.rs:101-121).  Coordinates are strings holding a decimal or a 0x-prefixed
 * hexadecimal number (str_to_fq -> FS::parse, format.rs:33-36, compiler/src/algebra/fs.rs:43-55) below q, points must lie
 * on their curve (try_from_coordinates; G2 also in the order-r subgroup); public inputs are what Fr::from_str accepts
 * (canonical decimal below r) -- anything else is ZG_E_BAD_FORMAT with a Rust-Debug-shaped message in
 * zg_last_error_thread() (`BadFormat("bad format")`, `Json("missing field `ic`")`, ...).  Every struct field must be
 * present (serde), unknown fields are ignored.
 * zg_vk_from_json writes bellman VerifyingKey::write bytes (the form zg_verify / zg_vk_json / zg_verifier_solidity
 * take); *written always receives the size needed.  zg_proof_from_json writes the 256-byte proof and n canonical-LE
 * public inputs (ZG_E_BUFFER_TOO_SMALL if cap_inputs < 32 n; *n_public is set either way).  No GPU, no ctx. */
int zg_vk_from_json(const char* json, uint8_t* vk_out, size_t cap, size_t* written);
int zg_proof_from_json(const char* json, uint8_t proof_out[256], uint8_t* public_inputs32, size_t cap_inputs,
                       size_t* n_public);
/* generate_solidity (prover/src/groth16/ethereum.rs:216-261): the verifier contract `za setup` writes by default
 * (cli/src/main.rs:33), G2 coordinates printed c1 first (ethereum.rs:28-31,231-236), numbers as FqRepr prints them.
 * names: the public input names, n_names NUL-terminated strings back to back (`// input = [...]` and the length of
 * verifyTx's input array).  *written = bytes needed including the NUL. */
int zg_verifier_solidity(const uint8_t* vk_bytes, size_t vk_len, const char* names, size_t n_names, char* out,
                         size_t cap, size_t* written);
/* Message of the last failing call made by THIS thread (ctx-less entry points have no zg_last_error). Never NULL. */
const char* zg_last_error_thread(void);

/* ---- formats ---------------------------------------------------------------------------------
 * proof.json exactly as JsonProofAndInput::write emits it (prover/src/groth16/format.rs:80-128):
 * {"a":[x,y],"b":[[x.c0,x.c1],[y.c0,y.c1]],"c":[x,y],"public_inputs":[...]} with 0x-prefixed
 * 64-digit lowercase hex coordinates and decimal public inputs (n_public canonical-LE scalars,
 * ONE excluded).  Returns ZG_E_BUFFER_TOO_SMALL if cap is insufficient (cf. binding/c/native/src/lib.rs:22-30). */
int zg_proof_json(const uint8_t proof[256], const uint8_t* public_inputs32, size_t n_public,
                  char* out, size_t cap);

/* ---- timing of the last zg_prove* call on this ctx (milliseconds, host clock):
 * [0] witness bytes uploaded by zg_prove (the copy itself is part of [5]), [1] R1CS evaluation, [2] the H pipeline (seven
 * transforms in coefficient form, six with the key's h bases in evaluation form, four with the C rows folded: zg_last_stats[3]),
 * [3] MSMs, [4] combine, [5] total. */
int zg_last_timings(const zg_ctx* ctx, double ms[6]);
/* ---- timing of the last zg_params_load / zg_pk_load on this ctx (milliseconds, host clock): [0] verifying key + container
 * walk on the host, [1] the five queries: upload, decoding and (checked) validation on the device, [2] zg_pk_load only: the
 * constraint sections of the container -> CSR (host threads), [3] zg_pk_load only: zg_r1cs_load of that CSR. */
int zg_last_load_timings(const zg_ctx* ctx, double ms[4]);
/* ---- what the last zg_prove* call (or, for the phase API, everything since the last one) queued on the GPU:
 * [0] kernel launches (a key's FIRST proof also counts its one-time work: table builds, the group transforms of the h query), [1] bit mask of the witness queries (bit 1 l, 2 a, 3 b_g1, 4 b_g2) whose multiexps shared one digit
 * sort over the witness positions in the last batch of multiexps (0 = none); bit 8: the l multiexp ran in parts behind a
 * witness that was handed over in chunks (ZG_OPT_UPLOAD_GATES), [2] 1 if the b_g1 multiexp ran over the b_g2
 * multiexp's sorted entries (same scalars, same positions, tables of one layout), [3] 1 if the proof ran with four
 * transforms and its C rows folded into the l multiexp (ZG_OPT_C_FOLD). */
int zg_last_stats(const zg_ctx* ctx, uint64_t stats[4]);

/* ---- optional per-kernel timing (HIP events on the ctx stream), used by bench.py for the roofline
 * line.  id: 0 msm_accumulate_g1 (the witness queries' G1 jobs), 1 msm_accumulate_g2, 2 msm_digit_sort,
 * 3 msm_bucket_reduce, 4 ntt_pass, 5 ntt_pointwise, 6 r1cs_eval, 7 msm_accumulate_g1_dense (the h query's job and
 * dense-hinted standalone multiexps: every point reaches the kernel), 8 msm_digit_sort_dense and 9 msm_bucket_reduce_dense (the
 * same jobs' sort and reduction: they are not in 2 and 3).  alg_bytes = algorithmic HBM bytes of the timed launches.
 * zg_profile_enable: 0 off, 1 = classes 0, 1 and 7 only (the dominant kernels: about five event pairs per proof),
 * 2 = every class (about a hundred event pairs per proof, which cost 0.5 ms of a proof's wall time). */
#define ZG_PROFILE_KERNELS 10
int zg_profile_enable(zg_ctx* ctx, int on);
int zg_profile_reset(zg_ctx* ctx);
int zg_profile_get(zg_ctx* ctx, int id, double* ms, uint64_t* launches, uint64_t* alg_bytes);
/* work units of the timed launches: mixed additions for the msm_accumulate classes, butterflies for ntt_pass */
int zg_profile_get_units(zg_ctx* ctx, int id, uint64_t* units);
const char* zg_profile_name(int id);

#ifdef __cplusplus
}
#endif
#endif
