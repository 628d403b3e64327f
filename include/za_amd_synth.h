/* za_amd synthetic workloads and self-test hooks -- NOT part of the drop-in boundary, and not in the product library:
 * these symbols live in libza_amd_synth.so (csrc/synth.hip), which links against libza_amd.so.
 *
 * Generators for the benchmark configurations of BASELINE.json / SURVEY.md section 8(d): the za
 * compiler (which would produce the R1CS and witness) and bellman's generate_random_parameters
 * (which would produce the proving key) are out of scope for this round, so bench.py and the
 * full-size parity tests build their inputs here.  The prover treats them like any other input.
 */
#ifndef ZA_AMD_SYNTH_H
#define ZA_AMD_SYNTH_H
#include "za_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zg_synth zg_synth;

/* Synthetic satisfiable R1CS with n_constraints + n_inputs = 2^log_m exactly and a
 * consistent witness, SplitMix64-seeded.  Row i defines a fresh aux variable:
 *   with probability bit_fraction:  (2a)*(b) = a + b - c   (c = a XOR b over earlier bit variables),
 *   otherwise:                      (k1 x + k2 y)*(k3 z + k4 t) = c   with uniform Fr coefficients.
 * bit_fraction = 0.97 gives the SHA256-Merkle-like witness statistics of config 4; 0.0 the
 * uniform-Fr config 5.  64 seed aux variables precede the defined ones. */
int zg_synth_r1cs_new(uint32_t log_m, uint32_t n_inputs, double bit_fraction, uint64_t seed, zg_synth** out);
/* info: n_inputs, n_aux, n_constraints, nnz(A), nnz(B), nnz(C) */
int zg_synth_r1cs_info(const zg_synth* s, uint64_t info[6]);
/* which: 0-2 row_ptr[k] (u64), 3-5 col[k] (u32), 6-8 coeff[k] (32 B canonical LE), 9 witness (32 B canonical LE) */
const void* zg_synth_ptr(const zg_synth* s, int which);
void zg_synth_free(zg_synth* s);

/* What `za setup` and the evaluator would leave on disk for such a circuit: the proving.key container
 * (prover/src/groth16/format.rs:223-252) around a Parameters body -- signal id = variable index, no ignored signals, an
 * empty AST vector, C's coefficients negated (the container holds a*b + c = 0) -- and the .binwitness dump
 * (compiler/src/tester/embeeded.rs:47-57) of n 32-byte canonical values.  Inputs of libza2c's prove() at bench sizes. */
int zg_synth_pk_container_write(const char* path, uint32_t n_inputs, uint32_t n_aux, uint64_t n_constraints,
                                const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                                const uint8_t* const coeff[3], const uint8_t* params, size_t params_len);
int zg_synth_binwitness_write(const char* path, const uint8_t* witness32, size_t n);

/* d_out[i] = k_i * G with k_i = splitmix64(seed + i) (a known discrete log, so MSM results can be
 * checked as (sum s_i k_i) * G).  Affine Montgomery LE, 64 B (G1) / 128 B (G2) per point. */
int zg_synth_bases_g1_dev(zg_ctx* ctx, void* d_out, size_t n, uint64_t seed);
int zg_synth_bases_g2_dev(zg_ctx* ctx, void* d_out, size_t n, uint64_t seed);
/* Proving key whose queries are zg_synth_bases_* arrays (seeds seed+1.. for h, l, a, b_g1; b_g2 uses
 * the b_g1 discrete logs) and vk elements alpha = 3G, beta = 5G, delta = 7G.  Not a valid CRS: it
 * exercises exactly the prover's arithmetic, with every discrete log known to the tests. */
int zg_synth_params_dev(zg_ctx* ctx, uint64_t n_h, uint64_t n_l, uint64_t n_a, uint64_t n_b, uint64_t seed,
                        int rank, int world, zg_params** out);
/* Chunk `rank` of `world` of every query of an unsharded resident key, as a new handle (multi-GPU bench). */
int zg_params_shard(zg_ctx* ctx, const zg_params* full, int rank, int world, zg_params** out);
/* k_i for the generators above */
uint64_t zg_synth_dlog(uint64_t seed, uint64_t i);
/* d_out[i] = uniform scalar below 2^253 from splitmix64 (canonical LE), for standalone MSM runs */
int zg_synth_scalars_dev(zg_ctx* ctx, void* d_out, size_t n, uint64_t seed);

/* Copy this handle's resident shard of query `which` (0 h, 1 l, 2 a, 3 b_g1, 4 b_g2), or with
 * which = -1 the five vk points (448 bytes, zg_params_from_mont order), back to host memory. */
int zg_params_download(const zg_params* p, int which, void* out, size_t bytes);

/* ---- self-test hooks (tests only) ----------------------------------------------------------------
 * which: 0 Fq, 1 Fr; op: 0 mul, 1 add, 2 sub, 3 inv, 4 to_mont, 5 from_mont, 6 neg, 7 sqr; operands are
 * n 32-byte Montgomery-LE values.  _host runs the library's host build of csrc/fp.h, _dev the gfx950 one. */
int zg_selftest_field_host(int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
int zg_selftest_field_dev(zg_ctx* ctx, int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
/* out = k*p (+ q when q != NULL) with the host curve code; group 1 = G1 (64 B affine), 2 = G2 (128 B). */
int zg_selftest_curve_host(int group, const uint8_t* p, const uint8_t* k32, const uint8_t* q, uint8_t* out);
/* The DEVICE point arithmetic (csrc/ecd.h) operation by operation over n operand pairs.  group 1 = G1, 2 = G2.
 * lanes: 1 = one lane per point; 2 = G2 on lane pairs (Fq2PL); 4 = G1 on four lanes (FqQL).
 * p, q: affine points (64 / 128 B, Montgomery LE, all zero = identity); zp, zq: non-zero field elements (32 / 64 B) that
 * choose the XYZZ representation (x z^2, y z^3, z^2, z^3) the operands are given to the operation in.
 * op: 0 P + Q (XYZZ + XYZZ), 1 P + Q (XYZZ + affine), 2 P - Q (XYZZ + affine with y negated lazily), 3 2P,
 *     4 (P + Q) + Q (mixed, chained), 5 (P + Q) + P (full, chained).   out: the result, affine. */
int zg_selftest_curve_dev(zg_ctx* ctx, int group, int lanes, int op, const uint8_t* p, const uint8_t* zp,
                          const uint8_t* q, const uint8_t* zq, uint8_t* out, size_t n);
/* zg_prove's host-side packer (csrc/prover.hip HostPacker) on plain memory, no GPU: `threads` host threads turn n
 * 32-byte values into class bytes ((n + 3) / 4) and the other values' (index, bytes) back to back (room for n / 8).
 * *n_others = their number, or SIZE_MAX when the witness is not of the kind that packs (more than 12.5 % other values
 * over all, or one 2^14-value stretch more than half of them) -- then zg_prove uploads the array as it is. */
int zg_selftest_pack_mt(const uint8_t* witness32, size_t n, int threads, uint8_t* classes, uint32_t* other_index,
                        uint8_t* others32, size_t* n_others);
/* The same pass sharing the array with a stand-in for the copy engine (HostPacker::RawCopier: the engine takes pieces from the
 * front while the threads pack from the back, until they meet): pieces are memcpy'd into raw_out; the stand-in accepts a
 * piece every `ready_every`-th time it is asked and takes at most cap_permille / 1000 of the values.  *first = the values it
 * took ([0, first) of raw_out are written); classes[i >> 2] describes value i for i >= first, the list holds positions
 * >= first.  ZG_E_BAD_FORMAT if the pieces were not handed over in order from value 0. */
int zg_selftest_pack_split(const uint8_t* witness32, size_t n, int threads, int cap_permille, int ready_every, uint8_t* raw_out,
                           uint8_t* classes, uint32_t* other_index, uint8_t* others32, size_t* n_others, size_t* first);
/* prod_i e(P_i, Q_i) with the library's host pairing (csrc/pairing.hip): slow = 1 the definition (points untwisted into Fq12,
 * plain powers), 0 = the path zg_verify runs.  g1: n x 64 B, g2: n x 128 B (Montgomery LE affine), n <= 4; gt_out (384 B,
 * nullable): the product's value in the library's tower representation; *is_one = 1 iff it is 1. */
int zg_selftest_pairing_product(const uint8_t* g1, const uint8_t* g2, size_t n, int slow, uint8_t* gt_out, int* is_one);
/* Parse + validate a bellman Parameters body on the host (no GPU) and check write(read(x)) == x.
 * counts: |ic|, |h|, |l|, |a|, |b_g1|, |b_g2|, bytes consumed. */
int zg_params_inspect(const uint8_t* buf, size_t len, int checked, uint64_t counts[7], char* err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif
