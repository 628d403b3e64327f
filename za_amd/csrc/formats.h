// Wire formats on the za side of the boundary (host only).
#pragma once
#include <sys/mman.h>
#include <type_traits>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace zg {

// Parsed bellman Parameters<Bn256> body, points converted to Montgomery affine.
struct ParsedParams {
    G1Affine alpha_g1, beta_g1, delta_g1;
    G2Affine beta_g2, gamma_g2, delta_g2;
    std::vector<G1Affine> ic, h, l, a, b_g1;
    std::vector<G2Affine> b_g2;
};

// Parameters::read(reader, checked) -- reference call site prover/src/groth16/format.rs:285.  Everything on the host: the
// cross-check of the device path below (zg_params_inspect) and the reader of verifying keys.
int parse_bellman_params(Ctx* ctx, const uint8_t* buf, size_t len, bool checked, ParsedParams* out, size_t* consumed);
// Parameters::write -- reference call site prover/src/groth16/format.rs:250.
void write_bellman_params(const ParsedParams& p, std::vector<uint8_t>* out);
// VerifyingKey::read / write: the prefix of the Parameters body up to and including ic (alpha_g1, beta_g1, beta_g2,
// gamma_g2, delta_g1, delta_g2, u32be |ic|, ic); points are curve- and subgroup-checked.
int parse_bellman_vk(Ctx* ctx, const uint8_t* buf, size_t len, ParsedParams* out);
void write_bellman_vk(const ParsedParams& p, std::vector<uint8_t>* out);
// za proving-key container header (format.rs:223-247): returns offset of the Parameters body.
int pk_params_offset(const uint8_t* pk, size_t len, size_t* offset);
// proof.json (format.rs:80-128)
int proof_json(const uint8_t proof[256], const uint8_t* public_inputs32, size_t n_public, std::string* out);

// A host array that is NOT zero-filled when it is sized (the threads that fill it touch it first).  From 4 MB on it is a
// mapping of its own that asks for transparent huge pages: the constraints of a 2^22 circuit are 0.6 GB of such arrays, and in
// 4 KB pages giving them back cost 0.13 s (150 000 pages unmapped one by one) of a 0.39 s key read.
template <class T>
struct RawBuf {
    T* ptr = nullptr;
    size_t n = 0, mapped = 0;   // mapped: bytes of the mapping (0: from the heap)
    RawBuf() = default;
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    RawBuf(RawBuf&& o) noexcept : ptr(o.ptr), n(o.n), mapped(o.mapped) { o.ptr = nullptr; o.n = o.mapped = 0; }
    RawBuf& operator=(RawBuf&& o) noexcept {
        if (this != &o) {
            release();
            ptr = o.ptr; n = o.n; mapped = o.mapped;
            o.ptr = nullptr; o.n = o.mapped = 0;
        }
        return *this;
    }
    ~RawBuf() { release(); }
    void release() {
        if (ptr) {
            if (mapped) (void)munmap(ptr, mapped);
            else delete[] ptr;
        }
        ptr = nullptr;
        n = mapped = 0;
    }
    void alloc(size_t k) {
        static_assert(std::is_trivial<T>::value, "RawBuf holds plain data");
        release();
        const size_t bytes = (k ? k : 1) * sizeof(T);
        constexpr size_t HUGE = (size_t)2 << 20;
        if (bytes >= 2 * HUGE) {
            const size_t len = (bytes + HUGE - 1) & ~(HUGE - 1);
            void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, len, MADV_HUGEPAGE);   // (a hint: without it the array works as before)
                ptr = static_cast<T*>(m);
                mapped = len;
            }
        }
        if (!ptr) ptr = new T[k ? k : 1];
        n = k;
    }
    T* data() const { return ptr; }
    size_t size() const { return n; }
};

// R1CS recovered from a za proving key (format.rs:223-247) in the variable numbering of
// CircomCircuit::synthesize (prover.rs:45-103)
struct PkR1cs {
    uint32_t n_inputs = 0, n_aux = 0;
    uint64_t n_constraints = 0;
    size_t params_offset = 0;
    std::vector<uint32_t> var_of_signal;   // UINT32_MAX for ignored signals
    RawBuf<uint64_t> row_ptr[3];
    RawBuf<uint32_t> col[3];
    RawBuf<uint8_t> coeff[3];
};
struct PkWalk {                        // where the container's sections lie
    std::vector<size_t> at;            // constraint i spans [at[i], at[i + 1] - 4)
    std::vector<uint32_t> ignore;      // ignore_signals
    size_t params_offset = 0;          // the bellman Parameters body
};
int pk_walk(Ctx* ctx, const uint8_t* pk, size_t len, PkWalk* w);
int parse_pk_r1cs(Ctx* ctx, const uint8_t* pk, size_t len, const PkWalk& walk, uint32_t n_public, uint32_t n_aux, PkR1cs* out);
int witness_from_binwitness(Ctx* ctx, const std::vector<uint32_t>& var_of_signal, uint32_t n_vars, const uint8_t* bw,
                            size_t len, uint8_t* witness_out);

// ---- Parameters::read with the five queries decoded and validated on the device (keyload.hip) -------------------------
// What is wrong with a point, in the order pairing_ce's readers test it; KEY_PT_INFINITY = a well-formed identity inside a
// query (bellman: "point at infinity").
enum KeyPointStatus { KEY_PT_OK = 0, KEY_PT_COMPRESSED, KEY_PT_BAD_INFINITY, KEY_PT_NOT_IN_FIELD, KEY_PT_NOT_ON_CURVE,
                      KEY_PT_NOT_IN_SUBGROUP, KEY_PT_INFINITY };
// the error both parsers raise for a bad point of `what` ("h", "vk.alpha_g1", ...): ZG_E_IO + bellman's wording
int key_point_error(Ctx* ctx, const char* what, int status);
const char* key_query_name(int which);
struct KeyQuery {
    size_t off = 0;     // first point, from the start of the Parameters body
    uint64_t n = 0;
};
// The host's part: the verifying key (parsed, validated and converted on the host: 6 points + ic), the walk over the five
// length-prefixed queries (bounds only).  out receives the vk and ic; its query vectors stay empty.
int parse_bellman_header(Ctx* ctx, const uint8_t* buf, size_t len, bool checked, ParsedParams* out, KeyQuery q[5], size_t* consumed);
bool peek_bellman_counts(const uint8_t* buf, size_t len, uint32_t* n_ic, uint32_t* n_l);   // |ic|, |l| from the length prefixes
struct KeyStage {   // the raw bytes' way through HBM: two chunk buffers used in turn
    static constexpr size_t CHUNK_BYTES = (size_t)64 << 20;
    DevBuf buf[2];
    hipEvent_t ev[2] = {nullptr, nullptr};
    int next = 0;
    ~KeyStage() {
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};
int key_decode_query_dev(Ctx* ctx, int which, const uint8_t* host_raw, uint64_t n, bool checked, uint64_t lo, uint64_t hi,
                         void* d_out, unsigned long long* d_verdict, KeyStage* stage);
int key_verdict_to_error(Ctx* ctx, const unsigned long long verdict[5]);
// Parameters::write of one query on the device: n resident points -> n * 64 / 128 bytes of pairing_ce's uncompressed encoding
int key_encode_query_dev(Ctx* ctx, int which, const void* d_points, uint64_t n, uint8_t* host_out, KeyStage* stage);

Fq2 g2_coeff_b();                       // 3/(9+u), Montgomery
bool g2_in_subgroup(const G2Affine& p); // [r]P == identity (the twist has a large cofactor; G1's is 1)
bool scalar_is_canonical(const uint32_t v[8]);   // v < r
void fq_to_be(uint8_t out[32], const Fq& mont);
bool fq_from_be(const uint8_t in[32], Fq* mont);  // false if >= q

}  // namespace zg
