// libza2c.so -- the outer drop-in boundary: the four symbols of za's C binding, served by libza_amd.so.
//
// Mirrors binding/c/native/src/lib.rs (crate `za2c`, the library every other binding of za wraps; cgo prototypes
// binding/go/lib.go:6-9):
//     void verbose(int on)                                                                   lib.rs:32-48
//     int  setup (circuit_path, pk_path, verifier_type, verifier_buf, n, err_buf, en)        lib.rs:50-79
//     int  prove (pk_path, inputs, proof_buf, n, err_buf, en)                                lib.rs:81-99
//     int  verify(vk_json, proof_with_inputs_json, err_buf, en)                              lib.rs:101-117
// Return codes (lib.rs:10-13): 0 ok, 1 a caller buffer is too small (nothing is written), 2 verification failed,
// 100 any other error with Rust-Debug-shaped text in err_buf.  Strings are copied only if len < size (lib.rs:22-30).
//
// Two documented deviations, both forced by scope (the za compiler and evaluator are CPU interpreters outside the hot
// path):
//   setup: `circuit_path` names an already compiled circuit -- a za proving.key container (format.rs:223-252), whose
//          constraints + ignore_signals are re-keyed with fresh toxic waste on the GPU -- instead of a .za source file.
//          Public input names come from `<circuit_path>.names` (one per line) or default to main.public[i].
//   prove: `inputs` is {"@binwitness": "<path>"} (or the bare path): the evaluator's .binwitness dump
//          (compiler/src/tester/embeeded.rs:47-57) instead of the values of the main component's inputs.
// Everything after that -- constraint check, create_random_proof, self-verification, proof.json -- follows
// helper::prove / generate_verified_proof (helper.rs:91-147, prover.rs:139-208) on the GPU.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

#include "../../include/za_amd.h"

namespace {

const int ERR_NONE = 0, ERR_BUFFER_TOO_SMALL = 1, ERR_VERIFICATION_FAILED = 2, ERR_CUSTOM = 100;
int g_verbose = 0;

int return_string(const std::string& s, char* buffer, size_t size, int ret) {
    if (s.size() >= size || !buffer) return ERR_BUFFER_TOO_SMALL;
    memcpy(buffer, s.c_str(), s.size() + 1);
    return ret;
}
void info(const char* fmt, double v) {
    if (g_verbose) fprintf(stderr, fmt, v);
}
double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}
bool read_file(const char* path, std::vector<uint8_t>* out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        *err = std::string("Io(\"") + path + ": " + strerror(errno) + "\")";
        return false;
    }
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    if (n < 0) {   // a directory or an unseekable path
        *err = std::string("Io(\"") + path + ": not a regular file\")";
        return false;
    }
    f.seekg(0);
    out->resize((size_t)n);
    if (n) f.read((char*)out->data(), n);
    if (!f) {
        *err = std::string("Io(\"") + path + ": short read\")";
        return false;
    }
    return true;
}
// A file as read-only memory: the proving key (1.4 GB at 2^22 constraints) goes from the page cache to the GPU without
// a copy into a heap buffer first.  Falls back to read_file for what cannot be mapped.
struct FileBytes {
    const uint8_t* p = nullptr;
    size_t n = 0;
    void* map = nullptr;
    std::vector<uint8_t> heap;
    ~FileBytes() {
        if (map) munmap(map, n);
    }
    bool open(const char* path, std::string* err) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) {
            *err = std::string("Io(\"") + path + ": " + strerror(errno) + "\")";
            return false;
        }
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);   // (pre-faulted: one call instead of a fault per 64 KB)
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                map = m;
                p = (const uint8_t*)m;
                n = (size_t)st.st_size;
                close(fd);
                return true;
            }
        }
        close(fd);
        if (!read_file(path, &heap, err)) return false;
        p = heap.data();
        n = heap.size();
        return true;
    }
};
// Error text in the shape `format!("{:?}", err)` has for za's Error enum (prover/src/groth16/error.rs:1-12)
std::string err_text(zg_ctx* ctx, int rc) {
    const char* detail = ctx ? zg_last_error(ctx) : zg_last_error_thread();
    switch (rc) {
        case ZG_E_ASSIGNMENT_MISSING: case ZG_E_DIVISION_BY_ZERO: case ZG_E_UNSATISFIABLE:
        case ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE: case ZG_E_UNEXPECTED_IDENTITY: case ZG_E_MALFORMED_VERIFYING_KEY:
        case ZG_E_UNCONSTRAINED_VARIABLE:
            return std::string("Synthesis(") + zg_strerror(rc) + ")";
        case ZG_E_IO:
            return std::string("Io(\"") + detail + "\")";
        case ZG_E_BAD_FORMAT:
            // the readers already produce `BadFormat("..")` / `Json("..")` / `Algebra(..)` texts
            if (!strncmp(detail, "BadFormat(", 10) || !strncmp(detail, "Json(", 5) || !strncmp(detail, "Algebra(", 8)) return detail;
            return std::string("BadFormat(\"") + detail + "\")";
        default:
            return std::string("Unexpected(\"") + zg_strerror(rc) + ": " + detail + "\")";
    }
}
struct Handles {   // released in reverse order of creation
    zg_ctx* ctx = nullptr;
    zg_r1cs* r1cs = nullptr;
    zg_params* old_params = nullptr;
    zg_params* params = nullptr;
    ~Handles() {
        if (params) zg_params_free(params);
        if (old_params) zg_params_free(old_params);
        if (r1cs) zg_r1cs_free(r1cs);
        if (ctx) zg_free(ctx);
    }
};
std::vector<std::string> input_names(const char* circuit_path, uint64_t n_public) {
    std::vector<std::string> names;
    std::ifstream f(std::string(circuit_path) + ".names");
    std::string line;
    while (f && std::getline(f, line) && names.size() < n_public)
        if (!line.empty()) names.push_back(line);
    for (uint64_t i = names.size(); i < n_public; i++) names.push_back("main.public[" + std::to_string(i) + "]");
    return names;
}
std::string nul_joined(const std::vector<std::string>& v) {
    std::string o;
    for (auto& s : v) {
        o += s;
        o.push_back('\0');
    }
    return o;
}
// A JSON string starting at s[q0] == '"', with its escapes decoded (\uXXXX as UTF-8): a path may contain '"' or '\\'.
bool json_string(const std::string& s, size_t q0, std::string* out) {
    out->clear();
    for (size_t i = q0 + 1; i < s.size(); i++) {
        const char ch = s[i];
        if (ch == '"') return true;
        if (ch != '\\') {
            out->push_back(ch);
            continue;
        }
        if (++i >= s.size()) return false;
        switch (s[i]) {
            case '"': out->push_back('"'); break;
            case '\\': out->push_back('\\'); break;
            case '/': out->push_back('/'); break;
            case 'b': out->push_back('\b'); break;
            case 'f': out->push_back('\f'); break;
            case 'n': out->push_back('\n'); break;
            case 'r': out->push_back('\r'); break;
            case 't': out->push_back('\t'); break;
            case 'u': {
                // as serde_json: a high surrogate must be followed by a \uXXXX low surrogate and the pair is ONE code point
                // (4 bytes of UTF-8); a lone surrogate is an error.  \u0000 is refused too: the string is a path, and an
                // embedded NUL would silently cut it short at fopen.
                auto hex4 = [&](size_t at, unsigned* v) {
                    if (at + 3 >= s.size()) return false;
                    unsigned x = 0;
                    for (int d = 0; d < 4; d++) {
                        const char h = s[at + d];
                        x <<= 4;
                        if (h >= '0' && h <= '9') x |= (unsigned)(h - '0');
                        else if (h >= 'a' && h <= 'f') x |= (unsigned)(h - 'a' + 10);
                        else if (h >= 'A' && h <= 'F') x |= (unsigned)(h - 'A' + 10);
                        else return false;
                    }
                    *v = x;
                    return true;
                };
                unsigned cp = 0;
                if (!hex4(i + 1, &cp)) return false;
                i += 4;
                if (cp >= 0xDC00 && cp <= 0xDFFF) return false;   // a low surrogate on its own
                if (cp >= 0xD800 && cp <= 0xDBFF) {
                    unsigned lo = 0;
                    if (i + 2 >= s.size() || s[i + 1] != '\\' || s[i + 2] != 'u' || !hex4(i + 3, &lo)) return false;
                    if (lo < 0xDC00 || lo > 0xDFFF) return false;
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    i += 6;
                }
                if (cp == 0) return false;
                if (cp < 0x80) out->push_back((char)cp);
                else if (cp < 0x800) {
                    out->push_back((char)(0xC0 | (cp >> 6)));
                    out->push_back((char)(0x80 | (cp & 0x3F)));
                } else if (cp < 0x10000) {
                    out->push_back((char)(0xE0 | (cp >> 12)));
                    out->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                    out->push_back((char)(0x80 | (cp & 0x3F)));
                } else {
                    out->push_back((char)(0xF0 | (cp >> 18)));
                    out->push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
                    out->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                    out->push_back((char)(0x80 | (cp & 0x3F)));
                }
                break;
            }
            default: return false;
        }
    }
    return false;   // no closing quote
}
// {"@binwitness": "<path>"} (a JSON string, escapes honoured) or a bare path
bool witness_path(const char* inputs, std::string* path, std::string* err) {
    std::string s(inputs);
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) {
        *err = "Json(\"EOF while parsing a value\")";
        return false;
    }
    if (s[a] != '{') {
        size_t b = s.find_last_not_of(" \t\r\n");
        *path = s.substr(a, b - a + 1);
        return true;
    }
    const std::string key = "\"@binwitness\"";
    size_t k = s.find(key);
    if (k != std::string::npos) {
        size_t c = s.find(':', k + key.size());
        size_t q0 = c == std::string::npos ? c : s.find_first_not_of(" \t\r\n", c + 1);
        if (q0 != std::string::npos && s[q0] == '"') {
            if (json_string(s, q0, path)) return true;
            *err = "Json(\"invalid string for @binwitness\")";
            return false;
        }
    }
    *err = "Evaluator(\"witness generation needs the za evaluator, which is outside this library: pass "
           "{\\\"@binwitness\\\": <path of the evaluator's .binwitness dump>}\")";
    return false;
}

}  // namespace

extern "C" void verbose(int on) {
    // the reference re-initialises stderrlog here and would panic on a second call (lib.rs:36-46); this is idempotent
    g_verbose = on != 0;
}

static int setup_impl(const char* circuit_path, const char* pk_path, const char* verifier_type, char* verifier_buffer,
                     size_t verifier_buffer_size, char* error_buffer, size_t error_buffer_size) {
    if (!circuit_path || !pk_path || !verifier_type)
        return return_string("Unexpected(\"null argument\")", error_buffer, error_buffer_size, ERR_CUSTOM);
    const bool want_json = !strcmp(verifier_type, "json"), want_sol = !strcmp(verifier_type, "solidity");
    if (!want_json && !want_sol) return return_string("invalid validator type", error_buffer, error_buffer_size, ERR_CUSTOM);
    std::string err;
    std::vector<uint8_t> circuit;
    if (!read_file(circuit_path, &circuit, &err)) return return_string(err, error_buffer, error_buffer_size, ERR_CUSTOM);
    Handles h;
    int rc = zg_init(0, &h.ctx);
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    double t0 = now_s();
    rc = zg_pk_load(h.ctx, circuit.data(), circuit.size(), 0, 0, 1, &h.r1cs, &h.old_params);
    if (rc) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    zg_params_free(h.old_params);   // only the constraints are kept: the key is made afresh
    h.old_params = nullptr;
    size_t header = 0;
    rc = zg_pk_params_offset(circuit.data(), circuit.size(), &header);
    if (rc) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    info("Compilation time: %.3fs (constraints read from the container)\n", now_s() - t0);
    t0 = now_s();
    rc = zg_setup_random(h.ctx, h.r1cs, &h.params, nullptr, 0, nullptr);   // the key is made and stays resident in HBM
    if (rc) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    size_t need = 0;
    rc = zg_params_write(h.ctx, h.params, nullptr, 0, &need);
    if (rc != ZG_E_BUFFER_TOO_SMALL) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    std::vector<uint8_t> pbytes(need);
    rc = zg_params_write(h.ctx, h.params, pbytes.data(), pbytes.size(), &need);
    if (rc) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    info("Setup time: %.3fs\n", now_s() - t0);
    t0 = now_s();
    {
        std::ofstream f(pk_path, std::ios::binary);
        if (f) f.write((const char*)circuit.data(), (std::streamsize)header);
        if (f) f.write((const char*)pbytes.data(), (std::streamsize)need);
        if (!f) return return_string(std::string("Io(\"") + pk_path + ": " + strerror(errno) + "\")", error_buffer, error_buffer_size, ERR_CUSTOM);
    }
    info("Proving key write time: %.3fs\n", now_s() - t0);
    uint64_t inf[7];
    zg_r1cs_info(h.r1cs, inf);
    const std::vector<std::string> names = input_names(circuit_path, inf[0] - 1);
    const std::string joined = nul_joined(names);
    std::string verifier;
    if (want_sol) {
        size_t n = 0;
        rc = zg_verifier_solidity(pbytes.data(), pbytes.size(), joined.data(), names.size(), nullptr, 0, &n);
        if (rc == ZG_E_BUFFER_TOO_SMALL) {
            verifier.resize(n);
            rc = zg_verifier_solidity(pbytes.data(), pbytes.size(), joined.data(), names.size(), &verifier[0], n, &n);
            verifier.resize(strlen(verifier.c_str()));
        }
    } else {
        verifier.resize(8192 + 400 * (size_t)inf[0]);
        rc = zg_vk_json(pbytes.data(), pbytes.size(), joined.data(), names.size(), &verifier[0], verifier.size());
        verifier.resize(strlen(verifier.c_str()));
    }
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    return return_string(verifier, verifier_buffer, verifier_buffer_size, ERR_NONE);
}

static int prove_impl(const char* pk_path, const char* inputs, char* proof_buffer, size_t proof_buffer_size,
                     char* error_buffer, size_t error_buffer_size) {
    if (!pk_path || !inputs) return return_string("Unexpected(\"null argument\")", error_buffer, error_buffer_size, ERR_CUSTOM);
    std::string err, wpath;
    if (!witness_path(inputs, &wpath, &err)) return return_string(err, error_buffer, error_buffer_size, ERR_CUSTOM);
    const double t_all = now_s();
    FileBytes pk, bw;
    if (!pk.open(pk_path, &err) || !bw.open(wpath.c_str(), &err))
        return return_string(err, error_buffer, error_buffer_size, ERR_CUSTOM);
    Handles h;
    double t0 = now_s();
    int rc = zg_init(0, &h.ctx);
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    info("GPU context time: %.3fs\n", now_s() - t0);
    zg_set_option(h.ctx, ZG_OPT_ONE_SHOT, 1);   // read the key, prove once, return: no per-key tables
    t0 = now_s();
    rc = zg_pk_load(h.ctx, pk.p, pk.n, /*checked=*/1, 0, 1, &h.r1cs, &h.params);   // read_pk: Parameters::read(pk, true)
    if (rc) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    info("Proving key read time: %.3fs\n", now_s() - t0);
    if (g_verbose) {
        double lt[4] = {0, 0, 0, 0};
        zg_last_load_timings(h.ctx, lt);
        fprintf(stderr, "  verifying key + container walk %.3fs, queries checked on the GPU %.3fs, constraints %.3fs, R1CS to the GPU %.3fs\n",
                lt[0] / 1e3, lt[1] / 1e3, lt[2] / 1e3, lt[3] / 1e3);
    }
    uint64_t inf[7];
    zg_r1cs_info(h.r1cs, inf);
    const size_t nv = (size_t)(inf[0] + inf[1]);
    t0 = now_s();
    std::unique_ptr<uint8_t[]> wbuf(new uint8_t[nv * 32]);   // (not zeroed here: zg_witness_from_binwitness writes all of it)
    uint8_t* const w = wbuf.get();
    rc = zg_witness_from_binwitness(h.ctx, h.r1cs, bw.p, bw.n, w, nv * 32);
    if (rc) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    info("Witness read time: %.3fs\n", now_s() - t0);
    t0 = now_s();
    uint64_t bad = 0;
    rc = zg_r1cs_check(h.ctx, h.r1cs, w, &bad);
    if (rc) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    info("Constraint check time: %.3fs\n", now_s() - t0);
    if (bad != UINT64_MAX)   // the reference panics here: expect("check_constrains_eval_zero failed"), prover.rs:155-157
        return return_string("Unexpected(\"check_constrains_eval_zero failed: constraint " + std::to_string(bad) + "\")",
                             error_buffer, error_buffer_size, ERR_CUSTOM);
    t0 = now_s();
    uint8_t proof[256];
    rc = zg_prove_random(h.ctx, h.params, h.r1cs, w, proof);
    if (rc) return return_string(err_text(h.ctx, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    info("Proof generation time: %.3fs\n", now_s() - t0);
    t0 = now_s();
    // the verifying key from the resident key's host copy (walking the container's 4 M length prefixes again to find the
    // Parameters body cost 0.11 s at 2^22: most of this step)
    size_t vk_len = 0;
    zg_params_write_vk(h.params, nullptr, 0, &vk_len);
    std::vector<uint8_t> vkb(vk_len);
    rc = zg_params_write_vk(h.params, vkb.data(), vkb.size(), &vk_len);
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    const size_t n_public = (size_t)inf[0] - 1;
    int ok = 0;
    rc = zg_verify(vkb.data(), vkb.size(), proof, w + 32, n_public, &ok);
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    if (!ok) return return_string("Unexpected(\"the proof does not verify against its own key\")", error_buffer, error_buffer_size, ERR_CUSTOM);
    std::string json(1024 + 90 * n_public, '\0');
    rc = zg_proof_json(proof, w + 32, n_public, &json[0], json.size());
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    json.resize(strlen(json.c_str()));
    info("Proof verification time: %.3fs\n", now_s() - t0);
    info("prove() time: %.3fs\n", now_s() - t_all);
    return return_string(json, proof_buffer, proof_buffer_size, ERR_NONE);
}

static int verify_impl(const char* verifying_key, const char* proof_with_inputs, char* error_buffer, size_t error_buffer_size) {
    if (!verifying_key || !proof_with_inputs)
        return return_string("Unexpected(\"null argument\")", error_buffer, error_buffer_size, ERR_CUSTOM);
    size_t vk_len = 0;
    int rc = zg_vk_from_json(verifying_key, nullptr, 0, &vk_len);
    if (rc && rc != ZG_E_BUFFER_TOO_SMALL) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    std::vector<uint8_t> vk(vk_len);
    rc = zg_vk_from_json(verifying_key, vk.data(), vk.size(), &vk_len);
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    uint8_t proof[256];
    size_t n_public = 0;
    std::vector<uint8_t> pub(32 * 16);
    rc = zg_proof_from_json(proof_with_inputs, proof, pub.data(), pub.size(), &n_public);
    if (rc == ZG_E_BUFFER_TOO_SMALL) {
        pub.resize(32 * n_public);
        rc = zg_proof_from_json(proof_with_inputs, proof, pub.data(), pub.size(), &n_public);
    }
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    int ok = 0;
    rc = zg_verify(vk.data(), vk.size(), proof, pub.data(), n_public, &ok);
    if (rc) return return_string(err_text(nullptr, rc), error_buffer, error_buffer_size, ERR_CUSTOM);
    return ok ? ERR_NONE : ERR_VERIFICATION_FAILED;
}

// No C++ exception leaves the four symbols (std::bad_alloc for a huge key, std::length_error ...): the C / Go / Rust
// host gets code 100 with an Unexpected(..) text instead of std::terminate.
template <class Fn>
static int guarded(char* error_buffer, size_t error_buffer_size, Fn&& fn) {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        return return_string("Unexpected(\"out of memory\")", error_buffer, error_buffer_size, ERR_CUSTOM);
    } catch (const std::exception& e) {
        return return_string(std::string("Unexpected(\"") + e.what() + "\")", error_buffer, error_buffer_size, ERR_CUSTOM);
    } catch (...) {
        return return_string("Unexpected(\"unknown exception\")", error_buffer, error_buffer_size, ERR_CUSTOM);
    }
}
extern "C" int setup(const char* circuit_path, const char* pk_path, const char* verifier_type, char* verifier_buffer,
                     size_t verifier_buffer_size, char* error_buffer, size_t error_buffer_size) {
    return guarded(error_buffer, error_buffer_size, [&] {
        return setup_impl(circuit_path, pk_path, verifier_type, verifier_buffer, verifier_buffer_size, error_buffer, error_buffer_size);
    });
}
extern "C" int prove(const char* pk_path, const char* inputs, char* proof_buffer, size_t proof_buffer_size,
                     char* error_buffer, size_t error_buffer_size) {
    return guarded(error_buffer, error_buffer_size,
                   [&] { return prove_impl(pk_path, inputs, proof_buffer, proof_buffer_size, error_buffer, error_buffer_size); });
}
extern "C" int verify(const char* verifying_key, const char* proof_with_inputs, char* error_buffer, size_t error_buffer_size) {
    return guarded(error_buffer, error_buffer_size,
                   [&] { return verify_impl(verifying_key, proof_with_inputs, error_buffer, error_buffer_size); });
}
