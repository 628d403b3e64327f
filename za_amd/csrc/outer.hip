// Text formats on the outer side of the boundary: the JSON readers for the verifying key and for
// proof + public inputs, and the Solidity verifier export.  Host code only (no kernels).
//
// Replaces, on the za side,
//   JsonVerifyingKey::{from_json, to_bellman}        prover/src/groth16/format.rs:169-194
//   JsonProofAndInput::json_to_bellman               prover/src/groth16/format.rs:101-121
//   str_to_fq -> FS::parse (decimal or 0x hex)       prover/src/groth16/format.rs:33-36, compiler/src/algebra/fs.rs:43-55
//   generate_solidity                                prover/src/groth16/ethereum.rs:216-261
// so that za2c's `verify(vk_json, proof_json, ...)` and `setup(..., "solidity" | "json", ...)`
// (binding/c/native/src/lib.rs:50-117) can be served by this library.
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "formats.h"

namespace zg {
namespace {

// ---------------------------------------------------------------------------------------------- a small JSON reader
// Enough of RFC 8259 for serde_json::from_str on the two structs above: objects, arrays, strings (with escapes),
// numbers (kept as text), true / false / null.  Unknown object keys are ignored and a duplicate key is an error,
// as serde's derive does for structs.
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    std::string s;                                   // STR: decoded text, NUM: the literal
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* key) const {
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* p;
    const char* end;
    const char* begin;
    std::string err;
    bool fail_at(const char* what) {
        if (err.empty()) err = std::string(what) + " at offset " + std::to_string((size_t)(p - begin));
        return false;
    }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    static void utf8(std::string* o, uint32_t cp) {
        if (cp < 0x80) o->push_back((char)cp);
        else if (cp < 0x800) { o->push_back((char)(0xC0 | (cp >> 6))); o->push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            o->push_back((char)(0xE0 | (cp >> 12))); o->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            o->push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            o->push_back((char)(0xF0 | (cp >> 18))); o->push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            o->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o->push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    bool hex4(uint32_t* v) {
        if (end - p < 4) return fail_at("EOF while parsing a string");
        uint32_t x = 0;
        for (int i = 0; i < 4; i++) {
            char c = *p++;
            x <<= 4;
            if (c >= '0' && c <= '9') x |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') x |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') x |= (uint32_t)(c - 'A' + 10);
            else return fail_at("invalid escape");
        }
        *v = x;
        return true;
    }
    bool string(std::string* out) {
        if (p >= end || *p != '"') return fail_at("expected a string");
        p++;
        for (;;) {
            if (p >= end) return fail_at("EOF while parsing a string");
            unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c < 0x20) return fail_at("control character in a string");
            if (c != '\\') { out->push_back((char)c); continue; }
            if (p >= end) return fail_at("EOF while parsing a string");
            char e = *p++;
            switch (e) {
                case '"': out->push_back('"'); break;
                case '\\': out->push_back('\\'); break;
                case '/': out->push_back('/'); break;
                case 'b': out->push_back('\b'); break;
                case 'f': out->push_back('\f'); break;
                case 'n': out->push_back('\n'); break;
                case 'r': out->push_back('\r'); break;
                case 't': out->push_back('\t'); break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(&cp)) return false;
                    if (cp >= 0xD800 && cp < 0xDC00) {   // surrogate pair
                        uint32_t lo;
                        if (end - p < 2 || p[0] != '\\' || p[1] != 'u') return fail_at("lone leading surrogate");
                        p += 2;
                        if (!hex4(&lo)) return false;
                        if (lo < 0xDC00 || lo >= 0xE000) return fail_at("lone leading surrogate");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp < 0xE000) {
                        return fail_at("lone trailing surrogate");
                    }
                    utf8(out, cp);
                    break;
                }
                default: return fail_at("invalid escape");
            }
        }
    }
    bool value(JVal* v, int depth) {
        if (depth > 128) return fail_at("recursion limit exceeded");
        ws();
        if (p >= end) return fail_at("EOF while parsing a value");
        char c = *p;
        if (c == '{') {
            v->kind = JVal::OBJ;
            p++;
            ws();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!string(&key)) return fail_at("key must be a string");
                ws();
                if (p >= end || *p != ':') return fail_at("expected `:`");
                p++;
                JVal child;
                if (!value(&child, depth + 1)) return false;
                v->obj.emplace_back(std::move(key), std::move(child));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; return true; }
                return fail_at("expected `,` or `}`");
            }
        }
        if (c == '[') {
            v->kind = JVal::ARR;
            p++;
            ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                JVal child;
                if (!value(&child, depth + 1)) return false;
                v->arr.push_back(std::move(child));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail_at("expected `,` or `]`");
            }
        }
        if (c == '"') {
            v->kind = JVal::STR;
            return string(&v->s);
        }
        if (c == '-' || (c >= '0' && c <= '9')) {
            v->kind = JVal::NUM;
            const char* s0 = p;
            if (*p == '-') p++;
            if (p >= end || *p < '0' || *p > '9') return fail_at("invalid number");
            if (*p == '0') p++;
            else while (p < end && *p >= '0' && *p <= '9') p++;
            if (p < end && *p == '.') {
                p++;
                if (p >= end || *p < '0' || *p > '9') return fail_at("invalid number");
                while (p < end && *p >= '0' && *p <= '9') p++;
            }
            if (p < end && (*p == 'e' || *p == 'E')) {
                p++;
                if (p < end && (*p == '+' || *p == '-')) p++;
                if (p >= end || *p < '0' || *p > '9') return fail_at("invalid number");
                while (p < end && *p >= '0' && *p <= '9') p++;
            }
            v->s.assign(s0, p);
            return true;
        }
        auto lit = [&](const char* w, JVal::Kind k) {
            size_t n = strlen(w);
            if ((size_t)(end - p) < n || memcmp(p, w, n) != 0) return false;
            p += n;
            v->kind = k;
            return true;
        };
        if (c == 't') { if (lit("true", JVal::BOOL)) { v->s = "true"; return true; } }
        else if (c == 'f') { if (lit("false", JVal::BOOL)) { v->s = "false"; return true; } }
        else if (c == 'n') { if (lit("null", JVal::NUL)) return true; }
        return fail_at("expected value");
    }
};

// Error::Json(..) / Error::BadFormat(..) / Error::Algebra(..) texts in the shape of Rust's Debug output, which is
// what za2c copies into the caller's error buffer (binding/c/native/src/lib.rs:97,114)
int bad(int code, const std::string& msg) { return fail(nullptr, code, "%s", msg.c_str()); }

int parse_json(const char* text, JVal* root) {
    JParser ps{text, text + strlen(text), text, {}};
    if (!ps.value(root, 0)) return bad(ZG_E_BAD_FORMAT, "Json(\"" + ps.err + "\")");
    ps.ws();
    if (ps.p != ps.end) return bad(ZG_E_BAD_FORMAT, "Json(\"trailing characters at offset " + std::to_string((size_t)(ps.p - text)) + "\")");
    return ZG_OK;
}

// serde's struct visitor: every named field must be present exactly once; unknown fields are skipped
int field(const JVal& obj, const char* name, const JVal** out) {
    const JVal* found = nullptr;
    for (auto& kv : obj.obj)
        if (kv.first == name) {
            if (found) return bad(ZG_E_BAD_FORMAT, std::string("Json(\"duplicate field `") + name + "`\")");
            found = &kv.second;
        }
    if (!found) return bad(ZG_E_BAD_FORMAT, std::string("Json(\"missing field `") + name + "`\")");
    *out = found;
    return ZG_OK;
}

// FS::parse (compiler/src/algebra/fs.rs:43-55): "0x" + hex digits, otherwise decimal digits; the value itself is
// not reduced.  str_to_fq then feeds its decimal form to Fq::from_str, which only exists below q (the reference
// unwrap()s there; here it is a BadFormat).  256-bit accumulator, overflow = "not in the field".
int parse_uint256(const std::string& s, uint32_t out[8], bool allow_hex) {
    for (int i = 0; i < 8; i++) out[i] = 0;
    const bool hex = allow_hex && s.size() >= 2 && s[0] == '0' && s[1] == 'x';
    const size_t from = hex ? 2 : 0;
    if (s.size() == from)
        return bad(ZG_E_BAD_FORMAT, "Algebra(InvalidFormat(\"" + s + (hex ? " is not hexadecimal\"))" : " is not decimal\"))"));
    for (size_t i = from; i < s.size(); i++) {
        const char c = s[i];
        uint32_t d;
        if (c >= '0' && c <= '9') d = (uint32_t)(c - '0');
        else if (hex && c >= 'a' && c <= 'f') d = (uint32_t)(c - 'a' + 10);
        else if (hex && c >= 'A' && c <= 'F') d = (uint32_t)(c - 'A' + 10);
        else return bad(ZG_E_BAD_FORMAT, "Algebra(InvalidFormat(\"" + s + (hex ? " is not hexadecimal\"))" : " is not decimal\"))"));
        uint64_t carry = d;
        const uint32_t mul = hex ? 16u : 10u;
        for (int k = 0; k < 8; k++) {
            uint64_t t = (uint64_t)out[k] * mul + carry;
            out[k] = (uint32_t)t;
            carry = t >> 32;
        }
        if (carry) return bad(ZG_E_BAD_FORMAT, "BadFormat(\"" + s + " is not a field element\")");
    }
    return ZG_OK;
}

int json_fq(const JVal& v, Fq* out) {
    if (v.kind != JVal::STR) return bad(ZG_E_BAD_FORMAT, "Json(\"invalid type: expected a string\")");
    uint32_t w[8];
    ZG_TRY(parse_uint256(v.s, w, true));
    bool lt = false;
    for (int i = 7; i >= 0; i--) {
        if (w[i] < FqParams::MOD[i]) { lt = true; break; }
        if (w[i] > FqParams::MOD[i]) break;
    }
    if (!lt) return bad(ZG_E_BAD_FORMAT, "BadFormat(\"" + v.s + " is not a field element\")");
    Fq c;
    for (int i = 0; i < 8; i++) c.l[i] = w[i];
    *out = c.to_mont();
    return ZG_OK;
}
int json_arr(const JVal& v, size_t n, const char* what) {
    if (v.kind != JVal::ARR) return bad(ZG_E_BAD_FORMAT, std::string("Json(\"invalid type: expected ") + what + "\")");
    if (v.arr.size() != n)
        return bad(ZG_E_BAD_FORMAT, "Json(\"invalid length " + std::to_string(v.arr.size()) + ", expected " + what + "\")");
    return ZG_OK;
}
std::string fq_dec_pair(const Fq& x, const Fq& y);

// g1_jstruct_to_bellman (format.rs:38-44): try_from_coordinates = on-curve test
int json_g1(const JVal& v, G1Affine* out) {
    ZG_TRY(json_arr(v, 2, "an array of length 2"));
    G1Affine p;
    ZG_TRY(json_fq(v.arr[0], &p.x));
    ZG_TRY(json_fq(v.arr[1], &p.y));
    if (p.is_inf() || !on_curve(p, Fq::from_u64(3))) return bad(ZG_E_BAD_FORMAT, "BadFormat(\"bad coordinates (" + fq_dec_pair(p.x, p.y) + ")\")");
    *out = p;
    return ZG_OK;
}
// g2_jstruct_to_bellman (format.rs:54-67): [[x.c0, x.c1], [y.c0, y.c1]]
int json_g2(const JVal& v, G2Affine* out) {
    ZG_TRY(json_arr(v, 2, "an array of length 2"));
    ZG_TRY(json_arr(v.arr[0], 2, "an array of length 2"));
    ZG_TRY(json_arr(v.arr[1], 2, "an array of length 2"));
    G2Affine p;
    ZG_TRY(json_fq(v.arr[0].arr[0], &p.x.c0));
    ZG_TRY(json_fq(v.arr[0].arr[1], &p.x.c1));
    ZG_TRY(json_fq(v.arr[1].arr[0], &p.y.c0));
    ZG_TRY(json_fq(v.arr[1].arr[1], &p.y.c1));
    if (p.is_inf() || !on_curve(p, g2_coeff_b()) || !g2_in_subgroup(p))
        return bad(ZG_E_BAD_FORMAT, "BadFormat(\"bad coordinates (G2 point is not on the curve / in the subgroup)\")");
    *out = p;
    return ZG_OK;
}

std::string fq_hex0x(const Fq& m) {   // FqRepr's Display: 0x + 64 lowercase hex digits
    uint8_t be[32];
    fq_to_be(be, m);
    static const char* d = "0123456789abcdef";
    std::string s = "0x";
    for (int i = 0; i < 32; i++) { s.push_back(d[be[i] >> 4]); s.push_back(d[be[i] & 15]); }
    return s;
}
std::string fq_dec_pair(const Fq& x, const Fq& y) { return fq_hex0x(x) + "," + fq_hex0x(y); }

void replace_all(std::string* s, const std::string& from, const std::string& to) {
    size_t at = 0;
    while ((at = s->find(from, at)) != std::string::npos) {
        s->replace(at, from.size(), to);
        at += to.size();
    }
}

// Rust's {:?} of a String: quotes, backslash escapes, \u{..} for control characters
std::string rust_debug_str(const char* p) {
    std::string o = "\"";
    for (; *p; p++) {
        const unsigned char c = (unsigned char)*p;
        if (c == '"') o += "\\\"";
        else if (c == '\\') o += "\\\\";
        else if (c == '\n') o += "\\n";
        else if (c == '\r') o += "\\r";
        else if (c == '\t') o += "\\t";
        else if (c < 0x20 || c == 0x7f) {
            char buf[16];
            snprintf(buf, sizeof(buf), "\\u{%x}", c);
            o += buf;
        } else o.push_back((char)c);
    }
    o += "\"";
    return o;
}

const char SOL_TEMPLATE[] =
#include "verifier_sol.inc"
    ;

}  // namespace
}  // namespace zg

using namespace zg;

// JsonVerifyingKey::from_json + to_bellman (format.rs:169-194), output = VerifyingKey::write bytes (what zg_verify
// and zg_vk_json take).  `input_names` must be present (serde has no default for it) and is otherwise ignored.
extern "C" int zg_vk_from_json(const char* json, uint8_t* vk_out, size_t cap, size_t* written) try {
    if (!json || !written) return ZG_E_INVALID_ARG;
    JVal root;
    ZG_TRY(parse_json(json, &root));
    if (root.kind != JVal::OBJ) return bad(ZG_E_BAD_FORMAT, "Json(\"invalid type: expected struct JsonVerifyingKey\")");
    ParsedParams pp;
    const JVal* f;
    ZG_TRY(field(root, "alpha_g1", &f)); const JVal* f_alpha = f;
    ZG_TRY(field(root, "beta_g1", &f)); const JVal* f_beta1 = f;
    ZG_TRY(field(root, "beta_g2", &f)); const JVal* f_beta2 = f;
    ZG_TRY(field(root, "delta_g1", &f)); const JVal* f_delta1 = f;
    ZG_TRY(field(root, "delta_g2", &f)); const JVal* f_delta2 = f;
    ZG_TRY(field(root, "gamma_g2", &f)); const JVal* f_gamma2 = f;
    ZG_TRY(field(root, "ic", &f)); const JVal* f_ic = f;
    ZG_TRY(field(root, "input_names", &f));
    if (f->kind != JVal::ARR) return bad(ZG_E_BAD_FORMAT, "Json(\"invalid type: expected a sequence\")");
    for (auto& nm : f->arr)
        if (nm.kind != JVal::STR) return bad(ZG_E_BAD_FORMAT, "Json(\"invalid type: expected a string\")");
    if (f_ic->kind != JVal::ARR) return bad(ZG_E_BAD_FORMAT, "Json(\"invalid type: expected a sequence\")");
    // to_bellman converts ic first, then the named points in declaration order (format.rs:169-185)
    pp.ic.resize(f_ic->arr.size());
    for (size_t i = 0; i < f_ic->arr.size(); i++) ZG_TRY(json_g1(f_ic->arr[i], &pp.ic[i]));
    ZG_TRY(json_g1(*f_alpha, &pp.alpha_g1));
    ZG_TRY(json_g1(*f_beta1, &pp.beta_g1));
    ZG_TRY(json_g2(*f_beta2, &pp.beta_g2));
    ZG_TRY(json_g1(*f_delta1, &pp.delta_g1));
    ZG_TRY(json_g2(*f_delta2, &pp.delta_g2));
    ZG_TRY(json_g2(*f_gamma2, &pp.gamma_g2));
    std::vector<uint8_t> bytes;
    write_bellman_vk(pp, &bytes);
    *written = bytes.size();
    if (!vk_out || cap < bytes.size()) return ZG_E_BUFFER_TOO_SMALL;
    memcpy(vk_out, bytes.data(), bytes.size());
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

// JsonProofAndInput::json_to_bellman (format.rs:101-121).  Coordinates: decimal or 0x hex (str_to_fq); public
// inputs: what Fr::from_str accepts -- canonical decimal below r, no sign, no leading zero -- else BadFormat("bad format").
extern "C" int zg_proof_from_json(const char* json, uint8_t proof_out[256], uint8_t* public_inputs32, size_t cap_inputs,
                                  size_t* n_public) try {
    if (!json || !proof_out || !n_public) return ZG_E_INVALID_ARG;
    JVal root;
    ZG_TRY(parse_json(json, &root));
    if (root.kind != JVal::OBJ) return bad(ZG_E_BAD_FORMAT, "Json(\"invalid type: expected struct JsonProofAndInput\")");
    const JVal *fa, *fb, *fc, *fp;
    ZG_TRY(field(root, "a", &fa));
    ZG_TRY(field(root, "b", &fb));
    ZG_TRY(field(root, "c", &fc));
    ZG_TRY(field(root, "public_inputs", &fp));
    if (fp->kind != JVal::ARR) return bad(ZG_E_BAD_FORMAT, "Json(\"invalid type: expected a sequence\")");
    for (auto& s : fp->arr)
        if (s.kind != JVal::STR) return bad(ZG_E_BAD_FORMAT, "Json(\"invalid type: expected a string\")");
    G1Affine A, Cc;
    G2Affine B;
    ZG_TRY(json_g1(*fa, &A));
    ZG_TRY(json_g2(*fb, &B));
    ZG_TRY(json_g1(*fc, &Cc));
    fq_to_be(proof_out, A.x); fq_to_be(proof_out + 32, A.y);
    fq_to_be(proof_out + 64, B.x.c0); fq_to_be(proof_out + 96, B.x.c1);
    fq_to_be(proof_out + 128, B.y.c0); fq_to_be(proof_out + 160, B.y.c1);
    fq_to_be(proof_out + 192, Cc.x); fq_to_be(proof_out + 224, Cc.y);
    *n_public = fp->arr.size();
    if (fp->arr.size() && (!public_inputs32 || cap_inputs < 32 * fp->arr.size())) return ZG_E_BUFFER_TOO_SMALL;
    for (size_t i = 0; i < fp->arr.size(); i++) {
        const std::string& s = fp->arr[i].s;
        uint32_t w[8];
        // Fr::from_str: empty -> None; "0" -> zero; a leading '0' otherwise -> None; any non-digit -> None; >= r -> None
        bool ok = !s.empty() && (s == "0" || s[0] != '0');
        for (char ch : s) ok = ok && ch >= '0' && ch <= '9';
        if (ok) {
            if (parse_uint256(s, w, false) != ZG_OK) ok = false;
            else ok = scalar_is_canonical(w);
        }
        if (!ok) return bad(ZG_E_BAD_FORMAT, "BadFormat(\"bad format\")");
        memcpy(public_inputs32 + 32 * i, w, 32);
    }
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}

// generate_solidity (ethereum.rs:216-261): the template's <%...%> fields filled from the verifying key; G2 points are
// printed c1 first (:231-236); `names` = the public input names (n_names NUL-terminated strings back to back) printed
// as Rust prints a Vec<String> with {:?}.
extern "C" int zg_verifier_solidity(const uint8_t* vk_bytes, size_t vk_len, const char* names, size_t n_names, char* out,
                                    size_t cap, size_t* written) try {
    if (!vk_bytes || !written) return ZG_E_INVALID_ARG;
    ParsedParams pp;
    ZG_TRY(parse_bellman_vk(nullptr, vk_bytes, vk_len, &pp));
    auto g1s = [&](const G1Affine& p) { return fq_hex0x(p.x) + "," + fq_hex0x(p.y); };
    auto g2s = [&](const G2Affine& p) {
        return "[" + fq_hex0x(p.x.c1) + "," + fq_hex0x(p.x.c0) + "],[" + fq_hex0x(p.y.c1) + "," + fq_hex0x(p.y.c0) + "]";
    };
    // try_to_coordinates().expect("non-infinite point expected")
    if (pp.alpha_g1.is_inf() || pp.beta_g2.is_inf() || pp.gamma_g2.is_inf() || pp.delta_g2.is_inf())
        return bad(ZG_E_BAD_FORMAT, "BadFormat(\"non-infinite point expected\")");
    for (auto& p : pp.ic)
        if (p.is_inf()) return bad(ZG_E_BAD_FORMAT, "BadFormat(\"non-infinite point expected\")");
    std::string names_dbg = "[";
    {
        const char* p = names;
        for (size_t i = 0; i < n_names && p; i++) {
            if (i) names_dbg += ", ";
            names_dbg += rust_debug_str(p);
            p += strlen(p) + 1;
        }
        names_dbg += "]";
    }
    std::string pts;
    for (size_t i = 0; i < pp.ic.size(); i++) {
        if (i) pts += "\n";
        pts += "vk.gammaABC[" + std::to_string(i) + "] = Pairing.G1Point(" + g1s(pp.ic[i]) + ");";
    }
    std::string c = SOL_TEMPLATE;
    // same replacement order as the reference: a replaced value never contains another placeholder
    replace_all(&c, "<%vk_a%>", g1s(pp.alpha_g1));
    replace_all(&c, "<%vk_b%>", g2s(pp.beta_g2));
    replace_all(&c, "<%vk_gamma%>", g2s(pp.gamma_g2));
    replace_all(&c, "<%vk_delta%>", g2s(pp.delta_g2));
    replace_all(&c, "<%vk_inputs_length%>", std::to_string(n_names));
    replace_all(&c, "<%vk_inputs%>", names_dbg);
    replace_all(&c, "<%vk_gammaABC_length%>", std::to_string(pp.ic.size()));
    replace_all(&c, "<%vk_gammaABC_pts%>", pts);
    *written = c.size() + 1;
    if (!out || cap < c.size() + 1) return ZG_E_BUFFER_TOO_SMALL;
    memcpy(out, c.c_str(), c.size() + 1);
    return ZG_OK;
} catch (...) {
    return zg::on_exception(nullptr);
}
