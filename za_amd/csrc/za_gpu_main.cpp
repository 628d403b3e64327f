// za-gpu -- command-line driver over libza2c.so with the sub-commands, flags and default file names of the `za`
// CLI's setup / prove arms (cli/src/main.rs:25-33 defaults, :120-154 flags, :196-226 and :249-269 behaviour), plus a
// `verify` arm for the binding-level verify (binding/c/native/src/lib.rs:101-117).  `compile` and `test` belong to the
// za compiler and are not provided.  Deviations are those of libza2c (see za2c.cpp): --circuit names a compiled
// constraint container, and prove takes the evaluator's witness dump through --witness (or {"@binwitness": ..} inside
// the --input file).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

extern "C" {
void verbose(int);
int setup(const char*, const char*, const char*, char*, size_t, char*, size_t);
int prove(const char*, const char*, char*, size_t, char*, size_t);
int verify(const char*, const char*, char*, size_t);
}

static const char* DEFAULT_CIRCUIT = "circuit.za";
static const char* DEFAULT_PROVING_KEY = "proving.key";
static const char* DEFAULT_INPUT = "input.json";
static const char* DEFAULT_PROOF = "proof.json";
static const char* DEFAULT_VERIFIER_SOLIDITY = "verifier.sol";
static const char* DEFAULT_VERIFIER_JSON = "verifier.json";

static bool slurp(const std::string& path, std::string* out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    *out = ss.str();
    return true;
}
static bool spit(const std::string& path, const char* text) {
    std::ofstream f(path, std::ios::binary);
    f << text;
    return (bool)f;
}
static int usage() {
    fprintf(stderr,
            "za-gpu <setup|prove|verify> [flags]\n"
            "  setup  --circuit <compiled circuit, default circuit.za> --pk <proving.key> --verifier <file>\n"
            "         --verifiertype <solidity|json> (default solidity)\n"
            "  prove  --pk <proving.key> --input <input.json> --proof <proof.json> [--witness <file.binwitness>]\n"
            "  verify --vk <verifier.json> --proof <proof.json>\n");
    return 2;
}

int main(int argc, char** argv) {
    if (argc < 2) return usage();
    const std::string cmd = argv[1];
    std::string circuit, pk, verifier_file, verifier_type, input, proof, witness, vk;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](std::string* dst) {
            if (i + 1 >= argc) {
                fprintf(stderr, "%s needs a value\n", a.c_str());
                exit(2);
            }
            *dst = argv[++i];
        };
        if (a == "--circuit") val(&circuit);
        else if (a == "--pk") val(&pk);
        else if (a == "--verifier") val(&verifier_file);
        else if (a == "--verifiertype") val(&verifier_type);
        else if (a == "--input") val(&input);
        else if (a == "--proof") val(&proof);
        else if (a == "--witness") val(&witness);
        else if (a == "--vk") val(&vk);
        else return usage();
    }
    verbose(1);   // the za CLI always logs (cli/src/main.rs:182-188)
    const size_t ERR = 1 << 16;
    std::vector<char> err(ERR);
    if (cmd == "setup") {
        if (circuit.empty()) circuit = DEFAULT_CIRCUIT;
        if (pk.empty()) pk = DEFAULT_PROVING_KEY;
        if (verifier_type.empty()) verifier_type = "solidity";
        if (verifier_type != "solidity" && verifier_type != "json") {
            fprintf(stderr, "unknown verifier type\n");   // the reference panics here (main.rs:210)
            return 101;
        }
        if (verifier_file.empty()) verifier_file = verifier_type == "solidity" ? DEFAULT_VERIFIER_SOLIDITY : DEFAULT_VERIFIER_JSON;
        const size_t cap = (size_t)256 << 20;
        char* out = (char*)malloc(cap);
        if (!out) return 101;
        int rc = setup(circuit.c_str(), pk.c_str(), verifier_type.c_str(), out, cap, err.data(), ERR);
        if (rc) {
            fprintf(stderr, "unable to create proof: %s (code %d)\n", err.data(), rc);   // .expect() text of main.rs:220
            return 101;
        }
        if (!spit(verifier_file, out)) {
            fprintf(stderr, "cannot write verifier file\n");
            return 101;
        }
        free(out);
        return 0;
    }
    if (cmd == "prove") {
        if (pk.empty()) pk = DEFAULT_PROVING_KEY;
        if (input.empty()) input = DEFAULT_INPUT;
        if (proof.empty()) proof = DEFAULT_PROOF;
        std::string inputs_json;
        if (!witness.empty()) inputs_json = witness;   // prove() takes a bare path as it is: no JSON quoting to get wrong
        else if (!slurp(input, &inputs_json)) {
            fprintf(stderr, "cannot open inputs file\n");
            return 101;
        }
        const size_t cap = (size_t)64 << 20;
        char* out = (char*)malloc(cap);
        if (!out) return 101;
        int rc = prove(pk.c_str(), inputs_json.c_str(), out, cap, err.data(), ERR);
        if (rc) {
            fprintf(stderr, "cannot generate proof: %s (code %d)\n", err.data(), rc);
            return 101;
        }
        if (!spit(proof, out)) {
            fprintf(stderr, "cannot write proof file\n");
            return 101;
        }
        free(out);
        return 0;
    }
    if (cmd == "verify") {
        if (vk.empty()) vk = DEFAULT_VERIFIER_JSON;
        if (proof.empty()) proof = DEFAULT_PROOF;
        std::string vk_json, proof_json;
        if (!slurp(vk, &vk_json) || !slurp(proof, &proof_json)) {
            fprintf(stderr, "cannot open the verifying key / proof file\n");
            return 101;
        }
        int rc = verify(vk_json.c_str(), proof_json.c_str(), err.data(), ERR);
        if (rc == 0) {
            printf("proof is valid\n");
            return 0;
        }
        if (rc == 2) {
            printf("proof is NOT valid\n");
            return 1;
        }
        fprintf(stderr, "verify failed: %s (code %d)\n", err.data(), rc);
        return 101;
    }
    return usage();
}
