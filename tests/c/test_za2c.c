/* The outer boundary called the way a C / cgo host calls za2c (binding/go/lib.go:6-9 prototypes): caller-supplied
 * buffers, integer codes 0 / 1 / 2 / 100 (binding/c/native/src/lib.rs:10-13).
 * argv: vk_json_file proof_json_file bad_proof_json_file */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void verbose(int);
int setup(const char*, const char*, const char*, char*, size_t, char*, size_t);
int prove(const char*, const char*, char*, size_t, char*, size_t);
int verify(const char*, const char*, char*, size_t);

static char* slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(3); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* b = malloc((size_t)n + 1);
    if (fread(b, 1, (size_t)n, f) != (size_t)n) exit(3);
    b[n] = 0;
    fclose(f);
    return b;
}
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) return 3;
    char* vk = slurp(argv[1]);
    char* proof = slurp(argv[2]);
    char* bad = slurp(argv[3]);
    char err[512];
    verbose(0);
    verbose(0);                                           /* idempotent (the reference would panic on the second call) */
    memset(err, 'x', sizeof err);
    CHECK(verify(vk, proof, err, sizeof err) == 0);       /* ERR_NONE */
    CHECK(err[0] == 'x');                                 /* nothing written on success */
    CHECK(verify(vk, bad, err, sizeof err) == 2);         /* ERR_VERIFICATION_FAILED: wrong public input */
    CHECK(verify("{not json", proof, err, sizeof err) == 100);   /* ERR_CUSTOM + Debug text */
    CHECK(strncmp(err, "Json(", 5) == 0);
    CHECK(verify("{not json", proof, err, 4) == 1);       /* ERR_BUFFER_TOO_SMALL: the message does not fit */
    CHECK(verify(vk, "{\"a\":[\"1\",\"2\"]}", err, sizeof err) == 100);
    CHECK(strstr(err, "missing field") != NULL);
    /* setup: the verifier type is checked before anything else (lib.rs:65-69) */
    char out[64];
    CHECK(setup("nofile", "nopk", "yaml", out, sizeof out, err, sizeof err) == 100);
    CHECK(strcmp(err, "invalid validator type") == 0);
    CHECK(setup("/nonexistent/circuit", "/tmp/pk", "json", out, sizeof out, err, sizeof err) == 100);
    CHECK(strncmp(err, "Io(", 3) == 0);
    /* prove: inputs that carry no witness dump are an Evaluator error; a missing key file an Io error */
    CHECK(prove("/nonexistent/proving.key", "{\"p\": 2, \"q\": 3}", out, sizeof out, err, sizeof err) == 100);
    CHECK(strncmp(err, "Evaluator(", 10) == 0);
    CHECK(prove("/nonexistent/proving.key", "{\"@binwitness\": \"/nonexistent/w\"}", out, sizeof out, err, sizeof err) == 100);
    CHECK(strncmp(err, "Io(", 3) == 0);
    printf("za2c ok\n");
    return 0;
}
