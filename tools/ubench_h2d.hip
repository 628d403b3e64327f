// Host-to-device copy rate from pinned and from pageable memory (the witness hand-over of zg_prove):
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_h2d.hip -o tools/ubench_h2d && tools/ubench_h2d [MiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 128) << 20;
    void *d, *pinned;
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    void* pageable = malloc(bytes);
    memset(pinned, 1, bytes);
    memset(pageable, 1, bytes);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int kind = 0; kind < 2; kind++) {
        const void* src = kind ? pageable : pinned;
        double best = 1e30;
        for (int r = 0; r < 8; r++) {
            const double t0 = now_ms();
            CK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            const double t = now_ms() - t0;
            if (r > 0 && t < best) best = t;
        }
        printf("%-8s %zu MiB: %.3f ms = %.1f GB/s\n", kind ? "pageable" : "pinned", bytes >> 20, best, bytes / best / 1e6);
    }
    return 0;
}
