// Launch-cost microbenchmark: a chain of N small dependent kernels queued on a stream one by one, and the same chain
// captured once in a hipGraph and replayed.  Prints host time to queue and time until the stream drains.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_graph.hip -o tools/ubench_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ int g_spin = 1;
__global__ void k_step(uint32_t* p, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = p[i];
    for (int k = 0; k < g_spin; k++) v = v * 1664525u + 1013904223u;   // a dependent chain: ~spin x 8 cycles
    p[i] = v;
}
static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32, blocks = argc > 2 ? atoi(argv[2]) : 64, reps = 200;
    const int spin = argc > 3 ? atoi(argv[3]) : 1;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_spin), &spin, sizeof(int)));
    uint32_t* d;
    CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    CK(hipMemset(d, 1, (size_t)blocks * 256 * 4));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto chain = [&]() { for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, st, d, (uint32_t)blocks * 256); };
    for (int w = 0; w < 5; w++) chain();
    CK(hipStreamSynchronize(st));
    double q = 0, t = 0;
    for (int r = 0; r < reps; r++) {
        double t0 = now_us();
        chain();
        double t1 = now_us();
        CK(hipStreamSynchronize(st));
        q += t1 - t0; t += now_us() - t0;
    }
    printf("stream: %d kernels x %d blocks: queue %.1f us, drain %.1f us (%.2f us per kernel)\n", N, blocks, q / reps, t / reps, t / reps / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(st, &g));
    double i0 = now_us();
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("instantiate %.1f us\n", now_us() - i0);
    for (int w = 0; w < 5; w++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    q = t = 0;
    for (int r = 0; r < reps; r++) {
        double t0 = now_us();
        CK(hipGraphLaunch(ge, st));
        double t1 = now_us();
        CK(hipStreamSynchronize(st));
        q += t1 - t0; t += now_us() - t0;
    }
    printf("graph : %d kernels x %d blocks: queue %.1f us, drain %.1f us (%.2f us per kernel)\n", N, blocks, q / reps, t / reps, t / reps / N);
    // four streams at once (the prover's side streams): queue all, then drain all
    hipStream_t ss[4]; hipGraphExec_t ges[4];
    for (int s = 0; s < 4; s++) { CK(hipStreamCreateWithFlags(&ss[s], hipStreamNonBlocking)); CK(hipGraphInstantiate(&ges[s], g, nullptr, nullptr, 0)); }
    for (int mode = 0; mode < 2; mode++) {
        q = t = 0;
        for (int r = 0; r < reps + 5; r++) {
            double t0 = now_us();
            for (int s = 0; s < 4; s++) {
                if (mode) CK(hipGraphLaunch(ges[s], ss[s]));
                else for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, ss[s], d, (uint32_t)blocks * 256);
            }
            double t1 = now_us();
            for (int s = 0; s < 4; s++) CK(hipStreamSynchronize(ss[s]));
            if (r >= 5) { q += t1 - t0; t += now_us() - t0; }
        }
        printf("%s x4 streams: queue %.1f us, drain %.1f us\n", mode ? "graph " : "stream", q / reps, t / reps);
    }
    return 0;
}
