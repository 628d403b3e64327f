// 64-byte random gathers (the access pattern of k_accumulate) against the SPAN they range over: is the ceiling of that
// pattern (1.55 TB/s over a 4 GiB table, tools/ubench_gather.hip) a property of the access size, or of the footprint
// (TLB reach, Infinity Cache)?  If it rises for spans of a few hundred MB, ordering a bucket's entries by table row --
// so that the waves in flight touch one row of the window table at a time -- would pay.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gather_span.hip -o tools/ubench_gather_span && tools/ubench_gather_span
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct alignas(16) Rec { uint4 v[4]; };
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// every thread makes `per` gathers; gather k of all threads falls into window (k * nwin / per) of the table: `nwin` windows of
// `span` records each, visited one after the other (nwin = 1: the whole table at random, as k_accumulate does today)
__global__ void __launch_bounds__(256) k_gather(const Rec* __restrict__ table, uint64_t span, uint32_t nwin, uint32_t per,
                                                uint32_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint64_t w = (uint64_t)k * nwin / per;
        const Rec r = table[w * span + mix(t * per + k) % span];
        acc ^= r.v[0].x ^ r.v[1].y ^ r.v[2].z ^ r.v[3].w;
    }
    out[t] = acc;
}
// the window table of the h multiexp: `rows` rows of `row_len` records; a thread's gather k goes to row (k mod rows) -- or, with
// `ordered`, to rows in ascending order -- at a random column below `cols` (cols = row_len / 2: what ONE part of the job touches)
__global__ void __launch_bounds__(256) k_gather_rows(const Rec* __restrict__ table, uint64_t row_len, uint32_t rows, uint64_t cols,
                                                     uint32_t per, int ordered, uint32_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint64_t row = ordered ? (uint64_t)k * rows / per : mix(t * 977 + k) % rows;
        const Rec r = table[row * row_len + mix(t * per + k) % cols];
        acc ^= r.v[0].x ^ r.v[1].y ^ r.v[2].z ^ r.v[3].w;
    }
    out[t] = acc;
}
int main() {
    const uint64_t n_rec = (uint64_t)1 << 26;   // 4 GiB
    const uint64_t threads = (uint64_t)1 << 21;
    const uint32_t per = 32;
    Rec* d_table;
    uint32_t* d_out;
    CK(hipMalloc(&d_table, n_rec * sizeof(Rec)));
    CK(hipMalloc(&d_out, threads * 4));
    CK(hipMemset(d_table, 0x5a, n_rec * sizeof(Rec)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t nwin : {1u, 2u, 4u, 8u, 16u, 32u}) {
        const uint64_t span = n_rec / nwin;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            k_gather<<<(unsigned)(threads / 256), 256>>>(d_table, span, nwin, per, d_out);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("windows %2u of %6.0f MB: %.3f ms, %.1f GB/s of 64-byte gathers\n", nwin, span * 64 / 1e6, ms, threads * per * 64 / 1e6 / ms);
        }
    }
    // the same with the whole kernel confined to one small span (everything in the Infinity Cache / TLB)
    for (uint64_t span : {(uint64_t)1 << 19, (uint64_t)1 << 21, (uint64_t)1 << 22, (uint64_t)1 << 23, (uint64_t)1 << 24}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            k_gather<<<(unsigned)(threads / 256), 256>>>(d_table, span, 1, per, d_out);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("one span of %6.0f MB: %.3f ms, %.1f GB/s\n", span * 64 / 1e6, ms, threads * per * 64 / 1e6 / ms);
        }
    }
    {   // 13 rows of 2^22 records = 3.5 GB, as the h query's table at 2^22
        const uint64_t row_len = (uint64_t)1 << 22;
        const uint32_t rows = 13, per2 = 26;
        for (int ordered = 0; ordered < 2; ordered++)
            for (uint64_t cols : {row_len, row_len / 2, row_len / 4, row_len / 8}) {
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    k_gather_rows<<<(unsigned)(threads / 256), 256>>>(d_table, row_len, rows, cols, per2, ordered, d_out);
                    CK(hipEventRecord(e1));
                    CK(hipDeviceSynchronize());
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) printf("13 rows x 2^22, columns below %8llu (%5.0f MB touched in 13 pieces), rows %s: %.3f ms, %.1f GB/s\n",
                                    (unsigned long long)cols, rows * cols * 64 / 1e6, ordered ? "in order " : "at random", ms,
                                    threads * per2 * 64 / 1e6 / ms);
                }
            }
    }
    return 0;
}
