// Calibration of rocprofv3's FETCH_SIZE for the access pattern of k_accumulate: every lane gathers one 64-byte record
// (4 x global_load_dwordx4) at a random index of a table far larger than the 256 MiB Infinity Cache.  The algorithmic
// bytes are known (64 B per gather), so  correction = known bytes / (FETCH_SIZE * 1024)  -- MI355X_MICROARCH.md (HBM
// section) validates the "x 2" only for wide coalesced streaming reads and asks for exactly this calibration.
// Also runs a coalesced streaming read of the same table for reference.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o tools/ubench_gather
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/ubench_gather
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
__global__ void __launch_bounds__(256) k_gather64(const Rec* __restrict__ table, uint64_t n_rec, uint64_t gathers_per_thread,
                                                  uint32_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t k = 0; k < gathers_per_thread; k++) {
        const Rec r = table[mix(t * gathers_per_thread + k) % n_rec];
        acc ^= r.v[0].x ^ r.v[1].y ^ r.v[2].z ^ r.v[3].w;
    }
    out[t] = acc;
}
__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ table, uint64_t n16, uint32_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = table[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const uint64_t n_rec = (uint64_t)1 << 26;          // 4 GiB table
    const uint64_t threads = (uint64_t)1 << 22, per = 16;   // 2^26 gathers = 4 GiB of algorithmic bytes
    Rec* d_table;
    uint32_t* d_out;
    CK(hipMalloc(&d_table, n_rec * sizeof(Rec)));
    CK(hipMalloc(&d_out, threads * 4));
    CK(hipMemset(d_table, 0x5a, n_rec * sizeof(Rec)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        k_gather64<<<(unsigned)(threads / 256), 256>>>(d_table, n_rec, per, d_out);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("gather64: %llu gathers, %.3f GB algorithmic, %.3f ms, %.1f GB/s\n", (unsigned long long)(threads * per),
               threads * per * 64 / 1e9, ms, threads * per * 64 / 1e6 / ms);
    }
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        k_stream<<<256 * 16, 256>>>((const uint4*)d_table, n_rec * 4, d_out);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream: %.3f GB, %.3f ms, %.1f GB/s\n", n_rec * 64 / 1e9, ms, n_rec * 64 / 1e6 / ms);
    }
    return 0;
}
