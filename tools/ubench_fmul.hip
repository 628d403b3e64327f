// Field-multiplication throughput on gfx950 for several formulations of the 9x29 Montgomery product (fd.h):
//   0  Fd::operator* as the compiler schedules it (next column's products summed apart, joined with a 64-bit add)
//   1  the same columns as ONE dependent v_mad_u64_u32 chain (inline asm): no joins
//   2  Fd::sqr          3  Fd::mac2 (two products, one reduction)
// Each thread runs CHAINS independent dependency chains; occupancy is limited with dynamic LDS so that the numbers can
// be read at 1, 2, 3 ... waves per SIMD (k_accumulate<Fq> runs at 3).
//   hipcc -O3 --offload-arch=gfx950 -I za_amd/csrc tools/ubench_fmul.hip -o tools/ubench_fmul && tools/ubench_fmul
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../za_amd/csrc/ecd.h"
using namespace zg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void mad(uint64_t& acc, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mad_s(uint64_t& acc, uint32_t a, uint32_t b_const) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b_const) : "vcc");
}
__device__ __forceinline__ FqD mul_serial(const FqD& a, const FqD& b) {
    constexpr uint32_t MASK = FqD::MASK;
    uint32_t m[9];
    FqD r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mad_s(acc, m[i], Fq29::P[k - i]);
        m[k] = ((uint32_t)acc * Fq29::NINV) & MASK;
        mad_s(acc, m[k], Fq29::P[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 8; i <= 8; i++) mad_s(acc, m[i], Fq29::P[k - i]);
        r.l[k - 9] = (uint32_t)acc & MASK;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

template <int V, int CHAINS>
__global__ void __launch_bounds__(256) bench(const Fq* in, Fq* out, int iters) {
    extern __shared__ uint32_t pad[];
    FqD x[CHAINS], y[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        x[c] = FqD::load_red(&in[(threadIdx.x + 7 * c) & 255]);
        y[c] = FqD::load_red(&in[(threadIdx.x + 13 * c + 1) & 255]);
    }
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (V == 0) x[c] = x[c] * y[c];
            if (V == 1) x[c] = mul_serial(x[c], y[c]);
            if (V == 2) x[c] = x[c].sqr();
            if (V == 3) x[c] = FqD::mac2(x[c], y[c], y[c], x[c]);
        }
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (V == 0) y[c] = y[c] * x[c];
            if (V == 1) y[c] = mul_serial(y[c], x[c]);
            if (V == 2) y[c] = y[c].sqr();
            if (V == 3) y[c] = FqD::mac2(y[c], x[c], x[c], y[c]);
        }
    }
    FqD s = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; c++) s = s + x[c] + y[c];
    if (pad[threadIdx.x & 15] == 12345u) s = s + y[0];
    s.reduce().store(&out[blockIdx.x * blockDim.x + threadIdx.x]);
}

template <int V, int CHAINS>
void run(const char* name, const Fq* d_in, Fq* d_out, int waves_per_simd) {
    // 256 threads = 1 wave per SIMD per block; blocks per CU limited through LDS: 160 KB / k
    const size_t lds = waves_per_simd >= 8 ? 1024 : (size_t)(160 * 1024 / waves_per_simd) - 512;
    const int blocks = 256 * waves_per_simd * 4, iters = 300;
    CK(hipFuncSetAttribute((const void*)bench<V, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bench<V, CHAINS><<<blocks, 256, lds>>>(d_in, d_out, 3);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    bench<V, CHAINS><<<blocks, 256, lds>>>(d_in, d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double muls = (double)blocks * 256 * iters * 2 * CHAINS;          // lane multiplications
    const double wave_muls_per_simd = muls / 64 / 1024;                      // per SIMD (1024 SIMDs)
    printf("%-34s chains %d waves/SIMD %d: %8.3f ms  %7.2f G lane-mul/s  %7.1f ns per wave-mul per SIMD\n", name, CHAINS,
           waves_per_simd, ms, muls / ms * 1e-6, ms * 1e6 / wave_muls_per_simd);
}

int main() {
    Fq h[256];
    for (int i = 0; i < 256; i++)
        for (int j = 0; j < 8; j++) h[i].l[j] = (uint32_t)(0x9E3779B9u * (i * 8 + j + 1)) & (j == 7 ? 0x0fffffffu : 0xffffffffu);
    Fq *d_in, *d_out;
    CK(hipMalloc(&d_in, sizeof h));
    CK(hipMalloc(&d_out, sizeof(Fq) * 256 * 256 * 8 * 4));
    CK(hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice));
    for (int w : {1, 2, 3, 4}) {
        run<0, 1>("operator* (compiler schedule)", d_in, d_out, w);
        run<1, 1>("serial v_mad chain (asm)", d_in, d_out, w);
        run<0, 2>("operator* (compiler schedule)", d_in, d_out, w);
        run<1, 2>("serial v_mad chain (asm)", d_in, d_out, w);
        run<2, 2>("sqr", d_in, d_out, w);
        run<3, 2>("mac2", d_in, d_out, w);
    }
    // results equal?
    return 0;
}
