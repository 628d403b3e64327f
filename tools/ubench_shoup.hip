// Twiddle multiplication of the Fr transform on gfx950: Montgomery (fd.h operator*) against a Shoup-style product with a
// precomputed companion ws = floor(w 2^261 / p) -- q = floor(a ws / 2^261) from the product's columns 7..16 (53
// multiply-adds), r = low9(a w) - low9(q p) (45 + 45): 143 multiply-adds instead of 162 + 9.  Same dependency-chain
// harness as ubench_fmul.hip; the second kernel checks the two products against each other on random operands.
//   hipcc -O3 --offload-arch=gfx950 -I za_amd/csrc tools/ubench_shoup.hip -o tools/ubench_shoup && tools/ubench_shoup
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../za_amd/csrc/ecd.h"
using namespace zg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Tw { uint32_t w[9], ws[9]; };

// a: limbs below 2^31 (lazy sums allowed), value below 2^261 * 3; w canonical (< p), ws = floor(w 2^261 / p).
// Result: a w mod p, normalised, below 3 p.
__device__ __forceinline__ FrD mul_shoup(const FrD& a, const uint32_t* w, const uint32_t* ws) {
    constexpr uint32_t MASK = FrD::MASK;
    uint32_t q[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 7; k < 17; k++) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) acc += (uint64_t)a.l[i] * ws[k - i];
        if (k >= 9) q[k - 9] = (uint32_t)acc & MASK;
        acc >>= 29;
    }
    q[8] = (uint32_t)acc;
    FrD r;
    int64_t t = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) t += (int64_t)(int32_t)a.l[i] * (int32_t)w[k - i];
#pragma unroll
        for (int i = 0; i <= k; i++) t += (int64_t)(int32_t)q[i] * (-(int32_t)Fr29::P[k - i]);
        r.l[k] = (uint32_t)t & MASK;
        t >>= 29;
    }
    return r;
}

template <int V, int CHAINS>
__global__ void __launch_bounds__(256) bench(const Fr* in, const Tw* tw, Fr* out, int iters) {
    extern __shared__ uint32_t pad[];
    FrD x[CHAINS];
    Tw T = tw[threadIdx.x & 63];
    FrD wm;   // the Montgomery form of the same twiddle (for V = 0 any canonical value does)
#pragma unroll
    for (int i = 0; i < 9; i++) wm.l[i] = T.w[i];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = FrD::load_red(&in[(threadIdx.x + 7 * c) & 255]);
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (V == 0) x[c] = x[c] * wm;
            if (V == 1) x[c] = mul_shoup(x[c], T.w, T.ws);
        }
    }
    FrD s = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; c++) s = s + x[c];
    if (pad[threadIdx.x & 15] == 12345u) s = s + x[0];
    s.reduce().store(&out[blockIdx.x * blockDim.x + threadIdx.x]);
}

// out[0] = (a * wm) canonical, out[1] = shoup(a, w, ws) canonical, where wm = w 2^261 mod p
__global__ void check(const Fr* a, const Tw* tw, const Fr* wm, Fr* o0, Fr* o1, int n, int lazy) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FrD x = FrD::load_red(&a[i]);
    if (lazy) x = FrD::add_lazy(FrD::add_lazy(x, x), x);   // limbs up to 3 * 2^29
    const FrD m = FrD::load_canon(&wm[i & 63]);
    (x * m).canonical().store(&o0[i]);
    mul_shoup(x, tw[i & 63].w, tw[i & 63].ws).canonical().store(&o1[i]);
}

template <int V, int CHAINS>
void run(const char* name, const Fr* d_in, const Tw* d_tw, Fr* d_out, int waves_per_simd) {
    const size_t lds = waves_per_simd >= 8 ? 1024 : (size_t)(160 * 1024 / waves_per_simd) - 512;
    const int blocks = 256 * waves_per_simd * 4, iters = 600;
    CK(hipFuncSetAttribute((const void*)bench<V, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bench<V, CHAINS><<<blocks, 256, lds>>>(d_in, d_tw, d_out, 3);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    bench<V, CHAINS><<<blocks, 256, lds>>>(d_in, d_tw, d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double muls = (double)blocks * 256 * iters * CHAINS;
    printf("%-28s chains %d waves/SIMD %d: %8.3f ms  %7.1f ns per wave-mul per SIMD\n", name, CHAINS, waves_per_simd, ms,
           ms * 1e6 / (muls / 64 / 1024));
}

#include "ubench_shoup_tw.inc"   // static const Tw TWS[64]; static const uint32_t TWM[64][8];  (tools/gen_shoup_tw.py)

int main() {
    Fr h[256];
    for (int i = 0; i < 256; i++)
        for (int j = 0; j < 8; j++) h[i].l[j] = (uint32_t)(0x9E3779B9u * (i * 8 + j + 1)) & (j == 7 ? 0x0fffffffu : 0xffffffffu);
    Fr *d_in, *d_out, *d_o0, *d_o1, *d_wm;
    Tw* d_tw;
    CK(hipMalloc(&d_in, sizeof h));
    CK(hipMalloc(&d_out, sizeof(Fr) * 256 * 256 * 8 * 4));
    CK(hipMalloc(&d_o0, sizeof h)); CK(hipMalloc(&d_o1, sizeof h));
    CK(hipMalloc(&d_tw, sizeof TWS)); CK(hipMalloc(&d_wm, sizeof TWM));
    CK(hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, TWS, sizeof TWS, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wm, TWM, sizeof TWM, hipMemcpyHostToDevice));
    for (int lazy = 0; lazy < 2; lazy++) {
        check<<<1, 256>>>(d_in, d_tw, d_wm, d_o0, d_o1, 256, lazy);
        Fr a[256], b[256];
        CK(hipMemcpy(a, d_o0, sizeof a, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b, d_o1, sizeof b, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; i++) bad += memcmp(&a[i], &b[i], 32) != 0;
        printf("check (lazy operand %d): %d of 256 differ\n", lazy, bad);
    }
    for (int w : {1, 2, 3, 4}) {
        run<0, 1>("Montgomery operator*", d_in, d_tw, d_out, w);
        run<1, 1>("Shoup companion", d_in, d_tw, d_out, w);
        run<0, 2>("Montgomery operator*", d_in, d_tw, d_out, w);
        run<1, 2>("Shoup companion", d_in, d_tw, d_out, w);
    }
    return 0;
}
