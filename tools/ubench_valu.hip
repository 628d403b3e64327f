// VALU instruction-throughput microbenchmark for gfx950 (informs the Montgomery-multiply design:
// which integer-multiply primitive is cheapest per bit-product).  Standalone: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ void __launch_bounds__(256) bench(uint64_t* out, int iters, uint32_t x, uint32_t y) {
    uint64_t a[8];
    uint32_t b[8], c[8];
    double d[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 77 + i; b[i] = x + i + threadIdx.x; c[i] = y + 3 * i; d[i] = (double)(threadIdx.x + i) * 1.0000001; }
    double dx = (double)x * 1e-9 + 1.0, dy = (double)y * 1e-12;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]) : "vcc");
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 4) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(b[i]), "+v"(c[i]) : "v"(c[(i + 1) & 7]), "v"(b[(i + 2) & 7]) : "vcc");
                if (OP == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(b[i]) : "v"(c[i]));
                if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 7) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 8) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dx), "v"(dy));
                if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(c[i]), "v"(c[(i + 1) & 7]));
                if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 12) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "s"(x) : "vcc");
                if (OP == 13) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(b[i]) : "v"(c[i]) : "vcc");
                if (OP == 14) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dx));
                if (OP == 15) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 16) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(a[i]));
                if (OP == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 18) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(b[i]) : "v"(c[i]));
                if (OP == 19) asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(b[i]));
                if (OP == 20) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(a[i]));
                if (OP == 21) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]) : "vcc");
            }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + b[i] + c[i] + (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint64_t* d_out) {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bench<OP><<<blocks, threads>>>(d_out, 10, 3, 5);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    bench<OP><<<blocks, threads>>>(d_out, iters, 3, 5);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double insts = (double)blocks * threads / 64 * iters * 32;   // wave-instructions
    double per_simd_per_s = insts / (ms * 1e-3) / (256 * 4);
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    printf("%-28s %8.3f ms  %7.2f Gwave-inst/s  ~%5.2f cyc/wave-inst/SIMD @2.4GHz  %7.2f Tlane-op/s\n", name, ms, insts / ms * 1e-6,
           2.4e9 / per_simd_per_s, insts * 64 / ms * 1e-9);
}

int main() {
    uint64_t* d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * 8 * 256));
    run<10>("v_fma_f32", d_out);
    run<0>("v_mad_u64_u32", d_out);
    run<12>("v_mad_u64_u32 (sgpr src)", d_out);
    run<1>("v_mul_lo_u32", d_out);
    run<2>("v_mul_hi_u32", d_out);
    run<3>("v_lshl_add_u64", d_out);
    run<4>("v_add_co+v_addc_co (pair)", d_out);
    run<13>("v_add_co_u32", d_out);
    run<5>("v_mov_b32", d_out);
    run<16>("v_lshrrev_b64", d_out);
    run<20>("v_ashrrev_i64", d_out);
    run<17>("v_and_b32", d_out);
    run<18>("v_alignbit_b32", d_out);
    run<19>("v_lshrrev_b32", d_out);
    run<21>("v_mad_i64_i32", d_out);
    run<8>("v_fma_f64", d_out);
    run<6>("v_mad_u32_u24", d_out);
    run<15>("v_mad_i32_i24", d_out);
    run<7>("v_mul_hi_u32_u24", d_out);
    run<11>("v_mul_u32_u24", d_out);
    run<9>("v_add3_u32", d_out);
    run<8>("v_fma_f64", d_out);
    run<14>("v_mul_f64", d_out);
    return 0;
}
