// Diagnostic micro-benchmark (not part of the product): issue rate of packed f32 VALU instructions on gfx950.
// Each wave runs ITER iterations of 16 independent instructions of one kind; the chip is filled with 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_rate_probe.hip -o /tmp/pk_rate_probe && /tmp/pk_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(S) S S S S S S S S
template <int KIND>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed)
{
    f2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {   // 16 scalar fma (two per "pair")
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0.y), "+v"(a1.y), "+v"(a2.y), "+v"(a3.y), "+v"(a4.y), "+v"(a5.y), "+v"(a6.y), "+v"(a7.y) : "v"(m.y), "v"(c.y));
        } else if (KIND == 1) {   // 16 packed fma
            REP8(asm volatile(
                "v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n"
                : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));
                asm volatile("" : "+v"(a0), "+v"(a1));)
        } else if (KIND == 2) {   // 16 packed fma on 8 independent registers
            asm volatile(
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 3) {   // 16 packed mul
            asm volatile(
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (KIND == 4) {   // 16 packed add
            asm volatile(
                "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (KIND == 5) {   // 16 scalar mul
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x));
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                : "+v"(a0.y), "+v"(a1.y), "+v"(a2.y), "+v"(a3.y), "+v"(a4.y), "+v"(a5.y), "+v"(a6.y), "+v"(a7.y) : "v"(m.y));
        } else if (KIND == 6) {   // 16 sqrt (transcendental unit)
            asm volatile(
                "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));
            asm volatile(
                "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                : "+v"(a0.y), "+v"(a1.y), "+v"(a2.y), "+v"(a3.y), "+v"(a4.y), "+v"(a5.y), "+v"(a6.y), "+v"(a7.y));
        } else if (KIND == 7) {   // 8 sqrt interleaved with 8 fma: does the transcendental unit overlap?
            asm volatile(
                "v_sqrt_f32 %0, %0\n v_fma_f32 %4, %4, %8, %9\n v_sqrt_f32 %1, %1\n v_fma_f32 %5, %5, %8, %9\n"
                "v_sqrt_f32 %2, %2\n v_fma_f32 %6, %6, %8, %9\n v_sqrt_f32 %3, %3\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));
            asm volatile(
                "v_sqrt_f32 %0, %0\n v_fma_f32 %4, %4, %8, %9\n v_sqrt_f32 %1, %1\n v_fma_f32 %5, %5, %8, %9\n"
                "v_sqrt_f32 %2, %2\n v_fma_f32 %6, %6, %8, %9\n v_sqrt_f32 %3, %3\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0.y), "+v"(a1.y), "+v"(a2.y), "+v"(a3.y), "+v"(a4.y), "+v"(a5.y), "+v"(a6.y), "+v"(a7.y) : "v"(m.y), "v"(c.y));
        } else if (KIND == 8) {   // 16 f64 multiplies
            double d0 = a0.x, d1 = a1.x;
            REP8(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n" : "+v"(d0), "+v"(d1) : "v"((double)m.x));)
            a0.x = (float)d0; a1.x = (float)d1;
        } else if (KIND == 9) {   // 16 v_cndmask / compare pairs
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n" : "+v"(a0.x) : "v"(a1.x) : "vcc");)
        }
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 123.456f) out[threadIdx.x] = s.x;
}

template <int KIND>
void run(float* d, const char* name, double flops_per_instr_lane)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 4096, blocks = 256 * 8, reps = 5;   // 8 blocks of 4 waves per CU: 8 waves per SIMD
    probe<KIND><<<blocks, 256>>>(d, 16, 1.0f);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) probe<KIND><<<blocks, 256>>>(d, iters, 1.0f + i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    double winstr = 16.0 * iters * blocks * 4;   // wave-instructions
    double per_cu_clk = winstr / (ms * 1e-3) / 256 / 2.4e9;
    printf("%-44s %.3f ms  %.3e wave-instr/s  %.3f per CU per clock (2.4 GHz)  %.1f TFLOP/s\n", name, ms, winstr / (ms * 1e-3),
           per_cu_clk, winstr * 64 * flops_per_instr_lane / (ms * 1e-3) / 1e12);
}

int main()
{
    float* d; CK(hipMalloc(&d, 4096));
    run<0>(d, "v_fma_f32", 2);
    run<1>(d, "v_pk_fma_f32 (2 registers, dependent pairs)", 4);
    run<2>(d, "v_pk_fma_f32 (8 independent registers)", 4);
    run<3>(d, "v_pk_mul_f32", 2);
    run<4>(d, "v_pk_add_f32", 2);
    run<5>(d, "v_mul_f32", 1);
    run<6>(d, "v_sqrt_f32", 1);
    run<7>(d, "v_sqrt_f32 + v_fma_f32 interleaved", 1.5);
    run<8>(d, "v_mul_f64", 1);
    run<9>(d, "v_cmp_lt_f32 + v_cndmask_b32", 0);
    return 0;
}
