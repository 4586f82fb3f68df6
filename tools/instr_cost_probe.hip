// Diagnostic micro-benchmark (not part of the product): issue cost of the individual VALU instruction forms the exact
// sweep is made of, on gfx950.  Every kernel runs ITER x 16 instructions of one form on 8 independent registers with
// 8 waves per SIMD on every CU; the cost is printed relative to v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/instr_cost_probe.hip -o /tmp/instr_cost_probe && /tmp/instr_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define R8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)
#define TWICE(S) S S
// one line = 8 instructions, one per register
#define L8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                     OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"

// the same with the extra operand first (VOP2: literals and constants are src0)
#define L8P(OP, PRE) OP " %0, " PRE ", %0\n" OP " %1, " PRE ", %1\n" OP " %2, " PRE ", %2\n" OP " %3, " PRE ", %3\n" \
                     OP " %4, " PRE ", %4\n" OP " %5, " PRE ", %5\n" OP " %6, " PRE ", %6\n" OP " %7, " PRE ", %7\n"

template <int KIND>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) asm volatile(TWICE(L8("v_fma_f32", ", %8, %9")) R8);
        if (KIND == 1) asm volatile(TWICE(L8("v_fma_f32", ", -%8, %9")) R8);                       // neg modifier
        if (KIND == 2) asm volatile(TWICE(L8P("v_mul_f32", "0x3e4ccccd")) R8);                    // 32-bit literal
        if (KIND == 3) asm volatile(TWICE(L8P("v_add_f32", "-0.5")) R8);                          // inline constant
        if (KIND == 4) asm volatile(TWICE(L8P("v_add_u32", "-1")) R8);
        if (KIND == 5) asm volatile(TWICE(L8("v_min_f32", ", %8")) R8);
        if (KIND == 6) asm volatile(TWICE(L8("v_min3_f32", ", |%8|, |%9|")) R8);
        if (KIND == 7)   // compares into four SGPR pairs (VOP3)
            asm volatile(TWICE("v_cmp_ge_f32_e64 s[20:21], 0, %0\n v_cmp_lt_f32_e64 s[22:23], 0, %1\n v_cmp_ge_f32_e64 s[24:25], 0, %2\n"
                               "v_cmp_lt_f32_e64 s[26:27], 0, %3\n v_cmp_ge_f32_e64 s[20:21], 0, %4\n v_cmp_lt_f32_e64 s[22:23], 0, %5\n"
                               "v_cmp_ge_f32_e64 s[24:25], 0, %6\n v_cmp_lt_f32_e64 s[26:27], 0, %7\n")
                         R8 : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (KIND == 8)   // compares into VCC (VOP2/VOPC encoding)
            asm volatile(TWICE("v_cmp_ge_f32 vcc, 0, %0\n v_cmp_lt_f32 vcc, 0, %1\n v_cmp_ge_f32 vcc, 0, %2\n v_cmp_lt_f32 vcc, 0, %3\n"
                               "v_cmp_ge_f32 vcc, 0, %4\n v_cmp_lt_f32 vcc, 0, %5\n v_cmp_ge_f32 vcc, 0, %6\n v_cmp_lt_f32 vcc, 0, %7\n")
                         R8 : "vcc");
        if (KIND == 9)   // selects on an SGPR-pair mask set once (VOP3)
            asm volatile("v_cmp_ge_f32_e64 s[20:21], 0, %8\n" TWICE(L8("v_cndmask_b32_e64", ", %8, s[20:21]")) R8 : "s20", "s21");
        if (KIND == 10)  // selects on VCC set once
            asm volatile("v_cmp_ge_f32 vcc, 0, %8\n" TWICE(L8("v_cndmask_b32", ", %8, vcc")) R8 : "vcc");
        if (KIND == 11)  // the dependent pair as the fix-up uses it: compare -> SGPR pair -> select, eight independent pairs
            asm volatile("v_cmp_ge_f32_e64 s[20:21], 0, %0\n v_cmp_lt_f32_e64 s[22:23], 0, %1\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n"
                         "v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cmp_ge_f32_e64 s[20:21], 0, %2\n v_cmp_lt_f32_e64 s[22:23], 0, %3\n"
                         "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n"
                         "v_cmp_ge_f32_e64 s[20:21], 0, %4\n v_cmp_lt_f32_e64 s[22:23], 0, %5\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n"
                         "v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n v_cmp_ge_f32_e64 s[20:21], 0, %6\n v_cmp_lt_f32_e64 s[22:23], 0, %7\n"
                         "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]\n"
                         R8 : "s20", "s21", "s22", "s23");
        if (KIND == 12) asm volatile(TWICE(L8("v_sqrt_f32", "")) R8);
        if (KIND == 13) asm volatile(TWICE(L8("v_rcp_f32", "")) R8);
        if (KIND == 14) asm volatile(TWICE(L8("v_rsq_f32", "")) R8);
        if (KIND == 15) asm volatile(TWICE(L8("v_mul_f32", ", %8")) R8);
        if (KIND == 16) asm volatile(TWICE(L8("v_max_f32", ", %8")) R8);
        if (KIND == 18) asm volatile(TWICE(L8("v_med3_f32", ", %8, %9")) R8);
        if (KIND == 19) asm volatile(TWICE(L8("v_sub_f32", ", %8")) R8);
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 123.456f) out[threadIdx.x] = s;
}

static double base_ms = 0;
template <int KIND>
void run(float* d, const char* name)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 4096, blocks = 256 * 8, reps = 5;
    probe<KIND><<<blocks, 256>>>(d, 16, 1.0f);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) probe<KIND><<<blocks, 256>>>(d, iters, 1.0f + i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    if (KIND == 0) base_ms = ms;
    double winstr = 16.0 * iters * blocks * 4;
    printf("%-52s %.3f ms  %.3e wave-instr/s  x%.2f of v_fma_f32\n", name, ms, winstr / (ms * 1e-3), ms / base_ms);
}

int main()
{
    float* d; CK(hipMalloc(&d, 4096));
    run<0>(d, "v_fma_f32");
    run<1>(d, "v_fma_f32 with a negated operand");
    run<15>(d, "v_mul_f32");
    run<2>(d, "v_mul_f32 with a 32-bit literal");
    run<3>(d, "v_add_f32 with an inline constant");
    run<19>(d, "v_sub_f32");
    run<4>(d, "v_add_u32");
    run<5>(d, "v_min_f32");
    run<16>(d, "v_max_f32");
    run<6>(d, "v_min3_f32 with |abs| modifiers");
    run<18>(d, "v_med3_f32");
    run<7>(d, "v_cmp_*_f32_e64 -> SGPR pair");
    run<8>(d, "v_cmp_*_f32 -> VCC");
    run<9>(d, "v_cndmask_b32_e64 (SGPR-pair mask)");
    run<10>(d, "v_cndmask_b32 (VCC mask)");
    run<11>(d, "compare -> SGPR pair -> select (dependent pairs)");
    run<12>(d, "v_sqrt_f32");
    run<13>(d, "v_rcp_f32");
    run<14>(d, "v_rsq_f32");
    return 0;
}
