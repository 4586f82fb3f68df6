// Diagnostic (not part of the product): for which of these sequences is n / d, with d's reciprocal refined once and
// shared, equal to IEEE division for EVERY pair of float mantissas?  (The quotient's bits depend on the mantissas only
// while nothing underflows: dcsg_normalize_fast holds operands to 2^-60 <= |n|, d <= 2^60, |n| <= d (1 + 2^-22).)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div_short_probe.hip -o /tmp/div_probe && /tmp/div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// bad[0]: one residual correction (3 ops per quotient); bad[1]: two (5 ops, what the compiler's division does between
// v_div_scale and v_div_fixup, shipped in rounds 1-2); bad[2]: one correction with the UNREFINED v_rcp_f32
__global__ void __launch_bounds__(256) probe(unsigned long long* bad, unsigned int d_first, unsigned int n_slices_log2)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned int dm = d_first + (unsigned int)(t >> n_slices_log2);          // mantissa of d
    const unsigned int slice = (unsigned int)(t & ((1u << n_slices_log2) - 1u));
    const unsigned int per = (1u << 23) >> n_slices_log2;
    const float d = __uint_as_float(0x3f800000u | dm);                              // [1, 2)
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    float r1 = __builtin_fmaf(e, r, r);
    unsigned int b0 = 0, b1 = 0, b2 = 0;
    for (unsigned int k = 0; k < per; k++) {
        const float n = __uint_as_float(0x3f800000u | (slice * per + k));           // [1, 2): n / d in (0.5, 2)
        const float want = n / d;
        float q = n * r1;
        float e2 = __builtin_fmaf(-d, q, n);
        float q1 = __builtin_fmaf(e2, r1, q);
        float e3 = __builtin_fmaf(-d, q1, n);
        float q2 = __builtin_fmaf(e3, r1, q1);
        float qa = n * r;
        float ea = __builtin_fmaf(-d, qa, n);
        float qb = __builtin_fmaf(ea, r, qa);
        b0 += __float_as_uint(q1) != __float_as_uint(want);
        b1 += __float_as_uint(q2) != __float_as_uint(want);
        b2 += __float_as_uint(qb) != __float_as_uint(want);
    }
    if (b0) atomicAdd(&bad[0], (unsigned long long)b0);
    if (b1) atomicAdd(&bad[1], (unsigned long long)b1);
    if (b2) atomicAdd(&bad[2], (unsigned long long)b2);
}

int main(int argc, char** argv)
{
    unsigned long long* bad; CK(hipMalloc(&bad, 24)); CK(hipMemset(bad, 0, 24));
    const unsigned int slices_log2 = 5, d_per_launch = 1u << 16;   // 2^21 threads per launch, 2^18 numerators each
    const unsigned int d_total = argc > 1 ? (unsigned int)atoi(argv[1]) : (1u << 23);
    for (unsigned int d0 = 0; d0 < d_total; d0 += d_per_launch) {
        probe<<<(d_per_launch << slices_log2) / 256, 256>>>(bad, d0, slices_log2);
        if ((d0 / d_per_launch) % 16 == 15) {
            CK(hipDeviceSynchronize());
            unsigned long long h[3]; CK(hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost));
            printf("denominator mantissas < %u done: mismatches one-correction %llu, two-corrections %llu, one-correction/raw rcp %llu\n",
                   d0 + d_per_launch, h[0], h[1], h[2]);
            fflush(stdout);
        }
    }
    CK(hipDeviceSynchronize());
    unsigned long long h[3]; CK(hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost));
    printf("all %u x 2^23 mantissa pairs: mismatches against IEEE n / d: one correction %llu, two corrections %llu, one correction with raw rcp %llu\n",
           d_total, h[0], h[1], h[2]);
    return 0;
}
