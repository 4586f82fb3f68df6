// Diagnostic: the division n / d as the compiler's IEEE sequence minus v_div_scale / v_div_fixup (which are no-ops
// for operands well inside the normal range) against the `/` operator, bit for bit, over random and structured
// operand pairs with 2^-60 <= |n|, d <= 2^60 and |n| <= 2 d (what dcsg_normalize feeds it).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fast_div_probe.hip -o tools/_build/fast_div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float fast_div(float n, float d, float r1)
{
    float q = n * r1;
    float e2 = __builtin_fmaf(-d, q, n);
    float q1 = __builtin_fmaf(e2, r1, q);
    float e3 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e3, r1, q1);
}
__device__ __forceinline__ float refined_rcp(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ unsigned long long splitmix(unsigned long long& s)
{
    unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// mode 0: d random in [2^-60, 2^60], n = random sign, random magnitude in [2^-60, 2 d]
// mode 1: d random mantissa with exponent near 0 (lengths of gradients), n any float with |n| <= d (the common case)
// mode 2: d = every mantissa at a few exponents x n from a small set of hard ratios (n = d * k / 2^m patterns)
__global__ void probe(int mode, unsigned long long per_thread, unsigned long long seed, unsigned long long* counts)
{
    unsigned long long s = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ull;
    unsigned long long bad = 0, n_done = 0;
    for (unsigned long long it = 0; it < per_thread; it++) {
        unsigned long long a = splitmix(s), b = splitmix(s);
        float d, n;
        if (mode == 0) {
            int ed = (int)(a % 121) - 60;                       // exponent of d
            d = ldexpf(1.0f + (float)((a >> 8) & 0x7fffff) * 0x1p-23f, ed);
            int en = -60 + (int)(b % (unsigned)(ed + 60 + 2));   // up to exponent of 2d
            n = ldexpf(1.0f + (float)((b >> 8) & 0x7fffff) * 0x1p-23f, en);
            if (fabsf(n) > 2.0f * d) n = d;
        } else if (mode == 1) {
            d = ldexpf(1.0f + (float)((a >> 8) & 0x7fffff) * 0x1p-23f, (int)(a % 5) - 2);
            n = ldexpf(1.0f + (float)((b >> 8) & 0x7fffff) * 0x1p-23f, -(int)(b % 40)) * d;
            if (fabsf(n) > d) n = d;
            if (fabsf(n) < 0x1p-60f) n = 0x1p-60f;
        } else {
            d = __uint_as_float(0x3f800000u + (unsigned)(a & 0x7fffff) + (((unsigned)(a >> 23) % 3u) << 23));
            unsigned k = (unsigned)(b & 0xff) + 1u;
            n = d * (float)k * 0x1p-8f;                          // quotients near k/256: many exact or tie-adjacent cases
            n = __uint_as_float(__float_as_uint(n) + (unsigned)((b >> 8) % 3u) - 1u);
        }
        if (b >> 63) n = -n;
        if (!(d >= 0x1p-60f && d <= 0x1p60f && fabsf(n) >= 0x1p-60f && fabsf(n) <= 0x1p61f)) continue;
        float want = n / d;
        float got = fast_div(n, d, refined_rcp(d));
        n_done++;
        if (__float_as_uint(want) != __float_as_uint(got)) bad++;
    }
    atomicAdd(&counts[0], n_done);
    atomicAdd(&counts[1], bad);
}

int main()
{
    unsigned long long* dcnt;
    CK(hipMalloc(&dcnt, 16));
    for (int mode = 0; mode < 3; mode++) {
        CK(hipMemset(dcnt, 0, 16));
        probe<<<8192, 256>>>(mode, 2048, 0x1234567ull + mode, dcnt);   // 4.3e9 pairs per mode
        unsigned long long h[2];
        CK(hipMemcpy(h, dcnt, 16, hipMemcpyDeviceToHost));
        printf("mode %d: %llu operand pairs, %llu results differ from n / d\n", mode, h[0], h[1]);
    }
    return 0;
}
