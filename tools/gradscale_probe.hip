// Diagnostic (not part of the product): is the double-precision gradient scale of get_normal (k2.cl:175-177),
// (float)((1.0 / (double)twoE) * (double)D), reproduced for EVERY float D by float-only arithmetic?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/gradscale_probe.hip -o /tmp/gradscale_probe && /tmp/gradscale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) probe(unsigned long long* bad, unsigned int* first_bad, float kh, float kl, double k, unsigned int* by_exp)
{
    unsigned long long n0 = 0, n1 = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b < (1ull << 32); b += (unsigned long long)gridDim.x * 256) {
        float D = __uint_as_float((unsigned int)b);
        float ref = (float)(k * (double)D);
        float c0 = __builtin_fmaf(kh, D, kl * D);                 // candidate 0
        float p = kh * D, e = __builtin_fmaf(kh, D, -p);          // candidate 1: two-product + low part
        float c1 = p + (e + kl * D);
        bool nan_ok = (ref != ref);
        if (!(nan_ok ? (c0 != c0) : (__float_as_uint(c0) == __float_as_uint(ref)))) { n0++; atomicMin(first_bad, (unsigned int)b); atomicAdd(&by_exp[((unsigned int)b >> 23) & 0xff], 1u); }
        if (!(nan_ok ? (c1 != c1) : (__float_as_uint(c1) == __float_as_uint(ref)))) n1++;
    }
    if (n0) atomicAdd(&bad[0], n0);
    if (n1) atomicAdd(&bad[1], n1);
}

int main()
{
    const float twoE = (float)(2.0 * 0.005);
    const double k = 1.0 / (double)twoE;
    const float kh = (float)k;
    const float kl = (float)(k - (double)kh);
    printf("twoE=%a k=%.17g kh=%a kl=%a\n", twoE, k, kh, kl);
    unsigned long long* bad; unsigned int* fb;
    CK(hipMalloc(&bad, 16)); CK(hipMalloc(&fb, 4));
    CK(hipMemset(bad, 0, 16)); CK(hipMemset(fb, 0xff, 4));
    unsigned int* be; CK(hipMalloc(&be, 1024)); CK(hipMemset(be, 0, 1024));
    probe<<<256 * 16, 256>>>(bad, fb, kh, kl, k, be);
    CK(hipDeviceSynchronize());
    unsigned long long h[2]; unsigned int hf;
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost));
    printf("mismatches over all 2^32 inputs: fma(kh, D, kl*D): %llu (first 0x%08x)   two-product: %llu\n", h[0], hf, h[1]);
    unsigned int hb[256]; CK(hipMemcpy(hb, be, 1024, hipMemcpyDeviceToHost));
    for (int e = 0; e < 256; e++) if (hb[e]) printf("  biased exponent %d: %u mismatches (candidate 0)\n", e, hb[e]);
    return 0;
}
