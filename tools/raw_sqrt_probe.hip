// Diagnostic: is the hardware v_sqrt_f32 correctly rounded on this GPU?  Compares it with (float)sqrt((double)x)
// (correctly rounded: 53 >= 2*24+2) for every positive normal float.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/raw_sqrt_probe.hip -o tools/_build/raw_sqrt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void probe(unsigned long long* counts)
{
    unsigned long long bad = 0, worse = 0, n = 0;
    const unsigned long long total = 0x7f800000ull - 0x00800000ull;  // positive normals
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned int bits = 0x00800000u + (unsigned int)i;
        float x = __uint_as_float(bits);
        float hw = __builtin_amdgcn_sqrtf(x);
        float cr = (float)sqrt((double)x);
        n++;
        if (__float_as_uint(hw) != __float_as_uint(cr)) {
            bad++;
            int d = (int)__float_as_uint(hw) - (int)__float_as_uint(cr);
            if (d > 1 || d < -1) worse++;
        }
    }
    atomicAdd(&counts[0], n);
    atomicAdd(&counts[1], bad);
    atomicAdd(&counts[2], worse);
}

int main()
{
    unsigned long long* d;
    CK(hipMalloc(&d, 24));
    CK(hipMemset(d, 0, 24));
    probe<<<4096, 256>>>(d);
    unsigned long long h[3];
    CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
    printf("v_sqrt_f32 vs correctly rounded over %llu positive normal floats: %llu differ (%.3f %%), %llu by more than 1 ulp\n",
           h[0], h[1], 100.0 * (double)h[1] / (double)h[0], h[2]);
    return 0;
}
