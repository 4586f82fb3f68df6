// Diagnostic micro-benchmark (not part of the product), round 3: why does the fused sweep run FASTER with every square
// root behind its own compare-and-branch than with seven of them in one basic block and one merged test?
// Each wave evaluates, per iteration, C = 7 independent "stencil" chains of the product's square root
//     s = a*a + b;  y = rsq(s);  g = s*y;  h = y/2;  d = fma(-g, g, s);  r = fma(d, h, g)
// in three code shapes, at 8 waves per SIMD on every CU, no memory traffic in the loop:
//   SEQ     chains one after the other, a scheduling barrier between them (one dependent chain at a time)
//   MIXED   no barriers: the compiler interleaves the seven chains as it likes (what the merged form got)
//   BRANCH  as SEQ, each chain followed by the product's guard: range test on g, ballot, never-taken branch
//   hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o tools/_build/chain_probe && tools/_build/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { SEQ, MIXED, BRANCH, MERGED };

template <int SHAPE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed)
{
    float a[7], acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 7; k++) a[k] = seed + 0.001f * (float)(threadIdx.x + 64 * k);
    const float b = 0.25f;
    for (int i = 0; i < iters; i++) {
        unsigned int gmn = 0x3f800000u, gmx = 0x3f800000u;
        float r[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            float s = a[k] * a[k] + b;
            float y = __builtin_amdgcn_rsqf(s);
            float g = s * y;
            float h = 0.5f * y;
            float d = __builtin_fmaf(-g, g, s);
            r[k] = __builtin_fmaf(d, h, g);
            if (SHAPE == BRANCH) {
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__float_as_uint(g) - 0x27800000u < 0x58000000u)) != 0ull, 0)) r[k] = __builtin_sqrtf(s);
            }
            if (SHAPE == MERGED) {
                const unsigned int gb = __float_as_uint(g);
                gmn = gb < gmn ? gb : gmn;
                gmx = gb > gmx ? gb : gmx;
            }
            if (SHAPE == SEQ) __builtin_amdgcn_sched_barrier(0);
        }
        if (SHAPE == MERGED) {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(gmn < 0x27800000u || gmx >= 0x7f800000u) != 0ull, 0)) {
#pragma unroll
                for (int k = 0; k < 7; k++) r[k] = __builtin_sqrtf(a[k] * a[k] + b);
            }
        }
#pragma unroll
        for (int k = 0; k < 7; k++) {
            acc += r[k];
            a[k] = a[k] * 1.0000001f + 1e-7f;  // keeps the chains alive and different per iteration
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int SHAPE>
void run(float* out, const char* name)
{
    const int iters = 20000, blocks = 256 * 8;  // 8 resident blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) probe<SHAPE><<<blocks, 256>>>(out, iters, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) probe<SHAPE><<<blocks, 256>>>(out, iters, 1.0f + i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double chains = (double)blocks * 4 * iters * 7;  // wave-level chains
    printf("%-7s %.3f ms   %.2f ns per wave-chain   %.3e wave-chains/s\n", name, ms, ms * 1e6 / chains * 1.0, chains / (ms * 1e-3));
}

int main()
{
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    for (int rep = 0; rep < 2; rep++) {
        run<SEQ>(out, "SEQ");
        run<MIXED>(out, "MIXED");
        run<BRANCH>(out, "BRANCH");
        run<MERGED>(out, "MERGED");
    }
    return 0;
}
