// Diagnostic micro-benchmark (not part of the product), round 3: sustained HBM write bandwidth of the store patterns the
// lattice sweeps use or could use, measured the way bench.py measures the sweeps -- after a settle phase of back-to-back
// launches (the r02 table was taken from idle clocks and under-read every pattern by 15-25 %), with plain and with
// non-temporal stores, in the product's launch shape (256-thread blocks, grid-stride over 1024-point tiles, 64 blocks
// per CU; or a given number of tiles per block).
//   hipcc --offload-arch=gfx950 -O3 tools/write_patterns2.hip -o tools/_build/write_patterns2 && tools/_build/write_patterns2
//
// Patterns (x fastest, flat index; a wave owns runs of consecutive points):
//   sdf1    4 B/pt   one dword per lane per pass                      (dcsg_sweep_*_d today)
//   sdf4    4 B/pt   lane owns 4 consecutive points: one dwordx4      (1024 contiguous bytes per instruction)
//   f1     16 B/pt   dword SDF + 3 dword normal stores at stride 12   (dcsg_sweep_*_f today)
//   f4     16 B/pt   lane owns 4 consecutive points: dwordx4 SDF + 3 x dwordx4 normals (48 contiguous bytes per lane)
//   f4t    16 B/pt   as f4, normals transposed across the wave so that every instruction writes 1024 contiguous bytes
//                    (what an LDS transpose in front of the stores would produce)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT, class T>
__device__ __forceinline__ void st(T v, T* p)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

enum { SDF1, SDF4, F1, F4, F4T };

// tile = 1024 points (4 waves x 256): MODE *1: 4 passes of 64 lanes; MODE *4: one pass, 4 points per lane
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) pat(float* __restrict__ sdf, float* __restrict__ nrm, unsigned n_points, float seed)
{
    const unsigned tiles = (n_points + 1023u) / 1024u;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float v = seed + (float)threadIdx.x;
    for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const unsigned base = tile * 1024u + wave * 256u;
        if (base + 256u > n_points) continue;  // the ragged tail is left out (0.0008 % of the points)
        if (MODE == SDF1 || MODE == F1) {
#pragma unroll
            for (unsigned k = 0; k < 4; k++) {
                const unsigned i = base + k * 64u + lane;
                st<NT>(v, sdf + i);
                if (MODE == F1) {
                    float* o = nrm + (size_t)i * 3;
                    st<NT>(v + 1, o);
                    st<NT>(v + 2, o + 1);
                    st<NT>(v + 3, o + 2);
                }
            }
        } else {
            const unsigned i = base + 4u * lane;
            f4 s = {v, v + 1, v + 2, v + 3};
            // the lattice base is 16-byte aligned and base is a multiple of 4 floats
            st<NT>(s, (f4*)(sdf + i));
            if (MODE == F4) {
                f4* o = (f4*)(nrm + (size_t)i * 3);
                st<NT>(s, o);
                st<NT>(s, o + 1);
                st<NT>(s, o + 2);
            } else if (MODE == F4T) {
                f4* o = (f4*)(nrm + (size_t)base * 3) + lane;
                st<NT>(s, o);
                st<NT>(s, o + 64);
                st<NT>(s, o + 128);
            }
        }
        v += 1.0f;
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int MODE, bool NT>
void run(float* sdf, float* nrm, unsigned n1, unsigned blocks_per_cu, const char* name)
{
    const unsigned n = n1 * n1 * n1, tiles = (n + 1023u) / 1024u;
    unsigned grid = blocks_per_cu * 256u;
    if (grid > tiles) grid = tiles;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // settle: 0.25 s of launches, as bench.py does before it times anything
    double t_end = now() + 0.25;
    while (now() < t_end) {
        for (int i = 0; i < 8; i++) pat<MODE, NT><<<grid, 256>>>(sdf, nrm, n, 1.0f);
        CK(hipDeviceSynchronize());
    }
    const int reps = 30;
    float total = 0;
    for (int i = 0; i < reps; i++) {   // an event pair around each launch, like bench.py's kernel time
        CK(hipEventRecord(a));
        pat<MODE, NT><<<grid, 256>>>(sdf, nrm, n, 1.0f + i);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        total += ms;
    }
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) pat<MODE, NT><<<grid, 256>>>(sdf, nrm, n, 1.0f + i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms_b2b; CK(hipEventElapsedTime(&ms_b2b, a, b));
    ms_b2b /= reps;
    const float ms = total / reps;
    const double bytes = ((MODE == SDF1 || MODE == SDF4) ? 4.0 : 16.0) * n;
    printf("n1=%-4u %-6s %-3s %3u blocks/CU  per-launch %.4f ms %5.0f GB/s (%.3f of 8 TB/s)   back-to-back %.4f ms %5.0f GB/s (%.3f)\n",
           n1, name, NT ? "nt" : "", blocks_per_cu, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0, ms_b2b, bytes / ms_b2b / 1e6,
           bytes / ms_b2b / 1e6 / 8000.0);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main(int argc, char** argv)
{
    for (unsigned n1 : {513u, 1025u}) {
        float *sdf, *nrm;
        const size_t n = (size_t)n1 * n1 * n1;
        CK(hipMalloc(&sdf, 4 * n + 4096));
        CK(hipMalloc(&nrm, 12 * n + 4096));
        for (unsigned bpc : {64u, 16u, 256u}) {
            if (bpc != 64u && n1 == 1025u && argc < 2) continue;
            run<SDF1, false>(sdf, nrm, n1, bpc, "sdf1");
            run<SDF1, true>(sdf, nrm, n1, bpc, "sdf1");
            run<SDF4, false>(sdf, nrm, n1, bpc, "sdf4");
            run<SDF4, true>(sdf, nrm, n1, bpc, "sdf4");
            run<F1, false>(sdf, nrm, n1, bpc, "f1");
            run<F1, true>(sdf, nrm, n1, bpc, "f1");
            run<F4, false>(sdf, nrm, n1, bpc, "f4");
            run<F4, true>(sdf, nrm, n1, bpc, "f4");
            run<F4T, false>(sdf, nrm, n1, bpc, "f4t");
            run<F4T, true>(sdf, nrm, n1, bpc, "f4t");
        }
        CK(hipFree(sdf)); CK(hipFree(nrm));
    }
    return 0;
}
