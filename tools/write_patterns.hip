// Diagnostic micro-benchmark (not part of the product): sustained HBM write bandwidth of the store patterns a
// multi-point-per-thread lattice sweep could use, on a (n1)^3 float lattice.
//   hipcc --offload-arch=gfx950 -O3 tools/write_patterns.hip -o tools/_build/write_patterns && tools/_build/write_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode 0: flat, one dword per thread.
// mode 1: Z planes per thread (stores back to back).     grid.y = plane group, flat tiles inside the plane
// mode 2: Z rows per thread (adjacent in y).             grid.y = plane, items = row groups x n1 inside the plane
// mode 4: one thread = 4 consecutive floats (4 dword stores), flat.
// mode 5: one thread = 4 consecutive floats, one 16-byte store (lattice base is 16-byte aligned), flat.
// All index arithmetic is 32-bit; divisions by n1 use a host-computed multiply-high magic.
struct Div { unsigned magic, shift; };
__device__ __forceinline__ unsigned divu(unsigned n, Div d) { unsigned q = __umulhi(n, d.magic); return (((n - q) >> 1) + q) >> d.shift; }
static Div make_div(unsigned d) { unsigned lg = 0; while ((1ull << lg) < d) lg++; Div r; r.magic = (unsigned)((((1ull << lg) - d) << 32) / d + 1); r.shift = lg ? lg - 1 : 0; return r; }

template <int Z>
__global__ void __launch_bounds__(256) pat(float* out, unsigned n1, unsigned planes, int mode, Div dn1, float seed)
{
    const unsigned plane = n1 * n1;
    float v = seed + (float)threadIdx.x;
    if (mode == 0) {
        const unsigned total = plane * planes;   // < 2^32 for n1 <= 1025
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) out[i] = v;
    } else if (mode == 1) {   // persistent blocks, grid-stride over (plane group, flat position in plane)
        const unsigned groups = (planes + Z - 1) / Z, items = groups * plane;
        for (unsigned it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
            unsigned g = divu(divu(it, dn1), dn1), f = it - g * plane;
#pragma unroll
            for (int k = 0; k < Z; k++) {
                unsigned p = g * Z + k;
                if (p < planes) out[(size_t)p * plane + f] = v + k;
            }
        }
    } else if (mode == 2) {   // persistent blocks, grid-stride over (plane, row group, ix); gy = row groups per plane
        const unsigned gy = (n1 + Z - 1) / Z, items = planes * gy * n1;
        for (unsigned it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
            unsigned t = divu(it, dn1), ix = it - t * n1;
            unsigned pl = t / gy, iy0 = (t - pl * gy) * Z;
#pragma unroll
            for (int k = 0; k < Z; k++)
                if (iy0 + k < n1) out[(size_t)pl * plane + (iy0 + k) * n1 + ix] = v + k;
        }
    } else if (mode == 6) {   // z-march with almost no index arithmetic: a thread owns plane positions f, f+stride, ...
        // and walks the planes of its z-chunk (gridDim.y chunks); Z stores per step, pointer increments only
        const unsigned chunk = (planes + gridDim.y - 1) / gridDim.y, p0 = blockIdx.y * chunk;
        const unsigned p1 = p0 + chunk < planes ? p0 + chunk : planes;
        for (unsigned f = blockIdx.x * 256 + threadIdx.x; f < plane; f += gridDim.x * 256) {
            float* o = out + (size_t)p0 * plane + f;
            for (unsigned p = p0; p < p1; p += Z) {
#pragma unroll
                for (int k = 0; k < Z; k++)
                    if (p + k < p1) o[(size_t)k * plane] = v + k;
                o += (size_t)Z * plane;
            }
        }
    } else if (mode == 4) {
        const unsigned total = plane * planes, quads = (total + 3) / 4;
        for (unsigned q = blockIdx.x * 256 + threadIdx.x; q < quads; q += gridDim.x * 256)
            for (int k = 0; k < 4; k++) if (4 * q + k < total) out[4 * q + k] = v + k;
    } else {
        const unsigned total = plane * planes, quads = total / 4;
        float4* o4 = (float4*)out;
        for (unsigned q = blockIdx.x * 256 + threadIdx.x; q < quads; q += gridDim.x * 256) o4[q] = make_float4(v, v + 1, v + 2, v + 3);
    }
}

template <int Z>
void run(float* d, unsigned n1, int mode, const char* name)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 10;
    dim3 grid(8192, 1);
    if (mode == 6) grid = dim3(512, 16);
    Div dn1 = make_div(n1);
    for (int i = 0; i < 2; i++) pat<Z><<<grid, 256>>>(d, n1, n1, mode, dn1, 1.0f);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) pat<Z><<<grid, 256>>>(d, n1, n1, mode, dn1, 1.0f + i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    double bytes = 4.0 * n1 * n1 * n1;
    printf("n1=%u %-34s Z=%d  grid %ux%u  %.4f ms  %.0f GB/s\n", n1, name, Z, grid.x, grid.y, ms, bytes / ms / 1e6);
}

// The fused sweep's output: 4 B SDF + 12 B normal per lattice point, flat (mode 0) or z-marching (mode 1).
__global__ void __launch_bounds__(256) pat16(float* sdf, float* nrm, unsigned n1, unsigned planes, int mode, float seed)
{
    const unsigned plane = n1 * n1;
    float v = seed + (float)threadIdx.x;
    if (mode == 0) {
        const unsigned total = plane * planes;
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            sdf[i] = v;
            float* o = nrm + (size_t)i * 3;
            o[0] = v + 1; o[1] = v + 2; o[2] = v + 3;
        }
    } else {
        const unsigned chunk = (planes + gridDim.y - 1) / gridDim.y, p0 = blockIdx.y * chunk;
        const unsigned p1 = p0 + chunk < planes ? p0 + chunk : planes;
        for (unsigned f = blockIdx.x * 256 + threadIdx.x; f < plane; f += gridDim.x * 256) {
            size_t i = (size_t)p0 * plane + f;
            for (unsigned p = p0; p < p1; p++, i += plane) {
                sdf[i] = v;
                float* o = nrm + i * 3;
                o[0] = v + 1; o[1] = v + 2; o[2] = v + 3;
            }
        }
    }
}

void run16(float* sdf, float* nrm, unsigned n1, int mode, dim3 grid, const char* name)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 10;
    for (int i = 0; i < 2; i++) pat16<<<grid, 256>>>(sdf, nrm, n1, n1, mode, 1.0f);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) pat16<<<grid, 256>>>(sdf, nrm, n1, n1, mode, 1.0f + i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    double bytes = 16.0 * n1 * n1 * n1;
    printf("n1=%u 16 B/point %-28s grid %ux%u  %.4f ms  %.0f GB/s\n", n1, name, grid.x, grid.y, ms, bytes / ms / 1e6);
}

int main()
{
    for (unsigned n1 : {513u, 1025u}) {
        float* d; CK(hipMalloc(&d, 4ull * n1 * n1 * n1 + 64));
        run<1>(d, n1, 0, "flat dword");
        run<1>(d, n1, 4, "flat, 4 dword stores per thread");
        run<1>(d, n1, 5, "flat, one 16-byte store per thread");
        run<1>(d, n1, 1, "z-run (Z=1: plane-blocked flat)");
        run<2>(d, n1, 1, "z-run");
        run<4>(d, n1, 1, "z-run");
        run<8>(d, n1, 1, "z-run");
        run<1>(d, n1, 6, "z-march, pointer increments only");
        run<2>(d, n1, 6, "z-march, pointer increments only");
        run<4>(d, n1, 6, "z-march, pointer increments only");
        run<8>(d, n1, 6, "z-march, pointer increments only");
        run<1>(d, n1, 2, "y-run (Z=1)");
        run<2>(d, n1, 2, "y-run");
        run<4>(d, n1, 2, "y-run");
        run<8>(d, n1, 2, "y-run");
        CK(hipFree(d));
        float *sd, *nr;
        CK(hipMalloc(&sd, 4ull * n1 * n1 * n1 + 64));
        CK(hipMalloc(&nr, 12ull * n1 * n1 * n1 + 64));
        run16(sd, nr, n1, 0, dim3(8192, 1), "flat");
        run16(sd, nr, n1, 1, dim3(512, 16), "z-march 16 chunks");
        run16(sd, nr, n1, 1, dim3(1029, 8), "z-march 8 chunks");
        run16(sd, nr, n1, 1, dim3(1029, 32), "z-march 32 chunks");
        run16(sd, nr, n1, 1, dim3(256, 32), "z-march 32 chunks");
        CK(hipFree(sd)); CK(hipFree(nr));
    }
    return 0;
}
