// What a first export pays for its buffers: hipMalloc / hipHostMalloc / hipFree times by size (ms).
//   hipcc --offload-arch=gfx950 tools/alloc_probe.hip -o /tmp/alloc_probe && /tmp/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(nullptr);
    for (int rep = 0; rep < 2; rep++) {
        for (size_t mb : {1, 16, 64, 256, 1024, 4300}) {
            void* p = nullptr;
            double a = now();
            hipMalloc(&p, mb << 20);
            double b = now();
            hipMemset(p, 0, 4096);
            hipDeviceSynchronize();
            double c = now();
            hipFree(p);
            double d = now();
            printf("hipMalloc %5zu MiB: %.3f ms (+ first touch %.3f), hipFree %.3f\n", mb, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3);
        }
        for (size_t mb : {1, 4, 16, 64}) {
            void* p = nullptr;
            double a = now();
            hipHostMalloc(&p, mb << 20, hipHostMallocDefault);
            double b = now();
            hipHostFree(p);
            double c = now();
            printf("hipHostMalloc %3zu MiB: %.3f ms, hipHostFree %.3f\n", mb, (b - a) * 1e3, (c - b) * 1e3);
        }
    }
    return 0;
}
