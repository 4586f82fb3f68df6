// Where the PLY writer's time goes: the formatting loop of dcsg_write_ply (float -> double vertices, face records) into
// (a) pre-faulted anonymous memory, (b) a fresh shared file mapping (page faults included), (c) the same mapping again
// (pages already mapped), for several thread counts.   g++ -O2 -pthread tools/fmt_probe.cpp -o /tmp/fmt_probe && /tmp/fmt_probe <dir>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void format(unsigned char* vb, unsigned char* fb, const float* tris, uint64_t t0, uint64_t t1)
{
    unsigned char* o = vb + t0 * 72;
    for (uint64_t i = t0 * 9; i < t1 * 9; i++, o += 8) {
        double v = (double)tris[i];
        memcpy(o, &v, 8);
    }
    unsigned char* q = fb + t0 * 13;
    for (uint64_t t = t0; t < t1; t++, q += 13) {
        q[0] = 3;
        uint32_t idx[3] = {(uint32_t)(3 * t), (uint32_t)(3 * t + 1), (uint32_t)(3 * t + 2)};
        memcpy(q + 1, idx, 12);
    }
}

static double run(unsigned char* base, const float* tris, uint64_t n, unsigned nt)
{
    double a = now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back([=] { format(base, base + n * 72, tris, n * t / nt, n * (t + 1) / nt); });
    for (auto& x : th) x.join();
    return (now() - a) * 1e3;
}

int main(int argc, char** argv)
{
    const uint64_t n = 1600000, bytes = n * 85;
    std::vector<float> tris(n * 9);
    for (size_t i = 0; i < tris.size(); i++) tris[i] = (float)(i % 1000) * 0.01f;
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    for (unsigned nt : {1u, 2u, 4u, 8u, 16u}) {
        unsigned char* anon = (unsigned char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        double a0 = run(anon, tris.data(), n, nt), a1 = run(anon, tris.data(), n, nt);
        munmap(anon, bytes);
        char path[512];
        snprintf(path, sizeof(path), "%s/fmt_probe_%d.bin", dir, (int)getpid());
        int fd = open(path, O_RDWR | O_CREAT, 0644);
        if (fallocate(fd, 0, 0, (off_t)bytes) != 0) perror("fallocate");
        unsigned char* m = (unsigned char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double f0 = run(m, tris.data(), n, nt), f1 = run(m, tris.data(), n, nt);
        munmap(m, bytes);
        m = (unsigned char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   // the file's pages are in the page cache now
        double f2 = run(m, tris.data(), n, nt);
        munmap(m, bytes);
        m = (unsigned char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
        double p0 = now();
        madvise(m, bytes, MADV_POPULATE_WRITE);
        double pop = (now() - p0) * 1e3;
        double f3 = run(m, tris.data(), n, nt);
        munmap(m, bytes);
        close(fd);
        unlink(path);
        printf("%2u threads: anonymous first %.1f again %.1f | new file %.1f, same mapping again %.1f, remapped (cached pages) %.1f | populate-write %.1f then format %.1f ms\n",
               nt, a0, a1, f0, f1, f2, pop, f3);
    }
    return 0;
}
