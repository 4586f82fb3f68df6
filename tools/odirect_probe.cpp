// Can direct I/O beat the page cache for the export's files?  Writes 136 MB in 4 MiB aligned pieces from 1 / 2 / 4 / 8 threads
// with O_DIRECT (falls back to reporting the error), against buffered pwrite of the same pieces.
//   g++ -O2 -pthread tools/odirect_probe.cpp -o /tmp/odirect_probe && /tmp/odirect_probe <dir>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include <cerrno>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    const size_t piece = 4u << 20, total = 136u << 20, n = total / piece;
    void* buf = nullptr;
    if (posix_memalign(&buf, 4096, piece)) return 1;
    memset(buf, 0x5a, piece);
    char path[512];
    snprintf(path, sizeof(path), "%s/odirect_probe_%d.bin", dir, (int)getpid());
    for (int direct = 0; direct < 2; direct++)
        for (unsigned nt : {1u, 2u, 4u, 8u}) {
            int fd = open(path, O_WRONLY | O_CREAT | (direct ? O_DIRECT : 0), 0644);
            if (fd < 0) { printf("%s open: %s\n", direct ? "O_DIRECT" : "buffered", strerror(errno)); continue; }
            if (fallocate(fd, 0, 0, (off_t)total) != 0) perror("fallocate");
            for (int rep = 0; rep < 3; rep++) {
                double a = now();
                std::vector<std::thread> th;
                bool ok = true;
                for (unsigned t = 0; t < nt; t++)
                    th.emplace_back([&, t] {
                        for (size_t i = t; i < n; i += nt)
                            if (pwrite(fd, buf, piece, (off_t)(i * piece)) != (ssize_t)piece) ok = false;
                    });
                for (auto& x : th) x.join();
                double ms = (now() - a) * 1e3;
                if (rep == 2 || !ok) printf("%-9s %u threads: %.1f ms (%.1f GB/s)%s\n", direct ? "O_DIRECT" : "buffered", nt, ms, total / ms / 1e6, ok ? "" : " FAILED");
            }
            close(fd);
        }
    unlink(path);
    return 0;
}
