// Thread-sanitizer harness for the host side of libdcsg_hip (tools/sanitize_host.sh --thread): everything the library does
// on more than one thread that does not need a GPU.
//   1. two host threads JIT the same scene and two different scenes at once into a cold cache (comgr loader, cache files)
//   2. the background build of dcsg_load_scene: started, abandoned by an "early return", joined by the next load / destroy,
//      while the calling thread changes the JIT's environment switches between calls
//   3. the writers' internal threads (dcsg_write_stl / _ply), and 2-8 caller threads writing byte ranges of one file pair
// Exits non-zero when a call fails; the sanitizer reports races on stderr (and makes the exit code non-zero).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/dcsg.h"

static std::string slurp(const std::string& p)
{
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
struct SceneFiles {
    std::string st, bp, cl;
};
static SceneFiles scene(const std::string& root, const char* name)
{
    const std::string d = root + "/tests/golden/scenes/" + name + "/";
    return {slurp(d + "scene.txt"), slurp(d + "buildprocedure.txt"), slurp(d + "scene.cl")};
}
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            exit(2);                                                         \
        }                                                                    \
    } while (0)

int main(int argc, char** argv)
{
    CHECK(argc >= 3);
    const std::string root = argv[1], tmp = argv[2];
    setenv("DCSG_JIT_CACHE", (tmp + "/cache").c_str(), 1);
    const SceneFiles sphere = scene(root, "sphere"), design1 = scene(root, "design1"), prefabs = scene(root, "prefabs"),
                     ct = scene(root, "const_table");

    // 1. concurrent JIT: the same scene twice, then two different ones, all into the cold cache
    auto prebuild = [](const SceneFiles* s) {
        char log[4096];
        int cached = 0;
        CHECK(dcsg_jit_prebuild_scene(s->st.c_str(), s->bp.c_str(), s->cl.c_str(), log, sizeof(log), &cached) == 0);
    };
    {
        std::thread a(prebuild, &sphere), b(prebuild, &sphere);
        a.join();
        b.join();
        std::thread c(prebuild, &design1), d(prebuild, &prefabs);
        c.join();
        d.join();
    }
    puts("1. concurrent prebuilds done");

    // 2. background build: abandon, churn the environment, load the next scene, destroy
    {
        void* slot = nullptr;
        char log[4096];
        setenv("DCSG_JIT_CACHE", (tmp + "/cache2").c_str(), 1);  // cold again
        CHECK(dcsg_debug_background_build(&slot, ct.st.c_str(), ct.bp.c_str(), ct.cl.c_str(), 1, log, sizeof(log)) == 0);  // returns, still building
        setenv("DCSG_JIT_EXTRA_OPTS", "-DTSAN_CHURN=1", 1);
        setenv("DCSG_JIT_PROFILE", "1", 1);
        unsetenv("DCSG_JIT_PROFILE");
        unsetenv("DCSG_JIT_EXTRA_OPTS");
        CHECK(dcsg_debug_background_build(&slot, design1.st.c_str(), design1.bp.c_str(), design1.cl.c_str(), 0, log, sizeof(log)) == 0);
        CHECK(dcsg_debug_background_build(&slot, sphere.st.c_str(), sphere.bp.c_str(), sphere.cl.c_str(), 1, log, sizeof(log)) == 0);
        setenv("DCSG_JIT_NO_CACHE", "1", 1);
        unsetenv("DCSG_JIT_NO_CACHE");
        CHECK(dcsg_debug_background_build(&slot, nullptr, nullptr, nullptr, 2, log, sizeof(log)) == 0);  // destroy joins
        // two contexts at once, as dcsg_export_multi's hosts have them
        void *s1 = nullptr, *s2 = nullptr;
        setenv("DCSG_JIT_CACHE", (tmp + "/cache3").c_str(), 1);
        std::thread a([&] {
            char l[4096];
            CHECK(dcsg_debug_background_build(&s1, prefabs.st.c_str(), prefabs.bp.c_str(), prefabs.cl.c_str(), 0, l, sizeof(l)) == 0);
        });
        std::thread b([&] {
            char l[4096];
            CHECK(dcsg_debug_background_build(&s2, ct.st.c_str(), ct.bp.c_str(), ct.cl.c_str(), 0, l, sizeof(l)) == 0);
        });
        a.join();
        b.join();
        CHECK(dcsg_debug_background_build(&s1, nullptr, nullptr, nullptr, 2, log, sizeof(log)) == 0);
        CHECK(dcsg_debug_background_build(&s2, nullptr, nullptr, nullptr, 2, log, sizeof(log)) == 0);
    }
    puts("2. background builds done");

    // 3. writers
    {
        const uint64_t n = 300000;
        std::vector<float> tris(n * 9);
        for (size_t i = 0; i < tris.size(); i++) tris[i] = (float)((i * 2654435761u) % 100003) * 1e-3f;
        const std::string stl = tmp + "/w.stl", ply = tmp + "/w.ply";
        CHECK(dcsg_write_stl(stl.c_str(), tris.data(), n) == 0);
        CHECK(dcsg_write_ply(ply.c_str(), tris.data(), n) == 0);
        for (int threads : {2, 5, 8}) {
            const std::string rs = tmp + "/r.stl", rp = tmp + "/r.ply";
            CHECK(dcsg_write_stl_range(rs.c_str(), nullptr, 0, 0, n, DCSG_WRITE_BEGIN) == 0);
            CHECK(dcsg_write_ply_range(rp.c_str(), nullptr, 0, 0, n, DCSG_WRITE_BEGIN) == 0);
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; t++)
                pool.emplace_back([&, t] {
                    const uint64_t lo = n * t / threads, hi = n * (t + 1) / threads;
                    CHECK(dcsg_write_stl_range(rs.c_str(), tris.data() + lo * 9, hi - lo, lo, n, 0) == 0);
                    CHECK(dcsg_write_ply_range(rp.c_str(), tris.data() + lo * 9, hi - lo, lo, n, 0) == 0);
                });
            for (auto& t : pool) t.join();
            CHECK(dcsg_write_stl_range(rs.c_str(), nullptr, 0, 0, n, DCSG_WRITE_END) == 0);
            CHECK(dcsg_write_ply_range(rp.c_str(), nullptr, 0, 0, n, DCSG_WRITE_END) == 0);
            CHECK(slurp(rs) == slurp(stl) && slurp(rp) == slurp(ply));
        }
    }
    puts("3. writers done");
    return 0;
}
