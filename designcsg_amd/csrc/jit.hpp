// jit.hpp -- per-scene code object builder (see jit.cpp).
#pragma once
#include <string>
#include <vector>

namespace dcsg {
std::string library_dir();
bool read_file(const std::string& path, std::string& out);
// scene_cl: the generated scene.cl text.  On success hsaco holds a gfx950 code object containing
// every kernel of kernels.hip linked against the scene's sdf_bank.  On failure log holds the
// compiler diagnostics (what Evaluator::build returns as the build log, Evaluator.cpp:64-89).
// extra_cl: generated OpenCL C appended after the epilogue (csrc/specialise.cpp).
// relaxed: the opt-in relaxed-numerics module (kernels.hip built with DCSG_RELAXED: hardware sqrt / rsq, FMA
// contraction, no IEEE division sequences) instead of the bit-exact one.
// spec_set: link the lean kernel set of a specialised scene (the _sp kernels + mesher) instead of the interpreters'.
// part: every kernel set exists in two parts, each its own code object (kernels.hip, DCSG_PART_*): JIT_PART_EXPORT =
// what dcsg_export launches (stage A, node-wise sweep, mesher, projection, de-duplication), JIT_PART_REST = dense
// sweeps, point evaluation, T-junction repair, test kernels, preview.  A cold build compiles the export part first and
// the rest while the export already runs (dcsg_api.cpp: load_rest_part).
// cache_only: never compile; returns false (log empty) when the code object is not in the cache.
enum { JIT_PART_EXPORT = 0, JIT_PART_REST = 1 };
// The JIT's environment switches, read ONCE on the caller's thread: a background build must not call getenv while the
// host (or a test) may call setenv -- a libc data race.  env == nullptr: read the environment now.
struct JitEnv {
    std::string extra_opts;  // DCSG_JIT_EXTRA_OPTS
    std::string cache_dir;   // the resolved cache directory; empty = no cache (DCSG_JIT_NO_CACHE, or no usable directory)
    bool profile = false;    // DCSG_JIT_PROFILE
    std::vector<std::string> comgr_paths;  // where libamd_comgr is looked for (DCSG_COMGR_PATH, ROCM_PATH, defaults)
    static JitEnv from_environment();
};
bool jit_scene_module(const std::string& scene_cl, const std::string& extra_cl, std::string& hsaco, std::string& log,
                      bool* from_cache, bool relaxed = false, bool spec_set = false, int part = JIT_PART_EXPORT,
                      bool cache_only = false, const JitEnv* env = nullptr);
}  // namespace dcsg
