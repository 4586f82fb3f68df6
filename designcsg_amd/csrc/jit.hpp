// jit.hpp -- per-scene code object builder (see jit.cpp).
#pragma once
#include <string>

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
bool jit_scene_module(const std::string& scene_cl, const std::string& extra_cl, std::string& hsaco, std::string& log,
                      bool* from_cache, bool relaxed = false, bool spec_set = false);
}  // namespace dcsg
