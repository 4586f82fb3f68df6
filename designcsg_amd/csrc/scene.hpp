// scene.hpp -- host-side parsing of the compiled scene format (the Python <-> native contract).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dcsg.h"

namespace dcsg {

// scene.txt + buildprocedure.txt as BasicDrawPane::loadScene reads them (DrawPane.cpp:267-371),
// validated (the reference has no bounds checks) and packed into
// the blob the kernels stage into LDS.
struct Scene {
    int n_objects = 0, n_steps = 0, n_slots = 0;
    std::vector<uint8_t> shape;                // [n_objects]
    std::vector<uint8_t> material;             // [n_objects] second column of scene.txt (preview only)
    std::vector<float> pos, right, up, fwd;    // [3*n_objects]
    std::vector<int32_t> steps;                // [4*n_steps] opcode,left,right,dest  (as in buildprocedure.txt)
    bool invalid_program = false;              // optimise_program met a slot outside 0..DCSG_MAX_SLOTS-1 (cannot happen after parse_scene)
    std::vector<int32_t> prog;                 // internal program: fused + slots renamed (optimise_program)
    int n_prog = 0;                            // steps in prog
    int n_regs = 0;                            // stack registers prog needs (<= n_slots)
    std::vector<int32_t> stack_prog;           // stack program (kernels.hip SOP_*), empty if prog does not nest
    int n_stack_prog = 0, stack_depth = 0;
    // Top-level items of the stack program for the tile culling (kernels.hip): {first step, step count, kind, 0};
    // kind = the accumulate opcode a leaf step applies (33..35), the POP opcode a group is joined with (36..38),
    // 32 for the root PUSH, 39 for EXPORT_TOP.  Empty when the program is deeper than one group level.
    std::vector<int32_t> items;
    int n_items = 0;
    int max_brush_id = 0;                      // over IMPORT steps
    bool has_smooth = false;                   // uses the extension opcodes SMOOTH_MIN / SMOOTH_MAX (6, 7; SET_K 8)
    std::vector<uint32_t> blob;                // uint4-granular, see kernels.hip "Scene tables in LDS"
    int blob_vec4() const { return (int)(blob.size() / 4); }
    size_t lds_bytes() const;                  // blob + interpreter stack
};

bool parse_scene(const char* scene_txt, const char* buildprocedure_txt, Scene& out, std::string& err);
// Value-preserving rewrite of the build procedure for the device interpreters:
//  * "IMPORT b,o -> R; MIN a R -> a" (also MAX, and NEGATE R -> R; MAX a R -> a) with R dead afterwards
//    becomes one accumulate step (opcodes 16/17/18 of kernels.hip): same compare, same operands;
//  * stack slots are renamed by liveness so that the program needs as few registers as possible.
void optimise_program(Scene& s);
void build_blob(Scene& s, bool null_brush_is_const);

// True when scene_cl defines sd0..sd4 exactly as the compiler always does
// (scenecompiler.py:424-425, DesignCSG.py:9-22) and has no preprocessor lines that could change
// their meaning: then the HIP-native brush bank applies and nothing needs compiling.
bool scene_cl_is_builtin_only(const std::string& scene_cl);
// True when sd0's body is `return MAX_DISTANCE;` and MAX_DISTANCE is not redefined.
bool scene_cl_null_brush_is_const(const std::string& scene_cl);

bool parse_export_config(const char* text, dcsg_export_config& out, std::string& err);

// lookupTable.txt (readLookupTable.hpp:32-76 + getIndexTriangleStrip, geometry.hpp:228-248) expanded to
// table[case*16] = triangle count, then 3 edge ids per triangle.
bool parse_lookup_table(const char* text, std::vector<uint8_t>& table, int& total_triangles, std::string& err);

// Largest float q with (float)acos((double)q) > threshold, or -2 if none: folds
// Vector3f::angleBetweenVectors' acosf (geometry.hpp:118-124) + the comparison of mesh.hpp:253
// into one device-side compare.
float angle_threshold_to_q_star(float threshold);

// csrc/specialise.cpp
// simplify: the scene uses only the five standard brushes (see specialise.cpp for what that licenses)
// threaded: object evaluations as calls of per-brush functions (large programs, csrc/specialise.cpp: emit_threaded_prelude)
std::string specialised_source(const Scene& s, bool null_brush_const, bool enable, bool simplify, bool threaded = false);
const char* builtin_scene_cl();

uint64_t fnv1a64(const void* data, size_t n, uint64_t seed);

}  // namespace dcsg
