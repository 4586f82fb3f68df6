// specialise.cpp -- emits a scene's stack program as straight-line OpenCL C (see kernels.hip,
// "Per-scene specialisation").  The text is appended to the brush source before it goes through the
// OpenCL C front-end, so the compiler sees the whole evaluation: constants fold into operands, the seven
// stencil transforms share sub-expressions, nothing is interpreted.  Every expression is written in the
// operand order of the interpreters (k2.cl:105-111 for the object transform, T_min/T_max as a<b?a:b), with
// literals as exact hexadecimal floats, so results are bit-identical to them.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "scene.hpp"

namespace dcsg {

static std::string hexf(float f)
{
    char buf[64];
    if (f != f) return "NAN";
    if (f - f != 0.0f) return f > 0 ? "INFINITY" : "(-INFINITY)";
    snprintf(buf, sizeof(buf), "(%af)", (double)f);  // exact: hex float literal with f suffix
    return buf;
}

// N stencil points passed as float8 lanes (N = 7) or one point (N = 1).
// threaded: every object evaluation is a CALL of an out-of-line function of its brush (emit_threaded_prelude) with a pointer
// to the object's constants, instead of the transform and the brush written out in place.
static void emit_function(const Scene& s, bool null_brush_const, int N, bool simplify, const char* name1, std::string& out,
                          bool threaded = false)
{
    enum {
        PUSH = 32, ACC_MIN = 33, ACC_MAX = 34, ACC_NEGMAX = 35, POP_MIN = 36, POP_MAX = 37, POP_NEGMAX = 38, EXPORT_TOP = 39,
        SETK = 40, ACC_SMIN = 41, ACC_SMAX = 42, ACC_NEGSMAX = 43, POP_SMIN = 44, POP_SMAX = 45, POP_NEGSMAX = 46
    };
    char line[768];
    std::string kk = "0.0f", ikk = "0.0f";  // literals of the blend radius in force and of its reciprocal
    auto P = [&](const char* fmt, auto... a) {
        snprintf(line, sizeof(line), fmt, a...);
        out += line;
    };
    static const char* hexdigit = "0123456789abcdef";
    // always_inline: whether the evaluation is inlined into a kernel must not depend on how many kernels of the module
    // call it (the stencil points share sub-expressions with their caller only when it is)
    const bool exact_copy = std::string(name1) == "dcsg_spec_eval1x";  // rare path, kept out of line by its caller
    // (the threaded form is a long chain of calls: one copy per module, called)
    const char* inl = threaded ? "__attribute__((noinline)) " : (exact_copy ? "" : "__attribute__((always_inline)) ");
    if (N == 7)
        out += std::string(inl) + "float8 dcsg_spec_eval7(float8 px, float8 py, float8 pz){\n";
    else if (N == 4)
        out += std::string(inl) + "float4 dcsg_spec_eval4(float4 px, float4 py, float4 pz){\n";
    else
        out += std::string(inl) + "float " + name1 + "(float x, float y, float z){\n";
    auto X = [&](int k) { return N > 1 ? std::string("px.s") + hexdigit[k] : std::string("x"); };
    auto Y = [&](int k) { return N > 1 ? std::string("py.s") + hexdigit[k] : std::string("y"); };
    auto Z = [&](int k) { return N > 1 ? std::string("pz.s") + hexdigit[k] : std::string("z"); };
    for (int k = 0; k < N; k++) P("  float o%d = (float)MAX_DISTANCE;\n", k);
    for (int d = 0; d < (s.stack_depth > 0 ? s.stack_depth : 1); d++)
        for (int k = 0; k < N; k++) P("  float s%d_%d = 0.0f;\n", d, k);
    int tmp = 0;
    for (int c = 0; c < s.n_stack_prog; c++) {
        int op = s.stack_prog[4 * c], brush = s.stack_prog[4 * c + 1] & 0xff, obj = s.stack_prog[4 * c + 2],
            depth = s.stack_prog[4 * c + 3];
        if (op == SETK) {
            float k, ik;
            memcpy(&k, &s.stack_prog[4 * c + 1], 4);
            memcpy(&ik, &s.stack_prog[4 * c + 2], 4);
            kk = hexf(k);
            ikk = hexf(ik);
            continue;
        }
        // smooth joins (extension; same expression order as dcsg_smin / dcsg_smax in kernels.hip)
        auto smooth = [&](const std::string& dst, const std::string& a, const std::string& b, bool is_min) {
            P("    { const float sa = %s, sb = %s; float st = %s - fabs(sa - sb); st = st > 0.0f ? st : 0.0f; "
              "const float sh = st * %s; %s = (sa %s sb ? sa : sb) %s ((sh * sh) * %s) * 0.25f; }\n",
              a.c_str(), b.c_str(), kk.c_str(), ikk.c_str(), dst.c_str(), is_min ? "<" : ">", is_min ? "-" : "+", kk.c_str());
        };
        auto S = [&](int d, int k) { return "s" + std::to_string(d) + "_" + std::to_string(k); };
        if ((op >= PUSH && op <= ACC_NEGMAX) || (op >= ACC_SMIN && op <= ACC_NEGSMAX)) {
            int t = tmp++;
            bool obj_simplified = brush == 0 && null_brush_const;  // set below when the object's transform was simplified
            out += "  {\n";
            if (brush == 0 && null_brush_const) {
                for (int k = 0; k < N; k++) P("    const float r%d = (float)MAX_DISTANCE;\n", k);
            } else if (threaded) {
                if (N > 1) {
                    P("    const float%d rr = dcsg_spec_b%d_%d(px, py, pz, dcsg_spec_T + %d);\n", N == 7 ? 8 : 4, brush, N, 12 * obj);
                    for (int k = 0; k < N; k++) P("    const float r%d = rr.s%d;\n", k, k);
                } else {
                    P("    const float r0 = dcsg_spec_b%d_1(x, y, z, dcsg_spec_T + %d);\n", brush, 12 * obj);
                }
            } else {
                const float* o = &s.pos[3 * obj];
                const float* r = &s.right[3 * obj];
                const float* u = &s.up[3 * obj];
                const float* f = &s.fwd[3 * obj];
                P("    const float ox=%s, oy=%s, oz=%s;\n", hexf(o[0]).c_str(), hexf(o[1]).c_str(), hexf(o[2]).c_str());
                P("    const float rx=%s, ry=%s, rz=%s;\n", hexf(r[0]).c_str(), hexf(r[1]).c_str(), hexf(r[2]).c_str());
                P("    const float ux=%s, uy=%s, uz=%s;\n", hexf(u[0]).c_str(), hexf(u[1]).c_str(), hexf(u[2]).c_str());
                P("    const float fx=%s, fy=%s, fz=%s;\n", hexf(f[0]).c_str(), hexf(f[1]).c_str(), hexf(f[2]).c_str());
                // `simplify` (scenes made of the five standard brushes only): terms whose coefficient is exactly zero are
                // left out, "* 1.0f" and "- 0.0f" are dropped.  For finite coordinates (|v| <= 1e30, guaranteed by the
                // callers; anything else takes dcsg_spec_eval1x) such a product is +-0 and adding it changes nothing but,
                // possibly, the sign of a zero sum -- which no standard brush can see (squares, fabs).  Same bits out.
                // The argument needs every product and three-term sum to stay finite (an overflow to +-inf would turn
                // "+ 0 * dx" into NaN in the full form and into nothing in the short one): coordinates are <= 1e30 in
                // magnitude, so coefficients and origins are held to 1e3 -- max |term| <= 1e3 * (1e30 + 1e3) and a sum of
                // three stays below FLT_MAX.  Objects beyond that keep the written-out form.
                auto finite_small = [](float v) { return v - v == 0.0f && v <= 1e3f && v >= -1e3f; };
                bool simp = simplify;
                for (int q = 0; q < 3; q++) simp = simp && finite_small(o[q]) && finite_small(r[q]) && finite_small(u[q]) && finite_small(f[q]);
                obj_simplified = simp;
                auto row = [&](int k, const float* c, const char* cn) {
                    std::string e;
                    const char* d[3] = {"dx", "dy", "dz"};
                    const char* ax[3] = {"x", "y", "z"};
                    for (int q = 0; q < 3; q++) {
                        std::string t = std::string(d[q]) + std::to_string(k);
                        if (simp && c[q] == 0.0f) continue;
                        if (simp && c[q] == -1.0f) t = "(-" + t + ")";
                        else if (!(simp && c[q] == 1.0f)) t += std::string("*") + cn + ax[q];
                        e += e.empty() ? t : " + " + t;
                    }
                    return e.empty() ? std::string("0.0f") : e;
                };
                for (int k = 0; k < N; k++) {
                    const std::string p3[3] = {X(k), Y(k), Z(k)};
                    const char* d[3] = {"dx", "dy", "dz"};
                    const char* on[3] = {"ox", "oy", "oz"};
                    out += "    const float ";
                    for (int q = 0; q < 3; q++) {
                        out += std::string(d[q]) + std::to_string(k) + " = " + p3[q];
                        if (!(simp && o[q] == 0.0f)) out += std::string(" - ") + on[q];
                        out += q < 2 ? ", " : ";\n";
                    }
                    P("    const float r%d = sdf_bank((float3)(%s, %s, %s), (unsigned char)%d);\n", k, row(k, r, "r").c_str(),
                      row(k, u, "u").c_str(), row(k, f, "f").c_str(), brush);
                }
            }
            // `simplify` scenes: right after "PUSH null brush" the accumulator is the constant MAX_DISTANCE and the value
            // joined to it cannot be NaN (standard brushes, finite coordinates), so T_min(64, r) = (64 < r ? 64 : r) is
            // the IEEE minimum: one v_min_f32 instead of compare + select.  Same value for every non-NaN r.
            const bool acc_is_max_distance = simplify && obj_simplified && op == ACC_MIN && c > 0 && s.stack_prog[4 * (c - 1)] == PUSH &&
                                             (s.stack_prog[4 * (c - 1) + 1] & 0xff) == 0 && null_brush_const &&
                                             s.stack_prog[4 * (c - 1) + 3] == depth - 1;
            for (int k = 0; k < N; k++) {
                if (op == PUSH) P("    s%d_%d = r%d;\n", depth, k, k);
                else if (acc_is_max_distance) P("    s%d_%d = __builtin_fminf(r%d, (float)MAX_DISTANCE);\n", depth - 1, k, k);
                else if (op == ACC_MIN) P("    s%d_%d = s%d_%d < r%d ? s%d_%d : r%d;\n", depth - 1, k, depth - 1, k, k, depth - 1, k, k);
                else if (op == ACC_MAX) P("    s%d_%d = s%d_%d > r%d ? s%d_%d : r%d;\n", depth - 1, k, depth - 1, k, k, depth - 1, k, k);
                else if (op == ACC_NEGMAX) P("    { const float nb = -r%d; s%d_%d = s%d_%d > nb ? s%d_%d : nb; }\n", k, depth - 1, k, depth - 1, k, depth - 1, k);
                else if (op == ACC_SMIN) smooth(S(depth - 1, k), S(depth - 1, k), "r" + std::to_string(k), true);
                else if (op == ACC_SMAX) smooth(S(depth - 1, k), S(depth - 1, k), "r" + std::to_string(k), false);
                else smooth(S(depth - 1, k), S(depth - 1, k), "(-r" + std::to_string(k) + ")", false);
            }
            out += "  }\n";
            (void)t;
        } else if ((op >= POP_MIN && op <= POP_NEGMAX) || (op >= POP_SMIN && op <= POP_NEGSMAX)) {
            for (int k = 0; k < N; k++) {
                int a = depth - 2, b = depth - 1;
                if (op == POP_MIN) P("  s%d_%d = s%d_%d < s%d_%d ? s%d_%d : s%d_%d;\n", a, k, a, k, b, k, a, k, b, k);
                else if (op == POP_MAX) P("  s%d_%d = s%d_%d > s%d_%d ? s%d_%d : s%d_%d;\n", a, k, a, k, b, k, a, k, b, k);
                else if (op == POP_NEGMAX) P("  { const float nb = -s%d_%d; s%d_%d = s%d_%d > nb ? s%d_%d : nb; }\n", b, k, a, k, a, k, a, k);
                else if (op == POP_SMIN) smooth(S(a, k), S(a, k), S(b, k), true);
                else if (op == POP_SMAX) smooth(S(a, k), S(a, k), S(b, k), false);
                else smooth(S(a, k), S(a, k), "(-" + S(b, k) + ")", false);
            }
        } else if (op == EXPORT_TOP) {
            for (int k = 0; k < N; k++) P("  o%d = s%d_%d;\n", k, depth - 1, k);
        }
    }
    if (N == 7)
        out += "  return (float8)(o0, o1, o2, o3, o4, o5, o6, 0.0f);\n}\n";
    else if (N == 4)
        out += "  return (float4)(o0, o1, o2, o3);\n}\n";
    else
        out += "  return o0;\n}\n";
}

// The threaded form's constants and brush functions: the objects' twelve numbers each in a __constant table (exact
// hexadecimal literals), and per brush in use one out-of-line function for seven points and one for a single point that
// read an object's numbers through a pointer (uniform: scalar loads), transform the points exactly as the interpreters do
// (k2.cl:105-111, every term written out) and evaluate the brush.  The program itself then is a chain of calls and joins:
// a few instructions per object whatever the brushes cost, so the code a large scene adds to a module -- and the time to
// compile it -- stays small, while nothing is decoded at run time.  Same operations in the same order: same bits.
static void emit_threaded_prelude(const Scene& s, bool null_brush_const, std::string& out)
{
    char line[256];
    out += "__constant float dcsg_spec_T[] = {\n";
    for (int o = 0; o < s.n_objects; o++) {
        const float* v[4] = {&s.pos[3 * o], &s.right[3 * o], &s.up[3 * o], &s.fwd[3 * o]};
        out += "  ";
        for (auto p : v)
            for (int q = 0; q < 3; q++) out += hexf(p[q]) + ", ";
        out += "\n";
    }
    out += "  0.0f};\n";
    bool used[256] = {};
    for (int c = 0; c < s.n_stack_prog; c++) {
        const int op = s.stack_prog[4 * c], brush = s.stack_prog[4 * c + 1] & 0xff;
        if (((op >= 32 && op <= 35) || (op >= 41 && op <= 43)) && !(brush == 0 && null_brush_const)) used[brush] = true;
    }
    const char* load = "  const float ox=T[0], oy=T[1], oz=T[2], rx=T[3], ry=T[4], rz=T[5], ux=T[6], uy=T[7], uz=T[8], fx=T[9], fy=T[10], fz=T[11];\n";
    for (int b = 0; b < 256; b++) {
        if (!used[b]) continue;
        for (int n : {7, 4}) {
            const int w = n == 7 ? 8 : 4;
            snprintf(line, sizeof(line), "__attribute__((noinline)) float%d dcsg_spec_b%d_%d(float%d px, float%d py, float%d pz, __constant float* T){\n", w,
                     b, n, w, w, w);
            out += line;
            out += load;
            snprintf(line, sizeof(line), "  float%d r = (float%d)(0.0f);\n", w, w);
            out += line;
            for (int k = 0; k < n; k++) {
                snprintf(line, sizeof(line),
                         "  { const float dx = px.s%d - ox, dy = py.s%d - oy, dz = pz.s%d - oz;\n"
                         "    r.s%d = sdf_bank((float3)(dx*rx + dy*ry + dz*rz, dx*ux + dy*uy + dz*uz, dx*fx + dy*fy + dz*fz), (unsigned char)%d); }\n",
                         k, k, k, k, b);
                out += line;
            }
            out += "  return r;\n}\n";
        }
        snprintf(line, sizeof(line), "__attribute__((noinline)) float dcsg_spec_b%d_1(float x, float y, float z, __constant float* T){\n", b);
        out += line;
        out += load;
        snprintf(line, sizeof(line),
                 "  const float dx = x - ox, dy = y - oy, dz = z - oz;\n"
                 "  return sdf_bank((float3)(dx*rx + dy*ry + dz*rz, dx*ux + dy*uy + dz*uz, dx*fx + dy*fy + dz*fz), (unsigned char)%d);\n}\n", b);
        out += line;
    }
}

// OpenCL C for dcsg_spec_eval7 / dcsg_spec_eval1.  With `enable` false (or a program that is not
// stack-disciplined) it emits stubs, which only satisfy the linker: the host never selects the
// specialised kernels for such a scene.
std::string specialised_source(const Scene& s, bool null_brush_const, bool enable, bool simplify, bool threaded)
{
    std::string out = "\n/* ---- generated by csrc/specialise.cpp ---- */\n";
    if (!enable || s.n_stack_prog == 0) {
        out += "float8 dcsg_spec_eval7(float8 px, float8 py, float8 pz){ return (float8)(0.0f); }\n";
        out += "float4 dcsg_spec_eval4(float4 px, float4 py, float4 pz){ return (float4)(0.0f); }\n";
        out += "float dcsg_spec_eval1(float x, float y, float z){ return 0.0f; }\n";
        out += "float dcsg_spec_eval1x(float x, float y, float z){ return 0.0f; }\n";
        out += "int dcsg_spec_small(void){ return 0; }\n";
        out += "int dcsg_spec_threaded(void){ return 0; }\n";
        return out;
    }
    if (threaded) {  // large programs: calls instead of inline code (every term written out: eval1 serves any input)
        emit_threaded_prelude(s, null_brush_const, out);
        emit_function(s, null_brush_const, 7, false, "", out, true);
        emit_function(s, null_brush_const, 4, false, "", out, true);   // four points: the SDF-only sweeps' runs
        emit_function(s, null_brush_const, 1, false, "dcsg_spec_eval1", out, true);
        emit_function(s, null_brush_const, 1, false, "dcsg_spec_eval1x", out, true);
    } else {
        emit_function(s, null_brush_const, 7, simplify, "", out);
        emit_function(s, null_brush_const, 1, simplify, "dcsg_spec_eval1", out);
        emit_function(s, null_brush_const, 1, false, "dcsg_spec_eval1x", out);  // every term written out: any input
        out += "float4 dcsg_spec_eval4(float4 px, float4 py, float4 pz){ return (float4)(0.0f); }\n";  // threaded form only
    }
    // dcsg_spec_small: the scene evaluates at most four objects, so the row-invariant part of its seven stencil
    // evaluations fits in registers and the sweep may keep it across the passes of a run (kernels.hip, dcsg_sweep_body)
    int evals = 0;
    for (int c = 0; c < s.n_stack_prog; c++) {
        const int op = s.stack_prog[4 * c], brush = s.stack_prog[4 * c + 1] & 0xff;
        if (((op >= 32 && op <= 35) || (op >= 41 && op <= 43)) && !(brush == 0 && null_brush_const)) evals++;
    }
    out += std::string("__attribute__((always_inline)) int dcsg_spec_small(void){ return ") + (evals <= 4 ? "1" : "0") + "; }\n";
    // dcsg_spec_threaded: the evaluation is a chain of calls -- the SDF-only sweeps then send four points through one chain
    out += std::string("__attribute__((always_inline)) int dcsg_spec_threaded(void){ return ") + (threaded ? "1" : "0") + "; }\n";
    return out;
}

// The scene.cl the scene compiler writes for a design that defines no brushes of its own
// (scenecompiler.py:424-425 + DesignCSG.py:9-22), reduced to what the export path needs.
const char* builtin_scene_cl()
{
    return "float sd0( float3 v){ return MAX_DISTANCE; }\n"
           "float sd1( float3 v){ return 0.0; }\n"
           "float sd2( float3 v){ return length(v)-0.5; }\n"
           "float sd3( float3 v){ v=fabs(v); float x = length((float2)(v.x,v.z)); float y = v.y; return max(x-0.5,y-0.5); }\n"
           "float sd4( float3 v){ v=fabs(v); return max(v.x-0.5,max(v.y-0.5,v.z-0.5)); }\n"
           "float sdf_bank(float3 v, unsigned char shape_id){\n"
           "  switch(shape_id){ case 0: return sd0(v); case 1: return sd1(v); case 2: return sd2(v);\n"
           "                    case 3: return sd3(v); case 4: return sd4(v); }\n"
           "  return 0.0;\n}\n"
           // the two materials every scene starts with (scenecompiler.py:426-427), for the preview
           "float3 shader0 (float3 gv, float3 lv, float3 n){ return fabs(n); }\n"
           "float3 shader1 (float3 gv, float3 lv, float3 n){ float3 n_g = n.x*rgt_g+n.y*upp_g+n.z*fwd_g;\n"
           "  float L = dot(n_g,(float3)(0.0,0.0,-1.0)); return (float3)(L,L,L); }\n"
           "float3 shader_bank(float3 gv, float3 lv, float3 n, unsigned char material_id){\n"
           "  switch(material_id){ case 0: return shader0(gv,lv,n); case 1: return shader1(gv,lv,n); }\n"
           "  return (float3)(1.0, 1.0, 1.0);\n}\n";
}

}  // namespace dcsg
