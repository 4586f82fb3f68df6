// kernels.hip -- gfx950 (CDNA4) device code of the DesignCSG export path.
//
// Replaces the reference's OpenCL kernel `k2` (k2.cl:234-280) and the host-side
// lattice cache / octree mesher / projection loop that pulled samples through it
// (ISV.hpp, cms/main/Headers/mesh.hpp) with:
//   dcsg_sweep            fused SDF (+ central-difference normal) over a dense lattice slab
//   dcsg_eval_points      k2-equivalent evaluation at arbitrary points
//   dcsg_bbox             stage-A bounding-box probe (DesignCSG.cpp:668-712)
//   dcsg_mesh_classify /  level-synchronous octree mesher (mesh.hpp:148-311) with
//   dcsg_scan_blocks  /   deterministic prefix-sum compaction of children and triangles
//   dcsg_mesh_emit
//   dcsg_project          vertex projection ("gradient descent", mesh.hpp:531-593)
//
// Built twice by designcsg_amd/build.py:
//   * stand-alone code object with the five built-in brushes written in HIP
//   * device bitcode with DCSG_EXTERN_BANK, linked at scene-load time against the
//     user's scene.cl (OpenCL C brush bodies) by csrc/jit.cpp
//
// Numerics contract (tests/ compare bit-for-bit with oracle/): IEEE binary32/64,
// -ffp-contract=off, correctly rounded sqrt and divide, the operation order of
// the reference expressions.  No MFMA: there is no dense contraction here.
//
// wave = 64 lanes; blocks are 256 threads (4 waves, one per SIMD).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dcsg_device_params.h"

// Scene modules are built in two parts (csrc/jit.cpp), so that a cold JIT compiles first what an export launches:
// DCSG_PART_EXPORT = stage A, node-wise sweep, mesher, projection, de-duplication; DCSG_PART_REST = dense sweeps, point
// evaluation, T-junction repair, test kernels, preview.  Neither defined (the stand-alone module): everything.
#if !defined(DCSG_PART_EXPORT) && !defined(DCSG_PART_REST)
#define DCSG_PART_EXPORT 1
#define DCSG_PART_REST 1
#endif

// k2.cl:2-4,12-17
#define K2_MAX_DISTANCE 64.0
#define K2_NORMAL_EPSILON 0.005
#define OP_IMPORT 0
#define OP_EXPORT 1
#define OP_MIN 2
#define OP_MAX 3
#define OP_NEGATE 4
#define OP_IDENTITY 5
// Extension opcodes (not in the reference; DESIGN.md section 11, oracle/dcsg_oracle.c OP_SMOOTH_*): smooth joins
// with the blend radius loaded by SET_K {k bits, bits of 1.0f/k}.
#define OP_SMOOTH_MIN 6
#define OP_SMOOTH_MAX 7
#define OP_SET_K 8

// ---------------------------------------------------------------------------
// Correctly rounded sqrtf from the reciprocal square root: y = v_rsq_f32(x) (1 ulp), g = x y, h = y / 2, then ONE
// residual correction g + (x - g g) h with both steps as FMAs.  tools/rsq_sqrt_probe.hip compares it with IEEE sqrtf
// for every positive normal float on gfx950: no difference for any x >= 2^-102 (the residual is exact there and the
// hardware's rsq is accurate enough that the corrected value never lands on the wrong side of a rounding boundary);
// below that the result can be off by an ulp, and for 0 / subnormal / inf / negative / NaN inputs (where rsq or x y is
// inf or NaN) it is NaN or inf -- so a wave in which some lane's first estimate g is not a finite number >= 2^-48
// (covers every x < 2^-96 and all of those) takes the compiler's full IEEE sequence instead.  Issue cost measured on the chip (tools/instr_cost_probe.hip): rsq 2.4 fma
// slots + 2 multiplies + 2 FMAs + the guard (integer add + compare), against 2.4 + nine instructions (two integer adds, two FMAs, three
// compares, two selects -- compares and selects cost 1.4 fma slots each) for the v_sqrt_f32 + neighbour test form the
// compiler uses and rounds 1-2 shipped.  Eight square roots per lattice point in the fused sweep.
// tests/test_gpu_parity.py::test_sqrt_all_float_inputs checks all 2^32 bit patterns against IEEE sqrtf.
// ---------------------------------------------------------------------------
#ifdef DCSG_RELAXED
// Opt-in relaxed numerics (dcsg_set_numerics(ctx, DCSG_NUMERICS_RELAXED); built as a separate bitcode with FMA
// contraction allowed): the hardware square root (1 ulp), normalize through v_rsq_f32, no double-precision scale of
// the gradient.  Results are NOT bit-identical to the reference arithmetic; tests hold them to 1e-5.
__device__ __forceinline__ float dcsg_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
__device__ __forceinline__ float dcsg_sqrt(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    float d = __builtin_fmaf(-g, g, x);  // x - g*g, exact for x >= 2^-102
    float r = __builtin_fmaf(d, h, g);
    // rare: some lane is below 2^-96 (or zero, subnormal, inf, negative, NaN): override with the compiler's full sequence
    // (the estimate g is a finite number >= 2^-48 exactly when x is an ordinary input: tiny x gives a smaller g, zeros,
    // +inf, negative numbers and NaNs give NaN, subnormals of either sign +inf; one unsigned range test on its bits)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__float_as_uint(g) - 0x27800000u < 0x58000000u)) != 0ull, 0)) r = __builtin_sqrtf(x);
    return r;
}
#endif

// N square roots with ONE guard: the fast form for all of them, the range tests OR-ed together (scalar unit), one branch.
// The HIP-native brush bank evaluates the seven stencil points of an object side by side, so this replaces seven
// compare-and-branch pairs per object by seven compares and one branch and leaves the seven chains free to overlap.
template <int N>
__device__ __forceinline__ void dcsg_sqrt_n(const float (&x)[N], float (&r)[N])
{
#ifdef DCSG_RELAXED
#pragma unroll
    for (int k = 0; k < N; k++) r[k] = dcsg_sqrt(x[k]);
#else
    bool bad = false;
#pragma unroll
    for (int k = 0; k < N; k++) {
        float y = __builtin_amdgcn_rsqf(x[k]);
        float g = x[k] * y;
        float h = 0.5f * y;
        float d = __builtin_fmaf(-g, g, x[k]);
        r[k] = __builtin_fmaf(d, h, g);
        bad = bad || !(__float_as_uint(g) - 0x27800000u < 0x58000000u);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0ull, 0)) {
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = __builtin_sqrtf(x[k]);
    }
#endif
}

// ---------------------------------------------------------------------------
// wave / block reductions and scans
// ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_min(int v)
{
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(v, o, DCSG_WAVE);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(v, o, DCSG_WAVE);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ unsigned int wave_sum(unsigned int v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, DCSG_WAVE);
    return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ unsigned int wave_scan_incl(unsigned int v)
{
    const int lane = threadIdx.x & (DCSG_WAVE - 1);
    for (int o = 1; o < DCSG_WAVE; o <<= 1) {
        unsigned int t = __shfl_up(v, o, DCSG_WAVE);
        if (lane >= o) v += t;
    }
    return v;
}

// ---------------------------------------------------------------------------
// Brush bank
// ---------------------------------------------------------------------------
#ifdef DCSG_EXTERN_BANK
// OpenCL C float3 is a 16-byte ext-vector, not HIP's 12-byte struct.
typedef float dcsg_cl_float3 __attribute__((ext_vector_type(3)));
extern "C" __device__ float sdf_bank(dcsg_cl_float3 v, unsigned char shape_id);  // generated scene.cl
extern "C" __device__ void dcsg_scene_set_ad(__attribute__((address_space(1))) float* p);  // scene_prelude.cl
extern "C" __device__ float dcsg_sqrt_ieee(float x);                                        // scene_prelude.cl: the brushes' sqrt / length
// preview only: the scene's materials and the camera basis they may read (scene_epilogue.cl)
extern "C" __device__ dcsg_cl_float3 shader_bank(dcsg_cl_float3 gv, dcsg_cl_float3 lv, dcsg_cl_float3 n, unsigned char material_id);
extern "C" __device__ void dcsg_scene_set_camera(dcsg_cl_float3 r, dcsg_cl_float3 u, dcsg_cl_float3 f);
__device__ __forceinline__ float dcsg_bank(float x, float y, float z, int shape_id)
{
    dcsg_cl_float3 v = {x, y, z};
    return sdf_bank(v, (unsigned char)shape_id);
}
// Per-scene specialisation (csrc/specialise.cpp): the scene's stack program emitted as straight-line
// OpenCL C next to the brush text -- object constants as literals, one variable per stack level and
// stencil point -- for 7 points (elements 0..6 of the float8s) and for 1 point.  Same arithmetic in the same
// order as the interpreters, so the results are bit-identical; no decode, no tables, no LDS.
#ifndef DCSG_SET_INTERP
typedef float dcsg_cl_float8 __attribute__((ext_vector_type(8)));
extern "C" __device__ dcsg_cl_float8 dcsg_spec_eval7(dcsg_cl_float8 px, dcsg_cl_float8 py, dcsg_cl_float8 pz);
extern "C" __device__ float dcsg_spec_eval1(float x, float y, float z);
// dcsg_spec_eval1 with every term of every transform written out: valid for any input (the three above may
// leave out terms with a zero coefficient, which is exact for finite coordinates only -- specialise.cpp)
extern "C" __device__ float dcsg_spec_eval1x(float x, float y, float z);
extern "C" __device__ int dcsg_spec_small();  // 1: few objects, row-invariant stencil terms may stay in registers
extern "C" __device__ int dcsg_spec_threaded();  // 1: the evaluation is a chain of out-of-line calls (large programs)
typedef float dcsg_cl_float4 __attribute__((ext_vector_type(4)));
extern "C" __device__ dcsg_cl_float4 dcsg_spec_eval4(dcsg_cl_float4 px, dcsg_cl_float4 py, dcsg_cl_float4 pz);  // threaded form: four points
#define DCSG_HAVE_SPEC 1
#endif
#else
// OpenCL max(gentype x, gentype y): y if x < y, otherwise x
__device__ __forceinline__ float cl_max_f(float x, float y) { return x < y ? y : x; }
// The five brushes every scene starts with (scenecompiler.py:424-425, DesignCSG.py:9-22).
// Their unsuffixed literals are double in OpenCL C ("length(v)-0.5" subtracts in double and narrows at
// `return`).  A single +,- on float operands evaluated in double and rounded to float equals the float
// operation (53 >= 2*24+2 bits: the double rounding is innocuous), and rounding is monotone, so
// max-then-narrow equals narrow-then-max: the float forms below return the same bits and keep the
// f64 pipe out of the hot loop.  tests/test_gpu_parity.py::test_builtin_bank_equals_jit_bank pins this
// against the literal OpenCL text compiled by the OpenCL front-end.
__device__ __forceinline__ float dcsg_bank(float x, float y, float z, int shape_id)
{
    switch (shape_id & 0xff) {  // wave-uniform
    case 0: return (float)K2_MAX_DISTANCE;
    case 1: return 0.0f;
    case 2: return dcsg_sqrt(x * x + y * y + z * z) - 0.5f;
    case 3: {
        float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y), az = __builtin_fabsf(z);
        float r = dcsg_sqrt(ax * ax + az * az);
        return cl_max_f(r - 0.5f, ay - 0.5f);
    }
    case 4: {
        float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y), az = __builtin_fabsf(z);
        return cl_max_f(ax - 0.5f, cl_max_f(ay - 0.5f, az - 0.5f));
    }
    default: return 0.0f;  // sdf_bank's trailing "return 0.0;"
    }
}
#endif

// ---------------------------------------------------------------------------
// Scene tables in LDS.
// Blob (built by csrc/scene.cpp): [0] header {n_steps,n_objects,n_slots,flags}, flags bit 0 =
// "brush 0 is the constant null brush `return MAX_DISTANCE;`" (always true for the built-in bank;
// for a JIT-linked scene.cl only when the host verified sd0's body); [1] header
// {n_stack_steps, stack_depth, n_items, 0}; then the register program, the stack program, the culling
// items, the objects:
// steps {opcode,left,right,dest}; 4 x float4 per object:
// {pos.xyz, brush id | material id << 8}, {right.xyz, Lipschitz bound}, {up.xyz,0}, {forward.xyz,0}.
// After the blob: the interpreter stack, slot-major (slot*DCSG_BLOCK + tid), so
// every access is one conflict-free ds_read/ds_write_b32.
// ---------------------------------------------------------------------------
extern __shared__ uint4 dcsg_lds[];

struct SceneLds {
    const int4* steps;
    const float4* obj;
    const int4* stack_steps;  // stack program (empty when the program does not nest)
    const int4* items;        // top-level items of the stack program (tile culling)
    int n_items;
    float* stack;  // generic interpreter's LDS stack, already offset by threadIdx.x
    int n_steps;
    int n_stack_steps;
    int n_regs;
    int flags;
    int n_vec4;  // size of the blob: dcsg_scene_view(n_vec4) finds the tables again
};

// the tables as they lie in LDS once dcsg_stage_scene has copied them
__device__ __forceinline__ SceneLds dcsg_scene_view(int n_vec4)
{
    SceneLds s;
    int n_steps = __builtin_amdgcn_readfirstlane((int)dcsg_lds[0].x);
    s.n_steps = n_steps;
    s.flags = __builtin_amdgcn_readfirstlane((int)dcsg_lds[0].w);
    s.n_regs = __builtin_amdgcn_readfirstlane((int)dcsg_lds[0].z);
    int n_stack = __builtin_amdgcn_readfirstlane((int)dcsg_lds[1].x);
    s.n_stack_steps = n_stack;
    s.steps = reinterpret_cast<const int4*>(dcsg_lds + 2);
    int n_items = __builtin_amdgcn_readfirstlane((int)dcsg_lds[1].z);
    s.n_items = n_items;
    s.stack_steps = reinterpret_cast<const int4*>(dcsg_lds + 2 + n_steps);
    s.items = reinterpret_cast<const int4*>(dcsg_lds + 2 + n_steps + n_stack);
    s.obj = reinterpret_cast<const float4*>(dcsg_lds + 2 + n_steps + n_stack + n_items);
    s.stack = reinterpret_cast<float*>(dcsg_lds + n_vec4) + threadIdx.x;
    s.n_vec4 = n_vec4;
    return s;
}
__device__ __forceinline__ SceneLds dcsg_stage_scene(const uint4* __restrict__ blob, int n_vec4)
{
    for (int i = threadIdx.x; i < n_vec4; i += DCSG_BLOCK) dcsg_lds[i] = blob[i];
    __syncthreads();
    return dcsg_scene_view(n_vec4);
}

// Internal opcodes: the host (csrc/scene.cpp, fuse_program) rewrites the reference's
// "IMPORT -> R0; MIN a R0 -> a" / "...; MAX a R0 -> a" / "...; NEGATE R0 -> R0; MAX a R0 -> a"
// triples, when R0 is dead afterwards, into one accumulate step.  Same comparisons, same operands.
#define OP_IMPORT_MIN 16     // stack[dest] = T_min(stack[dest], sdf)
#define OP_IMPORT_MAX 17     // stack[dest] = T_max(stack[dest], sdf)
#define OP_IMPORT_NEGMAX 18  // stack[dest] = T_max(stack[dest], -sdf)
#define OP_IMPORT_SMIN 19    // the same three with the smooth joins
#define OP_IMPORT_SMAX 20
#define OP_IMPORT_NEGSMAX 21

// Polynomial smooth minimum / maximum with blend radius k (extension, see OP_SMOOTH_MIN): float, left to right.
__device__ __forceinline__ float dcsg_smooth_term(float a, float b, float k, float inv_k)
{
    float t = k - __builtin_fabsf(a - b);
    t = t > 0.0f ? t : 0.0f;
    float h = t * inv_k;
    return ((h * h) * k) * 0.25f;
}
__device__ __forceinline__ float dcsg_smin(float a, float b, float k, float inv_k)
{
    return (a < b ? a : b) - dcsg_smooth_term(a, b, k, inv_k);
}
__device__ __forceinline__ float dcsg_smax(float a, float b, float k, float inv_k)
{
    return (a > b ? a : b) + dcsg_smooth_term(a, b, k, inv_k);
}

// ---------------------------------------------------------------------------
// Generic interpreter: one point, stack in LDS (any slot count up to 64).
// primary_sdf, k2.cl:47-144.  Step words are wave-uniform: read once from LDS (broadcast) and
// moved to SGPRs so the opcode switch and the brush switch are scalar branches.
// Used by the mesher's on-demand samples and as the exact path for waves that hold a -0.0
// coordinate (see dcsg_stencil).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dcsg_import(const SceneLds& s, int shape, int obj, float x, float y, float z)
{
    if (shape == 0 && (s.flags & 1)) return (float)K2_MAX_DISTANCE;  // sd0 ignores v: skip the dead transform
    float4 q0 = s.obj[4 * obj + 0];
    float4 q1 = s.obj[4 * obj + 1];
    float4 q2 = s.obj[4 * obj + 2];
    float4 q3 = s.obj[4 * obj + 3];
    float dx = x - q0.x, dy = y - q0.y, dz = z - q0.z;  // k2.cl:105-111
    float A = dx * q1.x + dy * q1.y + dz * q1.z;        // dot(v-o, right)
    float B = dx * q2.x + dy * q2.y + dz * q2.z;        // dot(v-o, up)
    float C = dx * q3.x + dy * q3.y + dz * q3.z;        // dot(v-o, forward)
    return dcsg_bank(A, B, C, shape);
}

__device__ __forceinline__ float dcsg_primary_sdf(const SceneLds& s, float x, float y, float z)
{
    float export_value = (float)K2_MAX_DISTANCE;
    float k = 0.0f, inv_k = 0.0f;  // blend radius of the smooth joins (wave-uniform)
    for (int c = 0; c < s.n_steps; c++) {
        int4 st = s.steps[c];
        int op = __builtin_amdgcn_readfirstlane(st.x);
        int left = __builtin_amdgcn_readfirstlane(st.y);
        int right = __builtin_amdgcn_readfirstlane(st.z);
        int dest = __builtin_amdgcn_readfirstlane(st.w);
        switch (op) {
        case OP_IMPORT: s.stack[dest * DCSG_BLOCK] = dcsg_import(s, left & 0xff, right, x, y, z); break;
        case OP_IMPORT_MIN: {
            float a = s.stack[dest * DCSG_BLOCK], b = dcsg_import(s, left & 0xff, right, x, y, z);
            s.stack[dest * DCSG_BLOCK] = a < b ? a : b;
        } break;
        case OP_IMPORT_MAX: {
            float a = s.stack[dest * DCSG_BLOCK], b = dcsg_import(s, left & 0xff, right, x, y, z);
            s.stack[dest * DCSG_BLOCK] = a > b ? a : b;
        } break;
        case OP_IMPORT_NEGMAX: {
            float a = s.stack[dest * DCSG_BLOCK], b = -dcsg_import(s, left & 0xff, right, x, y, z);
            s.stack[dest * DCSG_BLOCK] = a > b ? a : b;
        } break;
        case OP_EXPORT: export_value = s.stack[left * DCSG_BLOCK]; break;
        case OP_MIN: {  // T_min(a,b) (a<b?a:b), k2.cl:29
            float a = s.stack[left * DCSG_BLOCK], b = s.stack[right * DCSG_BLOCK];
            s.stack[dest * DCSG_BLOCK] = a < b ? a : b;
        } break;
        case OP_MAX: {  // T_max(a,b) (a>b?a:b), k2.cl:30
            float a = s.stack[left * DCSG_BLOCK], b = s.stack[right * DCSG_BLOCK];
            s.stack[dest * DCSG_BLOCK] = a > b ? a : b;
        } break;
        case OP_NEGATE: s.stack[dest * DCSG_BLOCK] = -s.stack[left * DCSG_BLOCK]; break;
        case OP_IDENTITY: s.stack[dest * DCSG_BLOCK] = s.stack[left * DCSG_BLOCK]; break;
        case OP_SET_K:
            k = __int_as_float(left);
            inv_k = __int_as_float(right);
            break;
        case OP_SMOOTH_MIN:
            s.stack[dest * DCSG_BLOCK] = dcsg_smin(s.stack[left * DCSG_BLOCK], s.stack[right * DCSG_BLOCK], k, inv_k);
            break;
        case OP_SMOOTH_MAX:
            s.stack[dest * DCSG_BLOCK] = dcsg_smax(s.stack[left * DCSG_BLOCK], s.stack[right * DCSG_BLOCK], k, inv_k);
            break;
        case OP_IMPORT_SMIN:
            s.stack[dest * DCSG_BLOCK] = dcsg_smin(s.stack[dest * DCSG_BLOCK], dcsg_import(s, left & 0xff, right, x, y, z), k, inv_k);
            break;
        case OP_IMPORT_SMAX:
            s.stack[dest * DCSG_BLOCK] = dcsg_smax(s.stack[dest * DCSG_BLOCK], dcsg_import(s, left & 0xff, right, x, y, z), k, inv_k);
            break;
        case OP_IMPORT_NEGSMAX:
            s.stack[dest * DCSG_BLOCK] = dcsg_smax(s.stack[dest * DCSG_BLOCK], -dcsg_import(s, left & 0xff, right, x, y, z), k, inv_k);
            break;
        default: break;
        }
    }
    return export_value;
}

// n / d, correctly rounded, through d's reciprocal refined once (r1, shared by the three components) and ONE residual
// correction of the quotient.  The compiler's IEEE-754 division does two corrections between v_div_scale_f32 and
// v_div_fixup_f32 (rounds 1-2 shipped that); tools/div_short_probe.hip compares both with `/` for every pair of float
// mantissas (2^46 pairs, 40 s on the GPU): neither ever differs, so the second correction never changes a bit.  The
// quotient's bits depend on the mantissas alone while nothing underflows or overflows, which the caller guarantees
// (2^-60 <= |n|, d <= 2^60 and |n| <= 2 d: quotient, reciprocal and residual are normal).
__device__ __forceinline__ float dcsg_div_by(float n, float d, float r1)
{
    float q = n * r1;
    float e2 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e2, r1, q);
}

#define DCSG_KEY_ORDINARY (2u * 0x21800000u - 2u)  // dcsg_component_key of 2^-60
#ifdef DCSG_RELAXED
__device__ __forceinline__ void dcsg_normalize(float& x, float& y, float& z)
{
    float q = x * x + y * y + z * z;
    float r = q > 0.0f ? __builtin_amdgcn_rsqf(q) : 0.0f;  // zero vector stays zero
    x *= r;
    y *= r;
    z *= r;
}
__device__ __forceinline__ void dcsg_finish_normal(float Dx, float Dy, float Dz, float& nx, float& ny, float& nz)
{
    nx = Dx, ny = Dy, nz = Dz;  // the scale 1/(2 eps) cancels in the normalisation
    dcsg_normalize(nx, ny, nz);
}
#else
// OpenCL normalize(float3) pinned as v / length(v), zero vector stays zero.
// dcsg_normalize_fast: the quotients through one shared reciprocal when every lane of the wave has ordinary operands
// (returns true); otherwise nothing is written and the caller takes dcsg_normalize_plain.
// Ordinary: every component is a zero or at least 2^-60 in magnitude, and the length is at most 2^60.  An exactly zero
// component is common -- lattice points on a symmetry plane of the scene, where the two stencil values of an axis are
// equal (one 64-point pass in eight at 513 points per row with the box centred on a sphere) -- and harmless: 0 * r1 = 0,
// the residual is 0, the quotient is 0 (the sign of a -0 numerator is put back by the copysign; for every other
// numerator it is a no-op, the divisor being positive).
// 2 |c| - 2 with |c| as an integer (one v_lshl_add_u32: the shift drops the sign), zeros wrapped to the top:
// key >= DCSG_KEY_ORDINARY <=> c is a zero or at least 2^-60 in magnitude
__device__ __forceinline__ unsigned int dcsg_component_key(float c)
{
    return (__float_as_uint(c) << 1) - 2u;
}
__device__ __forceinline__ bool dcsg_normalize_fast(float& x, float& y, float& z, float& len)
{
    // the length's square root (dcsg_sqrt) written out, its operand test and the length's range test folded into ONE
    // unsigned range test of the estimate g: 2^-48 <= g < 2^60 (g is within an ulp of the root, so the root is positive and
    // at most 2^60; NaN, zero, tiny and huge squared lengths all fall outside -- two compares and a branch fewer per point
    // than testing the root's operand, then 0 < len <= 2^60).
    // components are at most len (1 + 2^-22) in magnitude; a tiny, huge or NaN operand anywhere in the wave sends the
    // wave through the plain divisions (a NaN component makes g NaN; the zero vector has g NaN too and is left to the
    // plain path, which returns it unchanged)
    const float q = x * x + y * y + z * z;
    const float ry = __builtin_amdgcn_rsqf(q), g = q * ry;
    unsigned int k0 = dcsg_component_key(x), k1 = dcsg_component_key(y), k2 = dcsg_component_key(z);
    unsigned int smallest = k0 < k1 ? k0 : k1;
    smallest = k2 < smallest ? k2 : smallest;
    const bool plain = !(smallest >= DCSG_KEY_ORDINARY && __float_as_uint(g) - 0x27800000u < 0x5d800000u - 0x27800000u);
    if (__builtin_amdgcn_ballot_w64(plain) != 0ull) {
        len = dcsg_sqrt(q);
        return false;
    }
    len = __builtin_fmaf(__builtin_fmaf(-g, g, q), 0.5f * ry, g);  // dcsg_sqrt's fast path: same operations, same bits
    float r = __builtin_amdgcn_rcpf(len);
    float e = __builtin_fmaf(-len, r, 1.0f);
    float r1 = __builtin_fmaf(e, r, r);
    x = __builtin_copysignf(dcsg_div_by(x, len, r1), x);
    y = __builtin_copysignf(dcsg_div_by(y, len, r1), y);
    z = __builtin_copysignf(dcsg_div_by(z, len, r1), z);
    return true;
}
__device__ __forceinline__ void dcsg_normalize_plain(float& x, float& y, float& z, float len)
{
    if (len == 0.0f) {
        x = y = z = 0.0f;
    } else {
        x = x / len;
        y = y / len;
        z = z / len;
    }
}
__device__ __forceinline__ void dcsg_normalize(float& x, float& y, float& z)
{
    float len;
    if (!dcsg_normalize_fast(x, y, z, len)) dcsg_normalize_plain(x, y, z, len);
}

// tail of get_normal, k2.cl:175-177: "1.0/twoE*Dx" is a double product narrowed by the float3 cast.
// With k = 1 / (double)twoE split into kh + kl (two floats), fma(kh, D, kl * D) equals (float)(k * (double)D) for EVERY
// float D with |D| >= 2^-111, and for zeros, infinities and NaNs (all 2^32 bit patterns compared on the GPU:
// tools/gradscale_probe.hip, tests/test_gpu_parity.py::test_normal_all_float_inputs); below that kl * D loses bits
// to underflow.  Such a component scales to less than 2^-104, which dcsg_normalize_fast refuses (2^-60), and a wave
// it refuses redoes the scale in double precision: three multiplies and three FMAs replace six conversions and three
// double multiplies on every wave that matters.
__device__ __forceinline__ void dcsg_finish_normal(float Dx, float Dy, float Dz, float& nx, float& ny, float& nz)
{
    const float twoE = (float)(2.0 * K2_NORMAL_EPSILON);
    const double k = 1.0 / (double)twoE;
    const float kh = (float)k, kl = (float)(k - (double)kh);
    nx = __builtin_fmaf(kh, Dx, kl * Dx);
    ny = __builtin_fmaf(kh, Dy, kl * Dy);
    nz = __builtin_fmaf(kh, Dz, kl * Dz);
    float len;
    if (dcsg_normalize_fast(nx, ny, nz, len)) return;
    nx = (float)(k * (double)Dx);
    ny = (float)(k * (double)Dy);
    nz = (float)(k * (double)Dz);
    len = dcsg_sqrt(nx * nx + ny * ny + nz * nz);
    dcsg_normalize_plain(nx, ny, nz, len);
}
#endif

// get_normal, k2.cl:149-179
__device__ __forceinline__ void dcsg_get_normal(const SceneLds& s, float x, float y, float z, float& nx, float& ny,
                                                float& nz)
{
    const float e = (float)K2_NORMAL_EPSILON;
    const float zero = 0.0f;  // v+dx adds +0.0 on the other axes (k2.cl:167-173)
    float Dx = dcsg_primary_sdf(s, x + e, y + zero, z + zero) - dcsg_primary_sdf(s, x - e, y - zero, z - zero);
    float Dy = dcsg_primary_sdf(s, x + zero, y + e, z + zero) - dcsg_primary_sdf(s, x - zero, y - e, z - zero);
    float Dz = dcsg_primary_sdf(s, x + zero, y + zero, z + e) - dcsg_primary_sdf(s, x - zero, y - zero, z - e);
    dcsg_finish_normal(Dx, Dy, Dz, nx, ny, nz);
}

// Out-of-line copies for the rare exact path of the fast kernels (keeps their code small).
// The scene tables are found again from their place in LDS (dcsg_scene_view) instead of through a reference to the
// caller's SceneLds: a struct whose address is passed to a real call has to live in scratch memory, 80 bytes per lane
// that rounds 1-2 carried in every kernel with this path.
__device__ __noinline__ float dcsg_primary_sdf_slow(int n_vec4, float x, float y, float z)
{
    SceneLds s = dcsg_scene_view(n_vec4);
    return dcsg_primary_sdf(s, x, y, z);
}

// ---------------------------------------------------------------------------
// Fast interpreter: N points per thread in ONE pass over the program, stack in registers.
//
// The central-difference normal needs primary_sdf at 7 stencil points.  Interpreting the program
// once for all of them pays the step decode once, and -- because the unperturbed coordinates of
// the stencil are passed as the SAME values -- lets the compiler share every common sub-expression
// of the seven object transforms (value numbering; the arithmetic and its order are unchanged, so
// results stay bit-identical).
// ---------------------------------------------------------------------------
// Brush evaluation at N points: ONE wave-uniform dispatch, then N independent straight-line bodies in a
// single basic block (lets the scheduler interleave them and use packed f32 math).
template <int N>
__device__ __forceinline__ void dcsg_bank_n(const float (&A)[N], const float (&B)[N], const float (&C)[N], int shape,
                                            float (&r)[N])
{
#ifdef DCSG_EXTERN_BANK
#pragma unroll
    for (int k = 0; k < N; k++) r[k] = dcsg_bank(A[k], B[k], C[k], shape);
#else
    switch (shape & 0xff) {
    case 0:
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = (float)K2_MAX_DISTANCE;
        break;
    case 2: {
        float q[N], len[N];
#pragma unroll
        for (int k = 0; k < N; k++) q[k] = A[k] * A[k] + B[k] * B[k] + C[k] * C[k];
        dcsg_sqrt_n<N>(q, len);
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = len[k] - 0.5f;
        break;
    }
    case 3: {
        float q[N], rad[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            float ax = __builtin_fabsf(A[k]), az = __builtin_fabsf(C[k]);
            q[k] = ax * ax + az * az;
        }
        dcsg_sqrt_n<N>(q, rad);
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = cl_max_f(rad[k] - 0.5f, __builtin_fabsf(B[k]) - 0.5f);
        break;
    }
    case 4:
#pragma unroll
        for (int k = 0; k < N; k++) {
            float ax = __builtin_fabsf(A[k]), ay = __builtin_fabsf(B[k]), az = __builtin_fabsf(C[k]);
            r[k] = cl_max_f(ax - 0.5f, cl_max_f(ay - 0.5f, az - 0.5f));
        }
        break;
    default:  // brush 1 ("return 0.0;") and sdf_bank's trailing "return 0.0;"
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = 0.0f;
        break;
    }
#endif
}

// Stack program (built by the host from the register program when its live ranges nest, which is
// what the reference scene compiler always emits -- scenecompiler.py:353-387):
//   PUSH b,o        push sdf(object o, brush b)                      ("IMPORT b,o -> group slot")
//   ACC_* b,o       top = T_min/T_max(top, +-sdf)                    (fused leaf joins, see above)
//   POP_*           v = pop(); top = T_min/T_max(top, +-v)           (group joins)
//   EXPORT_TOP      result = top
// PUSH/POP carry the stack depth before the step, so the register shifts they need are selected by a
// wave-uniform compare and every stack access names a fixed register: no indexing, no memory.
#define SOP_PUSH 32
#define SOP_ACC_MIN 33
#define SOP_ACC_MAX 34
#define SOP_ACC_NEGMAX 35
#define SOP_POP_MIN 36
#define SOP_POP_MAX 37
#define SOP_POP_NEGMAX 38
#define SOP_EXPORT_TOP 39
#define SOP_SET_K 40        // extension (smooth joins): k = as_float(left), 1/k = as_float(right)
#define SOP_ACC_SMIN 41
#define SOP_ACC_SMAX 42
#define SOP_ACC_NEGSMAX 43
#define SOP_POP_SMIN 44
#define SOP_POP_SMAX 45
#define SOP_POP_NEGSMAX 46

// `list` (LDS, wave-uniform contents) optionally restricts the pass to a subset of the steps, in program
// order: the active-step list of the tile culling below.
template <int D, int N>
__device__ __forceinline__ void dcsg_interp(const SceneLds& s, const float (&px)[N], const float (&py)[N],
                                            const float (&pz)[N], float (&out)[N],
                                            const unsigned short* list = nullptr, int n_list = 0)
{
    float stk[D][N];  // stk[0] = top of stack
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int k = 0; k < N; k++) stk[i][k] = 0.0f;
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = (float)K2_MAX_DISTANCE;
    float bk = 0.0f, inv_bk = 0.0f;  // blend radius of the smooth joins (wave-uniform)
    const int n_run = list ? n_list : s.n_stack_steps;
    for (int k_run = 0; k_run < n_run; k_run++) {
        const int c = list ? __builtin_amdgcn_readfirstlane((int)list[k_run]) : k_run;
        int4 st = s.stack_steps[c];
        int op = __builtin_amdgcn_readfirstlane(st.x);
        int left = __builtin_amdgcn_readfirstlane(st.y);
        int right = __builtin_amdgcn_readfirstlane(st.z);
        int depth = __builtin_amdgcn_readfirstlane(st.w);
        if (op <= SOP_ACC_NEGMAX || (op >= SOP_ACC_SMIN && op <= SOP_ACC_NEGSMAX)) {  // PUSH / ACC_*: evaluate the object at the N points
            float r[N];
            int shape = left & 0xff;
            if (shape == 0 && (s.flags & 1)) {
#pragma unroll
                for (int k = 0; k < N; k++) r[k] = (float)K2_MAX_DISTANCE;
            } else {
                float4 q0 = s.obj[4 * right + 0];
                float4 q1 = s.obj[4 * right + 1];
                float4 q2 = s.obj[4 * right + 2];
                float4 q3 = s.obj[4 * right + 3];
                float A[N], B[N], C[N];
#pragma unroll
                for (int k = 0; k < N; k++) {
                    float dx = px[k] - q0.x, dy = py[k] - q0.y, dz = pz[k] - q0.z;  // k2.cl:105-111
                    A[k] = dx * q1.x + dy * q1.y + dz * q1.z;
                    B[k] = dx * q2.x + dy * q2.y + dz * q2.z;
                    C[k] = dx * q3.x + dy * q3.y + dz * q3.z;
                }
                dcsg_bank_n<N>(A, B, C, shape, r);
            }
            if (op == SOP_PUSH) {
#pragma unroll
                for (int lvl = D - 1; lvl >= 1; lvl--) {
                    if (depth >= lvl) {
#pragma unroll
                        for (int k = 0; k < N; k++) stk[lvl][k] = stk[lvl - 1][k];
                    }
                }
#pragma unroll
                for (int k = 0; k < N; k++) stk[0][k] = r[k];
            } else if (op == SOP_ACC_MIN) {
#pragma unroll
                for (int k = 0; k < N; k++) stk[0][k] = stk[0][k] < r[k] ? stk[0][k] : r[k];
            } else if (op == SOP_ACC_MAX) {
#pragma unroll
                for (int k = 0; k < N; k++) stk[0][k] = stk[0][k] > r[k] ? stk[0][k] : r[k];
            } else if (op == SOP_ACC_NEGMAX) {
#pragma unroll
                for (int k = 0; k < N; k++) {
                    float nb = -r[k];
                    stk[0][k] = stk[0][k] > nb ? stk[0][k] : nb;
                }
            } else if (op == SOP_ACC_SMIN) {
#pragma unroll
                for (int k = 0; k < N; k++) stk[0][k] = dcsg_smin(stk[0][k], r[k], bk, inv_bk);
            } else if (op == SOP_ACC_SMAX) {
#pragma unroll
                for (int k = 0; k < N; k++) stk[0][k] = dcsg_smax(stk[0][k], r[k], bk, inv_bk);
            } else {
#pragma unroll
                for (int k = 0; k < N; k++) stk[0][k] = dcsg_smax(stk[0][k], -r[k], bk, inv_bk);
            }
        } else if (op <= SOP_POP_NEGMAX || (op >= SOP_POP_SMIN && op <= SOP_POP_NEGSMAX)) {  // join the top group into its parent (stk[1])
#pragma unroll
            for (int k = 0; k < N; k++) {
                float a = stk[1][k], b = stk[0][k];
                if (op == SOP_POP_MIN) stk[0][k] = a < b ? a : b;
                else if (op == SOP_POP_MAX) stk[0][k] = a > b ? a : b;
                else if (op == SOP_POP_NEGMAX) {
                    float nb = -b;
                    stk[0][k] = a > nb ? a : nb;
                } else if (op == SOP_POP_SMIN) stk[0][k] = dcsg_smin(a, b, bk, inv_bk);
                else if (op == SOP_POP_SMAX) stk[0][k] = dcsg_smax(a, b, bk, inv_bk);
                else stk[0][k] = dcsg_smax(a, -b, bk, inv_bk);
            }
#pragma unroll
            for (int lvl = 1; lvl <= D - 2; lvl++) {
                if (depth > lvl + 1) {
#pragma unroll
                    for (int k = 0; k < N; k++) stk[lvl][k] = stk[lvl + 1][k];
                }
            }
        } else if (op == SOP_EXPORT_TOP) {
#pragma unroll
            for (int k = 0; k < N; k++) out[k] = stk[0][k];
        } else if (op == SOP_SET_K) {
            bk = __int_as_float(left);
            inv_bk = __int_as_float(right);
        }
    }
}

__device__ __forceinline__ bool dcsg_is_neg_zero(float v) { return __float_as_uint(v) == 0x80000000u; }

// SDF and/or normal at one point through the 7-point stencil (or 1 point when only the SDF is wanted).
// The reference forms the stencil as v +- (eps,0,0): the other coordinates get "+ 0.0f", which changes
// bits only for -0.0.  A wave holding such a coordinate takes the generic path, every other wave passes
// the unperturbed coordinates as shared values (see dcsg_interp).
template <int K>
__device__ __forceinline__ void dcsg_stencil(const SceneLds& s, float x, float y, float z, bool want_sdf,
                                             bool want_normal, float& sdf, float& nx, float& ny, float& nz,
                                             const unsigned short* list = nullptr, int n_list = 0)
{
    if (!want_normal) {
        float p0[1] = {x}, p1[1] = {y}, p2[1] = {z}, o[1];
        dcsg_interp<K, 1>(s, p0, p1, p2, o, list, n_list);
        sdf = o[0];
        return;
    }
    bool nzero = dcsg_is_neg_zero(x) || dcsg_is_neg_zero(y) || dcsg_is_neg_zero(z);
    if (__builtin_amdgcn_ballot_w64(nzero) != 0ull) {
        if (want_sdf) sdf = dcsg_primary_sdf_slow(s.n_vec4, x, y, z);
        const float e = (float)K2_NORMAL_EPSILON, zero = 0.0f;
        float Dx = dcsg_primary_sdf_slow(s.n_vec4, x + e, y + zero, z + zero) - dcsg_primary_sdf_slow(s.n_vec4, x - e, y - zero, z - zero);
        float Dy = dcsg_primary_sdf_slow(s.n_vec4, x + zero, y + e, z + zero) - dcsg_primary_sdf_slow(s.n_vec4, x - zero, y - e, z - zero);
        float Dz = dcsg_primary_sdf_slow(s.n_vec4, x + zero, y + zero, z + e) - dcsg_primary_sdf_slow(s.n_vec4, x - zero, y - zero, z - e);
        dcsg_finish_normal(Dx, Dy, Dz, nx, ny, nz);
        return;
    }
    const float e = (float)K2_NORMAL_EPSILON;
    float xp = x + e, xm = x - e, yp = y + e, ym = y - e, zp = z + e, zm = z - e;
    float px[7] = {x, xp, xm, x, x, x, x};
    float py[7] = {y, y, y, yp, ym, y, y};
    float pz[7] = {z, z, z, z, z, zp, zm};
    float o[7];
    dcsg_interp<K, 7>(s, px, py, pz, o, list, n_list);
    sdf = o[0];
    dcsg_finish_normal(o[1] - o[2], o[3] - o[4], o[5] - o[6], nx, ny, nz);
}

#ifndef DCSG_EXTERN_BANK
// ---------------------------------------------------------------------------
// Tile culling for many-object scenes (flat programs over the built-in brushes).
//
// A wave evaluates 64 points that sit close together (a run of the lattice, a cluster of mesh vertices).
// Before interpreting the program for them it bounds every object's value over the tile,
//     s_i(p) in [c_i - m_i, c_i + m_i],  c_i = s_i(tile centre),  m_i = L_i * R + float margin,
// (L_i = Lipschitz constant of brush_i o transform_i: the built-in brushes have gradient norm <= 1, so
// L_i <= ||[right;up;forward]||_F, precomputed by the host; R = radius of the tile including the normal
// stencil), and pushes an interval [lo, hi] that contains the accumulator at every point of the tile through
// the program.  A step "acc = T_min(acc, s_i)" cannot change acc anywhere in the tile when c_i - m_i > hi
// (T_min(a,b) = a<b?a:b returns a whenever a < b); likewise T_max / T_max(-s).  Only the steps that may matter
// are written to an LDS list and interpreted -- with exactly the same arithmetic, so results stay
// bit-identical (tests/test_gpu_parity.py, tests/test_gpu_fullsize.py compare with the oracle).
// 64 objects are bounded per pass (one per lane); the interval walk is a wave-uniform loop.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_min_f(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, DCSG_WAVE));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, DCSG_WAVE));
    return v;
}

// The interval [lo, hi] of the accumulator evolves by clamps x -> min(max(x, P), Q): T_min with a value in
// [lb, ub] maps lo by (P,Q) = (-inf, lb) and hi by (-inf, ub); T_max by (lb, +inf) / (ub, +inf); T_max(-s) by
// (-ub, +inf) / (-lb, +inf); PUSH is the constant clamp (lb, lb) / (ub, ub).  Clamps are closed under
// composition -- (p1,q1) then (p2,q2) is (max(p1,p2), min(max(q1,p2), q2)) -- so the interval before every step
// is an exclusive prefix scan, and the kept steps are compacted with a ballot: no sequential walk.
struct DcsgClamp2 {
    float plo, qlo, phi, qhi;  // clamp acting on lo, clamp acting on hi
};
__device__ __forceinline__ DcsgClamp2 dcsg_clamp_identity()
{
    const float inf = __builtin_inff();
    return {-inf, inf, -inf, inf};
}
// `a` first, then `b`
__device__ __forceinline__ DcsgClamp2 dcsg_clamp_compose(const DcsgClamp2& a, const DcsgClamp2& b)
{
    return {fmaxf(a.plo, b.plo), fminf(fmaxf(a.qlo, b.plo), b.qlo), fmaxf(a.phi, b.phi), fminf(fmaxf(a.qhi, b.phi), b.qhi)};
}
__device__ __forceinline__ DcsgClamp2 dcsg_clamp_shfl_up(const DcsgClamp2& v, int o)
{
    return {__shfl_up(v.plo, o, DCSG_WAVE), __shfl_up(v.qlo, o, DCSG_WAVE), __shfl_up(v.phi, o, DCSG_WAVE),
            __shfl_up(v.qhi, o, DCSG_WAVE)};
}
// inclusive scan of clamps across the 64 lanes of a wave (lane order = program order)
__device__ __forceinline__ DcsgClamp2 dcsg_clamp_wave_scan(DcsgClamp2 v)
{
    const int lane = threadIdx.x & (DCSG_WAVE - 1);
    for (int o = 1; o < DCSG_WAVE; o <<= 1) {
        DcsgClamp2 t = dcsg_clamp_shfl_up(v, o);
        if (lane >= o) v = dcsg_clamp_compose(t, v);
    }
    return v;
}

// keep test of a step given the interval [lo, hi] of the accumulator before it (k = blend radius in force)
__device__ __forceinline__ bool dcsg_step_keep(int op, float lb, float ub, float lo, float hi, float k = 0.0f)
{
    if (op == SOP_ACC_MIN) return !(lb > hi);    // T_min(a, s) = a wherever a < s
    if (op == SOP_ACC_MAX) return !(ub < lo);    // T_max(a, s) = a wherever a > s
    if (op == SOP_ACC_NEGMAX) return !(-lb < lo);
    // smooth joins return the accumulator unchanged wherever |a - s| >= k on the accumulator's side (h = 0)
    if (op == SOP_ACC_SMIN) return !(lb - hi > k);
    if (op == SOP_ACC_SMAX) return !(lo - ub > k);
    if (op == SOP_ACC_NEGSMAX) return !(lo + lb > k);
    return op >= 0;                              // PUSH, POP, EXPORT, SET_K: always
}

// value of one PUSH / ACC step at the tile centre with its margin: s in [lb, ub] over the tile
__device__ __forceinline__ void dcsg_step_bounds(const SceneLds& s, const int4& st, float cx, float cy, float cz, float R,
                                                 float& lb, float& ub)
{
    int shape = st.y & 0xff;
    float val, L;
    if (shape == 0 && (s.flags & 1)) {
        val = (float)K2_MAX_DISTANCE;
        L = 0.0f;
    } else {
        float4 q1 = s.obj[4 * st.z + 1];
        val = dcsg_import(s, shape, st.z, cx, cy, cz);
        L = q1.w;
    }
    float m = L * R * 1.0001f + 1e-3f + 1e-5f * __builtin_fabsf(val);
    lb = val - m;
    ub = val + m;
}

// Bounds of the program's steps over a tile, computed ONE STEP PER LANE before the items are walked: a group's leaves are
// then read back instead of being evaluated one after the other by the group's single lane (grouped256: 16 lanes of a wave
// walking 16 leaves each -- 256 brush evaluations in sequence on a quarter of the lanes -- became 4 rounds of 64).
// cols < 0: element e at base[e] (the block list's work area); otherwise the wave's own columns of the generic interpreter's
// LDS stack (row e / 64, column cols + e % 64 of rows of DCSG_BLOCK floats): nothing else of this wave uses them while a
// list is being built, and the other waves' columns are theirs.
struct DcsgStepCache {
    float* base;
    int cols;
};
__device__ __forceinline__ float* dcsg_step_cache_at(const DcsgStepCache& c, int e)
{
    return c.cols < 0 ? c.base + e : c.base + (e >> 6) * DCSG_BLOCK + c.cols + (e & 63);
}
// Bounds of top-level item i over the tile (centre cx,cy,cz, radius R) and the clamp it applies to the root
// accumulator's interval.  A leaf is one accumulate step; a group "PUSH, ACC..., POP_x" is walked step by step
// (interval arithmetic on its own accumulator) and then acts like an accumulate of its result with the POP's
// operator.  op = -1 for "no item"; first/count = the item's steps.
// mask: bit j set = step first+j of the item has to run.  Inside a group a leaf whose bound strictly decides its
// comparison against the GROUP's accumulator interval cannot change that accumulator anywhere in the tile and
// is dropped (groups longer than 64 steps keep all their steps).
__device__ __forceinline__ DcsgClamp2 dcsg_item_clamp(const SceneLds& s, int i, float cx, float cy, float cz, float R, int& op,
                                                      float& lb, float& ub, int& first, int& count,
                                                      unsigned long long& mask, const DcsgStepCache cache = {nullptr, -1})
{
    op = -1;
    lb = ub = 0.0f;
    first = count = 0;
    mask = 1ull;
    DcsgClamp2 k = dcsg_clamp_identity();
    if (i >= s.n_items) return k;
    int4 it = s.items[i];
    first = it.x;
    count = it.y;
    op = it.z;
    const float inf = __builtin_inff();
    if (op == SOP_EXPORT_TOP) return k;
    if (count == 1) {
        if (cache.base) lb = *dcsg_step_cache_at(cache, 2 * first), ub = *dcsg_step_cache_at(cache, 2 * first + 1);
        else dcsg_step_bounds(s, s.stack_steps[first], cx, cy, cz, R, lb, ub);
    } else {
        float glo = 0.0f, ghi = 0.0f, gk = 0.0f, gq = 0.0f;
        const bool small = count <= 64;
        mask = small ? (1ull << (count - 1)) : ~0ull;  // the POP always runs
        // A group that is one fold -- PUSH then T_min steps only, or PUSH then T_max / T_max(-s) steps only -- has the value
        // min (max) over its leaves whatever their order, ties going to the last of the equal ones: a leaf that lies
        // strictly above the smallest upper bound (below the largest lower bound) of ALL the group's leaves, not only of
        // those before it, can neither be that value nor take part in a tie (dcsg_top_min_upper, one level down).  Needs
        // the steps' bounds twice, so only with the cache.
        // (which groups are folds is the host's finding: items[i].w = 1 min, 2 max, unless the scene is a plain union at the
        // root, whose items carry their brush order there)
        const int fold = (cache.base && !(s.flags & 2)) ? it.w : 0;
        float fold_min_ub = inf, fold_max_lb = -inf;
        if (fold == 1) {
            for (int j = 0; j < count - 1; j++) fold_min_ub = fminf(fold_min_ub, *dcsg_step_cache_at(cache, 2 * (first + j) + 1));
        } else if (fold == 2) {
            for (int j = 0; j < count - 1; j++) {
                const float b0 = *dcsg_step_cache_at(cache, 2 * (first + j)), b1 = *dcsg_step_cache_at(cache, 2 * (first + j) + 1);
                fold_max_lb = fmaxf(fold_max_lb, s.stack_steps[first + j].x == SOP_ACC_NEGMAX ? -b1 : b0);
            }
        }
        for (int j = 0; j < count - 1; j++) {
            int4 st = s.stack_steps[first + j];
            if (st.x == SOP_SET_K) {  // radius of the smooth joins that follow; always runs
                gk = __int_as_float(st.y);
                gq = 0.25f * gk * 1.001f + 1e-6f;  // >= the largest amount a smooth join moves past min / max
                if (small) mask |= 1ull << j;
                continue;
            }
            float b0, b1;
            if (cache.base) b0 = *dcsg_step_cache_at(cache, 2 * (first + j)), b1 = *dcsg_step_cache_at(cache, 2 * (first + j) + 1);
            else dcsg_step_bounds(s, st, cx, cy, cz, R, b0, b1);
            bool keep = dcsg_step_keep(st.x, b0, b1, fold == 2 ? fmaxf(glo, fold_max_lb) : glo, fold == 1 ? fminf(ghi, fold_min_ub) : ghi, gk);
            if (keep && small) mask |= 1ull << j;
            if (st.x >= SOP_ACC_SMIN) {  // smooth accumulates: unchanged when dropped, else min / max widened by gq
                if (keep) {
                    if (st.x == SOP_ACC_SMIN) {
                        glo = fminf(glo, b0) - gq;
                        ghi = fminf(ghi, b1);
                    } else if (st.x == SOP_ACC_SMAX) {
                        glo = fmaxf(glo, b0);
                        ghi = fmaxf(ghi, b1) + gq;
                    } else {
                        glo = fmaxf(glo, -b1);
                        ghi = fmaxf(ghi, -b0) + gq;
                    }
                }
            } else if (st.x == SOP_PUSH) {
                glo = b0;
                ghi = b1;
            } else if (st.x == SOP_ACC_MIN) {
                glo = fminf(glo, b0);
                ghi = fminf(ghi, b1);
            } else if (st.x == SOP_ACC_MAX) {
                glo = fmaxf(glo, b0);
                ghi = fmaxf(ghi, b1);
            } else {
                glo = fmaxf(glo, -b1);
                ghi = fmaxf(ghi, -b0);
            }
        }
        lb = glo;
        ub = ghi;
        op = op - (SOP_POP_MIN - SOP_ACC_MIN);  // POP_x joins like ACC_x
    }
    if (op == SOP_PUSH) k = {lb, lb, ub, ub};
    else if (op == SOP_ACC_MIN) k = {-inf, lb, -inf, ub};
    else if (op == SOP_ACC_MAX) k = {lb, inf, ub, inf};
    else k = {-ub, inf, -lb, inf};  // T_max with -s, -s in [-ub, -lb]
    return k;
}

// Scenes whose root accumulator is a plain union (scene flag bit 1: PUSH, then T_min steps and groups joined by T_min): the
// result is the minimum of the items' values, ties going to the LAST of the equal ones, whatever the order of the rest -- so
// an item whose lower bound lies above the smallest UPPER bound of all items (not only of those before it in the program,
// which is what the prefix scan knows) is strictly larger than some other item everywhere in the tile, can neither be the
// minimum nor take part in a tie, and is dropped.  With objects in random order the prefix rule keeps every "record so
// far" (about ln n of them) besides the ones that matter.  Returns that bound for one lane's item (+inf for other items).
__device__ __forceinline__ float dcsg_top_min_upper(int op, float ub)
{
    return (op == SOP_PUSH || op == SOP_ACC_MIN) ? ub : __builtin_inff();
}
// one step's bounds into the cache (steps without an object -- SET_K, POP, EXPORT -- store zeros nobody reads)
__device__ __forceinline__ void dcsg_cache_step_bounds(const SceneLds& s, const DcsgStepCache& c, int t, float cx, float cy, float cz, float R)
{
    if (t >= s.n_stack_steps) return;
    const int4 st = s.stack_steps[t];
    float lb = 0.0f, ub = 0.0f;
    const bool leaf = st.x == SOP_PUSH || (st.x >= SOP_ACC_MIN && st.x <= SOP_ACC_NEGMAX) || (st.x >= SOP_ACC_SMIN && st.x <= SOP_ACC_NEGSMAX);
    if (leaf) dcsg_step_bounds(s, st, cx, cy, cz, R, lb, ub);
    *dcsg_step_cache_at(c, 2 * t) = lb;
    *dcsg_step_cache_at(c, 2 * t + 1) = ub;
}
// The list for an explicit tile: centre (cx, cy, cz), half extents (hx, hy, hz) INCLUDING the stencil's pad.
__device__ __forceinline__ int dcsg_build_active_list_box(const SceneLds& s, float cx, float cy, float cz, float hx, float hy, float hz,
                                                          unsigned short* list)
{
    const int lane = threadIdx.x & (DCSG_WAVE - 1);
    const float inf = __builtin_inff();
    float R = __builtin_sqrtf(hx * hx + hy * hy + hz * hz) * 1.0001f + 1e-6f;
    const int n = s.n_items;
    float top_upper = inf;  // plain unions: the smallest upper bound of all items (dcsg_top_min_upper)
    if (s.flags & 2) {
        for (int base = 0; base < n; base += DCSG_WAVE) {
            int op, first, count;
            float lb, ub;
            unsigned long long imask;
            const int i = base + lane;  // order-free: the items sorted by brush (items[i].w)
            dcsg_item_clamp(s, i < n ? s.items[i].w : n, cx, cy, cz, R, op, lb, ub, first, count, imask);
            top_upper = fminf(top_upper, dcsg_top_min_upper(op, ub));
        }
        top_upper = wave_min_f(top_upper);
    }
    // every step's bounds first, one step per lane (scene flag bit 2: the wave's columns of the generic interpreter's stack hold
    // them), when the items are walked below (plain unions bound each item once, above)
    DcsgStepCache cache = {reinterpret_cast<float*>(dcsg_lds + s.n_vec4), (int)(threadIdx.x & ~(DCSG_WAVE - 1u))};
    const bool cached = (s.flags & 6) == 4;
    if (cached) {
        for (int t = lane; t < s.n_stack_steps; t += DCSG_WAVE) dcsg_cache_step_bounds(s, cache, t, cx, cy, cz, R);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    DcsgClamp2 carry = dcsg_clamp_identity();  // composition of all items of earlier chunks
    int n_active = 0;
    for (int base = 0; base < n; base += DCSG_WAVE) {
        int op, first, count;
        float lb, ub;
        unsigned long long imask;
        DcsgClamp2 mine = dcsg_item_clamp(s, base + lane, cx, cy, cz, R, op, lb, ub, first, count, imask, {cached ? cache.base : nullptr, cache.cols});
        float lo = -inf, hi = top_upper;  // plain unions: the bound of all items, no interval walk
        DcsgClamp2 incl = mine;
        if (!(s.flags & 2)) {
            incl = dcsg_clamp_wave_scan(mine);
            DcsgClamp2 excl = dcsg_clamp_shfl_up(incl, 1);
            if (lane == 0) excl = dcsg_clamp_identity();
            DcsgClamp2 before = dcsg_clamp_compose(carry, excl);
            // a program starts with PUSH, so the clamp before any accumulate is constant: apply it to 0
            lo = fminf(fmaxf(0.0f, before.plo), before.qlo), hi = fminf(fmaxf(0.0f, before.phi), before.qhi);
        }
        bool keep = dcsg_step_keep(op, lb, ub, lo, hi);
        const bool all_steps = count > 64;
        unsigned int sz = keep ? (all_steps ? (unsigned int)count : (unsigned int)__popcll(imask)) : 0u;
        unsigned int inc = wave_scan_incl(sz);
        unsigned int off = (unsigned int)n_active + inc - sz;
        if (keep) {
            if (all_steps) {
                for (int j = 0; j < count; j++) list[off + j] = (unsigned short)(first + j);
            } else {
                unsigned int w = off;
                for (unsigned long long m = imask; m; m &= m - 1ull) list[w++] = (unsigned short)(first + __builtin_ctzll(m));
            }
        }
        n_active += (int)__shfl(inc, DCSG_WAVE - 1, DCSG_WAVE);
        if (!(s.flags & 2)) {
            DcsgClamp2 last = {__shfl(incl.plo, DCSG_WAVE - 1, DCSG_WAVE), __shfl(incl.qlo, DCSG_WAVE - 1, DCSG_WAVE),
                               __shfl(incl.phi, DCSG_WAVE - 1, DCSG_WAVE), __shfl(incl.qhi, DCSG_WAVE - 1, DCSG_WAVE)};
            carry = dcsg_clamp_compose(carry, last);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return n_active;
}
// The wave's list for the tile that contains the valid lanes' points (x, y, z); `pad` is added to every half extent (the
// stencil's epsilon).  Returns the number of steps kept (wave-uniform).
__device__ __forceinline__ int dcsg_build_active_list(const SceneLds& s, bool valid, float x, float y, float z, float pad,
                                                      unsigned short* list)
{
    const float inf = __builtin_inff();
    float x0 = wave_min_f(valid ? x : inf), x1 = wave_max_f(valid ? x : -inf);
    float y0 = wave_min_f(valid ? y : inf), y1 = wave_max_f(valid ? y : -inf);
    float z0 = wave_min_f(valid ? z : inf), z1 = wave_max_f(valid ? z : -inf);
    return dcsg_build_active_list_box(s, 0.5f * (x0 + x1), 0.5f * (y0 + y1), 0.5f * (z0 + z1), 0.5f * (x1 - x0) + pad, 0.5f * (y1 - y0) + pad,
                                      0.5f * (z1 - z0) + pad, list);
}
#define DCSG_HAVE_CULL 1
#endif

#ifdef DCSG_HAVE_SPEC
// Out-of-line exact evaluation for the rare waves that hold a coordinate the fast specialised code must not see:
// -0.0 (the stencil's "+ 0.0f" changes it) or a non-finite / huge value (zero-coefficient terms are not zero then).
__device__ __noinline__ float dcsg_spec_exact(float x, float y, float z) { return dcsg_spec_eval1x(x, y, z); }
__device__ __forceinline__ bool dcsg_spec_odd(float v) { return !(__builtin_fabsf(v) <= 1e30f) || dcsg_is_neg_zero(v); }

// dcsg_stencil for the specialised scene module (K = -1 in the kernel bodies below).  LATTICE: the coordinates are
// lattice positions of a box the host has checked to be finite, so only -0.0 can be odd, and only for the stencil.
// no_neg_zero (LATTICE only, wave-uniform): the host has checked that no coordinate of the lattice is -0.0
// (DCSG_SWEEP_NO_NEG_ZERO), so the per-pass test is skipped (3 vector instructions of the fused sweep's 150 per point).
template <bool LATTICE = false>
__device__ __forceinline__ void dcsg_stencil_spec(float x, float y, float z, bool want_sdf, bool want_normal, float& sdf,
                                                  float& nx, float& ny, float& nz, bool no_neg_zero = false)
{
    const float e = (float)K2_NORMAL_EPSILON;
    bool exact_path = false;
    if (!(LATTICE && no_neg_zero)) {
        const bool odd = LATTICE ? (want_normal && (dcsg_is_neg_zero(x) || dcsg_is_neg_zero(y) || dcsg_is_neg_zero(z)))
                                 : (dcsg_spec_odd(x) || dcsg_spec_odd(y) || dcsg_spec_odd(z));
        exact_path = __builtin_amdgcn_ballot_w64(odd) != 0ull;
    }
    if (exact_path) {
        const float zero = 0.0f;  // v+dx adds +0.0 on the other axes, v-dx subtracts it (k2.cl:167-173)
        if (want_sdf || !want_normal) sdf = dcsg_spec_exact(x, y, z);
        if (want_normal) {
            float Dx = dcsg_spec_exact(x + e, y + zero, z + zero) - dcsg_spec_exact(x - e, y - zero, z - zero);
            float Dy = dcsg_spec_exact(x + zero, y + e, z + zero) - dcsg_spec_exact(x - zero, y - e, z - zero);
            float Dz = dcsg_spec_exact(x + zero, y + zero, z + e) - dcsg_spec_exact(x - zero, y - zero, z - e);
            dcsg_finish_normal(Dx, Dy, Dz, nx, ny, nz);
        }
        return;
    }
    if (!want_normal) {
        sdf = dcsg_spec_eval1(x, y, z);
        return;
    }
    float xp = x + e, xm = x - e, yp = y + e, ym = y - e, zp = z + e, zm = z - e;
    dcsg_cl_float8 px = {x, xp, xm, x, x, x, x, 0.0f};
    dcsg_cl_float8 py = {y, y, y, yp, ym, y, y, 0.0f};
    dcsg_cl_float8 pz = {z, z, z, z, z, zp, zm, 0.0f};
    dcsg_cl_float8 o = dcsg_spec_eval7(px, py, pz);
    sdf = o[0];
    dcsg_finish_normal(o[1] - o[2], o[3] - o[4], o[5] - o[6], nx, ny, nz);
}
#endif


// ISV3D64::getPoint, ISV.hpp:103-109
__device__ __forceinline__ float dcsg_lattice_coord(const DcsgLattice& L, int axis, int i)
{
    float t = L.diam[axis] * (float)i;
    t = L.w_pow2 ? t * L.inv_w : t / L.w;
    return L.base[axis] + t;
}

// ISV3D64::getCoords, ISV.hpp:91-96 (float arithmetic, truncation toward zero)
__device__ __forceinline__ int dcsg_snap(const DcsgLattice& L, int axis, float p)
{
    float t = L.w * (p - L.center[axis] + L.half[axis]) / L.diam[axis];
    int i = (int)t;
    i = i < 0 ? 0 : (i > L.res ? L.res : i);
    return i;
}

// n / (res+1) for any 32-bit n by multiply-high (round-up method; magic and shift from the host)
__device__ __forceinline__ unsigned int dcsg_div_n1(const DcsgLattice& L, unsigned int n)
{
    unsigned int q = __umulhi(n, L.n1_magic);
    return (((n - q) >> 1) + q) >> L.n1_shift;
}

// ---------------------------------------------------------------------------
// Wave-level lattice indexing for the dense sweeps.
//
// A wave's 64 points are consecutive in the flat index, so (ix,iy,iz) of the wave's FIRST point is wave-uniform:
// it is decomposed once per wave on the scalar unit (s_mul_hi_u32 magic divisions, no VALU work), and a lane's
// point is (ix0 + lane, iy0, iz0) -- or, past the end of the row, the next row (rows hold res+1 >= 64 points, so a
// wave crosses at most one row end).  The three coordinates come from per-axis tables of ISV3D64::getPoint
// (ISV.hpp:103-109) that each block computes once into LDS with exactly the arithmetic of dcsg_lattice_coord:
// same bits, and the per-point cost of index -> position drops from ~30 VALU instructions (two per-lane magic
// divisions, three int -> float conversions, six multiplies / adds) to one add plus the LDS reads.
// Lattices with fewer than 64 points per row, or too large for the tables to fit in LDS (axes_off < 0), take the
// per-lane path.
// ---------------------------------------------------------------------------
// XO ("x only"): the block fills the x table alone and y / z, which are wave-uniform per row, are computed where they are
// needed (dcsg_axis) -- a third of the start-up work of a block, for the SDF-only sweep of per-scene code, whose 0.1 ms
// launch pays that start-up eight times per CU.
template <bool XO = false>
__device__ __forceinline__ const float* dcsg_stage_axes(const DcsgLattice& L, int axes_off)
{
    if (axes_off < 0 || L.res + 1 < DCSG_WAVE) return nullptr;
    float* tab = reinterpret_cast<float*>(reinterpret_cast<char*>(dcsg_lds) + axes_off);
    const int n1 = L.res + 1;
    for (int k = 0; k < (XO ? 1 : 3); k++)
        for (int i = threadIdx.x; i < n1; i += DCSG_BLOCK) tab[k * n1 + i] = dcsg_lattice_coord(L, k, i);
    __syncthreads();
    return tab;
}
// coordinate i of axis 1 (y) or 2 (z): from the table, or computed (same arithmetic, same bits)
template <bool XO>
__device__ __forceinline__ float dcsg_axis(const DcsgLattice& L, const float* ax, int axis, unsigned int i)
{
    return XO ? dcsg_lattice_coord(L, axis, (int)i) : ax[(unsigned int)axis * ((unsigned int)L.res + 1u) + i];
}

// Position of this lane's point when the wave's first point is (ix0, iy0) of lattice plane plane0 (all wave-uniform).
// `ax` = the LDS tables.  wrapped: the lane sits in the next row.
struct DcsgWavePoint {
    float x, y, z;
    bool wrapped;  // lane is past the end of the wave's first row
};
template <bool XO = false>
__device__ __forceinline__ DcsgWavePoint dcsg_wave_point(const DcsgLattice& L, const float* ax, unsigned int ix0, unsigned int iy0,
                                                         unsigned int plane0, unsigned int lane)
{
    const unsigned int n1 = (unsigned int)L.res + 1u;
    DcsgWavePoint w;
    if (ix0 + (DCSG_WAVE - 1u) < n1) {  // wave-uniform: the whole wave lies in one row
        w.x = ax[ix0 + lane];
        w.y = dcsg_axis<XO>(L, ax, 1, iy0);
        w.z = dcsg_axis<XO>(L, ax, 2, plane0);
        w.wrapped = false;
    } else {
        unsigned int iy1 = iy0 + 1u, plane1 = plane0;
        if (iy1 == n1) {
            iy1 = 0u;
            plane1 = plane0 + 1u;
        }
        // plane1 can be res + 1 only for lanes past the end of the lattice, which store nothing: keep the read in range
        const unsigned int plane1c = plane1 > (unsigned int)L.res ? (unsigned int)L.res : plane1;
        unsigned int ix = ix0 + lane;
        w.wrapped = ix >= n1;
        ix = w.wrapped ? ix - n1 : ix;
        w.x = ax[ix];
        w.y = dcsg_axis<XO>(L, ax, 1, w.wrapped ? iy1 : iy0);
        w.z = dcsg_axis<XO>(L, ax, 2, w.wrapped ? plane1c : plane0);
    }
    return w;
}
// (ix0, iy0, plane0) of the point 64 flat indices further on (scalar unit; rows hold >= 64 points)
__device__ __forceinline__ void dcsg_wave_advance(unsigned int n1, unsigned int& ix0, unsigned int& iy0, unsigned int& plane0)
{
    ix0 += DCSG_WAVE;
    if (ix0 >= n1) {
        ix0 -= n1;
        iy0 += 1u;
        if (iy0 == n1) {
            iy0 = 0u;
            plane0 += 1u;
        }
    }
}

// write-once streaming output: non-temporal stores (nothing in the sweeps reads the lattice back)
#define DCSG_STREAM_STORE(v, p) __builtin_nontemporal_store((v), (p))
// The sweeps' stores as "scalar base + 32-bit lane offset" (global_store ... s[base:base+1]): a wave's base is uniform (SGPRs,
// advanced by scalar adds), so a pass spends no vector instruction on addresses -- left to itself the compiler keeps
// "pointer + lane" in a VGPR pair and pays two 64-bit vector adds per pass (a v_mad_u64_u32 in the general path).  Sphere
// fused 513^3 0.458 -> 0.450 ms, 1025^3 3.54 -> 3.47 (profiles/r03_ab/r03_ab_stores2.txt, r03_ab_saddr.txt).  Used by the
// SDF + normal sweep of per-scene code only: the SDF-only sweep (+3-4 %) and the interpreters (+1 %) were slower with it.
// `base` MUST be wave-uniform.
// SGPR copy of a wave-uniform pointer (folds away where the compiler already knows the value to be uniform), made opaque
// by an empty asm so that the compiler cannot fold "base + lane" back into a loop-carried VGPR pointer: the stores below
// are then ordinary compiler-generated instructions of the form global_store ... v_off, v_data, s[base:base+1] (all of the
// chip's hazards -- the two wait states after a dwordx3 store, SGPRs written by v_readfirstlane -- are the compiler's to
// handle; a hand-written asm store was measured first and needed them spelled out).
typedef __attribute__((address_space(1))) float* dcsg_global_ptr;  // global memory, as the kernel arguments are
__device__ __forceinline__ dcsg_global_ptr dcsg_uniform_ptr(float* p)
{
    const unsigned long long u = (unsigned long long)p;
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)u);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(u >> 32));
    dcsg_global_ptr q = (dcsg_global_ptr)(((unsigned long long)hi << 32) | lo);
    asm("" : "+s"(q));
    return q;
}
typedef __attribute__((address_space(1))) char* dcsg_global_bytes;
// the lane's byte offset: loop-invariant (one VGPR across the passes), made opaque where it is used so that its zero-extension
// stays next to the store and the instruction selector sees "SGPR base + zext(32-bit VGPR)"
// (lane_bytes: the caller's variable, redefined in place -- no copy)
__device__ __forceinline__ dcsg_global_ptr dcsg_lane_address(float* base, unsigned int& lane_bytes)
{
    asm volatile("" : "+v"(lane_bytes));
    return (dcsg_global_ptr)((dcsg_global_bytes)dcsg_uniform_ptr(base) + lane_bytes);
}
// lane4 = 4 * lane, lane12 = 12 * lane: variables of the caller that live across its passes
__device__ __forceinline__ void dcsg_store_lane(float* base, unsigned int& lane4, float v)
{
    DCSG_STREAM_STORE(v, dcsg_lane_address(base, lane4));
}
__device__ __forceinline__ void dcsg_store3_lane(float* base, unsigned int& lane12, float a, float b, float c)
{
    dcsg_global_ptr o = dcsg_lane_address(base, lane12);
    DCSG_STREAM_STORE(a, o);
    DCSG_STREAM_STORE(b, o + 1);
    DCSG_STREAM_STORE(c, o + 2);
}

// ---------------------------------------------------------------------------
// Kernel 1: fused lattice sweep.  One thread per lattice point, x fastest, so a wave stores
// 256 contiguous bytes of SDF and 768 contiguous bytes of normals per pass; a wave owns a run of
// DCSG_RUN = 4 x 64 consecutive points (one scalar index decomposition per run, then scalar increments).
// `first` is the flat index of this launch's first point inside the slab buffer, plane z0 is the slab's first plane.
// Algorithmic traffic: 16 B written per point, nothing read (positions come from the index, the
// op and object tables from LDS).
// K = stack depth of the register-resident stack interpreter; K = 0 instantiates the generic
// LDS-stack interpreter (programs that are not stack-disciplined, or deeper than 8); K < 0 the per-scene
// specialised code.  MODE fixes the outputs at compile time (1: SDF only, 3: SDF + normals; no mode tests or dead
// paths in the loop); MODE 0 reads them off the pointers at run time (normals only, and the kernels kept small).
// ---------------------------------------------------------------------------
#define DCSG_RUN (4u * DCSG_WAVE)

template <int K, int MODE>
__device__ __forceinline__ void dcsg_sweep_body(const uint4* __restrict__ blob, int n_vec4, const DcsgLattice& L, int z0,
                                                unsigned int first, unsigned int n_points,
                                                float* __restrict__ sdf_out, float* __restrict__ nrm_out, int axes_off,
                                                unsigned int flags)
{
    const unsigned int n1 = (unsigned int)L.res + 1u;
    const unsigned int tile_points = DCSG_RUN * (DCSG_BLOCK / DCSG_WAVE);
    const unsigned int tiles = (n_points + tile_points - 1u) / tile_points;
    const unsigned int lane = threadIdx.x & (DCSG_WAVE - 1u);
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / DCSG_WAVE));
    // per-scene specialised code has the scene in its instructions: no tables to stage (a block's start-up is then the
    // axis tables alone -- the blob's global loads and their barrier cost every block about a microsecond of latency,
    // eight times per CU in a 0.1 ms SDF-only sweep: sphere 513^3 0.108 -> 0.104 ms).  Dropping the axis tables as well
    // (positions from dcsg_lattice_coord per point, no LDS, no barrier, finer grids) was measured and not kept: 0.110 ms
    // at 513^3 and 0.82 against 0.75 ms at 1025^3 (profiles/r03_ab_notables.txt).
    SceneLds s = K < 0 ? SceneLds{} : dcsg_stage_scene(blob, n_vec4);
    constexpr bool XO = K < 0 && MODE == 1;  // SDF-only sweep of per-scene code: x table only (dcsg_stage_axes)
    const float* ax = dcsg_stage_axes<XO>(L, axes_off);
    const bool want_sdf = MODE == 0 ? sdf_out != nullptr : true;
    const bool want_nrm = MODE == 0 ? nrm_out != nullptr : MODE == 3;
    const bool no_neg_zero = (flags & DCSG_SWEEP_NO_NEG_ZERO) != 0u;  // host-checked: no -0.0 coordinate on this lattice
    unsigned int lane4 = lane * 4u, lane12 = lane * 12u;              // byte offsets of the lane's outputs (dcsg_store_lane)
    // (Tiles of T consecutive 1 024-point groups, to decouple the number of blocks from the number of write fronts, were
    // measured and not kept: profiles/r03_ab/r03_ab_groups.txt.)
    for (unsigned int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const unsigned int local0 = tile * tile_points + wave * DCSG_RUN;  // wave-uniform from here ...
        if (local0 >= n_points) continue;
        const unsigned int left = n_points - local0;
        unsigned int idx0 = first + local0;
        unsigned int ix0 = 0u, iy0 = 0u, plane0 = 0u;
        if (ax) {
            const unsigned int t0 = dcsg_div_n1(L, idx0);
            ix0 = idx0 - t0 * n1;
            const unsigned int iz0 = dcsg_div_n1(L, t0);
            iy0 = t0 - iz0 * n1;
            plane0 = iz0 + (unsigned int)z0;                                // ... to here
        }
        // Stack interpreters, SDF only: ONE interpretation of the program serves the run's four 64-point passes (the lane's
        // point of each) -- the step decode (LDS record, readfirstlane, opcode / brush / depth branches: scalar work,
        // ~35 instructions per step) is paid per 256 points instead of per 64: sphere 513^3 0.546 -> 0.234 ms.  Same
        // arithmetic per point: same bits.  (SDF + normals, two passes x seven stencil points per interpretation: the
        // kernel's registers go from 95 to 184 and the sweep from 1.04 to 1.45 ms -- measured, not kept.)
        if (K > 0 && MODE == 1 && ax && left >= DCSG_RUN && (flags & DCSG_SWEEP_MULTIPASS)) {
            float px[4], py[4], pz[4], o[4];
            if (ix0 + DCSG_RUN <= n1) {  // the run lies inside one row (wave-uniform): one y, one z, no per-pass row tests
                const float y = dcsg_axis<XO>(L, ax, 1, iy0), z = dcsg_axis<XO>(L, ax, 2, plane0);
                const float* xr = ax + ix0 + lane;
#pragma unroll
                for (int j = 0; j < 4; j++) px[j] = xr[64 * j], py[j] = y, pz[j] = z;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    DcsgWavePoint w = dcsg_wave_point<XO>(L, ax, ix0, iy0, plane0, lane);
                    px[j] = w.x, py[j] = w.y, pz[j] = w.z;
                    dcsg_wave_advance(n1, ix0, iy0, plane0);
                }
            }
            dcsg_interp<(K <= 0 ? 4 : K), 4>(s, px, py, pz, o);
#pragma unroll
            for (int j = 0; j < 4; j++) DCSG_STREAM_STORE(o[j], sdf_out + idx0 + lane + 64 * j);
            continue;
        }
#ifdef DCSG_HAVE_SPEC
        // Threaded per-scene code (large programs: a chain of calls, one per object), SDF only: the run's four passes go
        // through ONE chain (dcsg_spec_eval4) -- a call's cost (its
        // constants' loads, the jump) is paid per 256 points instead of per 64, and four evaluations overlap.  Same
        // arithmetic per point (the threaded form writes every term out: valid for any coordinate): same bits.
        if (K < 0 && MODE == 1 && ax && left >= DCSG_RUN && dcsg_spec_threaded()) {
            float px[4], py[4], pz[4];
            if (ix0 + DCSG_RUN <= n1) {
                const float y = dcsg_axis<XO>(L, ax, 1, iy0), z = dcsg_axis<XO>(L, ax, 2, plane0);
                const float* xr = ax + ix0 + lane;
#pragma unroll
                for (int j = 0; j < 4; j++) px[j] = xr[64 * j], py[j] = y, pz[j] = z;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    DcsgWavePoint w = dcsg_wave_point<XO>(L, ax, ix0, iy0, plane0, lane);
                    px[j] = w.x, py[j] = w.y, pz[j] = w.z;
                    dcsg_wave_advance(n1, ix0, iy0, plane0);
                }
            }
            const dcsg_cl_float4 vx = {px[0], px[1], px[2], px[3]}, vy = {py[0], py[1], py[2], py[3]}, vz = {pz[0], pz[1], pz[2], pz[3]};
            const dcsg_cl_float4 o = dcsg_spec_eval4(vx, vy, vz);
#pragma unroll
            for (int j = 0; j < 4; j++) DCSG_STREAM_STORE(o[j], sdf_out + idx0 + lane + 64 * j);
            continue;
        }
        // Small scenes: passes that lie inside one lattice row share y and z, so everything the seven stencil evaluations
        // derive from those two alone is computed once for the passes of a row segment (loop-invariant code motion: same
        // operations, same bits), and the per-pass row tests disappear.
        auto row_pass = [&](const float* xrow, float y, float z, unsigned int k) {
            float sd = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
            dcsg_stencil_spec<true>(xrow[k * DCSG_WAVE], y, z, true, MODE == 3, sd, nx, ny, nz, no_neg_zero);
            if (MODE == 3) {
                dcsg_store_lane(sdf_out + idx0 + k * DCSG_WAVE, lane4, sd);
                dcsg_store3_lane(nrm_out + ((size_t)idx0 + k * DCSG_WAVE) * 3, lane12, nx, ny, nz);
            } else {
                DCSG_STREAM_STORE(sd, (sdf_out + idx0) + lane + k * DCSG_WAVE);
            }
        };
        if (K < 0 && MODE == 1 && ax && left >= DCSG_RUN && ix0 + DCSG_RUN <= n1 && dcsg_spec_small()) {
            // SDF only: a whole run inside one row (half of the runs at 513 points per row, three quarters at 1025), its
            // four x reads issued together, four passes unrolled
            const float y = dcsg_axis<XO>(L, ax, 1, iy0), z = dcsg_axis<XO>(L, ax, 2, plane0);
            const float* xrow = ax + ix0 + lane;
#pragma unroll
            for (unsigned int k = 0u; k < DCSG_RUN / DCSG_WAVE; k++) row_pass(xrow, y, z, k);
            continue;
        }
#endif
        auto pass = [&](unsigned int done) {
            float x, y, z;
            if (ax) {
                DcsgWavePoint w = dcsg_wave_point<XO>(L, ax, ix0, iy0, plane0, lane);
                x = w.x, y = w.y, z = w.z;
                dcsg_wave_advance(n1, ix0, iy0, plane0);
            } else {
                unsigned int idx = idx0 + lane;
                unsigned int t = dcsg_div_n1(L, idx);
                unsigned int ix = idx - t * n1;
                unsigned int iz = dcsg_div_n1(L, t);
                unsigned int iy = t - iz * n1;
                x = dcsg_lattice_coord(L, 0, (int)ix);
                y = dcsg_lattice_coord(L, 1, (int)iy);
                z = dcsg_lattice_coord(L, 2, (int)iz + z0);
            }
            const unsigned int at = idx0;  // this pass's first point; advanced here, ahead of the divergent return, so that
            idx0 += DCSG_WAVE;             // idx0 stays a wave-uniform value for the compiler
            if (left - done < DCSG_WAVE && lane >= left - done) return;  // the launch's last wave
            float sd = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
            if (K == 0) {
                if (want_sdf) sd = dcsg_primary_sdf(s, x, y, z);
                if (want_nrm) dcsg_get_normal(s, x, y, z, nx, ny, nz);
#ifdef DCSG_HAVE_SPEC
            } else if (K < 0) {
                dcsg_stencil_spec<true>(x, y, z, want_sdf, want_nrm, sd, nx, ny, nz, no_neg_zero);
#endif
            } else {
                dcsg_stencil<(K <= 0 ? 4 : K)>(s, x, y, z, want_sdf, want_nrm, sd, nx, ny, nz);
            }
            // write-once streaming output: non-temporal stores (nothing in this kernel reads the lattice back)
            if (K < 0 && MODE == 3) {
                dcsg_store_lane(sdf_out + at, lane4, sd);
                dcsg_store3_lane(nrm_out + (size_t)at * 3, lane12, nx, ny, nz);
            } else {
                // wave-uniform base + lane: the store addresses cost no per-lane multiplies
                if (want_sdf) DCSG_STREAM_STORE(sd, (sdf_out + at) + lane);
                if (want_nrm) {
                    float* o = nrm_out + (size_t)at * 3 + lane * 3u;
                    DCSG_STREAM_STORE(nx, o);
                    DCSG_STREAM_STORE(ny, o + 1);
                    DCSG_STREAM_STORE(nz, o + 2);
                }
            }
        };
#ifdef DCSG_HAVE_SPEC
        if (K < 0 && MODE == 3 && ax && left >= DCSG_RUN && dcsg_spec_small()) {
            // SDF + normals: the run is cut into row segments (the passes up to the end of the current row); a pass that
            // straddles a row end takes the general code.  7 of 8 passes are inside a row at 513 points per row.
            unsigned int k = 0u;
            while (k < DCSG_RUN / DCSG_WAVE) {
                unsigned int m = (n1 - ix0) / DCSG_WAVE;  // whole passes left in this row (ix0 < n1)
                if (m > DCSG_RUN / DCSG_WAVE - k) m = DCSG_RUN / DCSG_WAVE - k;
                if (m == 0u) {
                    pass(k * DCSG_WAVE);
                    k++;
                    continue;
                }
                const float y = dcsg_axis<XO>(L, ax, 1, iy0), z = dcsg_axis<XO>(L, ax, 2, plane0);
                const float* xrow = ax + ix0 + lane;
#pragma unroll 1
                for (unsigned int q = 0u; q < m; q++) row_pass(xrow, y, z, q);
                idx0 += m * DCSG_WAVE;
                ix0 += m * DCSG_WAVE;
                if (ix0 >= n1) {  // only when the row length is a multiple of 64
                    ix0 -= n1;
                    iy0 += 1u;
                    if (iy0 == n1) {
                        iy0 = 0u;
                        plane0 += 1u;
                    }
                }
                k += m;
            }
            continue;
        }
#endif
        if (K < 0 && MODE == 1 && left >= DCSG_RUN) {
            // the cheapest kernel (SDF only, per-scene code): four independent passes in flight hide the square root's
            // latency and quarter the loop overhead
#pragma unroll
            for (unsigned int done = 0u; done < DCSG_RUN; done += DCSG_WAVE) pass(done);
        } else {
#pragma unroll 1
            for (unsigned int done = 0u; done < DCSG_RUN && done < left; done += DCSG_WAVE) pass(done);
        }
    }
}

// SDF-only kernels of per-scene code (K < 0, MODE 1): at least 8 waves per SIMD, i.e. at most 64 VGPRs -- the allocator
// otherwise lets the four unrolled passes drift to 65, one wave per SIMD less (sphere: 23 under the bound, Design2: 47,
// no spill).  Not for SDF + normals: Design2's seven stencil evaluations of a 63-box brush spill under it.
#ifndef DCSG_SWEEP_MIN_WAVES
#define DCSG_SWEEP_MIN_WAVES 1  // A/B builds: minimum waves per SIMD asked of the other sweep kernels
#endif
#define DCSG_SWEEP_KERNEL(NAME, K, MODE)                                                                            \
    extern "C" __global__ void __launch_bounds__(DCSG_BLOCK, ((K) < 0 && (MODE) == 1 ? 8 : DCSG_SWEEP_MIN_WAVES))   \
    NAME(const uint4* __restrict__ blob, int n_vec4, DcsgLattice L, int z0, unsigned int first,                     \
         unsigned int n_points, float* __restrict__ sdf_out, float* __restrict__ nrm_out, int axes_off,             \
         unsigned int flags)                                                                                        \
    {                                                                                                               \
        dcsg_sweep_body<K, MODE>(blob, n_vec4, L, z0, first, n_points, sdf_out, nrm_out, axes_off, flags);          \
    }
// NAME: outputs chosen at run time; NAME_d: SDF only; NAME_f: SDF + normals
#define DCSG_SWEEP_KERNELS(NAME, K)                                                                                 \
    DCSG_SWEEP_KERNEL(NAME, K, 0)                                                                                   \
    DCSG_SWEEP_KERNEL(NAME##_d, K, 1)                                                                               \
    DCSG_SWEEP_KERNEL(NAME##_f, K, 3)
#ifdef DCSG_PART_REST
#ifndef DCSG_SET_SPEC  // the lean module of a specialised scene holds the _sp kernels only (csrc/jit.cpp)
DCSG_SWEEP_KERNELS(dcsg_sweep, 0)
DCSG_SWEEP_KERNELS(dcsg_sweep_s4, 4)
DCSG_SWEEP_KERNELS(dcsg_sweep_s8, 8)
#endif
#ifdef DCSG_HAVE_SPEC
DCSG_SWEEP_KERNELS(dcsg_sweep_sp, -1)
#endif
#endif  // DCSG_PART_REST


// ---------------------------------------------------------------------------
// Node-wise sweep (dcsg_export): the lattice values of the octree nodes of one level -- the nodes that reached the
// mesher at the first level where the cull rule can drop a node for good -- instead of the whole lattice.  This is
// the reference's lazy block cache (ISV.hpp:15-60) turned into one launch: a node of level m covers (2^(g-m)+1)^3
// lattice points, of which the sweep writes those the mesher can read (dcsg_in_sweep_pattern); blocks take the node list
// eight siblings at a time (dcsg_node_group); points shared with a neighbour are written by both (same bits).  The node's
// index on the level's grid comes from its snapped centre; node_mask records it for dcsg_lattice_point_written.
// ---------------------------------------------------------------------------
// i0: first lattice index of node n per axis; ic: the lattice point its centre snaps to (what the mesher's cull test
// samples, mesh.hpp:167); slot: the node's byte in node_mask.
__device__ __forceinline__ void dcsg_node_origin(const DcsgLattice& L, const float* __restrict__ nodes, unsigned int n, int shift,
                                                 int level, int (&i0)[3], int (&ic)[3], size_t& slot)
{
    const int N = 1 << level;
    int a[3];
    for (int k = 0; k < 3; k++) {
        ic[k] = dcsg_snap(L, k, nodes[(size_t)n * 3 + k]);
        a[k] = ic[k] >> shift;
        a[k] = a[k] > N - 1 ? N - 1 : a[k];
        i0[k] = a[k] << shift;
    }
    slot = ((size_t)a[2] * N + (size_t)a[1]) * N + (size_t)a[0];
}
// The cull rule of mesh.hpp:167-170 on the value at the node's snapped centre: a node it drops is not swept (the
// mesher then samples that one point on demand and drops the node itself).
__device__ __forceinline__ bool dcsg_node_culled(float d, float hx, float hy, float hz)
{
    float mag = dcsg_sqrt(hx * hx + hy * hy + hz * hz);
    return __builtin_fabsf(d) > mag * 1.1f;
}

// Z-slab exports (dcsg_export_slab): the level's node list also holds nodes outside this rank's slab -- all 64 roots at
// level 2, every child of a parent that straddles slabs -- and the slab buffer holds planes z_own_lo .. z_own_hi only.
// A node whose planes [i0z, i0z + 2^shift] do not lie inside the slab is not swept: nothing is stored for it and its
// mask byte stays 0 (the mesher classifies such a node as "outside this rank" before it samples anything).
__device__ __forceinline__ bool dcsg_node_outside_slab(int i0z, int shift, int z_own_lo, int z_own_hi)
{
    return i0z < z_own_lo || i0z + (1 << shift) > z_own_hi;
}

// Local point index p of a node block -> (lx, ly, lz), p = (lz*side + ly)*side + lx, without integer division:
// side = 2^shift + 1 <= 33 and p < side^3 < 2^16, for which floor(p * (floor(2^32/d) + 1) / 2^32) = p / d exactly.
// `pattern` (n_pattern > 0): the node's points that the mesher can read, as lx | ly << 5 | lz << 10 -- the sweep then writes
// those and nothing else (dcsg_in_sweep_pattern has the rule; the host builds the list from the same rule).
struct DcsgNodeGrid {
    unsigned int side, ss, pts, m_side, m_ss;
    const unsigned short* pattern;
};
__device__ __forceinline__ DcsgNodeGrid dcsg_node_grid(int shift, const unsigned short* pattern = nullptr, unsigned int n_pattern = 0u)
{
    DcsgNodeGrid g;
    g.side = (1u << shift) + 1u;
    g.ss = g.side * g.side;
    g.pts = n_pattern ? n_pattern : g.ss * g.side;
    g.m_side = 0xffffffffu / g.side + 1u;
    g.m_ss = 0xffffffffu / g.ss + 1u;
    g.pattern = n_pattern ? pattern : nullptr;
    return g;
}
__device__ __forceinline__ void dcsg_node_local(const DcsgNodeGrid& g, unsigned int p, unsigned int& lx, unsigned int& ly, unsigned int& lz)
{
    if (g.pattern) {  // kernel argument: uniform
        const unsigned int v = g.pattern[p];
        lx = v & 31u, ly = (v >> 5) & 31u, lz = v >> 10;
        return;
    }
    lz = __umulhi(p, g.m_ss);
    unsigned int r = p - lz * g.ss;
    ly = __umulhi(r, g.m_side);
    lx = r - ly * g.side;
}
// The nodes of a group (eight siblings, or one node) that are swept, and the flattened walk over their points.
// Phase 1 -- every WAVE on its own, no barrier: lane l looks at node l & 7 of the group (origin, slab, the cull rule on the
// value at its snapped centre; one evaluation serves the eight centres), the ballot's low eight bits are the nodes to sweep.
// Phase 2: the block walks the kept nodes' points as ONE index range t = rank * pts + p (rank-th kept node, p-th point of
// the node's pattern), 256 or 1 024 at a time, so that a node of 297 points does not leave half of a second pass idle; the
// node's origin comes from the lane that looked at it (lane exchange), the coordinates from the lattice's own formula.
struct DcsgNodeGroup {
    unsigned int mask;      // bit k: node k of the group is swept (wave-uniform)
    int ox, oy, oz;         // lane l: first lattice index of node l & 7
};
template <class CentreEval>
__device__ __forceinline__ DcsgNodeGroup dcsg_node_group(const DcsgLattice& L, const float* __restrict__ nodes, unsigned int n_first,
                                                         unsigned int n_end, int shift, int level, float hdx, float hdy, float hdz,
                                                         int z_own_lo, int z_own_hi, unsigned char* __restrict__ node_mask, CentreEval eval)
{
    const unsigned int k = threadIdx.x & 7u;
    const bool exists = n_first + k < n_end;
    int i0[3], ic[3];
    size_t slot;
    dcsg_node_origin(L, nodes, exists ? n_first + k : n_first, shift, level, i0, ic, slot);
    const bool inside = exists && !dcsg_node_outside_slab(i0[2], shift, z_own_lo, z_own_hi);
    const float d = eval(dcsg_lattice_coord(L, 0, ic[0]), dcsg_lattice_coord(L, 1, ic[1]), dcsg_lattice_coord(L, 2, ic[2]));
    const bool keep = inside && !dcsg_node_culled(d, hdx, hdy, hdz);
    DcsgNodeGroup G;
    G.mask = (unsigned int)(__builtin_amdgcn_ballot_w64(keep) & 0xffull);
    G.ox = i0[0], G.oy = i0[1], G.oz = i0[2];
    if (threadIdx.x < 8u && keep) node_mask[slot] = 1;  // lanes 0 .. 7 of the block's first wave
    return G;
}
// point t of the walk (t < popcount(mask) * pts, clamped by the caller): its coordinates and its place in the lattice
__device__ __forceinline__ void dcsg_node_group_point(const DcsgLattice& L, const DcsgNodeGrid& g, const DcsgNodeGroup& G, unsigned int m_pts,
                                                      unsigned int t, float& x, float& y, float& z, size_t& at)
{
    const unsigned int rank = __umulhi(t, m_pts), p = t - rank * g.pts;  // t / pts, exact below 2^16
    unsigned int c = G.mask;
    for (unsigned int i = 0u; i < rank; i++) c &= c - 1u;
    const int child = __builtin_ctz(c | 0x100u);
    unsigned int lx, ly, lz;
    dcsg_node_local(g, p, lx, ly, lz);
    const int ix = __shfl(G.ox, child, DCSG_WAVE) + (int)lx, iy = __shfl(G.oy, child, DCSG_WAVE) + (int)ly,
              iz = __shfl(G.oz, child, DCSG_WAVE) + (int)lz;
    x = dcsg_lattice_coord(L, 0, ix), y = dcsg_lattice_coord(L, 1, iy), z = dcsg_lattice_coord(L, 2, iz);
    const size_t n1 = (size_t)L.res + 1u;
    at = ((size_t)iz * n1 + (size_t)iy) * n1 + (size_t)ix;
}

template <int K>
__device__ __forceinline__ void dcsg_sweep_nodes_body(const uint4* __restrict__ blob, int n_vec4, const DcsgLattice& L,
                                                      const float* __restrict__ nodes, unsigned int n_nodes, int shift, int level,
                                                      float* __restrict__ sdf_out, float* __restrict__ nrm_out,
                                                      unsigned char* __restrict__ node_mask, float hdx, float hdy, float hdz,
                                                      int z_own_lo, int z_own_hi, unsigned int* __restrict__ n_swept, int group,
                                                      const unsigned short* __restrict__ pattern, unsigned int n_pattern)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    const DcsgNodeGrid g = dcsg_node_grid(shift, pattern, n_pattern);
    const unsigned int m_pts = 0xffffffffu / g.pts + 1u;
    const unsigned int per = (unsigned int)group, n_groups = (n_nodes + per - 1u) / per;
    unsigned int swept_here = 0u;  // block-uniform; one atomic per block at the end (one per node, all on the same word, retire one at a time)
    for (unsigned int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const unsigned int n_first = grp * per, n_end = n_first + per < n_nodes ? n_first + per : n_nodes;
        const DcsgNodeGroup G = dcsg_node_group(L, nodes, n_first, n_end, shift, level, hdx, hdy, hdz, z_own_lo, z_own_hi, node_mask,
                                                [&](float cx, float cy, float cz) {
                                                    float d = 0.0f, u0, u1, u2;
                                                    if (K == 0) d = dcsg_primary_sdf(s, cx, cy, cz);
#ifdef DCSG_HAVE_SPEC
                                                    else if (K < 0) dcsg_stencil_spec<true>(cx, cy, cz, true, false, d, u0, u1, u2);
#endif
                                                    else dcsg_stencil<(K <= 0 ? 4 : K)>(s, cx, cy, cz, true, false, d, u0, u1, u2);
                                                    return d;
                                                });
        if (G.mask == 0u) continue;
        swept_here += (unsigned int)__popc(G.mask);
        const unsigned int total = (unsigned int)__popc(G.mask) * g.pts;
        unsigned int t0 = 0u;
#ifdef DCSG_HAVE_SPEC
        if (K < 0 && sdf_out && !nrm_out && dcsg_spec_threaded()) {
            // threaded per-scene code, SDF only: four points per lane through ONE chain of calls (dcsg_spec_eval4), while at
            // least three of a pass's four quarters have work
            for (; t0 + 3u * DCSG_BLOCK <= total; t0 += 4u * DCSG_BLOCK) {
                dcsg_cl_float4 vx, vy, vz;
                size_t at[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned int t = t0 + (unsigned int)j * DCSG_BLOCK + threadIdx.x;
                    float x, y, z;
                    dcsg_node_group_point(L, g, G, m_pts, t < total ? t : total - 1u, x, y, z, at[j]);  // idle lanes: the last point, nothing stored
                    vx[j] = x, vy[j] = y, vz[j] = z;
                }
                const dcsg_cl_float4 o = dcsg_spec_eval4(vx, vy, vz);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (t0 + (unsigned int)j * DCSG_BLOCK + threadIdx.x < total) sdf_out[at[j]] = o[j];
            }
        }
#endif
        for (; t0 < total; t0 += DCSG_BLOCK) {
            const unsigned int t = t0 + threadIdx.x;
            float x, y, z;
            size_t idx;
            dcsg_node_group_point(L, g, G, m_pts, t < total ? t : total - 1u, x, y, z, idx);
            float sd = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
            if (K == 0) {
                if (sdf_out) sd = dcsg_primary_sdf(s, x, y, z);
                if (nrm_out) dcsg_get_normal(s, x, y, z, nx, ny, nz);
#ifdef DCSG_HAVE_SPEC
            } else if (K < 0) {
                dcsg_stencil_spec<true>(x, y, z, sdf_out != nullptr, nrm_out != nullptr, sd, nx, ny, nz);
#endif
            } else {
                dcsg_stencil<(K <= 0 ? 4 : K)>(s, x, y, z, sdf_out != nullptr, nrm_out != nullptr, sd, nx, ny, nz);
            }
            if (t < total) {
                if (sdf_out) sdf_out[idx] = sd;
                if (nrm_out) {
                    float* o = nrm_out + idx * 3;
                    o[0] = nx;
                    o[1] = ny;
                    o[2] = nz;
                }
            }
        }
    }
    if (threadIdx.x == 0 && swept_here) atomicAdd(n_swept, swept_here);
}

#define DCSG_SWEEP_NODES_KERNEL(NAME, K)                                                                             \
    extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)                                                         \
    NAME(const uint4* __restrict__ blob, int n_vec4, DcsgLattice L, const float* __restrict__ nodes,                 \
         unsigned int n_nodes, int shift, int level, float* __restrict__ sdf_out, float* __restrict__ nrm_out,       \
         unsigned char* __restrict__ node_mask, float hdx, float hdy, float hdz, int z_own_lo, int z_own_hi,        \
         unsigned int* __restrict__ n_swept, int group,                                                             \
         const unsigned short* __restrict__ pattern, unsigned int n_pattern)                                         \
    {                                                                                                                \
        dcsg_sweep_nodes_body<K>(blob, n_vec4, L, nodes, n_nodes, shift, level, sdf_out, nrm_out, node_mask, hdx,    \
                                 hdy, hdz, z_own_lo, z_own_hi, n_swept, group, pattern, n_pattern);                  \
    }
#ifdef DCSG_PART_EXPORT
#ifndef DCSG_SET_SPEC  // the lean module of a specialised scene holds the _sp kernels only (csrc/jit.cpp)
DCSG_SWEEP_NODES_KERNEL(dcsg_sweep_nodes, 0)
DCSG_SWEEP_NODES_KERNEL(dcsg_sweep_nodes_s4, 4)
DCSG_SWEEP_NODES_KERNEL(dcsg_sweep_nodes_s8, 8)
#endif
#ifdef DCSG_HAVE_SPEC
DCSG_SWEEP_NODES_KERNEL(dcsg_sweep_nodes_sp, -1)

#endif
#endif  // DCSG_PART_EXPORT


// ---------------------------------------------------------------------------
// Kernel 2: k2-equivalent point evaluation (Evaluator::eval_sdf_at_points /
// eval_normal_at_points).  mode 0 = SDF (1 float/pt), 1 = normal (3 floats/pt).
// ---------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void dcsg_eval_points_body(const uint4* __restrict__ blob, int n_vec4,
                                                      const float* __restrict__ xyz, unsigned int n, int mode,
                                                      float* __restrict__ out)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    const unsigned int tiles = (n + DCSG_BLOCK - 1) / DCSG_BLOCK;
    for (unsigned int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        unsigned int i = tile * DCSG_BLOCK + threadIdx.x;
        if (i >= n) continue;
        float x = xyz[(size_t)i * 3 + 0], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
        float sd = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
        if (K == 0) {
            if (mode == 0) sd = dcsg_primary_sdf(s, x, y, z);
            else dcsg_get_normal(s, x, y, z, nx, ny, nz);
#ifdef DCSG_HAVE_SPEC
        } else if (K < 0) {
            dcsg_stencil_spec(x, y, z, mode == 0, mode != 0, sd, nx, ny, nz);
#endif
        } else {
            dcsg_stencil<(K <= 0 ? 4 : K)>(s, x, y, z, mode == 0, mode != 0, sd, nx, ny, nz);
        }
        if (mode == 0) {
            out[i] = sd;
        } else {
            out[(size_t)i * 3 + 0] = nx;
            out[(size_t)i * 3 + 1] = ny;
            out[(size_t)i * 3 + 2] = nz;
        }
    }
}

#define DCSG_EVAL_KERNEL(NAME, K)                                                                                   \
    extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)                                                        \
    NAME(const uint4* __restrict__ blob, int n_vec4, const float* __restrict__ xyz, unsigned int n, int mode,       \
         float* __restrict__ out)                                                                                   \
    {                                                                                                               \
        dcsg_eval_points_body<K>(blob, n_vec4, xyz, n, mode, out);                                                  \
    }
#ifdef DCSG_PART_REST
#ifndef DCSG_SET_SPEC  // the lean module of a specialised scene holds the _sp kernels only (csrc/jit.cpp)
DCSG_EVAL_KERNEL(dcsg_eval_points, 0)
DCSG_EVAL_KERNEL(dcsg_eval_points_s4, 4)
DCSG_EVAL_KERNEL(dcsg_eval_points_s8, 8)
#endif
#ifdef DCSG_HAVE_SPEC
DCSG_EVAL_KERNEL(dcsg_eval_points_sp, -1)
#endif
#endif  // DCSG_PART_REST


// ---------------------------------------------------------------------------
// Kernel 3: stage A bounding-box probe, DesignCSG.cpp:668-712.
// 256^3 points, p = -cd/2 + cd*(float)i, i in [-128,127], z innermost.  Position is
// monotone in the index, so min/max positions reduce as integer indices.
// result: int[0..2] min index, int[3..5] max index (both seeded by the host with the
// "no interior point" sentinels), unsigned long long at byte 32: interior count.
// ---------------------------------------------------------------------------
// Stage A's reduction: wave min / max / sum, one combine per block through LDS, then at most seven atomics per
// block -- and none from blocks that saw no interior point (most of the probe cube).  Same-address atomics
// serialise in L2: issued per wave they cost more than the 16.7 M evaluations (1.34 ms -> see DESIGN.md).
// zh: the block's histogram over the 256 probe planes in z of the probes NEAR the surface (|sdf| < one probe spacing), added
// to result[16 .. 271]: what a Z-slab export balances its cuts on (work follows the surface, and every rank computes the
// same integers without talking to anyone).
__device__ __forceinline__ void dcsg_bbox_commit(int (&mn)[3], int (&mx)[3], unsigned int cnt, int* __restrict__ result,
                                                 unsigned int* zh)
{
    __shared__ int part[DCSG_BLOCK / DCSG_WAVE][8];
    // DCSG_BBOX_COPIES copies of the result, a block adding to copy blockIdx % copies and the host reducing them: every block
    // of the launch lands its seven words and its share of the 256 histogram bins on the same few cache lines, and atomics
    // on one line retire one at a time (sphere: 0.2 ms of stage A at 16 blocks per CU, 0.4 at 32, for 0.1 ms of probes)
    result += (blockIdx.x % DCSG_BBOX_COPIES) * DCSG_BBOX_WORDS;
    for (int k = 0; k < 3; k++) {
        mn[k] = wave_min(mn[k]);
        mx[k] = wave_max(mx[k]);
    }
    cnt = wave_sum(cnt);
    const int wave = threadIdx.x / DCSG_WAVE;
    if ((threadIdx.x & (DCSG_WAVE - 1)) == 0) {
        for (int k = 0; k < 3; k++) {
            part[wave][k] = mn[k];
            part[wave][3 + k] = mx[k];
        }
        part[wave][6] = (int)cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int total = 0;
        for (int w = 0; w < DCSG_BLOCK / DCSG_WAVE; w++) {
            for (int k = 0; k < 3; k++) {
                mn[k] = part[w][k] < mn[k] ? part[w][k] : mn[k];
                mx[k] = part[w][3 + k] > mx[k] ? part[w][3 + k] : mx[k];
            }
            total += (unsigned int)part[w][6];
        }
        if (total) {
            for (int k = 0; k < 3; k++) {
                atomicMin(&result[k], mn[k]);
                atomicMax(&result[3 + k], mx[k]);
            }
            atomicAdd(reinterpret_cast<unsigned long long*>(result + 8), (unsigned long long)total);
        }
    }
    // (the barrier above also orders every thread's zh updates before these reads)
    if (zh[threadIdx.x]) atomicAdd(reinterpret_cast<unsigned int*>(result) + 16 + threadIdx.x, zh[threadIdx.x]);
}
#define DCSG_BBOX_ZHIST_DECL                    \
    __shared__ unsigned int zh[DCSG_BLOCK];     \
    zh[threadIdx.x] = 0u;                       \
    __syncthreads();

template <int K>
__device__ __forceinline__ void dcsg_bbox_body(const uint4* __restrict__ blob, int n_vec4, float cd, float half_neg,
                                               double bb_eps, int* __restrict__ result)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    DCSG_BBOX_ZHIST_DECL
    const unsigned int total = 256u * 256u * 256u;
    int mn[3] = {1 << 20, 1 << 20, 1 << 20};
    int mx[3] = {-(1 << 20), -(1 << 20), -(1 << 20)};
    unsigned int cnt = 0;
    auto probe = [&](unsigned int i, float& x, float& y, float& z, int& ix, int& iy, int& iz) {
        iz = (int)(i & 255u) - 128;
        iy = (int)((i >> 8) & 255u) - 128;
        ix = (int)(i >> 16) - 128;
        x = half_neg + cd * (float)ix;
        y = half_neg + cd * (float)iy;
        z = half_neg + cd * (float)iz;
    };
    auto account = [&](float v, int ix, int iy, int iz) {
        if ((double)__builtin_fabsf(v) < bb_eps) atomicAdd(&zh[iz + 128], 1u);
        if ((double)v < bb_eps) {  // "p.second < BB_EPSILON", float vs double
            cnt++;
            mn[0] = ix < mn[0] ? ix : mn[0];
            mx[0] = ix > mx[0] ? ix : mx[0];
            mn[1] = iy < mn[1] ? iy : mn[1];
            mx[1] = iy > mx[1] ? iy : mx[1];
            mn[2] = iz < mn[2] ? iz : mn[2];
            mx[2] = iz > mx[2] ? iz : mx[2];
        }
    };
#ifdef DCSG_HAVE_SPEC
    if (K < 0 && dcsg_spec_threaded()) {
        // threaded per-scene code: four probes per lane through one chain of calls (the launch's grid is a power of two, so
        // the strided loop covers the 2^24 probes in whole groups of four)
        const unsigned int stride = gridDim.x * DCSG_BLOCK;
        if (total % (4u * stride) == 0u) {
            for (unsigned int i = blockIdx.x * DCSG_BLOCK + threadIdx.x; i < total; i += 4u * stride) {
                dcsg_cl_float4 vx, vy, vz;
                int jx[4], jy[4], jz[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float x, y, z;
                    probe(i + (unsigned int)j * stride, x, y, z, jx[j], jy[j], jz[j]);
                    vx[j] = x, vy[j] = y, vz[j] = z;
                }
                const dcsg_cl_float4 o = dcsg_spec_eval4(vx, vy, vz);
#pragma unroll
                for (int j = 0; j < 4; j++) account(o[j], jx[j], jy[j], jz[j]);
            }
            dcsg_bbox_commit(mn, mx, cnt, result, zh);
            return;
        }
    }
#endif
    for (unsigned int i = blockIdx.x * DCSG_BLOCK + threadIdx.x; i < total; i += gridDim.x * DCSG_BLOCK) {
        int ix, iy, iz;
        float x, y, z;
        probe(i, x, y, z, ix, iy, iz);
        float v, u0, u1, u2;
        if (K == 0) v = dcsg_primary_sdf(s, x, y, z);
#ifdef DCSG_HAVE_SPEC
        else if (K < 0) v = dcsg_spec_eval1(x, y, z);
#endif
        else dcsg_stencil<(K <= 0 ? 4 : K)>(s, x, y, z, true, false, v, u0, u1, u2);
        account(v, ix, iy, iz);
    }
    dcsg_bbox_commit(mn, mx, cnt, result, zh);
}

#define DCSG_BBOX_KERNEL(NAME, K)                                                                                   \
    extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)                                                        \
    NAME(const uint4* __restrict__ blob, int n_vec4, float cd, float half_neg, double bb_eps,                       \
         int* __restrict__ result)                                                                                  \
    {                                                                                                               \
        dcsg_bbox_body<K>(blob, n_vec4, cd, half_neg, bb_eps, result);                                              \
    }
#ifdef DCSG_PART_EXPORT
#ifndef DCSG_SET_SPEC  // the lean module of a specialised scene holds the _sp kernels only (csrc/jit.cpp)
DCSG_BBOX_KERNEL(dcsg_bbox, 0)
DCSG_BBOX_KERNEL(dcsg_bbox_s4, 4)
DCSG_BBOX_KERNEL(dcsg_bbox_s8, 8)
#endif
#ifdef DCSG_HAVE_SPEC
DCSG_BBOX_KERNEL(dcsg_bbox_sp, -1)
#endif
#endif  // DCSG_PART_EXPORT


// ---------------------------------------------------------------------------
// Mesher (Mesh::getSurface, mesh.hpp:148-311) on the dense lattice.
// Level-synchronous: one thread per node of the level.
// ---------------------------------------------------------------------------
// Box3f::getCorners(s), geometry.hpp:264-279: corner k = center + ((hd*sign_k)*s) with
// sign_k = (-,-,+),(+,-,+),(+,-,-),(-,-,-),(-,+,+),(+,+,+),(+,+,-),(-,+,-).  hd*(+-1) and
// the scaling by s in {1.0f, 0.5f} are exact, so this is center +- s*hd per axis.  Signs come
// from the bits of k (no runtime-indexed private array, which would live in scratch).
__device__ __forceinline__ void dcsg_corner(int k, float s, const float c[3], const float hd[3], float out[3])
{
    int q = k & 3;
    float sx = (q == 1 || q == 2) ? 1.0f : -1.0f;
    float sy = k >= 4 ? 1.0f : -1.0f;
    float sz = q < 2 ? 1.0f : -1.0f;
    out[0] = c[0] + s * (hd[0] * sx);
    out[1] = c[1] + s * (hd[1] * sy);
    out[2] = c[2] + s * (hd[2] * sz);
}
// 12 edges: 0-3 bottom ring (i,(i+1)%4), 4-7 top ring, 8-11 verticals (i,i+4); mesh.hpp:187-209
__device__ __forceinline__ int dcsg_edge_a(int e) { return e < 8 ? e : e - 8; }
__device__ __forceinline__ int dcsg_edge_b(int e)
{
    return e < 4 ? ((e + 1) & 3) : (e < 8 ? (((e - 3) & 3) + 4) : e - 4);
}

struct Sampler {
    const DcsgMeshParams* P;
    const float* sdf;
    const SceneLds* scene;
    // Node-wise lattice, node below the sweep level: the lattice indices of its ancestor on that level, lo[k] .. hi[k].  The
    // ancestor was subdivided, so the sweep did not cull it (same rule on the same value, dcsg_node_culled) and wrote every
    // point of that range: a sample inside it needs no look at the node mask (up to eight dependent byte loads per sample,
    // nine to ninety-three samples per node).  lo[0] > hi[0]: no such knowledge.
    int lo[3], hi[3];
};
__device__ __forceinline__ bool dcsg_sample_in_swept_node(const Sampler& S, int ix, int iy, int iz)
{
    return ix >= S.lo[0] && ix <= S.hi[0] && iy >= S.lo[1] && iy <= S.hi[1] && iz >= S.lo[2] && iz <= S.hi[2];
}
// Node-wise lattice: which points of a swept node the sweep wrote.  The mesher reads, inside a node of the sweep level,
// (A) corners and centres of nodes down to the maximum level -- with s = 2^(gridLevel - maximumOctreeLevel) lattice steps per
// such node: all three indices multiples of s, or all three congruent s/2 -- and (B) the edge-ambiguity samples of the levels
// that make the test (minimum ... maximum - 1): every lattice point along an edge of such a node, i.e. at least two indices
// multiples of 2 s.  Everything else in the node is never looked at (at 7/9/10: 297 of 729 points), and the sweep leaves it
// out.  A sample outside the pattern -- possible only if float rounding moved it -- is evaluated on demand like any other
// point the lattice does not hold.  P.pattern_step = s (1: every point), P.pattern_edges = rule (B) applies.
__device__ __forceinline__ bool dcsg_in_sweep_pattern(const DcsgMeshParams& P, int ix, int iy, int iz)
{
    const int s = P.pattern_step;
    if (s <= 1) return true;
    const int m = s - 1, h = s >> 1, rx = ix & m, ry = iy & m, rz = iz & m;
    if ((rx | ry | rz) == 0) return true;
    if (rx == h && ry == h && rz == h) return true;
    if (P.pattern_edges) {
        const int e = 2 * s - 1;
        if (((ix & e) == 0 ? 1 : 0) + ((iy & e) == 0 ? 1 : 0) + ((iz & e) == 0 ? 1 : 0) >= 2) return true;
    }
    return false;
}

// Node-wise sweep: is lattice point (ix,iy,iz) inside a swept node?  A point on a node face belongs to two nodes per
// axis, so up to eight mask bytes are looked at (one for a point strictly inside a node).
__device__ __forceinline__ bool dcsg_lattice_point_written(const DcsgMeshParams& P, int ix, int iy, int iz)
{
    if (!P.node_mask) return true;
    const int sh = P.mask_shift, N = 1 << P.mask_level, low = (1 << sh) - 1;
    int c[3] = {ix, iy, iz}, a[3], n_alt[3];
    for (int k = 0; k < 3; k++) {
        a[k] = c[k] >> sh;
        n_alt[k] = ((c[k] & low) == 0 && a[k] > 0) ? 2 : 1;  // also the node below
        if (a[k] > N - 1) {                                   // index res: only the last node holds it
            a[k] = N - 1;
            n_alt[k] = 1;
        }
    }
    for (int dz = 0; dz < n_alt[2]; dz++)
        for (int dy = 0; dy < n_alt[1]; dy++)
            for (int dx = 0; dx < n_alt[0]; dx++)
                if (P.node_mask[((size_t)(a[2] - dz) * N + (size_t)(a[1] - dy)) * N + (size_t)(a[0] - dx)]) return true;
    return false;
}

// The value AT THE SNAPPED LATTICE POINT (ISV3D64::getValue, ISV.hpp:15-60).  Planes that
// are not resident on this rank are evaluated directly: a sample is a pure function of its
// lattice index, so both paths give the same bits.
__device__ __forceinline__ float sample_sdf(const Sampler& S, const float p[3])
{
    const DcsgLattice& L = S.P->lat;
    int ix = dcsg_snap(L, 0, p[0]), iy = dcsg_snap(L, 1, p[1]), iz = dcsg_snap(L, 2, p[2]);
    if (iz >= S.P->z_lo && iz <= S.P->z_hi && dcsg_in_sweep_pattern(*S.P, ix, iy, iz) &&
        (dcsg_sample_in_swept_node(S, ix, iy, iz) || dcsg_lattice_point_written(*S.P, ix, iy, iz))) {
        size_t n1 = (size_t)L.res + 1;
        return S.sdf[((size_t)(iz - S.P->z_lo) * n1 + iy) * n1 + ix];
    }
#ifdef DCSG_HAVE_SPEC
    // the module's per-scene code (same bits as the interpreter, a fraction of its instructions); a module of the
    // specialised kernel set holds no interpreter copy here: every copy is compiled when a scene is first loaded
    float sd, u0, u1, u2;
    dcsg_stencil_spec<true>(dcsg_lattice_coord(L, 0, ix), dcsg_lattice_coord(L, 1, iy), dcsg_lattice_coord(L, 2, iz), true, false,
                            sd, u0, u1, u2);
    return sd;
#else
    return dcsg_primary_sdf(*S.scene, dcsg_lattice_coord(L, 0, ix), dcsg_lattice_coord(L, 1, iy),
                            dcsg_lattice_coord(L, 2, iz));
#endif
}

// Node codes in info[i] (low byte; lookup case in the high byte)
#define NODE_SKIP 0u       // culled (mesh.hpp:170) or outside this rank's Z-slab
#define NODE_SUBDIVIDE 1u  // 8 children go to the next level
#define NODE_LEAF 2u       // emits trsMap[lookup] triangles
#define NODE_LEAF_REMOTE 3u  // leaf whose triangles another rank emits

// primary_sdf at a point the lattice does not hold, for the lanes that `need` it; called by ALL lanes of a wave (the cell
// lists' interpreter runs the wave as long as its longest list).  Many-object scenes of built-in brushes (CL.cells != NULL:
// interpreter 4) read the list of the cell the point is in, as the projection does -- a handful of entries instead of
// every object of the scene; other scenes take their module's evaluation.
#ifdef DCSG_HAVE_CULL
template <int N>
__device__ __forceinline__ void dcsg_interp_lanes(const SceneLds& s, const unsigned int* __restrict__ entries, unsigned int first,
                                                  unsigned int count, float x, float y, float z, float (&out)[N]);
__device__ __forceinline__ uint2 dcsg_cell_list_at(const DcsgCellLists& CL, float x, float y, float z);
#endif
__device__ __forceinline__ float dcsg_point_sdf(const SceneLds& scene, const DcsgCellLists& CL, bool need, float x, float y, float z)
{
#ifdef DCSG_HAVE_CULL
    if (CL.cells) {  // kernel argument: uniform
        const uint2 cl = need ? dcsg_cell_list_at(CL, x, y, z) : make_uint2(0u, 0u);
        float o[1];
        dcsg_interp_lanes<1>(scene, CL.entries, cl.x, cl.y, x, y, z, o);
        return o[0];
    }
#endif
    float sd = 0.0f;
    if (need) {
#ifdef DCSG_HAVE_SPEC
        float u0, u1, u2;
        dcsg_stencil_spec<false>(x, y, z, true, false, sd, u0, u1, u2);
#else
        sd = dcsg_primary_sdf(scene, x, y, z);
#endif
    }
    return sd;
}

// get_normal at such a point (k2.cl:149-179), same calling rule
__device__ __forceinline__ void dcsg_point_normal(const SceneLds& scene, const DcsgCellLists& CL, bool need, float x, float y, float z,
                                                  float& nx, float& ny, float& nz)
{
#ifdef DCSG_HAVE_CULL
    if (CL.cells) {  // the cell's list serves the stencil too: cells are built over their box widened by epsilon
        const uint2 cl = need ? dcsg_cell_list_at(CL, x, y, z) : make_uint2(0u, 0u);
        float o[7];
        dcsg_interp_lanes<7>(scene, CL.entries, cl.x, cl.y, x, y, z, o);
        dcsg_finish_normal(o[1] - o[2], o[3] - o[4], o[5] - o[6], nx, ny, nz);
        return;
    }
#endif
    nx = ny = nz = 0.0f;
    if (need) {
#ifdef DCSG_HAVE_SPEC
        float sd;
        dcsg_stencil_spec<true>(x, y, z, false, true, sd, nx, ny, nz);
#else
        dcsg_get_normal(scene, x, y, z, nx, ny, nz);
#endif
    }
}

#ifdef DCSG_PART_EXPORT
// Node classification.  block_sums[2*b+0] = children of block b, [2*b+1] = triangles of
// block b.  block_stats[3*b ..] = {culled, subdivided, leaves} of block b, counting
// only nodes this rank owns so that per-rank stats add up to the single-GPU ones.
// The complex-edge test (mesh.hpp:244-258) is NOT made here: a node that reaches it is filed as the leaf it is unless that
// test says otherwise, and listed in pending[1 ..] (pending[0] = count; entry = node | owned << 31) for
// dcsg_mesh_complex_edges, which runs between this kernel and the scan.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_mesh_classify(const uint4* __restrict__ blob, int n_vec4, DcsgMeshParams P, const float* __restrict__ nodes,
                   unsigned int n_nodes, const float* __restrict__ sdf, const float* __restrict__ nrm,
                   const unsigned char* __restrict__ tri_table, unsigned short* __restrict__ info,
                   unsigned int* __restrict__ block_sums, unsigned int* __restrict__ block_stats,
                   unsigned int* __restrict__ pending, DcsgCellLists CL, const unsigned int* __restrict__ n_nodes_dev)
{
    // the first levels run without a host round trip: n_nodes is then an upper bound (eight per node of the level before)
    // and the level's real node count is what the previous level's scan left on the device
    if (n_nodes_dev) n_nodes = *n_nodes_dev < n_nodes ? *n_nodes_dev : n_nodes;
    SceneLds scene = dcsg_stage_scene(blob, n_vec4);
    Sampler S = {&P, sdf, &scene, {1, 1, 1}, {0, 0, 0}};
    unsigned int i = blockIdx.x * DCSG_BLOCK + threadIdx.x;
    unsigned int code = NODE_SKIP, lookup = 0, n_tri = 0;
    bool owned = false, culled = false, intersects = false, pend = false;
    float c[3] = {0.0f, 0.0f, 0.0f};
    if (i < n_nodes) {
        c[0] = nodes[(size_t)i * 3 + 0], c[1] = nodes[(size_t)i * 3 + 1], c[2] = nodes[(size_t)i * 3 + 2];
        // Z-slab ownership: the node's plane range [zlo, zhi]
        int zlo = dcsg_snap(P.lat, 2, c[2] + 1.0f * (P.hd[2] * -1.0f));
        int zhi = dcsg_snap(P.lat, 2, c[2] + 1.0f * (P.hd[2] * 1.0f));
        intersects = zlo < P.z_own_hi && zhi > P.z_own_lo;
        owned = zlo >= P.z_own_lo && zlo < P.z_own_hi;
    }
    // mesh.hpp:167-170: the value at the (snapped) centre.  The first levels' centres, and those of the nodes the sweep
    // level drops, are not in a node-wise lattice: those are evaluated, the whole wave together (dcsg_point_sdf)
    float d = 0.0f;
    {
        const DcsgLattice& L = P.lat;
        const int ix = dcsg_snap(L, 0, c[0]), iy = dcsg_snap(L, 1, c[1]), iz = dcsg_snap(L, 2, c[2]);
        if (P.node_mask && P.level >= P.mask_level && P.mask_check == 0) {
            // the sweep-level node this one lies in (itself at that level), as the sweep indexed it (dcsg_node_origin)
            const int idx[3] = {ix, iy, iz}, N = 1 << P.mask_level;
            for (int k = 0; k < 3; k++) {
                int a = idx[k] >> P.mask_shift;
                a = a > N - 1 ? N - 1 : a;
                S.lo[k] = a << P.mask_shift;
                S.hi[k] = S.lo[k] + (1 << P.mask_shift);
            }
        }
        // (at the sweep level itself the node may be one the sweep culled: its centre asks the mask)
        const bool in_swept = P.level > P.mask_level && dcsg_sample_in_swept_node(S, ix, iy, iz);
        const bool held = intersects && iz >= P.z_lo && iz <= P.z_hi && dcsg_in_sweep_pattern(P, ix, iy, iz) &&
                          (in_swept || dcsg_lattice_point_written(P, ix, iy, iz));
        if (held) {
            const size_t n1 = (size_t)L.res + 1;
            d = sdf[((size_t)(iz - P.z_lo) * n1 + iy) * n1 + ix];
        }
        const float e = dcsg_point_sdf(scene, CL, intersects && !held, dcsg_lattice_coord(L, 0, ix), dcsg_lattice_coord(L, 1, iy),
                                       dcsg_lattice_coord(L, 2, iz));
        if (!held) d = e;
    }
    if (intersects) {
        float mag = dcsg_sqrt(P.hd[0] * P.hd[0] + P.hd[1] * P.hd[1] + P.hd[2] * P.hd[2]);
        culled = __builtin_fabsf(d) > mag * 1.1f;
        if (!culled) {
            // mesh.hpp:176-183; the corner case is only ever used by a leaf, and a node above the minimum level
            // is always subdivided (mesh.hpp:214-217), so its corners are not sampled
            if (P.level >= P.min_level) {
                // rolled: one copy of the (per-scene, possibly large) evaluation code instead of eight -- the scene
                // module is compiled when a scene is first loaded, and this kernel was the largest part of that
#pragma unroll 1
                for (int k = 0; k < 8; k++) {
                    float p[3];
                    dcsg_corner(k, 1.0f, c, P.hd, p);
                    lookup |= (sample_sdf(S, p) < 0.0f ? 1u : 0u) << k;
                }
            }
            bool subdivide = false;
            if (P.level < P.min_level) {  // mesh.hpp:214-217
                subdivide = true;
            } else if (P.level != P.max_level) {  // at max level the tests cannot change the outcome (:265-267)
                // edge ambiguity, mesh.hpp:221-238
                int n_along = 1 << (P.grid_level - P.level);
                for (int e = 0; e < 12 && !subdivide; e++) {
                    float a[3], b[3];
                    dcsg_corner(dcsg_edge_a(e), 1.0f, c, P.hd, a);
                    dcsg_corner(dcsg_edge_b(e), 1.0f, c, P.hd, b);
                    float dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
                    for (int j = 1; j < n_along; j++) {
                        float f = (float)j / (float)n_along;
                        float t[3] = {a[0] + f * dx, a[1] + f * dy, a[2] + f * dz};
                        if (sample_sdf(S, t) < 0.0f) {
                            subdivide = true;
                            break;
                        }
                    }
                }
                pend = !subdivide;  // complex edges, mesh.hpp:244-258: dcsg_mesh_complex_edges
            }
            if (subdivide) {
                code = NODE_SUBDIVIDE;
            } else if (owned) {
                code = NODE_LEAF;
                n_tri = tri_table[lookup * 16];
            } else {
                code = NODE_LEAF_REMOTE;
            }
        }
    }
    if (i < n_nodes) info[i] = (unsigned short)(code | (lookup << 8));
    // block totals for the compaction scan + per-level statistics; the nodes the complex-edge test has to see
    const unsigned long long pm = __builtin_amdgcn_ballot_w64(pend);
    unsigned int child = code == NODE_SUBDIVIDE ? 8u : 0u;
    unsigned int wc = wave_sum(child), wt = wave_sum(n_tri);
    unsigned int ws = wave_sum((owned && code == NODE_SUBDIVIDE) ? 1u : 0u);
    unsigned int wl = wave_sum(code == NODE_LEAF ? 1u : 0u);
    unsigned int wq = wave_sum((owned && culled) ? 1u : 0u);
    __shared__ unsigned int part[6][DCSG_BLOCK / DCSG_WAVE];
    __shared__ unsigned int pend_base;
    int wave = threadIdx.x / DCSG_WAVE;
    if ((threadIdx.x & (DCSG_WAVE - 1)) == 0) {
        part[0][wave] = wc;
        part[1][wave] = wt;
        part[2][wave] = wl;
        part[3][wave] = wq;
        part[4][wave] = ws;
        part[5][wave] = (unsigned int)__popcll(pm);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int cs = 0, t = 0, l = 0, q = 0, sb = 0, np = 0;
        for (int k = 0; k < DCSG_BLOCK / DCSG_WAVE; k++) {
            cs += part[0][k];
            t += part[1][k];
            l += part[2][k];
            q += part[3][k];
            sb += part[4][k];
            np += part[5][k];
        }
        block_sums[2 * blockIdx.x + 0] = cs;
        block_sums[2 * blockIdx.x + 1] = t;
        // the level's statistics per block, summed by dcsg_scan_blocks (four atomics per block on the same four words
        // retired one at a time: 0.4 ms for the 13 000 blocks of a 1024^3 export's last level)
        block_stats[3 * blockIdx.x + 0] = q;
        block_stats[3 * blockIdx.x + 1] = sb;
        block_stats[3 * blockIdx.x + 2] = l;
        pend_base = np ? atomicAdd(pending, np) : 0u;  // one per block, for the same reason
    }
    __syncthreads();
    if (pend) {  // in any order: the test's verdict on a node does not depend on the others
        unsigned int off = pend_base + (unsigned int)__popcll(pm & ((1ull << (threadIdx.x & (DCSG_WAVE - 1u))) - 1ull));
        for (int k = 0; k < wave; k++) off += part[5][k];
        pending[1u + off] = i | (owned ? 0x80000000u : 0u);
    }
}

// Complex edges, mesh.hpp:244-258 (angleBetweenVectors, geometry.hpp:118-124): a node without an ambiguous edge is still
// subdivided when the surface normals at the two ends of one of its edges differ by more than the threshold.  The ends are
// the node's 8 corners, and a normal that the lattice does not hold is six evaluations of the scene: 48 per node, for the
// nodes that get this far -- scattered over the level, so that inside dcsg_mesh_classify (rounds 1-3) a wave ran the eight
// corners one after the other for as few as one of its 64 nodes while the rest of the launch had long finished.  Here the
// pending nodes are dense: a wave takes eight of them, a lane ONE corner of one node (lane = 8 * node + corner), and the 12
// edge tests are made by the node's eight lanes exchanging normals (lane k: edge k, and edge 8 + k for k < 4).  Same
// points, same arithmetic, same verdict as the reference's loop.  A node found complex becomes NODE_SUBDIVIDE and the sums
// dcsg_mesh_classify wrote for it as a leaf are taken back (integer atomics: any order gives the same totals).
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_mesh_complex_edges(const uint4* __restrict__ blob, int n_vec4, DcsgMeshParams P, const float* __restrict__ nodes,
                        const unsigned int* __restrict__ pending, const float* __restrict__ nrm,
                        const unsigned char* __restrict__ tri_table, unsigned short* __restrict__ info,
                        unsigned int* __restrict__ block_sums, unsigned int* __restrict__ block_stats, DcsgCellLists CL)
{
    SceneLds scene = dcsg_stage_scene(blob, n_vec4);
    const unsigned int lane = threadIdx.x & (DCSG_WAVE - 1u);
    const unsigned int waves = DCSG_BLOCK / DCSG_WAVE;
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / DCSG_WAVE));
    const unsigned int n_pending = pending[0];
    const DcsgLattice& L = P.lat;
    const unsigned int slot = lane >> 3, group0 = lane & ~7u;
    const int k = (int)(lane & 7u);
    for (unsigned int w0 = (blockIdx.x * waves + wave) * 8u; w0 < n_pending; w0 += gridDim.x * waves * 8u) {
        const bool live = w0 + slot < n_pending;
        const unsigned int ent = pending[1u + (live ? w0 + slot : w0)], i = ent & 0x7fffffffu;
        const bool owned = (ent >> 31) != 0u;
        const float c[3] = {nodes[(size_t)i * 3 + 0], nodes[(size_t)i * 3 + 1], nodes[(size_t)i * 3 + 2]};
        float p[3];
        dcsg_corner(k, 1.0f, c, P.hd, p);
        // the normal AT THE SNAPPED corner (ISV3D64::getValue semantics, as for the SDF); read where the lattice holds normals
        const int ix = dcsg_snap(L, 0, p[0]), iy = dcsg_snap(L, 1, p[1]), iz = dcsg_snap(L, 2, p[2]);
        const bool held = P.has_normals && iz >= P.z_lo && iz <= P.z_hi && dcsg_in_sweep_pattern(P, ix, iy, iz) &&
                          dcsg_lattice_point_written(P, ix, iy, iz);
        float nx, ny, nz;
        dcsg_point_normal(scene, CL, live && !held, dcsg_lattice_coord(L, 0, ix), dcsg_lattice_coord(L, 1, iy), dcsg_lattice_coord(L, 2, iz), nx,
                          ny, nz);
        if (held) {
            const size_t n1 = (size_t)L.res + 1;
            const float* q = nrm + (((size_t)(iz - P.z_lo) * n1 + iy) * n1 + ix) * 3;
            nx = q[0], ny = q[1], nz = q[2];
        }
        const float len = dcsg_sqrt(nx * nx + ny * ny + nz * nz);
        // edge e joins corners a(e) and b(e) (mesh.hpp:187-209): a(e) = e for the two rings, a(8 + j) = j for the verticals
        auto edge_test = [&](int other) {
            const int from = (int)group0 + other;
            const float bx = __shfl(nx, from, DCSG_WAVE), by = __shfl(ny, from, DCSG_WAVE), bz = __shfl(nz, from, DCSG_WAVE),
                        bl = __shfl(len, from, DCSG_WAVE);
            const float mm = len * bl;
            if (mm < 1e-6f) return false;
            const float q = (nx * bx + ny * by + nz * bz) / mm;
            // (float)acos((double)q) > threshold, folded by the host into q_star
            return q >= -1.0f && q <= P.q_star;
        };
        bool complex_edge = edge_test(dcsg_edge_b(k));
        const bool vertical = edge_test(k < 4 ? dcsg_edge_b(8 + k) : k);
        complex_edge = complex_edge || (k < 4 && vertical);
        const unsigned long long votes = __builtin_amdgcn_ballot_w64(live && complex_edge);
        if (live && k == 0 && ((votes >> group0) & 0xffull) != 0ull) {
            const unsigned int lookup = (unsigned int)info[i] >> 8;
            info[i] = (unsigned short)(NODE_SUBDIVIDE | (lookup << 8));
            const unsigned int b = i / DCSG_BLOCK;
            atomicAdd(&block_sums[2u * b], 8u);
            if (owned) {  // it was counted as a leaf of this rank
                const unsigned int n_tri = tri_table[lookup * 16u];
                atomicSub(&block_sums[2u * b + 1u], n_tri);
                atomicAdd(&block_stats[3u * b + 1u], 1u);
                atomicSub(&block_stats[3u * b + 2u], 1u);
            }
        }
    }
}

// Exclusive scan of the per-block {children, triangles} pairs, single block.
// totals[0] = children, totals[1] = triangles; level_stats[0..3] = {culled, subdivided, leaves, triangles} of the level
// (block_stats summed; NULL: not wanted).
extern "C" __global__ void __launch_bounds__(1024)
dcsg_scan_blocks(unsigned int* __restrict__ block_sums, unsigned int n_blocks, unsigned int* __restrict__ totals,
                 const unsigned int* __restrict__ block_stats, unsigned long long* __restrict__ level_stats)
{
    __shared__ unsigned int wsum[2][1024 / DCSG_WAVE];
    __shared__ unsigned int carry[2];
    __shared__ unsigned long long stat[3];
    if (threadIdx.x == 0) carry[0] = carry[1] = 0;
    if (threadIdx.x < 3) stat[threadIdx.x] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & (DCSG_WAVE - 1), wave = threadIdx.x / DCSG_WAVE;
    unsigned long long mine[3] = {0ull, 0ull, 0ull};
    for (unsigned int base = 0; base < n_blocks; base += 1024) {
        unsigned int i = base + threadIdx.x;
        unsigned int c = i < n_blocks ? block_sums[2 * i] : 0u;
        unsigned int t = i < n_blocks ? block_sums[2 * i + 1] : 0u;
        if (block_stats && i < n_blocks)
            for (int k = 0; k < 3; k++) mine[k] += block_stats[3 * i + k];
        unsigned int ic = wave_scan_incl(c), it = wave_scan_incl(t);
        if (lane == DCSG_WAVE - 1) {
            wsum[0][wave] = ic;
            wsum[1][wave] = it;
        }
        __syncthreads();
        unsigned int oc = carry[0], ot = carry[1];
        for (int k = 0; k < wave; k++) {
            oc += wsum[0][k];
            ot += wsum[1][k];
        }
        if (i < n_blocks) {
            block_sums[2 * i] = oc + ic - c;
            block_sums[2 * i + 1] = ot + it - t;
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            carry[0] = oc + ic;
            carry[1] = ot + it;
        }
        __syncthreads();
    }
    if (level_stats) {
        for (int k = 0; k < 3; k++) {
            unsigned long long v = mine[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, DCSG_WAVE);
            if (lane == 0 && v) atomicAdd(&stat[k], v);  // LDS, 16 waves
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = carry[0];
        totals[1] = carry[1];
        if (level_stats) {
            level_stats[0] = stat[0];
            level_stats[1] = stat[1];
            level_stats[2] = stat[2];
            level_stats[3] = carry[1];
        }
    }
}

// Emit: children of subdivided nodes (Node::subdivideIntoQueue, octree.hpp:24-32) into the
// next level's node list, triangles of leaves (mesh.hpp:283-307) into the triangle buffer.
// Offsets = scanned block base + in-block wave prefix sums, so the output order is a pure
// function of the node order: deterministic, unlike the reference's thread interleaving.
// The 256-case triangle table (lookupTable.txt expanded) is staged in LDS.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_mesh_emit(DcsgMeshParams P, const float* __restrict__ nodes, unsigned int n_nodes,
               const unsigned short* __restrict__ info, const unsigned int* __restrict__ block_base,
               const unsigned char* __restrict__ tri_table, float* __restrict__ next_nodes,
               float* __restrict__ tris, unsigned long long tri_base, const unsigned int* __restrict__ n_nodes_dev)
{
    if (n_nodes_dev) n_nodes = *n_nodes_dev < n_nodes ? *n_nodes_dev : n_nodes;  // as in dcsg_mesh_classify
    __shared__ unsigned char table[256 * 16];  // per case: count, then up to 5 x 3 edge ids
    for (int k = threadIdx.x; k < 256 * 16 / 4; k += DCSG_BLOCK)
        reinterpret_cast<unsigned int*>(table)[k] = reinterpret_cast<const unsigned int*>(tri_table)[k];
    __shared__ unsigned int wsum[2][DCSG_BLOCK / DCSG_WAVE];
    unsigned int i = blockIdx.x * DCSG_BLOCK + threadIdx.x;
    unsigned int code = NODE_SKIP, lookup = 0;
    if (i < n_nodes) {
        unsigned int v = info[i];
        code = v & 0xffu;
        lookup = v >> 8;
    }
    __syncthreads();
    unsigned int child = code == NODE_SUBDIVIDE ? 8u : 0u;
    unsigned int n_tri = code == NODE_LEAF ? table[lookup * 16] : 0u;
    unsigned int ic = wave_scan_incl(child), it = wave_scan_incl(n_tri);
    const int lane = threadIdx.x & (DCSG_WAVE - 1), wave = threadIdx.x / DCSG_WAVE;
    if (lane == DCSG_WAVE - 1) {
        wsum[0][wave] = ic;
        wsum[1][wave] = it;
    }
    __syncthreads();
    unsigned int oc = block_base[2 * blockIdx.x], ot = block_base[2 * blockIdx.x + 1];
    for (int k = 0; k < wave; k++) {
        oc += wsum[0][k];
        ot += wsum[1][k];
    }
    oc += ic - child;
    ot += it - n_tri;
    if (code != NODE_SUBDIVIDE && n_tri == 0) return;
    float c[3] = {nodes[(size_t)i * 3 + 0], nodes[(size_t)i * 3 + 1], nodes[(size_t)i * 3 + 2]};
    if (code == NODE_SUBDIVIDE) {
        // child centre = getCorners(0.5)[k]; child half diameter = 0.5f*hd (host side)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float p[3];
            dcsg_corner(k, 0.5f, c, P.hd, p);
            float* o = next_nodes + (size_t)(oc + k) * 3;
            o[0] = p[0];
            o[1] = p[1];
            o[2] = p[2];
        }
    } else {
        const unsigned char* row = table + lookup * 16;
        for (unsigned int t = 0; t < n_tri; t++) {
            float* o = tris + (size_t)(tri_base + ot + t) * 9;
            for (int v = 0; v < 3; v++) {
                int e = row[1 + 3 * t + v];
                float a[3], b[3];
                dcsg_corner(dcsg_edge_a(e), 1.0f, c, P.hd, a);
                dcsg_corner(dcsg_edge_b(e), 1.0f, c, P.hd, b);
                // Vector3f::midpoint = a.scaled(0.5) + b.scaled(0.5), geometry.hpp:91-93
                o[3 * v + 0] = 0.5f * a[0] + 0.5f * b[0];
                o[3 * v + 1] = 0.5f * a[1] + 0.5f * b[1];
                o[3 * v + 2] = 0.5f * a[2] + 0.5f * b[2];
            }
        }
    }
}

#endif  // DCSG_PART_EXPORT (mesher kernels)

// ---------------------------------------------------------------------------
// Kernel: vertex projection, performGradientDescent mesh.hpp:531-593.
// p <- p + n(p) * (-sdf(p)), `steps` times, at the true (unsnapped) positions.  All steps of
// a vertex run in registers: 12 B read + 12 B written per vertex for the whole loop instead of
// the reference's 100 blocking evaluator round trips.
// ---------------------------------------------------------------------------
// performGradientDescent (mesh.hpp:531-593) iterates p <- F(p) = p - sdf(p) * normal(p), a pure function of the
// position's bits, a fixed number of times.  Once F(p) == p bit for bit the remaining steps cannot change anything;
// once F(F(p)) == p the positions alternate and the last one follows from the parity of the steps that are left.
// Both are detected per vertex and the vertex is retired with exactly the position the full iteration would end
// on; a wave stops when all its vertices are retired (after a handful of steps: projection onto a distance field
// converges fast, then the update falls below half an ulp of the coordinates).
struct DcsgDescent {
    float x, y, z;        // current position p_k
    float px, py, pz;     // p_(k-1)
    bool have_prev, done;
};
__device__ __forceinline__ bool dcsg_same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
// q = F(p_k) was just computed as step number k (0-based) of `steps`
__device__ __forceinline__ void dcsg_descent_advance(DcsgDescent& d, float qx, float qy, float qz, int k, int steps)
{
    if (d.done) return;
    if (dcsg_same_bits(qx, d.x) && dcsg_same_bits(qy, d.y) && dcsg_same_bits(qz, d.z)) {
        d.done = true;  // fixed point: p stays
    } else if (d.have_prev && dcsg_same_bits(qx, d.px) && dcsg_same_bits(qy, d.py) && dcsg_same_bits(qz, d.pz)) {
        // period two: after this step the position is q = p_(k-1); each further step flips between q and p_k
        if (((steps - 1 - k) & 1) == 0) d.x = qx, d.y = qy, d.z = qz;
        d.done = true;
    } else {
        d.px = d.x, d.py = d.y, d.pz = d.z;
        d.x = qx, d.y = qy, d.z = qz;
        d.have_prev = true;
    }
}

template <int K>
__device__ __forceinline__ void dcsg_project_body(const uint4* __restrict__ blob, int n_vec4, float* __restrict__ verts,
                                                  unsigned int n_verts, int steps)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    const unsigned int tiles = (n_verts + DCSG_BLOCK - 1) / DCSG_BLOCK;
    for (unsigned int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        unsigned int i = tile * DCSG_BLOCK + threadIdx.x;
        if (i >= n_verts) continue;
        DcsgDescent d = {verts[(size_t)i * 3 + 0], verts[(size_t)i * 3 + 1], verts[(size_t)i * 3 + 2], 0.0f, 0.0f, 0.0f, false, false};
        for (int k = 0; k < steps; k++) {
            if (__builtin_amdgcn_ballot_w64(!d.done) == 0ull) break;
            const float x = d.x, y = d.y, z = d.z;
            float sd, nx, ny, nz;
            if (K == 0) {
                sd = dcsg_primary_sdf(s, x, y, z);
                dcsg_get_normal(s, x, y, z, nx, ny, nz);
#ifdef DCSG_HAVE_SPEC
            } else if (K < 0) {
                dcsg_stencil_spec(x, y, z, true, true, sd, nx, ny, nz);
#endif
            } else {
                dcsg_stencil<(K <= 0 ? 4 : K)>(s, x, y, z, true, true, sd, nx, ny, nz);
            }
            float m = -sd;  // v3f_scale(n, -s) then v3f_add, CVector.cpp:128-149
            dcsg_descent_advance(d, x + m * nx, y + m * ny, z + m * nz, k, steps);
        }
        verts[(size_t)i * 3 + 0] = d.x;
        verts[(size_t)i * 3 + 1] = d.y;
        verts[(size_t)i * 3 + 2] = d.z;
    }
}

#ifdef DCSG_HAVE_SPEC
// The projection of per-scene code with a LANE PER STENCIL POINT: a wave holds eight vertices, lane 8 v + j the j-th point of
// vertex v's stencil (0: the vertex, 1 .. 6: +x, -x, +y, -y, +z, -z as get_normal forms them, k2.cl:167-173; 7 idles on a
// copy of the vertex).  One single-point evaluation per lane and step instead of a seven-point one, the seven values
// exchanged within the group: a step costs a wave a seventh of the instructions, and fifty DEPENDENT steps are what a
// projection launch is made of -- with a vertex per lane a launch over a quarter of Design2's 512^3 vertices (0.8 waves per
// SIMD) took 0.6 ms whatever its size.  Every lane of a group computes the same normal and the same update from the same
// seven values, so the group's lanes stay identical copies of the vertex's state; a wave ends when its eight vertices have
// retired (the slowest of 8, not of 64).  Same points, same arithmetic as dcsg_stencil_spec: same bits.
__device__ __forceinline__ void dcsg_project_points_body(float* __restrict__ verts, unsigned int n_verts, int steps)
{
    const unsigned int lane = threadIdx.x & (DCSG_WAVE - 1u), j = lane & 7u, group0 = lane & ~7u;
    const unsigned int per_block = DCSG_BLOCK / 8u;
    const unsigned int tiles = (n_verts + per_block - 1u) / per_block;
    const float e = (float)K2_NORMAL_EPSILON, zero = 0.0f;
    for (unsigned int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const unsigned int i = tile * per_block + threadIdx.x / 8u;
        const bool live = i < n_verts;
        const unsigned int ci = live ? i : n_verts - 1u;
        DcsgDescent d = {verts[(size_t)ci * 3 + 0], verts[(size_t)ci * 3 + 1], verts[(size_t)ci * 3 + 2], 0.0f, 0.0f, 0.0f, false, !live};
        for (int k = 0; k < steps; k++) {
            if (__builtin_amdgcn_ballot_w64(!d.done) == 0ull) break;
            const float x = d.x, y = d.y, z = d.z;
            // the lane's point: v + dx adds +0.0 on the other axes, v - dx subtracts it
            const bool minus = j == 2u || j == 4u || j == 6u;
            const float sx = (j == 1u || j == 2u) ? e : zero, sy = (j == 3u || j == 4u) ? e : zero, sz = (j == 5u || j == 6u) ? e : zero;
            const bool centre = j == 0u || j == 7u;
            const float qx = centre ? x : (minus ? x - sx : x + sx), qy = centre ? y : (minus ? y - sy : y + sy),
                        qz = centre ? z : (minus ? z - sz : z + sz);
            float f, u0, u1, u2;
            dcsg_stencil_spec<false>(qx, qy, qz, true, false, f, u0, u1, u2);
            const float sd = __shfl(f, (int)group0, DCSG_WAVE);
            const float Dx = __shfl(f, (int)group0 + 1, DCSG_WAVE) - __shfl(f, (int)group0 + 2, DCSG_WAVE);
            const float Dy = __shfl(f, (int)group0 + 3, DCSG_WAVE) - __shfl(f, (int)group0 + 4, DCSG_WAVE);
            const float Dz = __shfl(f, (int)group0 + 5, DCSG_WAVE) - __shfl(f, (int)group0 + 6, DCSG_WAVE);
            float nx, ny, nz;
            dcsg_finish_normal(Dx, Dy, Dz, nx, ny, nz);
            const float m = -sd;  // v3f_scale(n, -s) then v3f_add, CVector.cpp:128-149
            dcsg_descent_advance(d, x + m * nx, y + m * ny, z + m * nz, k, steps);
        }
        if (live && j == 0u) {
            verts[(size_t)i * 3 + 0] = d.x;
            verts[(size_t)i * 3 + 1] = d.y;
            verts[(size_t)i * 3 + 2] = d.z;
        }
    }
}
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_project_sp_points(const uint4* __restrict__ /*blob*/, int /*n_vec4*/, float* __restrict__ verts, unsigned int n_verts, int steps)
{
    dcsg_project_points_body(verts, n_verts, steps);
}
#endif

#define DCSG_PROJECT_KERNEL(NAME, K)                                                                                \
    extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)                                                        \
    NAME(const uint4* __restrict__ blob, int n_vec4, float* __restrict__ verts, unsigned int n_verts, int steps)    \
    {                                                                                                               \
        dcsg_project_body<K>(blob, n_vec4, verts, n_verts, steps);                                                  \
    }
#ifdef DCSG_PART_EXPORT
#ifndef DCSG_SET_SPEC  // the lean module of a specialised scene holds the _sp kernels only (csrc/jit.cpp)
DCSG_PROJECT_KERNEL(dcsg_project, 0)
DCSG_PROJECT_KERNEL(dcsg_project_s4, 4)
DCSG_PROJECT_KERNEL(dcsg_project_s8, 8)
#endif
#ifdef DCSG_HAVE_SPEC
DCSG_PROJECT_KERNEL(dcsg_project_sp, -1)
#endif
#endif  // DCSG_PART_EXPORT


#ifdef DCSG_HAVE_CULL
// ---------------------------------------------------------------------------
// Culling variants of the evaluation kernels (built-in brush bank, flat programs; see dcsg_build_active_list).
// The per-wave active-step lists live in LDS behind the generic interpreter's stack.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned short* dcsg_wave_list(const SceneLds& s, int n_vec4)
{
    unsigned short* lists =
        reinterpret_cast<unsigned short*>(reinterpret_cast<float*>(dcsg_lds + n_vec4) + s.n_regs * DCSG_BLOCK);
    int cap = (s.n_stack_steps + 7) & ~7;
    return lists + (threadIdx.x / DCSG_WAVE) * cap;
}

// Block-level list for the lattice sweep: the block owns an 8 x 8 x 4 brick of lattice points (radius ~5 lattice
// steps instead of the 32 of a 64-point run); every thread bounds one step, the clamps are scanned across the
// block (wave scans + a 4-entry combine through LDS), kept steps are compacted with ballots.
// LDS behind the generic stack: list[cap] u16, then 4 wave totals (clamp + count).
__device__ __forceinline__ int dcsg_build_block_list(const SceneLds& s, int n_vec4, float cx, float cy, float cz, float R,
                                                     const unsigned short*& list_out)
{
    const int n = s.n_items;
    const int cap = (s.n_stack_steps + 7) & ~7;
    const int lane = threadIdx.x & (DCSG_WAVE - 1), wave = threadIdx.x / DCSG_WAVE;
    unsigned short* list = reinterpret_cast<unsigned short*>(reinterpret_cast<float*>(dcsg_lds + n_vec4) + s.n_regs * DCSG_BLOCK);
    float* wtot = reinterpret_cast<float*>(list + cap);  // [4 waves][4] clamp totals, then [4] counts (as int)
    int* wcnt = reinterpret_cast<int*>(wtot + 16);
    DcsgClamp2 carry = dcsg_clamp_identity();
    int n_active = 0;
    __syncthreads();  // the previous brick's readers are done with the list
    float top_upper = __builtin_inff();  // plain unions: the smallest upper bound of all items (dcsg_top_min_upper)
    // ... and the items' lower bounds, kept in the work area behind the counters for the pass that writes the list (the area
    // holds 12 bytes per step, Scene::lds_bytes: 2 for the list, 4 here)
    float* lower = reinterpret_cast<float*>(wcnt + 4);
    if (s.flags & 2) {
        for (int base = 0; base < n; base += DCSG_BLOCK) {
            int op, first, count;
            float lb, ub;
            unsigned long long imask;
            // order-free: thread i bounds item items[i].w (the items sorted by brush: one brush's code per wave)
            const int i = base + (int)threadIdx.x, j = i < n ? s.items[i].w : n;
            dcsg_item_clamp(s, j, cx, cy, cz, R, op, lb, ub, first, count, imask);
            top_upper = fminf(top_upper, dcsg_top_min_upper(op, ub));
            if (j < n) lower[j] = lb;
        }
        top_upper = wave_min_f(top_upper);
        if (lane == 0) wtot[wave] = top_upper;
        __syncthreads();
        top_upper = fminf(fminf(wtot[0], wtot[1]), fminf(wtot[2], wtot[3]));
        __syncthreads();  // wtot is free for the scan below
    }
    // the walk: every step's bounds first, one step per thread, into the work area (the plain unions' `lower`)
    DcsgStepCache cache = {lower, -1};
    if (!(s.flags & 2)) {
        for (int t = (int)threadIdx.x; t < s.n_stack_steps; t += DCSG_BLOCK) dcsg_cache_step_bounds(s, cache, t, cx, cy, cz, R);
        __syncthreads();
    }
    for (int base = 0; base < n; base += DCSG_BLOCK) {
        int op, first, count;
        float lb, ub;
        unsigned long long imask;
        const bool walk = !(s.flags & 2);  // block-uniform
        DcsgClamp2 mine = dcsg_clamp_identity();
        bool have = false;
        if (!walk) {  // plain unions: a single step's bound was computed above; groups are walked again for their step masks
            const int i = base + (int)threadIdx.x;
            op = -1, lb = ub = 0.0f, first = count = 0, imask = 1ull;
            have = i >= n;
            if (i < n) {
                const int4 it = s.items[i];
                if (it.y == 1) {
                    first = it.x, count = 1, op = it.z, lb = lower[i];
                    have = true;
                }
            }
        }
        if (!have) mine = dcsg_item_clamp(s, base + threadIdx.x, cx, cy, cz, R, op, lb, ub, first, count, imask, {walk ? cache.base : nullptr, -1});
        float lo = -__builtin_inff(), hi = top_upper;  // plain unions: the bound of all items, no interval walk
        if (walk) {
            DcsgClamp2 incl = dcsg_clamp_wave_scan(mine);
            if (lane == DCSG_WAVE - 1) {
                wtot[4 * wave + 0] = incl.plo;
                wtot[4 * wave + 1] = incl.qlo;
                wtot[4 * wave + 2] = incl.phi;
                wtot[4 * wave + 3] = incl.qhi;
            }
            __syncthreads();
            DcsgClamp2 pre = carry;  // everything before this wave: earlier chunks, then earlier waves of this chunk
            for (int w = 0; w < wave; w++)
                pre = dcsg_clamp_compose(pre, {wtot[4 * w], wtot[4 * w + 1], wtot[4 * w + 2], wtot[4 * w + 3]});
            DcsgClamp2 excl = dcsg_clamp_shfl_up(incl, 1);
            if (lane == 0) excl = dcsg_clamp_identity();
            DcsgClamp2 before = dcsg_clamp_compose(pre, excl);
            lo = fminf(fmaxf(0.0f, before.plo), before.qlo), hi = fminf(fmaxf(0.0f, before.phi), before.qhi);
        }
        bool keep = dcsg_step_keep(op, lb, ub, lo, hi);
        const bool all_steps = count > 64;
        unsigned int sz = keep ? (all_steps ? (unsigned int)count : (unsigned int)__popcll(imask)) : 0u;
        unsigned int inc = wave_scan_incl(sz);
        if (lane == DCSG_WAVE - 1) wcnt[wave] = (int)inc;
        if (walk)
            for (int w = 0; w < DCSG_BLOCK / DCSG_WAVE; w++)
                carry = dcsg_clamp_compose(carry, {wtot[4 * w], wtot[4 * w + 1], wtot[4 * w + 2], wtot[4 * w + 3]});
        __syncthreads();
        unsigned int off = (unsigned int)n_active + inc - sz;
        for (int w = 0; w < wave; w++) off += (unsigned int)wcnt[w];
        if (keep) {
            if (all_steps) {
                for (int j = 0; j < count; j++) list[off + j] = (unsigned short)(first + j);
            } else {
                unsigned int wpos = off;
                for (unsigned long long m = imask; m; m &= m - 1ull) list[wpos++] = (unsigned short)(first + __builtin_ctzll(m));
            }
        }
        for (int w = 0; w < DCSG_BLOCK / DCSG_WAVE; w++) n_active += wcnt[w];
        __syncthreads();  // list entries visible; wtot / wcnt free for the next chunk
    }
    list_out = list;
    return __builtin_amdgcn_readfirstlane(n_active);
}

// Lattice sweep with block-brick culling.  The launch covers whole planes (first = multiple of (res+1)^2).
// A block owns a brick of lattice points: ONE list for the brick (every thread bounds one item, dcsg_build_block_list), then
// passes of 256 points each.  `shape` = depth | wide << 8:
//   wide    16 x 16 x depth: a pass is a z plane of the brick (thread = (x, y); a wave writes 4 rows of 16 consecutive points:
//           64-B / 192-B segments);
//   narrow   8 x  8 x depth (a multiple of 4): a pass is 8 x 8 x 4 (a wave writes 8 rows of 8 points).
// Rounds 1-3 built a list per 8 x 8 x 4 brick, i.e. per 256 points: bounding the scene's items cost as much as evaluating
// what was left of them (flat256: ~400 of 1 100 vector instructions per point).  A brick of 2 048 points pays for its list
// an eighth as often, and the list of a brick twice as wide is hardly longer where objects are many times the size of a
// brick; where every extra entry is dear (smooth joins with normals: the blend radius widens every bound) the host keeps
// the small brick (dcsg_api.cpp, sweep_device).
// MODE as in dcsg_sweep_body: 0 = outputs read off the pointers, 1 = SDF only, 3 = SDF + normals (own kernel names, so
// that a profile never averages the 4 B/point and the 16 B/point sweep under one name)
template <int MODE>
__device__ __forceinline__ void dcsg_sweep_c4_body(const uint4* __restrict__ blob, int n_vec4, const DcsgLattice& L, int z0, unsigned int first,
                                                   unsigned int n_points, float* __restrict__ sdf_out, float* __restrict__ nrm_arg, int shape)
{
    float* __restrict__ nrm_out = MODE == 1 ? nullptr : nrm_arg;
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    const bool wide = (shape >> 8) != 0;                                          // kernel argument: uniform
    const unsigned int side = wide ? 16u : 8u, depth = (unsigned int)(shape & 0xff), pass_planes = wide ? 1u : 4u;
    const unsigned int n1 = (unsigned int)L.res + 1u;
    const unsigned int plane = n1 * n1;
    const unsigned int planes = dcsg_div_n1(L, dcsg_div_n1(L, n_points));
    const unsigned int first_plane = dcsg_div_n1(L, dcsg_div_n1(L, first));
    const unsigned int bx = (n1 + side - 1u) / side, by = bx, bz = (planes + depth - 1u) / depth;
    const unsigned int bricks = bx * by * bz;
    const float pad = nrm_out ? (float)K2_NORMAL_EPSILON * 1.01f : 0.0f;
    const unsigned int lx = threadIdx.x & (side - 1u), ly = wide ? threadIdx.x >> 4 : (threadIdx.x >> 3) & 7u, lz = wide ? 0u : threadIdx.x >> 6;
    for (unsigned int brick = blockIdx.x; brick < bricks; brick += gridDim.x) {
        unsigned int tx = brick % bx, t2 = brick / bx;
        unsigned int ty = t2 % by, tz = t2 / by;
        // brick bounds in lattice indices (clipped) -> centre and radius in world units
        unsigned int ix0 = tx * side, iy0 = ty * side, izl0 = tz * depth;
        unsigned int ix1 = ix0 + side - 1u < n1 - 1u ? ix0 + side - 1u : n1 - 1u;
        unsigned int iy1 = iy0 + side - 1u < n1 - 1u ? iy0 + side - 1u : n1 - 1u;
        unsigned int izl1 = izl0 + depth - 1u < planes - 1u ? izl0 + depth - 1u : planes - 1u;
        float xa = dcsg_lattice_coord(L, 0, (int)ix0), xb = dcsg_lattice_coord(L, 0, (int)ix1);
        float ya = dcsg_lattice_coord(L, 1, (int)iy0), yb = dcsg_lattice_coord(L, 1, (int)iy1);
        float za = dcsg_lattice_coord(L, 2, (int)(first_plane + izl0) + z0);
        float zb = dcsg_lattice_coord(L, 2, (int)(first_plane + izl1) + z0);
        float hx = 0.5f * __builtin_fabsf(xb - xa) + pad, hy = 0.5f * __builtin_fabsf(yb - ya) + pad,
              hz = 0.5f * __builtin_fabsf(zb - za) + pad;
        float R = __builtin_sqrtf(hx * hx + hy * hy + hz * hz) * 1.0001f + 1e-6f;
        const unsigned short* list;
        int n_list = dcsg_build_block_list(s, n_vec4, 0.5f * (xa + xb), 0.5f * (ya + yb), 0.5f * (za + zb), R, list);

        const unsigned int ix = ix0 + lx, iy = iy0 + ly;
        const bool valid_xy = ix < n1 && iy < n1;
        // invalid lanes evaluate a clamped point (keeps the wave's control flow uniform) and store nothing
        const unsigned int cix = ix < n1 ? ix : n1 - 1u, ciy = iy < n1 ? iy : n1 - 1u;
        const float x = dcsg_lattice_coord(L, 0, (int)cix);
        const float y = dcsg_lattice_coord(L, 1, (int)ciy);
#pragma unroll 1
        for (unsigned int pz = izl0; pz <= izl1; pz += pass_planes) {  // block-uniform bounds
            const unsigned int izl = pz + lz, cizl = izl < planes ? izl : planes - 1u;
            const float z = dcsg_lattice_coord(L, 2, (int)(first_plane + cizl) + z0);
            float sd = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
            dcsg_stencil<4>(s, x, y, z, sdf_out != nullptr, nrm_out != nullptr, sd, nx, ny, nz, list, n_list);
            if (valid_xy && izl <= izl1) {
                unsigned int idx = first + izl * plane + iy * n1 + ix;
                if (sdf_out) sdf_out[idx] = sd;
                if (nrm_out) {
                    float* o = nrm_out + (size_t)idx * 3;
                    o[0] = nx;
                    o[1] = ny;
                    o[2] = nz;
                }
            }
        }
    }
}

#ifndef DCSG_C4_WAVES
#define DCSG_C4_WAVES 1  // A/B builds: minimum waves per SIMD asked of the culling sweeps
#endif
#ifndef DCSG_C4D_WAVES
#define DCSG_C4D_WAVES 1  // A/B builds: the SDF-only culling sweep takes 77 VGPRs (6 waves per SIMD); asking for 7 or 8 makes it spill
#endif
#define DCSG_SWEEP_C4_KERNEL(NAME, MODE)                                                                               \
    extern "C" __global__ void __launch_bounds__(DCSG_BLOCK, ((MODE) == 1 ? DCSG_C4D_WAVES : DCSG_C4_WAVES))         \
    NAME(const uint4* __restrict__ blob, int n_vec4, DcsgLattice L, int z0, unsigned int first, unsigned int n_points, \
         float* __restrict__ sdf_out, float* __restrict__ nrm_out, int shape /*axes_off of the other sweeps*/,          \
         unsigned int /*flags*/)                                                                                        \
    {                                                                                                                  \
        dcsg_sweep_c4_body<MODE>(blob, n_vec4, L, z0, first, n_points, sdf_out, nrm_out, shape);                       \
    }
DCSG_SWEEP_C4_KERNEL(dcsg_sweep_c4, 0)
DCSG_SWEEP_C4_KERNEL(dcsg_sweep_c4_d, 1)
DCSG_SWEEP_C4_KERNEL(dcsg_sweep_c4_f, 3)

// Stage A of a culling scene with per-wave lists: a wave takes bricks of 8 x 8 x 8 probes -- ONE list for the brick's box, then
// eight passes of 4 x 4 x 4 probes (rounds 1-3: a list per 64 probes; bounding the scene's items was most of the kernel).
// The result does not depend on the visiting order.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_bbox_c4(const uint4* __restrict__ blob, int n_vec4, float cd, float half_neg, double bb_eps, int* __restrict__ result)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    DCSG_BBOX_ZHIST_DECL
    unsigned short* list = dcsg_wave_list(s, n_vec4);
    const unsigned int lane = threadIdx.x & (DCSG_WAVE - 1u), waves = DCSG_BLOCK / DCSG_WAVE;
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / DCSG_WAVE));
    int mn[3] = {1 << 20, 1 << 20, 1 << 20};
    int mx[3] = {-(1 << 20), -(1 << 20), -(1 << 20)};
    unsigned int cnt = 0;
    for (unsigned int brick = blockIdx.x * waves + wave; brick < 32u * 32u * 32u; brick += gridDim.x * waves) {
        const int bx = (int)(brick & 31u) * 8 - 128, by = (int)((brick >> 5) & 31u) * 8 - 128, bz = (int)(brick >> 10) * 8 - 128;
        // the box of the brick's probes, from the probes' own coordinates (the same expressions as below: monotone in the index)
        const float x0 = half_neg + cd * (float)bx, x1 = half_neg + cd * (float)(bx + 7);
        const float y0 = half_neg + cd * (float)by, y1 = half_neg + cd * (float)(by + 7);
        const float z0 = half_neg + cd * (float)bz, z1 = half_neg + cd * (float)(bz + 7);
        const int n_list = dcsg_build_active_list_box(s, 0.5f * (x0 + x1), 0.5f * (y0 + y1), 0.5f * (z0 + z1), 0.5f * __builtin_fabsf(x1 - x0),
                                                      0.5f * __builtin_fabsf(y1 - y0), 0.5f * __builtin_fabsf(z1 - z0), list);
#pragma unroll 1
        for (unsigned int pass = 0u; pass < 8u; pass++) {
            const int ix = bx + (int)((pass & 1u) * 4u + (lane & 3u));
            const int iy = by + (int)(((pass >> 1) & 1u) * 4u + ((lane >> 2) & 3u));
            const int iz = bz + (int)((pass >> 2) * 4u + (lane >> 4));
            float x = half_neg + cd * (float)ix;
            float y = half_neg + cd * (float)iy;
            float z = half_neg + cd * (float)iz;
            float v, u0, u1, u2;
            dcsg_stencil<4>(s, x, y, z, true, false, v, u0, u1, u2, list, n_list);
            if ((double)__builtin_fabsf(v) < bb_eps) atomicAdd(&zh[iz + 128], 1u);
            if ((double)v < bb_eps) {
                cnt++;
                mn[0] = ix < mn[0] ? ix : mn[0];
                mx[0] = ix > mx[0] ? ix : mx[0];
                mn[1] = iy < mn[1] ? iy : mn[1];
                mx[1] = iy > mx[1] ? iy : mx[1];
                mn[2] = iz < mn[2] ? iz : mn[2];
                mx[2] = iz > mx[2] ? iz : mx[2];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the list area is rewritten by the next brick
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    dcsg_bbox_commit(mn, mx, cnt, result, zh);
}

// Node-wise sweep with tile culling.  The tile is the box of `tile` consecutive nodes of the list: 8 = the children of one
// parent (the mesher emits them together, and the 64 roots come in eights too), so that the scene's items are bounded
// once per eight nodes -- the list of a box twice as wide is hardly longer where objects are many times the size of a node,
// and bounding 256 items cost more than evaluating the three that were left (flat256 1024^3: 191 000 nodes, 1.85 ms); 1 =
// a list per node (scenes with smooth joins, where every extra entry is dear).  The nodes' points are walked as in
// dcsg_sweep_nodes_body (dcsg_node_group).
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK, DCSG_C4_WAVES)
dcsg_sweep_nodes_c4(const uint4* __restrict__ blob, int n_vec4, DcsgLattice L, const float* __restrict__ nodes,
                    unsigned int n_nodes, int shift, int level, float* __restrict__ sdf_out, float* __restrict__ nrm_out,
                    unsigned char* __restrict__ node_mask, float hdx, float hdy, float hdz, int z_own_lo, int z_own_hi,
                    unsigned int* __restrict__ n_swept, int group, const unsigned short* __restrict__ pattern,
                    unsigned int n_pattern)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    const DcsgNodeGrid g = dcsg_node_grid(shift, pattern, n_pattern);
    const unsigned int side = g.side;
    const unsigned int m_pts = 0xffffffffu / g.pts + 1u;
    const float pad = nrm_out ? (float)K2_NORMAL_EPSILON * 1.01f : 0.0f;
    const unsigned int per = (unsigned int)group, n_groups = (n_nodes + per - 1u) / per;
    unsigned int swept_here = 0u;  // block-uniform; one atomic per block at the end
    for (unsigned int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const unsigned int n_first = grp * per, n_end = n_first + per < n_nodes ? n_first + per : n_nodes;
        // the group's box in lattice indices (block-uniform: every thread reads the same nodes)
        int b0[3] = {1 << 30, 1 << 30, 1 << 30}, b1[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
        bool any_inside = false;
        for (unsigned int n = n_first; n < n_end; n++) {
            int i0[3], ic[3];
            size_t slot;
            dcsg_node_origin(L, nodes, n, shift, level, i0, ic, slot);
            if (dcsg_node_outside_slab(i0[2], shift, z_own_lo, z_own_hi)) continue;
            any_inside = true;
            for (int k = 0; k < 3; k++) {
                b0[k] = i0[k] < b0[k] ? i0[k] : b0[k];
                b1[k] = i0[k] + (int)side - 1 > b1[k] ? i0[k] + (int)side - 1 : b1[k];
            }
        }
        if (!any_inside) continue;
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            lo[k] = dcsg_lattice_coord(L, k, b0[k]);
            hi[k] = dcsg_lattice_coord(L, k, b1[k]);
        }
        float hx = 0.5f * __builtin_fabsf(hi[0] - lo[0]) + pad, hy = 0.5f * __builtin_fabsf(hi[1] - lo[1]) + pad,
              hz = 0.5f * __builtin_fabsf(hi[2] - lo[2]) + pad;
        float R = __builtin_sqrtf(hx * hx + hy * hy + hz * hz) * 1.0001f + 1e-6f;
        const unsigned short* list;
        int n_list = dcsg_build_block_list(s, n_vec4, 0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2]), R, list);
        // cull rule on the snapped centres (inside the nodes' boxes, so the group's list serves them)
        const DcsgNodeGroup G = dcsg_node_group(L, nodes, n_first, n_end, shift, level, hdx, hdy, hdz, z_own_lo, z_own_hi, node_mask,
                                                [&](float cx, float cy, float cz) {
                                                    float d, u0, u1, u2;
                                                    dcsg_stencil<4>(s, cx, cy, cz, true, false, d, u0, u1, u2, list, n_list);
                                                    return d;
                                                });
        if (G.mask == 0u) continue;
        swept_here += (unsigned int)__popc(G.mask);
        const unsigned int total = (unsigned int)__popc(G.mask) * g.pts;
        unsigned int t0 = 0u;
        if (!nrm_out && sdf_out) {
            // SDF only (what an export sweeps): a lane carries FOUR points through one interpretation of the list -- the entry
            // decode (LDS record, readfirstlane, brush branch: a chain of dependent LDS reads and scalar moves per entry) is
            // paid per 1 024 points instead of per 256, and four independent evaluations overlap -- while at least three of a
            // pass's four quarters have work.  Same arithmetic per point: same bits.
            for (; t0 + 3u * DCSG_BLOCK <= total; t0 += 4u * DCSG_BLOCK) {
                float px[4], py[4], pz[4], o[4];
                size_t at[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned int t = t0 + (unsigned int)j * DCSG_BLOCK + threadIdx.x;
                    dcsg_node_group_point(L, g, G, m_pts, t < total ? t : total - 1u, px[j], py[j], pz[j], at[j]);  // idle lanes: the last point
                }
                dcsg_interp<4, 4>(s, px, py, pz, o, list, n_list);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (t0 + (unsigned int)j * DCSG_BLOCK + threadIdx.x < total) sdf_out[at[j]] = o[j];
            }
        }
        for (; t0 < total; t0 += DCSG_BLOCK) {
            const unsigned int t = t0 + threadIdx.x;
            float x, y, z;
            size_t idx;
            dcsg_node_group_point(L, g, G, m_pts, t < total ? t : total - 1u, x, y, z, idx);
            float sd = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
            dcsg_stencil<4>(s, x, y, z, sdf_out != nullptr, nrm_out != nullptr, sd, nx, ny, nz, list, n_list);
            if (t < total) {
                if (sdf_out) sdf_out[idx] = sd;
                if (nrm_out) {
                    float* o = nrm_out + idx * 3;
                    o[0] = nx;
                    o[1] = ny;
                    o[2] = nz;
                }
            }
        }
    }
    if (threadIdx.x == 0 && swept_here) atomicAdd(n_swept, swept_here);
}

#ifdef DCSG_PART_EXPORT
// ---------------------------------------------------------------------------
// Cell lists: the projection of many-object scenes (and whatever else evaluates them at scattered points).
//
// Rounds 1-3 built the projection's list per WAVE, over the box of the wave's 64 vertices: a wave whose vertices lie far apart
// (the order of the distinct vertices follows the octree, which jumps; vertices hop to other objects while they descend)
// gets the whole scene for all its steps and sets the kernel's duration on its own (round 3: 24 576 waves of 0.17 M
// cycles each in a kernel of 25 M cycles, VALU 5 % busy).  Here the tile belongs to SPACE, not to a wave: a cube of g^3
// cells over the scene, every cell with the list of the program's steps that can matter anywhere inside it (the same
// bounds and the same keep rules as the wave lists: dcsg_build_active_list_box over the cell widened by the stencil),
// built once per scene by dcsg_cells_build.  A vertex reads the list of the cell it is in -- per LANE: the interpreter
// below decodes a lane's own entry, so what a lane costs depends on where IT is and on nothing else, and lanes that
// retire are refilled from a queue of the vertices still waiting (a wave is busy until the queue is empty, not until its
// slowest vertex has done its fifty steps).  Same steps in the same order with the same arithmetic: same bits.
// Entry: object | brush << 16 | (opcode - 32) << 24 | (stack depth before the step) << 30; SET_K carries its step index
// in place of the object (its operands are the two floats of the step record).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned int dcsg_cell_entry(const int4& st, unsigned int step_index)
{
    const unsigned int op = (unsigned int)(st.x - 32) & 0x3fu;
    const unsigned int payload = st.x == SOP_SET_K ? step_index : (unsigned int)st.z & 0xffffu;
    return payload | (((unsigned int)st.y & 0xffu) << 16) | (op << 24) | (((unsigned int)st.w & 1u) << 30);
}

#define DCSG_CELL_NOT_BUILT 0xffffffffu
#define DCSG_CELL_CHUNK 512u
#define DCSG_CELL_COARSE 4u  // a coarse cell = 4 x 4 x 4 fine ones (the grid size is a multiple of 4)
__device__ __forceinline__ unsigned int dcsg_cell_of(const DcsgCellGrid& G, float x, float y, float z, unsigned int& ix, unsigned int& iy,
                                                     unsigned int& iz)
{
    // points outside the cube, and NaNs: the pseudo-cell g^3 (the whole program)
    const unsigned int g = (unsigned int)G.g;
    const float fx = (x - G.x0) * G.inv_cell, fy = (y - G.y0) * G.inv_cell, fz = (z - G.z0) * G.inv_cell, gf = (float)g;
    const bool inside = fx >= 0.0f && fx < gf && fy >= 0.0f && fy < gf && fz >= 0.0f && fz < gf;
    if (!inside) return g * g * g;
    ix = (unsigned int)fx, iy = (unsigned int)fy, iz = (unsigned int)fz;
    return (iz * g + iy) * g + ix;
}

// The list that serves a point: its fine cell where that is built, else the coarse cell around it (all of those exist), the
// whole program outside the cube.
__device__ __forceinline__ uint2 dcsg_cell_list_at(const DcsgCellLists& CL, float x, float y, float z)
{
    const unsigned int g = (unsigned int)CL.G.g, n_cells = g * g * g, gc = g / DCSG_CELL_COARSE;
    unsigned int cx = 0u, cy = 0u, cz = 0u;
    const unsigned int cell = dcsg_cell_of(CL.G, x, y, z, cx, cy, cz);
    const uint2* cells = reinterpret_cast<const uint2*>(CL.cells);
    uint2 cl = cells[cell];
    if (cl.y == DCSG_CELL_NOT_BUILT) cl = cells[n_cells + 1u + ((cz / DCSG_CELL_COARSE) * gc + cy / DCSG_CELL_COARSE) * gc + cx / DCSG_CELL_COARSE];
    return cl;
}

// Cells are built where vertices are: want[cell] = 1 asks for a cell, 2 = built.  The cells of the vertices and their 26
// neighbours (most steps of a descent stay that close).
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_cells_mark(const float* __restrict__ verts, unsigned int n_verts, DcsgCellGrid G, unsigned char* __restrict__ want)
{
    const int g = G.g;
    for (unsigned int v = blockIdx.x * DCSG_BLOCK + threadIdx.x; v < n_verts; v += gridDim.x * DCSG_BLOCK) {
        unsigned int ix = 0, iy = 0, iz = 0;
        if (dcsg_cell_of(G, verts[(size_t)v * 3], verts[(size_t)v * 3 + 1], verts[(size_t)v * 3 + 2], ix, iy, iz) == (unsigned int)(g * g * g)) continue;
        for (int dz = -1; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int cx = (int)ix + dx, cy = (int)iy + dy, cz = (int)iz + dz;
                    if (cx < 0 || cy < 0 || cz < 0 || cx >= g || cy >= g || cz >= g) continue;
                    unsigned char* w = want + ((size_t)cz * g + cy) * g + cx;
                    if (*w == 0) *w = 1;  // racing writers all write 1; a built cell (2) is left alone
                }
    }
}

// The wanted cells (want == 1) as a list, so that the build below can deal them out evenly: they are clustered (a surface),
// and a wave that walked its own stretch of the cube would find all of them or none.  work[0] = count, then the cells.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_cells_collect(const unsigned char* __restrict__ want, unsigned int n_cells_plus_one, unsigned int* __restrict__ work)
{
    const unsigned int lane = threadIdx.x & (DCSG_WAVE - 1u);
    for (unsigned int c0 = blockIdx.x * DCSG_BLOCK + (threadIdx.x & ~(DCSG_WAVE - 1u)); c0 < n_cells_plus_one; c0 += gridDim.x * DCSG_BLOCK) {
        const unsigned int cell = c0 + lane;
        const bool wanted = cell < n_cells_plus_one && want[cell] == 1;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(wanted);
        if (m == 0ull) continue;
        unsigned int base = 0u;
        if (lane == 0u) base = atomicAdd(work, (unsigned int)__popcll(m));
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (wanted) work[1u + base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = cell;
    }
}

// One wave per wanted cell of dcsg_cells_collect's list.  Behind the g^3 fine cells: cells[g^3] = the whole program, for
// points outside the cube, then the (g / 4)^3 coarse cells that serve a vertex whose fine cell has not been built (the host
// asks for all of these once per scene).  counter[0] = entries used, counter[3] = cells built.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_cells_build(const uint4* __restrict__ blob, int n_vec4, DcsgCellGrid G, uint2* __restrict__ cells,
                 unsigned int* __restrict__ entries, unsigned int* __restrict__ counter, unsigned int capacity,
                 unsigned char* __restrict__ want, const unsigned int* __restrict__ work)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    unsigned short* list = dcsg_wave_list(s, n_vec4);
    const unsigned int lane = threadIdx.x & (DCSG_WAVE - 1u);
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / DCSG_WAVE));
    const unsigned int g = (unsigned int)G.g, n_cells = g * g * g, waves = DCSG_BLOCK / DCSG_WAVE;
    // a cell's tile: the cell, a hair more than the cell (a point on a face may be filed under either neighbour), the stencil
    const float h = 0.5f * G.cell * 1.001f + (float)K2_NORMAL_EPSILON * 1.01f;
    const unsigned int n_work = work[0];
    // Entry space in chunks: one atomic per DCSG_CELL_CHUNK entries instead of one per cell (every wave of the launch adds
    // to the same word, and same-address atomics retire one at a time: 75 000 cells took 1.7 ms of which 1.5 were this).
    unsigned int chunk_at = 0u, chunk_left = 0u, built = 0u;  // wave-uniform
    for (unsigned int w = blockIdx.x * waves + wave; w < n_work; w += gridDim.x * waves) {
        const unsigned int cell = work[1u + w];
        int n;
        if (cell == n_cells) {
            n = s.n_stack_steps;
        } else if (cell > n_cells) {  // a coarse cell: DCSG_CELL_COARSE^3 fine ones
            const unsigned int gc = g / DCSG_CELL_COARSE, cc = cell - n_cells - 1u;
            const unsigned int ix = cc % gc, iy = (cc / gc) % gc, iz = cc / (gc * gc);
            const float side = G.cell * (float)DCSG_CELL_COARSE, hc = 0.5f * side * 1.001f + (float)K2_NORMAL_EPSILON * 1.01f;
            n = dcsg_build_active_list_box(s, G.x0 + ((float)ix + 0.5f) * side, G.y0 + ((float)iy + 0.5f) * side,
                                           G.z0 + ((float)iz + 0.5f) * side, hc, hc, hc, list);
        } else {
            const unsigned int ix = cell % g, iy = (cell / g) % g, iz = cell / (g * g);
            n = dcsg_build_active_list_box(s, G.x0 + ((float)ix + 0.5f) * G.cell, G.y0 + ((float)iy + 0.5f) * G.cell,
                                           G.z0 + ((float)iz + 0.5f) * G.cell, h, h, h, list);
        }
        if ((unsigned int)n > chunk_left) {  // what is left of the old chunk stays unused
            const unsigned int take = (unsigned int)n > DCSG_CELL_CHUNK ? (unsigned int)n : DCSG_CELL_CHUNK;
            unsigned int base = 0u;
            if (lane == 0u) base = atomicAdd(counter, take);
            chunk_at = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            chunk_left = take;
        }
        if (chunk_at + (unsigned int)n <= capacity && chunk_at + (unsigned int)n >= chunk_at) {
            // (otherwise the cell stays wanted: the host sees the counter and makes room)
            for (unsigned int k = lane; k < (unsigned int)n; k += DCSG_WAVE) {
                const unsigned int c = cell == n_cells ? k : (unsigned int)list[k];
                entries[chunk_at + k] = dcsg_cell_entry(s.stack_steps[c], c);
            }
            if (lane == 0u) {
                cells[cell] = make_uint2(chunk_at, (unsigned int)n);
                want[cell] = 2;
            }
            built++;
        }
        chunk_at += (unsigned int)n;
        chunk_left -= (unsigned int)n;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the list area is rewritten by the next cell
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0u && built) atomicAdd(counter + 3, built);  // cells built (diagnostic)
}

// primary_sdf at the seven stencil points of get_normal (k2.cl:149-179), each lane running ITS OWN list of steps
// (entries[first .. first + count), dcsg_cell_entry).  The stack of a culling scene is at most two deep (root accumulator,
// one group: csrc/scene.cpp builds items only then).  The stencil as the reference forms it, v +- (eps, 0, 0): the
// other coordinates of a "+" point are c + 0.0f, of a "-" point c - 0.0f = c -- kept apart here, so a -0.0 coordinate
// needs no special path.
template <int N>  // N = 7: the stencil of get_normal around (x, y, z); N = 1: the point itself
__device__ __forceinline__ void dcsg_interp_lanes(const SceneLds& s, const unsigned int* __restrict__ entries, unsigned int first,
                                                  unsigned int count, float x, float y, float z, float (&out)[N])
{
    const float e = (float)K2_NORMAL_EPSILON, zero = 0.0f;
    const float xz = x + zero, yz = y + zero, zz = z + zero;
    const float px7[7] = {x, x + e, x - e, xz, x, xz, x};
    const float py7[7] = {y, yz, y, y + e, y - e, yz, y};
    const float pz7[7] = {z, zz, z, zz, z, z + e, z - e};
    float px[N], py[N], pz[N];
#pragma unroll
    for (int k = 0; k < N; k++) px[k] = px7[k], py[k] = py7[k], pz[k] = pz7[k];
    float top[N], below[N];
#pragma unroll
    for (int k = 0; k < N; k++) top[k] = below[k] = 0.0f, out[k] = (float)K2_MAX_DISTANCE;
    float bk = 0.0f, inv_bk = 0.0f;
    const unsigned int longest = (unsigned int)wave_max((int)count);
    unsigned int next = count ? entries[first] : 0u;
    for (unsigned int it = 0u; it < longest; it++) {
        const unsigned int ent = next;
        if (it + 1u < count) next = entries[first + it + 1u];  // the next entry is on its way while this one runs
        if (it >= count) continue;
        const int op = (int)((ent >> 24) & 0x3fu) + 32;
        const bool leaf = op <= SOP_ACC_NEGMAX || (op >= SOP_ACC_SMIN && op <= SOP_ACC_NEGSMAX);
        if (leaf) {
            float r[N];
            const int shape = (int)((ent >> 16) & 0xffu);
            if (shape == 0 && (s.flags & 1)) {
#pragma unroll
                for (int k = 0; k < N; k++) r[k] = (float)K2_MAX_DISTANCE;
            } else {
                const unsigned int obj = ent & 0xffffu;
                const float4 q0 = s.obj[4u * obj + 0u], q1 = s.obj[4u * obj + 1u], q2 = s.obj[4u * obj + 2u], q3 = s.obj[4u * obj + 3u];
                float A[N], B[N], C[N];
#pragma unroll
                for (int k = 0; k < N; k++) {
                    float dx = px[k] - q0.x, dy = py[k] - q0.y, dz = pz[k] - q0.z;  // k2.cl:105-111
                    A[k] = dx * q1.x + dy * q1.y + dz * q1.z;
                    B[k] = dx * q2.x + dy * q2.y + dz * q2.z;
                    C[k] = dx * q3.x + dy * q3.y + dz * q3.z;
                }
                dcsg_bank_n<N>(A, B, C, shape, r);
            }
            if (op == SOP_PUSH) {
                if (ent >> 30 & 1u) {
#pragma unroll
                    for (int k = 0; k < N; k++) below[k] = top[k];
                }
#pragma unroll
                for (int k = 0; k < N; k++) top[k] = r[k];
            } else if (op == SOP_ACC_MIN) {
#pragma unroll
                for (int k = 0; k < N; k++) top[k] = top[k] < r[k] ? top[k] : r[k];
            } else if (op == SOP_ACC_MAX) {
#pragma unroll
                for (int k = 0; k < N; k++) top[k] = top[k] > r[k] ? top[k] : r[k];
            } else if (op == SOP_ACC_NEGMAX) {
#pragma unroll
                for (int k = 0; k < N; k++) {
                    float nb = -r[k];
                    top[k] = top[k] > nb ? top[k] : nb;
                }
            } else if (op == SOP_ACC_SMIN) {
#pragma unroll
                for (int k = 0; k < N; k++) top[k] = dcsg_smin(top[k], r[k], bk, inv_bk);
            } else if (op == SOP_ACC_SMAX) {
#pragma unroll
                for (int k = 0; k < N; k++) top[k] = dcsg_smax(top[k], r[k], bk, inv_bk);
            } else {
#pragma unroll
                for (int k = 0; k < N; k++) top[k] = dcsg_smax(top[k], -r[k], bk, inv_bk);
            }
        } else if (op <= SOP_POP_NEGMAX || (op >= SOP_POP_SMIN && op <= SOP_POP_NEGSMAX)) {
#pragma unroll
            for (int k = 0; k < N; k++) {
                float a = below[k], b = top[k];
                if (op == SOP_POP_MIN) top[k] = a < b ? a : b;
                else if (op == SOP_POP_MAX) top[k] = a > b ? a : b;
                else if (op == SOP_POP_NEGMAX) {
                    float nb = -b;
                    top[k] = a > nb ? a : nb;
                } else if (op == SOP_POP_SMIN) top[k] = dcsg_smin(a, b, bk, inv_bk);
                else if (op == SOP_POP_SMAX) top[k] = dcsg_smax(a, b, bk, inv_bk);
                else top[k] = dcsg_smax(a, -b, bk, inv_bk);
            }
        } else if (op == SOP_EXPORT_TOP) {
#pragma unroll
            for (int k = 0; k < N; k++) out[k] = top[k];
        } else if (op == SOP_SET_K) {
            const int4 st = s.stack_steps[ent & 0xffffu];
            bk = __int_as_float(st.y);
            inv_bk = __int_as_float(st.z);
        }
    }
}

// performGradientDescent over the cell lists.  A lane owns one vertex at a time; idle lanes are refilled from `queue`
// (a counter into the vertex array) whenever `refill_at` lanes of the wave are idle, and the wave ends when the queue is
// empty and its last vertex has retired.  A vertex in a fine cell that has not been built (it has left the neighbourhood
// it started in) reads the list of the COARSE cell around it -- all of those exist -- and asks for the fine cell
// (want = 1: the next projection of this scene builds it before it starts).  Every vertex runs exactly the steps
// dcsg_project_body would run for it.
#ifndef DCSG_PROJECT_CELLS_WAVES
#define DCSG_PROJECT_CELLS_WAVES 4
#endif
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK, DCSG_PROJECT_CELLS_WAVES)
dcsg_project_cells(const uint4* __restrict__ blob, int n_vec4, float* __restrict__ verts, unsigned int n, int steps,
                   DcsgCellGrid G, const uint2* __restrict__ cells, const unsigned int* __restrict__ entries,
                   unsigned int* __restrict__ queue, int refill_at, unsigned char* __restrict__ want)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    const unsigned int lane = threadIdx.x & (DCSG_WAVE - 1u);
    const unsigned int g = (unsigned int)G.g, n_cells = g * g * g, gc = g / DCSG_CELL_COARSE;
    const unsigned int total_waves = gridDim.x * (DCSG_BLOCK / DCSG_WAVE);
    DcsgDescent d = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, false, true};
    unsigned int idx = 0u;
    int k = 0;
    bool queue_empty = false;  // wave-uniform
    // The wave draws vertices from the queue a batch at a time (every wave of the launch adds to the same word, and
    // same-address atomics retire one at a time) and hands them to its idle lanes as they come free.  Batches shrink with
    // what is left -- a quarter of an even share, between 16 and 256 -- so that the last ones are small and the waves end
    // together.
    unsigned int batch_at = 0u, batch_left = 0u, seen = 0u;  // wave-uniform
    for (;;) {
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(d.done);
        const int n_idle = __popcll(idle);
        if (!queue_empty && (n_idle >= refill_at || idle == ~0ull)) {
            if (batch_left == 0u) {
                unsigned int take = (seen < n ? n - seen : 0u) / (4u * total_waves);
                take = take < 16u ? 16u : (take > 256u ? 256u : take);
                unsigned int base = 0u;
                if (lane == 0u) base = atomicAdd(queue, take);
                batch_at = seen = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                batch_left = batch_at < n ? (n - batch_at < take ? n - batch_at : take) : 0u;
                if (batch_left == 0u) queue_empty = true;
            }
            const unsigned int rank = (unsigned int)__popcll(idle & ((1ull << lane) - 1ull));
            if (d.done && rank < batch_left) {
                idx = batch_at + rank;
                d = {verts[(size_t)idx * 3 + 0], verts[(size_t)idx * 3 + 1], verts[(size_t)idx * 3 + 2], 0.0f, 0.0f, 0.0f, false, false};
                k = 0;
            }
            const unsigned int took = (unsigned int)n_idle < batch_left ? (unsigned int)n_idle : batch_left;
            batch_at += took;
            batch_left -= took;
        }
        if (__builtin_amdgcn_ballot_w64(!d.done) == 0ull) {
            if (queue_empty) break;
            continue;
        }
        const float x = d.x, y = d.y, z = d.z;
        unsigned int ix = 0u, iy = 0u, iz = 0u;
        const unsigned int cell = dcsg_cell_of(G, x, y, z, ix, iy, iz);
        uint2 cl = d.done ? make_uint2(0u, 0u) : cells[cell];
        if (!d.done && cl.y == DCSG_CELL_NOT_BUILT) {
            want[cell] = 1;
            cl = cells[n_cells + 1u + ((iz / DCSG_CELL_COARSE) * gc + iy / DCSG_CELL_COARSE) * gc + ix / DCSG_CELL_COARSE];
        }
        float o[7];
        dcsg_interp_lanes<7>(s, entries, cl.x, cl.y, x, y, z, o);
        if (!d.done) {
            float nx, ny, nz;
            dcsg_finish_normal(o[1] - o[2], o[3] - o[4], o[5] - o[6], nx, ny, nz);
            const float m = -o[0];  // v3f_scale(n, -s) then v3f_add, CVector.cpp:128-149
            dcsg_descent_advance(d, x + m * nx, y + m * ny, z + m * nz, k, steps);
            k++;
            if (k == steps) d.done = true;
            if (d.done) {
                verts[(size_t)idx * 3 + 0] = d.x;
                verts[(size_t)idx * 3 + 1] = d.y;
                verts[(size_t)idx * 3 + 2] = d.z;
            }
        }
    }
}

#endif  // DCSG_PART_EXPORT (cell lists)
#endif

#ifdef DCSG_PART_EXPORT
// ---------------------------------------------------------------------------
// Vertex de-duplication for the projection.  The mesher emits an un-indexed triangle soup like the reference
// (mesh.hpp:283-307), so every surface vertex appears about six times; performGradientDescent moves each copy
// by the same function of its position (mesh.hpp:564-570), so projecting one representative per distinct
// position and copying the result is exact.  Keys are the three float bit patterns; open addressing with
// atomicCAS; the owner of a position is its first copy in the buffer (lowest index).  Owners are compacted in buffer
// order, which keeps neighbours in the mesh neighbours in the compact list (spatial coherence for the tile culling).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned int dcsg_hash3(unsigned int a, unsigned int b, unsigned int c)
{
    unsigned int h = a * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
    h ^= h >> 16;
    return h * 0x27D4EB2Fu;
}

// Pass 1: every vertex finds the table slot of its position; the slot ends up holding the LOWEST index among the
// copies (atomicMin), so that the owner of a position is its first occurrence in the buffer -- independent of the
// order in which threads arrive, and what lets the export project and write the mesh in chunks (every triangle before
// a cut refers only to owners before it).  rep[v] = the slot, resolved to the owner by dcsg_dedup_resolve.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_dedup_insert(const float* __restrict__ verts, unsigned int n, unsigned int* __restrict__ table, unsigned int mask,
                  unsigned int* __restrict__ rep)
{
    const unsigned int* vb = reinterpret_cast<const unsigned int*>(verts);
    for (unsigned int v = blockIdx.x * DCSG_BLOCK + threadIdx.x; v < n; v += gridDim.x * DCSG_BLOCK) {
        unsigned int a = vb[(size_t)v * 3], b = vb[(size_t)v * 3 + 1], c = vb[(size_t)v * 3 + 2];
        unsigned int h = dcsg_hash3(a, b, c) & mask;
        for (;;) {
            unsigned int old = atomicCAS(&table[h], 0xffffffffu, v);
            if (old == 0xffffffffu) break;
            // whichever copy the slot names at this moment has the slot's position
            if (vb[(size_t)old * 3] == a && vb[(size_t)old * 3 + 1] == b && vb[(size_t)old * 3 + 2] == c) {
                if (v < old) atomicMin(&table[h], v);  // the slot's index only ever decreases: nothing to do for a later copy
                break;
            }
            h = (h + 1u) & mask;
        }
        rep[v] = h;
    }
}
// Pass 2: rep[v] = the owner (lowest-index copy) of v's position
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_dedup_resolve(const unsigned int* __restrict__ table, unsigned int n, unsigned int* __restrict__ rep)
{
    for (unsigned int v = blockIdx.x * DCSG_BLOCK + threadIdx.x; v < n; v += gridDim.x * DCSG_BLOCK) rep[v] = table[rep[v]];
}

// per-block count of owners (rep[v] == v) in block_sums[2*b] (second word unused: dcsg_scan_blocks scans pairs)
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_dedup_count(const unsigned int* __restrict__ rep, unsigned int n, unsigned int* __restrict__ block_sums)
{
    unsigned int v = blockIdx.x * DCSG_BLOCK + threadIdx.x;
    unsigned int own = (v < n && rep[v] == v) ? 1u : 0u;
    unsigned int w = wave_sum(own);
    __shared__ unsigned int part[DCSG_BLOCK / DCSG_WAVE];
    if ((threadIdx.x & (DCSG_WAVE - 1)) == 0) part[threadIdx.x / DCSG_WAVE] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int k = 0; k < DCSG_BLOCK / DCSG_WAVE; k++) t += part[k];
        block_sums[2 * blockIdx.x] = t;
        block_sums[2 * blockIdx.x + 1] = 0;
    }
}

// owners -> compact[slot] (positions) in buffer order; slot_of[v] = slot for owners
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_dedup_gather(const float* __restrict__ verts, const unsigned int* __restrict__ rep, unsigned int n,
                  const unsigned int* __restrict__ block_base, float* __restrict__ compact,
                  unsigned int* __restrict__ slot_of)
{
    __shared__ unsigned int wsum[DCSG_BLOCK / DCSG_WAVE];
    unsigned int v = blockIdx.x * DCSG_BLOCK + threadIdx.x;
    unsigned int own = (v < n && rep[v] == v) ? 1u : 0u;
    unsigned int inc = wave_scan_incl(own);
    const int lane = threadIdx.x & (DCSG_WAVE - 1), wave = threadIdx.x / DCSG_WAVE;
    if (lane == DCSG_WAVE - 1) wsum[wave] = inc;
    __syncthreads();
    unsigned int off = block_base[2 * blockIdx.x] + inc - own;
    for (int k = 0; k < wave; k++) off += wsum[k];
    if (own) {
        slot_of[v] = off;
        compact[(size_t)off * 3 + 0] = verts[(size_t)v * 3 + 0];
        compact[(size_t)off * 3 + 1] = verts[(size_t)v * 3 + 1];
        compact[(size_t)off * 3 + 2] = verts[(size_t)v * 3 + 2];
    }
}

// every copy takes the projected position of its representative
// (copies [first, first + n): the export projects and writes the mesh in chunks)
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_dedup_scatter(float* __restrict__ verts, const unsigned int* __restrict__ rep, const unsigned int* __restrict__ slot_of,
                   const float* __restrict__ compact, unsigned int first, unsigned int n)
{
    for (unsigned int v = first + blockIdx.x * DCSG_BLOCK + threadIdx.x; v < first + n; v += gridDim.x * DCSG_BLOCK) {
        unsigned int s = slot_of[rep[v]];
        verts[(size_t)v * 3 + 0] = compact[(size_t)s * 3 + 0];
        verts[(size_t)v * 3 + 1] = compact[(size_t)s * 3 + 1];
        verts[(size_t)v * 3 + 2] = compact[(size_t)s * 3 + 2];
    }
}

#endif  // DCSG_PART_EXPORT (de-duplication)

#ifdef DCSG_PART_REST
// ---------------------------------------------------------------------------
// Opt-in: T-junction repair in the spirit of cms::retopologize (mesh.hpp:432-529).  The reference's function is
// undefined behaviour and unfinished; oracle/dcsg_oracle.c (oracle_retopologize) defines the semantics these kernels
// are held to, bit for bit: every cell (Indexer, mesh.hpp:413-421) remembers the mesh vertex of lowest index inside
// it; every triangle samples P points along each of its edges, and the occupant of a sample's cell joins the
// triangle's polygon when it is not an end of the edge, lies within joinDistance of the sample and was not accepted
// just before; the polygon is triangulated by the strip rule (geometry.hpp:228-248).  n polygon points give n - 2
// triangles; a triangle with nothing on its edges is reproduced as it is.  Three passes: mark (hash map insert,
// atomicMin of the vertex index), count (per-triangle accept masks + block sums for the scan), emit.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long dcsg_retopo_key(const DcsgRetopo& R, float x, float y, float z)
{
    const float v[3] = {x, y, z};
    unsigned long long key = 0ull;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float f = ((v[k] - R.c[k]) + R.hd[k]) / R.hd[k];  // diff, sum, termQuotient (mesh.hpp:415)
        f = f * R.grid_points;                            // scaled (:417)
        int i = (int)f;                                   // Vector3i(float, float, float) (:418)
        key = (key << 21) | (unsigned long long)((unsigned int)i & 0x1fffffu);
    }
    return key + 1ull;  // 0 marks an empty slot
}
__device__ __forceinline__ unsigned int dcsg_retopo_hash(unsigned long long k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    return (unsigned int)k;
}

// rep[] must be preset to 0xffffffff, keys[] to 0
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_retopo_mark(const float* __restrict__ verts, unsigned int n_verts, DcsgRetopo R, unsigned long long* __restrict__ keys,
                 unsigned int* __restrict__ rep, unsigned int mask)
{
    for (unsigned int v = blockIdx.x * DCSG_BLOCK + threadIdx.x; v < n_verts; v += gridDim.x * DCSG_BLOCK) {
        unsigned long long key = dcsg_retopo_key(R, verts[(size_t)v * 3], verts[(size_t)v * 3 + 1], verts[(size_t)v * 3 + 2]);
        unsigned int h = dcsg_retopo_hash(key) & mask;
        for (;;) {
            unsigned long long old = atomicCAS(&keys[h], 0ull, key);
            if (old == 0ull || old == key) break;
            h = (h + 1u) & mask;
        }
        atomicMin(&rep[h], v);  // the occupant is the vertex of lowest index: independent of the insertion order
    }
}

// occupant of the cell with this key, or 0xffffffff
__device__ __forceinline__ unsigned int dcsg_retopo_occupant(const unsigned long long* __restrict__ keys,
                                                             const unsigned int* __restrict__ rep, unsigned int mask,
                                                             unsigned long long key)
{
    unsigned int h = dcsg_retopo_hash(key) & mask;
    for (;;) {
        unsigned long long t = keys[h];
        if (t == key) return rep[h];
        if (t == 0ull) return 0xffffffffu;
        h = (h + 1u) & mask;
    }
}
// sample i of the edge a -> b (mesh.hpp:502-505): start + delta * ((float)i / pointsAlongEdge)
__device__ __forceinline__ void dcsg_retopo_sample(const float* a, const float* b, int i, int P, float q[3])
{
    float f = (float)i / (float)P;
    q[0] = a[0] + (b[0] - a[0]) * f;
    q[1] = a[1] + (b[1] - a[1]) * f;
    q[2] = a[2] + (b[2] - a[2]) * f;
}
__device__ __forceinline__ bool dcsg_same_bits3(const float* a, const float* b)
{
    return dcsg_same_bits(a[0], b[0]) && dcsg_same_bits(a[1], b[1]) && dcsg_same_bits(a[2], b[2]);
}
// the vertex that joins the polygon at sample i of edge a -> b, or 0xffffffff (`last`: the one accepted before it)
__device__ __forceinline__ unsigned int dcsg_retopo_joined(const float* __restrict__ verts, const DcsgRetopo& R,
                                                           const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ rep,
                                                           unsigned int mask, const float* a, const float* b, int i, unsigned int last)
{
    float q[3];
    dcsg_retopo_sample(a, b, i, R.points_along_edge, q);
    unsigned int r = dcsg_retopo_occupant(keys, rep, mask, dcsg_retopo_key(R, q[0], q[1], q[2]));
    if (r == 0xffffffffu || r == last) return 0xffffffffu;
    const float V[3] = {verts[(size_t)r * 3], verts[(size_t)r * 3 + 1], verts[(size_t)r * 3 + 2]};
    float ux = V[0] - q[0], uy = V[1] - q[1], uz = V[2] - q[2];
    if (dcsg_same_bits3(V, a) || dcsg_same_bits3(V, b) || !(ux * ux + uy * uy + uz * uz <= R.join_distance2)) return 0xffffffffu;
    return r;
}

// accept[3*t + e]: bit i set = the occupant of sample i's cell joins edge e of triangle t (bit 0: the start vertex).
// block_sums[2*b] = output triangles of block b.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_retopo_count(const float* __restrict__ tris, unsigned int n_tris, DcsgRetopo R, const unsigned long long* __restrict__ keys,
                  const unsigned int* __restrict__ rep, unsigned int mask, unsigned int* __restrict__ accept,
                  unsigned int* __restrict__ block_sums)
{
    unsigned int t = blockIdx.x * DCSG_BLOCK + threadIdx.x;
    unsigned int n_out = 0;
    if (t < n_tris) {
        float T[9];
        for (int k = 0; k < 9; k++) T[k] = tris[(size_t)t * 9 + k];
        unsigned int n = 0;
        for (int e = 0; e < 3; e++) {
            const float* a = T + 3 * e;
            const float* b = T + 3 * ((e + 1) % 3);
            unsigned int m = 1u, last = 0xffffffffu;
            for (int i = 1; i < R.points_along_edge; i++) {
                unsigned int r = dcsg_retopo_joined(tris, R, keys, rep, mask, a, b, i, last);
                if (r != 0xffffffffu) {
                    m |= 1u << i;
                    last = r;
                }
            }
            accept[(size_t)t * 3 + e] = m;
            n += (unsigned int)__popc(m);
        }
        n_out = n - 2;  // the strip rule yields n - 2 triangles (n >= 3)
    }
    unsigned int w = wave_sum(n_out);
    __shared__ unsigned int part[DCSG_BLOCK / DCSG_WAVE];
    if ((threadIdx.x & (DCSG_WAVE - 1)) == 0) part[threadIdx.x / DCSG_WAVE] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int s = 0;
        for (int k = 0; k < DCSG_BLOCK / DCSG_WAVE; k++) s += part[k];
        block_sums[2 * blockIdx.x] = s;
        block_sums[2 * blockIdx.x + 1] = 0;
    }
}

// point number k (0-based, edge order then sample order) of triangle T's polygon
__device__ __forceinline__ void dcsg_retopo_point(const float* __restrict__ verts, const DcsgRetopo& R,
                                                  const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ rep,
                                                  unsigned int mask, const float T[9], const unsigned int m[3], int k, float q[3])
{
    int e = 0;
    int c0 = __popc(m[0]), c1 = __popc(m[1]);
    if (k >= c0 + c1) {
        e = 2;
        k -= c0 + c1;
    } else if (k >= c0) {
        e = 1;
        k -= c0;
    }
    unsigned int mm = e == 0 ? m[0] : (e == 1 ? m[1] : m[2]);
    for (int j = 0; j < k; j++) mm &= mm - 1u;  // drop the k lowest set bits
    int i = __ffs((int)mm) - 1;
    const float* a = T + 3 * e;
    const float* b = T + 3 * ((e + 1) % 3);
    if (i == 0) {
        q[0] = a[0], q[1] = a[1], q[2] = a[2];
        return;
    }
    // the occupant of sample i's cell (it was accepted in the count pass: no test needed again)
    float s[3];
    dcsg_retopo_sample(a, b, i, R.points_along_edge, s);
    unsigned int r = dcsg_retopo_occupant(keys, rep, mask, dcsg_retopo_key(R, s[0], s[1], s[2]));
    q[0] = verts[(size_t)r * 3], q[1] = verts[(size_t)r * 3 + 1], q[2] = verts[(size_t)r * 3 + 2];
}

extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_retopo_emit(const float* __restrict__ tris, unsigned int n_tris, DcsgRetopo R, const unsigned long long* __restrict__ keys,
                 const unsigned int* __restrict__ rep, unsigned int mask, const unsigned int* __restrict__ accept,
                 const unsigned int* __restrict__ block_base, float* __restrict__ out)
{
    __shared__ unsigned int wsum[DCSG_BLOCK / DCSG_WAVE];
    unsigned int t = blockIdx.x * DCSG_BLOCK + threadIdx.x;
    unsigned int m[3] = {0u, 0u, 0u};
    int n = 0;
    if (t < n_tris) {
        for (int e = 0; e < 3; e++) {
            m[e] = accept[(size_t)t * 3 + e];
            n += __popc(m[e]);
        }
    }
    unsigned int n_out = n >= 3 ? (unsigned int)(n - 2) : 0u;
    unsigned int inc = wave_scan_incl(n_out);
    const int lane = threadIdx.x & (DCSG_WAVE - 1), wave = threadIdx.x / DCSG_WAVE;
    if (lane == DCSG_WAVE - 1) wsum[wave] = inc;
    __syncthreads();
    unsigned int off = block_base[2 * blockIdx.x] + inc - n_out;
    for (int k = 0; k < wave; k++) off += wsum[k];
    if (n_out == 0) return;
    float T[9];
    for (int k = 0; k < 9; k++) T[k] = tris[(size_t)t * 9 + k];
    // getIndexTriangleStrip (geometry.hpp:228-248) over the loop 0 .. n-1
    auto emit = [&](int a, int b, int c) {
        float* o = out + (size_t)off * 9;
        float q[3];
        dcsg_retopo_point(tris, R, keys, rep, mask, T, m, a, q);
        o[0] = q[0], o[1] = q[1], o[2] = q[2];
        dcsg_retopo_point(tris, R, keys, rep, mask, T, m, b, q);
        o[3] = q[0], o[4] = q[1], o[5] = q[2];
        dcsg_retopo_point(tris, R, keys, rep, mask, T, m, c, q);
        o[6] = q[0], o[7] = q[1], o[8] = q[2];
        off++;
    };
    int start = 0;
    if (n & 1) {
        emit(0, 1, n - 1);
        start = 1;  // pop_front
    }
    const int len = n - start;
    for (int A = 0; A < len / 2 - 1; A++) {
        int B = A + 1, D = len - 1 - A, C = D - 1;
        emit(start + A, start + B, start + C);
        emit(start + C, start + D, start + A);
    }
}

// Diagnostic: dcsg_sqrt over an array (exhaustive test of the custom square root).  In a scene module the copy the
// brush text uses (scene_prelude.cl) is evaluated as well and any difference between the two is reported as -1.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_test_sqrt(const float* __restrict__ in, float* __restrict__ out, unsigned int n)
{
    for (unsigned int i = blockIdx.x * DCSG_BLOCK + threadIdx.x; i < n; i += gridDim.x * DCSG_BLOCK) {
        float r = dcsg_sqrt(in[i]);
#ifdef DCSG_EXTERN_BANK
        float c = dcsg_sqrt_ieee(in[i]);
        if (__float_as_uint(c) != __float_as_uint(r) && !(c != c && r != r)) r = -1.0f;
#endif
        out[i] = r;
    }
}

// Diagnostic: dcsg_finish_normal(in[i], dy, dz) over an array (exhaustive test of the float-only gradient scale).
// mixed bit 0: odd lanes take the gradient (1, 2, 3) instead, so a wave holds ordinary and extreme operands side by side.
extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_test_normal(const float* __restrict__ in, float* __restrict__ out, unsigned int n, float dy, float dz, int mixed)
{
    for (unsigned int i = blockIdx.x * DCSG_BLOCK + threadIdx.x; i < n; i += gridDim.x * DCSG_BLOCK) {
        float nx, ny, nz;
        const bool other = (mixed & 1) && (i & 1u);
        const float Dx = other ? 1.0f : in[i], Dy = other ? 2.0f : dy, Dz = other ? 3.0f : dz;
        dcsg_finish_normal(Dx, Dy, Dz, nx, ny, nz);
        out[3u * i] = nx, out[3u * i + 1u] = ny, out[3u * i + 2u] = nz;
    }
}

#endif  // DCSG_PART_REST (T-junction repair, test kernels)

#if defined(DCSG_EXTERN_BANK) && !defined(DCSG_SET_SPEC) && defined(DCSG_PART_REST)
// ---------------------------------------------------------------------------
// Preview: kernel k1 (k1.cl:479-578) -- sphere tracing of the scene plus the three axis markers into a 640 x 480
// RGB image, one thread per pixel, the build procedure through the generic interpreter.  Scene modules only (the
// materials are OpenCL C text of scene.cl).  Arithmetic as the kernel text states it: float, left to right,
// comparisons against the double literals in double.  Checked against oracle_render_preview.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dcsg_axes_cylinder(float r, float h, float half_length, float radius)
{
    float a = __builtin_fabsf(h) - half_length, b = r - radius;  // axes_cylinderSDF, k1.cl:46-48
    return a > b ? a : b;
}

// primary_sdf of k1.cl:121-283: k2's build procedure, then the axis markers
__device__ __noinline__ float dcsg_k1_sdf(const SceneLds& s, float x, float y, float z)
{
    float ev = dcsg_primary_sdf(s, x, y, z);
    float vx = x / 5.0f, vy = y / 5.0f, vz = z / 5.0f;  // k1.cl:239
    const float radius = (float)0.015;                  // AXES_RADIUS
    float c;
    c = dcsg_axes_cylinder(dcsg_sqrt(vy * vy + vz * vz), vx - 0.5f, 0.5f, radius);
    ev = ev < c ? ev : c;
    c = dcsg_axes_cylinder(dcsg_sqrt(vx * vx + vz * vz), vy - 0.5f, 0.5f, radius);
    ev = ev < c ? ev : c;
    c = dcsg_axes_cylinder(dcsg_sqrt(vx * vx + vy * vy), vz - 0.5f, 0.5f, radius);
    ev = ev < c ? ev : c;
    return ev;
}

__device__ __forceinline__ unsigned char dcsg_colour_byte(float c)
{
    int v = (int)(255.0 * (double)c);  // RCOMP, k1.cl:8; v_cvt_i32_f64 saturates, NaN -> 0
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

extern "C" __global__ void __launch_bounds__(DCSG_BLOCK)
dcsg_preview(const uint4* __restrict__ blob, int n_vec4, DcsgCamera cam, int n_objects, unsigned char* __restrict__ out)
{
    SceneLds s = dcsg_stage_scene(blob, n_vec4);
    const int tid = (int)(blockIdx.x * DCSG_BLOCK + threadIdx.x);
    if (tid >= 640 * 480) return;
    {
        dcsg_cl_float3 r = {cam.r[0], cam.r[1], cam.r[2]}, u = {cam.u[0], cam.u[1], cam.u[2]}, f = {cam.f[0], cam.f[1], cam.f[2]};
        dcsg_scene_set_camera(r, u, f);  // rgt_g = rgt; ... every work-item, k1.cl:521-523
    }
    const int ix = tid % 640, iy = tid / 640;
    const float uvx = (float)(ix - 320) / 320.0f, uvy = -(float)(iy - 240) / 320.0f;  // k1.cl:515
    const float ox = cam.o[0] * cam.r[0] + cam.o[1] * cam.r[1] + cam.o[2] * cam.r[2];
    const float oy = cam.o[0] * cam.u[0] + cam.o[1] * cam.u[1] + cam.o[2] * cam.u[2];
    const float oz = cam.o[0] * cam.f[0] + cam.o[1] * cam.f[1] + cam.o[2] * cam.f[2];
    const float rx = uvx * cam.r[0] + uvy * cam.r[1] + 1.0f * cam.r[2];  // r = (uv.x, uv.y, IFOV)
    const float ry = uvx * cam.u[0] + uvy * cam.u[1] + 1.0f * cam.u[2];
    const float rz = uvx * cam.f[0] + uvy * cam.f[1] + 1.0f * cam.f[2];
    // march, k1.cl:419-470
    float d = 0.0f, vx = ox, vy = oy, vz = oz, hit = -1.0f;
    for (int i = 0; i < 512; i++) {
        float st = (float)((double)dcsg_k1_sdf(s, vx, vy, vz) * 0.85);
        if ((double)st < 0.005) {
            hit = d;
            break;
        }
        vx = vx + st * rx;
        vy = vy + st * ry;
        vz = vz + st * rz;
        d = d + st;
        if ((double)d > 64.0) break;
    }
    float cr = 1.0f, cg = 1.0f, cb = 1.0f;
    if ((double)hit > 0.0) {
        const float px = ox + hit * rx, py = oy + hit * ry, pz = oz + hit * rz;
        const float e = (float)K2_NORMAL_EPSILON, zero = 0.0f;
        float Dx = dcsg_k1_sdf(s, px + e, py + zero, pz + zero) - dcsg_k1_sdf(s, px - e, py - zero, pz - zero);
        float Dy = dcsg_k1_sdf(s, px + zero, py + e, pz + zero) - dcsg_k1_sdf(s, px - zero, py - e, pz - zero);
        float Dz = dcsg_k1_sdf(s, px + zero, py + zero, pz + e) - dcsg_k1_sdf(s, px - zero, py - zero, pz - e);
        float nx, ny, nz;
        dcsg_finish_normal(Dx, Dy, Dz, nx, ny, nz);
        // shade, k1.cl:285-379: the LAST object within 2*SDF_EPSILON supplies the material
        int match = -1;
        float ma = 0.0f, mb = 0.0f, mc = 0.0f;
        for (int i = 0; i < n_objects; i++) {
            float4 q0 = s.obj[4 * i + 0], q1 = s.obj[4 * i + 1], q2 = s.obj[4 * i + 2], q3 = s.obj[4 * i + 3];
            float dx = px - q0.x, dy = py - q0.y, dz = pz - q0.z;
            float A = dx * q1.x + dy * q1.y + dz * q1.z;
            float B = dx * q2.x + dy * q2.y + dz * q2.z;
            float C = dx * q3.x + dy * q3.y + dz * q3.z;
            float sd = dcsg_bank(A, B, C, __float_as_int(q0.w) & 0xff);
            if ((double)sd < 0.005 * 2.0) {
                match = i;
                ma = A, mb = B, mc = C;
            }
        }
        if (match != -1) {
            dcsg_cl_float3 gv = {px, py, pz}, lv = {ma, mb, mc}, n = {nx, ny, nz};
            int material = (__float_as_int(s.obj[4 * match].w) >> 8) & 0xff;
            dcsg_cl_float3 c = shader_bank(gv, lv, n, (unsigned char)material);
            cr = c[0], cg = c[1], cb = c[2];
        } else {
            float ax = px / 5.0f, ay = py / 5.0f, az = pz / 5.0f;
            const float radius = (float)0.025;
            cr = (float)(239.0 / 255.0), cg = (float)(66.0 / 255.0), cb = 0.0f;  // "245/255" is an integer division
            if ((double)dcsg_axes_cylinder(dcsg_sqrt(ay * ay + az * az), ax - 0.5f, 0.5f, radius) < 0.005 * 2.0) {
                cr = 1.0f, cg = 0.0f, cb = 0.0f;
            } else if ((double)dcsg_axes_cylinder(dcsg_sqrt(ax * ax + az * az), ay - 0.5f, 0.5f, radius) < 0.005 * 2.0) {
                cr = 0.0f, cg = 1.0f, cb = 0.0f;
            } else if ((double)dcsg_axes_cylinder(dcsg_sqrt(ax * ax + ay * ay), az - 0.5f, 0.5f, radius) < 0.005 * 2.0) {
                cr = 0.0f, cg = 0.0f, cb = 1.0f;
            }
        }
    }
    out[tid * 3 + 0] = dcsg_colour_byte(cr);
    out[tid * 3 + 1] = dcsg_colour_byte(cg);
    out[tid * 3 + 2] = dcsg_colour_byte(cb);
}
#endif

#ifdef DCSG_EXTERN_BANK
// Points the scene module's program-scope `arbitrary_data` (k2.cl:41,251) at the uploaded table.
extern "C" __global__ void dcsg_set_arbitrary_data(float* ad)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        dcsg_scene_set_ad((__attribute__((address_space(1))) float*)ad);
}
#endif
