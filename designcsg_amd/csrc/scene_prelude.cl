/*
 * scene_prelude.cl -- what the MI355X scene module puts in front of a generated scene.cl.
 *
 * The user's brush bodies are OpenCL C text inside the compiled scene format
 * (scenecompiler.py:476-522); they go through the OpenCL C front-end to gfx950 bitcode
 * and are linked into the HIP kernels of kernels.hip (csrc/jit.cpp).  This file restates the
 * environment k2.cl gives them (k2.cl:2,22,29-41: MAX_DISTANCE, getAD, T_min/T_max,
 * scaleFloat3, the bank prototypes, the program-scope globals) and pins the geometric
 * built-ins the reference leaves to the OpenCL vendor (k2.cl:111,177; DesignCSG.py:9-22)
 * as correctly rounded, left-to-right compositions, so that results are reproducible
 * bit for bit (tests compare with the CPU oracle).  Every other built-in (sin, pow, ...)
 * resolves to the ROCm device library.
 */
#define MAX_DISTANCE 64.0
#define getAD(name,offset) (arbitrary_data[name+offset])
#define T_min(a,b) (a<b?a:b)
#define T_max(a,b) (a>b?a:b)

__global float3 rgt_g;
__global float3 upp_g;
__global float3 fwd_g;
__global float * arbitrary_data;

float3 scaleFloat3(float s, float3 v){ return (float3)(s*v.x,s*v.y,s*v.z); }
float sdf_bank(float3 v, unsigned char shape_id);
float3 shader_bank(float3 gv, float3 lv, float3 n, unsigned char material_id);

#define DCSG_OVL __attribute__((overloadable))
/* Correctly rounded sqrtf: same sequence as dcsg_sqrt in kernels.hip (rsq + one FMA residual correction for waves
 * whose inputs are all in [2^-96, inf), the compiler's full IEEE expansion otherwise). */
#ifdef DCSG_RELAXED
/* opt-in relaxed numerics (dcsg_set_numerics): the hardware square root, 1 ulp */
float dcsg_sqrt_ieee(float x){ return __builtin_amdgcn_sqrtf(x); }
#else
float dcsg_sqrt_ieee(float x){
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    float d = __builtin_fmaf(-g, g, x);
    float r = __builtin_fmaf(d, h, g);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(as_uint(g) - 0x27800000u < 0x58000000u)) != 0ul, 0)) r = __builtin_sqrtf(x);
    return r;
}
#endif
float  DCSG_OVL sqrt(float x){ return dcsg_sqrt_ieee(x); }
double DCSG_OVL sqrt(double x){ return __builtin_sqrt(x); }
float  DCSG_OVL fabs(float x){ return __builtin_fabsf(x); }
double DCSG_OVL fabs(double x){ return __builtin_fabs(x); }
float2 DCSG_OVL fabs(float2 v){ return (float2)(__builtin_fabsf(v.x),__builtin_fabsf(v.y)); }
float3 DCSG_OVL fabs(float3 v){ return (float3)(__builtin_fabsf(v.x),__builtin_fabsf(v.y),__builtin_fabsf(v.z)); }
float4 DCSG_OVL fabs(float4 v){ return (float4)(__builtin_fabsf(v.x),__builtin_fabsf(v.y),__builtin_fabsf(v.z),__builtin_fabsf(v.w)); }
float  DCSG_OVL dot(float2 a, float2 b){ return a.x*b.x+a.y*b.y; }
float  DCSG_OVL dot(float3 a, float3 b){ return a.x*b.x+a.y*b.y+a.z*b.z; }
float  DCSG_OVL dot(float4 a, float4 b){ return a.x*b.x+a.y*b.y+a.z*b.z+a.w*b.w; }
float  DCSG_OVL length(float a){ return __builtin_fabsf(a); }
float  DCSG_OVL length(float2 a){ return dcsg_sqrt_ieee(a.x*a.x+a.y*a.y); }
float  DCSG_OVL length(float3 a){ return dcsg_sqrt_ieee(a.x*a.x+a.y*a.y+a.z*a.z); }
float  DCSG_OVL length(float4 a){ return dcsg_sqrt_ieee(a.x*a.x+a.y*a.y+a.z*a.z+a.w*a.w); }
float  DCSG_OVL distance(float3 a, float3 b){ return length(a-b); }
float3 DCSG_OVL normalize(float3 v){
    float l = length(v);
    if (l == 0.0f) return (float3)(0.0f,0.0f,0.0f);
    return (float3)(v.x/l, v.y/l, v.z/l);
}
float  DCSG_OVL max(float x, float y){ return x < y ? y : x; }
float  DCSG_OVL min(float x, float y){ return y < x ? y : x; }
double DCSG_OVL max(double x, double y){ return x < y ? y : x; }
double DCSG_OVL min(double x, double y){ return y < x ? y : x; }
int    DCSG_OVL max(int x, int y){ return x < y ? y : x; }
int    DCSG_OVL min(int x, int y){ return y < x ? y : x; }
float3 DCSG_OVL max(float3 a, float3 b){ return (float3)(max(a.x,b.x),max(a.y,b.y),max(a.z,b.z)); }
float3 DCSG_OVL min(float3 a, float3 b){ return (float3)(min(a.x,b.x),min(a.y,b.y),min(a.z,b.z)); }
float3 DCSG_OVL max(float3 a, float b){ return (float3)(max(a.x,b),max(a.y,b),max(a.z,b)); }
float3 DCSG_OVL min(float3 a, float b){ return (float3)(min(a.x,b),min(a.y,b),min(a.z,b)); }
float  DCSG_OVL fmax(float x, float y){ return x < y ? y : x; }
float  DCSG_OVL fmin(float x, float y){ return y < x ? y : x; }
float  DCSG_OVL clamp(float x, float lo, float hi){ return min(max(x,lo),hi); }
uint   DCSG_OVL abs(int a){ return (uint)(a < 0 ? -a : a); }
float  DCSG_OVL floor(float x){ return __builtin_floorf(x); }
float  DCSG_OVL ceil(float x){ return __builtin_ceilf(x); }
float  DCSG_OVL trunc(float x){ return __builtin_truncf(x); }
float2 DCSG_OVL max(float2 a, float2 b){ return (float2)(max(a.x,b.x),max(a.y,b.y)); }
float2 DCSG_OVL min(float2 a, float2 b){ return (float2)(min(a.x,b.x),min(a.y,b.y)); }
float2 DCSG_OVL max(float2 a, float b){ return (float2)(max(a.x,b),max(a.y,b)); }
float2 DCSG_OVL min(float2 a, float b){ return (float2)(min(a.x,b),min(a.y,b)); }
float2 DCSG_OVL clamp(float2 x, float lo, float hi){ return min(max(x,lo),hi); }
float3 DCSG_OVL clamp(float3 x, float lo, float hi){ return min(max(x,lo),hi); }
float3 DCSG_OVL clamp(float3 x, float3 lo, float3 hi){ return min(max(x,lo),hi); }
int    DCSG_OVL clamp(int x, int lo, int hi){ return min(max(x,lo),hi); }
float  DCSG_OVL mix(float x, float y, float a){ return x+(y-x)*a; }
float3 DCSG_OVL mix(float3 x, float3 y, float a){ return (float3)(mix(x.x,y.x,a),mix(x.y,y.y,a),mix(x.z,y.z,a)); }
float  DCSG_OVL step(float edge, float x){ return x < edge ? 0.0f : 1.0f; }
float  DCSG_OVL sign(float x){ return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f)); }
float  DCSG_OVL mad(float a, float b, float c){ return a*b+c; }
float  DCSG_OVL distance(float2 a, float2 b){ return length(a-b); }
float3 DCSG_OVL cross(float3 a, float3 b){ return (float3)(a.y*b.z-a.z*b.y, a.z*b.x-a.x*b.z, a.x*b.y-a.y*b.x); }
