# Own design script for the brush JIT: program-scope __constant tables, a typedef'd struct, helper
# functions with loops and early returns, integer indexing, vector built-ins (cross / mix / clamp / step /
# sign / fabs / max on float2 and float3) and arbitrary data next to them -- the OpenCL C a design may put
# into define_auxillary_function / define_brush (scenecompiler.py:476-522).
from DesignCSG import *
from designlibrary import *
import numpy as np

addArbitraryData("RING_RADII", [0.55, 0.08, 0.35, 0.05])

add_preprocessor_define(define="#define N_PEGS 6")
add_preprocessor_define(define="#define PEG_R 0.07f")

define_auxillary_function(function="""
__constant float PEG_XY[2*N_PEGS] = { 0.45f, 0.0f,  0.225f, 0.3897f,  -0.225f, 0.3897f,
                                      -0.45f, 0.0f, -0.225f, -0.3897f, 0.225f, -0.3897f };
__constant int PEG_ORDER[N_PEGS] = { 3, 0, 5, 2, 4, 1 };

typedef struct { float3 c; float3 h; } slab_t;

float slab_sdf(float3 v, slab_t s){
    float3 q = fabs(v - s.c) - s.h;
    return length(max(q, 0.0f)) + min(max(q.x, max(q.y, q.z)), 0.0f);
}

float torus_sdf(float3 v, float R, float r){
    float2 q = (float2)(length((float2)(v.x, v.z)) - R, v.y);
    return length(q) - r;
}

float pegs_sdf(float3 v){
    float d = MAX_DISTANCE;
    for(int k = 0; k < N_PEGS; k++){
        int i = PEG_ORDER[k];
        float2 c = (float2)(PEG_XY[2*i], PEG_XY[2*i+1]);
        float2 w = (float2)(length((float2)(v.x, v.z) - c) - PEG_R, fabs(v.y) - (0.1f + 0.03f*(float)i));
        float s = min(max(w.x, w.y), 0.0f) + length(max(w, 0.0f));
        if(s < d) d = s;
    }
    return d;
}
""")

define_auxillary_function(function="""
float wedge_sdf(float3 v){
    float3 n = cross((float3)(0.0f, 1.0f, 0.25f), (float3)(1.0f, 0.0f, 0.0f));
    float t = clamp(v.y * 2.0f + 0.5f, 0.0f, 1.0f);
    float w = mix(0.30f, 0.12f, t) * (0.5f + 0.5f*step(0.0f, v.x)) ;
    slab_t s;
    s.c = (float3)(0.0f, 0.0f, 0.0f);
    s.h = (float3)(w, 0.25f, 0.2f);
    return T_max(slab_sdf(v, s), dot(v, n) * sign(n.z) - 0.22f);
}
""")

ring_brush = define_brush(body="""
    float a = torus_sdf(v, getAD(AD_RING_RADII,0), getAD(AD_RING_RADII,1));
    float b = torus_sdf((float3)(v.x, v.y - 0.12f, v.z), getAD(AD_RING_RADII,2), getAD(AD_RING_RADII,3));
    return T_min(a, b);
""")
pegs_brush = define_brush(body="return pegs_sdf(v);")
wedge_brush = define_brush(body="return wedge_sdf(v);")

draw(ring_brush, Transform.initial(np.array([0.0, 0.0, 0.0]), 0.0, 0.0, 0.0, np.array([1.0, 1.0, 1.0])))
draw(pegs_brush, Transform.initial(np.array([0.0, 0.02, 0.0]), 0.3, 0.0, 0.0, np.array([1.0, 1.2, 1.0])))
draw(wedge_brush, Transform.initial(np.array([0.0, 0.1, 0.0]), 0.0, 0.2, 0.1, np.array([0.9, 0.9, 0.9])))
erase(box_brush, Transform.initial(np.array([0.3, 0.1, 0.0]), 0.5, 0.0, 0.0, np.array([0.08, 0.4, 0.08])))

setExportConfig(2.0, 4, 6, 7, np.pi / 4, 20, 16, 512, 4096)
commit()
