# Own design script: 128 draws of the design's OWN brushes (a torus, a rounded box and a capped cone written as OpenCL C
# in define_brush, scenecompiler.py:227-241), a few of them carved -- more object evaluations than is inlined into every
# kernel (48), so the scene takes the chunked per-scene code (csrc/specialise.cpp: emit_chunked).
from DesignCSG import *
from designlibrary import *
import numpy as np

define_auxillary_function(function="""
float ub_round_box(float3 v, float3 h, float r){
    float3 q = fabs(v) - h;
    return length(max(q, 0.0f)) + min(max(q.x, max(q.y, q.z)), 0.0f) - r;
}
""")

torus_brush = define_brush(body="""
    float2 q = (float2)(length((float2)(v.x, v.z)) - 0.35f, v.y);
    return length(q) - 0.12f;
""")
round_box_brush = define_brush(body="""
    return ub_round_box(v, (float3)(0.32f, 0.22f, 0.27f), 0.08f);
""")
cone_brush = define_brush(body="""
    float r = length((float2)(v.x, v.z));
    float a = r - (0.25f - 0.4f * v.y) ;
    float b = fabs(v.y) - 0.4f;
    return T_max(a * 0.9284767f, b);
""")

rng = np.random.default_rng(4242)
brushes = [torus_brush, round_box_brush, cone_brush]
for i in range(128):
    p = rng.uniform(-0.8, 0.8, 3)
    s = rng.uniform(0.08, 0.22, 3)
    a = rng.uniform(0, np.pi, 3)
    t = Transform.initial(p, a[0], a[1], a[2], s)
    if i % 16 == 15:
        erase(brushes[i % 3], t)
    else:
        draw(brushes[i % 3], t)

setExportConfig(2.0, 6, 8, 9, np.pi / 4, 50, 16, 512, 4096)
commit()
