# Own design script exercising PROGRAM-SCOPE MUTABLE __global VARIABLES in brush text, the pattern of the reference's
# Designs/Logo.py:183,270,293: an auxiliary-function block declares `__global int LETTER_AD_OFFS = -1;`, every letter
# brush assigns its own value to it and then calls a shared function that reads it back.  Here: two globals (an int
# selector and a float gain), three brushes that set them to different values before calling the shared shape function.
# (Like Logo.py, a brush always writes before it reads; what one work-item leaves behind is never read by another.)
from DesignCSG import *
from designlibrary import *
import numpy as np

define_auxillary_function(function="""
__global int SHAPE_SEL = -1;
__global float SHAPE_GAIN = 0.0;

float selected_shape(float3 v){
    float3 a = fabs(v);
    float g = SHAPE_GAIN;
    if(SHAPE_SEL==0){
        return length(v)-0.5*g;
    }
    if(SHAPE_SEL==1){
        return T_max(a.x-0.5*g,T_max(a.y-0.25,a.z-0.5*g));
    }
    if(SHAPE_SEL==2){
        return T_max(length((float2)(v.x,v.z))-0.5*g,a.y-0.5);
    }
    return MAX_DISTANCE;
}
""")

ball_brush = define_brush(body="SHAPE_SEL = 0; SHAPE_GAIN = 0.75; return selected_shape(v);")
slab_brush = define_brush(body="SHAPE_SEL = 1; SHAPE_GAIN = 1.0; return selected_shape(v);")
rod_brush = define_brush(body="SHAPE_SEL = 2; SHAPE_GAIN = 0.5; return selected_shape(v);")

draw(slab_brush, Transform.initial(np.array([0.0, -0.2, 0.0]), 0.3, 0.0, 0.1, np.array([1.4, 0.8, 1.2])))
draw(ball_brush, Transform.initial(np.array([0.2, 0.15, -0.1]), 0.0, 0.0, 0.0, np.array([0.9, 0.9, 0.9])))
erase(rod_brush, Transform.initial(np.array([-0.1, 0.0, 0.1]), 0.0, 0.4, 0.0, np.array([0.5, 2.0, 0.5])))
draw(ball_brush, Transform.initial(np.array([-0.45, 0.3, 0.35]), 0.0, 0.0, 0.0, np.array([0.4, 0.4, 0.4])))

setExportConfig(2.0, 4, 6, 7, np.pi / 4, 20, 16, 512, 4096)
commit()
