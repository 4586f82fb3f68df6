# Own design script exercising the arbitrary-data table (addArbitraryData / getAD, k2.cl:22): a brush
# that unions spheres whose centres and radii are read from the table, plus a preprocessor define and an
# auxiliary function, as Logo.py-style designs do.
from DesignCSG import *
from designlibrary import *
import numpy as np

rng = np.random.default_rng(2024)
N = 12
centres = rng.uniform(-0.4, 0.4, (N, 3))
radii = rng.uniform(0.05, 0.15, N)
addArbitraryData("BLOB_CENTRES", list(centres.reshape(-1)))
addArbitraryData("BLOB_RADII", list(radii))

add_preprocessor_define(define="#define BLOB_COUNT %d" % N)

define_auxillary_function(function="""
float blob_sdf(float3 v){
    float d = MAX_DISTANCE;
    for(int i=0;i<BLOB_COUNT;i++){
        float3 c = (float3)(getAD(AD_BLOB_CENTRES,3*i+0),getAD(AD_BLOB_CENTRES,3*i+1),getAD(AD_BLOB_CENTRES,3*i+2));
        float s = length(v-c)-getAD(AD_BLOB_RADII,i);
        d = T_min(d,s);
    }
    return d;
}
""")

blob_brush = define_brush(body="return blob_sdf(v);")

draw(blob_brush, Transform.initial(np.array([0.0, 0.1, 0.0]), 0.2, 0.1, 0.0, np.array([1.2, 1.0, 1.1])))
erase(sphere_brush, Transform.initial(np.array([0.1, 0.0, 0.2]), 0.0, 0.0, 0.0, np.array([0.3, 0.3, 0.3])))

setExportConfig(2.0, 4, 6, 7, np.pi / 4, 20, 16, 512, 4096)
commit()
