
        
#define AD_RING_RADII 0


        #define N_PEGS 6
#define PEG_R 0.07f

         

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


float wedge_sdf(float3 v){
    float3 n = cross((float3)(0.0f, 1.0f, 0.25f), (float3)(1.0f, 0.0f, 0.0f));
    float t = clamp(v.y * 2.0f + 0.5f, 0.0f, 1.0f);
    float w = mix(0.30f, 0.12f, t) * (0.5f + 0.5f*step(0.0f, v.x)) ;
    slab_t s;
    s.c = (float3)(0.0f, 0.0f, 0.0f);
    s.h = (float3)(w, 0.25f, 0.2f);
    return T_max(slab_sdf(v, s), dot(v, n) * sign(n.z) - 0.22f);
}


        
        float sd0( float3 v){

            return MAX_DISTANCE;

        }
        

        float sd1( float3 v){

            return 0.0;

        }
        

        float sd2( float3 v){

            return length(v)-0.5;

        }
        

        float sd3( float3 v){

            

    v=fabs(v);
    float x = length((float2)(v.x,v.z));
    float y = v.y;
    return max(x-0.5,y-0.5);



        }
        

        float sd4( float3 v){

            
    v=fabs(v);
    return max(v.x-0.5,max(v.y-0.5,v.z-0.5));


        }
        

        float sd5( float3 v){

            
    float a = torus_sdf(v, getAD(AD_RING_RADII,0), getAD(AD_RING_RADII,1));
    float b = torus_sdf((float3)(v.x, v.y - 0.12f, v.z), getAD(AD_RING_RADII,2), getAD(AD_RING_RADII,3));
    return T_min(a, b);


        }
        

        float sd6( float3 v){

            return pegs_sdf(v);

        }
        

        float sd7( float3 v){

            return wedge_sdf(v);

        }
        

        
        float3 shader0 (float3 gv, float3 lv, float3 n){

            return fabs(n);

        }
        

        float3 shader1 (float3 gv, float3 lv, float3 n){

            
        
        float3 n_g = n.x*rgt_g+n.y*upp_g+n.z*fwd_g;

        float L = dot(n_g,(float3)(0.0,0.0,-1.0)); return (float3)(L,L,L);



        

        }
        


        float sdf_bank(float3 v, unsigned char shape_id){

            switch(shape_id){

                
case 0: return sd0(v); break;


case 1: return sd1(v); break;


case 2: return sd2(v); break;


case 3: return sd3(v); break;


case 4: return sd4(v); break;


case 5: return sd5(v); break;


case 6: return sd6(v); break;


case 7: return sd7(v); break;


            }

            return 0.0;

        }

        float3 shader_bank(float3 gv, float3 lv, float3 n, unsigned char material_id){


            switch(material_id){

                
case 0: return shader0(gv,lv,n); break;


case 1: return shader1(gv,lv,n); break;


            }

            return (float3)(1.0, 1.0, 1.0);
        }
        
        
        