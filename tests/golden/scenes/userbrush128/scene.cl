
        


        

         

float ub_round_box(float3 v, float3 h, float r){
    float3 q = fabs(v) - h;
    return length(max(q, 0.0f)) + min(max(q.x, max(q.y, q.z)), 0.0f) - r;
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

            
    float2 q = (float2)(length((float2)(v.x, v.z)) - 0.35f, v.y);
    return length(q) - 0.12f;


        }
        

        float sd6( float3 v){

            
    return ub_round_box(v, (float3)(0.32f, 0.22f, 0.27f), 0.08f);


        }
        

        float sd7( float3 v){

            
    float r = length((float2)(v.x, v.z));
    float a = r - (0.25f - 0.4f * v.y) ;
    float b = fabs(v.y) - 0.4f;
    return T_max(a * 0.9284767f, b);


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
        
        
        