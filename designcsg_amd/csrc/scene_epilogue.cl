
/* scene_epilogue.cl -- appended after the generated scene.cl (csrc/jit.cpp). */
void dcsg_scene_set_ad(__global float* p){ arbitrary_data = p; }
/* preview (kernel k1): the program-scope camera basis materials may read (k1.cl:521-523) */
void dcsg_scene_set_camera(float3 r, float3 u, float3 f){ rgt_g = r; upp_g = u; fwd_g = f; }
