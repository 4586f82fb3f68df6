
/* scene_epilogue_cpu.cl -- scalar entry points the C oracle binds (TEST ONLY) */
float dcsg_cpu_sdf_bank(float x, float y, float z, int shape_id){
    return sdf_bank((float3)(x,y,z), (unsigned char)shape_id);
}
void dcsg_cpu_set_arbitrary_data(__global float* p){ arbitrary_data = p; }
/* preview hooks (k1.cl:521-523, shader_bank) */
void dcsg_cpu_shader_bank(__global const float* gv, __global const float* lv, __global const float* n, int material_id,
                          __global float* out){
    float3 c = shader_bank((float3)(gv[0],gv[1],gv[2]), (float3)(lv[0],lv[1],lv[2]), (float3)(n[0],n[1],n[2]),
                           (unsigned char)material_id);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void dcsg_cpu_set_camera(__global const float* r, __global const float* u, __global const float* f){
    rgt_g = (float3)(r[0],r[1],r[2]); upp_g = (float3)(u[0],u[1],u[2]); fwd_g = (float3)(f[0],f[1],f[2]);
}
