
/* scene_epilogue_cpu.cl -- scalar entry points the C oracle binds (TEST ONLY) */
float dcsg_cpu_sdf_bank(float x, float y, float z, int shape_id){
    return sdf_bank((float3)(x,y,z), (unsigned char)shape_id);
}
void dcsg_cpu_set_arbitrary_data(__global float* p){ arbitrary_data = p; }
