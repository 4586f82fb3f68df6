
/* scene_epilogue.cl -- appended after the generated scene.cl (csrc/jit.cpp). */
void dcsg_scene_set_ad(__global float* p){ arbitrary_data = p; }
