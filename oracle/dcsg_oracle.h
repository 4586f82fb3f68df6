/*
 * dcsg_oracle.h -- CPU restatement of the DesignCSG export path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under designcsg_amd/ may include, link,
 * import or execute this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * Every function restates one piece of the reference (mikeandike523/DesignCSG,
 * paths relative to /root/reference/master) and cites it.  Parity pin: the
 * golden vectors of SURVEY.md section 8(c) (tests/test_oracle_golden.py).
 */
#ifndef DCSG_ORACLE_H
#define DCSG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* float sdf_bank(float3 v, unsigned char shape_id) flattened to scalars
 * (generated scene.cl, scenecompiler.py:489-499). */
typedef float (*oracle_bank_fn)(float x, float y, float z, int shape_id);

typedef struct oracle_scene {
    int n_objects;            /* scene.txt rows        (DrawPane.cpp:267-353) */
    unsigned char *shape_id;  /* [n_objects]                                  */
    float *pos, *right, *up, *fwd; /* [3*n_objects] each                      */
    int n_steps;              /* buildprocedure.txt rows (DrawPane.cpp:355-371) */
    int *steps;               /* [4*n_steps] opcode,left,right,dest           */
    oracle_bank_fn bank;
    unsigned char *material_id; /* [n_objects] second column of scene.txt (preview only) */
    /* preview (k1.cl) hooks into a compiled scene.cl; NULL = the two materials every scene starts with */
    void (*shader)(const float gv[3], const float lv[3], const float n[3], int material_id, float out[3]);
    void (*set_camera)(const float right[3], const float up[3], const float forward[3]);
} oracle_scene;

typedef struct oracle_box {   /* box_t{center,diameters} CVector.h:74-78 */
    float center[3];
    float diameters[3];
} oracle_box;

typedef struct oracle_export_config { /* exportConfig.txt, DesignCSG.cpp:827-835 */
    float bounding_box_half_diameter;
    int min_level, max_level, grid_level;
    float complex_surface_threshold;
    int gradient_descent_steps;
    int cache_subdivision, queries_before_gc, queries_before_free;
} oracle_export_config;

typedef struct oracle_mesh_stats {
    uint64_t nodes_visited[32];      /* per octree level */
    uint64_t nodes_culled[32];
    uint64_t nodes_subdivided[32];
    uint64_t triangles[32];          /* emitted by leaves of that level */
    uint64_t sdf_samples, normal_samples; /* unique lattice points evaluated */
} oracle_mesh_stats;

/* ---- scene ---------------------------------------------------------- */
oracle_scene *oracle_scene_parse(const char *scene_txt, const char *buildprocedure_txt);
void oracle_scene_free(oracle_scene *s);
void oracle_scene_set_bank(oracle_scene *s, oracle_bank_fn bank);
/* the five built-in brushes sd0..sd4 (scenecompiler.py:424-425, DesignCSG.py:9-22) */
float oracle_builtin_bank(float x, float y, float z, int shape_id);
int oracle_parse_export_config(const char *text, oracle_export_config *out);

/* ---- k2.cl ---------------------------------------------------------- */
float oracle_primary_sdf(const oracle_scene *s, float x, float y, float z);
void oracle_get_normal(const oracle_scene *s, float x, float y, float z, float out[3]);
void oracle_eval_sdf(const oracle_scene *s, const float *xyz, size_t n, float *out, int threads);
void oracle_eval_normal(const oracle_scene *s, const float *xyz, size_t n, float *out_xyz, int threads);

/* ---- ISV.hpp lattice ------------------------------------------------ */
void oracle_lattice_point(const oracle_box *bb, int res, int64_t ix, int64_t iy, int64_t iz, float out[3]);
void oracle_lattice_coords(const oracle_box *bb, int res, const float p[3], int64_t out[3]);
/* dense planes z in [z0,z1) of the (res+1)^3 lattice, x fastest; either output may be NULL */
void oracle_eval_lattice(const oracle_scene *s, const oracle_box *bb, int res, int z0, int z1,
                         float *sdf, float *normals, int threads);

/* ---- DesignCSG.cpp:668-712 stage A ---------------------------------- */
void oracle_estimate_bounding_box(const oracle_scene *s, float bb_half_diameter, oracle_box *out,
                                  uint64_t *interior_points, int threads);

/* ---- lookupTable.txt (readLookupTable.hpp:32-76, geometry.hpp:228-248) */
/* fills tri_count[256] and tris[256][5][3] (edge ids); returns total triangles or -1 */
int oracle_parse_lookup_table(const char *text, int tri_count[256], int tris[256][5][3]);

/* ---- mesh.hpp -------------------------------------------------------- */
/* getSurface: returns number of triangles; *out = malloc'd 9 floats per triangle (A,B,C xyz) */
size_t oracle_get_surface(const oracle_scene *s, const oracle_box *bb, const oracle_export_config *cfg,
                          const int tri_count[256], const int tris[256][5][3],
                          float **out, oracle_mesh_stats *stats, int threads);
void oracle_gradient_descent(const oracle_scene *s, int steps, float *tris, size_t n_tris, int threads);
/* T-junction repair in the spirit of cms::retopologize (mesh.hpp:432-529): OWN DEFINITION, PARITY UNPINNED (the
 * reference's function is undefined behaviour and unfinished; see dcsg_oracle.c).  Returns the triangle count; *out is
 * malloc'd (oracle_free). */
float oracle_retopo_join_distance(const float half_diameter[3], int grid_level);
size_t oracle_retopologize(const float *tris, size_t n_tris, const float center[3], const float half_diameter[3],
                           int min_level, int grid_level, float **out);
void oracle_free(void *p);

/* ---- utils.hpp writers ---------------------------------------------- */
/* kernel k1 (k1.cl:479-578): the 640x480 RGB preview image, rows top to bottom, 3 bytes per pixel.
 * PARITY UNPINNED: the reference holds no image fixture and its OpenCL kernel cannot run in the build
 * container; this is a restatement of the kernel text only.                                            */
void oracle_render_preview(const oracle_scene *s, const float campos[3], const float right[3], const float up[3],
                           const float forward[3], unsigned char *rgb_out, int threads);
void oracle_scene_set_shader(oracle_scene *s,
                             void (*shader)(const float *, const float *, const float *, int, float *),
                             void (*set_camera)(const float *, const float *, const float *));

int oracle_write_stl(const char *path, const float *tris, size_t n_tris);
int oracle_write_ply(const char *path, const float *tris, size_t n_tris);

#ifdef __cplusplus
}
#endif
#endif
