/*
 * dcsg.h -- C ABI of libdcsg_hip.so: the MI355X-native export path of DesignCSG.
 *
 * Drop-in boundary.  The reference has no FFI; its export path sits behind
 *   (1) the compiled scene files written by the Python design API
 *       (scene.txt, buildprocedure.txt, scene.cl, arbitrary_data.hex, exportConfig.txt;
 *        scenecompiler.py:466-582, DesignCSG.py:205-237), and
 *   (2) class Evaluator (Evaluator.h:20-57), which the GUI's export thread
 *       (DesignCSG.cpp:638-790, OnExportInner) drives.
 * Each entry point below names the reference interface it replaces
 * (paths relative to /root/reference/master).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative dcsg_status and records a message retrievable with dcsg_last_error().  "host"
 * pointers are ordinary memory; "_device" entry points take HIP device pointers (e.g.
 * torch tensor data_ptr()) and a hipStream_t passed as void*, and do not synchronise.
 * A context is bound to one GPU and is not re-entrant (the reference serialises every
 * evaluator call behind one global mutex, Evaluator.cpp:10,120,170); use one context per
 * thread or lock around it.
 */
#ifndef DCSG_H
#define DCSG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCSG_ARBITRARY_DATA_POINTS 131072 /* Evaluator.h:17 */
#define DCSG_MAX_LEVELS 16

typedef struct dcsg_ctx dcsg_ctx;

typedef enum dcsg_status {
    DCSG_OK = 0,
    DCSG_ERR_INVALID = -1,   /* bad argument / malformed scene file */
    DCSG_ERR_NO_DEVICE = -2, /* no usable HIP device: there is no CPU fallback */
    DCSG_ERR_HIP = -3,       /* HIP runtime error */
    DCSG_ERR_BUILD = -4,     /* scene.cl failed to compile; log in dcsg_last_error (Evaluator.cpp:64-89) */
    DCSG_ERR_NO_SCENE = -5,  /* evaluation before dcsg_load_scene ("Cannot Export Yet.", DesignCSG.cpp:786) */
    DCSG_ERR_IO = -6
} dcsg_status;

/* exportConfig.txt, parsed positionally like DesignCSG.cpp:827-835 */
typedef struct dcsg_export_config {
    float bounding_box_half_diameter; /* already multiplied by 5 by setExportConfig (DesignCSG.py:225) */
    int32_t min_level, max_level, grid_level;
    float complex_surface_threshold;  /* radians */
    int32_t gradient_descent_steps;
    int32_t cache_subdivision, queries_before_gc, queries_before_free; /* accepted, unused: no ISV cache */
} dcsg_export_config;

typedef struct dcsg_mesh_stats {
    uint64_t nodes[DCSG_MAX_LEVELS];      /* nodes visited per octree level */
    uint64_t culled[DCSG_MAX_LEVELS];     /* mesh.hpp:170 */
    uint64_t subdivided[DCSG_MAX_LEVELS];
    uint64_t leaves[DCSG_MAX_LEVELS];
    uint64_t triangles[DCSG_MAX_LEVELS];  /* emitted by leaves of that level */
} dcsg_mesh_stats;

typedef struct dcsg_export_stats {
    float box_center[3];
    float box_diameter;
    uint64_t interior_points;  /* stage A: samples with sdf < bbHalf/256 */
    uint64_t lattice_points;   /* (res+1)^3 */
    uint64_t triangles;
    dcsg_mesh_stats mesh;
    /* wall-clock seconds per stage (host clock, device synchronised at stage ends) */
    double t_bbox, t_sweep, t_mesh, t_project, t_download, t_write_stl, t_write_ply, t_total;
    /* device time of the lattice sweep kernel(s), milliseconds (HIP events) */
    double sweep_kernel_ms;
    /* lattice points actually evaluated by the sweep: lattice_points for a dense sweep, fewer for the node-wise
     * sweep (nodes x (2^(grid-min)+1)^3, shared faces counted per node) */
    uint64_t swept_points;
    uint64_t swept_nodes;      /* 0 for a dense sweep */
    /* the lattice planes this export owned: [slab_z_lo, slab_z_hi) of 0 .. 2^gridLevel (the whole range unless dcsg_export_slab) */
    int32_t slab_z_lo, slab_z_hi;
} dcsg_export_stats;

/* ---- life cycle ------------------------------------------------------------------- */
/* Evaluator::Evaluator (Evaluator.cpp:14-40): binds a device, creates staging buffers. */
int dcsg_create(dcsg_ctx** out, int device_id);
void dcsg_destroy(dcsg_ctx* ctx);
/* Message of the last failure on ctx (or of dcsg_create when ctx is NULL). */
const char* dcsg_last_error(const dcsg_ctx* ctx);

/* ---- scene ------------------------------------------------------------------------ */
/* BasicDrawPane::loadScene parsing (DrawPane.cpp:267-371) + Evaluator::build
 * (Evaluator.cpp:45-112: clBuildProgram of k2.cl+scene.cl) + setArbitraryData.
 * scene_cl may be NULL when every brush id in scene.txt is one of the five built-ins 0..4.
 * arbitrary_data may be NULL (all zeros); n_arbitrary <= DCSG_ARBITRARY_DATA_POINTS.
 * On a brush compile error returns DCSG_ERR_BUILD with the compiler log as the error string,
 * like Evaluator::build returning (-1, log).
 * A scene's code comes in two code objects: the part dcsg_export launches (stage A, node-wise sweep, mesher, projection)
 * is ready when this call returns; the rest (dense sweeps, point evaluation, T-junction repair) is loaded on first use --
 * from the on-disk cache, or, after a cold build, from a host thread that compiles it in the background while the caller
 * already exports (the first dcsg_eval_* / dcsg_sweep_* call then waits for that thread; a compile error of the rest
 * surfaces there as DCSG_ERR_BUILD).  DCSG_JIT_SYNC=1 in the environment builds both parts inside this call. */
int dcsg_load_scene(dcsg_ctx* ctx, const char* scene_txt, const char* buildprocedure_txt, const char* scene_cl,
                    const float* arbitrary_data, size_t n_arbitrary);
/* Arithmetic of the scenes loaded AFTER this call (no reference counterpart; default DCSG_NUMERICS_EXACT).
 * EXACT: IEEE operations in the reference's order, bit-identical to the oracle.  RELAXED (opt-in): hardware square
 * root and reciprocal square root (1 ulp), FMA contraction, float-only normals -- faster, results within ~1e-6 of the
 * exact ones (tests/test_gpu_relaxed.py holds SDF values, normals and exported vertices to 1e-5). */
#define DCSG_NUMERICS_EXACT 0
#define DCSG_NUMERICS_RELAXED 1
int dcsg_set_numerics(dcsg_ctx* ctx, int mode);
/* Evaluator::setArbitraryData (Evaluator.cpp:213-225): copies `items` floats, uploads the whole table. */
int dcsg_set_arbitrary_data(dcsg_ctx* ctx, const float* data, size_t items);
/* Scene facts after loading: object / build-step counts, interpreter registers in use, whether scene.cl
 * was compiled (1) or the HIP-native brush bank applies (0), which interpreter runs the evaluation
 * kernels (0 generic LDS stack, 1 register stack depth <= 4, 2 depth <= 8, 3 per-scene specialised code,
 * 4 register stack with tile culling)
 * and the program's stack depth
 * (0 when it is not stack-disciplined).  Any out pointer may be NULL. */
int dcsg_scene_info(const dcsg_ctx* ctx, int* n_objects, int* n_steps, int* n_slots, int* jit_compiled,
                    int* interpreter, int* stack_depth);

/* ---- point evaluation = kernel k2 (k2.cl:234-280) ---------------------------------- */
/* Evaluator::eval_sdf_at_points (Evaluator.cpp:117-165): xyz = n x 3 floats, out = n floats. */
int dcsg_eval_sdf(dcsg_ctx* ctx, const float* xyz, size_t n, float* out);
/* Evaluator::eval_normal_at_points (Evaluator.cpp:167-211): out_xyz = n x 3 floats. */
int dcsg_eval_normal(dcsg_ctx* ctx, const float* xyz, size_t n, float* out_xyz);
int dcsg_eval_sdf_device(dcsg_ctx* ctx, const float* d_xyz, size_t n, float* d_out, void* stream);
int dcsg_eval_normal_device(dcsg_ctx* ctx, const float* d_xyz, size_t n, float* d_out_xyz, void* stream);

/* ---- dense lattice = ISV3D64 (ISV.hpp:8-138) evaluated everywhere, once ------------- */
/* Fused SDF + central-difference-normal sweep of lattice planes z in [z0,z1) of the (res+1)^3
 * lattice over box{center,diameters} (ISV.hpp:103-109), x fastest.  d_sdf: (z1-z0)*(res+1)^2
 * floats or NULL; d_normals: 3x that (AoS xyz) or NULL. */
int dcsg_sweep_lattice_device(dcsg_ctx* ctx, const float center[3], const float diameters[3], int res, int z0, int z1,
                              float* d_sdf, float* d_normals, void* stream);
/* Same with host outputs (allocates, sweeps, copies back). */
int dcsg_eval_lattice(dcsg_ctx* ctx, const float center[3], const float diameters[3], int res, int z0, int z1,
                      float* sdf, float* normals);

/* ---- export stages (DesignCSG.cpp:638-790) ------------------------------------------ */
/* Stage A (DesignCSG.cpp:668-712): 256^3 probe, cube = max extent, always contains the origin. */
int dcsg_estimate_bounding_box(dcsg_ctx* ctx, float bounding_box_half_diameter, float center[3], float* diameter,
                               uint64_t* interior_points);
/* cms::getIndexTrianglesFromTable (readLookupTable.hpp:32-76): replaces the packaged table. */
int dcsg_set_lookup_table(dcsg_ctx* ctx, const char* lookup_table_txt);
/* cms::Mesh::getSurface (mesh.hpp:82-389) over resident lattice planes z_lo..z_hi (inclusive) held in
 * d_sdf (and d_normals, or NULL to evaluate normals on demand).  Nodes are restricted to the Z-slab
 * [z_own_lo, z_own_hi) (0,res for the whole box).  Triangles (9 floats each: A,B,C xyz) stay in a
 * context-owned device buffer; *n_triangles receives the count.  stats may be NULL. */
int dcsg_mesh_lattice_device(dcsg_ctx* ctx, const float center[3], float diameter, const dcsg_export_config* cfg,
                             const float* d_sdf, const float* d_normals, int z_lo, int z_hi, int z_own_lo,
                             int z_own_hi, uint64_t* n_triangles, dcsg_mesh_stats* stats, void* stream);
/* Opt-in: T-junction repair in the spirit of cms::retopologize (mesh.hpp:432-529) on the context's triangle buffer.
 * The reference's own function cannot be a specification: it is undefined behaviour (by-value parameters captured by
 * reference, :414,:424; it segfaults), termQuotient multiplies (geometry.hpp:71-73), and its de-duplication helper
 * (:445-462) is computed and never used; with only the captures repaired it is the identity on the reference's designs,
 * which is what the default export reproduces.  This stage does what the function is evidently for, built from its own
 * ingredients (cells of Indexer, P = 2^(grid - min) samples per edge, joinDistance, the strip rule): a mesh vertex of
 * another triangle lying on an edge becomes a vertex of this triangle's polygon.  Semantics defined by
 * oracle_retopologize (oracle/dcsg_oracle.c), parity unpinned by construction.  gridLevel - minimumOctreeLevel <= 5. */
int dcsg_retopologize(dcsg_ctx* ctx, const float center[3], float diameter, int min_level, int grid_level, void* stream);
/* Replace the context's triangles with caller-provided ones (host memory, 9 floats each). */
int dcsg_upload_triangles(dcsg_ctx* ctx, const float* triangles, uint64_t n_triangles);
/* cms::performGradientDescent (mesh.hpp:531-593) on the context's triangle buffer, in place. */
int dcsg_project_triangles(dcsg_ctx* ctx, int steps, void* stream);
/* Same on caller-owned device vertices (n_vertices x 3 floats). */
int dcsg_project_vertices_device(dcsg_ctx* ctx, float* d_xyz, size_t n_vertices, int steps, void* stream);
/* Device pointer / host copy of the context's triangle buffer. */
int dcsg_triangles_device(dcsg_ctx* ctx, const float** d_triangles, uint64_t* n_triangles);
int dcsg_download_triangles(dcsg_ctx* ctx, float* out, uint64_t max_triangles);
/* Asynchronous device-to-device copy of the triangle buffer into caller-owned device memory (e.g. a torch
 * tensor that is then gathered over RCCL). */
int dcsg_copy_triangles_device(dcsg_ctx* ctx, float* d_out, uint64_t max_triangles, void* stream);

/* cms::writeTrianglesToSTL (utils.hpp:41-103) / writeTrianglesToPLY (utils.hpp:106-154). */
int dcsg_write_stl(const char* path, const float* triangles, uint64_t n_triangles);
int dcsg_write_ply(const char* path, const float* triangles, uint64_t n_triangles);
/* Byte-range forms for sharded exports (one process per GPU): both formats are fixed-size records once
 * the total count is known, so every rank writes triangles [first, first + n) of a file of `total` triangles at their
 * final offsets.  flags: DCSG_WRITE_BEGIN -- create / open for overwrite, mark the file incomplete and reserve its
 * size (one rank, before any payload; then a barrier); DCSG_WRITE_END -- cut the file to its final length and write
 * the real header (one rank, after a barrier behind all payload).  dcsg_write_stl(path, t, n) is
 * dcsg_write_stl_range(path, t, n, 0, n, BEGIN | END).  A file is valid only after END (crash safety: a failed or
 * interrupted export never leaves a valid header over a stale body). */
#define DCSG_WRITE_BEGIN 1u
#define DCSG_WRITE_END 2u
int dcsg_write_stl_range(const char* path, const float* triangles, uint64_t n, uint64_t first, uint64_t total, unsigned flags);
int dcsg_write_ply_range(const char* path, const float* triangles, uint64_t n, uint64_t first, uint64_t total, unsigned flags);
/* Opt-in writers beyond the reference's (its STL has zero normals, its PLY repeats every vertex per face):
 * STL with facet normals normalize(cross(B-A, C-A)) of the vertices as written; PLY with every distinct
 * position once (float x y z, order of first appearance) and faces indexing them. */
int dcsg_write_stl_normals(const char* path, const float* triangles, uint64_t n_triangles);
int dcsg_write_ply_indexed(const char* path, const float* triangles, uint64_t n_triangles, uint64_t* n_unique_vertices);

/* Kernel k1 (k1.cl:479-578, host side DrawPane.cpp:373-470): the 640 x 480 preview image of the loaded scene by
 * sphere tracing, RGB, 3 bytes per pixel, rows top to bottom, into rgb_out[640*480*3] (host memory).
 * campos / right / up / forward are the four float[3] buffers the reference uploads per frame. */
int dcsg_render_preview(dcsg_ctx* ctx, const float campos[3], const float right[3], const float up[3],
                        const float forward[3], unsigned char* rgb_out);

int dcsg_parse_export_config(const char* export_config_txt, dcsg_export_config* out);
/* MyFrame::OnExportInner (DesignCSG.cpp:638-790) end to end on one GPU: stage A, lattice sweep,
 * mesher, retopologize (identity), projection, both writers.
 * stl_path / ply_path may be NULL to skip a writer.  flags: DCSG_EXPORT_*. */
#define DCSG_EXPORT_DENSE_NORMALS 1u  /* sweep normals for every lattice point instead of on demand */
#define DCSG_EXPORT_SPARSE_NORMALS 2u /* normals on demand: the default (kept for callers of rounds 1-3, when culled scenes swept densely) */
#define DCSG_EXPORT_DENSE_SWEEP 16u   /* sweep the whole lattice even where the node-wise sweep would do */
#define DCSG_EXPORT_INDEXED_PLY 4u    /* ply_path gets dcsg_write_ply_indexed instead of the reference layout */
#define DCSG_EXPORT_STL_NORMALS 8u    /* stl_path gets dcsg_write_stl_normals instead of zero normals */
#define DCSG_EXPORT_EQUAL_SLABS 64u   /* dcsg_export_slab / _multi: equal Z slabs instead of the work-balanced cuts */
#define DCSG_EXPORT_RETOPO 32u        /* run dcsg_retopologize between the mesher and the projection (default: identity);
                                         whole-mesh exports only: DCSG_ERR_INVALID from dcsg_export_slab / _multi with
                                         world > 1 (a slab sees only its own vertices, T-junctions on slab borders would stay) */
int dcsg_export(dcsg_ctx* ctx, const char* export_config_txt, const char* stl_path, const char* ply_path,
                unsigned flags, dcsg_export_stats* stats);

/* Sharded export: Z slab `rank` of `world` (world divides 2^gridLevel), stages A .. projection, no files.
 * The cut planes are WORK-BALANCED by default: stage A counts, per probe plane in z, the probes near the surface (every rank
 * the same integers, no communication), and the N - 1 cuts put an N-th of that count into each slab, on node boundaries of
 * the sweep level; DCSG_EXPORT_EQUAL_SLABS restores equal slabs.  stats->slab_z_lo / _hi report the planes owned.
 * The rank owns the octree nodes whose lowest lattice plane lies in its slab; it evaluates every
 * lattice value its nodes read itself (including the plane shared with the slab above), so no exchange is needed here.
 * One context per GPU: one process per GPU over torch.distributed (designcsg_amd/sharded.py), or one host thread per
 * context in a single process.  The union of the slabs' triangles is the triangle set of dcsg_export. */
int dcsg_export_slab(dcsg_ctx* ctx, const char* export_config_txt, int rank, int world, unsigned flags,
                     dcsg_export_stats* stats);
/* The context's triangles into stl_path / ply_path (either may be NULL) as triangles [first, first + own count) of a
 * file of `total` triangles, through the range writers above (flags: DCSG_WRITE_BEGIN / DCSG_WRITE_END). */
int dcsg_write_slab(dcsg_ctx* ctx, const char* stl_path, const char* ply_path, uint64_t first, uint64_t total, unsigned flags);

/* The sharded export inside one process: ctxs[i] (one context per GPU, created with dcsg_create(&ctx, device_i) and
 * with the same scene loaded) runs Z slab i of n on its own host thread; then every thread writes its byte range of
 * stl_path / ply_path (either may be NULL).  n must divide 2^gridLevel.  stats: n entries (per slab) or NULL.  This is
 * how a single-process host (the reference's is one: DesignCSG.cpp:638-790) uses several GPUs through this boundary;
 * the contexts may also share a device (rehearsal). */
int dcsg_export_multi(dcsg_ctx* const* ctxs, int n, const char* export_config_txt, const char* stl_path, const char* ply_path,
                      unsigned flags, dcsg_export_stats* stats);

/* Diagnostic: host-side timing breakdown, "stage=milliseconds;..." -- which = 0: dcsg_create of this context,
 * 1: its last dcsg_load_scene.  The string lives until the next load / destroy. */
const char* dcsg_load_profile(const dcsg_ctx* ctx, int which);

/* Device time in ms of the most recent TIMED lattice sweep (HIP events on its stream; synchronises): the sweeps of
 * dcsg_eval_lattice and of the exports always are; those of dcsg_sweep_lattice_device only after
 * dcsg_set_sweep_timing(ctx, 1) -- the two event records put several microseconds between back-to-back launches, which an
 * asynchronous caller should not pay by default.  If the most recent sweep was NOT timed the call fails with
 * DCSG_ERR_INVALID (it never reports an older sweep's duration). */
int dcsg_last_sweep_ms(dcsg_ctx* ctx, double* ms);
int dcsg_set_sweep_timing(dcsg_ctx* ctx, int on);

/* ---- host-only diagnostics (no device needed; used by the CPU test-suite) ---------------------- */
/* What dcsg_load_scene makes of a build procedure: the register program after fusion and slot renaming
 * (4 ints per step: opcode,left,right,dest; opcodes 0-5 of k2.cl:12-17 plus 16 IMPORT_MIN, 17 IMPORT_MAX,
 * 18 IMPORT_NEGMAX) and, when the program is stack-disciplined, the stack program (32 PUSH, 33-35 ACC_MIN/
 * MAX/NEGMAX, 36-38 POP_MIN/MAX/NEGMAX, 39 EXPORT_TOP; 4th int = stack depth before the step).
 * cap = capacity of each output array in steps.  Returns DCSG_ERR_INVALID for a malformed scene. */
int dcsg_compile_program(const char* scene_txt, const char* buildprocedure_txt, int32_t* reg_prog, int* n_reg_steps,
                         int* n_regs, int32_t* stack_prog, int* n_stack_steps, int* stack_depth, int cap);
/* lookupTable.txt expanded as the kernels see it: table[case*16] = triangle count, then 3 edge ids per
 * triangle (cms::getIndexTrianglesFromTable + getIndexTriangleStrip). */
int dcsg_expand_lookup_table(const char* lookup_table_txt, uint8_t table[4096], int* total_triangles);
/* Diagnostic: the fold of mesh.hpp:253 `angleBetweenVectors(a,b) > threshold` (geometry.hpp:118-124) into one compare:
 * the largest float q with (float)acos((double)q) > threshold, or -2 if there is none. */
float dcsg_angle_q_star(float complex_surface_threshold);

/* Diagnostic: the flags word of the loaded scene's device tables (csrc/scene.cpp build_blob): bit 0 = the null brush is a
 * constant, bit 1 = the root accumulator is a plain union (the tile culling's order-free rule applies). */
int dcsg_debug_scene_flags(const dcsg_ctx* ctx, int* flags);
/* Diagnostic, no device needed: the work-balanced cut planes of dcsg_export_slab for a given stage-A histogram (256 counts of
 * near-surface probes per probe plane z = half_neg + cd (k - 128)), a box of `diameter` centred at center_z in z, and the
 * export levels: cuts[0 .. world], rank r owning lattice planes [cuts[r], cuts[r + 1]).  Returns DCSG_ERR_INVALID (cuts = the
 * equal slabs) when the equal slabs are kept. */
int dcsg_debug_slab_cuts(const uint32_t hist[256], float cd, float half_neg, float center_z, float diameter, int min_level, int grid_level,
                         int world, int* cuts);
/* Diagnostic: the cell lists of the loaded many-object scene (vertex projection, DESIGN.md "Cell lists"): cells per axis
 * (0 = no grid yet), cells built (fine, coarse and the pseudo-cell of the whole program) and list entries handed out so far,
 * host milliseconds spent building cells since the scene was loaded. */
int dcsg_debug_cells(const dcsg_ctx* ctx, int* grid, uint64_t* cells_built, uint64_t* entries, double* build_ms);
/* Diagnostic: the kernels' correctly rounded square root over a device array (exhaustive GPU test). */
int dcsg_debug_sqrt_device(dcsg_ctx* ctx, const float* d_in, size_t n, float* d_out, void* stream);
/* Diagnostic: the tail of get_normal (k2.cl:175-177: gradient scale in double, normalize) as the kernels compute it,
 * d_out[3i..3i+2] = normal of the gradient (d_in[i], dy, dz); mixed bit 0: odd elements take (1, 2, 3) instead, so a
 * wave holds ordinary and extreme operands together (exhaustive GPU test of the float-only scale). */
int dcsg_debug_normal_device(dcsg_ctx* ctx, const float* d_in, size_t n, float dy, float dz, int mixed, float* d_out, void* stream);

/* Diagnostic, no device needed: the host-side sequence of a cold dcsg_load_scene (export part compiled on the calling
 * thread, rest part by the background job) on a context that never touches HIP, kept in *slot between calls.  mode 0: wait
 * for the background build; 1: return with it still running (what an early error return leaves behind); 2: destroy the
 * slot's context.  For the thread-sanitizer harness (tools/sanitize_host.sh --thread) and tests/test_host_threads.py. */
int dcsg_debug_background_build(void** slot, const char* scene_txt, const char* buildprocedure_txt, const char* scene_cl, int mode,
                                char* log_out, size_t log_cap);

/* The compile half of Evaluator::build without a device: compiles scene_cl into the on-disk code
 * object cache (used at install time so that known scenes load without compiling).  The compiler
 * log is copied to log_out; *from_cache = 1 when the code object already existed. */
int dcsg_jit_prebuild(const char* scene_cl, char* log_out, size_t log_cap, int* from_cache);
/* Same for a whole scene: compiles exactly the code objects dcsg_load_scene would ask for (both parts: brush text plus
 * the specialisation of its build procedure), if it needs any. */
int dcsg_jit_prebuild_scene(const char* scene_txt, const char* buildprocedure_txt, const char* scene_cl, char* log_out,
                            size_t log_cap, int* from_cache);

/* ... with the arithmetic of dcsg_set_numerics. */
int dcsg_jit_prebuild_scene_numerics(const char* scene_txt, const char* buildprocedure_txt, const char* scene_cl,
                                     int numerics, char* log_out, size_t log_cap, int* from_cache);

#ifdef __cplusplus
}
#endif
#endif
