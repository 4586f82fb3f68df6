// dcsg_device_params.h -- kernel parameter blocks shared by kernels.hip and the host side.
// Plain C layout; passed by value through hipModuleLaunchKernel.
#ifndef DCSG_DEVICE_PARAMS_H
#define DCSG_DEVICE_PARAMS_H

#define DCSG_BLOCK 256
#define DCSG_BBOX_WORDS 272   // stage A's result: [0..5] min / max indices, [8..9] interior count (u64), [16..271] probes near the surface per z plane
#define DCSG_BBOX_COPIES 16   // ... kept in that many copies (kernels.hip, dcsg_bbox_commit)
#define DCSG_WAVE 64
#define DCSG_MAX_SLOTS 64  /* STACK_MEMORY_PER_PIXEL, Evaluator.cpp:7 */
/* dcsg_sweep* `flags`: bit 0 = the lattice holds no -0.0 coordinate (checked by the host), so a sweep of per-scene code
 * may share the unperturbed stencil coordinates between the seven points without looking (dcsg_stencil_spec) */
#define DCSG_SWEEP_NO_NEG_ZERO 1u
/* bit 1 = the stack interpreters may serve several 64-point passes with one interpretation of the program */
#define DCSG_SWEEP_MULTIPASS 2u

struct DcsgLattice {  // ISV3D64 geometry, ISV.hpp:91-109
    float base[3];    // center - 0.5f*diameters   (ISV.hpp:104)
    float diam[3];    // bb.diameters
    float center[3];  // bb.center
    float half[3];    // bb.diameters / 2.0f       (ISV.hpp:92)
    float w;          // (float)res
    float inv_w;      // 1/w, exact when res is a power of two
    int res;          // lattice has (res+1)^3 points
    int w_pow2;       // 1 -> x/w == x*inv_w bit-for-bit
    unsigned int n1_magic, n1_shift;  // n/(res+1) = (((n - mulhi(n,magic)) >> 1) + mulhi(n,magic)) >> shift
};

struct DcsgMeshParams {
    DcsgLattice lat;
    float hd[3];       // node half diameter at this level
    int level, min_level, max_level, grid_level;
    float q_star;      // angle test: acos(q) > threshold  <=>  -1 <= q <= q_star
    int z_lo, z_hi;    // lattice planes resident in sdf[] / nrm[]: z_lo..z_hi inclusive
    int has_normals;   // nrm[] resident; otherwise normals are evaluated on demand
    int z_own_lo, z_own_hi;  // Z-slab owned by this rank: nodes that do not intersect it are dropped,
                             // leaves emit triangles only where their lowest plane is in [z_own_lo, z_own_hi)
    // Node-wise sweep (dcsg_export): the lattice holds values only inside the octree nodes of level `mask_level`
    // that reached the mesher; node_mask[(nz*N + ny)*N + nx] != 0 (N = 2^mask_level) marks them, node (nx,ny,nz)
    // covering lattice indices [n << mask_shift, (n+1) << mask_shift] per axis.  NULL = the lattice is dense.
    const unsigned char* node_mask;
    int mask_level, mask_shift;
    int use_spec;  // on-demand samples through the module's per-scene specialised code (spec kernel set only)
    int pattern_step, pattern_edges;  // node-wise lattice: which points of a swept node were written (kernels.hip, dcsg_in_sweep_pattern); step 1 = all
    int mask_check;  // diagnostic (DCSG_MESH_MASK_ALWAYS): every sample asks the node mask, also inside its own swept node
};

struct DcsgRetopo {  // T-junction repair in the spirit of cms::retopologize (mesh.hpp:432-529; oracle_retopologize)
    float c[3], hd[3];   // bounds.center, bounds.halfDiameter
    float grid_points;   // (float)((1 << gridLevel) + 1)
    int points_along_edge;  // 1 << (gridLevel - minimumOctreeLevel), <= 32
    float join_distance2;   // joinDistance^2, joinDistance = 0.25 * min_k(2 hd_k / grid_points) (mesh.hpp:445)
};

// Cell lists of many-object scenes (kernels.hip, "Cell lists"): a cube of g^3 cells of side `cell` with its low corner at
// (x0, y0, z0); cells[c] = {first entry, entry count} of cell c = (iz * g + iy) * g + ix, cells[g^3] = the whole program
// (points outside the cube).
struct DcsgCellGrid {
    float x0, y0, z0;
    float cell, inv_cell;
    int g;
};

// What a kernel outside the cell-list section needs to read them (the mesher's on-demand samples)
struct DcsgCellLists {
    DcsgCellGrid G;
    const void* cells;            // uint2 per cell: {first entry, count}; NULL: the scene has no cell lists
    const unsigned int* entries;
};

struct DcsgCamera {  // kernel k1's campos / right / up / forward buffers (k1.cl:481-486)
    float o[3], r[3], u[3], f[3];
};

#endif
