/*
 * dcsg_oracle.c -- CPU restatement of the DesignCSG export path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see dcsg_oracle.h).  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fopenmp).  All arithmetic is IEEE binary32 /
 * binary64 with no contraction and no reassociation, in the operation order
 * of the reference expressions; citations are file:line under
 * /root/reference/master.
 *
 * OpenCL C built-ins the reference calls (dot, length, normalize, fabs, max,
 * sqrt -- k2.cl:111,177 and the generated brush bodies) come from whatever
 * OpenCL runtime the user has; the reference pins nothing at the ulp level
 * ("parity unpinned" for the last bit).  This oracle fixes them as
 * correctly-rounded, left-to-right compositions (same choice as the survey's
 * probe); the HIP kernels use the same definitions.
 */
#include "dcsg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* k2.cl:1-17 constants                                                */
#define K2_MAX_DISTANCE 64.0
#define K2_NORMAL_EPSILON 0.005
#define OP_IMPORT 0
#define OP_EXPORT 1
#define OP_MIN 2
#define OP_MAX 3
#define OP_NEGATE 4
#define OP_IDENTITY 5
/* EXTENSION opcodes (not in the reference: scenecompiler.py:25-32 stops at IDENTITY, k2.cl's switch ignores
 * anything else).  BASELINE.json's fifth configuration asks for "smooth-min unions"; they are defined HERE, and
 * parity for them is GPU vs this restatement only (no reference output exists):
 *   SET_K  bits          k = as_float(bits) (finite, > 0), inv_k = 1.0f / k
 *   SMOOTH_MIN a b -> d  d = T_min(a,b) - ((h*h)*k)*0.25f,  h = T_max(k - fabs(a-b), 0) * inv_k
 *   SMOOTH_MAX a b -> d  d = T_max(a,b) + ((h*h)*k)*0.25f
 * all in float, left to right, no contraction (the polynomial smooth minimum with blend radius k).       */
#define OP_SMOOTH_MIN 6
#define OP_SMOOTH_MAX 7
#define OP_SET_K 8
#define STACK_MEMORY_PER_PIXEL 64 /* Evaluator.cpp:7 */

/* OpenCL dot(float3,float3): x*x' + y*y' + z*z', left to right */
static inline float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return ax * bx + ay * by + az * bz;
}
static inline float length3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }
static inline float length2(float x, float y) { return sqrtf(x * x + y * y); }
/* OpenCL max(gentype x, gentype y): y if x < y, otherwise x */
static inline double cl_max_d(double x, double y) { return x < y ? y : x; }

/* ------------------------------------------------------------------ */
/* Built-in brushes.  Unsuffixed literals are double in OpenCL C, so the
 * subtraction happens in double and narrows at `return`.                */
float oracle_builtin_bank(float x, float y, float z, int shape_id)
{
    switch ((unsigned char)shape_id) {
    case 0: /* "return MAX_DISTANCE;"  scenecompiler.py:424 */
        return (float)K2_MAX_DISTANCE;
    case 1: /* "return 0.0;"           scenecompiler.py:425 */
        return (float)0.0;
    case 2: /* "return length(v)-0.5;" DesignCSG.py:9 */
        return (float)((double)length3(x, y, z) - 0.5);
    case 3: { /* cylinder, DesignCSG.py:10-17 */
        float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
        float r = length2(ax, az);
        float h = ay;
        return (float)cl_max_d((double)r - 0.5, (double)h - 0.5);
    }
    case 4: { /* box, DesignCSG.py:19-22 */
        float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
        return (float)cl_max_d((double)ax - 0.5, cl_max_d((double)ay - 0.5, (double)az - 0.5));
    }
    default: /* sdf_bank falls through to "return 0.0;" scenecompiler.py:497 */
        return (float)0.0;
    }
}

/* ------------------------------------------------------------------ */
/* Scene parsing: DrawPane.cpp:267-371 (whitespace-separated %d %d 12x%f per
 * scene.txt line; "%d %d %d %d" per buildprocedure.txt line).            */
static int count_lines(const char *t)
{
    int n = 0;
    const char *p = t;
    while (*p) {
        const char *e = strchr(p, '\n');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        int blank = 1;
        for (size_t i = 0; i < len; i++)
            if (p[i] != ' ' && p[i] != '\r' && p[i] != '\t')
                blank = 0;
        if (!blank)
            n++;
        if (!e)
            break;
        p = e + 1;
    }
    return n;
}

oracle_scene *oracle_scene_parse(const char *scene_txt, const char *buildprocedure_txt)
{
    oracle_scene *s = (oracle_scene *)calloc(1, sizeof(*s));
    int cap_o = count_lines(scene_txt), cap_s = count_lines(buildprocedure_txt);
    s->shape_id = (unsigned char *)calloc(cap_o + 1, 1);
    s->material_id = (unsigned char *)calloc(cap_o + 1, 1);
    s->pos = (float *)calloc(3 * (cap_o + 1), sizeof(float));
    s->right = (float *)calloc(3 * (cap_o + 1), sizeof(float));
    s->up = (float *)calloc(3 * (cap_o + 1), sizeof(float));
    s->fwd = (float *)calloc(3 * (cap_o + 1), sizeof(float));
    s->steps = (int *)calloc(4 * (cap_s + 1), sizeof(int));
    s->bank = oracle_builtin_bank;

    const char *p = scene_txt;
    while (*p && s->n_objects < cap_o) {
        char *end;
        long brush = strtol(p, &end, 10);
        if (end == p)
            break;
        p = end;
        long material = strtol(p, &end, 10); /* material id: used by the preview only */
        p = end;
        float v[12];
        for (int k = 0; k < 12; k++) {
            v[k] = strtof(p, &end);
            p = end;
        }
        int i = s->n_objects++;
        s->shape_id[i] = (unsigned char)brush;
        s->material_id[i] = (unsigned char)material;
        for (int k = 0; k < 3; k++) {
            s->pos[i * 3 + k] = v[k];
            s->right[i * 3 + k] = v[3 + k];
            s->up[i * 3 + k] = v[6 + k];
            s->fwd[i * 3 + k] = v[9 + k];
        }
        while (*p == ' ' || *p == '\r' || *p == '\n' || *p == '\t')
            p++;
    }
    p = buildprocedure_txt;
    while (*p && s->n_steps < cap_s) {
        char *end;
        int ok = 1;
        int v[4];
        for (int k = 0; k < 4; k++) {
            v[k] = (int)strtol(p, &end, 10);
            if (end == p)
                ok = 0;
            p = end;
        }
        if (!ok)
            break;
        memcpy(&s->steps[4 * s->n_steps++], v, sizeof(v));
        while (*p == ' ' || *p == '\r' || *p == '\n' || *p == '\t')
            p++;
    }
    return s;
}

void oracle_scene_free(oracle_scene *s)
{
    if (!s)
        return;
    free(s->shape_id);
    free(s->material_id);
    free(s->pos);
    free(s->right);
    free(s->up);
    free(s->fwd);
    free(s->steps);
    free(s);
}

void oracle_scene_set_bank(oracle_scene *s, oracle_bank_fn bank) { s->bank = bank ? bank : oracle_builtin_bank; }

void oracle_scene_set_shader(oracle_scene *s,
                             void (*shader)(const float *, const float *, const float *, int, float *),
                             void (*set_camera)(const float *, const float *, const float *))
{
    s->shader = shader;
    s->set_camera = set_camera;
}

/* exportConfig.txt: 9 lines, std::stof / std::stoi positional, DesignCSG.cpp:827-835 */
int oracle_parse_export_config(const char *text, oracle_export_config *c)
{
    char *end;
    const char *p = text;
    c->bounding_box_half_diameter = strtof(p, &end);
    if (end == p)
        return -1;
    p = end;
    c->min_level = (int)strtol(p, &end, 10);
    p = end;
    c->max_level = (int)strtol(p, &end, 10);
    p = end;
    c->grid_level = (int)strtol(p, &end, 10);
    p = end;
    c->complex_surface_threshold = strtof(p, &end);
    p = end;
    c->gradient_descent_steps = (int)strtol(p, &end, 10);
    p = end;
    c->cache_subdivision = (int)strtol(p, &end, 10);
    p = end;
    c->queries_before_gc = (int)strtol(p, &end, 10);
    p = end;
    c->queries_before_free = (int)strtol(p, &end, 10);
    if (end == p)
        return -1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* primary_sdf, k2.cl:47-144                                           */
float oracle_primary_sdf(const oracle_scene *s, float x, float y, float z)
{
    float stack[STACK_MEMORY_PER_PIXEL];
    memset(stack, 0, sizeof(stack));
    float export_value = (float)K2_MAX_DISTANCE; /* k2.cl:91 */
    float k = 0.0f, inv_k = 0.0f;                /* extension state, see OP_SET_K */
    for (int c = 0; c < s->n_steps; c++) {
        int op = s->steps[c * 4 + 0];
        int left = s->steps[c * 4 + 1];
        int right = s->steps[c * 4 + 2];
        int dest = s->steps[c * 4 + 3];
        int i = right;
        switch (op) {
        case OP_IMPORT: { /* k2.cl:104-113 */
            float dx = x - s->pos[i * 3 + 0], dy = y - s->pos[i * 3 + 1], dz = z - s->pos[i * 3 + 2];
            float A = dot3(dx, dy, dz, s->right[i * 3 + 0], s->right[i * 3 + 1], s->right[i * 3 + 2]);
            float B = dot3(dx, dy, dz, s->up[i * 3 + 0], s->up[i * 3 + 1], s->up[i * 3 + 2]);
            float C = dot3(dx, dy, dz, s->fwd[i * 3 + 0], s->fwd[i * 3 + 1], s->fwd[i * 3 + 2]);
            stack[dest] = s->bank(A, B, C, (unsigned char)left);
        } break;
        case OP_EXPORT: /* k2.cl:116-118 */
            export_value = stack[left];
            break;
        case OP_MIN: /* T_min(a,b) (a<b?a:b), k2.cl:29,120-122 */
            stack[dest] = stack[left] < stack[right] ? stack[left] : stack[right];
            break;
        case OP_MAX: /* T_max(a,b) (a>b?a:b), k2.cl:30,124-126 */
            stack[dest] = stack[left] > stack[right] ? stack[left] : stack[right];
            break;
        case OP_NEGATE: /* k2.cl:128-130 */
            stack[dest] = -stack[left];
            break;
        case OP_IDENTITY: /* k2.cl:132-134 */
            stack[dest] = stack[left];
            break;
        case OP_SET_K: { /* extension */
            int32_t bits = (int32_t)left;
            memcpy(&k, &bits, 4);
            inv_k = 1.0f / k;
        } break;
        case OP_SMOOTH_MIN:
        case OP_SMOOTH_MAX: { /* extension */
            float a = stack[left], b = stack[right];
            float t = k - fabsf(a - b);
            t = t > 0.0f ? t : 0.0f;
            float h = t * inv_k;
            float q = ((h * h) * k) * 0.25f;
            if (op == OP_SMOOTH_MIN)
                stack[dest] = (a < b ? a : b) - q;
            else
                stack[dest] = (a > b ? a : b) + q;
        } break;
        default:
            break;
        }
    }
    return export_value;
}

/* OpenCL normalize(float3): v / length(v); the zero vector stays zero
 * (platform-dependent in OpenCL).          */
static inline void normalize3(float v[3])
{
    float len = length3(v[0], v[1], v[2]);
    if (len == 0.0f) {
        v[0] = v[1] = v[2] = 0.0f;
        return;
    }
    v[0] = v[0] / len;
    v[1] = v[1] / len;
    v[2] = v[2] / len;
}

/* get_normal, k2.cl:149-179 */
void oracle_get_normal(const oracle_scene *s, float x, float y, float z, float out[3])
{
    const float e = (float)K2_NORMAL_EPSILON; /* (float3)(NORMAL_EPSILON,0.0,0.0) */
    const float zero = (float)0.0;
    float Dx = oracle_primary_sdf(s, x + e, y + zero, z + zero) - oracle_primary_sdf(s, x - e, y - zero, z - zero);
    float Dy = oracle_primary_sdf(s, x + zero, y + e, z + zero) - oracle_primary_sdf(s, x - zero, y - e, z - zero);
    float Dz = oracle_primary_sdf(s, x + zero, y + zero, z + e) - oracle_primary_sdf(s, x - zero, y - zero, z - e);
    float twoE = (float)(2.0 * K2_NORMAL_EPSILON);           /* k2.cl:175 */
    out[0] = (float)(1.0 / (double)twoE * (double)Dx);       /* k2.cl:177: double product, narrowed */
    out[1] = (float)(1.0 / (double)twoE * (double)Dy);
    out[2] = (float)(1.0 / (double)twoE * (double)Dz);
    normalize3(out);
}

static int pick_threads(int threads)
{
#ifdef _OPENMP
    if (threads <= 0)
        threads = omp_get_max_threads();
    return threads;
#else
    (void)threads;
    return 1;
#endif
}

/* kernel k2 in SDF mode over a point batch (k2.cl:234-280, Evaluator.cpp:117-165) */
void oracle_eval_sdf(const oracle_scene *s, const float *xyz, size_t n, float *out, int threads)
{
    threads = pick_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++)
        out[i] = oracle_primary_sdf(s, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}

/* kernel k2 in NORMAL mode (Evaluator.cpp:167-211) */
void oracle_eval_normal(const oracle_scene *s, const float *xyz, size_t n, float *out, int threads)
{
    threads = pick_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++)
        oracle_get_normal(s, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &out[3 * i]);
}

/* ------------------------------------------------------------------ */
/* ISV3D64::getPoint, ISV.hpp:103-109 (w = h = d = res)                 */
void oracle_lattice_point(const oracle_box *bb, int res, int64_t ix, int64_t iy, int64_t iz, float out[3])
{
    float w = (float)res;
    out[0] = (bb->center[0] - 0.5f * bb->diameters[0]) + bb->diameters[0] * (float)ix / w;
    out[1] = (bb->center[1] - 0.5f * bb->diameters[1]) + bb->diameters[1] * (float)iy / w;
    out[2] = (bb->center[2] - 0.5f * bb->diameters[2]) + bb->diameters[2] * (float)iz / w;
}

/* ISV3D64::getCoords, ISV.hpp:91-96: float arithmetic, truncation toward zero */
void oracle_lattice_coords(const oracle_box *bb, int res, const float p[3], int64_t out[3])
{
    float w = (float)(int64_t)res;
    for (int k = 0; k < 3; k++)
        out[k] = (int64_t)(w * (p[k] - bb->center[k] + bb->diameters[k] / 2.0f) / bb->diameters[k]);
}

void oracle_eval_lattice(const oracle_scene *s, const oracle_box *bb, int res, int z0, int z1, float *sdf,
                         float *normals, int threads)
{
    threads = pick_threads(threads);
    const int64_t n1 = (int64_t)res + 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) collapse(2)
    for (int64_t iz = z0; iz < z1; iz++)
        for (int64_t iy = 0; iy < n1; iy++)
            for (int64_t ix = 0; ix < n1; ix++) {
                float p[3];
                oracle_lattice_point(bb, res, ix, iy, iz, p);
                int64_t o = ((iz - z0) * n1 + iy) * n1 + ix;
                if (sdf)
                    sdf[o] = oracle_primary_sdf(s, p[0], p[1], p[2]);
                if (normals)
                    oracle_get_normal(s, p[0], p[1], p[2], &normals[3 * o]);
            }
}

/* ------------------------------------------------------------------ */
/* Stage A, DesignCSG.cpp:668-712                                       */
void oracle_estimate_bounding_box(const oracle_scene *s, float bb_half_diameter, oracle_box *out,
                                  uint64_t *interior_points, int threads)
{
    threads = pick_threads(threads);
    const int R = 256;                                   /* BOUNDING_BOX_RESOLUTION */
    const float cd = (float)bb_half_diameter / R;        /* cube_diameter, :673 */
    const double bb_eps = (double)((float)bb_half_diameter / R) * 1.0; /* BB_EPSILON, :670 */
    const float half = -cd / 2;
    float mn[3] = {0.0f, 0.0f, 0.0f}, mx[3] = {0.0f, 0.0f, 0.0f}; /* :688-693 */
    uint64_t interior = 0;
#pragma omp parallel num_threads(threads)
    {
        float lmn[3] = {0.0f, 0.0f, 0.0f}, lmx[3] = {0.0f, 0.0f, 0.0f};
        uint64_t lint = 0;
#pragma omp for schedule(dynamic, 1) nowait
        for (int ix = -R / 2; ix < R / 2; ix++)
            for (int iy = -R / 2; iy < R / 2; iy++)
                for (int iz = -R / 2; iz < R / 2; iz++) {
                    float p[3] = {half + cd * (float)ix, half + cd * (float)iy, half + cd * (float)iz}; /* :679 */
                    float v = oracle_primary_sdf(s, p[0], p[1], p[2]);
                    if ((double)v < bb_eps) { /* :687 */
                        lint++;
                        for (int k = 0; k < 3; k++) {
                            if (p[k] < lmn[k])
                                lmn[k] = p[k];
                            if (p[k] > lmx[k])
                                lmx[k] = p[k];
                        }
                    }
                }
#pragma omp critical
        {
            interior += lint;
            for (int k = 0; k < 3; k++) {
                if (lmn[k] < mn[k])
                    mn[k] = lmn[k];
                if (lmx[k] > mx[k])
                    mx[k] = lmx[k];
            }
        }
    }
    float d[3];
    for (int k = 0; k < 3; k++) {
        out->center[k] = (float)((double)(mn[k] + mx[k]) * 0.5); /* :703 */
        d[k] = mx[k] - mn[k];                                    /* :704 */
    }
    float inner = d[1] > d[2] ? d[1] : d[2];  /* T_max, :706 */
    float mbx = d[0] > inner ? d[0] : inner;
    out->diameters[0] = out->diameters[1] = out->diameters[2] = mbx; /* :707 */
    if (interior_points)
        *interior_points = interior;
}

/* ------------------------------------------------------------------ */
/* lookupTable.txt -> index triangles                                   */
/* getIndexTriangleStrip, geometry.hpp:228-248 */
static int triangle_strip(const int *loop, int n, int out[][3])
{
    int cnt = 0;
    int start = 0;
    if (n % 2 == 1) {
        out[cnt][0] = loop[0];
        out[cnt][1] = loop[1];
        out[cnt][2] = loop[n - 1];
        cnt++;
        start = 1; /* pop_front */
    }
    const int *dq = loop + start;
    int m = n - start;
    for (int A = 0; A < m / 2 - 1; A++) {
        int B = A + 1;
        int D = m - 1 - A;
        int C = D - 1;
        out[cnt][0] = dq[A];
        out[cnt][1] = dq[B];
        out[cnt][2] = dq[C];
        cnt++;
        out[cnt][0] = dq[C];
        out[cnt][1] = dq[D];
        out[cnt][2] = dq[A];
        cnt++;
    }
    return cnt;
}

/* getIndexTrianglesFromTable, readLookupTable.hpp:32-76: lines split on '\n'
 * (empty lines skipped by the tokenizer), a line equal to "\t" is the empty
 * case, loops split on ';', indices on ','.                              */
int oracle_parse_lookup_table(const char *text, int tri_count[256], int tris[256][5][3])
{
    int line = 0, total = 0;
    const char *p = text;
    memset(tri_count, 0, 256 * sizeof(int));
    while (*p && line < 256) {
        const char *e = strchr(p, '\n');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == 0) { /* find_first_not_of skips empty tokens */
            if (!e)
                break;
            p = e + 1;
            continue;
        }
        if (!(len == 1 && p[0] == '\t')) {
            size_t i = 0;
            while (i < len) {
                int loop[16], n = 0;
                while (i < len && p[i] != ';') {
                    char *end;
                    long v = strtol(p + i, &end, 10);
                    if (end == p + i) {
                        i++;
                        continue;
                    }
                    if (n < 16)
                        loop[n++] = (int)v;
                    i = (size_t)(end - p);
                    if (i < len && p[i] == ',')
                        i++;
                }
                if (i < len && p[i] == ';')
                    i++;
                if (n >= 3) {
                    int st[8][3];
                    int c = triangle_strip(loop, n, st);
                    for (int k = 0; k < c; k++) {
                        if (tri_count[line] >= 5)
                            return -1;
                        memcpy(tris[line][tri_count[line]++], st[k], 3 * sizeof(int));
                        total++;
                    }
                }
            }
        }
        line++;
        if (!e)
            break;
        p = e + 1;
    }
    return line == 256 ? total : -1;
}

/* ------------------------------------------------------------------ */
/* Lattice sample cache.  The reference's ISV (ISV.hpp:15-60) evaluates whole
 * blocks lazily and returns the value AT THE SNAPPED LATTICE POINT; a sample
 * is a pure function of its lattice index, so any caching granularity gives
 * the same values.  Here: per-point lazy evaluation in direct-address arrays
 * (calloc'd, committed on touch), shared by all threads.                  */
typedef struct lattice_cache {
    const oracle_scene *scene;
    oracle_box bb;
    int res;
    int64_t n1;
    float *sdf;          /* [(res+1)^3] */
    uint32_t *sdf_done;  /* bitmap */
    float *nrm;          /* [3*(res+1)^3], allocated on first use */
    uint32_t *nrm_done;
    uint64_t n_sdf, n_nrm;
} lattice_cache;

static int cache_init(lattice_cache *c, const oracle_scene *s, const oracle_box *bb, int res)
{
    memset(c, 0, sizeof(*c));
    c->scene = s;
    c->bb = *bb;
    c->res = res;
    c->n1 = (int64_t)res + 1;
    size_t n = (size_t)(c->n1 * c->n1 * c->n1);
    c->sdf = (float *)calloc(n, sizeof(float));
    c->sdf_done = (uint32_t *)calloc((n + 31) / 32, sizeof(uint32_t));
    c->nrm = (float *)calloc(3 * n, sizeof(float));
    c->nrm_done = (uint32_t *)calloc((n + 31) / 32, sizeof(uint32_t));
    return (c->sdf && c->sdf_done && c->nrm && c->nrm_done) ? 0 : -1;
}

static void cache_free(lattice_cache *c)
{
    free(c->sdf);
    free(c->sdf_done);
    free(c->nrm);
    free(c->nrm_done);
}

static inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline int64_t cache_index(const lattice_cache *c, const float p[3])
{
    int64_t ic[3];
    oracle_lattice_coords(&c->bb, c->res, p, ic);
    /* octree points lie inside the box, so 0..res; clamp guards the arrays */
    ic[0] = clampi(ic[0], 0, c->res);
    ic[1] = clampi(ic[1], 0, c->res);
    ic[2] = clampi(ic[2], 0, c->res);
    return (ic[2] * c->n1 + ic[1]) * c->n1 + ic[0];
}

static float cache_sdf(lattice_cache *c, const float p[3])
{
    int64_t idx = cache_index(c, p);
    uint32_t bit = 1u << (idx & 31);
    if (__atomic_load_n(&c->sdf_done[idx >> 5], __ATOMIC_ACQUIRE) & bit)
        return c->sdf[idx]; /* written before the releasing fetch_or below */
    int64_t iz = idx / (c->n1 * c->n1), iy = (idx / c->n1) % c->n1, ix = idx % c->n1;
    float q[3];
    oracle_lattice_point(&c->bb, c->res, ix, iy, iz, q);
    float v = oracle_primary_sdf(c->scene, q[0], q[1], q[2]);
    c->sdf[idx] = v;
    if (!(__atomic_fetch_or(&c->sdf_done[idx >> 5], bit, __ATOMIC_RELEASE) & bit))
        __atomic_fetch_add(&c->n_sdf, 1, __ATOMIC_RELAXED);
    return v;
}

static void cache_normal(lattice_cache *c, const float p[3], float out[3])
{
    int64_t idx = cache_index(c, p);
    uint32_t bit = 1u << (idx & 31);
    if (__atomic_load_n(&c->nrm_done[idx >> 5], __ATOMIC_ACQUIRE) & bit) {
        memcpy(out, &c->nrm[3 * idx], 3 * sizeof(float));
        return;
    }
    int64_t iz = idx / (c->n1 * c->n1), iy = (idx / c->n1) % c->n1, ix = idx % c->n1;
    float q[3];
    oracle_lattice_point(&c->bb, c->res, ix, iy, iz, q);
    oracle_get_normal(c->scene, q[0], q[1], q[2], out);
    memcpy(&c->nrm[3 * idx], out, 3 * sizeof(float));
    if (!(__atomic_fetch_or(&c->nrm_done[idx >> 5], bit, __ATOMIC_RELEASE) & bit))
        __atomic_fetch_add(&c->n_nrm, 1, __ATOMIC_RELAXED);
}

/* ------------------------------------------------------------------ */
/* Octree mesher: Mesh::getSurface, mesh.hpp:82-389                     */
typedef struct node {
    int level;
    float c[3];  /* bounds.center       */
    float hd[3]; /* bounds.halfDiameter */
} node;

/* Box3f::getCorners(s), geometry.hpp:264-279: center + ((hd*sign)*s) */
static const float CORNER_SIGN[8][3] = {{-1, -1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1},
                                        {-1, 1, 1},  {1, 1, 1},  {1, 1, -1},  {-1, 1, -1}};

static void get_corners(const node *nd, float s, float out[8][3])
{
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 3; k++)
            out[i][k] = nd->c[k] + (s * (nd->hd[k] * CORNER_SIGN[i][k]));
}

/* Node::subdivideIntoQueue, octree.hpp:24-32 */
static void subdivide(const node *nd, node out[8])
{
    float corners[8][3];
    get_corners(nd, 0.5f, corners);
    for (int i = 0; i < 8; i++) {
        out[i].level = nd->level + 1;
        for (int k = 0; k < 3; k++) {
            out[i].c[k] = corners[i][k];
            out[i].hd[k] = 0.5f * nd->hd[k];
        }
    }
}

typedef struct tri_vec {
    float *v;
    size_t n, cap; /* in triangles */
} tri_vec;

static void tri_push(tri_vec *t, const float a[3], const float b[3], const float c[3])
{
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->v = (float *)realloc(t->v, t->cap * 9 * sizeof(float));
    }
    float *o = &t->v[t->n++ * 9];
    memcpy(o, a, 12);
    memcpy(o + 3, b, 12);
    memcpy(o + 6, c, 12);
}

/* Vector3f::angleBetweenVectors, geometry.hpp:118-124.  acosf is evaluated as
 * (float)acos((double)q): identical to a correctly rounded acosf except on
 * double-rounding ties; chosen so host libm and device libm agree
 * (SURVEY.md appendix B item 17).                                        */
static float angle_between(const float a[3], const float b[3], float tolerance)
{
    float ma = sqrtf(dot3(a[0], a[1], a[2], a[0], a[1], a[2]));
    float mb = sqrtf(dot3(b[0], b[1], b[2], b[0], b[1], b[2]));
    if (ma * mb < tolerance)
        return 0.0f;
    float q = dot3(a[0], a[1], a[2], b[0], b[1], b[2]) / (ma * mb);
    return (float)acos((double)q);
}

static void process_subtree(const node *root, lattice_cache *cache, const oracle_export_config *cfg,
                            const int tri_count[256], const int tris[256][5][3], tri_vec *mesh,
                            oracle_mesh_stats *st)
{
    size_t cap = 1024, head = 0, tail = 0;
    node *q = (node *)malloc(cap * sizeof(node));
    q[tail++] = *root;
    while (head < tail) {
        node nd = q[head++];
        st->nodes_visited[nd.level]++;

        /* mesh.hpp:167-170 */
        float d = cache_sdf(cache, nd.c);
        float mag = sqrtf(dot3(nd.hd[0], nd.hd[1], nd.hd[2], nd.hd[0], nd.hd[1], nd.hd[2]));
        if (fabsf(d) > mag * 1.1f) {
            st->nodes_culled[nd.level]++;
            continue;
        }

        /* mesh.hpp:176-183 */
        float corner[8][3];
        get_corners(&nd, 1.0f, corner);
        int lookup = 0;
        for (int i = 0; i < 8; i++)
            lookup |= (cache_sdf(cache, corner[i]) < 0.0f ? 1 : 0) << i;

        /* mesh.hpp:185-209: 12 edges, vertex = midpoint = 0.5*A + 0.5*B (geometry.hpp:91-93) */
        int e0[12], e1[12];
        for (int i = 0; i < 4; i++) {
            e0[i] = i;
            e1[i] = (i + 1) % 4;
            e0[4 + i] = i + 4;
            e1[4 + i] = (i + 1) % 4 + 4;
            e0[8 + i] = i;
            e1[8 + i] = i + 4;
        }
        float mid[12][3];
        for (int e = 0; e < 12; e++)
            for (int k = 0; k < 3; k++)
                mid[e][k] = 0.5f * corner[e0[e]][k] + 0.5f * corner[e1[e]][k];

        int should_subdivide = 0;
        if (nd.level < cfg->min_level) { /* mesh.hpp:214-217 */
            should_subdivide = 1;
        } else {
            /* edge ambiguity, mesh.hpp:221-238 */
            int points_along_edge = 1 << (cfg->grid_level - nd.level);
            for (int e = 0; e < 12 && !should_subdivide; e++) {
                const float *start = corner[e0[e]], *end = corner[e1[e]];
                float delta[3] = {end[0] - start[0], end[1] - start[1], end[2] - start[2]};
                for (int i = 1; i < points_along_edge; i++) {
                    float f = (float)i / (float)points_along_edge;
                    float tp[3] = {start[0] + f * delta[0], start[1] + f * delta[1], start[2] + f * delta[2]};
                    if (cache_sdf(cache, tp) < 0.0f) {
                        should_subdivide = 1;
                        break;
                    }
                }
            }
            /* complex edges, mesh.hpp:244-258 (skipped by the goto when the edge test fired) */
            for (int e = 0; e < 12 && !should_subdivide; e++) {
                float na[3], nb[3];
                cache_normal(cache, corner[e0[e]], na);
                cache_normal(cache, corner[e1[e]], nb);
                float angle = angle_between(na, nb, 1e-6f);
                if (angle > cfg->complex_surface_threshold)
                    should_subdivide = 1;
            }
        }
        if (nd.level == cfg->max_level) /* mesh.hpp:265-267 */
            should_subdivide = 0;

        if (should_subdivide) { /* mesh.hpp:269-278 */
            st->nodes_subdivided[nd.level]++;
            if (tail + 8 > cap) {
                if (head > 0) { /* compact */
                    memmove(q, q + head, (tail - head) * sizeof(node));
                    tail -= head;
                    head = 0;
                }
                while (tail + 8 > cap) {
                    cap *= 2;
                    q = (node *)realloc(q, cap * sizeof(node));
                }
            }
            subdivide(&nd, &q[tail]);
            tail += 8;
        } else { /* mesh.hpp:283-307 */
            for (int t = 0; t < tri_count[lookup]; t++) {
                tri_push(mesh, mid[tris[lookup][t][0]], mid[tris[lookup][t][1]], mid[tris[lookup][t][2]]);
                st->triangles[nd.level]++;
            }
        }
    }
    free(q);
}

size_t oracle_get_surface(const oracle_scene *s, const oracle_box *bb, const oracle_export_config *cfg,
                          const int tri_count[256], const int tris[256][5][3], float **out,
                          oracle_mesh_stats *stats, int threads)
{
    threads = pick_threads(threads);
    int res = 1 << cfg->grid_level;
    lattice_cache cache;
    if (cache_init(&cache, s, bb, res) != 0) {
        cache_free(&cache);
        *out = NULL;
        return 0;
    }

    /* mesh.hpp:110-125: levels 0 and 1 are subdivided unconditionally
     * (meshSubdivision = 2), leaving the 64 level-2 nodes as work items. */
    node roots[64];
    {
        node root;
        root.level = 0;
        for (int k = 0; k < 3; k++) {
            root.c[k] = bb->center[k];
            root.hd[k] = bb->diameters[k] / 2.0f; /* DesignCSG.cpp:718 */
        }
        node l1[8];
        subdivide(&root, l1);
        for (int i = 0; i < 8; i++)
            subdivide(&l1[i], &roots[8 * i]);
    }

    tri_vec part[64];
    oracle_mesh_stats pst[64];
    memset(part, 0, sizeof(part));
    memset(pst, 0, sizeof(pst));
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int r = 0; r < 64; r++)
        process_subtree(&roots[r], &cache, cfg, tri_count, tris, &part[r], &pst[r]);

    size_t total = 0;
    for (int r = 0; r < 64; r++)
        total += part[r].n;
    float *all = (float *)malloc((total ? total : 1) * 9 * sizeof(float));
    size_t o = 0;
    oracle_mesh_stats agg;
    memset(&agg, 0, sizeof(agg));
    for (int r = 0; r < 64; r++) {
        if (part[r].n)
            memcpy(all + 9 * o, part[r].v, part[r].n * 9 * sizeof(float));
        o += part[r].n;
        free(part[r].v);
        for (int l = 0; l < 32; l++) {
            agg.nodes_visited[l] += pst[r].nodes_visited[l];
            agg.nodes_culled[l] += pst[r].nodes_culled[l];
            agg.nodes_subdivided[l] += pst[r].nodes_subdivided[l];
            agg.triangles[l] += pst[r].triangles[l];
        }
    }
    agg.sdf_samples = cache.n_sdf;
    agg.normal_samples = cache.n_nrm;
    if (stats)
        *stats = agg;
    cache_free(&cache);
    *out = all;
    return total;
}

/* performGradientDescent, mesh.hpp:531-593: p <- p + n*(-s), true positions */
void oracle_gradient_descent(const oracle_scene *s, int steps, float *tris, size_t n_tris, int threads)
{
    threads = pick_threads(threads);
    int64_t nv = (int64_t)n_tris * 3;
    for (int step = 0; step < steps; step++) {
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t i = 0; i < nv; i++) {
            float *p = &tris[3 * i];
            float sd = oracle_primary_sdf(s, p[0], p[1], p[2]);
            float n[3];
            oracle_get_normal(s, p[0], p[1], p[2], n);
            float m = -sd; /* v3f_scale(normals[i], -sdfs[i]) = (s*a.x, ...), CVector.cpp:146-149 */
            p[0] = p[0] + m * n[0];
            p[1] = p[1] + m * n[1];
            p[2] = p[2] + m * n[2];
        }
    }
}

/* ------------------------------------------------------------------ */
/* T-junction repair in the spirit of cms::retopologize (mesh.hpp:432-529) -- PARITY UNPINNED, OWN DEFINITION.
 *
 * The reference's function cannot serve as a specification.  As written it is undefined behaviour (Indexer / Deindexer
 * return lambdas that capture by-value parameters by reference, :414,:424: it segfaults under g++) and
 * Vector3f::termQuotient multiplies (geometry.hpp:71-73).  With only the captures repaired it is the identity on the
 * reference's designs (SURVEY a-9) because the mis-scaled cells are far smaller than the sample spacing; with the
 * quotient repaired as well it pushes every edge sample that falls into the start vertex's own cell into the polygon
 * (measured: sphere 4/6/7 33 608 -> 169 608 triangles, all new ones slivers) -- its de-duplication helper
 * (joinDistance, Vector3fIsInVector, :447-462) is computed and never used.  The function is unfinished.
 *
 * What it is evidently FOR -- a vertex of a finer neighbour lying on a triangle's edge (a T-junction) should become a
 * vertex of that triangle -- is defined here from its own ingredients, and the HIP kernels are held to THIS:
 *   cell(v)      = (int3)( ((v - center) + hd) / hd * (float)(2^grid + 1) )           Indexer, quotient as named
 *   occupant(c)  = the mesh vertex of lowest index in the soup that lies in cell c    (occupied map, :489-494)
 *   joinDistance = 0.25 * min over axes of 2 hd / (2^grid + 1)                        (:445)
 *   per triangle, per edge a -> b, for i = 1 .. P-1, P = 2^(grid - min) (:496-505): q = a + (b - a) * ((float)i / P);
 *       V = occupant(cell(q)); V joins the polygon after a (in order of i) when it is neither a nor b, lies within
 *       joinDistance of q, and is not the vertex accepted just before it on this edge
 *   polygon = A, [joined on AB], B, [joined on BC], C, [joined on CA]; triangulated by getIndexTriangleStrip
 *       (geometry.hpp:228-248) as the reference does (:515-523)
 * The inserted points are MESH VERTICES (not sample points), so the repaired triangle shares its sub-edges with its
 * neighbours exactly; a triangle with nothing on its edges is reproduced bit for bit.  The default export does not run
 * this stage (it reproduces the reference, whose repaired function is the identity): DCSG_EXPORT_RETOPO opts in.      */
typedef struct cell_map {
    uint64_t *keys; /* 0 = empty; key = 1 + packed cell */
    uint32_t *rep;  /* lowest vertex index in the cell */
    size_t mask;
} cell_map;

static inline uint64_t retopo_cell_key(const float v[3], const float c[3], const float hd[3], float grid_points)
{
    uint64_t key = 0;
    for (int k = 0; k < 3; k++) {
        float f = ((v[k] - c[k]) + hd[k]) / hd[k]; /* diff, sum, termQuotient */
        f = f * grid_points;                       /* scaled */
        int i = (int)f;                            /* Vector3i(float, float, float) */
        key = (key << 21) | ((uint64_t)(uint32_t)i & 0x1fffffu);
    }
    return key + 1;
}
static inline size_t cell_hash(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    return (size_t)k;
}
static void cell_insert(cell_map *s, uint64_t key, uint32_t vertex)
{
    size_t h = cell_hash(key) & s->mask;
    while (s->keys[h] && s->keys[h] != key) h = (h + 1) & s->mask;
    if (!s->keys[h] || vertex < s->rep[h]) s->rep[h] = vertex;
    s->keys[h] = key;
}
static int64_t cell_occupant(const cell_map *s, uint64_t key)
{
    size_t h = cell_hash(key) & s->mask;
    while (s->keys[h]) {
        if (s->keys[h] == key) return (int64_t)s->rep[h];
        h = (h + 1) & s->mask;
    }
    return -1;
}
static inline int same_bits3(const float *a, const float *b) { return memcmp(a, b, 12) == 0; }

float oracle_retopo_join_distance(const float half_diameter[3], int grid_level)
{
    float gp = (float)((1 << grid_level) + 1);
    float a = (float)(2.0 * half_diameter[0] / gp), b = (float)(2.0 * half_diameter[1] / gp), c = (float)(2.0 * half_diameter[2] / gp);
    float m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    return (float)(0.25 * m);
}

size_t oracle_retopologize(const float *tris, size_t n_tris, const float center[3], const float half_diameter[3],
                           int min_level, int grid_level, float **out)
{
    const float grid_points = (float)((1 << grid_level) + 1);
    const int P = 1 << (grid_level - min_level);
    const float jd = oracle_retopo_join_distance(half_diameter, grid_level), jd2 = jd * jd;
    cell_map map;
    size_t cap = 64;
    while (cap < 6 * n_tris + 16) cap <<= 1;
    map.keys = (uint64_t *)calloc(cap, sizeof(uint64_t));
    map.rep = (uint32_t *)calloc(cap, sizeof(uint32_t));
    map.mask = cap - 1;
    for (size_t v = 0; v < 3 * n_tris; v++) cell_insert(&map, retopo_cell_key(tris + 3 * v, center, half_diameter, grid_points), (uint32_t)v);
    tri_vec o = {0};
    float (*ngon)[3] = (float (*)[3])malloc((size_t)(3 * P) * sizeof(*ngon));
    int *loop = (int *)malloc((size_t)(3 * P) * sizeof(int));
    int (*st)[3] = (int (*)[3])malloc((size_t)(3 * P + 4) * sizeof(*st));
    for (size_t t = 0; t < n_tris; t++) {
        const float *T = tris + 9 * t;
        int n = 0;
        for (int e = 0; e < 3; e++) {
            const float *a = T + 3 * e, *b = T + 3 * ((e + 1) % 3);
            float d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            memcpy(ngon[n++], a, 12);
            int64_t last = -1;
            for (int i = 1; i < P; i++) {
                float f = (float)i / (float)P;
                float q[3] = {a[0] + d[0] * f, a[1] + d[1] * f, a[2] + d[2] * f};
                int64_t r = cell_occupant(&map, retopo_cell_key(q, center, half_diameter, grid_points));
                if (r < 0 || r == last) continue;
                const float *V = tris + 3 * r;
                float ux = V[0] - q[0], uy = V[1] - q[1], uz = V[2] - q[2];
                if (same_bits3(V, a) || same_bits3(V, b) || !(ux * ux + uy * uy + uz * uz <= jd2)) continue;
                memcpy(ngon[n++], V, 12);
                last = r;
            }
        }
        for (int i = 0; i < n; i++) loop[i] = i;
        int k = triangle_strip(loop, n, st);
        for (int j = 0; j < k; j++) tri_push(&o, ngon[st[j][0]], ngon[st[j][1]], ngon[st[j][2]]);
    }
    free(ngon);
    free(loop);
    free(st);
    free(map.keys);
    free(map.rep);
    *out = o.v;
    return o.n;
}

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------ */
/* Preview: kernel k1 (k1.cl).  PARITY UNPINNED (see dcsg_oracle.h).       */

/* axes_cylinderSDF, k1.cl:46-48 */
static inline float axes_cylinder_sdf(float r, float h, float half_length, float radius)
{
    float a = fabsf(h) - half_length, b = r - radius;
    return a > b ? a : b; /* T_max */
}

/* primary_sdf of k1.cl:121-283 = the build procedure (identical to k2.cl's) + the three axis markers */
static float k1_primary_sdf(const oracle_scene *s, float x, float y, float z)
{
    float export_value = oracle_primary_sdf(s, x, y, z);
    /* k1.cl:239: v/5.0 -- a double division narrowed by the float3 cast, equal to the float division */
    float vx = x / 5.0f, vy = y / 5.0f, vz = z / 5.0f;
    const float radius = (float)0.015; /* AXES_RADIUS */
    float c;
    c = axes_cylinder_sdf(sqrtf(vy * vy + vz * vz), vx - 0.5f, 0.5f, radius); /* k1.cl:243-250 */
    export_value = export_value < c ? export_value : c;
    c = axes_cylinder_sdf(sqrtf(vx * vx + vz * vz), vy - 0.5f, 0.5f, radius); /* k1.cl:253-260 */
    export_value = export_value < c ? export_value : c;
    c = axes_cylinder_sdf(sqrtf(vx * vx + vy * vy), vz - 0.5f, 0.5f, radius); /* k1.cl:264-271 */
    export_value = export_value < c ? export_value : c;
    return export_value;
}

/* get_normal of k1.cl:381-417: the stencil of k2.cl over k1's primary_sdf */
static void k1_get_normal(const oracle_scene *s, float x, float y, float z, float out[3])
{
    const float e = (float)K2_NORMAL_EPSILON;
    const float zero = (float)0.0;
    float Dx = k1_primary_sdf(s, x + e, y + zero, z + zero) - k1_primary_sdf(s, x - e, y - zero, z - zero);
    float Dy = k1_primary_sdf(s, x + zero, y + e, z + zero) - k1_primary_sdf(s, x - zero, y - e, z - zero);
    float Dz = k1_primary_sdf(s, x + zero, y + zero, z + e) - k1_primary_sdf(s, x - zero, y - zero, z - e);
    float twoE = (float)(2.0 * K2_NORMAL_EPSILON);
    out[0] = (float)(1.0 / (double)twoE * (double)Dx);
    out[1] = (float)(1.0 / (double)twoE * (double)Dy);
    out[2] = (float)(1.0 / (double)twoE * (double)Dz);
    normalize3(out);
}

/* the two materials every scene starts with (scenecompiler.py:426-427 and the basic-lighting text) */
static void default_shader(const float cam[9], const float n[3], int material_id, float out[3])
{
    if (material_id == 0) { /* "return fabs(n);" */
        out[0] = fabsf(n[0]);
        out[1] = fabsf(n[1]);
        out[2] = fabsf(n[2]);
    } else if (material_id == 1) { /* n_g = n.x*rgt_g+n.y*upp_g+n.z*fwd_g; L = dot(n_g,(0,0,-1)) */
        float g[3];
        for (int k = 0; k < 3; k++)
            g[k] = (n[0] * cam[k] + n[1] * cam[3 + k]) + n[2] * cam[6 + k];
        float L = (g[0] * 0.0f + g[1] * 0.0f) + g[2] * -1.0f;
        out[0] = out[1] = out[2] = L;
    } else { /* shader_bank's trailing return */
        out[0] = out[1] = out[2] = 1.0f;
    }
}

/* shade, k1.cl:285-379 */
static void k1_shade(const oracle_scene *s, const float cam[9], const float v[3], const float n[3], float out[3])
{
    int match = -1;
    float abc_out[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < s->n_objects; i++) {
        float dx = v[0] - s->pos[i * 3 + 0], dy = v[1] - s->pos[i * 3 + 1], dz = v[2] - s->pos[i * 3 + 2];
        float A = dot3(dx, dy, dz, s->right[i * 3 + 0], s->right[i * 3 + 1], s->right[i * 3 + 2]);
        float B = dot3(dx, dy, dz, s->up[i * 3 + 0], s->up[i * 3 + 1], s->up[i * 3 + 2]);
        float C = dot3(dx, dy, dz, s->fwd[i * 3 + 0], s->fwd[i * 3 + 1], s->fwd[i * 3 + 2]);
        float d = s->bank(A, B, C, s->shape_id[i]);
        if ((double)d < 0.005 * 2.0) { /* SDF_EPSILON*TOLERANCE_FACTOR_MATERIAL; the LAST match wins */
            match = i;
            abc_out[0] = A;
            abc_out[1] = B;
            abc_out[2] = C;
        }
    }
    if (match != -1) {
        if (s->shader)
            s->shader(v, abc_out, n, s->material_id[match], out);
        else
            default_shader(cam, n, s->material_id[match], out);
        return;
    }
    float vx = v[0] / 5.0f, vy = v[1] / 5.0f, vz = v[2] / 5.0f;
    const float radius = (float)0.025;
    if ((double)axes_cylinder_sdf(sqrtf(vy * vy + vz * vz), vx - 0.5f, 0.5f, radius) < 0.005 * 2.0) {
        out[0] = 1.0f, out[1] = 0.0f, out[2] = 0.0f;
        return;
    }
    if ((double)axes_cylinder_sdf(sqrtf(vx * vx + vz * vz), vy - 0.5f, 0.5f, radius) < 0.005 * 2.0) {
        out[0] = 0.0f, out[1] = 1.0f, out[2] = 0.0f;
        return;
    }
    if ((double)axes_cylinder_sdf(sqrtf(vx * vx + vy * vy), vz - 0.5f, 0.5f, radius) < 0.005 * 2.0) {
        out[0] = 0.0f, out[1] = 0.0f, out[2] = 1.0f;
        return;
    }
    out[0] = (float)(239.0 / 255.0); /* k1.cl:377; "245/255" is an integer division: 0 */
    out[1] = (float)(66.0 / 255.0);
    out[2] = (float)(245 / 255);
}

/* RCOMP/GCOMP/BCOMP + clip, k1.cl:8-10,472-478: (int)(255.0*c) in double, then clamped to a byte.  The
 * double -> int conversion saturates and maps NaN to 0 (what the GPU's conversion does; C leaves it undefined). */
static unsigned char colour_byte(float c)
{
    double x = 255.0 * (double)c;
    int v;
    if (!(x == x))
        v = 0;
    else if (x <= -2147483648.0)
        v = (int)(-2147483647 - 1);
    else if (x >= 2147483647.0)
        v = 2147483647;
    else
        v = (int)x;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

void oracle_render_preview(const oracle_scene *s, const float campos[3], const float right[3], const float up[3],
                           const float forward[3], unsigned char *rgb_out, int threads)
{
    threads = pick_threads(threads);
    float cam[9];
    for (int k = 0; k < 3; k++)
        cam[k] = right[k], cam[3 + k] = up[k], cam[6 + k] = forward[k];
    if (s->set_camera)
        s->set_camera(right, up, forward); /* rgt_g, upp_g, fwd_g: k1.cl:521-523 */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64)
    for (int tid = 0; tid < 640 * 480; tid++) {
        int ix = tid % 640, iy = tid / 640;
        /* k1.cl:515: uv = ((float)(ix-320), -(float)(iy-240)) / (320, 320) */
        float uvx = (float)(ix - 640 / 2) / 320.0f, uvy = -(float)(iy - 480 / 2) / 320.0f;
        float r[3] = {uvx, uvy, 1.0f}; /* IFOV */
        float colour[3] = {1.0f, 1.0f, 1.0f};
        /* march, k1.cl:419-470 */
        float ox = dot3(campos[0], campos[1], campos[2], right[0], right[1], right[2]);
        float oy = dot3(campos[0], campos[1], campos[2], up[0], up[1], up[2]);
        float oz = dot3(campos[0], campos[1], campos[2], forward[0], forward[1], forward[2]);
        float rx = dot3(r[0], r[1], r[2], right[0], right[1], right[2]);
        float ry = dot3(r[0], r[1], r[2], up[0], up[1], up[2]);
        float rz = dot3(r[0], r[1], r[2], forward[0], forward[1], forward[2]);
        float d = 0.0f, vx = ox, vy = oy, vz = oz, hit = -1.0f;
        for (int i = 0; i < 512; i++) { /* MAX_STEPS */
            float st = (float)((double)k1_primary_sdf(s, vx, vy, vz) * 0.85); /* TOLERANCE_FACTOR_MARCHSTEP */
            if ((double)st < 0.005) { /* SDF_EPSILON */
                hit = d;
                break;
            }
            vx = vx + st * rx;
            vy = vy + st * ry;
            vz = vz + st * rz;
            d = d + st;
            if ((double)d > 64.0) /* MAX_DISTANCE */
                break;
        }
        if ((double)hit > 0.0) { /* k1.cl:548 */
            float p[3] = {ox + hit * rx, oy + hit * ry, oz + hit * rz};
            float n[3];
            k1_get_normal(s, p[0], p[1], p[2], n);
            k1_shade(s, cam, p, n, colour);
        }
        rgb_out[tid * 3 + 0] = colour_byte(colour[0]);
        rgb_out[tid * 3 + 1] = colour_byte(colour[1]);
        rgb_out[tid * 3 + 2] = colour_byte(colour[2]);
    }
}

/* ------------------------------------------------------------------ */
/* writeTrianglesToSTL, utils.hpp:41-103: 80 zero bytes, u32 count, per
 * triangle zero normal then A,B,C each as (x, z, y), u16 0.              */
int oracle_write_stl(const char *path, const float *tris, size_t n_tris)
{
    FILE *f = fopen(path, "wb");
    if (!f)
        return -1;
    unsigned char header[80] = {0};
    fwrite(header, 1, 80, f);
    uint32_t n = (uint32_t)n_tris;
    fwrite(&n, 4, 1, f);
    for (size_t t = 0; t < n_tris; t++) {
        const float *v = &tris[9 * t];
        float rec[12] = {0.0f, 0.0f, 0.0f, v[0], v[2], v[1], v[3], v[5], v[4], v[6], v[8], v[7]};
        uint16_t zero = 0;
        fwrite(rec, 4, 12, f);
        fwrite(&zero, 2, 1, f);
    }
    fclose(f);
    return 0;
}

/* writeTrianglesToPLY, utils.hpp:106-154 through happly.h:1998-2039 (header),
 * 1538-1570 (double properties), 1640-1670 (uchar-count uint lists).       */
int oracle_write_ply(const char *path, const float *tris, size_t n_tris)
{
    FILE *f = fopen(path, "wb");
    if (!f)
        return -1;
    fprintf(f,
            "ply\nformat binary_little_endian 1.0\n"
            "comment Written with hapPLY (https://github.com/nmwsharp/happly)\n"
            "element vertex %zu\nproperty double x\nproperty double y\nproperty double z\n"
            "element face %zu\nproperty list uchar uint vertex_indices\nend_header\n",
            n_tris * 3, n_tris);
    for (size_t i = 0; i < n_tris * 3; i++) {
        double d[3] = {(double)tris[3 * i], (double)tris[3 * i + 1], (double)tris[3 * i + 2]};
        fwrite(d, 8, 3, f);
    }
    for (size_t t = 0; t < n_tris; t++) {
        unsigned char c = 3;
        uint32_t idx[3] = {(uint32_t)(3 * t), (uint32_t)(3 * t + 1), (uint32_t)(3 * t + 2)};
        fwrite(&c, 1, 1, f);
        fwrite(idx, 4, 3, f);
    }
    fclose(f);
    return 0;
}
