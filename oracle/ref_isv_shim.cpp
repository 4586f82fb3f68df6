// ref_isv_shim.cpp -- C entry points onto the REFERENCE's lattice cache, ISV::ISV3D64 (ISV.hpp:8-138).
//
// TEST INFRASTRUCTURE.  No reference code is copied: this file #includes CVector.h and ISV.hpp from where they lie under
// /root/reference (oracle/Makefile target `ref`) and instantiates the template exactly as the export driver does
// (DesignCSG.cpp:731-733: W = H = D = res, WW = HH = DD = res / cacheSubdivision, the stage-A box, a std::function
// data source, queriesBeforeFree, queriesBeforeGC).  Everything the lattice contract consists of -- getCoords (the
// truncating snap, ISV.hpp:91-96), hash (the block key, :85-87), the block point list (z outer, x inner, :28-33),
// getIndex (:97-102), getPoint (:103-109), getValue (:15-60: the value AT THE SNAPPED POINT, block evaluated on a miss,
// garbage collection) -- runs as the reference wrote it.
//
// One caveat, stated where it applies: ISV.hpp calls v3f / v3f_add / v3f_sub / v3f_scale, which CVector.h declares and
// CVector.cpp defines (:98-101, :128-133, :146-149) -- and CVector.cpp includes <windows.h>, so it cannot be compiled
// in this image and writing a stand-in header is not allowed.  The four helpers are component-wise one-liners; they are
// defined below so that the template links.  getCoords / hash / getIndex / getValue do not touch them; getPoint's
// expression structure (the order of the float operations) is the reference's, only "a.x + b.x" style bodies are ours.
// Kept in its own translation unit: CVector.h's global Vector3f / T_min / T_max must not meet the cms:: headers of
// ref_shim.cpp.
#include <cstdint>
#include <functional>
#include <vector>

#include "CVector.h"
#include "ISV.hpp"

// CVector.cpp:98-101, 128-133, 146-149 (see the caveat above)
Vector3f v3f(float x, float y, float z)
{
    Vector3f r = {x, y, z};
    return r;
}
Vector3f v3f_add(Vector3f a, Vector3f b) { return v3f(a.x + b.x, a.y + b.y, a.z + b.z); }
Vector3f v3f_sub(Vector3f a, Vector3f b) { return v3f(a.x - b.x, a.y - b.y, a.z - b.z); }
Vector3f v3f_scale(Vector3f a, float s) { return v3f(s * a.x, s * a.y, s * a.z); }

namespace {
using Source = std::function<std::vector<float>(std::vector<v3f_t>&)>;
using Lattice = ISV::ISV3D64<float, Source>;

box_t make_box(const float c[3], const float d[3])
{
    box_t b{};
    b.center = v3f(c[0], c[1], c[2]);
    b.diameters = v3f(d[0], d[1], d[2]);
    return b;
}
}  // namespace

extern "C" {

// getCoords of n points (ISV.hpp:91-96); out[3*i..] = (ix, iy, iz), not clamped (the reference does not clamp)
void ref_isv_coords(const float center[3], const float diam[3], int res, int cache_subdivision, const float* pts,
                    unsigned long long n, long long* out)
{
    Source none = [](std::vector<v3f_t>&) { return std::vector<float>(); };
    Lattice L(res, res, res, res / cache_subdivision, res / cache_subdivision, res / cache_subdivision, make_box(center, diam), none,
              4096, -1);
    for (unsigned long long i = 0; i < n; i++) {
        ISV::I643 c = L.getCoords(v3f(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        out[3 * i] = std::get<0>(c), out[3 * i + 1] = std::get<1>(c), out[3 * i + 2] = std::get<2>(c);
    }
}

// getPoint of n index triples (ISV.hpp:103-109)
void ref_isv_points(const float center[3], const float diam[3], int res, const int* idx, unsigned long long n, float* out)
{
    Source none = [](std::vector<v3f_t>&) { return std::vector<float>(); };
    Lattice L(res, res, res, res, res, res, make_box(center, diam), none, 4096, -1);
    for (unsigned long long i = 0; i < n; i++) {
        v3f_t p = L.getPoint(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]);
        out[3 * i] = p.x, out[3 * i + 1] = p.y, out[3 * i + 2] = p.z;
    }
}

// getValue (ISV.hpp:15-60) of n query points through ONE lattice object, as a mesher thread uses it.  The data source
// hands every block's point list to `eval` (a C callback: n points xyz -> n values).  Per query: the value returned and
// sampled_at = the lattice POINT whose value it is (recovered by a second lattice whose data source returns the running
// index of each block point, so nothing but the reference's own hash / getIndex decides which point that is).
// Returns the number of blocks the data source was asked for.
typedef void (*ref_isv_eval_fn)(const float* xyz, unsigned long long n, float* out, void* user);
long long ref_isv_values(const float center[3], const float diam[3], int res, int cache_subdivision, int queries_before_free,
                         int queries_before_gc, const float* pts, unsigned long long n, ref_isv_eval_fn eval, void* user,
                         float* values, float* sampled_at)
{
    long long blocks = 0;
    Source src = [&](std::vector<v3f_t>& grid) {
        blocks++;
        std::vector<float> xyz(grid.size() * 3), out(grid.size());
        for (size_t i = 0; i < grid.size(); i++) xyz[3 * i] = grid[i].x, xyz[3 * i + 1] = grid[i].y, xyz[3 * i + 2] = grid[i].z;
        eval(xyz.data(), grid.size(), out.data(), user);
        return out;
    };
    std::vector<v3f_t> last_grid;
    Source index_src = [&](std::vector<v3f_t>& grid) {
        last_grid = grid;
        std::vector<float> out(grid.size());
        for (size_t i = 0; i < grid.size(); i++) out[i] = (float)i;  // blocks hold at most 64^3 < 2^24 points: exact
        return out;
    };
    const int w = res / cache_subdivision;
    Lattice L(res, res, res, w, w, w, make_box(center, diam), src, queries_before_free, queries_before_gc);
    for (unsigned long long i = 0; i < n; i++) {
        v3f_t p = v3f(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        values[i] = L.getValue(p);
        Lattice probe(res, res, res, w, w, w, make_box(center, diam), index_src, 4096, -1);  // fresh: always a miss
        size_t k = (size_t)probe.getValue(p);
        v3f_t q = last_grid.at(k);
        sampled_at[3 * i] = q.x, sampled_at[3 * i + 1] = q.y, sampled_at[3 * i + 2] = q.z;
    }
    return blocks;
}

}  // extern "C"
