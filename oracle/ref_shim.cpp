// ref_shim.cpp -- C entry points onto the parts of the REFERENCE that compile on their own.
//
// TEST INFRASTRUCTURE.  This file contains no reference code: it #includes the reference's headers from where they
// lie under /root/reference (oracle/Makefile target `ref`, output oracle/_ref/libdcsg_ref.so) and forwards to them.
// What builds without stand-ins (standard headers + the vendored happly.h only): cms/main/Headers/geometry.hpp
// (Vector3f, Box3f::getCorners, getIndexTriangleStrip), octree.hpp (Node::subdivideIntoQueue), readLookupTable.hpp
// (getIndexTrianglesFromTable) and utils.hpp (writeTrianglesToSTL / writeTrianglesToPLY).  mesh.hpp, ISV.hpp and
// Evaluator.* are NOT here: they need wx/wx.h, Win32 headers and an OpenCL runtime (Oracle).
// tests/test_oracle_ref.py checks the oracle and the product's host code against these functions.
#include <cmath>
#include <cstdio>
#include <deque>
#include <map>
#include <vector>

#define logRoutine(...) ((void)0)  // the headers print progress through this macro (utils.hpp:8-10)
#include "utils.hpp"               // pulls geometry.hpp and happly.h
#include "octree.hpp"
#include "readLookupTable.hpp"

static std::vector<cms::Triangle3f> soup(const float* t, unsigned long long n)
{
    std::vector<cms::Triangle3f> v;
    v.reserve(n);
    for (unsigned long long i = 0; i < n; i++, t += 9)
        v.push_back(cms::Triangle3f(cms::Vector3f(t[0], t[1], t[2]), cms::Vector3f(t[3], t[4], t[5]), cms::Vector3f(t[6], t[7], t[8])));
    return v;
}

extern "C" {

int ref_write_stl(const char* path, const float* tris, unsigned long long n)
{
    int written = 0;
    cms::writeTrianglesToSTL(path, soup(tris, n), &written);
    return written;
}

int ref_write_ply(const char* path, const float* tris, unsigned long long n)
{
    int written = 0;
    cms::writeTrianglesToPLY(path, soup(tris, n), &written);
    return written;
}

// table[case][k][3] edge ids of triangle k, count[case]; returns the number of cases present in the file
int ref_lookup_table(const char* path, int count[256], int tris[256][5][3])
{
    std::map<int, std::vector<cms::IndexTriangle>> m = cms::getIndexTrianglesFromTable(path);
    for (int c = 0; c < 256; c++) count[c] = 0;
    for (auto& kv : m) {
        if (kv.first < 0 || kv.first > 255) return -1;
        int k = 0;
        for (cms::IndexTriangle& t : kv.second) {
            if (k >= 5) return -2;
            tris[kv.first][k][0] = t.x;
            tris[kv.first][k][1] = t.y;
            tris[kv.first][k][2] = t.z;
            k++;
        }
        count[kv.first] = k;
    }
    return (int)m.size();
}

void ref_box_corners(const float c[3], const float hd[3], float s, float out[8][3])
{
    cms::Box3f b(cms::Vector3f(c[0], c[1], c[2]), cms::Vector3f(hd[0], hd[1], hd[2]));
    std::vector<cms::Vector3f> k = b.getCorners(s);
    for (int i = 0; i < 8; i++) out[i][0] = k[i].x, out[i][1] = k[i].y, out[i][2] = k[i].z;
}

void ref_subdivide(const float c[3], const float hd[3], float child_centers[8][3], float child_hd[3])
{
    cms::Node n(cms::Box3f(cms::Vector3f(c[0], c[1], c[2]), cms::Vector3f(hd[0], hd[1], hd[2])), 0);
    std::deque<cms::Node> q;
    n.subdivideIntoQueue(q);
    for (int i = 0; i < 8; i++) {
        child_centers[i][0] = q[i].bounds.center.x;
        child_centers[i][1] = q[i].bounds.center.y;
        child_centers[i][2] = q[i].bounds.center.z;
    }
    child_hd[0] = q[0].bounds.halfDiameter.x;
    child_hd[1] = q[0].bounds.halfDiameter.y;
    child_hd[2] = q[0].bounds.halfDiameter.z;
}

float ref_angle_between(const float a[3], const float b[3], float tolerance)
{
    return cms::Vector3f::angleBetweenVectors(cms::Vector3f(a[0], a[1], a[2]), cms::Vector3f(b[0], b[1], b[2]), tolerance);
}

float ref_magnitude(const float a[3]) { return cms::Vector3f(a[0], a[1], a[2]).magnitude(); }

}  // extern "C"
