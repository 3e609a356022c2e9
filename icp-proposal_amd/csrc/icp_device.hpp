// icp_device.hpp — device-side building blocks shared by the HIP kernels (gfx950 only).
//
// Exact-geometry functions are written so that every IEEE-double operation happens in a fixed order with
// NO fused multiply-add (the translation units are compiled with -ffp-contract=off): the reference runs on
// the JVM, where a*b+c rounds twice, and the correspondence indices must be bit-identical to a CPU evaluation
// of the same expressions (BASELINE.json north_star).  Where fusing is harmless (pruning bounds, r-space
// linear algebra) the code calls fma() explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icp {

// A pointer READ FROM A RECORD IN MEMORY (the wide step's resident argument records, the batch records) is a generic pointer to the
// compiler — only kernel arguments are known to point at global memory — and every access through it a FLAT instruction.  FLAT loads
// come back in no fixed order against each other (memory vs LDS aperture), so a wait for one of them is `s_waitcnt vmcnt(0) lgkmcnt(0)`:
// a wait for ALL — no round of loads can stay in flight across the use of an earlier one.  A pointer of type global_ptr<T> states what
// it points at: global_load / global_store, counted waits.  (A cast to address space 1 and back is folded away: the TYPE has to stay.)
template <class T> using global_ptr = __attribute__((address_space(1))) T*;
template <class T> __device__ __forceinline__ global_ptr<T> as_global(T* p) { return (global_ptr<T>)p; }

struct Pose {      // ModelFittingParameters.scala:79-90 — scale ∘ translation ∘ rotation about a centre
  double R[9];     // Rz(phi)·Ry(theta)·Rx(psi), computed on the host (libm sin/cos)
  double t[3];
  double ctr[3];
  double s;
};

struct d3 {
  double x, y, z;
};

__device__ __forceinline__ d3 ld3(const double* __restrict__ p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ d3 sub(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ d3 normalized(d3 v) {
  double n = sqrt(dot(v, v));
  return {v.x / n, v.y / n, v.z / n};
}

// Closest point on triangle (a,b,c) to p: Voronoi-region method (Ericson, Real-Time Collision Detection §5.1.5).
// Replaces the point/triangle kernel inside Scalismo's closestPointOnSurface (NonRigidIcpProposal.scala:97).
__device__ __forceinline__ d3 closest_point_triangle(d3 p, d3 a, d3 b, d3 c) {
  d3 ab = sub(b, a), ac = sub(c, a), ap = sub(p, a);
  double d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0 && d2 <= 0.0) return a;
  d3 bp = sub(p, b);
  double d3_ = dot(ab, bp), d4 = dot(ac, bp);
  if (d3_ >= 0.0 && d4 <= d3_) return b;
  double vc = d1 * d4 - d3_ * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) {
    double v = d1 / (d1 - d3_);
    return {a.x + v * ab.x, a.y + v * ab.y, a.z + v * ab.z};
  }
  d3 cp = sub(p, c);
  double d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0 && d5 <= d6) return c;
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    double w = d2 / (d2 - d6);
    return {a.x + w * ac.x, a.y + w * ac.y, a.z + w * ac.z};
  }
  double va = d3_ * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0) {
    double w = (d4 - d3_) / ((d4 - d3_) + (d5 - d6));
    return {b.x + w * (c.x - b.x), b.y + w * (c.y - b.y), b.z + w * (c.z - b.z)};
  }
  double denom = 1.0 / ((va + vb) + vc);
  double v = vb * denom, w = vc * denom;
  return {(a.x + ab.x * v) + ac.x * w, (a.y + ab.y * v) + ac.y * w, (a.z + ab.z * v) + ac.z * w};
}

// squared distance from p to triangle `t` of (verts, tris); optionally returns the closest point
__device__ __forceinline__ double tri_dist2(d3 p, const double* __restrict__ verts, const int* __restrict__ tris, int t, d3* cp_out) {
  int ia = tris[3 * t], ib = tris[3 * t + 1], ic = tris[3 * t + 2];
  d3 cp = closest_point_triangle(p, ld3(verts + 3 * ia), ld3(verts + 3 * ib), ld3(verts + 3 * ic));
  if (cp_out) *cp_out = cp;
  d3 d = sub(p, cp);
  return dot(d, d);
}

// unit normal of triangle t (Scalismo cell normal: (b−a)×(c−a) normalised)
__device__ __forceinline__ d3 cell_normal(const double* __restrict__ x, const int* __restrict__ tris, int t) {
  d3 a = ld3(x + 3 * tris[3 * t]), b = ld3(x + 3 * tris[3 * t + 1]), c = ld3(x + 3 * tris[3 * t + 2]);
  return normalized(cross(sub(b, a), sub(c, a)));
}

// vertex normal = normalised sum of adjacent unit cell normals in ascending triangle id
__device__ __forceinline__ d3 vertex_normal(const double* __restrict__ x, const int* __restrict__ tris,
                                            const int* __restrict__ adj_off, const int* __restrict__ adj, int v) {
  d3 n = {0.0, 0.0, 0.0};
  for (int k = adj_off[v]; k < adj_off[v + 1]; ++k) {
    d3 cn = cell_normal(x, tris, adj[k]);
    n.x += cn.x; n.y += cn.y; n.z += cn.z;
  }
  return normalized(n);
}

// non-negative doubles order like their bit patterns: lets a 64-bit integer atomicMin pick the exact minimum
__device__ __forceinline__ unsigned long long d2bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ double bits2d(unsigned long long b) { return __longlong_as_double((long long)b); }

// the triangle ids behind a sphere list of T entries (they follow the spheres in the same buffer: launch_tri_spheres)
__host__ __device__ __forceinline__ int* sphere_triangles(float4* spheres, int T) { return (int*)(spheres + T); }
__host__ __device__ __forceinline__ const int* sphere_triangles(const float4* spheres, int T) { return (const int*)(spheres + T); }

constexpr unsigned long long kInfBits = 0x7FF0000000000000ull;
constexpr int kNoIndex = 0x7FFFFFFF;

}  // namespace icp
