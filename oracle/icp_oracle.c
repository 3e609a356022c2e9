/*
 * icp_oracle.c — CPU restatement (IEEE double, single thread, brute force) of the closest-point-proposal
 * hot path of unibas-gravis/icp-proposal.
 *
 *   *** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 *   leg may load this library.  The product (icp-proposal_amd/) never links, imports or calls it.
 *
 *   *** PARITY UNPINNED ***  The reference is Scala on Scalismo 0.90.0 / Breeze (build.sbt:14-19), none of
 *   which exists in this image (no JVM, no jars), it has no tests and no golden vectors, and its one sample
 *   log is a missing blob.  This file follows the reference's own source line by line where the logic is in
 *   the reference (citations "ref:" are relative to /root/reference/src/main/scala/), and the published
 *   Scalismo semantics (all tagged SCALISMO-UNVERIFIED) where the reference delegates.
 *   It is cross-checked against independent numpy/scipy formulations in tests/, never against Scalismo.
 *
 * Deliberately the LONG forms (N-point regressions, explicit eigen-decomposition of the posterior) — the
 * device path uses r-space closed forms, so agreement between the two is a real check.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  No FMA contraction: the reference
 * runs on the JVM, whose a*b+c is two roundings.
 */
#include "../include/icp_sincos.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "icp_spatial.h"
#ifdef _OPENMP
#include <omp.h>
#else
static inline int omp_get_max_threads(void) { return 1; }
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ small helpers */

static inline double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void sub3(const double *a, const double *b, double *o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void cross3(const double *a, const double *b, double *o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void normalize3(double *v) {
  double n = sqrt(dot3(v, v));
  v[0] /= n; v[1] /= n; v[2] /= n;
}

/* ------------------------------------------------------------------ opaque objects */

typedef struct {
  int N, T, r;
  double *ref;    /* [N*3] reference vertices x̄ */
  double *mean;   /* [N*3] mean deformation μ */
  double *Q;      /* [3N*r] Φ·diag(√λ), row-major */
  double *phi;    /* [3N*r] unscaled Φ */
  double *lambda; /* [r] */
  int *tris;      /* [T*3] */
  int *adj_off;   /* [N+1] CSR vertex -> adjacent triangles (ascending triangle id) */
  int *adj;
  unsigned char *boundary; /* [N] */
} orc_model;

typedef struct {
  int M, T;
  double *pts;
  int *tris;
  unsigned char *boundary;
  int n_boundary;
} orc_mesh;

static int cmp_i64(const void *x, const void *y) {
  int64_t a = *(const int64_t *)x, b = *(const int64_t *)y;
  return (a > b) - (a < b);
}

/* vertex lies on an edge owned by exactly one triangle (Scalismo pointIsOnBoundary) */
static void boundary_flags(int V, int T, const int *tris, unsigned char *flags) {
  /* count directed+undirected edge multiplicity with a sort over 3T keys */
  int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)T);
  for (int t = 0; t < T; ++t)
    for (int e = 0; e < 3; ++e) {
      int64_t a = tris[3 * t + e], b = tris[3 * t + (e + 1) % 3];
      if (a > b) { int64_t s = a; a = b; b = s; }
      keys[3 * t + e] = a * (int64_t)V + b;
    }
  qsort(keys, 3 * (size_t)T, sizeof(int64_t), cmp_i64);
  memset(flags, 0, (size_t)V);
  size_t n = 3 * (size_t)T, i = 0;
  while (i < n) {
    size_t j = i + 1;
    while (j < n && keys[j] == keys[i]) ++j;
    if (j - i == 1) {
      flags[keys[i] / V] = 1;
      flags[keys[i] % V] = 1;
    }
    i = j;
  }
  free(keys);
}

static void build_adjacency(int V, int T, const int *tris, int **off_out, int **adj_out) {
  int *off = (int *)calloc((size_t)V + 1, sizeof(int));
  for (int i = 0; i < 3 * T; ++i) off[tris[i] + 1]++;
  for (int v = 0; v < V; ++v) off[v + 1] += off[v];
  int *adj = (int *)malloc(sizeof(int) * (size_t)(3 * T > 0 ? 3 * T : 1));
  int *fill = (int *)calloc((size_t)V, sizeof(int));
  for (int t = 0; t < T; ++t)
    for (int e = 0; e < 3; ++e) {
      int v = tris[3 * t + e];
      adj[off[v] + fill[v]++] = t;
    }
  free(fill);
  *off_out = off;
  *adj_out = adj;
}

ORC_API orc_model *orc_model_create(int N, int T, int r, const double *ref, const double *mean_def,
                                    const double *basis, const double *lambda, const int *tris) {
  orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
  m->N = N; m->T = T; m->r = r;
  m->ref = (double *)malloc(sizeof(double) * 3 * N);
  m->mean = (double *)calloc(3 * (size_t)N, sizeof(double));
  m->phi = (double *)malloc(sizeof(double) * 3 * (size_t)N * r);
  m->Q = (double *)malloc(sizeof(double) * 3 * (size_t)N * r);
  m->lambda = (double *)malloc(sizeof(double) * r);
  m->tris = (int *)malloc(sizeof(int) * 3 * T);
  memcpy(m->ref, ref, sizeof(double) * 3 * N);
  if (mean_def) memcpy(m->mean, mean_def, sizeof(double) * 3 * N);
  memcpy(m->phi, basis, sizeof(double) * 3 * (size_t)N * r);
  memcpy(m->lambda, lambda, sizeof(double) * r);
  memcpy(m->tris, tris, sizeof(int) * 3 * T);
  /* Q = Φ·D, D = diag(√λ)  (SURVEY App. A notation; Scalismo genericRegressionComputations scales φ_j by √λ_j) */
  for (size_t i = 0; i < 3 * (size_t)N; ++i)
    for (int j = 0; j < r; ++j) m->Q[i * r + j] = m->phi[i * r + j] * sqrt(lambda[j]);
  build_adjacency(N, T, m->tris, &m->adj_off, &m->adj);
  m->boundary = (unsigned char *)malloc((size_t)N);
  boundary_flags(N, T, m->tris, m->boundary);
  return m;
}

ORC_API void orc_model_destroy(orc_model *m) {
  if (!m) return;
  free(m->ref); free(m->mean); free(m->Q); free(m->phi); free(m->lambda); free(m->tris);
  free(m->adj_off); free(m->adj); free(m->boundary); free(m);
}

ORC_API orc_mesh *orc_mesh_create(int M, int T, const double *pts, const int *tris) {
  orc_mesh *m = (orc_mesh *)calloc(1, sizeof(orc_mesh));
  m->M = M; m->T = T;
  m->pts = (double *)malloc(sizeof(double) * 3 * M);
  m->tris = (int *)malloc(sizeof(int) * 3 * T);
  memcpy(m->pts, pts, sizeof(double) * 3 * M);
  memcpy(m->tris, tris, sizeof(int) * 3 * T);
  m->boundary = (unsigned char *)malloc((size_t)M);
  boundary_flags(M, T, m->tris, m->boundary);
  m->n_boundary = 0;
  for (int i = 0; i < M; ++i) m->n_boundary += m->boundary[i];
  return m;
}

ORC_API void orc_mesh_destroy(orc_mesh *m) {
  if (!m) return;
  free(m->pts); free(m->tris); free(m->boundary); free(m);
}

ORC_API void orc_model_boundary(const orc_model *m, unsigned char *out) { memcpy(out, m->boundary, (size_t)m->N); }
ORC_API void orc_mesh_boundary(const orc_mesh *m, unsigned char *out) { memcpy(out, m->boundary, (size_t)m->M); }

/* ------------------------------------------------------------------ a1/a2: parameters -> mesh
 * theta = [s | t(3) | phi,theta,psi | centre(3) | c(r)]   ref: api/sampling/ModelFittingParameters.scala:27-36,64 */

/* Scalismo Rotation(phi,theta,psi,centre): R = Rz(phi)·Ry(theta)·Rx(psi)  [SCALISMO-UNVERIFIED, SURVEY App. B8] */
/* sines and cosines: include/icp_sincos.h — ONE plain-arithmetic implementation shared with the library's host and device code, so
 * that the three rotation matrices of an Euler triple are the same bits (the device's libm is not the host's) */
ORC_API void orc_rotation_matrix(double phi, double theta, double psi, double *R) { icp_rotation_matrix(phi, theta, psi, R); }
ORC_API void orc_sincos(double x, double *s, double *c) { icp_sincos(x, s, c); }

/* ref: ModelFittingParameters.scala:79-110 — x = scale(pose(shape(x̄))).
 * shape: x̄ + (μ + Σ_j Q_ij c_j) accumulated in basis order (GP instance via NearestNeighborInterpolator
 * evaluated at a reference vertex returns that vertex's rows);
 * pose: R(u − ctr) + ctr + t (:79-86); scale: s·x (:88-90). */
ORC_API void orc_instance(const orc_model *m, const double *theta, double *x) {
  const int r = m->r;
  const double s = theta[0], *t = theta + 1, *ctr = theta + 7, *c = theta + 10;
  double R[9];
  orc_rotation_matrix(theta[4], theta[5], theta[6], R);
  for (int i = 0; i < m->N; ++i) {
    double u[3], v[3];
    for (int d = 0; d < 3; ++d) {
      const double *q = m->Q + (size_t)(3 * i + d) * r;
      double acc = m->mean[3 * i + d];
      for (int j = 0; j < r; ++j) acc = acc + q[j] * c[j];
      u[d] = m->ref[3 * i + d] + acc;
    }
    sub3(u, ctr, v);
    for (int d = 0; d < 3; ++d) {
      double w = (R[3 * d] * v[0] + R[3 * d + 1] * v[1]) + R[3 * d + 2] * v[2];
      x[3 * i + d] = s * ((w + ctr[d]) + t[d]);
    }
  }
}

/* inverse of the RIGID pose only (ref: NonRigidIcpProposal.scala:142): R^T((q − t) − ctr) + ctr */
static void inverse_pose(const double *theta, const double *q, double *o) {
  double R[9], v[3];
  orc_rotation_matrix(theta[4], theta[5], theta[6], R);
  for (int d = 0; d < 3; ++d) v[d] = (q[d] - theta[1 + d]) - theta[7 + d];
  for (int d = 0; d < 3; ++d) o[d] = ((R[d] * v[0] + R[3 + d] * v[1]) + R[6 + d] * v[2]) + theta[7 + d];
}

/* Vertex normal = normalised sum of the unit normals of the adjacent triangles, ascending triangle id
 * (Scalismo TriangleMesh3D.vertexNormals, SURVEY App. A.2 [SCALISMO-UNVERIFIED]) */
static void cell_normal(const double *x, const int *tri, double *n) {
  double e1[3], e2[3];
  sub3(x + 3 * tri[1], x + 3 * tri[0], e1);
  sub3(x + 3 * tri[2], x + 3 * tri[0], e2);
  cross3(e1, e2, n);
  normalize3(n);
}

static void vertex_normal(const double *x, const int *tris, const int *adj_off, const int *adj, int v, double *n) {
  n[0] = n[1] = n[2] = 0.0;
  for (int k = adj_off[v]; k < adj_off[v + 1]; ++k) {
    double cn[3];
    cell_normal(x, tris + 3 * adj[k], cn);
    n[0] += cn[0]; n[1] += cn[1]; n[2] += cn[2];
  }
  normalize3(n);
}

ORC_API void orc_vertex_normals(const orc_model *m, const double *x, double *normals) {
  for (int v = 0; v < m->N; ++v) vertex_normal(x, m->tris, m->adj_off, m->adj, v, normals + 3 * v);
}

/* ------------------------------------------------------------------ brute-force spatial queries */

/* Scalismo UnstructuredPoints.findClosestPoint: exact NN; here brute force, lowest index wins ties.
 * d² evaluated as (dx·dx + dy·dy) + dz·dz. */
static void nearest_vertex_scan(const double *q, int M, const double *pts, int *idx, double *d2) {
  double best = INFINITY;
  int bi = -1;
  for (int i = 0; i < M; ++i) {
    double d[3];
    sub3(q, pts + 3 * i, d);
    double dd = dot3(d, d);
    if (dd < best) { best = dd; bi = i; }
  }
  *idx = bi;
  if (d2) *d2 = best;
}

ORC_API void orc_nearest_vertex(int K, const double *q, int M, const double *pts, int *idx, double *d2) {
  const int backend = orc_get_search_backend();  /* icp_spatial.h: the CPU-baseline variants B1 / B2, same results */
  if (backend == ORC_SEARCH_TREES) {
    for (int k = 0; k < K; ++k) {
      double dd;
      idx[k] = spatial_nearest_vertex(q + 3 * k, M, pts, &dd);
      if (d2) d2[k] = dd;
    }
    return;
  }
  if (backend == ORC_SEARCH_BRUTE_OMP && K > 1) {
    /* B2: the K independent queries of a batch over the host cores, every query the sequential scan below (same result, same
     * tie-break: lowest index).  (Round 2 split the scan of ONE query over the threads with a critical section at the end:
     * 32 threads were slower than one.) */
    const int nt = orc_get_search_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt > 0 ? nt : omp_get_max_threads())
    for (int k = 0; k < K; ++k) nearest_vertex_scan(q + 3 * k, M, pts, idx + k, d2 ? d2 + k : NULL);
    return;
  }
  for (int k = 0; k < K; ++k) nearest_vertex_scan(q + 3 * k, M, pts, idx + k, d2 ? d2 + k : NULL);
}

/* Closest point on triangle (a,b,c) to p — Voronoi-region method (Ericson, Real-Time Collision Detection §5.1.5).
 * Stands in for Scalismo's closestPointOnSurface point/triangle kernel [SCALISMO-UNVERIFIED]. */
static void closest_point_triangle(const double *p, const double *a, const double *b, const double *c, double *o) {
  double ab[3], ac[3], ap[3], bp[3], cp[3];
  sub3(b, a, ab); sub3(c, a, ac); sub3(p, a, ap);
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0.0 && d2 <= 0.0) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; return; }
  sub3(p, b, bp);
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0.0 && d4 <= d3) { o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; return; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
    double v = d1 / (d1 - d3);
    for (int k = 0; k < 3; ++k) o[k] = a[k] + v * ab[k];
    return;
  }
  sub3(p, c, cp);
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0.0 && d5 <= d6) { o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; return; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    double w = d2 / (d2 - d6);
    for (int k = 0; k < 3; ++k) o[k] = a[k] + w * ac[k];
    return;
  }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int k = 0; k < 3; ++k) o[k] = b[k] + w * (c[k] - b[k]);
    return;
  }
  double denom = 1.0 / ((va + vb) + vc);
  double v = vb * denom, w = vc * denom;
  for (int k = 0; k < 3; ++k) o[k] = (a[k] + ab[k] * v) + ac[k] * w;
}

/* mesh.operations.closestPointOnSurface(p).point: brute force over all triangles, lowest (d², triangle id) wins */
static void closest_on_surface_scan(const double *q, const double *pts, int T, const int *tris, double *cp, int *tri_idx, double *d2) {
  double best = INFINITY, bp[3] = {0, 0, 0};
  int bi = -1;
  for (int t = 0; t < T; ++t) {
    double o[3], d[3];
    closest_point_triangle(q, pts + 3 * tris[3 * t], pts + 3 * tris[3 * t + 1], pts + 3 * tris[3 * t + 2], o);
    sub3(q, o, d);
    double dd = dot3(d, d);
    if (dd < best) { best = dd; bi = t; bp[0] = o[0]; bp[1] = o[1]; bp[2] = o[2]; }
  }
  cp[0] = bp[0]; cp[1] = bp[1]; cp[2] = bp[2];
  if (tri_idx) *tri_idx = bi;
  if (d2) *d2 = best;
}

ORC_API void orc_closest_point_on_surface(int K, const double *q, const double *pts, int T, const int *tris,
                                          double *cp, int *tri_idx, double *d2) {
  const int backend = orc_get_search_backend();  /* icp_spatial.h */
  if (backend == ORC_SEARCH_TREES) {
    for (int k = 0; k < K; ++k) {
      double dd;
      const int bi = spatial_closest_on_surface(q + 3 * k, pts, T, tris, closest_point_triangle, cp + 3 * k, &dd);
      if (tri_idx) tri_idx[k] = bi;
      if (d2) d2[k] = dd;
    }
    return;
  }
  if (backend == ORC_SEARCH_BRUTE_OMP && K > 1) { /* B2: queries over the cores (see orc_nearest_vertex) */
    const int nt = orc_get_search_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt > 0 ? nt : omp_get_max_threads())
    for (int k = 0; k < K; ++k) closest_on_surface_scan(q + 3 * k, pts, T, tris, cp + 3 * k, tri_idx ? tri_idx + k : NULL, d2 ? d2 + k : NULL);
    return;
  }
  for (int k = 0; k < K; ++k) closest_on_surface_scan(q + 3 * k, pts, T, tris, cp + 3 * k, tri_idx ? tri_idx + k : NULL, d2 ? d2 + k : NULL);
}

/* ------------------------------------------------------------------ a6: surface-normal dependent noise
 * ref: api/sampling/SurfaceNoiseHelpers.scala:32-60 (including the inverted tangent selection at :46) */
ORC_API void orc_surface_noise_cov(const double *normal, double sd_normal, double sd_tangent, double *cov) {
  double n[3] = {normal[0], normal[1], normal[2]}, ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, cand[3], t1[3], t2[3];
  normalize3(n);                                   /* :38 */
  cross3(n, ex, cand);                             /* :45 */
  if (dot3(cand, cand) < 0.0001) { t1[0] = cand[0]; t1[1] = cand[1]; t1[2] = cand[2]; }
  else cross3(n, ey, t1);                          /* :46 (sic) */
  normalize3(t1);                                  /* :47 */
  cross3(n, t1, t2);
  normalize3(t2);                                  /* :50 */
  double vn = sd_normal * sd_normal, vt = sd_tangent * sd_tangent; /* :52-53 */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      cov[3 * i + j] = (n[i] * n[j]) * vn + (t1[i] * t1[j]) * vt + (t2[i] * t2[j]) * vt; /* :55-59 */
}

static void inv3(const double *a, double *o) {
  double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = 1.0 / det;
  o[0] = c00 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  o[3] = c01 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = c02 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

/* ------------------------------------------------------------------ dense r×r algebra (no BLAS/LAPACK) */

/* in-place lower Cholesky of SPD a[n*n]; returns 0 ok */
static int cholesky(int n, double *a) {
  for (int j = 0; j < n; ++j) {
    double s = a[j * n + j];
    for (int k = 0; k < j; ++k) s -= a[j * n + k] * a[j * n + k];
    if (!(s > 0.0)) return -1;
    double l = sqrt(s);
    a[j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double v = a[i * n + j];
      for (int k = 0; k < j; ++k) v -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = v / l;
    }
  }
  return 0;
}
static void chol_solve(int n, const double *l, double *b) {
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= l[i * n + k] * b[k];
    b[i] = v / l[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = b[i];
    for (int k = i + 1; k < n; ++k) v -= l[k * n + i] * b[k];
    b[i] = v / l[i * n + i];
  }
}
/* inverse of SPD matrix via Cholesky (Breeze pinv of a well-conditioned SPD matrix ≡ inverse) */
static int spd_inverse(int n, const double *a, double *inv) {
  double *l = (double *)malloc(sizeof(double) * n * n), *e = (double *)malloc(sizeof(double) * n);
  memcpy(l, a, sizeof(double) * n * n);
  int rc = cholesky(n, l);
  if (rc == 0)
    for (int j = 0; j < n; ++j) {
      memset(e, 0, sizeof(double) * n);
      e[j] = 1.0;
      chol_solve(n, l, e);
      for (int i = 0; i < n; ++i) inv[i * n + j] = e[i];
    }
  free(l); free(e);
  return rc;
}

/* cyclic Jacobi eigen-decomposition of symmetric a[n*n]: a = V diag(w) V^T, w sorted DESCENDING (as the
 * singular values Breeze svd returns), each eigenvector's largest-|.| component made positive (canonical sign;
 * LAPACK's sign is not reproducible, SURVEY §7 "hard parts"). */
ORC_API void orc_sym_eigen(int n, const double *a_in, double *w, double *V) {
  double *a = (double *)malloc(sizeof(double) * n * n);
  memcpy(a, a_in, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += a[i * n + i] * a[i * n + i];
      for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    }
    if (off <= 1e-60 * diag || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = a[p * n + q];
        if (apq == 0.0) continue;
        double tau = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) {
          double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  /* sort descending (selection sort on columns) */
  for (int i = 0; i < n; ++i) w[i] = a[i * n + i];
  for (int i = 0; i < n - 1; ++i) {
    int m = i;
    for (int j = i + 1; j < n; ++j)
      if (w[j] > w[m]) m = j;
    if (m != i) {
      double tw = w[i]; w[i] = w[m]; w[m] = tw;
      for (int k = 0; k < n; ++k) { double tv = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = tv; }
    }
  }
  for (int j = 0; j < n; ++j) {
    int m = 0;
    for (int k = 1; k < n; ++k)
      if (fabs(V[k * n + j]) > fabs(V[m * n + j])) m = k;
    if (V[m * n + j] < 0.0)
      for (int k = 0; k < n; ++k) V[k * n + j] = -V[k * n + j];
  }
  free(a);
}

/* ------------------------------------------------------------------ a4/a5/a7: the ICP posterior */

typedef struct {
  double step_length;      /* ref: NonRigidIcpProposal.scala:33 */
  double tangential_noise; /* :34 */
  double noise_along_normal; /* :35 */
  int direction;           /* 0 = ModelSampling, 1 = TargetSampling   (ref: api/other/IcpProjectionDirection.scala:19-25) */
  int boundary_aware;      /* :38 */
  int n_model_ids;         /* K_m: decimatedModel.referenceMesh.pointSet.pointIds = 0 until K_m */
  int n_target_pts;        /* K_t */
  const double *target_pts; /* decimatedTarget.pointSet.points [K_t*3] */
} orc_proposal_params;

typedef struct {
  int K;          /* number of candidate correspondences (before the boundary filter) */
  int *corr_id;   /* [K] model vertex id of each candidate */
  int *corr_aux;  /* [K] ModelSampling: nearest TARGET vertex id of the surface point (:98); TargetSampling: -1 */
  double *corr_pt;/* [K*3] target-side point (:97 / :117) */
  unsigned char *keep; /* [K] survives the boundary filter (:104 / :124) */
  double *alpha;  /* [r] posterior mean coefficients */
  double *M;      /* [r*r] I + Σ Q_i^T Σ_i^-1 Q_i */
  double *Minv;   /* [r*r] */
  double *V;      /* [r*r] eigenvectors of D·Minv·D (columns) */
  double *S;      /* [r] eigenvalues, descending */
} orc_posterior;

/* ref: NonRigidIcpProposal.scala:88-153 (icpPosterior); regression per Scalismo
 * LowRankGaussianProcess.regression / genericRegressionComputations (SURVEY App. A.4 [SCALISMO-UNVERIFIED]). */
ORC_API int orc_icp_posterior(const orc_model *m, const orc_mesh *tgt, const orc_proposal_params *pp,
                              const double *theta, orc_posterior *out) {
  const int r = m->r, N = m->N;
  double *x = (double *)malloc(sizeof(double) * 3 * N);
  orc_instance(m, theta, x);                                      /* :141 currentMesh */
  const int K = pp->direction == 1 ? pp->n_target_pts : pp->n_model_ids;
  out->K = K;
  double *obs = (double *)malloc(sizeof(double) * 3 * (K > 0 ? K : 1));   /* y_i */
  double *Linv = (double *)malloc(sizeof(double) * 9 * (K > 0 ? K : 1));  /* Σ_i^-1 */
  /* the K searches first, as ONE batch of independent queries (the CPU-baseline back ends may spread them over cores; per query
   * the result is the same scan), then the per-correspondence arithmetic in the reference's order */
  int *ids = (int *)malloc(sizeof(int) * (K > 0 ? K : 1)), *tids = (int *)malloc(sizeof(int) * (K > 0 ? K : 1));
  double *tps = (double *)malloc(sizeof(double) * 3 * (K > 0 ? K : 1));
  if (pp->direction == 1) { /* targetBasedClosestPointsEstimation :112-131 */
    orc_nearest_vertex(K, pp->target_pts, N, x, ids, NULL);         /* :118 */
    memcpy(tps, pp->target_pts, sizeof(double) * 3 * K);            /* :117 */
    for (int k = 0; k < K; ++k) tids[k] = -1;
  } else { /* modelBasedClosestPointsEstimation :89-110 */
    for (int k = 0; k < K; ++k) ids[k] = k;                         /* :94 ids 0 until K_m index the FULL mesh (:96) */
    orc_closest_point_on_surface(K, x, tgt->pts, tgt->T, tgt->tris, tps, NULL, NULL); /* :97 (x + 3·id, id = k) */
    orc_nearest_vertex(K, tps, tgt->M, tgt->pts, tids, NULL);       /* :98 */
  }
  for (int k = 0; k < K; ++k) {
    const int id = ids[k];
    double nrm[3], cov[9], back[3];
    const double *tp = tps + 3 * k;
    /* :119 currentMesh has the model's triangulation / :99 */
    const int on_boundary = pp->direction == 1 ? m->boundary[id] : tgt->boundary[tids[k]];
    out->corr_aux[k] = tids[k];
    vertex_normal(x, m->tris, m->adj_off, m->adj, id, nrm);         /* :100 / :120 */
    orc_surface_noise_cov(nrm, pp->noise_along_normal, pp->tangential_noise, cov);
    inv3(cov, Linv + 9 * k);
    out->corr_id[k] = id;
    out->corr_pt[3 * k] = tp[0]; out->corr_pt[3 * k + 1] = tp[1]; out->corr_pt[3 * k + 2] = tp[2];
    out->keep[k] = pp->boundary_aware ? !on_boundary : 1;           /* :104 / :124 */
    inverse_pose(theta, tp, back);                                  /* :108 / :129 */
    for (int d = 0; d < 3; ++d) obs[3 * k + d] = back[d] - m->ref[3 * id + d];
  }
  free(ids); free(tids); free(tps);
  /* regression: M = Q^T L Q + I,  α = Minv (Q^T L)(y − m) */
  double *M = out->M, *b = (double *)calloc((size_t)r, sizeof(double));
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < r; ++j) M[i * r + j] = (i == j);
  double *qtl = (double *)malloc(sizeof(double) * 3 * r);
  for (int k = 0; k < K; ++k) {
    if (!out->keep[k]) continue;
    const int id = out->corr_id[k];
    const double *Qi = m->Q + (size_t)3 * id * r, *Li = Linv + 9 * k;
    for (int j = 0; j < r; ++j)
      for (int d = 0; d < 3; ++d) /* QtL block = Q_i^T · Σ_i^-1  (r×3) */
        qtl[j * 3 + d] = (Qi[0 * r + j] * Li[0 * 3 + d] + Qi[1 * r + j] * Li[1 * 3 + d]) + Qi[2 * r + j] * Li[2 * 3 + d];
    for (int i = 0; i < r; ++i) {
      for (int j = 0; j < r; ++j)
        M[i * r + j] += (qtl[i * 3] * Qi[j] + qtl[i * 3 + 1] * Qi[r + j]) + qtl[i * 3 + 2] * Qi[2 * r + j];
      double e0 = obs[3 * k] - m->mean[3 * id], e1 = obs[3 * k + 1] - m->mean[3 * id + 1], e2 = obs[3 * k + 2] - m->mean[3 * id + 2];
      b[i] += (qtl[i * 3] * e0 + qtl[i * 3 + 1] * e1) + qtl[i * 3 + 2] * e2;
    }
  }
  int rc = spd_inverse(r, M, out->Minv);
  for (int i = 0; i < r; ++i) {
    double s = 0.0;
    for (int j = 0; j < r; ++j) s += out->Minv[i * r + j] * b[j];
    out->alpha[i] = s;
  }
  /* posterior KL basis: D·Minv·D = V S V^T */
  double *sig = (double *)malloc(sizeof(double) * r * r);
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < r; ++j) sig[i * r + j] = sqrt(m->lambda[i]) * out->Minv[i * r + j] * sqrt(m->lambda[j]);
  /* symmetrise against rounding before the eigen-solve */
  for (int i = 0; i < r; ++i)
    for (int j = i + 1; j < r; ++j) { double v = 0.5 * (sig[i * r + j] + sig[j * r + i]); sig[i * r + j] = sig[j * r + i] = v; }
  orc_sym_eigen(r, sig, out->S, out->V);
  free(sig); free(qtl); free(b); free(obs); free(Linv); free(x);
  return rc;
}

/* regularised least-squares coefficients of a displacement field `disp` [3N] in a (scaled) basis B [3N×r]:
 * (B^T B/σ² + I)^-1 B^T disp/σ², σ² = 1e-5 — Scalismo DiscreteLowRankGaussianProcess.coefficients
 * (SURVEY App. A.5 [SCALISMO-UNVERIFIED]) */
static int lsq_coefficients(int N3, int r, const double *B, const double *disp, double *c) {
  const double sigma2 = 1e-5;
  double *A = (double *)calloc((size_t)r * r, sizeof(double));
  for (int j = 0; j < r; ++j) c[j] = 0.0;
  for (int i = 0; i < N3; ++i) {
    const double *bi = B + (size_t)i * r;
    for (int j = 0; j < r; ++j) {
      double bj = bi[j] / sigma2; /* QtL = Q^T · (I/σ²) */
      c[j] += bj * disp[i];
      for (int k = 0; k < r; ++k) A[j * r + k] += bj * bi[k];
    }
  }
  for (int j = 0; j < r; ++j) A[j * r + j] += 1.0;
  for (int j = 0; j < r; ++j)
    for (int k = j + 1; k < r; ++k) { double v = 0.5 * (A[j * r + k] + A[k * r + j]); A[j * r + k] = A[k * r + j] = v; }
  int rc = cholesky(r, A);
  if (rc == 0) chol_solve(r, A, c);
  free(A);
  return rc;
}

static orc_posterior posterior_alloc(int K, int r) {
  orc_posterior p;
  p.K = K;
  p.corr_id = (int *)malloc(sizeof(int) * (K > 0 ? K : 1));
  p.corr_aux = (int *)malloc(sizeof(int) * (K > 0 ? K : 1));
  p.corr_pt = (double *)malloc(sizeof(double) * 3 * (K > 0 ? K : 1));
  p.keep = (unsigned char *)malloc((size_t)(K > 0 ? K : 1));
  p.alpha = (double *)malloc(sizeof(double) * r);
  p.M = (double *)malloc(sizeof(double) * r * r);
  p.Minv = (double *)malloc(sizeof(double) * r * r);
  p.V = (double *)malloc(sizeof(double) * r * r);
  p.S = (double *)malloc(sizeof(double) * r);
  return p;
}
static void posterior_free(orc_posterior *p) {
  free(p->corr_id); free(p->corr_aux); free(p->corr_pt); free(p->keep);
  free(p->alpha); free(p->M); free(p->Minv); free(p->V); free(p->S);
}
static int proposal_K(const orc_proposal_params *pp) { return pp->direction == 1 ? pp->n_target_pts : pp->n_model_ids; }

/* a8 — ref: NonRigidIcpProposal.scala:53-68.  z = the r standard normals posterior.sample() would draw (:55). */
ORC_API int orc_propose_from_posterior(const orc_model *m, const orc_proposal_params *pp, const orc_posterior *post,
                                       const double *theta, const double *z, double *theta_out) {
  const int r = m->r, N3 = 3 * m->N;
  /* sampled field at every reference vertex: μ + Q α + Φ V √S z  (posterior mean + Σ_j √S_j z_j (ΦV)_j) */
  double *w = (double *)malloc(sizeof(double) * r), *disp = (double *)malloc(sizeof(double) * N3);
  for (int i = 0; i < r; ++i) {
    double s = 0.0;
    for (int j = 0; j < r; ++j) s += post->V[i * r + j] * (sqrt(post->S[j]) * z[j]);
    w[i] = s;
  }
  for (int i = 0; i < N3; ++i) {
    const double *qi = m->Q + (size_t)i * r, *pi = m->phi + (size_t)i * r;
    double s = 0.0;
    for (int j = 0; j < r; ++j) s += qi[j] * post->alpha[j] + pi[j] * w[j];
    disp[i] = s; /* (field − μ): model.coefficients subtracts the model mean (:59) */
  }
  double *cnew = (double *)malloc(sizeof(double) * r);
  int rc = lsq_coefficients(N3, r, m->Q, disp, cnew);           /* :59 */
  memcpy(theta_out, theta, sizeof(double) * (10 + r));
  for (int j = 0; j < r; ++j) theta_out[10 + j] = theta[10 + j] + (cnew[j] - theta[10 + j]) * pp->step_length; /* :61-62 */
  free(w); free(disp); free(cnew);
  return rc;
}

ORC_API int orc_propose(const orc_model *m, const orc_mesh *tgt, const orc_proposal_params *pp, const double *theta,
                        const double *z, double *theta_out) {
  orc_posterior post = posterior_alloc(proposal_K(pp), m->r);
  int rc = orc_icp_posterior(m, tgt, pp, theta, &post);          /* :54 */
  if (rc == 0) rc = orc_propose_from_posterior(m, pp, &post, theta, z, theta_out);
  posterior_free(&post);
  return rc;
}

/* a9 — ref: NonRigidIcpProposal.scala:71-85 */
ORC_API int orc_log_transition_from_posterior(const orc_model *m, const orc_proposal_params *pp, const orc_posterior *post,
                                              const double *from, const double *to, double *out) {
  const int r = m->r, N3 = 3 * m->N;
  for (int i = 0; i < 10; ++i)
    if (from[i] != to[i]) { *out = -INFINITY; return 0; }        /* :72-74 */
  /* posterior model discretised on the reference: scaled basis Q_p = Φ V √S (:77) */
  double *Qp = (double *)malloc(sizeof(double) * (size_t)N3 * r), *disp = (double *)malloc(sizeof(double) * N3);
  double *ct = (double *)malloc(sizeof(double) * r), *beta = (double *)malloc(sizeof(double) * r);
  for (int j = 0; j < r; ++j) ct[j] = from[10 + j] + (to[10 + j] - from[10 + j]) / pp->step_length; /* :79 */
  for (int i = 0; i < N3; ++i) {
    const double *pi = m->phi + (size_t)i * r, *qi = m->Q + (size_t)i * r;
    for (int j = 0; j < r; ++j) {
      double s = 0.0;
      for (int k = 0; k < r; ++k) s += pi[k] * post->V[k * r + j];
      Qp[(size_t)i * r + j] = s * sqrt(post->S[j]);
    }
    double s = 0.0; /* instance(c̃) (:80) minus the posterior mean: Q (c̃ − α) */
    for (int j = 0; j < r; ++j) s += qi[j] * (ct[j] - post->alpha[j]);
    disp[i] = s;
  }
  int rc = lsq_coefficients(N3, r, Qp, disp, beta);              /* :82 */
  double nn = 0.0;
  for (int j = 0; j < r; ++j) nn += beta[j] * beta[j];
  *out = -0.5 * nn - 0.5 * r * log(2.0 * M_PI);                  /* :83 standard-normal logpdf, no log-det */
  free(Qp); free(disp); free(ct); free(beta);
  return rc;
}

ORC_API int orc_log_transition(const orc_model *m, const orc_mesh *tgt, const orc_proposal_params *pp,
                               const double *from, const double *to, double *out) {
  for (int i = 0; i < 10; ++i)
    if (from[i] != to[i]) { *out = -INFINITY; return 0; }
  orc_posterior post = posterior_alloc(proposal_K(pp), m->r);
  int rc = orc_icp_posterior(m, tgt, pp, from, &post);           /* :76 */
  if (rc == 0) rc = orc_log_transition_from_posterior(m, pp, &post, from, to, out);
  posterior_free(&post);
  return rc;
}

/* ------------------------------------------------------------------ next row 1: deterministic non-rigid ICP
 * ref: api/other/IcpBasedSurfaceFitting.scala:46-126.  For every sigma2 of the sequence (:36-40: 1, 0.1, 0.01) the
 * recursion (:55-104) runs for nbIterations = numIterations .. 0, i.e. numIterations + 1 times:
 *   instance of the current coefficients (:61), correspondences in ONE direction (:71-79; the random alternation of :66-69 is
 *   the caller's), model.posterior(correspondences, sigma2) with isotropic noise (:81), its MEAN (:82), model.coefficients of
 *   that mean (:84), step (:85).  Sample ids / target sample points come from Scalismo's UniformMeshSampler3D (:51-53) and are
 *   therefore inputs.  The regression works on the UNTRANSFORMED model with the world-space points, as the reference does (:81). */
typedef struct {
  int direction;            /* 0 = ModelSampling, 1 = TargetSampling */
  int n_model_ids;
  const int *model_ids;     /* pointIds (:53) */
  int n_target_pts;
  const double *target_pts; /* targetPointSamples (:51) */
  double step_length;       /* :32 */
} orc_fit_params;

ORC_API int orc_fit_deterministic(const orc_model *m, const orc_mesh *tgt, const orc_fit_params *fp, const double *theta_init,
                                  int n_iterations, int n_sigma, const double *sigma2_seq, double *theta_out) {
  const int r = m->r, N = m->N, N3 = 3 * N;
  const int K = fp->direction == 1 ? fp->n_target_pts : fp->n_model_ids;
  double *theta = theta_out;
  memcpy(theta, theta_init, sizeof(double) * (10 + r));
  double *x = (double *)malloc(sizeof(double) * N3), *M = (double *)malloc(sizeof(double) * r * r);
  double *b = (double *)malloc(sizeof(double) * r), *disp = (double *)malloc(sizeof(double) * N3), *cnew = (double *)malloc(sizeof(double) * r);
  int rc = 0;
  for (int si = 0; si < n_sigma && rc == 0; ++si) {
    const double w = 1.0 / sigma2_seq[si];
    for (int it = 0; it <= n_iterations && rc == 0; ++it) {
      orc_instance(m, theta, x);                                                  /* :61 */
      for (int i = 0; i < r; ++i) { b[i] = 0.0; for (int j = 0; j < r; ++j) M[i * r + j] = (i == j); }
      for (int k = 0; k < K; ++k) {
        int id; double tp[3];
        if (fp->direction == 1) {                                                 /* :76-78 */
          memcpy(tp, fp->target_pts + 3 * k, sizeof(tp));
          orc_nearest_vertex(1, tp, N, x, &id, NULL);
        } else {                                                                  /* :72-74 */
          id = fp->model_ids[k];
          orc_closest_point_on_surface(1, x + 3 * id, tgt->pts, tgt->T, tgt->tris, tp, NULL, NULL);
        }
        const double *Qi = m->Q + (size_t)3 * id * r;
        double e[3];
        for (int d = 0; d < 3; ++d) e[d] = (tp[d] - m->ref[3 * id + d]) - m->mean[3 * id + d];
        for (int i = 0; i < r; ++i) {
          for (int j = 0; j < r; ++j) M[i * r + j] += w * ((Qi[i] * Qi[j] + Qi[r + i] * Qi[r + j]) + Qi[2 * r + i] * Qi[2 * r + j]);
          b[i] += w * ((Qi[i] * e[0] + Qi[r + i] * e[1]) + Qi[2 * r + i] * e[2]);
        }
      }
      rc = cholesky(r, M);
      if (rc) break;
      chol_solve(r, M, b);                                                        /* alpha = M^-1 b: posterior mean coefficients */
      for (int i = 0; i < N3; ++i) {                                              /* posterior.mean − model mean (:82,84) */
        const double *qi = m->Q + (size_t)i * r;
        double sacc = 0.0;
        for (int j = 0; j < r; ++j) sacc += qi[j] * b[j];
        disp[i] = sacc;
      }
      rc = lsq_coefficients(N3, r, m->Q, disp, cnew);                             /* :84 */
      for (int j = 0; j < r; ++j) theta[10 + j] = theta[10 + j] + (cnew[j] - theta[10 + j]) * fp->step_length;  /* :85 */
    }
  }
  free(x); free(M); free(b); free(disp); free(cnew);
  return rc;
}

/* ------------------------------------------------------------------ next row 3: posterior variability maps
 * ref: apps/util/PosteriorVariability.scala:30-73.  meshes = transformedMesh of every logged sample.
 *   mode 0 (computeDistanceMapFromMeshesTotal :30-49): trace of the per-vertex sample covariance, mean = (Σ s)·(1/n),
 *          cov = (Σ (s−mean)(s−mean)ᵀ)·(1/(n−1));
 *   mode 1 (computeDistanceMapFromMeshesNormal, sumNormals = false :51-72): (Σ (n·(s−mean))²)·(1/(n−1)) with n = unit vertex
 *          normal of the mesh of theta_ref;
 *   mode 2 (sumNormals = true): n = (Σ unit vertex normals of the samples)·(1/n), NOT renormalised (:63-65). */
ORC_API int orc_posterior_variability(const orc_model *m, int n_samples, const double *thetas, int mode, const double *theta_ref,
                                      double *out) {
  const int N = m->N, P = 10 + m->r, S = n_samples;
  if (S < 2) return 1;
  double *X = (double *)malloc(sizeof(double) * (size_t)S * 3 * N), *nrm = (double *)calloc((size_t)3 * N, sizeof(double));
  for (int s = 0; s < S; ++s) orc_instance(m, thetas + (size_t)s * P, X + (size_t)s * 3 * N);
  if (mode == 1) {
    double *xr = (double *)malloc(sizeof(double) * 3 * N);
    orc_instance(m, theta_ref, xr);
    orc_vertex_normals(m, xr, nrm);
    free(xr);
  } else if (mode == 2) {
    double *tmp = (double *)malloc(sizeof(double) * 3 * N);
    for (int s = 0; s < S; ++s) {
      orc_vertex_normals(m, X + (size_t)s * 3 * N, tmp);
      for (int i = 0; i < 3 * N; ++i) nrm[i] += tmp[i];
    }
    for (int i = 0; i < 3 * N; ++i) nrm[i] *= 1.0 / S;
    free(tmp);
  }
  for (int i = 0; i < N; ++i) {
    double mean[3] = {0, 0, 0};
    for (int s = 0; s < S; ++s)
      for (int d = 0; d < 3; ++d) mean[d] += X[(size_t)s * 3 * N + 3 * i + d];
    for (int d = 0; d < 3; ++d) mean[d] *= 1.0 / S;
    if (mode == 0) {
      double c[3] = {0, 0, 0};
      for (int s = 0; s < S; ++s)
        for (int d = 0; d < 3; ++d) { double v = X[(size_t)s * 3 * N + 3 * i + d] - mean[d]; c[d] += v * v; }
      out[i] = (c[0] * (1.0 / (S - 1)) + c[1] * (1.0 / (S - 1))) + c[2] * (1.0 / (S - 1));
    } else {
      double acc = 0.0;
      for (int s = 0; s < S; ++s) {
        const double *x = X + (size_t)s * 3 * N + 3 * i;
        double p = (nrm[3 * i] * (x[0] - mean[0]) + nrm[3 * i + 1] * (x[1] - mean[1])) + nrm[3 * i + 2] * (x[2] - mean[2]);
        acc += p * p;
      }
      out[i] = acc * (1.0 / (S - 1));
    }
  }
  free(X); free(nrm);
  return 0;
}

/* ------------------------------------------------------------------ a11-a14: evaluators */

/* Breeze Gaussian(mu, sigma).logPdf / Exponential(rate).logPdf */
static double gauss_logpdf(double x, double mu, double sigma) {
  double d = (x - mu) / sigma;
  return -d * d / 2.0 - (log(sqrt(2.0 * M_PI)) + log(sigma));
}
static double expo_logpdf(double x, double rate) { return -rate * x + log(rate); }

typedef struct {
  int kind;            /* 0 IndependentPointDistance, 1 Hausdorff, 2 CollectiveAverageHausdorffBoundaryAware */
  int mode;            /* 0 ModelToTarget, 1 TargetToModel, 2 Symmetric   (ref: evaluators/EvaluationModeType.scala:20-26) */
  int n_model_ids;     /* K_e model-side ids = 0 until K_e */
  int n_target_pts;
  const double *target_pts;
  double p0, p1, p2;   /* kind 0: Gaussian(p0=mean, p1=sigma); kind 1: Exponential(p0=rate);
                          kind 2: Gaussian(p0=mean, p1=sigma) for the average, Exponential(p2=rate) for the max */
} orc_evaluator_params;

/* ref: evaluators/ModelPriorEvaluator.scala:24-31 — MVN(0, I_r).logpdf(c) */
ORC_API double orc_prior_log_value(int r, const double *theta) {
  double nn = 0.0;
  for (int j = 0; j < r; ++j) nn += theta[10 + j] * theta[10 + j];
  return -0.5 * nn - 0.5 * r * log(2.0 * M_PI);
}

/* distances of K points to a surface, the K searches as ONE batch of independent queries (see orc_icp_posterior):
 * dist[k] = |closestPointOnSurface(p_k) − p_k|, cp (optional) = the closest points */
static void point_surface_distances(int K, const double *p, const double *pts, int T, const int *tris, double *dist, double *cp_out) {
  double *cp = cp_out ? cp_out : (double *)malloc(sizeof(double) * 3 * (K > 0 ? K : 1));
  orc_closest_point_on_surface(K, p, pts, T, tris, cp, NULL, NULL);
  for (int k = 0; k < K; ++k) {
    double d[3];
    sub3(cp + 3 * k, p + 3 * k, d);
    dist[k] = sqrt(dot3(d, d));
  }
  if (!cp_out) free(cp);
}

ORC_API int orc_evaluator_log_value(const orc_model *m, const orc_mesh *tgt, const orc_evaluator_params *ep,
                                    const double *theta, double *out) {
  const int N = m->N;
  double *x = (double *)malloc(sizeof(double) * 3 * N);
  orc_instance(m, theta, x);
  int rc = 0;
  const int Kmax = ep->kind == 1 ? (N > tgt->M ? N : tgt->M) : (ep->n_model_ids > ep->n_target_pts ? ep->n_model_ids : ep->n_target_pts);
  double *dist = (double *)malloc(sizeof(double) * (Kmax > 0 ? Kmax : 1));
  if (ep->kind == 0) {
    /* ref: evaluators/IndependentPointDistanceEvaluator.scala:40-66 */
    double m2t = 0.0, t2m = 0.0;
    if (ep->mode == 0 || ep->mode == 2) {
      point_surface_distances(ep->n_model_ids, x, tgt->pts, tgt->T, tgt->tris, dist, NULL); /* :41 ids 0 until K_e of the full sample mesh, :43 */
      for (int k = 0; k < ep->n_model_ids; ++k) m2t += gauss_logpdf(dist[k], ep->p0, ep->p1);
    }
    if (ep->mode == 1 || ep->mode == 2) {
      point_surface_distances(ep->n_target_pts, ep->target_pts, x, m->T, m->tris, dist, NULL); /* :50-52 */
      for (int k = 0; k < ep->n_target_pts; ++k) t2m += gauss_logpdf(dist[k], ep->p0, ep->p1);
    }
    *out = ep->mode == 0 ? m2t : ep->mode == 1 ? t2m : 0.5 * m2t + 0.5 * t2m; /* :60-64 */
  } else if (ep->kind == 1) {
    /* ref: evaluators/HausdorffDistanceEvaluator.scala:31-35; MeshMetrics.hausdorffDistance = max over both
     * directions of vertex -> closest-surface-point distance (SURVEY App. B7 [SCALISMO-UNVERIFIED]) */
    double hd = 0.0;
    point_surface_distances(N, x, tgt->pts, tgt->T, tgt->tris, dist, NULL);
    for (int i = 0; i < N; ++i)
      if (dist[i] > hd) hd = dist[i];
    point_surface_distances(tgt->M, tgt->pts, x, m->T, m->tris, dist, NULL);
    for (int i = 0; i < tgt->M; ++i)
      if (dist[i] > hd) hd = dist[i];
    *out = expo_logpdf(hd, ep->p0);
  } else {
    /* ref: evaluators/CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator.scala:40-78 */
    double avg[2] = {0, 0}, mx[2] = {0, 0};
    double *cps = (double *)malloc(sizeof(double) * 3 * (Kmax > 0 ? Kmax : 1));
    int *vids = (int *)malloc(sizeof(int) * (Kmax > 0 ? Kmax : 1));
    if (ep->mode == 0 || ep->mode == 2) {                        /* distModelToTarget :40-52 */
      double sum = 0.0, mxx = -INFINITY;
      int cnt = 0;
      point_surface_distances(ep->n_model_ids, x, tgt->pts, tgt->T, tgt->tris, dist, cps); /* :45 */
      orc_nearest_vertex(ep->n_model_ids, cps, tgt->M, tgt->pts, vids, NULL);              /* :46 */
      for (int k = 0; k < ep->n_model_ids; ++k) {
        if (tgt->boundary[vids[k]]) continue;                                              /* :47 */
        sum += dist[k]; cnt++;
        if (dist[k] > mxx) mxx = dist[k];
      }
      avg[0] = sum / cnt; mx[0] = mxx;                           /* :51 (empty list: NaN / exception in the reference) */
      if (cnt == 0) rc = -2;
    }
    if (ep->mode == 1 || ep->mode == 2) {                        /* distTargetToModel :54-65 */
      double sum = 0.0, mxx = -INFINITY;
      int cnt = 0;
      point_surface_distances(ep->n_target_pts, ep->target_pts, x, m->T, m->tris, dist, cps); /* :57 */
      orc_nearest_vertex(ep->n_target_pts, cps, N, x, vids, NULL);                            /* :58 id in modelSample */
      for (int k = 0; k < ep->n_target_pts; ++k) {
        /* :59 (sic) tests the TARGET mesh's boundary flag with a model-sample vertex id;
         * an id beyond the target's vertex count is treated as "not on boundary". */
        if (vids[k] < tgt->M && tgt->boundary[vids[k]]) continue;
        sum += dist[k]; cnt++;
        if (dist[k] > mxx) mxx = dist[k];
      }
      avg[1] = sum / cnt; mx[1] = mxx;
      if (cnt == 0) rc = -2;
    }
    double a, h;
    if (ep->mode == 0) { a = avg[0]; h = mx[0]; }
    else if (ep->mode == 1) { a = avg[1]; h = mx[1]; }
    else { a = 0.5 * avg[0] + 0.5 * avg[1]; h = fmax(mx[0], mx[1]); } /* :71-75 */
    *out = gauss_logpdf(a, ep->p0, ep->p1) + expo_logpdf(h, ep->p2);  /* :77 */
    free(cps); free(vids);
  }
  free(dist);
  free(x);
  return rc;
}

/* ------------------------------------------------------------------ a16: the caller — Metropolis–Hastings harness
 * Mirrors (for measurement and golden chains only) Scalismo MetropolisHastings.next + MixtureProposal as used
 * by ref: api/sampling/SamplingRegistration.scala:52-85 with the mixture of ref:
 * apps/femur/IcpProposalRegistration.scala:70-72 (SURVEY §8 a16 / App. B1-B2 [SCALISMO-UNVERIFIED]).
 * Randomness: counter-based generator shared bit-for-bit with the host harness (icp-proposal_amd/host). */

static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
/* uniform in (0,1): stream (seed, step, lane) */
ORC_API double orc_rng_uniform(uint64_t seed, uint64_t step, uint64_t lane) {
  uint64_t h = splitmix64(splitmix64(splitmix64(seed) ^ (step * 0xD1342543DE82EF95ull)) ^ (lane * 0x2545F4914F6CDD1Dull));
  return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
ORC_API double orc_rng_normal(uint64_t seed, uint64_t step, uint64_t lane) {
  double u1 = orc_rng_uniform(seed, step, 2 * lane + 1000), u2 = orc_rng_uniform(seed, step, 2 * lane + 1001);
  return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

typedef struct {
  int n_icp;                       /* ICP components (0..2) */
  orc_proposal_params icp[2];
  double icp_weight[2];            /* inner mixture weights (0.5/0.5, ref: MixedProposalDistributions.scala:50) */
  double w_icp, w_rw;              /* outer mixture (0.9/0.1, ref: IcpProposalRegistration.scala:72) */
  double rw_sigma;                 /* RandomShapeUpdateProposal stdev (ref: RandomShapeUpdateProposal.scala:25-35) */
  orc_evaluator_params eval;       /* likelihood; the prior evaluator is always multiplied in (ProductEvaluators.scala:38-55) */
  /* mixedRandomPoseProposal (ref: MixedProposalDistributions.scala:29-39): six equally weighted one-dimensional Gaussian walks,
   * in the reference's order Yaw, Pitch, Roll, X, Y, Z; the outer mixture is then (pose, ICP, shape walk) as in ref:
   * apps/bfm/BfmFittingPartial.scala:70 (0.4 / 0.55 / 0.05).  w_pose = 0: no pose proposals. */
  double w_pose;
  double pose_rot_sigma[3];        /* rotYaw, rotPitch, rotRoll */
  double pose_trans_sigma[3];      /* transX, transY, transZ */
} orc_chain_config;

static double logsumexp_mix(int n, const double *w, const double *t) {
  double mx = -INFINITY;
  for (int i = 0; i < n; ++i)
    if (t[i] > mx) mx = t[i];
  if (mx == -INFINITY) return -INFINITY;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += w[i] * exp(t[i] - mx);
  return log(s) + mx;
}

static double rw_log_transition(int r, double sigma, const double *from, const double *to) {
  /* ref: RandomShapeUpdateProposal.scala:37-45: MVN(0, σ²I).logpdf(to − from) */
  for (int i = 0; i < 10; ++i)
    if (from[i] != to[i]) return -INFINITY;
  double nn = 0.0;
  for (int j = 0; j < r; ++j) { double d = to[10 + j] - from[10 + j]; nn += d * d; }
  return -0.5 * nn / (sigma * sigma) - 0.5 * (r * log(2.0 * M_PI) + r * log(sigma * sigma));
}

/* The pose walks of ref: api/sampling/proposals/PoseProposals.scala:31-90 in the order of
 * ref: MixedProposalDistributions.scala:31-36: Yaw, Pitch, Roll, X, Y, Z.  allParameters = [s | t(3) | rotation._1, _2, _3 |
 * centre(3) | c] (ref: ModelFittingParameters.scala:28-36,64), and RollAxis -> _1, PitchAxis -> _2, YawAxis -> _3
 * (ref: PoseProposals.scala:39-41): component a perturbs theta[pose_param_index[a]]. */
static const int pose_param_index[6] = {6, 5, 4, 1, 2, 3};

/* ref: PoseProposals.scala:46-60 (rotation), :77-88 (translation).  -inf only when a parameter OUTSIDE the proposal's own
 * group differs: the comparison of :47 / :78 is made after the WHOLE rotation triple / translation vector of `to` has been reset to
 * `from`'s.  Inside the group only the component's own axis enters the residual, so e.g. the Yaw component evaluated on a Roll
 * move returns logPdf(0), finite.  breeze Gaussian(0, sd).logPdf(x) = -(x/sd)^2/2 - (log sqrt(2 pi) + log sd). */
ORC_API double orc_pose_log_transition(int P, int component, double sd, const double *from, const double *to) {
  const int idx = pose_param_index[component];
  const int g0 = component < 3 ? 4 : 1, g1 = g0 + 3;
  for (int i = 0; i < P; ++i)
    if ((i < g0 || i >= g1) && from[i] != to[i]) return -INFINITY;
  const double d = (to[idx] - from[idx]) / sd;
  return -d * d / 2.0 - (log(sqrt(2.0 * M_PI)) + log(sd));
}

/* MixtureProposal.logTransitionProbability of mixedRandomPoseProposal: six components of weight 0.5 each (normalised: 1/6) */
ORC_API double orc_pose_mixture_log_transition(int P, const double rot_sigma[3], const double trans_sigma[3], const double *from,
                                               const double *to) {
  double t[6], w[6], ws = 0.0;
  for (int a = 0; a < 6; ++a) t[a] = orc_pose_log_transition(P, a, a < 3 ? rot_sigma[a] : trans_sigma[a - 3], from, to);
  for (int a = 0; a < 6; ++a) ws += 0.5;
  for (int a = 0; a < 6; ++a) w[a] = 0.5 / ws;
  return logsumexp_mix(6, w, t);
}

/* the outer mixture's components in the order the reference builds them (pose, ICP, shape walk; absent ones skipped) with their
 * normalised weights; returns their number.  kind: 0 pose, 1 ICP, 2 shape walk */
static int chain_outer_components(const orc_chain_config *cfg, int *kind, double *w) {
  int n = 0;
  double raw[3], wsum = 0.0;
  if (cfg->w_pose > 0) { kind[n] = 0; raw[n++] = cfg->w_pose; }
  if (cfg->n_icp > 0 && cfg->w_icp > 0) { kind[n] = 1; raw[n++] = cfg->w_icp; }
  if (cfg->w_rw > 0) { kind[n] = 2; raw[n++] = cfg->w_rw; }
  for (int i = 0; i < n; ++i) wsum += raw[i];
  for (int i = 0; i < n; ++i) w[i] = raw[i] / wsum;
  return n;
}

/* log-sum-exp over ALL mixture components of the transition density from -> to, given the ICP posteriors of
 * `from` (Scalismo MixtureProposal.logTransitionProbability) */
static int chain_log_transition(const orc_model *m, const orc_chain_config *cfg, const orc_posterior *post_from,
                                const double *from, const double *to, double *out) {
  double outer_t[3], outer_w[3];
  int kind[3], rc = 0;
  const int n_outer = chain_outer_components(cfg, kind, outer_w);
  for (int o = 0; o < n_outer; ++o) {
    if (kind[o] == 0) {
      outer_t[o] = orc_pose_mixture_log_transition(10 + m->r, cfg->pose_rot_sigma, cfg->pose_trans_sigma, from, to);
    } else if (kind[o] == 1) {
      double t[2], w[2], ws = 0;
      for (int i = 0; i < cfg->n_icp; ++i) ws += cfg->icp_weight[i];
      for (int i = 0; i < cfg->n_icp; ++i) {
        w[i] = cfg->icp_weight[i] / ws;
        rc |= orc_log_transition_from_posterior(m, &cfg->icp[i], &post_from[i], from, to, &t[i]);
      }
      outer_t[o] = logsumexp_mix(cfg->n_icp, w, t);
    } else {
      /* mixedRandomShapeProposal is itself a 1-component mixture (weight normalised to 1) */
      outer_t[o] = rw_log_transition(m->r, cfg->rw_sigma, from, to);
    }
  }
  *out = logsumexp_mix(n_outer, outer_w, outer_t);
  return rc;
}

/* the whole chain mixture's transition density from -> to (posteriors of `from` computed here): for tests of the harness */
ORC_API int orc_chain_log_transition(const orc_model *m, const orc_mesh *tgt, const orc_chain_config *cfg, const double *from,
                                     const double *to, double *out) {
  orc_posterior pf[2];
  int rc = 0;
  for (int i = 0; i < cfg->n_icp; ++i) {
    pf[i] = posterior_alloc(proposal_K(&cfg->icp[i]), m->r);
    rc |= orc_icp_posterior(m, tgt, &cfg->icp[i], from, &pf[i]);
  }
  if (rc == 0) rc = chain_log_transition(m, cfg, pf, from, to, out);
  for (int i = 0; i < cfg->n_icp; ++i) posterior_free(&pf[i]);
  return rc;
}

/* first component whose cumulative normalised weight reaches u (Scalismo MixtureProposal) */
static int mixture_pick(int n, const double *w_normalised, double u) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) {
    acc += w_normalised[i];
    if (acc >= u) return i;
  }
  return n - 1;
}

/* Runs n_steps MH steps from theta0.  Outputs per step: accepted flag, component index (0/1 = ICP component,
 * 2 = random walk, 3..8 = the pose walks Yaw, Pitch, Roll, X, Y, Z), log posterior value of the state after the step, and the
 * state itself.
 * The ICP posteriors and the likelihood of the CURRENT state are carried over between steps, which is what the
 * reference's Memoize caches achieve (ref: NonRigidIcpProposal.scala:49, evaluators/EvaluationCaching.scala:32).
 * RNG lanes per step: uniform lane 0 outer mixture draw, 1 inner mixture draw, 2 accept draw; normal lane j = z_j (a pose walk
 * perturbs its parameter by sigma * z_0). */
ORC_API int orc_run_chain(const orc_model *m, const orc_mesh *tgt, const orc_chain_config *cfg, const double *theta0,
                          uint64_t seed, int n_steps, unsigned char *accepted, int *component, double *logp,
                          double *states /* [n_steps*(10+r)] */) {
  const int r = m->r, P = 10 + r;
  double *cur = (double *)malloc(sizeof(double) * P), *prop = (double *)malloc(sizeof(double) * P);
  double *z = (double *)malloc(sizeof(double) * r);
  orc_posterior pc[2], pp[2];
  memcpy(cur, theta0, sizeof(double) * P);
  double cur_like, cur_p;
  int rc = orc_evaluator_log_value(m, tgt, &cfg->eval, cur, &cur_like);
  cur_p = orc_prior_log_value(r, cur) + cur_like;
  for (int i = 0; i < cfg->n_icp; ++i) {
    pc[i] = posterior_alloc(proposal_K(&cfg->icp[i]), r);
    pp[i] = posterior_alloc(proposal_K(&cfg->icp[i]), r);
    rc |= orc_icp_posterior(m, tgt, &cfg->icp[i], cur, &pc[i]);
  }
  int kind[3];
  double outer_w[3];
  const int n_outer = chain_outer_components(cfg, kind, outer_w);
  for (int s = 0; s < n_steps && rc == 0; ++s) {
    const double u = orc_rng_uniform(seed, (uint64_t)s, 0);
    int comp;
    for (int j = 0; j < r; ++j) z[j] = orc_rng_normal(seed, (uint64_t)s, (uint64_t)j);
    const int outer = kind[mixture_pick(n_outer, outer_w, u)];
    if (outer == 1) {
      double w[2], ws = 0, u2 = orc_rng_uniform(seed, (uint64_t)s, 1);
      for (int i = 0; i < cfg->n_icp; ++i) ws += cfg->icp_weight[i];
      for (int i = 0; i < cfg->n_icp; ++i) w[i] = cfg->icp_weight[i] / ws;
      comp = mixture_pick(cfg->n_icp, w, u2);
      rc |= orc_propose_from_posterior(m, &cfg->icp[comp], &pc[comp], cur, z, prop);
    } else if (outer == 2) {
      comp = 2;
      memcpy(prop, cur, sizeof(double) * P);
      for (int j = 0; j < r; ++j) prop[10 + j] = cur[10 + j] + cfg->rw_sigma * z[j];
    } else { /* ref: PoseProposals.scala:36-44, :71-76 */
      double w[6], ws = 0, u2 = orc_rng_uniform(seed, (uint64_t)s, 1);
      for (int a = 0; a < 6; ++a) ws += 0.5;
      for (int a = 0; a < 6; ++a) w[a] = 0.5 / ws;
      const int a = mixture_pick(6, w, u2);
      comp = 3 + a;
      memcpy(prop, cur, sizeof(double) * P);
      prop[pose_param_index[a]] = cur[pose_param_index[a]] + (a < 3 ? cfg->pose_rot_sigma[a] : cfg->pose_trans_sigma[a - 3]) * z[0];
    }
    double prop_like, fw, bw;
    rc |= orc_evaluator_log_value(m, tgt, &cfg->eval, prop, &prop_like);
    double prop_p = orc_prior_log_value(r, prop) + prop_like;
    for (int i = 0; i < cfg->n_icp; ++i) rc |= orc_icp_posterior(m, tgt, &cfg->icp[i], prop, &pp[i]);
    rc |= chain_log_transition(m, cfg, pc, cur, prop, &fw);
    rc |= chain_log_transition(m, cfg, pp, prop, cur, &bw);
    double t = (fw == -INFINITY && bw == -INFINITY) ? 0.0 : fw - bw;
    double a = prop_p - cur_p - t;
    int acc = (a > 0.0) || (orc_rng_uniform(seed, (uint64_t)s, 2) < exp(a));
    if (acc) {
      memcpy(cur, prop, sizeof(double) * P);
      cur_p = prop_p;
      for (int i = 0; i < cfg->n_icp; ++i) { orc_posterior tmp = pc[i]; pc[i] = pp[i]; pp[i] = tmp; }
    }
    accepted[s] = (unsigned char)acc;
    component[s] = comp;
    logp[s] = cur_p;
    memcpy(states + (size_t)s * P, cur, sizeof(double) * P);
  }
  for (int i = 0; i < cfg->n_icp; ++i) { posterior_free(&pc[i]); posterior_free(&pp[i]); }
  free(cur); free(prop); free(z);
  return rc;
}
