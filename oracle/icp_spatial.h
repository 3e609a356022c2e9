/*
 * icp_spatial.h — search back ends of the CPU oracle's baseline variants (TEST / BENCHMARK INFRASTRUCTURE ONLY).
 *
 * BASELINE.md §3 defines two CPU baselines beside the plain restatement in icp_oracle.c:
 *   B1  "reference-shaped": one thread per chain, KD-tree for findClosestPoint and a bounding-volume hierarchy for
 *       closestPointOnSurface, the structures of the CURRENT model instance rebuilt for every new state — what Scalismo's
 *       lazily built spatial indices amount to for a mesh that changes every step (call sites: NonRigidIcpProposal.scala:97-98,118;
 *       IndependentPointDistanceEvaluator.scala:43,51) [SCALISMO-UNVERIFIED: the tree types are Scalismo internals];
 *   B2  the restatement's brute-force scans spread over all host cores with OpenMP: the independent queries of a batch (the K
 *       correspondences of a posterior, the K_e points of an evaluator) go to different threads, each the sequential scan.
 * Both return EXACTLY what the brute-force scans of icp_oracle.c return — same squared-distance expressions, ties to the
 * lowest index (tests/test_oracle.py::test_search_backends_agree) — so every other line of the oracle is shared.
 */
#ifndef ICP_SPATIAL_H
#define ICP_SPATIAL_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_SEARCH_BRUTE = 0, ORC_SEARCH_TREES = 1 /* B1 */, ORC_SEARCH_BRUTE_OMP = 2 /* B2 */ };

/* per-thread: which back end orc_nearest_vertex / orc_closest_point_on_surface use from now on (n_threads: B2 only, 0 = all) */
void orc_set_search_backend(int backend, int n_threads);
int orc_get_search_backend(void);
int orc_get_search_threads(void);
/* statistics of the calling thread's tree cache: number of KD-trees / hierarchies built so far */
void orc_search_stats(long *kd_builds, long *bvh_builds);

/* the two queries, one point each; `closest_point_triangle` is the oracle's own kernel */
typedef void (*orc_tri_kernel)(const double *p, const double *a, const double *b, const double *c, double *o);
int spatial_nearest_vertex(const double *q, int M, const double *pts, double *d2_out);
int spatial_closest_on_surface(const double *q, const double *pts, int T, const int *tris, orc_tri_kernel kernel, double *cp_out,
                               double *d2_out);

#ifdef __cplusplus
}
#endif
#endif
