/*
 * icp_spatial.c — KD-tree / bounding-volume-hierarchy / OpenMP search back ends of the CPU baselines B1 and B2
 * (see icp_spatial.h; TEST / BENCHMARK INFRASTRUCTURE ONLY, never linked into the product).
 *
 * Exactness.  The brute-force scans define the result: minimum of d² = (dx·dx + dy·dy) + dz·dz over all elements, the lowest
 * index among equal values.  A subtree is skipped only when a LOWER BOUND of every d² inside it exceeds the best value so
 * far (strictly), and the bound is evaluated with the same expression on per-axis distances to the node's box: floating-point
 * multiplication and addition are monotone, so bound <= d² holds in floating point, not just in exact arithmetic.  For
 * triangles the computed closest point may sit an ulp outside the triangle's box, so that bound is relaxed by 1e-12 relative
 * (only more nodes are visited).  Candidates are compared lexicographically by (d², index).
 */
#include "icp_spatial.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread int t_backend = ORC_SEARCH_BRUTE;
static __thread int t_threads = 0;
static __thread long t_kd_builds = 0, t_bvh_builds = 0;

void orc_set_search_backend(int backend, int n_threads) { t_backend = backend; t_threads = n_threads; }
int orc_get_search_backend(void) { return t_backend; }
int orc_get_search_threads(void) { return t_threads; }
void orc_search_stats(long *kd, long *bvh) { if (kd) *kd = t_kd_builds; if (bvh) *bvh = t_bvh_builds; }

static inline double d2_of(const double *a, const double *b) {
  const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}
static inline double box_bound(const double *q, const double *lo, const double *hi) {
  double d[3];
  for (int a = 0; a < 3; ++a) d[a] = q[a] < lo[a] ? lo[a] - q[a] : (q[a] > hi[a] ? q[a] - hi[a] : 0.0);
  return (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
}

/* ---------------------------------------------------------------- tree over items with a representative point + a box */
typedef struct {
  double lo[3], hi[3];
  int left, right;   /* children (internal) */
  int start, count;  /* items (leaf: count > 0) */
} node_t;

typedef struct {
  int n_items, n_nodes, cap_nodes;
  node_t *nodes;
  int *items;        /* permutation */
  double *rep;       /* [n_items*3] representative point (vertex itself / triangle centroid) */
  double *ilo, *ihi; /* [n_items*3] item boxes */
  /* identity of the indexed data */
  const double *pts;
  const int *tris;
  int M, T;
  uint64_t print;
  uint64_t stamp;
} tree_t;

static void tree_free(tree_t *t) {
  free(t->nodes); free(t->items); free(t->rep); free(t->ilo); free(t->ihi);
  memset(t, 0, sizeof(*t));
}

static __thread int cmp_axis;
static __thread const double *cmp_rep;
static int cmp_items(const void *a, const void *b) {
  const double x = cmp_rep[3 * *(const int *)a + cmp_axis], y = cmp_rep[3 * *(const int *)b + cmp_axis];
  if (x < y) return -1;
  if (x > y) return 1;
  return *(const int *)a - *(const int *)b;
}

static int build_node(tree_t *t, int start, int count, int leaf_size) {
  const int id = t->n_nodes++;
  node_t *nd = &t->nodes[id];
  for (int a = 0; a < 3; ++a) { nd->lo[a] = INFINITY; nd->hi[a] = -INFINITY; }
  double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = start; i < start + count; ++i) {
    const int it = t->items[i];
    for (int a = 0; a < 3; ++a) {
      if (t->ilo[3 * it + a] < nd->lo[a]) nd->lo[a] = t->ilo[3 * it + a];
      if (t->ihi[3 * it + a] > nd->hi[a]) nd->hi[a] = t->ihi[3 * it + a];
      if (t->rep[3 * it + a] < clo[a]) clo[a] = t->rep[3 * it + a];
      if (t->rep[3 * it + a] > chi[a]) chi[a] = t->rep[3 * it + a];
    }
  }
  nd->left = nd->right = -1;
  nd->start = start;
  nd->count = count;
  if (count <= leaf_size) return id;
  int axis = 0;
  for (int a = 1; a < 3; ++a)
    if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
  if (!(chi[axis] > clo[axis])) return id;  /* all representatives coincide (or are not finite): keep as a leaf */
  cmp_axis = axis;
  cmp_rep = t->rep;
  qsort(t->items + start, (size_t)count, sizeof(int), cmp_items);
  const int half = count / 2;
  const int l = build_node(t, start, half, leaf_size);
  const int r = build_node(t, start + half, count - half, leaf_size);
  nd = &t->nodes[id];  /* (the array does not move: allocated up front) */
  nd->left = l;
  nd->right = r;
  nd->count = 0;
  return id;
}

static void tree_build(tree_t *t, int n_items, int leaf_size) {
  t->n_items = n_items;
  t->cap_nodes = 2 * (n_items > 0 ? n_items : 1) + 2;
  t->nodes = (node_t *)malloc(sizeof(node_t) * (size_t)t->cap_nodes);
  t->items = (int *)malloc(sizeof(int) * (size_t)(n_items > 0 ? n_items : 1));
  for (int i = 0; i < n_items; ++i) t->items[i] = i;
  t->n_nodes = 0;
  if (n_items > 0) build_node(t, 0, n_items, leaf_size);
}

static uint64_t mix64(uint64_t h, uint64_t v) {
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  return h;
}
/* fingerprint of a coordinate array: 96 samples spread over it (a new model instance moves every vertex) */
static uint64_t fingerprint(const double *pts, int M) {
  uint64_t h = (uint64_t)M;
  if (M <= 0) return h;
  for (int s = 0; s < 96; ++s) {
    const size_t i = ((size_t)s * (size_t)(3 * M - 1)) / 95;
    uint64_t bits;
    memcpy(&bits, &pts[i], 8);
    h = mix64(h, bits);
  }
  return h;
}

#define CACHE_SLOTS 4
static __thread tree_t t_kd[CACHE_SLOTS], t_bvh[CACHE_SLOTS];
static __thread uint64_t t_clock = 0;

static tree_t *cache_find(tree_t *slots, const double *pts, const int *tris, int M, int T, uint64_t print) {
  for (int i = 0; i < CACHE_SLOTS; ++i)
    if (slots[i].nodes && slots[i].pts == pts && slots[i].tris == tris && slots[i].M == M && slots[i].T == T && slots[i].print == print) {
      slots[i].stamp = ++t_clock;
      return &slots[i];
    }
  return NULL;
}
static tree_t *cache_victim(tree_t *slots) {
  tree_t *v = &slots[0];
  for (int i = 0; i < CACHE_SLOTS; ++i) {
    if (!slots[i].nodes) { v = &slots[i]; break; }
    if (slots[i].stamp < v->stamp) v = &slots[i];
  }
  if (v->nodes) tree_free(v);
  v->stamp = ++t_clock;
  return v;
}

/* ---------------------------------------------------------------- KD-tree over vertices */
static tree_t *kd_for(const double *pts, int M) {
  const uint64_t fp = fingerprint(pts, M);
  tree_t *t = cache_find(t_kd, pts, NULL, M, 0, fp);
  if (t) return t;
  t = cache_victim(t_kd);
  t->pts = pts; t->tris = NULL; t->M = M; t->T = 0; t->print = fp;
  t->rep = (double *)malloc(sizeof(double) * 3 * (size_t)(M > 0 ? M : 1));
  t->ilo = t->rep;  /* a vertex is its own box */
  t->ihi = t->rep;
  memcpy(t->rep, pts, sizeof(double) * 3 * (size_t)M);
  tree_build(t, M, 8);
  t->ilo = t->ihi = NULL;  /* (aliases of rep: not freed twice) */
  ++t_kd_builds;
  return t;
}

int spatial_nearest_vertex(const double *q, int M, const double *pts, double *d2_out) {
  tree_t *t = kd_for(pts, M);
  double best = INFINITY;
  int bi = -1;
  if (t->n_nodes > 0) {
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
      const node_t *nd = &t->nodes[stack[--sp]];
      if (box_bound(q, nd->lo, nd->hi) > best) continue;
      if (nd->left < 0) {
        for (int i = nd->start; i < nd->start + nd->count; ++i) {
          const int v = t->items[i];
          const double dd = d2_of(q, pts + 3 * v);
          if (dd < best || (dd == best && v < bi)) { best = dd; bi = v; }
        }
      } else {
        const double bl = box_bound(q, t->nodes[nd->left].lo, t->nodes[nd->left].hi);
        const double br = box_bound(q, t->nodes[nd->right].lo, t->nodes[nd->right].hi);
        if (bl <= br) { stack[sp++] = nd->right; stack[sp++] = nd->left; }  /* nearer child on top */
        else { stack[sp++] = nd->left; stack[sp++] = nd->right; }
      }
    }
  }
  if (d2_out) *d2_out = best;
  return bi;
}

/* ---------------------------------------------------------------- bounding-volume hierarchy over triangles */
static tree_t *bvh_for(const double *pts, int T, const int *tris) {
  int M = 0;
  /* the vertex count is not passed down: fingerprint the coordinates through a spread of triangle corners */
  uint64_t fp = (uint64_t)T;
  for (int s = 0; s < 96 && T > 0; ++s) {
    const size_t i = ((size_t)s * (size_t)(3 * T - 1)) / 95;
    const double *p = pts + 3 * tris[i];
    uint64_t bits;
    memcpy(&bits, &p[s % 3], 8);
    fp = mix64(fp, bits);
  }
  tree_t *t = cache_find(t_bvh, pts, tris, M, T, fp);
  if (t) return t;
  t = cache_victim(t_bvh);
  t->pts = pts; t->tris = tris; t->M = M; t->T = T; t->print = fp;
  const size_t n = (size_t)(T > 0 ? T : 1);
  t->rep = (double *)malloc(sizeof(double) * 3 * n);
  t->ilo = (double *)malloc(sizeof(double) * 3 * n);
  t->ihi = (double *)malloc(sizeof(double) * 3 * n);
  for (int k = 0; k < T; ++k) {
    const double *a = pts + 3 * tris[3 * k], *b = pts + 3 * tris[3 * k + 1], *c = pts + 3 * tris[3 * k + 2];
    for (int d = 0; d < 3; ++d) {
      t->rep[3 * k + d] = (a[d] + b[d] + c[d]) / 3.0;
      t->ilo[3 * k + d] = fmin(a[d], fmin(b[d], c[d]));
      t->ihi[3 * k + d] = fmax(a[d], fmax(b[d], c[d]));
    }
  }
  tree_build(t, T, 4);
  ++t_bvh_builds;
  return t;
}

int spatial_closest_on_surface(const double *q, const double *pts, int T, const int *tris, orc_tri_kernel kernel, double *cp_out,
                               double *d2_out) {
  tree_t *t = bvh_for(pts, T, tris);
  double best = INFINITY, bp[3] = {0.0, 0.0, 0.0};
  int bi = -1;
  if (t->n_nodes > 0) {
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
      const node_t *nd = &t->nodes[stack[--sp]];
      if (box_bound(q, nd->lo, nd->hi) > best * (1.0 + 1e-12) + 1e-300) continue;
      if (nd->left < 0) {
        for (int i = nd->start; i < nd->start + nd->count; ++i) {
          const int k = t->items[i];
          double o[3];
          kernel(q, pts + 3 * tris[3 * k], pts + 3 * tris[3 * k + 1], pts + 3 * tris[3 * k + 2], o);
          const double dd = d2_of(q, o);
          if (dd < best || (dd == best && k < bi)) { best = dd; bi = k; bp[0] = o[0]; bp[1] = o[1]; bp[2] = o[2]; }
        }
      } else {
        const double bl = box_bound(q, t->nodes[nd->left].lo, t->nodes[nd->left].hi);
        const double br = box_bound(q, t->nodes[nd->right].lo, t->nodes[nd->right].hi);
        if (bl <= br) { stack[sp++] = nd->right; stack[sp++] = nd->left; }
        else { stack[sp++] = nd->left; stack[sp++] = nd->right; }
      }
    }
  }
  cp_out[0] = bp[0]; cp_out[1] = bp[1]; cp_out[2] = bp[2];
  if (d2_out) *d2_out = best;
  return bi;
}

/* ---------------------------------------------------------------- B2: the brute-force scans over all cores
 * live in icp_oracle.c (orc_nearest_vertex / orc_closest_point_on_surface): the K independent queries of a batch are spread over
 * the threads, every query the restatement's own sequential scan. */
