"""Input data for the closest-point-proposal path: statistical mesh model, target meshes, synthetic targets.

Everything here is host-side preparation that the reference delegates to Scalismo IO
(`StatisticalModelIO.readStatisticalMeshModel`, `MeshIO.readMesh`, `LandmarkIO.readLandmarksJson`,
reference `apps/femur/LoadTestData.scala:32-50`).  The model arrays come from the build-owned fixtures in
`tests/golden/femur/` (bit-exact float32 copies of the reference's bundled data).

Conventions:
  ref_points  x̄  [N,3] f64     reference mesh vertices
  mean_def    μ   [N,3] f64     mean deformation (Statismo stores the mean *shape*; μ = mean − x̄)
  basis       Φ   [3N,r] f64    unscaled eigenfunctions (Statismo v0.9 layout), row 3i+d = vertex i, axis d
  variance    λ   [r]   f64     eigenvalues;  Q = Φ·diag(√λ)
  cells           [T,3] i32     triangles
"""
from __future__ import annotations

import dataclasses
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE_DIR = os.path.join(os.path.dirname(_HERE), "tests", "golden", "femur")


@dataclasses.dataclass
class TriangleMesh:
    """Counterpart of Scalismo's TriangleMesh3D as far as this path needs it: points + triangulation."""
    points: np.ndarray  # [V,3] f64
    cells: np.ndarray   # [T,3] i32

    def __post_init__(self):
        self.points = np.ascontiguousarray(self.points, dtype=np.float64)
        self.cells = np.ascontiguousarray(self.cells, dtype=np.int32)

    @property
    def n_points(self):
        return self.points.shape[0]

    @property
    def n_cells(self):
        return self.cells.shape[0]

    def transform(self, fn):
        return TriangleMesh(fn(self.points), self.cells)


@dataclasses.dataclass
class StatisticalMeshModel:
    """Counterpart of Scalismo's StatisticalMeshModel restricted to what crosses the C-ABI boundary."""
    ref_points: np.ndarray
    cells: np.ndarray
    mean_def: np.ndarray
    basis: np.ndarray
    variance: np.ndarray

    def __post_init__(self):
        self.ref_points = np.ascontiguousarray(self.ref_points, dtype=np.float64)
        self.cells = np.ascontiguousarray(self.cells, dtype=np.int32)
        self.mean_def = np.ascontiguousarray(self.mean_def, dtype=np.float64)
        self.basis = np.ascontiguousarray(self.basis, dtype=np.float64)
        self.variance = np.ascontiguousarray(self.variance, dtype=np.float64)
        assert self.basis.shape == (3 * self.ref_points.shape[0], self.variance.shape[0])

    @property
    def rank(self):
        return self.variance.shape[0]

    @property
    def n_points(self):
        return self.ref_points.shape[0]

    @property
    def n_cells(self):
        return self.cells.shape[0]

    @property
    def reference_mesh(self):
        return TriangleMesh(self.ref_points, self.cells)

    def instance(self, coeffs):
        """x̄ + μ + Q c (SURVEY App. A.1 without pose); numpy convenience, not used by the device path."""
        q = self.basis * np.sqrt(self.variance)[None, :]
        return self.ref_points + self.mean_def + (q @ np.asarray(coeffs, dtype=np.float64)).reshape(-1, 3)


def load_femur_model(n_components: int = 50, fixture_dir: str = FIXTURE_DIR) -> StatisticalMeshModel:
    """femur GPMM with `n_components`+1 basis functions (reference: data/femur/femur_gp_model_*-components.h5)."""
    z = np.load(os.path.join(fixture_dir, f"femur_gp_model_{n_components}.npz"))
    pts = z["points"].astype(np.float64)
    mean = z["mean"].astype(np.float64)
    return StatisticalMeshModel(pts, z["cells"], mean - pts, z["pcaBasis"].astype(np.float64),
                                z["pcaVariance"].astype(np.float64))


def load_femur_mesh(name: str, fixture_dir: str = FIXTURE_DIR):
    """name in {"femur_reference","femur_target"} -> (TriangleMesh, landmark ids, landmarks [L,3])."""
    z = np.load(os.path.join(fixture_dir, name + ".npz"))
    return TriangleMesh(z["points"].astype(np.float64), z["cells"]), [str(s) for s in z["landmark_ids"]], z["landmarks"]


def rigid_landmark_transform(src: np.ndarray, dst: np.ndarray):
    """Least-squares rigid (R, t) with dst ≈ R·src + t (Kabsch).

    Stands in for Scalismo `LandmarkRegistration.rigid3DLandmarkRegistration` as called from the reference's
    `apps/util/AlignmentTransforms.scala:25-30` [SCALISMO-UNVERIFIED]; the optimum is unique, so only
    rounding differs.
    """
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    cs, cd = src.mean(0), dst.mean(0)
    h = (src - cs).T @ (dst - cd)
    u, _, vt = np.linalg.svd(h)
    d = np.sign(np.linalg.det(vt.T @ u.T))
    rot = vt.T @ np.diag([1.0, 1.0, d]) @ u.T
    return rot, cd - rot @ cs


def load_femur_model_and_target(n_components: int = 50):
    """Counterpart of reference `apps/femur/LoadTestData.scala:32-50`: model + landmark-aligned target mesh."""
    model = load_femur_model(n_components)
    _, ref_ids, ref_lms = load_femur_mesh("femur_reference")
    target, tgt_ids, tgt_lms = load_femur_mesh("femur_target")
    common = [i for i in tgt_ids if i in ref_ids]
    a = np.asarray([tgt_lms[tgt_ids.index(i)] for i in common])
    b = np.asarray([ref_lms[ref_ids.index(i)] for i in common])
    rot, t = rigid_landmark_transform(a, b)
    return model, target.transform(lambda p: p @ rot.T + t)


def subdivide(mesh: TriangleMesh, n: int) -> TriangleMesh:
    """n-way edge subdivision: every triangle -> n² triangles; shared edge/corner vertices are merged.

    V' = V + (n-1)·E + (n-1)(n-2)/2·F.  Used to synthesise the "~50k-vertex target" of BASELINE.json
    (n=6 on the femur target gives 58,322 vertices / 116,640 triangles).
    """
    pts, cells = mesh.points, mesh.cells
    V = pts.shape[0]
    new_pts = [pts]
    count = V
    edge_ids = {}

    def edge_points(a, b):
        nonlocal count
        key = (a, b) if a < b else (b, a)
        ids = edge_ids.get(key)
        if ids is None:
            lo, hi = key
            w = (np.arange(1, n, dtype=np.float64) / n)[:, None]
            new_pts.append(pts[lo] * (1.0 - w) + pts[hi] * w)
            ids = np.arange(count, count + n - 1, dtype=np.int64)
            count += n - 1
            edge_ids[key] = ids
        return ids if a < b else ids[::-1]

    out_cells = []
    for a, b, c in cells.tolist():
        # barycentric lattice: node (i,j) = a + (b-a)·i/n + (c-a)·j/n, i+j<=n
        idx = -np.ones((n + 1, n + 1), dtype=np.int64)
        idx[0, 0], idx[n, 0], idx[0, n] = a, b, c
        eab, eac, ebc = edge_points(a, b), edge_points(a, c), edge_points(b, c)
        for i in range(1, n):
            idx[i, 0] = eab[i - 1]
            idx[0, i] = eac[i - 1]
            idx[n - i, i] = ebc[i - 1]
        inner = [(i, j) for i in range(1, n) for j in range(1, n - i)]
        if inner:
            ij = np.asarray(inner, dtype=np.float64)
            p = pts[a] + (pts[b] - pts[a]) * (ij[:, :1] / n) + (pts[c] - pts[a]) * (ij[:, 1:] / n)
            new_pts.append(p)
            for k, (i, j) in enumerate(inner):
                idx[i, j] = count + k
            count += len(inner)
        for i in range(n):
            for j in range(n - i):
                out_cells.append((idx[i, j], idx[i + 1, j], idx[i, j + 1]))
                if i + j < n - 1:
                    out_cells.append((idx[i + 1, j], idx[i + 1, j + 1], idx[i, j + 1]))
    return TriangleMesh(np.concatenate(new_pts, axis=0), np.asarray(out_cells, dtype=np.int32))


def synthetic_femur_target(n_subdiv: int = 6, jitter_mm: float = 0.05, seed: int = 1024,
                           n_components: int = 50):
    """BASELINE.json metric config (`configs[1]`): femur model + the aligned bundled target subdivided
    `n_subdiv`-way with seeded N(0, jitter) vertex noise (removes coplanar ties).  SURVEY.md §8d."""
    model, target = load_femur_model_and_target(n_components)
    big = subdivide(target, n_subdiv) if n_subdiv > 1 else target
    if jitter_mm > 0:
        rng = np.random.Generator(np.random.PCG64(seed))
        big = TriangleMesh(big.points + rng.normal(0.0, jitter_mm, size=big.points.shape), big.cells)
    return model, big


def decimated_point_subset(mesh: TriangleMesh, k: int) -> np.ndarray:
    """Stand-in for Scalismo `mesh.operations.decimate(k).pointSet.points` (VTK quadric decimation; not
    reproducible here).  Only the resulting point list crosses the C-ABI, so any host-side
    choice is valid input; we take k vertices by a deterministic stride over the vertex list."""
    k = min(k, mesh.n_points)
    idx = (np.arange(k, dtype=np.int64) * mesh.n_points) // k
    return mesh.points[idx].copy()


def boundary_vertex_flags(mesh: TriangleMesh) -> np.ndarray:
    """vertex lies on an edge owned by exactly one triangle (Scalismo `pointIsOnBoundary`)."""
    c = mesh.cells.astype(np.int64)
    e = np.concatenate([c[:, [0, 1]], c[:, [1, 2]], c[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    key = e[:, 0] * mesh.n_points + e[:, 1]
    uniq, cnt = np.unique(key, return_counts=True)
    b = uniq[cnt == 1]
    flags = np.zeros(mesh.n_points, dtype=np.uint8)
    flags[b // mesh.n_points] = 1
    flags[b % mesh.n_points] = 1
    return flags


# ---------------------------------------------------------------------------------------------------------------
# Synthetic stand-in for the Basel Face Model configurations (BASELINE.json configs[3], configs[4]).
# The BFM-2017 files are not redistributable and absent from the reference tree (README.md:60-70, .gitignore:26-29);
# SURVEY.md §8d prescribes a procedural model of the same SIZE: an open height-field "face" patch with smooth
# low-frequency deformation modes, and a partial target with a hole (the reference crops its scans around the nose,
# apps/bfm/AlignShapes.scala:90-92).  Nothing here claims to resemble the BFM statistically.

def synthetic_face_model(grid: int = 169, rank: int = 200, seed: int = 2017) -> StatisticalMeshModel:
    """grid×grid vertices (169 -> N = 28,561, T = 56,448; BFM face12 has 28,588 / 56,572), `rank` deformation modes.

    Geometry: a 150 mm × 200 mm patch with a nose-like bump.  Basis: 2-D cosine modes cos(pi p u)·cos(pi q v) on cell-centred
    coordinates (exactly orthogonal on the grid), ordered by frequency; every mode deforms along a seeded random unit
    direction; columns are scaled to squared norm N like Statismo's unscaled eigenfunctions; the variances decay
    with frequency from 25 down to ~0.05 mm²."""
    g = int(grid)
    u = (np.arange(g) + 0.5) / g
    uu, vv = np.meshgrid(u, u, indexing="ij")
    x = 150.0 * (uu - 0.5)
    y = 200.0 * (vv - 0.5)
    z = 45.0 * np.exp(-((uu - 0.5) ** 2 + (vv - 0.45) ** 2) / 0.012) + 12.0 * np.cos(np.pi * (uu - 0.5)) * np.cos(np.pi * (vv - 0.5))
    ref = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    idx = np.arange(g * g).reshape(g, g)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    cells = np.concatenate([np.stack([a, b, c], axis=1), np.stack([b, d, c], axis=1)], axis=0).astype(np.int32)
    n = g * g
    modes = sorted(((p * p + q * q, p, q) for p in range(32) for q in range(32)))[1:rank + 1]   # skip the constant mode
    rng = np.random.Generator(np.random.PCG64(seed))
    basis = np.empty((3 * n, rank))
    var = np.empty(rank)
    for j, (f2, p, q) in enumerate(modes):
        phi = (np.cos(np.pi * p * uu) * np.cos(np.pi * q * vv)).reshape(-1)
        dirn = rng.normal(size=3)
        dirn /= np.linalg.norm(dirn)
        col = (phi[:, None] * dirn[None, :]).reshape(-1)
        basis[:, j] = col * np.sqrt(n / np.dot(col, col))
        var[j] = 25.0 * np.exp(-f2 / 40.0) + 0.05
    return StatisticalMeshModel(ref, cells, np.zeros_like(ref), basis, var)


def synthetic_partial_target(model: StatisticalMeshModel, seed: int = 7, n_remove: int = 1000, coeff_scale: float = 0.5,
                             jitter_mm: float = 0.02) -> TriangleMesh:
    """A model sample (coefficients ~ coeff_scale·N(0, I), seeded) with the `n_remove` vertices nearest to the nose tip cut
    out (and seeded vertex jitter against exact ties): a target with an inner boundary, as in apps/bfm/BfmFittingPartial.scala."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pts = model.instance(coeff_scale * rng.normal(size=model.rank))
    pts = pts + rng.normal(0.0, jitter_mm, size=pts.shape)
    tip = int(np.argmax(model.ref_points[:, 2]))
    drop = np.zeros(model.n_points, dtype=bool)
    drop[np.argsort(np.linalg.norm(model.ref_points - model.ref_points[tip], axis=1))[:n_remove]] = True
    keep_cells = model.cells[~drop[model.cells].any(axis=1)]
    used = np.zeros(model.n_points, dtype=bool)
    used[keep_cells.ravel()] = True
    remap = -np.ones(model.n_points, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    return TriangleMesh(pts[used], remap[keep_cells].astype(np.int32))

