#!/usr/bin/env python3
"""Convert the reference's bundled femur DATA files into small build-owned fixtures.

Runs only in the build container (needs /root/reference and /opt/conda/bin/h5dump).
Data only: the statistical model arrays (HDF5, Statismo layout), the two STL meshes and the
landmark JSON files under /root/reference/data/femur.  No reference source
text is read or copied.  Output: tests/golden/femur/*.npz (float32/int32, bit-exact copies).
"""
import json, os, struct, subprocess, sys, tempfile
import numpy as np

REF = "/root/reference/data/femur"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "femur")
H5DUMP = "/opt/conda/bin/h5dump"


def h5_dataset(path, name, dtype, shape):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run([H5DUMP, "-d", name, "-b", "LE", "-o", tmp.name, path],
                       check=True, stdout=subprocess.DEVNULL)
        a = np.fromfile(tmp.name, dtype=dtype)
    return a.reshape(shape)


def read_stl_binary(path):
    """Binary STL -> (vertices merged in first-occurrence order [V,3] f32, cells [T,3] i32)."""
    raw = open(path, "rb").read()
    (ntri,) = struct.unpack_from("<I", raw, 80)
    rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                        count=ntri, offset=84)
    corners = rec["v"].reshape(-1, 3)
    keys = corners.view(np.uint32).reshape(-1, 3)
    seen, verts, cells = {}, [], np.empty(ntri * 3, dtype=np.int32)
    for i, k in enumerate(map(tuple, keys)):
        j = seen.get(k)
        if j is None:
            j = len(verts)
            seen[k] = j
            verts.append(corners[i])
        cells[i] = j
    return np.asarray(verts, dtype=np.float32), cells.reshape(ntri, 3)


def landmarks(path):
    lm = json.load(open(path))
    return [l["id"] for l in lm], np.asarray([l["coordinates"] for l in lm], dtype=np.float64)


def main():
    os.makedirs(OUT, exist_ok=True)
    for n in (50, 100, 200):
        f = os.path.join(REF, f"femur_gp_model_{n}-components.h5")
        r = n + 1
        pts = h5_dataset(f, "/representer/points", "<f4", (3, -1)).T.copy()
        N = pts.shape[0]
        cells = h5_dataset(f, "/representer/cells", "<i4", (3, -1)).T.copy()
        mean = h5_dataset(f, "/model/mean", "<f4", (N, 3))
        basis = h5_dataset(f, "/model/pcaBasis", "<f4", (3 * N, r))
        var = h5_dataset(f, "/model/pcaVariance", "<f4", (r,))
        noise = h5_dataset(f, "/model/noiseVariance", "<f4", (1,))
        # (round 6: the 200-component model — rank 201, the reference's largest, apps/femur/CreateGPModel.scala:93 — is a fixture as
        # well: tests/test_gpu_rank201.py, 3.7 MB)
        np.savez_compressed(os.path.join(OUT, f"femur_gp_model_{n}.npz"), points=pts, cells=cells,
                            mean=mean, pcaBasis=basis, pcaVariance=var, noiseVariance=noise)
        print(n, "N", N, "T", cells.shape[0], "rank", r)
    for name in ("femur_reference", "femur_target"):
        v, c = read_stl_binary(os.path.join(REF, name + ".stl"))
        ids, lm = landmarks(os.path.join(REF, name + ".json"))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), points=v, cells=c,
                            landmark_ids=np.asarray(ids), landmarks=lm)
        print(name, v.shape, c.shape, ids)


if __name__ == "__main__":
    main()
