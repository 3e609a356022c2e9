#!/usr/bin/env python3
"""developer check of the wide step's on-device loop: the same chains through icp_chains_run_on_device (ICP_HOST_DEVICE_LOOP=1) and
host-stepped (=0), records compared.  usage: r5_wide_loop.py <kind> <B> <n> [out.npz]   (run by itself it does both modes)"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(kind, B, n, out):
    import __graft_entry__ as graft
    pkg = graft.load_package()
    if kind == "face100":      # apps/bfm/BfmFittingPartial.scala:62-96: open target, pose walks + ICP + shape walk, collective evaluator
        model = pkg.data.synthetic_face_model(grid=41, rank=100)
        target = pkg.data.synthetic_partial_target(model, n_remove=90, seed=7)
        mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
    elif kind == "face200":
        model = pkg.data.synthetic_face_model(grid=41, rank=200)
        target = pkg.data.synthetic_partial_target(model, n_remove=90, seed=7)
        mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
    elif kind == "facefull":   # configs[3] / configs[4] size: N = 28,561, rank 200
        model = pkg.data.synthetic_face_model()
        target = pkg.data.synthetic_partial_target(model, seed=100)
        mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
    elif kind == "hausdorff":
        model = pkg.data.synthetic_face_model(grid=41, rank=100)
        target = pkg.data.synthetic_partial_target(model, n_remove=90, seed=7)
        mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="hausdorff", fused=2)
    elif kind == "face40":     # rank <= 64: the warm-started Jacobi iteration inside the wide loop
        model = pkg.data.synthetic_face_model(grid=31, rank=40)
        target = pkg.data.synthetic_partial_target(model, n_remove=60, seed=7)
        mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
    elif kind == "femur100":   # apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala: two ICP directions + shape walk at rank 101 (closed target)
        model, target = pkg.data.load_femur_model_and_target(100)
        mk = lambda: pkg.femur_icp_proposal_registration(model, target, fused=2)
    elif kind == "femur200":   # the reference's largest model (rank 201: apps/femur/CreateGPModel.scala:93), two ICP directions + shape walk
        model, target = pkg.data.load_femur_model_and_target(200)
        mk = lambda: pkg.femur_icp_proposal_registration(model, target, fused=2)
    else:
        raise SystemExit("unknown kind")
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
    chains = [pkg.SamplingRegistration(ctxs[i], mk(), pkg.random_initial_parameters(model, i), seed=300 + i) for i in range(B)]
    t0 = time.time()
    a = pkg.run_chains_batched(chains, n)
    dt = time.time() - t0
    single = chains[0].run(5)
    b = pkg.run_chains_batched(chains[1:], 10) if B > 1 else []
    paths = [c.step_paths() for c in ctxs]
    print(kind, "mode", os.environ.get("ICP_HOST_DEVICE_LOOP"), "%.0f it/s" % (B * n / dt), paths[0], "acc %.2f" % np.stack(a)[:, :, 1].mean(),
          "leaves", sorted(set(np.stack(a)[:, :, 2].astype(int).ravel())), pkg._native.runtime_stats(), flush=True)
    np.savez(out, a=np.stack(a), single=single, b=np.stack(b) if B > 1 else np.zeros(1))
    [c.close() for c in chains]; [c.close() for c in ctxs]

if __name__ == "__main__":
    kind, B, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4:
        child(kind, B, n, sys.argv[4])
        sys.exit(0)
    out = {}
    for mode in ("1", "0"):
        path = "/tmp/wl_%s_%s.npz" % (kind, mode)
        rc = subprocess.run([sys.executable, __file__, kind, str(B), str(n), path], env={**os.environ, "ICP_HOST_DEVICE_LOOP": mode}, timeout=900).returncode
        if rc != 0:
            print("mode", mode, "failed rc", rc); sys.exit(1)
        out[mode] = np.load(path)
    for key in ("a", "single", "b"):
        d, h = out["1"][key], out["0"][key]
        same = np.array_equal(d, h)
        print(key, "identical" if same else "DIFFERENT", end=" ")
        if not same and d.shape == h.shape and d.ndim == 3:
            dec = np.array_equal(d[:, :, 1:3], h[:, :, 1:3])
            first = np.argwhere(np.any(d != h, axis=2))
            print("| decisions", "same" if dec else "differ", "| first differing (chain, step):", first[0] if len(first) else None,
                  "| max rel state diff %.2e" % (np.abs(d[:, :, 14:] - h[:, :, 14:]).max() / np.abs(h[:, :, 14:]).max()), end="")
        print()
