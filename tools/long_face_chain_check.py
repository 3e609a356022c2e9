"""Developer check: a long face-configuration chain (pose + ICP + random walk; rank above 64: tridiagonal route) with the four-stream
step and decompositions started ahead (default) against the same chain with ICP_SPECULATION=0 (everything on one stream, nothing
ahead): the records must be identical — the schedule must not change a number."""
import os, subprocess, sys
import numpy as np
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as graft
pkg = graft.load_package()
n_steps, rank, evaluator, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
model = pkg.data.synthetic_face_model(grid=41, rank=rank)
target = pkg.data.synthetic_partial_target(model, n_remove=90)
setup = pkg.bfm_fitting_partial(model, target, evaluator=evaluator)
ctx = pkg.IcpContext(model, target, device=0)
chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), 77)
np.save(out, chain.run(n_steps))
'''
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for rank in (72, 150, 200):
    for evaluator in ("hausdorff", "collective"):
        recs = []
        for spec in ("default", "0"):
            env = dict(os.environ)
            if spec != "default": env["ICP_SPECULATION"] = spec
            out = "/tmp/face_chain_%s.npy" % spec
            subprocess.run([sys.executable, "-c", CHILD, str(n_steps), str(rank), evaluator, out], check=True, env=env)
            recs.append(np.load(out))
        a, b = recs
        same = np.array_equal(a[:, 1:3], b[:, 1:3])
        print("rank %d %s: %d steps, accepted %d | decisions identical: %s | max |state difference| %.2e | max |log value difference| %.2e"
              % (rank, evaluator, n_steps, int(a[:, 1].sum()), same, np.abs(a[:, 14:] - b[:, 14:]).max(), np.abs(a[:, 3] - b[:, 3]).max()))
