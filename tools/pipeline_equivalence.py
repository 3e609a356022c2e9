"""dev: the pipelined chain step (default) against the unpipelined one (ICP_NO_PIPELINE=1), record for record."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.synthetic_femur_target()
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
ctx = pkg.IcpContext(model, target, device=0)
chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), 1024)
recs = [chain.run(n) for n in (1500, 1, 2, 1497)]   # (several runs: a pending half step is dropped and rebuilt in between)
np.save({out!r}, np.concatenate(recs))
chain.close(); ctx.close()
"""
outs = []
for env in ({}, {"ICP_NO_PIPELINE": "1"}):
    out = os.path.join(tempfile.mkdtemp(), "rec.npy")
    subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT, out=out)], check=True, env={**os.environ, **env})
    outs.append(np.load(out))
a, b = outs
print("records", a.shape, "identical:", np.array_equal(a, b), "| accepted", int(a[:, 1].sum()),
      "| max |diff|", float(np.abs(a - b).max()))
