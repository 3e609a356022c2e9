"""GPU: the library beside ANOTHER PROCESS that keeps the device busy.

The merged step orders its streams ON THE DEVICE: a step's first launch waits for the decomposition it draws from, a decomposition
started ahead waits for the regression's word, a batched step's gate waits for its decompositions to become resident
(kernels_step.hip, kernels_posterior.hip).  Those waits rely on the waited-for launch becoming resident while the waiting one spins —
true on an idle device, not guaranteed when a second tenant's workgroups hold the compute units (the realistic case of "8 chains on 8
GPUs" started by a JVM pool next to other jobs).  Every wait has a time-out that ends in a slower, equivalent schedule and a counter
(icp_ctx_runtime_stats).  This test runs the metric chain (BASELINE.json configs[1], 58k-vertex target) and a 16-chain batch while
tests/support/gpu_hog — a separate process, started before this process' children touch the GPU — saturates the device, and demands:
identical records to the idle run (every decision, state and log value), no hang, and every fall-back accounted for in the counters
(a redo only with a time-out behind it).  The rates of both runs are printed; the busy one must stay within a generous factor of the
idle one (the device is time-shared with a tenant that fills it: a factor, not a fraction)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

HOG = os.path.join(ROOT, "tests", "support", "gpu_hog")

_SCRIPT = r"""
import sys, time, json, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.synthetic_femur_target()
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
out = {{}}
# ---- one chain: the pipelined single-chain step (device-side waits between its two streams and the eigen stream)
ctx = pkg.IcpContext(model, target, device=0)
ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
ch.run(5, want_records=False)
t0 = time.perf_counter(); rec1 = ch.run({n1}); out["single_it_s"] = {n1} / (time.perf_counter() - t0)
out["single_stats"] = ctx.runtime_stats()
ch.close(); ctx.close()
# ---- sixteen chains in lockstep groups (the batch gate) and, for the second call, inside the on-device loop
B = 16
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=2048 + i) for i in range(B)]
t0 = time.perf_counter(); rec2 = pkg.run_chains_batched(chains, {n2}); out["batched_it_s"] = B * {n2} / (time.perf_counter() - t0)
out["process_stats"] = pkg._native.runtime_stats()
[c.close() for c in chains]; [c.close() for c in ctxs]
np.savez({out!r}, rec1=rec1, rec2=np.stack(rec2))
print(json.dumps(out))
"""


def _run(tmp_path, tag, env_extra):
    path = str(tmp_path / (tag + ".npz"))
    done = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT, n1=300, n2=40, out=path)], check=True, capture_output=True, text=True,
                          env={**os.environ, **env_extra}, timeout=900)
    info = json.loads(done.stdout.strip().splitlines()[-1])
    return np.load(path), info


@pytest.mark.parametrize("loop", ["0", "1"])
def test_chains_beside_a_process_that_saturates_the_device(tmp_path, loop):
    assert os.path.exists(HOG), "build the test support programs (python -c 'import __graft_entry__ as g; g.build()')"
    env = {"ICP_HOST_DEVICE_LOOP": loop}
    idle, idle_info = _run(tmp_path, "idle", env)
    hog = subprocess.Popen([HOG, "120"], stdout=subprocess.PIPE, text=True)
    try:
        assert hog.stdout.readline().strip() == "running"
        t0 = time.time()
        busy, busy_info = _run(tmp_path, "busy", env)
        took = time.time() - t0
        assert hog.poll() is None, "the second tenant ended before the chains did: the device was not busy throughout (%.0f s)" % took
    finally:
        hog.kill()
        hog.wait()
    print("idle:", json.dumps(idle_info), "\nbusy:", json.dumps(busy_info))
    for key in ("rec1", "rec2"):
        assert np.array_equal(idle[key], busy[key]), "records beside a busy device differ from the idle run: " + key
    assert all(v == 0 for v in idle_info["process_stats"].values()), idle_info
    for info in (busy_info["single_stats"], busy_info["process_stats"]):
        timeouts = info["wait_timeouts"] + info["speculation_giveups"] + info["gate_timeouts"]
        assert info["step_redos"] <= timeouts + info["pipeline_fallbacks"], info  # (no step is done twice without a counted cause)
    # The tenant fills every compute unit with 40 µs workgroups, 16 launches of 4,096 of them queued per stream: each of a chain's small
    # launches queues behind a tenant launch's dispatch, and how the command processor alternates between the two processes' queues
    # varies from run to run — measured: the single chain 7x slower than idle in one run (1,700 it/s against 12,400), 90x in the
    # next (138 it/s); the 16-chain batch 3-5x.  The ratio is a property of the device's scheduler under an adversarial neighbour, not
    # of this library (no time-out fired either way): it is printed, and only a collapse (a hang that a time-out then resolves:
    # steps of 50 ms and more) fails the test.
    print("slow-down beside the tenant: single chain %.1fx, 16 chains %.1fx" % (
        idle_info["single_it_s"] / busy_info["single_it_s"], idle_info["batched_it_s"] / busy_info["batched_it_s"]))
    assert busy_info["single_it_s"] >= 20.0 and busy_info["batched_it_s"] >= 16 * 20.0, (idle_info, busy_info)
    # … and as RATIOS (verdict r05): the 16-chain batch — whose launches carry enough work to hold their own against the tenant's — keeps at
    # least a tenth of its idle rate (measured 3-5x slower); the lone chain at least 1/150 (measured 7-90x: every one of its ~6 small
    # launches per step queues behind a 40 µs x 4,096-workgroup launch of the tenant)
    assert busy_info["batched_it_s"] >= idle_info["batched_it_s"] / 10.0, (idle_info, busy_info)
    assert busy_info["single_it_s"] >= idle_info["single_it_s"] / 150.0, (idle_info, busy_info)
