"""GPU: the JNI shim (bindings/jni/icp_jni.c) RUN — without a JVM.  tests/support/jni_mock is a JNI test double (the functions the shim
calls, implemented over plain C arrays); the natives of api.gpu.NativeIcp are called through it the way Scalismo's chain would call them
and compared with the C ABI driven directly (ctypes, api.py) on contexts of their own: the three plug-in methods over a bound chain
(MetropolisHastings.next's call order, api/sampling/SamplingRegistration.scala:52-58), the batched step in one call and in two halves,
the whole loop on the device (the replacement of apps/femur/RunMHRandomInitComparison.scala:66-87), re-targeting, and the exceptions."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
from test_bindings_cpu import jni_definitions

pytestmark = pytest.mark.gpu

_CT = {"jint": C.c_int32, "jlong": C.c_int64, "jdouble": C.c_double, "jboolean": C.c_uint8, "void": None,
       "jdoubleArray": C.c_void_p, "jintArray": C.c_void_p, "jlongArray": C.c_void_p}


class Jvm:
    """The natives of NativeIcp over the test double: Python lists / numpy arrays in, handles kept alive for the call, results copied out."""

    def __init__(self, pkg):
        pkg._native.lib()  # (the C ABI library first: the shim links against it)
        so = os.path.join(ROOT, "tests", "support", "jni_mock", "libicp_jni_mock.so")
        if not os.path.exists(so):  # (built by __graft_entry__.build(); the box has the same gcc)
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "support"), "jni_mock/libicp_jni_mock.so"], check=True)
        self.L = C.CDLL(so)
        self.L.mock_env.restype = C.c_void_p
        self.L.mock_new_array.restype = C.c_void_p
        self.L.mock_new_array.argtypes = [C.c_int, C.c_int32, C.c_void_p]
        self.L.mock_array_data.restype = C.c_void_p
        self.L.mock_array_data.argtypes = [C.c_void_p]
        self.L.mock_array_length.argtypes = [C.c_void_p]
        self.L.mock_free_array.argtypes = [C.c_void_p]
        self.env = self.L.mock_env()
        self.sig = jni_definitions()
        for name, (args, ret) in self.sig.items():
            fn = getattr(self.L, "Java_api_gpu_NativeIcp_00024_" + name)
            fn.restype = _CT[ret]
            fn.argtypes = [C.c_void_p, C.c_void_p] + [_CT[a] for a in args]

    def _arr(self, kind, a):
        dt = {1: np.float64, 2: np.int32, 3: np.int64}[kind]
        a = np.ascontiguousarray(a, dtype=dt)
        return self.L.mock_new_array(kind, a.size, a.ctypes.data_as(C.c_void_p))

    def _read(self, h, kind):
        n = self.L.mock_array_length(h)
        ct = {1: C.c_double, 2: C.c_int32, 3: C.c_int64}[kind]
        return np.ctypeslib.as_array(C.cast(self.L.mock_array_data(h), C.POINTER(ct)), shape=(n,)).copy() if n else np.zeros(0)

    def exception(self):
        cls, msg = C.create_string_buffer(64), C.create_string_buffer(512)
        return (cls.value.decode(), msg.value.decode()) if self.L.mock_take_exception(cls, 64, msg, 512) else None

    def call(self, name, *args, expect_exception=False):
        """Arrays: numpy arrays (copied into mock arrays; every array is copied back into the numpy array afterwards — JNI semantics:
        the native decides which ones it writes), None = null."""
        kinds = {"jdoubleArray": 1, "jintArray": 2, "jlongArray": 3}
        jargs, live = [], []
        for a, t in zip(args, self.sig[name][0]):
            if t in kinds:
                if a is None:
                    jargs.append(None)
                else:
                    h = self._arr(kinds[t], a)
                    live.append((h, kinds[t], a))
                    jargs.append(h)
            else:
                jargs.append(a)
        assert len(args) == len(self.sig[name][0]), name
        out = getattr(self.L, "Java_api_gpu_NativeIcp_00024_" + name)(self.env, None, *jargs)
        for h, k, a in live:
            if isinstance(a, np.ndarray) and a.flags.writeable:
                a[...] = self._read(h, k).reshape(a.shape)
            self.L.mock_free_array(h)
        exc = self.exception()
        if expect_exception:
            return exc
        assert exc is None, (name, exc)
        ret = self.sig[name][1]
        if ret in kinds and out:
            v = self._read(out, kinds[ret])
            self.L.mock_free_array(out)
            return v
        return out


def _jni_chain(jvm, pkg, model, target, key=77):
    r = model.rank
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).ravel()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64).ravel()
    ctx = jvm.call("ctxCreateKeyed", model.n_points, model.n_cells, r, f64(model.ref_points), f64(model.mean_def), f64(model.basis),
                   f64(model.variance), i32(model.cells), target.n_points, target.n_cells, f64(target.points), i32(target.cells), 0, key)
    assert ctx and jvm.call("ctxRank", ctx) == r
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    props = [jvm.call("proposalCreate", ctx, 0.1, 10.0, 5.0, 0, 1, 2 * r, None),
             jvm.call("proposalCreate", ctx, 0.1, 10.0, 5.0, 1, 1, 0, f64(tp))]
    ev = jvm.call("evaluatorCreate", ctx, 0, 0, 4 * r, f64(pkg.data.decimated_point_subset(target, 4 * r)), 0.0, 2.0, 1.0)
    return ctx, props, ev


def _abi_chain(pkg, model, target):
    r = model.rank
    ctx = pkg.IcpContext(model, target, device=0)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    props = [pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, "ModelSampling", True),
             pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, "TargetSampling", True, decimatedTargetPoints=tp)]
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, pkg.ModelToTargetEvaluation, 4 * r,
                                               decimatedTargetPoints=pkg.data.decimated_point_subset(target, 4 * r))
    return ctx, props, ev


def test_plugin_methods_through_the_jni_shim_over_a_bound_chain(pkg, femur50):
    model, target = femur50
    r, P = model.rank, 10 + model.rank
    jvm = Jvm(pkg)
    jctx, jprops, jev = _jni_chain(jvm, pkg, model, target)
    ctx, props, ev = _abi_chain(pkg, model, target)
    jvm.call("chainBind", jev, np.array(jprops, dtype=np.int64))
    rng = np.random.default_rng(3)
    cur = pkg.initial_parameters(model)
    cur[10:] = 0.3 * rng.normal(size=r)
    for step in range(6):
        z = rng.normal(size=r)
        cur_value = jvm.call("logValue", jev, cur)
        if step % 3 < 2:
            prop = np.zeros(P)
            jvm.call("propose", jprops[step % 3], cur, z, prop)
            assert np.allclose(prop, props[step % 3].propose(cur, z), rtol=1e-10, atol=1e-11)
        else:
            prop = np.concatenate([cur[:10], cur[10:] + 0.1 * z])
        prop_value = jvm.call("logValue", jev, prop)
        fwd = [jvm.call("logTransition", p, cur, prop) for p in jprops]
        bwd = [jvm.call("logTransition", p, prop, cur) for p in jprops]
        assert cur_value == ev.logValue(cur) and prop_value == ev.logValue(prop)
        assert fwd == [p.logTransitionProbability(cur, prop) for p in props]
        assert bwd == [p.logTransitionProbability(prop, cur) for p in props]
        if step % 2 == 0:
            cur = prop
    st = jvm.call("chainBindStats", jev)
    assert list(st) == [4, 2, 24], st
    assert list(jvm.call("stepPaths", jctx))[0] == 6 and not any(jvm.call("runtimeStats", jctx))
    assert jvm.call("chainStepPath", jev, np.array(jprops, dtype=np.int64)) == 0
    # a pose walk: -inf is a VALUE of logTransition (NonRigidIcpProposal.scala:72-74), not an exception
    moved = cur.copy()
    moved[5] += 0.01
    assert jvm.call("logTransition", jprops[0], cur, moved) == -np.inf
    # exceptions: a NaN state -> RuntimeException carrying icp_last_error(); a wrong array length -> RuntimeException, nothing written
    bad = cur.copy()
    bad[12] = np.nan
    exc = jvm.call("logValue", jev, bad, expect_exception=True)
    assert exc and exc[0] == "java/lang/RuntimeException" and "finite" in exc[1], exc
    exc = jvm.call("propose", jprops[0], cur, rng.normal(size=r), np.zeros(P - 1), expect_exception=True)
    assert exc and exc[0] == "java/lang/RuntimeException", exc
    # mesh metrics and re-targeting (icp_ctx_set_target is refused while a proposal / evaluator lives)
    m = jvm.call("meshMetrics", jctx, cur)
    want = pkg.evaluate_reconstruction_to_ground_truth(ctx, cur)
    assert m.shape == (5,) and m[0] == want["average2surface"] and m[1] == want["hausdorff"], (m, want)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64).ravel()
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).ravel()
    exc = jvm.call("ctxSetTarget", jctx, target.n_points, target.n_cells, f64(target.points), i32(target.cells), expect_exception=True)
    assert exc and exc[0] == "java/lang/RuntimeException", exc
    for p in jprops:
        jvm.call("proposalDestroy", p)
    jvm.call("evaluatorDestroy", jev)
    shifted = target.points + np.array([0.5, 0.0, 0.0])
    jvm.call("ctxSetTarget", jctx, target.n_points, target.n_cells, f64(shifted), i32(target.cells))
    jev2 = jvm.call("evaluatorCreate", jctx, 0, 0, 4 * r, f64(pkg.data.decimated_point_subset(target, 4 * r) + np.array([0.5, 0.0, 0.0])), 0.0, 2.0, 1.0)
    v2 = jvm.call("logValue", jev2, cur)
    assert np.isfinite(v2) and v2 != ev.logValue(cur)
    jvm.call("evaluatorDestroy", jev2)
    jvm.call("ctxDestroy", jctx)
    for o in props + [ev]:
        o.close()
    ctx.close()


def test_batched_steps_and_the_device_loop_through_the_jni_shim(pkg, femur50):
    model, target = femur50
    r, P, B, n = model.rank, 10 + model.rank, 3, 14
    jvm = Jvm(pkg)
    jc = [_jni_chain(jvm, pkg, model, target, key=91) for _ in range(B)]
    ac = [_abi_chain(pkg, model, target) for _ in range(B)]
    evs = np.array([c[2] for c in jc], dtype=np.int64)
    prs = np.array([p for c in jc for p in c[1]], dtype=np.int64)
    rng = np.random.default_rng(8)
    thetas = np.stack([pkg.random_initial_parameters(model, b) for b in range(B)])
    # ---- icp_chain_step_batched in one call and in two halves, against icp_chain_step chain by chain
    for halves in (False, True):
        gen = np.array([0, 1, -1], dtype=np.int32)
        z = rng.normal(size=(B, r))
        prop = thetas.copy()
        prop[2, 10:] += 0.1 * z[2]
        lv, fwd, bwd, status = np.zeros(B), np.zeros(B * 2), np.zeros(B * 2), np.zeros(B, dtype=np.int32)
        if not halves:
            rc = jvm.call("chainStepBatched", evs, 2, prs, gen, thetas.ravel(), z.ravel(), prop.reshape(-1), lv, fwd, bwd, status)
        else:
            ticket = jvm.call("chainStepBatchedIssue", evs, 2, prs, gen, thetas.ravel(), z.ravel(), prop.reshape(-1), 0)
            assert ticket
            exc = jvm.call("logValue", int(evs[0]), thetas[0], expect_exception=True)  # a member context is busy until the ticket is collected
            assert exc and "batch in flight" in exc[1], exc
            rc = jvm.call("chainStepBatchedCollect", ticket, prop.reshape(-1), lv, fwd, bwd, status)
        assert rc == 0 and not status.any()
        for b in range(B):
            got, val, f, bw = pkg.chain_step(ac[b][2], ac[b][1], thetas[b], generator=int(gen[b]), z=z[b], theta_prop=prop[b] if gen[b] < 0 else None)
            assert np.allclose(prop[b], got, rtol=1e-10, atol=1e-11)
            val2, f2, bw2 = pkg.chain_eval_step(ac[b][2], ac[b][1], thetas[b], prop[b])  # (at the SAME proposed state: no decomposition behind these)
            assert lv[b] == val2 and list(fwd[2 * b:2 * b + 2]) == list(f2) and list(bwd[2 * b:2 * b + 2]) == list(bw2)
        thetas = prop.copy()
    # an abandoned ticket records nothing and frees the contexts
    ticket = jvm.call("chainStepBatchedIssue", evs, 2, prs, np.zeros(B, dtype=np.int32), thetas.ravel(), rng.normal(size=B * r), thetas.ravel().copy(), 0)
    jvm.call("chainStepBatchedAbandon", ticket)
    # ---- icp_chains_run_on_device: records identical to the C ABI driven directly
    prior = pkg.ModelPriorEvaluator(r)
    start = np.stack([pkg.random_initial_parameters(model, 10 + b) for b in range(B)])
    logp = np.array([prior.logValue(start[b]) + jvm.call("logValue", int(evs[b]), start[b]) for b in range(B)])
    mixture = np.array([0.5, 0.5, 0.9, 0.1, 0.1, 0.0, 0.01, 0.01, 0.01, 0.1, 0.1, 0.1])
    seeds, first = np.arange(700, 700 + B, dtype=np.int64), np.zeros(B, dtype=np.int64)
    j_theta, j_logp = start.ravel().copy(), logp.copy()
    j_rec, j_acc = np.zeros(B * n * (4 + P)), np.zeros(B, dtype=np.int64)
    jvm.call("chainsRunOnDevice", evs, 2, prs, mixture, seeds, first, j_theta, j_logp, n, j_rec, j_acc)
    nat = pkg._native
    mix = nat.MhMixture(C.sizeof(nat.MhMixture), (C.c_double * 2)(0.5, 0.5), 0.9, 0.1, 0.1, 0.0, (C.c_double * 3)(0.01, 0.01, 0.01),
                        (C.c_double * 3)(0.1, 0.1, 0.1))
    a_theta = [start[b].copy() for b in range(B)]
    a_logp = np.array([prior.logValue(start[b]) + ac[b][2].logValue(start[b]) for b in range(B)])
    assert np.array_equal(a_logp, logp)
    a_rec = [np.zeros((n, 4 + P)) for _ in range(B)]
    a_acc = (C.c_int64 * B)()
    eh = (C.c_void_p * B)(*[c[2].h for c in ac])
    ph = (C.c_void_p * (2 * B))(*[p.h for c in ac for p in c[1]])
    thp = (nat.c_double_p * B)(*[t.ctypes.data_as(nat.c_double_p) for t in a_theta])
    rcp = (nat.c_double_p * B)(*[t.ctypes.data_as(nat.c_double_p) for t in a_rec])
    sd = (C.c_uint64 * B)(*[int(s) for s in seeds])
    fs = (C.c_int64 * B)(*[0] * B)
    rc = nat.lib().icp_chains_run_on_device(B, eh, 2, ph, C.byref(mix), sd, fs, thp, a_logp.ctypes.data_as(nat.c_double_p), n, rcp, a_acc)
    assert rc == 0
    # (the two sets of contexts have different decomposition histories — the warm-started Jacobi iteration starts from another basis:
    # decisions identical, states to rounding)
    jr, ar = j_rec.reshape(B, n, 4 + P), np.stack(a_rec)
    assert np.array_equal(jr[:, :, :3], ar[:, :, :3]), "index / decision / mixture component differ"
    assert np.allclose(jr[:, :, 3:], ar[:, :, 3:], rtol=1e-9, atol=1e-10)
    assert np.allclose(j_theta.reshape(B, P), np.stack(a_theta), rtol=1e-9, atol=1e-10) and np.allclose(j_logp, a_logp, rtol=1e-9)
    assert list(j_acc) == list(a_acc) and np.array_equal(j_theta.reshape(B, P), jr[:, -1, 4:])
    assert j_acc.sum() >= 2 and list(jvm.call("stepPaths", jc[0][0]))[3] == n
    for ctx, props, ev in jc:
        for p in props:
            jvm.call("proposalDestroy", p)
        jvm.call("evaluatorDestroy", ev)
        jvm.call("ctxDestroy", ctx)
    for ctx, props, ev in ac:
        for o in props + [ev]:
            o.close()
        ctx.close()
    jvm.call("releaseCachedModels")
