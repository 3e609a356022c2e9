"""CPU: the drop-in boundary itself — the C ABI library builds for gfx950, loads, and exports every symbol that
include/icp_proposal.h declares; without a GPU it refuses to work instead of falling back to a CPU path."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_theta


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "icp_proposal.h")).read()
    return sorted(set(re.findall(r"ICP_API\s+[\w\s\*]+?\b(icp_\w+)\s*\(", text)))


def test_header_symbols_exported(pkg):
    lib = pkg._native.lib()
    names = declared_symbols()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/icp_proposal.h but not exported"
    assert set(names) == set(pkg._native.SIGNATURES), "ctypes binding and header disagree"


def test_library_is_gfx950_only(pkg):
    """the fat binary inside the shared library carries exactly one GPU target: gfx950"""
    blob = open(pkg._native.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_struct_layouts_match_header(pkg):
    nat = pkg._native
    assert ctypes.sizeof(nat.ModelDesc) == 56 and ctypes.sizeof(nat.MeshDesc) == 24
    assert ctypes.sizeof(nat.ProposalParams) == 48 and ctypes.sizeof(nat.EvaluatorParams) == 48
    assert ctypes.sizeof(nat.PosteriorView) == 72 and ctypes.sizeof(nat.KernelStat) == 72
    assert ctypes.sizeof(nat.RuntimeStats) == 64


def test_runtime_stats_start_at_zero(pkg):
    """icp_ctx_runtime_stats(NULL): the process-wide fall-back counters (no device needed)."""
    st = pkg._native.runtime_stats()
    assert set(st) == {"wait_timeouts", "speculation_giveups", "pipeline_fallbacks", "step_redos", "gate_timeouts"}
    assert all(v == 0 for v in st.values())
    assert pkg._native.lib().icp_ctx_runtime_stats(None, None) == -1


def test_prior_is_host_arithmetic(pkg, femur50):
    model, _ = femur50
    theta = make_theta(model, 3)
    got = pkg.ModelPriorEvaluator(model.rank).logValue(theta)
    assert np.isclose(got, -0.5 * theta[10:] @ theta[10:] - 0.5 * model.rank * np.log(2 * np.pi), rtol=1e-14)


def test_status_strings(pkg):
    lib = pkg._native.lib()
    assert lib.icp_status_string(0) == b"ok" and b"HIP" in lib.icp_status_string(-2)


def _gpu_present():
    try:
        return subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=30).stdout.count("gfx950") > 0
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="a GPU is present; this checks the no-GPU behaviour")
def test_no_cpu_fallback(pkg, femur50):
    model, target = femur50
    with pytest.raises(pkg._native.IcpNativeError) as e:
        pkg.IcpContext(model, target)
    assert e.value.status == -2 and "no CPU fallback" in str(e.value)


def test_invalid_arguments_rejected_before_device(pkg, femur50):
    model, target = femur50
    lib, nat = pkg._native.lib(), pkg._native
    h = ctypes.c_void_p()
    assert lib.icp_ctx_create(None, None, 0, ctypes.byref(h)) == -1
    bad = model.cells.copy()
    bad[0, 0] = model.n_points + 5
    md = nat.ModelDesc(model.n_points, bad.shape[0], model.rank, model.ref_points.ctypes.data_as(nat.c_double_p), None,
                       model.basis.ctypes.data_as(nat.c_double_p), model.variance.ctypes.data_as(nat.c_double_p),
                       bad.ctypes.data_as(nat.c_int_p))
    td = nat.MeshDesc(target.n_points, target.n_cells, target.points.ctypes.data_as(nat.c_double_p),
                      target.cells.ctypes.data_as(nat.c_int_p))
    assert lib.icp_ctx_create(ctypes.byref(md), ctypes.byref(td), 0, ctypes.byref(h)) == -1
    assert b"out of range" in lib.icp_last_error()


def test_batched_step_rejects_bad_arguments_before_device(pkg):
    """icp_chain_step_batched / _issue / _collect: null and empty argument lists are refused with ICP_ERR_INVALID_ARG and leave
    no ticket behind (no device is touched: this runs without a GPU)."""
    lib = pkg._native.lib()
    assert lib.icp_chain_step_batched(0, None, 0, None, None, None, None, None, None, None, None, None) == -1
    assert b"null argument" in lib.icp_last_error()
    status = (ctypes.c_int32 * 1)()
    assert lib.icp_chain_step_batched(1, None, 0, None, None, None, None, None, None, None, None, status) == -1
    ticket = ctypes.c_void_p(1234)
    assert lib.icp_chain_step_batched_issue(1, None, 0, None, None, None, None, None, None, None, None, status, None,
                                            ctypes.byref(ticket)) == -1
    assert ticket.value is None
    assert lib.icp_chain_step_batched_collect(None) == -1
    assert lib.icp_chain_step_batched_abandon(None) == -1
    assert lib.icp_ctx_set_rotation(None, None, None) == -1


def test_round3_entry_points_reject_bad_arguments_before_device(pkg):
    """icp_chains_run_on_device, icp_proposal_set_sampler, icp_ctx_profile_search_counters: null / empty argument lists come back with
    ICP_ERR_INVALID_ARG and touch no device (this runs without a GPU)."""
    lib, nat = pkg._native.lib(), pkg._native
    mix = nat.MhMixture(ctypes.sizeof(nat.MhMixture), (ctypes.c_double * 2)(0.5, 0.5), 0.9, 0.1, 0.1)
    assert lib.icp_chains_run_on_device(0, None, 1, None, ctypes.byref(mix), None, None, None, None, 10, None, None) == -1
    assert lib.icp_chains_run_on_device(1, None, 1, None, None, None, None, None, None, 10, None, None) == -1
    assert b"null argument" in lib.icp_last_error()
    assert lib.icp_proposal_set_sampler(None, 1) == -1
    assert lib.icp_ctx_profile_search_counters(None, 1) == -1
    assert ctypes.sizeof(nat.MhMixture) == 104  # (round 4: + w_pose and the six pose walks' sigmas; round 5: struct_size in front)


def test_round6_entry_points_reject_bad_arguments_before_device(pkg):
    """icp_chain_bind / icp_chain_bind_stats / icp_ctx_expect: null handles and bad counts are refused without touching a device; the
    hint for a device that does not exist says so instead of starting a helper thread."""
    lib = pkg._native.lib()
    assert lib.icp_chain_bind(None, 0, None) == -1 and b"null argument" in lib.icp_last_error()
    out = (ctypes.c_int64 * 3)()
    assert lib.icp_chain_bind_stats(None, out) == -1
    assert lib.icp_ctx_expect(0, -1) == -1
    assert lib.icp_ctx_expect(0, 8) == -2   # ICP_ERR_DEVICE: no HIP device in this container (on a GPU box: 0)


def test_model_arrays_are_frozen_once_the_model_key_is_taken(pkg, femur50):
    """api._model_key vouches for the arrays it hashed (icp_ctx_create_keyed: equal keys mean equal arrays): an in-place edit afterwards
    would meet the stale device copy, so the arrays go read-only and a changed model has to be a new object (advisor, round 5)."""
    import copy
    model = copy.deepcopy(femur50[0])
    k1 = pkg.api._model_key(model)
    assert k1 == pkg.api._model_key(model) and not model.basis.flags.writeable
    with pytest.raises(ValueError):
        model.basis[0, 0] += 1.0
    other = copy.deepcopy(femur50[0])
    other.basis = other.basis.copy()
    other.basis[0, 0] += 1.0
    assert pkg.api._model_key(other) != k1


def test_synthetic_target_sizes(pkg):
    """BASELINE.json configs[1]: 6-way subdivision of the femur target -> 58,322 vertices / 116,640 triangles."""
    _, big = pkg.data.synthetic_femur_target()
    assert (big.n_points, big.n_cells) == (58322, 116640)
    assert pkg.data.boundary_vertex_flags(big).sum() == 0
    _, again = pkg.data.synthetic_femur_target()
    assert np.array_equal(big.points, again.points)  # seeded
