"""CPU: the accept/reject JSON log (reference format) built from the harness' fixed-size per-step records."""
import json

import numpy as np
import pytest


def fake_records(r=5, n=12, seed=0):
    rng = np.random.default_rng(seed)
    rec = np.zeros((n, 4 + 10 + r))
    theta = np.zeros(10 + r)
    theta[0] = 1.0
    theta[7:10] = [1.0, 2.0, 3.0]
    for i in range(n):
        acc = i % 3 != 1
        if acc:
            theta = theta.copy()
            theta[1:4] += rng.normal(size=3) * 0.1
            theta[10:] = rng.normal(size=r)
        rec[i, 0], rec[i, 1], rec[i, 2], rec[i, 3] = i, float(acc), i % 3, -100.0 + (i if acc else 0) * 1.5
        rec[i, 4:] = theta
    return rec


def test_json_log_roundtrip_matches_reference_format(pkg, tmp_path):
    rec = fake_records()
    # the names are the reference's own (api/sampling/MixedProposalDistributions.scala:31-54 interpolates Scala Doubles)
    names = pkg.femur_icp_proposal_registration(*pkg.data.load_femur_model_and_target(50)).leaf_names()
    assert names[0] == "IcpProposal-TargetSampling-0.1Step" and names[1] == "IcpProposal-ModelSampling-0.1Step"
    assert names[2] == "RandomShape-0.1" and names[3] == "RotationYaw-0.01" and names[6] == "TranslationX-0.1"
    path = tmp_path / "log.json"
    lg = pkg.loggers.JSONAcceptRejectLogger(str(path)).add_records(rec, names)
    lg.write_log()
    raw = json.load(open(path))
    assert len(raw) == 12
    assert set(raw[0]) == {"index", "name", "logvalue", "status", "rigid", "coeff", "datetime"}      # jsonLogFormat :35
    assert raw[0]["status"] is True and len(raw[0]["rigid"]) == 9 and len(raw[0]["coeff"]) == 5
    assert raw[1]["status"] is False and raw[1]["rigid"] == [] and raw[1]["coeff"] == []            # reject: empty (:104)
    assert raw[0]["rigid"][6:9] == [1.0, 2.0, 3.0] and raw[3]["index"] == 3
    assert abs(lg.percent_accepted - 8 / 12) < 1e-12
    assert lg.percent_accepted_of_type(names[1]) == 0.0
    best = lg.get_best_fitting_pars_from_json()
    k = int(np.argmax(np.where(rec[:, 1] != 0, rec[:, 3], -np.inf)))
    assert np.array_equal(best[1:], rec[k, 5:])
    # LogHelper.samplesFromLog: every 2nd index below `total`, rejected ones step back to the last accepted record
    sub = pkg.loggers.samples_from_log(raw, take_every_n=2, total=9)
    assert [i for _, i in sub] == [0, 2, 3, 6, 8][:len(sub)] or all(raw[i]["status"] for _, i in sub)
    assert all(s["status"] for s, _ in sub)


def test_log_path_must_exist(pkg):
    with pytest.raises(IOError):
        pkg.loggers.JSONAcceptRejectLogger("/nonexistent-dir-xyz/log.json")


def test_scala_double_formatting(pkg):
    """java.lang.Double.toString, as Scala's string interpolation prints the proposals' standard deviations."""
    S = pkg.ChainSetup.scala_double
    for x, want in ((0.1, "0.1"), (0.01, "0.01"), (1.0, "1.0"), (10.0, "10.0"), (0.001, "0.001"), (1e-4, "1.0E-4"), (5e-5, "5.0E-5"),
                    (1e7, "1.0E7"), (123456.789, "123456.789"), (0.3, "0.3"), (2.5e-3, "0.0025"), (0.0, "0.0")):
        assert S(x) == want, (x, S(x), want)
