"""Accept/reject JSON log of a chain — writer, reader, best sample, sub-sampling.

Mirrors the reference's on-disk format so that its replay tools can consume chains produced here:
  jsonLogFormat(index, name, logvalue{name -> value}, status, rigid[9], coeff[r], datetime)
      api/sampling/loggers/JSONAcceptRejectLogger.scala:35
  accept: rigid = pose parameters (translation(3), rotation(3), centre(3)), coeff = shape coefficients   :93-98
  reject: EMPTY rigid / coeff, logvalue of the CURRENT state                                           :100-106
  getBestFittingParsFromJSON: accepted record with the largest "product" value                         :142-146
  LogHelper.samplesFromLog (apps/util/LogHelper.scala:27-38): every N-th index, stepping back to the last accepted record.

Input here = the fixed-size per-step records of the host harness (host/icp_host.h): [index, status, leaf id, log value,
theta(10 + r)], which is also what the multi-GPU gather ships.  Host I/O only; nothing here touches the device."""
from __future__ import annotations

import datetime as _dt
import json
import os

import numpy as np


class JSONAcceptRejectLogger:
    def __init__(self, file_path=None):
        self.file_path = file_path
        if file_path is not None:
            parent = os.path.dirname(os.path.abspath(file_path))
            if not os.path.isdir(parent):
                raise IOError(f"JSON log path does not exist: {parent}!")   # :52-54
        self.log_status = []

    # ---- filling
    def add_records(self, records: np.ndarray, leaf_names, evaluator_name: str = "product"):
        """Append the records of icp_host_chain_run (one row per MH step)."""
        stamp = _dt.datetime.now().strftime("%Y-%m-%d %H:%M:%S")
        for rec in np.asarray(records, dtype=np.float64):
            accepted = bool(rec[1] != 0.0)
            theta = rec[4:]
            self.log_status.append({
                "index": len(self.log_status),                      # totalSamples at the time of logging (:96,104)
                "name": leaf_names[int(rec[2])],
                "logvalue": {evaluator_name: float(rec[3])},
                "status": accepted,
                # theta = [s | t(3) | phi,theta,psi | centre(3) | c(r)]  ->  rigid = t, rotation, centre (:133-140)
                "rigid": [float(v) for v in theta[1:10]] if accepted else [],
                "coeff": [float(v) for v in theta[10:]] if accepted else [],
                "datetime": stamp,
            })
        return self

    # ---- statistics (:108-110, :148-170)
    @property
    def total_samples(self):
        return len(self.log_status)

    @property
    def percent_accepted(self):
        return sum(1 for r in self.log_status if r["status"]) / max(1, len(self.log_status))

    def percent_accepted_of_type(self, name: str):
        sel = [r for r in self.log_status if r["name"] == name]
        return sum(1 for r in sel if r["status"]) / len(sel) if sel else float("nan")

    # ---- I/O (:112-127)
    def write_log(self):
        with open(self.file_path, "w") as f:
            json.dump(self.log_status, f, indent=2)

    def load_log(self):
        with open(self.file_path) as f:
            return json.load(f)

    @staticmethod
    def sample_to_model_parameters(sample, scale: float = 1.0) -> np.ndarray:
        """jsonLogFormat -> allParameters vector (:133-140); the scale parameter is not logged by the reference."""
        rigid = sample["rigid"]
        return np.concatenate([[scale], rigid[0:3], rigid[3:6], rigid[6:9], sample["coeff"]]).astype(np.float64)

    def get_best_fitting_pars_from_json(self, evaluator_name: str = "product") -> np.ndarray:
        accepted = [r for r in self.load_log() if r["status"]]
        best = max(accepted, key=lambda r: r["logvalue"][evaluator_name])
        return self.sample_to_model_parameters(best)


def samples_from_log(log, take_every_n: int = 50, total: int = 100, burn_in: int = 0):
    """LogHelper.samplesFromLog (apps/util/LogHelper.scala:27-38) — including its use of `total` as the upper index bound."""
    def get_log_index(i):
        while not log[i]["status"]:
            i -= 1
            if i < 0:
                raise IndexError("no accepted sample before the requested index")
        return i
    idx = [get_log_index(i) for i in range(burn_in, min(len(log), total), take_every_n)]
    return [(log[i], i) for i in idx][:min(total, len(idx))]
