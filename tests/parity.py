"""Observed-error log of the GPU parity tests: every model-level comparison appends
{"test", "err" (max |got - want| / max(1, |want|max)), "tol"} to $PPLHIP_PARITY_LOG (default gpurun_out/parity_errors.jsonl when
that directory exists), so that the tolerances written in the tests can be held against what the hardware actually produced
(profiles/rNN_parity_errors.jsonl is a committed copy)."""
import json
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record_err(name, err, tol):
    path = os.environ.get("PPLHIP_PARITY_LOG")
    if not path:
        d = os.path.join(_ROOT, "gpurun_out")
        if not os.path.isdir(d):
            return
        path = os.path.join(d, "parity_errors.jsonl")
    try:
        with open(path, "a") as f:
            f.write(json.dumps({"test": name, "err": float(err), "tol": float(tol)}) + "\n")
    except OSError:
        pass
