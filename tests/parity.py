"""Observed-error log of the GPU parity tests: every model-level comparison appends
{"test", "err" (max |got - want| / max(1, |want|max)), "tol"} to $PPLHIP_PARITY_LOG (default gpurun_out/parity_errors.jsonl when
that directory exists), so that the tolerances written in the tests can be held against what the hardware actually produced
(profiles/rNN_parity_errors.jsonl is a committed copy)."""
import json
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_noise(models, step):
    """the oracle's own noise floor on this step: the same forward in a different, equally valid fp32 summation order
    (ref.MODE_ALT_ORDER) -- how far two correct fp16 implementations of the specification drift apart on these inputs.  The KV
    slabs are restored afterwards (the alternative run writes its own K/V)."""
    from oracle import ref
    saved = []
    for mm in models:
        a, b = mm.kv_array(0), mm.kv_array(1)
        saved.append((a.copy(), None if b is None else b.copy()))
    with ref.mode(ref.MODE_ALT_ORDER):
        alt = ref.forward(models, step)
    for mm, (a, b) in zip(models, saved):
        mm.kv_array(0)[:] = a
        if b is not None:
            mm.kv_array(1)[:] = b
    return alt


def record_err(name, err, tol, noise=None):
    path = os.environ.get("PPLHIP_PARITY_LOG")
    if not path:
        d = os.path.join(_ROOT, "gpurun_out")
        if not os.path.isdir(d):
            return
        path = os.path.join(d, "parity_errors.jsonl")
    try:
        with open(path, "a") as f:
            rec = {"test": name, "err": float(err), "tol": float(tol)}
            if noise is not None:
                rec["oracle_noise_floor"] = float(noise)
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
