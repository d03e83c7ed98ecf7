"""Diagnosis (not a test): per-layer error curves of the full-depth 7B model on the device against the oracle, beside the
oracle's own noise floor (alternative summation order) and its distance from exact arithmetic (fp32 activations + double
accumulation).  python profiles/probes/fulldepth_probe.py [wq kvq]  ->  gpurun_out/fulldepth_probe_<wq>_<kvq>.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests.conftest import load_pplhip  # noqa: E402

DIMS = dict(hidden_dim=4096, intermediate_dim=11008, num_layers=32, num_heads=32, num_kv_heads=32, vocab_size=32000)
LENS = (33, 1, 48, 17, 5, 24)


def rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, float(np.abs(b).max())))


def layers(a, b):
    return [rel(a[l], b[l]) for l in range(a.shape[0])]


def main():
    wq, kvq = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 8)
    m = load_pplhip()
    desc = ref.make_desc(max_position=2048, cache_quant_bit=kvq, cache_quant_group=8 if kvq else 1, cache_layout=3, cache_mode=0,
                         weight_quant_bit=wq, weight_quant_group=128, **DIMS)
    rm = ref.RefModel(desc)
    rm.init_synthetic(4321)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=8, max_tokens_per_step=256)
    ctx.init_synthetic(0, 4321)
    ctx.kv_alloc(0, 1024)
    rng = np.random.RandomState(11)
    prompts = [rng.randint(3, 32000, size=n).astype(np.int64) for n in LENS]
    n = len(prompts)
    cache_idx = (np.arange(n) * 128).astype(np.int64)

    def trace(runner, feed=None):
        tok = np.concatenate(prompts)
        seq = np.concatenate([[0], np.cumsum(LENS)])
        sp = np.zeros(n, dtype=np.int64)
        out = []
        for s in range(4):
            dec = 0 if s == 0 else n
            logits, dump = runner(tok, seq, sp, dec, s)
            out.append((logits, dump))
            nxt = logits.argmax(-1) if feed is None else feed[s]
            sp = sp + (seq[1:] - seq[:-1])
            tok = nxt.astype(np.int64)
            seq = np.arange(n + 1)
        return out

    def oracle_runner(tok, seq, sp, dec, s):
        return ref.forward([rm], ref.make_step(tok, seq, sp, cache_idx, dec), dump_hidden=True)

    def device_runner(tok, seq, sp, dec, s):
        ctx.set_inputs(0, m.make_step(tok, seq, sp, cache_idx, dec, req_list_changed=int(s == 0)))
        d = ctx.run_dump(0, len(tok))
        return ctx.copy_logits(n), d

    res = {}
    with ref.mode(0):
        rm.kv_alloc(1024)
        o0 = trace(oracle_runner)
    feed = [o[0].argmax(-1) for o in o0]
    dev = trace(device_runner, feed)
    with ref.mode(ref.MODE_ALT_ORDER):
        rm.kv_alloc(1024)
        o4 = trace(oracle_runner, feed)
    with ref.mode(ref.MODE_FP32_ACT | ref.MODE_F64_ACC):
        rm.kv_alloc(1024)
        tr = trace(oracle_runner, feed)
    for s in range(4):
        res[f"step{s}"] = {
            "logits": {"dev_vs_o0": rel(dev[s][0], o0[s][0]), "o4_vs_o0": rel(o4[s][0], o0[s][0]),
                       "dev_vs_exact": rel(dev[s][0], tr[s][0]), "o0_vs_exact": rel(o0[s][0], tr[s][0]),
                       "o4_vs_exact": rel(o4[s][0], tr[s][0])},
            "hidden_dev_vs_o0": layers(dev[s][1], o0[s][1]), "hidden_o4_vs_o0": layers(o4[s][1], o0[s][1]),
            "hidden_dev_vs_exact": layers(dev[s][1], tr[s][1]), "hidden_o0_vs_exact": layers(o0[s][1], tr[s][1]),
            "greedy_agree_dev_o0": int((dev[s][0].argmax(-1) == o0[s][0].argmax(-1)).sum()),
        }
        print(s, json.dumps(res[f"step{s}"]["logits"]))
        for k in ("hidden_dev_vs_o0", "hidden_o4_vs_o0", "hidden_dev_vs_exact", "hidden_o0_vs_exact"):
            print("   ", k, " ".join(f"{e*1e3:.2f}" for e in res[f"step{s}"][k][::4]), f"| last {res[f'step{s}'][k][-1]*1e3:.2f}  (x1e-3)")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"fulldepth_probe_{wq}_{kvq}.json"), "w"))


if __name__ == "__main__":
    main()
