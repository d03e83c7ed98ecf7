"""The FULL-DEPTH model of the benchmark configuration against the oracle: LLaMA-2-7B dimensions, all 32 layers,
synthetic weights (the same counter-based generator on both sides), W8A16 + int8-g8 KV (config 2's arithmetic) and fp16
weights + fp16 KV (config 1's arithmetic).  A packed prefill of six prompts followed by three greedy decode steps;
device logits vs oracle/llama_ref.c at every step, plus the residual stream after every layer (pplhip_debug_run_dump vs
the oracle's hidden_dump) so that a difference can be bisected to the layer where it appears.

The oracle runs this size at ~30 tokens/s on the GPU box's 16 host cores (bench.py's cpu_baseline leg times exactly
this model), so the whole file takes about two minutes.  Every comparison appends its observed error to the parity log
(tests/parity.py); the per-layer curve goes to gpurun_out/fulldepth_layers.jsonl when that directory exists."""
import json
import os

import numpy as np
import pytest

from oracle import ref
from tests.conftest import ROOT, load_pplhip
from tests.parity import record_err

pytestmark = pytest.mark.gpu

DIMS = dict(hidden_dim=4096, intermediate_dim=11008, num_layers=32, num_heads=32, num_kv_heads=32, vocab_size=32000)
PROMPT_LENS = (33, 1, 48, 17, 5, 24)
SEED = 4321
KV_TOKENS = 1024


def _layer_log(name, errs):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "fulldepth_layers.jsonl"), "a") as f:
            f.write(json.dumps({"case": name, "rel_err_per_layer": [float(e) for e in errs]}) + "\n")


def _pair(m, wq, kvq, layers=32):
    kw = dict(DIMS)
    kw["num_layers"] = layers
    desc = ref.make_desc(max_position=2048, cache_quant_bit=kvq, cache_quant_group=8 if kvq else 1, cache_layout=3,
                         cache_mode=0, weight_quant_bit=wq, weight_quant_group=128, **kw)
    rm = ref.RefModel(desc)
    rm.init_synthetic(SEED)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=8, max_tokens_per_step=256)
    ctx.init_synthetic(0, SEED)
    ctx.kv_alloc(0, KV_TOKENS)
    return desc, rm, ctx


def _rel(got, want):
    return float(np.abs(got - want).max() / max(1.0, float(np.abs(want).max())))


def _layer_errs(got, want):
    """per layer: max |a - b| of the residual stream, relative to the layer's largest |value|"""
    return [_rel(got[l], want[l]) for l in range(got.shape[0])]


# The bar at full depth.  CORRECT fp16 implementations of the specification -- the oracle, and the oracle with its fp32 dot products
# summed in two other orders -- drift apart by ~1.3e-2 of the logit scale over these 32 layers with int8 KV (6e-3 with fp16 KV): the
# residual stream is rounded to fp16 64 times, every flipped rounding is carried to the end, and a flipped int8 cache byte is worth
# eight fp16 roundings (profiles/r03_fulldepth_probe.txt).  "Within 1e-3" therefore cannot be asked of ANY implementation at this
# depth; what is asked of the device is
#   (a) to be no further from the oracle than RATIO_NOISE x that noise floor, measured on the same inputs in the same test, for the
#       logits and for the residual stream after every layer, and
#   (b) to be no further from EXACT arithmetic (the oracle with fp32 activations and double accumulation, the mode pinned against
#       HuggingFace at 1e-5) than RATIO_EXACT x the fp16 oracles are.
# Round 4: the noise floor is the LARGEST of the three pairwise distances between three summation orders of the
# oracle (it was one distance between two), x_ref the largest of three oracle-vs-exact distances, and the ratios are FROZEN at the values
# below -- a kernel change that needs them raised is a regression to explain, not a bound to move.  Observed with three orders:
# profiles/r04_parity_errors.jsonl.
RATIO_NOISE_LOGITS, RATIO_NOISE_HIDDEN, RATIO_EXACT = 1.3, 1.6, 1.2   # observed (r04, three orders): <= 1.13 / <= 1.03
# greedy tokens: with synthetic weights the top-2 margin of a row is a few per cent of the logit scale, the order
# of the noise floor, so "equal wherever the margin is safe" compared few rows or none.  The lm_head rows of 24 chosen tokens are therefore
# ENGINEERED from the oracle's own final hidden states (teacher-forced on the chosen continuation) so that row (request r, step s) has
# its chosen token MARGIN_X safety thresholds above every other token; the greedy continuation of both sides must then BE the chosen
# tokens, and MIN_SAFE_FRACTION of the rows must be compared token for token (asserted and logged).
MARGIN_X, MIN_SAFE_FRACTION = 4.0, 0.75


def _final_norm(rm, resid):
    w = rm.get_tensor("norm.weight", np.float16).astype(np.float64)
    r = resid.astype(np.float64)
    return r / np.sqrt((r * r).mean(-1, keepdims=True) + float(rm.desc.norm_eps)) * w


def _traces(m, rm, ctx, name):
    """packed prefill + 3 greedy decode steps, five ways: oracle (the specification), device, the oracle in two other summation
    orders, oracle in exact arithmetic; the specification's greedy tokens feed all of them"""
    rng = np.random.RandomState(11)
    V = DIMS["vocab_size"]
    prompts = [rng.randint(3, V, size=n).astype(np.int64) for n in PROMPT_LENS]
    n = len(prompts)
    cache_idx = (np.arange(n) * 128).astype(np.int64)
    chosen = (1000 + 37 * np.arange(4 * n)).reshape(4, n).astype(np.int64)   # the continuation the engineered lm_head must produce

    def trace(runner, feed=None):
        tok = np.concatenate(prompts)
        seq = np.concatenate([[0], np.cumsum(PROMPT_LENS)])
        sp = np.zeros(n, dtype=np.int64)
        out = []
        for s in range(4):
            logits, dump = runner(tok, seq, sp, 0 if s == 0 else n, s)
            out.append((logits, dump, seq[1:] - 1))
            nxt = logits.argmax(-1) if feed is None else feed[s]
            sp = sp + (seq[1:] - seq[:-1])
            tok = nxt.astype(np.int64)
            seq = np.arange(n + 1)
        return out

    def oracle(tok, seq, sp, dec, s):
        return ref.forward([rm], ref.make_step(tok, seq, sp, cache_idx, dec), dump_hidden=True)

    def device(tok, seq, sp, dec, s):
        ctx.set_inputs(0, m.make_step(tok, seq, sp, cache_idx, dec, req_list_changed=int(s == 0)))
        dump = ctx.run_dump(0, len(tok))
        return ctx.copy_logits(n), dump

    # ---- phase A: teacher-forced on the chosen tokens with the synthetic lm_head: final hidden states of the 24 (row, step) pairs and a
    # first noise estimate; then the chosen tokens' lm_head rows are solved for (least-norm W with W . Y^T = T)
    rm.kv_alloc(KV_TOKENS)
    a_spec = trace(oracle, chosen)
    with ref.mode(ref.MODE_ALT_ORDER):
        rm.kv_alloc(KV_TOKENS)
        a_alt = trace(oracle, chosen)
    noise0 = max(_rel(a_alt[s][0], a_spec[s][0]) for s in range(4))
    Y = np.concatenate([_final_norm(rm, a_spec[s][1][-1][a_spec[s][2]]) for s in range(4)])      # [4 n, hidden], row p = s * n + r
    other = np.concatenate([a_spec[s][0] for s in range(4)]).astype(np.float64)                  # [4 n, V]
    other[:, chosen.reshape(-1)] = -np.inf
    top_other = other.max(-1)
    scale = max(1.0, float(np.abs(np.concatenate([a_spec[s][0] for s in range(4)])).max()))
    margin = MARGIN_X * 2 * RATIO_NOISE_LOGITS * noise0 * scale
    T = np.zeros((4 * n, 4 * n))
    T[np.arange(4 * n), np.arange(4 * n)] = top_other + margin
    W = (T @ np.linalg.inv(Y @ Y.T) @ Y).astype(np.float16)
    head = rm.get_tensor("output.weight", np.float16).reshape(V, -1).copy()
    head[chosen.reshape(-1)] = W
    rm.set_tensor("output.weight", head)
    ctx.set_tensor(0, "output.weight", head)
    _layer_log(f"{name}_engineered_head", [noise0, margin / scale, float(np.abs(W.astype(np.float32)).max())])

    # ---- phase B: the comparison proper
    rm.kv_alloc(KV_TOKENS)
    spec = trace(oracle)
    feed = [o[0].argmax(-1) for o in spec]
    dev = trace(device, feed)
    with ref.mode(ref.MODE_ALT_ORDER):
        rm.kv_alloc(KV_TOKENS)
        alt = trace(oracle, feed)
    with ref.mode(ref.MODE_ALT_ORDER2):
        rm.kv_alloc(KV_TOKENS)
        alt2 = trace(oracle, feed)
    with ref.mode(ref.MODE_FP32_ACT | ref.MODE_F64_ACC):
        rm.kv_alloc(KV_TOKENS)
        exact = trace(oracle, feed)
    return spec, dev, alt, alt2, exact, chosen


def _check(name, spec, dev, alt, alt2, exact, chosen):
    n_safe = n_rows = 0
    for s in range(4):
        e_dev = _rel(dev[s][0], spec[s][0])
        e_noise = max(_rel(alt[s][0], spec[s][0]), _rel(alt2[s][0], spec[s][0]), _rel(alt2[s][0], alt[s][0]))
        x_dev = _rel(dev[s][0], exact[s][0])
        x_ref = max(_rel(spec[s][0], exact[s][0]), _rel(alt[s][0], exact[s][0]), _rel(alt2[s][0], exact[s][0]))
        h_dev = _layer_errs(dev[s][1], spec[s][1])
        h_noise = [max(a, b, c) for a, b, c in zip(_layer_errs(alt[s][1], spec[s][1]), _layer_errs(alt2[s][1], spec[s][1]),
                                                   _layer_errs(alt2[s][1], alt[s][1]))]
        _layer_log(f"{name}_step{s}_device_vs_oracle", h_dev)
        _layer_log(f"{name}_step{s}_oracle_noise_floor", h_noise)
        record_err(f"fulldepth_{name}_step{s}_logits", e_dev, RATIO_NOISE_LOGITS * e_noise, noise=e_noise)
        record_err(f"fulldepth_{name}_step{s}_logits_vs_exact_arithmetic", x_dev, RATIO_EXACT * x_ref, noise=x_ref)
        assert h_dev[0] == 0.0                                            # embedding gather: bit exact
        assert e_dev <= RATIO_NOISE_LOGITS * e_noise, (name, s, e_dev, e_noise)
        assert x_dev <= RATIO_EXACT * x_ref, (name, s, x_dev, x_ref)
        for l in range(1, len(h_dev)):
            assert h_dev[l] <= RATIO_NOISE_HIDDEN * max(h_noise[l], 1e-3), (name, s, l, h_dev[l], h_noise[l])
        # greedy tokens: equal wherever the specification's top-2 margin is outside twice the tolerance -- by construction of the
        # lm_head (MARGIN_X) that is nearly every row, and the continuation is the chosen one
        want = spec[s][0]
        srt = np.sort(want, -1)
        safe = (srt[:, -1] - srt[:, -2]) > 2 * RATIO_NOISE_LOGITS * e_noise * max(1.0, float(np.abs(want).max()))
        n_safe += int(safe.sum())
        n_rows += len(safe)
        assert (dev[s][0].argmax(-1)[safe] == want.argmax(-1)[safe]).all(), (name, s)
        assert (want.argmax(-1)[safe] == chosen[s][safe]).all(), (name, s)
    record_err(f"fulldepth_{name}_greedy_rows_compared_fraction", n_safe / n_rows, MIN_SAFE_FRACTION)
    assert n_safe >= MIN_SAFE_FRACTION * n_rows, (name, n_safe, n_rows)


def test_7b_w8a16_int8kv_32_layers_vs_oracle():
    """config 2's arithmetic at full depth"""
    m = load_pplhip()
    desc, rm, ctx = _pair(m, wq=8, kvq=8)
    try:
        _check("w8a16_int8kv", *_traces(m, rm, ctx, "w8a16_int8kv"))
    finally:
        ctx.close()
        rm.close()


def test_7b_fp16_fp16kv_32_layers_vs_oracle():
    """config 1's arithmetic (fp16 weights, fp16 KV) on the device at full depth"""
    m = load_pplhip()
    desc, rm, ctx = _pair(m, wq=0, kvq=0)
    try:
        _check("fp16_fp16kv", *_traces(m, rm, ctx, "fp16_fp16kv"))
    finally:
        ctx.close()
        rm.close()


def test_7b_prefill_and_decode_paths_both_match_the_oracle():
    """tests/test_gpu_properties.py compares two DEVICE paths with each other at full depth (prefill of n+1 tokens vs prefill
    of n tokens + one decode step: 7e-3 apart in round 2, and a prefix-cache hit vs a cold prefill: 5.8e-3).  Here both paths
    are held against the oracle (which computes the two the same way: it always reads K/V back from the slab) and against the
    oracle's noise floor on the same prompt: the two device paths are two more correct implementations, as far from the
    oracle and from each other as the oracle is from itself in another summation order."""
    m = load_pplhip()
    desc, rm, ctx = _pair(m, wq=8, kvq=8)
    try:
        rng = np.random.RandomState(5)
        p = rng.randint(3, 32000, size=130).astype(np.int64)
        nxt = (7 * int(p[-1]) + 11) % 32000
        ext = np.concatenate([p, [nxt]])
        rm.kv_alloc(KV_TOKENS)
        want_full = ref.forward([rm], ref.make_step(ext, [0, 131], [0], [0], 0))[0]
        ref.forward([rm], ref.make_step(p, [0, 130], [0], [512], 0))
        want_dec = ref.forward([rm], ref.make_step([nxt], [0, 1], [130], [512], 1))[0]
        assert (want_dec == want_full).all()                               # the oracle: one computation either way
        with ref.mode(ref.MODE_ALT_ORDER):
            rm.kv_alloc(KV_TOKENS)
            alt_full = ref.forward([rm], ref.make_step(ext, [0, 131], [0], [0], 0))[0]
        ctx.set_inputs(0, m.make_step(ext, [0, 131], [0], [0], 0))
        ctx.run(0)
        got_full = ctx.copy_logits(1)[0]
        ctx.set_inputs(0, m.make_step(p, [0, 130], [0], [512], 0))
        ctx.run(0)
        ctx.set_inputs(0, m.make_step([nxt], [0, 1], [130], [512], 1))
        ctx.run(0)
        got_dec = ctx.copy_logits(1)[0]
        noise = _rel(alt_full, want_full)
        e_full, e_dec, e_dev = _rel(got_full, want_full), _rel(got_dec, want_full), _rel(got_dec, got_full)
        record_err("fulldepth_prefill131_vs_oracle", e_full, RATIO_NOISE_LOGITS * noise, noise=noise)
        record_err("fulldepth_prefill130_decode1_vs_oracle", e_dec, RATIO_NOISE_LOGITS * noise, noise=noise)
        record_err("fulldepth_device_prefill_vs_device_decode", e_dev, 2 * noise, noise=noise)
        assert e_full <= RATIO_NOISE_LOGITS * noise and e_dec <= RATIO_NOISE_LOGITS * noise, (e_full, e_dec, noise)
        assert e_dev <= 2 * noise, (e_dev, noise)
    finally:
        ctx.close()
        rm.close()


def _paged_pair(m, page=16, kv_tokens=1024):
    desc = ref.make_desc(max_position=2048, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=1, page_size=page,
                         weight_quant_bit=8, weight_quant_group=128, **DIMS)
    rm = ref.RefModel(desc)
    rm.init_synthetic(SEED)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=8, max_tokens_per_step=256)
    ctx.init_synthetic(0, SEED)
    ctx.kv_alloc(0, kv_tokens)
    return desc, rm, ctx


def _three_orders(rm, kv_tokens, steps):
    """the LAST step's logits of a sequence of oracle steps, in the specification's order and two other summation orders (each on a
    fresh slab): [spec, alt, alt2] and the noise floor = the largest pairwise distance"""
    outs = []
    for md in (None, ref.MODE_ALT_ORDER, ref.MODE_ALT_ORDER2):
        def run():
            rm.kv_alloc(kv_tokens)
            o = None
            for st in steps:
                o = ref.forward([rm], ref.make_step(*st))
            return o
        if md is None:
            outs.append(run())
        else:
            with ref.mode(md):
                outs.append(run())
    noise = max(_rel(outs[1], outs[0]), _rel(outs[2], outs[0]), _rel(outs[2], outs[1]))
    return outs, noise


def test_7b_cache_prefill_behind_cached_pages_vs_oracle_32_layers():
    """Config 5's cache-prefill step against the ORACLE at full depth (weak item 3: the 32-layer evidence of
    tests/test_gpu_config5_tokens.py compares the device with itself).  W8A16, int8-g8 KV on SHUFFLED 16-token pages: 96 tokens are
    prefilled cold (the cached prefix: 6 pages), then the remaining 35 tokens run as a cache-prefill step at start_pos 96
    (/root/reference/src/engine/llm_engine.cc:114, /root/reference/src/generator/llm_generator.cc:233-241) on pages of their own.  Each
    side fills its own cache (no slab copied across).  Bar: 1.3 x the noise floor measured on the same two steps with three summation
    orders of the oracle -- the frozen full-depth ratio; the oracle's cold prefill of all 131 tokens gives the same bits (it always reads
    K / V back from the slab), asserted."""
    m = load_pplhip()
    desc, rm, ctx = _paged_pair(m)
    try:
        rng = np.random.RandomState(23)
        prompt = rng.randint(3, DIMS["vocab_size"], size=131).astype(np.int64)
        npg = 9                                                                     # ceil(131 / 16)
        pages = rng.permutation(KV_TOKENS // 16)[:npg].reshape(1, npg).astype(np.int64)
        steps = [(prompt[:96], [0, 96], [0], pages, 0, npg), (prompt[96:], [0, 35], [96], pages, 0, npg)]
        (want, alt, alt2), noise = _three_orders(rm, KV_TOKENS, steps)
        rm.kv_alloc(KV_TOKENS)
        cold = ref.forward([rm], ref.make_step(prompt, [0, 131], [0], pages, 0, npg))
        assert (cold == want).all()                                                  # the oracle: one computation either way
        ctx.set_inputs(0, m.make_step(*steps[0]))
        ctx.run(0)
        ctx.set_inputs(0, m.make_step(*steps[1]))
        ctx.run(0, cache_prefill=1)
        got = ctx.copy_logits(1)
        err = _rel(got, want)
        record_err("fulldepth_cache_prefill_35_behind_96_cached_paged_vs_oracle", err, RATIO_NOISE_LOGITS * noise, noise=noise)
        assert err <= RATIO_NOISE_LOGITS * noise, (err, noise)
        # the int8 K / V bytes both steps wrote: layer 0's (cache layout 3 = [L, 2, h, N, d]: the first 2 h N d bytes) see inputs that are
        # equal on both sides up to one RMSNorm + GEMM -- a few LSB on few bytes; deeper layers inherit the residual stream's drift
        # (the logits bound above is what holds them), so there only the fraction of differing bytes is recorded
        gk, rk = ctx.kv_read(0, 0), rm.kv_array(0)
        n0 = 2 * DIMS["num_kv_heads"] * KV_TOKENS * (DIMS["hidden_dim"] // DIMS["num_heads"])
        assert (np.abs(gk[:n0].astype(np.int32) - rk[:n0].astype(np.int32)) <= 3).all()
        assert (gk[:n0] != rk[:n0]).mean() < 0.01
        record_err("fulldepth_cache_prefill_kv_bytes_differing_fraction_all_layers", float((gk != rk).mean()), 1.0)
    finally:
        ctx.close()
        rm.close()


@pytest.mark.parametrize("as_decode_row", [True, False])
def test_7b_full_prefix_hit_single_row_vs_oracle_32_layers(as_decode_row):
    """A FULL prefix-cache hit at full depth: every page of the 128-token prompt is cached, so the generator re-runs only the last token at
    start_pos = len - 1 (/root/reference/src/generator/llm_generator.cc:233-236) -- ONE row over 127 cached keys + itself.  The row goes
    through the decode-attention kernel when the step counts it as a decoding request, through the cache-prefill kernel (one query row)
    otherwise; both are held against the oracle's logits of the same step AND against the cold prefill's last row, inside 1.3 x the
    oracle's noise floor on the same steps (three summation orders)."""
    m = load_pplhip()
    desc, rm, ctx = _paged_pair(m)
    try:
        rng = np.random.RandomState(29)
        n = 128
        prompt = rng.randint(3, DIMS["vocab_size"], size=n).astype(np.int64)
        npg = n // 16
        pages = rng.permutation(KV_TOKENS // 16)[:npg].reshape(1, npg).astype(np.int64)
        dec = 1 if as_decode_row else 0
        steps = [(prompt, [0, n], [0], pages, 0, npg), (prompt[-1:], [0, 1], [n - 1], pages, dec, npg)]
        (want, alt, alt2), noise = _three_orders(rm, KV_TOKENS, steps)
        rm.kv_alloc(KV_TOKENS)
        cold = ref.forward([rm], ref.make_step(*steps[0]))
        assert (cold == want).all()                                                  # oracle: the hit row recomputes the same bits
        ctx.set_inputs(0, m.make_step(*steps[0]))
        ctx.run(0)
        got_cold = ctx.copy_logits(1).copy()
        ctx.set_inputs(0, m.make_step(*steps[1]))
        ctx.run(0, cache_prefill=0 if as_decode_row else 1)
        got = ctx.copy_logits(1)
        tag = "decode_kernel" if as_decode_row else "cache_prefill_kernel"
        e_hit, e_cold, e_dev = _rel(got, want), _rel(got_cold, want), _rel(got, got_cold)
        record_err(f"fulldepth_full_prefix_hit_row_{tag}_vs_oracle", e_hit, RATIO_NOISE_LOGITS * noise, noise=noise)
        record_err(f"fulldepth_full_prefix_hit_row_{tag}_vs_device_cold_prefill", e_dev, 2 * noise, noise=noise)
        assert e_cold <= RATIO_NOISE_LOGITS * noise, (e_cold, noise)
        assert e_hit <= RATIO_NOISE_LOGITS * noise, (e_hit, noise)
        assert e_dev <= 2 * noise, (e_dev, noise)                                    # two device paths: two more samples of the same noise
    finally:
        ctx.close()
        rm.close()
