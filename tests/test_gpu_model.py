"""End-to-end parity of the decoder forward (pplhip_set_inputs / pplhip_run / pplhip_sample) against the CPU oracle
and the HF golden vectors: packed ragged prefill, decode steps, all cache layouts/modes, fp16 and int8 KV,
fp16 / W8A16 / W4A16 weights.

Tolerance on logits (round 3).  The north star asks for "logits within 1e-3 fp16".  Two CORRECT fp16 implementations of this
specification do not always meet that between themselves: every model-level comparison here also runs the oracle a second time
with its fp32 dot products summed in another order (tests/parity.py oracle_noise, ref.MODE_ALT_ORDER) and records how far the two
oracles are apart -- the noise floor of the case (0.2e-3 .. 2e-3 on these tiny models, 1.3e-2 at 32 layers,
tests/test_gpu_fulldepth.py).  The device must satisfy BOTH
    |d| <= max(1e-3, NOISE_RATIO x noise floor) x max(1, |logit|max)      (no further from the oracle than ~2 oracles from each other)
    |d| <= 1e-3 x k x max(1, |logit|max)                                  (a fixed cap per test, k from the observed errors)
and greedy tokens must be equal wherever the oracle's top-2 margin exceeds twice the tolerance.  Observed errors and noise floors
of every comparison: profiles/r03_parity_errors.jsonl.  Causes of the larger k:
  * int8-g8 KV: quantisation is discontinuous -- K/V inputs that differ from the oracle's by one fp16 rounding can flip a cache
    byte by one LSB (test_hf_fixture_model checks <= 3 LSB on < 5 % of the bytes), worth eight fp16 roundings;
  * the prefill attention kernel feeds dequantised int8 K/V to the MFMA as fp16 (one rounding of q x scale; P is an exact hi + lo
    pair since round 3, and the grouped-query decode kernel is exact in all three operands)."""
import os
import tempfile

import numpy as np
import pytest

from oracle import ref
from tests.conftest import load_pplhip
from tests.parity import oracle_noise, record_err
from tests.test_oracle_hf import desc_from_meta, load_fixture

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
# imported at collection time, before any device work: importing `transformers` (dozens of extension modules) late, from inside a
# test of a process whose HIP runtime and OpenMP pools are already running, segfaulted intermittently (2 runs of 6, round 3)
try:
    from tests import test_export_hf as te
except (Exception, pytest.skip.Exception):   # transformers missing
    te = None


def plan_cache(desc, lens_total, max_tokens, seed=0):
    n = len(lens_total)
    lens_total = np.asarray(lens_total)
    if desc.cache_mode == 0:
        idx = np.concatenate([[0], np.cumsum(lens_total)[:-1]]).astype(np.int64)
        return idx, 0
    P = desc.page_size
    npg = (lens_total + P - 1) // P
    mp = int(npg.max())
    idx = np.full((n, mp), np.iinfo(np.int64).max, dtype=np.int64)
    order = np.random.RandomState(seed).permutation(max_tokens // P)
    k = 0
    for i in range(n):
        idx[i, :npg[i]] = order[k:k + npg[i]]
        k += npg[i]
    return idx, mp


def generate_both(m, ctx, models, desc, prompts, steps, max_tokens):
    """runs `steps` greedy steps (packed prefill, then decode) on the device and on the oracle with the SAME inputs
    (the oracle's tokens drive both, so a single flipped argmax cannot cascade)."""
    n = len(prompts)
    lens = np.array([len(p) for p in prompts])
    cache_idx, max_pages = plan_cache(desc, lens + steps, max_tokens)
    tok = np.concatenate(prompts).astype(np.int64)
    seq_starts = np.concatenate([[0], np.cumsum(lens)])
    start_pos = np.zeros(n, dtype=np.int64)
    res = []
    for s in range(steps):
        dec = 0 if s == 0 else n
        st_r = ref.make_step(tok, seq_starts, start_pos, cache_idx, dec, max_pages)
        st_g = m.make_step(tok, seq_starts, start_pos, cache_idx, dec, max_pages, req_list_changed=1 if s == 0 else 0)
        want = ref.forward(models, st_r)
        alt = oracle_noise(models, st_r)
        ctx.set_inputs(0, st_g)
        ctx.run(0)
        gtok, glp = ctx.sample(n, top_k=1)
        got = ctx.copy_logits(n)
        wtok, wlp = ref.sample(want, top_k=1)
        res.append((got, want, gtok, wtok, glp, wlp, alt))
        start_pos = start_pos + (seq_starts[1:] - seq_starts[:-1])
        tok = wtok.astype(np.int64)
        seq_starts = np.arange(n + 1)
    return res


NOISE_RATIO = 2.5
MIN_SAFE_FRACTION = 0.75


def check_steps(res, k, name=None):
    if name is None:
        name = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]
    # the noise floor of the TRACE (largest over its steps): the device's KV slab carries its differences from step to step
    noise = max(float(np.abs(r[6] - r[1]).max() / max(1.0, np.abs(r[1]).max())) for r in res)
    n_safe = n_rows = 0
    for s, (got, want, gtok, wtok, glp, wlp, alt) in enumerate(res):
        scale = max(1.0, np.abs(want).max())
        rel = min(1e-3 * k, max(1e-3, NOISE_RATIO * noise)) if noise > 1e-5 else 1e-3 * k   # (integer GEMMs have no summation-order noise)
        tol = rel * scale
        err = np.abs(got - want).max()
        record_err(name, err / scale, rel, noise=noise)
        assert err <= tol, (s, err / scale, rel, noise)
        srt = np.sort(want, -1)
        safe = (srt[:, -1] - srt[:, -2]) > 2 * tol
        n_safe += int(safe.sum())
        n_rows += len(safe)
        assert (gtok[safe] == wtok[safe]).all(), s
        if safe.any():
            assert np.abs(glp[safe] - wlp[safe]).max() < 5 * tol
    # the greedy-token comparison must not be vacuous: most rows of the trace have a top-2 margin outside
    # twice the tolerance and ARE compared token for token; the fraction goes to the parity log
    record_err(name + ":greedy_rows_compared_fraction", n_safe / n_rows, MIN_SAFE_FRACTION)
    assert n_safe >= MIN_SAFE_FRACTION * n_rows, (name, n_safe, n_rows)


@pytest.mark.parametrize("name", ["mha", "gqa"])
@pytest.mark.parametrize("layout,mode,quant", [(3, 0, 0), (0, 0, 8), (1, 1, 0), (2, 1, 8), (3, 1, 8)])
def test_hf_fixture_model(golden_dir, name, layout, mode, quant):
    m = load_pplhip()
    meta, weights, prompts, hf_logits, hf_tokens, _ = load_fixture(os.path.join(golden_dir, f"hf_tiny_{name}.npz"))
    kw = dict(cache_layout=layout, cache_mode=mode, page_size=4 if mode else 0, cache_quant_bit=quant,
              cache_quant_group=8 if quant else 1)
    desc = desc_from_meta(meta, **kw)
    gdesc = m.copy_desc(desc)
    rm = ref.RefModel(desc)
    ctx = m.Context(gdesc, max_running_batch=8, max_tokens_per_step=64)
    for k, v in weights.items():
        rm.set_tensor(k, v)
        ctx.set_tensor(0, k, v)
    max_tokens = 256
    rm.kv_alloc(max_tokens)
    ctx.kv_alloc(0, max_tokens)
    steps = hf_logits.shape[1]
    res = generate_both(m, ctx, [rm], desc, prompts, steps, max_tokens)
    # observed (r03): mha 0.74e-3 / 1.28e-3 (fp16 / int8 KV; noise floor 0.49e-3 / 0.67e-3), gqa 1.34e-3 / 3.27e-3 (0.83e-3 / 2.05e-3)
    check_steps(res, k={("mha", 0): 1.5, ("mha", 8): 2, ("gqa", 0): 2, ("gqa", 8): 4.5}[(name, quant)])
    if quant == 0:
        # and against the independent HF vectors (fp32 model vs fp16 activations)
        got = np.stack([r[0] for r in res], 1)
        assert np.abs(got - hf_logits).max() < 2e-2 * max(1.0, np.abs(hf_logits).max())
    # KV slab written by the device equals the oracle's (fp16 bits / int8 bytes + scales), up to rare rounding ties
    gk, rk = ctx.kv_read(0, 0), rm.kv_array(0)
    if quant == 0:
        # the device computed k, v from ITS hidden states (fp32 accumulation order differs): same tolerance class as logits
        d = np.abs(gk.astype(np.float32) - rk.astype(np.float32))
        assert d.max() <= 4e-3 * max(1.0, np.abs(rk.astype(np.float32)).max()), d.max()
        assert ((gk != 0) == (rk != 0)).mean() > 0.999  # same slots written
    else:
        # int8 bytes: inputs differ by fp16 rounding noise, so allow a few LSB on a small fraction of bytes
        assert (np.abs(gk.astype(np.int32) - rk.astype(np.int32)) <= 3).all()
        assert (gk != rk).mean() < 0.05
    ctx.close()


@pytest.mark.parametrize("wq,kvq,mode,inter", [(8, 8, 1, 512), (4, 8, 0, 512), (0, 0, 0, 512), (8, 0, 1, 512),
                                               (8, 8, 0, 176), (0, 0, 1, 336)])
def test_synthetic_model(wq, kvq, mode, inter):
    """synthetic weights generated ON THE DEVICE equal the oracle's generator (else logits could not agree).
    inter % 64 != 0 (a tensor-parallel slice such as 11008 / 8 = 1376) exercises the zero-padded w2 rows."""
    m = load_pplhip()
    desc = ref.make_desc(hidden_dim=256, intermediate_dim=inter, num_layers=2, num_heads=4, num_kv_heads=4, vocab_size=1024,
                         max_position=512, cache_quant_bit=kvq, cache_quant_group=8 if kvq else 1, cache_layout=3,
                         cache_mode=mode, page_size=16 if mode else 0, weight_quant_bit=wq, weight_quant_group=128)
    rm = ref.RefModel(desc)
    rm.init_synthetic(1234)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=16, max_tokens_per_step=256)
    ctx.init_synthetic(0, 1234)
    rm.kv_alloc(1024)
    ctx.kv_alloc(0, 1024)
    rng = np.random.RandomState(7)
    prompts = [rng.randint(3, 1024, size=n) for n in (70, 3, 129, 1, 16)]
    res = generate_both(m, ctx, [rm], desc, prompts, 4, 1024)
    check_steps(res, k=1)   # observed (r02): <= 3.5e-4 for every weight / KV format
    ctx.close()


def test_uploaded_weights_with_padded_w2_rows():
    """upload path (pplhip_rank_set_tensor) for a slice whose intermediate size is not a multiple of the GEMM k-tile:
    the oracle's tensors are uploaded byte for byte; w2 lands in zero-padded rows on the device."""
    m = load_pplhip()
    desc = ref.make_desc(hidden_dim=256, intermediate_dim=176, num_layers=2, num_heads=4, num_kv_heads=2, vocab_size=512,
                         max_position=256, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=0,
                         weight_quant_bit=8)
    rm = ref.RefModel(desc)
    rm.init_synthetic(77)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=8, max_tokens_per_step=128)
    for name in ref.tensor_names(desc):
        ctx.set_tensor(0, name, rm.get_tensor(name, np.uint8))
    rm.kv_alloc(512)
    ctx.kv_alloc(0, 512)
    rng = np.random.RandomState(3)
    prompts = [rng.randint(3, 512, size=n) for n in (33, 2, 65)]
    check_steps(generate_both(m, ctx, [rm], desc, prompts, 3, 512), k=1)   # observed 2.6e-4
    ctx.close()


@pytest.mark.parametrize("init", ["all", "unique_id"])
def test_comm_path_and_chunked_overlap_single_gpu(monkeypatch, init):
    """PPLHIP_FORCE_COMM=1 builds the RCCL communicator at world size 1 (every collective is an identity) so that the
    call sequence of the tensor-parallel step -- all-reduce after wo and w2, all-gather of the logits, the
    communication stream and its events, the two-chunk schedule -- runs on a one-GPU box.  Same logits as the plain
    path: bit for bit with the collectives on the compute stream, oracle tolerance for the chunked schedule (the
    chunks may take another GEMM kernel)."""
    m = load_pplhip()
    desc = ref.make_desc(hidden_dim=256, intermediate_dim=512, num_layers=3, num_heads=4, num_kv_heads=2, vocab_size=1024,
                         max_position=512, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=1, page_size=16,
                         weight_quant_bit=8)
    rng = np.random.RandomState(11)
    prompts = [rng.randint(3, 1024, size=n) for n in (40, 3, 129, 1, 16, 77)]
    rm = ref.RefModel(desc)
    rm.init_synthetic(99)
    rm.kv_alloc(2048)

    def run(env):
        for k in ("PPLHIP_FORCE_COMM", "PPLHIP_TP_OVERLAP", "PPLHIP_TP_OVERLAP_MIN_TOKENS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        uid = m.get_unique_id() if (env and init == "unique_id") else None
        ctx = m.Context(m.copy_desc(desc), max_running_batch=16, max_tokens_per_step=512, unique_id=uid)
        ctx.init_synthetic(0, 99)
        ctx.kv_alloc(0, 2048)
        rm_kv = rm.kv_array(0); rm_kv[:] = 0
        res = generate_both(m, ctx, [rm], desc, prompts, 4, 2048)
        ctx.close()
        return res

    plain = run({})
    same_stream = run({"PPLHIP_FORCE_COMM": "1", "PPLHIP_TP_OVERLAP": "0"})
    chunked = run({"PPLHIP_FORCE_COMM": "1", "PPLHIP_TP_OVERLAP": "1", "PPLHIP_TP_OVERLAP_MIN_TOKENS": "2"})
    for a, b in zip(plain, same_stream):
        assert (a[0] == b[0]).all() and (a[2] == b[2]).all()
    check_steps(chunked, k=1)   # observed 5.4e-4
    for a, b in zip(plain, chunked):
        assert np.abs(a[0] - b[0]).max() <= 1e-3 * max(1.0, np.abs(a[0]).max())


def test_container_load_and_errors(golden_dir):
    m = load_pplhip()
    meta, weights, prompts, hf_logits, _, _ = load_fixture(os.path.join(golden_dir, "hf_tiny_mha.npz"))
    desc = desc_from_meta(meta)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=4, max_tokens_per_step=32)
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "model_slice_0"))
        m.write_container(os.path.join(td, "model_slice_0", "weights.pplhip"), weights)
        ctx.load(0, os.path.join(td, "model_slice_0"))
        with pytest.raises(m.PplHipError):
            ctx.load(0, os.path.join(td, "model_slice_1"))
    with pytest.raises(m.PplHipError):
        ctx.set_tensor(0, "no.such.tensor", np.zeros(4, dtype=np.float16))
    with pytest.raises(m.PplHipError):
        ctx.set_tensor(0, "norm.weight", np.zeros(3, dtype=np.float16))
    ctx.kv_alloc(0, 64)
    p = prompts[0]
    st = m.make_step(p, [0, len(p)], [0], [0], 0)
    ctx.set_inputs(0, st)
    ctx.run(0)
    got = ctx.copy_logits(1)
    assert np.abs(got[0] - hf_logits[0, 0]).max() < 2e-2 * max(1.0, np.abs(hf_logits).max())
    # a step larger than the context was sized for is rejected, not truncated
    big = m.make_step(np.zeros(40, dtype=np.int64), [0, 40], [0], [0], 0)
    with pytest.raises(m.PplHipError):
        ctx.set_inputs(0, big)
    ctx.close()


def test_prefix_cache_hit_equals_cold_prefill():
    """cache-prefill (K7): re-using cached pages for a shared prefix gives the same logits as a cold prefill,
    bit for bit in the KV slab and within fp16 rounding in the logits (SURVEY.md section 10 worked example)."""
    m = load_pplhip()
    desc = ref.make_desc(hidden_dim=256, intermediate_dim=512, num_layers=2, num_heads=4, num_kv_heads=2, vocab_size=512,
                         max_position=512, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=1, page_size=4,
                         weight_quant_bit=8)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=4, max_tokens_per_step=64)
    ctx.init_synthetic(0, 5)
    ctx.kv_alloc(0, 256)
    rng = np.random.RandomState(1)
    prompt = rng.randint(3, 512, size=10)
    I64MAX = np.iinfo(np.int64).max
    # cold: pages [7, 9, 11]
    st = m.make_step(prompt, [0, 10], [0], np.array([[7, 9, 11]]), 0, max_pages=3)
    ctx.set_inputs(0, st); ctx.run(0)
    cold = ctx.copy_logits(1)
    # hit: first 8 tokens (pages 7, 9) are cached; only tokens 8..9 are fed, start_pos = 8, new page 20
    st = m.make_step(prompt[8:], [0, 2], [8], np.array([[7, 9, 20]]), 0, max_pages=3)
    ctx.set_inputs(0, st); ctx.run(0, cache_prefill=1)
    hit = ctx.copy_logits(1)
    assert np.abs(cold - hit).max() <= 2e-3 * max(1.0, np.abs(cold).max())
    assert cold.argmax() == hit.argmax()
    ctx.close()


def test_penalty_and_sampling_through_the_abi():
    """K12: repetition / presence / frequency penalties + temperature on the logits of a real step, count map kept
    across steps per batch slot (src/backends/cuda/post_processor.cc:221-281), against ref_penalty."""
    import ctypes as C
    m = load_pplhip()
    desc = ref.make_desc(hidden_dim=256, intermediate_dim=512, num_layers=1, num_heads=4, num_kv_heads=4, vocab_size=1024,
                         max_position=256, cache_quant_bit=0, cache_quant_group=1, cache_layout=3, cache_mode=0)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=8, max_tokens_per_step=64, enable_penalty=True)
    ctx.init_synthetic(0, 3)
    ctx.kv_alloc(0, 256)
    rng = np.random.RandomState(2)
    prompts = [rng.randint(3, 1024, size=n) for n in (12, 5, 9)]
    n = len(prompts)
    lens = np.array([len(p) for p in prompts])
    slots = np.array([5, 0, 2], dtype=np.int64)
    temps = np.array([0.7, 1.0, 1.3], dtype=np.float32)
    rep = np.array([1.2, 1.0, 1.5], dtype=np.float32)
    pres = np.array([0.1, 0.0, 0.3], dtype=np.float32)
    freq = np.array([0.05, 0.2, 0.0], dtype=np.float32)
    count_map = np.zeros((8, 1024), dtype=np.uint16)
    tok = np.concatenate(prompts).astype(np.int64)
    seq = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    sp = np.zeros(n, dtype=np.int64)
    ci = np.array([0, 64, 128], dtype=np.int64)
    for s in range(3):
        ctx.set_inputs(0, m.make_step(tok, seq, sp, ci, 0 if s == 0 else n, req_list_changed=int(s == 0)))
        ctx.run(0)
        raw = ctx.copy_logits(n)
        ctx.penalty(temps, rep, pres, freq, slots, req_list_changed=(s == 0))
        got = ctx.copy_logits(n)
        want = raw.copy()
        ref.lib().ref_penalty(want.ctypes.data, temps.ctypes.data, rep.ctypes.data, pres.ctypes.data, freq.ctypes.data,
                              slots.ctypes.data, tok.ctypes.data, seq.ctypes.data, sp.ctypes.data, n, 1024, count_map.ctypes.data)
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
        gtok, _ = ctx.sample(n, top_k=1, temperatures=temps, enable_penalty=True, req_list_changed=(s == 0))
        wtok, _ = ref.sample(want, top_k=1)
        srt = np.sort(want, -1)
        safe = (srt[:, -1] - srt[:, -2]) > 1e-4
        assert (gtok[safe] == wtok[safe]).all()
        sp = sp + (seq[1:] - seq[:-1])
        tok = wtok.astype(np.int64)
        seq = np.arange(n + 1, dtype=np.int64)
    ctx.close()


def test_exported_hf_checkpoint_runs_on_the_device(tmp_path_factory):
    """HF checkpoint -> export_hf_llama.py (W8A16) -> pplhip_rank_load on the device == the oracle on the same containers."""
    if te is None:
        pytest.skip("transformers is not importable")
    import json
    m = load_pplhip()
    d, prompt, _ = te.make_hf_checkpoint(tmp_path_factory.mktemp("hf"))
    out = str(tmp_path_factory.mktemp("exported"))
    te.exp.main(["--model-dir", d, "--out", out, "--quant", "w8a16", "--cache-quant-bit", "8"])
    want = te.oracle_logits(out, 1, prompt)
    p = json.load(open(os.path.join(out, "params.json")))
    desc = m.make_desc(hidden_dim=p["hidden_dim"], intermediate_dim=p["intermediate_dim"], num_layers=p["num_layers"],
                       num_heads=p["num_heads"], num_kv_heads=p["num_kv_heads"], vocab_size=p["vocab_size"], max_position=p["max_position"],
                       cache_quant_bit=8, cache_quant_group=8, cache_layout=p["cache_layout"], cache_mode=p["cache_mode"], page_size=0,
                       weight_quant_bit=8, weight_quant_group=p["weight_quant_group"], norm_eps=p["norm_eps"], rope_theta=p["rope_theta"])
    ctx = m.Context(desc, max_running_batch=4, max_tokens_per_step=32)
    ctx.load(0, os.path.join(out, "model_slice_0"))
    ctx.kv_alloc(0, 64)
    ctx.set_inputs(0, m.make_step(np.array(prompt), [0, len(prompt)], [0], [0], 0))
    ctx.run(0)
    got = ctx.copy_logits(1)[0]
    record_err("exported_hf_checkpoint", np.abs(got - want).max() / max(1.0, np.abs(want).max()), 2.5e-3)   # observed (r02) 1.7e-3, int8 KV
    assert np.abs(got - want).max() <= 2.5e-3 * max(1.0, np.abs(want).max())
    ctx.close()


@pytest.mark.parametrize("wq,kvq,batch", [(8, 8, 1), (8, 8, 3), (4, 0, 2), (0, 0, 4), (8, 8, 5)])
def test_small_batch_decode_on_the_streaming_gemv(wq, kvq, batch):
    """decode steps of 1..5 rows (the streaming GEMV of k_gemv.hip up to its row limit, the half-height tiles above it) against the oracle.
    (Round 4 also ran these steps with both RMSNorms folded into the GEMVs -- bit-identical, measured slower; that variant and its switch
    were removed in round 5.)"""
    m = load_pplhip()
    desc = ref.make_desc(hidden_dim=512, intermediate_dim=1408, num_layers=3, num_heads=4, num_kv_heads=4, vocab_size=1024,
                         max_position=512, cache_quant_bit=kvq, cache_quant_group=8 if kvq else 1, cache_layout=3,
                         cache_mode=0, weight_quant_bit=wq, weight_quant_group=128)
    rm = ref.RefModel(desc)
    rm.init_synthetic(99)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=8, max_tokens_per_step=64)
    ctx.init_synthetic(0, 99)
    rm.kv_alloc(512)
    ctx.kv_alloc(0, 512)
    rng = np.random.RandomState(batch)
    prompts = [rng.randint(3, 1024, size=n) for n in (9, 4, 17, 1, 6)[:batch]]
    res = generate_both(m, ctx, [rm], desc, prompts, 5, 512)
    check_steps(res, k=1)
    ctx.close()
    rm.close()


def _defer_case_logits(wq, kvq, batch, steps=3, hkv=16):
    m = load_pplhip()
    desc = ref.make_desc(hidden_dim=2048, intermediate_dim=5632, num_layers=2, num_heads=16, num_kv_heads=hkv, vocab_size=1024,
                         max_position=512, cache_quant_bit=kvq, cache_quant_group=8 if kvq else 1, cache_layout=3,
                         cache_mode=0, weight_quant_bit=wq, weight_quant_group=128)
    rm = ref.RefModel(desc)
    rm.init_synthetic(31)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=256, max_tokens_per_step=1024)
    ctx.init_synthetic(0, 31)
    ntok = batch * 16 + 64
    rm.kv_alloc(ntok)
    ctx.kv_alloc(0, ntok)
    rng = np.random.RandomState(batch)
    prompts = [rng.randint(3, 1024, size=int(n)) for n in rng.randint(1, 9, size=batch)]
    res = generate_both(m, ctx, [rm], desc, prompts, steps, ntok)
    ctx.close()
    rm.close()
    return res


@pytest.mark.parametrize("wq,kvq,batch", [(8, 8, 6), (8, 8, 40), (0, 0, 24), (8, 0, 200), (4, 8, 12), (8, 8, 130), (8, 8, 100), (8, 0, 3)])
def test_split_k_slabs_reduced_by_the_consuming_kernel(wq, kvq, batch):
    """tensor-parallel size 1, 4 < M <= 256: the split-K slabs of wqkv / wo / w2 are summed by RoPE + KV write and by the (Skip)RMSNorms
    that consume them instead of a reduce kernel of their own (kernels.h SplitSlabs) -- against the oracle, and BIT-identical to the
    same steps with PPLHIP_DEFER_REDUCE=0 (child process: the switch is read once)."""
    import subprocess, sys
    res = _defer_case_logits(wq, kvq, batch)
    check_steps(res, k=3)   # (this geometry's noise floor: 1.2e-3)
    logits = np.stack([r[0] for r in res])
    code = ("import sys, numpy as np\n"
            "from tests.test_gpu_model import _defer_case_logits\n"
            f"res = _defer_case_logits({wq}, {kvq}, {batch})\n"
            "np.save(sys.argv[1], np.stack([r[0] for r in res]))\n")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "plain.npy")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, PPLHIP_DEFER_REDUCE="0"), cwd=root, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        plain = np.load(out)
    assert (logits.view(np.uint32) == plain.view(np.uint32)).all(), float(np.abs(logits - plain).max())


@pytest.mark.parametrize("wq,kvq,batch,hkv", [(8, 8, 40, 16), (4, 8, 12, 16), (8, 8, 130, 16), (8, 0, 200, 16), (0, 0, 24, 16), (8, 8, 96, 2), (0, 8, 120, 2)])
def test_two_stream_decode_steps(wq, kvq, batch, hkv):
    """PPLHIP_DUAL_STREAM=1 (child process: the switch is read at context creation): pure-decode steps run as two half-batches on two
    streams (pplhip.cc run_launches), each half with its own split-K workspace, attention workspace and deferred-slab state -- the second
    half's RoPE + KV write and norms read slabs whose rows count from the half's first row.  The child checks its logits against the oracle
    like every model test; here they are compared with the one-stream step's (the halves' GEMMs pick other tile shapes than the whole
    batch's, so the two agree within the summation-order noise, not bit for bit).  (Grouped-query cases use int8 / fp16 weights: the
    synthetic W4 model with 8 : 1 grouped queries has single ill-conditioned rows at 120 requests in which the ORACLE is 3e-2 .. 8e-2 away
    from itself in another summation order -- profiles/r04_late_experiments.md section 4.)"""
    import subprocess, sys
    res = _defer_case_logits(wq, kvq, batch, hkv=hkv)
    check_steps(res, k=3)
    logits = np.stack([r[0] for r in res])
    code = ("import sys, numpy as np\n"
            "from tests.test_gpu_model import _defer_case_logits, check_steps\n"
            f"res = _defer_case_logits({wq}, {kvq}, {batch}, hkv={hkv})\n"
            "check_steps(res, k=3, name='test_two_stream_decode_steps:child')\n"
            "np.save(sys.argv[1], np.stack([r[0] for r in res]))\n")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dual.npy")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, PPLHIP_DUAL_STREAM="1", PPLHIP_DUAL_MIN_ROWS="8", PPLHIP_VERBOSE="1"),
                           cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert "two-stream decode" in r.stderr, r.stderr[-2000:]     # the path under test did run
        dual = np.load(out)
    assert (logits[0].view(np.uint32) == dual[0].view(np.uint32)).all()    # (the prefill step is not split)
    scale = max(1.0, float(np.abs(logits).max()))
    assert float(np.abs(logits - dual).max()) <= 3e-3 * scale


def _long_kv_case_logits(kvq, nreq, steps=5):
    m = load_pplhip()
    desc = ref.make_desc(hidden_dim=512, intermediate_dim=1408, num_layers=2, num_heads=4, num_kv_heads=4, vocab_size=1024,
                         max_position=1024, cache_quant_bit=kvq, cache_quant_group=8 if kvq else 1, cache_layout=3,
                         cache_mode=0, weight_quant_bit=8, weight_quant_group=128)
    rm = ref.RefModel(desc)
    rm.init_synthetic(57)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=16, max_tokens_per_step=4096)
    ctx.init_synthetic(0, 57)
    lens = [515, 701, 530, 644, 519, 770][:nreq]
    ntok = sum(lens) + nreq * (steps + 2) + 64
    rm.kv_alloc(ntok)
    ctx.kv_alloc(0, ntok)
    rng = np.random.RandomState(nreq)
    prompts = [rng.randint(3, 1024, size=n) for n in lens]
    res = generate_both(m, ctx, [rm], desc, prompts, steps, ntok)
    ctx.close()
    rm.close()
    return res


@pytest.mark.parametrize("kvq,nreq", [(8, 1), (8, 3), (0, 6)])
def test_decode_rows_over_long_histories_with_k_splits(kvq, nreq):
    """a few decode rows over 515-770 cached tokens: the runtime splits the keys of every (request, head) over 4-8 blocks + the merge
    kernel (pplhip.cc decode_split), five steps in a row, against the oracle.  (Round 4 measured merging the partial rows in the same
    launch -- last block to arrive at a per-(request, head) counter -- bit-identical and SLOWER: batch 1 / 2 / 4 2.37 / 2.57 / 3.00 ->
    2.62 / 3.17 / 4.20 ms; the device-scope release / acquire every block then needs costs more than the ~4 us launch it saves.)"""
    check_steps(_long_kv_case_logits(kvq, nreq), k=2)
