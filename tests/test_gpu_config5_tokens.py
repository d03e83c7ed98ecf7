"""BASELINE config 5 (prefix cache, 8192-token prompts sharing 6144 tokens) at FULL depth, token for token: the 32-layer LLaMA-2-7B
geometry, W8A16, int8-g8 KV on shuffled 16-token pages.  The same 8192-token prompt is answered twice on the device --

  cold   : one prefill step of all 8192 tokens, then greedy decode steps;
  cached : a second request whose first 6144 tokens sit in the pages the cold request filled (what a prefix-cache hit hands the engine:
           /root/reference/src/generator/llm_generator.cc:233-241,496-551) -- a cache-prefill step of the 2048-token tail at start_pos 6144
           (ENGINE_CONF_CACHE_PREFILL, /root/reference/src/engine/llm_engine.cc:114), then greedy decode steps

-- and both must emit the SAME tokens.  With synthetic weights the top-2 margin of a logits row is of the order of the difference between
two correct fp16 evaluation orders (profiles/r04_prefix_cache_benchmark.log: 13 of 64 synthetic requests answered differently after a
cache hit), so "equal" would be a coin toss.  As in tests/test_gpu_fulldepth.py the lm_head rows of a CHOSEN continuation are therefore
engineered from the final hidden states of a teacher-forced run so that every chosen token wins its row by a wide margin; both paths must
then produce exactly the chosen tokens.

What this test is and is not: cold prefill and the partial-hit cache-prefill are BIT-IDENTICAL on the device --
both read K / V back from the slab through the same kernel on the same 16-token-aligned tiles -- so the logits distance between the two
paths is asserted to be exactly 0: a PROPERTY of the design (a prefix-cache hit cannot change an answer), stronger than any margin
fraction, and not a parity claim.  Parity of these paths against the ORACLE at 32 layers is in tests/test_gpu_fulldepth.py (cache-prefill
behind cached pages; a full hit's single row through the decode kernel and through the cache-prefill kernel), at this size with 2 layers
in tests/test_gpu_config34_shape.py; the oracle cannot run 8192 tokens x 32 layers of 7B in test time."""
import numpy as np
import pytest

from oracle import ref
from tests.conftest import load_pplhip
from tests.parity import record_err

pytestmark = pytest.mark.gpu

DIMS = dict(hidden_dim=4096, intermediate_dim=11008, num_layers=32, num_heads=32, num_kv_heads=32, vocab_size=32000)
P, SHARED, TOTAL, GEN = 16, 6144, 8192, 8
MIN_SAFE_FRACTION = 0.75


def test_cold_prefill_and_prefix_cache_hit_emit_the_same_tokens_at_32_layers():
    m = load_pplhip()
    V, hd = DIMS["vocab_size"], DIMS["hidden_dim"]
    desc = ref.make_desc(max_position=TOTAL + 64, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=1, page_size=P,
                         weight_quant_bit=8, weight_quant_group=128, **DIMS)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=2, max_tokens_per_step=TOTAL)
    ctx.init_synthetic(0, 55)
    n_tok = 3 * TOTAL
    ctx.kv_alloc(0, n_tok)
    try:
        rng = np.random.RandomState(5)
        prompt = rng.randint(3, V, size=TOTAL).astype(np.int64)
        chosen = (2000 + 97 * np.arange(GEN)).astype(np.int64)           # the continuation both paths must produce
        npg = (TOTAL + GEN + P - 1) // P
        perm = rng.permutation(n_tok // P)
        pages_cold = perm[:npg].reshape(1, npg)
        pages_hit = pages_cold.copy()
        pages_hit[0, SHARED // P:] = perm[npg:npg + npg - SHARED // P]     # shared prefix pages, own pages for the tail and the answer

        def answer(path, feed):
            """logits rows [GEN, V] of the answer's steps; feed = tokens to continue with (teacher forcing) or None (greedy)"""
            rows = []
            if path == "cold":
                ctx.set_inputs(0, m.make_step(prompt, [0, TOTAL], [0], pages_cold, 0, npg))
                ctx.run(0)
                pages = pages_cold
            else:
                ctx.set_inputs(0, m.make_step(prompt[SHARED:], [0, TOTAL - SHARED], [SHARED], pages_hit, 0, npg))
                ctx.run(0, cache_prefill=1)
                pages = pages_hit
            rows.append(ctx.copy_logits(1)[0].copy())
            for s in range(1, GEN):
                tok = feed[s - 1] if feed is not None else int(rows[-1].argmax())
                ctx.set_inputs(0, m.make_step(np.array([tok]), [0, 1], [TOTAL + s - 1], pages, 1, npg, req_list_changed=0))
                ctx.run(0)
                rows.append(ctx.copy_logits(1)[0].copy())
            return np.stack(rows)

        # ---- phase A: final hidden states (after the final norm) of the teacher-forced answer, read through an identity lm_head
        eye = np.zeros((V, hd), dtype=np.float16)
        eye[np.arange(hd), np.arange(hd)] = 1.0
        ctx.set_tensor(0, "output.weight", eye)
        y_cold = answer("cold", chosen)[:, :hd].astype(np.float64)       # [GEN, hidden]
        y_hit = answer("hit", chosen)[:, :hd].astype(np.float64)
        # ---- an lm_head of small random rows in which the chosen tokens' rows are solved for: row s of the answer has its chosen token
        # a quarter of the logit scale above every other token
        head = (rng.uniform(-0.0346, 0.0346, size=(V, hd))).astype(np.float16)
        other = y_cold @ head.astype(np.float64).T
        other[:, chosen] = -np.inf
        top_other = other.max(-1)
        scale = max(1.0, float(np.abs(other[np.isfinite(other)]).max()))
        T = np.zeros((GEN, GEN))
        T[np.arange(GEN), np.arange(GEN)] = top_other + 0.25 * scale
        W = (T @ np.linalg.inv(y_cold @ y_cold.T) @ y_cold).astype(np.float16)
        head[chosen] = W
        ctx.set_tensor(0, "output.weight", head)

        # ---- phase B: both paths, greedy
        l_cold = answer("cold", None)
        l_hit = answer("hit", None)
        t_cold, t_hit = l_cold.argmax(-1), l_hit.argmax(-1)
        dist = float(np.abs(l_cold - l_hit).max())
        srt = np.sort(l_cold, -1)
        margin = srt[:, -1] - srt[:, -2]
        safe = margin > 2 * dist
        lscale = max(1.0, float(np.abs(l_cold).max()))
        record_err("config5_cold_vs_prefix_hit_logits_32_layers_bit_identical", dist / lscale, 0.0,
                   noise=float(np.abs(y_cold - y_hit).max() / max(1.0, np.abs(y_cold).max())))
        assert (t_cold == chosen).all(), (t_cold, chosen)
        assert (t_hit == chosen).all(), (t_hit, chosen)                  # the prefix-cache hit answers token for token like the cold request
        assert dist == 0.0, dist / lscale                                # ... because it computes the same bits (see the docstring)
        assert (margin > 0).all() and safe.mean() >= MIN_SAFE_FRACTION
    finally:
        ctx.close()
