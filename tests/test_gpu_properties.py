"""Size-independent properties at the FULL model size of the benchmark configuration (LLaMA-2-7B dimensions, 32 layers,
W8A16, int8-g8 KV): invariants that must hold exactly (bit-for-bit) or to rounding, whatever the oracle says.  (The direct
comparison of this model with the oracle -- logits and the residual stream after every layer, against the oracle's own noise
floor -- is tests/test_gpu_fulldepth.py; the two "to rounding" properties below are two more samples of that noise floor:
1.2-1.6e-2 between two correct fp16 implementations at this depth.)
  * cache-layout / cache-mode invariance: the four KV layouts and contiguous vs paged slots change addressing only;
  * batch-permutation equivariance: permuting the requests of a step permutes the logits rows, bit for bit;
  * prefill / decode consistency: logits after prefilling n+1 tokens == prefilling n tokens then decoding 1;
  * prefix-cache hit == cold prefill (cache-prefill kernel path) at an 8-page shared prefix;
  * determinism: the same step twice gives identical logits;
  * GEMM linearity (x -> 2x doubles every fp16 output exactly) and split-K / tile-path agreement."""
import numpy as np
import pytest

from tests.conftest import load_pplhip
from tests.parity import record_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

DIMS = dict(hidden_dim=4096, intermediate_dim=11008, num_layers=32, num_heads=32, num_kv_heads=32, vocab_size=32000)
I64MAX = np.iinfo(np.int64).max


def make_ctx(m, layout=3, mode=0, page=16, batch=16, tokens=2048, kv_tokens=8192, wq=8, **over):
    kw = dict(DIMS)
    kw.update(over)
    desc = m.make_desc(max_position=4096, cache_quant_bit=8, cache_quant_group=8, cache_layout=layout, cache_mode=mode,
                       page_size=page if mode else 0, weight_quant_bit=wq, weight_quant_group=128, **kw)
    ctx = m.Context(desc, max_running_batch=batch, max_tokens_per_step=tokens)
    ctx.init_synthetic(0, 4321)
    ctx.kv_alloc(0, kv_tokens)
    return ctx


def slots(mode, lens_total, page, kv_tokens, seed=0):
    n = len(lens_total)
    if mode == 0:
        return np.concatenate([[0], np.cumsum(lens_total)[:-1]]).astype(np.int64), 0
    npg = (np.asarray(lens_total) + page - 1) // page
    mp = int(npg.max())
    idx = np.full((n, mp), I64MAX, dtype=np.int64)
    order = np.random.RandomState(seed).permutation(kv_tokens // page)
    k = 0
    for i in range(n):
        idx[i, :npg[i]] = order[k:k + npg[i]]
        k += npg[i]
    return idx, mp


def run_two_steps(m, ctx, prompts, mode, page, kv_tokens, order=None):
    """packed prefill of all prompts, then one decode step fed with fixed tokens; returns both logits arrays"""
    n = len(prompts)
    if order is None:
        order = np.arange(n)
    ps = [prompts[i] for i in order]
    lens = np.array([len(p) for p in ps])
    ci, mp = slots(mode, lens + 2, page, kv_tokens)
    seq = np.concatenate([[0], np.cumsum(lens)])
    ctx.set_inputs(0, m.make_step(np.concatenate(ps), seq, np.zeros(n, dtype=np.int64), ci, 0, max_pages=mp))
    ctx.run(0)
    l0 = ctx.copy_logits(n)
    nxt = np.array([(7 * int(p[-1]) + 11) % 32000 for p in ps], dtype=np.int64)
    ctx.set_inputs(0, m.make_step(nxt, np.arange(n + 1), lens, ci, n, max_pages=mp, req_list_changed=0))
    ctx.run(0)
    l1 = ctx.copy_logits(n)
    return l0, l1


@pytest.fixture(scope="module")
def prompts():
    rng = np.random.RandomState(5)
    return [rng.randint(3, 32000, size=k).astype(np.int64) for k in (37, 1, 130, 64, 5, 17)]


def test_layout_and_mode_invariance_bit_exact(prompts):
    m = load_pplhip()
    ref_l = None
    for layout, mode in [(3, 0), (0, 0), (1, 1), (2, 1), (3, 1)]:
        ctx = make_ctx(m, layout=layout, mode=mode)
        l0, l1 = run_two_steps(m, ctx, prompts, mode, 16, 8192)
        ctx.close()
        assert np.isfinite(l0).all() and np.isfinite(l1).all()
        if ref_l is None:
            ref_l = (l0, l1)
        else:
            assert (l0 == ref_l[0]).all() and (l1 == ref_l[1]).all(), (layout, mode)


def test_permutation_determinism_and_prefill_decode_consistency(prompts):
    m = load_pplhip()
    ctx = make_ctx(m)
    a0, a1 = run_two_steps(m, ctx, prompts, 0, 16, 8192)
    b0, b1 = run_two_steps(m, ctx, prompts, 0, 16, 8192)
    assert (a0 == b0).all() and (a1 == b1).all()                       # determinism
    order = np.array([3, 0, 5, 1, 4, 2])
    c0, c1 = run_two_steps(m, ctx, prompts, 0, 16, 8192, order=order)
    assert (c0 == a0[order]).all() and (c1 == a1[order]).all()         # permutation equivariance, bit for bit
    # prefill n+1 == prefill n, then decode 1 (different kernels: MFMA prefill attention vs the decode kernel, tile
    # GEMM vs skinny GEMM) -> equal to fp16 rounding, same argmax unless the top-2 margin is inside the noise
    p = prompts[2]
    nxt = (7 * int(p[-1]) + 11) % 32000
    ext = np.concatenate([p, [nxt]])
    ctx.set_inputs(0, m.make_step(ext, [0, len(ext)], [0], [4096], 0))
    ctx.run(0)
    full = ctx.copy_logits(1)[0]
    step = a1[2]
    scale = max(1.0, np.abs(full).max())
    # two different kernel paths over 32 layers with int8 KV (MFMA prefill attention + tile GEMM vs the VALU decode kernel +
    # skinny GEMM): observed 7.0e-3 -- half the oracle's own summation-order noise at this depth (tests/test_gpu_fulldepth.py)
    record_err("7b_prefill_vs_decode_consistency", np.abs(full - step).max() / scale, 1e-2)
    assert np.abs(full - step).max() <= 1e-2 * scale
    srt = np.sort(full)
    if srt[-1] - srt[-2] > 2e-2 * scale:
        assert full.argmax() == step.argmax()
    ctx.close()


def test_prefix_hit_equals_cold_prefill_full_size():
    m = load_pplhip()
    ctx = make_ctx(m, mode=1, page=16, tokens=1024)
    rng = np.random.RandomState(9)
    prompt = rng.randint(3, 32000, size=150).astype(np.int64)
    pages = np.arange(100, 110, dtype=np.int64)[None, :]                # 10 pages for 150 (+ margin) tokens
    ctx.set_inputs(0, m.make_step(prompt, [0, 150], [0], pages, 0, max_pages=10))
    ctx.run(0)
    cold = ctx.copy_logits(1)[0]
    # 8 full pages (128 tokens) are "cached": feed only tokens 128..149 with start_pos 128, new pages for the tail
    hit_pages = np.concatenate([pages[0, :8], [200, 201]])[None, :]
    ctx.set_inputs(0, m.make_step(prompt[128:], [0, 22], [128], hit_pages, 0, max_pages=10))
    ctx.run(0, cache_prefill=1)
    hit = ctx.copy_logits(1)[0]
    scale = max(1.0, np.abs(cold).max())
    record_err("7b_prefix_hit_vs_cold", np.abs(cold - hit).max() / scale, 1e-2)   # observed (r02) 5.8e-3
    assert np.abs(cold - hit).max() <= 1e-2 * scale
    ctx.close()


def test_gemm_linearity_and_path_agreement():
    """x -> 2x doubles fp16 outputs exactly (power-of-two scaling commutes with every rounding); the skinny, split-K and
    tiled kernels agree on shared rows to accumulation-order noise."""
    m = load_pplhip()
    rng = np.random.RandomState(1)
    N, K = 12288, 4096
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
    sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half()
    outs = {}
    for M in (8, 100, 1024):
        x = (torch.randn(M, K, device="cuda") * 0.25).half()
        x[:8] = torch.from_numpy((rng.randn(8, K) * 0.25).astype(np.float16)).cuda() if M == 8 else outs["x8"]
        if M == 8:
            outs["x8"] = x[:8].clone()
        y1 = torch.empty(M, N, dtype=torch.float16, device="cuda")
        y2 = torch.empty(M, N, dtype=torch.float16, device="cuda")
        x2 = (x.float() * 2).half()
        assert m.lib().pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y1.data_ptr(), 0) == 0
        assert m.lib().pplhip_op_linear(None, x2.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y2.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        # linearity: bit exact wherever the fp16 result is a normal number (subnormal outputs round differently when doubled)
        normal = y1.float().abs() >= 2.0 ** -13
        assert torch.equal(y2.float()[normal], (y1.float() * 2)[normal])
        assert (y2.float() - y1.float() * 2).abs().max().item() <= 2.0 ** -23
        outs[M] = y1[:8].float().cpu().numpy()
    ref8 = outs[8]
    for M in (100, 1024):
        assert np.abs(outs[M] - ref8).max() <= 2e-3 * max(1.0, np.abs(ref8).max())


def test_grouped_query_w4_model_invariances(prompts):
    """the same invariants on a grouped-query, W4A16-g128 model at 70B-like head geometry (8 query heads per KV head,
    head_dim 128, hidden 4096, 8 layers): decode rows run the MFMA grouped-query kernel, the GEMMs the int4 tile path."""
    m = load_pplhip()
    over = dict(num_kv_heads=4, num_layers=8, wq=4)
    ref_l = None
    for layout, mode in [(3, 0), (1, 1), (3, 1)]:
        ctx = make_ctx(m, layout=layout, mode=mode, **over)
        l0, l1 = run_two_steps(m, ctx, prompts, mode, 16, 8192)
        if ref_l is None:
            order = np.array([3, 0, 5, 1, 4, 2])
            c0, c1 = run_two_steps(m, ctx, prompts, mode, 16, 8192, order=order)
            assert (c0 == l0[order]).all() and (c1 == l1[order]).all()     # permutation equivariance, bit for bit
            # prefill n+1 == prefill n, then decode 1 (MFMA prefill kernel vs MFMA grouped-query decode kernel)
            p = prompts[2]
            ext = np.concatenate([p, [(7 * int(p[-1]) + 11) % 32000]])
            ctx.set_inputs(0, m.make_step(ext, [0, len(ext)], [0], [4096], 0))
            ctx.run(0)
            full = ctx.copy_logits(1)[0]
            scale = max(1.0, np.abs(full).max())
            record_err("gqa_w4_prefill_vs_decode_consistency", np.abs(full - l1[2]).max() / scale, 1e-3)   # observed (r02) 1e-4
            assert np.abs(full - l1[2]).max() <= 1e-3 * scale
        ctx.close()
        assert np.isfinite(l0).all() and np.isfinite(l1).all()
        if ref_l is None:
            ref_l = (l0, l1)
        else:
            assert (l0 == ref_l[0]).all() and (l1 == ref_l[1]).all(), (layout, mode)


@pytest.mark.parametrize("N,K,swiglu", [(12288, 4096, 0), (4096, 11008, 0), (22016, 4096, 1)])
@pytest.mark.parametrize("M", [254, 40])
def test_gemm_output_does_not_depend_on_the_row_position(M, N, K, swiglu):
    """every row of x is the same vector, so every output row must be the same bits -- whichever MFMA sub-tile, wave and
    epilogue slot the row lands in.  (Regression: the compiler fused `acc * scale -> fp16` into v_fma_mixlo_f16 -- one rounding
    -- for some epilogue slots and into mul + cvt -- two roundings -- for others; rows 112..127 of a tile then differed from the
    rest on fp16 rounding ties, about one column in 10^4.)"""
    import ctypes as C
    m = load_pplhip()
    torch.manual_seed(N + M)
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
    sc = (torch.rand(N, device="cuda") * 0.001).half()
    for trial in range(3):
        x = (torch.randn(1, K, device="cuda") * 0.5).half().repeat(M, 1).contiguous()
        y = torch.empty(M, N // 2 if swiglu else N, device="cuda", dtype=torch.float16)
        if swiglu:
            rc = m.lib().pplhip_op_linear_swiglu(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 0, M, N, K, y.data_ptr())
        else:
            rc = m.lib().pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 0, M, N, K, y.data_ptr(), 0)
        assert rc == 0
        torch.cuda.synchronize()
        assert bool((y == y[0:1]).all()), (trial, torch.nonzero((y != y[0:1]).any(1)).flatten().tolist()[:20])
