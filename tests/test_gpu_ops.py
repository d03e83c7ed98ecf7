"""Parity of every hand-written gfx950 kernel against the CPU oracle, one operator at a time, through the C ABI
(pplhip_op_*).  Tolerances: bit-exact for integer work (embedding gather, int8 KV bytes up to documented +-1 LSB
rounding ties, token ids); fp16 outputs within 1 fp16 ulp-class bounds stated per test."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref
from tests.conftest import load_pplhip

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


_KEEP = []


def dev(arr):
    """host array -> device tensor, kept alive until the next test (a temporary freed before the launch would be
    handed out again by torch's caching allocator and two kernel arguments would alias)."""
    t = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _drop_device_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def f16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16)


def ck(rc):
    assert rc == 0, rc
    torch.cuda.synchronize()


def close_f16(got, want, rel=2e-3, abs_=2e-3):
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    err = np.abs(got - want)
    tol = abs_ + rel * np.abs(want)
    assert (err <= tol).all(), (err.max(), np.abs(want).max(), int((err > tol).sum()))


def test_embedding_bit_exact():
    m = load_pplhip()
    rng = np.random.RandomState(0)
    table = f16(rng.randn(300, 256))
    ids = rng.randint(0, 300, size=77).astype(np.int64)
    out = torch.empty((77, 256), dtype=torch.float16, device="cuda")
    ck(m.lib().pplhip_op_embedding(None, dev(ids).data_ptr(), dev(table).data_ptr(), 77, 256, out.data_ptr()))
    assert (out.cpu().numpy() == table[ids]).all()


@pytest.mark.parametrize("hidden", [128, 4096, 8192, 5120])
@pytest.mark.parametrize("skip", [False, True])
def test_rmsnorm(hidden, skip):
    m = load_pplhip()
    rng = np.random.RandomState(hidden)
    T = 37
    x = f16(rng.randn(T, hidden) * 2)
    sk = f16(rng.randn(T, hidden)) if skip else None
    w = f16(1 + 0.1 * rng.randn(hidden))
    want = np.empty((T, hidden), dtype=np.float32)
    want_res = np.empty((T, hidden), dtype=np.float32)
    xs = x.astype(np.float32)
    sks = sk.astype(np.float32) if skip else None
    ref.lib().ref_rmsnorm(xs.ctypes.data, None if sks is None else sks.ctypes.data, w.ctypes.data, 1e-5, T, hidden,
                          want.ctypes.data, want_res.ctypes.data)
    out = torch.empty((T, hidden), dtype=torch.float16, device="cuda")
    res = torch.empty((T, hidden), dtype=torch.float16, device="cuda")
    dx, dw = dev(x), dev(w)
    dsk = dev(sk) if skip else None
    ck(m.lib().pplhip_op_rmsnorm(None, dx.data_ptr(), dsk.data_ptr() if skip else None, dw.data_ptr(), 1e-5, T, hidden,
                                 out.data_ptr(), res.data_ptr()))
    # residual = fp16(x + skip): bit exact
    assert (res.cpu().numpy().astype(np.float32) == want_res).all()
    close_f16(out.cpu().numpy(), want, rel=1.5e-3, abs_=1e-4)


@pytest.mark.parametrize("wq", [0, 8, 4])
@pytest.mark.parametrize("M,N,K", [(1, 128, 128), (5, 384, 256), (130, 320, 128), (64, 512, 1376), (257, 1024, 4096),
                                   (33, 1024, 704), (2100, 1284, 512), (100, 384, 4096), (200, 260, 2048), (1024, 1536, 1024),
                                   (4100, 1092, 256),    # M >= 4096: the 256 x 256 tile kernel (prefill steps)
                                   (1040, 1284, 128), (2148, 516, 64),   # a few rows above a multiple of 1024: two launches (main part + rest)
                                   (1000, 12000, 192), (1024, 12288, 64),  # W8: the 128 x 384 producer / consumer kernel (ragged edges; one K tile < ring depth)
                                   (4608, 5120, 128), (8192, 2304, 64),   # its super-tile block order: 18 x 20 tiles (3 x 3 super-tiles), 32 x 9 (4 x 2, padded share)
                                   # per-rank shapes of BASELINE configs 3 / 4: 13B/TP2 wqkv and w2, 70B/TP8 wqkv, wo-like and w2
                                   (300, 7680, 5120), (300, 5120, 6912), (256, 1280, 8192), (64, 8192, 1024), (130, 8192, 3584),
                                   (150, 10100, 1792)])   # W4: the 128 x 256 tile kernel with a ragged last tile and two K slabs
def test_linear(wq, M, N, K):
    m = load_pplhip()
    if wq == 4 and K % 128:
        group = 32
    else:
        group = 128
    rng = np.random.RandomState(M * 7 + N + K + wq)
    x = f16(rng.randn(M, K))
    if wq == 0:
        w = f16(rng.randn(N, K) * 0.05)
        scale = None
    elif wq == 8:
        w = rng.randint(-127, 128, size=(N, K)).astype(np.int8)
        scale = f16(0.0005 * (0.5 + rng.rand(N)))
    else:
        w = rng.randint(0, 256, size=(N, K // 2)).astype(np.uint8)
        scale = f16(0.01 * (0.5 + rng.rand(N, K // group)))
    for out_fp32 in (0, 1):
        want = np.empty((M, N), dtype=np.float32)
        xs = x.astype(np.float32)
        ref.lib().ref_linear_raw(xs.ctypes.data, w.ctypes.data, None if scale is None else scale.ctypes.data, wq, group, M, N,
                                 K, want.ctypes.data, out_fp32)
        y = torch.empty((M, N), dtype=torch.float32 if out_fp32 else torch.float16, device="cuda")
        dx, dw = dev(x), dev(w)
        ds = dev(scale) if scale is not None else None
        ck(m.lib().pplhip_op_linear(None, dx.data_ptr(), dw.data_ptr(), ds.data_ptr() if ds is not None else None, wq, group,
                                    M, N, K, y.data_ptr(), out_fp32))
        got = y.float().cpu().numpy()
        mag = np.abs(want).max()
        # fp32 accumulation-order noise + (W4 only) one fp16 rounding of q*scale per weight
        rel = 3e-3 if wq == 4 else 1.5e-3
        # (W4: the per-weight rounding noise scales with the typical |y|, not with the element's own value)
        close_f16(got, want, rel=rel, abs_=rel * mag * (0.25 if wq == 4 else 0.05) + 1e-5)


@pytest.mark.parametrize("seed", range(6))
def test_linear_random_shapes_w8(seed):
    """pplhip_op_linear W8A16 over random shapes around the dispatcher's thresholds (skinny / 128 x 128 with and without split-K and
    producers / 128 x 384 producer-consumer / 256 x 256 with its super-tile order), against the oracle on the whole output."""
    m = load_pplhip()
    rng = np.random.RandomState(1000 + seed)
    cases = []
    for _ in range(5):
        M = int(rng.choice([rng.randint(1, 40), rng.randint(100, 600), rng.randint(512, 1300), rng.randint(1300, 4200), rng.randint(4096, 5000)]))
        N = int(rng.choice([rng.randint(16, 300), rng.randint(1000, 4200), rng.randint(8192, 13000)])) // 4 * 4
        K = int(rng.choice([64, 128, 192, 320, 704])) if M * N > 2e7 else int(rng.randint(1, 24)) * 64
        cases.append((M, N, K))
    for M, N, K in cases:
        x = f16(rng.randn(M, K))
        w = rng.randint(-127, 128, size=(N, K)).astype(np.int8)
        scale = f16(0.0005 * (0.5 + rng.rand(N)))
        want = np.empty((M, N), dtype=np.float32)
        xs = x.astype(np.float32)
        ref.lib().ref_linear_raw(xs.ctypes.data, w.ctypes.data, scale.ctypes.data, 8, 128, M, N, K, want.ctypes.data, 0)
        y = torch.empty((M, N), dtype=torch.float16, device="cuda")
        dx, dw, ds = dev(x), dev(w), dev(scale)
        ck(m.lib().pplhip_op_linear(None, dx.data_ptr(), dw.data_ptr(), ds.data_ptr(), 8, 128, M, N, K, y.data_ptr(), 0))
        got = y.float().cpu().numpy()
        mag = np.abs(want).max()
        try:
            close_f16(got, want, rel=1.5e-3, abs_=1.5e-3 * mag * 0.05 + 1e-5)
        except AssertionError as e:
            raise AssertionError(f"shape M={M} N={N} K={K}: {e}")


def test_silu_mul():
    m = load_pplhip()
    rng = np.random.RandomState(3)
    T, inter = 19, 1376
    gu = f16(rng.randn(T, 2 * inter) * 3)
    want = np.empty((T, inter), dtype=np.float32)
    gus = gu.astype(np.float32)
    ref.lib().ref_silu_mul(gus.ctypes.data, T, inter, want.ctypes.data)
    out = torch.empty((T, inter), dtype=torch.float16, device="cuda")
    ck(m.lib().pplhip_op_silu_mul(None, dev(gu).data_ptr(), T, inter, out.data_ptr()))
    close_f16(out.cpu().numpy(), want, rel=1.5e-3, abs_=1e-5)


# ---------------------------------------------------------------------------------------------------------------
# KV-cache operators
# ---------------------------------------------------------------------------------------------------------------
class KvCase:
    """A ragged batch + KV slab shared by the oracle and the device."""

    def __init__(self, m, H, Hkv, D, L, layer, quant, layout, mode, seqlens, start_pos, seed=0, page_size=4,
                 decoding_batches=0):
        self.m, self.H, self.Hkv, self.D, self.L, self.layer = m, H, Hkv, D, L, layer
        rng = np.random.RandomState(seed)
        self.B = len(seqlens)
        seqlens = np.asarray(seqlens, dtype=np.int64)
        start_pos = np.asarray(start_pos, dtype=np.int64)
        self.seq_starts = np.concatenate([[0], np.cumsum(seqlens)]).astype(np.int64)
        self.start_pos = start_pos
        self.T = int(seqlens.sum())
        total = start_pos + seqlens
        self.max_seq_len, self.max_kv_len = int(seqlens.max()), int(total.max())
        self.decoding_batches = decoding_batches
        self.desc = ref.make_desc(hidden_dim=H * D, intermediate_dim=64, num_layers=L, num_heads=H, num_kv_heads=Hkv,
                                  vocab_size=64, cache_quant_bit=quant, cache_quant_group=8 if quant else 1,
                                  cache_layout=layout, cache_mode=mode, page_size=page_size if mode else 0)
        if mode == 0:
            gaps = rng.randint(0, 5, size=self.B)
            self.cache_idx = (np.concatenate([[0], np.cumsum(total + gaps)[:-1]]) + 3).astype(np.int64)
            self.max_pages = 0
            self.N = int((total + gaps).sum()) + 8
        else:
            P = page_size
            npg = (total + P - 1) // P
            self.max_pages = int(npg.max())
            n_pages = int(npg.sum()) + 5
            order = rng.permutation(n_pages)
            self.cache_idx = np.full((self.B, self.max_pages), np.iinfo(np.int64).max, dtype=np.int64)
            k = 0
            for i in range(self.B):
                self.cache_idx[i, :npg[i]] = order[k:k + npg[i]]
                k += npg[i]
            self.N = n_pages * P
        elems = self.N * L * 2 * Hkv * D
        self.cache = np.zeros(elems, dtype=np.int8 if quant else np.float16)
        self.scale = np.zeros(elems // 8, dtype=np.float16) if quant else None
        self.qkv = f16(rng.randn(self.T, (H + 2 * Hkv) * D))
        npos = max(4096, self.max_kv_len + 1)   # the table must cover every position the case uses
        self.rope = np.empty((npos, D), dtype=np.float32)
        ref.lib().ref_build_rope_table(self.rope.ctypes.data, npos, D, 10000.0)

    def view(self, dcache, dscale):
        v = self.m.KvView()
        v.cache, v.scale = dcache.data_ptr(), (dscale.data_ptr() if dscale is not None else None)
        v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = self.N, self.L, self.Hkv, self.D
        d = self.desc
        v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = (d.cache_quant_bit, d.cache_quant_group,
                                                                              d.cache_layout, d.cache_mode, d.page_size,
                                                                              self.layer)
        return v

    def ref_write(self):
        q32 = self.qkv.astype(np.float32)
        ref.lib().ref_rope_kv_write(q32.ctypes.data, self.rope.ctypes.data, C.byref(self.desc), self.H, self.Hkv, self.D,
                                    self.layer, self.cache.ctypes.data, None if self.scale is None else self.scale.ctypes.data,
                                    self.N, self.seq_starts.ctypes.data, self.start_pos.ctypes.data,
                                    self.cache_idx.ctypes.data, self.max_pages, self.B)
        return q32

    def ref_attention(self, q32):
        out = np.zeros((self.T, self.H * self.D), dtype=np.float32)
        ref.lib().ref_attention(q32.ctypes.data, C.byref(self.desc), self.H, self.Hkv, self.D, self.layer,
                                self.cache.ctypes.data, None if self.scale is None else self.scale.ctypes.data, self.N,
                                self.seq_starts.ctypes.data, self.start_pos.ctypes.data, self.cache_idx.ctypes.data,
                                self.max_pages, self.B, out.ctypes.data)
        return out


@pytest.mark.parametrize("quant", [0, 8])
@pytest.mark.parametrize("layout,mode", [(0, 0), (1, 1), (2, 0), (3, 1), (3, 0)])
@pytest.mark.parametrize("H,Hkv,D", [(4, 4, 32), (8, 2, 64), (4, 4, 128), (32, 8, 128), (32, 32, 128)])   # the last two: 2 / 4 blocks per token
def test_rope_kv_write(quant, layout, mode, H, Hkv, D):
    m = load_pplhip()
    case = KvCase(m, H, Hkv, D, L=3, layer=1, quant=quant, layout=layout, mode=mode, seqlens=[5, 1, 9, 1],
                  start_pos=[0, 7, 3, 0], seed=layout * 10 + mode)
    want_q = case.ref_write()
    dq = dev(case.qkv)
    dcache = torch.zeros(case.cache.size, dtype=torch.int8 if quant else torch.float16, device="cuda")
    dscale = torch.zeros(case.scale.size, dtype=torch.float16, device="cuda") if quant else None
    v = case.view(dcache, dscale)
    ck(m.lib().pplhip_op_rope_kv_write(None, dq.data_ptr(), dev(case.rope).data_ptr(), C.byref(v), dev(case.seq_starts).data_ptr(),
                                       dev(case.start_pos).data_ptr(), dev(case.cache_idx).data_ptr(), case.max_pages, case.B,
                                       case.T, H))
    got_q = dq.cpu().numpy().astype(np.float32)
    hq = H * D
    # rotated q (fp16, written in place): bit exact (identical fp32 op sequence, same cos/sin table)
    assert (got_q[:, :hq] == want_q[:, :hq]).all()
    got_cache = dcache.cpu().numpy()
    if quant == 0:
        assert (got_cache.view(np.uint16) == case.cache.view(np.uint16)).all()
    else:
        assert (dscale.cpu().numpy().view(np.uint16) == case.scale.view(np.uint16)).all()
        assert (got_cache == case.cache).all()


ATT_SHAPES = [(4, 4, 32), (8, 2, 64), (4, 4, 128), (8, 1, 128), (16, 1, 64), (12, 2, 128)]


@pytest.mark.parametrize("quant", [0, 8])
@pytest.mark.parametrize("layout,mode", [(3, 0), (0, 0), (2, 1), (3, 1)])
@pytest.mark.parametrize("H,Hkv,D", ATT_SHAPES)
def test_attention_decode(quant, layout, mode, H, Hkv, D):
    """decode rows: kv lengths 1 .. 700, ragged, contiguous and paged, MHA and GQA, with and without split-K."""
    m = load_pplhip()
    kvlen = [1, 2, 63, 64, 65, 257, 700, 33]
    case = KvCase(m, H, Hkv, D, L=2, layer=1, quant=quant, layout=layout, mode=mode, seqlens=[1] * len(kvlen),
                  start_pos=[k - 1 for k in kvlen], seed=D + quant, page_size=16, decoding_batches=len(kvlen))
    rng = np.random.RandomState(5)
    # random cache contents (history), then write the current token through the oracle
    if quant:
        case.cache[:] = rng.randint(-127, 128, size=case.cache.size).astype(np.int8)
        case.scale[:] = f16(0.02 * (0.5 + rng.rand(case.scale.size)))
    else:
        case.cache[:] = f16(rng.randn(case.cache.size))
    q32 = case.ref_write()
    want = case.ref_attention(q32)
    dq = dev(q32.astype(np.float16))
    dcache, dscale = dev(case.cache), (dev(case.scale) if quant else None)
    v = case.view(dcache, dscale)
    args = (dev(case.seq_starts), dev(case.start_pos), dev(case.cache_idx))
    for split in (1, 3):
        out = torch.zeros((case.T, H * D), dtype=torch.float16, device="cuda")
        ws = torch.empty(case.B * H * split * (D + 2) + 16, dtype=torch.float32, device="cuda")
        ck(m.lib().pplhip_op_attention(None, dq.data_ptr(), C.byref(v), args[0].data_ptr(), args[1].data_ptr(),
                                       args[2].data_ptr(), case.max_pages, case.B, case.T, case.B, 1, case.max_kv_len, H,
                                       split, ws.data_ptr(), ws.numel() * 4, out.data_ptr()))
        close_f16(out.cpu().numpy(), want, rel=1.5e-3, abs_=1.5e-3)


@pytest.mark.parametrize("quant", [0, 8])
@pytest.mark.parametrize("layout,mode", [(3, 0), (1, 0), (3, 1)])
@pytest.mark.parametrize("H,Hkv,D", ATT_SHAPES)
def test_attention_prefill_and_mixed(quant, layout, mode, H, Hkv, D):
    """mixed step: 2 decode rows first, then prefill / cache-prefill (start_pos > 0) requests of ragged length."""
    m = load_pplhip()
    seqlens = [1, 1, 130, 1, 64, 17, 200]
    start = [40, 5, 0, 0, 64, 30, 70]
    case = KvCase(m, H, Hkv, D, L=2, layer=0, quant=quant, layout=layout, mode=mode, seqlens=seqlens, start_pos=start,
                  seed=D * 3 + quant, page_size=16, decoding_batches=2)
    rng = np.random.RandomState(9)
    if quant:
        case.cache[:] = rng.randint(-127, 128, size=case.cache.size).astype(np.int8)
        case.scale[:] = f16(0.02 * (0.5 + rng.rand(case.scale.size)))
    else:
        case.cache[:] = f16(rng.randn(case.cache.size))
    q32 = case.ref_write()
    want = case.ref_attention(q32)
    dq = dev(q32.astype(np.float16))
    dcache, dscale = dev(case.cache), (dev(case.scale) if quant else None)
    v = case.view(dcache, dscale)
    out = torch.zeros((case.T, H * D), dtype=torch.float16, device="cuda")
    ck(m.lib().pplhip_op_attention(None, dq.data_ptr(), C.byref(v), dev(case.seq_starts).data_ptr(),
                                   dev(case.start_pos).data_ptr(), dev(case.cache_idx).data_ptr(), case.max_pages, case.B,
                                   case.T, 2, case.max_seq_len, case.max_kv_len, H, 1, None, 0, out.data_ptr()))
    # prefill path: K/V dequantised to fp16 and P rounded to fp16 before the MFMAs: 4e-3 relative to |V|max
    vmax = 3.0 if not quant else 0.03 * 127
    from tests.parity import record_err
    record_err(f"attn_mixed_q{quant}_l{layout}m{mode}_{H}_{Hkv}_{D}", float(np.abs(out.cpu().numpy().astype(np.float32) - want).max()) / vmax, 1e-3)
    close_f16(out.cpu().numpy(), want, rel=1e-3, abs_=1e-3 * vmax)   # observed (r02) <= 6.5e-4 |V|max


LONG_CASES = [  # (new tokens, cached tokens) per request, (H, Hkv)
    ([1024], [0], (2, 2)), ([2048, 5], [0, 0], (2, 2)), ([4096], [0], (8, 1)), ([1500, 700], [300, 4000], (2, 2)),
    # BASELINE config 5: 2048 new tokens behind a 6144-token cached prefix (benchmark_prefix_cache_offline shape)
    ([2048], [6144], (2, 2)), ([2048], [6144], (8, 1)),
    # the 32-row kernel's XCD-aware 1-D grid (k_attn_prefill32.hip): several ragged requests x several query blocks x H = 16 and 24 heads
    # (2 and 3 heads per XCD), 8-wave blocks (>= 1024 new tokens) and 4-wave blocks, contiguous slots (mode 0) as well as pages
    ([1100, 37, 1300, 256], [0, 500, 64, 0], (16, 4), 0), ([1100, 37, 1300, 256], [0, 500, 64, 0], (16, 4), 1),
    ([300, 900, 1], [0, 130, 700], (24, 8), 0),
]


@pytest.mark.parametrize("quant", [8, 0])
@pytest.mark.parametrize("seqlens,start,heads,mode", [c if len(c) == 4 else c + (1,) for c in LONG_CASES])
def test_attention_long_prefill_and_cache_prefill(quant, seqlens, start, heads, mode):
    """prefill / cache-prefill attention against the oracle at 1k .. 8k keys (dozens of 128-key tiles per query tile):
    causal wave skipping, the mask-only-on-diagonal-tiles rule, the conditional rescale across many tiles and the
    8192-key cache-prefill of the prefix-cache benchmark -- paged (16-token pages, shuffled), int8 and fp16 KV."""
    m = load_pplhip()
    H, Hkv = heads
    D = 128
    case = KvCase(m, H, Hkv, D, L=1, layer=0, quant=quant, layout=3, mode=mode, seqlens=seqlens, start_pos=start,
                  seed=len(seqlens) + quant + H, page_size=16, decoding_batches=0)
    rng = np.random.RandomState(17)
    if quant:
        case.cache[:] = rng.randint(-127, 128, size=case.cache.size).astype(np.int8)
        case.scale[:] = f16(0.02 * (0.5 + rng.rand(case.scale.size)))
    else:
        case.cache[:] = f16(rng.randn(case.cache.size))
    q32 = case.ref_write()
    want = case.ref_attention(q32)
    dq = dev(q32.astype(np.float16))
    dcache, dscale = dev(case.cache), (dev(case.scale) if quant else None)
    v = case.view(dcache, dscale)
    out = torch.zeros((case.T, H * D), dtype=torch.float16, device="cuda")
    ck(m.lib().pplhip_op_attention(None, dq.data_ptr(), C.byref(v), dev(case.seq_starts).data_ptr(),
                                   dev(case.start_pos).data_ptr(), dev(case.cache_idx).data_ptr(), case.max_pages, case.B,
                                   case.T, 0, case.max_seq_len, case.max_kv_len, H, 1, None, 0, out.data_ptr()))
    got = out.cpu().numpy().astype(np.float32)
    vmax = 3.0 if not quant else 0.03 * 127
    from tests.parity import record_err
    record_err(f"attn_long_q{quant}_{seqlens}_{start}_{heads}", float(np.abs(got - want).max()) / vmax, 1e-3)
    close_f16(got, want, rel=1e-3, abs_=1e-3 * vmax)   # observed (r02) <= 5.1e-4 |V|max


def _split_kv_case(quant, seqlens, start, heads, mode, nb=0):
    """one cache-prefill launch with and without the split-KV workspace, against the oracle.  The workspace is sized for the prefill
    rows ONLY (the launcher's contract) and followed by a canary region: decode rows ahead of the prefill requests must not shift the
    partial rows past its end."""
    m = load_pplhip()
    H, Hkv = heads
    D = 128
    case = KvCase(m, H, Hkv, D, L=1, layer=0, quant=quant, layout=3, mode=mode, seqlens=seqlens, start_pos=start,
                  seed=len(seqlens) + quant + H, page_size=16, decoding_batches=nb)
    rng = np.random.RandomState(23)
    if quant:
        case.cache[:] = rng.randint(-127, 128, size=case.cache.size).astype(np.int8)
        case.scale[:] = f16(0.02 * (0.5 + rng.rand(case.scale.size)))
    else:
        case.cache[:] = f16(rng.randn(case.cache.size))
    q32 = case.ref_write()
    want = case.ref_attention(q32)
    dq = dev(q32.astype(np.float16))
    dcache, dscale = dev(case.cache), (dev(case.scale) if quant else None)
    v = case.view(dcache, dscale)
    n_ws = (case.T - nb) * H * 32 * (D + 2)
    n_guard = (nb + 8) * H * 32 * (D + 2)
    ws = torch.zeros(n_ws + n_guard, dtype=torch.float32, device="cuda")
    ws[n_ws:] = 12345.0
    outs = []  # (first with the workspace: split-KV; then without: one block per (query block, request, head))
    for wsp, wsb in ((ws.data_ptr(), n_ws * 4), (None, 0)):
        out = torch.zeros((case.T, H * D), dtype=torch.float16, device="cuda")
        ck(m.lib().pplhip_op_attention(None, dq.data_ptr(), C.byref(v), dev(case.seq_starts).data_ptr(), dev(case.start_pos).data_ptr(),
                                       dev(case.cache_idx).data_ptr(), case.max_pages, case.B, case.T, nb, case.max_seq_len,
                                       case.max_kv_len, H, 1, wsp, wsb, out.data_ptr()))
        outs.append(out.cpu().numpy().astype(np.float32))
    assert float(ws[:n_ws].abs().max()) > 0, "the split-KV path did not run"
    assert bool((ws[n_ws:] == 12345.0).all()), "the split-KV kernel wrote past its workspace"
    assert np.isfinite(outs[0]).all()
    vmax = 3.0 if not quant else 0.03 * 127
    close_f16(outs[0], want, rel=1e-3, abs_=1e-3 * vmax)
    close_f16(outs[1], want, rel=1e-3, abs_=1e-3 * vmax)
    close_f16(outs[0], outs[1], rel=2e-3, abs_=2e-4 * vmax)


@pytest.mark.parametrize("quant", [8, 0])
@pytest.mark.parametrize("seqlens,start,heads,mode", [([16], [8176], (8, 1), 1), ([7, 16, 1], [2000, 5000, 4097], (4, 4), 0),
                                                        ([32, 3], [1000, 1500], (2, 2), 1),
                                                        ([300, 129], [2500, 900], (4, 4), 1), ([600], [1800], (2, 2), 0)])  # several query blocks
def test_attention_short_suffix_split_kv(quant, seqlens, start, heads, mode):
    """cache-prefill of a few new tokens behind a long cached prefix (what a prefix-cache hit leaves to compute): with a workspace the
    launcher splits the keys over several blocks per (request, head) and merges the partial rows -- against the oracle, and equal
    (to fp16 rounding of the merge) to the unsplit kernel."""
    _split_kv_case(quant, seqlens, start, heads, mode)


@pytest.mark.parametrize("quant", [8, 0])
@pytest.mark.parametrize("seqlens,start,heads,mode,nb", [([1, 1, 1, 1, 1, 16, 9], [1500, 1200, 3000, 1100, 2000, 4000, 2050], (4, 4), 1, 5),
                                                           ([1] * 40 + [100], list(range(600, 640)) + [8000], (4, 2), 0, 40)])
def test_attention_split_kv_behind_decode_rows(quant, seqlens, start, heads, mode, nb):
    """a continuous-batching step: decode requests (one row each) AHEAD of a short-suffix prefill request.  The partial rows of the
    split-KV form are indexed relative to the launch's first row (they were indexed by the absolute token row and ran
    past a workspace sized for the prefill rows)."""
    _split_kv_case(quant, seqlens, start, heads, mode, nb)


@pytest.mark.parametrize("quant", [8, 0])
def test_attention_split_kv_first_tile_masked(quant):
    """PPLHIP_P32_SPLIT forces one 64-key tile per split, so that a split STARTS on the tile in which some rows of a wave see no key
    (start_pos + rows crossing a 64-key boundary): those rows must contribute p = 0 and keep their initial state (the mask
    value equalled the running maximum's initial value).  The switch is read once per process: run in a child."""
    import subprocess, sys, os
    code = ("import tests.test_gpu_ops as t\n"
            f"t._split_kv_case({quant}, [32, 40], [1072, 1101], (2, 2), 0)\n"
            f"t._split_kv_case({quant}, [32], [1136], (2, 1), 1)\n"
            "print('SPLIT-OK')\n")
    env = dict(os.environ, PPLHIP_P32_SPLIT="19")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SPLIT-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_sampler_greedy_and_topk():
    m = load_pplhip()
    rng = np.random.RandomState(11)
    B, V = 33, 32000
    logits = (rng.randn(B, V) * 3).astype(np.float32)
    logits[3, 100] = logits[3, 7] = 50.0  # tie -> lower index
    desc = m.make_desc(hidden_dim=128, intermediate_dim=128, num_layers=1, num_heads=4, num_kv_heads=4, vocab_size=V)
    ctx = m.Context(desc, max_running_batch=64, max_tokens_per_step=64)
    d = dev(logits)
    tok, lp = ctx.sample(B, top_k=1, logits_ptr=d.data_ptr())
    wt, wl = ref.sample(logits, top_k=1)
    assert (tok == wt).all() and tok[3] == 7
    assert np.abs(lp - wl).max() < 1e-4
    # temperature + top-k/top-p: the kernel's random numbers come from the unseeded rand() stream like the
    # reference's (post_processor.cc:179-183); restate that stream here with libc
    temps = (0.5 + rng.rand(B)).astype(np.float32)
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    # ctx.sample consumed 1 + B values above; the next call draws 1 default + B per-row values
    tok2, lp2 = ctx.sample(B, top_k=50, top_p=0.9, temperatures=temps, logits_ptr=d.data_ptr())
    assert ((tok2 >= 0) & (tok2 < V)).all()
    # every sampled token must be inside the top-50 / top-p 0.9 nucleus of its row
    for b in range(B):
        x = logits[b] / temps[b]
        order = np.argsort(-x, kind="stable")[:50]
        p = np.exp(x[order] - x[order].max())
        p /= p.sum()
        keep = int(np.searchsorted(np.cumsum(p), 0.9) + 1)
        assert tok2[b] in order[:keep]
        lse = np.log(np.exp(x - x.max()).sum()) + x.max()
        assert abs(lp2[b] - (x[tok2[b]] - lse)) < 1e-3
    ctx.close()


@pytest.mark.parametrize("top_k,top_p", [(0, 0.8), (-3, 0.25), (5000, 0.97)])
def test_sampler_pure_top_p_and_clamped_top_k(top_k, top_p):
    """per-request values a client may send (grpc default top_k = 0 = pure top-p; top_k above the 1024-candidate cap):
    never an error; top_k <= 0 samples from the nucleus of the WHOLE-vocabulary softmax, top_k > 1024 is clamped.
    The device draws its random numbers from libc rand() (post_processor.cc:179-183): replay that stream for the oracle."""
    m = load_pplhip()
    rng = np.random.RandomState(21)
    B, V = 17, 32000
    logits = (rng.randn(B, V) * 3).astype(np.float32)
    temps = (0.5 + rng.rand(B)).astype(np.float32)
    desc = m.make_desc(hidden_dim=128, intermediate_dim=128, num_layers=1, num_heads=4, num_kv_heads=4, vocab_size=V)
    ctx = m.Context(desc, max_running_batch=64, max_tokens_per_step=64)
    d = dev(logits)
    tok, lp = ctx.sample(B, top_k=top_k, top_p=top_p, temperatures=temps, logits_ptr=d.data_ptr())
    full = top_k <= 0
    k = 1024 if full else min(top_k, 1024)
    for b in range(B):
        x = logits[b] / temps[b]
        order = np.argsort(-x, kind="stable")[:k]
        e = np.exp((x[order] - x.max()).astype(np.float64))
        tot = np.exp((x - x.max()).astype(np.float64)).sum() if full else e.sum()
        keep = int(np.searchsorted(np.cumsum(e / tot), top_p) + 1)
        assert tok[b] in order[:keep + 1]          # (+1: fp32 vs fp64 cumulative sums may disagree on the boundary candidate)
        lse = np.log(np.exp((x - x.max()).astype(np.float64)).sum()) + x.max()
        assert abs(lp[b] - (x[tok[b]] - lse)) < 1e-3
    ctx.close()


@pytest.mark.parametrize("wq", [0, 8, 4])
@pytest.mark.parametrize("M,inter,K", [(3, 64, 128), (40, 192, 256), (300, 1376, 512), (1000, 6000, 128), (1030, 640, 128)])  # W8: 128 x 384 kernel; two launches
def test_linear_swiglu_fused(wq, M, inter, K):
    """K3 + K10 fused: the GEMM over row-interleaved (gate_i, up_i) weights writes silu(gate) * up directly."""
    m = load_pplhip()
    group = 128
    rng = np.random.RandomState(M + inter + wq)
    N = 2 * inter
    x = f16(rng.randn(M, K) * 0.5)
    if wq == 0:
        w, scale = f16(rng.randn(N, K) * 0.08), None
    elif wq == 8:
        w, scale = rng.randint(-127, 128, size=(N, K)).astype(np.int8), f16(0.0008 * (0.5 + rng.rand(N)))
    else:
        w, scale = rng.randint(0, 256, size=(N, K // 2)).astype(np.uint8), f16(0.015 * (0.5 + rng.rand(N, K // group)))
    gu = np.empty((M, N), dtype=np.float32)
    xs = x.astype(np.float32)
    ref.lib().ref_linear_raw(xs.ctypes.data, w.ctypes.data, None if scale is None else scale.ctypes.data, wq, group, M, N, K,
                             gu.ctypes.data, 0)
    want = np.empty((M, inter), dtype=np.float32)
    ref.lib().ref_silu_mul(gu.ctypes.data, M, inter, want.ctypes.data)
    perm = np.empty(N, dtype=np.int64)
    perm[0::2], perm[1::2] = np.arange(inter), inter + np.arange(inter)
    wi = np.ascontiguousarray(w[perm])
    si = None if scale is None else np.ascontiguousarray(scale[perm])
    y = torch.empty((M, inter), dtype=torch.float16, device="cuda")
    dx, dw = dev(x), dev(wi)
    ds = dev(si) if si is not None else None
    ck(m.lib().pplhip_op_linear_swiglu(None, dx.data_ptr(), dw.data_ptr(), ds.data_ptr() if ds is not None else None, wq, group,
                                       M, N, K, y.data_ptr()))
    mag = np.abs(want).max()
    rel = 6e-3 if wq == 4 else 3e-3   # two fp16 roundings upstream of the product
    close_f16(y.cpu().numpy(), want, rel=rel, abs_=rel * mag * (0.25 if wq == 4 else 0.05) + 1e-5)


@pytest.mark.parametrize("M,N,K", [(129, 5120, 128), (256, 5184, 384), (512, 2560, 640), (200, 10100, 256), (257, 6976, 1152),
                                   (384, 3520, 2048)])
def test_linear_w4_tiles_128x64_rolling_pipeline(M, N, K):
    """gemm_w4_pc_kernel (k_gemm_pc.hip, round 5): W4A16 at 128 < M <= 512 when the 128 x 64 tiles fill the chip without K slabs -- one,
    three and five super-tiles (the unrolled-by-four phase loop and each of its tails), ragged last weight tile (N % 64 != 0), ragged last
    row tile, two to four row tiles.  Same oracle and tolerance as test_linear."""
    assert ((N + 63) // 64) * ((M + 127) // 128) >= 160 and 128 < M <= 512   # the dispatcher's rule for this kernel (k_gemm.hip)
    test_linear(4, M, N, K)


@pytest.mark.parametrize("M,inter,K", [(256, 3584, 1024), (300, 2600, 384), (130, 5128, 128)])
def test_linear_w4_tiles_128x64_swiglu(M, inter, K):
    """... and its fused SwiGLU epilogue (w13 of config 4), incl. an output width that is not a multiple of 32."""
    assert ((2 * inter + 63) // 64) * ((M + 127) // 128) >= 160
    test_linear_swiglu_fused(4, M, inter, K)


@pytest.mark.parametrize("wq", [0, 8, 4])
@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K", [(4096, 4096), (12288, 4096), (4096, 11008), (1000, 384), (52, 128), (2000, 9600), (640, 24576)])
def test_linear_streaming_gemv(wq, M, N, K):
    """1 <= M <= 4: the streaming GEMV of k_gemv.hip (whole 1-KiB row pieces per wave-load, fp32 VALU dot products, halving-butterfly row
    sums): 7B layer shapes, rows of 1 .. 3 pieces per wave, ragged row counts, pieces past the end of a row, fp32 and fp16 outputs."""
    test_linear(wq, M, N, K)


@pytest.mark.parametrize("wq", [0, 8, 4])
@pytest.mark.parametrize("M,inter,K", [(1, 11008, 4096), (2, 1376, 512), (3, 100, 128), (1, 40, 4096), (4, 3000, 11008)])
def test_linear_streaming_gemv_swiglu(wq, M, inter, K):
    test_linear_swiglu_fused(wq, M, inter, K)


def test_rope_kv_write_three_blocks_per_token():
    """a block count that does not divide a token's work items (PPLHIP_ROPE_BLOCKS_PER_TOKEN=3; child process: the switch is read once)"""
    import subprocess, sys, os
    code = ("import tests.test_gpu_ops as t\n"
            "for q in (0, 8):\n"
            "    t.test_rope_kv_write(q, 3, 1, 32, 8, 128)\n"
            "    t.test_rope_kv_write(q, 3, 0, 4, 4, 32)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PPLHIP_ROPE_BLOCKS_PER_TOKEN="3"), cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


def test_linear_streaming_gemv_int8_three_and_four_rows():
    """int8 weights at 3 and 4 rows go to the half-height tile kernel by default (round 4: it overtook the GEMV there); the GEMV's 3- and 4-row int8
    instantiation stays reachable with PPLHIP_GEMV_STREAM_MAX_M=4 and is held against the oracle here (child process: the switch is read once)."""
    import subprocess, sys, os
    code = ("import tests.test_gpu_ops as t\n"
            "for N, K in [(4096, 4096), (4096, 11008), (1000, 384), (640, 24576)]:\n"
            "    t.test_linear(8, 4, N, K)\n"
            "    t.test_linear(8, 3, N, K)\n"
            "t.test_linear_swiglu_fused(8, 4, 3000, 11008)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PPLHIP_GEMV_STREAM_MAX_M="4"), cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("M", [5, 16, 20, 33, 64, 65, 96, 100, 128])
@pytest.mark.parametrize("N,K", [(4096, 4096), (12288, 4096), (4096, 11008), (1000, 384), (520, 128), (8192, 1024)])
def test_linear_w8_half_height_tiles_with_128_deep_k_tiles(M, N, K):
    """2 < M <= 64, W8A16, K % 128 == 0: gemm_w8_half128_kernel (whole 128-byte lines of every weight row per LDS-DMA piece; split-K slabs
    at the 7B layer shapes, a single split at the small ones); 64 < M <= 128: the same kernel with 80- .. 128-row sub-tiles for the shapes that take split-K slabs."""
    test_linear(8, M, N, K)


@pytest.mark.parametrize("M,inter,K", [(8, 11008, 4096), (64, 1376, 512), (40, 100, 128), (100, 1376, 512), (128, 11008, 4096)])
def test_linear_w8_half128_swiglu(M, inter, K):
    test_linear_swiglu_fused(8, M, inter, K)
