"""Is the prefill / decode attention output of a request bit-identical when the request sits at another row offset of the step
and another slot base of the KV slab?  (it must be: nothing in the algorithm depends on either)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
m = load_pplhip()
H = HKV = 32; D = 128; N = 4096
torch.manual_seed(0)
def run(qlen, kvlen, r0, s0, T, nreq_before):
    """request with `qlen` query rows at rows [r0, r0+qlen), start_pos = kvlen - qlen, KV slots [s0, s0+kvlen)"""
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    kd = torch.randint(-127, 128, (2, HKV, kvlen, D), dtype=torch.int8, device="cuda", generator=g)
    ks = (torch.rand(2, HKV, kvlen, D // 8, device="cuda", generator=g) * 0.02 + 0.01).half()
    q = torch.randn(qlen, 3 * H * D, device="cuda", generator=g).half()
    cache = torch.zeros(2, HKV, N, D, dtype=torch.int8, device="cuda"); scale = torch.zeros(2, HKV, N, D // 8, dtype=torch.float16, device="cuda")
    cache[:, :, s0:s0 + kvlen] = kd; scale[:, :, s0:s0 + kvlen] = ks
    qkv = torch.randn(T, 3 * H * D, device="cuda").half(); qkv[r0:r0 + qlen] = q
    # requests: `nreq_before` one-row prefill requests in front (rows 0..), filler request(s) to reach r0, then ours
    lens = [1] * nreq_before + ([r0 - nreq_before] if r0 > nreq_before else []) + [qlen] + ([T - r0 - qlen] if T > r0 + qlen else [])
    me = nreq_before + (1 if r0 > nreq_before else 0)
    seq = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device="cuda")
    B = len(lens)
    sp = torch.zeros(B, dtype=torch.int64, device="cuda"); sp[me] = kvlen - qlen
    ci = torch.full((B,), 3000, dtype=torch.int64, device="cuda"); ci[me] = s0
    dec = 0
    if qlen == 1:   # decode row must come first
        assert r0 == 0; dec = 1
    v = m.KvView()
    v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = cache.data_ptr(), scale.data_ptr(), N, 1, HKV, D
    v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, 0, 0, 0
    out = torch.zeros(T, H * D, device="cuda", dtype=torch.float16)
    ws = torch.empty(1 << 20, device="cuda", dtype=torch.float32)
    rc = m.lib().pplhip_op_attention(None, qkv.data_ptr(), C.byref(v), seq.data_ptr(), sp.data_ptr(), ci.data_ptr(), 0, B, T, dec, max(lens), kvlen + 200,
                                     H, 1, ws.data_ptr(), ws.numel() * 4, out.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    return out[r0:r0 + qlen].clone()
for qlen, kvlen in [(130, 130), (5, 5), (17, 17), (64, 64), (130, 200)]:
    base = run(qlen, kvlen, 0, 0, 300, 0)
    for r0, s0, nb in [(0, 10, 0), (0, 16, 0), (38, 0, 0), (124, 0, 0), (119, 127, 3), (38, 42, 2)]:
        o = run(qlen, kvlen, r0, s0, 300, nb)
        d = (o.float() - base.float()).abs().max().item()
        print(f"qlen {qlen} kv {kvlen}: row offset {r0}, slot base {s0}, {nb} requests before -> max diff {d:.3e}")
