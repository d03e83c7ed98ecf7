#!/usr/bin/env python3
"""One-off robustness run (round 5): pplhip_op_rope_kv_write + pplhip_op_attention on random MIXED steps -- a random number of decode rows
(kv 1 .. 1500) in front of a random number of prefill / cache-prefill requests (1 .. 400 new tokens behind 0 .. 1200 cached ones), random
head geometry (multi-head, 2 : 1 and grouped-query 4 : 1 .. 16 : 1), head size, fp16 / int8-g8 KV, all four cache layouts, contiguous and
paged, random decode split -- against the oracle (ref_rope_kv_write / ref_attention) with the tolerances of tests/test_gpu_ops.py.
usage: python profiles/probes/attention_fuzz.py [seconds] [seed]"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_gpu_ops as T
from tests.conftest import load_pplhip

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
m = load_pplhip()
GEOM = [(4, 4, 32), (8, 2, 64), (4, 4, 128), (8, 1, 128), (16, 1, 64), (12, 2, 128), (8, 8, 128), (6, 3, 64), (16, 2, 128), (32, 4, 128)]
t0, n, bad, worst = time.time(), 0, [], 0.0
while time.time() - t0 < budget:
    H, Hkv, D = GEOM[rng.randint(len(GEOM))]
    quant = int(rng.choice([0, 8]))
    mode = int(rng.rand() < 0.5)
    layout = int(rng.choice([0, 1, 2, 3]))
    nd = int(rng.choice([0, 1, 2, 5, 9, 17]))
    npf = int(rng.choice([0, 1, 2, 3])) if nd else int(rng.choice([1, 2, 4]))
    kvlen = [int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 700, 1500]) if rng.rand() < 0.5 else rng.randint(1, 1500)) for _ in range(nd)]
    new = [int(rng.choice([1, 2, 31, 32, 33, 64, 127, 128, 129, 400]) if rng.rand() < 0.5 else rng.randint(1, 400)) for _ in range(npf)]
    cached = [int(rng.choice([0, 0, 1, 16, 100, 1200])) for _ in range(npf)]
    seqlens = [1] * nd + new
    start = [k - 1 for k in kvlen] + cached
    split = int(rng.choice([1, 1, 2, 3, 5]))
    case_id = (H, Hkv, D, quant, layout, mode, kvlen, new, cached, split)
    try:
        case = T.KvCase(m, H, Hkv, D, L=2, layer=int(rng.randint(2)), quant=quant, layout=layout, mode=mode, seqlens=seqlens, start_pos=start,
                        seed=int(rng.randint(1 << 30)), page_size=int(rng.choice([4, 16])), decoding_batches=nd)
        if quant:
            case.cache[:] = rng.randint(-127, 128, size=case.cache.size).astype(np.int8)
            case.scale[:] = T.f16(0.02 * (0.5 + rng.rand(case.scale.size)))
        else:
            case.cache[:] = T.f16(rng.randn(case.cache.size))
        hist_cache, hist_scale = case.cache.copy(), (case.scale.copy() if quant else None)
        q32 = case.ref_write()                       # oracle: rope + KV write (case.cache now holds the new tokens)
        want = case.ref_attention(q32)
        dq = T.dev(case.qkv)
        dcache, dscale = T.dev(hist_cache), (T.dev(hist_scale) if quant else None)
        v = case.view(dcache, dscale)
        args = (T.dev(case.seq_starts), T.dev(case.start_pos), T.dev(case.cache_idx))
        T.ck(m.lib().pplhip_op_rope_kv_write(None, dq.data_ptr(), T.dev(case.rope).data_ptr(), C.byref(v), args[0].data_ptr(), args[1].data_ptr(),
                                             args[2].data_ptr(), case.max_pages, case.B, case.T, H))
        # rope + KV write: bit exact
        assert (dq.cpu().numpy().astype(np.float32)[:, :H * D] == q32[:, :H * D]).all(), "rotated q"
        if quant:
            assert (dcache.cpu().numpy() == case.cache).all() and (dscale.cpu().numpy().view(np.uint16) == case.scale.view(np.uint16)).all(), "cache"
        else:
            assert (dcache.cpu().numpy().view(np.uint16) == case.cache.view(np.uint16)).all(), "cache"
        out = torch.zeros((case.T, H * D), dtype=torch.float16, device="cuda")
        ws = torch.empty(max(1, case.B) * H * split * (D + 2) + (64 << 20) // 4, dtype=torch.float32, device="cuda")
        T.ck(m.lib().pplhip_op_attention(None, dq.data_ptr(), C.byref(v), args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(), case.max_pages,
                                         case.B, case.T, nd, case.max_seq_len, case.max_kv_len, H, split, ws.data_ptr(), ws.numel() * 4, out.data_ptr()))
        got = out.cpu().numpy().astype(np.float32)
        vmax = 3.0 if not quant else 0.03 * 127
        if nd: T.close_f16(got[:nd], want[:nd], rel=1.5e-3, abs_=1.5e-3)
        if npf:
            T.close_f16(got[nd:], want[nd:], rel=1e-3, abs_=1e-3 * vmax)
            worst = max(worst, float(np.abs(got[nd:] - want[nd:]).max()) / vmax)
    except AssertionError as e:
        bad.append((case_id, str(e)[:200])); print("FAIL", case_id, str(e)[:200], flush=True)
    except Exception as e:
        bad.append((case_id, repr(e)[:200])); print("ERROR", case_id, repr(e)[:200], flush=True)
    n += 1
print(f"{n} cases in {time.time() - t0:.0f} s, {len(bad)} failures; worst prefill error {worst:.2e} of |V|max (tolerance 1e-3)")
