import numpy as np, sys
sys.path.insert(0, '.')
from oracle import ref
from tests.test_gpu_model import plan_cache
def trace(wq, kvq, batch, hkv, mode, tokens=None, steps=3):
    desc = ref.make_desc(hidden_dim=2048, intermediate_dim=5632, num_layers=2, num_heads=16, num_kv_heads=hkv, vocab_size=1024,
                         max_position=512, cache_quant_bit=kvq, cache_quant_group=8 if kvq else 1, cache_layout=3,
                         cache_mode=0, weight_quant_bit=wq, weight_quant_group=128)
    with ref.mode(mode):
        rm = ref.RefModel(desc); rm.init_synthetic(31)
        ntok = batch * 16 + 64
        rm.kv_alloc(ntok)
        rng = np.random.RandomState(batch)
        prompts = [rng.randint(3, 1024, size=int(n)) for n in rng.randint(1, 9, size=batch)]
        n = len(prompts); lens = np.array([len(p) for p in prompts])
        cache_idx, max_pages = plan_cache(desc, lens + steps, ntok)
        tok = np.concatenate(prompts).astype(np.int64)
        seq = np.concatenate([[0], np.cumsum(lens)]); sp = np.zeros(n, dtype=np.int64)
        outs, toks = [], []
        for s in range(steps):
            dec = 0 if s == 0 else n
            want = ref.forward([rm], ref.make_step(tok, seq, sp, cache_idx, dec, max_pages))
            outs.append(want)
            wtok = ref.sample(want, top_k=1)[0].astype(np.int64) if tokens is None else tokens[s]
            toks.append(wtok)
            sp = sp + (seq[1:] - seq[:-1]); tok = wtok; seq = np.arange(n + 1)
        rm.close()
    return outs, toks
for case in [(4, 0, 120, 2), (4, 8, 120, 2)]:
    base, toks = trace(*case, ref.MODE_FP16)
    for name, md in [("alt", ref.MODE_ALT_ORDER), ("alt2", ref.MODE_ALT_ORDER2), ("fp32act+f64acc", ref.MODE_FP32_ACT | ref.MODE_F64_ACC)]:
        o, _ = trace(*case, md, tokens=toks)
        line = []
        for s in range(3):
            sc = max(1.0, np.abs(base[s]).max()); e = np.abs(o[s] - base[s]).max(-1) / sc
            line.append((round(float(e.max()), 5), int(e.argmax()), int((e > 5e-3).sum())))
        print(case, name, line, flush=True)
