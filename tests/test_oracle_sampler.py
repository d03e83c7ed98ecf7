"""CPU: the oracle's sampler (ref_sample) for the per-request parameter values a client may legally send --
top_k <= 0 ("pure top-p": nucleus over the whole vocabulary) and top_k above the candidate cap -- against a plain
numpy restatement.  (The GPU kernel is compared with the same rules in tests/test_gpu_ops.py.)"""
import numpy as np

from oracle import ref


def nucleus(x, k, top_p, full):
    order = np.argsort(-x, kind="stable")[:k]
    e = np.exp((x[order] - x.max()).astype(np.float64))
    tot = np.exp((x - x.max()).astype(np.float64)).sum() if full else e.sum()
    keep = int(np.searchsorted(np.cumsum(e / tot), top_p) + 1)
    return order[:min(keep, len(order))], e


def test_pure_top_p_and_clamped_top_k():
    rng = np.random.RandomState(3)
    B, V = 9, 5000
    logits = (rng.randn(B, V) * 2.5).astype(np.float32)
    temps = (0.6 + rng.rand(B)).astype(np.float32)
    for top_k, top_p in [(0, 0.8), (-1, 0.3), (4000, 0.95), (50, 0.9)]:
        for trial in range(4):
            rnd = rng.rand(B).astype(np.float32)
            tok, lp = ref.sample(logits, top_k=top_k, top_p=top_p, temperatures=temps, rnd=rnd)
            for b in range(B):
                x = logits[b] / temps[b]
                full = top_k <= 0
                k = 1024 if full else min(top_k, 1024)
                cand, e = nucleus(x, k, top_p, full)
                assert tok[b] in cand
                # the pick is the first candidate whose cumulative mass exceeds rnd * kept mass
                c = np.cumsum(e[:len(cand)])
                want = cand[min(int(np.searchsorted(c, rnd[b] * c[-1], side="right")), len(cand) - 1)]
                assert tok[b] == want
                lse = np.log(np.exp((x - x.max()).astype(np.float64)).sum()) + x.max()
                assert abs(lp[b] - (x[tok[b]] - lse)) < 1e-4


def test_greedy_is_unchanged():
    rng = np.random.RandomState(4)
    logits = rng.randn(5, 300).astype(np.float32)
    tok, _ = ref.sample(logits, top_k=1)
    assert (tok == logits.argmax(-1)).all()
