"""Token-exact stochastic sampling: the device sampler vs the oracle on the SAME random numbers.

The reference draws its per-row numbers from the C library's unseeded rand() (src/backends/cuda/post_processor.cc:179-183:
one default value, then one per row, every step) and hands them to sample_topk_topp (:185-193).  libpplhip draws them the
same way, in-process, so the test restarts the generator (srand) through ctypes, replays rand() in Python to learn the
numbers the library will draw next, restarts it again, samples on the device, and feeds the oracle (ref.sample(rnd=...)) the
replayed numbers.  Tokens must be EQUAL on every row whose decision is not a cumulative-mass tie: the device accumulates
in fp32, the oracle in fp64, so a row whose target (rand * kept mass) or whose top-p boundary lies within 3e-6 of a
cumulative-mass edge may legitimately land on the neighbouring candidate; such rows are counted, not compared."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref
from tests.conftest import load_pplhip

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

RAND_MAX = 2147483647
libc = C.CDLL(None)
libc.rand.restype = C.c_int
libc.srand.argtypes = [C.c_uint]


def replay(seed, calls, batch):
    """the float32 numbers `calls` consecutive pplhip_sample calls of `batch` rows draw after srand(seed)"""
    libc.srand(seed)
    out = []
    for _ in range(calls):
        libc.rand()                                                       # the default value (post_processor.cc:179)
        r = np.array([libc.rand() for _ in range(batch)], dtype=np.int64)
        out.append((r.astype(np.float32) / np.float32(RAND_MAX)).astype(np.float32))
    libc.srand(seed)
    return out


def decision_margins(x, top_k, top_p, rnd):
    """fp64 restatement of the pick rule; returns (token, distance of the two decisions from a cumulative-mass edge)"""
    full = top_k <= 0
    k = 1024 if full else min(top_k, 1024)
    order = np.argsort(-x, kind="stable")[:k]
    e = np.exp((x[order] - x.max()).astype(np.float64))
    tot = np.exp((x - x.max()).astype(np.float64)).sum() if full else e.sum()
    cum = np.cumsum(e / tot)
    keep = min(int(np.searchsorted(cum, top_p, side="left") + 1), len(order))
    m_keep = float(np.abs(cum[:keep] - top_p).min())
    c = np.cumsum(e[:keep])
    target = float(rnd) * c[-1]
    sel = min(int(np.searchsorted(c, target, side="right")), keep - 1)
    m_pick = float(np.abs(c - target).min() / c[-1])
    return order[sel], min(m_keep, m_pick)


def make_ctx(m, V, B):
    desc = m.make_desc(hidden_dim=128, intermediate_dim=128, num_layers=1, num_heads=4, num_kv_heads=4, vocab_size=V)
    return m.Context(desc, max_running_batch=B, max_tokens_per_step=B)


@pytest.mark.parametrize("top_k", [8, 40, 0])
@pytest.mark.parametrize("top_p", [0.25, 0.9])
@pytest.mark.parametrize("with_temperature", [False, True])
def test_stochastic_sampling_is_token_exact_vs_oracle(top_k, top_p, with_temperature):
    m = load_pplhip()
    rng = np.random.RandomState(100 + top_k + int(top_p * 100) + int(with_temperature))
    B, V = 64, 32000
    logits = (rng.randn(B, V) * 2.5).astype(np.float32)
    logits[5, 77] = logits[5, 4000] = logits[5].max() + 1.0                # exact value tie at the top: lower index first
    temps = (0.5 + rng.rand(B)).astype(np.float32) if with_temperature else None
    ctx = make_ctx(m, V, B)
    d = torch.from_numpy(logits).cuda()
    steps = 3
    rnds = replay(1, steps, B)
    tied = 0
    for s in range(steps):
        tok, lp = ctx.sample(B, top_k=top_k, top_p=top_p, temperatures=temps, logits_ptr=d.data_ptr())
        wtok, wlp = ref.sample(logits, top_k=top_k, top_p=top_p, temperatures=temps, rnd=rnds[s])
        for b in range(B):
            x = logits[b] / (temps[b] if temps is not None else np.float32(1.0))
            t64, margin = decision_margins(x, top_k, top_p, rnds[s][b])
            if margin < 3e-6:
                tied += 1
                continue
            assert wtok[b] == t64, (s, b)                                  # the oracle agrees with the fp64 restatement
            assert tok[b] == wtok[b], (s, b, int(tok[b]), int(wtok[b]), float(rnds[s][b]))
            assert abs(lp[b] - wlp[b]) < 2e-4
    assert tied <= 0.04 * steps * B, tied
    ctx.close()


def test_per_row_top_p_list_and_q3_temperatures_only_on_changed_steps():
    """per-request top_p values, and quirk Q3: temperatures / top-p reach the kernel only on steps whose
    batch changed (post_processor.cc:154-177) -- on the other steps the kernel runs WITHOUT them (temperature 1, the default
    top_p), which is what the oracle is fed here."""
    m = load_pplhip()
    rng = np.random.RandomState(7)
    B, V = 48, 32000
    logits = (rng.randn(B, V) * 2.0).astype(np.float32)
    temps = (0.5 + rng.rand(B)).astype(np.float32)
    topp = (0.2 + 0.7 * rng.rand(B)).astype(np.float32)
    ctx = make_ctx(m, V, B)
    d = torch.from_numpy(logits).cuda()
    rnds = replay(3, 2, B)
    for s, changed in enumerate([True, False]):
        tok, lp = ctx.sample(B, top_k=40, top_p=0.6, temperatures=temps, top_p_list=topp, req_list_changed=changed,
                             logits_ptr=d.data_ptr())
        if changed:
            wtok, _ = ref.sample(logits, top_k=40, top_p=0.6, temperatures=temps, top_p_list=topp, rnd=rnds[s])
        else:
            wtok, _ = ref.sample(logits, top_k=40, top_p=0.6, rnd=rnds[s])
        n_cmp = 0
        for b in range(B):
            x = logits[b] / (temps[b] if changed else np.float32(1.0))
            _, margin = decision_margins(x, 40, float(topp[b]) if changed else 0.6, rnds[s][b])
            if margin < 3e-6:
                continue
            n_cmp += 1
            assert tok[b] == wtok[b], (s, b)
        assert n_cmp >= B - 2
    ctx.close()


def test_full_vocabulary_sampling_top_p_one():
    """top_k <= 0 with top_p = 1 and a flat distribution: the nucleus is wider than the 1024-candidate cap, so the pick is
    made among the 1024 most probable tokens with the whole-vocabulary mass as the denominator (DESIGN.md "sampler") -- on
    the device and in the oracle alike; must stay fast (it used to cost 1024 whole-row passes)."""
    m = load_pplhip()
    rng = np.random.RandomState(9)
    B, V = 32, 32000
    logits = (rng.randn(B, V) * 0.7).astype(np.float32)
    ctx = make_ctx(m, V, B)
    d = torch.from_numpy(logits).cuda()
    rnds = replay(5, 1, B)
    tok, lp = ctx.sample(B, top_k=0, top_p=1.0, logits_ptr=d.data_ptr())
    wtok, wlp = ref.sample(logits, top_k=0, top_p=1.0, rnd=rnds[0])
    same = 0
    for b in range(B):
        _, margin = decision_margins(logits[b], 0, 1.0, rnds[0][b])
        if margin >= 3e-6:
            assert tok[b] == wtok[b], b
            same += 1
    assert same >= B - 2
    ctx.close()


@pytest.mark.parametrize("top_k,n_ties,above", [(50, 300, 19), (8, 5000, 0), (1024, 40, 1000), (0, 2000, 500)])
def test_equal_values_straddling_the_candidate_cut_are_taken_in_index_order(top_k, n_ties, above):
    """`above` logits lie over a plateau of `n_ties` EQUAL logits that the candidate cut (top_k, or 1024 in pure top-p mode) falls
    into: the candidates are the plateau's LOWEST indices (DESIGN.md "sampler": ties -> lower index first) -- the index-selection
    passes of the radix select, which random logits never reach.  With top_p = 1 the plateau holds most of the candidate mass,
    so most picks land on plateau members; tokens must equal the oracle's."""
    m = load_pplhip()
    rng = np.random.RandomState(top_k + n_ties)
    B, V = 16, 32000
    logits = (rng.randn(B, V) * 0.5 - 6.0).astype(np.float32)
    for b in range(B):
        idx = rng.permutation(V)
        logits[b, idx[:above]] = (2.0 + rng.rand(above)).astype(np.float32)          # clearly above the plateau
        logits[b, idx[above:above + n_ties]] = np.float32(1.25)                      # the plateau
    ctx = make_ctx(m, V, B)
    d = torch.from_numpy(logits).cuda()
    seq = replay(11, 3, B)
    for s in range(3):
        tok, _ = ctx.sample(B, top_k=top_k, top_p=1.0, logits_ptr=d.data_ptr())
        wtok, _ = ref.sample(logits, top_k=top_k, top_p=1.0, rnd=seq[s])
        k = 1024 if top_k <= 0 else min(top_k, 1024)
        for b in range(B):
            _, margin = decision_margins(logits[b], top_k, 1.0, seq[s][b])
            if margin < 3e-6:
                continue
            assert tok[b] == wtok[b], (s, b, int(tok[b]), int(wtok[b]))
            # and a plateau pick is one of the plateau's k - above lowest indices
            if logits[b, tok[b]] == np.float32(1.25) and k > above:
                plateau = np.flatnonzero(logits[b] == np.float32(1.25))
                assert tok[b] in plateau[:k - above]
    ctx.close()
