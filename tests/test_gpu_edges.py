"""Edge cases of the step boundary (pplhip_set_inputs / pplhip_run): empty step, exact capacity, last rope position, and the
inputs the library must REJECT instead of indexing the rope table, the embedding table or the KV slab out of range."""
import numpy as np
import pytest

from oracle import ref
from tests.conftest import load_pplhip

pytestmark = pytest.mark.gpu

I64MAX = np.iinfo(np.int64).max


def make(m, mode=0, batch=8, tokens=64, max_position=128, kv_tokens=256):
    desc = ref.make_desc(hidden_dim=128, intermediate_dim=256, num_layers=2, num_heads=4, num_kv_heads=2, vocab_size=320,
                         max_position=max_position, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=mode,
                         page_size=8 if mode else 0, weight_quant_bit=8)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=batch, max_tokens_per_step=tokens)
    ctx.init_synthetic(0, 21)
    ctx.kv_alloc(0, kv_tokens)
    rm = ref.RefModel(desc)
    rm.init_synthetic(21)
    rm.kv_alloc(kv_tokens)
    return desc, ctx, rm


def test_empty_step_is_a_no_op():
    m = load_pplhip()
    _, ctx, _ = make(m)
    st = m.make_step(np.zeros(0, dtype=np.int64), [0], np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), 0)
    ctx.set_inputs(0, st)
    ctx.run(0)
    ctx.sync(0)
    ctx.close()


def test_exact_capacity_and_last_position():
    m = load_pplhip()
    desc, ctx, rm = make(m, batch=4, tokens=40, max_position=64, kv_tokens=4 * 64)
    rng = np.random.RandomState(0)
    # T == max_tokens_per_step, B == max_running_batch; request 3 ends exactly at max_position after the decode step
    lens = [10, 10, 10, 10]
    sp = np.array([0, 0, 0, 53], dtype=np.int64)
    tok = rng.randint(3, 320, size=40)
    seq = np.concatenate([[0], np.cumsum(lens)])
    ci = np.arange(4, dtype=np.int64) * 64
    st = m.make_step(tok, seq, sp, ci, 0)
    ctx.set_inputs(0, st); ctx.run(0)
    got = ctx.copy_logits(4)
    want = ref.forward([rm], ref.make_step(tok, seq, sp, ci, 0))
    assert np.abs(got - want).max() <= 8e-3 * max(1.0, np.abs(want).max())
    nxt = rng.randint(3, 320, size=4)
    sp2 = sp + 10                                      # request 3 now sits at position 63 = max_position - 1
    st = m.make_step(nxt, np.arange(5), sp2, ci, 4, req_list_changed=0)
    ctx.set_inputs(0, st); ctx.run(0)
    got = ctx.copy_logits(4)
    want = ref.forward([rm], ref.make_step(nxt, np.arange(5), sp2, ci, 4))
    assert np.abs(got - want).max() <= 8e-3 * max(1.0, np.abs(want).max())
    ctx.close()


@pytest.mark.parametrize("what", ["seq_starts", "token", "position", "kv_range", "decode_len", "no_tokens", "dec_batches"])
def test_invalid_steps_are_rejected(what):
    m = load_pplhip()
    _, ctx, _ = make(m, batch=4, tokens=32, max_position=64, kv_tokens=128)
    tok = np.array([5, 6, 7, 8, 9, 10], dtype=np.int64)
    seq, sp, ci, dec = np.array([0, 4, 6]), np.array([0, 0], dtype=np.int64), np.array([0, 64], dtype=np.int64), 0
    if what == "seq_starts":
        seq = np.array([0, 4, 5])                      # does not end at num_tokens
    elif what == "token":
        tok = tok.copy(); tok[2] = 320                 # == vocab_size
    elif what == "position":
        sp = np.array([0, 63], dtype=np.int64)         # 63 + 2 > max_position
    elif what == "kv_range":
        ci = np.array([0, 127], dtype=np.int64)        # 127 + 2 > 128 slab tokens
    elif what == "decode_len":
        dec = 1                                        # request 0 has 4 tokens but is flagged as a decode row
    elif what == "no_tokens":
        seq = np.array([0, 6, 6])
    elif what == "dec_batches":
        dec = 3
    with pytest.raises(m.PplHipError):
        ctx.set_inputs(0, m.make_step(tok, seq, sp, ci, dec))
    # the context stays usable
    ctx.set_inputs(0, m.make_step(np.array([5, 6, 7, 8, 9, 10]), [0, 4, 6], [0, 0], [0, 64], 0))
    ctx.run(0)
    assert np.isfinite(ctx.copy_logits(2)).all()
    ctx.close()


def test_invalid_page_lists_are_rejected():
    m = load_pplhip()
    _, ctx, _ = make(m, mode=1, batch=4, tokens=32, max_position=64, kv_tokens=128)   # 16 pages of 8 tokens
    tok = np.arange(3, 13, dtype=np.int64)
    seq, sp = np.array([0, 10]), np.array([0], dtype=np.int64)
    with pytest.raises(m.PplHipError):                                                 # page 16 is outside the slab
        ctx.set_inputs(0, m.make_step(tok, seq, sp, np.array([[3, 16]]), 0, max_pages=2))
    with pytest.raises(m.PplHipError):                                                 # 10 tokens need 2 pages
        ctx.set_inputs(0, m.make_step(tok, seq, sp, np.array([[3]]), 0, max_pages=1))
    with pytest.raises(m.PplHipError):                                                 # padding where a page is needed
        ctx.set_inputs(0, m.make_step(tok, seq, sp, np.array([[3, I64MAX]]), 0, max_pages=2))
    ctx.set_inputs(0, m.make_step(tok, seq, sp, np.array([[3, 15, I64MAX]]), 0, max_pages=3))
    ctx.run(0)
    assert np.isfinite(ctx.copy_logits(1)).all()
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_decode_steps_replayed_as_hip_graphs_are_bit_identical(monkeypatch, mode):
    """PPLHIP_DECODE_GRAPH=1: the second decode step of a shape is captured, later ones are replayed -- token ids, positions,
    cache slots and page lists come from the step buffer, so the replay must give exactly the eager logits, also across a page
    boundary (the page-table width is part of the shape key) and after the batch shrank and grew back."""
    m = load_pplhip()

    def run(graph):
        monkeypatch.setenv("PPLHIP_DECODE_GRAPH", "1" if graph else "0")
        desc, ctx, _ = make(m, mode=mode, batch=6, tokens=64, max_position=128, kv_tokens=6 * 48)
        rng = np.random.RandomState(4)
        lens = np.array([5, 9, 3, 7, 2, 6])
        ci = (np.arange(6, dtype=np.int64) * 48) if mode == 0 else np.arange(6 * 6, dtype=np.int64).reshape(6, 6)
        mp = 6 if mode else 0
        tok = rng.randint(3, 320, size=int(lens.sum()))
        ctx.set_inputs(0, m.make_step(tok, np.concatenate([[0], np.cumsum(lens)]), np.zeros(6, dtype=np.int64), ci, 0, max_pages=mp))
        ctx.run(0)
        out = [ctx.copy_logits(6)]
        sp = lens.copy()
        rows = np.arange(6)
        for i in range(14):
            if i == 6:
                rows = np.array([0, 2, 3, 5])          # two requests leave ...
            if i == 10:
                rows = np.arange(6)                    # ... and the same shape comes back
            nxt = rng.randint(3, 320, size=len(rows))
            st = m.make_step(nxt, np.arange(len(rows) + 1), sp[rows], ci[rows], len(rows), max_pages=mp,
                             req_list_changed=int(i in (0, 6, 10)))
            ctx.set_inputs(0, st)
            ctx.run(0)
            out.append(ctx.copy_logits(len(rows)))
            sp[rows] += 1
        ctx.close()
        return out

    eager, graph = run(False), run(True)
    assert len(eager) == len(graph)
    for a, b in zip(eager, graph):
        assert (a == b).all()
