"""Host logic of the hot path (hashing, prefix cache, allocators, request scheduler, ModelInput packing):
  1. the Python oracle (oracle/host_logic.py) against golden values captured from the reference's own code;
  2. the C++ implementation (ppl.llm.serving_amd/src, driven by tests/host/sched_trace with a fake backend)
     against the golden values and, step by step and bit-exactly, against the oracle on scripted request traces;
  3. where /root/reference exists (the build container): THE REFERENCE'S OWN llm_generator.cc / llm_engine.cc, compiled in place
     against the ppl.nn / ppl.common surface of src/compat (tests/host/ref_sched_trace.cc, `make ref`), on the same traces --
     what its engine binds by index and copies per step equals what the repo's generator packs."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import host_logic as hl
from tests.conftest import ROOT

PKG = os.path.join(ROOT, "ppl.llm.serving_amd")
TRACE = os.path.join(PKG, "build", "sched_trace")


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "host_logic.json")))


@pytest.fixture(scope="module")
def trace_bin():
    subprocess.check_call(["make", "-s", "-C", PKG, "build/sched_trace"])
    return TRACE


def test_oracle_hash_combine_golden(golden):
    for case in golden["hash_combine"]:
        assert hl.hash_combine(int(case["prev"]), case["vec"]) == int(case["expect"])


def test_oracle_prefix_cache_golden(golden):
    g = golden["prefix_cache"]
    p = hl.PrefixCacheManager()
    for h, page in g["insert"]:
        p.insert(h, page)
    p.dec_ref(g["dec_ref"])
    assert p.evict(g["evict"]) == g["expect_evicted"]
    assert p.size() == g["expect_size"]
    for k, v in g["expect_find"].items():
        assert p.find(int(k)) == v


def test_oracle_scheduler_example_golden(golden):
    ex = golden["scheduler_example"]
    steps, _, _ = hl.simulate(ex["scenario"])
    for want in ex["expect_steps"]:
        got = steps[want["step"]]
        for k, v in want.items():
            assert got[k] == v, (k, got[k], v)


def test_cpp_unit_known_answers(trace_bin, golden):
    u = json.loads(subprocess.check_output([trace_bin, "--unit"], stderr=subprocess.DEVNULL))
    hc = golden["hash_combine"]
    assert [u["hash_a"], u["hash_b"], u["hash_p1"], u["hash_p2"]] == [c["expect"] for c in hc]
    g = golden["prefix_cache"]
    assert u["prefix_evicted"] == g["expect_evicted"] and u["prefix_size"] == g["expect_size"]
    assert u["prefix_find2"] == 13 and u["prefix_find0"] == -1
    assert u["prefix_evicted2"] == [14] and u["prefix_size2"] == 1          # a re-referenced page leaves the LRU
    assert u["index_allocs"] == [0, 30, 60, hl.INT64_MAX, 30, hl.INT64_MAX] and u["index_avail"] == 60
    assert u["pages"] == [0, 1, 2, 1, 3] and u["pages_rc"] == 3 and u["pages_avail"] == 0
    assert u["cfg_ok"] == 1 and u["cfg_kv_heads"] == 32 and u["cfg_page"] == 16 and u["cfg_wq"] == 8
    assert u["cfg_missing"] == 0 and u["cfg_nopage"] == 0                   # missing mandatory keys are errors
    assert u["mpsc_popped"] == 1000 and u["mpsc_stash_same"] == 1 and u["mpsc_fifo"] == 1 and u["mpsc_pending"] == 0
    assert u["parse_tokens"] == [2, 7, 13]


def run_cpp(trace_bin, sc):
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(sc, f)
        path = f.name
    try:
        out = subprocess.check_output([trace_bin, path], stderr=subprocess.DEVNULL, timeout=int(os.environ.get("SCHED_TRACE_TIMEOUT", "120"))).decode().strip().split("\n")
    finally:
        os.unlink(path)
    lines = [json.loads(l) for l in out]
    return lines[:-1], lines[-1]


def compare(trace_bin, sc):
    steps, responses, failed = hl.simulate(sc)
    csteps, cfinal = run_cpp(trace_bin, sc)
    assert len(csteps) == len(steps), (len(csteps), len(steps))
    for a, b in zip(csteps, steps):
        for k in b:
            assert a[k] == b[k], (b["step"], k, a[k], b[k])
    assert {int(k): v for k, v in cfinal["responses"].items()} == responses
    assert {int(k): v for k, v in cfinal["failed"].items()} == failed
    return steps, responses, failed


def rand_requests(rng, n, vocab, max_prompt, max_gen, shared_prefix=None):
    reqs = []
    for i in range(n):
        toks = rng.randint(3, vocab, size=rng.randint(1, max_prompt + 1)).tolist()
        if shared_prefix is not None and rng.rand() < 0.7:
            toks = shared_prefix[:rng.randint(1, len(shared_prefix) + 1)] + toks
        reqs.append({"id": i, "tokens": toks, "generation_length": int(rng.randint(1, max_gen + 1))})
    return reqs


def test_cpp_scheduler_example(trace_bin, golden):
    steps, responses, _ = compare(trace_bin, golden["scheduler_example"]["scenario"])
    assert len(responses[0]["tokens"]) == 3 and responses[0]["finish"] == 1     # FinishFlag::LENGTH
    assert len(responses[1]["tokens"]) == 2


@pytest.mark.parametrize("seed", range(6))
def test_cpp_contiguous_mode_with_kv_pressure(trace_bin, seed):
    """cache_mode 0; a KV pool far smaller than the demand exercises the cool-down path (llm_generator.cc:488-492,
    637, 727-728), the token budget (Q7) and the batch limits."""
    rng = np.random.RandomState(seed)
    sc = {"model": {"cache_mode": 0, "vocab_size": 997},
          "generator": {"max_running_batch": 6, "max_tokens_per_step": 48, "max_prefill_batch": 3, "max_cooldown_request": 2,
                        "stop_tokens": [5, 6, 7, 8, 9, 10, 11, 12]},
          "kv_cache_max_tokens": 160, "requests": rand_requests(rng, 30, 997, 20, 12)}
    for r in sc["requests"][::4]:
        r["stop_tokens"] = list(range(100, 140))
    sc["requests"][3]["early_stopping"] = False
    steps, responses, failed = compare(trace_bin, sc)
    assert len(responses) + len(failed) == 30
    assert max(len(s["start_pos"]) for s in steps) <= 6


@pytest.mark.parametrize("seed", range(4))
def test_cpp_paged_mode(trace_bin, seed):
    rng = np.random.RandomState(100 + seed)
    sc = {"model": {"cache_mode": 1, "page_size": 4, "vocab_size": 500},
          "generator": {"max_running_batch": 8, "max_tokens_per_step": 64, "max_prefill_batch": 4, "enable_penalty": True},
          "kv_cache_max_tokens": 256, "requests": rand_requests(rng, 24, 500, 18, 9)}
    compare(trace_bin, sc)


@pytest.mark.parametrize("seed", range(4))
def test_cpp_prefix_cache(trace_bin, seed):
    """prefix cache on (max_prefill_batch forced to 1): chained page hashes, hits, partial hits, full hits
    (start_pos = hit - 1), LRU eviction under page pressure, dec-ref on finish."""
    rng = np.random.RandomState(200 + seed)
    shared = rng.randint(3, 300, size=24).tolist()
    reqs = rand_requests(rng, 20, 300, 10, 6, shared_prefix=shared)
    reqs.append({"id": 20, "tokens": shared[:16], "generation_length": 3})      # whole prompt cached
    reqs.append({"id": 21, "tokens": shared[:16], "generation_length": 2})
    sc = {"model": {"cache_mode": 1, "page_size": 4, "vocab_size": 300},
          "generator": {"max_running_batch": 6, "max_tokens_per_step": 128, "enable_prefix_cache": True},
          "kv_cache_max_tokens": 120, "requests": reqs}
    steps, _, _ = compare(trace_bin, sc)
    assert any(s["prefix_hit"] for s in steps)


def test_cpp_prefix_worked_example(trace_bin):
    """SURVEY.md section 10 worked example: 10-token prompt, first 8 tokens cached in two pages -> start_pos 8,
    token_inputs = tokens[8:10], kv_starts [0, 10], one new page appended after the cached ones."""
    base = [50, 51, 52, 53, 54, 55, 56, 57]
    sc = {"model": {"cache_mode": 1, "page_size": 4, "vocab_size": 1000},
          "generator": {"enable_prefix_cache": True, "max_running_batch": 4},
          "kv_cache_max_tokens": 64,
          "requests": [{"id": 0, "tokens": base + [1, 2], "generation_length": 1},
                       {"id": 1, "tokens": base + [90, 91], "generation_length": 3}]}
    steps, _, _ = compare(trace_bin, sc)
    s = [x for x in steps if x["prefix_hit"]][0]
    assert s["token_inputs"] == [90, 91] and s["start_pos"] == [8] and s["kv_starts"] == [0, 10]
    assert s["seq_starts"] == [0, 2] and s["max_seq_len"] == 2 and s["max_kv_len"] == 10 and s["max_pages"] == 3
    assert s["page_list"][:2] == steps[0]["page_list"][:2]          # the two cached pages are reused


def test_cpp_limits_and_failures(trace_bin):
    sc = {"model": {"cache_mode": 0, "vocab_size": 400},
          "generator": {"max_running_batch": 4, "max_input_tokens_per_request": 8, "max_output_tokens_per_request": 5,
                        "max_total_tokens_per_request": 10, "max_tokens_per_step": 32},
          "kv_cache_max_tokens": 128,
          "requests": [{"id": 0, "tokens": list(range(10, 19)), "generation_length": 2},     # prompt too long -> failure
                       {"id": 1, "tokens": [7, 8, 9], "generation_length": 50},                # total clamp wins: 10 - 3 = 7
                       {"id": 2, "tokens": list(range(20, 28)), "generation_length": 4},       # total clamp -> 2
                       {"id": 3, "tokens": [], "generation_length": 3},                        # empty prompt -> failure
                       {"id": 4, "tokens": [5], "generation_length": 0},                       # nothing to generate
                       {"id": 5, "tokens": [3, 4], "generation_length": 1}]}
    steps, responses, failed = compare(trace_bin, sc)
    assert sorted(failed) == [0, 3, 4] and all(v == 2 for v in failed.values())
    # reference quirk kept (llm_generator.cc:465-476): the total-length clamp is computed from the REQUESTED length and
    # overrides the max_output clamp, so request 1 generates 7 (> max_output_tokens_per_request = 5) tokens
    assert len(responses[1]["tokens"]) == 7 and len(responses[2]["tokens"]) == 2 and len(responses[5]["tokens"]) == 1


def test_cpp_cancel_and_execute_failure(trace_bin):
    rng = np.random.RandomState(7)
    sc = {"model": {"cache_mode": 1, "page_size": 8, "vocab_size": 600}, "generator": {"max_running_batch": 8},
          "kv_cache_max_tokens": 512, "requests": rand_requests(rng, 6, 600, 12, 10),
          "cancel": [{"at_step": 1, "id": 2}, {"at_step": 2, "id": 99}, {"at_step": 2, "id": 0}]}
    for r in sc["requests"]:
        r["generation_length"] = 8
    steps, responses, _ = compare(trace_bin, sc)
    assert len(responses[2]["tokens"]) == 2 and responses[2]["finish"] == 0     # cancelled by the connection
    sc2 = dict(sc)
    sc2.pop("cancel")
    sc2["fail_at_run"] = 3
    steps, responses, failed = compare(trace_bin, sc2)
    assert sorted(failed) == [0, 1, 2, 3, 4, 5] and len(steps) == 4


def test_cpp_cancel_on_a_quiet_step_marks_the_batch_changed(trace_bin):
    """paged mode, three running requests, empty queue, one of them cancelled by the connection while nothing finishes
    and nothing is admitted: the rows shift, so the NEXT step must carry req_list_changed = 1 and a repacked page list
    (otherwise the device keeps the old rows' page table and the survivors read / write another request's pages)."""
    sc = {"model": {"cache_mode": 1, "page_size": 4, "vocab_size": 700}, "generator": {"max_running_batch": 8},
          "kv_cache_max_tokens": 256,
          "requests": [{"id": i, "tokens": [10 + i, 20 + i, 30 + i, 40 + i, 50 + i], "generation_length": 12, "early_stopping": False}
                       for i in range(3)],
          "cancel": [{"at_step": 3, "id": 0}]}
    steps, responses, _ = compare(trace_bin, sc)
    assert [s["req_list_changed"] for s in steps[:6]] == [1, 0, 0, 0, 1, 0]
    before, after = steps[3], steps[4]
    mp = before["max_pages"]
    assert after["page_list"] == before["page_list"][mp:]                 # rows 1, 2 moved up with THEIR pages
    assert len(responses[0]["tokens"]) == 4 and len(responses[1]["tokens"]) == 12 and len(responses[2]["tokens"]) == 12
    # survivors' tokens equal an uncancelled run's
    sc2 = dict(sc)
    sc2.pop("cancel")
    _, responses2, _ = compare(trace_bin, sc2)
    assert responses[1] == responses2[1] and responses[2] == responses2[2]


# ---------------------------------------------------------------------------------------------------------------
# 3. the reference's own generator + engine, compiled in place (boundary proof)
# ---------------------------------------------------------------------------------------------------------------
REFERENCE = "/root/reference"
REF_TRACE = os.path.join(PKG, "build", "ref_sched_trace")


@pytest.fixture(scope="module")
def ref_trace_bin():
    if not os.path.isdir(os.path.join(REFERENCE, "src", "generator")):
        pytest.skip("the reference tree is not on this machine (GPU box): nothing to compile")
    subprocess.check_call(["make", "-s", "-C", PKG, "ref"])
    return REF_TRACE


def compare_with_reference(ref_trace_bin, sc, skip_upload_flag_at=()):
    """the reference-compiled generator on scenario `sc` against the oracle (== the repo's generator, tests above)"""
    steps, responses, failed = hl.simulate(sc)
    sc = dict(sc, expect_done=sum(1 for r in responses.values() if r["finish"]) + len(failed))
    rsteps, rfinal = run_cpp(ref_trace_bin, sc)
    assert len(rsteps) == len(steps), (len(rsteps), len(steps))
    mode = sc["model"].get("cache_mode", 0)
    for a, b in zip(rsteps, steps):
        # (the driver's "step" counts Execute calls; the generator's own loop_step restarts with every Generate() call and is not
        # visible to a backend)
        for k in ("decoding_batches", "max_seq_len", "max_kv_len", "token_inputs", "seq_starts", "kv_starts", "start_pos", "prefix_hit"):
            assert a[k] == b[k], (b["step"], k, a[k], b[k])
        if mode == 0:
            assert a["cache_indices"] == b["cache_indices"], b["step"]
        else:
            assert a["max_pages"] == b["max_pages"] and a["page_list"] == b["page_list"], b["step"]
            if b["step"] not in skip_upload_flag_at:
                assert a["pages_uploaded"] == b["req_list_changed"], b["step"]   # llm_engine.cc:67-71
    assert {int(k): v for k, v in rfinal["responses"].items()} == responses
    assert sorted(int(k) for k in rfinal["failed"]) == sorted(failed)
    return rsteps


def test_reference_sources_scheduler_example(ref_trace_bin, golden):
    compare_with_reference(ref_trace_bin, golden["scheduler_example"]["scenario"])


@pytest.mark.parametrize("seed", range(3))
def test_reference_sources_contiguous_mode_with_kv_pressure(ref_trace_bin, seed):
    rng = np.random.RandomState(seed)
    sc = {"model": {"cache_mode": 0, "vocab_size": 997},
          "generator": {"max_running_batch": 6, "max_tokens_per_step": 48, "max_prefill_batch": 3, "max_cooldown_request": 2,
                        "stop_tokens": [5, 6, 7, 8, 9, 10, 11, 12]},
          "kv_cache_max_tokens": 160, "requests": rand_requests(rng, 30, 997, 20, 12)}
    for r in sc["requests"][::4]:
        r["stop_tokens"] = list(range(100, 140))
    sc["requests"][3]["early_stopping"] = False
    compare_with_reference(ref_trace_bin, sc)


@pytest.mark.parametrize("seed", range(3))
def test_reference_sources_paged_mode(ref_trace_bin, seed):
    rng = np.random.RandomState(100 + seed)
    sc = {"model": {"cache_mode": 1, "page_size": 4, "vocab_size": 500},
          "generator": {"max_running_batch": 8, "max_tokens_per_step": 64, "max_prefill_batch": 4, "enable_penalty": True},
          "kv_cache_max_tokens": 256, "requests": rand_requests(rng, 24, 500, 18, 9)}
    compare_with_reference(ref_trace_bin, sc)


@pytest.mark.parametrize("seed", range(3))
def test_reference_sources_prefix_cache(ref_trace_bin, seed):
    rng = np.random.RandomState(200 + seed)
    shared = rng.randint(3, 300, size=24).tolist()
    reqs = rand_requests(rng, 20, 300, 10, 6, shared_prefix=shared)
    reqs.append({"id": 20, "tokens": shared[:16], "generation_length": 3})
    reqs.append({"id": 21, "tokens": shared[:16], "generation_length": 2})
    sc = {"model": {"cache_mode": 1, "page_size": 4, "vocab_size": 300},
          "generator": {"max_running_batch": 6, "max_tokens_per_step": 128, "enable_prefix_cache": True},
          "kv_cache_max_tokens": 120, "requests": reqs}
    steps = compare_with_reference(ref_trace_bin, sc)
    assert any(s["prefix_hit"] for s in steps)


def test_reference_sources_execute_failure(ref_trace_bin):
    rng = np.random.RandomState(7)
    sc = {"model": {"cache_mode": 1, "page_size": 8, "vocab_size": 600}, "generator": {"max_running_batch": 8},
          "kv_cache_max_tokens": 512, "requests": rand_requests(rng, 6, 600, 12, 10), "fail_at_run": 3}
    for r in sc["requests"]:
        r["generation_length"] = 8
    compare_with_reference(ref_trace_bin, sc)


def test_reference_sources_cancel_differs_only_in_the_documented_flag(ref_trace_bin):
    """a cancel on a quiet step: same packing, same tokens -- but the reference does NOT re-upload the page table on the next
    step (the hole the repo closes on purpose, llm_generator.cc DeleteTasks): that one flag is the whole difference."""
    sc = {"model": {"cache_mode": 1, "page_size": 4, "vocab_size": 700}, "generator": {"max_running_batch": 8},
          "kv_cache_max_tokens": 256,
          "requests": [{"id": i, "tokens": [10 + i, 20 + i, 30 + i, 40 + i, 50 + i], "generation_length": 12, "early_stopping": False}
                       for i in range(3)],
          "cancel": [{"at_step": 3, "id": 0}]}
    steps, responses, failed = hl.simulate(sc)
    sc2 = dict(sc, expect_done=2)
    rsteps, rfinal = run_cpp(ref_trace_bin, sc2)
    assert len(rsteps) == len(steps)
    for a, b in zip(rsteps, steps):
        for k in ("decoding_batches", "token_inputs", "seq_starts", "kv_starts", "start_pos"):
            assert a[k] == b[k], (b["step"], k)
    assert steps[4]["req_list_changed"] == 1 and rsteps[4]["pages_uploaded"] == 0     # the repo re-uploads, the reference does not
    assert rsteps[4]["page_list"] != steps[4]["page_list"]                              # ... and so runs rows 1, 2 on row 0's stale table


# ---------------------------------------------------------------------------------------------------------------
# 4. detokenise-and-send (reference DecodeAndSendTask, src/generator/llm_generator.cc:58-112) with a real tokenizer
# ---------------------------------------------------------------------------------------------------------------
def test_cpp_text_request_streams_pieces_and_buffers_split_characters(trace_bin):
    """a TEXT request through the C++ generator with the SentencePiece tokenizer of src/tokenizer (LLaMA-style BPE model with
    byte fallback, tests/golden/spm_bpe.model): the prompt is tokenised with BOS in front; every step's token is detokenised on
    its own; a piece that decodes to U+FFFD (one byte of a multi-byte character) is held back, and once three such tokens are
    buffered they are decoded together (llm_generator.cc:84-99) -- here the three byte pieces of U+4E16."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "spm_cases.json")))["spm_bpe.model"]
    by_text = {t["text"]: t for t in cases["texts"]}
    prompt = by_text["Hello, my name is"]
    seq = {tuple(s["ids"]): s["decoded"] for s in cases["id_sequences"]}
    e4b896 = next(list(k) for k, v in seq.items() if v == "世" and len(k) == 3)
    word = by_text["a b c d e f g"]["ids"][:2]           # two ordinary pieces after the character
    emit = e4b896 + word
    chain, last = [], prompt["ids"][-1]
    for t in emit:
        chain.append({"from": last, "to": t})
        last = t
    sc = {"model": {"cache_mode": 0, "vocab_size": cases["vocab_size"]}, "generator": {"max_running_batch": 4},
          "kv_cache_max_tokens": 256, "tokenizer": os.path.join(ROOT, "tests", "golden", "spm_bpe.model"), "chain": chain,
          "requests": [{"id": 0, "prompt": "Hello, my name is", "generation_length": len(emit), "early_stopping": False}]}
    csteps, final = run_cpp(trace_bin, sc)
    assert csteps[0]["token_inputs"] == [cases["bos"]] + prompt["ids"]                 # LlamaTokenizer: BOS first
    assert final["responses"]["0"]["tokens"] == emit
    texts = [bytes.fromhex(h).decode("utf-8") for h in final["texts_hex"]["0"]]
    pieces = by_text["a b c d e f g"]
    want_tail = [(" " + d) if (p.startswith("▁") and d and d[0] != " ") else d for p, d in zip(pieces["pieces"][:2], pieces["per_token"][:2])]
    assert texts == ["", "", "世"] + want_tail


# ---------------------------------------------------------------------------------------------------------------
# 5. Generate() re-entry and a fixed sample of the differential fuzz (oracle/sched_fuzz.py; 1000 scenarios agreed three ways)
# ---------------------------------------------------------------------------------------------------------------
def test_idle_generator_is_reentered_for_the_stashed_request(trace_bin):
    """llm_generator.cc:628 charges the PREVIOUS step's batch size (finished rows included) against max_tokens_per_step, so a
    prompt that exactly fills the budget is refused right after a step that finished everything; Generate() then returns
    (:657-660) and GeneratorThreadFunc (:342-366) enters it again with fresh locals, where the request is admitted."""
    sc = {"model": {"cache_mode": 0, "vocab_size": 997}, "generator": {"max_running_batch": 4, "max_tokens_per_step": 12, "max_prefill_batch": 2},
          "kv_cache_max_tokens": 64,
          "requests": [{"id": 0, "tokens": list(range(10, 20)), "generation_length": 1}, {"id": 1, "tokens": [30, 31], "generation_length": 1},
                       {"id": 2, "tokens": list(range(40, 52)), "generation_length": 1}]}
    steps, responses, failed = compare(trace_bin, sc)
    assert [s["step"] for s in steps] == [0, 0] and [len(s["token_inputs"]) for s in steps] == [12, 12]
    assert sorted(responses) == [0, 1, 2] and not failed


def test_scheduler_fuzz_sample(trace_bin):
    from oracle.sched_fuzz import scenario
    rng = np.random.RandomState(5)
    for _ in range(40):
        compare(trace_bin, scenario(rng))


def test_reference_sources_scheduler_fuzz_sample(ref_trace_bin):
    from oracle.sched_fuzz import scenario
    rng = np.random.RandomState(5)
    for _ in range(40):
        compare_with_reference(ref_trace_bin, scenario(rng))
