"""Differential fuzz of the scheduler / input packing: random scenarios through
  (a) oracle/host_logic.py (the restatement),
  (b) build/sched_trace      -- this tree's C++ LLMGenerator + LLMEngine on a recording backend,
  (c) build/ref_sched_trace  -- the REFERENCE's own llm_generator.cc / llm_engine.cc compiled in place (`make ref`; only where
                                /root/reference exists),
and compares every step's packed ModelInput, the responses and the failures.  Test infrastructure (build container); run as
    python oracle/sched_fuzz.py [n_scenarios] [seed]
tests/test_host_logic.py::test_scheduler_fuzz_sample runs a small fixed sample of it."""
import os
import signal
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def scenario(rng):
    mode = int(rng.randint(0, 2))
    page = int(rng.choice([2, 4, 8, 16])) if mode else 0
    prefix = bool(mode == 1 and rng.rand() < 0.4)
    vocab = int(rng.choice([50, 300, 997]))
    n = int(rng.randint(1, 36))
    max_prompt = int(rng.choice([3, 12, 30, 70]))
    max_gen = int(rng.choice([1, 4, 12, 25]))
    shared = rng.randint(3, vocab, size=int(rng.randint(4, 40))).tolist() if rng.rand() < 0.6 else None
    reqs = []
    for i in range(n):
        toks = rng.randint(3, vocab, size=rng.randint(1, max_prompt + 1)).tolist()
        if shared is not None and rng.rand() < 0.7:
            toks = shared[:rng.randint(1, len(shared) + 1)] + (toks if rng.rand() < 0.8 else [])
        r = {"id": i, "tokens": toks, "generation_length": int(rng.randint(1, max_gen + 1))}
        if rng.rand() < 0.2:
            r["stop_tokens"] = rng.randint(3, vocab, size=rng.randint(1, 30)).tolist()
        if rng.rand() < 0.15:
            r["early_stopping"] = False
        reqs.append(r)
    gen = {"max_running_batch": int(rng.randint(1, 11)), "max_tokens_per_step": int(rng.choice([8, 24, 48, 128, 512])),
           "max_prefill_batch": int(rng.randint(1, 7)), "max_cooldown_request": int(rng.randint(1, 5))}
    if rng.rand() < 0.5:
        gen["stop_tokens"] = rng.randint(3, vocab, size=rng.randint(1, 12)).tolist()
    if rng.rand() < 0.3:
        gen["enable_penalty"] = True
    if prefix:
        gen["enable_prefix_cache"] = True
    if rng.rand() < 0.25:
        gen["max_input_tokens_per_request"] = int(rng.randint(2, max_prompt + 10))
    if rng.rand() < 0.25:
        gen["max_output_tokens_per_request"] = int(rng.randint(1, max_gen + 3))
    if rng.rand() < 0.25:
        gen["max_total_tokens_per_request"] = int(rng.randint(4, max_prompt + max_gen + 10))
    kv = int(rng.choice([32, 64, 160, 512, 4096]))
    # a request that can never be admitted (prompt longer than the step budget, or a lifetime reservation larger than the whole
    # pool -- Q7 / Q8) waits forever in the reference and here alike: keep every request schedulable on its own
    longest = max(len(r["tokens"]) for r in reqs)
    gen["max_tokens_per_step"] = max(gen["max_tokens_per_step"], longest)
    need = max(len(r["tokens"]) + r["generation_length"] for r in reqs) + (page or 1)
    kv = max(kv, need)
    if mode:
        kv = (kv + page - 1) // page * page + page
    sc = {"model": {"cache_mode": mode, "vocab_size": vocab}, "generator": gen, "kv_cache_max_tokens": kv, "requests": reqs}
    if mode:
        sc["model"]["page_size"] = page
    # (not with the prefix cache: ReleaseResource, llm_generator.cc:368-385, drops the cache without returning the pages of
    # finished requests it still held, so a later request can wait forever -- in the reference, here and in the restatement alike)
    if rng.rand() < 0.1 and not prefix:
        sc["fail_at_run"] = int(rng.randint(0, 6))
    return sc


def _alarm(*_):
    raise TimeoutError("oracle simulate() did not terminate within 20 s")


def main():
    signal.signal(signal.SIGALRM, _alarm)
    from tests import test_host_logic as T
    from oracle import host_logic as hl
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    trace_bin = os.path.join(T.PKG, "build", "sched_trace")
    ref_bin = T.REF_TRACE if os.path.exists(T.REF_TRACE) else None
    rng = np.random.RandomState(seed)
    bad = 0
    for i in range(n):
        sc = scenario(rng)
        signal.alarm(20)             # a scenario that never terminates is a finding, not a hang of the fuzz
        try:
            hl.simulate(sc)
        except Exception as e:      # a scenario the restatement itself rejects is a finding too
            print(f"[{i}] oracle raised {type(e).__name__}: {e}")
            import json
            json.dump(sc, open(f"/tmp/sched_fuzz_{seed}_{i}_oracle.json", "w"))
            bad += 1
            continue
        finally:
            signal.alarm(0)
        for name, fn in (("cpp", lambda: T.compare(trace_bin, sc)), ("reference", (lambda: T.compare_with_reference(ref_bin, sc)) if ref_bin else None)):
            if fn is None:
                continue
            try:
                fn()
            except Exception as e:
                bad += 1
                print(f"[{i}] {name}: {type(e).__name__}: {str(e)[:300]}")
                import json
                json.dump(sc, open(f"/tmp/sched_fuzz_{seed}_{i}_{name}.json", "w"))
    print(f"scenarios {n} seed {seed} reference {'yes' if ref_bin else 'no'} mismatches {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
