#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s of the batched decode hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[1] -- LLaMA-2-7B, W8A16 weights, int8 group-8 KV cache
(cache_layout 3, cache_mode 0), tensor parallel over the N GPUs, running batch 1024, every request at context
length --kv-len (512 = the mean of a 1024-token sequence) when the timed region starts; synthetic weights and
synthetic KV history generated on the device (no checkpoints/datasets are available offline).  One "step" is one
pass of the hot path over the batch: pplhip_set_inputs -> pplhip_run (32 layers) -> pplhip_sample (greedy), i.e.
LLMEngine::Execute (reference src/engine/llm_engine.cc:171-236); each step emits 1024 tokens.
value = 1024 * K / (max over ranks of the wall time of the K timed steps).

Extra objects on the JSON line:
  roofline     -- the dominant kernel (decode attention): algorithmic KV bytes / HIP-event duration vs 8 TB/s
  cpu_baseline -- the CPU restatement (oracle/, "port") timed on the host cores, on a bounded sample
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def load_pplhip():
    spec = importlib.util.spec_from_file_location("pplhip_binding", os.path.join(ROOT, "ppl.llm.serving_amd", "pplhip.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pplhip_binding"] = mod
    spec.loader.exec_module(mod)
    return mod


MODELS = {
    "llama2-7b": dict(hidden_dim=4096, intermediate_dim=11008, num_layers=32, num_heads=32, num_kv_heads=32, vocab_size=32000),
    "llama2-13b": dict(hidden_dim=5120, intermediate_dim=13824, num_layers=40, num_heads=40, num_kv_heads=40, vocab_size=32000),
    "llama2-70b": dict(hidden_dim=8192, intermediate_dim=28672, num_layers=80, num_heads=64, num_kv_heads=8, vocab_size=32000),
}


def attn_bytes_per_launch(B, kv_lens_sum, H, Hkv, D, kv_quant):
    """per layer  sum_b kv_len_b * 2 * Hkv * (D*e + (D/8)*2 [int8])  +  B*H*D*2*2 (q read, o write)."""
    per_tok = 2 * Hkv * (D * (1 if kv_quant else 2) + ((D // 8) * 2 if kv_quant else 0))
    return kv_lens_sum * per_tok + B * H * D * 2 * 2


def cpu_decode_sample(model_kw, kv_len, B, wq, kvq, max_steps, budget_s):
    """decode steps of the oracle (oracle/llama_ref.c, OpenMP) on the host cores: (tokens/s, steps, setup seconds, threads)"""
    from oracle import ref  # sets OMP_NUM_THREADS to the CPUs this container may really use (cgroup quota)
    desc = ref.make_desc(max_position=2048, cache_quant_bit=kvq, cache_quant_group=8 if kvq else 1, cache_layout=3, cache_mode=0,
                         weight_quant_bit=wq, **model_kw)
    t0 = time.time()
    m = ref.RefModel(desc)
    m.init_synthetic(1234)
    tokens = B * (kv_len + 8)
    m.kv_alloc(tokens)
    kc, ks = m.kv_array(0), m.kv_array(1)
    rng = np.random.RandomState(0)
    if kvq:
        kc[:] = rng.randint(-127, 128, size=kc.size, dtype=np.int8)
        ks[:] = (0.01 + 0.02 * rng.rand(ks.size)).astype(np.float16)
    else:
        kc[:] = (rng.standard_normal(kc.size).astype(np.float32) * 0.5).astype(np.float16)
    setup_s = time.time() - t0
    cache_idx = (np.arange(B) * (kv_len + 8)).astype(np.int64)
    tok = rng.randint(3, desc.vocab_size, size=B).astype(np.int64)
    steps, t_total = 0, 0.0
    while steps < max_steps and (steps == 0 or t_total + t_total / steps < budget_s):
        st = ref.make_step(tok, np.arange(B + 1), np.full(B, kv_len + steps), cache_idx, B)
        t1 = time.time()
        logits = ref.forward([m], st)
        tok = logits.argmax(-1).astype(np.int64)
        t_total += time.time() - t1
        steps += 1
    cores = ref.lib().ref_num_threads()
    m.close()
    return B * steps / t_total, steps, setup_s, cores


def cpu_baseline(model_kw, kv_len):
    """The reference has no CPU path; the baseline is this build's CPU restatement ("port"), timed on the
    host cores on a bounded sample.  value = BASELINE config 1 as SURVEY.md D5 states it (7B fp16 weights, fp16 KV, batch 1,
    greedy decode); the W8A16 / int8-KV batch-8 figure of the benchmark configuration is reported beside it."""
    v1, s1, set1, cores = cpu_decode_sample(model_kw, kv_len, B=1, wq=0, kvq=0, max_steps=8, budget_s=12.0)
    v8, s8, set8, _ = cpu_decode_sample(model_kw, kv_len, B=8, wq=8, kvq=8, max_steps=4, budget_s=10.0)
    return {"value": round(v1, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"BASELINE config 1: {s1} greedy decode steps of batch 1 at kv_len {kv_len}, LLaMA-2-7B fp16 weights / fp16 KV "
                      f"(oracle/llama_ref.c, OpenMP; setup {set1:.1f}s not timed)",
            "batch8_w8a16_int8kv": {"value": round(v8, 3), "unit": "tokens/s",
                                    "sample": f"{s8} decode steps of batch 8 at kv_len {kv_len}, W8A16 / int8-g8 KV (setup {set8:.1f}s not timed)"}}


def ragged_kv_lengths(B, seed=1234, cap=1024):
    """running-batch context lengths of a samples_1024.json-shaped load in steady state (prompt log-normal
    median 30 / mean 104, answer median 286 / mean 315, both clipped to [4, 1024], prompt + answer <= cap): every request
    is somewhere between its first and its last generated token."""
    rng = np.random.RandomState(seed)
    p = np.clip(rng.lognormal(np.log(30.0), 1.577, size=B), 4, 1024).astype(np.int64)
    o = np.clip(rng.lognormal(np.log(286.0), 0.44, size=B), 4, 1024).astype(np.int64)
    p = np.minimum(p, cap - 4)
    o = np.minimum(o, cap - p)
    return p + (rng.rand(B) * o).astype(np.int64)


def i8i8_leg(args):
    """the same decode steps in the reference's OTHER int8 mode, --quant-method online_i8i8 (W8A8: src/backends/cuda/resource_manager.cc:51-52)
    -- a secondary number beside the W8A16 headline BASELINE.json's configs name; its own process, after this one released the GPU"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--act-quant", "8", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--kv-len", str(args.kv_len), "--kv-quant", str(args.kv_quant), "--cache-mode", str(args.cache_mode), "--no-cpu-baseline",
           "--no-serving-leg", "--no-i8i8-leg", "--ragged-steps", "0", "--breakdown"]
    try:
        out = subprocess.run(cmd, capture_output=True, timeout=600)
        lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            return {"error": f"rc {out.returncode}: {out.stderr.decode()[-300:]}"}
        r = json.loads(lines[-1])
        return {"what": "same workload, int8 activations x int8 weights (per-token / per-output-row scales), int32 accumulate",
                "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                "gemm_and_quantisers_ms_per_step": r["breakdown_ms_per_step"]["gemm"],
                "attn_decode_ms_per_step": r["breakdown_ms_per_step"]["attn_decode"], "prefill_step_ms": r.get("prefill_step_ms")}
    except Exception as e:
        return {"error": repr(e)}


def serving_leg(model_kw, args):
    """BASELINE's metric is decode tokens/s + p50 TTFT: the samples_1024-shaped token-in/out load through the C++ generator
    + engine + hip backend (tools/offline_inference --workload samples1024, the in-process counterpart of the reference's
    client_qps_measure run): all 1024 requests submitted at once, TTFT = first response - submit per request
    (tools/client_qps_measure.cc:285-287), p50 / p90 over requests (:337-340), tokens out per second (:331)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "ppl.llm.serving_amd", "build", "offline_inference")
    if not os.path.exists(exe):
        return {"error": "build/offline_inference missing (run __graft_entry__.build())"}
    paced = None
    with tempfile.TemporaryDirectory() as td:
        params = dict(num_heads=model_kw["num_heads"], num_kv_heads=model_kw["num_kv_heads"], num_layers=model_kw["num_layers"],
                      hidden_dim=model_kw["hidden_dim"], intermediate_dim=model_kw["intermediate_dim"], vocab_size=model_kw["vocab_size"],
                      cache_quant_bit=args.kv_quant, cache_quant_group=8 if args.kv_quant else 1, cache_layout=3, cache_mode=0,
                      dynamic_batching=True, auto_causal=True, weight_quant_bit=args.weight_quant, weight_quant_group=128,
                      max_position=2048)
        path = os.path.join(td, "params.json")
        json.dump(params, open(path, "w"))
        cmd = [exe, "--model-param-path", path, "--synthetic-weights", "--workload", "samples1024", "--num-requests", "1024",
               "--max-seq-len", "1024", "--max-running-batch", str(args.batch), "--max-tokens-per-step", "8192"]
        if args.act_quant == 8:
            cmd += ["--quant-method", "online_i8i8"]
        t0 = time.time()
        out = subprocess.run(cmd, capture_output=True, timeout=900)
        wall = time.time() - t0
        # the same load at a finite arrival rate (client_qps_measure's --request_rate): what a request sees when the server is
        # not handed its whole day's work at once -- a short second run, 256 requests
        if out.returncode == 0 and not args.no_paced_leg:
            cmd2 = [exe, "--model-param-path", path, "--synthetic-weights", "--workload", "samples1024", "--num-requests", "256",
                    "--request-rate", "64", "--max-seq-len", "1024", "--max-running-batch", str(args.batch), "--max-tokens-per-step", "8192"]
            if args.act_quant == 8:
                cmd2 += ["--quant-method", "online_i8i8"]
            out2 = subprocess.run(cmd2, capture_output=True, timeout=900)
            l2 = [l for l in out2.stdout.decode().splitlines() if l.startswith("{")]
            if out2.returncode == 0 and l2:
                r2 = json.loads(l2[-1])
                paced = {"workload": "256 requests, Poisson arrivals at 64 requests/s", "failed": r2["failed"], "ttft_p50_ms": r2["ttft_ms"]["p50"],
                         "ttft_p90_ms": r2["ttft_ms"]["p90"], "ttft_p99_ms": r2["ttft_ms"]["p99"],
                         "decode_ms_per_token_p50": r2["decode_ms_per_token"]["p50"], "tokens_out_per_s": r2["tokens_out_per_s"]}
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": f"offline_inference rc {out.returncode}: {out.stderr.decode()[-300:]}"}
    r = json.loads(lines[-1])
    return {"workload": r["workload"] + ", submitted at once, max-running-batch %d" % args.batch, "requests": r["requests"], "failed": r["failed"],
            "tokens_out_per_s": r["tokens_out_per_s"], "ttft_p50_ms": r["ttft_ms"]["p50"], "ttft_p90_ms": r["ttft_ms"]["p90"],
            "ttft_p99_ms": r["ttft_ms"]["p99"], "decode_ms_per_token_p50": r["decode_ms_per_token"]["p50"], "steps": r["steps"],
            "max_running": r["max_running"], "process_wall_s": round(wall, 1), "paced": paced}


def dry_run(args, P, dist, rank, world, guard=None):
    """the N>1 control plane without a GPU: rendezvous, unique-id broadcast, barrier-bracketed timing, MAX over ranks,
    one JSON line from rank 0.  (tests/test_tp_gloo.py)"""
    uid = None
    if rank == 0:
        try:
            uid = P.get_unique_id()
        except Exception:
            uid = os.urandom(P.UNIQUE_ID_BYTES)
    agreed = True
    if dist is not None:
        if guard is not None:
            guard.at("RCCL unique-id broadcast")
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
        if args.dry_run_fail_rank == rank:   # test hook: a rank that dies before the handle exchange (tests/test_tp_gloo.py)
            os._exit(17)
        if guard is not None:
            guard.at("IPC-handle exchange")
        allv = [None] * world
        dist.all_gather_object(allv, uid)
        agreed = all(v == allv[0] for v in allv) and len(uid) == P.UNIQUE_ID_BYTES
        dist.barrier()
    if guard is not None:
        guard.disarm()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    own = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [own / max(args.steps, 1) * 1e3]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        tmin = torch.tensor([own], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        rank_ms = [float(tmin.item()) / max(args.steps, 1) * 1e3, float(t.item()) / max(args.steps, 1) * 1e3]
        elapsed = float(t.item())
    if rank == 0:
        emit({"metric": "decode tokens/sec, LLaMA-7B int8 (W8A16), max-running-batch 1024", "value": 0.0,
                          "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dry_run": True, "unique_id_agreed": agreed,
                          "ms_per_step_ranks": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4)},
                          "schedule": "none (dry run)",
                          "collectives": {"mode": "none (dry run)", "selftest": "not run", "schedule": "none (dry run)", "rccl_communicator": False,
                                          "two_stream_rows": [0, 0], "fallbacks": "", "allreduce_us": {"rows": 0, "bytes": 0, "chosen_path": None, "rccl": None}},
                          "config": {"workload": "dry run (no device work)", "parallelism": f"tp{world}"}})
    if dist is not None:
        dist.destroy_process_group()


_JSON_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries (the RCCL version banner, for one) write to file
    descriptor 1 through their own stdio buffers, which are flushed at exit -- after our line.  Keep a private copy of
    the real stdout for the JSON line and point fd 1 at stderr for everything else."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    _JSON_OUT.write(json.dumps(obj) + "\n")
    _JSON_OUT.flush()



class StartupGuard:
    """Wall-clock guard around a multi-process start-up (rendezvous, library init, IPC-handle exchange, collective self-test, warm-up).
    A rank that never reaches the rendezvous, a peer that died, an RCCL bootstrap that waits for ever: without this the driver's
    `bench.py --gpus N` hangs until ITS limit and leaves nothing to read.  With it rank 0 prints the contract's ONE JSON line carrying
    "error" (what was being waited for, after how long) and whatever the collectives had reported so far ("collectives.fallbacks"),
    and every rank exits non-zero.  Fired by a timer thread (the main thread may sit in a blocking C call), by SIGTERM (the launcher
    tearing the group down because a peer failed) and by an exception on the way.  Disarmed when the timed region starts."""

    def __init__(self, args, rank, world, timeout_s):
        import threading
        self.args, self.rank, self.world, self.timeout_s = args, rank, world, timeout_s
        self.phase, self.t0, self.info, self.done = "start", time.perf_counter(), None, False
        self.lock = threading.Lock()
        self.timer = None
        if world > 1 and timeout_s > 0:
            self.timer = threading.Timer(timeout_s, self.fire, args=(f"start-up guard: no progress past '{{phase}}' within {timeout_s:.0f} s",))
            self.timer.daemon = True
            self.timer.start()
            import signal
            signal.signal(signal.SIGTERM, lambda *_: self.fire("terminated by the launcher during '{phase}' (a peer rank failed?)", code=143))

    def at(self, phase):
        self.phase = phase

    def disarm(self):
        self.done = True
        if self.timer is not None:
            self.timer.cancel()

    def fire(self, why, code=3):
        with self.lock:
            if self.done:
                return
            self.done = True
        msg = why.replace("{phase}", self.phase)
        print(f"[bench] rank {self.rank}: {msg} (after {time.perf_counter() - self.t0:.1f} s)", file=sys.stderr, flush=True)
        if self.rank == 0:
            co = self.info if isinstance(self.info, dict) else {"mode": "unknown (start-up did not get that far)", "fallbacks": ""}
            try:
                emit({"metric": "decode tokens/sec, LLaMA-7B int8 (W8A16), max-running-batch 1024", "value": 0.0, "unit": "tokens/s",
                      "n_gpus": self.world, "steps": self.args.steps, "warmup": self.args.warmup, "ms_per_step": None,
                      "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "error": msg, "phase": self.phase,
                      "elapsed_s": round(time.perf_counter() - self.t0, 1), "collectives": co,
                      "config": {"workload": "not run (start-up failed)", "parallelism": f"tp{self.world}"}})
            except Exception:
                pass
        os._exit(code)


def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: become `python -m torch.distributed.run ... bench.py --gpus N ...`
    (one process per GPU, the same command line the driver uses).  stdout stays the rank-0 JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="llama2-7b", choices=sorted(MODELS))
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--kv-len", type=int, default=512)
    ap.add_argument("--weight-quant", type=int, default=8)
    ap.add_argument("--kv-quant", type=int, default=8)
    ap.add_argument("--act-quant", type=int, default=0, choices=[0, 8],
                    help="8 = the reference's --quant-method online_i8i8 (W8A8); NOT the configuration of the headline metric")
    ap.add_argument("--layers", type=int, default=0, help="debug only: override the layer count (result is then INVALID)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serving-leg", action="store_true", help="skip the samples_1024-shaped serving run (TTFT)")
    ap.add_argument("--no-paced-leg", action="store_true", help="serving leg: skip the second run at 64 requests/s")
    ap.add_argument("--breakdown", action="store_true",
                    help="also time the GEMM launches (events around every kernel class: costs the step ~3 %%; default: decode attention only)")
    ap.add_argument("--breakdown-steps", type=int, default=4,
                    help="extra, untimed decode steps with every kernel class bracketed, for breakdown_ms_per_step.gemm (0: skip)")
    ap.add_argument("--no-i8i8-leg", action="store_true",
                    help="skip the secondary run of the same decode step in the reference's other int8 mode (--quant-method online_i8i8)")
    ap.add_argument("--ragged-steps", type=int, default=4, help="decode steps at a samples_1024-shaped ragged kv_len batch (0: skip)")
    ap.add_argument("--tpb", type=int, default=0)
    ap.add_argument("--prefill-sample", type=int, default=1, help="also time one 8192-token prefill step (TTFT proxy)")
    ap.add_argument("--cache-mode", type=int, default=0, choices=[0, 1],
                    help="0: contiguous KV ranges (default, as the metric is quoted), 1: paged KV with shuffled 16-token pages")
    ap.add_argument("--emulate-tp", type=int, default=0,
                    help="profiling only: run ONE rank's slice of a tp-way step on one GPU (collectives are local "
                         "identities, logits incomplete); the printed line is marked invalid as a throughput number")
    ap.add_argument("--startup-timeout", type=float, default=float(os.environ.get("PPLHIP_BENCH_STARTUP_TIMEOUT", "900")),
                    help="N > 1: seconds the start-up (rendezvous .. end of warm-up) may take before rank 0 prints the JSON line with \"error\" and every rank exits non-zero (0: no guard)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="test hook (--dry-run): this rank exits before the handle exchange")
    ap.add_argument("--dry-run", action="store_true",
                    help="no device work: exercises only the multi-process control plane (CPU test of the N>1 path)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args.gpus)  # does not return
    claim_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    guard = StartupGuard(args, rank, world, args.startup_timeout)
    try:
        return run(args, guard, world, rank, local_rank)
    except SystemExit:
        raise
    except BaseException as e:   # (a peer that vanished shows up here as a gloo / library error on the survivors)
        if world > 1:
            guard.fire(f"{type(e).__name__} during '{{phase}}': {str(e)[:300]}", code=1)
        raise


def run(args, guard, world, rank, local_rank):
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one node by contract: keep both control planes (gloo here, the RCCL bootstrap inside libpplhip) on the loopback
        # interface so that neither depends on the container's hostname resolving or on an external NIC
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL P2P setup)
        import datetime
        guard.at("rendezvous (torch.distributed, gloo)")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=max(30.0, args.startup_timeout or 1800.0)))

    guard.at("loading libpplhip")
    P = load_pplhip()
    if args.dry_run:
        return dry_run(args, P, dist, rank, world, guard)
    mk = dict(MODELS[args.model])
    if args.layers:
        mk["num_layers"] = args.layers
    B, K, W = args.batch, args.steps, args.warmup
    total_len = args.kv_len + K + W + 2 + (0 if args.breakdown else args.breakdown_steps)
    desc = P.make_desc(max_position=max(2048, total_len + 1), cache_quant_bit=args.kv_quant,
                       cache_quant_group=8 if args.kv_quant else 1, cache_layout=3, cache_mode=args.cache_mode,
                       page_size=16 if args.cache_mode else 0,
                       weight_quant_bit=args.weight_quant, act_quant_bit=args.act_quant, **mk)
    # ---- bring-up (context, collectives, weights, KV slab), warm-up, and -- N > 1 only -- ONE retry on RCCL when the warm-up steps fail on the
    # direct collectives although their self-test passed (first contact with real links: a timed-out spin, a device error).  Every rank
    # takes the same decision (gloo MIN over "my warm-up ran"); the retry is reported in collectives.fallbacks.  Failures BEFORE the
    # warm-up (rendezvous, init, handle exchange, self-test) are the start-up guard's business, not retried.
    ctx = collectives = comm_mode = uid = None
    tp = args.emulate_tp if args.emulate_tp > 1 else world
    kv_tokens = rag_kv = None
    H = Hkv = D = 0

    def bring_up():
        nonlocal ctx, collectives, comm_mode, uid, kv_tokens, rag_kv, H, Hkv, D
        uid = None
        p2p_only = os.environ.get("PPLHIP_COMM") == "p2p"
        guard.at("RCCL unique-id broadcast")
        if world > 1 and not p2p_only:
            box = [P.get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        elif args.emulate_tp > 1:
            os.environ["PPLHIP_EMULATE_TP"] = "1"
        elif os.environ.get("PPLHIP_FORCE_COMM"):
            uid = P.get_unique_id()  # single-GPU self-test of the RCCL path (world size 1: every collective is an identity)
        # PPLHIP_BENCH_ONE_DEVICE=1 (tests): every rank on device 0 -- the multi-process plumbing (IPC handles, direct
        # collectives) on a one-GPU box; needs PPLHIP_COMM=p2p because RCCL refuses two ranks on one device
        dev = 0 if os.environ.get("PPLHIP_BENCH_ONE_DEVICE") else local_rank
        guard.at("pplhip_init (streams, RCCL communicator, buffers)")
        ctx = P.Context(desc, max_running_batch=B, max_tokens_per_step=max(8192, B), n_local_ranks=1, world_size=tp,
                        rank_base=rank, device_ids=[dev], unique_id=uid, profiling=1 if args.breakdown else 2, tpb=args.tpb)
        if world > 1 and os.environ.get("PPLHIP_COMM") != "rccl":
            # one process per GPU: exchange the IPC handles of the exchange regions, then the collective self-test decides
            # between the direct kernels and RCCL (the same decision on every rank)
            handles = [None] * world
            guard.at("IPC-handle exchange")
            try:
                mine = ctx.comm_export(0)
            except RuntimeError as e:   # no IPC handle for the exchange region on this rank: every rank then stays on RCCL
                print(f"[bench] rank {rank}: {e}", file=sys.stderr)
                mine = None
            dist.all_gather_object(handles, mine)
            if all(h is not None for h in handles):
                guard.at("pplhip_comm_connect (peer mapping + collective self-test)")
                ctx.comm_connect(handles)
        comm_mode = {0: "none", 1: "rccl", 2: "direct xGMI kernels (two-shot, all links)"}[ctx.comm_mode()]
        # what this run's collectives are and cost (outside the timed region): mode, self-test verdict, schedule of the timed step, fallbacks
        # taken, and a timed micro-loop of the step's own all-reduce message on the chosen path AND on RCCL -- so that a bad scaling curve can
        # be read from the JSON line alone
        collectives = ctx.comm_info(B)
        guard.info = collectives
        guard.at("timed all-reduce micro-loop")
        if world > 1 or os.environ.get("PPLHIP_FORCE_COMM"):
            try:
                collectives["allreduce_us"] = {"rows": B, "bytes": B * desc.hidden_dim * 2, "chosen_path": ctx.comm_allreduce_us(B, 20, 0),
                                               "rccl": ctx.comm_allreduce_us(B, 20, 1)}
            except Exception as e:   # reporting only
                collectives["allreduce_us"] = {"error": repr(e)}
        guard.at("weights and KV slab")
        ctx.init_synthetic(0, 1234)
        kv_tokens = B * total_len if args.cache_mode == 0 else B * ((total_len + 15) // 16) * 16
        rag_kv = ragged_kv_lengths(B) if (args.ragged_steps > 0 and args.cache_mode == 0) else None
        if rag_kv is not None:  # the ragged leg re-plans the same slab: request b owns kv_b + steps + 1 contiguous slots
            kv_tokens = max(kv_tokens, int((rag_kv + args.ragged_steps + 1).sum()))
        cap = ctx.kv_capacity(0.94)
        if kv_tokens > cap:
            sys.exit(f"KV slab needs {kv_tokens} tokens but only {cap} fit")
        ctx.kv_alloc(0, kv_tokens)
        ctx.kv_fill_synthetic(0, 99)
        H, Hkv, D = desc.num_heads // tp, desc.num_kv_heads // tp, desc.hidden_dim // desc.num_heads


    rng = np.random.RandomState(1234)
    cache_idx = (np.arange(B, dtype=np.int64) * total_len)
    max_pages = 0
    if args.cache_mode == 1:  # every request owns ceil(total_len / 16) pages drawn from a shuffled pool
        max_pages = (total_len + 15) // 16
        kv_need = B * max_pages * 16
        pool = np.random.RandomState(7).permutation(kv_need // 16).astype(np.int64)
        cache_idx = pool[:B * max_pages].reshape(B, max_pages)
    seq_starts = np.arange(B + 1, dtype=np.int64)
    tok = rng.randint(3, desc.vocab_size, size=B).astype(np.int64)

    def barrier():
        ctx.sync(0)
        if dist is not None:
            dist.barrier()

    def step(i, tok):
        st = P.make_step(tok, seq_starts, np.full(B, args.kv_len + i, dtype=np.int64), cache_idx, B, max_pages=max_pages,
                         req_list_changed=1 if i == 0 else 0)
        ctx.set_inputs(0, st)
        ctx.run(0)
        out, _ = ctx.sample(B, top_k=1, req_list_changed=(i == 0))
        return out.astype(np.int64)


    def agree(ok):
        if dist is None:
            return ok
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    tok0 = tok.copy()
    bench_fallback = ""
    retry_to = os.environ.get("PPLHIP_BENCH_FALLBACK_TO", "rccl")        # (tests on one device retry on "p2p": RCCL refuses two ranks per device)
    may_retry = world > 1 and not args.emulate_tp and (os.environ.get("PPLHIP_COMM", "auto") in ("auto", "") or "PPLHIP_BENCH_FALLBACK_TO" in os.environ)
    for attempt in range(2):
        bring_up()
        if bench_fallback:
            collectives["fallbacks"] = (collectives.get("fallbacks", "") + " | " if collectives.get("fallbacks") else "") + bench_fallback
            guard.info = collectives
        guard.at("warm-up steps" + (" (retry)" if attempt else ""))
        err, tok = None, tok0.copy()
        try:
            if attempt == 0 and os.environ.get("PPLHIP_BENCH_FAIL_WARMUP_ONCE") == str(rank):   # test hook: this rank's first warm-up fails
                raise RuntimeError("injected warm-up failure (PPLHIP_BENCH_FAIL_WARMUP_ONCE)")
            for i in range(W):
                tok = step(i, tok)
            ctx.sync(0)
        except Exception as e:
            err = f"{type(e).__name__}: {str(e)[:200]}"
            print(f"[bench] rank {rank}: warm-up failed on '{comm_mode}': {err}", file=sys.stderr, flush=True)
        if agree(err is None):                   # (doubles as the barrier behind the warm-up)
            break
        if attempt == 1 or not may_retry or ctx.comm_mode() != 2:
            raise RuntimeError(f"warm-up steps failed on {comm_mode}: {err or 'on another rank'}")
        bench_fallback = f"bench: warm-up steps failed on the direct collectives ({err or 'on another rank'}): every rank re-initialised on {retry_to}"
        print(f"[bench] rank {rank}: {bench_fallback}", file=sys.stderr, flush=True)
        try:
            ctx.close()
        except Exception:
            pass
        os.environ["PPLHIP_COMM"] = retry_to
    guard.disarm()   # every rank is past its first steps: from here a failure is a step failure and raises as such
    ctx.profile_reset(0)
    t0 = time.perf_counter()
    for i in range(W, W + K):
        tok = step(i, tok)
    ctx.sync(0)
    own = time.perf_counter() - t0              # this rank's own time, before the closing barrier: the spread over ranks is reported
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [own / max(K, 1) * 1e3]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        tmin = torch.tensor([own], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        rank_ms = [float(tmin.item()) / max(K, 1) * 1e3, float(t.item()) / max(K, 1) * 1e3]
        elapsed = float(t.item())

    n_attn, ms_attn = ctx.profile_get(P.PROF_ATTN_DECODE)
    n_gemm, ms_gemm = ctx.profile_get(P.PROF_GEMM)
    n_run, ms_run = ctx.profile_get(P.PROF_RUN)
    kv_sum = sum(B * (args.kv_len + i + 1) for i in range(W, W + K))  # keys read per layer over the timed steps
    bytes_total = attn_bytes_per_launch(B * K, kv_sum, H, Hkv, D, args.kv_quant) * desc.num_layers
    achieved = bytes_total / (ms_attn * 1e-3) / 1e9 if ms_attn > 0 else 0.0
    # HBM bytes per launch from the PMC counters: collected OFFLINE on this very script (profiles/collect_r03.sh: rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes) and stored PER KV LENGTH (every
    # step of a run launches the kernel at one kv length, 32 layers each); reported when the table covers every kv length of the
    # timed steps for the same batch / KV format / cache mode, whatever --steps / --warmup were -- never borrowed otherwise
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "attn_decode_traffic.json")
    if os.path.exists(tpath) and tp == 1:
        try:
            tj = json.load(open(tpath))
            table = tj.get("hbm_bytes_per_launch_by_kv_len", {})
            want_kv = [str(args.kv_len + i + 1) for i in range(W, W + K)]
            same = (tj.get("batch") == B and tj.get("kv_quant") == args.kv_quant and tj.get("cache_mode") == args.cache_mode
                    and tj.get("model") == args.model and all(k in table for k in want_kv))
            if same:
                traffic = int(sum(table[k] for k in want_kv) / len(want_kv))
            traffic_source = tj.get("source_short", "profiles/attn_decode_traffic.json") + ("" if same else " -- does not cover this launch shape, omitted")
        except Exception:
            traffic = None

    # GEMM share of the step: 4 more decode steps, NOT part of the timed region, with every kernel class bracketed by events
    # (an event record is a barrier packet: bracketing inside the timed region would cost it ~3 %)
    gemm_ms_per_step, gemm_note = None, None
    if args.breakdown:
        gemm_ms_per_step, gemm_note = ms_gemm / K, "events around every kernel class inside the timed region (--breakdown)"
    elif args.breakdown_steps > 0:
        ctx.profile_mode(1)
        ctx.profile_reset(0)
        for i in range(W + K, W + K + args.breakdown_steps):
            tok = step(i, tok)
        barrier()
        n_g, ms_g = ctx.profile_get(P.PROF_GEMM)
        gemm_ms_per_step = ms_g / args.breakdown_steps
        gemm_note = f"GEMM: {args.breakdown_steps} extra steps after the timed region with events around every kernel class ({n_g} launches)"
        ctx.profile_mode(2)
        ctx.profile_reset(0)

    # ragged leg: the same batch size at a samples_1024-shaped spread of context lengths (4 .. 1024 in ONE step)
    ragged = None
    if rag_kv is not None:
        r_idx = np.concatenate([[0], np.cumsum(rag_kv + args.ragged_steps + 1)[:-1]]).astype(np.int64)
        tok_r = rng.randint(3, desc.vocab_size, size=B).astype(np.int64)

        def rstep(i, tok_r):
            st = P.make_step(tok_r, seq_starts, rag_kv - 1 + i, r_idx, B, req_list_changed=1 if i == 0 else 0)
            ctx.set_inputs(0, st)
            ctx.run(0)
            out, _ = ctx.sample(B, top_k=1, req_list_changed=(i == 0))
            return out.astype(np.int64)

        tok_r = rstep(0, tok_r)   # untimed
        barrier()
        ctx.profile_reset(0)
        t1 = time.perf_counter()
        for i in range(1, args.ragged_steps + 1):
            tok_r = rstep(i, tok_r)
        barrier()
        dt_r = time.perf_counter() - t1
        n_r, ms_r = ctx.profile_get(P.PROF_ATTN_DECODE)
        kv_sum_r = sum(int((rag_kv + i).sum()) for i in range(1, args.ragged_steps + 1))
        bytes_r = attn_bytes_per_launch(B * args.ragged_steps, kv_sum_r, H, Hkv, D, args.kv_quant) * desc.num_layers
        ragged = {"kv_len": {"min": int(rag_kv.min()), "p50": int(np.median(rag_kv)), "mean": round(float(rag_kv.mean()), 1), "max": int(rag_kv.max())},
                  "steps": args.ragged_steps, "ms_per_step": round(dt_r / args.ragged_steps * 1e3, 3),
                  "tokens_per_s": round(B * args.ragged_steps / dt_r, 1),
                  "attn_decode_GBps": round(bytes_r / (ms_r * 1e-3) / 1e9, 1) if ms_r > 0 else None,
                  "attn_decode_frac_of_8TBps": round(bytes_r / (ms_r * 1e-3) / 8e12, 4) if ms_r > 0 else None,
                  "attn_avg_launch_ms": round(ms_r / max(n_r, 1), 4), "algorithmic_bytes_per_launch": int(bytes_r / max(n_r, 1))}
    extra = {}
    if args.prefill_sample:
        # TTFT proxy: one admission step of 16 x 512-token prompts (max_tokens_per_step 8192), cold cache slots
        nreq, plen = 16, 512
        if nreq * plen <= kv_tokens:
            ptok = rng.randint(3, desc.vocab_size, size=nreq * plen).astype(np.int64)
            if args.cache_mode == 0:
                pci, pmp = (np.arange(nreq, dtype=np.int64) * total_len), 0
            else:
                pmp = (plen + 15) // 16
                pci = cache_idx[:nreq, :pmp] if max_pages >= pmp else None
            st = P.make_step(ptok, np.arange(nreq + 1) * plen, np.zeros(nreq, dtype=np.int64), pci, 0, max_pages=pmp)
            for rep in range(2):
                barrier()
                t1 = time.perf_counter()
                ctx.set_inputs(0, st)
                ctx.run(0)
                ctx.sample(nreq, top_k=1)
                barrier()
                dt = time.perf_counter() - t1
            extra = {"prefill_step_ms": round(dt * 1e3, 3), "prefill_tokens_per_s": round(nreq * plen / dt, 1),
                     "prefill_shape": f"{nreq} prompts x {plen} tokens in one step"}

    if rank == 0:
        res = {
            "metric": "decode tokens/sec, LLaMA-7B int8 (W8A16), max-running-batch 1024",
            "value": round(B * K / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None,
            "ms_per_step_ranks": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4)},
            "schedule": collectives["schedule"], "collectives": collectives,
            "dtype": ("int8 activations x int8 weights (online_i8i8, W8A8), int32 accumulate" if args.act_quant == 8 else
                      "fp16 activations, int8 weights (W8A16), int8-g8 KV, fp32 accumulate"),
            "data": "synthetic (device-generated weights and KV history, random token ids)",
            "config": {"workload": f"{args.model} W{args.weight_quant or 16}A{args.act_quant or 16} decode, batch {B}, kv_len {args.kv_len}"
                                   f"..{args.kv_len + K + W}, greedy top_k=1, cache_layout 3 / cache_mode {args.cache_mode}, "
                                   f"kv int{args.kv_quant or 16}", "global_batch": B, "seq_len": 1024,
                       "parallelism": f"tp{world}", "layers": desc.num_layers, "collectives": comm_mode},
            "roofline": {"kernel": ("attn_decode_gqa_kernel" if 4 <= H // max(Hkv, 1) <= 16 else "attn_decode_kernel") + f"<{args.kv_quant},{D}>",
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "launches": n_attn, "avg_launch_ms": round(ms_attn / max(n_attn, 1), 4),
                         "algorithmic_bytes_per_launch": int(bytes_total / max(n_attn, 1))},
            "breakdown_ms_per_step": {"attn_decode": round(ms_attn / K, 3), "gemm": round(gemm_ms_per_step, 3) if gemm_ms_per_step is not None else None,
                                      "run_total_gpu": round(ms_run / K, 3), "note": gemm_note},
        }
        if gemm_ms_per_step:
            # the second kernel class of the step: W8A16 / W4A16 tile GEMMs (4 per layer + lm_head), MFMA-bound at this batch
            hd_, it_ = desc.hidden_dim, desc.intermediate_dim // tp
            per_layer = (H + 2 * Hkv) * D * hd_ + hd_ * H * D + 2 * it_ * hd_ + hd_ * it_
            flops = 2.0 * B * (desc.num_layers * per_layer + (desc.vocab_size // tp) * hd_)
            tf = flops / (gemm_ms_per_step * 1e-3) / 1e12
            res["roofline_gemm"] = {"kernel": "gemm_w8_wide_kernel (128x384x64, wqkv / w13) + gemm_dma_kernel (128x128x64, wo / w2), LDS-DMA rings", "bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0,
                                    "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4), "flops_per_step": flops,
                                    "note": "dense fp16 MFMA peak; durations from the bracketed extra steps (breakdown_ms_per_step.gemm).  On these operands (random fp16 x "
                                            "int8) the K loops sit on the chip's power limit: ~1.25 PFLOP/s in-loop for the compiler-scheduled and the hand-scheduled kernel "
                                            "alike, 1.96 PFLOP/s for the same hand-scheduled binary on zero operands; a bare MFMA stream reaches 1.73 PFLOP/s on such operands (2.49 on zeros): profiles/r04_gemm_asm_experiments.md"}
        res.update(extra)
        if ragged is not None:
            res["ragged_batch"] = ragged
        if args.layers:
            res["INVALID"] = "layer count overridden for debugging"
        if args.emulate_tp > 1:
            res["INVALID"] = f"one rank's slice of a tp{args.emulate_tp} step without its peers (profiling only)"
        ctx.close()   # the serving leg sizes its own KV slab from the free memory
        if world == 1 and not args.no_serving_leg and not args.layers and not args.emulate_tp:
            try:
                sv = serving_leg(mk, args)
            except Exception as e:
                sv = {"error": repr(e)}
            res["serving"] = sv
            res["ttft_p50_ms"] = sv.get("ttft_p50_ms")
        if (world == 1 and not args.no_i8i8_leg and not args.layers and not args.emulate_tp and args.act_quant == 0 and args.weight_quant == 8
                and args.model == "llama2-7b" and B == 1024):
            res["online_i8i8"] = i8i8_leg(args)
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(mk, args.kv_len)
            except Exception as e:  # the baseline is reporting only; never hide the GPU result
                res["cpu_baseline"] = {"error": repr(e)}
        emit(res)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
