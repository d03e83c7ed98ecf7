#!/usr/bin/env python3
"""Wire-level load generator for the gRPC service, the counterpart of the reference's
tools/client_qps_measure_token_in_out.cc (:51-360): one Generation call per request (batch size 1, :64), token ids in and
out, `ignore_eos_token = !early_stopping`, exponential inter-arrival times at --request-rate (Poisson arrivals, :117-125;
"inf" sends everything at once), and the same figures of merit: output tokens/s, (input+output) tokens/s, requests/s,
per-request latency percentiles -- plus the time to first token (first streamed Response - submit) percentiles that
BASELINE.json's metric names.

Requests: `--dataset file.json` ([{"input_ids": [...], "max_new_tokens": n}, ...]) or the synthetic samples_1024-shaped
workload of SURVEY.md 8(d) D2 (log-normal prompt / answer lengths, seed 1234), the same one `offline_inference --workload
samples1024` runs in process.

    python client_qps_measure_token_in_out.py --target 127.0.0.1:10086 --num-requests 1024 --request-rate inf
"""
import argparse
import asyncio
import json
import os
import sys
import time

import grpc
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import llm_proto as P  # noqa: E402


def synthetic_requests(n, vocab, max_seq_len, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        plen = int(np.clip(rng.lognormal(np.log(30.0), 1.577), 4, 1024))
        olen = int(np.clip(rng.lognormal(np.log(286.0), 0.44), 4, 1024))
        if plen + olen > max_seq_len:
            olen = max(4, max_seq_len - plen) if plen < max_seq_len - 4 else 4
            plen = min(plen, max_seq_len - olen)
        out.append((rng.randint(3, vocab, size=plen).tolist(), olen))
    return out


async def one_request(stub, rid, tokens, gen_len, early_stopping, rec):
    req = P.BatchedRequest()
    r = req.req.add()
    r.id = rid
    r.tokens.ids.extend(tokens)
    r.choosing_parameters.do_sample = False
    r.stopping_parameters.max_new_tokens = gen_len
    r.stopping_parameters.ignore_eos_token = not early_stopping
    t0 = time.perf_counter()
    first = last = None
    n_out = 0
    status = P.PROCESSING
    async for batch in stub(req):
        now = time.perf_counter()
        for rsp in batch.rsp:
            status = rsp.status
            if rsp.status == P.FAILED:
                continue
            if first is None:
                first = now
            last = now
            n_out += len(rsp.tokens.ids) if len(rsp.tokens.ids) else 1
    rec[rid] = dict(submit=t0, first=first, last=last, n_in=len(tokens), n_out=n_out, failed=status == P.FAILED or first is None)


async def run(a):
    if a.dataset:
        data = json.load(open(a.dataset))
        reqs = [(d["input_ids"], int(d.get("max_new_tokens", 64))) for d in data][:a.num_requests]
    else:
        reqs = synthetic_requests(a.num_requests, a.vocab_size, a.max_seq_len, a.seed)
    opts = [("grpc.max_receive_message_length", 64 << 20), ("grpc.max_send_message_length", 64 << 20)]
    async with grpc.aio.insecure_channel(a.target, options=opts) as ch:
        stub = ch.unary_stream(P.METHOD, request_serializer=P.BatchedRequest.SerializeToString,
                               response_deserializer=P.BatchedResponse.FromString)
        rec = {}
        rng = np.random.RandomState(a.seed + 1)
        rate = float("inf") if a.request_rate == "inf" else float(a.request_rate)
        tasks = []
        t_begin = time.perf_counter()
        for i, (tok, gl) in enumerate(reqs):
            tasks.append(asyncio.create_task(one_request(stub, i, tok, gl, a.early_stopping, rec)))
            if rate != float("inf"):
                await asyncio.sleep(rng.exponential(1.0 / rate))
            elif i % 64 == 63:
                await asyncio.sleep(0)     # let the calls start
        await asyncio.gather(*tasks)
        t_end = time.perf_counter()
    ok = [r for r in rec.values() if not r["failed"]]
    lat = np.array([(r["last"] - r["submit"]) * 1e3 for r in ok]) if ok else np.zeros(1)
    ttft = np.array([(r["first"] - r["submit"]) * 1e3 for r in ok]) if ok else np.zeros(1)
    wall = t_end - t_begin
    n_in, n_out = sum(r["n_in"] for r in ok), sum(r["n_out"] for r in ok)
    pct = lambda v, p: float(np.percentile(v, p))
    res = {"requests": len(reqs), "failed": len(reqs) - len(ok), "request_rate": a.request_rate,
           "avg_input_len": n_in / max(len(ok), 1), "avg_output_len": n_out / max(len(ok), 1),
           "out_tps": n_out / wall, "in_out_tps": (n_in + n_out) / wall, "qps": len(ok) / wall, "total_latency_s": wall,
           "latency_ms": {"avg": float(lat.mean()), "min": float(lat.min()), "p50": pct(lat, 50), "p90": pct(lat, 90),
                          "p99": pct(lat, 99), "max": float(lat.max())},
           "ttft_ms": {"p10": pct(ttft, 10), "p50": pct(ttft, 50), "p90": pct(ttft, 90), "p99": pct(ttft, 99), "max": float(ttft.max())}}
    print(json.dumps(res))
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--target", default="127.0.0.1:10086")
    ap.add_argument("--dataset", default="")
    ap.add_argument("--num-requests", type=int, default=1024)
    ap.add_argument("--request-rate", default="inf", help='requests per second (Poisson arrivals) or "inf"')
    ap.add_argument("--early-stopping", action="store_true", help="honour EOS (default: ignore_eos_token = true, as the benchmark)")
    ap.add_argument("--vocab-size", type=int, default=32000)
    ap.add_argument("--max-seq-len", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args(argv)
    return asyncio.run(run(a))


if __name__ == "__main__":
    main()
