#!/usr/bin/env python3
"""Text-mode load generator for the gRPC service, the counterpart of the reference's tools/client_qps_measure.cc (:54-96 request
sampling, :107-139 Poisson submission, :168-200 per-response bookkeeping, :275-340 figures of merit).

The dataset is the reference's conversation format (its tools/samples_1024.json): a JSON list whose entries carry
`conversations[0].value` (the prompt, sent as TEXT -- the server tokenises it) and `conversations[1].value` (the recorded answer: only its
token count is used, as `max_new_tokens`).  `--tokenizer` is the SentencePiece model the server runs; here it only counts tokens
(prompt length, expected output length), like the reference's client.  One Generation call per request (batch size 1),
`do_sample = false`, temperature 1, no penalties, `ignore_eos_token = !early_stopping`; exponential inter-arrival times at
--request_rate ("inf": everything at time 0).

Figures (the reference's [RESULT] lines, as one JSON object and as the same lines on stderr): benchmark time, request count, average /
total input length, real and expected generated tokens, time per token, average prefill latency (first streamed Response - send =
TTFT), average decoding latency per token ((finish - first) / (n - 1)), average latency per prompt, tokens out per second, tokens
in + out per second, requests per second, and the prefill / decode-step / per-prompt latency distributions at the reference's
quantiles (min, 1, 10, 25, 50, 75, 80, 90, 95, 99 %, max; indices `n * q` into the sorted list, :318-340).

    python client_qps_measure.py --target 127.0.0.1:10086 --tokenizer tokenizer.model --dataset samples_1024.json --request_rate inf
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

QUANTILES = (("min", None), ("1%", (1, 100)), ("10%", (1, 10)), ("25%", (1, 4)), ("50%", (1, 2)), ("75%", (3, 4)), ("80%", (8, 10)),
             ("90%", (9, 10)), ("95%", (95, 100)), ("99%", (99, 100)), ("max", None))  # integer index n * num / den, as the reference


def sample_requests(dataset_path, tokenizer):
    """client_qps_measure.cc:54-96: (prompt text, prompt tokens, expected output tokens) per dataset entry."""
    out = []
    for entry in json.load(open(dataset_path)):
        convs = entry["conversations"]
        prompt, ans = convs[0]["value"], convs[1]["value"]
        out.append((prompt, len(tokenizer.encode(prompt)), len(tokenizer.encode(ans))))
    return out


def distribution(values):
    """the reference's quantile rule: sorted list, element n * q (no interpolation), :318-340"""
    v = sorted(values)
    n = len(v)
    if n == 0:
        return {name: 0.0 for name, _ in QUANTILES}
    d = {}
    for name, q in QUANTILES:
        d[name] = v[0] if name == "min" else v[-1] if name == "max" else v[min(n - 1, n * q[0] // q[1])]
    return d


async def one_request(stub, rid, prompt, max_new, early_stopping, rec, step_lat):
    req = P.BatchedRequest()
    r = req.req.add()
    r.id = rid
    r.prompt = prompt
    c = r.choosing_parameters
    c.do_sample, c.temperature, c.repetition_penalty, c.presence_penalty, c.frequency_penalty = False, 1.0, 1.0, 0.0, 0.0
    r.stopping_parameters.max_new_tokens = max_new
    r.stopping_parameters.ignore_eos_token = not early_stopping
    send = time.perf_counter()
    first = prev = None
    n_out, failed, text = 0, False, []
    async for batch in stub(req):
        now = time.perf_counter()
        for rsp in batch.rsp:
            if rsp.status == P.FAILED:
                failed = True
                continue
            if first is None:
                first = prev = now                    # the first Response = the prefill's token (:176-180)
            else:
                step_lat.append((now - prev) * 1e3)   # (:181-184)
                prev = now
            text.append(rsp.generated)
            n_out += 1
    # (the reference never advances prev_time, :183, so its "decoding latency distribution" is the time since the FIRST token of the
    # request; this client advances it and reports step-to-step gaps.  The averages below follow the reference exactly.)
    rec[rid] = dict(send=send, first=first, finish=time.perf_counter(), n_out=n_out, failed=failed or first is None, text="".join(text))


async def run(a):
    import sentencepiece as spm
    tok = spm.SentencePieceProcessor()
    tok.load(a.tokenizer)
    print(f"VOCAB_SIZE: {tok.get_piece_size()}; BOS ID: {tok.bos_id()}; EOS ID: {tok.eos_id()}; PAD ID: {tok.pad_id()}", file=sys.stderr)
    reqs = sample_requests(a.dataset, tok)
    if a.num_requests > 0:
        reqs = reqs[:a.num_requests]
    rate = float("inf") if a.request_rate == "inf" else float(a.request_rate)
    rng = np.random.RandomState(a.seed)
    opts = [("grpc.max_receive_message_length", 64 << 20), ("grpc.max_send_message_length", 64 << 20)]
    rec, step_lat, tasks = {}, [], []
    async with grpc.aio.insecure_channel(a.target, options=opts) as ch:
        stub = ch.unary_stream(P.METHOD, request_serializer=P.BatchedRequest.SerializeToString,
                               response_deserializer=P.BatchedResponse.FromString)
        t_begin = time.perf_counter()
        for i, (prompt, _, exp_out) in enumerate(reqs):
            tasks.append(asyncio.create_task(one_request(stub, i, prompt, exp_out, a.early_stopping, rec, step_lat)))
            if rate != float("inf"):
                await asyncio.sleep(rng.exponential(1.0 / rate))
            elif i % 64 == 63:
                await asyncio.sleep(0)
        await asyncio.gather(*tasks)
        bench = time.perf_counter() - t_begin
    n = len(reqs)
    ok = {i: r for i, r in rec.items() if not r["failed"]}
    prefill = [(r["first"] - r["send"]) * 1e3 for r in ok.values()]
    per_prompt = [(r["finish"] - r["send"]) * 1e3 for r in ok.values()]
    dec_per_tok = [((r["finish"] - r["first"]) * 1e3 / (r["n_out"] - 1)) if r["n_out"] > 1 else 0.0 for r in ok.values()]
    tin = sum(reqs[i][1] for i in ok)
    texp = sum(reqs[i][2] for i in ok)
    tgen = sum(r["n_out"] for r in ok.values())
    m = max(len(ok), 1)
    res = {"benchmark_time_s": bench, "request_count": n, "failed": n - len(ok), "request_rate": a.request_rate,
           "avg_input_len": tin // m, "total_input_len": tin, "avg_gen_len": tgen // m, "real_total_gen_len": tgen,
           "expected_total_gen_len": texp, "time_per_token_ms": bench * 1e3 / max(tgen, 1),
           "avg_latency_prefill_ms": sum(prefill) / m, "avg_latency_decoding_ms": sum(dec_per_tok) / m,
           "avg_latency_per_prompt_ms": sum(per_prompt) / m, "tokens_out_per_sec": tgen / bench,
           "tokens_inout_per_sec": (tin + tgen) / bench, "requests_per_sec": len(ok) / bench,
           "prefill_latency_ms": distribution(prefill), "decode_step_ms": distribution(step_lat),
           "prompt_latency_ms": distribution(per_prompt)}
    e = sys.stderr
    print(f"[RESULT] benchmark time: {bench:.2f} s", file=e)
    print(f"[RESULT] request count: {n}", file=e)
    print(f"[RESULT] avg input len: {res['avg_input_len']}, total input len: {tin}", file=e)
    print(f"[RESULT] avg gen len: {res['avg_gen_len']}, real total gen len: {tgen}, expected total gen len: {texp}", file=e)
    print(f"[RESULT] time per token: {res['time_per_token_ms']:.2f} ms", file=e)
    print(f"[RESULT] avg latency prefill: {res['avg_latency_prefill_ms']:.2f} ms", file=e)
    print(f"[RESULT] avg latency decoding: {res['avg_latency_decoding_ms']:.2f} ms", file=e)
    print(f"[RESULT] avg latency per prompt: {res['avg_latency_per_prompt_ms']:.2f} ms", file=e)
    print(f"[RESULT] tokens out per sec: {res['tokens_out_per_sec']:.2f}", file=e)
    print(f"[RESULT] tokens inout per sec: {res['tokens_inout_per_sec']:.2f}", file=e)
    print(f"[RESULT] requests per sec: {res['requests_per_sec']:.2f}", file=e)
    for title, key in (("prefill latency", "prefill_latency_ms"), ("decoding latency", "decode_step_ms"), ("prompt latency", "prompt_latency_ms")):
        print(f"[RESULT] {title} distribution (ms): \n    " + ", ".join(f"{k}:[{v:.2f}]" for k, v in res[key].items()), file=e)
    if a.dump_answers:
        json.dump({str(i): r["text"] for i, r in sorted(ok.items())}, open(a.dump_answers, "w"))
    print(json.dumps(res))
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--target", default="localhost:23333", help="ip:port")
    ap.add_argument("--tokenizer", required=True, help="path to the SentencePiece model (counts tokens only; the server tokenises)")
    ap.add_argument("--dataset", required=True, help='JSON list of {"conversations": [{"value": prompt}, {"value": answer}]}')
    ap.add_argument("--request_rate", "--request-rate", dest="request_rate", default="inf",
                    help='requests per second (Poisson arrivals) or "inf" (all requests at time 0)')
    ap.add_argument("--early_stopping", "--early-stopping", dest="early_stopping", action="store_true",
                    help="stop at the end token (default: ignore_eos_token = true)")
    ap.add_argument("--num-requests", type=int, default=0, help="use only the first N dataset entries (0 = all)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dump-answers", default="", help="write {request id: generated text} to this JSON file")
    return asyncio.run(run(ap.parse_args(argv)))


if __name__ == "__main__":
    main()
