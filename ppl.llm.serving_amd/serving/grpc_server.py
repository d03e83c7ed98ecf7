#!/usr/bin/env python3
"""gRPC front end of the serving core: wire-compatible with the reference's `ppl_llm_server`
(src/serving/grpc/grpc_server.cc:88-341, proto in llm_proto.py), so `client_qps_measure[_token_in_out]` and other
clients of the reference can talk to this backend unchanged.

The reference's server is C++ on grpc++, which this image does not have; the front end is an I/O shell, so it is written
against `grpcio` and drives the C++ generator through the C ABI of src/capi/serving_c.h (build/libpplserving_c.so).
Behaviour kept from grpc_server.cc:
  * ParseRequest (:218-252): prompt vs tokens, do_sample false -> top_k 1 / top_p 0, top_p outside [0,1] -> 0,
    temperature 0 -> 1, repetition_penalty 0 -> 1, max_new_tokens, early_stopping = !ignore_eos_token, stop tokens of
    the request are NOT forwarded (the reference builds an empty set, :225);
  * every generated token is streamed as a Response {status PROCESSING | FINISHED, id = the client's id, tokens.ids =
    [token] (or `generated` text when a tokenizer is configured), detail {logprobs, is_special, finish_reason}} (:88-135);
  * a rejected request answers {status FAILED, id} (:137-150); the stream ends when every request of the call finished;
  * a client that goes away cancels its unfinished requests (NewCallThreadFunc -> on_disconnected_func, :293-312).

    python grpc_server.py --model-param-path params.json --model-dir <dir> [--host 127.0.0.1 --port 10086] [tool flags]
"""
import argparse
import asyncio
import ctypes as C
import os
import sys
import threading

import grpc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import llm_proto as P  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(HERE), "build", "libpplserving_c.so")


class Config(C.Structure):
    _fields_ = [("model_param_path", C.c_char_p), ("model_dir", C.c_char_p), ("tensor_parallel_size", C.c_int32),
                ("synthetic_weights", C.c_int32), ("synthetic_seed", C.c_uint64), ("kv_cache_max_tokens", C.c_uint64),
                ("max_tokens_scale", C.c_float), ("max_running_batch", C.c_int32), ("max_tokens_per_step", C.c_int32),
                ("max_input_tokens_per_request", C.c_int32), ("max_output_tokens_per_request", C.c_int32),
                ("max_total_tokens_per_request", C.c_int32), ("max_prefill_batch", C.c_int32), ("max_cooldown_request", C.c_int32),
                ("enable_prefix_cache", C.c_int32), ("enable_penalty", C.c_int32), ("stop_tokens", C.POINTER(C.c_int32)),
                ("n_stop_tokens", C.c_int32), ("tokenizer_path", C.c_char_p), ("tokenizer_type", C.c_char_p),
                ("model_type", C.c_char_p), ("quant_method", C.c_char_p), ("top_p", C.c_float), ("top_k", C.c_int32),
                ("decoding_attn_split_k", C.c_int32), ("decoding_attn_tpb", C.c_int32)]


class CRequest(C.Structure):
    _fields_ = [("id", C.c_uint64), ("tokens", C.POINTER(C.c_int32)), ("n_tokens", C.c_int32), ("temperature", C.c_float),
                ("top_p", C.c_float), ("top_k", C.c_int32), ("repetition_penalty", C.c_float), ("presence_penalty", C.c_float),
                ("frequency_penalty", C.c_float), ("generation_length", C.c_int32), ("early_stopping", C.c_int32),
                ("prompt", C.c_char_p), ("n_prompt", C.c_int32)]


class CResponse(C.Structure):
    _fields_ = [("id", C.c_uint64), ("token", C.c_int32), ("logprob", C.c_float), ("status", C.c_int32),
                ("finish_reason", C.c_int32), ("is_special", C.c_int32), ("text_len", C.c_int32), ("text_off", C.c_int64)]


def load_lib():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `make` in ppl.llm.serving_amd (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    L.pplsrv_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.pplsrv_submit.argtypes = [C.c_void_p, C.POINTER(CRequest), C.c_int32]
    L.pplsrv_poll.argtypes = [C.c_void_p, C.POINTER(CResponse), C.c_int32, C.c_int32]
    L.pplsrv_poll_text.argtypes = [C.c_void_p, C.POINTER(CResponse), C.c_int32, C.c_int32, C.c_char_p, C.c_int64]
    L.pplsrv_cancel.argtypes = [C.c_void_p, C.c_uint64]
    L.pplsrv_kv_cache_max_tokens.argtypes = [C.c_void_p]
    L.pplsrv_kv_cache_max_tokens.restype = C.c_uint64
    L.pplsrv_destroy.argtypes = [C.c_void_p]
    L.pplsrv_destroy.restype = None
    return L


def parse_request(pb):
    """grpc_server.cc:218-252 -> keyword values of pplsrv_request (tokens handled by the caller)"""
    cp, sp = pb.choosing_parameters, pb.stopping_parameters
    top_k, top_p = (cp.top_k, cp.top_p) if cp.do_sample else (1, 0.0)
    if top_p > 1 or top_p < 0:
        top_p = 0.0
    return dict(temperature=cp.temperature if cp.temperature != 0 else 1.0, top_k=int(top_k), top_p=float(top_p),
                repetition_penalty=cp.repetition_penalty if cp.repetition_penalty != 0 else 1.0,
                presence_penalty=cp.presence_penalty, frequency_penalty=cp.frequency_penalty,
                generation_length=int(sp.max_new_tokens), early_stopping=0 if sp.ignore_eos_token else 1)


class Serving:
    def __init__(self, lib, handle, has_tokenizer=False):
        # text requests are tokenised and detokenised INSIDE the C++ generator (src/tokenizer, LLMGenerator::Process and
        # DecodeAndSend with its U+FFFD buffering), like the reference's server; this shell only moves bytes
        self.lib, self.h, self.has_tok = lib, handle, has_tokenizer
        self.loop = None
        self.uuid_seq = 0                     # mapped ids: unique across calls (grpc_server.cc:176-178)
        self.routes = {}                      # mapped id -> (asyncio.Queue of the call, client id, text mode)
        self.lock = threading.Lock()
        self.stop = False
        self.poller = None

    # ---- response path: one thread drains the C queue and hands every call its responses ----------------------------
    def start(self, loop):
        self.loop = loop
        self.poller = threading.Thread(target=self._poll, daemon=True)
        self.poller.start()

    def _poll(self):
        buf = (CResponse * 4096)()
        tcap = 1 << 20
        tbuf = C.create_string_buffer(tcap)
        while not self.stop:
            n = self.lib.pplsrv_poll_text(self.h, buf, 4096, 50, tbuf, tcap)
            if n <= 0:
                continue
            per_call = {}
            with self.lock:
                for i in range(n):
                    r = buf[i]
                    route = self.routes.get(r.id)
                    if route is None:
                        continue              # the call is gone (client disconnected)
                    q, orig_id, text = route
                    if r.status != P.PROCESSING:
                        del self.routes[r.id]
                    # (C.string_at copies only the piece; tbuf.raw would copy the whole 1 MiB buffer for every response)
                    piece = (C.string_at(C.addressof(tbuf) + r.text_off, r.text_len).decode("utf-8", errors="replace")
                             if (text and r.text_off >= 0) else "")
                    per_call.setdefault(id(q), (q, []))[1].append((orig_id, r.token, r.logprob, r.status, r.finish_reason,
                                                                   r.is_special, text, piece))
            for q, items in per_call.values():
                self.loop.call_soon_threadsafe(q.put_nowait, items)

    def shutdown(self):
        self.stop = True
        if self.poller:
            self.poller.join(timeout=2)
        self.lib.pplsrv_destroy(self.h)

    # ---- request path ------------------------------------------------------------------------------------------------
    async def generation(self, request, context):
        n = len(request.req)
        if n == 0:
            return
        q = asyncio.Queue()
        creqs = (CRequest * n)()
        keep = []
        failed = []
        with self.lock:
            base = self.uuid_seq
            self.uuid_seq += n
        mapped = []
        for i, pb in enumerate(request.req):
            text = bool(pb.prompt)
            if text and not self.has_tok:
                failed.append(pb.id)          # no tokenizer configured: the text path cannot be served
                continue
            kw = parse_request(pb)
            c = creqs[len(mapped)]
            c.id = base + i
            if text:
                raw = pb.prompt.encode("utf-8")
                keep.append(raw)
                c.tokens, c.n_tokens, c.prompt, c.n_prompt = None, 0, raw, len(raw)
            else:
                tokens = list(pb.tokens.ids)
                arr = (C.c_int32 * max(len(tokens), 1))(*tokens)
                keep.append(arr)
                c.tokens, c.n_tokens, c.prompt, c.n_prompt = arr, len(tokens), None, 0
            for k, v in kw.items():
                setattr(c, k, v)
            mapped.append((base + i, pb.id, text))
        with self.lock:                        # routes exist before the generator can answer (grpc_server.cc:180-190)
            for mid, orig, text in mapped:
                self.routes[mid] = (q, orig, text)
        if mapped:
            rc = self.lib.pplsrv_submit(self.h, creqs, len(mapped))
            if rc != 0:
                failed += [orig for _, orig, _ in mapped]
                with self.lock:
                    for mid, _, _ in mapped:
                        self.routes.pop(mid, None)
                mapped = []
        if failed:
            out = P.BatchedResponse()
            for orig in failed:
                r = out.rsp.add()
                r.status, r.id = P.FAILED, orig
            yield out
        pending = len(mapped)
        try:
            while pending > 0:
                items = await q.get()
                out = P.BatchedResponse()
                for orig_id, token, logprob, status, reason, special, text, piece in items:
                    r = out.rsp.add()
                    r.status, r.id = status, orig_id
                    if status == P.FAILED:
                        pending -= 1
                        continue
                    if text:
                        r.generated = piece
                    else:
                        r.tokens.ids.append(token & 0xFFFFFFFF)
                    r.detail.logprobs, r.detail.is_special, r.detail.finish_reason = logprob, bool(special), reason
                    if status == P.FINISHED:
                        pending -= 1
                yield out
        finally:
            if pending > 0:                    # client went away: cancel what is still running
                with self.lock:
                    gone = [mid for mid, _, _ in mapped if mid in self.routes]
                    for mid in gone:
                        del self.routes[mid]
                for mid in gone:
                    self.lib.pplsrv_cancel(self.h, mid)


def make_config(a):
    stop = [int(t) for t in a.stop_tokens.split(",") if t.strip()] if a.stop_tokens else []
    arr = (C.c_int32 * max(len(stop), 1))(*stop)
    cfg = Config(model_param_path=a.model_param_path.encode(), model_dir=(a.model_dir or "").encode(),
                 tensor_parallel_size=a.tensor_parallel_size, synthetic_weights=int(a.synthetic_weights), synthetic_seed=a.synthetic_seed,
                 kv_cache_max_tokens=a.kv_cache_max_tokens, max_tokens_scale=a.max_tokens_scale, max_running_batch=a.max_running_batch,
                 max_tokens_per_step=a.max_tokens_per_step, max_input_tokens_per_request=a.max_input_tokens_per_request,
                 max_output_tokens_per_request=a.max_output_tokens_per_request,
                 max_total_tokens_per_request=a.max_total_tokens_per_request, max_prefill_batch=a.max_prefill_batch,
                 max_cooldown_request=a.max_cooldown_request, enable_prefix_cache=int(a.enable_prefix_cache),
                 enable_penalty=int(a.enable_penalty), stop_tokens=arr, n_stop_tokens=len(stop),
                 tokenizer_path=(a.tokenizer_path or "").encode(), tokenizer_type=(a.tokenizer_type or "sentencepiece").encode(),
                 model_type=(a.model_type or "llama").encode(), quant_method=(a.quant_method or "none").encode(),
                 top_p=a.top_p, top_k=a.top_k, decoding_attn_split_k=a.configure_decoding_attn_split_k + 1,
                 decoding_attn_tpb=a.specify_decoding_attn_tpb)
    cfg._keep = arr
    return cfg


def add_flags(ap):
    """the flag names of tools/llm_server.cc / offline_inference.cc:40-90 that apply here"""
    ap.add_argument("--model-param-path", required=True)
    ap.add_argument("--model-dir", default="")
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--max-tokens-scale", type=float, default=0.94)
    ap.add_argument("--max-input-tokens-per-request", type=int, default=4096)
    ap.add_argument("--max-output-tokens-per-request", type=int, default=4096)
    ap.add_argument("--max-total-tokens-per-request", type=int, default=8192)
    ap.add_argument("--max-running-batch", type=int, default=1024)
    ap.add_argument("--max-tokens-per-step", type=int, default=8192)
    ap.add_argument("--max-prefill-batch", type=int, default=64)
    ap.add_argument("--max-cooldown-request", type=int, default=2)
    ap.add_argument("--enable-prefix-cache", action="store_true")
    ap.add_argument("--enable-penalty", action="store_true")
    ap.add_argument("--stop-tokens", default="")
    ap.add_argument("--tokenizer-path", default="", help="sentencepiece model; without it only the token-in/token-out path is served")
    ap.add_argument("--tokenizer-type", default="sentencepiece")
    ap.add_argument("--model-type", default="llama")
    ap.add_argument("--quant-method", default="none", help="none | online_i8i8 (tools/llm_server.cc:62)")
    ap.add_argument("--top-p", type=float, default=0.0)
    ap.add_argument("--top-k", type=int, default=1)
    ap.add_argument("--configure-decoding-attn-split-k", type=int, default=1, help="always-on(2)/heuristic(1)/off(0)")
    ap.add_argument("--specify-decoding-attn-tpb", type=int, default=0, help="512/256/heuristic(0)")
    # flags of tools/llm_server.cc that have no effect on this backend: accepted so that a command line carries over unchanged
    ap.add_argument("--model-format", default="pplhip")
    ap.add_argument("--tokenizer-config-path", default="")
    ap.add_argument("--special-tokens", default="")
    ap.add_argument("--cublas-layout-hint", default="default")
    for flag in ("--disable-decoding-shm-mha", "--disable-decoding-inf-mha", "--disable-decoding-inf-gqa", "--disable-graph-fusion",
                 "--enable-profiling", "--enable-backtrace", "--version"):
        ap.add_argument(flag, action="store_true")
    ap.add_argument("--monitor-port", type=int, default=23333)
    ap.add_argument("--control-port", type=int, default=12345)
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--synthetic-seed", type=int, default=1234)
    ap.add_argument("--kv-cache-max-tokens", type=int, default=0)
    ap.add_argument("--host", default="127.0.0.1")            # tools/llm_server.cc:84-85
    ap.add_argument("--port", type=int, default=10086)


async def serve(a, ready=None):
    lib = load_lib()
    h = C.c_void_p()
    rc = lib.pplsrv_create(C.byref(make_config(a)), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"pplsrv_create failed: RetCode {-rc}")
    srv = Serving(lib, h, bool(a.tokenizer_path))
    srv.start(asyncio.get_running_loop())
    server = grpc.aio.server(options=[("grpc.max_receive_message_length", 64 << 20), ("grpc.max_send_message_length", 64 << 20)])
    handler = grpc.method_handlers_generic_handler(P.SERVICE, {
        "Generation": grpc.unary_stream_rpc_method_handler(srv.generation, request_deserializer=P.BatchedRequest.FromString,
                                                           response_serializer=P.BatchedResponse.SerializeToString)})
    server.add_generic_rpc_handlers((handler,))
    port = server.add_insecure_port(f"{a.host}:{a.port}")
    await server.start()
    print(f"ppl_llm_server (hip backend) listening on {a.host}:{port}, kv_cache_max_tokens {lib.pplsrv_kv_cache_max_tokens(h)}",
          file=sys.stderr, flush=True)
    if ready is not None:
        ready(port)
    try:
        await server.wait_for_termination()
    finally:
        await server.stop(0)
        srv.shutdown()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    add_flags(ap)
    a = ap.parse_args(argv)
    asyncio.run(serve(a))


if __name__ == "__main__":
    main()
