"""ctypes binding of oracle/libllama_ref.so (TEST INFRASTRUCTURE ONLY -- see oracle/llama_ref.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class ModelDesc(C.Structure):
    """Same layout as pplhip_model_desc (include/pplhip.h) and ref_model_desc (oracle/llama_ref.c)."""
    _fields_ = [("hidden_dim", C.c_int32), ("intermediate_dim", C.c_int32), ("num_layers", C.c_int32),
                ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32), ("vocab_size", C.c_int32),
                ("norm_eps", C.c_float), ("rope_theta", C.c_float), ("max_position", C.c_int32),
                ("cache_quant_bit", C.c_int32), ("cache_quant_group", C.c_int32), ("cache_layout", C.c_int32),
                ("cache_mode", C.c_int32), ("page_size", C.c_int32), ("weight_quant_bit", C.c_int32),
                ("weight_quant_group", C.c_int32), ("act_quant_bit", C.c_int32)]


class Step(C.Structure):
    """Same layout as pplhip_step / ref_step."""
    _fields_ = [("batch", C.c_int64), ("num_tokens", C.c_int64), ("decoding_batches", C.c_int64),
                ("max_seq_len", C.c_int64), ("max_kv_len", C.c_int64), ("max_pages", C.c_int64),
                ("token_inputs", C.c_void_p), ("seq_starts", C.c_void_p), ("kv_starts", C.c_void_p),
                ("start_pos", C.c_void_p), ("cache_indices", C.c_void_p), ("req_list_changed", C.c_int32)]


def make_desc(**kw):
    d = ModelDesc()
    defaults = dict(norm_eps=1e-5, rope_theta=10000.0, max_position=4096, cache_quant_bit=0, cache_quant_group=1,
                    cache_layout=3, cache_mode=0, page_size=0, weight_quant_bit=0, weight_quant_group=128, act_quant_bit=0)
    defaults.update(kw)
    if defaults.get("num_kv_heads") is None:
        defaults["num_kv_heads"] = defaults["num_heads"]
    for k, v in defaults.items():
        setattr(d, k, v)
    return d


def make_step(token_inputs, seq_starts, start_pos, cache_indices, decoding_batches, max_pages=0, req_list_changed=1):
    """Builds a Step plus the numpy arrays that keep its pointers alive (returned as step._keep)."""
    tok = np.ascontiguousarray(token_inputs, dtype=np.int64)
    ss = np.ascontiguousarray(seq_starts, dtype=np.int64)
    sp = np.ascontiguousarray(start_pos, dtype=np.int64)
    ci = np.ascontiguousarray(cache_indices, dtype=np.int64)
    B = len(sp)
    seqlens = ss[1:] - ss[:-1]
    kvs = np.zeros(B + 1, dtype=np.int64)
    kvs[1:] = np.cumsum(sp + seqlens)   # src/generator/llm_generator.cc:289
    st = Step()
    st.batch, st.num_tokens, st.decoding_batches = B, len(tok), decoding_batches
    st.max_seq_len = int(seqlens.max()) if B else 0
    st.max_kv_len = int((sp + seqlens).max()) if B else 0
    st.max_pages = max_pages
    st.token_inputs, st.seq_starts, st.kv_starts = tok.ctypes.data, ss.ctypes.data, kvs.ctypes.data
    st.start_pos, st.cache_indices = sp.ctypes.data, ci.ctypes.data
    st.req_list_changed = req_list_changed
    st._keep = (tok, ss, sp, ci, kvs)
    return st


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])


def usable_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup cpu.max quota).  The GPU boxes expose 256 logical
    CPUs but cap the container at a 16-CPU quota; 256 OpenMP threads on that quota are ~100x slower than 16."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, n)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_DIR, "libllama_ref.so")
        if not os.path.exists(path):
            build()
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))  # read by libgomp when the library loads
        L = C.CDLL(path)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.POINTER(ModelDesc), C.c_int, C.c_int]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        L.ref_get_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        L.ref_tensor_bytes.restype = C.c_int64
        L.ref_tensor_bytes.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_init_synthetic.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_kv_alloc.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_kv_ptr.restype = C.c_void_p
        L.ref_kv_ptr.argtypes = [C.c_void_p, C.c_int]
        L.ref_kv_bytes.restype = C.c_uint64
        L.ref_kv_bytes.argtypes = [C.c_void_p, C.c_int]
        L.ref_forward.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Step), C.c_void_p, C.c_void_p]
        L.ref_synth_fill.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, C.c_void_p]
        L.ref_build_rope_table.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float]
        L.ref_embedding.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ref_rmsnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_linear_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int]
        L.ref_linear_i8_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ref_quant_act_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_quant_weight_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_silu_mul.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ref_rope_kv_write.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int64]
        L.ref_attention.argtypes = [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_void_p]
        L.ref_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_void_p, C.c_void_p]
        L.ref_penalty.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p]
        L.ref_num_threads.restype = C.c_int
        L.ref_set_mode.argtypes = [C.c_int]
        _LIB = L
    return _LIB


MODE_FP16, MODE_FP32_ACT, MODE_F64_ACC, MODE_ALT_ORDER, MODE_ALT_ORDER2 = 0, 1, 2, 4, 8


class mode:
    """`with ref.mode(ref.MODE_FP32_ACT | ref.MODE_F64_ACC): ...` -- the oracle's arithmetic mode (llama_ref.c, g_mode) for the
    models CREATED and run inside the block (an unquantised KV slab allocated in fp32 mode holds fp32).  Mode 0 is the
    specification the device is compared with; the others exist to pin the algorithm (fp32 activations vs HuggingFace) and to
    measure the noise floor of the fp16 specification (alternative summation order)."""

    def __init__(self, m):
        self.m = m

    def __enter__(self):
        self.old = lib().ref_get_mode()
        lib().ref_set_mode(self.m)

    def __exit__(self, *a):
        lib().ref_set_mode(self.old)


def _p(a):
    return None if a is None else a.ctypes.data


class RefModel:
    """One tensor-parallel slice of the oracle model."""

    def __init__(self, desc, tp_size=1, tp_rank=0):
        self.desc, self.tp_size, self.tp_rank = desc, tp_size, tp_rank
        self.h = lib().ref_create(C.byref(desc), tp_size, tp_rank)
        self.kv_tokens = 0

    def close(self):
        if self.h:
            lib().ref_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_tensor(self, name, arr):
        arr = np.ascontiguousarray(arr)
        rc = lib().ref_set_tensor(self.h, name.encode(), arr.ctypes.data, arr.nbytes)
        if rc:
            raise RuntimeError(f"ref_set_tensor({name}) -> {rc} (bytes {arr.nbytes}, want {self.tensor_bytes(name)})")

    def tensor_bytes(self, name):
        return lib().ref_tensor_bytes(self.h, name.encode())

    def get_tensor(self, name, dtype):
        n = self.tensor_bytes(name)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n // np.dtype(dtype).itemsize, dtype=dtype)
        lib().ref_get_tensor(self.h, name.encode(), out.ctypes.data, n)
        return out

    def init_synthetic(self, seed):
        lib().ref_init_synthetic(self.h, seed)

    def kv_alloc(self, tokens):
        rc = lib().ref_kv_alloc(self.h, tokens)
        assert rc == 0
        self.kv_tokens = tokens

    def kv_array(self, which):
        n = lib().ref_kv_bytes(self.h, which)
        if n == 0:
            return None
        buf = (C.c_uint8 * n).from_address(lib().ref_kv_ptr(self.h, which))
        dt = np.float16 if which == 1 or self.desc.cache_quant_bit == 0 else np.int8
        if which == 0 and self.desc.cache_quant_bit == 0 and n == self.kv_tokens_elems() * 4:
            dt = np.float32                                            # slab allocated in MODE_FP32_ACT
        return np.frombuffer(buf, dtype=dt)

    def kv_tokens_elems(self):
        d = self.desc
        return self.kv_tokens * d.num_layers * 2 * (d.num_kv_heads // self.tp_size) * (d.hidden_dim // d.num_heads)


def tensor_names(desc):
    """Every tensor name of a slice (DESIGN.md "weight container")."""
    names = ["tok_embeddings.weight", "norm.weight", "output.weight"]
    for l in range(desc.num_layers):
        names += [f"layers.{l}.attention_norm.weight", f"layers.{l}.ffn_norm.weight"]
        for w in ("attention.wqkv", "attention.wo", "feed_forward.w13", "feed_forward.w2"):
            names.append(f"layers.{l}.{w}.weight")
            if desc.weight_quant_bit:
                names.append(f"layers.{l}.{w}.scale")
    return names


def forward(models, step, dump_hidden=False):
    """Runtime::Run() on `models` (list of TP slices). Returns fp32 logits [B, vocab] (and the residual dump)."""
    d = models[0].desc
    arr = (C.c_void_p * len(models))(*[m.h for m in models])
    logits = np.empty((step.batch, d.vocab_size), dtype=np.float32)
    dump = None
    if dump_hidden:
        dump = np.empty((d.num_layers + 1, step.num_tokens, d.hidden_dim), dtype=np.float32)
    rc = lib().ref_forward(arr, len(models), C.byref(step), logits.ctypes.data, _p(dump))
    assert rc == 0
    return (logits, dump) if dump_hidden else logits


def sample(logits, top_k=1, top_p=0.0, temperatures=None, top_p_list=None, rnd=None):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, V = logits.shape
    tok = np.empty(B, dtype=np.int32)
    lp = np.empty(B, dtype=np.float32)
    if rnd is None:
        rnd = np.zeros(B, dtype=np.float32)
    rnd = np.ascontiguousarray(rnd, dtype=np.float32)
    t = None if temperatures is None else np.ascontiguousarray(temperatures, dtype=np.float32)
    tp = None if top_p_list is None else np.ascontiguousarray(top_p_list, dtype=np.float32)
    lib().ref_sample(logits.ctypes.data, _p(t), _p(tp), rnd.ctypes.data, B, V, V, top_k, top_p, tok.ctypes.data,
                     lp.ctypes.data)
    return tok, lp


def f16_to_f32(a):
    return np.asarray(a, dtype=np.float16).astype(np.float32)
