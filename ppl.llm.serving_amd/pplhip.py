"""ctypes binding of libpplhip.so (include/pplhip.h) for the parity tests and bench.py.

This is NOT a fallback path: every call goes into the HIP library; if the library is missing or a device call
fails, a PplHipError is raised.  Device tensors used by the single-operator entry points are torch tensors
(PyTorch is only the device-memory allocator here).
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPLHIP_LIB") or os.path.join(_DIR, "csrc", "libpplhip.so")   # PPLHIP_LIB: diagnosis builds (profiles/)
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64
COMM_NONE, COMM_RCCL, COMM_P2P = 0, 1, 2
_LIB = None

STATUS = {0: "SUCCESS", -1: "OTHER_ERROR", -2: "INVALID_VALUE", -3: "OUT_OF_MEMORY", -4: "DEVICE_RUNTIME_ERROR",
          -5: "DEVICE_MEMORY_ERROR", -6: "NOT_FOUND", -7: "UNSUPPORTED"}

PROF_ATTN_DECODE, PROF_ATTN_PREFILL, PROF_GEMM, PROF_RUN = 0, 1, 2, 3


class PplHipError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [("hidden_dim", C.c_int32), ("intermediate_dim", C.c_int32), ("num_layers", C.c_int32),
                ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32), ("vocab_size", C.c_int32),
                ("norm_eps", C.c_float), ("rope_theta", C.c_float), ("max_position", C.c_int32),
                ("cache_quant_bit", C.c_int32), ("cache_quant_group", C.c_int32), ("cache_layout", C.c_int32),
                ("cache_mode", C.c_int32), ("page_size", C.c_int32), ("weight_quant_bit", C.c_int32),
                ("weight_quant_group", C.c_int32), ("act_quant_bit", C.c_int32)]


class Opts(C.Structure):
    _fields_ = [("n_local_ranks", C.c_int32), ("world_size", C.c_int32), ("rank_base", C.c_int32),
                ("device_ids", C.POINTER(C.c_int32)), ("nccl_unique_id", C.c_void_p),
                ("max_running_batch", C.c_int32), ("max_tokens_per_step", C.c_int32), ("enable_penalty", C.c_int32),
                ("decoding_attn_split_k", C.c_int32), ("decoding_attn_tpb", C.c_int32), ("enable_profiling", C.c_int32)]


class Step(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_tokens", C.c_int64), ("decoding_batches", C.c_int64),
                ("max_seq_len", C.c_int64), ("max_kv_len", C.c_int64), ("max_pages", C.c_int64),
                ("token_inputs", C.c_void_p), ("seq_starts", C.c_void_p), ("kv_starts", C.c_void_p),
                ("start_pos", C.c_void_p), ("cache_indices", C.c_void_p), ("req_list_changed", C.c_int32)]


class SampleArgs(C.Structure):
    _fields_ = [("temperatures", C.c_void_p), ("top_k", C.c_void_p), ("top_p", C.c_void_p), ("batch", C.c_int32),
                ("vocab_size", C.c_int32), ("batch_stride", C.c_int32), ("default_top_k", C.c_int32),
                ("default_top_p", C.c_float), ("req_list_changed", C.c_int32), ("enable_penalty", C.c_int32)]


class PenaltyArgs(C.Structure):
    _fields_ = [("temperatures", C.c_void_p), ("repetition_penalties", C.c_void_p), ("presence_penalties", C.c_void_p),
                ("frequency_penalties", C.c_void_p), ("batch_slots", C.c_void_p), ("batch", C.c_int32),
                ("vocab_size", C.c_int32), ("req_list_changed", C.c_int32)]


class KvView(C.Structure):
    _fields_ = [("cache", C.c_void_p), ("scale", C.c_void_p), ("max_tokens", C.c_int64), ("num_layers", C.c_int32),
                ("kv_heads", C.c_int32), ("head_dim", C.c_int32), ("quant_bit", C.c_int32), ("quant_group", C.c_int32),
                ("layout", C.c_int32), ("mode", C.c_int32), ("page_size", C.c_int32), ("layer", C.c_int32)]


COMM_MODES = {0: "none", 1: "rccl", 2: "direct xGMI kernels (two-shot, all links)"}
COMM_SCHEDULES = {0: "one-stream (collectives in stream)", 1: "two-stream (half-batches, a channel each)", 2: "chunked (collectives on the communication stream)"}


class CommInfo(C.Structure):
    _fields_ = [("mode", C.c_int32), ("selftest", C.c_int32), ("schedule", C.c_int32), ("has_rccl", C.c_int32),
                ("dual_min_rows", C.c_int64), ("dual_max_rows", C.c_int64), ("notes", C.c_char * 512)]


# every symbol include/pplhip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "pplhip_version", "pplhip_device_count", "pplhip_get_unique_id", "pplhip_init", "pplhip_destroy",
    "pplhip_comm_export", "pplhip_comm_connect", "pplhip_comm_mode", "pplhip_comm_fused_norm", "pplhip_comm_info", "pplhip_comm_allreduce_us",
    "pplhip_last_error", "pplhip_rank_load", "pplhip_rank_set_tensor", "pplhip_rank_init_synthetic", "pplhip_rank_tie_output",
    "pplhip_kv_block_bytes", "pplhip_kv_capacity", "pplhip_kv_alloc", "pplhip_kv_ptrs", "pplhip_kv_read",
    "pplhip_kv_write", "pplhip_kv_fill_synthetic", "pplhip_set_inputs", "pplhip_run", "pplhip_debug_run_dump", "pplhip_logits", "pplhip_copy_logits", "pplhip_sync",
    "pplhip_sample", "pplhip_penalty", "pplhip_profile_reset", "pplhip_profile_get", "pplhip_profile_mode", "pplhip_mem_info",
    "pplhip_op_embedding", "pplhip_op_rmsnorm", "pplhip_op_linear", "pplhip_op_linear_swiglu", "pplhip_op_rmsnorm_quant", "pplhip_op_quant_act", "pplhip_op_quant_weight",
    "pplhip_op_linear_i8", "pplhip_op_silu_mul", "pplhip_op_rope_kv_write",
    "pplhip_op_attention", "pplhip_build_rope_table",
]


def build():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_DIR, "csrc")])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PplHipError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (no CPU fallback exists)")
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
        L.pplhip_last_error.restype = C.c_char_p
        L.pplhip_last_error.argtypes = [vp, C.c_int]
        L.pplhip_get_unique_id.argtypes = [vp]
        L.pplhip_init.argtypes = [C.POINTER(ModelDesc), C.POINTER(Opts), C.POINTER(vp)]
        L.pplhip_destroy.argtypes = [vp]
        L.pplhip_destroy.restype = None
        L.pplhip_comm_export.argtypes = [vp, C.c_int, vp]
        L.pplhip_comm_connect.argtypes = [vp, vp]
        L.pplhip_comm_mode.argtypes = [vp]
        L.pplhip_comm_fused_norm.argtypes = [vp]
        L.pplhip_comm_info.argtypes = [vp, i64, C.POINTER(CommInfo)]
        L.pplhip_comm_allreduce_us.argtypes = [vp, C.c_int, i64, i32, i32, C.POINTER(f32)]
        L.pplhip_rank_load.argtypes = [vp, C.c_int, C.c_char_p]
        L.pplhip_rank_set_tensor.argtypes = [vp, C.c_int, C.c_char_p, vp, u64]
        L.pplhip_rank_init_synthetic.argtypes = [vp, C.c_int, u64]
        L.pplhip_rank_tie_output.argtypes = [vp, C.c_int, i64, u64, f32]
        L.pplhip_kv_block_bytes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.pplhip_kv_capacity.argtypes = [vp, f32, C.POINTER(u64)]
        L.pplhip_kv_alloc.argtypes = [vp, C.c_int, u64]
        L.pplhip_kv_ptrs.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]
        L.pplhip_kv_read.argtypes = [vp, C.c_int, C.c_int, u64, vp, u64]
        L.pplhip_kv_write.argtypes = [vp, C.c_int, C.c_int, u64, vp, u64]
        L.pplhip_kv_fill_synthetic.argtypes = [vp, C.c_int, u64]
        L.pplhip_set_inputs.argtypes = [vp, C.c_int, C.POINTER(Step)]
        L.pplhip_run.argtypes = [vp, C.c_int, C.c_int]
        L.pplhip_debug_run_dump.argtypes = [vp, C.c_int, vp]
        L.pplhip_logits.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(i64)]
        L.pplhip_copy_logits.argtypes = [vp, C.c_int, vp, i64]
        L.pplhip_sync.argtypes = [vp, C.c_int]
        L.pplhip_sample.argtypes = [vp, vp, C.POINTER(SampleArgs), vp, vp]
        L.pplhip_penalty.argtypes = [vp, vp, C.POINTER(PenaltyArgs)]
        L.pplhip_profile_reset.argtypes = [vp, C.c_int]
        L.pplhip_profile_get.argtypes = [vp, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(C.c_double)]
        L.pplhip_profile_mode.argtypes = [vp, C.c_int]
        L.pplhip_mem_info.argtypes = [vp, C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.pplhip_op_embedding.argtypes = [vp, vp, vp, i64, i32, vp]
        L.pplhip_op_rmsnorm.argtypes = [vp, vp, vp, vp, f32, i64, i32, vp, vp]
        L.pplhip_op_linear.argtypes = [vp, vp, vp, vp, i32, i32, i64, i32, i32, vp, i32]
        L.pplhip_op_linear_swiglu.argtypes = [vp, vp, vp, vp, i32, i32, i64, i32, i32, vp]
        L.pplhip_op_rmsnorm_quant.argtypes = [vp, vp, vp, vp, C.c_float, i64, i32, vp, vp, vp]
        L.pplhip_op_quant_act.argtypes = [vp, vp, i64, i32, vp, vp]
        L.pplhip_op_quant_weight.argtypes = [vp, vp, i32, i32, vp, vp]
        L.pplhip_op_linear_i8.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp, i32, i32]
        L.pplhip_op_silu_mul.argtypes = [vp, vp, i64, i32, vp]
        L.pplhip_op_rope_kv_write.argtypes = [vp, vp, vp, C.POINTER(KvView), vp, vp, vp, i64, i64, i64, i32]
        L.pplhip_op_attention.argtypes = [vp, vp, C.POINTER(KvView), vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, i32,
                                          vp, u64, vp]
        L.pplhip_build_rope_table.argtypes = [vp, i32, i32, f32]
        _LIB = L
    return _LIB


def make_desc(**kw):
    d = ModelDesc()
    defaults = dict(norm_eps=1e-5, rope_theta=10000.0, max_position=4096, cache_quant_bit=0, cache_quant_group=1,
                    cache_layout=3, cache_mode=0, page_size=0, weight_quant_bit=0, weight_quant_group=128, act_quant_bit=0)
    defaults.update(kw)
    for k, v in defaults.items():
        setattr(d, k, v)
    return d


def copy_desc(src, cls=ModelDesc):
    d = cls()
    for name, _ in ModelDesc._fields_:
        setattr(d, name, getattr(src, name))
    return d


def get_unique_id():
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    rc = lib().pplhip_get_unique_id(buf)
    if rc:
        raise PplHipError(f"pplhip_get_unique_id -> {STATUS.get(rc, rc)}")
    return bytes(buf)


def write_container(path, tensors):
    """weights.pplhip: "PPLHIPW1" | u32 count | count x { u32 name_len | name | u64 nbytes | pad to 64 | data }"""
    with open(path, "wb") as f:
        f.write(b"PPLHIPW1")
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            arr = np.ascontiguousarray(arr)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<Q", arr.nbytes))
            pad = (64 - f.tell() % 64) % 64
            f.write(b"\0" * pad)
            f.write(arr.tobytes())


class Context:
    """One pplhip_ctx.  n_local_ranks ranks in this process; world_size may be larger (one process per GPU)."""

    def __init__(self, desc, max_running_batch, max_tokens_per_step, n_local_ranks=1, world_size=None, rank_base=0,
                 device_ids=None, unique_id=None, enable_penalty=False, split_k=1, tpb=0, profiling=False):
        self.desc = desc
        o = Opts()
        o.n_local_ranks = n_local_ranks
        o.world_size = world_size if world_size else n_local_ranks
        o.rank_base = rank_base
        self._devs = None
        if device_ids is not None:
            self._devs = (C.c_int32 * len(device_ids))(*device_ids)
            o.device_ids = C.cast(self._devs, C.POINTER(C.c_int32))
        self._uid = None
        if unique_id is not None:
            self._uid = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
            o.nccl_unique_id = C.cast(self._uid, C.c_void_p)
        o.max_running_batch, o.max_tokens_per_step = max_running_batch, max_tokens_per_step
        o.enable_penalty = int(enable_penalty)
        o.decoding_attn_split_k, o.decoding_attn_tpb, o.enable_profiling = split_k, tpb, int(profiling)
        self.opts = o
        self.h = C.c_void_p()
        rc = lib().pplhip_init(C.byref(desc), C.byref(o), C.byref(self.h))
        if rc:
            raise PplHipError(f"pplhip_init -> {STATUS.get(rc, rc)}")
        self.n_local = n_local_ranks

    def _ck(self, rc, rank=0, what=""):
        if rc:
            msg = lib().pplhip_last_error(self.h, rank)
            raise PplHipError(f"{what} -> {STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def close(self):
        if self.h:
            lib().pplhip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # tensor-parallel collectives (one process per GPU: export -> all-gather by the launcher -> connect)
    def comm_export(self, rank=0):
        buf = (C.c_uint8 * IPC_HANDLE_BYTES)()
        self._ck(lib().pplhip_comm_export(self.h, rank, buf), rank, "comm_export")
        return bytes(buf)

    def comm_connect(self, all_handles):
        blob = b"".join(all_handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._ck(lib().pplhip_comm_connect(self.h, buf), -1, "comm_connect")

    def comm_mode(self):
        return lib().pplhip_comm_mode(self.h)

    def comm_info(self, rows):
        """what a multi-GPU run does at a pure-decode step of `rows` rows: mode, self-test verdict, schedule, fallbacks taken"""
        ci = CommInfo()
        self._ck(lib().pplhip_comm_info(self.h, rows, C.byref(ci)), -1, "comm_info")
        mode = COMM_MODES[ci.mode]
        if lib().pplhip_comm_fused_norm(self.h) == 1:
            mode += " + residual add and RMSNorm fused between the two shots (sequence-parallel residual stream)"
        return {"mode": mode, "selftest": {0: "not run", 1: "passed", -1: "failed"}[ci.selftest],
                "schedule": COMM_SCHEDULES[ci.schedule], "rccl_communicator": bool(ci.has_rccl),
                "two_stream_rows": [int(ci.dual_min_rows), int(ci.dual_max_rows)], "fallbacks": ci.notes.decode(errors="replace")}

    def comm_allreduce_us(self, rows, iters=20, path=0, rank=0):
        """average microseconds of one all-reduce of fp16 [rows, hidden] (collective call); None when the path does not exist"""
        us = C.c_float(-1.0)
        self._ck(lib().pplhip_comm_allreduce_us(self.h, rank, rows, iters, path, C.byref(us)), rank, "comm_allreduce_us")
        return None if us.value < 0 else float(us.value)

    # weights
    def set_tensor(self, rank, name, arr):
        arr = np.ascontiguousarray(arr)
        self._ck(lib().pplhip_rank_set_tensor(self.h, rank, name.encode(), arr.ctypes.data, arr.nbytes), rank, f"set_tensor({name})")

    def load(self, rank, slice_dir):
        self._ck(lib().pplhip_rank_load(self.h, rank, slice_dir.encode()), rank, "rank_load")

    def init_synthetic(self, rank, seed):
        self._ck(lib().pplhip_rank_init_synthetic(self.h, rank, seed), rank, "rank_init_synthetic")

    # kv
    def kv_block_bytes(self):
        kb, sb = C.c_uint64(), C.c_uint64()
        self._ck(lib().pplhip_kv_block_bytes(self.h, C.byref(kb), C.byref(sb)))
        return kb.value, sb.value

    def kv_capacity(self, scale):
        t = C.c_uint64()
        self._ck(lib().pplhip_kv_capacity(self.h, scale, C.byref(t)), 0, "kv_capacity")
        return t.value

    def kv_alloc(self, rank, tokens):
        self._ck(lib().pplhip_kv_alloc(self.h, rank, tokens), rank, "kv_alloc")
        self.kv_tokens = tokens

    def kv_read(self, rank, which):
        kb, sb = self.kv_block_bytes()
        n = self.kv_tokens * (sb if which else kb)
        if n == 0:
            return None
        dt = np.float16 if which == 1 or self.desc.cache_quant_bit == 0 else np.int8
        out = np.empty(n // np.dtype(dt).itemsize, dtype=dt)
        self._ck(lib().pplhip_kv_read(self.h, rank, which, 0, out.ctypes.data, n), rank, "kv_read")
        return out

    def kv_write(self, rank, which, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        self._ck(lib().pplhip_kv_write(self.h, rank, which, offset, arr.ctypes.data, arr.nbytes), rank, "kv_write")

    def kv_fill_synthetic(self, rank, seed):
        self._ck(lib().pplhip_kv_fill_synthetic(self.h, rank, seed), rank, "kv_fill_synthetic")

    # step
    def set_inputs(self, rank, step):
        self._ck(lib().pplhip_set_inputs(self.h, rank, C.byref(step)), rank, "set_inputs")

    def run(self, rank, cache_prefill=0):
        self._ck(lib().pplhip_run(self.h, rank, cache_prefill), rank, "run")

    def run_dump(self, rank, num_tokens):
        """diagnosis: run the step and return the residual stream after every layer, fp32 [L+1, T, hidden] (oracle convention)"""
        out = np.empty((self.desc.num_layers + 1, num_tokens, self.desc.hidden_dim), dtype=np.float32)
        self._ck(lib().pplhip_debug_run_dump(self.h, rank, out.ctypes.data), rank, "debug_run_dump")
        return out

    def sync(self, rank=0):
        self._ck(lib().pplhip_sync(self.h, rank), rank, "sync")

    def logits_ptr(self, rank=0):
        p, s = C.c_void_p(), C.c_int64()
        self._ck(lib().pplhip_logits(self.h, rank, C.byref(p), C.byref(s)))
        return p.value, s.value

    def copy_logits(self, batch, rank=0):
        out = np.empty((batch, self.desc.vocab_size), dtype=np.float32)
        self._ck(lib().pplhip_copy_logits(self.h, rank, out.ctypes.data, batch), rank, "copy_logits")
        return out

    def sample(self, batch, top_k=1, top_p=0.0, temperatures=None, top_p_list=None, req_list_changed=True,
               enable_penalty=False, logits_ptr=None):
        a = SampleArgs()
        keep = []
        if temperatures is not None:
            t = np.ascontiguousarray(temperatures, dtype=np.float32); keep.append(t); a.temperatures = t.ctypes.data
        if top_p_list is not None:
            t = np.ascontiguousarray(top_p_list, dtype=np.float32); keep.append(t); a.top_p = t.ctypes.data
        a.batch, a.vocab_size, a.batch_stride = batch, self.desc.vocab_size, self.desc.vocab_size
        a.default_top_k, a.default_top_p = top_k, top_p
        a.req_list_changed, a.enable_penalty = int(req_list_changed), int(enable_penalty)
        tok = np.empty(batch, dtype=np.int32)
        lp = np.empty(batch, dtype=np.float32)
        if logits_ptr is None:
            logits_ptr = self.logits_ptr(0)[0]
        self._ck(lib().pplhip_sample(self.h, logits_ptr, C.byref(a), tok.ctypes.data, lp.ctypes.data), 0, "sample")
        return tok, lp

    def penalty(self, temperatures, repetition, presence, frequency, batch_slots, req_list_changed=True):
        a = PenaltyArgs()
        B = len(batch_slots)
        t = np.ascontiguousarray(temperatures, dtype=np.float32)
        r = np.ascontiguousarray(repetition, dtype=np.float32)
        p = None if presence is None else np.ascontiguousarray(presence, dtype=np.float32)
        f = None if frequency is None else np.ascontiguousarray(frequency, dtype=np.float32)
        s = np.ascontiguousarray(batch_slots, dtype=np.int64)
        a.temperatures, a.repetition_penalties, a.batch_slots = t.ctypes.data, r.ctypes.data, s.ctypes.data
        a.presence_penalties = None if p is None else p.ctypes.data
        a.frequency_penalties = None if f is None else f.ctypes.data
        a.batch, a.vocab_size, a.req_list_changed = B, self.desc.vocab_size, int(req_list_changed)
        self._ck(lib().pplhip_penalty(self.h, self.logits_ptr(0)[0], C.byref(a)), 0, "penalty")

    # measurement
    def profile_reset(self, rank=0):
        self._ck(lib().pplhip_profile_reset(self.h, rank), rank, "profile_reset")

    def profile_get(self, cls, rank=0):
        n, ms = C.c_int64(), C.c_double()
        self._ck(lib().pplhip_profile_get(self.h, rank, cls, C.byref(n), C.byref(ms)), rank, "profile_get")
        return n.value, ms.value

    def profile_mode(self, mode):
        self._ck(lib().pplhip_profile_mode(self.h, mode), 0, "profile_mode")

    def mem_info(self, rank=0):
        f, t = C.c_uint64(), C.c_uint64()
        self._ck(lib().pplhip_mem_info(self.h, rank, C.byref(f), C.byref(t)), rank, "mem_info")
        return f.value, t.value


def make_step(token_inputs, seq_starts, start_pos, cache_indices, decoding_batches, max_pages=0, req_list_changed=1):
    tok = np.ascontiguousarray(token_inputs, dtype=np.int64)
    ss = np.ascontiguousarray(seq_starts, dtype=np.int64)
    sp = np.ascontiguousarray(start_pos, dtype=np.int64)
    ci = np.ascontiguousarray(cache_indices, dtype=np.int64)
    B = len(sp)
    seqlens = ss[1:] - ss[:-1]
    kvs = np.zeros(B + 1, dtype=np.int64)
    kvs[1:] = np.cumsum(sp + seqlens)
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


def shard_weights(weights, desc, tp, rank):
    """Export-side helper: slices an UNSHARDED fp16 weight dict (names of DESIGN.md section 3) for tensor-parallel rank
    `rank` of `tp` -- the partitioning of SURVEY.md 8(e): wqkv / w13 / output on the output dim (heads, inter, vocab),
    wo / w2 on the input dim; embeddings and norms replicated.  (Quantised models are quantised per slice AFTER this.)"""
    H, Hkv, hd = desc.num_heads, desc.num_kv_heads, desc.hidden_dim
    D, inter, V = hd // H, desc.intermediate_dim, desc.vocab_size
    h, hk, it, vl = H // tp, Hkv // tp, inter // tp, V // tp
    out = {}
    for name, w in weights.items():
        w = np.asarray(w)
        if name.endswith("attention.wqkv.weight"):
            w = w.reshape((H + 2 * Hkv) * D, hd)
            q, k, v = w[:H * D], w[H * D:(H + Hkv) * D], w[(H + Hkv) * D:]
            out[name] = np.concatenate([q[rank * h * D:(rank + 1) * h * D], k[rank * hk * D:(rank + 1) * hk * D],
                                        v[rank * hk * D:(rank + 1) * hk * D]], 0)
        elif name.endswith("attention.wo.weight"):
            out[name] = np.ascontiguousarray(w.reshape(hd, H * D)[:, rank * h * D:(rank + 1) * h * D])
        elif name.endswith("feed_forward.w13.weight"):
            w = w.reshape(2 * inter, hd)
            out[name] = np.concatenate([w[rank * it:(rank + 1) * it], w[inter + rank * it:inter + (rank + 1) * it]], 0)
        elif name.endswith("feed_forward.w2.weight"):
            out[name] = np.ascontiguousarray(w.reshape(hd, inter)[:, rank * it:(rank + 1) * it])
        elif name == "output.weight":
            out[name] = np.ascontiguousarray(w.reshape(V, hd)[rank * vl:(rank + 1) * vl])
        else:
            out[name] = w
    return out
