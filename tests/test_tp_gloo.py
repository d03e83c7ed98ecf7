"""Tensor parallelism without GPUs:
  1. the sharding spec: the oracle run on two slices produced by pplhip.shard_weights (+ explicit all-reduce /
     all-gather inside ref_forward) equals the unsharded oracle;
  2. world_size 2 over gloo: two PROCESSES each own one slice, run the per-rank half of every layer with the oracle's
     operators in exactly the order libpplhip's pplhip_run issues them, exchange through torch.distributed
     (all_reduce of the row-parallel outputs, all_gather of the vocab shards) and reproduce the unsharded logits -- in both step schedules:
     all-reduce + replicated norm (RCCL), and the direct collectives' sequence-parallel residual stream (reduce-scatter by rows, residual add +
     RMSNorm on the owned rows, all-gather of normed rows: csrc/k_comm.hip p2p_allreduce_norm_kernel, round 6);
  3. bench.py's multi-process control plane (rendezvous, unique-id broadcast, MAX-over-ranks timing) in --dry-run."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref
from tests.conftest import ROOT, load_pplhip
from tests.test_oracle_hf import desc_from_meta, load_fixture


def _setup(golden_dir):
    meta, weights, prompts, hf_logits, _, _ = load_fixture(os.path.join(golden_dir, "hf_tiny_gqa.npz"))
    desc = desc_from_meta(meta, cache_quant_bit=8, cache_quant_group=8)
    return desc, weights, prompts


def _step(prompts):
    lens = np.array([len(p) for p in prompts])
    return ref.make_step(np.concatenate(prompts), np.concatenate([[0], np.cumsum(lens)]), np.zeros(len(prompts), dtype=np.int64),
                         np.concatenate([[0], np.cumsum(lens + 4)[:-1]]), 0)


def test_sharded_oracle_equals_unsharded(golden_dir):
    m = load_pplhip()
    desc, weights, prompts = _setup(golden_dir)
    full = ref.RefModel(desc)
    for k, v in weights.items():
        full.set_tensor(k, v)
    full.kv_alloc(64)
    want = ref.forward([full], _step(prompts))
    slices = []
    for r in range(2):
        s = ref.RefModel(desc, tp_size=2, tp_rank=r)
        for k, v in m.shard_weights(weights, desc, 2, r).items():
            s.set_tensor(k, v)
        s.kv_alloc(64)
        slices.append(s)
    got = ref.forward(slices, _step(prompts))
    assert np.abs(got - want).max() < 4e-3 * max(1.0, np.abs(want).max())
    assert (got.argmax(-1) == want.argmax(-1)).all()


WORKER = r'''
import ctypes as C, os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import ref
from tests.conftest import load_pplhip
from tests.test_oracle_hf import desc_from_meta, load_fixture
from tests.test_tp_gloo import _setup, _step

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
m = load_pplhip()
desc, weights, prompts = _setup(os.path.join(sys.argv[1], "tests", "golden"))
sl = ref.RefModel(desc, tp_size=world, tp_rank=rank)
for k, v in m.shard_weights(weights, desc, world, rank).items():
    sl.set_tensor(k, v)
sl.kv_alloc(64)
st = _step(prompts)
L = ref.lib()
T, B, hd = st.num_tokens, st.batch, desc.hidden_dim
H, Hkv, D = desc.num_heads // world, desc.num_kv_heads // world, desc.hidden_dim // desc.num_heads
inter, vl = desc.intermediate_dim // world, desc.vocab_size // world
tok, ss, sp, ci, _ = st._keep
f32 = lambda *s: np.zeros(s, dtype=np.float32)
p = lambda a: a.ctypes.data
rh = lambda a: a.astype(np.float16).astype(np.float32)
def lin(x, name, N, K, out32=0):
    w = sl.get_tensor(name + ".weight", np.float16)
    y = f32(x.shape[0], N)
    L.ref_linear_raw(p(x), p(w), None, 0, 0, x.shape[0], N, K, p(y), out32)
    return y
def allreduce(x):                      # ncclAllReduce(fp16) of pplhip_run
    t = torch.from_numpy(x.copy()); dist.all_reduce(t); return rh(t.numpy())
FUSED = len(sys.argv) > 3 and sys.argv[3] == "fused"
per = (T + world - 1) // world
lo, hi = min(per * rank, T), min(per * rank + per, T)
def allreduce_norm(partial, h, wname):
    """the direct collectives' fused form (csrc/k_comm.hip p2p_allreduce_norm_kernel): reduce-scatter by ROWS (rank r owns rows [r per,
    (r + 1) per)), residual add + RMSNorm on the owned rows only, all-gather of the NORMED rows; h is valid on the owned rows only"""
    parts = [torch.zeros(per, hd) for _ in range(world)]
    padded = np.zeros((per * world, hd), dtype=np.float32); padded[:T] = partial
    mine = torch.zeros(per, hd)
    dist.reduce_scatter(mine, [torch.from_numpy(padded[r * per:(r + 1) * per].copy()) for r in range(world)])
    s_own = rh(mine.numpy())[:hi - lo]                                   # fp16(sum over ranks) of MY rows
    xn_own, h_own = f32(hi - lo, hd), np.ascontiguousarray(h[lo:hi])
    if hi > lo:
        L.ref_rmsnorm(p(h_own), p(np.ascontiguousarray(s_own)), p(sl.get_tensor(wname, np.float16)), desc.norm_eps, hi - lo, hd, p(xn_own), p(h_own))
    h[lo:hi] = h_own                                                     # the residual stream: my rows only
    send = torch.zeros(per, hd); send[:hi - lo] = torch.from_numpy(xn_own)
    got = [torch.zeros(per, hd) for _ in range(world)]
    dist.all_gather(got, send)
    return np.ascontiguousarray(torch.cat(got, 0).numpy()[:T])
rope = f32(desc.max_position, D); L.ref_build_rope_table(p(rope), desc.max_position, D, desc.rope_theta)
h = f32(T, hd); L.ref_embedding(p(tok), p(sl.get_tensor("tok_embeddings.weight", np.float16)), T, hd, p(h))
pending = None
xn = f32(T, hd)
for l in range(desc.num_layers):
    if not FUSED or l == 0:
        L.ref_rmsnorm(p(h), None if pending is None else p(pending), p(sl.get_tensor(f"layers.{l}.attention_norm.weight", np.float16)),
                      desc.norm_eps, T, hd, p(xn), p(h))
    qkv = lin(xn, f"layers.{l}.attention.wqkv", (H + 2 * Hkv) * D, hd)
    L.ref_rope_kv_write(p(qkv), p(rope), C.byref(desc), H, Hkv, D, l, L.ref_kv_ptr(sl.h, 0), L.ref_kv_ptr(sl.h, 1), 64,
                        p(ss), p(sp), p(ci), 0, B)
    att = f32(T, H * D)
    L.ref_attention(p(qkv), C.byref(desc), H, Hkv, D, l, L.ref_kv_ptr(sl.h, 0), L.ref_kv_ptr(sl.h, 1), 64, p(ss), p(sp), p(ci), 0, B, p(att))
    if FUSED:
        xn = allreduce_norm(lin(att, f"layers.{l}.attention.wo", hd, H * D), h, f"layers.{l}.ffn_norm.weight")
    else:
        part = allreduce(lin(att, f"layers.{l}.attention.wo", hd, H * D))
        L.ref_rmsnorm(p(h), p(part), p(sl.get_tensor(f"layers.{l}.ffn_norm.weight", np.float16)), desc.norm_eps, T, hd, p(xn), p(h))
    gu = lin(xn, f"layers.{l}.feed_forward.w13", 2 * inter, hd)
    act = f32(T, inter); L.ref_silu_mul(p(gu), T, inter, p(act))
    if FUSED:   # the collective behind w2 normalises for the NEXT consumer: the next layer's attention norm, or the final norm
        nxt = f"layers.{l + 1}.attention_norm.weight" if l + 1 < desc.num_layers else "norm.weight"
        xn = allreduce_norm(lin(act, f"layers.{l}.feed_forward.w2", hd, inter), h, nxt)
    else:
        pending = allreduce(lin(act, f"layers.{l}.feed_forward.w2", hd, inter))
last = ss[1:] - 1
if FUSED:
    hn = np.ascontiguousarray(xn[last])                                  # the final norm already ran on every row: last-token gather only
else:
    hl, pl, hn = np.ascontiguousarray(h[last]), np.ascontiguousarray(pending[last]), f32(B, hd)
    L.ref_rmsnorm(p(hl), p(pl), p(sl.get_tensor("norm.weight", np.float16)), desc.norm_eps, B, hd, p(hn), None)
mine = torch.from_numpy(lin(hn, "output", vl, hd, 1))
shards = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(shards, mine)          # ncclAllGather + strided copies of pplhip_run
logits = torch.cat(shards, 1).numpy()
if rank == 0:
    np.save(sys.argv[2], logits)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("schedule", ["plain", "fused"])
def test_world_size_2_gloo_matches_unsharded(golden_dir, tmp_path, schedule):
    """plain: all-reduce + replicated (Skip)RMSNorm (RCCL's schedule); fused: the direct collectives' sequence-parallel residual stream
    (reduce-scatter by rows, residual add + norm on the owned rows, all-gather of normed rows, last-token gather from the gathered matrix)"""
    desc, weights, prompts = _setup(golden_dir)
    full = ref.RefModel(desc)
    for k, v in weights.items():
        full.set_tensor(k, v)
    full.kv_alloc(64)
    want = ref.forward([full], _step(prompts))
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "logits.npy"
    env = dict(os.environ, OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                           "127.0.0.1", "--master-port", "29631" if schedule == "plain" else "29633", str(script), ROOT, str(out), schedule],
                          env=env, timeout=600,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    got = np.load(out)
    assert np.abs(got - want).max() < 4e-3 * max(1.0, np.abs(want).max())
    assert (got.argmax(-1) == want.argmax(-1)).all()


def test_bench_control_plane_world_size_2_dry_run():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29632", os.path.join(ROOT, "bench.py"),
                                   "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], env=env, timeout=600,
                                  stderr=subprocess.DEVNULL).decode()
    line = [l for l in out.splitlines() if l.startswith("{")]
    assert len(line) == 1                       # rank 0 prints ONE json line
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "strong"
    assert res["dry_run"] is True and res["unique_id_agreed"] is True and res["config"]["parallelism"] == "tp2"
    # the fields a multi-GPU run is diagnosed from: collectives' mode / self-test verdict / fallbacks / timed all-reduce
    # of the step's own message on the chosen path and on RCCL, the step's schedule, slowest and fastest rank
    co = res["collectives"]
    assert set(co) >= {"mode", "selftest", "schedule", "rccl_communicator", "two_stream_rows", "fallbacks", "allreduce_us"}
    assert set(co["allreduce_us"]) >= {"rows", "bytes", "chosen_path", "rccl"}
    assert "schedule" in res and res["ms_per_step_ranks"]["min"] <= res["ms_per_step_ranks"]["max"]
    # rank r sleeps (r + 1) ms per step: min = the fastest rank's own time (before the closing barrier), max = the headline time
    assert 0.9 <= res["ms_per_step_ranks"]["min"] < 1.9 <= res["ms_per_step_ranks"]["max"] and abs(res["ms_per_step"] - res["ms_per_step_ranks"]["max"]) < 1e-3


def test_bench_self_launches_without_a_launcher():
    """the driver's N = 1 verb with --gpus 2 and NO torch.distributed.run around it: bench.py re-executes itself under the
    launcher (one process per GPU) and rank 0 still prints exactly one JSON line on stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                   "--dry-run"], env=env, timeout=600, stderr=subprocess.DEVNULL).decode()
    line = [l for l in out.splitlines() if l.startswith("{")]
    assert len(line) == 1
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["dry_run"] is True and res["unique_id_agreed"] is True


def test_bench_rank_that_dies_before_the_handle_exchange_gives_one_error_line():
    """a peer that never reaches the IPC-handle exchange must not hang the driver.  Two bench.py processes WITHOUT the
    elastic launcher (nothing tears the survivor down from outside); rank 1 exits right before the exchange.  Rank 0 must exit non-zero
    within the start-up guard's limit, with exactly ONE JSON line that carries "error", the phase it was waiting in and the collectives'
    fallback notes; rank 1's exit code is the hook's."""
    import socket, time
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base.update(OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--dry-run-fail-rank", "1",
           "--startup-timeout", "45"]
    t0 = time.time()
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    out0, err0 = procs[0].communicate(timeout=300)
    procs[1].communicate(timeout=60)
    assert time.time() - t0 < 200
    assert procs[1].returncode == 17
    assert procs[0].returncode not in (0, None)
    lines = [l for l in out0.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out0.decode() + err0.decode()
    res = json.loads(lines[0])
    assert "error" in res and res["value"] == 0.0 and res["n_gpus"] == 2
    assert res["phase"] == "IPC-handle exchange"
    assert "fallbacks" in res["collectives"]
    assert sum(1 for l in err0.decode().splitlines() if l.startswith("[bench] rank 0:")) == 1   # one diagnostic line
