"""The W8A16 GEMMs of ONE RANK's slice of the headline decode step under tensor parallelism (LLaMA-2-7B, batch 1024, TP 4 / 8: what
`bench.py --emulate-tp N` and a real `--gpus N` run launch 32 times per step), at their own launch shapes against the oracle's
ref_linear_raw: N = 1536 / 2752 / 3072 leave the 128 x 128 grid with idle CUs, so these shapes take the k-split tiles of
csrc/k_gemm_ks.hip (128 x 96 and 64 x 96, four multiplying waves that split every K tile by k-step; round 6) -- plus ragged row counts,
channel counts that end inside a tile, short K loops, and the shapes next to them that stay on the other kernels.
Reference work: /root/reference/src/engine/llm_engine.cc:113-116 (the matmuls are ppl.nn's, absent from the tree)."""
import pytest

from tests.test_gpu_config34_shape import _linear_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,M,N,K,swiglu", [
    ("tp8_wqkv", 1024, 1536, 4096, False), ("tp8_wo", 1024, 4096, 512, False), ("tp8_w13", 1024, 2 * 1376, 4096, True),
    ("tp8_w2", 1024, 4096, 1408, False), ("tp4_wqkv", 1024, 3072, 4096, False), ("tp4_wo", 1024, 4096, 1024, False),
    ("tp4_w13", 1024, 2 * 2752, 4096, True), ("tp4_w2", 1024, 4096, 2752 + 64 - 2752 % 64, False)])
def test_7b_tensor_parallel_slice_gemms_at_batch_1024(name, M, N, K, swiglu):
    _linear_case(f"slice_gemm_{name}_m1024", 8, M, N, K, swiglu, N + K)


@pytest.mark.parametrize("M,N,K,swiglu", [
    (1000, 1536, 4096, False),     # rows end inside the last 64-row tile
    (700, 2752, 1024, True),       # 128 x 96 tiles, rows end inside a tile, channels end inside the last tile (2752 = 28 x 96 + 64)
    (520, 3072, 2048, False),      # 64 x 96 tiles at the lower end of the row window
    (1024, 1540, 1024, False),     # four channels in the last tile
    (513, 1444, 1088, False),      # 17 K tiles: the ring drains with an odd count
    (1024, 1536, 960, False),      # K below the window: the other kernels
    (1100, 1536, 4096, False)])    # rows above the window
def test_k_split_tiles_edges(M, N, K, swiglu):
    _linear_case(f"ks_edge_m{M}_n{N}_k{K}{'_swiglu' if swiglu else ''}", 8, M, N, K, swiglu, M + N + K)
