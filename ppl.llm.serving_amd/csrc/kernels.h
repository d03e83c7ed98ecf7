// Host-callable launchers of the hand-written gfx950 kernels.  All pointers are device pointers, all
// launches are asynchronous on `stream`.  Return value: hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "k_common.h"

namespace pplhip {

// A split-K GEMM result that has NOT been reduced yet: y[m][n] = fp16( (sum_z ws[(z M + m) N + n]) * scale[n] ) (scale NULL: 1).  The tile
// kernels hand it to the kernel that consumes y anyway -- (Skip)RMSNorm for wo / w2, RoPE + KV write for wqkv -- instead of launching
// splitk_reduce_kernel: one ~5 us kernel and one kernel boundary less per linear of a small-batch step (round 4).  Same arithmetic, same
// summation order, same fp16 rounding as the reduce kernel: bit-identical results.
struct SplitSlabs {
    const float* ws = nullptr;
    int splits = 0;              // 0: nothing deferred (y was written)
    const uint16_t* scale = nullptr;
    int N = 0;
    int64_t M = 0;
};
#ifdef __HIPCC__
// 8 consecutive outputs n .. n + 7 of row m, as fp16-rounded floats
__device__ __forceinline__ void slab_load8(const SplitSlabs& sl, int64_t m, int n, float* o) {
    // every slab's 32 bytes are requested before the first sum (one memory latency, not `splits` of them: a consumer row is one
    // workgroup); slabs past `splits` re-read slab 0 and are not added.  Summation order z = 0, 1, ... like splitk_reduce_kernel.
    const float* p = sl.ws + m * sl.N + n;
    const int64_t zs = sl.M * sl.N;
    float4 c[8], d[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) {
        const float* q = p + (z < sl.splits ? z : 0) * zs;
        c[z] = *reinterpret_cast<const float4*>(q);
        d[z] = *reinterpret_cast<const float4*>(q + 4);
    }
    float4 a = c[0], b = d[0];
#pragma unroll
    for (int z = 1; z < 8; ++z) {
        if (z < sl.splits) {
            a.x += c[z].x; a.y += c[z].y; a.z += c[z].z; a.w += c[z].w;
            b.x += d[z].x; b.y += d[z].y; b.z += d[z].z; b.w += d[z].w;
        }
    }
    if (sl.scale) {
        const h8 sh = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(sl.scale + n));
        a.x *= (float)sh[0]; a.y *= (float)sh[1]; a.z *= (float)sh[2]; a.w *= (float)sh[3];
        b.x *= (float)sh[4]; b.y *= (float)sh[5]; b.z *= (float)sh[6]; b.w *= (float)sh[7];
    }
    o[0] = round_h(a.x); o[1] = round_h(a.y); o[2] = round_h(a.z); o[3] = round_h(a.w);
    o[4] = round_h(b.x); o[5] = round_h(b.y); o[6] = round_h(b.z); o[7] = round_h(b.w);
}
#endif

// ---- k_elem.hip -------------------------------------------------------------------------------
hipError_t launch_embedding(hipStream_t s, const int64_t* token_ids, const uint16_t* table, int64_t T, int hidden,
                            uint16_t* out);
// out[r] = rmsnorm(x[src(r)] (+ skip[src(r)])) * w.  gather_seq_starts != NULL: src(r) = seq_starts[r+1]-1
// (last-token gather of K11), else src(r) = r.  residual_out (optional) receives fp16(x + skip) at row r.
hipError_t launch_rmsnorm(hipStream_t s, const uint16_t* x, const uint16_t* skip, const uint16_t* w, float eps,
                          int64_t rows, int hidden, const int64_t* gather_seq_starts, uint16_t* out,
                          uint16_t* residual_out, int8_t* qout = nullptr, float* sx = nullptr,
                          const SplitSlabs* skip_slabs = nullptr);  // skip_slabs: the skip operand as unreduced split-K slabs
hipError_t launch_silu_mul(hipStream_t s, const uint16_t* gate_up, int64_t T, int inter, uint16_t* out);
// out[r] = x[seq_starts[r + 1] - 1] (last-token gather of K11 when the final norm already ran on every row: fused tensor-parallel norm)
hipError_t launch_gather_last_rows(hipStream_t s, const uint16_t* x, const int64_t* seq_starts, int64_t B, int hidden, uint16_t* out);

// ---- k_rope_kv.hip ----------------------------------------------------------------------------
hipError_t launch_rope_kv_write(hipStream_t s, uint16_t* qkv, const float* cos_sin, const KvAddr& kv, int quant_bit,
                                int quant_group, const int64_t* seq_starts, const int64_t* start_pos,
                                const int64_t* cache_indices, int64_t max_pages, int64_t B, int64_t t0, int64_t T, int H,
                                int Hkv, int D, const SplitSlabs* qkv_slabs = nullptr);  // token rows [t0, t0 + T) of the step's B requests;
                                                   // qkv_slabs: the rows come from unreduced split-K slabs of a [T, N] launch (slab row = token row - t0); rotated q -> qkv

// ---- k_attn_decode.hip ------------------------------------------------------------------------
// rows [0, nb) of the batch are single-token queries; q row of request b is qkv row seq_starts[b].
// split > 1 uses `workspace` (fp32 [nb, H, split, D+2]).
size_t attn_decode_workspace_bytes(int64_t nb, int H, int D, int split);
hipError_t launch_attn_decode(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit,
                              const int64_t* seq_starts, const int64_t* start_pos, const int64_t* cache_indices,
                              int64_t max_pages, int64_t nb, int H, int Hkv, int D, int64_t max_kv_len, int split,
                              int threads, float* workspace, uint16_t* out, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

// ---- k_attn_prefill.hip -----------------------------------------------------------------------
// requests [b0, B): causal attention of their new tokens over the cache [0, start_pos + seqlen).
hipError_t launch_attn_prefill(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit,
                               const int64_t* seq_starts, const int64_t* start_pos, const int64_t* cache_indices,
                               int64_t max_pages, int64_t b0, int64_t B, int H, int Hkv, int D, int64_t max_seq_len,
                               uint16_t* out, int64_t max_kv_len = 0, float* ws = nullptr, size_t ws_bytes = 0, int64_t row0 = 0,
                               int64_t nrows = 0);  // max_kv_len .. nrows: optional, enable the split-KV form for short suffixes (k_attn_prefill32.hip)

// ---- k_attn_prefill32.hip ---------------------------------------------------------------------
// the same operation for head_dim 128 on the 32-row wave tile (mfma_f32_32x32x16_f16, 64-key tiles, one barrier per tile)
hipError_t launch_attn_prefill32(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit, const int64_t* seq_starts,
                                 const int64_t* start_pos, const int64_t* cache_indices, int64_t max_pages, int64_t b0, int64_t B,
                                 int H, int Hkv, int D, int64_t max_seq_len, uint16_t* out, int64_t max_kv_len = 0, float* ws = nullptr,
                                 size_t ws_bytes = 0, int64_t row0 = 0, int64_t nrows = 0);

// ---- k_attn_decode_gqa.hip --------------------------------------------------------------------
// grouped-query decode (4 <= H/Hkv <= 16): MFMA kernel, one block per (request, KV head[, split]); same workspace layout
// and reduce kernel as launch_attn_decode.  t0 / t1: optional start / stop events of the kernel's own dispatch packet.
bool attn_decode_gqa_supported(int quant_bit, int H, int Hkv, int D);
hipError_t launch_attn_decode_gqa(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit,
                                  const int64_t* seq_starts, const int64_t* start_pos, const int64_t* cache_indices,
                                  int64_t max_pages, int64_t nb, int H, int Hkv, int D, int split, float* workspace,
                                  uint16_t* out, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

// ---- k_gemm.hip -------------------------------------------------------------------------------
// y[M,N] = x[M,K] . W[N,K]^T (+ per-channel / per-group scales).  wq_bit 0/8/4.  out_fp32: y is float.
// ldy = row stride of y in elements.  ws (optional, fp32 scratch) enables split-K at small M.
// swiglu: the weight rows are interleaved (gate_i, up_i) and y[M, N/2] = silu(gate) * up (fused K10).
// W8A16 128 (m) x 384 (n) tile kernel (k_gemm_wide.hip); K % 64 == 0, N % 4 == 0; epi 0 fp16 / 1 fp32 / 2 fused SwiGLU
hipError_t launch_linear_w8_wide(hipStream_t s, const uint16_t* x, const int8_t* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                                 int64_t ldy, int epi, int nc);
// W4A16 (group 128) on 128 (m) x 64 (n) tiles, no K slabs (k_gemm_pc.hip): a few hundred rows; epi 0 fp16 / 2 fused SwiGLU
bool linear_w4_pc_supported(int group, int64_t M, int N, int K, const void* x, const void* w, const void* scale, const void* y, int64_t ldy, int epi);
hipError_t launch_linear_w4_pc(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                               int64_t ldy, int epi);
int linear_w8_wide_waves(int64_t M, int N);  // 12 when the 128 x 384 tiles fill rounds of 256 blocks well enough, else 0
hipError_t launch_linear(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int wq_bit, int group,
                         int64_t M, int N, int K, void* y, int64_t ldy, bool out_fp32, float* ws = nullptr, size_t ws_bytes = 0,
                         bool swiglu = false, SplitSlabs* defer = nullptr);  // defer: a split-K launch leaves its slabs unreduced there
// y = (sum of `splits` fp32 slabs [M][N] at ws) * scale (NULL: 1), epilogue epi (0 fp16 / 1 fp32 / 2 fused SwiGLU)
hipError_t launch_splitk_reduce(hipStream_t s, const float* ws, int splits, int64_t M, int N, const uint16_t* scale, void* y, int64_t ldy, int epi);
// ---- k_gemv.hip: streaming GEMV, 1 <= M <= 4 (whole 1-KiB row pieces per wave-load; VALU dot products) ----------------------------
int gemv_stream_max_m(int wq_bit, int group, int N, int K);  // largest M the kernel takes for this shape (0: none)
hipError_t launch_gemv_stream(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int wq_bit, int group, int64_t M, int N,
                              int K, void* y, int64_t ldy, int epi);
// ---- k_gemm_i8.hip: online_i8i8 (W8A8) ------------------------------------------------------------
// per-token int8 activations: q [M, ldq] (columns K..ldq-1 zeroed), sx [M] = max|x| / 127
hipError_t launch_quant_act(hipStream_t s, const uint16_t* x, int64_t M, int K, int64_t ldx, int8_t* q, int64_t ldq, float* sx);
// per-output-row int8 weights from an fp16 [N, K] matrix: q [N, ldq], scale [N] fp16
hipError_t launch_quant_weight(hipStream_t s, const uint16_t* w, int N, int K, int8_t* q, int64_t ldq, uint16_t* scale);
// y[M,N] = fp16/fp32( (sum_k xq * w) * sx[m] * scale[n] ); K = row stride of xq and w (K % 16 == 0)
hipError_t launch_linear_i8(hipStream_t s, const int8_t* xq, const float* sx, const int8_t* w, const uint16_t* scale, int64_t M, int N,
                            int K, void* y, int64_t ldy, bool out_fp32, bool swiglu);
// dst row r = src row perm(r): r even -> r/2 (gate), r odd -> half + r/2 (up).  row_bytes % 4 == 0.
hipError_t launch_interleave_rows(hipStream_t s, const void* src, void* dst, int rows, int64_t row_bytes);

// ---- k_comm.hip ------------------------------------------------------------------------------
// direct (all-links) tensor-parallel collectives over peer-mapped exchange regions; see the file header.
constexpr int P2P_MAX_RANKS = 8;
constexpr int P2P_MAX_BLOCKS = 64;
// exchange-region layout (bytes, identical on every rank): flag words first, data buffers after P2P_DATA_START
constexpr size_t P2P_FLAGS_START = 0;                                         // uint32 [P2P_MAX_BLOCKS][P2P_MAX_RANKS]
constexpr size_t P2P_FLAGS_MID = (size_t)P2P_MAX_BLOCKS * P2P_MAX_RANKS * 4;  // uint32 [P2P_MAX_BLOCKS][P2P_MAX_RANKS]
// two channels (flag sets): collectives of different channels may be in flight at the same time (the two half-batches of a two-stream
// decode step, pplhip.cc run_launches); channel c's words sit P2P_CHANNEL_FLAG_BYTES x c further on
constexpr int P2P_CHANNELS = 2;
constexpr size_t P2P_CHANNEL_FLAG_BYTES = (size_t)2 * P2P_MAX_BLOCKS * P2P_MAX_RANKS * 4;
constexpr size_t P2P_DATA_START = 8192;
static_assert(P2P_CHANNELS * P2P_CHANNEL_FLAG_BYTES <= P2P_DATA_START, "flag words of every channel in front of the data");
struct P2pPeers {
    char* base[P2P_MAX_RANKS];  // exchange region of every rank as mapped in this process (base[me] is the local one)
};
// all-reduce(sum) of fp16[count] at byte offset data_off of every rank's region (count % 4 == 0); scratch_off: a region
// offset with room for ceil(count / n) fp16 that no other collective in flight uses
hipError_t launch_p2p_allreduce(hipStream_t s, const P2pPeers& peers, int me, int n, size_t data_off, size_t scratch_off, int64_t count,
                                uint32_t epoch, uint64_t timeout_ticks, uint32_t* status, int channel = 0);   // epochs count per channel
// The same all-reduce fused with the residual add + RMSNorm that consumes it (sequence-parallel residual stream, k_comm.hip): `rows` rows of
// fp16 [rows, hidden] partial sums at data_off of every rank's region; rank `me` owns rows [me per, (me + 1) per), per = ceil(rows / n): on those
// it computes s = fp16(sum over ranks), h = fp16(h + s) (h: LOCAL residual rows, only the owned ones are read or written) and
// y = rmsnorm(h) * w; every rank ends with all rows of y in the local matrix xn.  scratch_off: room for per rows.  n == 1: local reference
// form (no peers, no barriers, every row owned; reads data_off of the own region).  hidden <= P2P_NORM_MAX_HIDDEN, hidden % 8 == 0
constexpr int P2P_NORM_MAX_HIDDEN = 8192;
hipError_t launch_p2p_allreduce_norm(hipStream_t s, const P2pPeers& peers, int me, int n, size_t data_off, size_t scratch_off, int64_t rows,
                                     int hidden, uint16_t* h, const uint16_t* w, float eps, uint16_t* xn, uint32_t epoch, uint64_t timeout_ticks,
                                     uint32_t* status, int channel = 0);
// rows x row_bytes at src_off of every rank's region (rank r's column block) -> columns [r * row_bytes, ...) of the local
// [rows, dst_row_bytes] matrix dst (ordinary memory)
hipError_t launch_p2p_allgather(hipStream_t s, const P2pPeers& peers, int me, int n, size_t src_off, void* dst, int64_t rows,
                                int64_t row_bytes, int64_t dst_row_bytes, uint32_t epoch, uint64_t timeout_ticks, uint32_t* status);

// stream hand-off through a device-memory epoch (k_comm.hip): signal = one-thread kernel after the producer; wait = one-wave kernel
// in front of the consumer (bounded spin; optionally raises a second flag when the wait is over)
hipError_t launch_handoff_signal(hipStream_t s, uint32_t* flag, uint32_t epoch);
hipError_t launch_handoff_wait(hipStream_t s, const uint32_t* wait_flag, uint32_t wait_epoch, uint32_t* done_flag, uint32_t done_epoch,
                               uint64_t timeout_ticks, uint32_t* status);

// self-test patterns: halfs[i] = value(i, g, round), floats[i] = value(i, g, round + 2)
float p2p_pattern_value(int64_t i, int g, int round);
hipError_t launch_p2p_pattern(hipStream_t s, uint16_t* halfs, int64_t cnt, float* floats, int64_t gcnt, int g, int round);

// ---- k_sample.hip -----------------------------------------------------------------------------
hipError_t launch_sample_greedy(hipStream_t s, const float* logits, const float* temperatures, int batch, int vocab,
                                int stride, int32_t* out_tok, float* out_logprob);
size_t sample_topk_workspace_bytes(int batch, int vocab, int top_k);
hipError_t launch_sample_topk_topp(hipStream_t s, const float* logits, const float* temperatures, const float* top_p,
                                   const float* rnd, int batch, int vocab, int stride, int top_k, float default_top_p,
                                   void* workspace, int32_t* out_tok, float* out_logprob);
hipError_t launch_penalty(hipStream_t s, float* logits, const float* temperatures, const float* rep,
                          const float* presence, const float* frequency, const int64_t* batch_slots,
                          const int64_t* token_inputs, const int64_t* seq_starts, const int64_t* start_pos, int batch,
                          int vocab, int stride, int decoding_batches, uint16_t* count_map);

// ---- synth.hip --------------------------------------------------------------------------------
// kinds as in oracle/llama_ref.c: 0 fp16 uniform(-amp,amp), 1 int8, 2 packed int4 (n bytes), 3 scale, 4 norm
hipError_t launch_synth_fill(hipStream_t s, int kind, uint64_t seed, uint32_t tensor_id, uint32_t stream_id, float amp,
                             uint64_t n, void* out);

}  // namespace pplhip
