// Streaming GEMV for 1 <= M <= 4 activation rows (single-stream and few-stream decode; round 4).
//
// The MFMA skinny kernel of k_gemm.hip fetches weights in the MFMA A-operand shape -- a wave-load is 16 rows x 64 bytes -- and streams the
// 7B layer at 2.9 TB/s (W8A16), 36 % of the HBM peak: every request is half a 128-byte line of a different row.  At M <= 4 the matrix
// unit buys nothing (one row of activations against every weight: 2 M flop per weight byte), so this kernel is a plain stream:
//   * a wave-load is ONE KiB of ONE weight row (16 contiguous bytes per lane): whole DRAM lines, the access pattern of the decode attention
//     kernel (6.4 TB/s);
//   * block = 4 or 8 waves, split nwk (along K) x nwr (along rows): wave (wk, wr) owns the 1-KiB pieces p == wk (mod nwk) of its 8 or 16
//     weight rows, so the activations it needs (16 / 32 / 8 elements per lane and piece for int8 / int4 / fp16 weights) sit in registers
//     while it walks its rows; short rows (few pieces) give the spare waves rows of their own, small matrices get 8-row waves so that every
//     CU still has several (launch_gemv_stream);
//   * 8 rows in flight per wave (8 KiB) before the first use; partial sums of 8 rows are folded across the 64 lanes by a halving
//     butterfly (10 cross-lane operations per 8 rows), across the waves along K through LDS; scales / SwiGLU / rounding by the first threads.
// Default range (gemv_stream_max_m): M <= 4; int8 weights with K % 128 == 0 only M <= 2 -- from 3 rows the tile kernel of k_gemm.hip is faster.
// Numerics as ref_linear (oracle/llama_ref.c): fp32 products and sums of fp16 x {int8 exact, fp16(nibble x group scale), fp16}; the
// per-channel W8 scale multiplies the finished sum.  Summation order differs from the oracle's (as in every GEMM kernel here).
#include <stdlib.h>
#include "k_gemm_dev.h"

namespace pplhip {

namespace {

constexpr int GV_NW = 8, GV_RB = 16, GV_U = 8;  // most waves per block, most weight rows per wave, rows in flight per wave

template <int WQ>
struct GvCfg {
    static constexpr int KL = WQ == 8 ? 16 : (WQ == 4 ? 32 : 8);   // k elements in a lane's 16 bytes
    static constexpr int KP = KL * 64;                              // k elements per 1-KiB piece
};

// 16 bytes of weights -> KL fp16 values (exact for int8; fp16(nibble x scale) for int4)
template <int WQ>
__device__ __forceinline__ void gv_unpack(const uint4& raw, h2 sc, h8* out) {
    if constexpr (WQ == 8) {
        out[0] = cvt_i8x8_f16(make_uint2(raw.x, raw.y));
        out[1] = cvt_i8x8_f16(make_uint2(raw.z, raw.w));
    } else if constexpr (WQ == 4) {
        out[0] = cvt_i4x8_f16(raw.x, sc);
        out[1] = cvt_i4x8_f16(raw.y, sc);
        out[2] = cvt_i4x8_f16(raw.z, sc);
        out[3] = cvt_i4x8_f16(raw.w, sc);
    } else {
        out[0] = __builtin_bit_cast(h8, raw);
    }
}

// (Round 4 also had a form with the consuming RMSNorm folded in -- bit-identical and measured slower, batch 1 2.42 -> 2.62 ms: every one of
// the ~768 short blocks paid the row reduction in front of its first product; removed in round 5.)
template <int WQ, int M, int EPI, int NW>  // NW waves per block (4 or 8)
__global__ __launch_bounds__(NW * 64) void gemv_stream_kernel(const uint16_t* __restrict__ x, const void* __restrict__ wv,
                                                              const uint16_t* __restrict__ scale, int N, int K, int group,
                                                              void* __restrict__ yv, int64_t ldy, int nwk_log2, int npw, int nb) {
    using C = GvCfg<WQ>;
    constexpr int KL = C::KL, KP = C::KP, NV = KL / 8;
    __shared__ float red[NW][GV_RB][M];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves along K (a row's pieces p == wk (mod nwk)) x waves along rows (8 nb rows each): short rows (few pieces) give the spare waves
    // rows of their own instead of leaving them idle.  npw = pieces per wave and row, nb = batches of 8 rows per wave (1 or 2: fewer rows per
    // wave when the matrix is small, so that every CU still gets several waves)
    const int nwk = 1 << nwk_log2, wk = wave & (nwk - 1), wr = wave >> nwk_log2, nwr = NW >> nwk_log2;
    const int rpw = nb * GV_U;                                  // rows per wave
    const int n0 = (blockIdx.x * nwr + wr) * rpw;
    const int64_t row_bytes = (int64_t)K * (WQ == 0 ? 16 : WQ) / 8;
    const char* wbase = reinterpret_cast<const char*>(wv);

    for (int bt = 0; bt < nb; ++bt) {
        float tot[M];  // this lane's folded sums of row (lane & 7) of the batch
#pragma unroll
        for (int m = 0; m < M; ++m) tot[m] = 0.f;
        for (int pp = 0; pp < npw; ++pp) {
            const int k0 = (wk + pp * nwk) * KP + lane * KL;       // first k of this lane's chunk
            const bool live = k0 < K;                              // (K is a multiple of KL: a chunk is inside the row or past it)
            uint4 raw[GV_U];
            h2 sc[GV_U];
            const int64_t coff = (int64_t)k0 * (WQ == 0 ? 16 : WQ) / 8;
#pragma unroll
            for (int u = 0; u < GV_U; ++u) {
                int n = n0 + bt * GV_U + u;
                if (n >= N) n = N - 1;
                raw[u] = live ? kv_stream_load(reinterpret_cast<const uint4*>(wbase + (int64_t)n * row_bytes + coff)) : make_uint4(0, 0, 0, 0);
                if constexpr (WQ == 4) {
                    const _Float16 s1 = live ? __builtin_bit_cast(_Float16, scale[(int64_t)n * (K / group) + k0 / group]) : (_Float16)0;
                    sc[u] = h2{s1, s1};
                } else {
                    sc[u] = h2{(_Float16)1, (_Float16)1};
                }
            }
            // activations of the chunk, packed fp16 (L1 / L2 resident: M rows of K halfs)
            h8 xv[M][NV];
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    xv[m][v] = live ? __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(x + (int64_t)m * K + k0 + v * 8)) : h8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            float part[GV_U][M];
#pragma unroll
            for (int u = 0; u < GV_U; ++u) {
                h8 wf[NV];
                gv_unpack<WQ>(raw[u], sc[u], wf);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    float a = 0.f;
#pragma unroll
                    for (int v = 0; v < NV; ++v)
#pragma unroll
                        for (int e = 0; e < 8; e += 2)   // v_dot2_f32_f16: two exact products added into the fp32 sum
                            a = __builtin_amdgcn_fdot2(h2{wf[v][e], wf[v][e + 1]}, h2{xv[m][v][e], xv[m][v][e + 1]}, a, false);
                    part[u][m] = a;
                }
            }
            // fold 8 row sums across the 64 lanes: three halving steps (lane bit b decides which half of the rows a lane keeps), then a
            // plain butterfly over the remaining three lane bits; lane l ends with the sum of row (l & 7)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v4[4], v2[2], v1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float keep = (lane & 4) ? part[i + 4][m] : part[i][m], give = (lane & 4) ? part[i][m] : part[i + 4][m];
                    v4[i] = keep + __shfl_xor(give, 4, 64);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float keep = (lane & 2) ? v4[i + 2] : v4[i], give = (lane & 2) ? v4[i] : v4[i + 2];
                    v2[i] = keep + __shfl_xor(give, 2, 64);
                }
                {
                    const float keep = (lane & 1) ? v2[1] : v2[0], give = (lane & 1) ? v2[0] : v2[1];
                    v1 = keep + __shfl_xor(give, 1, 64);
                }
                v1 += __shfl_xor(v1, 8, 64);
                v1 += __shfl_xor(v1, 16, 64);
                v1 += __shfl_xor(v1, 32, 64);
                tot[m] += v1;   // row inside the batch: bit 2 of the lane picked i + 4, bit 1 i + 2, bit 0 i + 1 -> row = lane & 7
            }
        }
        if (lane < 8) {
#pragma unroll
            for (int m = 0; m < M; ++m) red[wave][bt * GV_U + lane][m] = tot[m];
        }
    }
    __syncthreads();
    // final sums over the waves along K, scale, epilogue: one thread per (row of the block, activation row)
    const int rows = nwr * rpw, nb0 = blockIdx.x * rows;
    if constexpr (EPI == EPI_SWIGLU) {
        if (tid < (rows / 2) * M) {
            const int pr = tid % (rows / 2), m = tid / (rows / 2);
            const int rl = 2 * pr, w0 = (rl / rpw) * nwk, rr = rl % rpw;
            const int n = nb0 + rl;
            if (n + 1 < N) {
                float g = 0.f, u = 0.f;
                for (int w = 0; w < nwk; ++w) { g += red[w0 + w][rr][m]; u += red[w0 + w][rr + 1][m]; }
                if constexpr (WQ == 8) { g *= h2f(scale[n]); u *= h2f(scale[n + 1]); }
                const float gr = round_h(g), ur = round_h(u);
                reinterpret_cast<uint16_t*>(yv)[(int64_t)m * ldy + (n >> 1)] = f2h(gr / (1.0f + __expf(-gr)) * ur);
            }
        }
    } else {
        if (tid < rows * M) {
            const int rl = tid % rows, m = tid / rows;
            const int w0 = (rl / rpw) * nwk, rr = rl % rpw;
            const int n = nb0 + rl;
            if (n < N) {
                float v = 0.f;
                for (int w = 0; w < nwk; ++w) v += red[w0 + w][rr][m];
                if constexpr (WQ == 8) v *= h2f(scale[n]);
                if constexpr (EPI == EPI_F32) reinterpret_cast<float*>(yv)[(int64_t)m * ldy + n] = v;
                else reinterpret_cast<uint16_t*>(yv)[(int64_t)m * ldy + n] = f2h(v);
            }
        }
    }
}

}  // namespace

// largest M this kernel takes; 0 when the shape does not fit it (the caller then uses the tile kernels)
int gemv_stream_max_m(int wq_bit, int group, int N, int K) {
    // default: 4 rows; 2 for int8 weights that the half-height tile kernel of k_gemm.hip takes (K % 128 == 0) -- with 16-row activation
    // sub-tiles and one block per CU it overtakes the GEMV from 3 rows (7B decode step at batch 2 / 3 / 4 / 5: GEMV 2.48 / 2.75 / 3.00 / -,
    // tiles - / see DESIGN.md / 2.68 / 2.74 ms)
    static const int env_m = getenv("PPLHIP_GEMV_STREAM_MAX_M") ? atoi(getenv("PPLHIP_GEMV_STREAM_MAX_M")) : -1;
    const int max_m = env_m >= 0 ? env_m : (wq_bit == 8 && K % 128 == 0 ? 2 : 4);
    const int kl = wq_bit == 8 ? 16 : (wq_bit == 4 ? 32 : 8);
    if (K % kl) return 0;
    if (wq_bit == 4 && (group % 32 || K % group)) return 0;
    const int pieces = (K / kl + 63) / 64;
    if (pieces > 3 * GV_NW) return 0;  // at most three pieces per wave and row (K <= 24576 int8)
    (void)N;
    return max_m > 4 ? 4 : max_m;
}

hipError_t launch_gemv_stream(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int wq_bit, int group, int64_t M, int N,
                              int K, void* y, int64_t ldy, int epi) {
    const int kl = wq_bit == 8 ? 16 : (wq_bit == 4 ? 32 : 8);
    const int pieces = (K / kl + 63) / 64;
    int nwk_log2 = 0;
    while ((1 << nwk_log2) < pieces && nwk_log2 < 3) ++nwk_log2;
    const int nwk = 1 << nwk_log2, npw = (pieces + nwk - 1) / nwk;
    const int nw = nwk <= 4 ? 4 : 8;                                   // small blocks: the grid balances over the 256 CUs
    // 16 rows per wave when that still leaves >= 8 waves per CU, else 8
    const int64_t waves16 = (int64_t)((N + 15) / 16) * nwk;
    const int nb = waves16 >= 2048 ? 2 : 1;
    const int rows = (nw / nwk) * nb * GV_U;
    dim3 grid((unsigned)((N + rows - 1) / rows)), block(nw * 64);
#define GV_L(WQ, MM, E, W) hipLaunchKernelGGL((gemv_stream_kernel<WQ, MM, E, W>), grid, block, 0, s, x, w, scale, N, K, group, y, ldy, nwk_log2, npw, nb)
#define GV_P(WQ, MM, E) do { if (nw == 4) GV_L(WQ, MM, E, 4); else GV_L(WQ, MM, E, 8); } while (0)
#define GV_E(WQ, MM) do { if (epi == EPI_F32) GV_P(WQ, MM, EPI_F32); else if (epi == EPI_F16) GV_P(WQ, MM, EPI_F16); else GV_P(WQ, MM, EPI_SWIGLU); } while (0)
#define GV_M(WQ) do { if (M == 1) GV_E(WQ, 1); else if (M == 2) GV_E(WQ, 2); else if (M == 3) GV_E(WQ, 3); else GV_E(WQ, 4); } while (0)
    if (M < 1 || M > 4 || npw > 3) return hipErrorInvalidValue;
    if (wq_bit == 8) GV_M(8); else if (wq_bit == 4) GV_M(4); else if (wq_bit == 0) GV_M(0); else return hipErrorInvalidValue;
#undef GV_M
#undef GV_E
#undef GV_P
#undef GV_L
    return hipGetLastError();
}

}  // namespace pplhip
