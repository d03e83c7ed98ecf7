// K3 / K9 / K11 linear layers: y[M,N] = x[M,K] . W[N,K]^T with weight-only quantisation (W8A16 per output
// channel, W4A16 per K-group, or plain fp16 weights).  MFMA-bound at the batch sizes of the headline
// configuration (M = 1024 rows: ~2*M flop per weight byte >> machine balance), so the kernel is an LDS-tiled
// mfma_f32_16x16x32_f16 GEMM with the int8 / int4 weights dequantised to fp16 on their way into LDS.
//
//   block tile 128 (weight rows n) x 128 (activation rows m) x 64 (k); 4 waves as 2 x 2, each 64 x 64 =
//   4 x 4 MFMA tiles; the WEIGHT fragment is the MFMA A operand and the activation fragment the B operand, so
//   the accumulator layout gives each lane 4 consecutive n of one activation row -> 8-byte (fp16) stores and
//   an 8-byte load of the 4 per-channel scales in the epilogue.
//   LDS tiles are [row][64] fp16 with the 16-byte chunk index XOR-swizzled by (row>>1)&7 (128-byte rows: two
//   rows per 256-byte bank window) so ds_read_b128 fragment reads are conflict-free.
//   global -> register -> LDS staging with the next tile's loads in flight during the MFMAs of the current.
//   1-D grid remapped so that the M-tiles of one weight tile run on the same XCD (its L2 then serves the tile
//   to all of them; MI355X_MICROARCH: block b -> XCD b % 8).
// Numerics: fp32 accumulate; W8: y = scale[n] * sum (exact int8 -> fp16); W4: fp16(q * scale) per element
// (one rounding).  Oracle: ref_linear_fwd (oracle/llama_ref.c).
#include "kernels.h"

namespace pplhip {

constexpr int G_BN = 128, G_BM = 128, G_BK = 64;

__device__ __forceinline__ int g_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int WQ, bool OUT32>
__global__ __launch_bounds__(256) void gemm_kernel(const uint16_t* __restrict__ x, const void* __restrict__ wv,
                                                   const uint16_t* __restrict__ scale, int64_t M, int N, int K, int group,
                                                   void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    __shared__ __attribute__((aligned(16))) uint16_t Ws[G_BN * G_BK];
    __shared__ __attribute__((aligned(16))) uint16_t Xs[G_BM * G_BK];

    // XCD-aware tile mapping: XCD x = id % 8 walks weight tiles n = x, x+8, ... and for each all m tiles
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * G_BN;
    const int64_t m0 = (int64_t)mt * G_BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int wn = wave >> 1, wm = wave & 1;

    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    // staging registers
    uint4 xr[4];
    uint4 wr[WQ == 0 ? 4 : (WQ == 8 ? 2 : 1)];
    float wsc = 1.f;  // W4: group scale of this thread's chunk

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
            const int64_t m = m0 + row;
            const int k = k0 + cc * 8;
            xr[i] = (m < M && k < K) ? *reinterpret_cast<const uint4*>(x + m * K + k) : make_uint4(0, 0, 0, 0);
        }
        if constexpr (WQ == 0) {
            const uint16_t* w = reinterpret_cast<const uint16_t*>(wv);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
                const int n = n0 + row, k = k0 + cc * 8;
                wr[i] = (n < N && k < K) ? *reinterpret_cast<const uint4*>(w + (int64_t)n * K + k) : make_uint4(0, 0, 0, 0);
            }
        } else if constexpr (WQ == 8) {
            const int8_t* w = reinterpret_cast<const int8_t*>(wv);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + 256 * i, row = q >> 2, c16 = q & 3;
                const int n = n0 + row, k = k0 + c16 * 16;
                wr[i] = (n < N && k < K) ? *reinterpret_cast<const uint4*>(w + (int64_t)n * K + k) : make_uint4(0, 0, 0, 0);
            }
        } else {
            const uint8_t* w = reinterpret_cast<const uint8_t*>(wv);
            const int row = tid >> 1, c32 = tid & 1;
            const int n = n0 + row, k = k0 + c32 * 32;
            const bool ok = n < N && k < K;
            wr[0] = ok ? *reinterpret_cast<const uint4*>(w + ((int64_t)n * K + k) / 2) : make_uint4(0, 0, 0, 0);
            wsc = ok ? h2f(scale[(int64_t)n * (K / group) + k / group]) : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
            *reinterpret_cast<uint4*>(&Xs[row * G_BK + g_swz(row, cc) * 8]) = xr[i];
        }
        if constexpr (WQ == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
                *reinterpret_cast<uint4*>(&Ws[row * G_BK + g_swz(row, cc) * 8]) = wr[i];
            }
        } else if constexpr (WQ == 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + 256 * i, row = q >> 2, c16 = q & 3;
                const uint32_t w4[4] = {wr[i].x, wr[i].y, wr[i].z, wr[i].w};
#pragma unroll
                for (int hc = 0; hc < 2; ++hc) {
                    h8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int idx = hc * 8 + e;
                        v[e] = (_Float16)(int)(int8_t)(w4[idx >> 2] >> (8 * (idx & 3)));
                    }
                    *reinterpret_cast<uint4*>(&Ws[row * G_BK + g_swz(row, c16 * 2 + hc) * 8]) = __builtin_bit_cast(uint4, v);
                }
            }
        } else {
            const int row = tid >> 1, c32 = tid & 1;
            const uint32_t w4[4] = {wr[0].x, wr[0].y, wr[0].z, wr[0].w};
#pragma unroll
            for (int hc = 0; hc < 4; ++hc) {
                h8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int idx = hc * 8 + e;  // element 0..31; byte idx>>1, low nibble = even element
                    const uint32_t byte = (w4[idx >> 3] >> (8 * ((idx >> 1) & 3))) & 0xffu;
                    const int nib = (idx & 1) ? (int)(byte >> 4) : (int)(byte & 15u);
                    v[e] = (_Float16)((float)(nib - 8) * wsc);
                }
                *reinterpret_cast<uint4*>(&Ws[row * G_BK + g_swz(row, c32 * 4 + hc) * 8]) = __builtin_bit_cast(uint4, v);
            }
        }
    };

    const int ktiles = (K + G_BK - 1) / G_BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < ktiles; ++t) {
        if (t + 1 < ktiles) load_tile((t + 1) * G_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 a[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + l15;
                a[i] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&Ws[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + l15;
                bfr[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&Xs[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (t + 1 < ktiles) {
            store_tile();
            __syncthreads();
        }
    }

    // epilogue: lane holds, for activation row m = .. + l15, weight rows n = .. + kq*4 + r (r = 0..3)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + kq * 4;
        if (n >= N) continue;  // N is a multiple of 4 (checked by the launcher)
        float sc[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (WQ == 8) {
            const uint2 s2 = *reinterpret_cast<const uint2*>(scale + n);
            const h4 sh = __builtin_bit_cast(h4, s2);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = (float)sh[r];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + wm * 64 + j * 16 + l15;
            if (m >= M) continue;
            if constexpr (OUT32) {
                float4 o = make_float4(acc[i][j][0] * sc[0], acc[i][j][1] * sc[1], acc[i][j][2] * sc[2], acc[i][j][3] * sc[3]);
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + m * ldy + n) = o;
            } else {
                h4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc[i][j][r] * sc[r]);
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(yv) + m * ldy + n) = __builtin_bit_cast(uint2, o);
            }
        }
    }
}

hipError_t launch_linear(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int wq_bit, int group,
                         int64_t M, int N, int K, void* y, int64_t ldy, bool out_fp32) {
    if (M == 0) return hipSuccess;
    if (N % 4 || ldy % 4) return hipErrorInvalidValue;
    if (wq_bit == 0 && K % 8) return hipErrorInvalidValue;
    if (wq_bit == 8 && K % 16) return hipErrorInvalidValue;
    if (wq_bit == 4 && (K % 32 || group % 32 || K % group)) return hipErrorInvalidValue;
    const int n_tiles = (N + G_BN - 1) / G_BN;
    const int m_tiles = (int)((M + G_BM - 1) / G_BM);
    const int n_tiles_pad = (n_tiles + 7) / 8 * 8;
    dim3 grid((unsigned)(n_tiles_pad * m_tiles)), block(256);
#define GEMM_CASE(WQ, O32)                                                                                          \
    if (wq_bit == WQ && out_fp32 == O32) {                                                                          \
        hipLaunchKernelGGL((gemm_kernel<WQ, O32>), grid, block, 0, s, x, w, scale, M, N, K, group, y, ldy, n_tiles, \
                           m_tiles);                                                                                \
        return hipGetLastError();                                                                                   \
    }
    GEMM_CASE(0, false) GEMM_CASE(0, true) GEMM_CASE(8, false) GEMM_CASE(8, true) GEMM_CASE(4, false) GEMM_CASE(4, true)
#undef GEMM_CASE
    return hipErrorInvalidValue;
}

}  // namespace pplhip
