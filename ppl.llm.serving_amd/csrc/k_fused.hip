// One launch, two block roles: decode attention (K8, HBM-bound) for one half of the batch and a tile GEMM (K3/K9, MFMA-bound)
// for the other half.
//
// Why: a pure-decode step at batch 1024 spends 64 % of its time in the attention kernel at the HBM ceiling and 32 % in GEMMs on
// the matrix pipes -- complementary resources, but launched back to back.  Two streams do not help (the small attention workgroups
// back-fill every free wave slot and the GEMM workgroups wait; CU-masked streams cost the attention its bandwidth --
// profiles/overlap_probe*.py).  Inside ONE launch the dispatcher places workgroups in index order, so GEMM workgroups interleaved
// into the index space (every P-th index, P odd so that they spread over all CUs and XCDs) become co-resident with the attention
// workgroups by construction; profiles/probes/fused_roles_probe.hip measured a streaming role + a matrix role at 1.12x the
// streaming role alone (sum: 1.28x).  Every workgroup carries the GEMM role's resources (48 KiB LDS, its VGPRs): 3 workgroups
// = 12 waves per CU, at which the attention role still streams at ~90 % of its full-occupancy rate.
// The roles run the unchanged bodies of k_attn_decode_dev.h and k_gemm_dev.h (results are bit-identical to separate launches).
// Host schedule: pplhip.cc (run_decode_fused).
#include "k_attn_decode_dev.h"
#include "k_gemm_dev.h"
#include <stdlib.h>

namespace pplhip {

struct FusedAttnArgs {
    const uint16_t* qkv;
    KvAddr kv;
    const int64_t* seq_starts;   // already offset to the first request of the chunk
    const int64_t* start_pos;
    const int64_t* cache_indices;
    int64_t max_pages;
    int H, Hkv;
    uint16_t* out;               // row 0 = first request of the chunk
};
struct FusedGemmArgs {
    const uint16_t* x;
    const void* w;
    const uint16_t* scale;
    int64_t M;
    int N, K;
    void* y;
    int64_t ldy;
    int n_tiles, m_tiles, map_mode;
};

template <int QBIT, int EPI, int ST, int WL>
__global__ __launch_bounds__(WL == 5 ? 512 : 256) void fused_attn_gemm_kernel(FusedAttnArgs a, FusedGemmArgs g, int n_attn, int n_gemm, int P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // gemm_dma_lds_bytes<8, ST>()
    const int idx = blockIdx.x;
    const int grp = idx / P;
    if (idx - grp * P == 0 && grp < n_gemm) {  // GEMM role: tile `grp` of the (padded) tile grid, two-stage ring, 4 x 1 waves
        gemm_dma_body<8, EPI, ST, WL>(g.x, g.w, g.scale, g.M, g.N, g.K, g.y, g.ldy, g.n_tiles, g.m_tiles, g.map_mode, g.K / G_BK, nullptr, grp, 0, 1,
                                    smem);
        return;
    }
    const int before = (idx + P - 1) / P;      // GEMM workgroups with a smaller index
    const int aid = idx - (before < n_gemm ? before : n_gemm);
    if (aid >= n_attn) return;
    attn_decode_body<QBIT, 128>(a.qkv, a.kv, a.seq_starts, a.start_pos, a.cache_indices, a.max_pages, a.H, a.Hkv, 1, nullptr, a.out,
                                aid % a.H, (int64_t)(aid / a.H), 0, WL == 5 ? 8 : 4, reinterpret_cast<float*>(smem));
}

bool fused_attn_gemm_supported(int kv_quant_bit, int D, int H, int Hkv, int wq_bit, int K, int N) {
    return (kv_quant_bit == 8 || kv_quant_bit == 0) && D == 128 && H / Hkv < 4 && wq_bit == 8 && K % G_BK == 0 && N % 4 == 0;
}

hipError_t launch_fused_attn_gemm(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int kv_quant_bit, const int64_t* seq_starts,
                                  const int64_t* start_pos, const int64_t* cache_indices, int64_t max_pages, int64_t nb, int H,
                                  int Hkv, uint16_t* attn_out, const uint16_t* x, const void* w, const uint16_t* scale, int64_t M,
                                  int N, int K, void* y, int64_t ldy, bool swiglu) {
    FusedAttnArgs a{qkv, kv, seq_starts, start_pos, cache_indices, max_pages, H, Hkv, attn_out};
    const int n_tiles = (N + G_BN - 1) / G_BN, m_tiles = (int)((M + G_BM - 1) / G_BM);
    static const int ablate = getenv("PPLHIP_GEMM_ABLATE") ? atoi(getenv("PPLHIP_GEMM_ABLATE")) : 0;  // diagnosis only: wrong results
    FusedGemmArgs g{x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles, ablate << 8};
    const int n_gemm = (n_tiles + 7) / 8 * 8 * m_tiles;
    const int64_t n_attn64 = nb * H;
    if (n_attn64 <= 0 || n_attn64 > (1 << 30) || n_gemm <= 0) return hipErrorInvalidValue;
    const int n_attn = (int)n_attn64;
    // GEMM workgroups at every P-th index from the front: they live several times longer than attention workgroups, so they are
    // front-loaded (P = 3; uniform spreading leaves a tail of one GEMM tile after the last attention workgroup -- measured,
    // profiles/fused_microbench.py); odd, so that they visit every XCD / CU residue
    int P = (n_attn + n_gemm) / n_gemm;
    static const int forced_p = getenv("PPLHIP_FUSED_P") ? atoi(getenv("PPLHIP_FUSED_P")) : 3;
    if (forced_p > 0 && forced_p < P) P = forced_p;
    if (P < 1) P = 1;
    if (P > 1 && (P & 1) == 0) --P;
    static const int wl = getenv("PPLHIP_FUSED_WL") ? atoi(getenv("PPLHIP_FUSED_WL")) : 1;  // 5: producer / consumer waves
    const dim3 grid((unsigned)(n_attn + n_gemm)), block(wl == 5 ? 512 : 256);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static const int st = getenv("PPLHIP_FUSED_STAGES") ? atoi(getenv("PPLHIP_FUSED_STAGES")) : 2;  // GEMM role's ring depth
#define FUSED1(QB, E, ST, WL)                                                                                                     \
    do {                                                                                                                          \
        static bool attr_dev[64] = {false};  /* the attribute is per device */                                                   \
        bool& attr = attr_dev[dev & 63];                                                                                          \
        constexpr int lds = gemm_dma_lds_bytes<8, ST>();                                                                          \
        if (!attr) { (void)hipFuncSetAttribute((const void*)fused_attn_gemm_kernel<QB, E, ST, WL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
        hipLaunchKernelGGL((fused_attn_gemm_kernel<QB, E, ST, WL>), grid, block, lds, s, a, g, n_attn, n_gemm, P);                \
    } while (0)
#define FUSED2(QB, E, ST) do { if (wl == 5) FUSED1(QB, E, ST, 5); else FUSED1(QB, E, ST, 1); } while (0)
#define FUSED(QB, E) do { if (st == 3) FUSED2(QB, E, 3); else FUSED2(QB, E, 2); } while (0)
    if (kv_quant_bit == 8) { if (swiglu) FUSED(8, EPI_SWIGLU); else FUSED(8, EPI_F16); }
    else { if (swiglu) FUSED(0, EPI_SWIGLU); else FUSED(0, EPI_F16); }
#undef FUSED1
#undef FUSED2
#undef FUSED
    return hipGetLastError();
}

}  // namespace pplhip
