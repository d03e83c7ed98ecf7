// W4A16 (group 128) linear layers at a few hundred rows (round 5): converter waves beside MFMA waves.
//
// Why another tile kernel.  At M = 256 (config 4: LLaMA-2-70B W4A16, TP 8, batch 256) the 128 x 128 ring kernel of k_gemm_dev.h ran the
// layer's four GEMMs at 19 % of the MFMA peak (profiles/r04_late_experiments.md 3): 112 / 20 / 128 output tiles do not fill 256 CUs, so
// every launch was cut into K slabs (fp32 partial sums of the whole output written and read back: 2.8 x the algorithmic bytes on w13,
// 4 x on wo) and inside a wave each K tile was a serial wait -> barrier -> DMA issue -> int4 conversion -> fragment reads -> MFMA chain.
// This kernel changes the decomposition and the roles:
//   * block tile 128 (m) x 64 (n): 224 / 256 / 256 blocks for w13 / wo / w2 of that layer -- one per CU, NO K slabs, outputs written once
//     in fp16 (wqkv, 40 tiles, keeps slabs that RoPE + KV write sums anyway);
//   * 8 waves, two per SIMD with different jobs.  Waves 4..7 (producers / converters) own the data movement: the activation tiles go
//     global -> LDS by DMA into a 4-stage ring; the int4 weights of a 128-deep super-tile (= one quantisation group; 64 rows x 64 bytes,
//     16 rows per wave, a lane's 16 bytes = 32 nibbles of one row) and their group scales go global -> LDS by DMA into a raw ring five
//     super-tiles deep (HBM latency is hidden by depth, not by occupancy), are read back by the lane that fetched them, converted ONCE
//     per block to fp16(q * scale) and written as MFMA-ready fp16 rows into a double-buffered weight image;
//   * waves 0..3 (consumers, 2 (n) x 2 (m), each 32 (n) x 64 (m)) do nothing but read fragments (12 ds_read_b128 per K tile, a whole tile
//     ahead of their use, double-buffered in registers) and issue v_mfma_f32_32x32x16_f16: no conversion, no address arithmetic, no
//     vmcnt in their instruction stream.  One s_barrier per 64-deep K tile.
// LDS (121 KiB, one block per CU): X ring 4 x 16 KiB [128 rows][64 fp16], 16-byte chunk q of row r at position q ^ ((r >> 1) & 7);
// fp16 weights 2 super-tiles x 2 K tiles x 8 KiB [64 rows][64 fp16], chunk q of row r of K tile t at q ^ ((r >> 1) & 7) ^ 2 (t & 1)
// (conflict-free fragment reads AND conflict-free converter writes); raw ring 5 x 4 KiB; scale ring 5 x 1 KiB.
// Every memory operation of the producers is an LDS-DMA issued from inline asm (no VGPR destinations: nothing for hipcc to mis-wait),
// counted with s_waitcnt vmcnt(N); N is derived below from the fixed issue order, and the tail re-issues clamped loads so that the
// counts stay static.
// Numerics: the dequantised weight is the fp16 number fp16(q * scale) (cvt_i4x8_f16, as every W4 kernel here), fp32 accumulation in k
// order, one rounding of the sum to fp16.  Oracle: ref_linear_fwd (oracle/llama_ref.c).  Reference call site: the model's linear nodes
// behind runtime->Run() (/root/reference/src/engine/llm_engine.cc:113-116) under --quant-method of
// /root/reference/src/backends/cuda/resource_manager.cc:49-56.
#include <stdlib.h>

#include "k_gemm_dev.h"

namespace pplhip {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int PC_BM = 128, PC_BN = 64;
constexpr int PC_XB = PC_BM * 64 * 2;   // activation bytes per K tile: 16 KiB
constexpr int PC_WB = PC_BN * 64 * 2;   // converted weight bytes per K tile: 8 KiB
constexpr int PC_ST = 4;                // activation ring, K tiles
constexpr int PC_PD = 4;                // raw-weight prefetch distance, super-tiles
constexpr int PC_RD = PC_PD + 1;        // raw / scale ring slots
constexpr int PC_RAWB = PC_BN * 64;     // raw int4 bytes per super-tile: 4 KiB (1 KiB per producer wave)
constexpr int PC_SCB = 1024;            // scale slot: one dword per producer lane
constexpr int PC_LDS = PC_ST * PC_XB + 4 * PC_WB + PC_RD * (PC_RAWB + PC_SCB);
static_assert(PC_LDS <= 160 * 1024, "one block per CU");

// LDS-DMA, 16 / 4 bytes per lane: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset (the per-tile address update is
// then scalar arithmetic).  M0 written and restored in the same statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_wave_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}
__device__ __forceinline__ void glds4_s(uint32_t voff, const void* sbase, uint32_t lds_wave_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}

#ifdef PC_ABLATE_BUILD
#define PC_VMCNT(N) do { if (ABL & 12) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); } while (0)
#else
#define PC_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#endif
#define PC_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// EPI: EPI_F16 / EPI_SWIGLU.  SPLIT: K slabs (fp32 partial sums [split][M][N] at ws; the epilogue is applied by whoever sums them).
template <int EPI, bool SPLIT>
__global__ __launch_bounds__(512) void gemm_w4_pc_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ w,
                                                         const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                         void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles, int nst_per_split,
                                                         float* __restrict__ ws, int abl_arg) {
#ifdef PC_ABLATE_BUILD   // diagnosis builds only (profiles/probes/w4_pc_ablate.sh; WRONG results): 1 no MFMAs, 2 no conversion, 4 no activation
    const int ABL = abl_arg;   // refills, 8 no raw-weight refills, 16 no fragment reads
#else
    constexpr int ABL = 0;
#endif
    extern __shared__ __attribute__((aligned(128))) char smem_pc[];  // ONE shared object (a second one de-pipelines DMA kernels)
    char* const Xs = smem_pc;
    char* const Wf = smem_pc + PC_ST * PC_XB;
    char* const Raw = Wf + 4 * PC_WB;
    char* const Scl = Raw + PC_RD * PC_RAWB;

    // XCD id % 8 owns the weight tiles n == id (mod 8); the m tiles of a weight tile are neighbours on that XCD (its L2 serves the second)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles), mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * PC_BN;
    const int64_t m0 = (int64_t)mt * PC_BM;
    const int G = K >> 7;                              // super-tiles (= quantisation groups) along K
    const int split_id = blockIdx.y;
    const int g0 = split_id * nst_per_split;
    const int nst = (g0 + nst_per_split < G) ? nst_per_split : G - g0;   // >= 1 (launcher)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------------------------------
        // producer / converter wave pw: activation pieces P = pw + 4 j (8 rows x 128 B each) of every K tile; weight rows 16 pw .. + 16
        // ------------------------------------------------------------------------------------------------------------------------
        const int pw = wave - 4;
        uint32_t xoff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * (pw + 4 * j) + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            int64_t m = m0 + row;
            if (m >= M) m = M - 1;                     // rows past M are never stored
            xoff[j] = (uint32_t)((m * K + c * 8) * 2);
        }
        const int seg = lane & 3, R = 16 * pw + (lane >> 2);   // this lane's weight row inside the tile, its 32-nibble segment of a super-tile
        int n = n0 + R;
        if (n >= N) n = N - 1;
        const uint32_t woff = (uint32_t)((int64_t)n * (K >> 1) + seg * 16);
        const uint32_t nG = (uint32_t)n * (uint32_t)G;          // index of the row's first group scale
        const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(Xs) + pw * 1024);
        const uint32_t rdst = __builtin_amdgcn_readfirstlane(lds_addr(Raw) + pw * 1024);
        const uint32_t sdst = __builtin_amdgcn_readfirstlane(lds_addr(Scl) + pw * 256);
        const char* const xb = reinterpret_cast<const char*>(x) + (int64_t)g0 * 256;   // this split's first K tile
        const int kt_last = 2 * nst - 1;
        // converter addresses: chunk j of this lane's segment goes to K tile kt = seg >> 1, position ((seg & 1) * 4 + j) ^ f(R) ^ 2 kt
        const int kt_l = seg >> 1, bp = ((seg & 1) * 4) ^ ((R >> 1) & 7) ^ (2 * kt_l);
        char* const wdst = Wf + kt_l * PC_WB + R * 128;
        const char* const rsrc = Raw + pw * 1024 + lane * 16;
        const char* const ssrc = Scl + pw * 256 + lane * 4;

        auto issue_x = [&](int t) {                    // K tile t of this split -> ring stage t % PC_ST (past the end: tile kt_last again)
            if ((ABL & 4) && t >= PC_ST) return;
            const int tc = t < kt_last ? t : kt_last;
            const char* base = xb + (int64_t)tc * 128;
            const uint32_t d = xdst + (uint32_t)(t & (PC_ST - 1)) * PC_XB;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16_s(xoff[j], base, d + j * 4096);
        };
        auto issue_r = [&](int s, int rs) {            // raw weights + scales of super-tile s -> raw slot rs
            if ((ABL & 8) && s > PC_PD) return;
            const int g = g0 + (s < nst ? s : nst - 1);
            glds16_s(woff, w + (int64_t)g * 64, rdst + rs * PC_RAWB);
            glds4_s((2u * (nG + (uint32_t)g)) & ~3u, scale, sdst + rs * PC_SCB);
        };
        uint4 raw;                                     // the super-tile being converted: half of it in front of each of the two barriers
        h2 sc2;
        char* cd;
        auto convert_a = [&](int s, int rs) {          // raw slot rs (super-tile s) -> fp16 image (s & 1), chunks 0 and 1
            if (ABL & 2) return;
            const int g = g0 + s;
            raw = *reinterpret_cast<const uint4*>(rsrc + rs * PC_RAWB);
            const _Float16 sv = *reinterpret_cast<const _Float16*>(ssrc + rs * PC_SCB + ((nG + (uint32_t)g) & 1u) * 2);
            sc2 = h2{sv, sv};
            cd = wdst + (s & 1) * (2 * PC_WB);
            *reinterpret_cast<uint4*>(cd + ((bp ^ 0) << 4)) = __builtin_bit_cast(uint4, cvt_i4x8_f16(raw.x, sc2));
            *reinterpret_cast<uint4*>(cd + ((bp ^ 1) << 4)) = __builtin_bit_cast(uint4, cvt_i4x8_f16(raw.y, sc2));
        };
        auto convert_b = [&]() {                       // chunks 2 and 3
            if (ABL & 2) return;
            *reinterpret_cast<uint4*>(cd + ((bp ^ 2) << 4)) = __builtin_bit_cast(uint4, cvt_i4x8_f16(raw.z, sc2));
            *reinterpret_cast<uint4*>(cd + ((bp ^ 3) << 4)) = __builtin_bit_cast(uint4, cvt_i4x8_f16(raw.w, sc2));
        };

        // prologue.  Issue order (what the counts below are derived from): R(0) S(0) .. R(PD) S(PD), X(0) X(1) X(2);
        // then per iteration s: X(2s+3) [4], R(s+1+PD) S(s+1+PD) [2], X(2s+4) [4].
#pragma unroll
        for (int s = 0; s <= PC_PD; ++s) issue_r(s, s);
        issue_x(0); issue_x(1); issue_x(2);
        PC_VMCNT(2 * PC_PD + 12);                      // R(0), S(0) landed
        convert_a(0, 0);
        convert_b();
        PC_VMCNT(8);                                   // X(0) landed
        PC_LGKM0();
        __builtin_amdgcn_s_barrier();                  // B_0: tile 0 published
        int rs_new = 0;                                // raw slot of super-tile s + 1 + PD == slot of super-tile s
        int rs_cvt = 1;                                // raw slot of super-tile s + 1
#define PC_ITER(S, NR, NX1)                                                                                              \
    do {                                                                                                                 \
        issue_x(2 * (S) + 3);                                                                                            \
        issue_r((S) + 1 + PC_PD, rs_new);                                                                               \
        PC_VMCNT(NR);                                  /* R(s+1), S(s+1) landed */                                       \
        if ((S) + 1 < nst) convert_a((S) + 1, rs_cvt);                                                                   \
        PC_VMCNT(NX1);                                 /* X(2s+1) landed */                                              \
        __builtin_amdgcn_s_barrier();                  /* B_{2s+1} */                                                    \
        issue_x(2 * (S) + 4);                                                                                            \
        if ((S) + 1 < nst) convert_b();                                                                                  \
        if ((S) + 1 < nst) PC_VMCNT(10); else PC_VMCNT(0);   /* X(2s+2) landed; last iteration: everything (the epilogue reuses the ring) */ \
        PC_LGKM0();                                    /* the converted super-tile is written */                          \
        __builtin_amdgcn_s_barrier();                  /* B_{2s+2} */                                                    \
        rs_new = rs_new == PC_RD - 1 ? 0 : rs_new + 1;                                                                   \
        rs_cvt = rs_cvt == PC_RD - 1 ? 0 : rs_cvt + 1;                                                                   \
    } while (0)
        // the first PD iterations wait for loads of the prologue: R(s+1) has 2 (PD - s - 1) + 12 + 10 s + 6 younger operations
        static_assert(PC_PD == 4, "peeled iterations below");
        PC_ITER(0, 2 * PC_PD + 16, 10);
        if (nst > 1) PC_ITER(1, 2 * PC_PD + 24, 12);
        if (nst > 2) PC_ITER(2, 2 * PC_PD + 32, 12);
        if (nst > 3) PC_ITER(3, 2 * PC_PD + 40, 12);
        for (int s = PC_PD; s < nst; ++s) PC_ITER(s, 10 * PC_PD, 12);
#undef PC_ITER
    } else {
        // ------------------------------------------------------------------------------------------------------------------------
        // consumer wave (wn, wm): 32 weight rows x 64 activation rows; fragments of K tile t + 1 are read while tile t is multiplied
        // ------------------------------------------------------------------------------------------------------------------------
        const int wn = wave & 1, wm = wave >> 1;
        const int r = lane & 31, hh = lane >> 5, sw = (hh ^ ((r >> 1) & 7)) << 4;
        const char* const xrow = Xs + (64 * wm + r) * 128;
        const char* const wrow = Wf + (32 * wn + r) * 128;
        f16v acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        uint4 fw[2][4], fx[2][4][2];
        // phase p = t % 4 fixes every LDS offset of tile t: ring stage p, weight image (p >> 1) & 1, K tile p & 1 of it
#define PC_READ(BUF, P)                                                                                                   \
    do {                                                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                \
            fw[BUF][ks] = *reinterpret_cast<const uint4*>(wrow + (((P) >> 1) & 1) * (2 * PC_WB) + ((P) & 1) * PC_WB +    \
                                                          (sw ^ (ks << 5) ^ (((P) & 1) << 5)));                          \
            fx[BUF][ks][0] = *reinterpret_cast<const uint4*>(xrow + (P) * PC_XB + (sw ^ (ks << 5)));                      \
            fx[BUF][ks][1] = *reinterpret_cast<const uint4*>(xrow + (P) * PC_XB + 32 * 128 + (sw ^ (ks << 5)));          \
        }                                                                                                                 \
    } while (0)
#define PC_MMA(BUF)                                                                                                       \
    do {                                                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                \
            const h8 a = __builtin_bit_cast(h8, fw[BUF][ks]);                                                             \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(h8, fx[BUF][ks][0]), acc0, 0, 0, 0);      \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(h8, fx[BUF][ks][1]), acc1, 0, 0, 0);      \
        }                                                                                                                 \
    } while (0)
#define PC_PHASE(P)                                                                                                       \
    do {                                                                                                                  \
        PC_LGKM0();                                    /* my reads of tile t are complete: its stage may be refilled */   \
        __builtin_amdgcn_s_barrier();                  /* B_{t+1} */                                                      \
        __builtin_amdgcn_sched_barrier(0);             /* (hipcc otherwise hoists the next phase's wait + barrier above these MFMAs) */ \
        if (!(ABL & 16)) PC_READ(((P) + 1) & 1, ((P) + 1) & 3);         /* (unconditional: behind a branch hipcc waits lgkmcnt(0) at the join; past the last tile the reads fetch stale bytes that nobody uses) */ \
        if (!(ABL & 1)) PC_MMA((P) & 1);                                                                                  \
        /* the twelve fragment reads in the shadow of the first two MFMAs (left alone, hipcc re-uses the registers of issued MFMAs and */ \
        /* the reads land only at the end of the phase, in front of the wait) */                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                                                \
        asm volatile("" : "+v"(acc0), "+v"(acc1));     /* the MFMAs stay in their phase (an MFMA is register-only: neither "memory" nor sched_barrier holds it) */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    } while (0)
        const int kt_all = 2 * nst;
        __builtin_amdgcn_s_barrier();                  // B_0
        PC_READ(0, 0);
        int t0 = 0;
        for (; t0 + 4 <= kt_all; t0 += 4) {
            PC_PHASE(0);
            PC_PHASE(1);
            PC_PHASE(2);
            PC_PHASE(3);
        }
        if (t0 < kt_all) {                             // kt_all = 2 nst: two tiles left when nst is odd (a break inside the loop costs 16 accumulator copies per pass)
            PC_PHASE(0);
            PC_PHASE(1);
        }
#undef PC_PHASE
#undef PC_MMA
#undef PC_READ
        // ---- epilogue, part 1: accumulators -> LDS staging image (the rings are idle: every DMA has landed, every fragment is read).
        // accumulator j, value e of lane (r, hh): channel 32 wn + 8 (e >> 2) + 4 hh + (e & 3), row 64 wm + 32 j + r
        if constexpr (SPLIT) {
            constexpr int PITCH = PC_BN * 4 + 16;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f16v& a = j ? acc1 : acc0;
                char* d = smem_pc + (64 * wm + 32 * j + r) * PITCH + (32 * wn + 4 * hh) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(d + g * 32) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
            }
        } else if constexpr (EPI == EPI_F16) {
            constexpr int PITCH = PC_BN * 2 + 16;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f16v& a = j ? acc1 : acc0;
                char* d = smem_pc + (64 * wm + 32 * j + r) * PITCH + (32 * wn + 4 * hh) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const h4 o = {to_h(a[4 * g]), to_h(a[4 * g + 1]), to_h(a[4 * g + 2]), to_h(a[4 * g + 3])};
                    *reinterpret_cast<uint2*>(d + g * 16) = __builtin_bit_cast(uint2, o);
                }
            }
        } else {
            constexpr int PITCH = PC_BN + 16;          // 32 outputs per row
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f16v& a = j ? acc1 : acc0;
                char* d = smem_pc + (64 * wm + 32 * j + r) * PITCH + (16 * wn + 2 * hh) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float g0f = round_h(a[4 * g]), u0 = round_h(a[4 * g + 1]), g1f = round_h(a[4 * g + 2]), u1 = round_h(a[4 * g + 3]);
                    const h2 o = {to_h(g0f / (1.0f + __expf(-g0f)) * u0), to_h(g1f / (1.0f + __expf(-g1f)) * u1)};
                    *reinterpret_cast<uint32_t*>(d + g * 8) = __builtin_bit_cast(uint32_t, o);
                }
            }
        }
    }
    // ---- epilogue, part 2 (all eight waves): whole rows of the staging image -> global memory, 16 bytes per lane
    __syncthreads();
    if constexpr (SPLIT) {
        constexpr int PITCH = PC_BN * 4 + 16, CPR = PC_BN / 4;
        float* const slab = ws + (int64_t)split_id * M * N;
#pragma unroll
        for (int c = tid; c < PC_BM * CPR; c += 512) {
            const int row = c / CPR, ch = c - row * CPR;
            const int64_t m = m0 + row;
            const int col = n0 + ch * 4;
            if (m < M && col < N)                      // N % 4 == 0
                *reinterpret_cast<float4*>(slab + m * N + col) = *reinterpret_cast<const float4*>(smem_pc + row * PITCH + ch * 16);
        }
    } else {
        constexpr int OUTW = EPI == EPI_SWIGLU ? PC_BN / 2 : PC_BN, PITCH = OUTW * 2 + 16, CPR = OUTW / 8;
        uint16_t* const y = reinterpret_cast<uint16_t*>(yv);
        const int nout = EPI == EPI_SWIGLU ? N / 2 : N, c0 = EPI == EPI_SWIGLU ? n0 / 2 : n0;
#pragma unroll
        for (int c = tid; c < PC_BM * CPR; c += 512) {
            const int row = c / CPR, ch = c - row * CPR;
            const int64_t m = m0 + row;
            const int col = c0 + ch * 8;
            if (m < M) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem_pc + row * PITCH + ch * 16);
                if (col + 8 <= nout) {
                    *reinterpret_cast<uint4*>(y + m * ldy + col) = v;
                } else {
                    const uint16_t* e = reinterpret_cast<const uint16_t*>(&v);
                    for (int k = 0; k < 8; ++k)
                        if (col + k < nout) y[m * ldy + col + k] = e[k];
                }
            }
        }
    }
}

}  // namespace

// Shapes this kernel takes: int4 weights in groups of 128, K % 128 == 0, N % 4 == 0 (N % 16 == 0 with the fused SwiGLU), 16-byte aligned
// output rows, 32-bit byte offsets inside x and w.
bool linear_w4_pc_supported(int group, int64_t M, int N, int K, const void* y, int64_t ldy, int epi) {
    if (group != 128 || K % 128 || K < 128 || N % 4 || M < 1) return false;
    if (epi != EPI_F16 && epi != EPI_SWIGLU) return false;
    if (epi == EPI_SWIGLU && N % 16) return false;
    if (ldy % 8 || ((uintptr_t)y & 15)) return false;
    if ((uint64_t)M * (uint64_t)K * 2 >= (1ull << 32) || (uint64_t)N * (uint64_t)K / 2 >= (1ull << 32)) return false;
    if ((uint64_t)N * (uint64_t)(K / 128) >= (1ull << 30)) return false;
    return true;
}

// splits > 1: fp32 slabs [splits][M][N] at ws (the caller reduces them); nst_per_split super-tiles (128 k) per split
hipError_t launch_linear_w4_pc(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                               int64_t ldy, int epi, int splits, float* ws) {
    const int n_tiles = (N + PC_BN - 1) / PC_BN, m_tiles = (int)((M + PC_BM - 1) / PC_BM);
    const int G = K / 128;
    if (splits < 1) splits = 1;
    if (splits > G) splits = G;
    const int nst_per = (G + splits - 1) / splits;
    splits = (G + nst_per - 1) / nst_per;              // no empty split
    static bool attr_dev[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_dev[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)gemm_w4_pc_kernel<EPI_F16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_w4_pc_kernel<EPI_SWIGLU, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_w4_pc_kernel<EPI_F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        attr_dev[dev & 63] = true;
    }
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8 * m_tiles), (unsigned)splits), block(512);
    const uint8_t* wq = reinterpret_cast<const uint8_t*>(w);
#ifdef PC_ABLATE_BUILD
    const int abl = getenv("PPLHIP_PC_ABL") ? atoi(getenv("PPLHIP_PC_ABL")) : 0;
#else
    const int abl = 0;
#endif
    if (splits > 1)
        hipLaunchKernelGGL((gemm_w4_pc_kernel<EPI_F16, true>), grid, block, PC_LDS, s, x, wq, scale, M, N, K, y, ldy, n_tiles, m_tiles, nst_per, ws, abl);
    else if (epi == EPI_SWIGLU)
        hipLaunchKernelGGL((gemm_w4_pc_kernel<EPI_SWIGLU, false>), grid, block, PC_LDS, s, x, wq, scale, M, N, K, y, ldy, n_tiles, m_tiles, nst_per, ws, abl);
    else
        hipLaunchKernelGGL((gemm_w4_pc_kernel<EPI_F16, false>), grid, block, PC_LDS, s, x, wq, scale, M, N, K, y, ldy, n_tiles, m_tiles, nst_per, ws, abl);
    return hipGetLastError();
}

}  // namespace pplhip
