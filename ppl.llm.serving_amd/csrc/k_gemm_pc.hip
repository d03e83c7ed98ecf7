// W4A16 (group 128) linear layers at a few hundred rows (round 5): 128 x 64 tiles, DMA waves beside MFMA waves, a rolling register pipeline.
//
// Why another tile kernel.  At M = 256 (config 4: LLaMA-2-70B W4A16, TP 8, batch 256) the 128 x 128 ring kernel of k_gemm_dev.h ran the
// layer's four GEMMs at 19 % of the MFMA peak (profiles/r04_late_experiments.md 3): 112 / 20 / 128 output tiles do not fill 256 CUs, so
// every launch was cut into K slabs (fp32 partial sums of the whole output written and read back: 2.8 x the algorithmic bytes on w13,
// 4 x on wo) and inside a wave each K tile was a serial wait -> barrier -> DMA issue -> int4 conversion -> fragment reads -> MFMA chain.
// This kernel changes the decomposition and the schedule:
//   * block tile 128 (m) x 64 (n): 224 / 256 / 256 blocks for w13 / wo / w2 of that layer -- one per CU, NO K slabs, outputs written once
//     in fp16 through LDS (whole 128-byte rows); wqkv (40 tiles: slabs in any case) stays on the ring kernel, where it measured faster;
//   * a K step is a SUPER-TILE of 128 = one quantisation group: one s_barrier per 128 deep (the barrier round trip was ~150 cycles of a
//     64-deep tile whose MFMAs take 256, profiles/r05_w4_pc_experiments.md);
//   * waves 4..7 (producers) only issue LDS-DMA: the activation super-tile (two [128 rows][64 fp16] tiles, 8 one-KiB pieces per wave)
//     into a 4-slot ring, the raw int4 weights (64 rows x 64 bytes, one piece per wave) and their group scales into 4-slot rings;
//   * waves 0..3 (consumers on v_mfma_f32_32x32x16_f16: 2 (n) x 2 (k halves), each 32 (n) x 128 (m) x half of every super-tile's k, so that
//     every weight is converted once per block; the halves are added through LDS at the end) keep ONE register set of fragments that
//     rolls: behind the four MFMAs of k-step ks of super-tile s, the activation fragments of k-step ks of super-tile s + 1 are read into
//     the registers those MFMAs just released, and the int4 weights of that k-step (8 nibbles per lane, read a super-tile earlier) are
//     converted to fp16(q x scale) in registers (cvt_i4x8_f16) -- no fp16 weight image in LDS, no converter wave on the critical path
//     (the first form of this kernel had both: its converter waves needed ~840 cycles per 64-deep tile, ibid.).  Lane half h of wave half kh multiplies
//     k = 64 h + 32 kh + 8 ks .. + 8 of the super-tile in k-step ks (both operands use the same permutation of the contraction index), so that
//     a lane's 16 weight bytes are one LDS read and its activation chunk 4 kh + ks sits in the 64-deep tile h.
// LDS (148 KiB, one block per CU): X ring 4 x 32 KiB, 16-byte chunk q of row r of a tile at position q ^ ((r >> 1) & 7) (conflict-free
// ds_read_b128 by 32 rows); raw ring 4 x 4 KiB [64 rows][64 B], chunk c of row r at c ^ ((r >> 2) & 3); scale ring 4 x 1 KiB (one
// dword per producer lane).  Every global access of the producers is an LDS-DMA issued from inline asm (no VGPR destinations: nothing
// for hipcc to mis-wait), counted with s_waitcnt vmcnt(N); N follows from the fixed issue order, and the tail re-issues clamped loads
// so that the counts stay static.
// Numerics: the dequantised weight is the fp16 number fp16(q * scale) (cvt_i4x8_f16, as every W4 kernel here), fp32 accumulation, one
// rounding of the sum to fp16.  Oracle: ref_linear_fwd (oracle/llama_ref.c).  Reference call site: the model's linear nodes behind
// runtime->Run() (/root/reference/src/engine/llm_engine.cc:113-116) under --quant-method of
// /root/reference/src/backends/cuda/resource_manager.cc:49-56.
#include <stdlib.h>

#include "k_gemm_dev.h"

namespace pplhip {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int PC_BM = 128, PC_BN = 64;
constexpr int PC_XT = PC_BM * 64 * 2;   // one 64-deep activation tile: 16 KiB
constexpr int PC_XB = 2 * PC_XT;        // activation bytes per super-tile: 32 KiB
constexpr int PC_NS = 4;                // ring slots (activations, raw weights, scales), super-tiles
constexpr int PC_RAWB = PC_BN * 64;     // raw int4 bytes per super-tile: 4 KiB (1 KiB per producer wave)
constexpr int PC_SCB = 1024;            // scale slot: one dword per producer lane
constexpr int PC_LDS = PC_NS * (PC_XB + PC_RAWB + PC_SCB);
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
typedef uint32_t pc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 lds_ld16(uint32_t a) {   // ds_read_b128 from an LDS byte address
    const pc_u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) pc_u32x4*>((uintptr_t)a);
    return __builtin_bit_cast(uint4, v);
}
__device__ __forceinline__ uint32_t lds_ld4(uint32_t a) {
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)a);
}

#ifdef PC_ABLATE_BUILD
#define PC_VMCNT(N) do { if (ABL & 12) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); } while (0)
#else
#define PC_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#endif
#define PC_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifdef PC_TIME_BUILD   // diagnosis build (profiles/probes/w4_pc_timeline.sh): cycle stamps of every wave of block 0 at the protocol's points
__device__ uint64_t pc_time_log[8][256];
#define PC_T() do { if (blockIdx.x == 0 && lane == 0 && ti < 256) pc_time_log[wave][ti] = __builtin_readcyclecounter(); ++ti; } while (0)
#else
#define PC_T() do {} while (0)
#endif

// EPI: EPI_F16 / EPI_SWIGLU
template <int EPI>
__global__ __launch_bounds__(512) void gemm_w4_pc_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ w,
                                                         const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                         void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles, int abl_arg) {
#ifdef PC_ABLATE_BUILD   // diagnosis builds only (profiles/probes/w4_pc_ablate.sh; WRONG results): 1 no MFMAs, 2 no conversion, 4 no activation
    const int ABL = abl_arg;   // refills, 8 no raw-weight refills, 16 no fragment reads
#else
    constexpr int ABL = 0;
#endif
    extern __shared__ __attribute__((aligned(128))) char smem_pc[];  // ONE shared object (a second one de-pipelines DMA kernels)
    char* const Xs = smem_pc;
    char* const Raw = smem_pc + PC_NS * PC_XB;
    char* const Scl = Raw + PC_NS * PC_RAWB;

    // XCD id % 8 owns the weight tiles n == id (mod 8); the m tiles of a weight tile are neighbours on that XCD (its L2 serves the second)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles), mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * PC_BN;
    const int64_t m0 = (int64_t)mt * PC_BM;
    const int G = K >> 7;                              // super-tiles (= quantisation groups) along K
    constexpr int g0 = 0;
    const int nst = G;                                 // >= 1 (launcher)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ti = 0;
    (void)ti;
    PC_T();                                            // 0: start

    // Barrier B_j (j = 0 .. nst) "super-tile j is published": the producers have seen the activations of super-tile j and the raw
    // weights + scales of super-tile j + 1 land; the consumers have every fragment of super-tile j - 1 in registers (so the slots of
    // activations j - 1 and raw weights j are free behind it).
    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------------------------------
        // producer wave pw: activation pieces P = pw + 4 i (8 rows x 128 B) of both tiles of every super-tile; weight rows 16 pw .. + 16
        // ------------------------------------------------------------------------------------------------------------------------
        const int pw = wave - 4;
        uint32_t xoff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (pw + 4 * i) + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            int64_t m = m0 + row;
            if (m >= M) m = M - 1;                     // rows past M are never stored
            xoff[i] = (uint32_t)((m * K + c * 8) * 2);
        }
        const int R = 16 * pw + (lane >> 2);           // this lane's weight row inside the tile; it fetches LDS chunk position lane & 3 of it
        int n = n0 + R;
        if (n >= N) n = N - 1;
        const uint32_t woff = (uint32_t)((int64_t)n * (K >> 1) + (((lane & 3) ^ ((R >> 2) & 3)) << 4));
        const uint32_t nG = (uint32_t)n * (uint32_t)G;          // index of the row's first group scale
        const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(Xs) + pw * 1024);
        const uint32_t rdst = __builtin_amdgcn_readfirstlane(lds_addr(Raw) + pw * 1024);
        const uint32_t sdst = __builtin_amdgcn_readfirstlane(lds_addr(Scl) + pw * 256);
        const char* const xb = reinterpret_cast<const char*>(x);

        // DMA operations per super-tile and wave: raw weights (1), scales (1), activations (tile 0: 4 pieces, tile 1: 4 pieces).
        // Super-tiles past the end re-load the last one (into slots nobody reads any more): the wait counts stay static.
        auto issue_r = [&](int j) {
            if ((ABL & 8) && j >= PC_NS) return;
            const int g = g0 + (j < nst ? j : nst - 1), sl = j & (PC_NS - 1);
            glds16_s(woff, w + (int64_t)g * 64, rdst + sl * PC_RAWB);
            glds4_s((2u * (nG + (uint32_t)g)) & ~3u, scale, sdst + sl * PC_SCB);
        };
        auto issue_x = [&](int j) {
            if ((ABL & 4) && j >= PC_NS - 1) return;
            const char* base = xb + (int64_t)(j < nst ? j : nst - 1) * 256;
            const uint32_t d = xdst + (uint32_t)(j & (PC_NS - 1)) * PC_XB;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_s(xoff[i], base, d + i * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_s(xoff[i], base + 128, d + PC_XT + i * 4096);
        };
        // prologue: R S X(0) | R S X(1) | R S X(2) | R S (3); iteration j then issues R S (j + 4), X(j + 3)
        issue_r(0); issue_x(0); issue_r(1); issue_x(1); issue_r(2); issue_x(2); issue_r(3);
        PC_VMCNT(2 + 10);                              // x(0) [and raw(0), raw(1)] landed: younger = raw(3) (2) + all of super-tile 2 (10)
        PC_T();                                        // 1: prologue loads landed
        __builtin_amdgcn_s_barrier();                  // B_0
        __builtin_amdgcn_s_barrier();                  // B_0': the consumers hold super-tile 0 and raw(1) in registers (raw slot 0 is free)
        PC_T();                                        // 2: behind B_0'
        for (int j = 0; j < nst; ++j) {
            // behind B_j: raw slot j % 4 and activation slot (j - 1) % 4 are free
            issue_r(j + PC_NS);
            issue_x(j + PC_NS - 1);
            PC_T();                                    // 3 + 3 j: issued
            // x(j + 1) and raw(j + 2) landed.  x(j + 1) was issued two iterations ago (in the prologue for j < 2); younger than its last
            // piece: j = 0: R S of super-tile 3 (2) + x(2)'s group (8) + this iteration (10) = 20 -- and 20 in the steady state as well
            if (j + 1 < nst) PC_VMCNT(20); else PC_VMCNT(0);   // last iteration: everything (the epilogue reuses the rings)
            PC_T();                                    // 4 + 3 j: landed
            __builtin_amdgcn_s_barrier();              // B_{j+1}
            PC_T();                                    // 5 + 3 j: behind the barrier
        }
        __builtin_amdgcn_s_barrier();                  // (the consumers' barrier inside the exchange of their k halves)
    } else {
        // ------------------------------------------------------------------------------------------------------------------------
        // consumer wave (wn, kh): 32 weight rows x all 128 activation rows x HALF of every super-tile's k (lane half h, wave half kh:
        // k = 64 h + 32 kh + 8 ks .. + 8 in k-step ks = 0..3).  Splitting the block's work by k instead of by m converts every weight ONCE
        // per block: 76 conversion instructions per wave and super-tile beside 16 MFMAs -- a wave hides ~4 instructions per 32-cycle MFMA
        // and pays ~5.3 cycles for each further one (profiles/probes/valu_beside_mfma_probe.hip), which is what held the m-split form
        // (152 conversions per wave) at the old kernel's speed.  The two k halves are added through LDS at the end.
        // ------------------------------------------------------------------------------------------------------------------------
        const int wn = wave & 1, kh = wave >> 1;
        const int r = lane & 31, hh = lane >> 5;
        // activation fragment (ks, jm) of ring slot sl: tile hh of the super-tile, row 32 jm + r, chunk position (4 kh + ks) ^ f(r).
        // Slots 2 and 3 lie beyond the 16-bit offset field of ds_read: a second base
        const uint32_t xa = lds_addr(Xs) + hh * PC_XT + r * 128 + ((((r >> 1) & 7) ^ (4 * kh)) << 4);
        uint32_t xlo[4], xhi[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { xlo[ks] = xa ^ (uint32_t)(ks << 4); xhi[ks] = xlo[ks] + 2 * PC_XB; }
        const int R = 32 * wn + r;
        // raw weights: this lane's 16 bytes = chunk 2 hh + kh of row R, stored at position c ^ ((R >> 2) & 3)
        const uint32_t ra = lds_addr(Raw) + R * 64 + (((2 * hh + kh) ^ ((R >> 2) & 3)) << 4);
        int nrow = n0 + R;
        if (nrow >= N) nrow = N - 1;
        const uint32_t nG = (uint32_t)nrow * (uint32_t)G;
        const uint32_t sa = lds_addr(Scl) + (R >> 4) * 256 + (R & 15) * 16;   // the dword its row's producer lanes fetched

        f16v acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        uint4 xf[4][4];                                // activation fragments of the super-tile being multiplied (rolling)
        h8 wf[4];                                      // its converted weight fragments (rolling)
        uint4 rw;                                      // raw weights of the NEXT super-tile (words = its k-steps)
        uint32_t scw;                                  // ... and the dword that holds its scale

#define PC_XADDR(SL, KS, JM) (((SL) < 2 ? xlo[KS] : xhi[KS]) + (uint32_t)(((SL) & 1) * PC_XB + (JM) * 4096))
#define PC_RAW_READ(SL)                                                                                                   \
    do {                                                                                                                  \
        rw = lds_ld16(ra + (SL) * PC_RAWB);                                                                               \
        scw = lds_ld4(sa + (SL) * PC_SCB);                                                                                \
    } while (0)
        auto scale_of = [&](int g) {                   // the group scale in scw: element nG + g of the scale matrix
            const uint32_t v = ((nG + (uint32_t)g) & 1u) ? (scw >> 16) : (scw & 0xffffu);
            const _Float16 sv = __builtin_bit_cast(_Float16, (uint16_t)v);
            return h2{sv, sv};
        };

        // prologue: everything of super-tile 0, the raw weights of super-tile 1
        __builtin_amdgcn_s_barrier();                  // B_0
        PC_T();                                        // 1: behind B_0
        PC_RAW_READ(0);
        {
            const h2 sc2 = scale_of(g0);
            const uint32_t wv[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                wf[ks] = cvt_i4x8_f16(wv[ks], sc2);
#pragma unroll
                for (int jm = 0; jm < 4; ++jm) xf[ks][jm] = lds_ld16(PC_XADDR(0, ks, jm));
            }
        }
        PC_RAW_READ(1);
        PC_LGKM0();
        __builtin_amdgcn_s_barrier();                  // B_0'
        PC_T();                                        // 2: behind B_0'

        // phase of super-tile s (slot p = s % 4): MFMAs of s; behind k-step ks its registers take k-step ks of super-tile s + 1 (slot p + 1);
        // the raw weights of s + 2 (slot p + 2) are read at the end
#define PC_PHASE(P)                                                                                                       \
    do {                                                                                                                  \
        PC_T();                                        /* 3 + 3 s: phase top */                                           \
        PC_LGKM0();                                    /* every read of super-tile s + 1's predecessors has landed */     \
        PC_T();                                        /* 4 + 3 s: reads landed */                                        \
        __builtin_amdgcn_s_barrier();                  /* B_{s+1} */                                                      \
        PC_T();                                        /* 5 + 3 s: behind the barrier */                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        const h2 sc2 = scale_of(g0 + s0 + (P) + 1);                                                                       \
        const uint32_t wv[4] = {rw.x, rw.y, rw.z, rw.w};                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                \
            _Pragma("unroll") for (int jm = 0; jm < 4; ++jm) {                                                            \
                if (!(ABL & 1)) acc[jm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], __builtin_bit_cast(h8, xf[ks][jm]), acc[jm], 0, 0, 0); \
                if (!(ABL & 16)) xf[ks][jm] = lds_ld16(PC_XADDR(((P) + 1) & 3, ks, jm));                                  \
            }                                                                                                             \
            if (!(ABL & 2)) wf[ks] = cvt_i4x8_f16(wv[ks], sc2);                                                           \
        }                                                                                                                 \
        PC_RAW_READ(((P) + 2) & 3);                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* an MFMA */                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* the fragment read that refills its register */        \
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   /* a quarter of a k-step's conversion for the next super-tile */ \
        }                                                                                                                 \
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));   /* the MFMAs stay in their phase (register-only: "memory" does not hold them) */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    } while (0)
        int s0 = 0;
        for (; s0 + 4 <= nst; s0 += 4) {
            PC_PHASE(0);
            PC_PHASE(1);
            PC_PHASE(2);
            PC_PHASE(3);
        }
        if (s0 < nst) PC_PHASE(0);
        if (s0 + 1 < nst) PC_PHASE(1);
        if (s0 + 2 < nst) PC_PHASE(2);
#undef PC_PHASE
#undef PC_RAW_READ
#undef PC_XADDR
        // ---- epilogue, part 1: the k halves are added through LDS (the rings are idle: every DMA has landed, every fragment is read),
        // then accumulators -> LDS staging image.  accumulator j, value e of lane (r, hh): channel 32 wn + 8 (e >> 2) + 4 hh + (e & 3), row 32 j + r
        PC_T();                                        // loop done
        PC_LGKM0();                                    // (the last phase's look-ahead reads)
        {
            float4* const red = reinterpret_cast<float4*>(smem_pc + 64 * 1024);   // [wn][j][g][lane] float4: 32 KiB behind the staging image
            if (kh == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        red[((wn * 4 + j) * 4 + g) * 64 + lane] = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
            }
            __builtin_amdgcn_s_barrier();              // (all eight waves: the producers meet it below)
            if (kh == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 o = red[((wn * 4 + j) * 4 + g) * 64 + lane];
                        acc[j][4 * g] += o.x; acc[j][4 * g + 1] += o.y; acc[j][4 * g + 2] += o.z; acc[j][4 * g + 3] += o.w;
                    }
            }
        }
        if (kh == 0) {
            if constexpr (EPI == EPI_F16) {
                constexpr int PITCH = PC_BN * 2 + 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    char* d = smem_pc + (32 * j + r) * PITCH + (32 * wn + 4 * hh) * 2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const h4 o = {to_h(acc[j][4 * g]), to_h(acc[j][4 * g + 1]), to_h(acc[j][4 * g + 2]), to_h(acc[j][4 * g + 3])};
                        *reinterpret_cast<uint2*>(d + g * 16) = __builtin_bit_cast(uint2, o);
                    }
                }
            } else {
                constexpr int PITCH = PC_BN + 16;      // 32 outputs per row
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    char* d = smem_pc + (32 * j + r) * PITCH + (16 * wn + 2 * hh) * 2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float g0f = round_h(acc[j][4 * g]), u0 = round_h(acc[j][4 * g + 1]), g1f = round_h(acc[j][4 * g + 2]), u1 = round_h(acc[j][4 * g + 3]);
                        const h2 o = {to_h(g0f / (1.0f + __expf(-g0f)) * u0), to_h(g1f / (1.0f + __expf(-g1f)) * u1)};
                        *reinterpret_cast<uint32_t*>(d + g * 8) = __builtin_bit_cast(uint32_t, o);
                    }
                }
            }
        }
    }
    // ---- epilogue, part 2 (all eight waves): whole rows of the staging image -> global memory, 16 bytes per lane
    PC_T();                                            // staged
    __syncthreads();
    PC_T();
    {
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
    PC_T();                                            // stores issued
}

}  // namespace

// Shapes this kernel takes: int4 weights in groups of 128, K % 128 == 0, N % 4 == 0 (N % 16 == 0 with the fused SwiGLU), 16-byte aligned
// output rows, 32-bit byte offsets inside x and w.
bool linear_w4_pc_supported(int group, int64_t M, int N, int K, const void* x, const void* w, const void* scale, const void* y, int64_t ldy, int epi) {
    if (group != 128 || K % 128 || K < 128 || N % 4 || M < 1) return false;
    if (epi != EPI_F16 && epi != EPI_SWIGLU) return false;
    if (epi == EPI_SWIGLU && N % 16) return false;
    if (ldy % 8 || ((uintptr_t)y & 15)) return false;
    // the LDS-DMA fetches x and w in 16-byte pieces and the scale as the aligned dword around fp16 element n G + g
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)scale & 3)) return false;
    if ((uint64_t)M * (uint64_t)K * 2 >= (1ull << 32) || (uint64_t)N * (uint64_t)K / 2 >= (1ull << 32)) return false;
    if ((uint64_t)N * (uint64_t)(K / 128) >= (1ull << 30)) return false;
    return true;
}

hipError_t launch_linear_w4_pc(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                               int64_t ldy, int epi) {
    const int n_tiles = (N + PC_BN - 1) / PC_BN, m_tiles = (int)((M + PC_BM - 1) / PC_BM);
    static bool attr_dev[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_dev[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)gemm_w4_pc_kernel<EPI_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_w4_pc_kernel<EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        attr_dev[dev & 63] = true;
    }
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8 * m_tiles)), block(512);
    const uint8_t* wq = reinterpret_cast<const uint8_t*>(w);
#ifdef PC_ABLATE_BUILD
    const int abl = getenv("PPLHIP_PC_ABL") ? atoi(getenv("PPLHIP_PC_ABL")) : 0;
#else
    const int abl = 0;
#endif
    if (epi == EPI_SWIGLU)
        hipLaunchKernelGGL((gemm_w4_pc_kernel<EPI_SWIGLU>), grid, block, PC_LDS, s, x, wq, scale, M, N, K, y, ldy, n_tiles, m_tiles, abl);
    else
        hipLaunchKernelGGL((gemm_w4_pc_kernel<EPI_F16>), grid, block, PC_LDS, s, x, wq, scale, M, N, K, y, ldy, n_tiles, m_tiles, abl);
#ifdef PC_TIME_BUILD
    if (getenv("PPLHIP_PC_TIME")) {
        static int calls = 0;
        if (++calls == 5) {   // a warm call
            (void)hipStreamSynchronize(s);
            static uint64_t h[8][256];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(pc_time_log), sizeof(h));
            const int nst = K / 128;
            for (int w = 0; w < 8; ++w) {
                fprintf(stderr, "[pc_time] wave %d (%s) N=%d K=%d:", w, w < 4 ? "consumer" : "producer", N, K);
                const int n = 3 + 3 * nst + 4;
                for (int i = 1; i < n && i < 256; ++i) fprintf(stderr, " %lld", (long long)(h[w][i] - h[w][0]));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace pplhip
