// Round-3 GEMM probes (NOT part of the build): the two W8A16 tile-kernel variants measured against gemm_dma_body<8, *, 2, 1> and rejected,
// profiles/r03_gemm_experiments.md #11 (weights straight from global memory into registers) and #15 (192 x 128 tile, 48 x 128 per wave).
// They were compiled inside ppl.llm.serving_amd/csrc/k_gemm.hip (they use its helpers: glds16, lds_addr, g_swz, w_swz, cvt_i8x8_f16,
// store4, G_BM / G_BK) and selected in launch_linear by the hooks at the end of this file (PPLHIP_GEMM_WD=1 / PPLHIP_GEMM_N192=1).
// Both were correct on every `linear` test.  To rebuild: paste the kernels above gemm_w8_dma256_kernel and the hooks in front of the
// `forced_wl` line of launch_linear.

// ---------------------------------------------------------------------------------------------------------------
// W8A16, 128 x 128 x 64 tile, 4 waves of 32(n) x 128(m), WEIGHTS STRAIGHT FROM GLOBAL MEMORY INTO REGISTERS (round 3).
// In this wave layout a wave owns its 32 weight rows alone, so the weight tile needs no LDS at all: lane (row l15, quarter kq)
// loads the 16 bytes k = kq*16 .. +16 of its two rows of a K tile -- exactly the two MFMA A fragments it multiplies -- and a
// ring of three register sets replaces the LDS weight stages.  Against gemm_dma_body<8, *, 2, 1> this removes a third of the
// LDS-DMA instructions a wave issues per tile (4 instead of 6; the issue stalls the in-order wave ~100 cycles each), a third
// of the LDS write traffic, the weight-fragment ds_reads at the head of every iteration, and it fits a THREE-stage activation
// ring in the 48 KiB that keep three blocks per CU (the DMA form affords two stages there).
// The register loads are issued from inline asm at the same pipeline distance as the tile's LDS-DMA, so ONE counted
// s_waitcnt vmcnt(6 x younger tiles) covers both (hipcc counts neither); the wait statement names the registers it releases
// ("+v"), which keeps their consumers below it (cdna_hip_programming.md 5.7, form ii).
// ---------------------------------------------------------------------------------------------------------------
typedef uint32_t g_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ g_u32x4 gload16_asm(const void* p) {
    g_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w8_wd_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                         const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                         void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    constexpr int ST = 3, NT = 256, X_DMA = 4, NI = 2, NJ = 8;
    __shared__ __attribute__((aligned(16))) char smem[ST * G_BM * G_BK * 2];
    uint16_t* const Xs0 = reinterpret_cast<uint16_t*>(smem);
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles), mt = slot % m_tiles;  // XCD x owns weight tiles n == x (mod 8)
    if (nt >= n_tiles) return;
    const int n0 = nt * G_BN;
    const int64_t m0 = (int64_t)mt * G_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int nb = wave * 32;

    const uint16_t* xsrc[X_DMA];
#pragma unroll
    for (int j = 0; j < X_DMA; ++j) {
        const int p = j * NT + tid, row = p >> 3, pos = (p & 7) ^ ((row >> 1) & 7);
        const int c = ((pos & 3) << 1) | (pos >> 2);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;
        xsrc[j] = x + m * K + c * 8;
    }
    const int8_t* wsrc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int n = n0 + nb + i * 16 + l15;
        if (n >= N) n = N - 1;
        wsrc[i] = w + (int64_t)n * K + kq * 16;
    }
    const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(Xs0) + wave * 1024);
    auto issue_x = [&](int stage, int k0) {
#pragma unroll
        for (int j = 0; j < X_DMA; ++j) glds16(xsrc[j] + k0, xdst + stage * (G_BM * G_BK * 2) + j * (NT * 16));
    };

    f4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int ktiles = K / G_BK;
    g_u32x4 wa0, wa1, wb0, wb1, wc0, wc1;  // three register sets (scalars, not an array: no scratch, static names per stage)
    wa0 = wa1 = wb0 = wb1 = wc0 = wc1 = g_u32x4{0, 0, 0, 0};
    // prologue: tiles 0 and 1
    issue_x(0, 0);
    wa0 = gload16_asm(wsrc[0]); wa1 = gload16_asm(wsrc[1]);
    {
        const int k1 = ktiles > 1 ? G_BK : 0;
        issue_x(1, k1);
        wb0 = gload16_asm(wsrc[0] + k1); wb1 = gload16_asm(wsrc[1] + k1);
    }
    // one K tile: wait for it (LDS-DMA + register loads), publish, refill the stage two ahead, multiply.  The refill is
    // UNCONDITIONAL (past the end it re-reads the last tile into a stage nobody reads again): exactly one younger tile is in
    // flight at every wait, and the loop body has no control flow -- the register sets keep their names across the back edge, so
    // hipcc has no reason to copy a register whose load is still in flight (audited in the ISA: no v_mov of the ring registers)
    const int klast = (ktiles - 1) * G_BK;
    auto step = [&](int t, auto stag, g_u32x4& w0, g_u32x4& w1, g_u32x4& n0r, g_u32x4& n1r) {
        constexpr int st = decltype(stag)::value;
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(w0), "+v"(w1)::"memory");
        __syncthreads();
        {
            int k0 = (t + 2) * G_BK;
            k0 = k0 < klast ? k0 : klast;
            issue_x(st == 0 ? 2 : st - 1, k0);
            n0r = gload16_asm(wsrc[0] + k0);
            n1r = gload16_asm(wsrc[1] + k0);
        }
        const uint16_t* xs = Xs0 + st * (G_BM * G_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 a[NI], bfr[NJ];
            a[0] = cvt_i8x8_f16(ks == 0 ? make_uint2(w0.x, w0.y) : make_uint2(w0.z, w0.w));
            a[1] = cvt_i8x8_f16(ks == 0 ? make_uint2(w1.x, w1.y) : make_uint2(w1.z, w1.w));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = j * 16 + l15;
                bfr[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&xs[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    int t = 0;
    for (; t + 3 <= ktiles; t += 3) {
        step(t, S0{}, wa0, wa1, wc0, wc1);
        step(t + 1, S1{}, wb0, wb1, wa0, wa1);
        step(t + 2, S2{}, wc0, wc1, wb0, wb1);
    }
    if (t < ktiles) {
        step(t, S0{}, wa0, wa1, wc0, wc1);
        if (t + 1 < ktiles) step(t + 1, S1{}, wb0, wb1, wa0, wa1);
    }
    // the past-the-end refills are still in flight: their destination registers stay live (named here) until they have landed,
    // or hipcc would hand those registers to the epilogue while a load is about to overwrite them
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wa0), "+v"(wa1), "+v"(wb0), "+v"(wb1), "+v"(wc0), "+v"(wc1)::"memory");

#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n0 + nb + i * 16 + kq * 4;
        if (n >= N) continue;
        const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int64_t m = m0 + j * 16 + l15;
            if (m >= M) continue;
            store4<EPI>(yv, ldy, m, n, acc[i][j][0] * (float)sh[0], acc[i][j][1] * (float)sh[1], acc[i][j][2] * (float)sh[2],
                        acc[i][j][3] * (float)sh[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// W8A16, 192(n) x 128(m) x 64 tile, 4 waves of 48(n) x 128(m) (round 3).  The 128 x 128 kernel's compute structure alone tops out
// at 49 % of the MFMA peak (profiles/r03_gemm_experiments.md #14): 18 fragment reads per 32 MFMAs per wave and K tile, the block's
// waves re-synchronised every ~500 MFMA cycles.  A wave tile of 48 x 128 multiplies 48 MFMAs per 19 fragment reads (2.5 per read instead
// of 1.8) between two barriers, on 96 accumulator registers: two blocks per CU (512 slots) instead of three -- and 12288 / 192 x 8 = 512
// tiles is exactly one round for wqkv at M = 1024.  Same LDS-DMA ring (two stages of 16 KiB activations + 12 KiB int8 weights), same
// swizzles and fragment reads as gemm_dma_body<8, *, 2, 1>.
// ---------------------------------------------------------------------------------------------------------------
constexpr int N192_BN = 192;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w8_n192_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                           const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                           void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    constexpr int ST = 2, NT = 256, X_DMA = 4, W_DMA = 3, NI = 3, NJ = 8;
    constexpr int X_STAGE = G_BM * G_BK * 2, W_STAGE = N192_BN * G_BK;
    __shared__ __attribute__((aligned(16))) char smem[ST * (X_STAGE + W_STAGE)];
    uint16_t* const Xs0 = reinterpret_cast<uint16_t*>(smem);
    char* const Wq0 = smem + ST * X_STAGE;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles), mt = slot % m_tiles;  // XCD x owns weight tiles n == x (mod 8)
    if (nt >= n_tiles) return;
    const int n0 = nt * N192_BN;
    const int64_t m0 = (int64_t)mt * G_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int nb = wave * (NI * 16);

    const uint16_t* xsrc[X_DMA];
    const int8_t* wsrc[W_DMA];
#pragma unroll
    for (int j = 0; j < X_DMA; ++j) {
        const int p = j * NT + tid, row = p >> 3, pos = (p & 7) ^ ((row >> 1) & 7);
        const int c = ((pos & 3) << 1) | (pos >> 2);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;
        xsrc[j] = x + m * K + c * 8;
    }
#pragma unroll
    for (int j = 0; j < W_DMA; ++j) {
        const int p = j * NT + tid, row = p >> 2, c = (p & 3) ^ w_swz(row);
        int n = n0 + row;
        if (n >= N) n = N - 1;
        wsrc[j] = w + (int64_t)n * K + c * 16;
    }
    const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(Xs0) + wave * 1024);
    const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds_addr(Wq0) + wave * 1024);
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int j = 0; j < X_DMA; ++j) glds16(xsrc[j] + k0, xdst + stage * X_STAGE + j * (NT * 16));
#pragma unroll
        for (int j = 0; j < W_DMA; ++j) glds16(wsrc[j] + k0, wdst + stage * W_STAGE + j * (NT * 16));
    };

    f4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int ktiles = K / G_BK;
    issue(0, 0);
    for (int t = 0; t < ktiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile t is published; the stage read during iteration t-1 is free
        if (t + 1 < ktiles) issue((t + 1) & 1, (t + 1) * G_BK);
        const uint16_t* xs = Xs0 + (t & 1) * (G_BM * G_BK);
        const char* wq = Wq0 + (t & 1) * W_STAGE;
        uint4 wraw[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = nb + i * 16 + l15;
            wraw[i] = *reinterpret_cast<const uint4*>(&wq[row * G_BK + (kq ^ w_swz(row)) * 16]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 a[NI], bfr[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[i] = cvt_i8x8_f16(ks == 0 ? make_uint2(wraw[i].x, wraw[i].y) : make_uint2(wraw[i].z, wraw[i].w));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = j * 16 + l15;
                bfr[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&xs[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n0 + nb + i * 16 + kq * 4;
        if (n >= N) continue;
        const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int64_t m = m0 + j * 16 + l15;
            if (m >= M) continue;
            store4<EPI>(yv, ldy, m, n, acc[i][j][0] * (float)sh[0], acc[i][j][1] * (float)sh[1], acc[i][j][2] * (float)sh[2],
                        acc[i][j][3] * (float)sh[3]);
        }
    }
}


/* ---- launch hooks (inside launch_linear, before the WL selection) ----
        static const int n192_mode = getenv("PPLHIP_GEMM_N192") ? atoi(getenv("PPLHIP_GEMM_N192")) : 0;
        if (n192_mode && wq_bit == 8 && splits == 1 && N % 4 == 0 && !(map_mode & 0xff)) {
            const int nt192 = (N + N192_BN - 1) / N192_BN;
            if ((int64_t)nt192 * m_tiles > 256) {
                dim3 g192((unsigned)((nt192 + 7) / 8 * 8 * m_tiles));
#define L192(E) hipLaunchKernelGGL((gemm_w8_n192_kernel<E>), g192, block, 0, s, x, (const int8_t*)w, scale, M, N, K, y, ldy, nt192, m_tiles)
                if (epi == EPI_F32) L192(EPI_F32); else if (epi == EPI_F16) L192(EPI_F16); else L192(EPI_SWIGLU);
#undef L192
                return hipGetLastError();
            }
        }
        static const int wd_mode = getenv("PPLHIP_GEMM_WD") ? atoi(getenv("PPLHIP_GEMM_WD")) : 0;
        if (wd_mode && wq_bit == 8 && splits == 1 && tiles > 256 && N % 4 == 0 && !(map_mode & 0xff)) {
#define LWD(E) hipLaunchKernelGGL((gemm_w8_wd_kernel<E>), grid, block, 0, s, x, (const int8_t*)w, scale, M, N, K, y, ldy, n_tiles, m_tiles)
            if (epi == EPI_F32) LWD(EPI_F32); else if (epi == EPI_F16) LWD(EPI_F16); else LWD(EPI_SWIGLU);
#undef LWD
            return hipGetLastError();
        }
*/
