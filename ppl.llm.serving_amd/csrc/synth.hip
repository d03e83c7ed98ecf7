// Synthetic weights: the counter-based generator of oracle/llama_ref.c (ref_synth_fill), bit for bit, run on the
// device so that a 7B / 70B slice is filled in milliseconds instead of being generated on the host and uploaded.
#include "kernels.h"

namespace pplhip {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}
__device__ __forceinline__ uint64_t synth_val(uint64_t key, uint64_t idx) { return mix64(key + idx * 0x9E3779B97F4A7C15ull); }
__device__ __forceinline__ float synth_unit(uint64_t v) { return (float)(uint32_t)(v >> 40) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void synth_fill_kernel(int kind, uint64_t key, float amp, uint64_t n, void* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        switch (kind) {
            case 0: ((uint16_t*)out)[i] = f2h(__fmul_rn(__fmul_rn(__fsub_rn(synth_unit(synth_val(key, i)), 0.5f), 2.0f), amp)); break;
            case 1: ((int8_t*)out)[i] = (int8_t)((int)((uint32_t)(synth_val(key, i) >> 32) % 255u) - 127); break;
            case 2: {
                const uint32_t lo = (uint32_t)(synth_val(key, 2 * i) >> 32) & 15u;
                const uint32_t hi = (uint32_t)(synth_val(key, 2 * i + 1) >> 32) & 15u;
                ((uint8_t*)out)[i] = (uint8_t)(lo | (hi << 4));
            } break;
            case 3: ((uint16_t*)out)[i] = f2h(__fmul_rn(amp, __fadd_rn(0.5f, synth_unit(synth_val(key, i))))); break;
            default: ((uint16_t*)out)[i] = f2h(__fadd_rn(1.0f, __fmul_rn(0.1f, __fsub_rn(synth_unit(synth_val(key, i)), 0.5f)))); break;
        }
    }
}

hipError_t launch_synth_fill(hipStream_t s, int kind, uint64_t seed, uint32_t tensor_id, uint32_t stream_id, float amp,
                             uint64_t n, void* out) {
    if (n == 0) return hipSuccess;
    // host-side copy of mix64 for the key
    auto mix = [](uint64_t z) {
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27; z *= 0x94D049BB133111EBull;
        z ^= z >> 31; return z;
    };
    const uint64_t key = mix(seed ^ ((uint64_t)tensor_id * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)stream_id * 0xD1B54A32D192ED03ull));
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, kind, key, amp, n, out);
    return hipGetLastError();
}

}  // namespace pplhip
