// Counter-based synthetic data generator, bit-identical to oracle/oracle.py:synth_uniform /
// synth_small_int.  Used by bench.py and the GPU tests to build multi-GB corpora directly in HBM
// (host generation of 4 GB took 35 s in the survey container; SURVEY.md section 7 "hard parts").
#include "common.h"

namespace rl {

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float synth_value(uint64_t idx, uint64_t stream_key, int kind) {
    const uint32_t u = (uint32_t)(splitmix64(idx ^ stream_key) >> 40);  // top 24 bits
    if (kind == RL_SYNTH_SMALL_INT) return (float)(u % 7u) - 3.0f;
    return (float)u * 1.1920928955078125e-07f - 1.0f;  // u * 2^-23 - 1, exact in fp32
}

__global__ __launch_bounds__(256) void synth_kernel(float* __restrict__ dst, uint64_t start, int64_t count,
                                                     uint64_t stream_key, int kind) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += stride) {
        if (i + 4 <= count && ((reinterpret_cast<uintptr_t>(dst + i) & 15) == 0)) {
            float4 v;
            v.x = synth_value(start + i + 0, stream_key, kind);
            v.y = synth_value(start + i + 1, stream_key, kind);
            v.z = synth_value(start + i + 2, stream_key, kind);
            v.w = synth_value(start + i + 3, stream_key, kind);
            *reinterpret_cast<float4*>(dst + i) = v;
        } else {
            for (int64_t j = i; j < count && j < i + 4; ++j) dst[j] = synth_value(start + j, stream_key, kind);
        }
    }
}

int launch_synth(float* dst, int64_t start, int64_t count, uint64_t seed, int kind, hipStream_t s) {
    if (count <= 0) return RL_OK;
    const uint64_t stream_key = splitmix64(seed * 0xD1342543DE82EF95ull);
    const int64_t quads = (count + 3) / 4;
    const int blocks = (int)std::min<int64_t>((quads + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(synth_kernel, dim3(blocks), dim3(256), 0, s, dst, (uint64_t)start, count, stream_key, kind);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
