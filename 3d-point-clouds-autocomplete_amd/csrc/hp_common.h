// Shared device helpers for the HyperPocket gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HP_API extern "C" __attribute__((visibility("default")))

#define HP_WAVE 64

// Launch epilogue: the reference's launchers either check nothing (nndistance.cu:131-160) or throw
// std::runtime_error (approxmatch.cu:334-337).  The C-ABI returns the hipError_t instead.
#define HP_RETURN_LAST_ERROR() return (int)hipGetLastError()

#define HP_CHECK_ARG(cond) \
    do {                   \
        if (!(cond)) return -1; \
    } while (0)

// stream_order.hip: everything enqueued on `to` afterwards starts behind everything enqueued on `from` so far (an event per
// (device, from) under a mutex).  0 or a hipError_t.
int hp_order_streams(hipStream_t from, hipStream_t to);

namespace hp {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, HP_WAVE);
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, HP_WAVE);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, HP_WAVE));
    return v;
}

// Block-wide sum in a fixed order (wave partials combined by wave 0 in wave order): deterministic.
// `scratch` must hold blockDim.x/64 elements.  Result valid in thread 0.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T r = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; ++w) r += scratch[w];
    __syncthreads();
    return r;
}

// squared distance with the fma chain the oracle uses: fma(dz,dz,fma(dy,dy,dx*dx))
__device__ __forceinline__ float sqdist(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

}  // namespace hp
