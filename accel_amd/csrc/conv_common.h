// Device helpers shared by the convolution kernels (conv_igemm.hip, conv_wino.hip, conv_stem.hip): vector types,
// buffer-resource loads / stores (hardware bounds check = zero padding without branches), small exact division.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <set>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Buffer resources: out-of-range offsets return 0 on loads and are dropped on
// stores, so image borders, ragged tiles and padded K need no branches.
#define ACCEL_BUF_FLAGS 0x00020000   // gfx9 raw buffer, 32-bit data format
#define OOB 0xFFFFFFFFu

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, ACCEL_BUF_FLAGS);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ f32x3 buf_load3(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, 0);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, off, 0, 0);
}

// exact k -> (tap, ci) and tap -> (ky, kx) without integer division
__device__ __forceinline__ void divmod_small(int a, int d, float inv_d, int& q, int& r)
{
    q = (int)((float)a * inv_d);
    r = a - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (device, kernel): set once per pair, under a lock -- a second GPU in the
// process (the reference's multi-context executor group) or two threads finalising plans must each find it set before a launch
// that asks for more than 64 KB of dynamic LDS.  (Called from the launchers; the first launch of every kernel variant happens
// eagerly at plan finalisation, never inside a stream capture.)
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}
