// What does "note the largest |value| a kernel stored" cost a byte mover?  A copy over 268 MB (max-pool sized) in which every
// wavefront reduces max |v| of what it stored and publishes it, in the variants considered for the range slots of the fp16x2 form
// (DESIGN.md 5): hipcc --offload-arch=gfx950 -O3 amax_ubench.hip -o amax_ubench
//   0 plain copy                      1 one atomicMax per wavefront, same address
//   2 agent-scope load of the slot first, atomicMax only when larger          3 as 2, one atomic per BLOCK (LDS reduce)
//   4 one atomicMax per wavefront into one of 32 sub-slots of one 128-byte line    5 the same, sub-slots 256 bytes apart
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned wave_max(unsigned u)
{
    for (int o = 32; o; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
    return u;
}

template <int MODE>
__global__ __launch_bounds__(256) void copyk(const f4* __restrict__ src, f4* __restrict__ dst, size_t n16, unsigned* slot)
{
    __shared__ unsigned sm[4];
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
        f4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + 256 * k < n16) v[k] = src[i + 256 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + 256 * k < n16) {
            dst[i + 256 * k] = v[k];
            if (MODE) m = fmaxf(fmaxf(fmaxf(m, fabsf(v[k][0])), fmaxf(fabsf(v[k][1]), fabsf(v[k][2]))), fabsf(v[k][3]));
        }
    }
    if (MODE == 0) return;
    unsigned u = wave_max(__float_as_uint(m));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (MODE == 3) {
        if (lane == 0) sm[wave] = u;
        __syncthreads();
        if (threadIdx.x == 0) {
            u = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
            if (u > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, u);
        }
        return;
    }
    if (lane) return;
    if (MODE == 1) atomicMax(slot, u);
    else if (MODE == 2) { if (u > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, u); }
    else if (MODE == 4) atomicMax(slot + ((blockIdx.x * 4 + wave) & 31), u);
    else atomicMax(slot + 64 * ((blockIdx.x * 4 + wave) & 31), u);
}

template <int MODE>
void run(const char* name, const f4* s, f4* d, size_t n16, int blocks, unsigned* slot)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    for (int r = 0; r < 8; ++r) {
        hipMemsetAsync(slot, 0, 64 * 32 * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((copyk<MODE>), dim3(blocks), dim3(256), 0, 0, s, d, n16, slot);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r > 1) { sum += ms; if (ms < best) best = ms; }
    }
    unsigned h[64 * 32]; hipMemcpy(h, slot, sizeof h, hipMemcpyDeviceToHost);
    unsigned mx = 0; for (unsigned v : h) mx = v > mx ? v : mx;
    printf("%-58s blocks %6d: best %7.1f us  mean %7.1f us   max %g\n", name, blocks, best * 1e3, sum / 6 * 1e3, *(float*)&mx);
}

// the worst case for the pre-check: values GROW along the launch order, every wavefront sets a new record
__global__ void fill(float* p, size_t n, int ramp)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float r = (float)(h & 0xFFFF) / 65536.f;
        p[i] = ramp ? r + (float)i * 1e-6f : r * 4.f - 2.f;
    }
}

int main()
{
    const size_t bytes = (size_t)268 << 20, n16 = bytes / 16;
    f4 *s, *d; unsigned* slot;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMalloc(&slot, 64 * 32 * 4);
    for (int ramp = 0; ramp < 2; ++ramp) {
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float*)s, bytes / 4, ramp);
        printf("---- data: %s\n", ramp ? "growing along the launch order (every wavefront a new record)" : "uniform random");
        for (int blocks : {2048, 16384, 65536}) {
            run<0>("plain copy", s, d, n16, blocks, slot);
            run<1>("atomicMax per wavefront, one address", s, d, n16, blocks, slot);
            run<2>("agent-scope load first, atomicMax only when larger", s, d, n16, blocks, slot);
            run<3>("the same per block (LDS reduce)", s, d, n16, blocks, slot);
            run<4>("atomicMax per wavefront, 32 sub-slots in one line", s, d, n16, blocks, slot);
            run<5>("atomicMax per wavefront, 32 sub-slots 256 bytes apart", s, d, n16, blocks, slot);
        }
    }
    hipDeviceSynchronize();
    return 0;
}
