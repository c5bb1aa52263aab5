// Do the fp32 matrix pipe and the fp32 vector pipe add up?  Waves 0..MW-1 of each block run an MFMA-only loop,
// the remaining waves a v_pk_fma_f32-style FMA loop (registers only).  Reports the two rates alone and together.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int MW, int VW>
__global__ __launch_bounds__(64 * (MW + VW)) void k(float* out, int iters)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < MW) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        float a = threadIdx.x * 1e-3f, b = 1.0f - a;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
            a += 1e-6f;
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) r += acc[i][e];
    } else {
        f32x2 v[16];
        for (int i = 0; i < 16; ++i) v[i] = f32x2{threadIdx.x * 1e-3f + i, 1.0f};
        f32x2 m = {1.0001f, 0.9999f}, c = {1e-4f, -1e-4f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) v[u & 15] = v[u & 15] * m + c;      // 64 packed FMAs = 128 FMA per lane
        }
        for (int i = 0; i < 16; ++i) r += v[i].x + v[i].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MW, int VW>
int run(const char* name, int blocks, int iters)
{
    float* out; CK(hipMalloc(&out, (size_t)blocks * 64 * (MW + VW) * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MW, VW>), dim3(blocks), dim3(64 * (MW + VW)), 0, 0, out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MW, VW>), dim3(blocks), dim3(64 * (MW + VW)), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mf = (double)blocks * MW * iters * 16.0 * (2.0 * 32 * 32 * 2);
    const double vf = (double)blocks * VW * iters * 64.0 * 64 * 2 * 2;
    printf("%-34s %8.3f ms   MFMA %7.1f TF   VALU %7.1f TF   total %7.1f TF\n", name, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
    CK(hipFree(out)); return 0;
}

int main()
{
    run<4, 0>("MFMA only, 4 waves/block", 512, 4000);
    run<0, 4>("VALU only, 4 waves/block", 512, 4000);
    run<4, 4>("4 MFMA + 4 VALU waves/block", 512, 4000);
    run<4, 4>("4 MFMA + 4 VALU waves/block x1024", 1024, 4000);
    run<4, 8>("4 MFMA + 8 VALU waves/block", 512, 4000);
    return 0;
}
