// fp32 MFMA ceiling probe: registers only vs LDS-fed fragments vs +barrier (the 152 TFLOP/s figure in DESIGN.md).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_ubench.hip -o mfma_ubench && ./mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int V, int MI, int NI, int RND, int CH = 1>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDK = 36;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 256 * LDK; i += 256) { unsigned h = (unsigned)(i + blockIdx.x * 7919) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; smem[i] = RND ? ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) : (float)((i * 7) % 13) * 0.01f; }
    __syncthreads();
    f32x16 acc[CH][MI][NI];
    for (int c = 0; c < CH; ++c) for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int e = 0; e < 16; ++e) acc[c][i][j][e] = 0.f;
    const float* a = smem + ((wave >> 1) * MI * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    const float* b = smem + 128 * LDK + ((wave & 1) * NI * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    f32x4 fa[2][MI], fb[2][NI];
    for (int i = 0; i < MI; ++i) fa[0][i] = fa[1][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK);
    for (int j = 0; j < NI; ++j) fb[0][j] = fb[1][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK);
    const float* a0 = a; const float* b0 = b;
    for (int it = 0; it < iters; ++it) {
        a = a0 + (it & 1) * 256 * LDK; b = b0 + (it & 1) * 256 * LDK;   // alternate LDS buffers like the real kernel
        if (V >= 1) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = t & 1, n = c ^ 1;
            if (V >= 1 && t + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[n][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK + (t + 1) * 8);
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[n][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK + (t + 1) * 8);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[r % CH][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][r], fb[c][j][r], acc[r % CH][i][j], 0, 0, 0);
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int e = 0; e < 16; ++e) s += acc[c][i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V, int MI, int NI, int RND, int CH = 1>
int run(const char* name, int blocks, int iters)
{
    float* out; CK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t lds = 2 * 256 * 36 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V, MI, NI, RND, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k<V, MI, NI, RND, CH>), dim3(blocks), dim3(256), lds, 0, out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<V, MI, NI, RND, CH>), dim3(blocks), dim3(256), lds, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * 4 * iters * 16.0 * MI * NI * (2.0 * 32 * 32 * 2);
    printf("%-40s blocks %4d  %8.3f ms  %7.1f TF\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out); return 0;
}

int main()
{
    for (int blocks : {512, 1024}) {
        run<0, 2, 2, 0>("regs only, MI=NI=2, regular data", blocks, 4000);
        run<0, 2, 2, 1>("regs only, MI=NI=2, random data", blocks, 4000);
        run<2, 2, 2, 0>("+ds_read +barrier, MI=NI=2, regular data", blocks, 4000);
        run<2, 2, 2, 1>("+ds_read +barrier, MI=NI=2, random data", blocks, 4000);
        run<2, 1, 1, 1>("+ds_read +barrier, MI=NI=1, random data", blocks, 8000);
        run<2, 1, 1, 1, 2>("+ds_read +barrier, MI=NI=1, 2 acc chains", blocks, 8000);
        run<2, 1, 1, 1, 4>("+ds_read +barrier, MI=NI=1, 4 acc chains", blocks, 8000);
        run<2, 2, 1, 1, 1>("+ds_read +barrier, MI=2 NI=1, 1 chain each", blocks, 4000);
        run<2, 2, 1, 1, 2>("+ds_read +barrier, MI=2 NI=1, 2 chains each", blocks, 4000);
    }
    return 0;
}
