// Direct 7x7 / stride 2 / pad 3 stem convolution for 3-channel images (ResNet-101 `conv1`, ResNet-18/34 `conv0`:
// resnet_v1_101_flownet_deeplab.py:577-582, :112-117), 64 output channels, on v_mfma_f32_32x32x2_f32.
//
// Why not the implicit GEMM: its K axis is (tap, channel) with channels padded to 4 -- 49 x 4 = 196, rounded to 224 for
// the 32-wide K steps: a third of the matrix-core work multiplies zeros (70 TFLOP/s "useful" on these layers), and the
// loader fetches every input pixel 49/4 times through the cache hierarchy.  Here
//   * a block stages the input window of its 8 x 64 output pixels (21 rows x 133 pixels) ONCE into LDS, packed to
//     3 floats per pixel, so the (kx, channel) pairs of one kernel row are 21 CONSECUTIVE floats: the A operand of
//     K step (ky, pair) is the single float at  base(pixel) + ky*row + 2*pair + (lane >> 5)  -- one ds_read_b32 with
//     an immediate offset per MFMA pair, no address arithmetic in the loop.  K = 7 x 22 = 154 (one zero per kernel row
//     instead of 49 + 28);
//   * the whole weight matrix lives in REGISTERS for the lifetime of the block (154 values per lane: K/2 steps x 2
//     output-channel tiles, pre-arranged per lane on the host), so the loop is ds_read + MFMA only;
//   * blocks are persistent (one per CU, one wavefront per SIMD with the full 512-register budget: 154 weight + 128
//     accumulator registers) and walk the output tiles: weights are fetched once per block, the next tile's window is
//     loaded into registers while the current one is computed and written to the other LDS stage afterwards;
//   * the MFMA takes the weights as the A operand: a lane ends up with 4 consecutive output channels of one pixel per
//     accumulator quad -> 16-byte stores, scale/shift (BatchNorm) + ReLU fused.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "conv_common.h"

namespace {
constexpr int OTH = 8, OTW = 64;                 // output tile of a block
constexpr int IRH = 2 * OTH + 5, IRW = 2 * OTW + 5;   // input window: 21 x 133 pixels
constexpr int RWS = IRW * 3 + 1;                 // floats per staged row (+1: the zero-weight 22nd element of the last pixel)
constexpr int KS = 77;                           // K steps of 2: 7 kernel rows x 11 pairs
constexpr size_t STEM_LDS = (size_t)(2 * (IRH * RWS + 4) + 128) * sizeof(float);
}

__global__ __launch_bounds__(256, 1) void conv_stem_f32_kernel(ConvParams p, int tiles_x, int tiles_y, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [2][IRH * RWS + 4]
    constexpr int STAGE = IRH * RWS + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kk = lane >> 5;

    // ---- weights -> registers: wst[(s*2 + j)*64 + lane], s = K step, j = output-channel tile ----
    float wreg[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        wreg[s][0] = p.w[(s * 2 + 0) * 64 + lane];
        wreg[s][1] = p.w[(s * 2 + 1) * 64 + lane];
    }
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    float* ssc = smem + 2 * STAGE;            // [64 scale | 64 shift]
    if (tid < 64) { ssc[tid] = p.scale[tid]; ssc[64 + tid] = p.shift[tid]; }

    // input window of tile t: NHWC4 pixels, 11 per thread, all in flight together; written to LDS packed to 3 floats
    constexpr int NLD = (IRH * IRW + 255) / 256;
    f32x4 v[NLD];
    auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
        n = t / (tiles_x * tiles_y);
        const int r0 = t - n * tiles_x * tiles_y, ty = r0 / tiles_x;
        oy0 = ty * OTH; ox0 = (r0 - ty * tiles_x) * OTW;
    };
    auto load_window = [&](int t) {
        int n, oy0, ox0;
        tile_origin(t, n, oy0, ox0);
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        const unsigned kill = t < ntiles ? 0u : OOB;      // past the last tile: every offset out of range (no branch)
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            const int ry = i / IRW, rx = i - ry * IRW;
            const int iy = iy0 + ry, ix = ix0 + rx;
            const bool ok = i < IRH * IRW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            v[k] = buf_load4(xr, (ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.xCs) * 4) : OOB) | kill);
        }
    };
    auto store_window = [&](int buf) {
        float* sm = smem + buf * STAGE;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            if (i < IRH * IRW) {
                const int ry = i / IRW, rx = i - ry * IRW;
                float* d = sm + ry * RWS + rx * 3;
                d[0] = v[k][0]; d[1] = v[k][1]; d[2] = v[k][2];
            }
        }
        if (tid < IRH) sm[tid * RWS + IRW * 3] = 0.f;      // the 22nd element behind the last pixel of every row
    };

    int t = blockIdx.x, cur = 0;
    load_window(t);
    store_window(0);
    __syncthreads();
    for (; t < ntiles; t += gridDim.x, cur ^= 1) {
        int n, oy0, ox0;
        tile_origin(t, n, oy0, ox0);
#if !defined(STEM_ABL) || STEM_ABL != 2
        load_window(t + gridDim.x);           // the next tile's window travels while this one is computed
#endif

        // ---- this wavefront: output rows 2*wave, 2*wave+1 of the tile (one per pass), 64 columns = 2 pixel tiles of 32 ----
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            f32x16 acc[2][2];                 // [column half][channel tile]
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][j][e] = 0.f;
            const float* bx0 = smem + cur * STAGE + (2 * (2 * wave + half)) * RWS + (2 * col) * 3 + kk;
            const float* bx1 = bx0 + 64 * 3;
            // pixel operands run 3 K steps ahead of the MFMAs that consume them (ring of 4 registers per pixel tile)
            auto xoff = [](int s_) { return (s_ / 11) * RWS + 2 * (s_ % 11); };
            float xa[4], xb[4];
#pragma unroll
            for (int s = 0; s < 3; ++s) { xa[s] = bx0[xoff(s)]; xb[s] = bx1[xoff(s)]; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 3 < KS) {
                    xa[(s + 3) & 3] = bx0[xoff(s + 3)];
                    xb[(s + 3) & 3] = bx1[xoff(s + 3)];
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s][0], xa[s & 3], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s][1], xa[s & 3], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s][0], xb[s & 3], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s][1], xb[s & 3], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            // ---- epilogue: col = lane & 31 -> pixel, row = (e&3) + 8*(e>>2) + 4*kk -> channel ----
            const int oy = oy0 + 2 * wave + half;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int ox = ox0 + a * 32 + col;
                const bool ok = oy < p.Ho && ox < p.Wo;
                const unsigned pix = (unsigned)((n * p.Ho + oy) * p.Wo + ox);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = j * 32 + 8 * g + 4 * kk;
                        // scale / shift come from LDS (staged once per block): a global load here would sit behind the
                        // output stores in the in-order vector-memory counter and make every store wait for the previous one
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(ssc + co), sf = *reinterpret_cast<const f32x4*>(ssc + 64 + co);
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float u = acc[a][j][4 * g + e] * sc[e] + sf[e];
                            if (p.act == 1) u = fmaxf(u, 0.f);
                            else if (p.act == 2) u = u > 0.f ? u : u * p.slope;
                            o[e] = u;
                        }
                        buf_store4(yr, (ok && co < p.Cout_store) ? (pix * p.yCs + co) * 4u : OOB, o);
                    }
            }
        }
#if !defined(STEM_ABL) || STEM_ABL != 2
        store_window(cur ^ 1);                // nobody reads that stage: its last readers passed the previous barrier
#endif
        // LDS-only barrier: __syncthreads() would also wait for this tile's 128 KB of output stores to retire
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// layers the stem kernel takes: 7x7 / stride 2 / pad 3 on a 3-channel NHWC4 image, 64 output channels, single output
bool conv_stem_eligible(const ConvParams& p)
{
    return !p.deconv2x && !p.f16 && !p.narrow && p.kh == 7 && p.kw == 7 && p.sh == 2 && p.sw == 2 && p.dh == 1 && p.dw == 1 &&
           p.ph == 3 && p.pw == 3 && p.Cin == 4 && p.Cout_store == 64 && !p.res && !p.y2;
}

// OIHW (64, 3, 7, 7) -> wst[(s*2 + j)*64 + lane]: the value lane (col = lane & 31, kk = lane >> 5) feeds into K step
// s = ky*11 + pair for channel tile j, i.e. w[32j + col][c][ky][kx] with (kx, c) = divmod(2*pair + kk, 3); 0 past kx = 6
void conv_stem_pack(const float* w, int Cout, float* out)
{
    for (int s = 0; s < KS; ++s)
        for (int j = 0; j < 2; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int ky = s / 11, e = 2 * (s % 11) + (lane >> 5), kx = e / 3, c = e % 3, co = 32 * j + (lane & 31);
                out[(s * 2 + j) * 64 + lane] = (kx < 7 && co < Cout) ? w[((co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
            }
}

int conv_stem_pack_floats() { return KS * 2 * 64; }

hipError_t launch_conv_stem(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    if (!conv_stem_eligible(p) || !p.wstem) return hipErrorInvalidValue;
    p.w = p.wstem;
    const int N = p.M / (p.Ho * p.Wo);
    const int tiles_x = (p.Wo + OTW - 1) / OTW, tiles_y = (p.Ho + OTH - 1) / OTH;
    const int ntiles = N * tiles_x * tiles_y;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = ntiles < cus ? ntiles : cus;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem_f32_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)STEM_LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(conv_stem_f32_kernel, dim3(grid), dim3(256), STEM_LDS, st, p, tiles_x, tiles_y, ntiles);
    return hipGetLastError();
}
