// Direct 7x7 / stride 2 / pad 3 stem convolution for 3-channel images (ResNet-101 `conv1`, ResNet-18/34 `conv0`:
// resnet_v1_101_flownet_deeplab.py:577-582, :112-117), 64 output channels, on v_mfma_f32_32x32x2_f32.
//
// Why not the implicit GEMM: its K axis is (tap, channel) with channels padded to 4 -- 49 x 4 = 196, rounded to 224 for
// the 32-wide K steps: a third of the matrix-core work multiplies zeros (70 TFLOP/s "useful" on these layers), and the
// loader fetches every input pixel 49/4 times through the cache hierarchy.  Here
//   * a block stages the input window of its 8 x 64 output pixels (21 rows x 133 pixels) ONCE into LDS, packed to
//     3 floats per pixel, so the (kx, channel) pairs of one kernel row are a run of 21 floats: the pixel operand of
//     K step (ky, pair) is the single float at  base(pixel) + ky*row + 2*pair + pair/3 + (lane >> 5)  -- one
//     ds_read_b32 with an immediate offset per MFMA, no address arithmetic in the loop.  K = 7 x 22 = 154 (one zero
//     per kernel row instead of 49 + 28).  Pixel PAIRS are padded from 6 to 7 floats (the "+ pair/3" skips the pads;
//     a run always starts on an even pixel): consecutive output pixels are then 7 floats apart, distinct banks for the
//     32 lanes a ds_read_b32 serves per cycle -- the unpadded stride of 6 was a 2-way conflict on every read;
//   * the weights live in REGISTERS for the lifetime of the block: a wavefront owns ONE 32-channel tile (77 values per
//     lane, pre-arranged per lane on the host) and 4 output rows, so the loop is ds_read + MFMA only;
//   * blocks are persistent and TWO are resident per CU (two wavefronts per SIMD from different blocks, hence out of
//     phase: one block's staging / epilogue runs under the other's MFMAs); they walk the output tiles, the next tile's
//     window is loaded into registers while the current one is computed and written to the other LDS stage afterwards
//     (versions with both channel tiles in one wavefront -- 154 weight registers -- either spilled or fell back to one
//     wavefront per SIMD and stopped at 62 % MFMA-busy: 849 us per 8 images);
//   * the MFMA takes the weights as the A operand: a lane ends up with 4 consecutive output channels of one pixel per
//     accumulator quad -> 16-byte stores, 4x fewer store instructions than with one channel per lane; scale/shift
//     (BatchNorm) + ReLU fused with packed fmas.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "kernels.h"
#include "conv_common.h"
#include "range.h"

namespace {
constexpr int OTH = 8, OTW = 64;                 // output tile of a block
constexpr int IRH = 2 * OTH + 5, IRW = 2 * OTW + 5;   // input window: 21 x 133 pixels
constexpr int RWS = ((IRW + 1) / 2) * 7;         // floats per staged row: pixel PAIRS of 6 floats + 1 pad (see the header)
constexpr int KS = 77;                           // K steps of 2: 7 kernel rows x 11 pairs
constexpr size_t STEM_LDS = (size_t)(2 * (IRH * RWS + 4) + 128) * sizeof(float);      // 79.3 KB: two blocks per CU
}

__global__ __launch_bounds__(256, 2) void conv_stem_f32_kernel(ConvParams p, int tiles_x, int tiles_y, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [2][IRH * RWS + 4] + [128]
    constexpr int STAGE = IRH * RWS + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kk = lane >> 5;
    const int jt = wave & 1, rg = wave >> 1;        // this wavefront: output-channel tile jt (32 channels), output rows 4*rg .. 4*rg+3

    // ---- weights -> registers: wst[(s*2 + jt)*64 + lane], s = K step (77 values per lane) ----
    float wreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wreg[s] = p.w[(s * 2 + jt) * 64 + lane];
    float* ssc = smem + 2 * STAGE;            // [64 scale | 64 shift]
    if (tid < 64) { ssc[tid] = p.scale[tid]; ssc[64 + tid] = p.shift[tid]; }
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    unsigned rmax = 0u;

    // input window of tile t: NHWC4 pixels, 11 per thread, all in flight together; written to LDS packed to 3 floats
    constexpr int NLD = (IRH * IRW + 255) / 256;
    f32x3 v[NLD];                            // the 4th (padding) channel of a pixel is never fetched
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
        // the per-thread window coordinates are recomputed for every tile: hoisted out of the tile loop they would cost
        // 22 registers, spill, and every scratch reload would wait for ALL outstanding output stores (one in-order counter)
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid_ + 256 * k;
            const int ry = i / IRW, rx = i - ry * IRW;
            const int iy = iy0 + ry, ix = ix0 + rx;
            const bool ok = i < IRH * IRW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            v[k] = buf_load3(xr, (ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.xCs) * 4) : OOB) | kill);
        }
    };
    auto store_window = [&](int buf) {
        float* sm = smem + buf * STAGE;
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid_ + 256 * k;
            if (k < NLD - 1 || i < IRH * IRW) {      // only the last round runs past the window
                const int ry = i / IRW, rx = i - ry * IRW;
                float* d = sm + ry * RWS + (rx >> 1) * 7 + (rx & 1) * 3;
                d[0] = v[k][0]; d[1] = v[k][1]; d[2] = v[k][2];
            }
        }
        if (tid < IRH) sm[tid * RWS + (IRW >> 1) * 7 + 3] = 0.f;      // the (zero-weight) 22nd element behind the last pixel of every row
    };

#ifdef STEM_TIMING
    long long tacc[5] = {0, 0, 0, 0, 0};
#endif
    const bool leaky = p.act == 2;
    const float floor_ = p.act == 1 ? 0.f : -__builtin_inff();
    int t = blockIdx.x, cur = 0;
    load_window(t);
    store_window(0);
    __syncthreads();
    // a wait the compiler can see: without it the loop header inherits "weight loads may be pending" from this prologue
    // and every tile waits for the previous tile's output stores before its first MFMA
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    for (; t < ntiles; t += gridDim.x, cur ^= 1) {
        int n, oy0, ox0;
        tile_origin(t, n, oy0, ox0);
#ifdef STEM_TIMING
        long long tq0 = clock64();
#endif
        load_window(t + gridDim.x);           // the next tile's window travels while this one is computed
#ifdef STEM_TIMING
        long long tq1 = clock64(); tacc[0] += tq1 - tq0;
#endif

        // ---- two passes of 2 output rows x 64 columns = 4 pixel tiles of 32, one channel tile ----
#pragma unroll       // NOT a real loop: the compiler drains the vector-memory counter in front of any loop with stores
        for (int half = 0; half < 2; ++half) {
            f32x16 acc[4];                    // [row of the pass x column half]
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
#ifdef STEM_TIMING
            long long tp0 = clock64();
#endif
            const float* bx[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                bx[a] = smem + cur * STAGE + (2 * (4 * rg + 2 * half + (a >> 1))) * RWS + (col + 32 * (a & 1)) * 7 + kk;
            // pixel operands run 2 K steps ahead of the MFMAs that consume them (ring of 4 registers per pixel tile)
            auto xoff = [](int s_) { return (s_ / 11) * RWS + 2 * (s_ % 11) + (s_ % 11) / 3; };      // + pads skipped
            float xq[4][4];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 4; ++a) xq[a][s] = bx[a][xoff(s)];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 2 < KS) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) xq[a][(s + 2) & 3] = bx[a][xoff(s + 2)];
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s], xq[a][s & 3], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
#ifdef STEM_TIMING
            asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
            tacc[1] += clock64() - tp0;
#endif
            __builtin_amdgcn_sched_barrier(0);
            // The next window goes to the other LDS stage (nobody reads it: its last readers passed the previous
            // barrier) after the FIRST pass: its loads, and the previous tile's stores queued in front of them in the
            // in-order vector-memory counter, have had a whole pass to retire.  After the second epilogue the wait would
            // have to drain this tile's stores as well (more than the counter's 63 outstanding operations).
            if (half == 0) store_window(cur ^ 1);
            // ---- epilogue: col = lane & 31 -> pixel, row = (e&3) + 8*(e>>2) + 4*kk -> channel: 16-byte stores.
            // Kept to as FEW INSTRUCTIONS as possible (packed fma, one max per value): while the other wavefront of the
            // SIMD streams MFMAs this one is issued roughly one instruction per MFMA slot (measured ~50 cycles each), so
            // the epilogue's length in instructions, not its bytes, decides whether it fits under the other's pass.
            f32x2 sc2[8], sf2[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(ssc + jt * 32 + 8 * g + 4 * kk);
                const f32x4 f4 = *reinterpret_cast<const f32x4*>(ssc + 64 + jt * 32 + 8 * g + 4 * kk);
                sc2[2 * g] = f32x2{s4[0], s4[1]}; sc2[2 * g + 1] = f32x2{s4[2], s4[3]};
                sf2[2 * g] = f32x2{f4[0], f4[1]}; sf2[2 * g + 1] = f32x2{f4[2], f4[3]};
            }
            auto finish = [&](auto leaky_tag) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oy = oy0 + 4 * rg + 2 * half + (a >> 1), ox = ox0 + (a & 1) * 32 + col;
                const bool ok = oy < p.Ho && ox < p.Wo;
                const unsigned off0 = ok ? (unsigned)((((n * p.Ho + oy) * p.Wo + ox) * p.yCs + jt * 32 + 4 * kk) * 4) : OOB;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x2 lo = f32x2{acc[a][4 * g], acc[a][4 * g + 1]} * sc2[2 * g] + sf2[2 * g];
                    f32x2 hi = f32x2{acc[a][4 * g + 2], acc[a][4 * g + 3]} * sc2[2 * g + 1] + sf2[2 * g + 1];
                    f32x4 o = {lo[0], lo[1], hi[0], hi[1]};
                    if (decltype(leaky_tag)::value) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * p.slope;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], floor_);       // ReLU (0) or nothing (-inf)
                    }
                    buf_store4(yr, off0 | (unsigned)(32 * g), o);      // off0 is a multiple of 128 bytes, or all ones
                    if (p.yr && ok) {      // range slot of the output (range.h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(o[e]); rmax = b > rmax ? b : rmax; }
                    }
                }
            }
            };
            if (leaky) finish(std::true_type{}); else finish(std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef STEM_TIMING
        long long tq3 = clock64(); tacc[2] += tq3 - tq1;   // both passes incl. epilogues
#endif
        // LDS-only barrier: __syncthreads() would also wait for this tile's 128 KB of output stores to retire
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef STEM_TIMING
        long long tq4 = clock64(); tacc[3] += tq4 - tq3; tacc[4] += 1;
#endif
    }
#ifdef STEM_TIMING
    if (blockIdx.x == 300 && lane == 0) for (int i = 0; i < 5; ++i) p.y[wave * 8 + i] = (float)tacc[i];
#endif
    if (p.yr) range_note_wave(p.yr, rmax, (unsigned)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
}

// layers the stem kernel takes: 7x7 / stride 2 / pad 3 on a 3-channel NHWC4 image, 64 output channels, single output
bool conv_stem_eligible(const ConvParams& p)
{
    return !p.deconv2x && (!p.f16 || p.f16 == 3) && !p.narrow && p.kh == 7 && p.kw == 7 && p.sh == 2 && p.sw == 2 && p.dh == 1 && p.dw == 1 &&
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
    const int grid = ntiles < 2 * cus ? ntiles : 2 * cus;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_stem_f32_kernel), STEM_LDS); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_stem_f32_kernel, dim3(grid), dim3(256), STEM_LDS, st, p, tiles_x, tiles_y, ntiles);
    return hipGetLastError();
}
