// 7x7 / stride 2 / pad 3 stem convolution of a 3-channel image (ResNet-101 `conv1`, ResNet-18/34 `conv0`:
// resnet_v1_101_flownet_deeplab.py:577-582, :112-117) on the BF16 matrix cores with the exact three-term split of conv_b3r.hip
// (launch geometry 51): fp32 values, six v_mfma_f32_32x32x16_bf16 products per multiply-add, fp32 accumulate.
//
// Why: conv_stem_f32_kernel (geometry 50) runs at 0.65-0.67 of the fp32-MFMA peak and cannot go much further (the fp32 matrix
// instructions share the vector ALUs): 671-744 us per 8 images, five launches per step.  Six bf16 products cost 6/16 of
// the fp32 matrix time and leave the vector ALUs to the split.
//
// Layout of the contraction: K = 7 kernel rows x 24 (21 = 7 taps x 3 channels of one input row, padded to three chunks of 8) = 168,
// padded to 11 steps of 16.  A chunk of 8 consecutive K values lies inside ONE input row of the staged window, which is packed to
// 3 floats per pixel: the pixel operand of (output pixel, chunk) is 32 contiguous bytes at  row (2 oy + ky) , float 6 ox + 8 part  --
// four ds_read_b64 (8-byte aligned; consecutive output pixels are 24 bytes apart: the 32 lanes of a half wavefront touch 64
// distinct banks), split into its three bf16 planes by the wavefront that multiplies it.  The values a chunk reads past the 21st
// element of a row (the next pixel's channels) meet zero weights.
//   * block = 8 x 64 output pixels x 64 channels, 8 wavefronts; wavefront w owns output row w: 2 pixel tiles x 2 channel tiles,
//     24 matrix instructions per K step against 8 ds_read_b64 + 6 ds_read_b128 + the split of 16 values per lane;
//   * the weights (three bf16 planes in MFMA fragment order, 66 KB) are loaded into LDS once per persistent block and read as
//     fragments (the fp32 kernel keeps them in registers: 3 planes x 11 steps x 2 tiles would not fit);
//   * the input window of the NEXT tile is fetched into registers while the current one is multiplied and written to the other LDS
//     stage after the K loop (two stages of 33 KB);
//   * the MFMA takes the weights as the A operand: a lane ends up with 4 consecutive output channels of one pixel per accumulator
//     quad -> 16-byte stores; scale / shift (BatchNorm) + ReLU fused.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <vector>
#include <cstring>
#include <cstdint>
#include "kernels.h"
#include "conv_common.h"
#include "range.h"

namespace {
constexpr int OTH = 8, OTW = 64;                        // output tile of a block
constexpr int IRW = 2 * OTW + 5;                         // input window: (2 rows + 5) x 133 pixels
constexpr int RWS = 408;                                 // floats per staged row (133 x 3 = 399, + the overrun of the last chunk; even: 8-byte reads)
constexpr int NSTEP = 11;                                // K steps of 16 (22 chunks of 8; chunk 21 is all padding)
constexpr int WBYTES = NSTEP * 2 * 3 * 64 * 16;          // weight fragments: [step][channel tile][plane][lane][8 bf16]
constexpr int WBYTES_H2 = NSTEP * 2 * 2 * 64 * 16;       // the fp16x2 form: two half planes
constexpr int IRH = 2 * OTH + 5, WIN = IRH * RWS;       // input window rows; floats per window stage
constexpr size_t stem_lds(bool h2) { return (size_t)(h2 ? WBYTES_H2 : WBYTES) + 2 * WIN * sizeof(float) + 256 * sizeof(float); }
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8sb __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_pair_sb(float v0, float v1, unsigned& q0, unsigned& q1, unsigned& q2)
{
    const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
    q0 = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = v0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u), r1 = v1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    const unsigned s0 = __builtin_bit_cast(unsigned, r0), s1 = __builtin_bit_cast(unsigned, r1);
    q1 = __builtin_amdgcn_perm(s1, s0, 0x07060302);
    const float t0 = r0 - __builtin_bit_cast(float, s0 & 0xFFFF0000u), t1 = r1 - __builtin_bit_cast(float, s1 & 0xFFFF0000u);
    q2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0), 0x07060302);
}
}

// H2: the fp16x2 form (kernels.h): two half terms per operand, three v_mfma_f32_32x32x16_f16 products per multiply-add
template <bool H2>
__global__ __launch_bounds__(512, 2) void conv_stem_b3_kernel(ConvParams p, int tiles_x, int tiles_y, int ntiles)
{
    constexpr int NPS = H2 ? 2 : 3;
    constexpr int WB = H2 ? WBYTES_H2 : WBYTES;
    constexpr int NTHR = 512;
    constexpr int TSY = OTH, TSX = OTW;      // conv rows / columns between the origins of neighbouring tiles
    // fp16x2 form: the pixel scale from the range slot of the image tensor (range.h)
    RangeScale rs; rs.s = 1.f; rs.inv = 1.f;
    if constexpr (H2) rs = range_prologue(p.xr);
    const float xs = rs.s, xinv = rs.inv;
    unsigned rmax = 0u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sb[];
    float* win = reinterpret_cast<float*>(smem_sb + WB);           // [2][WIN]
    float* ssc = win + 2 * WIN;                                    // [64 scale | 64 shift]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kk = lane >> 5;

    // ---- weights -> LDS (once per persistent block), scale / shift -------------------------------------------------------
    {
        const i32x4* src = reinterpret_cast<const i32x4*>(p.w);
        i32x4* dst = reinterpret_cast<i32x4*>(smem_sb);
        for (int i = tid; i < WB / 16; i += NTHR) dst[i] = src[i];
        if (tid < 64) { ssc[tid] = p.scale[tid] * xinv; ssc[64 + tid] = p.shift[tid]; }
    }
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);

    // input window of tile t: NHWC4 pixels, 6 per thread, all in flight together; written to LDS packed to 3 floats
    constexpr int NLD = (IRH * IRW + NTHR - 1) / NTHR;
    f32x3 v[NLD];
    auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
        n = t / (tiles_x * tiles_y);
        const int r0 = t - n * tiles_x * tiles_y, ty = r0 / tiles_x;
        oy0 = ty * TSY; ox0 = (r0 - ty * tiles_x) * TSX;
    };
    auto load_window = [&](int t) {
        int n, oy0, ox0;
        tile_origin(t, n, oy0, ox0);
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        const unsigned kill = t < ntiles ? 0u : OOB;      // past the last tile: every offset out of range (no branch)
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + NTHR * k;
            const int ry = i / IRW, rx = i - ry * IRW;
            const int iy = iy0 + ry, ix = ix0 + rx;
            const bool ok = i < IRH * IRW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            v[k] = buf_load3(xr, (ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.xCs) * 4) : OOB) | kill);
        }
    };
    auto store_window = [&](int buf) {
        float* sm = win + buf * WIN;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + NTHR * k;
            if (k < NLD - 1 || i < IRH * IRW) {
                const int ry = i / IRW, rx = i - ry * IRW;
                float* d = sm + ry * RWS + rx * 3;
                d[0] = v[k][0]; d[1] = v[k][1]; d[2] = v[k][2];
            }
        }
        // the floats behind the last pixel of a row (read by the last chunk of the last pixels, multiplied by zero weights) must be finite
        if (tid < IRH * (RWS - IRW * 3)) win[buf * WIN + (tid / (RWS - IRW * 3)) * RWS + IRW * 3 + tid % (RWS - IRW * 3)] = 0.f;
    };

    const bool leaky = p.act == 2;
    const float floor_ = p.act == 1 ? 0.f : -__builtin_inff();
    int t = blockIdx.x, cur = 0;
    load_window(t);
    store_window(0);
    __syncthreads();
    for (; t < ntiles; t += gridDim.x, cur ^= 1) {
        int n, oy0, ox0;
        tile_origin(t, n, oy0, ox0);
        load_window(t + gridDim.x);           // the next tile's window travels while this one is computed

        f32x16 acc[2][2];                     // [pixel tile of the row][channel tile]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][j][e] = 0.f;
        // this lane's pixel of pixel tile a: output row `wave`, column 32 a + col; window row 2 wave + ky, float 6 (32 a + col) + 8 part
        const float* px0 = win + cur * WIN + (2 * wave) * RWS + 6 * col;
        auto read_x = [&](int s, i32x4 (&xb)[2][NPS]) {
            int ch = 2 * s + kk;                      // chunk of 8 K values: kernel row ch / 3, part ch % 3
            if (ch > 20) ch = 20;                     // chunk 21 is all padding (zero weights): read chunk 20's floats again
            const int ky = ch / 3, part = ch - 3 * ky;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x2* q = reinterpret_cast<const f32x2*>(px0 + ky * RWS + 192 * a + 8 * part);
                const f32x2 d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
                if constexpr (H2) {
                    const float v8[8] = {d0[0], d0[1], d1[0], d1[1], d2[0], d2[1], d3[0], d3[1]};
                    f16x8sb h, l;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        h[e] = (_Float16)(v8[e] * xs);
                        l[e] = (_Float16)__builtin_fmaf(v8[e], xs, -(float)h[e]);      // exact residual, then rounded to half
                    }
                    xb[a][0] = __builtin_bit_cast(i32x4, h);
                    xb[a][NPS - 1] = __builtin_bit_cast(i32x4, l);
                    continue;
                }
                unsigned x0, x1, x2;
                split3_pair_sb(d0[0], d0[1], x0, x1, x2); xb[a][0][0] = (int)x0; xb[a][1][0] = (int)x1; xb[a][NPS - 1][0] = (int)x2;
                split3_pair_sb(d1[0], d1[1], x0, x1, x2); xb[a][0][1] = (int)x0; xb[a][1][1] = (int)x1; xb[a][NPS - 1][1] = (int)x2;
                split3_pair_sb(d2[0], d2[1], x0, x1, x2); xb[a][0][2] = (int)x0; xb[a][1][2] = (int)x1; xb[a][NPS - 1][2] = (int)x2;
                split3_pair_sb(d3[0], d3[1], x0, x1, x2); xb[a][0][3] = (int)x0; xb[a][1][3] = (int)x1; xb[a][NPS - 1][3] = (int)x2;
            }
        };
        i32x4 xq[2][2][NPS];
        read_x(0, xq[0]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + 1 < NSTEP) read_x(s + 1, xq[(s + 1) & 1]);      // the next step's pixel operands are read and split beside this step's products
            i32x4 wf[2][NPS];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < NPS; ++pl)
                    wf[j][pl] = *reinterpret_cast<const i32x4*>(smem_sb + (((s * 2 + j) * NPS + pl) * 64 + lane) * 16);
#define SB_TERM(wp, xp)                                                                                                              \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                      \
        acc[a][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8s, wf[j][wp]), __builtin_bit_cast(bf16x8s, xq[s & 1][a][xp]), acc[a][j], 0, 0, 0);
#define SH_TERM(wp, xp)                                                                                                              \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                      \
        acc[a][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8sb, wf[j][wp]), __builtin_bit_cast(f16x8sb, xq[s & 1][a][xp]), acc[a][j], 0, 0, 0);
            if constexpr (H2) { SH_TERM(1, 0) SH_TERM(0, 1) SH_TERM(0, 0) }                          // the two cross terms, then hi * hi
            else { SB_TERM(1 % NPS, 1 % NPS) SB_TERM(0, 2 % NPS) SB_TERM(2 % NPS, 0) SB_TERM(0, 1 % NPS) SB_TERM(1 % NPS, 0) SB_TERM(0, 0) }      // smallest terms first
#undef SH_TERM
#undef SB_TERM
        }
        store_window(cur ^ 1);      // nobody reads that stage: its last readers passed the barrier that ended the previous tile

        // ---- epilogue: col = lane & 31 -> pixel, row = (e & 3) + 8 (e >> 2) + 4 kk -> channel: 16-byte stores ----------------
        const int oy = oy0 + wave;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ox = ox0 + 32 * a + col;
            const bool ok = oy < p.Ho && ox < p.Wo;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned off0 = ok ? (unsigned)((((n * p.Ho + oy) * p.Wo + ox) * p.yCs + j * 32 + 4 * kk) * 4) : OOB;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(ssc + j * 32 + 8 * g + 4 * kk);
                    const f32x4 f4 = *reinterpret_cast<const f32x4*>(ssc + 64 + j * 32 + 8 * g + 4 * kk);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = acc[a][j][4 * g + e] * s4[e] + f4[e];
                        o[e] = leaky ? (u > 0.f ? u : u * p.slope) : fmaxf(u, floor_);
                        if (p.yr && ok) { const unsigned b = range_abs_bits(o[e]); rmax = b > rmax ? b : rmax; }
                    }
                    buf_store4(yr, off0 | (unsigned)(32 * g), o);      // off0 is a multiple of 128 bytes, or all ones
                }
            }
        }
        // LDS-only barrier: __syncthreads() would also wait for this tile's output stores to retire
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (p.yr) range_note_wave(p.yr, rmax, (unsigned)(blockIdx.x * 8 + wave));      // once per persistent block and wavefront
}

bool conv_stem_b3_eligible(const ConvParams& p) { return conv_stem_eligible(p); }

// OIHW (64, 3, 7, 7) -> [step][channel tile][plane][lane][8 bf16]: the fragment lane (row = channel 32 j + (lane & 31), half = lane >> 5)
// feeds into K step s: chunk ch = 2 s + half -> kernel row ky = ch / 3, elements e = 8 (ch % 3) .. + 7 of that row's 21 (kx, c) pairs
// (e = 3 kx + c; zero for e >= 21 and for the padding chunk 21), each value split exactly into three bf16 terms
void conv_stem_b3_pack(const float* w, int Cout, std::vector<unsigned short>& out)
{
    out.assign((size_t)WBYTES / 2, 0);
    for (int s = 0; s < NSTEP; ++s)
        for (int j = 0; j < 2; ++j)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int ch = 2 * s + (lane >> 5), ky = ch / 3, e = 8 * (ch % 3) + i, kx = e / 3, c = e % 3, co = 32 * j + (lane & 31);
                    float r = (ch < 21 && e < 21 && co < Cout) ? w[((co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
                    for (int pl = 0; pl < 3; ++pl) {
                        uint32_t u; memcpy(&u, &r, 4);
                        u &= 0xFFFF0000u;
                        out[((((size_t)s * 2 + j) * 3 + pl) * 64 + lane) * 8 + i] = (unsigned short)(u >> 16);
                        float tt; memcpy(&tt, &u, 4);
                        r -= tt;
                    }
                }
}

// The fp16x2 form of the same fragments (ConvParams::wstemh): w * 2^q[co] as hi + lo, two half terms; q[co] puts the channel's largest
// weight into [2^14, 2^15) and is folded into the epilogue scale by the caller.
void conv_stem_b3_pack_h2(const float* w, int Cout, std::vector<unsigned short>& out, std::vector<int>& qexp)
{
    out.assign((size_t)WBYTES_H2 / 2, 0);
    qexp.assign(64, 0);
    for (int co = 0; co < Cout && co < 64; ++co) {
        float amax = 0.f;
        for (int i = 0; i < 147; ++i) amax = std::max(amax, std::fabs(w[co * 147 + i]));
        if (amax > 0.f && std::isfinite(amax)) { int e; std::frexp(amax, &e); qexp[co] = 15 - e; }
    }
    for (int s = 0; s < NSTEP; ++s)
        for (int j = 0; j < 2; ++j)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int ch = 2 * s + (lane >> 5), ky = ch / 3, e = 8 * (ch % 3) + i, kx = e / 3, c = e % 3, co = 32 * j + (lane & 31);
                    const float r = (ch < 21 && e < 21 && co < Cout) ? std::ldexp(w[((co * 3 + c) * 7 + ky) * 7 + kx], qexp[co]) : 0.f;
                    const _Float16 hi = (_Float16)r, lo = (_Float16)(r - (float)hi);
                    memcpy(&out[((((size_t)s * 2 + j) * 2 + 0) * 64 + lane) * 8 + i], &hi, 2);
                    memcpy(&out[((((size_t)s * 2 + j) * 2 + 1) * 64 + lane) * 8 + i], &lo, 2);
                }
}

hipError_t launch_conv_stem_b3(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    if (!conv_stem_b3_eligible(p) || !p.wstemb) return hipErrorInvalidValue;
    p.w = static_cast<const float*>(p.wstemb);
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
    const int grid = ntiles < cus ? ntiles : cus;      // 120-135 KB of LDS: one persistent block per CU
    if (p.f16 == 3) {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_stem_b3_kernel<true>), stem_lds(true)); e != hipSuccess) return e;
        hipLaunchKernelGGL((conv_stem_b3_kernel<true>), dim3(grid), dim3(512), stem_lds(true), st, p, tiles_x, tiles_y, ntiles);
        return hipGetLastError();
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_stem_b3_kernel<false>), stem_lds(false)); e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_stem_b3_kernel<false>), dim3(grid), dim3(512), stem_lds(false), st, p, tiles_x, tiles_y, ntiles);
    return hipGetLastError();
}
