// Winograd F(2x2,3x3) in the fp16x2 form with the transformed weights RESIDENT in registers (launch geometry 44): the 64 -> 64 channel
// 3x3 layers at quarter resolution (`res2*_branch2b`, resnet_v1_101_flownet_deeplab.py:589-645; the two units of the ResNet-18/34
// branch's first stage, :88-130) -- 19 launches of a step that geometry 43 runs at 250-280 us, bound by neither roof (1.5-2 TB/s,
// 0.16-0.2 of the fp16 peak).
//
// What geometry 43 pays for on these layers: every wavefront fetches the weight fragments of its positions from L2 again in every K
// step of every tile block -- 64 KB per block and step against 13 KB of patches.  With 64 input and 64 output channels the whole of
// U (16 positions x 64 x 64, two half planes) is 256 KB: 32 KB per wavefront if EIGHT wavefronts own two positions each -- 128
// registers.  So: one persistent block of 512 threads per CU, U loaded once, tile blocks (32 tiles = 128 output pixels) streamed past
// it; per K step a wavefront reads and splits the V fragments of its two positions and issues 12 matrix instructions from registers.
// The patch path, the V image, the split and the order of the three products are those of conv_wino_b3s.hip (bit-identical results);
// the exchange epilogue holds all 64 channels at once (16 x 32 x 64 floats, 128 KB: one round instead of two).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "conv_common.h"

typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
#define RAW_SWZW(x) (((((x) >> 2) & 1) << 1) | (((x) >> 3) & 1))      // as conv_wino_b3s.hip RAW_SWZ

namespace {
constexpr int TTW = 32;                // output tiles (2x2 pixels each) per tile block
constexpr int BKW = 16;                // input channels per K step
constexpr int NKW = 4;                 // K steps: 64 input channels
constexpr int VPSW = TTW * BKW + 8;    // floats per position of the V image
constexpr int VSTW = 16 * VPSW;
constexpr int RAWPXW = 6 * 34;
constexpr size_t WHW_LDS = 16 * 32 * 64 * sizeof(float);      // the exchange image; V stages (66 KB) + raw copy (13 KB) live inside it
static_assert((size_t)2 * VSTW * sizeof(float) + RAWPXW * 64 + 1024 <= WHW_LDS, "V stages + raw copy fit into the exchange image");

__device__ __forceinline__ float quad_2211w(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xF, 0xF, true));
}
}  // namespace

__global__ __launch_bounds__(512, 1) void conv_wino_h2w_kernel(ConvParams p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float xs = p.xs ? p.xs[0] * 0.25f : 1.f, xinv = p.xs ? p.xs[1] : 1.f;      // the factor 4 is in scale_h2w (host)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int TH = p.Ho >> 1, TW = p.Wo >> 1;
    const int bhs = p.wino_bhs, bws = 5 - bhs, BWm = (1 << bws) - 1;      // tile block = (1 << bhs) x (32 >> bhs) tiles
    const int RW = (2 << bws) + 2, RH = (2 << bhs) + 2;
    const int BX = (TW + BWm) >> bws, BY = (TH + (1 << bhs) - 1) >> bhs;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ur = make_rsrc(p.wub, p.wub_bytes);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res : p.y, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t y2r = make_rsrc(p.y2 ? p.y2 : p.y, p.y2 ? p.y2_bytes : 0u);

    // ---- U -> registers, once: [K step][own position][32-channel group][plane] -----------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    const int P0 = 2 * wave;
    i32x4 wreg[NKW][2][2][2];
    {
        const unsigned b_voff = (unsigned)(fr * 32 + fh * 16);
        const unsigned u_pos = (unsigned)p.wino_rows * 32u;
        const unsigned u_step = 16u * u_pos;
        const unsigned u_plane = (unsigned)NKW * u_step;
#pragma unroll
        for (int k = 0; k < NKW; ++k)
#pragma unroll
            for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        wreg[k][pi][jj][pl] = __builtin_amdgcn_raw_buffer_load_b128(
                            ur, b_voff, (unsigned)k * u_step + (unsigned)(P0 + pi) * u_pos + (unsigned)jj * 1024u + (unsigned)pl * u_plane, 0);
    }

    // ---- per-thread constants of the patch path (tile-block independent part) ---------------------------------------------------
    const int j = tid & 3, q = (tid >> 2) & 1, tl = (tid >> 3) & 31, it = tid >> 8;      // transform item: tile tl, patch column j, channel quads q / q + 2 (it)
    float* rawS = smem + 2 * VSTW;
    const int tyl = tl >> bws, txl = tl & BWm, pxx_t = 2 * txl + j;
    const int rr_off = (((2 * tyl) * RW + pxx_t) * 16 + ((q ^ RAW_SWZW(pxx_t)) << 2)) ^ (it ? 8 : 0);
    const int lsw = (tl >> 2) & 3;
    const int v_dst = j * VPSW + tl * BKW + ((q ^ lsw) << 2) + (it ? ((((q ^ lsw) & 2) ? -8 : 8)) : 0);
    const float sb = j == 1 ? 1.f : -1.f;
    const float inv_rw = 1.0f / (float)RW;
    int g_py[2], g_px[2], g_dst[2];
    bool g_in[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sl = tid + 512 * i, px = sl >> 2, c = sl & 3;
        int py, pxx;
        divmod_small(px, RW, inv_rw, py, pxx);
        g_py[i] = py; g_px[i] = pxx;
        g_in[i] = px < RH * RW;
        g_dst[i] = g_in[i] ? px * 16 + ((c ^ RAW_SWZW(pxx)) << 2) : RH * RW * 16 + lane * 4;
    }
    const int fsw = (fr >> 2) & 3;
    const int a_rd0 = fr * BKW + (((2 * fh) ^ fsw) << 2);
    const int et = tid >> 4, ecq = tid & 15;                // exchange: this thread finishes tile et, channels 4 ecq .. + 3
    const int x_rd = et * 64 + ((ecq ^ (et & 15)) << 2);
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    const int nblk = p.MT;
    for (int mt = blockIdx.x; mt < nblk; mt += gridDim.x) {
        const int un = mt / (BX * BY);
        const int rem = mt - un * (BX * BY);
        const int by = rem / BX;
        const int uty0 = by << bhs, utx0 = (rem - by * BX) << bws;

        unsigned g_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy = 2 * uty0 - 1 + g_py[i], ix = 2 * utx0 - 1 + g_px[i];
            const bool ok = g_in[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            g_off[i] = ok ? (unsigned)((((un * p.H + iy) * p.W + ix) * p.xCs + 4 * ((tid + 512 * i) & 3)) * 4) : OOB;
        }
        f32x4 g[2];
        auto load_g = [&](int k) {
            const unsigned ko = (unsigned)(k < NKW ? k : NKW - 1) * (BKW * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) g[i] = buf_load4(xr, g_off[i] != OOB ? g_off[i] + ko : OOB);
        };
        auto store_g = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(rawS + g_dst[i]) = g[i];
        };
        auto transform = [&](int stage) {
            float* vs = smem + stage * VSTW + v_dst;
            f32x4 d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = *reinterpret_cast<const f32x4*>(rawS + rr_off + r * RW * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 vo;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float t = i == 0 ? d[0][c] - d[2][c] : i == 1 ? d[1][c] + d[2][c] : i == 2 ? d[2][c] - d[1][c] : d[1][c] - d[3][c];
                    vo[c] = fmaf(sb, quad_2211w(t), t);
                }
                *reinterpret_cast<f32x4*>(vs + i * 4 * VPSW) = vo;
            }
        };
        f32x4 raw[2];
        auto read_raw = [&](int stage, int pos) {
            const float* v = smem + stage * VSTW + pos * VPSW;
            raw[0] = *reinterpret_cast<const f32x4*>(v + a_rd0);
            raw[1] = *reinterpret_cast<const f32x4*>(v + (a_rd0 ^ 4));
        };
        auto split_raw = [&](i32x4 (&a)[2]) {
            f16x8w h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = raw[e >> 2][e & 3];
                h[e] = (_Float16)(v * xs);
                l[e] = (_Float16)__builtin_fmaf(v, xs, -(float)h[e]);      // exact residual, then rounded to half
            }
            a[0] = __builtin_bit_cast(i32x4, h);
            a[1] = __builtin_bit_cast(i32x4, l);
        };
        f32x16 acc[2][2];      // [own position][channel group]
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[pi][jj][e] = 0.f;
        i32x4 aA[2], aB[2];

        // ---- prologue (the previous tile block's exchange image has been read: barrier at the end of the loop body) -------------
        load_g(0);
        store_g();
        load_g(1);
        lds_barrier();                       // raw copy = patches of step 0
        transform(0);
        lds_barrier();                       // V stage 0 complete, raw copy read by everybody
        store_g();
        load_g(2);
        lds_barrier();                       // raw copy = patches of step 1
        read_raw(0, P0);
        split_raw(aA);

#pragma unroll
        for (int k = 0; k < NKW; ++k) {
            const int cur = k & 1;
            auto mma = [&](int pi, int jj, const i32x4 (&a)[2]) {      // the two cross terms, then hi * hi (the order of conv_wino_b3s.hip)
                acc[pi][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, wreg[k][pi][jj][0]), __builtin_bit_cast(f16x8w, a[1]), acc[pi][jj], 0, 0, 0);
                acc[pi][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, wreg[k][pi][jj][1]), __builtin_bit_cast(f16x8w, a[0]), acc[pi][jj], 0, 0, 0);
                acc[pi][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, wreg[k][pi][jj][0]), __builtin_bit_cast(f16x8w, a[0]), acc[pi][jj], 0, 0, 0);
            };
            // phase 0: P0 channels 0-31 | the patch item of step k+1
            read_raw(cur, P0 + 1);
            mma(0, 0, aA);
            if (k + 1 < NKW) transform(cur ^ 1);
            // phase 1: P0 channels 32-63 | split of P1
            mma(0, 1, aA);
            split_raw(aB);
            lds_barrier();                  // V stage cur^1 complete; the raw copy has been read by everybody
            // phase 2: P1 channels 0-31 | P0's fragment of step k+1; the patches of step k+2 go to the raw copy
            if (k + 1 < NKW) read_raw(cur ^ 1, P0);
            mma(1, 0, aB);
            if (k + 2 < NKW) { store_g(); load_g(k + 3); }
            // phase 3: P1 channels 32-63 | split of P0 (step k+1)
            mma(1, 1, aB);
            if (k + 1 < NKW) split_raw(aA);
            lds_barrier();                  // raw copy = patches of step k+2; V stage cur is free (after the last step: the whole image is)
        }

        // ---- exchange + output transform + epilogue ---------------------------------------------------------------------------------
        const int ety = uty0 + (et >> bws), etx = utx0 + (et & BWm);
        const bool tile_ok = ety < TH && etx < TW;
        const unsigned pix00 = (unsigned)((un * p.Ho + 2 * ety) * p.Wo + 2 * etx);
        const unsigned pix[4] = {pix00, pix00 + 1, pix00 + (unsigned)p.Wo, pix00 + (unsigned)p.Wo + 1};
        const int co = 4 * ecq;
        const bool ok = co < p.Cout_store && tile_ok;
        f32x4 rv[4];
        if (p.res) {
#pragma unroll
            for (int o = 0; o < 4; ++o) rv[o] = buf_load4(rr, ok ? (pix[o] * p.resCs + co) * 4u : OOB);
        }
        float* X = smem;                                        // [16][32][64], chunk c of a tile row at slot c ^ (tile & 15)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int c = 8 * jj + 2 * gq + fh;
                    f32x4 v = {acc[pi][jj][4 * gq], acc[pi][jj][4 * gq + 1], acc[pi][jj][4 * gq + 2], acc[pi][jj][4 * gq + 3]};
                    *reinterpret_cast<f32x4*>(X + ((P0 + pi) * 32 + fr) * 64 + ((c ^ (fr & 15)) << 2)) = v;
                }
        lds_barrier();
        f32x4 v[4];
        {
            f32x4 s0[4], s1[4];      // column sums of the output transform, position row by position row (16 fragment reads, 8 live at a time)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(X + (c4) * 32 * 64 + x_rd);
                const f32x4 m1 = *reinterpret_cast<const f32x4*>(X + (4 + c4) * 32 * 64 + x_rd);
                const f32x4 m2 = *reinterpret_cast<const f32x4*>(X + (8 + c4) * 32 * 64 + x_rd);
                const f32x4 m3 = *reinterpret_cast<const f32x4*>(X + (12 + c4) * 32 * 64 + x_rd);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0[c4][e] = m0[e] + m1[e] + m2[e];
                    s1[c4][e] = m1[e] - m2[e] - m3[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[0][e] = s0[0][e] + s0[1][e] + s0[2][e];
                v[1][e] = s0[1][e] - s0[2][e] - s0[3][e];
                v[2][e] = s1[0][e] + s1[1][e] + s1[2][e];
                v[3][e] = s1[1][e] - s1[2][e] - s1[3][e];
            }
        }
        {
            const int cc = co < p.Cout_store ? co : 0;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + cc) * xinv, sf = *reinterpret_cast<const f32x4*>(p.shift + cc);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                v[o] = v[o] * sc + sf;
                if (p.res) v[o] += rv[o];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.act == 1) v[o][e] = fmaxf(v[o][e], 0.f);
                    else if (p.act == 2) v[o][e] = v[o][e] > 0.f ? v[o][e] : v[o][e] * p.slope;
                }
                buf_store4(yr, ok ? (pix[o] * p.yCs + co) * 4u : OOB, v[o]);
            }
            if (p.y2) {
                const f32x4 sc2 = *reinterpret_cast<const f32x4*>(p.scale2 + cc), sf2 = *reinterpret_cast<const f32x4*>(p.shift2 + cc);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    f32x4 u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[e] = fmaxf(v[o][e] * sc2[e] + sf2[e], 0.f);
                    buf_store4(y2r, ok ? (pix[o] * p.y2Cs + co) * 4u : OOB, u);
                }
            }
        }
        lds_barrier();                                          // the exchange image has been read: the next tile block may stage
    }
#endif
}

// layers geometry 44 takes: the fp16x2 form of a Winograd-eligible layer with 64 input channels and one 64-channel block of outputs
bool conv_wino_h2w_eligible(const ConvParams& p)
{
    return conv_wino_b3_eligible(p) && p.Cin == 64 && p.wino_rows == 64;
}

// p.wub = the two half planes of U (ConvParams::wubh), p.scale = scale_h2w, p.xs = the layer's range slot, p.f16 == 3 (set by the dispatcher)
hipError_t launch_conv_wino_h2w(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    if (!conv_wino_h2w_eligible(p) || !p.wub || p.f16 != 3 || p.ksplit > 1) return hipErrorInvalidValue;
    p.wino_T = p.M / 4;
    p.MT = (int)conv_wino_b3s_blocks(p, &p.wino_bhs);
    p.NT = 1;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_h2w_kernel), WHW_LDS); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_wino_h2w_kernel, dim3(p.MT < cus ? p.MT : cus), dim3(512), WHW_LDS, st, p);
    return hipGetLastError();
}
