// Implicit-GEMM convolution on the gfx950 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, fmaf-chain numerics, 157 TF/s peak).
//
// GEMM view:  D[m, co] = sum_k A[m, k] * Wt[co, k]
//   m  = output pixel (n, oy, ox), NHWC activations
//   k  = (ky, kx, ci) flattened, ci fastest, ci padded to a multiple of 4
//   Wt = weights repacked on the host to [Cout_pad][K_pad] (K contiguous)
//
// Block = 256 threads = 4 wavefronts (one per SIMD).  A BMxBK pixel tile and a
// BNxBK weight tile are staged global -> VGPR -> LDS (rows padded to BK+4
// floats, which makes the ds_read_b128 fragment reads bank-conflict free:
// 144-byte row stride visits all 16 16-byte slots of the 256-byte bank row),
// double buffered, one barrier per K step.  Each lane reads its MFMA operands
// as one float4 per 32-row tile: lanes 0-31 hold k = 8t..8t+3, lanes 32-63
// hold k = 8t+4..8t+7, so four consecutive MFMAs (register r = 0..3) contract
// k = 8t+r and 8t+4+r.  A and Wt use the same permutation, so the sum is
// unchanged.
//
// Epilogue (fused): v = acc*scale[co] + shift[co] (+ residual) -> act ->
// store NHWC; optional second output relu(v*scale2 + shift2) for the
// pre-activation ResNet units.  C/D layout of the 32x32 MFMA puts 32
// consecutive `co` in the 32 lanes of a half-wave: 128-byte coalesced stores.
//
// mode deconv2x: a 4x4/stride-2/pad-1 transposed convolution is run as four
// 2x2 sub-pixel convolutions (blockIdx.y = output parity class) whose results
// interleave into the 2x-upsampled output.
#include <hip/hip_runtime.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(ConvParams p)
{
    constexpr int BK = 32, LDK = BK + 4;
    constexpr int MI = BM / (WGM * 32), NI = BN / (WGN * 32);
    constexpr int AR = BM / 32, BR = BN / 32;       // rows each thread stages
    static_assert(WGM * WGN == 4, "4 waves per block");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;     // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (and its L2);
    // give each XCD a contiguous run of tiles, N-tiles fastest, so the pixel
    // tile is re-read from that L2 by its neighbours.
    const int nblk = p.MT * p.NT;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int nt = swz % p.NT, mt = swz / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    int kh = p.kh, kw = p.kw, ph = p.ph, pw = p.pw;
    const float* wbase = p.w;
    int py = 0, px = 0;
    if (p.deconv2x) {
        py = blockIdx.y >> 1; px = blockIdx.y & 1;
        ph = 1 - py; pw = 1 - px;
        wbase += (size_t)blockIdx.y * p.w_class_stride;
    }
    const int ntaps = kh * kw;

    // ---- staging coordinates --------------------------------------------------
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    int a_iy0[AR], a_ix0[AR], a_nb[AR];
    bool a_ok[AR];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_iy0[i] = oy * p.sh - ph;
        a_ix0[i] = ox * p.sw - pw;
        a_nb[i] = n * p.H * p.W;
    }
    int ci = scol, tap = 0, ky = 0, kx = 0;
    while (ci >= p.Cin) { ci -= p.Cin; ++tap; if (++kx == kw) { kx = 0; ++ky; } }

    const float* wrow0 = wbase + (size_t)(n0 + srow) * p.K_pad + scol;
    const size_t wrow_step = (size_t)32 * p.K_pad;

    f32x4 ra[AR], rb[BR];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = a_iy0[i] + ky * p.dh, ix = a_ix0[i] + kx * p.dw;
            const bool ok = a_ok[i] && tap < ntaps && (unsigned)iy < (unsigned)p.H &&
                            (unsigned)ix < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(
                        p.x + ((size_t)(a_nb[i] + iy * p.W + ix) * p.xCs + ci));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            rb[i] = *reinterpret_cast<const f32x4*>(wrow0 + i * wrow_step + k0);
    };
    auto advance = [&]() {
        ci += BK;
        while (ci >= p.Cin) { ci -= p.Cin; ++tap; if (++kx == kw) { kx = 0; ++ky; } }
    };
    auto store_tiles = [&](int buf) {
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BN * LDK;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<f32x4*>(a + (srow + 32 * i) * LDK + scol) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<f32x4*>(b + (srow + 32 * i) * LDK + scol) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int KT = p.K_pad / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 4;
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) { advance(); load_tiles((kt + 1) * BK); }
        const float* a = As + cur * BM * LDK + (wm * MI * 32 + frow) * LDK + fk;
        const float* b = Bs + cur * BN * LDK + (wn * NI * 32 + frow) * LDK + fk;
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            f32x4 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK + t * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK + t * 8);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- fused epilogue ---------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int co = n0 + (wn * NI + j) * 32 + (lane & 31);
        const bool cok = co < p.Cout_store;
        float sc = 1.f, sf = 0.f, sc2 = 1.f, sf2 = 0.f;
        if (cok) {
            sc = p.scale[co]; sf = p.shift[co];
            if (p.y2) { sc2 = p.scale2[co]; sf2 = p.shift2[co]; }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * MI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (!cok || m >= p.M) continue;
                size_t pix = (size_t)m;
                if (p.deconv2x) {
                    const int n = m / HoWo, rem = m - n * HoWo;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    pix = ((size_t)n * p.yH + (2 * oy + py)) * p.yW + (2 * ox + px);
                }
                float v = acc[i][j][e] * sc + sf;
                if (p.res) v += p.res[pix * p.resCs + co];
                if (p.act == 1) v = fmaxf(v, 0.f);
                else if (p.act == 2) v = v > 0.f ? v : v * p.slope;
                p.y[pix * p.yCs + co] = v;
                if (p.y2) p.y2[pix * p.y2Cs + co] = fmaxf(v * sc2 + sf2, 0.f);
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
static hipError_t launch_cfg(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.Cout_store + BN - 1) / BN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 36 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv_igemm_f32_kernel<BM, BN, WGM, WGN>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    dim3 grid(p.MT * p.NT, p.deconv2x ? 4 : 1);
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WGM, WGN>), grid, dim3(256), lds, st, p);
    return hipGetLastError();
}

// Tile choice: the chip has 256 CUs; prefer the largest tile that still gives
// >= ~2 blocks per CU, narrow-N tiles for the 2/19/72-channel layers.
int conv_pick_tile(const ConvParams& p)
{
    if (p.force_tile >= 0) return p.force_tile;
    const long classes = p.deconv2x ? 4 : 1;
    const int cs = p.Cout_store;
    if (cs <= 32) return 4;                              // 128x32
    auto blocks = [&](int bm, int bn) {
        return classes * ((p.M + bm - 1) / bm) * (long)((cs + bn - 1) / bn);
    };
    if (cs <= 64) return blocks(128, 64) >= 512 ? 1 : 3; // 128x64 or 64x64
    if (cs % 128 == 0 || cs > 256) {
        if (blocks(128, 128) >= 512) return 0;
        if (blocks(128, 64) >= 512) return 1;
        if (blocks(64, 128) >= 384) return 2;
        return 3;
    }
    if (blocks(128, 64) >= 512) return 1;
    return 3;
}

hipError_t launch_conv_igemm(const ConvParams& p, hipStream_t st)
{
    switch (conv_pick_tile(p)) {
        case 0: return launch_cfg<128, 128, 2, 2>(p, st);
        case 1: return launch_cfg<128, 64, 2, 2>(p, st);
        case 2: return launch_cfg<64, 128, 2, 2>(p, st);
        case 3: return launch_cfg<64, 64, 2, 2>(p, st);
        default: return launch_cfg<128, 32, 4, 1>(p, st);
    }
}
