// Winograd F(2x2,3x3) on the bf16 matrix cores, small-block form (launch geometry 43): the arithmetic and the patch path of
// conv_wino_b3.hip geometry 42 (exact three-term operands, the union of the block's patches loaded once into an LDS copy) with
// HALF the block -- 32 tiles (4x8 or 2x16) x 64 channels, 4 wavefronts, wavefront w owns positions 4w .. 4w+3 -- so that TWO
// blocks are resident per CU (80 KB of LDS and 256 registers per wavefront each).  The two wavefronts of a SIMD then belong to
// different blocks and run out of phase: the prologue (first patches, first transform), the barriers and the exchange epilogue
// of one block run under the other's matrix work.  That is what the layers with few input channels need (64 channels = 4 K
// steps between a prologue and an epilogue that geometry 42, one block per CU, cannot hide); the price is twice the weight-
// fragment traffic per MFMA (a fragment now multiplies 32 tiles instead of 64), which is why the tuner keeps 42 for the deep layers.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "conv_common.h"
#include "range.h"

typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
// chunk swizzle of the raw patch copy: chunk c of pixel column x at slot c ^ bitrev2((x >> 2) & 3).  The patch-column reads of a
// ds_read_b128 lane group (4 consecutive tiles, alternating channel quads: pixel columns x .. x+9) then hit 16 different slots.
#define RAW_SWZ(x) (((((x) >> 2) & 1) << 1) | (((x) >> 3) & 1))

namespace {
constexpr int TTS = 32;                // output tiles (2x2 pixels each) per block
constexpr int KKS = 64;                // output channels per block
constexpr int BKS = 16;                // input channels per K step
constexpr int VPSS = TTS * BKS + 8;    // floats per position of the V image (+32 B: the four patch columns of a quad hit different banks)
constexpr int VSTS = 16 * VPSS;
constexpr int RAWPX = 6 * 34;          // pixels of the largest patch union (2x16 tiles); 4x8 tiles: 10 x 18
constexpr size_t WBS_LDS = (size_t)2 * VSTS * sizeof(float) + RAWPX * 64 + 1024;      // 80640 B: two blocks per CU

__device__ __forceinline__ float quad_2211s(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xF, 0xF, true));
}
__device__ __forceinline__ void split3_pair_s(float v0, float v1, int& q0, int& q1, int& q2)
{
    const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
    q0 = (int)__builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = v0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u), r1 = v1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    const unsigned s0 = __builtin_bit_cast(unsigned, r0), s1 = __builtin_bit_cast(unsigned, r1);
    q1 = (int)__builtin_amdgcn_perm(s1, s0, 0x07060302);
    const float t0 = r0 - __builtin_bit_cast(float, s0 & 0xFFFF0000u), t1 = r1 - __builtin_bit_cast(float, s1 & 0xFFFF0000u);
    q2 = (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0), 0x07060302);
}
}  // namespace

// H2: the fp16x2 form (see conv_wino_b3.hip): two half terms per operand, three v_mfma_f32_32x32x16_f16 products per multiply-add
template <bool H2>
__global__ __launch_bounds__(256, 2) void conv_wino_b3s_kernel(ConvParams p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NPW = H2 ? 2 : 3;
    // fp16x2 form: the pixel scale from the range slot of the input tensor (range.h); the transformed patch B^T d B is up to 4x the
    // largest pixel, so V is split at a quarter of it (the factor 4 is in scale_h2w, host)
    RangeScale rs; rs.s = 1.f; rs.inv = 1.f;
    if constexpr (H2) rs = range_prologue(p.xr);
    const float xs = H2 ? rs.s * 0.25f : 1.f, xinv = rs.inv;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nblk = p.MT * p.NT, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int nt = swz % p.NT, mt = swz / p.NT;
    const int n0 = nt * KKS;

    const int TH = p.Ho >> 1, TW = p.Wo >> 1;
    const int bhs = p.wino_bhs, bws = 5 - bhs, BWm = (1 << bws) - 1;      // block = (1 << bhs) x (32 >> bhs) tiles
    const int RW = (2 << bws) + 2, RH = (2 << bhs) + 2;
    int un, uty0, utx0;
    {
        const int BX = (TW + BWm) >> bws, BY = (TH + (1 << bhs) - 1) >> bhs;
        un = mt / (BX * BY);
        const int rem = mt - un * (BX * BY);
        const int by = rem / BX;
        uty0 = by << bhs; utx0 = (rem - by * BX) << bws;
    }
    const int nk_all = p.Cin / BKS;
    const int kb = p.ksplit > 1 ? (int)blockIdx.y * p.kt_per_split : 0;
    const int nk = p.ksplit > 1 ? min(p.kt_per_split, nk_all - kb) : nk_all;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ur = make_rsrc(p.wub, p.wub_bytes);

    // ---- patch path (as geometry 42): loader slots -> raw copy -> patch columns -> B^T d B -> V ----
    const int j = tid & 3, q = (tid >> 2) & 1, tl = tid >> 3;      // transform item: tile tl (0..31), patch column j, channel quads q and q + 2
    float* rawS = smem + 2 * VSTS;
    unsigned g_off[4];
    int g_dst[4];
    f32x4 g[4];
#ifdef WKO_NO_LOADG
    for (int i = 0; i < 4; ++i) g[i] = f32x4{1.f, 2.f, 3.f, 4.f};
#endif
    {
        const float inv_rw = 1.0f / (float)RW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sl = tid + 256 * i, px = sl >> 2, c = sl & 3;
            int py, pxx;
            divmod_small(px, RW, inv_rw, py, pxx);
            const int iy = 2 * uty0 - 1 + py, ix = 2 * utx0 - 1 + pxx;
            const bool ok = px < RH * RW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            g_off[i] = ok ? (unsigned)((((un * p.H + iy) * p.W + ix) * p.xCs + 4 * c) * 4) : OOB;
            g_dst[i] = px < RH * RW ? px * 16 + ((c ^ RAW_SWZ(pxx)) << 2) : RH * RW * 16 + lane * 4;
        }
    }
    const int tyl = tl >> bws, txl = tl & BWm, pxx_t = 2 * txl + j;
    const int rr_off = ((2 * tyl) * RW + pxx_t) * 16 + ((q ^ RAW_SWZ(pxx_t)) << 2);
    const int lsw = (tl >> 2) & 3;      // chunk swizzle of the V image: see the fragment reads
    const int v_dst0 = j * VPSS + tl * BKS + ((q ^ lsw) << 2);
    const float sb = j == 1 ? 1.f : -1.f;
    auto load_g = [&](int k) {
#ifdef WKO_NO_LOADG
        return;
#endif
        const unsigned ko = (unsigned)(kb + min(k, nk - 1)) * (BKS * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = buf_load4(xr, g_off[i] != OOB ? g_off[i] + ko : OOB);
    };
    auto store_g = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(rawS + g_dst[i]) = g[i];
    };
    auto transform = [&](int stage, int it) {
#ifdef WKO_NO_XFORM
        return;
#endif
        float* vs = smem + stage * VSTS + (it ? v_dst0 + (((q ^ lsw) & 2) ? -8 : 8) : v_dst0);
        f32x4 d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = *reinterpret_cast<const f32x4*>(rawS + ((rr_off + r * RW * 16) ^ (it ? 8 : 0)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 vo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float t = i == 0 ? d[0][c] - d[2][c] : i == 1 ? d[1][c] + d[2][c] : i == 2 ? d[2][c] - d[1][c] : d[1][c] - d[3][c];
#ifdef WKO_XFORM_COPY
                vo[c] = d[i][c];
#else
                vo[c] = fmaf(sb, quad_2211s(t), t);
#endif
            }
            *reinterpret_cast<f32x4*>(vs + i * 4 * VPSS) = vo;
        }
    };

    // ---- fragments ----
    const int fr = lane & 31, fh = lane >> 5;
    // ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with the chunk of tile t at slot
    // c ^ ((t >> 2) & 3) the 16 tiles of a group hit 16 different 16-byte slots of the 256-byte bank row
    const int fsw = (fr >> 2) & 3;
    const int a_rd0 = fr * BKS + (((2 * fh) ^ fsw) << 2);
    const unsigned b_voff = (unsigned)((n0 + fr) * 32 + fh * 16);
    const unsigned u_pos = (unsigned)p.wino_rows * 32u;
    const unsigned u_step = 16u * u_pos;
    const unsigned u_plane = (unsigned)nk_all * u_step;
    i32x4 fb[4][NPW];       // weight fragments of four consecutive phases, each requested four phases (half a K step) ahead
#ifdef WKO_NO_LOADB
    for (int i = 0; i < 4; ++i) for (int pl = 0; pl < NPW; ++pl) fb[i][pl] = i32x4{0x3C003C00, 0x3C003C00, 0x3C003C00, 0x3C003C00};
#endif
    auto load_b = [&](int buf, int k, int pos, int jj) {
#ifdef WKO_NO_LOADB
        return;
#endif
        const unsigned so = (unsigned)(kb + min(k, nk - 1)) * u_step + (unsigned)pos * u_pos + (unsigned)jj * 1024u;
#pragma unroll
        for (int pl = 0; pl < NPW; ++pl) fb[buf][pl] = __builtin_amdgcn_raw_buffer_load_b128(ur, b_voff, so + (unsigned)pl * u_plane, 0);
    };
    f32x4 raw[2];
#ifdef WKO_NO_READRAW
    raw[0] = raw[1] = f32x4{1.f, 2.f, 3.f, 4.f};
#endif
    auto read_raw = [&](int stage, int pos) {
#ifdef WKO_NO_READRAW
        return;
#endif
        const float* v = smem + stage * VSTS + pos * VPSS;
        raw[0] = *reinterpret_cast<const f32x4*>(v + a_rd0);
        raw[1] = *reinterpret_cast<const f32x4*>(v + (a_rd0 ^ 4));
    };
    auto split_raw = [&](i32x4 (&a)[NPW]) {
#ifdef WKO_NO_SPLIT
        a[0] = __builtin_bit_cast(i32x4, raw[0]); a[NPW - 1] = __builtin_bit_cast(i32x4, raw[1]);
        return;
#endif
        if constexpr (H2) {
            f16x8s h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = raw[e >> 2][e & 3];
                h[e] = (_Float16)(v * xs);
                l[e] = (_Float16)__builtin_fmaf(v, xs, -(float)h[e]);      // exact residual, then rounded to half
            }
            a[0] = __builtin_bit_cast(i32x4, h);
            a[NPW - 1] = __builtin_bit_cast(i32x4, l);
            return;
        }
        int q0[4], q1[4], q2[4];
        split3_pair_s(raw[0][0], raw[0][1], q0[0], q1[0], q2[0]);
        split3_pair_s(raw[0][2], raw[0][3], q0[1], q1[1], q2[1]);
        split3_pair_s(raw[1][0], raw[1][1], q0[2], q1[2], q2[2]);
        split3_pair_s(raw[1][2], raw[1][3], q0[3], q1[3], q2[3]);
        a[0] = i32x4{q0[0], q0[1], q0[2], q0[3]};
        a[1] = i32x4{q1[0], q1[1], q1[2], q1[3]};
        a[NPW - 1] = i32x4{q2[0], q2[1], q2[2], q2[3]};
    };
    f32x16 acc[4][2];      // [own position][channel group]
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pi][jj][e] = 0.f;
    auto mma = [&](int pi, int jj, int buf, const i32x4 (&a)[NPW]) {
#ifdef WKO_NO_MMA
        asm volatile("" :: "v"(fb[buf][0]), "v"(fb[buf][NPW - 1]), "v"(a[0]), "v"(a[NPW - 1]));
        return;
#endif
        if constexpr (H2) {      // the two cross terms, then hi * hi
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
                acc[pi][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, fb[buf][PB[t]]),
                                                                     __builtin_bit_cast(f16x8s, a[PA[t]]), acc[pi][jj], 0, 0, 0);
        } else {
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc[pi][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8s, fb[buf][PB[t] % NPW]),
                                                                      __builtin_bit_cast(bf16x8s, a[PA[t] % NPW]), acc[pi][jj], 0, 0, 0);
        }
    };
#ifdef WKO_NO_BARRIER
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
#else
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#endif

    const int P0 = 4 * wave;
    i32x4 aA[NPW], aB[NPW];

    // ---- prologue ----
    load_g(0);
    load_b(0, 0, P0, 0); load_b(1, 0, P0, 1); load_b(2, 0, P0 + 1, 0); load_b(3, 0, P0 + 1, 1);
    store_g();
    load_g(1);
    lds_barrier();                       // raw copy = patches of step 0
    transform(0, 0);
    transform(0, 1);
    lds_barrier();                       // V stage 0 complete, raw copy read by everybody
    store_g();
    load_g(2);
    lds_barrier();                       // raw copy = patches of step 1
    read_raw(0, P0);
    split_raw(aA);

#define WS_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WS_INTERLEAVE(nv)                                                                             \
    do {                                                                                              \
        _Pragma("unroll") for (int g_ = 0; g_ < (H2 ? 3 : 6); ++g_) {                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x002, (H2 ? 2 : 1) * (nv), 0);                      \
        }                                                                                             \
    } while (0)
    // One K step = eight phases of 6 MFMAs (position x 32-channel group).  raw copy = patches of step k+1, g = patches of step k+2
    // (in flight); barriers: V stage complete (end of phase 3), raw copy complete (end of phase 7).
    for (int k = 0; k < nk; ++k) {
        const int cur = k & 1;
        // phase 0: P0 channels 0-31 | patch item 0 of step k+1
        read_raw(cur, P0 + 1);
        mma(0, 0, 0, aA);
        transform(cur ^ 1, 0);
        WS_INTERLEAVE(8);
        WS_FENCE();
        load_b(0, k, P0 + 2, 0);
        WS_FENCE();
        // phase 1: P0 channels 32-63 | split of P1
        mma(0, 1, 1, aA);
        split_raw(aB);
        WS_INTERLEAVE(8);
        WS_FENCE();
        load_b(1, k, P0 + 2, 1);
        WS_FENCE();
        // phase 2: P1 channels 0-31 | patch item 1
        read_raw(cur, P0 + 2);
        mma(1, 0, 2, aB);
        transform(cur ^ 1, 1);
        WS_INTERLEAVE(8);
        WS_FENCE();
        load_b(2, k, P0 + 3, 0);
        WS_FENCE();
        // phase 3: P1 channels 32-63 | split of P2
        mma(1, 1, 3, aB);
        split_raw(aA);
        WS_INTERLEAVE(8);
        WS_FENCE();
        load_b(3, k, P0 + 3, 1);
        lds_barrier();                  // V stage cur^1 complete; the raw copy has been read by everybody
        WS_FENCE();
        // phase 4: P2 channels 0-31
        read_raw(cur, P0 + 3);
        mma(2, 0, 0, aA);
        WS_FENCE();
        load_b(0, k + 1, P0, 0);
        WS_FENCE();
        // phase 5: P2 channels 32-63 | split of P3
        mma(2, 1, 1, aA);
        split_raw(aB);
        WS_INTERLEAVE(8);
        WS_FENCE();
        load_b(1, k + 1, P0, 1);
        WS_FENCE();
        // phase 6: P3 channels 0-31 | P0's fragment of step k+1; the patches of step k+2 go to the raw copy
        read_raw(cur ^ 1, P0);
        mma(3, 0, 2, aB);
        store_g();
        WS_FENCE();
        load_b(2, k + 1, P0 + 1, 0);
        load_g(k + 3);
        WS_FENCE();
        // phase 7: P3 channels 32-63 | split of P0 (step k+1)
        mma(3, 1, 3, aB);
        split_raw(aA);
        WS_INTERLEAVE(8);
        WS_FENCE();
        load_b(3, k + 1, P0 + 1, 1);
        lds_barrier();                  // raw copy = patches of step k+2; V stage cur is free
        WS_FENCE();
    }
#undef WS_FENCE
#undef WS_INTERLEAVE

#ifdef WKO_NO_EPI
    {
        f32x16 t = acc[0][0];
        for (int pi = 0; pi < 4; ++pi) for (int jj = 0; jj < 2; ++jj) t += acc[pi][jj];
        if (p.slope == 12345.f) for (int e = 0; e < 16; ++e) p.y[threadIdx.x * 16 + e] = t[e];
        return;
    }
#endif
    // ---- exchange + output transform + epilogue ----
    const int et = tid >> 3, ecq = tid & 7;                 // this thread finishes tile et, channels 4*ecq .. +3 of each round
    const int en = un, ety = uty0 + (et >> bws), etx = utx0 + (et & BWm);
    const bool tile_ok = ety < TH && etx < TW;
    const unsigned pix00 = (unsigned)((en * p.Ho + 2 * ety) * p.Wo + 2 * etx);
    const unsigned pix[4] = {pix00, pix00 + 1, pix00 + (unsigned)p.Wo, pix00 + (unsigned)p.Wo + 1};
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res : p.y, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t y2r = make_rsrc(p.y2 ? p.y2 : p.y, p.y2 ? p.y2_bytes : 0u);
    const size_t slab = (size_t)blockIdx.y * p.M * p.Cout_store;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.ksplit > 1 ? p.ws + slab : p.y, p.ksplit > 1 ? (unsigned)((size_t)p.M * p.Cout_store * 4) : 0u);
    float* X = smem;                                        // [16][32][32], chunk c of a tile row at slot c ^ (tile & 7)
    const int x_rd = et * 32 + ((ecq ^ (et & 7)) << 2);
    lds_barrier();                                          // every wavefront has read its last fragments
    unsigned rmax = 0u, rmax2 = 0u;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int co = n0 + jj * 32 + 4 * ecq;
        const bool ok = co < p.Cout_store && tile_ok;
        f32x4 rv[4];
        if (p.res && p.ksplit <= 1) {
#pragma unroll
            for (int o = 0; o < 4; ++o) rv[o] = buf_load4(rr, ok ? (pix[o] * p.resCs + co) * 4u : OOB);
        }
#pragma unroll
        for (int pi = 0; pi < 4; ++pi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int tile = fr, c = 2 * g + fh;
                f32x4 v = {acc[pi][jj][4 * g], acc[pi][jj][4 * g + 1], acc[pi][jj][4 * g + 2], acc[pi][jj][4 * g + 3]};
                *reinterpret_cast<f32x4*>(X + ((4 * wave + pi) * 32 + tile) * 32 + ((c ^ (tile & 7)) << 2)) = v;
            }
        lds_barrier();
        f32x4 m[16];
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) m[pp] = *reinterpret_cast<const f32x4*>(X + pp * 32 * 32 + x_rd);
        f32x4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s0[4], s1[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                s0[c4] = m[c4][e] + m[4 + c4][e] + m[8 + c4][e];
                s1[c4] = m[4 + c4][e] - m[8 + c4][e] - m[12 + c4][e];
            }
            v[0][e] = s0[0] + s0[1] + s0[2];
            v[1][e] = s0[1] - s0[2] - s0[3];
            v[2][e] = s1[0] + s1[1] + s1[2];
            v[3][e] = s1[1] - s1[2] - s1[3];
        }
        if (p.ksplit > 1) {
#pragma unroll
            for (int o = 0; o < 4; ++o) buf_store4(wr, ok ? (pix[o] * p.Cout_store + co) * 4u : OOB, v[o]);
        } else {
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
                if (p.yr && ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(v[o][e]); rmax = b > rmax ? b : rmax; }
                }
            }
            if (p.y2) {
                const f32x4 sc2 = *reinterpret_cast<const f32x4*>(p.scale2 + cc), sf2 = *reinterpret_cast<const f32x4*>(p.shift2 + cc);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    f32x4 u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[e] = fmaxf(v[o][e] * sc2[e] + sf2[e], 0.f);
                    buf_store4(y2r, ok ? (pix[o] * p.y2Cs + co) * 4u : OOB, u);
                    if (p.y2r && ok) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(u[e]); rmax2 = b > rmax2 ? b : rmax2; }
                    }
                }
            }
        }
        if (jj == 0) lds_barrier();                         // round 1 overwrites the image
    }
    // range slots of the outputs (range.h): the largest |value| this wavefront stored
    if (p.ksplit <= 1) {
        const unsigned key = (unsigned)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        if (p.yr) range_note_wave(p.yr, rmax, key);
        if (p.y2 && p.y2r) range_note_wave(p.y2r, rmax2, key);
    }
#endif
}

// geometry 43: tile-block shape for an output of TH x TW tiles, and the number of blocks
long conv_wino_b3s_blocks(const ConvParams& p, int* bhs)
{
    const int TH = p.Ho / 2, TW = p.Wo / 2;
    const int s = TH >= 4 ? 2 : 1;
    if (bhs) *bhs = s;
    const int BH = 1 << s, BW = 32 >> s;
    const long n = p.M / ((long)p.Ho * p.Wo);
    return n * ((TH + BH - 1) / BH) * ((TW + BW - 1) / BW);
}

hipError_t launch_conv_wino_b3s(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    if (!conv_wino_b3_eligible(p) || !p.wub) return hipErrorInvalidValue;
    p.wino_T = p.M / 4;
    p.MT = (int)conv_wino_b3s_blocks(p, &p.wino_bhs);
    p.NT = p.wino_rows / KKS;
    if (p.f16 == 3) {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_b3s_kernel<true>), WBS_LDS); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv_wino_b3s_kernel<true>, dim3(p.MT * p.NT, p.ksplit > 1 ? p.ksplit : 1), dim3(256), WBS_LDS, st, p);
    } else {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_b3s_kernel<false>), WBS_LDS); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv_wino_b3s_kernel<false>, dim3(p.MT * p.NT, p.ksplit > 1 ? p.ksplit : 1), dim3(256), WBS_LDS, st, p);
    }
    if (p.ksplit > 1) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_splitk_reduce(p, 1, st);
    }
    return hipGetLastError();
}
