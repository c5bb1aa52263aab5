// Winograd F(2x2, 3x3) on the bf16 matrix cores with exact three-term operands (launch geometry 41): the 16 position-GEMMs
// of conv_wino.hip computed the way conv_b3r.hip computes a direct convolution -- every fp32 operand split EXACTLY into
// three bf16 terms, six v_mfma_f32_32x32x16_bf16 products per multiply-add, fp32 accumulate.  For the 3x3 / stride 1 / pad 1
// layers this executes 16/36 of the direct kernel's matrix work AND 4/9 of its loader work (an input value is fetched
// and split for the 4 patches it belongs to instead of for 9 taps).
//
// Numerics: the same function as conv_wino_f32_kernel -- B^T d B is formed in fp32 by the loader (adds only), U = G g G^T
// comes from the host in double, rounded to fp32 once; the products of those fp32 values are exact to 2^-24 relative
// (three-term split of both operands, the three smallest cross terms dropped), the sums are fp32.
//
// Work split.  A block (512 threads = 8 wavefronts, two per SIMD) owns 64 output tiles (2x2 pixels each) x 64 output
// channels.  Unlike the fp32 kernel the 16 Winograd positions are spread over the wavefronts: wavefront w owns
// positions 2w and 2w+1 for ALL 64 tiles x 64 channels (2 x 2 MFMA tiles of 32x32 per position, 128 accumulators), so
//   * a transformed input value is read from LDS by exactly one wavefront, once; it is kept in LDS as fp32 (64 KB per
//     stage, two stages) and split into its three bf16 terms by the wavefront that multiplies it, between its MFMAs;
//   * the transformed weights never touch LDS: the host stores the three bf16 planes in MFMA fragment order
//     ([plane][K step][position][row][16]: the 16 bytes a lane feeds to one MFMA are contiguous), each wavefront loads the
//     fragments of its own positions global -> VGPR, one (position, 32-channel group) ahead;
//   * the output transform needs the 16 positions of a (tile, channel): after the K loop the accumulators are exchanged
//     through the (then free) LDS in two rounds of 32 channels, and every thread finishes one tile x 4 channels:
//     A^T M A, scale / shift, residual, activation, dual output, 16-byte stores.
//
// K loop: 16 input channels per step; one barrier per step.  Per step a wavefront runs four phases of 12 MFMAs
// (position P0 channels 0-31 / 32-63, position P1 likewise); behind them: the input transform of the NEXT step's patches
// (loader, as in conv_wino.hip: a thread owns one column of a 4x4 patch, B^T down the column, the row combination from
// its quad neighbours by DPP), the split of the next position's fragments, the patch loads of the step after next.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels.h"
#include "conv_common.h"
#include "range.h"

typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
// chunk swizzle of the raw patch copy: chunk c of pixel column x at slot c ^ bitrev2((x >> 2) & 3).  The patch-column reads of a
// ds_read_b128 lane group (4 consecutive tiles, alternating channel quads: pixel columns x .. x+9) then hit 16 different slots.
#define RAW_SWZ(x) (((((x) >> 2) & 1) << 1) | (((x) >> 3) & 1))

namespace {
constexpr int TT = 64;                 // output tiles (2x2 pixels each) per block
constexpr int KK = 64;                 // output channels per block
constexpr int BKC = 16;                // input channels per K step
constexpr int VPS = TT * BKC + 8;      // floats per position of the V image (+32 B: the four patch columns of a quad hit different banks)
constexpr int VSTAGE = 16 * VPS;
constexpr size_t WB_LDS_U = (size_t)2 * VSTAGE * sizeof(float) + 6 * 66 * 64 + 1024;      // geometry 42: two V stages + the raw copy (at most 6 x 66 pixels x 64 B) + a dump row
constexpr size_t WB_LDS = (size_t)2 * VSTAGE * sizeof(float) + 512 * 16;      // two V stages (132096 B; the exchange image [16][64][32] fp32 = 131072 B fits) + the patch offsets

__device__ __forceinline__ float quad_2211w(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xF, 0xF, true));
}

__device__ __forceinline__ void split3_pair_w(float v0, float v1, int& q0, int& q1, int& q2)
{
    const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
    q0 = (int)__builtin_amdgcn_perm(u1, u0, 0x07060302);                       // {top16(v1), top16(v0)}
    const float r0 = v0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u), r1 = v1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    const unsigned s0 = __builtin_bit_cast(unsigned, r0), s1 = __builtin_bit_cast(unsigned, r1);
    q1 = (int)__builtin_amdgcn_perm(s1, s0, 0x07060302);
    const float t0 = r0 - __builtin_bit_cast(float, s0 & 0xFFFF0000u), t1 = r1 - __builtin_bit_cast(float, s1 & 0xFFFF0000u);
    q2 = (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0), 0x07060302);
}
}  // namespace

// VAR = 0 is the product; the other values are timing ablations of the diagnostics build (bits listed at the launcher)
// UL (launch geometry 42): the block is a BH x BW rectangle of tiles (8x8, 4x16 or 2x32) and the UNION of its 64 patches --
// (2 BH + 2) x (2 BW + 2) pixels, 20-25 KB per K step instead of 64 KB -- is loaded once, 64 contiguous bytes per 4 lanes, into
// an LDS copy from which the threads take their patch columns (one more barrier per K step).
// H2: the fp16x2 form (kernels.h): V and U as two half terms each, three v_mfma_f32_32x32x16_f16 products per multiply-add.  The
// transformed patch B^T d B is up to 4x the largest pixel, so V is split at a quarter of the layer's calibrated pixel scale.
template <int VAR, bool UL = false, bool H2 = false>
__global__ __launch_bounds__(512) void conv_wino_b3_kernel(ConvParams p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NPW = H2 ? 2 : 3;      // planes per operand
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
    const int nt = (VAR & 1) ? swz / p.MT : swz % p.NT, mt = (VAR & 1) ? swz % p.MT : swz / p.NT;
    const int m0 = mt * TT, n0 = nt * KK;

    const int TH = p.Ho >> 1, TW = p.Wo >> 1, THW = TH * TW;
    const int T = p.wino_T;
    // UL: block rectangle (p.wino_bhs = log2 BH), its origin (image un, tile row uty0, tile column utx0)
    const int bhs = UL ? p.wino_bhs : 0, bws = 6 - bhs, BWm = (1 << bws) - 1;
    const int RW = (2 << bws) + 2, RH = (2 << bhs) + 2;
    int un = 0, uty0 = 0, utx0 = 0;
    if constexpr (UL) {
        const int BX = (TW + BWm) >> bws, BY = (TH + (1 << bhs) - 1) >> bhs;
        un = mt / (BX * BY);
        const int rem = mt - un * (BX * BY);
        const int by = rem / BX;
        uty0 = by << bhs; utx0 = (rem - by * BX) << bws;
    }
    const int nk_all = p.Cin / BKC;
    const int kb = p.ksplit > 1 ? (int)blockIdx.y * p.kt_per_split : 0;
    const int nk = p.ksplit > 1 ? min(p.kt_per_split, nk_all - kb) : nk_all;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ur = make_rsrc(p.wub, p.wub_bytes);

    // ---- loader: item = (tile, channel quad, patch column j); a thread owns quads q and q + 2 of its (tile, column) ----
    const int j = tid & 3, q = (tid >> 2) & 1, tl = tid >> 3;
    // the four row offsets of this thread's patch column are needed once per step only: they live in the spare LDS behind the
    // two V stages (one ds_read_b128 per step) instead of in four registers
    unsigned* aoff_lds = reinterpret_cast<unsigned*>(smem + 2 * VSTAGE) + tid * 4;
    if constexpr (!UL) {
        unsigned a_off[4];
        const int tg = m0 + tl;
        const bool ok = tg < T;
        const int tt = ok ? ((VAR & 1024) ? (tg & ~7) : tg) : 0;      // diagnostics 1024: the 8 tiles of a load instruction fetch the same patch
        const int n = tt / THW, rem = tt - n * THW;
        const int ty = rem / TW, tx = rem - ty * TW;
        // diagnostics 2048 (wrong results): the same bytes with 4 adjacent lanes on 64 contiguous bytes of one pixel
        const int ix = (VAR & 2048) ? 2 * tx - 1 + ((tid >> 2) & 1) : 2 * tx - 1 + j;
        const int qd = (VAR & 2048) ? (tid & 3) : q;
        const bool okx = ok && (unsigned)ix < (unsigned)p.W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * ty - 1 + r;
            a_off[r] = (okx && (unsigned)iy < (unsigned)p.H)
                           ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.xCs + 4 * qd) * 4) : OOB;
        }
        *reinterpret_cast<i32x4*>(aoff_lds) = i32x4{(int)a_off[0], (int)a_off[1], (int)a_off[2], (int)a_off[3]};
    }
    // UL: the raw copy [RH x RW pixels][16 channels] fp32 behind the two V stages; chunk c of pixel (y, x) at slot c ^ RAW_SWZ(x)
    // (the patch-column reads of 4 adjacent pixels x 2 chunks then hit 8 different bank groups).  Slot s = tid + 512 i of the
    // loader = (pixel s >> 2, chunk s & 3): 4 adjacent lanes fetch the 64 contiguous bytes of one pixel.
    float* rawS = smem + 2 * VSTAGE;
    unsigned g_off[4];
    int g_dst[4];
    f32x4 g[4];
    int rr_off = 0;
    if constexpr (UL) {
        const float inv_rw = 1.0f / (float)RW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sl = tid + 512 * i, px = sl >> 2, c = sl & 3;
            int py, pxx;
            divmod_small(px, RW, inv_rw, py, pxx);
            const int iy = 2 * uty0 - 1 + py, ix = 2 * utx0 - 1 + pxx;
            const bool ok = px < RH * RW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            g_off[i] = ok ? (unsigned)((((un * p.H + iy) * p.W + ix) * p.xCs + 4 * c) * 4) : OOB;
            g_dst[i] = px < RH * RW ? px * 16 + ((c ^ RAW_SWZ(pxx)) << 2) : RH * RW * 16 + lane * 4;      // slots past the rectangle: a dump row
        }
        const int tyl = tl >> bws, txl = tl & BWm, pxx = 2 * txl + j;
        rr_off = ((2 * tyl) * RW + pxx) * 16 + ((q ^ RAW_SWZ(pxx)) << 2);      // row r: + r * RW * 16; quad q + 2: ^ 8
    }
    auto load_g = [&](int k) {
        if constexpr ((VAR & 8) != 0) return;
        const unsigned ko = (unsigned)(kb + min(k, nk - 1)) * (BKC * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = buf_load4(xr, g_off[i] != OOB ? g_off[i] + ko : OOB);
    };
    auto store_g = [&]() {
        if constexpr ((VAR & 8) != 0) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(rawS + g_dst[i]) = g[i];
    };
    // V image: [position][tile][16 channels] fp32, the 16-byte chunk c of a tile row at physical slot c ^ ((tile >> 1) & 3):
    // the fragment reads (32 consecutive tiles, one chunk each) and these stores are then bank-conflict free
    const int lsw = (tl >> 2) & 3;      // chunk swizzle of the V image: see the fragment reads
    const int v_dst0 = j * VPS + tl * BKC + ((q ^ lsw) << 2);
    const float sb = j == 1 ? 1.f : -1.f;      // column 3 is stored negated, U negated to match (conv_wino.hip)

    f32x4 d[2][4];
    if constexpr ((VAR & 32) != 0) for (int it = 0; it < 2; ++it) for (int r = 0; r < 4; ++r) d[it][r] = f32x4{1.f, 2.f, (float)lane, 3.f};
    auto load_d = [&](int k, int it) {
        if constexpr ((VAR & 8) != 0 || (VAR & 32) != 0) return;
        const unsigned ko = (unsigned)(kb + min(k, nk - 1)) * (BKC * 4) + (unsigned)it * ((VAR & 2048) ? (unsigned)(2 * p.xCs * 4) : 32u);
        const i32x4 ao = *reinterpret_cast<const i32x4*>(aoff_lds);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            d[it][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)ao[r] != OOB ? (unsigned)ao[r] + ko : OOB, 0, (VAR & 256) ? 2 : 0));
    };
    auto transform = [&](int stage, int it, int i0, int i1) {      // rows i0 .. i1-1 of the transformed patch column
        if constexpr ((VAR & 8) != 0) return;
        if constexpr ((VAR & 64) != 0) { asm volatile("" :: "v"(d[it][0]), "v"(d[it][1]), "v"(d[it][2]), "v"(d[it][3])); return; }
        float* vs = smem + stage * VSTAGE + (it ? v_dst0 + (((q ^ lsw) & 2) ? -8 : 8) : v_dst0);      // quad q + 2: slot (q ^ lsw) ^ 2
        if constexpr (UL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) d[it][r] = *reinterpret_cast<const f32x4*>(rawS + ((rr_off + r * RW * 16) ^ (it ? 8 : 0)));
        }
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            f32x4 vo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float t = i == 0 ? d[it][0][c] - d[it][2][c] : i == 1 ? d[it][1][c] + d[it][2][c]
                              : i == 2 ? d[it][2][c] - d[it][1][c] : d[it][1][c] - d[it][3][c];
                vo[c] = fmaf(sb, quad_2211w(t), t);
            }
            *reinterpret_cast<f32x4*>(vs + i * 4 * VPS) = vo;
        }
    };

    // ---- fragments ----
    const int fr = lane & 31, fh = lane >> 5;
    // ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with the chunk of tile t at slot
    // c ^ ((t >> 2) & 3) the 16 tiles of a group hit 16 different 16-byte slots of the 256-byte bank row
    const int fsw = (fr >> 2) & 3;
    const int a_rd0 = fr * BKC + (((2 * fh) ^ fsw) << 2);            // channels 8h .. 8h+3 of tile fr
    const unsigned b_voff = (unsigned)((n0 + fr) * 32 + fh * 16);
    const unsigned u_pos = (unsigned)p.wino_rows * 32u;               // bytes of one position of one K step of one plane
    const unsigned u_step = 16u * u_pos;
    const unsigned u_plane = (unsigned)nk_all * u_step;

    i32x4 fb[4][NPW];       // weight fragments of the four phases of a step, each requested a whole step ahead (right after the
                            // MFMAs that used its registers): every wait on the in-order vector-memory counter is then for a load
                            // one step old, and the patch loads issued in between keep their lead
    if constexpr ((VAR & 4) != 0) for (int b_ = 0; b_ < 4; ++b_) for (int pl = 0; pl < NPW; ++pl) fb[b_][pl] = i32x4{lane, 1, 2, 3};
    auto load_b = [&](int buf, int k, int pos, int jj) {
        if constexpr ((VAR & 4) != 0) return;
        const unsigned so = (unsigned)(kb + min(k, nk - 1)) * u_step + (unsigned)pos * u_pos + (unsigned)jj * 1024u;
#pragma unroll
        for (int pl = 0; pl < NPW; ++pl) fb[buf][pl] = __builtin_amdgcn_raw_buffer_load_b128(ur, b_voff, so + (unsigned)pl * u_plane, (VAR & 128) ? 2 : 0);
    };
    f32x4 raw[2][2];        // fp32 fragment of one position: [mi][channel half]
    auto read_raw = [&](int stage, int pos, int mi) {
        const float* v = smem + stage * VSTAGE + pos * VPS + mi * 32 * BKC;
        raw[mi][0] = *reinterpret_cast<const f32x4*>(v + a_rd0);
        raw[mi][1] = *reinterpret_cast<const f32x4*>(v + (a_rd0 ^ 4));      // channels 8h+4 .. 8h+7: slot ^ 1
    };
    auto split_raw = [&](i32x4 (&a)[2][NPW], int mi) {
        if constexpr ((VAR & 16) != 0) {
            a[mi][0] = __builtin_bit_cast(i32x4, raw[mi][0]); a[mi][1] = __builtin_bit_cast(i32x4, raw[mi][1]); a[mi][NPW - 1] = a[mi][0] ^ a[mi][1];
            return;
        }
        if constexpr (H2) {
            f16x8w h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = raw[mi][e >> 2][e & 3];
                h[e] = (_Float16)(v * xs);
                l[e] = (_Float16)__builtin_fmaf(v, xs, -(float)h[e]);      // exact residual, then rounded to half
            }
            a[mi][0] = __builtin_bit_cast(i32x4, h);
            a[mi][NPW - 1] = __builtin_bit_cast(i32x4, l);
            return;
        }
        int q0[4], q1[4], q2[4];
        split3_pair_w(raw[mi][0][0], raw[mi][0][1], q0[0], q1[0], q2[0]);
        split3_pair_w(raw[mi][0][2], raw[mi][0][3], q0[1], q1[1], q2[1]);
        split3_pair_w(raw[mi][1][0], raw[mi][1][1], q0[2], q1[2], q2[2]);
        split3_pair_w(raw[mi][1][2], raw[mi][1][3], q0[3], q1[3], q2[3]);
        a[mi][0] = i32x4{q0[0], q0[1], q0[2], q0[3]};
        a[mi][1] = i32x4{q1[0], q1[1], q1[2], q1[3]};
        a[mi][NPW - 1] = i32x4{q2[0], q2[1], q2[2], q2[3]};
    };

    f32x16 acc[2][2][2];      // [own position][tile half mi][channel group jj]
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[pi][mi][jj][e] = 0.f;

    // 6 MFMAs on one accumulator tile: weight fragment = A operand (rows = channels), tile fragment = B operand (columns = tiles);
    // smallest terms first
    auto mma = [&](int pi, int jj, int buf, const i32x4 (&a)[2][NPW], int mi) {
        if constexpr (H2) {      // the two cross terms, then hi * hi
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if constexpr ((VAR & 2) != 0) { asm volatile("" :: "v"(fb[buf][PB[t]]), "v"(a[mi][PA[t]])); continue; }
                acc[pi][mi][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, fb[buf][PB[t]]),
                                                                         __builtin_bit_cast(f16x8w, a[mi][PA[t]]), acc[pi][mi][jj], 0, 0, 0);
            }
        } else {
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                if constexpr ((VAR & 2) != 0) { asm volatile("" :: "v"(fb[buf][PB[t] % NPW]), "v"(a[mi][PA[t] % NPW])); continue; }
                acc[pi][mi][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8w, fb[buf][PB[t] % NPW]),
                                                                          __builtin_bit_cast(bf16x8w, a[mi][PA[t] % NPW]), acc[pi][mi][jj], 0, 0, 0);
            }
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    const int P0 = 2 * wave, P1 = 2 * wave + 1;
    i32x4 a0[2][NPW], a1[2][NPW];

    // ---- prologue: stage 0 = step 0 ----
    if constexpr (UL) {
        load_g(0);
        load_b(0, 0, P0, 0); load_b(1, 0, P0, 1); load_b(2, 0, P1, 0); load_b(3, 0, P1, 1);
        store_g();
        load_g(1);
        lds_barrier();                       // raw copy = patches of step 0
        transform(0, 0, 0, 4);
        transform(0, 1, 0, 4);
        lds_barrier();                       // V stage 0 complete, raw copy read by everybody
        store_g();
        load_g(2);
        lds_barrier();                       // raw copy = patches of step 1
        read_raw(0, P0, 0);
        split_raw(a0, 0);
    } else {
    load_d(0, 0); load_d(0, 1);
        load_b(0, 0, P0, 0); load_b(1, 0, P0, 1); load_b(2, 0, P1, 0); load_b(3, 0, P1, 1);
        transform(0, 0, 0, 4);
        transform(0, 1, 0, 4);
        load_d(1, 0); load_d(1, 1);
        lds_barrier();
        read_raw(0, P0, 0);
        split_raw(a0, 0);

    }

    // One K step = four phases of 12 MFMAs (one position x one 32-channel group each).  What prepares the next fragments and
    // the next step is spread over the phases, and the scheduling groups at the end of each phase interleave it with the
    // MFMAs (one MFMA, then a handful of vector instructions issued while it runs).  Vector-memory loads are issued at the
    // phase ends in a fixed order, each a whole step before its use.
#define WB_FENCE() __builtin_amdgcn_sched_barrier(0)      /* the compiler keeps the order of what is on either side */
#define WB_INTERLEAVE(nv)                                                                             \
    do {                                                                                              \
        _Pragma("unroll") for (int g_ = 0; g_ < (H2 ? 6 : 12); ++g_) {                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x002, (H2 ? 2 : 1) * (nv), 0);                      \
        }                                                                                             \
    } while (0)
    if constexpr (UL) {
        // raw copy = patches of step k+1, g = patches of step k+2 (in flight); barriers: V (end of phase 1), raw (end of phase 3)
        for (int k = 0; k < nk; ++k) {
            const int cur = k & 1;
            // phase 0: P0, channels 0-31 | P0's second tile half, patch item 0 of step k+1: raw copy -> transform -> V stage cur^1
            read_raw(cur, P0, 1);
            mma(0, 0, 0, a0, 0);
            split_raw(a0, 1);
            mma(0, 0, 0, a0, 1);
            transform(cur ^ 1, 0, 0, 4);
            WB_INTERLEAVE(8);
            WB_FENCE();
            load_b(0, k + 1, P0, 0);
            WB_FENCE();
            // phase 1: P0, channels 32-63 | P1's first tile half, patch item 1
            read_raw(cur, P1, 0);
            mma(0, 1, 1, a0, 0);
            split_raw(a1, 0);
            mma(0, 1, 1, a0, 1);
            transform(cur ^ 1, 1, 0, 4);
            WB_INTERLEAVE(8);
            WB_FENCE();
            load_b(1, k + 1, P0, 1);
            lds_barrier();                  // V stage cur^1 complete; the raw copy has been read by everybody
            WB_FENCE();
            // phase 2: P1, channels 0-31 | P1's second tile half
            read_raw(cur, P1, 1);
            mma(1, 0, 2, a1, 0);
            split_raw(a1, 1);
            mma(1, 0, 2, a1, 1);
            WB_INTERLEAVE(4);
            WB_FENCE();
            load_b(2, k + 1, P1, 0);
            WB_FENCE();
            // phase 3: P1, channels 32-63 | P0's first tile half of step k+1; the patches of step k+2 go to the raw copy
            read_raw(cur ^ 1, P0, 0);
            mma(1, 1, 3, a1, 0);
            split_raw(a0, 0);
            mma(1, 1, 3, a1, 1);
            store_g();
            WB_INTERLEAVE(4);
            WB_FENCE();
            load_b(3, k + 1, P1, 1);
            load_g(k + 3);
            lds_barrier();                  // raw copy = patches of step k+2; V stage cur is free
            WB_FENCE();
        }
    } else
    for (int k = 0; k < nk; ++k) {
        const int cur = k & 1;
        // phase 0: P0, channels 0-31 | split of P0's second tile half, transform of patch item 0 of step k+1
        read_raw(cur, P0, 1);
        mma(0, 0, 0, a0, 0);
        split_raw(a0, 1);
        mma(0, 0, 0, a0, 1);
        transform(cur ^ 1, 0, 0, 4);
        WB_INTERLEAVE(8);
        WB_FENCE();
        load_b(0, k + 1, P0, 0);
        WB_FENCE();
        // phase 1: P0, channels 32-63 | P1's first tile half, first half of the transform of patch item 1
        read_raw(cur, P1, 0);
        mma(0, 1, 1, a0, 0);
        split_raw(a1, 0);
        mma(0, 1, 1, a0, 1);
        transform(cur ^ 1, 1, 0, 2);
        WB_INTERLEAVE(6);
        WB_FENCE();
        load_b(1, k + 1, P0, 1);
        load_d(k + 2, 0);
        WB_FENCE();
        // phase 2: P1, channels 0-31 | P1's second tile half, the rest of patch item 1
        read_raw(cur, P1, 1);
        mma(1, 0, 2, a1, 0);
        split_raw(a1, 1);
        mma(1, 0, 2, a1, 1);
        transform(cur ^ 1, 1, 2, 4);
        WB_INTERLEAVE(6);
        WB_FENCE();
        load_b(2, k + 1, P1, 0);
        load_d(k + 2, 1);
        lds_barrier();                      // stage cur^1 is complete; stage cur was last read at the top of phase 2
        WB_FENCE();
        // phase 3: P1, channels 32-63 | P0's fragments of step k+1 (the second half is split in phase 0)
        read_raw(cur ^ 1, P0, 0);
        mma(1, 1, 3, a1, 0);
        split_raw(a0, 0);
        mma(1, 1, 3, a1, 1);
        WB_INTERLEAVE(4);
        WB_FENCE();
        load_b(3, k + 1, P1, 1);
        WB_FENCE();
    }
#undef WB_FENCE
#undef WB_INTERLEAVE

    // ---- exchange + output transform + epilogue ----
    const int et = tid >> 3, ecq = tid & 7;                 // this thread finishes tile et, channels 4*ecq .. +3 of each round
    int tg, en, ety, etx;
    bool tile_ok;
    if constexpr (UL) {
        en = un; ety = uty0 + (et >> bws); etx = utx0 + (et & BWm);
        tile_ok = ety < TH && etx < TW;
        tg = 0;
    } else {
        tg = m0 + et;
        int erem;
        const float inv_thw = 1.0f / (float)THW, inv_tw = 1.0f / (float)TW;
        divmod_small(tg < T ? tg : 0, THW, inv_thw, en, erem);
        divmod_small(erem, TW, inv_tw, ety, etx);
        tile_ok = tg < T;
    }
    const unsigned pix00 = (unsigned)((en * p.Ho + 2 * ety) * p.Wo + 2 * etx);
    const unsigned pix[4] = {pix00, pix00 + 1, pix00 + (unsigned)p.Wo, pix00 + (unsigned)p.Wo + 1};
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res : p.y, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t y2r = make_rsrc(p.y2 ? p.y2 : p.y, p.y2 ? p.y2_bytes : 0u);
    const size_t slab = (size_t)blockIdx.y * p.M * p.Cout_store;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.ksplit > 1 ? p.ws + slab : p.y, p.ksplit > 1 ? (unsigned)((size_t)p.M * p.Cout_store * 4) : 0u);
    float* X = smem;                                        // [16][64][32], chunk c of a tile row at slot c ^ (tile & 7)
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
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int tile = mi * 32 + fr, c = 2 * g + fh;
                    f32x4 v = {acc[pi][mi][jj][4 * g], acc[pi][mi][jj][4 * g + 1], acc[pi][mi][jj][4 * g + 2], acc[pi][mi][jj][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(X + ((2 * wave + pi) * 64 + tile) * 32 + ((c ^ (tile & 7)) << 2)) = v;
                }
        lds_barrier();
        f32x4 m[16];
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) m[pp] = *reinterpret_cast<const f32x4*>(X + pp * 64 * 32 + x_rd);
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

// shapes the kernel takes: those of conv_wino.hip with input channels in multiples of 16
bool conv_wino_b3_eligible(const ConvParams& p)
{
    return conv_wino_eligible(p) && p.Cin % BKC == 0;
}

// U = G g G^T of an OIHW 3x3 weight (double, rounded to fp32 once), split exactly into three bf16 planes, each laid out
// [C/16][16 positions][rows][16]; rows = conv_wino_rows(Cout_store).  Column 3 negated as in conv_wino_pack.
void conv_wino_b3_pack(const float* w, int Cout, int Cin, int rows, std::vector<unsigned short>& out)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const size_t plane = (size_t)(Cin / BKC) * 16 * rows * BKC;
    out.assign(3 * plane, 0);
    auto top = [](float v) { unsigned u; memcpy(&u, &v, 4); u &= 0xFFFF0000u; float r; memcpy(&r, &u, 4); return r; };
    for (int k = 0; k < Cout; ++k)
        for (int c = 0; c < Cin; ++c) {
            const float* g = w + ((size_t)k * Cin + c) * 9;
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) tmp[i][b] = G[i][0] * g[0 * 3 + b] + G[i][1] * g[1 * 3 + b] + G[i][2] * g[2 * 3 + b];
            for (int i = 0; i < 4; ++i)
                for (int jj = 0; jj < 4; ++jj) {
                    double u = tmp[i][0] * G[jj][0] + tmp[i][1] * G[jj][1] + tmp[i][2] * G[jj][2];
                    if (jj == 3) u = -u;
                    const float f = (float)u;
                    const float h0 = top(f), r1 = f - h0, h1 = top(r1), r2 = r1 - h1, h2 = top(r2);
                    const float hs[3] = {h0, h1, h2};
                    const size_t at = (((size_t)(c / BKC) * 16 + (i * 4 + jj)) * rows + k) * BKC + (c % BKC);
                    for (int pl = 0; pl < 3; ++pl) {
                        unsigned bits; const float hv = hs[pl]; memcpy(&bits, &hv, 4);
                        out[pl * plane + at] = (unsigned short)(bits >> 16);
                    }
                }
        }
}

// The fp16x2 form of the same planes (ConvParams::wubh): U * 2^q[k] as hi + lo, two half terms, q[k] putting the largest |U| of output
// channel k into [2^14, 2^15) (see pack_h2r in accel_hip.cpp); qexp receives q per row.
void conv_wino_b3_pack_h2(const float* w, int Cout, int Cin, int rows, std::vector<unsigned short>& out, std::vector<int>& qexp)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const size_t plane = (size_t)(Cin / BKC) * 16 * rows * BKC;
    out.assign(2 * plane, 0);
    qexp.assign(rows, 0);
    std::vector<float> U((size_t)Cin * 16);
    for (int k = 0; k < Cout; ++k) {
        float amax = 0.f;
        for (int c = 0; c < Cin; ++c) {
            const float* g = w + ((size_t)k * Cin + c) * 9;
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) tmp[i][b] = G[i][0] * g[0 * 3 + b] + G[i][1] * g[1 * 3 + b] + G[i][2] * g[2 * 3 + b];
            for (int i = 0; i < 4; ++i)
                for (int jj = 0; jj < 4; ++jj) {
                    double u = tmp[i][0] * G[jj][0] + tmp[i][1] * G[jj][1] + tmp[i][2] * G[jj][2];
                    if (jj == 3) u = -u;
                    const float f = (float)u;
                    U[(size_t)c * 16 + i * 4 + jj] = f;
                    amax = std::max(amax, std::fabs(f));
                }
        }
        if (amax > 0.f && std::isfinite(amax)) { int e; std::frexp(amax, &e); qexp[k] = 15 - e; }
        for (int c = 0; c < Cin; ++c)
            for (int pos = 0; pos < 16; ++pos) {
                const float v = std::ldexp(U[(size_t)c * 16 + pos], qexp[k]);
                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                const size_t at = (((size_t)(c / BKC) * 16 + pos) * rows + k) * BKC + (c % BKC);
                memcpy(&out[at], &hi, 2);
                memcpy(&out[plane + at], &lo, 2);
            }
    }
}

// geometry 42: tile-block shape for an output of TH x TW tiles, and the number of blocks
long conv_wino_b3u_blocks(const ConvParams& p, int* bhs)
{
    const int TH = p.Ho / 2, TW = p.Wo / 2;
    const int s = TH >= 8 ? 3 : TH >= 4 ? 2 : 1;
    if (bhs) *bhs = s;
    const int BH = 1 << s, BW = 64 >> s;
    const long n = p.M / ((long)p.Ho * p.Wo);
    return n * ((TH + BH - 1) / BH) * ((TW + BW - 1) / BW);
}

hipError_t launch_conv_wino_b3(const ConvParams& p0, hipStream_t st, bool union_loader)
{
    ConvParams p = p0;
    if (!conv_wino_b3_eligible(p) || !p.wub) return hipErrorInvalidValue;
    p.wino_T = p.M / 4;
    p.MT = (p.wino_T + TT - 1) / TT;
    if (union_loader) p.MT = (int)conv_wino_b3u_blocks(p, &p.wino_bhs);
    p.NT = p.wino_rows / KK;
    const size_t lds = union_loader ? WB_LDS_U : WB_LDS;
    auto go = [&](auto kern) -> hipError_t {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(p.MT * p.NT, p.ksplit > 1 ? p.ksplit : 1), dim3(512), lds, st, p);
        return hipSuccess;
    };
    hipError_t le;
    if (union_loader) le = p.f16 == 3 ? go(&conv_wino_b3_kernel<0, true, true>) : go(&conv_wino_b3_kernel<0, true>);
    else le = p.f16 == 3 ? go(&conv_wino_b3_kernel<0, false, true>) : go(&conv_wino_b3_kernel<0>);
    if (le != hipSuccess) return le;
    if (p.ksplit > 1) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_splitk_reduce(p, 1, st);
    }
    return hipGetLastError();
}
