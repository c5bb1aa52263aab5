// 3x3 / stride 1 / dilation 1 or 2 convolutions with FEW output channels in the fp16x2 form (launch geometry 78): the offset branches of
// the deformable layers (512 -> 18 at dilation 2 in res5 of the ResNet-101 trunk, 512 -> 72 in the ResNet-18 one).
//
// On the implicit-GEMM kernels these layers are bound by their PIXEL path, whatever the width of the matrix tile: every K step gathers
// a 128-pixel x 32-channel tile for ONE tap, so every pixel is fetched from L2, split into its half pair and written to LDS nine times
// (1.2 GB through L2 for a 134 MB input at 8 clips per call: 190 us on geometries 9, 77 and a 128x32 strip alike,
// profiles/r05_strip_geometry78_experiment.log).  Here a block owns an 8 x 16 patch of output pixels and a strip of 32 output channels;
// per chunk of 32 input channels it stages the patch WITH ITS HALO -- (8 + 2d) x (16 + 2d) pixels -- once, already split, and the
// nine taps read their fragments from it at shifted addresses: 1.9x (d = 2) / 1.4x (d = 1) the input instead of 9x.
//
// Work split: the 18 sub-steps of a chunk (tap x half K step) are dealt round-robin to the four wavefronts, each of which multiplies
// its sub-step's weight fragment -- fetched global -> VGPR in MFMA order, the planes conv_b3r.hip uses -- with ALL four 32-pixel
// tiles of the patch: no fragment is fetched twice (with the wavefronts on different pixel tiles instead, every one of them would
// pull the same 36 KB of weights per chunk through the CU's vector cache: 590 KB per chunk and CU against 147).  A chunk's fragments
// are requested together, one chunk ahead and in front of the next halo's loads; the halo is staged in the other of two LDS stages
// (one barrier per chunk).  The four partial sums meet in the epilogue: a two-step exchange through LDS in a fixed order, after which wavefront w finishes pixel tile w.
// The range slot of the input gives the pixel scale, the outputs raise theirs (range.h).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "conv_common.h"
#include "range.h"

typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));

namespace {
constexpr int HT_H = 8, HT_W = 16;      // output pixels of a block
// LDS image of a plane: 64 bytes per pixel (32 halves, no padding), rows of HWP = 20 pixels (1280 bytes = five bank rows of 256), the
// 16-byte chunk c of the pixel in column hx stored at slot c ^ ((hx >> 2) & 3).  A ds_read_b128 is serviced 16 lanes at a time, and
// a group's lanes sit on 16 consecutive COLUMNS of two adjacent patch rows (8 + 8): with the row pitch a multiple of the bank row and
// the swizzle a function of the column, the 16 slots are distinct for every tap.  (First version: 80-byte pixels, rows of 16 + 2d --
// two-way conflicts on half of each group: 85 us for the 512 -> 18 layer at 8 clips per call.)
constexpr int HLDK = 32, HWP = 20;
constexpr size_t halo_lds(int d)      // two stages (the epilogue's exchange image, 32 KB, lives in them) + 16 bytes per thread for the loader slots past the halo
{
    return (size_t)2 * 2 * (HT_H + 2 * d) * HWP * HLDK * 2 + 4096;
}
}  // namespace

template <int D>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(ConvParams p, size_t wplane, int rowsB)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int HH = HT_H + 2 * D, HW = HT_W + 2 * D, NPX = HH * HW;
    constexpr int NS = (NPX * 4 + 255) / 256;      // loader slots per thread: a slot = 8 channels (32 bytes) of one halo pixel
    constexpr int PL = HH * HWP * HLDK;            // halves per plane
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];      // [stage][plane hi / lo][halo pixel][HLDK]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nblk = p.MT * p.NT, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int nt = swz % p.NT, mt = swz / p.NT;
    const int TXn = (p.Wo + HT_W - 1) / HT_W, TYn = (p.Ho + HT_H - 1) / HT_H;
    const int img = mt / (TXn * TYn), trem = mt - img * (TXn * TYn);
    const int y0 = (trem / TXn) * HT_H, x0 = (trem % TXn) * HT_W, n0 = nt * 32;

    const int nchunk = p.Cin / 32;
    const RangeScale rs = range_prologue(p.xr);
    const float xs = rs.s;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);

    // ---- halo loader ----
    unsigned g_off[NS];
    int g_dst[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = tid + 256 * i, hp = s >> 2, c8 = s & 3;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 - D + hy, ix = x0 - D + hx;
        const bool ok = hp < NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        g_off[i] = ok ? (unsigned)((((img * p.H + iy) * p.W + ix) * p.xCs + c8 * 8) * 4) : OOB;
        g_dst[i] = hp < NPX ? (hy * HWP + hx) * HLDK + ((c8 ^ ((hx >> 2) & 3)) << 3) : -1;
    }
    f32x4 g[NS][2];
#ifdef HKO_NO_LOADG
    for (int i = 0; i < NS; ++i) g[i][0] = g[i][1] = f32x4{1.f, 2.f, 3.f, (float)lane};
#endif
    auto load_g1 = [&](int i, int c) {      // slot i of chunk c
#ifdef HKO_NO_LOADG      // (HKO_*: knock-out builds for timing, WRONG results: scripts/ab_halo.sh)
        return;
#endif
        const unsigned off = g_off[i] != OOB ? g_off[i] + (unsigned)min(c, nchunk - 1) * 128u : OOB;
        g[i][0] = buf_load4(xr, off);
        g[i][1] = buf_load4(xr, off != OOB ? off + 16u : OOB);
    };
    auto store_g1 = [&](int i, int stage) {      // slot i: split into the half pair, into its place of the stage
#ifdef HKO_NO_STOREG
        return;
#endif
        unsigned short* dst = smem_h + stage * (2 * PL);
        f16x8h h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = g[i][e >> 2][e & 3];
            h[e] = (_Float16)(v * xs);
            l[e] = (_Float16)__builtin_fmaf(v, xs, -(float)h[e]);      // exact residual, then rounded to half
        }
        // (a slot past the halo -- last pass only -- writes to 16 bytes of its own behind the stages: the K loop has no branches, so the
        // compiler's vmcnt / lgkmcnt bookkeeping stays exact)
        unsigned short* dummy = smem_h + 4 * PL + tid * 8;
        *reinterpret_cast<i32x4*>(g_dst[i] >= 0 ? dst + g_dst[i] : dummy) = __builtin_bit_cast(i32x4, h);
        *reinterpret_cast<i32x4*>(g_dst[i] >= 0 ? dst + PL + g_dst[i] : dummy) = __builtin_bit_cast(i32x4, l);
    };

    // ---- fragments ----
    const int frow = lane & 31, half = lane >> 5;
    // pixel tile i (0..3) = patch rows 2i, 2i+1: lane frow holds pixel (2i + (frow >> 4), frow & 15)
    const int a_row = (frow >> 4) * HWP * HLDK, a_col = frow & 15;
    i32x4 fa[2][4][2];      // [buffer][pixel tile][plane]: the fragments of the next sub-step are read under the products of this one
#ifdef HKO_NO_READFA
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) fa[b][i][0] = fa[b][i][1] = i32x4{0x3C003C00, lane, 2, 3};
#endif
    auto read_fa = [&](int buf, int stage, int s) {
#ifdef HKO_NO_READFA
        return;
#endif
        const int t = s >> 1, kb = s & 1, ty = t / 3, tx = t - 3 * ty;
        const int hx = a_col + tx * D;
        const unsigned short* a = smem_h + stage * (2 * PL) + a_row + (ty * D * HWP + hx) * HLDK + (((2 * kb + half) ^ ((hx >> 2) & 3)) << 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[buf][i][0] = *reinterpret_cast<const i32x4*>(a + (2 * i * HWP) * HLDK);
            fa[buf][i][1] = *reinterpret_cast<const i32x4*>(a + PL + (2 * i * HWP) * HLDK);
        }
    };
    const unsigned short* wbase = reinterpret_cast<const unsigned short*>(p.w);
    const __amdgpu_buffer_rsrc_t wall = make_rsrc(wbase, (unsigned)(2 * wplane) + (unsigned)((size_t)rowsB * p.K_pad * 2) + (unsigned)(rowsB * 128));
    const unsigned b_voff = (unsigned)((n0 + frow) * 32 + half * 16);
    const unsigned hstep = (unsigned)rowsB * 32u, plane_b = (unsigned)(2 * wplane);
    // fragment j of a chunk = the one of this wavefront's j-th sub-step there; it is requested one chunk ahead, right behind the products
    // that used its registers.  (Vector-memory results return in order: every request below is placed so that what was requested
    // before it is about a chunk old when it is first waited for -- a fragment requested behind a fresh halo load would wait out
    // that load's HBM latency.)
    i32x4 fb[5][2];      // [sub-step of the chunk][plane]
#ifdef HKO_NO_LOADB
    for (int j = 0; j < 5; ++j) fb[j][0] = fb[j][1] = i32x4{0x3C003C00, lane, 2, 3};
#endif
    auto load_b1 = [&](int j, int c) {
#ifdef HKO_NO_LOADB
        return;
#endif
        const int cc = min(c, nchunk - 1), s = min(((wave + 2 * (cc & 1)) & 3) + 4 * j, 17);      // (a fifth fragment that is not used is requested all the same)
        const unsigned hs = 2u * (unsigned)((s >> 1) * nchunk + cc) + (unsigned)(s & 1);
        fb[j][0] = __builtin_amdgcn_raw_buffer_load_b128(wall, b_voff, hs * hstep, 0);
        fb[j][1] = __builtin_amdgcn_raw_buffer_load_b128(wall, b_voff, plane_b + hs * hstep, 0);
    };
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    auto mma = [&](int buf, const i32x4 (&f)[2]) {      // product-major: four independent accumulators between two dependent products
        const f16x8h b0 = __builtin_bit_cast(f16x8h, f[0]), b1 = __builtin_bit_cast(f16x8h, f[1]);
#ifdef HKO_NO_MMA
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[buf][i][0]), "v"(fa[buf][i][1]), "v"(b0), "v"(b1));
        return;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i)      // the two cross terms first
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8h, fa[buf][i][1]), b0, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8h, fa[buf][i][0]), b1, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8h, fa[buf][i][0]), b0, acc[i], 0, 0, 0);
    };

    // ---- K loop: chunk c's sub-steps s with (18 c + s) % 4 == wave; one barrier per chunk ----
    // Position j of a chunk: the products of sub-step j (fragments of sub-step j+1 read from LDS under them) | slot j of the NEXT
    // chunk's halo split and staged into the other stage | fragment j of the next chunk requested | slot j of the halo of the chunk
    // after that requested.
#pragma unroll
    for (int i = 0; i < NS; ++i) load_g1(i, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) load_b1(j, 0);
#pragma unroll
    for (int i = 0; i < NS; ++i) { store_g1(i, 0); load_g1(i, 1); }
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1, sbase = (wave + 2 * cur) & 3;
        read_fa(0, cur, sbase);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int s = sbase + 4 * j;
            if (j < 4) {
                read_fa((j + 1) & 1, cur, min(s + 4, 17));      // (j = 3: read whether or not a fifth sub-step follows)
                mma(j & 1, fb[j]);
            } else if (s < 18) mma(j & 1, fb[j]);              // wave-uniform; nothing but matrix instructions inside
            if (j < NS) store_g1(j, cur ^ 1);      // (that stage was last read in chunk c - 1, before the barrier that ended it; behind the last chunk: unused)
#ifdef HALO_SGB      // one matrix instruction, then three vector instructions of the split, twelve times
            for (int q_ = 0; q_ < 12; ++q_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
#endif
            __builtin_amdgcn_sched_barrier(0);
            load_b1(j, c + 1);
            if (j < NS) load_g1(j, c + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                            // stage cur^1 holds chunk c + 1; stage cur has been read by everybody
    }

    // ---- the four partial sums meet: wavefront w ends up with pixel tile w, summed in a fixed order ----
    float* X = reinterpret_cast<float*>(smem_h);      // (behind the loop's last barrier: no stage is read any more)
    f32x16 r0, r1;
    {
        float* mine = X + (wave * 2) * 1024 + lane;
        const float* theirs = X + ((wave ^ 2) * 2) * 1024 + lane;
        if (wave < 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { mine[e * 64] = acc[2][e]; mine[1024 + e * 64] = acc[3][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) { mine[e * 64] = acc[0][e]; mine[1024 + e * 64] = acc[1][e]; }
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { r0[e] = acc[0][e] + theirs[e * 64]; r1[e] = acc[1][e] + theirs[1024 + e * 64]; }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) { r0[e] = acc[2][e] + theirs[e * 64]; r1[e] = acc[3][e] + theirs[1024 + e * 64]; }
        }
        __syncthreads();
    }
    f32x16 f;
    {
        float* mine = X + wave * 1024 + lane;
        const float* theirs = X + (wave ^ 1) * 1024 + lane;
        if (wave & 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) mine[e * 64] = r0[e];
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) mine[e * 64] = r1[e];
        }
        __syncthreads();
        if (wave & 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) f[e] = r1[e] + theirs[e * 64];
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) f[e] = r0[e] + theirs[e * 64];
        }
    }

    // ---- epilogue of pixel tile `wave`: scale / shift -> activation -> store; C layout: column = lane & 31 (channel), row = (e&3) + 8 (e>>2) + 4 half ----
    const int co = n0 + frow;
    const bool cok = co < p.Cout_store;
    const int cc = cok ? co : 0;
    const float sc = p.scale[cc] * rs.inv, sf = p.shift[cc];
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    float vmax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ml = (e & 3) + 8 * (e >> 2) + 4 * half;
        const int y = y0 + 2 * wave + (ml >> 4), x = x0 + (ml & 15);
        const bool ok = cok && y < p.Ho && x < p.Wo;
        float v = f[e] * sc + sf;
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = v > 0.f ? v : v * p.slope;
        buf_store1(yr, ok ? (unsigned)((((img * p.Ho + y) * p.Wo + x) * p.yCs + co) * 4) : OOB, v);
        if (ok) vmax = fmaxf(vmax, fabsf(v));
    }
    if (p.yr) range_note_wave(p.yr, range_abs_bits(vmax), (unsigned)(blockIdx.x * 4 + wave));
#endif
}

// geometry 78 takes: 3x3, stride 1, dilation = padding = 1 or 2 (same-size output), whole chunks of 32 input channels, one output,
// no residual, and the fp16x2 planes of conv_b3r.hip (ConvParams::wh2r)
bool conv_halo_eligible(const ConvParams& p)
{
    return p.kh == 3 && p.kw == 3 && p.sh == 1 && p.sw == 1 && p.dh == p.dw && (p.dh == 1 || p.dh == 2) && p.ph == p.dh && p.pw == p.dw &&
           !p.deconv2x && !p.narrow && p.Cin % 32 == 0 && p.K_pad == 9 * p.Cin && p.Ho == p.H && p.Wo == p.W && !p.res && !p.y2 &&
           !p.x_half && !p.y_half && p.Cout_store <= 128;
}

hipError_t launch_conv_halo(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    if (!conv_halo_eligible(p) || p.f16 != 3 || p.ksplit > 1) return hipErrorInvalidValue;
    const long n = p.M / ((long)p.Ho * p.Wo);
    p.MT = (int)(n * ((p.Ho + HT_H - 1) / HT_H) * ((p.Wo + HT_W - 1) / HT_W));
    p.NT = (p.Cout_store + 31) / 32;
    const int rowsB = (int)(p.w_bytes / ((unsigned)p.K_pad * 2u));
    if (p.dh == 2) {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_halo_kernel<2>), halo_lds(2)); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv_halo_kernel<2>, dim3(p.MT * p.NT), dim3(256), halo_lds(2), st, p, p.w_plane, rowsB);
    } else {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_halo_kernel<1>), halo_lds(1)); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv_halo_kernel<1>, dim3(p.MT * p.NT), dim3(256), halo_lds(1), st, p, p.w_plane, rowsB);
    }
    return hipGetLastError();
}
