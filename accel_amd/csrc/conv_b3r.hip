// bf16x3 implicit GEMM, second generation ("b3r", launch geometries 76 / 77 / 79 / 80 / 81): fp32 values on the bf16 matrix cores as in
// conv_igemm_b3_kernel (each operand split EXACTLY into three bf16 terms, six v_mfma_f32_32x32x16_bf16 products per
// multiply-add, fp32 accumulate -- see conv_igemm.hip), with the staging reorganised around what bounded that kernel.
//
// What bounded it (profiles/r02): per K step a 128x128 block reads 144 KB of operand fragments from LDS and
// writes 48 KB (the three planes of both tiles) -- 1536 cycles of the CU's 128 B/clk LDS port against 1536 cycles of
// matrix work, so neither pipe could be kept busy while the other ran -- and its phases (multiply | barrier | split + stage
// | barrier | issue loads) were serial, overlapped only by a second resident block.
//
// Here:
//   * the WEIGHT fragments never touch LDS.  The host stores the three bf16 planes in MFMA fragment order
//     ([class][K step][half step][row][16]: the 16 bytes a lane feeds to one MFMA are contiguous, a wavefront's fetch is one
//     contiguous kilobyte), every wavefront loads its own B fragments global -> VGPR one half step ahead.  Weights are the
//     same for every pixel tile, so these loads are L2 hits; with a 2 x 4 wavefront grid the two M halves fetch the same
//     fragment twice, with the 1 x 8 / 1 x 4 grids (every wavefront owns 32 columns of the tile) nothing is fetched twice.
//     LDS traffic per K step drops to 96 KB of reads + 24 KB of writes (940 cycles), and nothing waits on a weight DMA.
//   * the pixel tile (split into its three planes by the loader, as before) is DOUBLE-buffered in LDS, 60 KB per block, two
//     blocks per CU: tile k+1 is split and stored while tile k is multiplied, ONE barrier per K step, and the loop body has
//     no branches (loads past the end are absorbed by the slack entries of the tap table / the slack rows of the planes),
//     so the compiler's in-order vmcnt accounting stays exact.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"
#include "conv_common.h"
#include "conv_epilogue.h"

typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8r __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_pair_r(float v0, float v1, unsigned& q0, unsigned& q1, unsigned& q2)
{
    const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
    q0 = __builtin_amdgcn_perm(u1, u0, 0x07060302);                       // {top16(v1), top16(v0)}
    const float r0 = v0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u), r1 = v1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    const unsigned s0 = __builtin_bit_cast(unsigned, r0), s1 = __builtin_bit_cast(unsigned, r1);
    q1 = __builtin_amdgcn_perm(s1, s0, 0x07060302);
    const float t0 = r0 - __builtin_bit_cast(float, s0 & 0xFFFF0000u), t1 = r1 - __builtin_bit_cast(float, s1 & 0xFFFF0000u);
    q2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0), 0x07060302);
}

// wplane: bf16 elements between the planes; rowsB: rows of a plane (Cout_store rounded up to 128)
// NPL = 3: bf16x3 (three planes per operand, six products); NPL = 1: the fp16-MFMA mode of plan option dtype=f16 (operands rounded to
// half -- weights on the host, pixels by the loader --, ONE plane, one v_mfma_f32_32x32x16_f16 product, fp32 accumulate) on the same staging
template <int BM, int BN, int WGM, int WGN, bool FAST, int ABL = 0, int NPL = 3>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN == 8 && BM * BN <= 128 * 128 ? 4 : 2)) void conv_b3r_kernel(ConvParams p, size_t wplane, int rowsB)
{
    constexpr int BK = 32, LDK = BK + 8;            // bf16 elements per staged row (80 bytes: conflict-free ds_read_b128)
    constexpr int NTHR = 64 * WGM * WGN;
    constexpr int CPR = BK / 8, RP = NTHR / CPR;    // 8-wide chunks per row, rows per pass
    constexpr int MI = BM / (WGM * 32), NI = BN / (WGN * 32);
    constexpr int AR = BM / RP;
    static_assert(BM % RP == 0, "tile / thread-count mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_r[];      // As[2][3][BM][LDK]
    constexpr int STAGE = NPL * BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nblk = p.MT * p.NT, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int nt = swz % p.NT, mt = swz / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    int ph = p.ph, pw = p.pw;
    const unsigned short* wbase = reinterpret_cast<const unsigned short*>(p.w);
    int py = 0, px = 0;
    if (p.deconv2x) {
        py = blockIdx.y >> 1; px = blockIdx.y & 1;
        ph = 1 - py; pw = 1 - px;
        wbase += (size_t)blockIdx.y * p.w_class_stride;
    }
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const int HoWo = p.Ho * p.Wo;
    // fp16x2 form: the power of two that centres the pixels in the half range, from the range slot of the input tensor (range.h)
    RangeScale rs; rs.s = 1.f; rs.inv = 1.f;
    if constexpr (NPL == 2) rs = range_prologue(p.xr);
    const float xs = rs.s;

    // ---- pixel-tile staging (same row order as conv_igemm_b3_kernel: rows of a group of 8 as 0,4,1,5,2,6,3,7) ------------
    const int t4 = tid / CPR;
    const int srow = (t4 & ~7) | ((t4 & 1) << 2) | ((t4 >> 1) & 3), scol = (tid % CPR) * 8;
    int a_iy0[AR], a_ix0[AR];
    unsigned a_base[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + RP * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_iy0[i] = ok ? oy * p.sh - ph : -(1 << 28);
        a_ix0[i] = ox * p.sw - pw;
        a_base[i] = (unsigned)(((n * p.H * p.W + a_iy0[i] * p.W + a_ix0[i]) * p.xCs) * 4);
    }
    const int KT_all = p.K_pad / BK;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per_split : 0;
    const int kt_end = p.ksplit > 1 ? min(KT_all, kt_begin + p.kt_per_split) : KT_all;
    const int nk = kt_end - kt_begin;

    f32x4 ralo[1][AR], rahi[1][AR];
    if constexpr ((ABL & 4) != 0) for (int i = 0; i < AR; ++i) { ralo[0][i] = f32x4{1.f, 2.f, 3.f, (float)lane}; rahi[0][i] = ralo[0][i]; }
    const int4* ktab = p.ktab + (p.deconv2x ? blockIdx.y * (p.K_pad / 4 + 48) : 0);
    int4 tk_next = FAST ? ktab[(kt_begin * BK) / 4] : ktab[(kt_begin * BK + scol) / 4];
    int4 tk2_next = ktab[(kt_begin * BK + scol) / 4 + 1];      // general case only
    auto load_a = [&](int set, int k0) {
        if constexpr ((ABL & 4) != 0) return;
        if constexpr (FAST) {
            int4 tk = tk_next;
            tk.z += scol * 4;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int iy = a_iy0[i] + tk.x, ix = a_ix0[i] + tk.y;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned off = ok ? a_base[i] + (unsigned)tk.z : OOB;
                ralo[set][i] = buf_load4(xr, off);
                rahi[set][i] = buf_load4(xr, ok ? off + 16u : OOB);
            }
            __builtin_amdgcn_sched_barrier(0);
            tk_next = ktab[(k0 + BK) / 4];
        } else {
            const int4 tk = tk_next, tk2 = tk2_next;
            tk_next = ktab[(k0 + BK + scol) / 4];
            tk2_next = ktab[(k0 + BK + scol) / 4 + 1];
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int iy = a_iy0[i] + tk.x, ix = a_ix0[i] + tk.y;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const int iy2 = a_iy0[i] + tk2.x, ix2 = a_ix0[i] + tk2.y;
                const bool ok2 = (unsigned)iy2 < (unsigned)p.H && (unsigned)ix2 < (unsigned)p.W;
                ralo[set][i] = buf_load4(xr, ok ? a_base[i] + (unsigned)tk.z : OOB);
                rahi[set][i] = buf_load4(xr, ok2 ? a_base[i] + (unsigned)tk2.z : OOB);
            }
        }
    };
    auto store_a = [&](int stage, int set) {
        if constexpr ((ABL & 8) != 0) return;
        unsigned short* a = smem_r + stage * STAGE;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            if constexpr (NPL == 1) {
                f16x8r h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = (_Float16)ralo[set][i][e]; h[4 + e] = (_Float16)rahi[set][i][e]; }
                *reinterpret_cast<i32x4*>(a + (srow + RP * i) * LDK + scol) = __builtin_bit_cast(i32x4, h);
            } else if constexpr (NPL == 2) {
                f16x8r h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < 4 ? ralo[set][i][e] : rahi[set][i][e - 4];
                    h[e] = (_Float16)(v * xs);
                    l[e] = (_Float16)__builtin_fmaf(v, xs, -(float)h[e]);      // exact residual (one fma), then rounded to half
                }
                *reinterpret_cast<i32x4*>(a + (srow + RP * i) * LDK + scol) = __builtin_bit_cast(i32x4, h);
                *reinterpret_cast<i32x4*>(a + (BM + srow + RP * i) * LDK + scol) = __builtin_bit_cast(i32x4, l);
            } else {
                i32x4 q[3];
                unsigned x0, x1, x2;
                split3_pair_r(ralo[set][i][0], ralo[set][i][1], x0, x1, x2); q[0][0] = (int)x0; q[1][0] = (int)x1; q[2][0] = (int)x2;
                split3_pair_r(ralo[set][i][2], ralo[set][i][3], x0, x1, x2); q[0][1] = (int)x0; q[1][1] = (int)x1; q[2][1] = (int)x2;
                split3_pair_r(rahi[set][i][0], rahi[set][i][1], x0, x1, x2); q[0][2] = (int)x0; q[1][2] = (int)x1; q[2][2] = (int)x2;
                split3_pair_r(rahi[set][i][2], rahi[set][i][3], x0, x1, x2); q[0][3] = (int)x0; q[1][3] = (int)x1; q[2][3] = (int)x2;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<i32x4*>(a + (pl * BM + srow + RP * i) * LDK + scol) = q[pl];
            }
        }
    };

    // ---- weight fragments: global -> VGPR in MFMA order ----------------------------------------------------------------
    // plane layout [K step][half step][row][16 bf16]: lane (frow, half) of N-subtile j reads 16 bytes at
    //   ((2 * ks + kb) * rowsB + n) * 32 + half * 16,   n = n0 + (wn * NI + j) * 32 + frow
    const int frow = lane & 31, half = lane >> 5;
    const __amdgpu_buffer_rsrc_t wall = make_rsrc(wbase, (unsigned)(2 * (NPL - 1) * wplane) + (unsigned)((size_t)rowsB * p.K_pad * 2) + (unsigned)(rowsB * 128));
    unsigned b_voff[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) b_voff[j] = (unsigned)((n0 + (wn * NI + j) * 32 + frow) * 32 + half * 16);
    const unsigned hstep = (unsigned)rowsB * 32u;      // bytes of one half step of one plane
    const unsigned plane_b = (unsigned)(2 * wplane);
    i32x4 fbr[2][NI][NPL];      // the two half steps of a K step; each is requested half a step of matrix work before its use
    if constexpr ((ABL & 2) != 0) for (int b_ = 0; b_ < 2; ++b_) for (int j = 0; j < NI; ++j) for (int pl = 0; pl < NPL; ++pl) fbr[b_][j][pl] = i32x4{lane, 1, 2, 3};
    auto load_b = [&](int hs /* global half-step index */, int buf) {
        if constexpr ((ABL & 2) != 0) return;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                fbr[buf][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(wall, b_voff[j], (unsigned)pl * plane_b + (unsigned)hs * hstep, 0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fk = half * 8;
    i32x4 fa[1][MI][NPL];
    auto read_fa = [&](int stage, int kb, int set) {
        const unsigned short* a = smem_r + stage * STAGE + (wm * MI * 32 + frow) * LDK + fk + kb * 16;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[set][i][pl] = *reinterpret_cast<const i32x4*>(a + (pl * BM + i * 32) * LDK);
    };
    auto mma = [&](int set, int buf) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                f32x16 c = acc[i][j];
                if constexpr (NPL == 1) {
                    if constexpr ((ABL & 1) != 0) { asm volatile("" :: "v"(fa[set][i][0]), "v"(fbr[buf][j][0])); continue; }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8r, fa[set][i][0]), __builtin_bit_cast(f16x8r, fbr[buf][j][0]), c, 0, 0, 0);
                } else if constexpr (NPL == 2) {
                    const f16x8r a0 = __builtin_bit_cast(f16x8r, fa[set][i][0]), a1 = __builtin_bit_cast(f16x8r, fa[set][i][1]);
                    const f16x8r b0 = __builtin_bit_cast(f16x8r, fbr[buf][j][0]), b1 = __builtin_bit_cast(f16x8r, fbr[buf][j][1]);
                    if constexpr ((ABL & 1) != 0) { asm volatile("" :: "v"(a0), "v"(a1), "v"(b0), "v"(b1)); continue; }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);      // the two cross terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
                } else {
                    const bf16x8r a0 = __builtin_bit_cast(bf16x8r, fa[set][i][0]), a1 = __builtin_bit_cast(bf16x8r, fa[set][i][1]),
                                  a2 = __builtin_bit_cast(bf16x8r, fa[set][i][2]);
                    const bf16x8r b0 = __builtin_bit_cast(bf16x8r, fbr[buf][j][0]), b1 = __builtin_bit_cast(bf16x8r, fbr[buf][j][1]),
                                  b2 = __builtin_bit_cast(bf16x8r, fbr[buf][j][2]);
                    if constexpr ((ABL & 1) != 0) {      // timing ablation: the operands are consumed, nothing is multiplied
                        asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2));
                        continue;
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);      // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
                }
            }
    };

    // ---- pipeline -------------------------------------------------------------------------------------------------------
    // One K step:  request the weight fragments of the second half | multiply the first half | split + stage pixel tile k+1
    // into the other LDS buffer (its loads were issued a step ago) | request pixel tile k+2 and the first-half fragments of
    // tile k+1 | multiply the second half | barrier (tile k+1 visible, tile k no longer read).  The body has no branches:
    // requests past the end fall on the slack entries of the tap table (out of range: zeros) / the slack rows of the planes.
    // Measured and NOT adopted (profiles/r03_b3r_microbench.md): the pixel loads behind the weight loads (-4 %); weight
    // fragments four half steps and pixel tiles two K steps ahead (no gain, spills at 128 registers); the barrier moved to
    // the middle of the step with every LDS fragment read issued half a step early (no gain, spills).
    // Repeated at the end of round 3 WITH the requests fenced in place (four weight buffers, every half step requested a whole K
    // step ahead, loop unrolled by two for the buffer parity): +-1 % on geometries 76 and 80 (profiles/r03_wino_b3/b3r_d*.log) --
    // request latency is not what binds this kernel.
    load_a(0, kt_begin * BK);
    load_b(2 * kt_begin, 0);
    store_a(0, 0);
    __syncthreads();
    load_a(0, (kt_begin + 1) * BK);
    // The scheduler sinks the weight / pixel requests to their first use unless fenced.  Keeping them where the pipeline wants
    // them pays on the 2x4 grid of 128x128 and on the 1x8 grid (1-3 %, up to 10 % on the 64-column layers) and costs 2-6 % on the
    // others (profiles/r03_wino_b3/b3r_fence.log vs b3r_nofence.log; single fences: b3r_p*.log): decided per geometry.
    // bit 0: behind the first weight request, bit 1: behind the stage of the next pixel tile, bit 2: behind the mid-step requests
    constexpr int FENCES = ((BM == 128 && BN == 128 && WGM == 2 && WGN == 4) || WGN == 8) ? 7      // 76, 80: all three
                           : (BN == 256 && WGM == 2) ? 1                                             // 79: the first only (-1 %)
                           : (BN == 128 && WGM == 1) ? 4 : 0;                                        // 81: the last only (-0.5 %)
#define B3R_FENCE_N(n) do { if constexpr ((FENCES & (n)) != 0) __builtin_amdgcn_sched_barrier(0); } while (0)
    for (int k = 0; k < nk; ++k) {
        const int cur = k & 1, ks = kt_begin + k;
        load_b(2 * ks + 1, 1);
        B3R_FENCE_N(1);
        read_fa(cur, 0, 0);
        mma(0, 0);
        store_a(cur ^ 1, 0);
        B3R_FENCE_N(2);
        load_a(0, (ks + 2) * BK);
        load_b(2 * ks + 2, 0);
        B3R_FENCE_N(4);
        read_fa(cur, 1, 0);
        mma(0, 1);
        if constexpr ((ABL & 16) == 0) __syncthreads();
    }
#undef B3R_FENCE_N
    if constexpr ((ABL & 32) != 0) {      // timing ablation: no epilogue (the accumulators are consumed, one dummy store that never happens)
        f32x16 t = acc[0][0];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) t += acc[i][j];
        if (p.slope == 12345.f)
            for (int e = 0; e < 16; ++e) p.y[threadIdx.x * 16 + e] = t[e];
        return;
    }
    conv_epilogue<MI, NI, WGN>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo, rs.inv);
}

template <int BM, int BN, int WGM, int WGN, int ABL = 0, int NPL = 3>
static hipError_t launch_b3r(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.Cout_store + BN - 1) / BN;
    constexpr size_t lds = (size_t)2 * NPL * BM * 40 * sizeof(unsigned short);
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_b3r_kernel<BM, BN, WGM, WGN, true, ABL, NPL>), lds); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_b3r_kernel<BM, BN, WGM, WGN, false, ABL, NPL>), lds); e != hipSuccess) return e;
    const int rowsB = (int)(p.w_bytes / ((unsigned)p.K_pad * 2u));
    dim3 grid(p.MT * p.NT, p.deconv2x ? 4 : 1, p.ksplit > 1 ? p.ksplit : 1);
    if (p.Cin % 32 == 0) hipLaunchKernelGGL((conv_b3r_kernel<BM, BN, WGM, WGN, true, ABL, NPL>), grid, dim3(64 * WGM * WGN), lds, st, p, p.w_plane, rowsB);
    else hipLaunchKernelGGL((conv_b3r_kernel<BM, BN, WGM, WGN, false, ABL, NPL>), grid, dim3(64 * WGM * WGN), lds, st, p, p.w_plane, rowsB);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit <= 1) return e;
    return launch_splitk_reduce(p, (int)grid.y, st);
}

// p.w = the fragment-ordered planes (ConvParams::wb3r), p.w_bytes = bytes of one plane of one class; p.f16 == 1: the one-plane fp16 form
hipError_t launch_conv_b3r(const ConvParams& p, int tile, hipStream_t st)
{
    if (p.f16 == 1) {
        switch (tile) {
            case CONV_TILE_B3R: return launch_b3r<128, 128, 2, 4, 0, 1>(p, st);
            case CONV_TILE_B3R + 1: return launch_b3r<128, 64, 2, 2, 0, 1>(p, st);
            case CONV_TILE_B3R + 3: return launch_b3r<128, 256, 2, 4, 0, 1>(p, st);
            case CONV_TILE_B3R + 4: return launch_b3r<128, 256, 1, 8, 0, 1>(p, st);
            case CONV_TILE_B3R + 5: return launch_b3r<128, 128, 1, 4, 0, 1>(p, st);
            default: return hipErrorInvalidValue;
        }
    }
    if (p.f16 == 3) {
        switch (tile) {
            case CONV_TILE_B3R: return launch_b3r<128, 128, 2, 4, 0, 2>(p, st);
            case CONV_TILE_B3R + 1: return launch_b3r<128, 64, 2, 2, 0, 2>(p, st);
            case CONV_TILE_B3R + 3: return launch_b3r<128, 256, 2, 4, 0, 2>(p, st);
            case CONV_TILE_B3R + 4: return launch_b3r<128, 256, 1, 8, 0, 2>(p, st);
            case CONV_TILE_B3R + 5: return launch_b3r<128, 128, 1, 4, 0, 2>(p, st);
#ifdef ACCEL_CONV_DIAG
            // timing-only ablations of geometry 76 in the fp16x2 form (WRONG results by design; diagnostics build only)
            case 90: return launch_b3r<128, 128, 2, 4, 1, 2>(p, st);      // no MFMAs
            case 91: return launch_b3r<128, 128, 2, 4, 2, 2>(p, st);      // no weight loads
            case 92: return launch_b3r<128, 128, 2, 4, 4, 2>(p, st);      // no pixel loads
            case 93: return launch_b3r<128, 128, 2, 4, 8, 2>(p, st);      // no split + LDS stores
            case 94: return launch_b3r<128, 128, 2, 4, 32, 2>(p, st);     // no epilogue
            case 95: return launch_b3r<128, 128, 2, 4, 33, 2>(p, st);     // no MFMAs, no epilogue: the loader side alone
            case 96: return launch_b3r<128, 128, 2, 4, 46, 2>(p, st);     // MFMAs + fragment reads + barrier only, no epilogue
#endif
            default: return hipErrorInvalidValue;
        }
    }
    switch (tile) {
        case CONV_TILE_B3R: return launch_b3r<128, 128, 2, 4>(p, st);
        case CONV_TILE_B3R + 1: return launch_b3r<128, 64, 2, 2>(p, st);       // 64-channel layers (res2, the ResNet-18 trunk's first stage)
        case CONV_TILE_B3R + 3: return launch_b3r<128, 256, 2, 4>(p, st);
        case CONV_TILE_B3R + 4: return launch_b3r<128, 256, 1, 8>(p, st);      // every wavefront owns 32 columns: no weight fragment is fetched twice
        case CONV_TILE_B3R + 5: return launch_b3r<128, 128, 1, 4>(p, st);
#ifdef ACCEL_CONV_DIAG
        // timing-only ablations of geometry 76 (WRONG results by design; diagnostics build only)
        case 90: return launch_b3r<128, 128, 2, 4, 1>(p, st);      // no MFMAs
        case 91: return launch_b3r<128, 128, 2, 4, 2>(p, st);      // no weight loads
        case 92: return launch_b3r<128, 128, 2, 4, 4>(p, st);      // no pixel loads
        case 93: return launch_b3r<128, 128, 2, 4, 8>(p, st);      // no split + LDS stores
        case 94: return launch_b3r<128, 128, 2, 4, 16>(p, st);     // no barrier
        case 95: return launch_b3r<128, 128, 2, 4, 14>(p, st);     // MFMAs + fragment reads + barrier only
        case 96: return launch_b3r<128, 128, 2, 4, 30>(p, st);     // MFMAs + fragment reads only
#endif
        default: return hipErrorInvalidValue;
    }
}
