// bf16x3 implicit GEMM, third generation ("b3d", launch geometries 82-87): the arithmetic of conv_b3r.hip (each fp32 operand split
// EXACTLY into three bf16 terms, six v_mfma_f32_32x32x16_bf16 products per multiply-add in the same order, fp32 accumulate: results
// are bit-identical to conv_b3r for the same split-K factor) with BOTH operands brought in by LDS-DMA and the pixel operand split
// AFTER its fragment read.
//
// What bounded conv_b3r (ablations of fc6 x 8 clips): the loader side -- fp32 pixel loads into registers, the VALU
// split, three ds_write_b128 per 8 values, weight fragments fetched from L2 by every M half of the block -- cost as much as the
// multiplication and overlapped with it only in part, because every wavefront alternates between the two roles.
//
// Here:
//   * the pixel tile stays fp32 in LDS and arrives by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write, zero
//     padding by out-of-range buffer offsets); a wavefront reads the 32 bytes of fp32 its MFMA fragment covers and splits them
//     in registers.  The wavefronts of a block are laid out along M first, so a pixel row is read and split by ONE wavefront
//     (WGN = 1) or two, and that split feeds 6 * NI matrix instructions;
//   * the weight planes (the fragment-ordered copy conv_b3r uses: [class][K step][half step][row][16], one contiguous kilobyte
//     per 32 rows and half step) arrive by LDS-DMA once per block instead of once per M half, and are read as fragments;
//   * a stage is ONE half step (16 of K): 4-deep ring, three stages in flight, one raw s_barrier per stage, waits counted
//     (`vmcnt(L)`), never zero inside the loop.  The fragments of pixel stage k+1 are read and split while the matrix
//     instructions of stage k run (the barrier of stage k also publishes stage k+1), so the vector ALU work sits beside the
//     matrix work of the SAME wavefront instead of in a phase of its own;
//   * LDS images are lane-linear (the DMA writes base + lane * 16); bank conflicts are removed on the SOURCE side: the 16-byte
//     slot s of pixel row r holds K quad s ^ ((r >> 2) & 3), the slot of weight row n holds half s ^ ((n >> 3) & 1) -- both
//     conflict-free for the 16-lane groups that serve a ds_read_b128 (MI355X_MICROARCH.md, LDS).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"
#include "conv_common.h"
#include "conv_epilogue.h"

typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8d __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_d;

__device__ __forceinline__ void split3_pair_d(float v0, float v1, unsigned& q0, unsigned& q1, unsigned& q2)
{
    const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
    q0 = __builtin_amdgcn_perm(u1, u0, 0x07060302);                       // {top16(v1), top16(v0)}
    const float r0 = v0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u), r1 = v1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    const unsigned s0 = __builtin_bit_cast(unsigned, r0), s1 = __builtin_bit_cast(unsigned, r1);
    q1 = __builtin_amdgcn_perm(s1, s0, 0x07060302);
    const float t0 = r0 - __builtin_bit_cast(float, s0 & 0xFFFF0000u), t1 = r1 - __builtin_bit_cast(float, s1 & 0xFFFF0000u);
    q2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0), 0x07060302);
}

// NPL = 3: bf16x3; NPL = 1: the fp16-MFMA mode (pixels rounded to half after the fragment read, one plane of half-rounded weights)
template <int BM, int BN, int WGM, int WGN, int NPL = 3>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv_b3d_kernel(ConvParams p, size_t wplane, int rowsB)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WGM * WGN, NS = 4;
    constexpr int MI = BM / (WGM * 32), NI = BN / (WGN * 32);
    constexpr int A_BYTES = BM * 64, B_BYTES = NPL * BN * 32, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM / 16, NB = NPL * BN / 32;      // DMA instructions (1 KB each) per stage
    static_assert(NA % NW == 0 && NB % NW == 0, "the pieces of a stage must split evenly over the wavefronts");
    constexpr int LA = NA / NW, LB = NB / NW, L = LA + LB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const __amdgpu_buffer_rsrc_t wall = make_rsrc(wbase, (unsigned)(2 * (NPL - 1) * wplane) + (unsigned)((size_t)rowsB * p.K_pad * 2) + (unsigned)(rowsB * 128));
    const int HoWo = p.Ho * p.Wo;

    // ---- DMA pieces of this wavefront ----------------------------------------------------------------------------------------
    // pixel piece q (rows 16q .. 16q+15): lane -> (row = lane >> 2, physical slot = lane & 3), fetches K quad slot ^ ((row >> 2) & 3)
    int a_iy0[LA], a_ix0[LA];
    unsigned a_off[LA];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int r = (wave * LA + j) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_iy0[j] = ok ? oy * p.sh - ph : -(1 << 28);
        a_ix0[j] = ox * p.sw - pw;
        a_off[j] = (unsigned)((((n * p.H + a_iy0[j]) * p.W + a_ix0[j]) * p.xCs + c * 4) * 4);
    }
    // weight piece q (plane q / (BN / 32), rows 32 (q % (BN / 32)) ..): lane -> (row = lane >> 1, slot = lane & 1), fetches half slot ^ ((row >> 3) & 1)
    unsigned b_off[LB];
    int b_lds[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int q = wave * LB + j, pl = q / (BN / 32), rb = q % (BN / 32);
        const int n = rb * 32 + (lane >> 1);
        const int h = (lane & 1) ^ ((n >> 3) & 1);
        b_off[j] = (unsigned)pl * (unsigned)(2 * wplane) + (unsigned)((n0 + n) * 32 + h * 16);
        b_lds[j] = A_BYTES + pl * BN * 32 + rb * 1024;
    }
    const unsigned hstep = (unsigned)rowsB * 32u;      // bytes of one half step of one plane

    const int KT_all = p.K_pad / 32;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per_split : 0;
    const int kt_end = p.ksplit > 1 ? min(KT_all, kt_begin + p.kt_per_split) : KT_all;
    const int s_begin = 2 * kt_begin, ns = 2 * (kt_end - kt_begin);      // stages = half steps of 16
    // the tap table through the CONSTANT address space: a scalar load (s_load_dwordx4) whatever the memory clobbers of the counted
    // waits below make the compiler assume -- as a vector load it would sit in the in-order vmcnt queue behind the DMAs and its
    // use would drain them
    typedef const int4 __attribute__((address_space(4)))* ktab_ptr;
    const ktab_ptr ktab = (ktab_ptr)(unsigned long long)(p.ktab + (p.deconv2x ? blockIdx.y * (p.K_pad / 4 + 48) : 0));

    int4 tk_next = ktab[s_begin * 4];        // {dy, dx, byte offset, 0} of k = 16 s (Cin % 16 == 0: the 16 values share a tap)
    auto issue = [&](int s /* global half-step index */, int slot) {
        const int4 tk = tk_next;
        tk_next = ktab[(s + 1) * 4];         // requested a stage ahead of its use
        unsigned char* st = smem_d + slot * STAGE;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int iy = a_iy0[j] + tk.x, ix = a_ix0[j] + tk.y;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_d)(st + (wave * LA + j) * 1024), 16,
                                                     ok ? a_off[j] + (unsigned)tk.z : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < LB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wall, (lds_ptr_d)(st + b_lds[j]), 16, b_off[j], (unsigned)s * hstep, 0, 0);
    };

    // ---- fragment addressing ----------------------------------------------------------------------------------------------------
    const int frow = lane & 31, fh = lane >> 5;
    int a_lo[MI], a_hi[MI], b_ad[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = (wm * MI + i) * 32 + frow, x = (r >> 2) & 3;
        a_lo[i] = r * 64 + ((2 * fh) ^ x) * 16;
        a_hi[i] = r * 64 + ((2 * fh + 1) ^ x) * 16;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = (wn * NI + j) * 32 + frow;
        b_ad[j] = A_BYTES + n * 32 + (fh ^ ((n >> 3) & 1)) * 16;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    i32x4 fa[2][MI][NPL];
    auto read_split_a = [&](int slot, int set) {
        const unsigned char* st = smem_d + slot * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(st + a_lo[i]), hi = *reinterpret_cast<const f32x4*>(st + a_hi[i]);
            if constexpr (NPL == 1) {
                f16x8d h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = (_Float16)lo[e]; h[4 + e] = (_Float16)hi[e]; }
                fa[set][i][0] = __builtin_bit_cast(i32x4, h);
            } else {
                unsigned x0, x1, x2;
                split3_pair_d(lo[0], lo[1], x0, x1, x2); fa[set][i][0][0] = (int)x0; fa[set][i][1][0] = (int)x1; fa[set][i][2][0] = (int)x2;
                split3_pair_d(lo[2], lo[3], x0, x1, x2); fa[set][i][0][1] = (int)x0; fa[set][i][1][1] = (int)x1; fa[set][i][2][1] = (int)x2;
                split3_pair_d(hi[0], hi[1], x0, x1, x2); fa[set][i][0][2] = (int)x0; fa[set][i][1][2] = (int)x1; fa[set][i][2][2] = (int)x2;
                split3_pair_d(hi[2], hi[3], x0, x1, x2); fa[set][i][0][3] = (int)x0; fa[set][i][1][3] = (int)x1; fa[set][i][2][3] = (int)x2;
            }
        }
    };
    auto mma = [&](int slot, int set) {
        const unsigned char* st = smem_d + slot * STAGE;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            i32x4 fb[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[pl] = *reinterpret_cast<const i32x4*>(st + b_ad[j] + pl * BN * 32);
            if constexpr (NPL == 1) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8d, fa[set][i][0]), __builtin_bit_cast(f16x8d, fb[0]), acc[i][j], 0, 0, 0);
            } else {
                const bf16x8d b0 = __builtin_bit_cast(bf16x8d, fb[0]), b1 = __builtin_bit_cast(bf16x8d, fb[1]), b2 = __builtin_bit_cast(bf16x8d, fb[2]);
                // the six terms, smallest first (the order of conv_b3r), the MI accumulators interleaved so that consecutive matrix
                // instructions do not wait for each other's result
#define B3D_TERM(ap, bp)                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8d, fa[set][i][ap]), bp, acc[i][j], 0, 0, 0);
                B3D_TERM(1, b1) B3D_TERM(0, b2) B3D_TERM(2, b0) B3D_TERM(0, b1) B3D_TERM(1, b0) B3D_TERM(0, b0)
#undef B3D_TERM
            }
        }
    };

    // ---- pipeline -----------------------------------------------------------------------------------------------------------------
    // stages s_begin .. s_begin + ns - 1; stage t lives in ring slot t % 4.  Requests past the end are issued as well (branch-free
    // body, exact counts): they fall on the slack entries of the tap table (zeros) and the slack rows of the weight planes.
    issue(s_begin, 0);
    issue(s_begin + 1, 1);
    issue(s_begin + 2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
    __builtin_amdgcn_s_barrier();
    read_split_a(0, 0);
    for (int k = 0; k < ns; k += 2) {
        // even stage: fragments in set 0
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");        // my pieces of stage k+1 have landed (k+2 may be in flight)
        __builtin_amdgcn_s_barrier();                                    // stage k+1 complete for everyone; slot (k+3) % 4 no longer read
        issue(s_begin + k + 3, (k + 3) & 3);
        read_split_a((k + 1) & 3, 1);
        mma(k & 3, 0);
        // odd stage: fragments in set 1
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(s_begin + k + 4, (k + 4) & 3);
        read_split_a((k + 2) & 3, 0);
        mma((k + 1) & 3, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA may land after the block has given its LDS back
    conv_epilogue<MI, NI, WGN, (MI * NI > 4)>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
#endif
}

template <int BM, int BN, int WGM, int WGN, int NPL = 3>
static hipError_t launch_b3d(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.Cout_store + BN - 1) / BN;
    constexpr size_t lds = (size_t)4 * (BM * 64 + NPL * BN * 32);
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_b3d_kernel<BM, BN, WGM, WGN, NPL>), lds); e != hipSuccess) return e;
    const int rowsB = (int)(p.w_bytes / ((unsigned)p.K_pad * 2u));
    dim3 grid(p.MT * p.NT, p.deconv2x ? 4 : 1, p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((conv_b3d_kernel<BM, BN, WGM, WGN, NPL>), grid, dim3(64 * WGM * WGN), lds, st, p, p.w_plane, rowsB);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit <= 1) return e;
    return launch_splitk_reduce(p, (int)grid.y, st);
}

bool conv_b3d_eligible(const ConvParams& p) { return p.Cin % 16 == 0 && p.ktab != nullptr; }

// p.w = the fragment-ordered planes (ConvParams::wb3r), p.w_bytes = bytes of one plane of one class; p.f16 == 1: the one-plane fp16 form
hipError_t launch_conv_b3d(const ConvParams& p, int tile, hipStream_t st)
{
    if (!conv_b3d_eligible(p)) return hipErrorInvalidValue;
    if (p.f16 == 1) {
        switch (tile) {
            case CONV_TILE_B3D: return launch_b3d<256, 256, 4, 2, 1>(p, st);
            case CONV_TILE_B3D + 1: return launch_b3d<128, 256, 4, 2, 1>(p, st);
            case CONV_TILE_B3D + 2: return launch_b3d<128, 128, 4, 1, 1>(p, st);
            case CONV_TILE_B3D + 3: return launch_b3d<128, 128, 2, 2, 1>(p, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (tile) {
        case CONV_TILE_B3D: return launch_b3d<256, 256, 4, 2>(p, st);          // 8 wavefronts, each 64 x 128: 160 KB of LDS, one block per CU
        case CONV_TILE_B3D + 1: return launch_b3d<128, 256, 4, 2>(p, st);      // 8 wavefronts, each 32 x 128
        case CONV_TILE_B3D + 2: return launch_b3d<128, 128, 4, 1>(p, st);      // 4 wavefronts, each 32 x 128: two blocks per CU
        case CONV_TILE_B3D + 3: return launch_b3d<128, 128, 2, 2>(p, st);      // 4 wavefronts, each 64 x 64
        case CONV_TILE_B3D + 4: return launch_b3d<256, 256, 8, 1>(p, st);      // 8 wavefronts, each 32 x 256: every pixel row split once
        case CONV_TILE_B3D + 5: return launch_b3d<128, 256, 2, 4>(p, st);      // 8 wavefronts, each 64 x 64
        default: return hipErrorInvalidValue;
    }
}
