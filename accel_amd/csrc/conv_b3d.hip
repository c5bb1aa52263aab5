// fp16-MFMA implicit GEMM with BOTH operands brought in by LDS-DMA ("b3d", launch geometries 82-85, 88, 89): the kernel of the f16 mode
// (plan option dtype=f16, BASELINE config 5) and the only one that reads / writes HALF activation views.
// (Rounds 4 carried the bf16x3 and fp16x2 forms of the fp32 layers on this staging as well: measured a tie / slower than conv_b3r.hip on
// every layer of the step -- profiles/r04_b3d_microbench.log, r04_b3d_h2_microbench.log -- and removed in round 5.)
//
//   * the pixel tile arrives by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write, zero padding by out-of-range buffer
//     offsets): fp32 pixels are read as the 32 bytes a fragment covers and rounded to half in registers; pixels STORED as half (XH) are
//     the MFMA fragment as the DMA brought it, no conversion at all.  The wavefronts of a block are laid out along M first;
//   * the weight plane (half-rounded, fragment order [class][K step][half step][row][16]: one contiguous kilobyte per 32 rows and half
//     step) arrives by LDS-DMA once per block, and is read as fragments;
//   * a ring stage holds KSUB half steps (16 of K each), NS-deep ring, one raw s_barrier per stage, waits counted (`vmcnt(L)`), never
//     zero inside the loop; the fragments of the next half step are read while the matrix instructions of the current one run;
//   * LDS images are lane-linear (the DMA writes base + lane * 16); bank conflicts are removed on the SOURCE side: the 16-byte
//     slot s of pixel row r holds K quad s ^ ((r >> 2) & 3), the slot of weight row n holds half s ^ ((n >> 3) & 1) -- both
//     conflict-free for the 16-lane groups that serve a ds_read_b128 (MI355X_MICROARCH.md, LDS).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "conv_common.h"
#include "conv_epilogue.h"

typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8d __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_d;

// NPL = 1 (the only form left): pixels rounded to half after the fragment read, one plane of half-rounded weights;
// XH: the pixel operand is STORED as half (ConvParams::x_half): rows of 32 bytes per stage, the fragment is what the
// DMA brought, no conversion at all; the epilogue of the NPL = 1 forms may read a half residual and write half (conv_epilogue_h)
// KSUB: half steps (16 of K) per ring stage.  A half step is MI * NI matrix instructions per wavefront: with one half step per barrier the fp16 form spent its time at barriers (0.13-0.18 of the fp16 peak on the
// deep-K layers of config 5), so its stages hold 2 or 4 half steps; NS = ring depth (3: two stages in flight, 4: three)
template <int BM, int BN, int WGM, int WGN, int NPL = 1, bool XH = false, int KSUB = 1, int NS = 4>
__global__ __launch_bounds__(64 * WGM * WGN, (NS * KSUB * (BM * (XH ? 32 : 64) + NPL * BN * 32) <= 81920 ? 2 : 1))      // two blocks per CU where the LDS allows it
void conv_b3d_kernel(ConvParams p, size_t wplane, int rowsB)
{
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(NPL == 1, "the fp16 form is the only one");
    static_assert(NS == 3 || NS == 4, "ring depth");
    constexpr int NW = WGM * WGN;
    constexpr int MI = BM / (WGM * 32), NI = BN / (WGN * 32);
    constexpr int A_ROW = XH ? 32 : 64;                   // bytes of a pixel row per half step (16 of K)
    constexpr int A_BYTES = BM * A_ROW, B_BYTES = NPL * BN * 32, SUB = A_BYTES + B_BYTES, STAGE = KSUB * SUB;
    constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024, NP1 = NA + NB, NP = KSUB * NP1;      // DMA pieces (1 KB = one wave instruction) per stage
    static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "tiles are whole kilobytes");
    // piece q of a HALF STEP belongs to wavefront q % NW, the same for every half step of a stage (so the per-lane source offsets and
    // the piece's kind do not depend on the half step): a wavefront issues KSUB * L1 or KSUB * (L1 - 1) pieces per stage
    constexpr int L1 = (NP1 + NW - 1) / NW, LMAX = KSUB * L1;
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
    constexpr int XB = XH ? 2 : 4;      // bytes per stored pixel value

    // ---- DMA pieces of this wavefront ----------------------------------------------------------------------------------------
    // fp32 pixel piece q (rows 16q ..): lane -> (row = lane >> 2, physical slot = lane & 3), fetches K quad slot ^ ((row >> 2) & 3)
    // half pixel piece q (rows 32q ..) and weight piece (plane q / (BN / 32), rows 32 (q % (BN / 32)) ..): lane -> (row = lane >> 1,
    // slot = lane & 1), fetches the 16-byte half slot ^ ((row >> 3) & 1)
    int pc_iy0[L1], pc_ix0[L1];
    unsigned pc_off[L1];
#pragma unroll
    for (int t = 0; t < L1; ++t) {
        const int q = wave + t * NW;        // wave-uniform; pieces q >= NP1 do not exist (this wavefront then issues L1 - 1 per half step)
        pc_iy0[t] = pc_ix0[t] = 0;
        pc_off[t] = 0;
        if (q < NA) {
            const int r = XH ? q * 32 + (lane >> 1) : q * 16 + (lane >> 2);
            const int cbyte = XH ? (((lane & 1) ^ ((r >> 3) & 1)) * 16) : (((lane & 3) ^ ((r >> 2) & 3)) * 16);
            const int m = m0 + r;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int n = mm / HoWo, rem = mm - n * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            pc_iy0[t] = ok ? oy * p.sh - ph : -(1 << 28);
            pc_ix0[t] = ox * p.sw - pw;
            pc_off[t] = (unsigned)((((n * p.H + pc_iy0[t]) * p.W + pc_ix0[t]) * p.xCs) * XB + cbyte);
        } else if (q < NP1) {
            const int qb = q - NA, pl = qb / (BN / 32), rb = qb % (BN / 32);
            const int n = rb * 32 + (lane >> 1);
            const int h = (lane & 1) ^ ((n >> 3) & 1);
            pc_off[t] = (unsigned)pl * (unsigned)(2 * wplane) + (unsigned)((n0 + n) * 32 + h * 16);
        }
    }
    const bool full = (NP1 % NW == 0) || wave < NP1 % NW;      // this wavefront issues L1 pieces per half step (else L1 - 1)
    const unsigned hstep = (unsigned)rowsB * 32u;      // bytes of one half step of one plane

    const int KT_all = p.K_pad / 32;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per_split : 0;
    const int kt_end = p.ksplit > 1 ? min(KT_all, kt_begin + p.kt_per_split) : KT_all;
    // stages of KSUB half steps; half steps past the end of the K range (the last stage of a range whose length is not a multiple of
    // KSUB) multiply zeros: their tap entries are the table's slack (out of range).  A split-K range must therefore END on a stage
    // boundary unless it is the last one (conv_plan_split rounds kt_per_split accordingly)
    const int s_begin = 2 * kt_begin / KSUB, ns = (2 * (kt_end - kt_begin) + KSUB - 1) / KSUB;
    // the tap table through the CONSTANT address space: a scalar load (s_load_dwordx4) whatever the memory clobbers of the counted
    // waits below make the compiler assume -- as a vector load it would sit in the in-order vmcnt queue behind the DMAs and its
    // use would drain them
    typedef const int4 __attribute__((address_space(4)))* ktab_ptr;
    const ktab_ptr ktab = (ktab_ptr)(unsigned long long)(p.ktab + (p.deconv2x ? blockIdx.y * (p.K_pad / 4 + 48) : 0));

    int4 tk_next[KSUB];                      // {dy, dx, byte offset, 0} of k = 16 h for the half steps h of a stage (Cin % 16 == 0: the 16 values share a tap)
#pragma unroll
    for (int u = 0; u < KSUB; ++u) tk_next[u] = ktab[(s_begin * KSUB + u) * 4];
    auto issue = [&](int s /* global stage index */, int slot) {
        unsigned char* st = smem_d + slot * STAGE;
#pragma unroll
        for (int u = 0; u < KSUB; ++u) {
            const int4 tk = tk_next[u];
            tk_next[u] = ktab[((s + 1) * KSUB + u) * 4];      // requested a stage ahead of its use
            const unsigned wsoff = (unsigned)(s * KSUB + u) * hstep;
#pragma unroll
            for (int t = 0; t < L1; ++t) {
                const int q = wave + t * NW;      // wave-uniform
                if (q >= NP1) continue;
                unsigned char* dst = st + u * SUB + q * 1024;      // per half step: pixel pieces first, then the weight planes (plane pl, row block rb at A_BYTES + (pl * (BN / 32) + rb) KB)
                if (q < NA) {
                    const int iy = pc_iy0[t] + tk.x, ix = pc_ix0[t] + tk.y;
                    const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_d)dst, 16, ok ? pc_off[t] + (unsigned)tk.z : OOB, 0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wall, (lds_ptr_d)dst, 16, pc_off[t], wsoff, 0, 0);
                }
            }
        }
    };
    // my pieces of the older stages have landed when at most `stages` newer stages of mine are outstanding
    auto wait_stages = [&](auto stages_c) {
        constexpr int n = decltype(stages_c)::value;
        static_assert(n * LMAX < 64, "vmcnt is a 6-bit counter");
        if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n * LMAX) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n * KSUB * (L1 - 1)) : "memory");
    };

    // ---- fragment addressing ----------------------------------------------------------------------------------------------------
    const int frow = lane & 31, fh = lane >> 5;
    int a_lo[MI], a_hi[MI], b_ad[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = (wm * MI + i) * 32 + frow;
        if constexpr (XH) {
            a_lo[i] = r * 32 + (fh ^ ((r >> 3) & 1)) * 16;
            a_hi[i] = 0;
        } else {
            const int x = (r >> 2) & 3;
            a_lo[i] = r * 64 + ((2 * fh) ^ x) * 16;
            a_hi[i] = r * 64 + ((2 * fh + 1) ^ x) * 16;
        }
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
    auto read_split_a = [&](int slot, int sub, int set) {
        const unsigned char* st = smem_d + slot * STAGE + sub * SUB;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if constexpr (XH) {
                fa[set][i][0] = *reinterpret_cast<const i32x4*>(st + a_lo[i]);
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(st + a_lo[i]), hi = *reinterpret_cast<const f32x4*>(st + a_hi[i]);
                f16x8d h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = (_Float16)lo[e]; h[4 + e] = (_Float16)hi[e]; }
                fa[set][i][0] = __builtin_bit_cast(i32x4, h);
            }
        }
    };
    auto mma = [&](int slot, int sub, int set) {
        const unsigned char* st = smem_d + slot * STAGE + sub * SUB;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            i32x4 fb[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[pl] = *reinterpret_cast<const i32x4*>(st + b_ad[j] + pl * BN * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8d, fa[set][i][0]), __builtin_bit_cast(f16x8d, fb[0]), acc[i][j], 0, 0, 0);
        }
    };

    // ---- pipeline -----------------------------------------------------------------------------------------------------------------
    // stages s_begin .. s_begin + ns - 1; stage t lives in ring slot t % NS, NS - 1 stages are requested ahead.  Requests past the end
    // are issued as well (branch-free body, exact counts): they fall on the slack entries of the tap table (zeros) and the slack
    // rows of the weight planes.  The fragments of the NEXT half step (the first one of stage k+1 at the end of stage k: the barrier
    // of stage k publishes stage k+1 as well) are read and split / converted while the matrix instructions of the current one run.
    issue(s_begin, 0);
    issue(s_begin + 1, 1);
    if constexpr (NS == 4) issue(s_begin + 2, 2);
    wait_stages(std::integral_constant<int, NS - 2>());
    __builtin_amdgcn_s_barrier();
    read_split_a(0, 0, 0);
    constexpr int UNR = (KSUB & 1) ? 2 : 1;      // stages per loop body: the fragment set alternates per half step
    for (int k = 0; k < ns; k += UNR) {
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int kk = k + r;
            if (UNR == 2 && r == 1 && kk >= ns) break;
            wait_stages(std::integral_constant<int, NS - 3>());              // my pieces of stage kk+1 have landed (NS = 4: kk+2 may be in flight)
            __builtin_amdgcn_s_barrier();                                    // stage kk+1 complete for everyone; slot (kk-1) % NS no longer read
            issue(s_begin + kk + NS - 1, (kk + NS - 1) % NS);
#pragma unroll
            for (int u = 0; u < KSUB; ++u) {
                const int set = (r * KSUB + u) & 1;
                if (u + 1 < KSUB) read_split_a(kk % NS, u + 1, set ^ 1);
                else read_split_a((kk + 1) % NS, 0, set ^ 1);
                mma(kk % NS, u, set);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA may land after the block has given its LDS back
    conv_epilogue_h<MI, NI, WGN>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
#endif
}

template <int BM, int BN, int WGM, int WGN, int NPL = 1, bool XH = false, int KSUB = 1, int NS = 4>
static hipError_t launch_b3d(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.Cout_store + BN - 1) / BN;
    constexpr size_t lds = (size_t)NS * KSUB * (BM * (XH ? 32 : 64) + NPL * BN * 32);
    static_assert(lds <= 163840, "160 KB of LDS per CU");
    if (p.ksplit > 1 && (2 * p.kt_per_split) % KSUB) return hipErrorInvalidValue;      // a split-K range ends on a stage boundary (conv_plan_split)
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_b3d_kernel<BM, BN, WGM, WGN, NPL, XH, KSUB, NS>), lds); e != hipSuccess) return e;
    const int rowsB = (int)(p.w_bytes / ((unsigned)p.K_pad * 2u));
    dim3 grid(p.MT * p.NT, p.deconv2x ? 4 : 1, p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((conv_b3d_kernel<BM, BN, WGM, WGN, NPL, XH, KSUB, NS>), grid, dim3(64 * WGM * WGN), lds, st, p, p.w_plane, rowsB);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit <= 1) return e;
    return launch_splitk_reduce(p, (int)grid.y, st);
}

bool conv_b3d_eligible(const ConvParams& p) { return p.Cin % 16 == 0 && p.ktab != nullptr; }

// p.w = the fragment-ordered half plane (ConvParams::wb3r of an f16-mode layer), p.w_bytes = its bytes per class; p.x_half: the pixel
// operand stored as half
hipError_t launch_conv_b3d(const ConvParams& p, int tile, hipStream_t st)
{
    if (!conv_b3d_eligible(p)) return hipErrorInvalidValue;
    if (p.f16 == 1 && p.x_half) {
        switch (tile) {
            // (stage = KSUB half steps, ring depth NS; LDS = NS * KSUB * (32 BM + 32 BN) bytes)
            case CONV_TILE_B3D: return launch_b3d<256, 256, 4, 2, 1, true, 2, 4>(p, st);        // 128 KB, 16 matrix instructions per wavefront and barrier
            case CONV_TILE_B3D + 1: return launch_b3d<128, 256, 4, 2, 1, true, 4, 3>(p, st);    // 144 KB, 16
            case CONV_TILE_B3D + 2: return launch_b3d<128, 128, 4, 1, 1, true, 2, 3>(p, st);    // 48 KB: three blocks per CU, 8
            case CONV_TILE_B3D + 3: return launch_b3d<128, 128, 2, 2, 1, true, 2, 3>(p, st);    // 48 KB, 8
            case CONV_TILE_B3D + 6: return launch_b3d<128, 64, 4, 1, 1, true, 4, 3>(p, st);     // 72 KB: two blocks per CU, 8
            case CONV_TILE_B3D + 7: return launch_b3d<256, 128, 4, 2, 1, true, 4, 3>(p, st);    // 144 KB, 16
            default: return hipErrorInvalidValue;
        }
    }
    if (p.f16 == 1) {
        switch (tile) {
            // fp32 pixels (64 bytes per row and half step), rounded to half after the fragment read
            case CONV_TILE_B3D: return launch_b3d<256, 256, 4, 2, 1, false, 2, 3>(p, st);       // 144 KB
            case CONV_TILE_B3D + 1: return launch_b3d<128, 256, 4, 2, 1, false, 2, 4>(p, st);   // 128 KB
            case CONV_TILE_B3D + 2: return launch_b3d<128, 128, 4, 1, 1, false, 2, 3>(p, st);   // 72 KB: two blocks per CU
            case CONV_TILE_B3D + 3: return launch_b3d<128, 128, 2, 2, 1, false, 2, 3>(p, st);
            case CONV_TILE_B3D + 6: return launch_b3d<128, 64, 4, 1, 1, false, 2, 3>(p, st);    // 60 KB
            case CONV_TILE_B3D + 7: return launch_b3d<256, 128, 4, 2, 1, false, 2, 3>(p, st);   // 120 KB
            default: return hipErrorInvalidValue;
        }
    }
    return hipErrorInvalidValue;      // fp32 layers run on conv_b3r.hip (the bf16x3 / fp16x2 forms of this staging were removed in round 5)
}
