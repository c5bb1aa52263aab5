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
#include <cstdlib>
#include "kernels.h"
#include "conv_common.h"

// Diagnostic build only (-DACCEL_CONV_TIMELINE, scripts/microbench/timeline.py): per-block timestamps of the pipelined
// kernel (entry / first tile in LDS / K loop done / stores retired) on the 100 MHz constant clock.
#ifdef ACCEL_CONV_TIMELINE
__device__ unsigned long long g_conv_timeline[8 * 16384];
#define CONV_TL(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 16384) \
        g_conv_timeline[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int accel_debug_conv_timeline(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_timeline), sizeof(unsigned long long) * (size_t)n);
}
#else
#define CONV_TL(slot) do { } while (0)
#endif

#include "conv_epilogue.h"

template <int BM, int BN, int WGM, int WGN, int BK, int MID, int ABL = 0, int FAST = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_f32_kernel(ConvParams p)
{
    constexpr int LDK = BK + 4;
    constexpr int NTHR = 64 * WGM * WGN;
    constexpr int CPR = BK / 4;                     // float4 columns per staged row
    constexpr int RP = NTHR / CPR;                  // rows staged per pass
    constexpr int MI = BM / (WGM * 32), NI = BN / (WGN * 32);
    constexpr int AR = BM / RP, BR = BN / RP;       // rows each thread stages
    static_assert(BM % RP == 0 && BN % RP == 0, "tile / thread-count mismatch");
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
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int nt = swz % p.NT, mt = swz / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    int kw = p.kw, ph = p.ph, pw = p.pw;
    const float* wbase = p.w;
    int py = 0, px = 0;
    if (p.deconv2x) {
        py = blockIdx.y >> 1; px = blockIdx.y & 1;
        ph = 1 - py; pw = 1 - px;
        wbase += (size_t)blockIdx.y * p.w_class_stride;
    }
    const int ntaps = p.kh * kw;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);

    // ---- staging coordinates --------------------------------------------------
    const int srow = tid / CPR, scol = (tid % CPR) * 4;
    int a_iy0[AR], a_ix0[AR], a_nb[AR];
    unsigned a_off[AR];                     // FAST: byte offset of (row, tap 0, ci = scol), may wrap below 0 at the border
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + RP * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_iy0[i] = ok ? oy * p.sh - ph : -(1 << 28);   // invalid rows fail the bounds test below
        a_ix0[i] = ox * p.sw - pw;
        a_nb[i] = n * p.H * p.W;
        a_off[i] = (unsigned)(((a_nb[i] + a_iy0[i] * p.W + a_ix0[i]) * p.xCs + scol) * 4);
    }
    // split-K: blockIdx.z owns K steps [kt_begin, kt_end)
    const int KT_all = p.K_pad / BK;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per_split : 0;
    const int kt_end = p.ksplit > 1 ? min(KT_all, kt_begin + p.kt_per_split) : KT_all;

    const float* wrow0 = wbase + (size_t)(n0 + srow) * p.K_pad + scol;
    const size_t wrow_step = (size_t)RP * p.K_pad;
    // FAST path (Cin % BK == 0): a K step never straddles two taps, so (dy, dx, byte offset) of the
    // step is wave-uniform and comes from a host-built table through the scalar unit; weights are
    // fetched through a buffer resource with the K offset in an SGPR (no per-step vector address math).
    const __amdgpu_buffer_rsrc_t wr_ = make_rsrc(wbase, p.w_bytes);
    unsigned b_off[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) b_off[i] = (unsigned)(((size_t)(n0 + srow + RP * i) * p.K_pad + scol) * 4);

    f32x4 ra[AR], rb[BR];
    const int4* ktab = p.ktab + (p.deconv2x ? blockIdx.y * (p.K_pad / 4 + 48) : 0);   // one entry per 4-wide K granule
    // prefetched one K step ahead: FAST = wave-uniform entry through the scalar unit, otherwise the lane's own granule
    int4 tk_next = ktab[(kt_begin * BK + (FAST ? 0 : scol)) / 4];
    auto load_tiles = [&](int k0) {
        if (FAST) {
            const int4 tk = tk_next;                                         // {dy, dx, byte offset, 0}
            tk_next = ktab[(k0 + BK) / 4];
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int iy = a_iy0[i] + tk.x, ix = a_ix0[i] + tk.y;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                ra[i] = buf_load4(xr, ok ? a_off[i] + (unsigned)tk.z : OOB);
            }
#pragma unroll
            for (int i = 0; i < BR; ++i)
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr_, b_off[i], k0 * 4, 0));
            return;
        }
        // generic path (Cin not a multiple of BK: RGB stems, ragged concats): each lane looks its own
        // 4-wide granule up in the same table (vector load, prefetched one step ahead) -- no divisions
        const int4 tk = tk_next;
        tk_next = ktab[(k0 + BK + scol) / 4];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = a_iy0[i] + tk.x, ix = a_ix0[i] + tk.y;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            ra[i] = buf_load4(xr, ok ? a_off[i] - scol * 4u + (unsigned)tk.z : OOB);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            rb[i] = *reinterpret_cast<const f32x4*>(wrow0 + i * wrow_step + k0);
    };
    auto store_tiles = [&](int buf) {
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BN * LDK;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<f32x4*>(a + (srow + RP * i) * LDK + scol) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<f32x4*>(b + (srow + RP * i) * LDK + scol) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31, fk = (lane >> 5) * 4;
    if constexpr (MID == 3) {
        // Deep-prefetch schedule.  Two register tiles are in flight: tile k+1 (loaded two K steps ago) is written to the
        // other LDS buffer in the SECOND quarter of step k's MFMAs, so the ds_write latency and the barrier sit behind
        // eight more MFMAs instead of one; the registers just freed receive tile k+3, giving every global load ~1.75
        // K steps (> HBM latency) to land.  The barrier is an LDS-only one (`s_waitcnt lgkmcnt(0); s_barrier`):
        // __syncthreads() would drain the global loads that are deliberately left in flight.  Scheduling fences
        // between the four quarters keep the compiler from hoisting the store/barrier chain back up.
        constexpr int T = BK / 8;
        static_assert(T == 4, "the deep-prefetch schedule is written for BK = 32");
        f32x4 fa[2][MI], fb[2][NI];
        f32x4 ra1[AR], rb1[BR];
        auto load_into = [&](f32x4 (&A)[AR], f32x4 (&B)[BR], int k0) {
            const int4 tk = tk_next;
            tk_next = ktab[(k0 + BK + (FAST ? 0 : scol)) / 4];
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int iy = a_iy0[i] + tk.x, ix = a_ix0[i] + tk.y;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                A[i] = buf_load4(xr, ok ? a_off[i] - (FAST ? 0u : scol * 4u) + (unsigned)tk.z : OOB);
            }
#pragma unroll
            for (int i = 0; i < BR; ++i)
                B[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr_, b_off[i], k0 * 4, 0));
        };
        auto store_from = [&](const f32x4 (&A)[AR], const f32x4 (&B)[BR], int buf) {
            float* a = As + buf * BM * LDK;
            float* b = Bs + buf * BN * LDK;
#pragma unroll
            for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(a + (srow + RP * i) * LDK + scol) = A[i];
#pragma unroll
            for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(b + (srow + RP * i) * LDK + scol) = B[i];
        };
        auto read_frags = [&](int buf, int t, int idx) {
            const float* a = As + buf * BM * LDK + (wm * MI * 32 + frow) * LDK + fk + t * 8;
            const float* b = Bs + buf * BN * LDK + (wn * NI * 32 + frow) * LDK + fk + t * 8;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[idx][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[idx][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK);
        };
        auto mfma_step = [&](int idx) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[idx][i][r], fb[idx][j][r], acc[i][j], 0, 0, 0);
        };
        auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        const int nk = kt_end - kt_begin;
        load_into(ra, rb, kt_begin * BK);              // tile 0
        load_into(ra1, rb1, (kt_begin + 1) * BK);      // tile 1 (past-the-end tiles read zeros / slack, never used)
        store_from(ra, rb, 0);
        lds_barrier();
        load_into(ra, rb, (kt_begin + 2) * BK);        // tile 2
        read_frags(0, 0, 0);
        // one K step: LDS buffer `cur` holds tile k, (An, Bn) hold tile k+1 and are refilled with tile k+3
        auto body = [&](int k, int cur, f32x4 (&An)[AR], f32x4 (&Bn)[BR]) {
            read_frags(cur, 1, 1);
            mfma_step(0);
            __builtin_amdgcn_sched_barrier(0);
            store_from(An, Bn, cur ^ 1);
            read_frags(cur, 2, 0);
            mfma_step(1);
            __builtin_amdgcn_sched_barrier(0);
            load_into(An, Bn, (kt_begin + k + 3) * BK);
            read_frags(cur, 3, 1);
            mfma_step(0);
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            read_frags(cur ^ 1, 0, 0);
            mfma_step(1);
            __builtin_amdgcn_sched_barrier(0);
        };
        int k = 0;
        for (; k + 1 < nk; k += 2) {
            body(k, 0, ra1, rb1);
            body(k + 1, 1, ra, rb);
        }
        if (k < nk) body(k, 0, ra1, rb1);
        conv_epilogue<MI, NI, WGN>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
        return;
    }
    if (MID == 2) {
        // Software-pipelined schedule.  Registers hold tile k+1 in flight for a whole K step; it is
        // written to the other LDS buffer behind the second-to-last fragment step, the (single) barrier
        // follows, then the global loads of tile k+2 are issued and the first fragments of tile k+1
        // are read -- all of that under the MFMAs of the last fragment step.  The serial chain
        // "ds_write -> wait -> barrier -> ds_read -> wait" no longer sits between two MFMA bursts.
        constexpr int T = BK / 8;
        f32x4 fa[2][MI], fb[2][NI];
        auto read_frags = [&](int buf, int t, int idx) {
            const float* a = As + buf * BM * LDK + (wm * MI * 32 + frow) * LDK + fk + t * 8;
            const float* b = Bs + buf * BN * LDK + (wn * NI * 32 + frow) * LDK + fk + t * 8;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[idx][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[idx][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK);
        };
        auto mfma_step = [&](int idx) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[idx][i][r], fb[idx][j][r], acc[i][j], 0, 0, 0);
        };
        const int nk = kt_end - kt_begin;
        CONV_TL(0);
        load_tiles(kt_begin * BK);
        store_tiles(0);
        __syncthreads();
        CONV_TL(1);
        if (nk > 1) load_tiles((kt_begin + 1) * BK);
        read_frags(0, 0, 0);
        int cur = 0;
        for (int k = 0; k < nk; ++k) {
#ifdef ACCEL_CONV_TIMELINE
            if (k == nk / 8) CONV_TL(4); else if (k == nk / 4) CONV_TL(5); else if (k == nk / 2) CONV_TL(6); else if (k == (3 * nk) / 4) CONV_TL(7);
#endif
#pragma unroll
            for (int t = 0; t < T - 1; ++t) {
                read_frags(cur, t + 1, (t + 1) & 1);
                mfma_step(t & 1);
            }
            if (FAST) {
                // branch-free tail (one scheduling region): past-the-end tiles are loaded/stored too --
                // the tap table and the weight buffer carry slack for two extra K steps -- so the
                // scheduler can spread the address math, loads and fragment reads between the MFMAs
                store_tiles(cur ^ 1);
                __syncthreads();
                load_tiles((kt_begin + k + 2) * BK);
                read_frags(cur ^ 1, 0, 0);
                mfma_step((T - 1) & 1);
#pragma unroll
                for (int q = 0; q < MI * NI * 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x006, (AR * 8 + 16) / (MI * NI * 4) + 1, 0);   // some VALU/SALU
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                      // 1 VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // 1 DS read
                }
            } else {
                if (k + 1 < nk) store_tiles(cur ^ 1);
                __syncthreads();
                if (k + 2 < nk) load_tiles((kt_begin + k + 2) * BK);
                if (k + 1 < nk) read_frags(cur ^ 1, 0, 0);
                mfma_step((T - 1) & 1);
            }
            cur ^= 1;
        }
        CONV_TL(2);
        conv_epilogue<MI, NI, WGN>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
#ifdef ACCEL_CONV_TIMELINE
        __builtin_amdgcn_s_waitcnt(0);
        CONV_TL(3);
#endif
        return;
    }

    load_tiles(kt_begin * BK);
    store_tiles(0);
    __syncthreads();

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (ABL & 1) ? false : kt + 1 < kt_end;   // ABL bit0: no global loads / LDS stores in the loop
        if (more && !(ABL & 8)) load_tiles((kt + 1) * BK);       // ABL bit3: LDS stores of stale registers, no loads
        const float* a = As + cur * BM * LDK + (wm * MI * 32 + frow) * LDK + fk;
        const float* b = Bs + cur * BN * LDK + (wn * NI * 32 + frow) * LDK + fk;
        // register double-buffered fragments: t+1 is read from LDS while t feeds the matrix core
        f32x4 fa[2][MI], fb[2][NI];
        if (ABL & 16) { if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }   // experiment: de-phase co-resident blocks
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK);
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            const int c = t & 1, n = c ^ 1;
            // the other LDS buffer was last read before the previous barrier, so the next tile can
            // be written into it in the middle of this tile's MFMA burst instead of serialising at the end
            if (MID && more && t == BK / 16 && !(ABL & 4)) store_tiles(cur ^ 1);
            if (t + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[n][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK + (t + 1) * 8);
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[n][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK + (t + 1) * 8);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][r], fb[c][j][r], acc[i][j], 0, 0, 0);
        }
        if (ABL & 16) __builtin_amdgcn_s_setprio(0);
        if (!MID && more && !(ABL & 4)) store_tiles(cur ^ 1);
        if (ABL & 4) {                                            // ABL bit2: loads issued and waited for, no LDS stores
#pragma unroll
            for (int i = 0; i < AR; ++i) asm volatile("" ::"v"(ra[i]));
#pragma unroll
            for (int i = 0; i < BR; ++i) asm volatile("" ::"v"(rb[i]));
        }
        if (!(ABL & 2)) __syncthreads();                          // ABL bit1: no barrier
        if (!(ABL & 1)) cur ^= 1;
    }

    conv_epilogue<MI, NI, WGN>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
}

// ---------------------------------------------------------------------------
// LDS-DMA variant (FAST shapes only: Cin % 32 == 0).
// Tiles go HBM/L2 -> LDS directly (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write
// pass, and an S-deep LDS ring so the loads of K step k+S-1 are in flight while step k computes.
// One barrier per K step; waits are counted (`vmcnt(L*(S-2))`), never 0 inside the loop.
// The DMA writes lane-linearly (wave-uniform base + lane*16 B), so the LDS image is unpadded
// 128-byte rows; bank conflicts are avoided with an XOR swizzle applied on the SOURCE side:
// physical 16-byte slot s of row r holds logical K-quad c = s ^ ((r >> 1) & 7), and fragment reads
// use the same XOR (conflict-free for the 16-lane groups of ds_read_b128).
// Out-of-image taps / ragged rows use out-of-range buffer offsets: the DMA writes zeros.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int BM, int BN, int WGM, int WGN, int S>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_dma_kernel(ConvParams p)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtin only type-checks in the device pass; the host needs the stub alone
    constexpr int BK = 32;
    constexpr int NW = WGM * WGN;
    constexpr int MI = BM / (WGM * 32), NI = BN / (WGN * 32);
    constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;      // DMA instructions per wave per stage (8 rows each)
    constexpr int L = AI + BI;
    constexpr int STAGE = (BM + BN) * BK;                  // floats per ring stage
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nblk = p.MT * p.NT;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int nt = swz % p.NT, mt = swz / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    int ph = p.ph, pw = p.pw;
    const float* wbase = p.w;
    int py = 0, px = 0;
    if (p.deconv2x) {
        py = blockIdx.y >> 1; px = blockIdx.y & 1;
        ph = 1 - py; pw = 1 - px;
        wbase += (size_t)blockIdx.y * p.w_class_stride;
    }
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t wr_ = make_rsrc(wbase, p.w_bytes);
    const int HoWo = p.Ho * p.Wo;

    // DMA lane assignment: instruction q covers tile rows 8q..8q+7; lane -> (row, physical slot)
    const int lrow = lane >> 3, lslot = lane & 7;
    int a_iy0[AI], a_ix0[AI];
    unsigned a_off[AI], b_off[BI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int r = (wave * AI + j) * 8 + lrow;
        const int c = lslot ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_iy0[j] = ok ? oy * p.sh - ph : -(1 << 28);
        a_ix0[j] = ox * p.sw - pw;
        a_off[j] = (unsigned)((((n * p.H + a_iy0[j]) * p.W + a_ix0[j]) * p.xCs + c * 4) * 4);
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int r = (wave * BI + j) * 8 + lrow;
        const int c = lslot ^ ((r >> 1) & 7);
        b_off[j] = (unsigned)(((size_t)(n0 + r) * p.K_pad + c * 4) * 4);
    }

    const int KT_all = p.K_pad / BK;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per_split : 0;
    const int kt_end = p.ksplit > 1 ? min(KT_all, kt_begin + p.kt_per_split) : KT_all;
    const int4* ktab = p.ktab + (p.deconv2x ? blockIdx.y * (p.K_pad / 4 + 48) : 0);

    auto issue = [&](int kt, int slot) {      // DMA K step `kt` into ring slot `slot`
        const int4 tk = ktab[kt * 8];         // {dy, dx, byte offset, 0}
        const int as = slot * STAGE, bs = as + BM * BK;     // float offsets into the ring (cast straight from smem: AS3)
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            const int iy = a_iy0[j] + tk.x, ix = a_ix0[j] + tk.y;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(smem + as + (wave * AI + j) * 8 * BK), 16,
                                                     ok ? a_off[j] + (unsigned)tk.z : OOB, 0, 0, 0);
        }
        const int koff = kt * BK * 4;
#pragma unroll
        for (int j = 0; j < BI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr_, (lds_void_ptr)(smem + bs + (wave * BI + j) * 8 * BK), 16,
                                                     b_off[j], koff, 0, 0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: S-1 stages in flight (past-the-end steps are issued too, as all-zero K: the tap
    // table marks them out of range and the weight rows beyond K_pad are never read -- guard instead)
    const int nk = kt_end - kt_begin;
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nk) issue(kt_begin + s, s);

    // fragment addressing: row r = lane&31 (+32*i), logical quad c = 2t + (lane>>5)
    const int frow = lane & 31, fh = lane >> 5;
    int a_rowoff[MI], b_rowoff[NI], a_x[MI], b_x[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) { const int r = (wm * MI + i) * 32 + frow; a_rowoff[i] = r * BK; a_x[i] = (r >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < NI; ++j) { const int r = (wn * NI + j) * 32 + frow; b_rowoff[j] = BM * BK + r * BK; b_x[j] = (r >> 1) & 7; }

    for (int k = 0; k < nk; ++k) {
        // my own DMAs of step k have landed once at most (S-2) newer stages are outstanding
        const int newer = min(S - 2, nk - 1 - k);
        if (newer >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L * (S - 2)) : "memory");
        else if (S > 2 && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everyone's step-k data is in LDS; slot (k-1)%S is free
        if (k + S - 1 < nk) issue(kt_begin + k + S - 1, (k + S - 1) % S);
        const float* st = smem + (k % S) * STAGE;
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            f32x4 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[i] = *reinterpret_cast<const f32x4*>(st + a_rowoff[i] + (((2 * t + fh) ^ a_x[i]) << 2));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                fb[j] = *reinterpret_cast<const f32x4*>(st + b_rowoff[j] + (((2 * t + fh) ^ b_x[j]) << 2));
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][r], fb[j][r], acc[i][j], 0, 0, 0);
        }
    }

    conv_epilogue<MI, NI, WGN>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
#endif
}

// ---------------------------------------------------------------------------
// fp16-MFMA variant (BASELINE config 5: "fp16 convs", fp32 accumulate).
// Activations stay fp32 NHWC in HBM; the loader converts each 8-float K chunk to half while
// staging it into LDS, weights are pre-packed as half [rows][K_pad].  v_mfma_f32_32x32x16_f16:
// lane l holds A[i = l&31][k = 8*(l>>5) .. +8] / B[k][j = l&31] as one 16-byte fragment.
// LDS rows are 32 halves + 8 pad (80 B: conflict-free ds_read_b128).  Needs Cin % 8 == 0 (a chunk
// never straddles two taps); other layers (RGB stems, ragged concats) run on the fp32 kernel.
// Same epilogue, same split-K, same software-pipelined schedule as the fp32 kernel.
// ---------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_f16_kernel(ConvParams p)
{
    constexpr int BK = 32, LDK = BK + 8;            // halves
    constexpr int NTHR = 64 * WGM * WGN;
    constexpr int CPR = BK / 8, RP = NTHR / CPR;    // 8-wide chunks per row, rows per pass
    constexpr int MI = BM / (WGM * 32), NI = BN / (WGN * 32);
    constexpr int AR = BM / RP, BR = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0, "tile / thread-count mismatch");
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* As = smem_h;                   // [2][BM][LDK]
    _Float16* Bs = smem_h + 2 * BM * LDK;    // [2][BN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nblk = p.MT * p.NT, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int nt = swz % p.NT, mt = swz / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    int kw = p.kw, ph = p.ph, pw = p.pw;
    const _Float16* wbase = reinterpret_cast<const _Float16*>(p.w);
    int py = 0, px = 0;
    if (p.deconv2x) {
        py = blockIdx.y >> 1; px = blockIdx.y & 1;
        ph = 1 - py; pw = 1 - px;
        wbase += (size_t)blockIdx.y * p.w_class_stride;
    }
    const int ntaps = p.kh * kw;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t wr_ = make_rsrc(wbase, p.w_bytes);
    const int HoWo = p.Ho * p.Wo;

    const int srow = tid / CPR, scol = (tid % CPR) * 8;
    int a_iy0[AR], a_ix0[AR], a_nb[AR];
    unsigned b_off[BR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + RP * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_iy0[i] = ok ? oy * p.sh - ph : -(1 << 28);
        a_ix0[i] = ox * p.sw - pw;
        a_nb[i] = n * p.H * p.W;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) b_off[i] = (unsigned)(((size_t)(n0 + srow + RP * i) * p.K_pad + scol) * 2);

    const int KT_all = p.K_pad / BK;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per_split : 0;
    const int kt_end = p.ksplit > 1 ? min(KT_all, kt_begin + p.kt_per_split) : KT_all;

    f16x8 ra[AR], rb[BR];
    const int4* ktab = p.ktab + (p.deconv2x ? blockIdx.y * (p.K_pad / 4 + 48) : 0);
    int4 tk_next = ktab[(kt_begin * BK + scol) / 4];
    unsigned a_base[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) a_base[i] = (unsigned)(((a_nb[i] + a_iy0[i] * p.W + a_ix0[i]) * p.xCs) * 4);
    auto load_tiles = [&](int k0) {
        const int4 tk = tk_next;                 // the lane's own 8-wide chunk = two 4-wide granules of one tap
        tk_next = ktab[(k0 + BK + scol) / 4];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = a_iy0[i] + tk.x, ix = a_ix0[i] + tk.y;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = ok ? a_base[i] + (unsigned)tk.z : OOB;
            const f32x4 lo = buf_load4(xr, off), hi = buf_load4(xr, ok ? off + 16u : OOB);
            f16x8 h;
            h[0] = (_Float16)lo[0]; h[1] = (_Float16)lo[1]; h[2] = (_Float16)lo[2]; h[3] = (_Float16)lo[3];
            h[4] = (_Float16)hi[0]; h[5] = (_Float16)hi[1]; h[6] = (_Float16)hi[2]; h[7] = (_Float16)hi[3];
            ra[i] = h;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            rb[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr_, b_off[i], k0 * 2, 0));
    };
    auto store_tiles = [&](int buf) {
        _Float16* a = As + buf * BM * LDK;
        _Float16* b = Bs + buf * BN * LDK;
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<f16x8*>(a + (srow + RP * i) * LDK + scol) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *reinterpret_cast<f16x8*>(b + (srow + RP * i) * LDK + scol) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const int nk = kt_end - kt_begin;
    load_tiles(kt_begin * BK);
    store_tiles(0);
    __syncthreads();
    if (nk > 1) load_tiles((kt_begin + 1) * BK);
    int cur = 0;
    for (int k = 0; k < nk; ++k) {
        const _Float16* a = As + cur * BM * LDK + (wm * MI * 32 + frow) * LDK + fk;
        const _Float16* b = Bs + cur * BN * LDK + (wn * NI * 32 + frow) * LDK + fk;
#pragma unroll
        for (int kb = 0; kb < BK / 16; ++kb) {
            f16x8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f16x8*>(a + i * 32 * LDK + kb * 16);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const f16x8*>(b + j * 32 * LDK + kb * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (k + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
        if (k + 2 < nk) load_tiles((kt_begin + k + 2) * BK);
        cur ^= 1;
    }
    conv_epilogue<MI, NI, WGN>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
}

// ---------------------------------------------------------------------------
// fp32 on the bf16 matrix cores ("bf16x3", f16 == 2): every fp32 operand is split EXACTLY into three bf16 terms
//     a = a0 + a1 + a2,   a0 = top 16 bits of a, a1 = top 16 bits of (a - a0), a2 = top 16 bits of (a - a0 - a1)
// (8 significant bits each: 24 together, the residuals are exact in fp32), and a product is taken as the six terms
//     a*b ~= a1*b1 + a0*b2 + a2*b0 + a0*b1 + a1*b0 + a0*b0          (the dropped terms are below 2^-23 |a*b|)
// each an exact fp32 value (8 x 8 bits) accumulated in fp32 by v_mfma_f32_32x32x16_bf16: the same accumulation the fp32
// MFMA does, with a relative operand error of 2^-23 instead of rounding -- fp32-equivalent results (op-level error
// against float64 is the same as the fp32 kernel's, tests/test_ops_gpu.py) from a pipe with 16x the fp32 MFMA rate,
// at 6 products: 2.6x the fp32 matrix peak, and it leaves the vector ALUs (which the fp32 MFMA occupies) to the loader.
// Weights are split on the host (three planes [3][rows][K_pad] of bf16), activations in the loader while they are staged
// into LDS (three planes per tile).  Same tiles, addressing table, split-K and epilogue as the fp16 kernel.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned& q0, unsigned& q1, unsigned& q2)
{
    const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
    q0 = __builtin_amdgcn_perm(u1, u0, 0x07060302);                       // {top16(v1), top16(v0)}
    const float r0 = v0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u), r1 = v1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    const unsigned s0 = __builtin_bit_cast(unsigned, r0), s1 = __builtin_bit_cast(unsigned, r1);
    q1 = __builtin_amdgcn_perm(s1, s0, 0x07060302);
    const float t0 = r0 - __builtin_bit_cast(float, s0 & 0xFFFF0000u), t1 = r1 - __builtin_bit_cast(float, s1 & 0xFFFF0000u);
    q2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0), 0x07060302);
}

// 16 bytes per lane, global -> LDS (lane i lands at lds + 16*i); outside the kernel template (see conv_1x1ws.hip)
__device__ __forceinline__ void b3_dma16(__amdgpu_buffer_rsrc_t r, unsigned short* lds, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// WDMA: the weight planes travel global -> LDS by DMA (no registers, no ds_write), double-buffered and one K step ahead;
// their rows are unpadded (64 bytes) with the 16-byte chunks XOR-swizzled by (row >> 2) & 3 on the SOURCE side, which
// keeps the fragment reads conflict-free.
// FAST (Cin % 32 == 0): a K step lies inside one tap, its table entry is wave-uniform and comes through the scalar unit.
// Per lane (the general case) the entry of the NEXT step is a vector load whose result the compiler waits for on the
// spot (a loop-carried copy): one exposed load latency per K step, 15-24 % of the kernel in the first version.
template <int BM, int BN, int WGM, int WGN, bool WDMA, bool FAST>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_b3_kernel(ConvParams p, size_t wplane)
{
    constexpr int BK = 32, LDK = BK + 8;            // bf16 elements
    constexpr int NTHR = 64 * WGM * WGN;
    constexpr int CPR = BK / 8, RP = NTHR / CPR;    // 8-wide chunks per row, rows per pass
    constexpr int MI = BM / (WGM * 32), NI = BN / (WGN * 32);
    constexpr int AR = BM / RP, BR = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0, "tile / thread-count mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_b[];
    unsigned short* As = smem_b;                      // [3][BM][LDK]: ONE stage, two blocks per CU (see the loop)
    unsigned short* Bs = smem_b + 3 * BM * LDK;       // [3][BN][LDK]; WDMA: [2][3][BN][32]

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
    __amdgpu_buffer_rsrc_t wr_[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wr_[pl] = make_rsrc(wbase + pl * wplane, p.w_bytes);
    const int HoWo = p.Ho * p.Wo;

    // Staging rows: a 16-byte LDS store is serviced 8 lanes at a time over 32 banks, and with 4 lanes per 64-byte row that is two
    // rows per group.  Rows r and r+1 (80 bytes apart) share 4 banks -- a conflict on EVERY store, 29 % of all LDS cycles
    // in the first version (SQ_LDS_BANK_CONFLICT) -- rows r and r+4 (320 bytes = 16 banks mod 32) share none: the rows of
    // each group of 8 are taken in the order 0,4,1,5,2,6,3,7.
    const int t4 = tid / CPR;
    const int srow = (t4 & ~7) | ((t4 & 1) << 2) | ((t4 >> 1) & 3), scol = (tid % CPR) * 8;
    int a_iy0[AR], a_ix0[AR], a_nb[AR];
    unsigned b_off[BR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + RP * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_iy0[i] = ok ? oy * p.sh - ph : -(1 << 28);
        a_ix0[i] = ox * p.sw - pw;
        a_nb[i] = n * p.H * p.W;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) b_off[i] = (unsigned)(((n0 + srow + RP * i) * 32 + scol) * 2);      // [K step][row][32] planes
    const unsigned b_step = p.w_bytes / (unsigned)(p.K_pad / BK);      // bytes of one K step of one plane: rows x 64

    const int KT_all = p.K_pad / BK;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per_split : 0;
    const int kt_end = p.ksplit > 1 ? min(KT_all, kt_begin + p.kt_per_split) : KT_all;

    f32x4 ralo[AR], rahi[AR];          // the activations travel as fp32 and are split when they are stored to LDS
    i32x4 rb[BR][3];
    const int4* ktab = p.ktab + (p.deconv2x ? blockIdx.y * (p.K_pad / 4 + 48) : 0);
    int4 tk_next = FAST ? ktab[(kt_begin * BK) / 4] : ktab[(kt_begin * BK + scol) / 4];
    int4 tk2_next = ktab[(kt_begin * BK + scol) / 4 + 1];      // general case only
    unsigned a_base[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) a_base[i] = (unsigned)(((a_nb[i] + a_iy0[i] * p.W + a_ix0[i]) * p.xCs) * 4);
    // WDMA: instruction g of the block covers plane g / (BN/16), rows 16 * (g % (BN/16)) .. +16 (4 chunks each)
    constexpr int NW = WGM * WGN, DPW = 3 * (BN / 16) / NW;      // DMA instructions per wavefront and K step
    static_assert(!WDMA || (3 * (BN / 16)) % NW == 0, "weight DMA pieces do not divide over the wavefronts");
    const unsigned dma_voff = (unsigned)(((n0 + (lane >> 2)) * 64) + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
    const __amdgpu_buffer_rsrc_t wall = make_rsrc(wbase, (unsigned)(4 * wplane) + p.w_bytes);      // the three planes of this class
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#ifdef B3_TIMING
    long long tsub[2] = {0, 0};
#endif
    auto issue_b = [&](int k0, int stage) {
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int g = wave_u * DPW + j, pl = g / (BN / 16), part = g % (BN / 16);
            b3_dma16(wall, Bs + ((stage * 3 + pl) * BN + part * 16) * 32, dma_voff + (unsigned)(part * 1024),
                     (unsigned)(pl * 2 * wplane) + (unsigned)((k0 / BK) * b_step));
        }
    };
    auto load_tiles = [&](int k0) {
        if constexpr (FAST) {
            int4 tk = tk_next;                   // uniform index: scalar loads, one K step ahead (re-issued at the END of this
            tk.z += scol * 4;                    // function: scalar loads return out of order, so the compiler's wait for `tk`
#pragma unroll                                   // would otherwise also wait for the one just issued -- a full miss latency)
            for (int i = 0; i < AR; ++i) {
                const int iy = a_iy0[i] + tk.x, ix = a_ix0[i] + tk.y;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned off = ok ? a_base[i] + (unsigned)tk.z : OOB;
                ralo[i] = buf_load4(xr, off);
                rahi[i] = buf_load4(xr, ok ? off + 16u : OOB);
            }
        } else {
            // the lane's own 8-wide chunk = two 4-wide granules, each with its own tap entry (with Cin % 8 == 4 the second
            // granule may belong to the next tap: FlowNet's deconvolutions read 386 / 770 / 1026 channels)
            const int4 tk = tk_next, tk2 = tk2_next;
            tk_next = ktab[(k0 + BK + scol) / 4];
            tk2_next = ktab[(k0 + BK + scol) / 4 + 1];
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int iy = a_iy0[i] + tk.x, ix = a_ix0[i] + tk.y;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const int iy2 = a_iy0[i] + tk2.x, ix2 = a_ix0[i] + tk2.y;
                const bool ok2 = (unsigned)iy2 < (unsigned)p.H && (unsigned)ix2 < (unsigned)p.W;
                ralo[i] = buf_load4(xr, ok ? a_base[i] + (unsigned)tk.z : OOB);
                rahi[i] = buf_load4(xr, ok2 ? a_base[i] + (unsigned)tk2.z : OOB);
            }
        }
        if (!WDMA) {
#pragma unroll
            for (int i = 0; i < BR; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) rb[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(wr_[pl], b_off[i], (k0 / BK) * b_step, 0);
        }
        if (FAST) {
            __builtin_amdgcn_sched_barrier(0);
            tk_next = ktab[(k0 + BK) / 4];
        }
    };
    auto store_tiles = [&]() {
        unsigned short* a = As;
        unsigned short* b = Bs;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            i32x4 q[3];
            unsigned x0, x1, x2;
            split3_pair(ralo[i][0], ralo[i][1], x0, x1, x2); q[0][0] = (int)x0; q[1][0] = (int)x1; q[2][0] = (int)x2;
            split3_pair(ralo[i][2], ralo[i][3], x0, x1, x2); q[0][1] = (int)x0; q[1][1] = (int)x1; q[2][1] = (int)x2;
            split3_pair(rahi[i][0], rahi[i][1], x0, x1, x2); q[0][2] = (int)x0; q[1][2] = (int)x1; q[2][2] = (int)x2;
            split3_pair(rahi[i][2], rahi[i][3], x0, x1, x2); q[0][3] = (int)x0; q[1][3] = (int)x1; q[2][3] = (int)x2;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<i32x4*>(a + (pl * BM + srow + RP * i) * LDK + scol) = q[pl];
        }
        if (!WDMA) {
#pragma unroll
            for (int i = 0; i < BR; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<i32x4*>(b + (pl * BN + srow + RP * i) * LDK + scol) = rb[i][pl];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const int nk = kt_end - kt_begin;
    if (WDMA) issue_b(kt_begin * BK, 0);
    load_tiles(kt_begin * BK);
    if (WDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_tiles();
    __syncthreads();
    if (nk > 1) load_tiles((kt_begin + 1) * BK);
    // One LDS stage for the pixel tile (two for the DMA'd weights): TWO blocks fit a CU, and while one multiplies (matrix
    // pipe) the other splits and stages its next tile (vector ALU + LDS writes) -- different units, so the phases overlap.
#ifdef B3_TIMING
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    tsub[0] = tsub[1] = 0;
#endif
    for (int k = 0; k < nk; ++k) {
#ifdef B3_TIMING
        const long long tq0 = clock64();
#endif
        if (WDMA && k + 1 < nk) issue_b((kt_begin + k + 1) * BK, (k + 1) & 1);      // lands while this tile is multiplied
        const unsigned short* a = As + (wm * MI * 32 + frow) * LDK + fk;
        const unsigned short* b = WDMA ? Bs + ((k & 1) * 3 * BN + wn * NI * 32 + frow) * 32
                                       : Bs + (wn * NI * 32 + frow) * LDK + fk;
#pragma unroll
        for (int kb = 0; kb < BK / 16; ++kb) {
            bf16x8 fa[MI][3], fb[NI][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][pl] = *reinterpret_cast<const bf16x8*>(a + (pl * BM + i * 32) * LDK + kb * 16);
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    fb[j][pl] = WDMA ? *reinterpret_cast<const bf16x8*>(b + (pl * BN + j * 32) * 32 + (((2 * kb + (lane >> 5)) ^ ((frow >> 2) & 3)) * 8))
                                     : *reinterpret_cast<const bf16x8*>(b + (pl * BN + j * 32) * LDK + kb * 16);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    f32x16 c = acc[i][j];      // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                }
        }
#ifdef B3_TIMING
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[MI - 1][NI - 1][15]));
        const long long tq1 = clock64(); tacc[0] += tq1 - tq0;
#endif
        __syncthreads();                       // everybody has read this tile
#ifdef B3_TIMING
        const long long tq2 = clock64(); tacc[1] += tq2 - tq1;
#endif
        // the pixel registers of tile k+1 AND this wavefront's part of the weight DMA of tile k+1: nothing else is in flight
        if (WDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef B3_TIMING
        const long long tq3 = clock64(); tacc[2] += tq3 - tq2;
#endif
        if (k + 1 < nk) store_tiles();
#ifdef B3_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long tq4 = clock64(); tacc[3] += tq4 - tq3;
#endif
        __syncthreads();
#ifdef B3_TIMING
        const long long tq5 = clock64(); tacc[4] += tq5 - tq4;
#endif
        if (k + 2 < nk) load_tiles((kt_begin + k + 2) * BK);
#ifdef B3_TIMING
        tacc[5] += clock64() - tq5;
#endif
    }
#ifdef B3_TIMING
    if (blockIdx.x == 777 && lane == 0) { for (int i = 0; i < 6; ++i) p.y[wave * 8 + i] = (float)tacc[i] / (float)nk; p.y[wave * 8 + 6] = (float)tsub[0] / (float)nk; p.y[wave * 8 + 7] = (float)tsub[1] / (float)nk; }
    if (blockIdx.x == 777) return;
#endif
    conv_epilogue<MI, NI, WGN>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
}

// Sum the split-K partials and apply the fused epilogue (one float4 of channels per thread).
__global__ void splitk_reduce_kernel(ConvParams p, int classes)
{
    const int N4 = p.Cout_store / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // fp16x2 form: the partial sums carry the pixel scale of the launch that wrote them -- the same slot, the same value (range.h)
    const float xinv = range_prologue(p.xr).inv;
    unsigned rmax = 0u, rmax2 = 0u;
    if (idx < (long)classes * p.M * N4) {
    const int c4 = (int)(idx % N4);
    const int m = (int)((idx / N4) % p.M);
    const int cls = (int)(idx / ((long)N4 * p.M));
    const size_t slab = (size_t)classes * p.M * p.Cout_store;
    const float* w = p.ws + ((size_t)cls * p.M + m) * p.Cout_store + c4 * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.ksplit; ++s) a += *reinterpret_cast<const f32x4*>(w + s * slab);
    size_t pix = (size_t)m;
    if (p.deconv2x) {
        const int HoWo = p.Ho * p.Wo;
        const int n = m / HoWo, rem = m - n * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int yy = 2 * oy + (cls >> 1), xx = 2 * ox + (cls & 1);
        if (yy >= p.yH || xx >= p.yW) pix = ~(size_t)0;           // cropped output row / column
        else pix = ((size_t)n * p.yH + yy) * p.yW + xx;
    }
    if (pix != ~(size_t)0) {
    const int co = c4 * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + co) * xinv;      // fp16x2 form: pixel exponent undone
    const f32x4 sf = *reinterpret_cast<const f32x4*>(p.shift + co);
    f32x4 v = a * sc + sf;
    typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));
    if (p.res && p.res_half) {
        const f16x4s r = *reinterpret_cast<const f16x4s*>(reinterpret_cast<const _Float16*>(p.res) + pix * p.resCs + co);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
    } else if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + pix * p.resCs + co);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p.act == 1) v[e] = fmaxf(v[e], 0.f);
        else if (p.act == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
    }
    if (p.y_half) {      // half storage: rounded (RTNE) after the whole epilogue, like conv_epilogue_h
        f16x4s h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
        *reinterpret_cast<f16x4s*>(reinterpret_cast<_Float16*>(p.y) + pix * p.yCs + co) = h;
    } else {
    *reinterpret_cast<f32x4*>(p.y + pix * p.yCs + co) = v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(v[e]); rmax = b > rmax ? b : rmax; }
    if (p.y2) {
        const f32x4 s2 = *reinterpret_cast<const f32x4*>(p.scale2 + co);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.shift2 + co);
        f32x4 u = v * s2 + b2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { u[e] = fmaxf(u[e], 0.f); const unsigned b = range_abs_bits(u[e]); rmax2 = b > rmax2 ? b : rmax2; }
        *reinterpret_cast<f32x4*>(p.y2 + pix * p.y2Cs + co) = u;
    }
    }
    }
    }
    // range slots of the outputs (every thread of the block takes part)
    if (p.yr && !p.y_half) range_note_block(p.yr, rmax, blockIdx.x);
    if (p.y2 && p.y2r) range_note_block(p.y2r, rmax2, blockIdx.x);
}

template <int BM, int BN, int WGM, int WGN, int BK, int MID, int ABL = 0, int FAST = 0>
static hipError_t launch_cfg2(const ConvParams& p0, hipStream_t st);

template <int BM, int BN, int WGM, int WGN, int BK, int MID, int ABL = 0>
static hipError_t launch_cfg(const ConvParams& p0, hipStream_t st)
{
    if (p0.Cin % BK == 0) return launch_cfg2<BM, BN, WGM, WGN, BK, MID, ABL, 1>(p0, st);
    return launch_cfg2<BM, BN, WGM, WGN, BK, MID, ABL, 0>(p0, st);
}

template <int BM, int BN, int WGM, int WGN, int BK, int MID, int ABL, int FAST>
static hipError_t launch_cfg2(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.Cout_store + BN - 1) / BN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_f32_kernel<BM, BN, WGM, WGN, BK, MID, ABL, FAST>), lds); e != hipSuccess) return e;
    dim3 grid(p.MT * p.NT, p.deconv2x ? 4 : 1, p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WGM, WGN, BK, MID, ABL, FAST>), grid, dim3(64 * WGM * WGN), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit <= 1) return e;
    const long total = (long)grid.y * p.M * (p.Cout_store / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, (int)grid.y);
    return hipGetLastError();
}

template <int BM, int BN, int WGM, int WGN, int S>
static hipError_t launch_dma(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.Cout_store + BN - 1) / BN;
    constexpr size_t lds = (size_t)S * (BM + BN) * 32 * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_dma_kernel<BM, BN, WGM, WGN, S>), lds); e != hipSuccess) return e;
    dim3 grid(p.MT * p.NT, p.deconv2x ? 4 : 1, p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, BN, WGM, WGN, S>), grid, dim3(64 * WGM * WGN), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit <= 1) return e;
    const long total = (long)grid.y * p.M * (p.Cout_store / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, (int)grid.y);
    return hipGetLastError();
}

template <int BM, int BN, int WGM, int WGN>
static hipError_t launch_f16(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.Cout_store + BN - 1) / BN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 40 * sizeof(_Float16);
    dim3 grid(p.MT * p.NT, p.deconv2x ? 4 : 1, p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, WGM, WGN>), grid, dim3(64 * WGM * WGN), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit <= 1) return e;
    const long total = (long)grid.y * p.M * (p.Cout_store / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, (int)grid.y);
    return hipGetLastError();
}

template <int BM, int BN, int WGM, int WGN>
static hipError_t launch_b3(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    p.MT = (p.M + BM - 1) / BM;
    p.NT = (p.Cout_store + BN - 1) / BN;
    constexpr bool WDMA = WGM * WGN == 8;      // measured: the DMA'd weight stream wins with 8 wavefronts (+5...14 %), loses with 4
    constexpr size_t lds = WDMA ? (size_t)(3 * BM * 40 + 2 * 3 * BN * 32) * sizeof(unsigned short) : (size_t)3 * (BM + BN) * 40 * sizeof(unsigned short);
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_b3_kernel<BM, BN, WGM, WGN, WDMA, true>), lds); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_igemm_b3_kernel<BM, BN, WGM, WGN, WDMA, false>), lds); e != hipSuccess) return e;
    dim3 grid(p.MT * p.NT, p.deconv2x ? 4 : 1, p.ksplit > 1 ? p.ksplit : 1);
    if (p.Cin % 32 == 0) hipLaunchKernelGGL((conv_igemm_b3_kernel<BM, BN, WGM, WGN, WDMA, true>), grid, dim3(64 * WGM * WGN), lds, st, p, p.w_plane);
    else hipLaunchKernelGGL((conv_igemm_b3_kernel<BM, BN, WGM, WGN, WDMA, false>), grid, dim3(64 * WGM * WGN), lds, st, p, p.w_plane);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit <= 1) return e;
    const long total = (long)grid.y * p.M * (p.Cout_store / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, (int)grid.y);
    return hipGetLastError();
}

// Tile choice: the chip has 256 CUs; prefer the largest tile that still gives
// >= ~2 blocks per CU, narrow-N tiles for the 2/19/72-channel layers.
int conv_pick_tile(const ConvParams& p)
{
    if (p.force_tile >= 0) return p.force_tile;
    if (p.x_half || p.y_half || p.res_half) return p.Cout_store <= 64 ? CONV_TILE_B3D + 6 : CONV_TILE_B3D + 2;   // half views: conv_b3d.hip only
    const long classes = p.deconv2x ? 4 : 1;
    const int cs = p.Cout_store;
    if (cs <= 32) return p.f16 ? 3 : 4;                  // 128x32 (fp32) / 64x64 (fp16 path has no 32-wide tile)
    auto blocks = [&](int bm, int bn) {
        return classes * ((p.M + bm - 1) / bm) * (long)((cs + bn - 1) / bn);
    };
    if (cs <= 64) return blocks(128, 64) >= 512 ? 1 : 3; // 128x64 or 64x64
    if (blocks(128, 128) < 64) return 3;                 // tiny M: small tiles + split-K
    if (cs % 128 == 0 || cs > 256) {
        if (blocks(128, 128) >= 512) return 0;
        if (blocks(128, 64) >= 512) return 1;
        if (blocks(64, 128) >= 384) return 2;
        return 3;
    }
    if (blocks(128, 64) >= 512) return 1;
    return 3;
}

hipError_t launch_splitk_reduce(const ConvParams& p, int classes, hipStream_t st)
{
    const long total = (long)classes * p.M * (p.Cout_store / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, classes);
    return hipGetLastError();
}

int conv_tile_bk(int tile) { return tile == 13 ? 64 : tile == 15 ? 16 : 32; }

// launch-geometry ids this build carries (all of them compute the same contraction)
bool conv_tile_valid(int tile)
{
#ifdef ACCEL_CONV_DIAG
    if ((tile >= 20 && tile <= 30) || (tile >= 90 && tile <= 96)) return true;
#endif
    return (tile >= 0 && tile <= 19) || (tile >= 31 && tile <= 35) || tile == CONV_TILE_WINO || tile == CONV_TILE_WINO_B3 || tile == CONV_TILE_WINO_B3U || tile == CONV_TILE_WINO_B3S || tile == CONV_TILE_STEM || tile == CONV_TILE_STEM_B3 || tile == CONV_TILE_WS || tile == CONV_TILE_HALO ||
           (tile >= CONV_TILE_B3 && tile < CONV_TILE_B3 + 8) || (tile >= CONV_TILE_B3R + 3 && tile <= CONV_TILE_B3R + 5) ||
           (tile >= CONV_TILE_B3D && tile < CONV_TILE_B3D + CONV_TILE_B3D_N);
}

static void tile_dims(int tile, int& bm, int& bn)
{
    static const int BMs[20] = {128, 128, 64, 64, 128, 128, 128, 64, 64, 128, 128, 128, 64, 64, 256, 128, 128, 64, 128, 64};
    static const int BNs[20] = {128, 64, 128, 64, 32, 128, 64, 128, 64, 32, 128, 64, 128, 64, 128, 128, 128, 64, 128, 64};
    if (tile == CONV_TILE_B3 + 5) { bm = 256; bn = 128; return; }
    if (tile == CONV_TILE_B3R || (tile >= 90 && tile <= 96)) { bm = 128; bn = 128; return; }
    if (tile == CONV_TILE_B3R + 1) { bm = 128; bn = 64; return; }
    if (tile == CONV_TILE_B3R + 3 || tile == CONV_TILE_B3R + 4) { bm = 128; bn = 256; return; }
    if (tile == CONV_TILE_B3R + 5) { bm = 128; bn = 128; return; }
    if (tile == CONV_TILE_B3D || tile == CONV_TILE_B3D + 4) { bm = 256; bn = 256; return; }
    if (tile == CONV_TILE_B3D + 1 || tile == CONV_TILE_B3D + 5) { bm = 128; bn = 256; return; }
    if (tile == CONV_TILE_B3D + 2 || tile == CONV_TILE_B3D + 3) { bm = 128; bn = 128; return; }
    if (tile == CONV_TILE_B3D + 6) { bm = 128; bn = 64; return; }
    if (tile == CONV_TILE_B3D + 7) { bm = 256; bn = 128; return; }
    if (tile >= CONV_TILE_B3 && tile < CONV_TILE_B3 + 5) { static const int g[5] = {0, 1, 2, 3, 10}; tile = g[tile - CONV_TILE_B3]; }
    if (tile >= 31 && tile <= 35) { static const int g[5] = {3, 0, 2, 1, 4}; tile = g[tile - 31]; }   // deep-prefetch variants
    if (tile >= 20) tile = (tile == 23 || (tile >= 26 && tile != 29)) ? 3 : 0;
    if (tile < 0 || tile > 19) tile = 3;
    bm = BMs[tile]; bn = BNs[tile];
}

// Fill the chip when the output grid alone cannot: split K across blockIdx.z.
// Returns the workspace bytes the launch needs (0 = no split).
size_t conv_plan_split(ConvParams& p)
{
    p.ksplit = 1; p.kt_per_split = 0;
    if (p.no_split || p.narrow) return 0;
    int bm, bn;
    const int tile = conv_pick_tile(p);
    if (tile == CONV_TILE_STEM || tile == CONV_TILE_STEM_B3 || tile == CONV_TILE_WS || tile == CONV_TILE_HALO) return 0;
    if (tile == CONV_TILE_WINO) {
        // Winograd blocks own 64 tiles (256 pixels) x 64 channels; the K loop runs in steps of 8 channels, unrolled by 2.
        // Splitting it over blockIdx.y leaves raw partial OUTPUTS (the output transform is linear) that the ordinary
        // split-K reduce kernel sums and finishes.
        const long blocks = ((p.M / 4 + 63) / 64) * (long)conv_wino_rows(p.Cout_store) / 64;
        const int KT = p.Cin / 8, min_steps = 4;
        const int min_blocks = p.split_target > 0 ? p.split_target : 256;
        const int target = p.split_target > 0 ? p.split_target : 512;
        if (blocks >= min_blocks || KT < 2 * min_steps) return 0;
        int want = (int)((target + blocks - 1) / blocks);
        int ks = want < KT / min_steps ? want : KT / min_steps;
        if (ks < 2) return 0;
        p.kt_per_split = ((KT + ks - 1) / ks + 1) / 2 * 2;             // even: the K loop is unrolled by 2
        p.ksplit = (KT + p.kt_per_split - 1) / p.kt_per_split;
        if (p.ksplit < 2) { p.ksplit = 1; p.kt_per_split = 0; return 0; }
        return (size_t)p.ksplit * p.M * p.Cout_store * sizeof(float);
    }
    if (tile == CONV_TILE_WINO_B3 || tile == CONV_TILE_WINO_B3U || tile == CONV_TILE_WINO_B3S) {
        // conv_wino_b3.hip: blocks of 64 tiles x 64 channels, K steps of 16 channels; raw partial outputs as for tile 40
        const long blocks = (tile == CONV_TILE_WINO_B3U ? conv_wino_b3u_blocks(p, nullptr) : tile == CONV_TILE_WINO_B3S ? conv_wino_b3s_blocks(p, nullptr) : (long)((p.M / 4 + 63) / 64)) * (long)conv_wino_rows(p.Cout_store) / 64;
        const int KT = p.Cin / 16, min_steps = 4;
        const int min_blocks = p.split_target > 0 ? p.split_target : 256;
        const int target = p.split_target > 0 ? p.split_target : 512;
        if (blocks >= min_blocks || KT < 2 * min_steps) return 0;
        int want = (int)((target + blocks - 1) / blocks);
        int ks = want < KT / min_steps ? want : KT / min_steps;
        if (ks < 2) return 0;
        p.kt_per_split = (KT + ks - 1) / ks;
        p.ksplit = (KT + p.kt_per_split - 1) / p.kt_per_split;
        if (p.ksplit < 2) { p.ksplit = 1; p.kt_per_split = 0; return 0; }
        return (size_t)p.ksplit * p.M * p.Cout_store * sizeof(float);
    }
    tile_dims(tile, bm, bn);
    const long classes = p.deconv2x ? 4 : 1;
    const long blocks = classes * ((p.M + bm - 1) / bm) * ((p.Cout_store + bn - 1) / bn);
    const int KT = p.K_pad / conv_tile_bk(tile);
    const int min_blocks = p.split_target > 0 ? p.split_target : 256;
    const int target = p.split_target > 0 ? p.split_target : 768;
    const int min_steps = 4;
    if (blocks >= min_blocks || KT < 2 * min_steps) return 0;
    int want = (int)((target + blocks - 1) / blocks);
    int ks = want < KT / min_steps ? want : KT / min_steps;
    if (ks < 2) return 0;
    p.kt_per_split = (KT + ks - 1) / ks;
    // the fp16 form of conv_b3d.hip works in stages of up to four half steps (two K steps): a split-K range ends on a stage boundary
    if (tile >= CONV_TILE_B3D && tile < CONV_TILE_B3D + CONV_TILE_B3D_N && p.f16 == 1) p.kt_per_split = (p.kt_per_split + 1) / 2 * 2;
    p.ksplit = (KT + p.kt_per_split - 1) / p.kt_per_split;
    if (p.ksplit < 2) { p.ksplit = 1; p.kt_per_split = 0; return 0; }
    return (size_t)p.ksplit * classes * p.M * p.Cout_store * sizeof(float);
}

hipError_t launch_conv_igemm(const ConvParams& p, hipStream_t st)
{
    // tile ids: 0-4 = geometry {128x128, 128x64, 64x128, 64x64, 128x32} with the plain schedule;
    // 5-9 the same geometries with the software-pipelined schedule (MID = 2);
    // +10 = 8-wave / BK-64 experiments (same geometry order)
    if (p.narrow) return launch_conv_narrow(p, st);
    if (p.force_tile == CONV_TILE_WINO) return launch_conv_wino(p, st);
    if (p.force_tile == CONV_TILE_WINO_B3 || p.force_tile == CONV_TILE_WINO_B3U || p.force_tile == CONV_TILE_WINO_B3S) {
        ConvParams q = p;
        if (p.wubh && !p.f16) {      // the fp16x2 form: two half planes of U, three products
            q.wub = p.wubh;
            q.wub_bytes = p.wub_bytes / 3 * 2;
            q.scale = p.scale_h2w;
            q.xr = p.xr_slot;
            q.f16 = 3;
        }
        if (p.force_tile == CONV_TILE_WINO_B3S) return launch_conv_wino_b3s(q, st);
        return launch_conv_wino_b3(q, st, p.force_tile == CONV_TILE_WINO_B3U);
    }
    if (p.force_tile == CONV_TILE_STEM) return launch_conv_stem(p, st);
    if (p.force_tile == CONV_TILE_STEM_B3) {
        if (p.wstemh && !p.f16) {      // the fp16x2 form
            ConvParams q = p;
            q.wstemb = p.wstemh;
            q.scale = p.scale_h2s;
            q.xr = p.xr_slot;
            q.f16 = 3;
            return launch_conv_stem_b3(q, st);
        }
        return launch_conv_stem_b3(p, st);
    }
    if (p.force_tile == CONV_TILE_WS) return launch_conv_ws(p, st);
    if (p.force_tile == CONV_TILE_HALO) {      // fp16x2 form only: the two half planes of conv_b3r.hip
        if (p.f16 || !p.wh2r || p.x_half || p.y_half || p.res_half) return hipErrorInvalidValue;
        ConvParams q = p;
        q.w = static_cast<const float*>(p.wh2r);
        q.w_bytes = p.w_bytes / 2;
        q.scale = p.scale_h2;
        q.xr = p.xr_slot;
        q.f16 = 3;
        return launch_conv_halo(q, st);
    }
    const int b3d_tile = (p.force_tile < 0 && (p.x_half || p.y_half || p.res_half)) ? conv_pick_tile(p) : p.force_tile;
    if (b3d_tile >= CONV_TILE_B3D && b3d_tile < CONV_TILE_B3D + CONV_TILE_B3D_N) {
        // conv_b3d.hip: the one-plane fp16 form of an f16-mode layer, both operands by LDS-DMA (fp32 / half views)
        if (!p.wb3r || p.f16 != 1) return hipErrorInvalidValue;
        ConvParams q = p;
        q.w = static_cast<const float*>(p.wb3r);
        return launch_conv_b3d(q, b3d_tile, st);
    }
    if (p.x_half || p.y_half || p.res_half) return hipErrorInvalidValue;      // no other kernel reads or writes half views
    if (p.force_tile >= CONV_TILE_B3R && p.force_tile < CONV_TILE_B3R + 6 && p.f16 == 1) {
        // fp16-MFMA mode on the staging of conv_b3r.hip: one plane of half-rounded weights in MFMA fragment order
        if (!p.wb3r) return hipErrorInvalidValue;
        ConvParams q = p;
        q.w = static_cast<const float*>(p.wb3r);
        return launch_conv_b3r(q, p.force_tile, st);
    }
    if (((p.force_tile >= CONV_TILE_B3R && p.force_tile < CONV_TILE_B3R + 6) || (p.force_tile >= 90 && p.force_tile <= 96)) && !p.f16 && p.wh2r) {      // (90-96: ablations, diagnostics build)
        // the fp16x2 form of the same staging: two half planes per operand, three products
        ConvParams q = p;
        q.w = static_cast<const float*>(p.wh2r);
        q.w_bytes = p.w_bytes / 2;
        q.scale = p.scale_h2;
        q.xr = p.xr_slot;
        q.f16 = 3;
        return launch_conv_b3r(q, p.force_tile, st);
    }
    if (((p.force_tile >= CONV_TILE_B3R && p.force_tile < CONV_TILE_B3R + 6) || (p.force_tile >= 90 && p.force_tile <= 96)) && !p.f16) {
        if (!p.wb3r) return hipErrorInvalidValue;
        ConvParams q = p;
        q.w = static_cast<const float*>(p.wb3r);
        q.w_bytes = p.w_bytes / 2;               // one bf16 plane of one parity class
        q.f16 = 2;
        return launch_conv_b3r(q, p.force_tile, st);
    }
    if (p.force_tile >= CONV_TILE_B3 && p.force_tile < CONV_TILE_B3 + 6 && !p.f16) {
        // an fp32 layer on the bf16 matrix cores: same inputs and outputs, the products taken as 3 x bf16 splits
        if (!p.wb3) return hipErrorInvalidValue;
        ConvParams q = p;
        q.w = static_cast<const float*>(p.wb3);
        q.w_bytes = p.w_bytes / 2;               // one bf16 plane of one parity class
        q.f16 = 2;
        switch (p.force_tile - CONV_TILE_B3) {
            case 0: return launch_b3<128, 128, 2, 2>(q, st);
            case 1: return launch_b3<128, 64, 2, 2>(q, st);
            case 2: return launch_b3<64, 128, 2, 2>(q, st);
            case 3: return launch_b3<64, 64, 2, 2>(q, st);
            case 5: return launch_b3<256, 128, 4, 2>(q, st);     // 92 KB of LDS: one block per CU, 1.5x the flops per byte fetched
            default: return launch_b3<128, 128, 2, 4>(q, st);
        }
    }
    if (p.f16 == 2) {      // fp32 split into three bf16 terms on the bf16 matrix cores: the same geometry ids
        switch (conv_pick_tile(p)) {
            case 0: case 5: return launch_b3<128, 128, 2, 2>(p, st);
            case 1: case 6: return launch_b3<128, 64, 2, 2>(p, st);
            case 2: case 7: return launch_b3<64, 128, 2, 2>(p, st);
            case 10: return launch_b3<128, 128, 2, 4>(p, st);
            default: return launch_b3<64, 64, 2, 2>(p, st);
        }
    }
    if (p.f16) {      // fp16-MFMA path: geometry ids 0-4 / 10-12 map onto the same tile shapes
        switch (conv_pick_tile(p)) {
            case 0: case 5: return launch_f16<128, 128, 2, 2>(p, st);
            case 1: case 6: return launch_f16<128, 64, 2, 2>(p, st);
            case 2: case 7: return launch_f16<64, 128, 2, 2>(p, st);
            case 10: return launch_f16<128, 128, 2, 4>(p, st);
            default: return launch_f16<64, 64, 2, 2>(p, st);   // incl. the narrow-N geometries
        }
    }
    const int tile = conv_pick_tile(p);
    if (p.K_pad % conv_tile_bk(tile)) return hipErrorInvalidValue;
    if (tile >= 16 && tile <= 19) {
        if (p.Cin % 32) return hipErrorInvalidValue;   // DMA variants need wave-uniform taps per K step
        switch (tile) {
            case 16: return launch_dma<128, 128, 2, 4, 3>(p, st);   // 8 waves, 3-stage ring (96 KB)
            case 17: return launch_dma<64, 64, 2, 2, 3>(p, st);     // 4 waves, 3 stages (48 KB)
            case 18: return launch_dma<128, 128, 2, 2, 2>(p, st);   // 4 waves, 2 stages (64 KB)
            default: return launch_dma<64, 64, 2, 2, 4>(p, st);     // 19: 4 stages (64 KB)
        }
    }
    switch (tile) {
        case 0: return launch_cfg<128, 128, 2, 2, 32, 0>(p, st);
        case 1: return launch_cfg<128, 64, 2, 2, 32, 0>(p, st);
        case 2: return launch_cfg<64, 128, 2, 2, 32, 0>(p, st);
        case 3: return launch_cfg<64, 64, 2, 2, 32, 0>(p, st);
        case 4: return launch_cfg<128, 32, 4, 1, 32, 0>(p, st);
        case 5: return launch_cfg<128, 128, 2, 2, 32, 2>(p, st);
        case 6: return launch_cfg<128, 64, 2, 2, 32, 2>(p, st);
        case 7: return launch_cfg<64, 128, 2, 2, 32, 2>(p, st);
        case 8: return launch_cfg<64, 64, 2, 2, 32, 2>(p, st);
        case 9: return launch_cfg<128, 32, 4, 1, 32, 2>(p, st);
        case 10: return launch_cfg<128, 128, 2, 4, 32, 2>(p, st);   // 8 waves, wave tile 64x32
        case 11: return launch_cfg<128, 64, 4, 2, 32, 2>(p, st);    // 8 waves, wave tile 32x32
        case 12: return launch_cfg<64, 128, 2, 4, 32, 2>(p, st);    // 8 waves, wave tile 32x32
        case 13: return launch_cfg<64, 64, 2, 2, 64, 2>(p, st);     // BK 64
        case 14: return launch_cfg<256, 128, 4, 2, 32, 2>(p, st);   // 8 waves, wave tile 64x64 (110 KB LDS, 1 block/CU)
        case 15: return launch_cfg<128, 128, 2, 2, 16, 0>(p, st);   // BK 16: half the LDS, 4 blocks/CU
        case 31: return launch_cfg<64, 64, 2, 2, 32, 3>(p, st);     // 31-35: deep-prefetch schedule (MID = 3)
        case 32: return launch_cfg<128, 128, 2, 4, 32, 3>(p, st);
        case 33: return launch_cfg<64, 128, 2, 4, 32, 3>(p, st);
        case 34: return launch_cfg<128, 64, 4, 2, 32, 3>(p, st);
        case 35: return launch_cfg<128, 32, 4, 1, 32, 3>(p, st);
#ifdef ACCEL_CONV_DIAG
        // timing-only ablation variants (they skip loads / stores / barriers and compute WRONG results): compiled into
        // the diagnostics build alone (`make -C accel_amd/csrc diag`, scripts/microbench), never into libaccel_hip.so
        case 20: return launch_cfg<128, 128, 2, 2, 32, 0, 1>(p, st);   // ablation: no loads
        case 21: return launch_cfg<128, 128, 2, 2, 32, 0, 3>(p, st);   // ablation: no loads, no barrier
        case 22: return launch_cfg<128, 128, 2, 2, 32, 0, 2>(p, st);   // ablation: no barrier (racy, timing only)
        case 23: return launch_cfg<64, 64, 2, 2, 32, 0, 3>(p, st);
        case 24: return launch_cfg<128, 128, 2, 2, 32, 0, 4>(p, st);   // ablation: loads but no LDS stores
        case 25: return launch_cfg<128, 128, 2, 2, 32, 0, 8>(p, st);   // ablation: LDS stores but no loads
        case 26: return launch_cfg<64, 64, 2, 2, 32, 0, 4>(p, st);
        case 27: return launch_cfg<64, 64, 2, 2, 32, 0, 8>(p, st);
        case 28: return launch_cfg<64, 64, 2, 2, 32, 0, 1>(p, st);
        case 29: return launch_cfg<128, 128, 2, 2, 32, 0, 16>(p, st);  // setprio experiment
        case 30: return launch_cfg<64, 64, 2, 2, 32, 0, 16>(p, st);
#endif
        default: return hipErrorInvalidValue;
    }
}
