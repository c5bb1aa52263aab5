// Winograd F(2x2, 3x3) convolution on the gfx950 fp32 matrix cores, fully fused:
// input transform in the loader, 16 position-GEMMs on v_mfma_f32_32x32x2_f32, output transform + the usual epilogue
// (scale/shift, residual, activation, dual output) from the accumulators.  For 3x3 / stride 1 / dilation 1 / pad 1
// layers (35 % of the Accel-18 step: res2-4 `branch2b`, the ResNet-18 trunk, FlowNet conv*_1) it executes 16 instead
// of 36 multiply-adds per 2x2 output tile and channel pair -- 2.25x fewer matrix-core cycles than the direct
// implicit GEMM, which already runs those layers at ~0.9 of the fp32 MFMA peak.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A       per 2x2 output tile, summed over input channels
//   B^T = |1 0 -1 0|   G = |1   0   0 |   A^T = |1 1  1  0|
//         |0 1  1 0|       |.5  .5  .5|         |0 1 -1 -1|
//         |0 -1 1 0|       |.5 -.5  .5|
//         |0 1 0 -1|       |0   0   1 |
//
// Work split.  A block (512 threads = 8 wavefronts, two per SIMD) owns 64 output tiles (= 256 pixels) x 64 output
// channels and ALL 16 Winograd positions of them: the output transform needs the 16 positions of one (tile, channel)
// in one lane, so each wavefront keeps, for its 32 tiles x 16 channels, 16 positions x two 16x16 accumulator tiles of
// v_mfma_f32_16x16x4_f32 = 128 accumulator registers -- half the 256-register budget of a wavefront that shares its
// SIMD with one partner.  (A first version used 32x32x2 tiles with 256 accumulators and ONE wavefront per SIMD: every
// LDS wait and every barrier then idles the matrix pipe; 58 % MFMA-busy against this version's two-wavefront overlap.)
//
// K loop, 8 input channels per step, double-buffered LDS (2 x 66 KB):
//   * weights U = G g G^T are transformed on the host and stored [C/8][16][rows][8]: per step the block's slice is 16
//     chunks of 2 KB, brought in by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass);
//   * the input transform is spread over (tile, channel quad, patch column): a thread loads the 4 rows of ONE column
//     of a 4x4 patch (4 x 16 B), applies B^T down the column, and gets the row combination from its quad neighbours
//     with DPP quad permutes (the 4 lanes of a quad hold the 4 columns of a patch) -- no LDS round trip for the raw
//     patch; the 4 transformed float4 go to V[position][tile][channel];
//   * both LDS images are unpadded 32-byte rows; fragment reads (ds_read_b128, lanes 0-31 rows, lane>>5 the channel
//     quad) are conflict-free through a one-bit XOR swizzle: physical quad slot = quad ^ ((row >> 3) & 1), applied on
//     the store side for V and on the SOURCE side of the DMA for U; the V image has 32 B of slack per position so
//     that the four patch columns of a quad hit different banks on the store side.
//   * per position three ds_read_b64 (two tile fragments, one weight fragment: lane>>4 selects a channel pair) feed
//     four MFMAs; the XOR swizzle above is conflict-free for them too (it moves whole quads, i.e. pairs of float2 slots).
//
// Numerics: fp32 throughout; U is computed in double and rounded once.  F(2x2,3x3) has transform matrices with
// entries 0, +-1, +-1/2 only: the error is that of direct convolution times a small constant (measured against the
// oracle in tests/test_ops_gpu.py at the op tolerance 1e-4).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "conv_common.h"
#include "range.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_void_ptr;

namespace {
constexpr int TT = 64;                 // output tiles (2x2 pixels each) per block
constexpr int KK = 64;                 // output channels per block
constexpr int BKC = 8;                 // input channels per K step
constexpr int MS = 2;                  // K steps per unrolled macro step = depth of the patch-register ring
constexpr int VPS = TT * BKC + 8;      // floats per position of the V image (+32 B: see header)
constexpr int UPS = KK * BKC;          // floats per position of the U image
constexpr int VSTAGE = 16 * VPS, USTAGE = 16 * UPS;
constexpr size_t WINO_LDS = (size_t)2 * (VSTAGE + USTAGE) * sizeof(float);

__device__ __forceinline__ float quad_2211(float v)
{
    // lane j of every quad receives the value of lane {2, 2, 1, 1}[j]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xF, 0xF, true));
}

}  // namespace

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void conv_wino_f32_kernel(ConvParams p)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtin only type-checks in the device pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                   // [2][16][VPS]
    float* Us = smem + 2 * VSTAGE;      // [2][16][UPS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;          // 2 tile halves x 4 channel groups of 16

    // XCD-aware block order (blocks b, b+8, .. share an XCD): contiguous runs of tiles per XCD, channel blocks fastest,
    // so the input patches of a tile block are re-read from that XCD's L2 by its channel-block neighbours
    const int nblk = p.MT * p.NT, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int nt = swz % p.NT, mt = swz / p.NT;
    const int m0 = mt * TT, n0 = nt * KK;

    const int TH = p.Ho >> 1, TW = p.Wo >> 1, THW = TH * TW;
    const int T = p.wino_T;
    const int kb = p.ksplit > 1 ? (int)blockIdx.y * p.kt_per_split : 0;       // first K step of this block
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ur = make_rsrc(p.wu, p.wu_bytes);

    // ---- loader coordinates: item = (tile, channel quad q, patch column j), one per thread ----
    const int j = tid & 3, q = (tid >> 2) & 1, tl = tid >> 3;
    unsigned a_off[4];
    {
        const int tg = m0 + tl;
        const bool ok = tg < T;
        const int tt = ok ? tg : 0;
        const int n = tt / THW, rem = tt - n * THW;
        const int ty = rem / TW, tx = rem - ty * TW;
        const int ix = 2 * tx - 1 + j;
        const bool okx = ok && (unsigned)ix < (unsigned)p.W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * ty - 1 + r;
            a_off[r] = (okx && (unsigned)iy < (unsigned)p.H)
                           ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.xCs + 4 * q) * 4) : OOB;
        }
    }
    const int v_dst = j * VPS + tl * BKC + ((q ^ ((tl >> 3) & 1)) << 2);
    // V[i][j] of a patch from the column-transformed t[i][0..3]: lane j combines its own value with one neighbour's
    //   j=0: t0 - t2   j=1: t1 + t2   j=2: t2 - t1   j=3: t3 - t1 (= -V[i][3])        (neighbour = lane {2,2,1,1}[j] of the quad)
    // column 3 is stored NEGATED -- own minus neighbour like columns 0 and 2 -- and the host negates U at the positions
    // 4i+3 to match (conv_wino_pack): one multiply-add per element, no per-lane sign on the own value.
    const float sb = j == 1 ? 1.f : -1.f;

    // ---- weight DMA: 32 instructions of 1 KB per K step, 4 per wavefront; lane -> (channel row, physical quad slot) ----
    unsigned u_off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = wave * 4 + e, pp = idx >> 1, hh = idx & 1;
        const int row = hh * 32 + (lane >> 1), slot = lane & 1;
        const int quad = slot ^ ((row >> 3) & 1);
        u_off[e] = (unsigned)((((size_t)pp * p.wino_rows + n0 + row) * BKC + quad * 4) * 4);
    }
    const unsigned u_step = (unsigned)((size_t)16 * p.wino_rows * BKC * 4);      // bytes between K steps

    // Patch registers: a ring of 2 K steps.  Step k issues the 4 loads of step k+2 behind its first MFMAs; they are consumed
    // by the transform that runs under the second half of step k+1 -- one and a half K steps (~4 us) of lead.  With half a
    // step of lead, as in a first version, the transform stalled on these loads for 20 % of the kernel (ablation: 345 us
    // with, 272 us without the transform consuming them; res4 branch2b x 8 clips).
    f32x4 d[2][4];
    auto load_ring = [&](int kstep_, int set_, int r) {      // patch row r of K step `kstep_` into ring set `set_` (= kstep_ & 1;
        d[set_][r] = buf_load4(xr, a_off[r] != OOB ? a_off[r] + (unsigned)(kb + kstep_) * (BKC * 4) : OOB);     // a constant after unrolling)
    };
    auto issue_u_one = [&](int k, int buf, int e) {   // one of this wavefront's 4 weight DMAs of step k (1 KB each)
        const int idx = wave * 4 + e, pp = idx >> 1, hh = idx & 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (lds_void_ptr)(Us + buf * USTAGE + pp * UPS + hh * 256), 16,
                                                 u_off[e], (unsigned)(kb + k) * u_step, 0, 0);
    };
    f32x4 vo;                                          // the float4 of V being assembled
    auto transform_one = [&](int buf, int ms, int piece) {    // ms = ring set; piece = (patch row i of V, channel c): 16 per step
        const int i = piece >> 2, c = piece & 3;
        // B^T down the patch column, one channel:  t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3
        const float t = i == 0 ? d[ms][0][c] - d[ms][2][c] : i == 1 ? d[ms][1][c] + d[ms][2][c]
                      : i == 2 ? d[ms][2][c] - d[ms][1][c] : d[ms][1][c] - d[ms][3][c];
        vo[c] = fmaf(sb, quad_2211(t), t);
        if (c == 3) *reinterpret_cast<f32x4*>(Vs + buf * VSTAGE + v_dst + i * 4 * VPS) = vo;
    };
    auto wait_all_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto wait_dma_barrier = [&]() { asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // all but the 4 youngest (patch loads)

    f32x4 acc[16][2];
#pragma unroll
    for (int pp = 0; pp < 16; ++pp)
#pragma unroll
        for (int sI = 0; sI < 2; ++sI)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[pp][sI][e] = 0.f;

    // fragment addressing of v_mfma_f32_16x16x4_f32: lanes 0-15 = 16 rows, lane >> 4 = one of 4 k; a lane reads a channel
    // PAIR (float2) of its row: .x feeds the MFMA over channels {0,2,4,6}, .y the one over {1,3,5,7} (A and B alike)
    const int frow = lane & 15, kq = lane >> 4;
    auto frag_off = [&](int row) { return row * BKC + (((kq >> 1) ^ ((row >> 3) & 1)) << 2) + ((kq & 1) << 1); };
    int a_base0 = frag_off(wm * 32 + frow), a_base1 = frag_off(wm * 32 + 16 + frow);
    // the two tile fragments sit 512 B apart; left to itself the compiler fuses their reads into ds_read2st64_b64, which
    // is serviced in 16-lane groups over 32 banks (this layout: 2-way conflicts, measured 41 % of the LDS cycles) at
    // half the bandwidth of two plain ds_read_b64.  An opaque copy hides the relation between the two offsets.
    asm volatile("" : "+v"(a_base1));
    const int b_base = frag_off(wn * 16 + frow);

    // One K step = 64 MFMAs (16 positions x 2 tile fragments x 2 channel groups) on LDS stage `cur`.  What prepares step
    // k+1 is placed BETWEEN the MFMAs, a few instructions behind each one: the 4 patch loads and the 4 weight DMAs behind
    // the first 8 (most of a step ahead of their first use), the 16 transform pieces (3 VALU each, one LDS store per 4)
    // behind MFMAs 32..47.  Scheduling fences pin that order; fragments of position p+1 are read before the MFMAs of
    // position p.  The partner wavefront of the SIMD fills the matrix pipe whenever this one waits.
    // K2 = k & 1 (compile time: the loop is unrolled by 2): LDS stage K2, transform of step k+1 from ring set K2 ^ 1, the 4
    // patch loads of step k+2 into ring set K2 (consumed at step k-1).
    auto kstep = [&](int k, auto k2_, auto pipe) {
        constexpr bool PIPE = decltype(pipe)::value;
        constexpr int K2 = decltype(k2_)::value, cur = K2;
        constexpr int NL = 4;                           // patch-load slots of this step
        const float* va = Vs + cur * VSTAGE;
        const float* ub = Us + cur * USTAGE + b_base;
        f32x2 fa0[2], fa1[2], fb[2];
        fa0[0] = *reinterpret_cast<const f32x2*>(va + a_base0);
        fa1[0] = *reinterpret_cast<const f32x2*>(va + a_base1);
        fb[0] = *reinterpret_cast<const f32x2*>(ub);
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const int c = pp & 1;
            if (pp + 1 < 16) {
                fa0[c ^ 1] = *reinterpret_cast<const f32x2*>(va + (pp + 1) * VPS + a_base0);
                fa1[c ^ 1] = *reinterpret_cast<const f32x2*>(va + (pp + 1) * VPS + a_base1);
                fb[c ^ 1] = *reinterpret_cast<const f32x2*>(ub + (pp + 1) * UPS);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sI = r & 1, kg = r >> 1;      // alternate the two accumulators: no back-to-back dependency
                const float av = sI ? fa1[c][kg] : fa0[c][kg];
                acc[pp][sI] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[c][kg], av, acc[pp][sI], 0, 0, 0);
                if (PIPE) {
                    const int slot = pp * 4 + r;
                    // weight DMAs first, patch loads behind them: the barrier at the end of the step then waits for the DMAs
                    // only (vmcnt(NL): the counter retires in order) and the patch loads keep their 1.5 steps of lead
                    if (slot < 4) issue_u_one(k + 1, cur ^ 1, slot);
                    else if (slot < 4 + NL) load_ring(k + 2, K2, slot - 4);
                    else if (slot >= 32 && slot < 48) transform_one(cur ^ 1, K2 ^ 1, slot - 32);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // split-K: blockIdx.y owns K steps [kb, kb + nk) (nk even), and leaves raw partial outputs in the workspace
    const int nk_all = p.Cin / BKC;       // a multiple of 2 (conv_wino_eligible)
    const int nk = p.ksplit > 1 ? min(p.kt_per_split, nk_all - kb) : nk_all;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        load_ring(0, 0, r);
        load_ring(1, 1, r);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) issue_u_one(0, 0, e);
#pragma unroll
    for (int pc = 0; pc < 16; ++pc) transform_one(0, 0, pc);
    wait_all_barrier();
    typedef std::true_type PIPE_T;
    typedef std::false_type LAST_T;
#define WINO_STEP(s, pipe_t, wait)                                             \
    do {                                                                       \
        kstep(k + (s), std::integral_constant<int, (s)>(), pipe_t());          \
        if (pipe_t::value) wait();                                             \
    } while (0)
    int k = 0;
    for (; k + 2 < nk; k += 2) {
        WINO_STEP(0, PIPE_T, wait_dma_barrier); WINO_STEP(1, PIPE_T, wait_dma_barrier);
    }
    // The patch loads of the step before last feed nobody (there is no step nk): the compiler deletes them, and a counted
    // wait that assumes them would let that step's weight DMAs be the "4 youngest" and stay in flight -- wait for everything.
    WINO_STEP(0, PIPE_T, wait_all_barrier); WINO_STEP(1, LAST_T, wait_all_barrier);
#undef WINO_STEP

    // ---- output transform + fused epilogue ---------------------------------------------------------------------
    // The MFMAs take the weight fragment as the A operand, so an accumulator tile is transposed: col = lane & 15 is the
    // TILE, row = 4*(lane>>4) + e the output channel.  A lane owns two tiles (one per fragment) and 4 consecutive
    // channels: two coordinate decodes per lane, 16-byte residual loads and stores.
    const int co = n0 + wn * 16 + 4 * kq;
    const bool cok = co < p.Cout_store;
    const int cc = cok ? co : 0;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + cc), sf = *reinterpret_cast<const f32x4*>(p.shift + cc);
    f32x4 sc2 = {1.f, 1.f, 1.f, 1.f}, sf2 = {0.f, 0.f, 0.f, 0.f};
    if (p.y2) { sc2 = *reinterpret_cast<const f32x4*>(p.scale2 + cc); sf2 = *reinterpret_cast<const f32x4*>(p.shift2 + cc); }
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res : p.y, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t y2r = make_rsrc(p.y2 ? p.y2 : p.y, p.y2 ? p.y2_bytes : 0u);
    const float inv_thw = 1.0f / (float)THW, inv_tw = 1.0f / (float)TW;
    // Vector-memory operations retire in order through one counter, so a residual load issued behind an output store
    // waits for that store's acknowledgement: ALL residual values are fetched before the first store.
    bool okS[2];
    unsigned pixS[2][4];
    f32x4 rv[2][4];
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
        const int tg = m0 + wm * 32 + sI * 16 + frow;
        int n, rem, ty, tx;
        divmod_small(tg < T ? tg : 0, THW, inv_thw, n, rem);
        divmod_small(rem, TW, inv_tw, ty, tx);
        okS[sI] = cok && tg < T;
        const unsigned pix00 = (unsigned)((n * p.Ho + 2 * ty) * p.Wo + 2 * tx);
        pixS[sI][0] = pix00; pixS[sI][1] = pix00 + 1; pixS[sI][2] = pix00 + (unsigned)p.Wo; pixS[sI][3] = pix00 + (unsigned)p.Wo + 1;
        if (p.res && p.ksplit <= 1) {
#pragma unroll
            for (int o = 0; o < 4; ++o) rv[sI][o] = buf_load4(rr, okS[sI] ? (pixS[sI][o] * p.resCs + co) * 4u : OOB);
        }
    }
    if (p.ksplit > 1) {
        // raw partial outputs of this K slice: ws[split][pixel][Cout_store]; scale/shift, residual and activation
        // are applied by the reduce kernel after summing the slices
        const size_t slab = (size_t)blockIdx.y * p.M * p.Cout_store;
        const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.ws + slab, (unsigned)((size_t)p.M * p.Cout_store * 4));
#pragma unroll
        for (int sI = 0; sI < 2; ++sI) {
            f32x4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s0[4], s1[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    s0[jj] = acc[jj][sI][e] + acc[4 + jj][sI][e] + acc[8 + jj][sI][e];
                    s1[jj] = acc[4 + jj][sI][e] - acc[8 + jj][sI][e] - acc[12 + jj][sI][e];
                }
                v[0][e] = s0[0] + s0[1] + s0[2];
                v[1][e] = s0[1] - s0[2] - s0[3];
                v[2][e] = s1[0] + s1[1] + s1[2];
                v[3][e] = s1[1] - s1[2] - s1[3];
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) buf_store4(wr, okS[sI] ? (pixS[sI][o] * p.Cout_store + co) * 4u : OOB, v[o]);
        }
        return;
    }
    unsigned rmax = 0u, rmax2 = 0u;
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
        const bool ok = okS[sI];
        f32x4 v[4];                          // [output pixel of the 2x2 tile][channel]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s0[4], s1[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                s0[jj] = acc[jj][sI][e] + acc[4 + jj][sI][e] + acc[8 + jj][sI][e];
                s1[jj] = acc[4 + jj][sI][e] - acc[8 + jj][sI][e] - acc[12 + jj][sI][e];
            }
            // column 3 of the transformed input was stored negated and U negated to match, so M[.][3] is the true
            // value: the output transform below is the textbook A^T M A
            v[0][e] = (s0[0] + s0[1] + s0[2]) * sc[e] + sf[e];
            v[1][e] = (s0[1] - s0[2] - s0[3]) * sc[e] + sf[e];
            v[2][e] = (s1[0] + s1[1] + s1[2]) * sc[e] + sf[e];
            v[3][e] = (s1[1] - s1[2] - s1[3]) * sc[e] + sf[e];
        }
        if (p.res) {
#pragma unroll
            for (int o = 0; o < 4; ++o) v[o] += rv[sI][o];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.act == 1) v[o][e] = fmaxf(v[o][e], 0.f);
                else if (p.act == 2) v[o][e] = v[o][e] > 0.f ? v[o][e] : v[o][e] * p.slope;
            }
            buf_store4(yr, ok ? (pixS[sI][o] * p.yCs + co) * 4u : OOB, v[o]);
            if (p.yr && ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(v[o][e]); rmax = b > rmax ? b : rmax; }
            }
        }
        if (p.y2) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                f32x4 u;
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = fmaxf(v[o][e] * sc2[e] + sf2[e], 0.f);
                buf_store4(y2r, ok ? (pixS[sI][o] * p.y2Cs + co) * 4u : OOB, u);
                if (p.y2r && ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(u[e]); rmax2 = b > rmax2 ? b : rmax2; }
                }
            }
        }
    }
    // range slots of the outputs (range.h)
    const unsigned key = (unsigned)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (p.yr) range_note_wave(p.yr, rmax, key);
    if (p.y2 && p.y2r) range_note_wave(p.y2r, rmax2, key);
#endif
}

// shapes the Winograd kernel takes: 3x3, stride 1, dilation 1, pad 1, even output size, channels in multiples of 8
bool conv_wino_eligible(const ConvParams& p)
{
    return !p.deconv2x && (!p.f16 || p.f16 == 3) && !p.narrow && p.kh == 3 && p.kw == 3 && p.sh == 1 && p.sw == 1 && p.dh == 1 && p.dw == 1 &&
           p.ph == 1 && p.pw == 1 && p.Cin % (BKC * MS) == 0 && p.Ho == p.H && p.Wo == p.W && !(p.Ho & 1) && !(p.Wo & 1) && p.Cin >= 16;
}

int conv_wino_rows(int cout_store) { return (cout_store + KK - 1) / KK * KK; }

// U = G g G^T of an OIHW 3x3 weight, laid out [C/8][16][rows][8] (C padded to `cin_pad`, rows = conv_wino_rows)
void conv_wino_pack(const float* w, int Cout, int Cin, int cin_pad, int rows, float* out)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int k = 0; k < Cout; ++k)
        for (int c = 0; c < Cin; ++c) {
            const float* g = w + ((size_t)k * Cin + c) * 9;
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) tmp[i][b] = G[i][0] * g[0 * 3 + b] + G[i][1] * g[1 * 3 + b] + G[i][2] * g[2 * 3 + b];
            for (int i = 0; i < 4; ++i)
                for (int jj = 0; jj < 4; ++jj) {
                    double u = tmp[i][0] * G[jj][0] + tmp[i][1] * G[jj][1] + tmp[i][2] * G[jj][2];
                    if (jj == 3) u = -u;      // the kernel stores column 3 of the transformed input negated
                    out[((((size_t)(c / BKC) * 16 + (i * 4 + jj)) * rows + k) * BKC) + (c % BKC)] = (float)u;
                }
        }
    (void)cin_pad;
}

hipError_t launch_conv_wino(const ConvParams& p0, hipStream_t st)
{
    ConvParams p = p0;
    if (!conv_wino_eligible(p) || !p.wu) return hipErrorInvalidValue;
    p.wino_T = p.M / 4;
    p.MT = (p.wino_T + TT - 1) / TT;
    p.NT = p.wino_rows / KK;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_f32_kernel), WINO_LDS); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_wino_f32_kernel, dim3(p.MT * p.NT, p.ksplit > 1 ? p.ksplit : 1), dim3(512), WINO_LDS, st, p);
    if (p.ksplit > 1) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_splitk_reduce(p, 1, st);
    }
    return hipGetLastError();
}
