// Weight-stationary streaming kernel for the 1x1 / stride-1 "expand" convolutions of the high-resolution ResNet stages
// (res2* branch1 / branch2c: 64 -> 256 channels on 256x512 maps, res3* branch2c: 128 -> 512 on 128x256;
// resnet_v1_101_flownet_deeplab.py:583-640): K = Cin is 64 or 128, so a block of the implicit GEMM does 2-4 K steps
// between a cold prologue and a 64 KB epilogue, nothing overlaps inside a block, and the layers ran at 2.7-3.9 TB/s
// of an HBM-bound budget.  Here
//   * a block is PERSISTENT (one per CU, one wavefront per SIMD) and owns BN output channels: their K x BN weights are
//     copied into LDS once (64 KB, host-packed in the LDS image) and stay there while the block streams pixel tiles;
//   * the pixel tile of the NEXT iteration travels global -> LDS by DMA (no registers) while this one is multiplied,
//     and the residual of THIS tile is fetched into registers before the MFMA loop, so the epilogue finds it there;
//   * LDS layout [K/4][pixel][4]: a lane's operand for 4 consecutive K steps is ONE ds_read_b128 (pixels / channels
//     16 bytes apart: conflict-free); K is split between the two lane halves of v_mfma_f32_32x32x2_f32 by HALVES
//     (lanes 0-31 take k, lanes 32-63 take K/2 + k) -- any split works as long as both operands use the same one;
//   * residual and output move as whole 128-byte lines in 16-byte accesses (through a small LDS transposition of the
//     MFMA result): 72 vector-memory operations per tile and wavefront.  Dword accesses (one channel per lane, no
//     transposition) are 264 -- more than the 64 a wavefront can have in flight, so it waits to ISSUE them instead of
//     multiplying (564 us on res2 branch2c) -- and 16-byte accesses straight from the MFMA layout touch 64 different
//     lines per instruction (652 us);
// Blocks that share pixel tiles (several column groups when Cout > BN) are placed on the same XCD and walk the tiles
// in the same order, so the input is fetched from HBM once and re-read from that XCD's L2.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "conv_common.h"
#include "range.h"

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// 16 bytes per lane, global -> LDS (lane i lands at lds + 16*i).  Kept out of the kernel template: with the builtin inside
// a template the host pass silently drops the kernel's instantiation (undefined __device_stub__ at load time).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* lds, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)lds, 16, voff, soff, 0, 0);
}

template <int K, int BM, int BN>
__global__ __launch_bounds__(256, 1) void conv1x1_ws_kernel(ConvParams p, int mtiles, int ngroups, int streams)
{
    constexpr int KQ = K / 4;                        // float4s per pixel / per output channel
    constexpr int WM = BM / 2, WN = BN / 2;          // wavefront tile
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int ASTAGE = KQ * BM * 4;              // floats per pixel-tile stage
    constexpr int NDMA = KQ * BM / 64 / 4;           // DMA instructions per wavefront and pixel tile (1 KB each)
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [K/4][BN][4] weights + 2 x [K/4][BM][4] pixels + 2 x [BN] scale, shift + 4 x [32][36] epilogue patches
    float* Ws = smem;
    float* As = smem + KQ * BN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // block -> (XCD, column group, tile stream): the column groups of one stream sit on one XCD (blocks are dealt to the
    // XCDs round-robin) and run concurrently
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int cg = slot % ngroups, stream = (slot / ngroups) * 8 + xcd;
    const int n0 = cg * BN;

    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res : p.y, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
    unsigned rmax = 0u;

    // ---- weights of this column group -> LDS (host layout = LDS image) ----
#pragma unroll
    for (int j = 0; j < KQ * BN / 64 / 4; ++j) {
        const int q = wave * (KQ * BN / 64 / 4) + j;
        dma16(wr, Ws + q * 256, (unsigned)(lane * 16), (unsigned)(((size_t)cg * KQ * BN + q * 64) * 16));
    }
    // per-lane byte offset of its 16 bytes inside a pixel tile, per DMA instruction: LDS float4 index q*64 + lane ->
    // (kq, pixel) = (idx / BM, idx % BM)
    unsigned a_off[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int idx = (wave * NDMA + j) * 64 + lane;
        a_off[j] = (unsigned)(((idx % BM) * p.xCs + (idx / BM) * 4) * 4);
    }
    auto issue_a = [&](int t, int stage) {
        const unsigned base = (unsigned)t * BM * p.xCs * 4;            // tiles past the end: out of range -> zeros
#pragma unroll
        for (int j = 0; j < NDMA; ++j)
            dma16(xr, As + stage * ASTAGE + (wave * NDMA + j) * 256, a_off[j], base);
    };
    float* ssc = As + 2 * ASTAGE;                   // [BN scale | BN shift] of this column group
    for (int i = tid; i < BN; i += 256) { ssc[i] = p.scale[n0 + i]; ssc[BN + i] = p.shift[n0 + i]; }
    const float floor_ = p.act == 1 ? 0.f : -__builtin_inff();

#ifdef WS_TIMING
    long long tacc[5] = {0, 0, 0, 0, 0};
#endif
    int t = stream, stage = 0;
    if (t < mtiles) issue_a(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; t < mtiles; t += streams, stage ^= 1) {
        const int m0 = t * BM;
#ifdef WS_TIMING
        const long long tq0 = clock64();
#endif
        if (t + streams < mtiles) issue_a(t + streams, stage ^ 1);      // its previous readers passed the barrier below
        // ---- residual of this tile -> registers, consumed after the MFMA loop, in the layout the output is stored in:
        // lane -> (pixel lane/8 of a group of 8, channel quad lane%8): one instruction moves 8 x 128 contiguous bytes
        f32x4 rv[MT][NT][4];
        unsigned ooff[MT];
        bool rowok[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pb = m0 + wm * WM + mt * 32 + (lane >> 3);
            ooff[mt] = (unsigned)(pb * p.yCs + n0 + wn * WN + 4 * (lane & 7)) * 4u;
#pragma unroll
            for (int i = 0; i < 4; ++i) rowok[mt][i] = pb + 8 * i < p.M;
            if (p.res) {
                const unsigned roff = (unsigned)(pb * p.resCs + n0 + wn * WN + 4 * (lane & 7)) * 4u;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        rv[mt][nt][i] = buf_load4(rr, rowok[mt][i] ? roff + (unsigned)((8 * i * p.resCs + nt * 32) * 4) : OOB);
            }
        }
#ifdef WS_TIMING
        const long long tq1 = clock64(); tacc[0] += tq1 - tq0;
#endif
        // ---- K/2 MFMA steps, 4 per LDS read ----
        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        const f32x4* Ab = reinterpret_cast<const f32x4*>(As + stage * ASTAGE) + (h * (KQ / 2)) * BM + wm * WM + col;
        const f32x4* Wb = reinterpret_cast<const f32x4*>(Ws) + (h * (KQ / 2)) * BN + wn * WN + col;
#pragma unroll
        for (int g = 0; g < KQ / 2; ++g) {
            f32x4 a[MT], bq[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = Ab[g * BM + mt * 32];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[nt] = Wb[g * BN + nt * 32];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[nt][r], a[mt][r], acc[mt][nt], 0, 0, 0);
        }
        // Everything this wavefront has in flight is older than the stores below: the residual, the next tile's DMA and
        // the previous tile's stores (which had this whole MFMA loop to retire).  After the barrier every wavefront's part
        // of the next tile has landed and nobody reads this stage any more; the stores then go out unwaited.
#ifdef WS_TIMING
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[MT - 1][NT - 1][15]));
        const long long tq2 = clock64(); tacc[1] += tq2 - tq1;
#endif
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef WS_TIMING
        const long long tq3 = clock64(); tacc[2] += tq3 - tq2;
#endif
        // ---- epilogue.  The MFMA leaves a lane with pixel `col` and channels 8g + 4h + (0..3) per accumulator quad;
        // 32 x 32 outputs at a time go through a per-wavefront LDS patch (16-byte writes, rows 36 floats apart) and come
        // back as 4 consecutive channels of pixel lane/8 (+8i): loads and stores are then whole 128-byte lines, 16 bytes
        // per lane.  (Stored straight from the MFMA layout -- 16 bytes per lane, every lane in another line -- the
        // layer was slower than the implicit GEMM: 652 vs 622 us.)
        float* scr = As + 2 * ASTAGE + 2 * BN + wave * (32 * 36);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(scr + col * 36 + 8 * g + 4 * h) =
                        f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(ssc + wn * WN + nt * 32 + 4 * (lane & 7));
                const f32x4 f4 = *reinterpret_cast<const f32x4*>(ssc + BN + wn * WN + nt * 32 + 4 * (lane & 7));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(scr + (8 * i + (lane >> 3)) * 36 + 4 * (lane & 7));
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = v[e] * s4[e] + f4[e];
                        if (p.res) u += rv[mt][nt][i][e];
                        o[e] = fmaxf(u, floor_);
                    }
                    buf_store4(yr, rowok[mt][i] ? ooff[mt] + (unsigned)((8 * i * p.yCs + nt * 32) * 4) : OOB, o);
                    if (p.yr && rowok[mt][i]) {      // range slot of the output (range.h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(o[e]); rmax = b > rmax ? b : rmax; }
                    }
                }
            }
#ifdef WS_TIMING
        tacc[3] += clock64() - tq3; tacc[4] += 1;
#endif
    }
#ifdef WS_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 77 && lane == 0) for (int i = 0; i < 5; ++i) p.y[wave * 8 + i] = (float)tacc[i];
#endif
    if (p.yr) range_note_wave(p.yr, rmax, (unsigned)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));      // once per persistent block and wavefront
}

template <int K, int BM, int BN>
static hipError_t launch_ws(const ConvParams& p0, hipStream_t st, int cus)
{
    ConvParams p = p0;
    p.w = p.wws;
    p.w_bytes = p.wws_bytes;
    const int mtiles = (p.M + BM - 1) / BM, ngroups = p.Cout_store / BN;
    // one block per CU; whole XCD-rounds of column groups only (every stream needs all its groups resident together)
    int streams = (cus / 8 / ngroups) * 8;
    if (streams < 8) streams = 8;
    if (streams > ((mtiles + 7) / 8) * 8) streams = ((mtiles + 7) / 8) * 8;
    constexpr size_t lds = (size_t)(K * BN + 2 * K * BM + 2 * BN + 4 * 32 * 36) * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv1x1_ws_kernel<K, BM, BN>), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((conv1x1_ws_kernel<K, BM, BN>), dim3(streams * ngroups), dim3(256), lds, st, p, mtiles, ngroups, streams);
    return hipGetLastError();
}

// layers the kernel takes: 1x1 / stride 1 / no padding, 64 -> k*256 or 128 -> k*128 channels, ReLU or no activation
bool conv_ws_eligible(const ConvParams& p)
{
    if (p.kh != 1 || p.kw != 1 || p.sh != 1 || p.sw != 1 || p.ph || p.pw || p.deconv2x || p.y2 || p.f16) return false;
    if (p.act != 0 && p.act != 1) return false;
    if (p.Cin == 64) return p.Cout_store % 256 == 0;
    if (p.Cin == 128) return p.Cout_store % 128 == 0;
    return false;
}

size_t conv_ws_pack_floats(int Cin, int cout_store) { return (size_t)Cin * cout_store; }

// w: [Cout][Cin] (1x1 kernel) -> per column group of BN channels the LDS image [Cin/4][BN][4]
void conv_ws_pack(const float* w, int Cout, int Cin, int cout_store, float* out)
{
    const int BN = Cin == 64 ? 256 : 128, KQ = Cin / 4;
    for (int n = 0; n < cout_store; ++n) {
        const int cg = n / BN, j = n % BN;
        for (int k = 0; k < Cin; ++k)
            out[(((size_t)cg * KQ + k / 4) * BN + j) * 4 + (k & 3)] = n < Cout ? w[(size_t)n * Cin + k] : 0.f;
    }
}

hipError_t launch_conv_ws(const ConvParams& p, hipStream_t st)
{
    if (!conv_ws_eligible(p) || !p.wws) return hipErrorInvalidValue;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return p.Cin == 64 ? launch_ws<64, 128, 256>(p, st, cus) : launch_ws<128, 64, 128>(p, st, cus);
}
