// Bandwidth-bound kernels of the Accel path (gfx950).  All of them are plain
// HBM streaming / gather kernels: 16-byte per-lane accesses along the NHWC
// channel axis (or along x for the NCHW boundary tensors), grid-stride free
// (one element quad per thread, grids >> 256 CUs).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <math.h>
#include "kernels.h"
#include "range.h"

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Range slots (range.h): every byte mover below whose output some fp16x2-form convolution reads raises that tensor's slot to the
// largest |value| it stored.  The kernels are written as a body that returns the bit pattern of this thread's largest stored |value|
// (0: nothing stored) and a wrapper that reduces it over the block -- one atomicMax per block -- so that every thread reaches the barrier.
#define RANGE_OF4(v) max(max(range_abs_bits((v).x), range_abs_bits((v).y)), max(range_abs_bits((v).z), range_abs_bits((v).w)))

// ---------------------------------------------------------------------------
// image boundary: NCHW 3xHxW -> NHWC4
// ---------------------------------------------------------------------------
// (all byte movers below: blockIdx.z = image of the batch, images a fixed stride apart)
// `slot` (may be null): device address of a pointer the host may redirect to a caller-owned frame between two runs of a captured
// plan (accel_model_bind_device): the image is then read where it lies, without a copy into the model's input buffer
__device__ __forceinline__ unsigned prep_rgb_body(const float* __restrict__ src, float* __restrict__ dst, int HW,
                                                  const float* scale, const float* shift, const float* const* slot)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return 0u;
    if (slot) src = *slot;
    src += (size_t)blockIdx.z * 3 * HW;
    dst += (size_t)blockIdx.z * 4 * HW;
    float r = src[i], g = src[HW + i], b = src[2 * HW + i];
    if (scale) {
        r = r * scale[0] + shift[0];
        g = g * scale[1] + shift[1];
        b = b * scale[2] + shift[2];
    }
    reinterpret_cast<float4*>(dst)[i] = make_float4(r, g, b, 0.f);
    return max(max(range_abs_bits(r), range_abs_bits(g)), range_abs_bits(b));
}
__global__ void prep_rgb_kernel(const float* __restrict__ src, float* __restrict__ dst, int HW,
                                const float* scale, const float* shift, const float* const* slot, unsigned* yr)
{
    const unsigned m = prep_rgb_body(src, dst, HW, scale, shift, slot);
    if (yr) range_note_block(yr, m, blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z);
}

hipError_t launch_prep_rgb(const float* src, float* dst, int H, int W, const float* scale3,
                           const float* shift3, int N, hipStream_t st, const float* const* slot, unsigned* yr)
{
    const int HW = H * W;
    hipLaunchKernelGGL(prep_rgb_kernel, dim3(cdiv(HW, 256), 1, N), dim3(256), 0, st, src, dst, HW, scale3, shift3, slot, yr);
    return hipGetLastError();
}

// FlowNet input: Concat(cur/255, prev/255) -> avg pool 2x2/2  (ref get_flownet :1752-1753)
__device__ __forceinline__ unsigned prep_flow_body(const float* __restrict__ cur, const float* __restrict__ prev,
                                                   float* __restrict__ dst, int H, int W, const float* const* cur_slot, const float* const* prev_slot)
{
    const int Wo = W >> 1, Ho = H >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ho * Wo) return 0u;
    if (cur_slot) cur = *cur_slot;
    if (prev_slot) prev = *prev_slot;
    const int oy = i / Wo, ox = i - oy * Wo;
    const size_t HW = (size_t)H * W;
    cur += blockIdx.z * 3 * HW; prev += blockIdx.z * 3 * HW;
    dst += (size_t)blockIdx.z * Ho * Wo * 8;
    float o[8];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float* pl = (c < 3 ? cur + c * HW : prev + (c - 3) * HW) + (size_t)(2 * oy) * W + 2 * ox;
        const float2 t = *reinterpret_cast<const float2*>(pl);
        const float2 b = *reinterpret_cast<const float2*>(pl + W);
        float acc = 0.f;
        acc += t.x / 255.0f; acc += t.y / 255.0f; acc += b.x / 255.0f; acc += b.y / 255.0f;
        o[c] = acc / 4.0f;
    }
    o[6] = o[7] = 0.f;
    float4* d = reinterpret_cast<float4*>(dst + (size_t)i * 8);
    d[0] = make_float4(o[0], o[1], o[2], o[3]);
    d[1] = make_float4(o[4], o[5], o[6], o[7]);
    unsigned m = 0u;
#pragma unroll
    for (int c = 0; c < 6; ++c) m = max(m, range_abs_bits(o[c]));
    return m;
}
__global__ void prep_flow_kernel(const float* __restrict__ cur, const float* __restrict__ prev,
                                 float* __restrict__ dst, int H, int W, const float* const* cur_slot, const float* const* prev_slot, unsigned* yr)
{
    const unsigned m = prep_flow_body(cur, prev, dst, H, W, cur_slot, prev_slot);
    if (yr) range_note_block(yr, m, blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z);
}

hipError_t launch_prep_flow(const float* cur, const float* prev, float* dst, int H, int W, int N, hipStream_t st,
                            const float* const* cur_slot, const float* const* prev_slot, unsigned* yr)
{
    const int n = (H / 2) * (W / 2);
    hipLaunchKernelGGL(prep_flow_kernel, dim3(cdiv(n, 256), 1, N), dim3(256), 0, st, cur, prev, dst, H, W, cur_slot, prev_slot, yr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Pooling (max 3x3/2 'full' or 'valid', avg 2x2/2), optional BN+ReLU epilogue
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned pool_body(const PoolParams& p)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.Ho * p.Wo * p.C4;
    if (idx >= total) return 0u;
    const int c4 = (int)(idx % p.C4);
    const int pix = (int)(idx / p.C4);
    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
    int hs = oy * p.sh - p.ph, ws = ox * p.sw - p.pw;
    int he = min(hs + p.kh, p.H + p.ph), we = min(ws + p.kw, p.W + p.pw);
    const int area = (he - hs) * (we - ws);
    hs = max(hs, 0); ws = max(ws, 0); he = min(he, p.H); we = min(we, p.W);
    float4 acc = p.is_max ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = hs; iy < he; ++iy)
        for (int ix = ws; ix < we; ++ix) {
            const float4 v = *reinterpret_cast<const float4*>(p.x + blockIdx.z * p.x_img + ((size_t)iy * p.W + ix) * p.xCs + c4 * 4);
            if (p.is_max) {
                acc.x = v.x > acc.x ? v.x : acc.x; acc.y = v.y > acc.y ? v.y : acc.y;
                acc.z = v.z > acc.z ? v.z : acc.z; acc.w = v.w > acc.w ? v.w : acc.w;
            } else {
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    if (!p.is_max) {
        const float a = (float)area;
        acc.x /= a; acc.y /= a; acc.z /= a; acc.w /= a;
    }
    if (p.scale) {
        const float4 s = *reinterpret_cast<const float4*>(p.scale + c4 * 4);
        const float4 b = *reinterpret_cast<const float4*>(p.shift + c4 * 4);
        acc.x = acc.x * s.x + b.x; acc.y = acc.y * s.y + b.y;
        acc.z = acc.z * s.z + b.z; acc.w = acc.w * s.w + b.w;
    }
    if (p.relu) {
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
        acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    *reinterpret_cast<float4*>(p.y + blockIdx.z * p.y_img + (size_t)pix * p.yCs + c4 * 4) = acc;
    return RANGE_OF4(acc);
}
__global__ void pool_kernel(PoolParams p)
{
    const unsigned m = pool_body(p);
    if (p.yr) range_note_block(p.yr, m, blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z);
}

// max 3x3 / stride 2 (both conventions: the windows are clipped to the image): the nine taps as nine independent 16-byte loads in
// flight (the general kernel's loops have run-time bounds: one load, one wait, one maximum at a time).  Same values: a maximum is exact.
__device__ __forceinline__ unsigned pool_max3x3s2_body(const PoolParams& p)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.Ho * p.Wo * p.C4;
    if (idx >= total) return 0u;
    const int c4 = (int)(idx % p.C4);
    const int pix = (int)(idx / p.C4);
    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
    const int hs = oy * 2 - p.ph, ws = ox * 2 - p.pw;
    const float* x = p.x + blockIdx.z * p.x_img + c4 * 4;
    float4 v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = hs + t / 3, ix = ws + t % 3;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int cy = ok ? iy : 0, cx = ok ? ix : 0;
        v[t] = *reinterpret_cast<const float4*>(x + ((size_t)cy * p.W + cx) * p.xCs);
        if (!ok) v[t] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    float4 acc = v[0];
#pragma unroll
    for (int t = 1; t < 9; ++t) {
        acc.x = v[t].x > acc.x ? v[t].x : acc.x; acc.y = v[t].y > acc.y ? v[t].y : acc.y;
        acc.z = v[t].z > acc.z ? v[t].z : acc.z; acc.w = v[t].w > acc.w ? v[t].w : acc.w;
    }
    if (p.scale) {
        const float4 s = *reinterpret_cast<const float4*>(p.scale + c4 * 4);
        const float4 b = *reinterpret_cast<const float4*>(p.shift + c4 * 4);
        acc.x = acc.x * s.x + b.x; acc.y = acc.y * s.y + b.y;
        acc.z = acc.z * s.z + b.z; acc.w = acc.w * s.w + b.w;
    }
    if (p.relu) {
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
        acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    *reinterpret_cast<float4*>(p.y + blockIdx.z * p.y_img + (size_t)pix * p.yCs + c4 * 4) = acc;
    return RANGE_OF4(acc);
}
__global__ void pool_max3x3s2_kernel(PoolParams p)
{
    const unsigned m = pool_max3x3s2_body(p);
    if (p.yr) range_note_block(p.yr, m, blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z);
}

hipError_t launch_pool(const PoolParams& p, hipStream_t st)
{
    const long total = (long)p.Ho * p.Wo * p.C4;
    if (p.is_max && p.kh == 3 && p.kw == 3 && p.sh == 2 && p.sw == 2) {
        hipLaunchKernelGGL(pool_max3x3s2_kernel, dim3(cdiv(total, 256), 1, p.N > 0 ? p.N : 1), dim3(256), 0, st, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pool_kernel, dim3(cdiv(total, 256), 1, p.N > 0 ? p.N : 1), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Flow warp = GridGenerator(transform_type='warp') + BilinearSampler
// (ref accel_18.py:174-175).  Same arithmetic sequence as the two MXNet ops:
//   gx = (x + fx) / ((W-1)/2) - 1 ;  x_real = (gx + 1) * (W-1) / 2
// In NHWC the four taps of a pixel are four contiguous channel rows, so the
// gather is fully coalesced and needs no cross-lane shuffles.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void flow_warp_body(const float* __restrict__ feat, int fCs,
                                               const float* __restrict__ flow, int flCs,
                                               float* __restrict__ out, int oCs, int C4, int H, int W,
                                               float* __restrict__ out2, int o2Cs, const float* __restrict__ bias, unsigned& m1, unsigned& m2)
{
    // (measured and not adopted, profiles/r05_ab_xcd_mapping.log: consecutive pixels on ONE XCD -- the mapping that helps dcn_cols9 below --
    // makes this kernel slower, 231 -> 276 us at 8 clips per call)
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)H * W * C4) return;
    {
        const size_t hw = (size_t)H * W * blockIdx.z;
        feat += hw * fCs; flow += hw * flCs; out += hw * oCs;
        if (out2) out2 += hw * o2Cs;
    }
    const int c4 = (int)(idx % C4);
    const int pix = (int)(idx / C4);
    const int y = pix / W, x = pix - y * W;
    const float2 f = *reinterpret_cast<const float2*>(flow + (size_t)pix * flCs);
    const float sx = (float)(W - 1) / 2.0f, sy = (float)(H - 1) / 2.0f;
    const float gx = (f.x + (float)x) / sx - 1.0f;
    const float gy = (f.y + (float)y) / sy - 1.0f;
    const float y_real = (gy + 1) * (H - 1) / 2;
    const float x_real = (gx + 1) * (W - 1) / 2;
    const int ty = (int)floorf(y_real), tx = (int)floorf(x_real);
    const float wy = 1.0f - (y_real - ty), wx = 1.0f - (x_real - tx);
    const bool x0 = tx >= 0 && tx <= W - 1, x1 = tx + 1 >= 0 && tx + 1 <= W - 1;
    const bool y0 = ty >= 0 && ty <= H - 1, y1 = ty + 1 >= 0 && ty + 1 <= H - 1;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 tl = z, tr = z, bl = z, br = z;
    const float* base = feat + c4 * 4;
    if (x0 && y0) tl = *reinterpret_cast<const float4*>(base + ((size_t)ty * W + tx) * fCs);
    if (x1 && y0) tr = *reinterpret_cast<const float4*>(base + ((size_t)ty * W + tx + 1) * fCs);
    if (x0 && y1) bl = *reinterpret_cast<const float4*>(base + ((size_t)(ty + 1) * W + tx) * fCs);
    if (x1 && y1) br = *reinterpret_cast<const float4*>(base + ((size_t)(ty + 1) * W + tx + 1) * fCs);
    const float w00 = wy * wx, w01 = wy * (1.0f - wx), w10 = (1.0f - wy) * wx, w11 = (1.0f - wy) * (1.0f - wx);
    float4 o;
    // tl*wy*wx evaluates as (tl*wy)*wx in the reference expression; keep that association
    o.x = tl.x * wy * wx + tr.x * wy * (1.0f - wx) + bl.x * (1.0f - wy) * wx + br.x * (1.0f - wy) * (1.0f - wx);
    o.y = tl.y * wy * wx + tr.y * wy * (1.0f - wx) + bl.y * (1.0f - wy) * wx + br.y * (1.0f - wy) * (1.0f - wx);
    o.z = tl.z * wy * wx + tr.z * wy * (1.0f - wx) + bl.z * (1.0f - wy) * wx + br.z * (1.0f - wy) * (1.0f - wx);
    o.w = tl.w * wy * wx + tr.w * wy * (1.0f - wx) + bl.w * (1.0f - wy) * wx + br.w * (1.0f - wy) * (1.0f - wx);
    (void)w00; (void)w01; (void)w10; (void)w11;
    *reinterpret_cast<float4*>(out + (size_t)pix * oCs + c4 * 4) = o;
    m1 = RANGE_OF4(o);
    if (out2) {
        const float4 b = *reinterpret_cast<const float4*>(bias + c4 * 4);
        float4 r;
        r.x = fmaxf(o.x + b.x, 0.f); r.y = fmaxf(o.y + b.y, 0.f); r.z = fmaxf(o.z + b.z, 0.f); r.w = fmaxf(o.w + b.w, 0.f);
        *reinterpret_cast<float4*>(out2 + (size_t)pix * o2Cs + c4 * 4) = r;
        m2 = RANGE_OF4(r);
    }
}
__global__ void flow_warp_kernel(const float* __restrict__ feat, int fCs,
                                 const float* __restrict__ flow, int flCs,
                                 float* __restrict__ out, int oCs, int C4, int H, int W,
                                 float* __restrict__ out2, int o2Cs, const float* __restrict__ bias, unsigned* yr, unsigned* y2r)
{
    unsigned m1 = 0u, m2 = 0u;
    flow_warp_body(feat, fCs, flow, flCs, out, oCs, C4, H, W, out2, o2Cs, bias, m1, m2);
    if (yr) range_note_block(yr, m1, blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z);
    if (out2 && y2r) range_note_block(y2r, m2, blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z);
}

hipError_t launch_flow_warp(const float* feat, int fCs, const float* flow, int flCs, float* out, int oCs,
                            int C, int H, int W, float* out2, int o2Cs, const float* bias, int N, hipStream_t st, unsigned* yr, unsigned* y2r)
{
    const int C4 = C / 4;
    const long total = (long)H * W * C4;
    hipLaunchKernelGGL(flow_warp_kernel, dim3(cdiv(total, 256), 1, N), dim3(256), 0, st, feat, fCs, flow, flCs,
                       out, oCs, C4, H, W, out2, o2Cs, bias, yr, y2r);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Deformable im2col (DCN v1 sampling rule, see oracle/accel_oracle.c
// orc_deform_im2col): writes col[pixel][tap][ci] (NHWC with 9*C channels) that
// the implicit-GEMM kernel then contracts as a 1x1 convolution.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned dcn_cols_body(const DcnColsParams& p)
{
    const int C4 = p.C / 4, taps = p.kh * p.kw;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)p.Ho * p.Wo * taps * C4) return 0u;
    const int c4 = (int)(idx % C4);
    const int tap = (int)((idx / C4) % taps);
    const int pix = (int)(idx / ((long)C4 * taps));
    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
    const int i = tap / p.kw, j = tap - i * p.kw;
    const int cpg = p.C / p.dg;
    const int g = (c4 * 4) / cpg;
    const size_t zn = blockIdx.z;      // image of the batch
    const float2 o = *reinterpret_cast<const float2*>(p.off + zn * p.Ho * p.Wo * p.offCs + (size_t)pix * p.offCs + g * 2 * taps + 2 * tap);
    const float oh = o.x, ow = o.y;
    const int h_in = oy * p.sh - p.ph, w_in = ox * p.sw - p.pw;
    const float h_im = (float)(h_in + i * p.dh) + oh;
    const float w_im = (float)(w_in + j * p.dw) + ow;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h_im >= 0 && w_im >= 0 && h_im < p.H && w_im < p.W) {
        float h = (float)(i * p.dh) + oh, w = (float)(j * p.dw) + ow;
        const int height = p.H - h_in, width = p.W - w_in;
        int h_low = (int)floorf(h), w_low = (int)floorf(w);
        int h_high, w_high;
        if (h_low >= height - 1) { h_high = h_low = height - 1; h = (float)h_low; } else h_high = h_low + 1;
        if (w_low >= width - 1) { w_high = w_low = width - 1; w = (float)w_low; } else w_high = w_low + 1;
        const float lh = h - h_low, lw = w - w_low;
        const float hh = 1 - lh, hw = 1 - lw;
        // absolute coordinates, clamped only for memory safety (no effect on in-range samples)
        const int y0 = min(max(h_in + h_low, 0), p.H - 1), y1 = min(max(h_in + h_high, 0), p.H - 1);
        const int x0 = min(max(w_in + w_low, 0), p.W - 1), x1 = min(max(w_in + w_high, 0), p.W - 1);
        const float* b = p.x + zn * p.H * p.W * p.xCs + c4 * 4;
        const float4 v1 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * p.W + x0) * p.xCs);
        const float4 v2 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * p.W + x1) * p.xCs);
        const float4 v3 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * p.W + x0) * p.xCs);
        const float4 v4 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * p.W + x1) * p.xCs);
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        val.x = w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
        val.y = w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
        val.z = w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
        val.w = w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
    }
    const size_t at = zn * p.Ho * p.Wo * p.colCs + (size_t)pix * p.colCs + (size_t)tap * p.C + c4 * 4;
    if (p.col_half) {
        typedef _Float16 f16x4m __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f16x4m*>(reinterpret_cast<_Float16*>(p.col) + at) = f16x4m{(_Float16)val.x, (_Float16)val.y, (_Float16)val.z, (_Float16)val.w};
    } else {
        *reinterpret_cast<float4*>(p.col + at) = val;
    }
    return RANGE_OF4(val);
}
__global__ void dcn_cols_kernel(DcnColsParams p)
{
    const unsigned m = dcn_cols_body(p);
    if (p.yr && !p.col_half) range_note_block(p.yr, m, blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z);
}

// The same function with ONE thread per (pixel, channel quad) walking all kh*kw = 9 taps: the nine offset pairs are fetched first,
// then the 36 corner fetches of the nine taps are in flight together, then nine stores -- the one-tap-per-thread form above has two
// dependent memory round trips (offset, corners) per 16 bytes written and reached 2.1 TB/s of column writes (round-4 profile: 559-574
// us for the 1.2 GB column buffer of a res5 layer at 8 clips); same arithmetic per value, same results bit for bit.
template <int TPT, bool NT>      // taps per thread: 3 (one kernel row; blockIdx.y = the row); NT: streaming stores
__device__ __forceinline__ unsigned dcn_cols9_body(const DcnColsParams& p)
{
    const int C4 = p.C / 4;
    // blocks are dealt to the eight XCDs round-robin: consecutive blocks of ONE XCD take consecutive pixels, so that the corners the
    // taps of neighbouring pixels share are fetched into one L2 instead of into all of them (round-5 PMC: 534 MB fetched per launch
    // for a 134 MB input with pixel groups dealt round-robin; same-box A/B: 365 -> 295-330 us per launch at 8 clips per call, +0.35 % on the step)
    const int nblk = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const long idx = (long)swz * blockDim.x + threadIdx.x;
    if (idx >= (long)p.Ho * p.Wo * C4) return 0u;
    unsigned rmax = 0u;
    const int c4 = (int)(idx % C4);
    const int pix = (int)(idx / C4);
    const int t0 = TPT * blockIdx.y;
    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
    const int cpg = p.C / p.dg;
    const int g = (c4 * 4) / cpg;
    const size_t zn = blockIdx.z;
    const float* offp = p.off + zn * p.Ho * p.Wo * p.offCs + (size_t)pix * p.offCs + g * 18;
    float2 o[TPT];
#pragma unroll
    for (int t = 0; t < TPT; ++t) o[t] = *reinterpret_cast<const float2*>(offp + 2 * (t0 + t));
    const int h_in = oy * p.sh - p.ph, w_in = ox * p.sw - p.pw;
    const float* b = p.x + zn * p.H * p.W * p.xCs + c4 * 4;
    float4 v1[TPT], v2[TPT], v3[TPT], v4[TPT];
    float w1[TPT], w2[TPT], w3[TPT], w4[TPT];
    bool inside[TPT];
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
        const int i = (t0 + t) / 3, j = (t0 + t) - 3 * i;
        const float oh = o[t].x, ow = o[t].y;
        const float h_im = (float)(h_in + i * p.dh) + oh;
        const float w_im = (float)(w_in + j * p.dw) + ow;
        const bool in = h_im >= 0 && w_im >= 0 && h_im < p.H && w_im < p.W;
        float h = (float)(i * p.dh) + oh, w = (float)(j * p.dw) + ow;
        const int height = p.H - h_in, width = p.W - w_in;
        int h_low = (int)floorf(h), w_low = (int)floorf(w);
        int h_high, w_high;
        if (h_low >= height - 1) { h_high = h_low = height - 1; h = (float)h_low; } else h_high = h_low + 1;
        if (w_low >= width - 1) { w_high = w_low = width - 1; w = (float)w_low; } else w_high = w_low + 1;
        const float lh = h - h_low, lw = w - w_low;
        const float hh = 1 - lh, hw = 1 - lw;
        const int y0 = min(max(h_in + h_low, 0), p.H - 1), y1 = min(max(h_in + h_high, 0), p.H - 1);
        const int x0 = min(max(w_in + w_low, 0), p.W - 1), x1 = min(max(w_in + w_high, 0), p.W - 1);
        v1[t] = *reinterpret_cast<const float4*>(b + ((size_t)y0 * p.W + x0) * p.xCs);
        v2[t] = *reinterpret_cast<const float4*>(b + ((size_t)y0 * p.W + x1) * p.xCs);
        v3[t] = *reinterpret_cast<const float4*>(b + ((size_t)y1 * p.W + x0) * p.xCs);
        v4[t] = *reinterpret_cast<const float4*>(b + ((size_t)y1 * p.W + x1) * p.xCs);
        // a sample outside the image contributes an exact zero (DCN v1): the fetches stay unconditional (clamped coordinates), the
        // RESULT is selected below -- zero weights would turn a non-finite feature value at the clamped position into NaN
        w1[t] = hh * hw; w2[t] = hh * lw; w3[t] = lh * hw; w4[t] = lh * lw;
        inside[t] = in;
    }
    const size_t at0 = zn * p.Ho * p.Wo * p.colCs + (size_t)pix * p.colCs + c4 * 4;
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
        float4 val;
        val.x = w1[t] * v1[t].x + w2[t] * v2[t].x + w3[t] * v3[t].x + w4[t] * v4[t].x;
        val.y = w1[t] * v1[t].y + w2[t] * v2[t].y + w3[t] * v3[t].y + w4[t] * v4[t].y;
        val.z = w1[t] * v1[t].z + w2[t] * v2[t].z + w3[t] * v3[t].z + w4[t] * v4[t].z;
        val.w = w1[t] * v1[t].w + w2[t] * v2[t].w + w3[t] * v3[t].w + w4[t] * v4[t].w;
        if (!inside[t]) val = make_float4(0.f, 0.f, 0.f, 0.f);
        rmax = max(rmax, RANGE_OF4(val));
        const size_t at = at0 + (size_t)(t0 + t) * p.C;
        if (p.col_half) {
            typedef _Float16 f16x4m __attribute__((ext_vector_type(4)));
            *reinterpret_cast<f16x4m*>(reinterpret_cast<_Float16*>(p.col) + at) = f16x4m{(_Float16)val.x, (_Float16)val.y, (_Float16)val.z, (_Float16)val.w};
        } else if (NT) {
            typedef float f32x4m __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(f32x4m{val.x, val.y, val.z, val.w}, reinterpret_cast<f32x4m*>(p.col + at));
        } else {
            *reinterpret_cast<float4*>(p.col + at) = val;
        }
    }
    return rmax;
}
template <int TPT, bool NT = false>
__global__ __launch_bounds__(256) void dcn_cols9_kernel(DcnColsParams p)
{
    const unsigned m = dcn_cols9_body<TPT, NT>(p);
    if (p.yr && !p.col_half) range_note_block(p.yr, m, blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z);
}

hipError_t launch_dcn_cols(const DcnColsParams& p, hipStream_t st)
{
    if (p.kh == 3 && p.kw == 3) {
        // measured (scripts/microbench/dcn_time.py, round 4; res5 of the key plan at 8 clips / of the ResNet-18 branch / at one clip): one tap per
        // thread 511 / 137 / 80 us, three taps (one kernel row) 431 / 82 / 55, nine 464 / 99 / 56, three with streaming stores 389 / 80 / 55:
        // a column buffer beyond the 256 MB Infinity Cache is written past the caches, a smaller one stays cached for the GEMM behind it
        const long total = (long)p.Ho * p.Wo * (p.C / 4);
        const size_t col_bytes = (size_t)(p.N > 0 ? p.N : 1) * p.Ho * p.Wo * p.colCs * (p.col_half ? 2 : 4);
        if (col_bytes > ((size_t)256 << 20)) hipLaunchKernelGGL((dcn_cols9_kernel<3, true>), dim3(cdiv(total, 256), 3, p.N > 0 ? p.N : 1), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(dcn_cols9_kernel<3>, dim3(cdiv(total, 256), 3, p.N > 0 ? p.N : 1), dim3(256), 0, st, p);
        return hipGetLastError();
    }
    // other kernel sizes (none on the Accel graphs; the operator entry point accepts them): one tap per thread
    const long total = (long)p.Ho * p.Wo * p.kh * p.kw * (p.C / 4);
    hipLaunchKernelGGL(dcn_cols_kernel, dim3(cdiv(total, 256), 1, p.N > 0 ? p.N : 1), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Fused score tail:
//   Deconvolution 32x32/16 group=ncls (+Crop offset 8,8) of one or two score
//   maps, Concat, `correction` 1x1 conv 2*ncls -> ncls (+bias), argmax.
//   (ref accel_18.py:193-197,223-235; demo.py:238,245)
// One thread per output pixel; logits are written NCHW (the boundary layout),
// coalesced along x; the label map (first maximal index) is written alongside.
// Accumulation orders match the oracle: deconv taps (ky asc, kx asc), then the
// 1x1 conv over channels ascending on top of the bias.
// ---------------------------------------------------------------------------
template <int NCLS>
__device__ __forceinline__ void upsample_px(const float* __restrict__ s, int Cs, const float* __restrict__ w,
                                            int Hs, int Ws, int Y, int X, float* o)
{
    const int yy = Y + 8, xx = X + 8;
    const int i0 = yy >> 4, ky0 = yy & 15, j0 = xx >> 4, kx0 = xx & 15;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) o[c] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {          // ky = ky0 (row i0), then ky0+16 (row i0-1)
        const int i = i0 - a, ky = ky0 + 16 * a;
        if (i < 0 || i >= Hs) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = j0 - b, kx = kx0 + 16 * b;
            if (j < 0 || j >= Ws) continue;
            const float* sp = s + ((size_t)i * Ws + j) * Cs;
            const float* wp = w + ky * 32 + kx;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) o[c] += wp[c * 1024] * sp[c];
        }
    }
}

template <int NCLS>
__global__ __launch_bounds__(256) void score_tail_kernel(ScoreTailParams p)
{
    {   // image of the batch
        const size_t zn = blockIdx.z;
        p.left += zn * p.Hs * p.Ws * p.lCs;
        if (p.right) p.right += zn * (p.rHs ? p.rHs : p.Hs) * (p.rWs ? p.rWs : p.Ws) * p.rCs;
        p.logits += zn * p.ncls * p.H * p.W;
        p.labels += zn * p.H * p.W;
    }
    const int X = blockIdx.x * 64 + (threadIdx.x & 63);
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= p.W || Y >= p.H) return;
    float sl[NCLS], out[NCLS];
    upsample_px<NCLS>(p.left, p.lCs, p.wl, p.Hs, p.Ws, Y, X, sl);
    if (p.right) {
        float sr[NCLS];
        upsample_px<NCLS>(p.right, p.rCs, p.wr, p.rHs ? p.rHs : p.Hs, p.rWs ? p.rWs : p.Ws, Y, X, sr);
#pragma unroll
        for (int k = 0; k < NCLS; ++k) {
            float v = p.cb[k];
            const float* cw = p.cw + k * 2 * NCLS;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) v += cw[c] * sl[c];
#pragma unroll
            for (int c = 0; c < NCLS; ++c) v += cw[NCLS + c] * sr[c];
            out[k] = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NCLS; ++k) out[k] = p.cb ? sl[k] + p.cb[k] : sl[k];
    }
    if (p.softmax) {      // SoftmaxOutput(multi_output=True), inference: softmax over the class axis
        float mx = out[0];
#pragma unroll
        for (int k = 1; k < NCLS; ++k) mx = fmaxf(mx, out[k]);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NCLS; ++k) { out[k] = expf(out[k] - mx); sum += out[k]; }
#pragma unroll
        for (int k = 0; k < NCLS; ++k) out[k] = out[k] / sum;
    }
    const size_t HW = (size_t)p.H * p.W, o = (size_t)Y * p.W + X;
    int best = 0;
    float bv = out[0];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) {
        p.logits[k * HW + o] = out[k];
        if (k > 0 && out[k] > bv) { bv = out[k]; best = k; }
    }
    p.labels[o] = (unsigned char)best;
}

// Same tail for the common case "one score map, one upsampling filter shared by all classes" (the reference freezes
// the 32x32/16 filters at the bilinear init; the two-head fusion has already run at score resolution): a thread owns
// 4 consecutive pixels -- they share their 2x2 source cell since (X+8) % 16 is a multiple of 4 -- so the filter taps
// are 4 float4 loads instead of 4*NCLS scalars and every class plane is written with 16-byte stores.
// Per-pixel arithmetic (tap order, fma form) is the one of score_tail_kernel.
template <int NCLS>
__global__ __launch_bounds__(256) void score_tail_uniform_kernel(ScoreTailParams p)
{
    {   // image of the batch
        const size_t zn = blockIdx.z;
        p.left += zn * p.Hs * p.Ws * p.lCs;
        if (p.right) p.right += zn * p.Hs * p.Ws * p.rCs;
        p.logits += zn * p.ncls * p.H * p.W;
        p.labels += zn * p.H * p.W;
    }
    const int X = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= p.W || Y >= p.H) return;
    const int yy = Y + 8, xx = X + 8;
    const int i0 = yy >> 4, ky0 = yy & 15, j0 = xx >> 4, kx0 = xx & 15;
    float o[NCLS][4];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) o[c][0] = o[c][1] = o[c][2] = o[c][3] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int i = i0 - a, ky = ky0 + 16 * a;
        if (i < 0 || i >= p.Hs) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = j0 - b, kx = kx0 + 16 * b;
            if (j < 0 || j >= p.Ws) continue;
            const float* sp = p.left + ((size_t)i * p.Ws + j) * p.lCs;
            const float4 w = *reinterpret_cast<const float4*>(p.wl + ky * 32 + kx);
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                const float s = sp[c];
                o[c][0] += w.x * s; o[c][1] += w.y * s; o[c][2] += w.z * s; o[c][3] += w.w * s;
            }
        }
    }
    if (p.cb) {
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            const float bb = p.cb[c];
            o[c][0] += bb; o[c][1] += bb; o[c][2] += bb; o[c][3] += bb;
        }
    }
    if (p.softmax) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float mx = o[0][q];
#pragma unroll
            for (int k = 1; k < NCLS; ++k) mx = fmaxf(mx, o[k][q]);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < NCLS; ++k) { o[k][q] = expf(o[k][q] - mx); sum += o[k][q]; }
#pragma unroll
            for (int k = 0; k < NCLS; ++k) o[k][q] = o[k][q] / sum;
        }
    }
    const size_t HW = (size_t)p.H * p.W, off = (size_t)Y * p.W + X;
    int best[4] = {0, 0, 0, 0};
    float bv[4] = {o[0][0], o[0][1], o[0][2], o[0][3]};
#pragma unroll
    for (int k = 0; k < NCLS; ++k) {
        *reinterpret_cast<float4*>(p.logits + k * HW + off) = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k > 0 && o[k][q] > bv[q]) { bv[q] = o[k][q]; best[q] = k; }
    }
    *reinterpret_cast<uchar4*>(p.labels + off) = make_uchar4((unsigned char)best[0], (unsigned char)best[1],
                                                             (unsigned char)best[2], (unsigned char)best[3]);
}

hipError_t launch_score_tail(const ScoreTailParams& p, hipStream_t st)
{
    if (p.uniform_w && !p.right && p.W % 4 == 0) {
        dim3 g4(cdiv(p.W / 4, 64), cdiv(p.H, 4), p.N > 0 ? p.N : 1);
        if (p.ncls == 19) hipLaunchKernelGGL(score_tail_uniform_kernel<19>, g4, dim3(256), 0, st, p);
        else if (p.ncls == 2) hipLaunchKernelGGL(score_tail_uniform_kernel<2>, g4, dim3(256), 0, st, p);
        else if (p.ncls == 21) hipLaunchKernelGGL(score_tail_uniform_kernel<21>, g4, dim3(256), 0, st, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    dim3 grid(cdiv(p.W, 64), cdiv(p.H, 4), p.N > 0 ? p.N : 1);
    if (p.ncls == 19) hipLaunchKernelGGL(score_tail_kernel<19>, grid, dim3(256), 0, st, p);
    else if (p.ncls == 2) hipLaunchKernelGGL(score_tail_kernel<2>, grid, dim3(256), 0, st, p);
    else if (p.ncls == 21) hipLaunchKernelGGL(score_tail_kernel<21>, grid, dim3(256), 0, st, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// layout converters for the NCHW boundary tensors (feat_key / warping_feat)
// ---------------------------------------------------------------------------
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int Cs, float* __restrict__ dst, int C, int HW)
{
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int pix = p0 + r, c = c0 + threadIdx.x;
        tile[r][threadIdx.x] = (pix < HW && c < C) ? src[(size_t)pix * Cs + c] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int c = c0 + r, pix = p0 + threadIdx.x;
        if (pix < HW && c < C) dst[(size_t)c * HW + pix] = tile[threadIdx.x][r];
    }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cs, int C, int HW)
{
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int c = c0 + r, pix = p0 + threadIdx.x;
        tile[r][threadIdx.x] = (pix < HW && c < C) ? src[(size_t)c * HW + pix] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int pix = p0 + r, c = c0 + threadIdx.x;
        if (pix < HW && c < Cs) dst[(size_t)pix * Cs + c] = tile[threadIdx.x][r];   // pad channels get 0
    }
}

hipError_t launch_nhwc_to_nchw(const float* src, int Cs, float* dst, int C, int H, int W, hipStream_t st)
{
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 32), cdiv(C, 32)), dim3(32, 8), 0, st, src, Cs, dst, C, HW);
    return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int Cs, int C, int H, int W, hipStream_t st, unsigned* yr)
{
    (void)yr;      // (no range epilogue: a reader of an imported tensor measures its view itself, accel_hip.cpp)
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 32), cdiv(Cs, 32)), dim3(32, 8), 0, st, src, dst, Cs, C, HW);
    return hipGetLastError();
}

__global__ void argmax_nchw_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int C, int HW)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    int best = 0;
    float bv = x[i];
    for (int c = 1; c < C; ++c) {
        const float v = x[(size_t)c * HW + i];
        if (v > bv) { bv = v; best = c; }
    }
    out[i] = (unsigned char)best;
}

hipError_t launch_argmax_nchw(const float* logits, unsigned char* labels, int C, int HW, hipStream_t st)
{
    hipLaunchKernelGGL(argmax_nchw_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, st, logits, labels, C, HW);
    return hipGetLastError();
}

// strided NHWC view copy (C floats per pixel, C % 4 == 0): used to persist the
// propagated feature when it was produced inside a concat buffer
__global__ void copy_view_kernel(const float* __restrict__ src, int sCs, float* __restrict__ dst, int dCs, int C4, long total, unsigned* yr)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned m = 0u;
    if (idx < total) {
        const int c4 = (int)(idx % C4);
        const long pix = idx / C4;
        const float4 v = *reinterpret_cast<const float4*>(src + pix * sCs + c4 * 4);
        *reinterpret_cast<float4*>(dst + pix * dCs + c4 * 4) = v;
        m = RANGE_OF4(v);
    }
    if (yr) range_note_block(yr, m, blockIdx.x);
}

hipError_t launch_copy_view(const float* src, int sCs, float* dst, int dCs, int C, int HW, hipStream_t st, unsigned* yr)
{
    const int C4 = (C + 3) / 4;
    const long total = (long)HW * C4;
    hipLaunchKernelGGL(copy_view_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, src, sCs, dst, dCs, C4, total, yr);
    return hipGetLastError();
}

// Flat device-to-device copy of a persistent buffer (accel_model_write with a source that already lives in HBM: the executor's
// copy-in of a resident frame).  hipMemcpyAsync(DeviceToDevice) runs as a runtime blit kernel in ~25 MB pieces at 1.2 TB/s
// (profiles/r04_rocprof_summary.md: __amd_rocclr_copyBuffer, 20 us per piece); this is one launch of 16-byte accesses, four in flight
// per thread, grid sized to the chip.
__global__ __launch_bounds__(256) void copy_bytes_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + 256 * k < n16) v[k] = src[i + 256 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + 256 * k < n16) dst[i + 256 * k] = v[k];
    }
}

hipError_t launch_copy_bytes(const void* src, void* dst, size_t bytes, hipStream_t st)
{
    if (bytes % 16 || (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16)
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    const size_t n16 = bytes / 16;
    const size_t blocks = (n16 + 1023) / 1024;
    hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st,
                       static_cast<const float4*>(src), static_cast<float4*>(dst), n16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Narrow-N convolution (Cout <= 4: FlowNet's 2-channel flow predictors).  An MFMA tile would be
// >= 87 % padding and the layer is pure latency, so: one wavefront per output pixel, lanes stride
// over the (tap, ci) reduction with 16-byte loads, four dot products per lane, wave reduction
// through cross-lane shuffles, fused scale/shift (+activation) epilogue, one float4 store.
// Weights are the same packed [rows][K_pad] matrix the implicit-GEMM kernel uses.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_narrow_kernel(ConvParams p)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int HoWo = p.Ho * p.Wo;
    const int n = m / HoWo, rem = m - n * HoWo;             // batched: image n, pixel (oy, ox)
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const float* xn = p.x + (size_t)n * p.H * p.W * p.xCs;
    const int C4 = p.Cin / 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    unsigned rmax = 0u;
    for (int ky = 0; ky < p.kh; ++ky) {
        const int iy = oy * p.sh - p.ph + ky * p.dh;
        if ((unsigned)iy >= (unsigned)p.H) continue;
        for (int kx = 0; kx < p.kw; ++kx) {
            const int ix = ox * p.sw - p.pw + kx * p.dw;
            if ((unsigned)ix >= (unsigned)p.W) continue;
            const float4* xp = reinterpret_cast<const float4*>(xn + ((size_t)iy * p.W + ix) * p.xCs);
            const float4* wp = reinterpret_cast<const float4*>(p.w + (size_t)(ky * p.kw + kx) * p.Cin);
            const size_t rs = (size_t)p.K_pad / 4;
            for (int c = lane; c < C4; c += 64) {
                const float4 v = xp[c];
                const float4 w0 = wp[c], w1 = wp[rs + c], w2 = wp[2 * rs + c], w3 = wp[3 * rs + c];
                a0 += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
                a1 += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
                a2 += v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
                a3 += v.x * w3.x + v.y * w3.y + v.z * w3.z + v.w * w3.w;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o); a3 += __shfl_xor(a3, o);
    }
    if (lane == 0) {
        float v[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = v[e] * p.scale[e] + p.shift[e];
            if (p.res) v[e] += p.res[(size_t)m * p.resCs + e];
            if (p.act == 1) v[e] = fmaxf(v[e], 0.f);
            else if (p.act == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
        }
        *reinterpret_cast<float4*>(p.y + (size_t)m * p.yCs) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(v[e]); rmax = b > rmax ? b : rmax; }
    }
    if (p.yr) range_note_wave(p.yr, rmax, (unsigned)m);      // range slot of the output (range.h); the wavefront is whole here
}

// The same layer for its common shape -- 3x3, stride 1, pad 1, at most 2 real output channels (every flow predictor of
// FlowNet-S, resnet_v1_101_flownet_deeplab.py:1776-1801) -- with the input reuse the one-pixel-per-wavefront kernel
// lacks: a wavefront owns a strip of TX consecutive output pixels of one row; lanes split the channels (one float4 per
// lane and pass); per kernel row it loads the TX + 2 input pixels once and uses each for up to 3 output pixels, and the
// 6 weight quads of the row for all TX pixels: 12 FMAs per 16-byte load instead of 3, every input pixel fetched
// 3 x (TX+2)/TX times through the L2 instead of 9, no work for the two zero padding rows of the weight matrix.
template <int TX>
__global__ __launch_bounds__(256) void conv_narrow3x3_kernel(ConvParams p, int strips)
{
    const int lane = threadIdx.x & 63;
    // the 4 wavefronts of a block take the SAME strip of 4 consecutive rows: the rows they share (each input row feeds
    // 3 output rows) are then fetched by one CU / one XCD's L2 instead of three different ones
    const int rows = p.M / p.Wo;                 // N * Ho
    const int rgrp = blockIdx.x / strips;
    const int row = rgrp * 4 + (threadIdx.x >> 6), ox0 = (blockIdx.x - rgrp * strips) * TX;
    if (row >= rows) return;
    const int n = row / p.Ho, oy = row - n * p.Ho;
    const float* xn = p.x + (size_t)n * p.H * p.W * p.xCs;
    const int C4 = p.Cin / 4;
    const size_t rs = (size_t)p.K_pad / 4;       // float4s per weight row
    float acc[TX][2];
#pragma unroll
    for (int t = 0; t < TX; ++t) acc[t][0] = acc[t][1] = 0.f;
    for (int c = lane; c < C4; c += 64) {
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy - 1 + ky;
            if ((unsigned)iy >= (unsigned)p.H) continue;       // wave-uniform
            const float* xrow = xn + (size_t)iy * p.W * p.xCs;
            float4 x[TX + 2];
#pragma unroll
            for (int i = 0; i < TX + 2; ++i) {
                const int ix = ox0 - 1 + i;                    // wave-uniform as well
                x[i] = (unsigned)ix < (unsigned)p.W ? reinterpret_cast<const float4*>(xrow + (size_t)ix * p.xCs)[c]
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const float4* wp = reinterpret_cast<const float4*>(p.w + (size_t)(ky * 3) * p.Cin) + c;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 w0 = wp[(size_t)kx * C4], w1 = wp[rs + (size_t)kx * C4];
#pragma unroll
                for (int t = 0; t < TX; ++t) {
                    const float4 v = x[t + kx];
                    acc[t][0] += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
                    acc[t][1] += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TX; ++t)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            acc[t][0] += __shfl_xor(acc[t][0], o);
            acc[t][1] += __shfl_xor(acc[t][1], o);
        }
    unsigned rmax = 0u;
    if (lane < TX && ox0 + lane < p.Wo) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int t = 0; t < TX; ++t) if (lane == t) { a0 = acc[t][0]; a1 = acc[t][1]; }
        const size_t m = (size_t)row * p.Wo + ox0 + lane;
        float v[4] = {a0, a1, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = v[e] * p.scale[e] + p.shift[e];
            if (p.res) v[e] += p.res[m * p.resCs + e];
            if (p.act == 1) v[e] = fmaxf(v[e], 0.f);
            else if (p.act == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
        }
        *reinterpret_cast<float4*>(p.y + m * p.yCs) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned b = range_abs_bits(v[e]); rmax = b > rmax ? b : rmax; }
    }
    if (p.yr) range_note_wave(p.yr, rmax, blockIdx.x * 4u + (threadIdx.x >> 6));
}

hipError_t launch_conv_narrow(const ConvParams& p, hipStream_t st)
{
    if (p.kh == 3 && p.kw == 3 && p.sh == 1 && p.sw == 1 && p.dh == 1 && p.dw == 1 && p.ph == 1 && p.pw == 1 &&
        p.Cout <= 2 && p.Ho == p.H && p.Wo == p.W && p.Cin % 4 == 0) {
        // strips of 8 pixels once the map is big enough to fill the chip with them (4 wavefronts per block), of 4 below
        // that; maps too small even for those keep the one-pixel-per-wavefront kernel (more wavefronts)
        const int rows = p.M / p.Wo;
        if ((long)p.M >= 8L * 4 * 1024) {
            const int strips = cdiv(p.Wo, 8);
            hipLaunchKernelGGL(conv_narrow3x3_kernel<8>, dim3(cdiv(rows, 4) * strips), dim3(256), 0, st, p, strips);
            return hipGetLastError();
        }
        if ((long)p.M >= 4L * 4 * 512) {
            const int strips = cdiv(p.Wo, 4);
            hipLaunchKernelGGL(conv_narrow3x3_kernel<4>, dim3(cdiv(rows, 4) * strips), dim3(256), 0, st, p, strips);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(conv_narrow_kernel, dim3(cdiv(p.M, 4)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// Low-resolution half of the fused score tail when both upsampling kernels are the same
// class-independent filter (the frozen bilinear init of the reference, accel_18.py:153): the
// 2*ncls -> ncls `correction` conv commutes with the per-class upsampling, so it runs on the
// H/16 x W/16 score maps (256x fewer pixels) and only ncls maps are upsampled afterwards.
__global__ void score_fuse_lowres_kernel(const float* __restrict__ left, int lCs, const float* __restrict__ right, int rCs,
                                         const float* __restrict__ cw, float* __restrict__ z, int zCs, int ncls, int npix)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * zCs) return;
    const int k = idx % zCs, pix = idx / zCs;
    float v = 0.f;
    if (k < ncls) {
        const float* w = cw + k * 2 * ncls;
        const float* l = left + (size_t)pix * lCs;
        const float* r = right + (size_t)pix * rCs;
        for (int c = 0; c < ncls; ++c) v += w[c] * l[c];
        for (int c = 0; c < ncls; ++c) v += w[ncls + c] * r[c];
    }
    z[(size_t)pix * zCs + k] = v;
}

hipError_t launch_score_fuse_lowres(const float* left, int lCs, const float* right, int rCs, const float* cw,
                                    float* z, int zCs, int ncls, int npix, hipStream_t st)
{
    hipLaunchKernelGGL(score_fuse_lowres_kernel, dim3(cdiv((long)npix * zCs, 256)), dim3(256), 0, st, left, lCs, right, rCs,
                       cw, z, zCs, ncls, npix);
    return hipGetLastError();
}

// one pointer written from a kernel argument: stream-ordered, no host staging buffer to keep alive
__global__ void set_slot_kernel(const void** slot, const void* value) { *slot = value; }
hipError_t launch_set_slot(const void** slot, const void* value, hipStream_t st)
{
    hipLaunchKernelGGL(set_slot_kernel, dim3(1), dim3(1), 0, st, slot, value);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Range slot of a VIEW measured by a pass of its own (range.h): in front of an fp16x2-form convolution whose input tensor has a
// writer without the range epilogue, or none inside the plan (a persistent buffer another plan or the host wrote).
__global__ __launch_bounds__(256) void range_amax_kernel(const float* __restrict__ x, long n4, int C4, int Cs, unsigned* __restrict__ slot)
{
    unsigned m = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C4;
        const int c4 = (int)(i - pix * C4);
        const float4 v = *reinterpret_cast<const float4*>(x + pix * Cs + 4 * c4);
        m = max(m, RANGE_OF4(v));
    }
    range_note_block(slot, m, blockIdx.x);
}

// The plan's slots back to zero at the start of a run.  (A kernel of its own, not hipMemsetAsync: as a node of a captured graph the
// runtime's memset filled parts of the table with a stale 16-byte pattern on some replays -- two device pointers alternating, surviving
// from run to run -- which the readers then took for a NaN range.)
__global__ __launch_bounds__(256) void range_clear_kernel(uint4* table, long n16)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) table[i] = make_uint4(0u, 0u, 0u, 0u);
}

hipError_t launch_range_clear(unsigned* table, int n_slots, hipStream_t st)
{
    const long n16 = (long)n_slots * RANGE_WORDS / 4;
    if (n16 > 0) hipLaunchKernelGGL(range_clear_kernel, dim3(cdiv(n16, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(table), n16);
    return hipGetLastError();
}

// word 0 of a slot = the maximum of its partial words (and of what word 0 held): one block, in front of the first reader after a write.
// A non-finite maximum is reported here, once, through the host-mapped flag: ONE word -- (the reader's op index + 1) | bit 31 for a NaN --
// so that the host, which reads it without a stream wait at the start of the next run, never sees half a report (ACCEL_ERR_RANGE).
__global__ __launch_bounds__(256) void range_fold_kernel(unsigned* slot, unsigned* rflag, int op_index)
{
    __shared__ unsigned sm[4];
    const uint4 v = reinterpret_cast<const uint4*>(slot + RANGE_PART_OFF)[threadIdx.x];      // RANGE_PART = 4 x 256
    const unsigned w0 = threadIdx.x == 0 ? slot[0] : 0u;      // (requested with the partial words: not a second round trip behind the barrier)
    unsigned m = range_wave_max(max(max(v.x, v.y), max(v.z, v.w)));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(max(sm[0], sm[1]), max(sm[2], sm[3])), w0);
        slot[0] = m;
        if (m >= 0x7F800000u && rflag) { atomicCAS(rflag, 0u, ((unsigned)op_index + 1u) | (m > 0x7F800000u ? 0x80000000u : 0u)); __threadfence_system(); }
    }
}

hipError_t launch_range_fold(unsigned* slot, unsigned* rflag, int op_index, hipStream_t st)
{
    static_assert(RANGE_PART == 1024, "range_fold_kernel reads four words per thread");
    hipLaunchKernelGGL(range_fold_kernel, dim3(1), dim3(256), 0, st, slot, rflag, op_index);
    return hipGetLastError();
}

hipError_t launch_range_amax(const float* x, long pixels, int C, int Cs, unsigned* slot, hipStream_t st)
{
    const long n4 = pixels * (C / 4);
    const int blocks = (int)std::min<long>(4096, (n4 + 1023) / 1024);
    if (blocks > 0) hipLaunchKernelGGL(range_amax_kernel, dim3(blocks), dim3(256), 0, st, x, n4, C / 4, Cs, slot);
    return hipGetLastError();
}
