/*
 * accel_oracle.c -- CPU restatement of the MXNet operators that make up the
 * Accel (dff_deeplab) inference path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The product path
 * (accel_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (SamvitJ/Accel) holds no tests, golden
 * vectors or checkpoints for this path, and its arithmetic lives in an
 * un-vendored third-party dependency (Apache MXNet @ commit 62ecb60 + cuDNN,
 * named only in prose at /root/reference/README.md:76,105-109), which cannot
 * be built in this image.  Every function below therefore restates the
 * published operator definition of that MXNet version and cites the reference
 * call site that uses it.  Where torch-CPU has an operator of identical
 * semantics the restatement is cross-checked against it in tests/ (that is a
 * consistency check, not a pin against the reference).
 *
 * Layout: fp32, NCHW, exactly what the reference's symbols operate on.
 * Accumulation order is fixed (documented per function) and independent of
 * the OpenMP thread count, so results are reproducible bit for bit.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#define IDX4(n, c, h, w, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (h)) * (W) + (w))

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

/* ------------------------------------------------------------------------
 * mx.symbol.Convolution  (cross-correlation, num_group = 1)
 * call sites: resnet_v1_101_flownet_deeplab.py:52-86,108-128,134-168,236-572,
 *             577-1298,1754-1803; accel_18.py:181-191,211-221,234
 * out = floor((in + 2p - d(k-1) - 1)/s) + 1; weight (Cout, Cin, kh, kw).
 * Accumulation order per output element: (ci, ky, kx) ascending, after the
 * bias initialisation.
 * ---------------------------------------------------------------------- */
void orc_conv2d(const float* x, int N, int C, int H, int W,
                const float* w, const float* b, int K,
                int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                float* y)
{
    const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
    const int Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            float* yo = y + IDX4(n, k, 0, 0, K, Ho, Wo);
            const float b0 = b ? b[k] : 0.0f;
            for (int i = 0; i < Ho * Wo; ++i) yo[i] = b0;
            for (int c = 0; c < C; ++c) {
                const float* xc = x + IDX4(n, c, 0, 0, C, H, W);
                for (int i = 0; i < kh; ++i)
                    for (int j = 0; j < kw; ++j) {
                        const float wv = w[(((size_t)k * C + c) * kh + i) * kw + j];
                        /* valid ox range: 0 <= ox*sw - pw + j*dw < W */
                        const int off = j * dw - pw;
                        int ox_lo = off >= 0 ? 0 : ceil_div(-off, sw);
                        int ox_hi = (W - 1 - off) >= 0 ? (W - 1 - off) / sw + 1 : 0;
                        if (ox_hi > Wo) ox_hi = Wo;
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * sh - ph + i * dh;
                            if (iy < 0 || iy >= H) continue;
                            const float* xr = xc + (size_t)iy * W + off;
                            float* yr = yo + (size_t)oy * Wo;
                            if (sw == 1) {
                                for (int ox = ox_lo; ox < ox_hi; ++ox) yr[ox] += wv * xr[ox];
                            } else {
                                for (int ox = ox_lo; ox < ox_hi; ++ox) yr[ox] += wv * xr[ox * sw];
                            }
                        }
                    }
            }
        }
}

/* ------------------------------------------------------------------------
 * mx.symbol.Deconvolution (transposed convolution, adj = 0, dilate = 1)
 * call sites: resnet_v1_101_flownet_deeplab.py:1775-1799 (4x4 s2 p0, bias);
 *             accel_18.py:204-206 (4x4 s2 p1, no bias);
 *             accel_18.py:193-195,223-225 (32x32 s16, num_group=19, no bias)
 * out = s(in-1) + k - 2p; weight (Cin, Cout/g, kh, kw);
 * out[co, s*i+ky-p, s*j+kx-p] += in[ci,i,j] * w[ci, co_in_group, ky, kx].
 * Accumulation order per output element: (ci, ky, kx) ascending.
 * ---------------------------------------------------------------------- */
void orc_deconv2d(const float* x, int N, int C, int H, int W,
                  const float* w, const float* b, int K, int groups,
                  int kh, int kw, int sh, int sw, int ph, int pw,
                  float* y)
{
    const int Ho = sh * (H - 1) + kh - 2 * ph;
    const int Wo = sw * (W - 1) + kw - 2 * pw;
    const int Cg = C / groups, Kg = K / groups;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            float* yo = y + IDX4(n, k, 0, 0, K, Ho, Wo);
            const float b0 = b ? b[k] : 0.0f;
            for (int i = 0; i < Ho * Wo; ++i) yo[i] = b0;
            const int g = k / Kg, kk = k % Kg;
            for (int c = g * Cg; c < (g + 1) * Cg; ++c) {
                const float* xc = x + IDX4(n, c, 0, 0, C, H, W);
                for (int i = 0; i < kh; ++i)
                    for (int j = 0; j < kw; ++j) {
                        const float wv = w[(((size_t)c * Kg + kk) * kh + i) * kw + j];
                        for (int iy = 0; iy < H; ++iy) {
                            const int oy = iy * sh + i - ph;
                            if (oy < 0 || oy >= Ho) continue;
                            float* yr = yo + (size_t)oy * Wo;
                            const float* xr = xc + (size_t)iy * W;
                            for (int ix = 0; ix < W; ++ix) {
                                const int ox = ix * sw + j - pw;
                                if (ox < 0 || ox >= Wo) continue;
                                yr[ox] += wv * xr[ix];
                            }
                        }
                    }
            }
        }
}

/* ------------------------------------------------------------------------
 * mx.contrib.symbol.DeformableConvolution (DCN v1), no bias, num_group = 1
 * call sites: resnet_v1_101_flownet_deeplab.py:1232-1237,1257-1262,1282-1287
 *             (dg=1) and :144-148,160-164,186-226,515-562 (dg=4)
 * Restates MXNet's deformable_im2col (src/operator/contrib/nn/
 * deformable_im2col.cuh of the DCN-v1 era) followed by a GEMM:
 *   offset channel 2*(i*kw+j)   = dy, 2*(i*kw+j)+1 = dx inside each deformable
 *   group (Cin/dg consecutive input channels share one group);
 *   sample at h_im = oy*s - p + i*d + dy, w_im likewise;
 *   value is 0 unless 0 <= h_im < H and 0 <= w_im < W;
 *   bilinear with the high neighbour clamped: floor(h) >= H-1 -> both rows
 *   = H-1 and the fractional part is dropped (same for columns);
 *   the interpolation runs in coordinates relative to (h_in, w_in), as the
 *   MXNet kernel does.
 * Accumulation order per output element: (ci, ky, kx) ascending.
 * ---------------------------------------------------------------------- */
static float dcn_bilinear(const float* base, int data_width, int height, int width,
                          float h, float w)
{
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high, w_high;
    if (h_low >= height - 1) { h_high = h_low = height - 1; h = (float)h_low; }
    else h_high = h_low + 1;
    if (w_low >= width - 1) { w_high = w_low = width - 1; w = (float)w_low; }
    else w_high = w_low + 1;
    const float lh = h - h_low, lw = w - w_low;
    const float hh = 1 - lh, hw = 1 - lw;
    const float v1 = base[(ptrdiff_t)h_low * data_width + w_low];
    const float v2 = base[(ptrdiff_t)h_low * data_width + w_high];
    const float v3 = base[(ptrdiff_t)h_high * data_width + w_low];
    const float v4 = base[(ptrdiff_t)h_high * data_width + w_high];
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* columns: (C*kh*kw, Ho*Wo), row index = (c*kh + i)*kw + j  -- exposed so the
 * tests can check the sampling stage on its own. */
void orc_deform_im2col(const float* x, int C, int H, int W, const float* offset,
                       int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                       int dg, float* col)
{
    const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
    const int Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
    const int cpg = C / dg;
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < C; ++c) {
        const int g = c / cpg;
        const float* off_g = offset + (size_t)g * 2 * kh * kw * Ho * Wo;
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const int h_in = oy * sh - ph, w_in = ox * sw - pw;
                const float* im = x + ((ptrdiff_t)c * H + h_in) * W + w_in;
                for (int i = 0; i < kh; ++i)
                    for (int j = 0; j < kw; ++j) {
                        const float oh = off_g[((size_t)(2 * (i * kw + j)) * Ho + oy) * Wo + ox];
                        const float ow = off_g[((size_t)(2 * (i * kw + j) + 1) * Ho + oy) * Wo + ox];
                        float val = 0.0f;
                        const float h_im = h_in + i * dh + oh;
                        const float w_im = w_in + j * dw + ow;
                        if (h_im >= 0 && w_im >= 0 && h_im < H && w_im < W) {
                            const float map_h = i * dh + oh, map_w = j * dw + ow;
                            val = dcn_bilinear(im, W, H - h_in, W - w_in, map_h, map_w);
                        }
                        col[(((size_t)c * kh + i) * kw + j) * Ho * Wo + (size_t)oy * Wo + ox] = val;
                    }
            }
    }
}

void orc_deform_conv2d(const float* x, int N, int C, int H, int W, const float* offset,
                       const float* w, int K, int kh, int kw, int sh, int sw,
                       int ph, int pw, int dh, int dw, int dg,
                       float* col_ws /* C*kh*kw*Ho*Wo floats */, float* y)
{
    const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
    const int Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
    const size_t P = (size_t)Ho * Wo;
    const int R = C * kh * kw;
    for (int n = 0; n < N; ++n) {
        orc_deform_im2col(x + IDX4(n, 0, 0, 0, C, H, W), C, H, W,
                          offset + (size_t)n * dg * 2 * kh * kw * P,
                          kh, kw, sh, sw, ph, pw, dh, dw, dg, col_ws);
#pragma omp parallel for schedule(dynamic, 1)
        for (int k = 0; k < K; ++k) {
            float* yo = y + IDX4(n, k, 0, 0, K, Ho, Wo);
            for (size_t i = 0; i < P; ++i) yo[i] = 0.0f;
            for (int r = 0; r < R; ++r) {
                const float wv = w[(size_t)k * R + r];
                const float* cr = col_ws + (size_t)r * P;
                for (size_t i = 0; i < P; ++i) yo[i] += wv * cr[i];
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * mx.symbol.BatchNorm, inference (use_global_stats / is_train=False)
 * call sites: every conv of the ResNets, e.g. resnet_v1_101_flownet_deeplab.py
 *             :50-75 (eps 2e-5), :108 (fix_gamma=True), :579 (eps 1e-5)
 * MXNet's mshadow inference expression:
 *   out = data * (g / sqrt(var + eps)) + (beta - g * mean / sqrt(var + eps)),
 *   g = 1 when fix_gamma.
 * scale/shift are produced here once so that callers (and the HIP path's
 * host-side folding) use the very same two numbers per channel.
 * ---------------------------------------------------------------------- */
void orc_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                 int C, float eps, int fix_gamma, float* scale, float* shift)
{
    for (int c = 0; c < C; ++c) {
        const float g = fix_gamma ? 1.0f : gamma[c];
        const float sd = sqrtf(var[c] + eps);
        scale[c] = g / sd;
        shift[c] = beta[c] - (g * mean[c]) / sd;
    }
}

void orc_scale_shift(const float* x, int N, int C, int HW, const float* scale,
                     const float* shift, float* y)
{
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float* xi = x + ((size_t)n * C + c) * HW;
            float* yo = y + ((size_t)n * C + c) * HW;
            const float a = scale[c], b = shift[c];
            for (int i = 0; i < HW; ++i) yo[i] = xi[i] * a + b;
        }
}

/* ------------------------------------------------------------------------
 * mx.symbol.Pooling, pool_type max|avg, pooling_convention valid|full
 * call sites: resnet_v1_101_flownet_deeplab.py:582,241 (max 3x3 s2 'full' p0),
 *             :117 (max 3x3 s2 'valid' p1), :1753,1802 (avg 2x2 s2 'full')
 * valid: out = floor((in + 2p - k)/s) + 1 ; full: out = ceil((in + 2p - k)/s) + 1
 * max ignores padding; avg divides by the window area clipped to
 * [ -p, in + p ) (MXNet pool.h pool_sum_2d_cpu); for the even sizes of every
 * BASELINE config both avg pools are exact 4-element means.
 * ---------------------------------------------------------------------- */
void orc_pool2d(const float* x, int N, int C, int H, int W, int is_max, int full,
                int kh, int kw, int sh, int sw, int ph, int pw, float* y)
{
    int Ho, Wo;
    if (full) {
        Ho = 1 + ceil_div(H + 2 * ph - kh, sh);
        Wo = 1 + ceil_div(W + 2 * pw - kw, sw);
    } else {
        Ho = 1 + (H + 2 * ph - kh) / sh;
        Wo = 1 + (W + 2 * pw - kw) / sw;
    }
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float* xc = x + IDX4(n, c, 0, 0, C, H, W);
            float* yo = y + IDX4(n, c, 0, 0, C, Ho, Wo);
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    int hs = oy * sh - ph, ws = ox * sw - pw;
                    int he = hs + kh, we = ws + kw;
                    if (he > H + ph) he = H + ph;
                    if (we > W + pw) we = W + pw;
                    const int area = (he - hs) * (we - ws);
                    if (hs < 0) hs = 0;
                    if (ws < 0) ws = 0;
                    if (he > H) he = H;
                    if (we > W) we = W;
                    float acc = is_max ? -INFINITY : 0.0f;
                    for (int iy = hs; iy < he; ++iy)
                        for (int ix = ws; ix < we; ++ix) {
                            const float v = xc[(size_t)iy * W + ix];
                            if (is_max) { if (v > acc) acc = v; }
                            else acc += v;
                        }
                    yo[(size_t)oy * Wo + ox] = is_max ? acc : acc / (float)area;
                }
        }
}

int orc_pool_out(int in, int k, int s, int p, int full)
{
    return full ? 1 + ceil_div(in + 2 * p - k, s) : 1 + (in + 2 * p - k) / s;
}

/* ------------------------------------------------------------------------
 * mx.sym.GridGenerator(transform_type='warp')   -- accel_18.py:174
 *   grid_x = (x + flow_x) / ((W-1)/2) - 1 ; grid_y = (y + flow_y) / ((H-1)/2) - 1
 *   flow channel 0 = x displacement, channel 1 = y displacement.
 * ---------------------------------------------------------------------- */
void orc_grid_generator_warp(const float* flow, int N, int H, int W, float* grid)
{
    const float sx = (float)(W - 1) / 2.0f, sy = (float)(H - 1) / 2.0f;
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t ix = IDX4(n, 0, y, x, 2, H, W), iy = IDX4(n, 1, y, x, 2, H, W);
                grid[ix] = (flow[ix] + (float)x) / sx - 1.0f;
                grid[iy] = (flow[iy] + (float)y) / sy - 1.0f;
            }
}

/* ------------------------------------------------------------------------
 * mx.sym.BilinearSampler   -- accel_18.py:175
 *   x_real = (gx + 1) * (W-1) / 2 ; y_real likewise; 4-tap bilinear; a tap
 *   outside [0,W-1]x[0,H-1] contributes 0 (MXNet bilinear_sampler.cc
 *   BilinearSamplerForward, `between(...)` guards).
 * ---------------------------------------------------------------------- */
void orc_bilinear_sampler(const float* data, int N, int C, int H, int W,
                          const float* grid, int Ho, int Wo, float* out)
{
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float* d = data + IDX4(n, c, 0, 0, C, H, W);
            for (int h = 0; h < Ho; ++h)
                for (int w = 0; w < Wo; ++w) {
                    const float gx = grid[IDX4(n, 0, h, w, 2, Ho, Wo)];
                    const float gy = grid[IDX4(n, 1, h, w, 2, Ho, Wo)];
                    const float y_real = (gy + 1) * (H - 1) / 2;
                    const float x_real = (gx + 1) * (W - 1) / 2;
                    const int ty = (int)floorf(y_real), tx = (int)floorf(x_real);
                    const float wy = 1.0f - (y_real - ty), wx = 1.0f - (x_real - tx);
                    float tl = 0, tr = 0, bl = 0, br = 0;
                    const int x0 = tx >= 0 && tx <= W - 1, x1 = tx + 1 >= 0 && tx + 1 <= W - 1;
                    const int y0 = ty >= 0 && ty <= H - 1, y1 = ty + 1 >= 0 && ty + 1 <= H - 1;
                    if (x0 && y0) tl = d[(ptrdiff_t)ty * W + tx];
                    if (x1 && y0) tr = d[(ptrdiff_t)ty * W + tx + 1];
                    if (x0 && y1) bl = d[(ptrdiff_t)(ty + 1) * W + tx];
                    if (x1 && y1) br = d[(ptrdiff_t)(ty + 1) * W + tx + 1];
                    out[IDX4(n, c, h, w, C, Ho, Wo)] =
                        tl * wy * wx + tr * wy * (1.0f - wx) +
                        bl * (1.0f - wy) * wx + br * (1.0f - wy) * (1.0f - wx);
                }
        }
}

/* mx.ndarray.argmax(axis=1) as used at demo.py:238,245: first maximal index,
 * returned as float by MXNet then cast to uint8 (demo.py:252). */
void orc_argmax_c(const float* x, int N, int C, int HW, unsigned char* out)
{
#pragma omp parallel for
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < HW; ++i) {
            int best = 0;
            float bv = x[((size_t)n * C) * HW + i];
            for (int c = 1; c < C; ++c) {
                const float v = x[((size_t)n * C + c) * HW + i];
                if (v > bv) { bv = v; best = c; }
            }
            out[(size_t)n * HW + i] = (unsigned char)best;
        }
}
