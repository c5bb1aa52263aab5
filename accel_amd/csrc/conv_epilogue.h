// Fused epilogue shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_b3r.hip): split-K partial store
// or scale/shift -> +residual -> activation -> store (+ dual output), from 32x32 MFMA accumulators.
#pragma once
#include "kernels.h"
#include "conv_common.h"
#include "range.h"
#include <type_traits>

// Shared tail of both conv kernels: split-K partial store or the fused epilogue
// (scale/shift -> +residual -> activation -> store, optional dual output).
// RES_PER_J: the residual values are fetched one 32-column block ahead of its stores instead of all at once (tiles with 128
// accumulator registers per lane cannot hold a second copy of them)
// xinv: fp16x2 form -- the inverse of the pixel scale the kernel derived from its input's range slot (an exact power of two)
// Range slots (range.h): where p.yr / p.y2r are set, the largest |value| this wavefront STORED (rows past M and channels past
// Cout_store do not count) is raised into them at the end.
template <int MI, int NI, int WGN, bool RES_PER_J = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[MI][NI], int m0, int n0, int wm, int wn,
                                              int lane, int py, int px, int HoWo, float xinv = 1.f)
{
    // ---- output coordinates of this lane's 16*MI accumulator rows ------------------
    // C/D layout of the 32x32 MFMA: col = lane & 31 (-> co), row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int rbase = m0 + wm * MI * 32 + 4 * (lane >> 5);

    // ---- split-K: raw partial sums to the workspace [split][class][M][Cout_store] ----
    if (p.ksplit > 1) {
        const size_t slab = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * (size_t)p.M * p.Cout_store;
        const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.ws + slab, (unsigned)((size_t)p.M * p.Cout_store * 4));
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int co = n0 + (wn * NI + j) * 32 + (lane & 31);
            const bool cok = co < p.Cout_store;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = rbase + i * 32 + (e & 3) + 8 * (e >> 2);
                    buf_store1(wr, (cok && m < p.M) ? (unsigned)((m * p.Cout_store + co) * 4) : OOB, acc[i][j][e]);
                }
        }
        return;
    }

    // ---- fused epilogue: scale/shift (+residual) -> activation -> store (+ dual output) ----
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res : p.y, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t y2r = make_rsrc(p.y2 ? p.y2 : p.y, p.y2 ? p.y2_bytes : 0u);
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
    auto pixel_of = [&](int m) -> unsigned {      // NHWC pixel index of GEMM row m, OOB past M
        if (m >= p.M) return OOB;
        if (!p.deconv2x) return (unsigned)m;
        int n, rem, oy, ox;
        divmod_small(m, HoWo, inv_howo, n, rem);
        divmod_small(rem, p.Wo, inv_wo, oy, ox);
        const int yy = 2 * oy + py, xx = 2 * ox + px;
        if (yy >= p.yH || xx >= p.yW) return OOB;      // output cropped to 2h-1 / 2w-1 (Crop after a pad-0 deconvolution)
        return (unsigned)((n * p.yH + yy) * p.yW + xx);
    };
    // Vector-memory operations retire IN ORDER through one counter: a residual load issued behind an output store cannot
    // be consumed before that store has been acknowledged.  Loading the residual group by group between the stores
    // (the obvious loop) therefore serialises 16*MI*NI/4 store round trips per wavefront -- measured as 4-11 us of
    // epilogue on 42 us residual layers.  So: every per-channel constant and EVERY residual value is fetched before the
    // first store is issued; after that the epilogue only computes and stores.
    float sc[NI], sf[NI], sc2[NI], sf2[NI];
    bool cok[NI];
    int co[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        co[j] = n0 + (wn * NI + j) * 32 + (lane & 31);
        cok[j] = co[j] < p.Cout_store;
        const int cc = cok[j] ? co[j] : 0;
        sc[j] = p.scale[cc] * xinv; sf[j] = p.shift[cc];
        sc2[j] = p.y2 ? p.scale2[cc] : 1.f; sf2[j] = p.y2 ? p.shift2[cc] : 0.f;
    }
    if (!RES_PER_J && p.res) {
        float rv[MI][NI][16];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned px_ = pixel_of(rbase + i * 32 + (e & 3) + 8 * (e >> 2));
                    rv[i][j][e] = buf_load1(rr, (cok[j] && px_ != OOB) ? (px_ * p.resCs + co[j]) * 4u : OOB);
                }
        // accumulators become acc*scale + shift + residual here, so the store loop below is shared with the plain case
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = acc[i][j][e] * sc[j] + sf[j] + rv[i][j][e];
    }
    // ---- range slots of the outputs (range.h) ----
    // A pass of its own over the accumulators, BEHIND the store loop and with nothing of it kept live: the values are recomputed from
    // acc (3-5 vector instructions each).  Woven into the store loop the same bookkeeping cost 30-40 registers per lane -- the
    // 128x128 tile went from 94 to 128 registers with scratch spills and lost a resident block, the 128x64 tile two (round 5, first
    // version: headline 626 -> 571 frames/s).  Rows past M do not count (masked where the tile is ragged), channels past Cout_store
    // neither; rows a cropped deconvolution drops (odd sizes) do: they are outputs of the same layer, the slot stays a bound and a
    // function of the run's data.
    // (RES_PER_J tiles -- the f16-mode kernels of conv_b3d.hip, 128-256 accumulator registers per lane -- carry no range epilogue at
    // all: with it they went from 163 to 256 registers and spilled 200; the library knows, and a reader of their output measures its
    // view itself: accel_hip.cpp resolve_range_flags)
#ifdef RANGE_AB_NO_NOTE
    const bool note = false, note2 = false;
#else
    const bool note = !RES_PER_J && p.yr != nullptr, note2 = !RES_PER_J && p.y2 && p.y2r != nullptr;
#endif
    unsigned rmax = 0u, rmax2 = 0u;
    // Branch-free, whatever the layer's constants, and 1.5 vector instructions per value: v = acc * a + b (a = 1, b = 0 behind a
    // residual: the accumulators already hold the pre-activation value; two values per packed multiply-add), a running MAXIMUM and a
    // running MINIMUM of v (three-operand forms: two values per instruction each).  Everything else follows from the two extremes,
    // once per column block: |act(v)| = max(v, c v) with c = 0 after a ReLU, -slope after a leaky one, -1 without an activation -- all
    // <= 0, so its maximum is max(vmax, c vmin); the second output relu(act(v) scale2 + shift2) is a monotone map of v followed by a
    // linear one, so its maximum sits at one of the two extremes.  (With the layer's constants tested inside the loop the pass compiled
    // to a scalar branch tree per value: 8400 lines of ISA for the 128x128 tile, 2.5 ms of a 64 ms step; as a switch over
    // straight-line variants it spilled 27 registers; per value with |act| taken inside 1.0 ms.  A NaN drops out of a floating-point
    // maximum: it reaches the outputs through the matrix instructions as in fp32 arithmetic; an infinity stays visible and is
    // reported by the fold.)
    auto note_tile = [&]() __attribute__((always_inline)) {
        const float c = p.act == 1 ? 0.f : p.act == 2 ? -p.slope : -1.f;
        const float d = p.act == 1 ? 0.f : p.act == 2 ? p.slope : 1.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const f32x2 a2 = {p.res ? 1.f : sc[j], p.res ? 1.f : sc[j]}, b2 = {p.res ? 0.f : sf[j], p.res ? 0.f : sf[j]};
            float vmax = -__builtin_inff(), vmin = __builtin_inff();
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 v = f32x2{acc[i][j][e], acc[i][j][e + 1]} * a2 + b2;
                    vmax = fmaxf(fmaxf(vmax, v[0]), v[1]);
                    vmin = fminf(fminf(vmin, v[0]), v[1]);
                }
            if (cok[j]) {
                const float mj = fmaxf(fmaxf(vmax, c * vmin), 0.f);
                const float r1 = fmaxf(vmax, d * vmax), r0 = fmaxf(vmin, d * vmin);      // act() at the two extremes
                const float mj2 = fmaxf(fmaxf(r1 * sc2[j] + sf2[j], r0 * sc2[j] + sf2[j]), 0.f);
                const unsigned bb = __builtin_bit_cast(unsigned, mj), b2_ = __builtin_bit_cast(unsigned, mj2);
                rmax = bb > rmax ? bb : rmax; rmax2 = b2_ > rmax2 ? b2_ : rmax2;
            }
        }
    };
    auto note_general = [&](int j) __attribute__((always_inline)) {      // one column block (RES_PER_J tiles: behind that block's residual add)
        const float c = p.act == 1 ? 0.f : p.act == 2 ? -p.slope : -1.f;
        const float d = p.act == 1 ? 0.f : p.act == 2 ? p.slope : 1.f;
        const float a = p.res ? 1.f : sc[j], b = p.res ? 0.f : sf[j];
        float mj = 0.f, mj2 = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e] * a + b;
                if (rbase + i * 32 + (e & 3) + 8 * (e >> 2) >= p.M) v = 0.f;
                mj = fmaxf(mj, fmaxf(v, v * c));
                mj2 = fmaxf(mj2, fmaxf(v, v * d) * sc2[j] + sf2[j]);
            }
        if (cok[j]) {
            const unsigned bb = __builtin_bit_cast(unsigned, mj), b2 = __builtin_bit_cast(unsigned, mj2);
            rmax = bb > rmax ? bb : rmax; rmax2 = b2 > rmax2 ? b2 : rmax2;
        }
    };
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        if constexpr (RES_PER_J) {
            if (p.res) {
                float rv[MI][16];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const unsigned px_ = pixel_of(rbase + i * 32 + (e & 3) + 8 * (e >> 2));
                        rv[i][e] = buf_load1(rr, (cok[j] && px_ != OOB) ? (px_ * p.resCs + co[j]) * 4u : OOB);
                    }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = acc[i][j][e] * sc[j] + sf[j] + rv[i][e];
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {          // 4 rows at a time keeps the live set small
                unsigned pix[4];
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned px_ = pixel_of(rbase + i * 32 + e + 8 * h);
                    pix[e] = cok[j] ? px_ : OOB;
                    v[e] = p.res ? acc[i][j][h * 4 + e] : acc[i][j][h * 4 + e] * sc[j] + sf[j];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.act == 1) v[e] = fmaxf(v[e], 0.f);
                    else if (p.act == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
                    buf_store1(yr, pix[e] != OOB ? (pix[e] * p.yCs + co[j]) * 4u : OOB, v[e]);
                }
                if (p.y2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        buf_store1(y2r, pix[e] != OOB ? (pix[e] * p.y2Cs + co[j]) * 4u : OOB, fmaxf(v[e] * sc2[j] + sf2[j], 0.f));
                }
            }
        }
    }
    // (behind the stores: the pass runs while they drain; the accumulators are still there, the values are recomputed from them)
#ifdef RANGE_AB_NO_VALU      // timing experiment: the atomics without the pass over the accumulators
    rmax = rmax2 = 0x3F800000u;
#else
    if constexpr (!RES_PER_J) {
        if (note || note2) {
            if (rbase - 4 * (lane >> 5) + MI * 32 > p.M) {      // wave-uniform: some rows of this wavefront lie past M -- value by value, masked
#pragma unroll
                for (int j = 0; j < NI; ++j) note_general(j);
            } else note_tile();
        }
    }
#endif
    // one atomic per wavefront into the partial word its index picks (1024 of them: no two wavefronts in flight meet)
    const unsigned key = (blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (note) range_note_wave(p.yr, rmax, key);
    if (note2) range_note_wave(p.y2r, rmax2, key);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Epilogue of the layers of an f16-mode plan that store / read HALF activations (conv_b3d.hip, NPL = 1; ConvParams::y_half,
// res_half).  Same function as conv_epilogue -- split-K partials (fp32) or scale / shift -> + residual -> activation -> store -- with
// the stored value rounded to half (RTNE) AFTER the whole epilogue, and a residual that may itself be stored as half.
// A lane of the 32x32 accumulator holds one channel of 16 rows, its neighbour lane ^ 1 the next channel: for every pair of rows the
// two lanes exchange one value each (one DPP row-xmask move), so that the even lane stores the channel pair of the first row and
// the odd lane the pair of the second row as ONE dword -- half as many store instructions as a 2-byte store per value would need,
// each covering whole 64-byte runs; the half residual is fetched the same way (one dword per lane and row pair) and exchanged back.
// No dual output (the plan keeps such layers in fp32).
// No range epilogue (range.h): half outputs have no fp16x2-form reader, and a reader of an fp32 output written from here measures its view
// itself (the library treats the conv_b3d.hip geometries as writers without the epilogue).
template <int MI, int NI, int WGN>
__device__ __forceinline__ void conv_epilogue_h(const ConvParams& p, f32x16 (&acc)[MI][NI], int m0, int n0, int wm, int wn,
                                                int lane, int py, int px, int HoWo)
{
    if (p.ksplit > 1 || (!p.y_half && !p.res_half)) {      // raw partial sums / an all-fp32 epilogue: the shared code
        conv_epilogue<MI, NI, WGN, true>(p, acc, m0, n0, wm, wn, lane, py, px, HoWo);
        return;
    }

    const int rbase = m0 + wm * MI * 32 + 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res : p.y, p.res ? p.res_bytes : 0u);
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
    auto pixel_of = [&](int m) -> unsigned {
        if (m >= p.M) return OOB;
        if (!p.deconv2x) return (unsigned)m;
        int n, rem, oy, ox;
        divmod_small(m, HoWo, inv_howo, n, rem);
        divmod_small(rem, p.Wo, inv_wo, oy, ox);
        const int yy = 2 * oy + py, xx = 2 * ox + px;
        if (yy >= p.yH || xx >= p.yW) return OOB;
        return (unsigned)((n * p.yH + yy) * p.yW + xx);
    };
    const bool odd = lane & 1;
    auto swap1 = [](float v) -> float {      // the value of lane ^ 1
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
    };
    auto h2f = [](unsigned bits16) -> float { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16); };
    auto f2h = [](float v) -> unsigned { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v); };
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int co = n0 + (wn * NI + j) * 32 + (lane & 31);
        const bool cok = co < p.Cout_store;      // Cout_store is a multiple of 4: a lane pair (even channel, odd channel) is valid together
        const int cc = cok ? co : 0;
        const float sc = p.scale[cc], sf = p.shift[cc];
        const int cpair = co & ~1;               // first channel of this lane pair
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // rows of this lane: r(e) = rbase + i*32 + (e & 3) + 8 * (e >> 2); row pairs (e, e + 1), e even
            float rv[16];
            if (p.res) {
                if (p.res_half) {
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        // even lane fetches {res[row e][c], res[row e][c+1]}, odd lane {res[row e+1][c-1], res[row e+1][c]}
                        const unsigned px_ = pixel_of(rbase + i * 32 + ((e + (odd ? 1 : 0)) & 3) + 8 * (e >> 2));
                        const unsigned d = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rr, (cok && px_ != OOB) ? (px_ * p.resCs + cpair) * 2u : OOB, 0, 0);
                        const unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)d, 0xB1, 0xF, 0xF, true);
                        // even lane (channel c): row e = lo(own), row e+1 = lo(partner); odd lane (channel c+1): row e = hi(partner), row e+1 = hi(own)
                        rv[e] = odd ? h2f(o >> 16) : h2f(d & 0xFFFFu);
                        rv[e + 1] = odd ? h2f(d >> 16) : h2f(o & 0xFFFFu);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const unsigned px_ = pixel_of(rbase + i * 32 + (e & 3) + 8 * (e >> 2));
                        rv[e] = buf_load1(rr, (cok && px_ != OOB) ? (px_ * p.resCs + co) * 4u : OOB);
                    }
                }
            }
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[e] = acc[i][j][e] * sc + sf + (p.res ? rv[e] : 0.f);
                if (p.act == 1) v[e] = fmaxf(v[e], 0.f);
                else if (p.act == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
            }
            if (p.y_half) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    // even lane writes row e: {own v[e], partner's v[e]}; odd lane writes row e+1: {partner's v[e+1], own v[e+1]}
                    const float got = swap1(odd ? v[e] : v[e + 1]);
                    const unsigned w = odd ? (f2h(got) | (f2h(v[e + 1]) << 16)) : (f2h(v[e]) | (f2h(got) << 16));
                    const unsigned px_ = pixel_of(rbase + i * 32 + ((e + (odd ? 1 : 0)) & 3) + 8 * (e >> 2));
                    __builtin_amdgcn_raw_buffer_store_b32((int)w, yr, (cok && px_ != OOB) ? (px_ * p.yCs + cpair) * 2u : OOB, 0, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned px_ = pixel_of(rbase + i * 32 + (e & 3) + 8 * (e >> 2));
                    buf_store1(yr, (cok && px_ != OOB) ? (px_ * p.yCs + co) * 4u : OOB, v[e]);
                }
            }
        }
    }
}
