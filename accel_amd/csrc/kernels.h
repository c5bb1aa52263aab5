// Internal kernel-launch interface of libaccel_hip (gfx950 only).
// Activations are fp32 NHWC with a channel stride `Cs` (multiple of 4) that
// may exceed the logical channel count (concat views, padded channels).
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stddef.h>

struct ConvParams {
    const float* x;        // NHWC input view (channel offset already applied)
    const float* w;        // packed weights [class][Cout_pad][K_pad]
    float* y;              // NHWC output view
    float* y2;             // optional second output: relu(v*scale2+shift2)
    const float* res;      // optional residual, added before the activation
    const float* scale; const float* shift;    // per output channel, length >= Cout_store
    const float* scale2; const float* shift2;
    int H, W;              // input spatial size
    int Cin;               // channels per tap (multiple of 4)
    int xCs;               // input channel stride
    int Ho, Wo;            // GEMM pixel grid (deconv2x: = input grid)
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int K_pad;             // flattened K (kh*kw*Cin) rounded up to 32
    int Cout_store;        // channels written (pad channels get exact zeros)
    int Cout;              // real output channels (<= Cout_store; the rows beyond it in the packed weights are zero)
    int yCs, y2Cs, resCs;
    int M;                 // N*Ho*Wo
    int act;               // 0 none, 1 relu, 2 leaky
    float slope;
    int deconv2x;          // 4x4 s2 p1 transposed conv as 4 sub-pixel 2x2 convs
    int yH, yW;            // output spatial size (deconv2x addressing)
    size_t w_class_stride; // floats between parity classes
    int MT, NT;            // filled by the launcher
    int force_tile;        // -1 = heuristic
    int ksplit;            // >1: split-K over blockIdx.z, partials in ws, reduce+epilogue kernel follows
    int kt_per_split;
    int no_split;
    int f16;               // 1: weights packed as half, loader converts activations: fp16 MFMA, fp32 accumulate;
                           // 2: "bf16x3" -- fp32 operands split exactly into three bf16 terms, six bf16 MFMA products (fp32-equivalent)
    size_t w_plane;        // f16 == 2: elements between the three bf16 planes of the packed weights
    int narrow;            // Cout <= 4 plain conv: wave-per-pixel dot-product kernel instead of the MFMA tile
    int split_target;      // >0: split K until the grid has about this many blocks (autotuner)
    float* ws;             // split-K workspace [ksplit][classes][M][Cout_store]
    unsigned x_bytes, y_bytes, y2_bytes, res_bytes;   // extents of the views (buffer-resource bounds)
    unsigned w_bytes;      // bytes of one weight class
    const int4* ktab;      // per K step (and parity class): {dy, dx, input byte offset, 0}; null -> generic path
    // Winograd F(2x2,3x3) variant (conv_wino.hip, tile id 40): host-transformed weights [C/8][16][wino_rows][8]
    const float* wu;       // null: the layer has no Winograd form (or ACCEL_WITHHOLD=winograd)
    unsigned wu_bytes;
    int wino_rows;         // output-channel rows of wu (Cout_store rounded up to the block's 64)
    int wino_T;            // 2x2 output tiles = M / 4, filled by the launcher
    int wino_bhs;          // geometry 42: log2 of the tile-block height (3 / 2 / 1 = 8x8 / 4x16 / 2x32 tiles), filled by the launcher
    // direct 7x7/2 stem variant (conv_stem.hip, tile id 50): weights pre-arranged per lane
    const float* wstem;    // null: not a 3-channel 7x7/2 stem (or ACCEL_WITHHOLD=stem)
    const void* wstemb;    // the same layer on the bf16 matrix cores (conv_stem_b3.hip, tile id 51): three bf16 planes in fragment order; null: not offered
    // weight-stationary streaming 1x1 variant (conv_1x1ws.hip, tile id 60): weights as the LDS image per column group
    // bf16x3 variant of an fp32 layer (launch geometries 70-74): the weights once more, split into three bf16 planes
    const void* wb3;       // null: Cin % 8 != 0, narrow output, or ACCEL_WITHHOLD=split
    const float* wws;      // null: not a 64 -> k*256 / 128 -> k*128 1x1 stride-1 layer (or ACCEL_WITHHOLD=ws1x1)
    unsigned wws_bytes;
    // second-generation bf16x3 kernel (conv_b3r.hip, launch geometries 76, 77, 79, 80, 81): the three bf16 planes once more, in MFMA
    // fragment order [class][K step][half step][row][16] (weights go global -> VGPR, never through LDS)
    const void* wb3r;      // null where wb3 is null (or ACCEL_WITHHOLD=b3r)
    const void* wub;       // conv_wino_b3.hip: U = G g G^T as three bf16 planes [plane][C/16][16][wino_rows][16]; null: not offered
    unsigned wub_bytes;
    // fp16x2 form of the bf16x3 kernels ("h2", ConvParams::f16 == 3 inside the launchers): every fp32 operand as hi + lo, two half
    // terms (hi = RTNE(v), lo = RTNE(v - hi): 22-23 significant bits), THREE v_mfma_f32_32x32x16_f16 products per multiply-add
    // (hi*hi + hi*lo + lo*hi), fp32 accumulate.  Operands are centred in the half range by exact powers of two: the weights per
    // output channel on the host (folded into scale_h2), the pixels by the power of two that the kernel derives IN ITS PROLOGUE from
    // the largest |pixel| of its input tensor -- measured in the same run by whoever produced that tensor (range slots, below) --
    // and undoes in the epilogue.  No state survives a run: the scale is a function of the frame's own data.
    const void* wh2r;      // the two half planes in the fragment order of wb3r; null: not offered
    const float* scale_h2; // scale[] with the weight exponents folded in
    const void* wstemh;    // conv_stem_b3.hip: the stem's fragments as two half planes; null: not offered
    const float* scale_h2s; // scale[] with the stem weights' exponents folded in
    const void* wubh;      // conv_wino_b3.hip / conv_wino_b3s.hip: U as two half planes in the layout of wub; null: not offered
    const float* scale_h2w; // scale[] with U's exponents (and the factor 4 of the quarter-scale V split) folded in
    // Range slots (range.h): a slot is one word the reader takes and RANGE_PART partial words, zero at the start of every run of the
    // plan; every kernel that writes a tensor some fp16x2-form convolution reads raises a partial word to the largest |value| it
    // stored (bit pattern of the absolute value, atomicMax: order-independent, so the result is deterministic); a one-block fold
    // kernel in front of the first reader takes their maximum into word 0.  A reader whose input has a writer without that epilogue
    // (or none inside the plan: a persistent buffer written by another plan or by the host) is preceded by a measuring launch
    // (misc.hip range_amax_kernel) of its view.
    const unsigned* xr;      // input range slot; set by the dispatcher for the fp16x2 launch alone (every other kernel of the layer sees null)
    const unsigned* xr_slot; // the layer's input range slot; null: the layer has no fp16x2 form
    unsigned* yr;            // range slot of the tensor `y` is (part of); null: no fp16x2-form convolution reads it
    unsigned* y2r;           // the same for y2
    // half activation storage (f16-mode plans; conv_b3d.hip NPL = 1 only): the view is stored as half (2 bytes per element, channel
    // strides in elements, x_bytes / y_bytes / res_bytes in bytes); values are rounded (RTNE) when stored, after the whole epilogue
    int x_half, y_half, res_half;
};

hipError_t launch_conv_igemm(const ConvParams& p, hipStream_t st);
int conv_pick_tile(const ConvParams& p);
int conv_tile_bk(int tile);
bool conv_tile_valid(int tile);
#define CONV_TILE_WINO 40
#define CONV_TILE_WINO_B3S 43     // 42 with half the block (32 tiles x 64 channels, 4 wavefronts): two blocks per CU (conv_wino_b3s.hip)
#define CONV_TILE_WINO_B3U 42     // the same with the union of the block's patches loaded once into an LDS copy (tile blocks 8x8 / 4x16 / 2x32)
#define CONV_TILE_WINO_B3 41      // Winograd F(2x2,3x3) on the bf16 matrix cores, three exact bf16 terms per operand (conv_wino_b3.hip)
bool conv_wino_eligible(const ConvParams& p);
int conv_wino_rows(int cout_store);
void conv_wino_pack(const float* w, int Cout, int Cin, int cin_pad, int rows, float* out);
hipError_t launch_conv_wino(const ConvParams& p, hipStream_t st);
bool conv_wino_b3_eligible(const ConvParams& p);
void conv_wino_b3_pack(const float* w, int Cout, int Cin, int rows, std::vector<unsigned short>& out);
void conv_wino_b3_pack_h2(const float* w, int Cout, int Cin, int rows, std::vector<unsigned short>& out, std::vector<int>& qexp);
hipError_t launch_conv_wino_b3(const ConvParams& p, hipStream_t st, bool union_loader = false);
long conv_wino_b3u_blocks(const ConvParams& p, int* bhs);
long conv_wino_b3s_blocks(const ConvParams& p, int* bhs);
hipError_t launch_conv_wino_b3s(const ConvParams& p, hipStream_t st);
hipError_t launch_splitk_reduce(const ConvParams& p, int classes, hipStream_t st);   // sums ws[split][class][M][Cout_store] + epilogue
#define CONV_TILE_STEM 50
#define CONV_TILE_STEM_B3 51      // the stem on the bf16 matrix cores, three exact bf16 terms per operand (conv_stem_b3.hip)
bool conv_stem_b3_eligible(const ConvParams& p);
void conv_stem_b3_pack(const float* w, int Cout, std::vector<unsigned short>& out);
void conv_stem_b3_pack_h2(const float* w, int Cout, std::vector<unsigned short>& out, std::vector<int>& qexp);
hipError_t launch_conv_stem_b3(const ConvParams& p, hipStream_t st);
bool conv_stem_eligible(const ConvParams& p);
void conv_stem_pack(const float* w, int Cout, float* out);
int conv_stem_pack_floats();
hipError_t launch_conv_stem(const ConvParams& p, hipStream_t st);
#define CONV_TILE_HALO 78                // conv_halo.hip: 3x3 / stride 1 / dilation 1 or 2 layers with few output channels, the patch staged once with its halo (fp16x2 form only)
bool conv_halo_eligible(const ConvParams& p);
hipError_t launch_conv_halo(const ConvParams& p, hipStream_t st);
#define CONV_TILE_B3 70                  // 70..75: the bf16x3 kernel (fp32 values, bf16 matrix cores) at geometry 0, 1, 2, 3, 10 and 256x128
#define CONV_TILE_B3R 76                 // conv_b3r.hip: 76 = 128x128 / 2x4 wavefronts, 77 = 128x64 / 2x2, 79 = 128x256 / 2x4, 80 = 128x256 / 1x8, 81 = 128x128 / 1x4
hipError_t launch_conv_b3r(const ConvParams& p, int tile, hipStream_t st);
#define CONV_TILE_B3D 82                 // conv_b3d.hip (f16-mode layers only; both operands by LDS-DMA, fp32 or half views): 82 = 256x256 / 4x2 wavefronts,
                                         // 83 = 128x256 / 4x2, 84 = 128x128 / 4x1, 85 = 128x128 / 2x2, 88 = 128x64 / 4x1, 89 = 256x128 / 4x2 (86 / 87 retired)
#define CONV_TILE_B3D_N 8
bool conv_b3d_eligible(const ConvParams& p);
hipError_t launch_conv_b3d(const ConvParams& p, int tile, hipStream_t st);
#define CONV_TILE_WS 60                  // weight-stationary streaming 1x1 (conv_1x1ws.hip)
bool conv_ws_eligible(const ConvParams& p);
size_t conv_ws_pack_floats(int Cin, int cout_store);
void conv_ws_pack(const float* w, int Cout, int Cin, int cout_store, float* out);
hipError_t launch_conv_ws(const ConvParams& p, hipStream_t st);
size_t conv_plan_split(ConvParams& p);   // sets ksplit/kt_per_split, returns workspace bytes

// ---- bandwidth-bound kernels (misc.hip) ---------------------------------------
// NCHW fp32 3xHxW image -> NHWC4 (c3 = 0); optional per-channel scale/shift (bn_data)
// N images: src image stride 3*H*W, dst image stride 4*H*W
hipError_t launch_prep_rgb(const float* src, float* dst, int H, int W,
                           const float* scale3, const float* shift3, int N, hipStream_t st, const float* const* slot = nullptr, unsigned* yr = nullptr);
// FlowNet input: avgpool2x2(concat(cur/255, prev/255)) -> NHWC8 at H/2 x W/2 (c6,c7 = 0)
hipError_t launch_prep_flow(const float* cur, const float* prev, float* dst, int H, int W,
                            int N, hipStream_t st, const float* const* cur_slot = nullptr, const float* const* prev_slot = nullptr, unsigned* yr = nullptr);
hipError_t launch_set_slot(const void** slot, const void* value, hipStream_t st);

struct PoolParams {
    const float* x; float* y;
    int C4;                // channel quads to process
    int xCs, yCs;
    int H, W, Ho, Wo;
    int kh, kw, sh, sw, ph, pw;
    int is_max;
    const float* scale; const float* shift;   // optional BN epilogue
    int relu;
    int N;                 // images (blockIdx.z); 0 = 1
    size_t x_img, y_img;   // floats between consecutive images
    unsigned* yr;          // range slot of the output tensor (ConvParams::yr); null: nobody needs it
};
hipError_t launch_pool(const PoolParams& p, hipStream_t st);

// GridGenerator(warp)+BilinearSampler fused; flow is NHWC (ch0 = dx, ch1 = dy)
// optional second output out2 = relu(warped + bias[c]) (bias: C floats)
hipError_t launch_flow_warp(const float* feat, int fCs, const float* flow, int flCs,
                            float* out, int oCs, int C, int H, int W,
                            float* out2, int o2Cs, const float* bias, int N, hipStream_t st, unsigned* yr = nullptr, unsigned* y2r = nullptr);

struct DcnColsParams {
    const float* x; const float* off; float* col;
    int C, xCs, offCs, colCs;
    int H, W, Ho, Wo;
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int dg;
    int N;                 // images (blockIdx.z), each H*W*xCs / Ho*Wo*offCs / Ho*Wo*colCs floats apart; 0 = 1
    int col_half;          // the column buffer is stored as half (colCs in elements): values rounded (RTNE) when stored
    unsigned* yr;          // range slot of the column buffer (ConvParams::yr); null: nobody needs it
};
hipError_t launch_dcn_cols(const DcnColsParams& p, hipStream_t st);

struct ScoreTailParams {
    const float* left;  int lCs;     // NHWC scores at H/16 x W/16
    const float* right; int rCs;     // nullptr for single-head graphs
    const float* wl; const float* wr;          // (ncls,1,32,32) deconvolution weights
    const float* cw; const float* cb;          // (ncls, 2*ncls), (ncls)
    float* logits;                   // NCHW ncls x H x W
    unsigned char* labels;           // H x W
    int ncls, Hs, Ws, H, W;
    int softmax;                     // apply softmax over classes to the written scores (deeplab test symbol)
    int uniform_w;                   // wl is the same 32x32 filter for every class: 4-pixel-per-thread kernel
    int N;                           // images (blockIdx.z), stacked in every buffer; 0 = 1
    int rHs, rWs;                    // size of the RIGHT score map when it differs from Hs x Ws (frame sizes that are multiples of 16
                                     // but not of 32: the stride-32 branch upsampled 2x is one row / column larger); 0 = same
};
hipError_t launch_score_tail(const ScoreTailParams& p, hipStream_t st);

hipError_t launch_nhwc_to_nchw(const float* src, int Cs, float* dst, int C, int H, int W, hipStream_t st);
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int Cs, int C, int H, int W, hipStream_t st, unsigned* yr = nullptr);
hipError_t launch_argmax_nchw(const float* logits, unsigned char* labels, int C, int HW, hipStream_t st);
hipError_t launch_copy_view(const float* src, int sCs, float* dst, int dCs, int C, int HW, hipStream_t st, unsigned* yr = nullptr);
// fp16x2 form: max |x| of a view (pixels x C channels, channel stride Cs) raised into a range slot -- in front of a convolution
// whose input tensor was not measured by its producers (misc.hip)
hipError_t launch_range_amax(const float* x, long pixels, int C, int Cs, unsigned* slot, hipStream_t st);
hipError_t launch_range_clear(unsigned* table, int n_slots, hipStream_t st);      // every slot back to zero (start of every run)
hipError_t launch_range_fold(unsigned* slot, unsigned* rflag, int op_index, hipStream_t st);      // word 0 of a slot = the maximum of its partial words; reports a non-finite one
hipError_t launch_copy_bytes(const void* src, void* dst, size_t bytes, hipStream_t st);      // flat device-to-device copy (one kernel, 16-byte accesses)
hipError_t launch_conv_narrow(const ConvParams& p, hipStream_t st);   // Cout_store == 4, plain conv, no dual output
hipError_t launch_score_fuse_lowres(const float* left, int lCs, const float* right, int rCs, const float* cw,
                                    float* z, int zCs, int ncls, int npix, hipStream_t st);
