/*
 * accel_hip.h -- C ABI of libaccel_hip.so, the MI355X (gfx950) execution engine
 * of the Accel (dff_deeplab) video-segmentation inference path.
 *
 * What it replaces in the reference (SamvitJ/Accel, /root/reference):
 *   the reference has NO C ABI on this path; its Python symbol files describe a
 *   graph and MXNet (un-vendored, README.md:76) executes it.  The boundary a
 *   maintainer binds is therefore the executor contract of
 *     dff_deeplab/core/tester.py:22-35        Predictor(bind, init_params, forward, get_outputs)
 *     dff_deeplab/core/module.py:791-845,1011-1044  bind at max shape / forward
 *     dff_deeplab/core/DataParallelExecutorGroup.py:18-27,330-378  copy-in, forward, collect
 *   and, per operator, the MXNet ops the symbol files call
 *     (dff_deeplab/symbols/resnet_v1_101_flownet_deeplab.py, accel_18.py:121-239).
 *   Each entry point below names the reference interface it stands in for.
 *
 * Conventions: every function returns 0 on success and a negative code on
 * failure (never throws, never prints); accel_last_error() gives the message of
 * the last failure on the calling thread.  Tensors crossing the boundary are
 * fp32 NCHW (label maps uint8 HxW), exactly the reference's layouts; `on_device`
 * flags whether a caller pointer is HBM or host memory.  The caller owns all
 * I/O buffers; the library owns weights, the activation arena and the named
 * persistent buffers.  Calls enqueue on the context's HIP stream; only
 * accel_sync() and copies to host memory block.  Objects are re-entrant across
 * instances, not thread-safe within one instance (same as one MXNet executor).
 */
#ifndef ACCEL_HIP_H
#define ACCEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct accel_ctx accel_ctx;       /* one GPU + one HIP stream            */
typedef struct accel_model accel_model;   /* parameters + persistent buffers     */
typedef struct accel_plan accel_plan;     /* one bound graph (key or cur)        */

#define ACCEL_OK 0
#define ACCEL_ERR_ARG (-1)
#define ACCEL_ERR_HIP (-2)
#define ACCEL_ERR_PLAN (-3)
#define ACCEL_ERR_PARAM (-4)
#define ACCEL_ERR_COMM (-5)
#define ACCEL_ERR_RANGE (-6)   /* fp16x2 form: a convolution input was not finite in an earlier run (accel_plan_run) */

const char* accel_last_error(void);
const char* accel_version(void);

/* context: stands in for `context=[mx.gpu(i)]` (demo.py:197,201) */
int accel_ctx_create(int device_id, accel_ctx** out);
int accel_ctx_destroy(accel_ctx* ctx);
int accel_sync(accel_ctx* ctx);                       /* = NDArray.asnumpy()'s implicit wait */
void* accel_ctx_stream(accel_ctx* ctx);               /* hipStream_t, for interop (RCCL / torch ExternalStream) */

/* model: parameter store shared by the key and cur graphs
 * (demo.py:192-195 arg_params/aux_params dicts; tester.py:30 init_params) */
int accel_model_create(accel_ctx* ctx, accel_model** out);
int accel_model_destroy(accel_model* m);
/* name = MXNet parameter name (`arg:`/`aux:` prefix already stripped), data = fp32
 * host memory in MXNet layout (OIHW conv, (Cin,Cout/g,kh,kw) deconv, (C,) BN);
 * the library repacks at plan finalisation. */
int accel_model_set_param(accel_model* m, const char* name, const float* data,
                          int ndim, const int64_t* shape);
int accel_model_has_param(accel_model* m, const char* name);

/* plan: a lowered, fused kernel sequence (text form, produced by
 * accel_amd/lower.py from the symbol graph) bound to static shapes -- the
 * counterpart of Module.bind (module.py:791-845).  `role` is "key", "cur" or any
 * label; plans of one model share the model's persistent buffers by name. */
int accel_model_add_plan(accel_model* m, const char* role, const char* plan_text, accel_plan** out);
int accel_plan_finalize(accel_plan* p);   /* repack weights, allocate arena, capture hipGraph */
/* enqueue one forward (Module.forward, module.py:1011).
 * Derived persistent buffers: a plan line `pbuf name=featG bytes=.. from=feat` declares featG a function of
 * `feat` that the plans keep in step with it (featG = fc6_weight * feat: non-key plans warp it instead of
 * re-running fc6 on the warped feature; Accel-101's plans have featC = corr_weight[:, :2048] * feat, the left half of
 * the feature fusion, the same way).  A plan that writes both leaves featG valid; any other write of `feat`
 * (accel_model_write, a raw pointer from accel_model_buffer, a plan that writes only `feat`) makes it stale, and
 * the next plan that READS featG first runs the plan registered under the role "init:featG" -- or fails with
 * ACCEL_ERR_PLAN if there is none.  Never silently reads a stale buffer. */
int accel_plan_run(accel_plan* p);
int accel_plan_num_ops(accel_plan* p);
/* kind: up to 31 chars + NUL; flops/bytes: algorithmic work of op i (0 if n/a) */
int accel_plan_op_info(accel_plan* p, int i, char* kind32, char* name64, double* flops, double* bytes);
/* launch decisions of op i after finalize (autotuned or heuristic): conv tile id (conv_igemm.hip, -1 for
 * non-conv ops), split-K factor, 1 if the narrow-N kernel runs it.  Diagnostic only. */
int accel_plan_op_launch(accel_plan* p, int i, int* tile, int* ksplit, int* narrow);
/* arithmetic mode of conv op i: 0 = fp32 layer (its launch geometry may still execute on the bf16 matrix cores as an exact
 * three-term split: tile 70-87), 1 = fp16-MFMA layer (plan option dtype=f16: ONE half product per multiply-add whatever the
 * geometry), 2 = bf16x3 layer (plan option dtype=bf16x3), 3 = fp32 layer whose matrix-core geometries run the fp16x2 form (two half
 * terms per operand, THREE half products per multiply-add; the default, plan option split=h2); -1 for non-conv ops.  Diagnostic only
 * (bench.py prices families by it). */
int accel_plan_op_mode(accel_plan* p, int i, int* mode);
/* fp16x2 form: the power of two conv op i's pixels were multiplied by before the split in the plan's LAST run (0 if the op has no such
 * form) and where the range behind it came from: 0 = no run yet / an all-zero input (scale 1), 1 = raised by the epilogues of the ops
 * that wrote the tensor, 2 = measured by a pass over the op's input view (tensors written outside the plan).  The scale is derived by
 * the convolution itself, in its prologue, from the largest |value| of its input tensor in THAT run (the largest lands in
 * [2^13, 2^14)): a plan run is a pure function of its inputs and parameters, like an executor forward of the reference
 * (dff_deeplab/core/module.py:1011-1044) -- nothing is calibrated, nothing survives a run.  A tensor such a convolution reads whose
 * largest stored |value| was an INFINITY in some run makes the NEXT accel_plan_run fail with ACCEL_ERR_RANGE (once); so does a NaN in a
 * tensor written by a byte mover (image converters, pools, warps, the deformable sampler, view copies) or measured by a pass of its
 * own.  A NaN that a matrix-core convolution produces mid-plan is NOT reported: its range epilogue takes floating-point maxima, which
 * drop NaNs (csrc/conv_epilogue.h); behind a ReLU the NaN itself becomes 0 (fmaxf -- the reference's relu `a > 0 ? a : 0` does the
 * same), without one it reaches the outputs through the matrix instructions as in fp32 arithmetic.
 * Diagnostic only. */
int accel_plan_op_range(accel_plan* p, int i, float* scale, int* source);
/* Diagnostic (scripts/debug/range_nan.py): the 1088 words of conv op i's input range slot after the plan's last run -- word 0 = the
 * largest |value| stored into the tensor, as a bit pattern (what the convolution read), words 64 .. 1087 = the partial words the
 * writers raised (csrc/range.h).  n_words must be at least 1088. */
int accel_plan_op_range_words(accel_plan* p, int i, unsigned* words, int n_words);
/* Launch geometries are REPRODUCIBLE: decisions come from the table shipped beside the library (tune/gfx950.tune, covers
 * the BASELINE workloads) or from the user's table ($ACCEL_TUNE_CACHE, else ~/.cache/accel_amd/gfx950.tune); a shape in
 * neither is timed once and appended to the user's table.  Counters of this process: decisions replayed, decisions
 * taken by timing, entries of the shipped table (0 = not found).  Diagnostic only. */
int accel_tune_stats(int* replayed, int* timed, int* shipped_entries);
/* runs the plan `iters` times eagerly with a HIP event pair around every op on
 * the context stream; ms[i] = mean duration of op i */
int accel_plan_profile(accel_plan* p, int iters, float* ms, int n_ms);

/* Diagnostics (scripts/debug, tests): the ops of a finalized plan launched one after the other on the context stream --
 * no captured graph -- followed by a host wait; and a host copy of a byte range of the plan's activation
 * arena (arena_bytes, if given, receives its size; host_dst may be NULL to query only).  Comparing the arena after a
 * normal accel_plan_run with the arena after accel_plan_run_serial of the same bound plan names the first op whose
 * output depends on the schedule. */
int accel_plan_run_serial(accel_plan* p);
int accel_plan_arena_read(accel_plan* p, size_t offset, void* host_dst, size_t bytes, size_t* arena_bytes);

/* persistent buffers (inputs `data`/`data_key`, outputs `logits`/`labels`, the
 * propagated feature `feat`): DataParallelExecutorGroup._load_general copy-in
 * (:18-27) and get_outputs (:357-378) */
int accel_model_write(accel_model* m, const char* buf, const void* src, size_t bytes, int src_on_device);
int accel_model_read(accel_model* m, const char* buf, void* dst, size_t bytes, int dst_on_device);
/* Zero-copy input: the image input `buf` (`data`, `data_key` -- a buffer that only the input-conversion kernels of the
 * finalized plans read) is read from the caller's device buffer `devptr` (`bytes` = the size of `buf`, same layout) by every
 * plan run that follows, until the next accel_model_write / accel_model_commit into `buf` or the next bind.  Stream-ordered
 * on the context stream like a write; the caller keeps `devptr` alive and unchanged until those runs have completed.
 * An EXTENSION, not the reference's behaviour: DataParallelExecutorGroup._load_general (executor_group.py:18-27) ALWAYS copies
 * the source array into the executor's bound input (d_src.copyto(d_targets)), also when the source already lives on the
 * device -- accel_model_write(src_on_device = 1) is that copy, and bench.py's headline uses it; the zero-copy binding is
 * reported separately (secondary.*_zero_copy_inputs).  accel_model_buffer(buf) and accel_model_read(buf) of a bound buffer:
 * a raw pointer hand-out ends the binding (the caller is about to write the model's own copy), a read returns the bytes the
 * plans would read (the bound frame). */
int accel_model_bind_device(accel_model* m, const char* buf, const void* devptr, size_t bytes);
int accel_model_buffer(accel_model* m, const char* buf, void** dev_ptr, size_t* bytes);
/* Write generation of a persistent buffer: starts at 0, bumped by every accel_plan_run of a plan that writes the
 * buffer, every accel_model_write / accel_model_commit into it and every raw-pointer hand-out.  A device handle to an
 * output (the `feat` a Predictor returns, tester.py:158-171) records the generation it was produced at; a consumer
 * that finds another generation knows the bytes are no longer that output (MXNet: "outputs are valid until the next
 * forward") and must fall back to a host copy or fail -- never read whatever the buffer holds now. */
int accel_model_buffer_generation(accel_model* m, const char* buf, uint64_t* generation);

/* ---- host I/O overlapped with compute ------------------------------------------------------------------------------
 * The reference's timed loop (demo.py:234-250) moves the frame in and the label map out synchronously through pageable
 * memory.  Here: page-locked staging memory, the NEXT frame's upload on a copy stream while the current frame computes,
 * and an asynchronous download of the outputs.
 *   accel_host_alloc / _free      page-locked host memory (hipHostMalloc)
 *   accel_model_prefetch(buf,src) enqueue H2D of `src` (page-locked, must stay valid until the matching commit) into a
 *                                 library-owned shadow of `buf` on the copy stream; returns immediately
 *   accel_model_commit(buf)       compute stream waits for the prefetch and copies the shadow into `buf` (HBM to HBM)
 *   accel_model_read_async        enqueue D2H of `buf` into page-locked `dst` on the compute stream, no host wait:
 *                                 the bytes are valid after accel_sync() */
int accel_host_alloc(size_t bytes, void** out);
int accel_host_free(void* p);
int accel_model_prefetch(accel_model* m, const char* buf, const void* pinned_src, size_t bytes);
int accel_model_commit(accel_model* m, const char* buf);
int accel_model_read_async(accel_model* m, const char* buf, void* pinned_dst, size_t bytes);

/* whole-frame entry points, the two Predictor.predict calls of the demo loop
 * (demo.py:235-245; tester.py:158-171 im_segment).  img_*: fp32 1x3xHxW already
 * mean-subtracted (lib/utils/image.py:224-235).  Any output pointer may be NULL.
 * feat_out / logits_out are NCHW fp32, labels_out uint8 HxW (first-max argmax).
 * The propagated feature stays in HBM between calls: in persistent buffer
 * `feat`, or -- when the non-key graph was bound as the plan pair `cur` /
 * `cur_b` (accel_amd/lower.py, feat_slot) -- alternately in `feat` and
 * `feat_b`; accel_cur_forward runs the plan that reads the buffer written
 * last, so callers never see the difference.  A host upload goes to `feat`. */
int accel_key_forward(accel_model* m, const float* img, int img_on_device,
                      float* feat_out, float* logits_out, uint8_t* labels_out, int out_on_device);
int accel_cur_forward(accel_model* m, const float* img_cur, const float* img_prev, int img_on_device,
                      float* feat_out, float* logits_out, uint8_t* labels_out, int out_on_device);

/* ---- operator level (host fp32 NCHW in / out; used by the parity tests) --------
 * Each is the single MXNet operator named, run through the same kernels and the
 * same weight repacking as the plans.  N >= 1 for conv / deconv / deformable conv / pool (the
 * convolution kernel runs the batch as one GEMM with M = N*Ho*Wo). */
/* mx.symbol.Convolution (+ optional per-channel scale/shift, residual, activation:
 * the fused epilogue).  act: 0 none, 1 relu, 2 leaky(slope).  force_tile: -1 = the library's choice, otherwise a launch
 * geometry id of conv_igemm.hip (every accepted id computes the same contraction; ids the build does not carry are
 * rejected with ACCEL_ERR_ARG -- timing-only ablation variants exist only in the diagnostics build). */
int accel_conv2d(accel_ctx* ctx, const float* x, int N, int C, int H, int W,
                 const float* w, const float* bias, int K, int kh, int kw,
                 int sh, int sw, int ph, int pw, int dh, int dw,
                 const float* scale, const float* shift, const float* residual,
                 int act, float slope, int force_tile, float* y);
/* mx.symbol.Deconvolution 4x4 stride 2 pad 1 (== pad 0 + Crop(offset 1,1)) */
int accel_deconv2d_4x4s2(accel_ctx* ctx, const float* x, int N, int C, int H, int W,
                         const float* w, const float* bias, int K, int act, float slope, float* y);
/* mx.contrib.symbol.DeformableConvolution, no bias */
int accel_deform_conv2d(accel_ctx* ctx, const float* x, int N, int C, int H, int W,
                        const float* offset, const float* w, int K, int kh, int kw,
                        int sh, int sw, int ph, int pw, int dh, int dw, int dg, float* y);
/* mx.symbol.Pooling; optional BN scale/shift + relu epilogue */
int accel_pool2d(accel_ctx* ctx, const float* x, int N, int C, int H, int W, int is_max, int full,
                 int kh, int kw, int sh, int sw, int ph, int pw,
                 const float* scale, const float* shift, int relu, float* y);
/* GridGenerator(transform_type='warp') + BilinearSampler == the "FlowWarp" op */
int accel_flow_warp(accel_ctx* ctx, const float* feat, int C, int H, int W, const float* flow, float* out);
/* Deconvolution 32x32/16 group=ncls + Crop(8,8) [+ Concat + correction 1x1] + argmax.
 * right/wr/cw/cb may be NULL (single head). Hs,Ws = score size; output H=16*Hs, W=16*Ws */
int accel_score_fuse(accel_ctx* ctx, const float* left, const float* right, int ncls, int Hs, int Ws,
                     const float* wl, const float* wr, const float* cw, const float* cb,
                     float* logits, uint8_t* labels);
/* mx.ndarray.argmax(axis=1) of an NCHW tensor (first maximal index) */
int accel_argmax_c(accel_ctx* ctx, const float* logits, int C, int H, int W, uint8_t* labels);
/* FlowNet input stage: avgpool2(concat(cur/255, prev/255)) -> 6 x H/2 x W/2 */
int accel_flow_input(accel_ctx* ctx, const float* cur, const float* prev, int H, int W, float* out);

/* ---- multi-GPU: the one collective of the path -------------------------------------------------------
 * Clips are sharded over ranks with no activation exchange; each frame's logits (or label map) are gathered to a root
 * rank.  Stands in for the host-side merge of per-GPU results in the reference's multi-GPU tester
 * (dff_rfcn/function/test_rcnn.py:62-82, dff_rfcn/core/tester.py:290-298: one thread per GPU, results appended on the
 * host).  One process per GPU; RCCL (librccl, resolved at run time) point-to-point send/recv over xGMI: every peer
 * uses its own direct link to the root, no ring.
 *   accel_comm_available   0 if THIS process can resolve librccl and every entry point the gather uses (no communicator, no
 *                          network activity); else an error whose text says what is missing.  Every rank calls it and the
 *                          ranks agree on the outcome BEFORE any of them enters accel_comm_create (ncclCommInitRank blocks
 *                          until all ranks have arrived: a rank that cannot load the library must not leave the others there)
 *   accel_comm_unique_id   128-byte id, made on one rank and distributed by the caller (file, socket, torch.distributed)
 *   accel_comm_create      communicator of `nranks` processes, this one being `rank`, bound to ctx's device; owns a
 *                          communication stream and two staging slots
 *   accel_gather_logits    every rank contributes `bytes` at `sendbuf` (HBM); the root receives rank r's block at
 *                          recvbuf + r*bytes (recvbuf is NULL on the other ranks).  Asynchronous: sendbuf is copied
 *                          into a staging slot in compute-stream order (so the next frame may overwrite it at once), the
 *                          transfer runs on the communication stream beside the next frame's kernels.  recvbuf must
 *                          stay untouched until accel_comm_sync() or the second-next gather.
 *   accel_comm_sync        host wait for all gathers issued so far */
typedef struct accel_comm accel_comm;
int accel_comm_available(void);
int accel_comm_unique_id(void* id128);
int accel_comm_create(accel_ctx* ctx, int rank, int nranks, const void* id128, accel_comm** out);
int accel_comm_destroy(accel_comm* comm);
int accel_gather_logits(accel_comm* comm, const void* sendbuf, void* recvbuf_or_null, size_t bytes, int root);
/* the same with a ROOT that contributes fewer bytes than a full slot (`bytes` = slot size = what every peer sends; the root, which
 * also receives all the others' frames, may be given fewer clips: send_bytes <= bytes on the root, == bytes elsewhere) */
int accel_gather_frames(accel_comm* comm, const void* sendbuf, size_t send_bytes, void* recvbuf_or_null, size_t bytes, int root);
/* The gather at SCORE resolution -- the default payload of bench.py --gpus N.  When the two upsampling filters are the same for every
 * class (the reference freezes them at the bilinear initialisation, accel_18.py:153) the plans fuse the two heads at score resolution
 * and leave the fused map in the model's persistent buffer `scores` ([images][H/16][W/16][ncls rounded up to 4] fp32, 0.66 MB per
 * 1024x2048 frame); the fp32 logits (159 MB) are a pure function of it.
 *   (A key plan's tail has ONE head and no correction bias, a non-key plan's the fused two: the expansion belongs to a PLAN.  Both leave
 *   their map in the same `scores` buffer; every rank passes the plan it has just run -- the ranks of a job run the same schedule.)
 *   accel_expand_scores   logits + labels of n_images maps at scores_dev (HBM, layout of `scores`) by the very launch plan p ends
 *                         with: bit-identical to the logits the producing GPU computed.  Enqueued on the model's compute
 *                         stream, or on a communicator's communication stream (on_comm_stream_of != NULL).
 *   accel_gather_scores   accel_gather_frames of the `scores` buffer (own_images of the slot_images a rank's slot holds; only the
 *                         root may contribute fewer) followed, on the root and on the communication stream, by the expansion of every
 *                         rank's block into logits_out / labels_out ([nranks * slot_images] images); valid after accel_comm_sync.
 * ACCEL_ERR_PLAN if the model has no `scores` buffer (filters not uniform): gather the logits then.  (Reference pattern: per-device
 * results merged on the host, dff_rfcn/function/test_rcnn.py:62-82.) */
int accel_expand_scores(accel_plan* p, const void* scores_dev, int n_images, float* logits_dev, unsigned char* labels_dev, accel_comm* on_comm_stream_of);
int accel_gather_scores(accel_comm* comm, accel_plan* p, int own_images, int slot_images, void* recv_scores_or_null, float* logits_out_or_null,
                        unsigned char* labels_out_or_null, int root);
int accel_comm_sync(accel_comm* comm);

#ifdef __cplusplus
}
#endif
#endif
