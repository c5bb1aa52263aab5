// libaccel_hip: plan executor + C ABI (see include/accel_hip.h).
//
// A plan is the text form of a lowered graph: one fused kernel op per line,
// `kind key=value ...`.  Buffer references are `space:byteoff:C:Cs:H:W` where
// space is `A` (the plan's activation arena; offsets assigned by the Python
// liveness planner) or the name of a model-level persistent buffer.
// Parameters are referenced by their MXNet names and repacked here.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/accel_hip.h"
#include "kernels.h"
#include "range.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return fail(ACCEL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                             \
    } while (0)

extern "C" const char* accel_last_error(void) { return g_err.c_str(); }
extern "C" const char* accel_version(void) { return "accel_hip 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------
struct accel_ctx {
    int device;
    hipStream_t stream;     // compute stream (the one callers order against)
    hipStream_t copy;       // host<->HBM prefetch stream (accel_model_prefetch)
};

struct HostParam {
    std::vector<int64_t> shape;
    std::vector<float> data;
    size_t numel() const { return data.size(); }
};

struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    // accel_model_bind_device: input buffers that only the prep kernels read are read THROUGH `slot` (a pointer in device
    // memory, initially = ptr), which the host may point at a caller-owned frame between two runs of a captured plan
    const void** slot = nullptr;
    const void* bound = nullptr;        // what the slot points at when it is not `ptr`
    int readers = 0, prep_readers = 0;  // plan operands resolved to this buffer / those of prep_rgb and prep_flow
};

struct accel_model {
    accel_ctx* ctx;
    std::map<std::string, HostParam> params;
    std::map<std::string, DevBuf> pbufs;
    std::vector<accel_plan*> plans;
    std::map<std::string, accel_plan*> roles;
    int feat_c = 0, feat_h = 0, feat_w = 0, feat_n = 1;   // shape of the propagated feature (`meta` line of the plans)
    // derived persistent buffers (`pbuf name=featG from=feat`): a linear image of another buffer that the plans keep
    // in step with it.  Writing the source from outside a plan that also writes the derived buffer makes it stale; a
    // plan that reads a stale derived buffer first runs the model's `init:<name>` plan (see accel_plan_run).
    std::map<std::string, std::string> derived_from;
    std::map<std::string, bool> derived_valid;
    std::map<std::string, uint64_t> generation;     // write generation per persistent buffer (accel_model_buffer_generation)
    struct Shadow { void* ptr = nullptr; size_t bytes = 0, filled = 0; hipEvent_t ready = nullptr, consumed = nullptr; bool was_consumed = false; };
    std::map<std::string, Shadow> shadows;          // prefetch targets (accel_model_prefetch / accel_model_commit)
    // which buffer holds the current propagated feature: 0 = `feat`, 1 = `feat_b` (non-key graphs may be bound as a pair of
    // plans `cur` / `cur_b` that ping-pong between the two instead of copying the warped feature back; whoever wrote last)
    int feat_slot = 0;
    void source_written(const std::string& src) {
        ++generation[src];
        if (src == "feat") feat_slot = 0;
        else if (src == "feat_b") feat_slot = 1;
        for (auto& kv : derived_from) if (kv.second == src) derived_valid[kv.first] = false;
    }
};

struct BufRef {
    std::string space;
    size_t off = 0;
    int C = 0, Cs = 0, H = 0, W = 0;
    int N = 1;              // images in the batch; image n starts n*H*W pixels (n*H*W*Cs floats) after image 0
    int esize = 4;          // bytes per element: 4 = fp32, 2 = half (":h" tag; f16-mode plans, convolution views only)
    bool set = false;
    float* ptr = nullptr;   // resolved at finalize
    size_t img() const { return (size_t)H * W * Cs; }   // floats per image
};

typedef std::map<std::string, std::string> KV;

enum OpKind { OP_PREP_RGB, OP_PREP_FLOW, OP_CONV, OP_POOL, OP_WARP, OP_DCN_COLS, OP_SCORE_TAIL,
              OP_COPY, OP_EXPORT_NCHW, OP_IMPORT_NCHW };

struct Op {
    OpKind kind;
    std::string kind_name, name;
    KV kv;
    double flops = 0, bytes = 0;
    // resolved launch state
    ConvParams conv;
    PoolParams pool;
    DcnColsParams dcn;
    ScoreTailParams tail;
    ScoreTailParams tail_lowres;   // set when the fusion can run at score resolution (uniform upsampling filter)
    float* tail_z = nullptr;
    bool tail_copy = false;      // single-head tail: tail_z is a copy of the one score map (so that `scores` always holds what the logits are a function of)
    BufRef a, b, c, d;          // generic buffer slots
    const float* p0 = nullptr;  // generic device param slots
    const float* p1 = nullptr;
    const float* const* slot_a = nullptr;   // prep ops: pointer slots of the image inputs (accel_model_bind_device)
    const float* const* slot_b = nullptr;
    size_t nbytes = 0;
    int H = 0, W = 0;
    // range slots (kernels.h / range.h): ids the lowering gave to the physical buffers this op reads as convolution input (`xr=`) and
    // writes (`yr=`, `y2r=`); -1: none given.  `measure`: an fp16x2-form convolution whose input tensor is not fully covered by range
    // epilogues of earlier ops of this plan -- its view is measured by a launch of its own right before it (assign_range_slots)
    int rs_in = -1, rs_out = -1, rs_out2 = -1;
    bool measure = false;
    bool fold = false;          // this op is the first reader of its input slot after a write: the slot's partial words are folded into word 0 first
    unsigned* yr = nullptr;     // resolved slots of the outputs (null: nobody needs them); convolutions, pools and the deformable sampler
    unsigned* y2r = nullptr;    // carry them in their parameter blocks as well
};

struct accel_plan {
    accel_model* m;
    std::string role;
    std::vector<Op> ops;
    size_t arena_bytes = 0;
    char* arena = nullptr;
    std::vector<void*> owned;       // packed weights etc.
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    bool finalized = false;
    bool allow_graph = true;
    bool allow_tune = true;
    int f16 = 0;                    // option dtype=f16: convolutions on the fp16 matrix cores; dtype=bf16x3: 2 (kernels.h)
    // fp32 layers on the matrix cores: split = 1 (default) the fp16x2 form (two half terms per operand, three products), 0 the
    // bf16x3 form (three bf16 terms, six products) -- plan option split=b3|h2, ACCEL_SPLIT=b3|h2.  The fp16x2 form centres every
    // convolution's pixels in the half range by a power of two that the kernel derives from the RANGE SLOT of its input tensor:
    // `range` holds n_slots slots of RANGE_WORDS words, zeroed at the start of every run (range_clear_kernel, the first node of the
    // captured graph: a memset node misbehaved, misc.hip) and raised by the epilogue of whoever writes the tensor -- a run's scales are
    // a function of that run's data, nothing survives it.
    // range_flag: host-mapped word an fp16x2-form convolution raises when its input's range is not finite (ACCEL_ERR_RANGE).
    int split = 1;
    unsigned* range = nullptr;
    int n_slots = 0;
    // Score-resolution tail (uniform upsampling filters): the plan leaves the map its logits are a pure function of -- the fused scores of
    // the two heads (score_fuse_lowres_kernel) or the one head's scores -- in the model's persistent buffer `scores`
    // ([images][H/16][W/16][ncls rounded up to 4] fp32) and `score_tpl` is the launch that expands such a map into logits + labels: the
    // plan's own last step, and what accel_expand_scores / accel_gather_scores run on a map that came from another GPU.
    ScoreTailParams score_tpl;
    bool has_score_tpl = false;
    int score_images = 0, score_zcs = 0;
    std::map<int, int> slot_of;      // lowering buffer id -> slot index
    std::vector<int> in_slot;        // per op: the slot a convolution with an fp16x2 form reads (-1: none)
    unsigned* range_flag = nullptr;       // host-mapped
    unsigned* check_scratch = nullptr;    // ACCEL_CHECK_FINITE: one more slot behind the table
    unsigned* range_flag_dev = nullptr;
    int n_h2 = 0;
    size_t ws_bytes = 0;            // split-K workspace shared by the convs of the plan (stream-ordered)
    float* ws = nullptr;
    std::vector<std::string> pbuf_reads, pbuf_writes;   // persistent buffers the ops read / write (derived-buffer tracking)
};

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
static int roundup(int a, int b) { return (a + b - 1) / b * b; }

// ACCEL_WITHHOLD="winograd,ws1x1,...": kernel families a plan does NOT offer to its tuner (A/B runs and diagnostics; a geometry a
// conv forces with tile= is packed regardless).  Families: split (every bf16x3 / fp16x2 geometry: the fp32 MFMA kernels remain),
// b3r (conv_b3r.hip), winograd, winograd_split (41-43), stem, stem_split (51), ws1x1, halo (78), deep (the deep-prefetch tiles 31-35).
static bool withheld(const char* family)
{
    const char* e = getenv("ACCEL_WITHHOLD");
    if (!e) return false;
    const size_t n = strlen(family);
    for (const char* q = e; *q;) {
        const char* c = strchr(q, ',');
        const size_t len = c ? (size_t)(c - q) : strlen(q);
        if (len == n && !strncmp(q, family, n)) return true;
        q += len + (c ? 1 : 0);
    }
    return false;
}

static bool kv_has(const KV& kv, const char* k) { return kv.find(k) != kv.end(); }
static std::string kv_str(const KV& kv, const char* k, const char* def = "")
{
    auto it = kv.find(k);
    return it == kv.end() ? std::string(def) : it->second;
}
static long kv_int(const KV& kv, const char* k, long def = 0)
{
    auto it = kv.find(k);
    return it == kv.end() ? def : strtol(it->second.c_str(), nullptr, 10);
}
static double kv_f(const KV& kv, const char* k, double def = 0)
{
    auto it = kv.find(k);
    return it == kv.end() ? def : strtod(it->second.c_str(), nullptr);
}
static void kv_pair(const KV& kv, const char* k, int& a, int& b, int da, int db)
{
    auto it = kv.find(k);
    a = da; b = db;
    if (it == kv.end()) return;
    sscanf(it->second.c_str(), "%d,%d", &a, &b);
}

static int parse_buf(const KV& kv, const char* key, BufRef& r)
{
    auto it = kv.find(key);
    if (it == kv.end()) { r.set = false; return 0; }
    std::vector<std::string> f;
    std::stringstream ss(it->second);
    std::string tok;
    while (std::getline(ss, tok, ':')) f.push_back(tok);
    r.esize = 4;
    if (f.size() == 8 && f[7] == "h") { r.esize = 2; f.pop_back(); }      // half storage: space:off:C:Cs:H:W:N:h
    if (f.size() != 6 && f.size() != 7) return fail(ACCEL_ERR_PLAN, "bad buffer reference %s=%s", key, it->second.c_str());
    r.space = f[0];
    r.off = strtoull(f[1].c_str(), nullptr, 10);
    r.C = atoi(f[2].c_str()); r.Cs = atoi(f[3].c_str()); r.H = atoi(f[4].c_str()); r.W = atoi(f[5].c_str());
    r.N = f.size() == 7 ? atoi(f[6].c_str()) : 1;
    if (r.N < 1) return fail(ACCEL_ERR_PLAN, "bad batch size in %s=%s", key, it->second.c_str());
    r.set = true;
    return 0;
}

static int dev_upload(accel_plan* p, const void* host, size_t bytes, void** out)
{
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes ? bytes : 4));
    if (bytes) HIP_TRY(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    p->owned.push_back(d);
    *out = d;
    return 0;
}

static const HostParam* get_param(accel_model* m, const std::string& name)
{
    auto it = m->params.find(name);
    return it == m->params.end() ? nullptr : &it->second;
}

// BatchNorm inference folding -- identical expressions to oracle orc_bn_fold
// (MXNet mshadow inference form): scale = g/sqrt(var+eps), shift = beta - g*mean/sqrt(var+eps)
static int bn_fold(accel_model* m, const std::string& bn, float eps, int fixg, int C,
                   std::vector<float>& scale, std::vector<float>& shift)
{
    const HostParam* g = get_param(m, bn + "_gamma");
    const HostParam* b = get_param(m, bn + "_beta");
    const HostParam* mu = get_param(m, bn + "_moving_mean");
    const HostParam* var = get_param(m, bn + "_moving_var");
    if (!b || !mu || !var || (!g && !fixg))
        return fail(ACCEL_ERR_PARAM, "BatchNorm %s: gamma/beta/moving_mean/moving_var not initialized", bn.c_str());
    if ((int)b->numel() != C || (int)mu->numel() != C || (int)var->numel() != C)
        return fail(ACCEL_ERR_PARAM, "BatchNorm %s: expected %d channels, got %zu", bn.c_str(), C, b->numel());
    scale.assign(C, 0.f); shift.assign(C, 0.f);
    for (int c = 0; c < C; ++c) {
        const float gg = fixg ? 1.0f : g->data[c];
        const float sd = sqrtf(var->data[c] + eps);
        scale[c] = gg / sd;
        shift[c] = b->data[c] - (gg * mu->data[c]) / sd;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// plan parsing
// ---------------------------------------------------------------------------
static int parse_plan(accel_plan* p, const char* text)
{
    std::stringstream ss(text);
    std::string line;
    int lineno = 0;
    while (std::getline(ss, line)) {
        ++lineno;
        if (line.empty() || line[0] == '#') continue;
        std::stringstream ls(line);
        std::string kind, tok;
        ls >> kind;
        KV kv;
        while (ls >> tok) {
            size_t eq = tok.find('=');
            if (eq == std::string::npos) return fail(ACCEL_ERR_PLAN, "line %d: token '%s' is not key=value", lineno, tok.c_str());
            kv[tok.substr(0, eq)] = tok.substr(eq + 1);
        }
        if (kind == "option") {
            if (kv_has(kv, "graph")) p->allow_graph = kv_int(kv, "graph", 1) != 0;
            if (kv_has(kv, "tune")) p->allow_tune = kv_int(kv, "tune", 1) != 0;
            if (kv_has(kv, "dtype")) p->f16 = kv_str(kv, "dtype") == "f16" ? 1 : kv_str(kv, "dtype") == "bf16x3" ? 2 : 0;
            if (kv_has(kv, "split")) p->split = kv_str(kv, "split") == "b3" ? 0 : 1;
            continue;
        }
        if (kind == "meta") {
            if (kv_has(kv, "feat_c")) { p->m->feat_c = (int)kv_int(kv, "feat_c"); p->m->feat_h = (int)kv_int(kv, "feat_h"); p->m->feat_w = (int)kv_int(kv, "feat_w"); p->m->feat_n = (int)kv_int(kv, "feat_n", 1); }
            continue;
        }
        if (kind == "arena") { p->arena_bytes = strtoull(kv_str(kv, "bytes", "0").c_str(), nullptr, 10); continue; }
        if (kind == "pbuf") {
            const std::string name = kv_str(kv, "name");
            const size_t bytes = strtoull(kv_str(kv, "bytes", "0").c_str(), nullptr, 10);
            if (name.empty() || !bytes) return fail(ACCEL_ERR_PLAN, "line %d: pbuf needs name and bytes", lineno);
            DevBuf& b = p->m->pbufs[name];
            if (!b.ptr) {
                if (hipMalloc(&b.ptr, bytes) != hipSuccess) return fail(ACCEL_ERR_HIP, "hipMalloc(%zu) for pbuf %s failed", bytes, name.c_str());
                b.bytes = bytes;
                if (hipMemsetAsync(b.ptr, 0, bytes, p->m->ctx->stream) != hipSuccess) return fail(ACCEL_ERR_HIP, "hipMemset pbuf failed");   // same stream as later writes: ordered
            } else if (b.bytes < bytes) {
                return fail(ACCEL_ERR_PLAN, "pbuf %s already exists with %zu bytes < %zu", name.c_str(), b.bytes, bytes);
            }
            if (kv_has(kv, "from")) {
                p->m->derived_from[name] = kv_str(kv, "from");
                if (!p->m->derived_valid.count(name)) p->m->derived_valid[name] = false;
            }
            continue;
        }
        Op op;
        op.kind_name = kind;
        op.kv = kv;
        op.name = kv_str(kv, "name");
        if (kv_int(kv, "stream", 0) != 0 || kv_has(kv, "wait"))
            return fail(ACCEL_ERR_PLAN, "op %s: plans run on one stream (the two-stream lowering was removed, DESIGN.md 7)", op.name.c_str());
        op.flops = kv_f(kv, "flops");
        op.bytes = kv_f(kv, "bytes");
        op.rs_in = (int)kv_int(kv, "xr", -1); op.rs_out = (int)kv_int(kv, "yr", -1); op.rs_out2 = (int)kv_int(kv, "y2r", -1);
        if (kind == "prep_rgb") op.kind = OP_PREP_RGB;
        else if (kind == "prep_flow") op.kind = OP_PREP_FLOW;
        else if (kind == "conv") op.kind = OP_CONV;
        else if (kind == "pool") op.kind = OP_POOL;
        else if (kind == "warp") op.kind = OP_WARP;
        else if (kind == "dcn_cols") op.kind = OP_DCN_COLS;
        else if (kind == "score_tail") op.kind = OP_SCORE_TAIL;
        else if (kind == "copy") op.kind = OP_COPY;
        else if (kind == "export_nchw") op.kind = OP_EXPORT_NCHW;
        else if (kind == "import_nchw") op.kind = OP_IMPORT_NCHW;
        else return fail(ACCEL_ERR_PLAN, "line %d: unknown op '%s'", lineno, kind.c_str());
        for (const auto& e : kv) {
            const size_t c0 = e.second.find(':');
            if (c0 == std::string::npos || c0 == 0) continue;
            const std::string space = e.second.substr(0, c0);
            if (space == "A" || !p->m->pbufs.count(space)) continue;
            const bool is_out = e.first == "out" || e.first == "out2" || e.first == "dst" || e.first == "logits" || e.first == "labels";
            auto& v = is_out ? p->pbuf_writes : p->pbuf_reads;
            if (std::find(v.begin(), v.end(), space) == v.end()) v.push_back(space);
        }
        p->ops.push_back(op);
    }
    return 0;
}

static int resolve(accel_plan* p, BufRef& r, const char* what, bool half_ok = false)
{
    if (!r.set) return fail(ACCEL_ERR_PLAN, "missing buffer '%s'", what);
    if (r.esize != 4 && !half_ok) return fail(ACCEL_ERR_PLAN, "buffer '%s': only convolutions of an f16-mode plan take half views", what);
    if (r.esize == 2 && (r.space != "A" || r.Cs % 8 || r.off % 16)) return fail(ACCEL_ERR_PLAN, "buffer '%s': a half view lives in the arena, 16-byte aligned, Cs %% 8 == 0", what);
    if (r.space == "A") {
        // a view's last pixel row may stop short of Cs; be lenient by C
        const size_t need_min = r.off + (((size_t)r.N * r.H * r.W - 1) * r.Cs + r.C) * (size_t)r.esize;
        if (need_min > p->arena_bytes) return fail(ACCEL_ERR_PLAN, "buffer '%s' exceeds arena (%zu > %zu)", what, need_min, p->arena_bytes);
        r.ptr = reinterpret_cast<float*>(p->arena + r.off);
    } else {
        auto it = p->m->pbufs.find(r.space);
        if (it == p->m->pbufs.end()) return fail(ACCEL_ERR_PLAN, "buffer '%s': unknown persistent buffer '%s'", what, r.space.c_str());
        r.ptr = reinterpret_cast<float*>(static_cast<char*>(it->second.ptr) + r.off);
        ++it->second.readers;
    }
    if (r.space == "A" && (r.Cs % 4 || (r.off % 16))) return fail(ACCEL_ERR_PLAN, "buffer '%s': Cs %d / offset %zu not 16-byte aligned", what, r.Cs, r.off);
    return 0;
}

// ---- weight repacking -------------------------------------------------------
// fp32 packed weights [classes][rows][K_pad] (+ slack) -> the three bf16 planes of the bf16x3 kernel.  Each weight is split
// exactly (top 16 bits, then of the residual, twice).  Inside a plane the order is [class][K step of 32][row][32]: the tile a
// block fetches per K step (BN rows x 32 k) is then ONE contiguous run of BN x 64 bytes -- whole cache lines, where the
// row-major order gave 64-byte pieces 2 * K_pad bytes apart.
static void pack_bf16x3(const std::vector<float>& packed, int classes, int rows, int K_pad, std::vector<uint16_t>& out, size_t& plane)
{
    const int steps = K_pad / 32;
    plane = (size_t)classes * rows * K_pad + 2 * (size_t)rows * 32;      // two K steps of slack: the kernel prefetches past the end
    out.assign(3 * plane, 0);
    for (int c = 0; c < classes; ++c)
        for (int n = 0; n < rows; ++n)
            for (int k = 0; k < K_pad; ++k) {
                float r = packed[((size_t)c * rows + n) * K_pad + k];
                const size_t dst = (((size_t)c * steps + k / 32) * rows + n) * 32 + (k & 31);
                for (int pl = 0; pl < 3; ++pl) {
                    uint32_t u; memcpy(&u, &r, 4);
                    u &= 0xFFFF0000u;
                    out[pl * plane + dst] = (uint16_t)(u >> 16);
                    float t; memcpy(&t, &u, 4);
                    r -= t;
                }
            }
}

// The same three planes in MFMA fragment order for conv_b3r.hip: [class][K step][half step][row][16] -- the 16 bytes lane
// (row, half) feeds to one v_mfma_f32_32x32x16_bf16 are contiguous (k = 32*step + 16*halfstep + 8*half .. +8), a
// wavefront's fragment fetch is one contiguous kilobyte.  Two K steps of slack: the kernel prefetches past the end.
static void pack_bf16x3r(const std::vector<float>& packed, int classes, int rows, int K_pad, std::vector<uint16_t>& out, size_t plane)
{
    const int steps = K_pad / 32;
    out.assign(3 * plane, 0);
    for (int c = 0; c < classes; ++c)
        for (int n = 0; n < rows; ++n)
            for (int k = 0; k < K_pad; ++k) {
                float r = packed[((size_t)c * rows + n) * K_pad + k];
                const int ks = k / 32, kb = (k >> 4) & 1, kk = k & 15;
                const size_t dst = ((((size_t)c * steps + ks) * 2 + kb) * rows + n) * 16 + kk;
                for (int pl = 0; pl < 3; ++pl) {
                    uint32_t u; memcpy(&u, &r, 4);
                    u &= 0xFFFF0000u;
                    out[pl * plane + dst] = (uint16_t)(u >> 16);
                    float t; memcpy(&t, &u, 4);
                    r -= t;
                }
            }
}

// fp16x2 form of the same fragment-ordered planes (ConvParams::wh2r): every weight as hi + lo, two half terms of w * 2^q[row], the
// exponent q[row] chosen so that the largest weight of the output channel (over all parity classes) lands in [2^14, 2^15) -- the
// top of the half range, where lo keeps its full 11 bits for every weight down to 2^-16 of the channel's largest (below that the
// absolute error is 2^-25, i.e. 2^-39 of the largest).  qexp receives q per row; the epilogue scale is multiplied by 2^-q (exact).
static void pack_h2r(const std::vector<float>& packed, int classes, int rows, int K_pad, std::vector<uint16_t>& out, size_t plane, std::vector<int>& qexp)
{
    const int steps = K_pad / 32;
    out.assign(2 * plane, 0);
    qexp.assign(rows, 0);
    for (int n = 0; n < rows; ++n) {
        float amax = 0.f;
        for (int c = 0; c < classes; ++c)
            for (int k = 0; k < K_pad; ++k) amax = std::max(amax, std::fabs(packed[((size_t)c * rows + n) * K_pad + k]));
        if (amax > 0.f && std::isfinite(amax)) { int e; std::frexp(amax, &e); qexp[n] = 15 - e; }
    }
    for (int c = 0; c < classes; ++c)
        for (int n = 0; n < rows; ++n)
            for (int k = 0; k < K_pad; ++k) {
                const float v = std::ldexp(packed[((size_t)c * rows + n) * K_pad + k], qexp[n]);
                const int ks = k / 32, kb = (k >> 4) & 1, kk = k & 15;
                const size_t dst = ((((size_t)c * steps + ks) * 2 + kb) * rows + n) * 16 + kk;
                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                memcpy(&out[dst], &hi, 2);
                memcpy(&out[plane + dst], &lo, 2);
            }
}

// conv weights (Cout, Cin, kh, kw) -> [rows][K_pad], k = (ky*kw + kx)*cin_pad + ci
static void pack_conv_w(const HostParam& w, int cin_pad, int rows, int K_pad, std::vector<float>& out)
{
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1], kh = (int)w.shape[2], kw = (int)w.shape[3];
    out.assign((size_t)rows * K_pad, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx)
                    out[(size_t)co * K_pad + (size_t)(ky * kw + kx) * cin_pad + ci] =
                        w.data[(((size_t)co * Cin + ci) * kh + ky) * kw + kx];
}

// deconv 4x4 s2 p1 weights (Cin, Cout, 4, 4) -> [4 classes][rows][K_pad]; class (py,px), tap (ty,tx):
// ky = 3 - py - 2*ty, kx = 3 - px - 2*tx ; k = (ty*2 + tx)*cin_pad + ci
static void pack_deconv2x_w(const HostParam& w, int cin_pad, int rows, int K_pad, std::vector<float>& out)
{
    const int Cin = (int)w.shape[0], Cout = (int)w.shape[1];
    out.assign((size_t)4 * rows * K_pad, 0.f);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            float* base = out.data() + (size_t)(py * 2 + px) * rows * K_pad;
            for (int co = 0; co < Cout; ++co)
                for (int ty = 0; ty < 2; ++ty)
                    for (int tx = 0; tx < 2; ++tx) {
                        const int ky = 3 - py - 2 * ty, kx = 3 - px - 2 * tx;
                        for (int ci = 0; ci < Cin; ++ci)
                            base[(size_t)co * K_pad + (size_t)(ty * 2 + tx) * cin_pad + ci] =
                                w.data[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx];
                    }
        }
}

static int finalize_conv(accel_plan* p, Op& op)
{
    accel_model* m = p->m;
    const KV& kv = op.kv;
    int rc;
    if ((rc = parse_buf(kv, "in", op.a)) || (rc = parse_buf(kv, "out", op.b)) ||
        (rc = parse_buf(kv, "out2", op.c)) || (rc = parse_buf(kv, "res", op.d))) return rc;
    if ((rc = resolve(p, op.a, "in", true)) || (rc = resolve(p, op.b, "out", true))) return rc;
    if (op.c.set && (rc = resolve(p, op.c, "out2"))) return rc;
    if (op.d.set && (rc = resolve(p, op.d, "res", true))) return rc;

    const std::string wname = kv_str(kv, "w");
    const HostParam* w = get_param(m, wname);
    if (!w) return fail(ACCEL_ERR_PARAM, "%s not initialized", wname.c_str());
    if (w->shape.size() != 4) return fail(ACCEL_ERR_PARAM, "%s: expected a 4-d weight", wname.c_str());
    const std::string mode = kv_str(kv, "mode", "conv");
    const bool deconv = mode == "deconv2x";
    const bool cols = mode == "cols";
    int kh, kw, sh, sw, ph, pw, dh, dw;
    kv_pair(kv, "k", kh, kw, 1, 1);
    kv_pair(kv, "s", sh, sw, 1, 1);
    kv_pair(kv, "p", ph, pw, 0, 0);
    kv_pair(kv, "d", dh, dw, 1, 1);
    const int cin = (int)kv_int(kv, "cin"), cout = (int)kv_int(kv, "cout");
    const int cin_pad = roundup(cin, 4);
    const int cout_store = roundup(cout, 4);
    const int rows = roundup(cout_store, 128);
    if (cout_store > op.b.Cs) return fail(ACCEL_ERR_PLAN, "conv %s: output view too narrow (%d > Cs %d)", op.name.c_str(), cout_store, op.b.Cs);

    ConvParams& c = op.conv;
    memset(&c, 0, sizeof c);
    std::vector<float> packed;
    if (deconv) {
        if (w->shape[0] != cin || w->shape[1] != cout || w->shape[2] != 4 || w->shape[3] != 4)
            return fail(ACCEL_ERR_PARAM, "shape inconsistent for %s: need (%d,%d,4,4)", wname.c_str(), cin, cout);
        c.kh = c.kw = 2; c.sh = c.sw = 1; c.dh = c.dw = 1; c.ph = c.pw = 0;
        c.Cin = cin_pad;
        c.K_pad = roundup(4 * cin_pad, 32);
        pack_deconv2x_w(*w, cin_pad, rows, c.K_pad, packed);
        c.deconv2x = 1;
        c.w_class_stride = (size_t)rows * c.K_pad;
        c.Ho = op.a.H; c.Wo = op.a.W;
        c.yH = op.b.H; c.yW = op.b.W;
        // 4x4/2 pad 1 gives exactly 2h x 2w; the reference's "pad 0 + Crop(offset 1) to the skip tensor" may keep one row /
        // column less when the skip tensor has an odd size (frame sizes that are not multiples of 128): 2h-1 / 2w-1
        if (op.b.H > 2 * op.a.H || op.b.H < 2 * op.a.H - 1 || op.b.W > 2 * op.a.W || op.b.W < 2 * op.a.W - 1)
            return fail(ACCEL_ERR_PLAN, "deconv2x %s: output must be 2x the input (or one row / column less)", op.name.c_str());
    } else {
        int wkh = kh, wkw = kw;
        if (cols) kv_pair(kv, "wk", wkh, wkw, 3, 3);
        if (w->shape[0] != cout || w->shape[1] != cin || w->shape[2] != wkh || w->shape[3] != wkw)
            return fail(ACCEL_ERR_PARAM, "shape inconsistent for %s: need (%d,%d,%d,%d) got (%ld,%ld,%ld,%ld)",
                        wname.c_str(), cout, cin, wkh, wkw, (long)w->shape[0], (long)w->shape[1], (long)w->shape[2], (long)w->shape[3]);
        const int Kreal = wkh * wkw * cin_pad;
        c.K_pad = roundup(Kreal, 32);
        pack_conv_w(*w, cin_pad, rows, c.K_pad, packed);
        if (cols) { c.kh = c.kw = 1; c.Cin = Kreal; }
        else { c.kh = kh; c.kw = kw; c.Cin = cin_pad; }
        c.sh = sh; c.sw = sw; c.ph = ph; c.pw = pw; c.dh = dh; c.dw = dw;
        c.Ho = op.b.H; c.Wo = op.b.W;
        c.yH = op.b.H; c.yW = op.b.W;
        const int eh = (op.a.H + 2 * c.ph - c.dh * (c.kh - 1) - 1) / c.sh + 1;
        const int ew = (op.a.W + 2 * c.pw - c.dw * (c.kw - 1) - 1) / c.sw + 1;
        if (eh != op.b.H || ew != op.b.W)
            return fail(ACCEL_ERR_PLAN, "conv %s: output %dx%d does not match geometry %dx%d", op.name.c_str(), op.b.H, op.b.W, eh, ew);
        if (c.Cin > op.a.Cs) return fail(ACCEL_ERR_PLAN, "conv %s: input view narrower than Cin", op.name.c_str());
    }
    packed.resize(packed.size() + 256, 0.f);   // slack: the pipelined kernel prefetches up to two K steps past the end
    // fp16-MFMA mode (plan option dtype=f16): only where a K chunk of 8 never straddles two taps
    const char* env_dt = getenv("ACCEL_CONV_DTYPE");   // "f16": also for the single-operator entry points
    const int dflt = (p->f16 == 2 || (env_dt && !strcmp(env_dt, "bf16x3"))) ? 2 : (p->f16 || (env_dt && !strcmp(env_dt, "f16"))) ? 1 : 0;
    const int want_f16 = (int)kv_int(kv, "f16", dflt);
    c.f16 = (want_f16 && c.Cin % (want_f16 == 2 ? 4 : 8) == 0 && cout_store > 4) ? want_f16 : 0;
    void* dw_ = nullptr;
    if (c.f16 == 2) {
        std::vector<uint16_t> pb;
        pack_bf16x3(packed, c.deconv2x ? 4 : 1, rows, c.K_pad, pb, c.w_plane);
        if ((rc = dev_upload(p, pb.data(), pb.size() * sizeof(uint16_t), &dw_))) return rc;
    } else if (c.f16) {
        std::vector<_Float16> ph(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) ph[i] = (_Float16)packed[i];
        if ((rc = dev_upload(p, ph.data(), ph.size() * sizeof(_Float16), &dw_))) return rc;
        if (!withheld("b3r")) {
            // the same half-rounded weights once more in MFMA fragment order [class][K step][half step][row][16] for the fp16 form
            // of conv_b3r.hip (launch geometries 76 / 77 / 79 / 80 / 81 of an f16 layer)
            const int classes = c.deconv2x ? 4 : 1, steps = c.K_pad / 32;
            c.w_plane = (size_t)classes * rows * c.K_pad + 2 * (size_t)rows * 32;
            std::vector<_Float16> pr(c.w_plane, (_Float16)0.f);
            for (int cl = 0; cl < classes; ++cl)
                for (int n = 0; n < rows; ++n)
                    for (int k = 0; k < c.K_pad; ++k)
                        pr[((((size_t)cl * steps + k / 32) * 2 + ((k >> 4) & 1)) * rows + n) * 16 + (k & 15)] = ph[((size_t)cl * rows + n) * c.K_pad + k];
            void* d4 = nullptr;
            if ((rc = dev_upload(p, pr.data(), pr.size() * sizeof(_Float16), &d4))) return rc;
            c.wb3r = d4;
        }
    } else if ((rc = dev_upload(p, packed.data(), packed.size() * sizeof(float), &dw_))) return rc;
    c.w = static_cast<const float*>(dw_);
    {
        // fp32 layers also get their weights as three bf16 planes: the bf16x3 kernel (conv_igemm.hip) is then one more
        // launch geometry (70-74) of the SAME fp32 convolution for the autotuner (ACCEL_WITHHOLD=split withholds it)
        const int ft = (int)kv_int(kv, "tile", -1);
        const bool forced = (ft >= CONV_TILE_B3 && ft < CONV_TILE_B3D + CONV_TILE_B3D_N) || (ft >= 90 && ft <= 96);
        if (!c.f16 && c.Cin % 4 == 0 && cout_store > 4 && (!withheld("split") || forced)) {
            std::vector<uint16_t> pb;
            pack_bf16x3(packed, c.deconv2x ? 4 : 1, rows, c.K_pad, pb, c.w_plane);
            void* d3 = nullptr;
            if ((rc = dev_upload(p, pb.data(), pb.size() * sizeof(uint16_t), &d3))) return rc;
            c.wb3 = d3;
            const bool forced_r = (ft >= CONV_TILE_B3R && ft < CONV_TILE_B3D + CONV_TILE_B3D_N) || (ft >= 90 && ft <= 96);
            if (!withheld("b3r") || forced_r) {      // the fragment-ordered copy for the second-generation kernel
                pack_bf16x3r(packed, c.deconv2x ? 4 : 1, rows, c.K_pad, pb, c.w_plane);
                void* d4 = nullptr;
                if ((rc = dev_upload(p, pb.data(), pb.size() * sizeof(uint16_t), &d4))) return rc;
                c.wb3r = d4;
            }
        } else if (forced && !(c.f16 == 1 && c.wb3r && ft >= CONV_TILE_B3R && ft < CONV_TILE_B3D + CONV_TILE_B3D_N)) {
            return fail(ACCEL_ERR_ARG, "conv %s: the bf16x3 kernel takes layers with more than "
                                       "4 output channels only", op.name.c_str());
        }
    }

    // epilogue scale / shift
    std::vector<float> scale(rows, 0.f), shift(rows, 0.f);
    for (int i = 0; i < cout; ++i) scale[i] = 1.f;
    if (kv_has(kv, "bn")) {
        std::vector<float> s, b;
        if ((rc = bn_fold(m, kv_str(kv, "bn"), (float)kv_f(kv, "eps", 1e-5), (int)kv_int(kv, "fixg", 0), cout, s, b))) return rc;
        for (int i = 0; i < cout; ++i) { scale[i] = s[i]; shift[i] = b[i]; }
    }
    if (kv_has(kv, "bias")) {
        const HostParam* bias = get_param(m, kv_str(kv, "bias"));
        if (!bias) return fail(ACCEL_ERR_PARAM, "%s not initialized", kv_str(kv, "bias").c_str());
        if ((int)bias->numel() != cout) return fail(ACCEL_ERR_PARAM, "shape inconsistent for %s", kv_str(kv, "bias").c_str());
        for (int i = 0; i < cout; ++i) shift[i] += bias->data[i] * scale[i];
    }
    if (kv_has(kv, "mul")) {
        const float mul = (float)kv_f(kv, "mul", 1.0);
        for (int i = 0; i < cout; ++i) { scale[i] *= mul; shift[i] *= mul; }
    }
    void *ds = nullptr, *db = nullptr;
    if ((rc = dev_upload(p, scale.data(), rows * sizeof(float), &ds)) ||
        (rc = dev_upload(p, shift.data(), rows * sizeof(float), &db))) return rc;
    c.scale = static_cast<const float*>(ds);
    c.shift = static_cast<const float*>(db);
    {
        const char* sp = getenv("ACCEL_SPLIT");
        const bool h2 = sp ? !strcmp(sp, "h2") : p->split == 1;
        if (c.wb3r && !c.f16 && h2) {
            std::vector<uint16_t> ph;
            std::vector<int> q;
            pack_h2r(packed, c.deconv2x ? 4 : 1, rows, c.K_pad, ph, c.w_plane, q);
            void *dh = nullptr, *dsh = nullptr;
            if ((rc = dev_upload(p, ph.data(), ph.size() * sizeof(uint16_t), &dh))) return rc;
            std::vector<float> sh(rows);
            for (int i = 0; i < rows; ++i) sh[i] = std::ldexp(scale[i], -q[i]);
            if ((rc = dev_upload(p, sh.data(), rows * sizeof(float), &dsh))) return rc;
            c.wh2r = dh;
            c.scale_h2 = static_cast<const float*>(dsh);
            ++p->n_h2;      // (the layer's input range slot, ConvParams::xr_slot, is assigned once every op is known: assign_range_slots)
        }
    }
    if (op.c.set) {
        std::vector<float> s2(rows, 0.f), b2(rows, 0.f), s, b;
        if (kv_has(kv, "bias2")) {          // out2 = relu(v + bias2): the biased, activated copy beside a raw linear output
            const HostParam* bias = get_param(m, kv_str(kv, "bias2"));
            if (!bias) return fail(ACCEL_ERR_PARAM, "%s not initialized", kv_str(kv, "bias2").c_str());
            if ((int)bias->numel() != cout) return fail(ACCEL_ERR_PARAM, "shape inconsistent for %s", kv_str(kv, "bias2").c_str());
            for (int i = 0; i < cout; ++i) { s2[i] = 1.f; b2[i] = bias->data[i]; }
        } else {
            if (!kv_has(kv, "bn2")) return fail(ACCEL_ERR_PLAN, "conv %s: out2 needs bn2 or bias2", op.name.c_str());
            if ((rc = bn_fold(m, kv_str(kv, "bn2"), (float)kv_f(kv, "eps2", 2e-5), (int)kv_int(kv, "fixg2", 0), cout, s, b))) return rc;
            for (int i = 0; i < cout; ++i) { s2[i] = s[i]; b2[i] = b[i]; }
        }
        void *d2 = nullptr, *e2 = nullptr;
        if ((rc = dev_upload(p, s2.data(), rows * sizeof(float), &d2)) ||
            (rc = dev_upload(p, b2.data(), rows * sizeof(float), &e2))) return rc;
        c.scale2 = static_cast<const float*>(d2);
        c.shift2 = static_cast<const float*>(e2);
        c.y2 = op.c.ptr; c.y2Cs = op.c.Cs;
    }
    c.x = op.a.ptr; c.xCs = op.a.Cs; c.H = op.a.H; c.W = op.a.W;
    c.y = op.b.ptr; c.yCs = op.b.Cs;
    if (op.d.set) { c.res = op.d.ptr; c.resCs = op.d.Cs; }
    c.Cout_store = cout_store;
    c.Cout = cout;
    // batch: the GEMM M dimension runs over (n, oy, ox); images are stacked pixel-major in every view
    if (op.b.N != op.a.N || (op.c.set && op.c.N != op.a.N) || (op.d.set && op.d.N != op.a.N))
        return fail(ACCEL_ERR_PLAN, "conv %s: batch sizes of the views differ", op.name.c_str());
    c.M = op.a.N * c.Ho * c.Wo;
    auto extent = [](const BufRef& r, int cuse) {
        const size_t b = (((size_t)r.N * r.H * r.W - 1) * r.Cs + cuse) * (size_t)r.esize;
        return (unsigned)(b > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : b);
    };
    c.x_half = op.a.esize == 2; c.y_half = op.b.esize == 2; c.res_half = op.d.set && op.d.esize == 2;
    if (c.x_half || c.y_half || c.res_half) {
        // half views are read / written by the fp16 form of conv_b3d.hip alone
        if (c.f16 != 1 || !c.wb3r || c.Cin % 16 || op.c.set)
            return fail(ACCEL_ERR_PLAN, "conv %s: half views need an f16-mode layer with Cin %% 16 == 0, more than 4 output channels and a "
                                        "single output (plan option dtype=f16, ACCEL_WITHHOLD without b3r)", op.name.c_str());
    }
    if ((size_t)op.a.N * op.a.img() * 4 > 0xFFFFFFF0ull || (size_t)op.b.N * op.b.img() * 4 > 0xFFFFFFF0ull)
        return fail(ACCEL_ERR_PLAN, "conv %s: a batched view exceeds the 4 GiB a buffer resource can address", op.name.c_str());
    c.x_bytes = extent(op.a, c.Cin > op.a.Cs ? op.a.Cs : c.Cin);
    c.y_bytes = extent(op.b, cout_store);
    if (op.c.set) c.y2_bytes = extent(op.c, cout_store);
    if (op.d.set) c.res_bytes = extent(op.d, cout_store);
    c.act = (int)kv_int(kv, "act", 0);
    c.slope = (float)kv_f(kv, "slope", 0.1);
    c.w_bytes = (unsigned)((size_t)rows * c.K_pad * (c.f16 ? 2 : 4));     // of one plane / class
    {
        // tap table, one entry per 4-wide K granule: (dy, dx, input byte offset relative to tap 0 / channel 0).
        // FAST shapes read it wave-uniformly through the scalar unit, the others per lane; padded K and the
        // slack entries (the pipelined kernels prefetch up to two K steps past the end) are marked out of range.
        const int G = c.K_pad / 4, classes = c.deconv2x ? 4 : 1, slack = 48;
        std::vector<int> tab((size_t)classes * (G + slack) * 4, 0);
        for (int cls = 0; cls < classes; ++cls)
            for (int g = 0; g < G + slack; ++g) {
                const int k = g * 4, tap = k / c.Cin, ci = k % c.Cin;
                const int ky = tap / c.kw, kx = tap % c.kw;
                int* t = &tab[((size_t)cls * (G + slack) + g) * 4];
                if (g >= G || tap >= c.kh * c.kw) { t[0] = -(1 << 28); continue; }
                t[0] = ky * c.dh; t[1] = kx * c.dw;
                t[2] = ((ky * c.dh * c.W + kx * c.dw) * c.xCs + ci) * op.a.esize;
            }
        void* dt = nullptr;
        if ((rc = dev_upload(p, tab.data(), tab.size() * sizeof(int), &dt))) return rc;
        c.ktab = static_cast<const int4*>(dt);
    }
    c.force_tile = (int)kv_int(kv, "tile", -1);
    if (c.force_tile >= 0 && !conv_tile_valid(c.force_tile))
        return fail(ACCEL_ERR_ARG, "conv %s: launch geometry id %d is not part of this build", op.name.c_str(), c.force_tile);
    {
        // Winograd F(2x2,3x3) form of the layer (3x3 / stride 1 / dilation 1 / pad 1): weights transformed here, in
        // double, once; offered to the autotuner beside the direct tiles (ACCEL_WITHHOLD=winograd withholds it)
        const bool want = !withheld("winograd") || c.force_tile == CONV_TILE_WINO;
        c.M = op.a.N * c.Ho * c.Wo;
        if (want && !cols && cin == cin_pad && conv_wino_eligible(c)) {
            c.wino_rows = conv_wino_rows(cout_store);
            std::vector<float> wu((size_t)(cin_pad / 8) * 16 * c.wino_rows * 8, 0.f);
            conv_wino_pack(w->data.data(), cout, cin, cin_pad, c.wino_rows, wu.data());
            void* du = nullptr;
            if ((rc = dev_upload(p, wu.data(), wu.size() * sizeof(float), &du))) return rc;
            c.wu = static_cast<const float*>(du);
            c.wu_bytes = (unsigned)(wu.size() * sizeof(float));
            // accuracy budget: Winograd evaluations carry about 3x the rounding error of a direct one, and the plan decides which
            // layers may take the split form (conv key wb3=0 withholds it)
            const bool wb3_ok = kv_int(kv, "wb3", 1) != 0;
            const bool wb3_forced = c.force_tile == CONV_TILE_WINO_B3 || c.force_tile == CONV_TILE_WINO_B3U || c.force_tile == CONV_TILE_WINO_B3S;
            if (((!withheld("split") && !withheld("winograd_split") && wb3_ok) || wb3_forced) && !c.f16 && conv_wino_b3_eligible(c)) {
                // the same transformed weights as three exact bf16 planes in MFMA fragment order: launch geometry 41
                std::vector<unsigned short> ub;
                conv_wino_b3_pack(w->data.data(), cout, cin, c.wino_rows, ub);
                void* db = nullptr;
                if ((rc = dev_upload(p, ub.data(), ub.size() * sizeof(unsigned short), &db))) return rc;
                c.wub = db;
                c.wub_bytes = (unsigned)(ub.size() * sizeof(unsigned short));
                if (c.wh2r) {      // fp16x2 form of the layer: the same planes as two half terms
                    std::vector<int> q;
                    conv_wino_b3_pack_h2(w->data.data(), cout, cin, c.wino_rows, ub, q);
                    void *dh = nullptr, *dsh = nullptr;
                    if ((rc = dev_upload(p, ub.data(), ub.size() * sizeof(unsigned short), &dh))) return rc;
                    std::vector<float> sh(rows, 0.f);
                    for (int i = 0; i < rows && i < c.wino_rows; ++i) sh[i] = std::ldexp(scale[i], 2 - q[i]);      // x 4: V is split at a quarter of the pixel scale
                    if ((rc = dev_upload(p, sh.data(), rows * sizeof(float), &dsh))) return rc;
                    c.wubh = dh;
                    c.scale_h2w = static_cast<const float*>(dsh);
                }
            }
        } else if (c.force_tile == CONV_TILE_WINO || c.force_tile == CONV_TILE_WINO_B3 || c.force_tile == CONV_TILE_WINO_B3U || c.force_tile == CONV_TILE_WINO_B3S) {
            return fail(ACCEL_ERR_ARG, "conv %s: the Winograd kernel takes 3x3 / stride 1 / dilation 1 / pad 1 layers with even output "
                                       "size and channels in multiples of 8 only", op.name.c_str());
        }
    }
    {
        // direct stem kernel (7x7 / stride 2 / pad 3, 3-channel image, 64 output channels): per-lane weight arrangement
        const bool want = !withheld("stem") || c.force_tile == CONV_TILE_STEM || c.force_tile == CONV_TILE_STEM_B3;
        c.Cout_store = cout_store;
        c.res = op.d.set ? op.d.ptr : nullptr;
        if (want && !cols && cin == 3 && cout == 64 && conv_stem_eligible(c)) {
            std::vector<float> ws((size_t)conv_stem_pack_floats(), 0.f);
            conv_stem_pack(w->data.data(), cout, ws.data());
            void* dsw = nullptr;
            if ((rc = dev_upload(p, ws.data(), ws.size() * sizeof(float), &dsw))) return rc;
            c.wstem = static_cast<const float*>(dsw);
            if ((!withheld("split") && !withheld("stem_split")) || c.force_tile == CONV_TILE_STEM_B3) {
                // the same layer for the bf16 matrix cores: three exact bf16 planes in MFMA fragment order (launch geometry 51)
                std::vector<unsigned short> wb;
                conv_stem_b3_pack(w->data.data(), cout, wb);
                void* dsb = nullptr;
                if ((rc = dev_upload(p, wb.data(), wb.size() * sizeof(unsigned short), &dsb))) return rc;
                c.wstemb = dsb;
                if (c.wh2r) {      // fp16x2 form of the layer
                    std::vector<int> q;
                    conv_stem_b3_pack_h2(w->data.data(), cout, wb, q);
                    void *dh = nullptr, *dsh = nullptr;
                    if ((rc = dev_upload(p, wb.data(), wb.size() * sizeof(unsigned short), &dh))) return rc;
                    std::vector<float> sh(rows, 0.f);
                    for (int i = 0; i < rows && i < 64; ++i) sh[i] = std::ldexp(scale[i], -q[i]);
                    if ((rc = dev_upload(p, sh.data(), rows * sizeof(float), &dsh))) return rc;
                    c.wstemh = dh;
                    c.scale_h2s = static_cast<const float*>(dsh);
                }
            }
        } else if (c.force_tile == CONV_TILE_STEM || c.force_tile == CONV_TILE_STEM_B3) {
            return fail(ACCEL_ERR_ARG, "conv %s: the stem kernel takes 7x7 / stride 2 / pad 3 layers on 3-channel images with 64 "
                                       "output channels only", op.name.c_str());
        }
    }
    {
        // weight-stationary streaming kernel for the 1x1 expand layers with K = 64 / 128 (ACCEL_WITHHOLD=ws1x1 withholds it)
        const bool want = !withheld("ws1x1") || c.force_tile == CONV_TILE_WS;
        c.y2 = op.c.set ? op.c.ptr : nullptr;
        if (want && !cols && cin == cin_pad && conv_ws_eligible(c)) {
            std::vector<float> ww(conv_ws_pack_floats(cin, cout_store), 0.f);
            conv_ws_pack(w->data.data(), cout, cin, cout_store, ww.data());
            void* dw = nullptr;
            if ((rc = dev_upload(p, ww.data(), ww.size() * sizeof(float), &dw))) return rc;
            c.wws = static_cast<const float*>(dw);
            c.wws_bytes = (unsigned)(ww.size() * sizeof(float));
        } else if (c.force_tile == CONV_TILE_WS) {
            return fail(ACCEL_ERR_ARG, "conv %s: the weight-stationary kernel takes 1x1 / stride 1 layers from 64 to a multiple of 256 "
                                       "or from 128 to a multiple of 128 channels (ReLU or no activation, one output) only", op.name.c_str());
        }
    }
    c.narrow = (cout_store == 4 && !c.deconv2x && !op.c.set && c.force_tile < 0 && kv_int(kv, "narrow", 1)) ? 1 : 0;
    c.no_split = (int)kv_int(kv, "nosplit", 0);
    c.split_target = (int)kv_int(kv, "split_target", 0);      // (tests: the tuner's "split until ~N blocks" variants, forced)
    const size_t ws = conv_plan_split(c);
    if (ws > p->ws_bytes) p->ws_bytes = ws;
    return 0;
}

static int upload_param(accel_plan* p, const std::string& name, size_t expect, const float** out)
{
    const HostParam* h = get_param(p->m, name);
    if (!h) return fail(ACCEL_ERR_PARAM, "%s not initialized", name.c_str());
    if (expect && h->numel() != expect) return fail(ACCEL_ERR_PARAM, "shape inconsistent for %s: %zu elements, expected %zu", name.c_str(), h->numel(), expect);
    void* d = nullptr;
    int rc = dev_upload(p, h->data.data(), h->numel() * sizeof(float), &d);
    if (rc) return rc;
    *out = static_cast<const float*>(d);
    return 0;
}

// the pointer slot of a whole persistent input buffer read by a prep kernel (null for arena views and partial views)
static int input_slot(accel_plan* p, const BufRef& r, const float* const** out)
{
    *out = nullptr;
    if (r.space == "A" || r.off != 0) return 0;
    DevBuf& b = p->m->pbufs[r.space];
    ++b.prep_readers;
    if (!b.slot) {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, sizeof(void*)));
        b.slot = static_cast<const void**>(d);
        HIP_TRY(hipMemcpy(d, &b.ptr, sizeof(void*), hipMemcpyHostToDevice));
    }
    *out = reinterpret_cast<const float* const*>(b.slot);
    return 0;
}

// the score-resolution map of a plan's tail lives in the model's persistent buffer `scores` (shared by the key and the non-key plan, like `logits`)
static int scores_buffer(accel_plan* p, Op& op, const ScoreTailParams& q)
{
    const int zCs = roundup(q.ncls, 4);
    const size_t zbytes = (size_t)op.a.N * q.Hs * q.Ws * zCs * sizeof(float);
    DevBuf& zb = p->m->pbufs["scores"];
    if (!zb.ptr) {
        HIP_TRY(hipMalloc(&zb.ptr, zbytes));
        zb.bytes = zbytes;
        HIP_TRY(hipMemsetAsync(zb.ptr, 0, zbytes, p->m->ctx->stream));      // the pad channel stays zero
    } else if (zb.bytes != zbytes) {
        return fail(ACCEL_ERR_PLAN, "score_tail %s: the model's `scores` buffer has %zu bytes, this plan needs %zu", op.name.c_str(), zb.bytes, zbytes);
    }
    op.tail_z = static_cast<float*>(zb.ptr);
    op.tail_lowres = q;
    op.tail_lowres.left = op.tail_z; op.tail_lowres.lCs = zCs;
    op.tail_lowres.right = nullptr; op.tail_lowres.wr = nullptr;
    op.tail_lowres.uniform_w = 1;
    p->score_tpl = op.tail_lowres;
    p->score_tpl.cw = nullptr;
    p->has_score_tpl = true;
    p->score_images = op.a.N;
    p->score_zcs = zCs;
    return 0;
}

static int finalize_op(accel_plan* p, Op& op)
{
    const KV& kv = op.kv;
    int rc;
    switch (op.kind) {
    case OP_CONV:
        return finalize_conv(p, op);
    case OP_PREP_RGB: {
        if ((rc = parse_buf(kv, "src", op.a)) || (rc = parse_buf(kv, "dst", op.b))) return rc;
        if ((rc = resolve(p, op.a, "src")) || (rc = resolve(p, op.b, "dst"))) return rc;
        if ((rc = input_slot(p, op.a, &op.slot_a))) return rc;
        op.H = (int)kv_int(kv, "H"); op.W = (int)kv_int(kv, "W");
        if (op.b.Cs != 4) return fail(ACCEL_ERR_PLAN, "prep_rgb: dst must be NHWC4");
        if (kv_has(kv, "bn")) {
            std::vector<float> s, b;
            if ((rc = bn_fold(p->m, kv_str(kv, "bn"), (float)kv_f(kv, "eps", 2e-5), (int)kv_int(kv, "fixg", 1), 3, s, b))) return rc;
            void *ds = nullptr, *db = nullptr;
            if ((rc = dev_upload(p, s.data(), 12, &ds)) || (rc = dev_upload(p, b.data(), 12, &db))) return rc;
            op.p0 = static_cast<const float*>(ds); op.p1 = static_cast<const float*>(db);
        }
        return 0;
    }
    case OP_PREP_FLOW: {
        if ((rc = parse_buf(kv, "cur", op.a)) || (rc = parse_buf(kv, "prev", op.b)) || (rc = parse_buf(kv, "dst", op.c))) return rc;
        if ((rc = resolve(p, op.a, "cur")) || (rc = resolve(p, op.b, "prev")) || (rc = resolve(p, op.c, "dst"))) return rc;
        if ((rc = input_slot(p, op.a, &op.slot_a)) || (rc = input_slot(p, op.b, &op.slot_b))) return rc;
        op.H = (int)kv_int(kv, "H"); op.W = (int)kv_int(kv, "W");
        if (op.c.Cs != 8 || (op.H & 1) || (op.W & 1)) return fail(ACCEL_ERR_PLAN, "prep_flow: dst must be NHWC8 and H,W even");
        return 0;
    }
    case OP_POOL: {
        if ((rc = parse_buf(kv, "in", op.a)) || (rc = parse_buf(kv, "out", op.b))) return rc;
        if ((rc = resolve(p, op.a, "in")) || (rc = resolve(p, op.b, "out"))) return rc;
        PoolParams& q = op.pool;
        memset(&q, 0, sizeof q);
        q.x = op.a.ptr; q.y = op.b.ptr; q.xCs = op.a.Cs; q.yCs = op.b.Cs;
        q.C4 = roundup(op.a.C, 4) / 4;
        q.H = op.a.H; q.W = op.a.W; q.Ho = op.b.H; q.Wo = op.b.W;
        kv_pair(kv, "k", q.kh, q.kw, 2, 2);
        kv_pair(kv, "s", q.sh, q.sw, 2, 2);
        kv_pair(kv, "p", q.ph, q.pw, 0, 0);
        q.is_max = kv_str(kv, "kind", "max") == "max";
        q.relu = (int)kv_int(kv, "act", 0) == 1;
        if (kv_has(kv, "bn")) {
            std::vector<float> s, b;
            if ((rc = bn_fold(p->m, kv_str(kv, "bn"), (float)kv_f(kv, "eps", 2e-5), (int)kv_int(kv, "fixg", 0), op.a.C, s, b))) return rc;
            s.resize(q.C4 * 4, 0.f); b.resize(q.C4 * 4, 0.f);
            void *ds = nullptr, *db = nullptr;
            if ((rc = dev_upload(p, s.data(), s.size() * 4, &ds)) || (rc = dev_upload(p, b.data(), b.size() * 4, &db))) return rc;
            q.scale = static_cast<const float*>(ds); q.shift = static_cast<const float*>(db);
        }
        return 0;
    }
    case OP_WARP: {
        if ((rc = parse_buf(kv, "feat", op.a)) || (rc = parse_buf(kv, "flow", op.b)) || (rc = parse_buf(kv, "out", op.c))) return rc;
        if ((rc = resolve(p, op.a, "feat")) || (rc = resolve(p, op.b, "flow")) || (rc = resolve(p, op.c, "out"))) return rc;
        if (op.a.C % 4) return fail(ACCEL_ERR_PLAN, "warp: C must be a multiple of 4");
        if (kv_has(kv, "out2")) {           // second output relu(warped + bias[c])
            if ((rc = parse_buf(kv, "out2", op.d)) || (rc = resolve(p, op.d, "out2"))) return rc;
            const HostParam* bias = get_param(p->m, kv_str(kv, "bias"));
            if (!bias) return fail(ACCEL_ERR_PARAM, "%s not initialized", kv_str(kv, "bias").c_str());
            if ((int)bias->numel() != op.a.C) return fail(ACCEL_ERR_PARAM, "shape inconsistent for %s", kv_str(kv, "bias").c_str());
            void* db = nullptr;
            if ((rc = dev_upload(p, bias->data.data(), (size_t)op.a.C * sizeof(float), &db))) return rc;
            op.p0 = static_cast<const float*>(db);
        }
        return 0;
    }
    case OP_DCN_COLS: {
        if ((rc = parse_buf(kv, "in", op.a)) || (rc = parse_buf(kv, "off", op.b)) || (rc = parse_buf(kv, "out", op.c))) return rc;
        // the column buffer may be a half view (f16-mode plans: its only reader, the GEMM, rounds the columns to half anyway)
        if ((rc = resolve(p, op.a, "in")) || (rc = resolve(p, op.b, "off")) || (rc = resolve(p, op.c, "out", p->f16 == 1))) return rc;
        DcnColsParams& q = op.dcn;
        memset(&q, 0, sizeof q);
        q.x = op.a.ptr; q.off = op.b.ptr; q.col = op.c.ptr;
        q.col_half = op.c.esize == 2;
        q.C = roundup(op.a.C, 4); q.xCs = op.a.Cs; q.offCs = op.b.Cs; q.colCs = op.c.Cs;
        q.H = op.a.H; q.W = op.a.W; q.Ho = op.c.H; q.Wo = op.c.W;
        kv_pair(kv, "k", q.kh, q.kw, 3, 3);
        kv_pair(kv, "s", q.sh, q.sw, 1, 1);
        kv_pair(kv, "p", q.ph, q.pw, 0, 0);
        kv_pair(kv, "d", q.dh, q.dw, 1, 1);
        q.dg = (int)kv_int(kv, "dg", 1);
        if (op.a.C % q.dg || (op.a.C / q.dg) % 4) return fail(ACCEL_ERR_PLAN, "dcn_cols: channels per deformable group must be a multiple of 4");
        if (q.colCs < q.kh * q.kw * q.C) return fail(ACCEL_ERR_PLAN, "dcn_cols: column buffer too narrow");
        return 0;
    }
    case OP_SCORE_TAIL: {
        if ((rc = parse_buf(kv, "left", op.a)) || (rc = parse_buf(kv, "right", op.b)) ||
            (rc = parse_buf(kv, "logits", op.c)) || (rc = parse_buf(kv, "labels", op.d))) return rc;
        if ((rc = resolve(p, op.a, "left")) || (rc = resolve(p, op.c, "logits")) || (rc = resolve(p, op.d, "labels"))) return rc;
        if (op.b.set && (rc = resolve(p, op.b, "right"))) return rc;
        ScoreTailParams& q = op.tail;
        memset(&q, 0, sizeof q);
        q.ncls = (int)kv_int(kv, "ncls", 19);
        q.softmax = (int)kv_int(kv, "softmax", 0);
        q.left = op.a.ptr; q.lCs = op.a.Cs; q.Hs = op.a.H; q.Ws = op.a.W;
        q.H = (int)kv_int(kv, "H"); q.W = (int)kv_int(kv, "W");
        if (q.H != 16 * q.Hs || q.W != 16 * q.Ws) return fail(ACCEL_ERR_PLAN, "score_tail: output must be 16x the score map");
        q.logits = op.c.ptr; q.labels = reinterpret_cast<unsigned char*>(op.d.ptr);
        const size_t wn = (size_t)q.ncls * 32 * 32;
        if ((rc = upload_param(p, kv_str(kv, "wl"), wn, &q.wl))) return rc;
        if (!op.b.set) {      // single head: the wide kernel applies when one filter serves every class
            const HostParam* hl = get_param(p->m, kv_str(kv, "wl"));
            bool uni = hl && kv_int(kv, "lowres", 1) != 0;
            for (int c = 1; c < q.ncls && uni; ++c) uni = !memcmp(hl->data.data(), hl->data.data() + (size_t)c * 1024, 4096);
            q.uniform_w = uni ? 1 : 0;
        }
        if (op.b.set) {
            q.right = op.b.ptr; q.rCs = op.b.Cs;
            // the right map may be one row / column larger than the left one (stride-32 branch upsampled 2x, frame size a multiple of
            // 16 only): Deconvolution 32x32/16 + Crop(8, 8) to the frame never reaches its last row / column
            if (op.b.H < op.a.H || op.b.W < op.a.W || op.b.H > op.a.H + 1 || op.b.W > op.a.W + 1)
                return fail(ACCEL_ERR_PLAN, "score_tail: right score map %dx%d does not fit the left one %dx%d", op.b.H, op.b.W, op.a.H, op.a.W);
            if (op.b.H != op.a.H || op.b.W != op.a.W) { q.rHs = op.b.H; q.rWs = op.b.W; }
            if ((rc = upload_param(p, kv_str(kv, "wr"), wn, &q.wr)) ||
                (rc = upload_param(p, kv_str(kv, "cw"), (size_t)q.ncls * 2 * q.ncls, &q.cw)) ||
                (rc = upload_param(p, kv_str(kv, "cb"), (size_t)q.ncls, &q.cb))) return rc;
            // fast path: both upsampling filters identical for every class (the reference freezes them at
            // the bilinear init) -> fuse at score resolution, then upsample ncls maps once
            const HostParam* hl = get_param(p->m, kv_str(kv, "wl"));
            const HostParam* hr = get_param(p->m, kv_str(kv, "wr"));
            bool uniform = kv_int(kv, "lowres", 1) != 0 && !q.rHs;      // fusing at score resolution needs the two maps on one grid
            for (int c = 0; c < q.ncls && uniform; ++c)
                uniform = !memcmp(hl->data.data(), hl->data.data() + (size_t)c * 1024, 4096) &&
                          !memcmp(hl->data.data(), hr->data.data() + (size_t)c * 1024, 4096);
            if (uniform) {
                if ((rc = scores_buffer(p, op, q))) return rc;
                op.tail_lowres.cw = nullptr;      // cb stays: the bias is added after upsampling
            }
        }
        if (!op.b.set && q.uniform_w) {      // single head with one filter for every class: its score map goes through `scores` as well (a copy)
            if ((rc = scores_buffer(p, op, q))) return rc;
            op.tail_copy = true;
        }
        return 0;
    }
    case OP_COPY: {
        if ((rc = parse_buf(kv, "src", op.a)) || (rc = parse_buf(kv, "dst", op.b))) return rc;
        if ((rc = resolve(p, op.a, "src")) || (rc = resolve(p, op.b, "dst"))) return rc;
        if (op.a.H != op.b.H || op.a.W != op.b.W || op.a.C != op.b.C || op.a.N != op.b.N) return fail(ACCEL_ERR_PLAN, "copy: view shapes differ");
        return 0;
    }
    case OP_EXPORT_NCHW:
    case OP_IMPORT_NCHW: {
        if ((rc = parse_buf(kv, "src", op.a)) || (rc = parse_buf(kv, "dst", op.b))) return rc;
        if ((rc = resolve(p, op.a, "src")) || (rc = resolve(p, op.b, "dst"))) return rc;
        return 0;
    }
    }
    return fail(ACCEL_ERR_PLAN, "unhandled op kind");
}

static int launch_op(accel_plan* p, Op& op)
{
    hipStream_t st = p->m->ctx->stream;
    hipError_t e = hipSuccess;
    switch (op.kind) {
    case OP_CONV: e = launch_conv_igemm(op.conv, st); break;
    // batch: blockIdx.z = image for the byte movers (fixed stride between images), M = N*Ho*Wo inside the convolution
    case OP_PREP_RGB: e = launch_prep_rgb(op.a.ptr, op.b.ptr, op.H, op.W, op.p0, op.p1, op.b.N, st, op.slot_a, op.yr); break;
    case OP_PREP_FLOW: e = launch_prep_flow(op.a.ptr, op.b.ptr, op.c.ptr, op.H, op.W, op.c.N, st, op.slot_a, op.slot_b, op.yr); break;
    case OP_POOL: {
        PoolParams q = op.pool;
        q.N = op.a.N; q.x_img = op.a.img(); q.y_img = op.b.img();
        e = launch_pool(q, st);
        break;
    }
    case OP_WARP:
        e = launch_flow_warp(op.a.ptr, op.a.Cs, op.b.ptr, op.b.Cs, op.c.ptr, op.c.Cs, op.a.C, op.a.H, op.a.W,
                             op.d.set ? op.d.ptr : nullptr, op.d.Cs, op.p0, op.a.N, st, op.yr, op.y2r);
        break;
    case OP_DCN_COLS: {
        DcnColsParams q = op.dcn;
        q.N = op.a.N;
        e = launch_dcn_cols(q, st);
        break;
    }
    case OP_SCORE_TAIL: {
        const int N = op.a.N;
        if (op.tail_z && op.tail_copy)
            e = launch_copy_view(op.tail.left, op.tail.lCs, op.tail_z, op.tail_lowres.lCs, op.tail.ncls, N * op.tail.Hs * op.tail.Ws, st, nullptr);
        else if (op.tail_z)     // low-resolution fusion is pixel-wise: one launch over all images
            e = launch_score_fuse_lowres(op.tail.left, op.tail.lCs, op.tail.right, op.tail.rCs, op.tail.cw, op.tail_z,
                                         op.tail_lowres.lCs, op.tail.ncls, N * op.tail.Hs * op.tail.Ws, st);
        if (e == hipSuccess) {
            ScoreTailParams q = op.tail_z ? op.tail_lowres : op.tail;
            q.N = N;
            e = launch_score_tail(q, st);
        }
        break;
    }
    case OP_COPY: e = launch_copy_view(op.a.ptr, op.a.Cs, op.b.ptr, op.b.Cs, op.a.C, op.a.N * op.a.H * op.a.W, st, op.yr); break;
    case OP_EXPORT_NCHW:
        for (int n = 0; n < op.a.N && e == hipSuccess; ++n)
            e = launch_nhwc_to_nchw(op.a.ptr + n * op.a.img(), op.a.Cs, op.b.ptr + (size_t)n * op.a.C * op.a.H * op.a.W, op.a.C, op.a.H, op.a.W, st);
        break;
    case OP_IMPORT_NCHW:
        for (int n = 0; n < op.b.N && e == hipSuccess; ++n)
            e = launch_nchw_to_nhwc(op.a.ptr + (size_t)n * op.b.C * op.b.H * op.b.W, op.b.ptr + n * op.b.img(), op.b.Cs, op.b.C, op.b.H, op.b.W, st);
        break;
    }
    if (e != hipSuccess) return fail(ACCEL_ERR_HIP, "launch of op %s (%s) failed: %s", op.kind_name.c_str(), op.name.c_str(), hipGetErrorString(e));
    return 0;
}

// Issues the plan: the range slots are zeroed, then every op in list order on the context's compute stream (under stream capture this
// becomes a linear graph).  An fp16x2-form convolution whose input tensor no earlier op of the plan measured (Op::measure) is preceded
// by the pass that does (misc.hip range_amax_kernel).
static int launch_measure(accel_plan* p, Op& op)      // what goes in front of an fp16x2-form convolution: the measuring pass (if any), the fold (if any)
{
    const ConvParams& c = op.conv;
    hipError_t e = hipSuccess;
    if (op.measure) e = launch_range_amax(c.x, (long)op.a.N * c.H * c.W, c.Cin, c.xCs, const_cast<unsigned*>(c.xr_slot), p->m->ctx->stream);
    if (e == hipSuccess && (op.measure || op.fold)) e = launch_range_fold(const_cast<unsigned*>(c.xr_slot), p->range_flag_dev, (int)(&op - p->ops.data()), p->m->ctx->stream);
    if (e != hipSuccess) return fail(ACCEL_ERR_HIP, "range pass of conv %s failed: %s", op.name.c_str(), hipGetErrorString(e));
    return 0;
}

static int run_eager(accel_plan* p)
{
    if (p->n_slots && launch_range_clear(p->range, p->n_slots, p->m->ctx->stream) != hipSuccess)
        return fail(ACCEL_ERR_HIP, "clearing the range slots of plan '%s' failed", p->role.c_str());
    for (size_t i = 0; i < p->ops.size(); ++i) {
        Op& op = p->ops[i];
        int rc = (op.measure || op.fold) ? launch_measure(p, op) : 0;
        if (!rc) rc = launch_op(p, op);
        if (rc) return rc;
        // ACCEL_CHECK_FINITE=1 (diagnostics, eager runs only -- set ACCEL_HIP_GRAPH=0): the first op of a run whose fp32 output holds a
        // non-finite value is named on stderr
        static const char* chk = getenv("ACCEL_CHECK_FINITE");
        if (chk && chk[0] == '1' && !p->gexec && (op.kind == OP_CONV || op.kind == OP_POOL || op.kind == OP_PREP_RGB) && p->n_slots) {
            const BufRef& o = op.b;
            if (o.esize == 4 && o.ptr && p->check_scratch) {      // (one slot of the plan's own allocation: freed with the plan)
                unsigned* scratch = p->check_scratch;
                hipStream_t st = p->m->ctx->stream;
                launch_range_clear(scratch, 1, st);
                launch_range_amax(o.ptr, (long)o.N * o.H * o.W, roundup(o.C, 4), o.Cs, scratch, st);
                launch_range_fold(scratch, nullptr, 0, st);
                unsigned bits = 0;
                hipStreamSynchronize(st);
                hipMemcpy(&bits, scratch, 4, hipMemcpyDeviceToHost);
                if (bits >= 0x7F800000u) fprintf(stderr, "[accel check] plan %s op %zu %s %s: output holds a non-finite value (largest |x| bits 0x%08x)\n", p->role.c_str(), i, op.kind_name.c_str(), op.name.c_str(), bits);
            }
        }
    }
    return 0;
}

// the word an fp16x2-form convolution raises when the range of its input is not finite: a loud error instead of NaN frames
static int range_check(accel_plan* p)
{
    if (!p->range_flag || !*p->range_flag) return 0;
    const unsigned w = *p->range_flag, i = (w & 0x7FFFFFFFu) - 1u;      // one word: (op index + 1) | bit 31 for a NaN (misc.hip range_fold_kernel)
    *p->range_flag = 0u;      // reported once
    return fail(ACCEL_ERR_RANGE, "plan '%s': the input of conv %s was not finite in an earlier run (the largest |value| stored into it was %s; fp16x2 "
                "form of the fp32 layers: the frames of that run are not trustworthy; ACCEL_SPLIT=b3 selects the bf16x3 form, which propagates "
                "non-finite values like fp32 arithmetic)",
                p->role.c_str(), i < p->ops.size() ? p->ops[i].name.c_str() : "?", (w & 0x80000000u) ? "a NaN" : "an infinity");
}

// Range slots (kernels.h, range.h): which tensors need one, who raises it, who has to measure for himself.
//   * a slot per physical buffer some convolution WITH an fp16x2 form reads as its input (`xr=<id>` from the lowering; a convolution
//     of a hand-written plan without an id gets a private slot): assign_range_slots, once every op is finalized and before anything
//     is launched (the tuner times the fp16x2 candidates);
//   * resolve_range_flags, before and again AFTER the tuner has chosen the launch geometries: a reader counts only if the geometry it
//     runs on executes the fp16x2 form; every op that writes into a buffer a counting reader reads (`yr=` / `y2r=`) and has the range
//     epilogue gets the slot's address (the others get none: their epilogue does nothing); a counting reader is marked `measure`
//     unless at least one earlier op of the plan wrote its buffer and every such writer has the epilogue (persistent buffers written
//     by another plan or by the host, tensors imported by import_nchw), and `fold` if it is the first reader after a write.
static bool conv_has_h2_form(const ConvParams& c) { return c.wh2r || c.wubh || c.wstemh; }
static bool conv_runs_h2(const ConvParams& c)
{
    if (c.f16 || c.narrow) return false;
    const int t = c.force_tile;
    return ((t == CONV_TILE_WINO_B3 || t == CONV_TILE_WINO_B3U || t == CONV_TILE_WINO_B3S) && c.wubh) || (t == CONV_TILE_STEM_B3 && c.wstemh) ||
           (((t >= CONV_TILE_B3R && t < CONV_TILE_B3R + 6) || (t >= 90 && t <= 96)) && c.wh2r);      // (90-96: the ablation builds of geometry 76, diagnostics library only)
}

static int assign_range_slots(accel_plan* p)
{
    std::map<int, int>& slot_of = p->slot_of;      // lowering id -> slot index
    int n = 0;
    p->in_slot.assign(p->ops.size(), -1);
    for (size_t i = 0; i < p->ops.size(); ++i) {
        Op& op = p->ops[i];
        if (op.kind != OP_CONV || !conv_has_h2_form(op.conv)) continue;
        if (op.rs_in < 0) p->in_slot[i] = n++;
        else {
            auto it = slot_of.find(op.rs_in);
            if (it == slot_of.end()) it = slot_of.insert({op.rs_in, n++}).first;
            p->in_slot[i] = it->second;
        }
    }
    p->n_slots = n;
    if (!n) return 0;
    HIP_TRY(hipMalloc((void**)&p->range, (size_t)(n + 1) * RANGE_WORDS * sizeof(unsigned)));      // (+ 1: the scratch slot of ACCEL_CHECK_FINITE)
    p->owned.push_back(p->range);
    HIP_TRY(hipMemset(p->range, 0, (size_t)(n + 1) * RANGE_WORDS * sizeof(unsigned)));
    p->check_scratch = p->range + (size_t)n * RANGE_WORDS;
    HIP_TRY(hipHostMalloc((void**)&p->range_flag, 2 * sizeof(unsigned), hipHostMallocMapped));      // [0]: (first offender's op index + 1) | bit 31 for a NaN
    p->range_flag[0] = p->range_flag[1] = 0u;
    HIP_TRY(hipHostGetDevicePointer((void**)&p->range_flag_dev, p->range_flag, 0));
    for (size_t i = 0; i < p->ops.size(); ++i)
        if (p->in_slot[i] >= 0) p->ops[i].conv.xr_slot = p->range + (size_t)p->in_slot[i] * RANGE_WORDS;
    return 0;
}

static void resolve_range_flags(accel_plan* p, bool (*counts)(const ConvParams&))
{
    const int n = p->n_slots;
    if (!n) return;
    std::vector<char> used(n, 0);      // slots with a counting reader
    for (size_t i = 0; i < p->ops.size(); ++i)
        if (p->in_slot[i] >= 0 && counts(p->ops[i].conv)) used[p->in_slot[i]] = 1;
    auto addr = [&](int id) -> unsigned* {
        auto it = id < 0 ? p->slot_of.end() : p->slot_of.find(id);
        return (it == p->slot_of.end() || !used[it->second]) ? nullptr : p->range + (size_t)it->second * RANGE_WORDS;
    };
    std::vector<int> state(n, 0);      // 0: no writer yet, 1: every writer so far had the epilogue, 2: some writer had not
    std::vector<char> dirty(n, 0);     // partial words written since the last fold
    auto wrote = [&](int id, bool has_epilogue) {
        auto it = id < 0 ? p->slot_of.end() : p->slot_of.find(id);
        if (it == p->slot_of.end()) return;
        int& st = state[it->second];
        st = (has_epilogue && st != 2) ? 1 : 2;
        dirty[it->second] = 1;
    };
    for (size_t i = 0; i < p->ops.size(); ++i) {
        Op& op = p->ops[i];
        op.measure = op.fold = false;
        if (p->in_slot[i] >= 0 && counts(op.conv)) {      // reader first: an op never feeds itself
            const int sl = p->in_slot[i];
            op.measure = state[sl] != 1;
            op.fold = op.measure || dirty[sl];      // (a measuring pass raises partial words as well)
            dirty[sl] = 0;
        }
        switch (op.kind) {
        case OP_CONV:
            // (half outputs: the lowering keeps a buffer half only if every reader is an f16-mode convolution -- none has an fp16x2 form)
            // and the f16-mode kernels of conv_b3d.hip have no range epilogue: a reader of their output measures its view)
            {
                const int t = op.conv.force_tile;
                const bool ep = !op.conv.y_half && !(t >= CONV_TILE_B3D && t < CONV_TILE_B3D + CONV_TILE_B3D_N) && !(op.conv.x_half || op.conv.res_half);
                op.conv.yr = ep ? addr(op.rs_out) : nullptr;
                op.conv.y2r = ep ? addr(op.rs_out2) : nullptr;
                wrote(op.rs_out, ep); wrote(op.rs_out2, ep);
            }
            break;
        case OP_POOL: op.pool.yr = addr(op.rs_out); wrote(op.rs_out, true); break;
        case OP_DCN_COLS: op.dcn.yr = op.dcn.col_half ? nullptr : addr(op.rs_out); wrote(op.rs_out, !op.dcn.col_half); break;
        case OP_WARP: case OP_PREP_RGB: case OP_PREP_FLOW: case OP_COPY:
            op.yr = addr(op.rs_out); op.y2r = addr(op.rs_out2);
            wrote(op.rs_out, true); wrote(op.rs_out2, true);
            break;
        default:
            wrote(op.rs_out, false); wrote(op.rs_out2, false);    // import_nchw, ...: no epilogue
            break;
        }
    }
}

// ---------------------------------------------------------------------------
// Per-shape autotuning of the conv launch geometry (tile variant x split-K
// policy), timed on the real buffers at finalisation.  All candidates compute
// the same contraction; only the schedule differs.  ACCEL_AUTOTUNE=0 keeps the
// static heuristic.
// ---------------------------------------------------------------------------
struct TuneKey {
    int v[16];
    bool operator<(const TuneKey& o) const { return memcmp(v, o.v, sizeof v) < 0; }
};
struct TuneVal { int tile, split_target, no_split; };
static std::map<TuneKey, TuneVal> g_tune_cache;
static std::map<TuneKey, TuneVal> g_tune_shipped;      // entries of the in-tree table (never written back to the user file)
static bool g_tune_loaded = false;
// ACCEL_POISON=1 (debugging): fresh arenas, persistent buffers and workspaces are filled with NaN bit patterns, so a kernel
// that consumes memory no op of the plan wrote shows up as NaNs in the outputs instead of as run-to-run noise
static void poison(void* ptr, size_t bytes)
{
    static const char* e = getenv("ACCEL_POISON");
    if (e && e[0] == '1' && ptr) { hipMemset(ptr, 0xFF, bytes); hipDeviceSynchronize(); }
}
static int g_tune_hits = 0, g_tune_timed = 0;     // decisions replayed from a table / taken by timing in this process

// Launch-geometry decisions are persisted so that they are REPRODUCIBLE: the summation order of a convolution (tile,
// split-K) decides its last bits, and a decision taken by timing can differ from process to process.
//   1. <directory of libaccel_hip.so>/tune/gfx950.tune  -- shipped with the sources, covers the BASELINE workloads;
//   2. the user file: $ACCEL_TUNE_CACHE, else $XDG_CACHE_HOME/accel_amd/gfx950.tune, else ~/.cache/accel_amd/gfx950.tune
//      -- shapes the shipped table lacks are timed ONCE, written there, and replayed by every later process.
// A file whose version tag differs from ACCEL_TUNE_VERSION (the tile-id set changed) is ignored.
// ACCEL_TUNE_SHIPPED=0 skips the shipped table (used when regenerating it), ACCEL_AUTOTUNE=0 disables timing altogether
// (static heuristic for every shape that is in neither file).
#define ACCEL_TUNE_VERSION "accel_hip-tune-9"

static std::string lib_dir()
{
    Dl_info info;
    if (!dladdr((const void*)&accel_last_error, &info) || !info.dli_fname) return ".";
    std::string path(info.dli_fname);
    const size_t slash = path.rfind('/');
    return slash == std::string::npos ? "." : path.substr(0, slash);
}

static std::string user_tune_path()
{
    if (const char* e = getenv("ACCEL_TUNE_CACHE")) return e;
    std::string dir;
    if (const char* x = getenv("XDG_CACHE_HOME")) dir = x;
    else if (const char* h = getenv("HOME")) dir = std::string(h) + "/.cache";
    else return "";
    return dir + "/accel_amd/gfx950.tune";
}

static void tune_file_load(const std::string& path, std::map<TuneKey, TuneVal>& into)
{
    if (path.empty()) return;
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return;
    char tag[64] = {0};
    if (fscanf(f, "# %63s", tag) != 1 || strcmp(tag, ACCEL_TUNE_VERSION)) { fclose(f); return; }
    TuneKey k; TuneVal v;
    for (;;) {
        int n = 0;
        for (int i = 0; i < 16; ++i) n += fscanf(f, "%d", &k.v[i]);
        n += fscanf(f, "%d %d %d", &v.tile, &v.split_target, &v.no_split);
        if (n != 19) break;
        into[k] = v;
    }
    fclose(f);
}

static void tune_cache_load()
{
    if (g_tune_loaded) return;
    g_tune_loaded = true;
    tune_file_load(user_tune_path(), g_tune_cache);
    const char* sh = getenv("ACCEL_TUNE_SHIPPED");
    if (!(sh && sh[0] == '0')) {
        tune_file_load(lib_dir() + "/tune/gfx950.tune", g_tune_shipped);
        for (const auto& kv : g_tune_shipped) g_tune_cache[kv.first] = kv.second;     // the shipped decision wins
    }
}

static void tune_cache_save()
{
    const std::string path = user_tune_path();
    if (path.empty()) return;
    const bool explicit_file = getenv("ACCEL_TUNE_CACHE") != nullptr;     // an explicit file receives every entry
    if (!explicit_file) {
        const size_t slash = path.rfind('/');
        std::string dir = path.substr(0, slash), up = dir.substr(0, dir.rfind('/'));
        mkdir(up.c_str(), 0755);
        mkdir(dir.c_str(), 0755);
    }
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "w");
    if (!f) return;
    fprintf(f, "# %s\n", ACCEL_TUNE_VERSION);
    for (const auto& kv : g_tune_cache) {
        if (!explicit_file && g_tune_shipped.count(kv.first)) continue;
        for (int i = 0; i < 16; ++i) fprintf(f, "%d ", kv.first.v[i]);
        fprintf(f, "%d %d %d\n", kv.second.tile, kv.second.split_target, kv.second.no_split);
    }
    fclose(f);
    rename(tmp.c_str(), path.c_str());      // atomic: concurrent ranks never read a half-written table
}

static size_t conv_apply(ConvParams& c, int tile, int split_target, int no_split)
{
    c.force_tile = tile; c.split_target = split_target; c.no_split = no_split;
    return conv_plan_split(c);
}

static int autotune_plan(accel_plan* p)
{
    const char* e = getenv("ACCEL_AUTOTUNE");
    if (e && e[0] == '0') return 0;
    hipStream_t st = p->m->ctx->stream;
    tune_cache_load();
    bool tuned_any = false;
    struct Cand { int tile, split_target, no_split; };
    // pass 1: workspace large enough for every candidate
    std::vector<std::vector<Cand>> cands(p->ops.size());
    size_t ws_need = p->ws_bytes;
    for (size_t i = 0; i < p->ops.size(); ++i) {
        Op& op = p->ops[i];
        if (op.kind != OP_CONV || op.conv.force_tile >= 0 || op.conv.narrow) continue;
        ConvParams c = op.conv;
        std::vector<Cand>& cs = cands[i];
        if (c.x_half || c.y_half || c.res_half) {
            // a layer with half views runs on the fp16 form of conv_b3d.hip: 82 / 83 / 84 / 85 / 88 / 89 by tile shape
            for (int t : {CONV_TILE_B3D, CONV_TILE_B3D + 1, CONV_TILE_B3D + 2, CONV_TILE_B3D + 3, CONV_TILE_B3D + 6, CONV_TILE_B3D + 7}) {
                if ((t == CONV_TILE_B3D || t == CONV_TILE_B3D + 1) && c.Cout_store <= 128) continue;      // 256-column tiles
                if ((t != CONV_TILE_B3D + 6) && c.Cout_store <= 64) continue;                              // 64 channels: the 128x64 tile only
                cs.push_back({t, 0, 0});
                ConvParams q = c;
                const size_t base = conv_apply(q, t, 0, 0);
                const int ks0 = q.ksplit;
                if (conv_apply(q, t, 1024, 0) && q.ksplit != ks0) cs.push_back({t, 1024, 0});
                if (base) cs.push_back({t, 0, 1});
            }
        }
        else if (c.f16 && c.Cout_store <= 32) { cs.push_back({3, 0, 0}); cs.push_back({3, 1024, 0}); }
        else if (c.Cout_store <= 32) {
            cs.push_back({4, 0, 0}); cs.push_back({4, 1024, 0}); cs.push_back({9, 0, 0}); cs.push_back({9, 1024, 0}); cs.push_back({3, 0, 0});
            if (c.wh2r && conv_halo_eligible(c) && !withheld("halo")) cs.push_back({CONV_TILE_HALO, 0, 1});      // the offset branches of res5 (conv_halo.hip)
        }
        else {
            // (a layer that runs in fp16-MFMA mode stays on the fp16 kernel: "operands of every convolution with Cin % 8 == 0 and
            // more than 4 output channels are rounded to half" is then the exact specification of the mode, which the oracle
            // restates -- oracle/graphs.py ROUND_F16 -- instead of depending on what the tuner happened to pick)
            if (c.wu && !c.f16) {
                cs.push_back({CONV_TILE_WINO, 0, 0});
                ConvParams q = c;
                const size_t base = conv_apply(q, CONV_TILE_WINO, 0, 0);
                const int ks0 = q.ksplit;
                if (conv_apply(q, CONV_TILE_WINO, 1024, 0) && q.ksplit != ks0) cs.push_back({CONV_TILE_WINO, 1024, 0});
                if (base) cs.push_back({CONV_TILE_WINO, 0, 1});
            }
            if (c.wub && !c.f16) {
                for (int wt : {CONV_TILE_WINO_B3, CONV_TILE_WINO_B3U, CONV_TILE_WINO_B3S}) {
                    cs.push_back({wt, 0, 0});
                    ConvParams q = c;
                    const size_t base = conv_apply(q, wt, 0, 0);
                    const int ks0 = q.ksplit;
                    if (conv_apply(q, wt, 1024, 0) && q.ksplit != ks0) cs.push_back({wt, 1024, 0});
                    if (base) cs.push_back({wt, 0, 1});
                }
            }
            if (c.wstem && !c.f16) cs.push_back({CONV_TILE_STEM, 0, 0});
            if (c.wstemb && !c.f16) cs.push_back({CONV_TILE_STEM_B3, 0, 0});
            if (c.wws && !c.f16) cs.push_back({CONV_TILE_WS, 0, 0});
            if (c.wh2r && !c.f16 && conv_halo_eligible(c) && !withheld("halo")) cs.push_back({CONV_TILE_HALO, 0, 1});
            const int nb3 = c.wb3 ? 5 : 0;
            static const int tiles[] = {0, 1, 2, 3, 5, 6, 7, 8, 10, 11, 12, 13, 31, 32, 33, 34, 35,
                                        CONV_TILE_B3, CONV_TILE_B3 + 1, CONV_TILE_B3 + 2, CONV_TILE_B3 + 3, CONV_TILE_B3 + 4, CONV_TILE_B3 + 5,
                                        CONV_TILE_B3R, CONV_TILE_B3R + 1, CONV_TILE_B3R + 3, CONV_TILE_B3R + 4, CONV_TILE_B3R + 5};
            const bool no_deep = withheld("deep");
            for (int t : tiles) {
                if (t >= CONV_TILE_B3 && t < CONV_TILE_B3R && !nb3) continue;
                if (t >= CONV_TILE_B3R && !c.wb3r) continue;
                if (no_deep && t >= 31 && t <= 35) continue;   // A/B switch: leave the deep-prefetch variants out
                if (c.K_pad % conv_tile_bk(t)) continue;      // BK-64 variants need K_pad % 64 == 0
                if (c.f16 && !(t <= 3 || t == 10 || (c.f16 == 1 && t >= CONV_TILE_B3R && c.wb3r))) continue;
                cs.push_back({t, 0, 0});
                ConvParams q = c;
                const size_t base = conv_apply(q, t, 0, 0);
                const int ks0 = q.ksplit;
                if (conv_apply(q, t, 1024, 0) && q.ksplit != ks0) cs.push_back({t, 1024, 0});
                if (base) cs.push_back({t, 0, 1});
            }
        }
        if (const char* fw = getenv("ACCEL_WB3_FORCE"); fw && c.wub && !c.f16) {
            // diagnostics: every layer that can take 41 / 42 / 43, does.  The geometry goes straight into the layer's parameters:
            // nothing is looked up in, inserted into or erased from the process-wide launch-geometry table (a forced run used to
            // replace shipped decisions for the rest of the process, so later tests of the same session ran on geometry 43)
            cs.clear();
            conv_apply(op.conv, atoi(fw), 0, 0);
            ConvParams q = op.conv;
            const size_t w = conv_plan_split(q);
            if (w > ws_need) ws_need = w;
            continue;
        }
        for (const Cand& k : cs) { ConvParams q = c; size_t w = conv_apply(q, k.tile, k.split_target, k.no_split); if (w > ws_need) ws_need = w; }
    }
    if (ws_need > p->ws_bytes) {
        float* nw = nullptr;
        HIP_TRY(hipMalloc((void**)&nw, ws_need));
        poison(nw, ws_need);
        p->owned.push_back(nw);
        p->ws = nw; p->ws_bytes = ws_need;
        for (Op& op : p->ops) if (op.kind == OP_CONV) op.conv.ws = p->ws;
    }
    struct TuneScratch {      // released on every way out of this function
        hipEvent_t e0 = nullptr, e1 = nullptr;
        void* scrub = nullptr;
        hipStream_t st;
        ~TuneScratch()
        {
            if (e0) hipEventDestroy(e0);
            if (e1) hipEventDestroy(e1);
            if (scrub) { hipStreamSynchronize(st); hipFree(scrub); }
        }
    } ts;
    ts.st = st;
    HIP_TRY(hipEventCreate(&ts.e0)); HIP_TRY(hipEventCreate(&ts.e1));
    hipEvent_t e0 = ts.e0, e1 = ts.e1;
    int rc = 0;
    const size_t scrub_bytes = (size_t)320 << 20;      // > L2 (8 x 4 MB) + Infinity Cache (256 MB)
    if (hipMalloc(&ts.scrub, scrub_bytes) != hipSuccess) ts.scrub = nullptr;
    void* const scrub = ts.scrub;
    for (size_t i = 0; i < p->ops.size() && !rc; ++i) {
        if (cands[i].empty()) continue;
        Op& op = p->ops[i];
        ConvParams& c = op.conv;
        TuneKey key; memset(&key, 0, sizeof key);
        int kk[16] = {c.H, c.W, c.Cin, c.xCs, c.Ho, c.Wo, c.kh * 16 + c.kw, c.sh * 16 + c.sw, c.dh * 16 + c.dw, c.K_pad,
                      c.Cout_store, c.yCs, c.res ? c.resCs : 0, c.y2 ? c.y2Cs : 0, c.act + 8 * c.deconv2x + 16 * (c.f16 == 1) + 256 * (c.f16 == 2) + 32 * (c.wu ? 1 : 0) + 64 * (c.wstem ? 1 : 0) + 128 * (c.wws ? 1 : 0) + 512 * (c.wb3 ? 1 : 0) + 1024 * (c.wub ? 1 : 0) + 2048 * c.x_half + 4096 * c.y_half + 8192 * c.res_half + 16384 * (c.wstemb ? 1 : 0) + 32768 * (c.wh2r ? 1 : 0), c.ph * 16 + c.pw + 65536 * (c.M / (c.Ho * c.Wo))};
        memcpy(key.v, kk, sizeof kk);
        auto it = g_tune_cache.find(key);
        if (it != g_tune_cache.end()) {
            // a table entry is only replayed if it names one of THIS layer's candidates (a table written by another build, or a
            // geometry the plan withholds from the layer, must not turn into an invalid launch): otherwise the layer is timed again
            bool known = false;
            for (const auto& k : cands[i]) known |= k.tile == it->second.tile;
            // (a shipped entry that is timed again leaves the shipped set: tune_cache_save skips shipped keys, and the new decision
            // must reach the user's table instead of being re-timed -- with another outcome, possibly -- in every process)
            if (!known) { g_tune_shipped.erase(it->first); g_tune_cache.erase(it); it = g_tune_cache.end(); }
        }
        if (it == g_tune_cache.end()) {
            // Each candidate is timed the way the launch will run inside the plan: weights cold (L2 and the 256 MB
            // Infinity Cache scrubbed by a memset of a larger scratch buffer), input freshly produced by the
            // preceding op.  Timing a hot loop of the same launch instead rewards tiles that only win with resident
            // weights and loses ~2 % on the whole plan.
            float best = 1e30f; TuneVal bv = {c.force_tile, 0, 0};
            if (c.xr_slot) {      // the fp16x2 candidates read the range of the input as it lies there now
                HIP_TRY(launch_range_clear(const_cast<unsigned*>(c.xr_slot), 1, st));
                const bool m0 = op.measure;
                op.measure = true;
                rc = launch_measure(p, op);
                op.measure = m0;
                if (rc) break;
            }
            for (const Cand& k : cands[i]) {
                ConvParams q = c;
                conv_apply(q, k.tile, k.split_target, k.no_split);
                hipError_t he = launch_conv_igemm(q, st);        // first launch of this variant: attribute set-up, code load
                float ms_min = 1e30f;
                for (int r = 0; r < 3 && he == hipSuccess; ++r) {
                    if (scrub) HIP_TRY(hipMemsetAsync(scrub, r, scrub_bytes, st));
                    if (i > 0 && (rc = launch_op(p, p->ops[i - 1]))) break;
                    HIP_TRY(hipEventRecord(e0, st));
                    he = launch_conv_igemm(q, st);
                    HIP_TRY(hipEventRecord(e1, st));
                    HIP_TRY(hipEventSynchronize(e1));
                    float ms = 0.f;
                    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < ms_min) ms_min = ms;
                }
                if (rc) break;
                if (he != hipSuccess) { rc = fail(ACCEL_ERR_HIP, "autotune launch failed: %s", hipGetErrorString(he)); break; }
                if (ms_min < best) { best = ms_min; bv = {k.tile, k.split_target, k.no_split}; }
            }
            if (rc) break;
            it = g_tune_cache.insert({key, bv}).first;
            tuned_any = true;
            ++g_tune_timed;
            if (e && !strcmp(e, "verbose"))
                fprintf(stderr, "[accel tune] %s: timed (not in a table): Cin %d Cout_store %d k %dx%d M %d -> geometry %d\n", op.name.c_str(),
                        c.Cin, c.Cout_store, c.kh, c.kw, c.M, bv.tile);
        } else {
            ++g_tune_hits;
        }
        conv_apply(c, it->second.tile, it->second.split_target, it->second.no_split);
    }
    if (tuned_any && !rc) tune_cache_save();
    return rc;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int accel_tune_stats(int* replayed, int* timed, int* shipped_entries)
{
    tune_cache_load();
    if (replayed) *replayed = g_tune_hits;
    if (timed) *timed = g_tune_timed;
    if (shipped_entries) *shipped_entries = (int)g_tune_shipped.size();
    return 0;
}

extern "C" int accel_ctx_create(int device_id, accel_ctx** out)
{
    if (!out) return fail(ACCEL_ERR_ARG, "accel_ctx_create: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(ACCEL_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(ACCEL_ERR_ARG, "device %d out of range (%d devices)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    accel_ctx* c = new accel_ctx();
    c->device = device_id;
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se == hipSuccess) se = hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return fail(ACCEL_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(se)); }
    *out = c;
    return 0;
}

extern "C" int accel_ctx_destroy(accel_ctx* ctx)
{
    if (!ctx) return 0;
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->copy);
    hipStreamDestroy(ctx->stream);
    hipStreamDestroy(ctx->copy);
    delete ctx;
    return 0;
}

extern "C" int accel_sync(accel_ctx* ctx)
{
    if (!ctx) return fail(ACCEL_ERR_ARG, "accel_sync: ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" void* accel_ctx_stream(accel_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int accel_model_create(accel_ctx* ctx, accel_model** out)
{
    if (!ctx || !out) return fail(ACCEL_ERR_ARG, "accel_model_create: NULL argument");
    accel_model* m = new accel_model();
    m->ctx = ctx;
    *out = m;
    return 0;
}

static void plan_free(accel_plan* p)
{
    if (p->gexec) hipGraphExecDestroy(p->gexec);
    if (p->graph) hipGraphDestroy(p->graph);
    for (void* d : p->owned) hipFree(d);
    if (p->arena) hipFree(p->arena);
    if (p->range_flag) hipHostFree(p->range_flag);
    delete p;
}

extern "C" int accel_model_destroy(accel_model* m)
{
    if (!m) return 0;
    hipStreamSynchronize(m->ctx->stream);
    for (accel_plan* p : m->plans) plan_free(p);
    for (auto& kv : m->pbufs) { hipFree(kv.second.ptr); if (kv.second.slot) hipFree(const_cast<void**>(kv.second.slot)); }
    for (auto& kv : m->shadows) { if (kv.second.ready) hipEventDestroy(kv.second.ready); if (kv.second.consumed) hipEventDestroy(kv.second.consumed); hipFree(kv.second.ptr); }
    delete m;
    return 0;
}

extern "C" int accel_model_set_param(accel_model* m, const char* name, const float* data, int ndim, const int64_t* shape)
{
    if (!m || !name || !data || ndim < 0 || ndim > 8 || (ndim && !shape)) return fail(ACCEL_ERR_ARG, "accel_model_set_param: bad argument");
    HostParam hp;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { hp.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    hp.data.assign(data, data + n);
    m->params[name] = std::move(hp);
    return 0;
}

extern "C" int accel_model_has_param(accel_model* m, const char* name)
{
    return m && name && m->params.count(name) ? 1 : 0;
}

extern "C" int accel_model_add_plan(accel_model* m, const char* role, const char* plan_text, accel_plan** out)
{
    if (!m || !plan_text || !out) return fail(ACCEL_ERR_ARG, "accel_model_add_plan: NULL argument");
    HIP_TRY(hipSetDevice(m->ctx->device));
    accel_plan* p = new accel_plan();
    p->m = m;
    p->role = role ? role : "";
    int rc = parse_plan(p, plan_text);
    if (rc) { delete p; return rc; }
    m->plans.push_back(p);
    if (role && *role) m->roles[role] = p;
    *out = p;
    return 0;
}

extern "C" int accel_plan_finalize(accel_plan* p)
{
    if (!p) return fail(ACCEL_ERR_ARG, "accel_plan_finalize: NULL plan");
    if (p->finalized) return 0;
    HIP_TRY(hipSetDevice(p->m->ctx->device));
    if (p->arena_bytes) {
        HIP_TRY(hipMalloc((void**)&p->arena, p->arena_bytes));
        poison(p->arena, p->arena_bytes);
        HIP_TRY(hipMemsetAsync(p->arena, 0, p->arena_bytes, p->m->ctx->stream));
    }
    for (Op& op : p->ops) {
        int rc = finalize_op(p, op);
        if (rc) return rc;
    }
    if (int rc = assign_range_slots(p)) return rc;
    resolve_range_flags(p, conv_has_h2_form);      // for the tuning launches: every layer that could run the form counts
    if (p->ws_bytes) {
        HIP_TRY(hipMalloc((void**)&p->ws, p->ws_bytes));
        poison(p->ws, p->ws_bytes);
        p->owned.push_back(p->ws);
        for (Op& op : p->ops) if (op.kind == OP_CONV) op.conv.ws = p->ws;
    }
    HIP_TRY(hipDeviceSynchronize());
    const char* g = getenv("ACCEL_HIP_GRAPH");
    const bool use_graph = p->allow_graph && !(g && g[0] == '0');
    // Autotuning and the warm-up run below EXECUTE ops of this plan on the real buffers.  Persistent buffers the plan
    // writes are live model state (a non-key plan warps `feat` / `featG` in place; a lazily bound plan is finalized
    // between two frames of a clip), so they are saved first and put back afterwards: finalizing a plan never changes
    // what the next forward sees.  Derived-buffer validity is untouched for the same reason.
    // (RAII: whatever way this function is left -- including every HIP_TRY early return below -- the snapshot is copied
    // back and freed.)
    struct Snapshot {
        accel_plan* p;
        struct Saved { std::string name; void* copy; size_t bytes; };
        std::vector<Saved> saved;
        int restore()
        {
            hipStream_t st = p->m->ctx->stream;
            int rc = 0;
            for (Saved& sv : saved)
                if (!rc && hipMemcpyAsync(p->m->pbufs[sv.name].ptr, sv.copy, sv.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    rc = fail(ACCEL_ERR_HIP, "restoring persistent buffer %s after plan finalisation failed", sv.name.c_str());
            if (!saved.empty() && hipStreamSynchronize(st) != hipSuccess && !rc) rc = fail(ACCEL_ERR_HIP, "sync after plan finalisation failed");
            for (Saved& sv : saved) hipFree(sv.copy);
            saved.clear();
            return rc;
        }
        ~Snapshot()
        {
            if (saved.empty()) return;
            const std::string keep = g_err;      // an error path: the message of the original failure stays
            restore();
            g_err = keep;
        }
    } snap{p, {}};
    if (p->allow_tune || use_graph) {
        for (const std::string& name : p->pbuf_writes) {
            DevBuf& b = p->m->pbufs[name];
            Snapshot::Saved sv{name, nullptr, b.bytes};
            if (hipMalloc(&sv.copy, b.bytes) != hipSuccess) return fail(ACCEL_ERR_HIP, "hipMalloc(%zu) for the snapshot of %s failed", b.bytes, name.c_str());
            snap.saved.push_back(sv);
            if (hipMemcpyAsync(sv.copy, b.ptr, b.bytes, hipMemcpyDeviceToDevice, p->m->ctx->stream) != hipSuccess) return fail(ACCEL_ERR_HIP, "snapshot of %s failed", name.c_str());
        }
    }
    if (p->allow_tune) { int trc = autotune_plan(p); if (trc) return trc; HIP_TRY(hipDeviceSynchronize()); }
    resolve_range_flags(p, conv_runs_h2);          // the geometries are final: only layers that run the fp16x2 form read a range
    if (use_graph) {
        hipStream_t st = p->m->ctx->stream;
        // one eager warm-up run: lazy module loading / function attributes must not happen under capture
        int rc = run_eager(p);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        rc = run_eager(p);
        hipError_t e = hipStreamEndCapture(st, &p->graph);
        if (rc) return rc;
        if (e != hipSuccess) return fail(ACCEL_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        HIP_TRY(hipGraphInstantiate(&p->gexec, p->graph, nullptr, nullptr, 0));
    }
    int rrc = snap.restore();
    if (rrc) return rrc;
    // the tuning launches and the warm-up run worked on whatever the buffers held (tuning: an op's candidates on the output of ONE
    // launch of its predecessor): a non-finite range seen there says nothing about a frame
    HIP_TRY(hipStreamSynchronize(p->m->ctx->stream));
    if (p->range_flag) p->range_flag[0] = p->range_flag[1] = 0u;
    p->finalized = true;
    return 0;
}

extern "C" int accel_plan_run(accel_plan* p)
{
    if (!p || !p->finalized) return fail(ACCEL_ERR_ARG, "accel_plan_run: plan not finalized");
    accel_model* m = p->m;
    // stale derived inputs are rebuilt from their source by the model's `init:<name>` plan (same stream: ordered)
    for (const auto& r : p->pbuf_reads) {
        auto d = m->derived_valid.find(r);
        if (d == m->derived_valid.end() || d->second) continue;
        auto ip = m->roles.find("init:" + r);
        if (ip == m->roles.end() || ip->second == p || !ip->second->finalized)
            return fail(ACCEL_ERR_PLAN, "persistent buffer '%s' is stale and the model has no finalized 'init:%s' plan", r.c_str(), r.c_str());
        int rc = accel_plan_run(ip->second);
        if (rc) return rc;
    }
    int rc = range_check(p);
    if (rc) return rc;
    if (p->gexec) {
        if (hipGraphLaunch(p->gexec, m->ctx->stream) != hipSuccess) return fail(ACCEL_ERR_HIP, "hipGraphLaunch failed");
    } else {
        rc = run_eager(p);
    }
    for (const auto& w : p->pbuf_writes) m->source_written(w);
    for (const auto& w : p->pbuf_writes)
        if (m->derived_valid.count(w)) m->derived_valid[w] = true;
    return rc;
}

extern "C" int accel_plan_num_ops(accel_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int accel_plan_op_info(accel_plan* p, int i, char* kind32, char* name64, double* flops, double* bytes)
{
    if (!p || i < 0 || i >= (int)p->ops.size()) return fail(ACCEL_ERR_ARG, "accel_plan_op_info: index out of range");
    const Op& op = p->ops[i];
    if (kind32) { strncpy(kind32, op.kind_name.c_str(), 31); kind32[31] = 0; }
    if (name64) { strncpy(name64, op.name.c_str(), 63); name64[63] = 0; }
    if (flops) *flops = op.flops;
    if (bytes) *bytes = op.bytes;
    return 0;
}

extern "C" int accel_plan_op_launch(accel_plan* p, int i, int* tile, int* ksplit, int* narrow)
{
    if (!p || i < 0 || i >= (int)p->ops.size()) return fail(ACCEL_ERR_ARG, "accel_plan_op_launch: index out of range");
    const Op& op = p->ops[i];
    const bool conv = op.kind == OP_CONV;
    if (tile) *tile = conv ? (op.conv.force_tile >= 0 ? op.conv.force_tile : conv_pick_tile(op.conv)) : -1;
    if (ksplit) *ksplit = conv ? op.conv.ksplit : 0;
    if (narrow) *narrow = conv ? op.conv.narrow : 0;
    return 0;
}

extern "C" int accel_plan_op_mode(accel_plan* p, int i, int* mode)
{
    if (!p || i < 0 || i >= (int)p->ops.size()) return fail(ACCEL_ERR_ARG, "accel_plan_op_mode: index out of range");
    const Op& op = p->ops[i];
    // 0: fp32 layer (bf16x3 form, six products on the bf16 pipe, where it runs on a matrix-core geometry), 1: f16-mode layer, 3: fp32 layer
    // whose matrix-core geometries run the fp16x2 form (three products on the fp16 pipe)
    if (mode) *mode = op.kind == OP_CONV ? (op.conv.f16 ? op.conv.f16 : (op.conv.wh2r ? 3 : 0)) : -1;
    return 0;
}

extern "C" int accel_plan_op_range(accel_plan* p, int i, float* scale, int* measured)
{
    if (!p || !p->finalized || i < 0 || i >= (int)p->ops.size()) return fail(ACCEL_ERR_ARG, "accel_plan_op_range: index out of range");
    const Op& op = p->ops[i];
    if (scale) *scale = 0.f;
    if (measured) *measured = 0;
    if (op.kind != OP_CONV || !op.conv.xr_slot) return 0;
    unsigned bits = 0u;
    HIP_TRY(hipStreamSynchronize(p->m->ctx->stream));
    HIP_TRY(hipMemcpy(&bits, op.conv.xr_slot, sizeof bits, hipMemcpyDeviceToHost));      // word 0: what the convolution read
    if (scale) *scale = range_scale(bits).s;
    if (measured) *measured = bits ? (op.measure ? 2 : 1) : 0;
    return 0;
}

// diagnostics (scripts/debug/range_nan.py; declared in include/accel_hip.h): the RANGE_WORDS words of conv op i's input slot (word 0 = what the convolution read, words RANGE_PART_OFF.. = the partial words)
extern "C" int accel_plan_op_range_words(accel_plan* p, int i, unsigned* words, int n_words)
{
    if (!p || !p->finalized || i < 0 || i >= (int)p->ops.size() || !words || n_words < RANGE_WORDS) return fail(ACCEL_ERR_ARG, "accel_plan_op_range_words: bad argument");
    const Op& op = p->ops[i];
    if (op.kind != OP_CONV || !op.conv.xr_slot) return fail(ACCEL_ERR_ARG, "accel_plan_op_range_words: op %d has no range slot", i);
    HIP_TRY(hipStreamSynchronize(p->m->ctx->stream));
    HIP_TRY(hipMemcpy(words, op.conv.xr_slot, RANGE_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int accel_plan_profile(accel_plan* p, int iters, float* ms, int n_ms)
{
    if (!p || !p->finalized || !ms || n_ms < (int)p->ops.size() || iters < 1) return fail(ACCEL_ERR_ARG, "accel_plan_profile: bad argument");
    hipStream_t st = p->m->ctx->stream;
    const size_t n = p->ops.size();
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    int rc = 0;
    for (int it = 0; it < iters && !rc; ++it) {
        if (p->n_slots) HIP_TRY(launch_range_clear(p->range, p->n_slots, st));
        for (size_t i = 0; i < n && !rc; ++i) {
            HIP_TRY(hipEventRecord(ev[2 * i], st));
            if (p->ops[i].measure || p->ops[i].fold) rc = launch_measure(p, p->ops[i]);      // the range pass / fold belongs to the op that needs it
            if (!rc) rc = launch_op(p, p->ops[i]);
            HIP_TRY(hipEventRecord(ev[2 * i + 1], st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t i = 0; i < n && !rc; ++i) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
            acc[i] += t;
        }
    }
    for (auto& e : ev) hipEventDestroy(e);
    for (size_t i = 0; i < n; ++i) ms[i] = (float)(acc[i] / iters);
    return rc;
}

// ---- diagnostics: the plan's ops one after the other, eagerly (no graph replay), then a host wait; and the arena -------------
extern "C" int accel_plan_run_serial(accel_plan* p)
{
    if (!p || !p->finalized) return fail(ACCEL_ERR_ARG, "accel_plan_run_serial: plan not finalized");
    int rc = range_check(p);
    if (rc) return rc;
    if ((rc = run_eager(p))) return rc;
    HIP_TRY(hipStreamSynchronize(p->m->ctx->stream));
    return range_check(p);
}

extern "C" int accel_plan_arena_read(accel_plan* p, size_t offset, void* host_dst, size_t bytes, size_t* arena_bytes)
{
    if (!p || !p->finalized) return fail(ACCEL_ERR_ARG, "accel_plan_arena_read: plan not finalized");
    if (arena_bytes) *arena_bytes = p->arena_bytes;
    if (!host_dst || !bytes) return 0;
    if (offset + bytes > p->arena_bytes) return fail(ACCEL_ERR_ARG, "accel_plan_arena_read: %zu + %zu bytes > arena (%zu)", offset, bytes, p->arena_bytes);
    HIP_TRY(hipStreamSynchronize(p->m->ctx->stream));
    HIP_TRY(hipMemcpy(host_dst, p->arena + offset, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// a write into the model's own buffer ends a binding made by accel_model_bind_device
static int unbind(accel_model* m, DevBuf& b)
{
    if (!b.bound) return 0;
    HIP_TRY(launch_set_slot(b.slot, b.ptr, m->ctx->stream));
    b.bound = nullptr;
    return 0;
}

extern "C" int accel_model_bind_device(accel_model* m, const char* buf, const void* devptr, size_t bytes)
{
    if (!m || !buf || !devptr) return fail(ACCEL_ERR_ARG, "accel_model_bind_device: NULL argument");
    auto it = m->pbufs.find(buf);
    if (it == m->pbufs.end()) return fail(ACCEL_ERR_ARG, "accel_model_bind_device: unknown buffer '%s'", buf);
    DevBuf& b = it->second;
    if (!b.slot || b.readers != b.prep_readers)
        return fail(ACCEL_ERR_ARG, "accel_model_bind_device: '%s' is not an image input of the finalized plans (only buffers that prep_rgb / "
                                   "prep_flow alone read can be bound)", buf);
    if (bytes != b.bytes) return fail(ACCEL_ERR_ARG, "accel_model_bind_device: %zu bytes, buffer '%s' has %zu", bytes, buf, b.bytes);
    if (b.bound != devptr) HIP_TRY(launch_set_slot(b.slot, devptr, m->ctx->stream));
    b.bound = devptr == b.ptr ? nullptr : devptr;
    m->source_written(buf);
    return 0;
}

extern "C" int accel_model_write(accel_model* m, const char* buf, const void* src, size_t bytes, int src_on_device)
{
    if (!m || !buf || !src) return fail(ACCEL_ERR_ARG, "accel_model_write: NULL argument");
    auto it = m->pbufs.find(buf);
    if (it == m->pbufs.end()) return fail(ACCEL_ERR_ARG, "accel_model_write: unknown buffer '%s'", buf);
    if (bytes > it->second.bytes) return fail(ACCEL_ERR_ARG, "accel_model_write: %zu bytes > buffer '%s' (%zu)", bytes, buf, it->second.bytes);
    if (int rc = unbind(m, it->second)) return rc;
    if (src_on_device) HIP_TRY(launch_copy_bytes(src, it->second.ptr, bytes, m->ctx->stream));      // own copy kernel: 4x the rate of the runtime's blit
    else HIP_TRY(hipMemcpyAsync(it->second.ptr, src, bytes, hipMemcpyHostToDevice, m->ctx->stream));
    if (!src_on_device) HIP_TRY(hipStreamSynchronize(m->ctx->stream));   // pageable source may be reused by the caller
    m->source_written(buf);
    return 0;
}

extern "C" int accel_model_read(accel_model* m, const char* buf, void* dst, size_t bytes, int dst_on_device)
{
    if (!m || !buf || !dst) return fail(ACCEL_ERR_ARG, "accel_model_read: NULL argument");
    auto it = m->pbufs.find(buf);
    if (it == m->pbufs.end()) return fail(ACCEL_ERR_ARG, "accel_model_read: unknown buffer '%s'", buf);
    if (bytes > it->second.bytes) return fail(ACCEL_ERR_ARG, "accel_model_read: %zu bytes > buffer '%s' (%zu)", bytes, buf, it->second.bytes);
    // a bound input is read where the plans read it (the caller's frame), not the model-owned copy an earlier write left behind
    const void* src = it->second.bound ? it->second.bound : it->second.ptr;
    if (dst_on_device) HIP_TRY(launch_copy_bytes(src, dst, bytes, m->ctx->stream));
    else HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, m->ctx->stream));
    if (!dst_on_device) HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    return 0;
}

extern "C" int accel_model_buffer(accel_model* m, const char* buf, void** dev_ptr, size_t* bytes)
{
    if (!m || !buf) return fail(ACCEL_ERR_ARG, "accel_model_buffer: NULL argument");
    auto it = m->pbufs.find(buf);
    if (it == m->pbufs.end()) return fail(ACCEL_ERR_ARG, "accel_model_buffer: unknown buffer '%s'", buf);
    // the caller is about to write the model-owned buffer through the raw pointer: a binding made by accel_model_bind_device
    // ends here (the plans read the model's copy again), otherwise that write would be silently ignored
    if (dev_ptr) if (int rc = unbind(m, it->second)) return rc;
    if (dev_ptr) *dev_ptr = it->second.ptr;
    if (bytes) *bytes = it->second.bytes;
    if (dev_ptr) m->source_written(buf);   // the caller may write through the raw pointer: derived buffers become stale
    return 0;
}

extern "C" int accel_model_buffer_generation(accel_model* m, const char* buf, uint64_t* generation)
{
    if (!m || !buf || !generation) return fail(ACCEL_ERR_ARG, "accel_model_buffer_generation: NULL argument");
    if (!m->pbufs.count(buf)) return fail(ACCEL_ERR_ARG, "accel_model_buffer_generation: unknown buffer '%s'", buf);
    auto it = m->generation.find(buf);
    *generation = it == m->generation.end() ? 0 : it->second;
    return 0;
}

extern "C" int accel_host_alloc(size_t bytes, void** out)
{
    if (!out || !bytes) return fail(ACCEL_ERR_ARG, "accel_host_alloc: bad argument");
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return 0;
}

extern "C" int accel_host_free(void* p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return 0;
}

extern "C" int accel_model_prefetch(accel_model* m, const char* buf, const void* pinned_src, size_t bytes)
{
    if (!m || !buf || !pinned_src) return fail(ACCEL_ERR_ARG, "accel_model_prefetch: NULL argument");
    auto it = m->pbufs.find(buf);
    if (it == m->pbufs.end()) return fail(ACCEL_ERR_ARG, "accel_model_prefetch: unknown buffer '%s'", buf);
    if (bytes > it->second.bytes) return fail(ACCEL_ERR_ARG, "accel_model_prefetch: %zu bytes > buffer '%s' (%zu)", bytes, buf, it->second.bytes);
    HIP_TRY(hipSetDevice(m->ctx->device));
    accel_model::Shadow& sh = m->shadows[buf];
    if (!sh.ptr) {
        HIP_TRY(hipMalloc(&sh.ptr, it->second.bytes));
        sh.bytes = it->second.bytes;
        HIP_TRY(hipEventCreateWithFlags(&sh.ready, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sh.consumed, hipEventDisableTiming));
    } else if (sh.was_consumed) {
        // the previous commit copied OUT of the shadow on the compute stream (event recorded right behind that copy,
        // ahead of the plan that followed): do not overwrite under it -- and wait for nothing else on that stream
        HIP_TRY(hipStreamWaitEvent(m->ctx->copy, sh.consumed, 0));
    }
    HIP_TRY(hipMemcpyAsync(sh.ptr, pinned_src, bytes, hipMemcpyHostToDevice, m->ctx->copy));
    HIP_TRY(hipEventRecord(sh.ready, m->ctx->copy));
    sh.filled = bytes;
    return 0;
}

extern "C" int accel_model_commit(accel_model* m, const char* buf)
{
    if (!m || !buf) return fail(ACCEL_ERR_ARG, "accel_model_commit: NULL argument");
    auto sh = m->shadows.find(buf);
    if (sh == m->shadows.end() || !sh->second.filled) return fail(ACCEL_ERR_ARG, "accel_model_commit: nothing was prefetched for '%s'", buf);
    HIP_TRY(hipStreamWaitEvent(m->ctx->stream, sh->second.ready, 0));
    if (int rc = unbind(m, m->pbufs[buf])) return rc;
    HIP_TRY(launch_copy_bytes(sh->second.ptr, m->pbufs[buf].ptr, sh->second.filled, m->ctx->stream));
    HIP_TRY(hipEventRecord(sh->second.consumed, m->ctx->stream));
    sh->second.was_consumed = true;
    sh->second.filled = 0;
    m->source_written(buf);
    return 0;
}

extern "C" int accel_model_read_async(accel_model* m, const char* buf, void* pinned_dst, size_t bytes)
{
    if (!m || !buf || !pinned_dst) return fail(ACCEL_ERR_ARG, "accel_model_read_async: NULL argument");
    auto it = m->pbufs.find(buf);
    if (it == m->pbufs.end()) return fail(ACCEL_ERR_ARG, "accel_model_read_async: unknown buffer '%s'", buf);
    if (bytes > it->second.bytes) return fail(ACCEL_ERR_ARG, "accel_model_read_async: %zu bytes > buffer '%s' (%zu)", bytes, buf, it->second.bytes);
    HIP_TRY(hipMemcpyAsync(pinned_dst, it->second.ptr, bytes, hipMemcpyDeviceToHost, m->ctx->stream));
    return 0;
}

static size_t pbuf_bytes(accel_model* m, const char* name)
{
    auto it = m->pbufs.find(name);
    return it == m->pbufs.end() ? 0 : it->second.bytes;
}

static int frame_outputs(accel_model* m, float* feat_out, float* logits_out, uint8_t* labels_out, int on_dev)
{
    int rc;
    if (feat_out) {
        // `feat` lives NHWC in HBM; the boundary layout is NCHW (res5c_relu_output / warping_feat_output)
        auto src = m->pbufs.find(m->feat_slot ? "feat_b" : "feat");
        if (src == m->pbufs.end() || !m->feat_c) return fail(ACCEL_ERR_ARG, "feat_out requested but the model has no propagated feature");
        const size_t img = (size_t)m->feat_c * m->feat_h * m->feat_w;
        const size_t bytes = img * m->feat_n * sizeof(float);
        DevBuf& tmp = m->pbufs["feat_nchw"];
        if (!tmp.ptr || tmp.bytes < bytes) {
            if (tmp.ptr) hipFree(tmp.ptr);
            if (hipMalloc(&tmp.ptr, bytes) != hipSuccess) return fail(ACCEL_ERR_HIP, "hipMalloc(feat_nchw) failed");
            tmp.bytes = bytes;
        }
        for (int n = 0; n < m->feat_n; ++n) {
            hipError_t e = launch_nhwc_to_nchw(static_cast<const float*>(src->second.ptr) + n * img, m->feat_c,
                                               static_cast<float*>(tmp.ptr) + n * img, m->feat_c, m->feat_h, m->feat_w, m->ctx->stream);
            if (e != hipSuccess) return fail(ACCEL_ERR_HIP, "feature export failed: %s", hipGetErrorString(e));
        }
        if ((rc = accel_model_read(m, "feat_nchw", feat_out, bytes, on_dev))) return rc;
    }
    if (logits_out && (rc = accel_model_read(m, "logits", logits_out, pbuf_bytes(m, "logits"), on_dev))) return rc;
    if (labels_out && (rc = accel_model_read(m, "labels", labels_out, pbuf_bytes(m, "labels"), on_dev))) return rc;
    return 0;
}

extern "C" int accel_key_forward(accel_model* m, const float* img, int img_on_device,
                                 float* feat_out, float* logits_out, uint8_t* labels_out, int out_on_device)
{
    if (!m) return fail(ACCEL_ERR_ARG, "accel_key_forward: NULL model");
    auto it = m->roles.find("key");
    if (it == m->roles.end()) return fail(ACCEL_ERR_ARG, "accel_key_forward: model has no 'key' plan");
    int rc;
    if (img && (rc = accel_model_write(m, "data", img, pbuf_bytes(m, "data"), img_on_device))) return rc;
    if ((rc = accel_plan_run(it->second))) return rc;
    return frame_outputs(m, feat_out, logits_out, labels_out, out_on_device);
}

extern "C" int accel_cur_forward(accel_model* m, const float* img_cur, const float* img_prev, int img_on_device,
                                 float* feat_out, float* logits_out, uint8_t* labels_out, int out_on_device)
{
    if (!m) return fail(ACCEL_ERR_ARG, "accel_cur_forward: NULL model");
    auto it = m->roles.find("cur");
    if (it == m->roles.end()) return fail(ACCEL_ERR_ARG, "accel_cur_forward: model has no 'cur' plan");
    if (m->feat_slot == 1) {      // the current feature sits in `feat_b`: the plan that reads it
        it = m->roles.find("cur_b");
        if (it == m->roles.end()) return fail(ACCEL_ERR_ARG, "accel_cur_forward: the feature is in 'feat_b' but the model has no 'cur_b' plan");
    }
    int rc;
    if (img_cur && (rc = accel_model_write(m, "data", img_cur, pbuf_bytes(m, "data"), img_on_device))) return rc;
    if (img_prev && (rc = accel_model_write(m, "data_key", img_prev, pbuf_bytes(m, "data_key"), img_on_device))) return rc;
    if ((rc = accel_plan_run(it->second))) return rc;
    return frame_outputs(m, feat_out, logits_out, labels_out, out_on_device);
}

// ---------------------------------------------------------------------------
// operator-level entry points: one-op plans over temporary models
// ---------------------------------------------------------------------------
namespace {
struct TempModel {
    accel_model* m = nullptr;
    ~TempModel() { if (m) accel_model_destroy(m); }
};

std::string bref(const char* space, size_t off, int C, int Cs, int H, int W, int N = 1)
{
    char b[176];
    if (N == 1) snprintf(b, sizeof b, "%s:%zu:%d:%d:%d:%d", space, off, C, Cs, H, W);
    else snprintf(b, sizeof b, "%s:%zu:%d:%d:%d:%d:%d", space, off, C, Cs, H, W, N);
    return b;
}

int set1(accel_model* m, const char* name, const float* d, std::initializer_list<int64_t> shp)
{
    std::vector<int64_t> s(shp);
    return accel_model_set_param(m, name, d, (int)s.size(), s.data());
}

int run_text(accel_model* m, const std::string& text)
{
    accel_plan* p = nullptr;
    int rc = accel_model_add_plan(m, "op", text.c_str(), &p);
    if (rc) return rc;
    if ((rc = accel_plan_finalize(p))) return rc;
    if ((rc = accel_plan_run(p))) return rc;
    return accel_sync(m->ctx);
}
}  // namespace

static size_t al(size_t b) { return (b + 255) / 256 * 256; }

extern "C" int accel_conv2d(accel_ctx* ctx, const float* x, int N, int C, int H, int W,
                            const float* w, const float* bias, int K, int kh, int kw,
                            int sh, int sw, int ph, int pw, int dh, int dw,
                            const float* scale, const float* shift, const float* residual,
                            int act, float slope, int force_tile, float* y)
{
    if (!ctx || !x || !w || !y) return fail(ACCEL_ERR_ARG, "accel_conv2d: NULL argument");
    if (N < 1) return fail(ACCEL_ERR_ARG, "accel_conv2d: N must be >= 1");
    const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
    if (Ho <= 0 || Wo <= 0) return fail(ACCEL_ERR_ARG, "accel_conv2d: empty output");
    TempModel t;
    int rc;
    if ((rc = accel_model_create(ctx, &t.m))) return rc;
    if ((rc = set1(t.m, "w_weight", w, {K, C, kh, kw}))) return rc;
    if (bias && (rc = set1(t.m, "w_bias", bias, {K}))) return rc;
    if (scale) {   // expressed as a BatchNorm with gamma=scale, beta=shift, mean=0, var=1-eps
        std::vector<float> zero(K, 0.f), var(K, 1.0f);
        if ((rc = set1(t.m, "e_gamma", scale, {K})) || (rc = set1(t.m, "e_beta", shift, {K})) ||
            (rc = set1(t.m, "e_moving_mean", zero.data(), {K})) || (rc = set1(t.m, "e_moving_var", var.data(), {K}))) return rc;
    }
    const int Cp = (C + 3) / 4 * 4, Kp = (K + 3) / 4 * 4;
    const size_t xin = (size_t)N * C * H * W * 4, yout = (size_t)N * K * Ho * Wo * 4;
    size_t off = 0;
    const size_t o_x = off; off += al((size_t)N * H * W * Cp * 4);
    const size_t o_y = off; off += al((size_t)N * Ho * Wo * Kp * 4);
    const size_t o_r = off; if (residual) off += al((size_t)N * Ho * Wo * Kp * 4);
    std::stringstream s;
    s << "option graph=0 tune=0\narena bytes=" << off << "\n";
    s << "pbuf name=x bytes=" << xin << "\npbuf name=y bytes=" << yout << "\n";
    if (residual) s << "pbuf name=r bytes=" << yout << "\n";
    s << "import_nchw src=" << bref("x", 0, C, C, H, W, N) << " dst=" << bref("A", o_x, C, Cp, H, W, N) << "\n";
    if (residual) s << "import_nchw src=" << bref("r", 0, K, K, Ho, Wo, N) << " dst=" << bref("A", o_r, K, Kp, Ho, Wo, N) << "\n";
    s << "conv name=op in=" << bref("A", o_x, C, Cp, H, W, N) << " out=" << bref("A", o_y, K, Kp, Ho, Wo, N)
      << " w=w_weight" << (bias ? " bias=w_bias" : "") << (scale ? " bn=e eps=0 fixg=0" : "")
      << " act=" << act << " slope=" << slope << " k=" << kh << "," << kw << " s=" << sh << "," << sw
      << " p=" << ph << "," << pw << " d=" << dh << "," << dw << " cin=" << C << " cout=" << K
      << " tile=" << force_tile;
    if (residual) s << " res=" << bref("A", o_r, K, Kp, Ho, Wo, N);
    s << "\nexport_nchw src=" << bref("A", o_y, K, Kp, Ho, Wo, N) << " dst=" << bref("y", 0, K, K, Ho, Wo, N) << "\n";
    accel_plan* p = nullptr;
    if ((rc = accel_model_add_plan(t.m, "op", s.str().c_str(), &p))) return rc;
    if ((rc = accel_model_write(t.m, "x", x, xin, 0))) return rc;
    if (residual && (rc = accel_model_write(t.m, "r", residual, yout, 0))) return rc;
    if ((rc = accel_plan_finalize(p)) || (rc = accel_plan_run(p))) return rc;
    return accel_model_read(t.m, "y", y, yout, 0);
}

extern "C" int accel_deconv2d_4x4s2(accel_ctx* ctx, const float* x, int N, int C, int H, int W,
                                    const float* w, const float* bias, int K, int act, float slope, float* y)
{
    if (!ctx || !x || !w || !y) return fail(ACCEL_ERR_ARG, "accel_deconv2d_4x4s2: NULL argument");
    if (N < 1) return fail(ACCEL_ERR_ARG, "accel_deconv2d_4x4s2: N must be >= 1");
    TempModel t;
    int rc;
    if ((rc = accel_model_create(ctx, &t.m))) return rc;
    if ((rc = set1(t.m, "w_weight", w, {C, K, 4, 4}))) return rc;
    if (bias && (rc = set1(t.m, "w_bias", bias, {K}))) return rc;
    const int Cp = (C + 3) / 4 * 4, Kp = (K + 3) / 4 * 4, Ho = 2 * H, Wo = 2 * W;
    const size_t xin = (size_t)N * C * H * W * 4, yout = (size_t)N * K * Ho * Wo * 4;
    const size_t o_x = 0, o_y = al((size_t)N * H * W * Cp * 4), tot = o_y + al((size_t)N * Ho * Wo * Kp * 4);
    std::stringstream s;
    s << "option graph=0 tune=0\narena bytes=" << tot << "\npbuf name=x bytes=" << xin << "\npbuf name=y bytes=" << yout << "\n";
    s << "import_nchw src=" << bref("x", 0, C, C, H, W, N) << " dst=" << bref("A", o_x, C, Cp, H, W, N) << "\n";
    s << "conv name=op mode=deconv2x in=" << bref("A", o_x, C, Cp, H, W, N) << " out=" << bref("A", o_y, K, Kp, Ho, Wo, N)
      << " w=w_weight" << (bias ? " bias=w_bias" : "") << " act=" << act << " slope=" << slope
      << " cin=" << C << " cout=" << K << "\n";
    s << "export_nchw src=" << bref("A", o_y, K, Kp, Ho, Wo, N) << " dst=" << bref("y", 0, K, K, Ho, Wo, N) << "\n";
    accel_plan* p = nullptr;
    if ((rc = accel_model_add_plan(t.m, "op", s.str().c_str(), &p))) return rc;
    if ((rc = accel_model_write(t.m, "x", x, xin, 0))) return rc;
    if ((rc = accel_plan_finalize(p)) || (rc = accel_plan_run(p))) return rc;
    return accel_model_read(t.m, "y", y, yout, 0);
}

extern "C" int accel_deform_conv2d(accel_ctx* ctx, const float* x, int N, int C, int H, int W,
                                   const float* offset, const float* w, int K, int kh, int kw,
                                   int sh, int sw, int ph, int pw, int dh, int dw, int dg, float* y)
{
    if (!ctx || !x || !offset || !w || !y) return fail(ACCEL_ERR_ARG, "accel_deform_conv2d: NULL argument");
    if (N < 1) return fail(ACCEL_ERR_ARG, "accel_deform_conv2d: N must be >= 1");
    const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
    TempModel t;
    int rc;
    if ((rc = accel_model_create(ctx, &t.m))) return rc;
    if ((rc = set1(t.m, "w_weight", w, {K, C, kh, kw}))) return rc;
    const int OC = 2 * kh * kw * dg;
    const int Cp = (C + 3) / 4 * 4, Kp = (K + 3) / 4 * 4, OCp = (OC + 3) / 4 * 4, CC = kh * kw * Cp;
    const size_t xin = (size_t)N * C * H * W * 4, oin = (size_t)N * OC * Ho * Wo * 4, yout = (size_t)N * K * Ho * Wo * 4;
    size_t off = 0;
    const size_t o_x = off; off += al((size_t)N * H * W * Cp * 4);
    const size_t o_o = off; off += al((size_t)N * Ho * Wo * OCp * 4);
    const size_t o_c = off; off += al((size_t)N * Ho * Wo * CC * 4);
    const size_t o_y = off; off += al((size_t)N * Ho * Wo * Kp * 4);
    std::stringstream s;
    s << "option graph=0 tune=0\narena bytes=" << off << "\npbuf name=x bytes=" << xin << "\npbuf name=o bytes=" << oin << "\npbuf name=y bytes=" << yout << "\n";
    s << "import_nchw src=" << bref("x", 0, C, C, H, W, N) << " dst=" << bref("A", o_x, C, Cp, H, W, N) << "\n";
    s << "import_nchw src=" << bref("o", 0, OC, OC, Ho, Wo, N) << " dst=" << bref("A", o_o, OC, OCp, Ho, Wo, N) << "\n";
    s << "dcn_cols in=" << bref("A", o_x, C, Cp, H, W, N) << " off=" << bref("A", o_o, OC, OCp, Ho, Wo, N)
      << " out=" << bref("A", o_c, CC, CC, Ho, Wo, N) << " k=" << kh << "," << kw << " s=" << sh << "," << sw
      << " p=" << ph << "," << pw << " d=" << dh << "," << dw << " dg=" << dg << "\n";
    s << "conv name=op mode=cols wk=" << kh << "," << kw << " in=" << bref("A", o_c, CC, CC, Ho, Wo, N)
      << " out=" << bref("A", o_y, K, Kp, Ho, Wo, N) << " w=w_weight act=0 k=1,1 cin=" << C << " cout=" << K << "\n";
    s << "export_nchw src=" << bref("A", o_y, K, Kp, Ho, Wo, N) << " dst=" << bref("y", 0, K, K, Ho, Wo, N) << "\n";
    accel_plan* p = nullptr;
    if ((rc = accel_model_add_plan(t.m, "op", s.str().c_str(), &p))) return rc;
    if ((rc = accel_model_write(t.m, "x", x, xin, 0)) || (rc = accel_model_write(t.m, "o", offset, oin, 0))) return rc;
    if ((rc = accel_plan_finalize(p)) || (rc = accel_plan_run(p))) return rc;
    return accel_model_read(t.m, "y", y, yout, 0);
}

extern "C" int accel_pool2d(accel_ctx* ctx, const float* x, int N, int C, int H, int W, int is_max, int full,
                            int kh, int kw, int sh, int sw, int ph, int pw,
                            const float* scale, const float* shift, int relu, float* y)
{
    if (!ctx || !x || !y) return fail(ACCEL_ERR_ARG, "accel_pool2d: NULL argument");
    if (N < 1) return fail(ACCEL_ERR_ARG, "accel_pool2d: N must be >= 1");
    auto po = [&](int in, int k, int s, int p) {
        return full ? 1 + (in + 2 * p - k + s - 1) / s : 1 + (in + 2 * p - k) / s;
    };
    const int Ho = po(H, kh, sh, ph), Wo = po(W, kw, sw, pw);
    TempModel t;
    int rc;
    if ((rc = accel_model_create(ctx, &t.m))) return rc;
    if (scale) {
        std::vector<float> zero(C, 0.f), var(C, 1.0f);
        if ((rc = set1(t.m, "e_gamma", scale, {C})) || (rc = set1(t.m, "e_beta", shift, {C})) ||
            (rc = set1(t.m, "e_moving_mean", zero.data(), {C})) || (rc = set1(t.m, "e_moving_var", var.data(), {C}))) return rc;
    }
    const int Cp = (C + 3) / 4 * 4;
    const size_t xin = (size_t)N * C * H * W * 4, yout = (size_t)N * C * Ho * Wo * 4;
    const size_t o_x = 0, o_y = al((size_t)N * H * W * Cp * 4), tot = o_y + al((size_t)N * Ho * Wo * Cp * 4);
    std::stringstream s;
    s << "option graph=0 tune=0\narena bytes=" << tot << "\npbuf name=x bytes=" << xin << "\npbuf name=y bytes=" << yout << "\n";
    s << "import_nchw src=" << bref("x", 0, C, C, H, W, N) << " dst=" << bref("A", o_x, C, Cp, H, W, N) << "\n";
    s << "pool in=" << bref("A", o_x, C, Cp, H, W, N) << " out=" << bref("A", o_y, C, Cp, Ho, Wo, N)
      << " kind=" << (is_max ? "max" : "avg") << " k=" << kh << "," << kw << " s=" << sh << "," << sw
      << " p=" << ph << "," << pw << (scale ? " bn=e eps=0 fixg=0" : "") << " act=" << (relu ? 1 : 0) << "\n";
    s << "export_nchw src=" << bref("A", o_y, C, Cp, Ho, Wo, N) << " dst=" << bref("y", 0, C, C, Ho, Wo, N) << "\n";
    accel_plan* p = nullptr;
    if ((rc = accel_model_add_plan(t.m, "op", s.str().c_str(), &p))) return rc;
    if ((rc = accel_model_write(t.m, "x", x, xin, 0))) return rc;
    if ((rc = accel_plan_finalize(p)) || (rc = accel_plan_run(p))) return rc;
    return accel_model_read(t.m, "y", y, yout, 0);
}

extern "C" int accel_flow_warp(accel_ctx* ctx, const float* feat, int C, int H, int W, const float* flow, float* out)
{
    if (!ctx || !feat || !flow || !out) return fail(ACCEL_ERR_ARG, "accel_flow_warp: NULL argument");
    if (C % 4) return fail(ACCEL_ERR_ARG, "accel_flow_warp: C must be a multiple of 4");
    TempModel t;
    int rc;
    if ((rc = accel_model_create(ctx, &t.m))) return rc;
    const size_t fb = (size_t)C * H * W * 4, flb = (size_t)2 * H * W * 4;
    const size_t o_f = 0, o_fl = al(fb), o_o = o_fl + al((size_t)H * W * 16), tot = o_o + al(fb);
    std::stringstream s;
    s << "option graph=0 tune=0\narena bytes=" << tot << "\npbuf name=f bytes=" << fb << "\npbuf name=fl bytes=" << flb << "\npbuf name=y bytes=" << fb << "\n";
    s << "import_nchw src=" << bref("f", 0, C, C, H, W) << " dst=" << bref("A", o_f, C, C, H, W) << "\n";
    s << "import_nchw src=" << bref("fl", 0, 2, 2, H, W) << " dst=" << bref("A", o_fl, 2, 4, H, W) << "\n";
    s << "warp feat=" << bref("A", o_f, C, C, H, W) << " flow=" << bref("A", o_fl, 2, 4, H, W) << " out=" << bref("A", o_o, C, C, H, W) << "\n";
    s << "export_nchw src=" << bref("A", o_o, C, C, H, W) << " dst=" << bref("y", 0, C, C, H, W) << "\n";
    accel_plan* p = nullptr;
    if ((rc = accel_model_add_plan(t.m, "op", s.str().c_str(), &p))) return rc;
    if ((rc = accel_model_write(t.m, "f", feat, fb, 0)) || (rc = accel_model_write(t.m, "fl", flow, flb, 0))) return rc;
    if ((rc = accel_plan_finalize(p)) || (rc = accel_plan_run(p))) return rc;
    return accel_model_read(t.m, "y", out, fb, 0);
}

extern "C" int accel_score_fuse(accel_ctx* ctx, const float* left, const float* right, int ncls, int Hs, int Ws,
                                const float* wl, const float* wr, const float* cw, const float* cb,
                                float* logits, uint8_t* labels)
{
    if (!ctx || !left || !wl || (!logits && !labels)) return fail(ACCEL_ERR_ARG, "accel_score_fuse: NULL argument");
    if (right && (!wr || !cw || !cb)) return fail(ACCEL_ERR_ARG, "accel_score_fuse: right branch needs wr, cw, cb");
    TempModel t;
    int rc;
    if ((rc = accel_model_create(ctx, &t.m))) return rc;
    if ((rc = set1(t.m, "wl", wl, {ncls, 1, 32, 32}))) return rc;
    if (right && ((rc = set1(t.m, "wr", wr, {ncls, 1, 32, 32})) || (rc = set1(t.m, "cw", cw, {ncls, 2 * ncls, 1, 1})) ||
                  (rc = set1(t.m, "cb", cb, {ncls})))) return rc;
    const int H = 16 * Hs, W = 16 * Ws, Cp = (ncls + 3) / 4 * 4;
    const size_t sb = (size_t)ncls * Hs * Ws * 4, lb = (size_t)ncls * H * W * 4, yb = (size_t)H * W;
    const size_t o_l = 0, o_r = al((size_t)Hs * Ws * Cp * 4), tot = 2 * o_r;
    std::stringstream s;
    s << "option graph=0 tune=0\narena bytes=" << tot << "\npbuf name=l bytes=" << sb << "\npbuf name=r bytes=" << sb
      << "\npbuf name=logits bytes=" << lb << "\npbuf name=labels bytes=" << al(yb) << "\n";
    s << "import_nchw src=" << bref("l", 0, ncls, ncls, Hs, Ws) << " dst=" << bref("A", o_l, ncls, Cp, Hs, Ws) << "\n";
    if (right) s << "import_nchw src=" << bref("r", 0, ncls, ncls, Hs, Ws) << " dst=" << bref("A", o_r, ncls, Cp, Hs, Ws) << "\n";
    s << "score_tail left=" << bref("A", o_l, ncls, Cp, Hs, Ws) << " wl=wl";
    if (right) s << " right=" << bref("A", o_r, ncls, Cp, Hs, Ws) << " wr=wr cw=cw cb=cb";
    s << " logits=" << bref("logits", 0, ncls, 4, H, W) << " labels=" << bref("labels", 0, 1, 4, H, W)
      << " H=" << H << " W=" << W << " ncls=" << ncls << "\n";
    accel_plan* p = nullptr;
    if ((rc = accel_model_add_plan(t.m, "op", s.str().c_str(), &p))) return rc;
    if ((rc = accel_model_write(t.m, "l", left, sb, 0))) return rc;
    if (right && (rc = accel_model_write(t.m, "r", right, sb, 0))) return rc;
    if ((rc = accel_plan_finalize(p)) || (rc = accel_plan_run(p))) return rc;
    if (logits && (rc = accel_model_read(t.m, "logits", logits, lb, 0))) return rc;
    if (labels && (rc = accel_model_read(t.m, "labels", labels, yb, 0))) return rc;
    return 0;
}

extern "C" int accel_argmax_c(accel_ctx* ctx, const float* logits, int C, int H, int W, uint8_t* labels)
{
    if (!ctx || !logits || !labels) return fail(ACCEL_ERR_ARG, "accel_argmax_c: NULL argument");
    float* d = nullptr;
    unsigned char* l = nullptr;
    const size_t lb = (size_t)C * H * W * 4;
    HIP_TRY(hipMalloc((void**)&d, lb));
    hipError_t e = hipMalloc((void**)&l, (size_t)H * W);
    if (e != hipSuccess) { hipFree(d); return fail(ACCEL_ERR_HIP, "hipMalloc failed"); }
    int rc = 0;
    if (hipMemcpy(d, logits, lb, hipMemcpyHostToDevice) != hipSuccess) rc = fail(ACCEL_ERR_HIP, "H2D copy failed");
    if (!rc && launch_argmax_nchw(d, l, C, H * W, ctx->stream) != hipSuccess) rc = fail(ACCEL_ERR_HIP, "argmax launch failed");
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ACCEL_ERR_HIP, "sync failed");
    if (!rc && hipMemcpy(labels, l, (size_t)H * W, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ACCEL_ERR_HIP, "D2H copy failed");
    hipFree(d); hipFree(l);
    return rc;
}

extern "C" int accel_flow_input(accel_ctx* ctx, const float* cur, const float* prev, int H, int W, float* out)
{
    if (!ctx || !cur || !prev || !out) return fail(ACCEL_ERR_ARG, "accel_flow_input: NULL argument");
    if ((H & 1) || (W & 1)) return fail(ACCEL_ERR_ARG, "accel_flow_input: H and W must be even");
    TempModel t;
    int rc;
    if ((rc = accel_model_create(ctx, &t.m))) return rc;
    const size_t ib = (size_t)3 * H * W * 4, ob = (size_t)6 * (H / 2) * (W / 2) * 4;
    std::stringstream s;
    s << "option graph=0 tune=0\narena bytes=" << al((size_t)(H / 2) * (W / 2) * 32) << "\npbuf name=data bytes=" << ib << "\npbuf name=data_key bytes=" << ib
      << "\npbuf name=y bytes=" << ob << "\n";
    s << "prep_flow cur=" << bref("data", 0, 3, 4, H, W) << " prev=" << bref("data_key", 0, 3, 4, H, W)
      << " dst=" << bref("A", 0, 6, 8, H / 2, W / 2) << " H=" << H << " W=" << W << "\n";
    s << "export_nchw src=" << bref("A", 0, 6, 8, H / 2, W / 2) << " dst=" << bref("y", 0, 6, 8, H / 2, W / 2) << "\n";
    accel_plan* p = nullptr;
    if ((rc = accel_model_add_plan(t.m, "op", s.str().c_str(), &p))) return rc;
    if ((rc = accel_model_write(t.m, "data", cur, ib, 0)) || (rc = accel_model_write(t.m, "data_key", prev, ib, 0))) return rc;
    if ((rc = accel_plan_finalize(p)) || (rc = accel_plan_run(p))) return rc;
    return accel_model_read(t.m, "y", out, ob, 0);
}

// ---------------------------------------------------------------------------
// multi-GPU gather over RCCL (resolved at run time: a process that never creates a communicator needs no librccl,
// and a process that already carries an RCCL -- torch.distributed's -- shares that copy instead of loading a second)
// ---------------------------------------------------------------------------
namespace {
struct RcclId { char internal[128]; };
typedef void* RcclComm;
struct RcclApi {
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};

RcclApi& rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (const char* e = getenv("ACCEL_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) { api.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return api; }
    auto sym = [&](const char* n) { void* f = dlsym(h, n); if (!f && api.why.empty()) api.why = std::string("librccl lacks ") + n; return f; };
    api.GetUniqueId = reinterpret_cast<int (*)(RcclId*)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<int (*)(RcclComm*, int, RcclId, int)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<int (*)(RcclComm)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<int (*)(const void*, size_t, int, int, RcclComm, hipStream_t)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<int (*)(void*, size_t, int, int, RcclComm, hipStream_t)>(sym("ncclRecv"));
    api.GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
    api.ok = api.why.empty();
    return api;
}
}  // namespace

struct accel_comm {
    accel_ctx* ctx;
    RcclComm comm = nullptr;
    int rank = 0, nranks = 1;
    hipStream_t stream = nullptr;             // communication stream
    void* stage[2] = {nullptr, nullptr};      // staging slots: the sender's copy of what is in flight
    size_t stage_bytes[2] = {0, 0};
    hipEvent_t staged[2] = {nullptr, nullptr};   // compute stream: slot filled
    hipEvent_t sent[2] = {nullptr, nullptr};     // communication stream: transfer out of the slot finished
    bool used[2] = {false, false};
    unsigned long n = 0;
};

#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        int r__ = (expr);                                                                                \
        if (r__ != 0) return fail(ACCEL_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r__) : "?"); \
    } while (0)

extern "C" int accel_comm_available(void)
{
    const char* off = getenv("ACCEL_RCCL_UNAVAILABLE");      // tests: this rank behaves as if librccl could not be loaded
    if (off && off[0] == '1') return fail(ACCEL_ERR_COMM, "librccl withheld by ACCEL_RCCL_UNAVAILABLE=1");
    if (!rccl().ok) return fail(ACCEL_ERR_COMM, "%s", rccl().why.c_str());
    return 0;
}

extern "C" int accel_comm_unique_id(void* id128)
{
    if (!id128) return fail(ACCEL_ERR_ARG, "accel_comm_unique_id: NULL argument");
    if (!rccl().ok) return fail(ACCEL_ERR_COMM, "%s", rccl().why.c_str());
    RcclId id;
    RCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return 0;
}

extern "C" int accel_comm_create(accel_ctx* ctx, int rank, int nranks, const void* id128, accel_comm** out)
{
    if (!ctx || !id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(ACCEL_ERR_ARG, "accel_comm_create: bad argument");
    if (!rccl().ok) return fail(ACCEL_ERR_COMM, "%s", rccl().why.c_str());
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<accel_comm> c(new accel_comm());
    c->ctx = ctx; c->rank = rank; c->nranks = nranks;
    RcclId id;
    memcpy(id.internal, id128, 128);
    RCCL_TRY(rccl().CommInitRank(&c->comm, nranks, id, rank));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int s = 0; s < 2; ++s) {
        HIP_TRY(hipEventCreateWithFlags(&c->staged[s], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->sent[s], hipEventDisableTiming));
    }
    *out = c.release();
    return 0;
}

extern "C" int accel_comm_sync(accel_comm* c)
{
    if (!c) return fail(ACCEL_ERR_ARG, "accel_comm_sync: NULL communicator");
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));      // staging copies
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int accel_comm_destroy(accel_comm* c)
{
    if (!c) return 0;
    hipStreamSynchronize(c->ctx->stream);
    hipStreamSynchronize(c->stream);
    if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
    for (int s = 0; s < 2; ++s) {
        if (c->stage[s]) hipFree(c->stage[s]);
        if (c->staged[s]) hipEventDestroy(c->staged[s]);
        if (c->sent[s]) hipEventDestroy(c->sent[s]);
    }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" int accel_gather_frames(accel_comm* c, const void* sendbuf, size_t send_bytes, void* recvbuf_or_null, size_t bytes, int root);

extern "C" int accel_gather_logits(accel_comm* c, const void* sendbuf, void* recvbuf_or_null, size_t bytes, int root)
{
    return accel_gather_frames(c, sendbuf, bytes, recvbuf_or_null, bytes, root);
}

// `bytes` = the slot every rank owns in the root's receive buffer; a PEER always fills its slot (send_bytes == bytes: the root
// posts its receives with that count), the ROOT may contribute fewer bytes (a root that is given fewer clips because it also
// receives everybody else's frames): its own block is a device copy of send_bytes, the rest of its slot is left untouched.
extern "C" int accel_gather_frames(accel_comm* c, const void* sendbuf, size_t send_bytes, void* recvbuf_or_null, size_t bytes, int root)
{
    if (!c || !sendbuf || !bytes || !send_bytes || send_bytes > bytes) return fail(ACCEL_ERR_ARG, "accel_gather_frames: bad argument");
    if (c->rank != root && send_bytes != bytes) return fail(ACCEL_ERR_ARG, "accel_gather_frames: only the root may send less than a full slot");
    if (root < 0 || root >= c->nranks) return fail(ACCEL_ERR_ARG, "accel_gather_frames: root %d out of range", root);
    if (c->rank == root && !recvbuf_or_null) return fail(ACCEL_ERR_ARG, "accel_gather_frames: the root needs a receive buffer");
    HIP_TRY(hipSetDevice(c->ctx->device));
    const int s = (int)(c->n & 1);
    hipStream_t compute = c->ctx->stream;
    if (c->stage_bytes[s] < bytes) {
        if (c->stage[s]) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->stage[s])); c->stage[s] = nullptr; }
        HIP_TRY(hipMalloc(&c->stage[s], bytes));
        c->stage_bytes[s] = bytes;
    }
    // (1) compute stream: wait until the transfer issued from this slot two calls ago has left it, then refill it
    if (c->used[s]) HIP_TRY(hipStreamWaitEvent(compute, c->sent[s], 0));
    HIP_TRY(launch_copy_bytes(sendbuf, c->stage[s], send_bytes, compute));      // (one kernel at HBM rate; the runtime's blit moves 25 MB pieces at 1.2 TB/s)
    HIP_TRY(hipEventRecord(c->staged[s], compute));
    // (2) communication stream: point-to-point to the root, every peer over its own link; the root copies its own block
    HIP_TRY(hipStreamWaitEvent(c->stream, c->staged[s], 0));
    char* recv = static_cast<char*>(recvbuf_or_null);
    // ACCEL_GATHER_SELF_SENDRECV=1 (tests on one GPU): the root's own block travels through ncclSend / ncclRecv to itself
    // inside the group instead of a device copy, so the resolved RCCL entry points, the dtype enum and the group
    // semantics of this function execute even in a world of one
    const char* self_sr = getenv("ACCEL_GATHER_SELF_SENDRECV");
    if (c->rank == root && self_sr && self_sr[0] == '1') {
        RCCL_TRY(rccl().GroupStart());
        RCCL_TRY(rccl().Send(c->stage[s], bytes, /*ncclUint8*/ 1, root, c->comm, c->stream));
        for (int r = 0; r < c->nranks; ++r)
            RCCL_TRY(rccl().Recv(recv + (size_t)r * bytes, bytes, /*ncclUint8*/ 1, r, c->comm, c->stream));
        RCCL_TRY(rccl().GroupEnd());
    } else if (c->rank == root) {
        HIP_TRY(launch_copy_bytes(c->stage[s], recv + (size_t)root * bytes, send_bytes, c->stream));
        if (c->nranks > 1) {
            RCCL_TRY(rccl().GroupStart());
            for (int r = 0; r < c->nranks; ++r)
                if (r != root) RCCL_TRY(rccl().Recv(recv + (size_t)r * bytes, bytes, /*ncclUint8*/ 1, r, c->comm, c->stream));
            RCCL_TRY(rccl().GroupEnd());
        }
    } else {
        RCCL_TRY(rccl().GroupStart());
        RCCL_TRY(rccl().Send(c->stage[s], bytes, /*ncclUint8*/ 1, root, c->comm, c->stream));
        RCCL_TRY(rccl().GroupEnd());
    }
    HIP_TRY(hipEventRecord(c->sent[s], c->stream));
    c->used[s] = true;
    ++c->n;
    return 0;
}

// ---- the gather at SCORE resolution ------------------------------------------------------------------------------------------------
// With uniform upsampling filters the fp32 logits of a frame (19 x H x W: 159 MB at 1024x2048) are a pure function of the fused score
// map the plan leaves in `scores` (20 x H/16 x W/16 floats: 0.66 MB).  Peers send that map; the root expands every block with the very
// launch its own plans end with (score_tail_uniform_kernel over the model's filter and bias), so the expanded logits are bit-identical
// to what the peer computed.  Round 4 measured the logits gather at 93 GB/s per peer link against ~77 sustainable: link-bound.
extern "C" int accel_expand_scores(accel_plan* p, const void* scores_dev, int n_images, float* logits_dev, unsigned char* labels_dev, accel_comm* on_comm_stream_of)
{
    if (!p || !p->finalized || !scores_dev || !logits_dev || !labels_dev || n_images < 1) return fail(ACCEL_ERR_ARG, "accel_expand_scores: bad argument");
    if (!p->has_score_tpl) return fail(ACCEL_ERR_PLAN, "accel_expand_scores: plan '%s' forms its logits at full resolution (no `scores` map: the upsampling "
                                                       "filters are not uniform, or the plan has no score tail)", p->role.c_str());
    HIP_TRY(hipSetDevice(p->m->ctx->device));
    ScoreTailParams q = p->score_tpl;
    q.left = static_cast<const float*>(scores_dev);
    q.logits = logits_dev;
    q.labels = labels_dev;
    q.N = n_images;
    const hipError_t e = launch_score_tail(q, on_comm_stream_of ? on_comm_stream_of->stream : p->m->ctx->stream);
    if (e != hipSuccess) return fail(ACCEL_ERR_HIP, "accel_expand_scores: launch failed: %s", hipGetErrorString(e));
    return 0;
}

// Every rank contributes the `scores` buffer its plan `p` has just filled (own_images of the slot_images a slot holds: only the root may
// contribute fewer; every rank passes the plan of the SAME role -- key and non-key frames expand differently: one head without a bias
// against the fused heads with the correction bias); the root receives the maps into recv_scores ([nranks][slot_images] maps) and
// expands rank r's block into logits_out / labels_out at image offset r * slot_images, on the communication stream right behind the
// receives -- no host synchronisation anywhere; the results are valid after accel_comm_sync.  Peers pass NULL for the three root-side buffers.
extern "C" int accel_gather_scores(accel_comm* c, accel_plan* p, int own_images, int slot_images, void* recv_scores, float* logits_out,
                                   unsigned char* labels_out, int root)
{
    if (!c || !p || !p->finalized || own_images < 1 || slot_images < own_images) return fail(ACCEL_ERR_ARG, "accel_gather_scores: bad argument");
    if (!p->has_score_tpl) return fail(ACCEL_ERR_PLAN, "accel_gather_scores: plan '%s' has no `scores` map (see accel_expand_scores)", p->role.c_str());
    if (own_images > p->score_images) return fail(ACCEL_ERR_ARG, "accel_gather_scores: %d images asked for, the plan produces %d", own_images, p->score_images);
    if (c->rank == root && (!recv_scores || !logits_out || !labels_out)) return fail(ACCEL_ERR_ARG, "accel_gather_scores: the root needs its three buffers");
    const ScoreTailParams& t = p->score_tpl;
    const size_t map_bytes = (size_t)t.Hs * t.Ws * p->score_zcs * sizeof(float);
    int rc = accel_gather_frames(c, p->m->pbufs["scores"].ptr, (size_t)own_images * map_bytes, recv_scores, (size_t)slot_images * map_bytes, root);
    if (rc || c->rank != root) return rc;
    const size_t img_logits = (size_t)t.ncls * t.H * t.W, img_labels = (size_t)t.H * t.W;
    for (int r = 0; r < c->nranks; ++r) {
        const int n = r == root ? own_images : slot_images;
        rc = accel_expand_scores(p, static_cast<const char*>(recv_scores) + (size_t)r * slot_images * map_bytes, n,
                                 logits_out + (size_t)r * slot_images * img_logits, labels_out + (size_t)r * slot_images * img_labels, c);
        if (rc) return rc;
    }
    return 0;
}
