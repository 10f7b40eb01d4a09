// Shared declarations of the UNet executor (unet.hip: registry + inference forward; unet_train.hip:
// training forward tape + backward).  Host code only.
#pragma once
#include <cstdlib>
#include <array>
#include <map>
#include <tuple>
#include <string>
#include <vector>
#include "blocks.h"
#include "det.h"

namespace ofd {

struct Param {
    std::string name;
    int ndim;
    int shape[4];
    size_t numel;
    size_t offset;   // floats into d_params (16-byte aligned)
    bool set;
};

struct ConvDesc {
    std::string wname;   // "<prefix>.weight"
    int Cout, Cin, Cin_pad, ksize;
    float ws_eps;        // < 0: plain conv
    int unshuffle;
    size_t w_off;        // bf16 elements into d_wbuf
    long phase_off = -1; // up-sample convs: 4 collapsed 2x2 kernels (bf16 elements into d_wbuf), else -1
    long pack8_off = -1; // 7x7 with <= 8 input channels: the tap-pair-packed weights for an 8-channel input (inference), else -1
};

struct Tensor {
    bf16_t* p = nullptr;
    int C = 0, H = 0, W = 0;
    bf16_t* g = nullptr;     // gradient buffer (training forward only)
};

enum ProfClass { PC_CONV3 = 0, PC_CONV3_64, PC_CONV3_PP, PC_CONV1, PC_CONV7, PC_GN, PC_RESOUT, PC_LN, PC_LINATTN, PC_FLASH, PC_MISC, PC_WGRAD3, PC_WGRAD1, PC_DGRAD3, PC_DGRAD1, PC_GNBWD, PC_LABWD, PC_FLASHBWD, PC_CONVUP, PC_COUNT };
// class = the group of layers the executor asked for.  The NAME of a class is the kernel that serves it under the library's current switches
// (prof_class_name, unet.hip): `<kernel family> [<layer group>]` -- classes served by the same kernel share the text before " [", which is
// what bench.py groups by when it picks the dominant kernel.
const char* prof_class_name(int cls);

struct ProfRec {
    int cls;
    hipEvent_t e0, e1;
    double flops, bytes;
    std::string label;
};


struct SrcSpec {
    Tensor t;
    int upsample = 0;
    int unshuffle = 0, p1 = 0, p2 = 0;
};

// one record per composite op of the training forward, replayed in reverse by the backward
enum TapeKind { TK_CONV = 0, TK_RES, TK_LINATTN, TK_MIDATTN };
struct TapeRec {
    int kind = TK_CONV;
    std::string name;
    std::vector<SrcSpec> srcs;       // conv / resblock inputs
    Tensor out, h1, h2, xn, qkv, ao, o2, x;
    Tensor act1;                     // resblock, training: SiLU(GroupNorm(h1)) as the forward materialised it (p == nullptr: the backward recomputes it)
    float *a1 = nullptr, *s1 = nullptr, *a2 = nullptr, *s2 = nullptr, *st1 = nullptr, *st2 = nullptr;
    float *ctx = nullptr, *ml = nullptr, *lse = nullptr;
};

struct TrainState {                  // what the backward needs from the last training forward
    bool valid = false;
    int B = 0, H = 0, W = 0;
    char* workspace = nullptr;
    size_t workspace_bytes = 0;
    const int64_t* t = nullptr;
    float *ss = nullptr, *temb = nullptr, *temb_silu = nullptr;
    Tensor xin, r, xf;
    size_t persist_used = 0;
};

}  // namespace ofd

using namespace ofd;

struct ofd_unet {
    ofd_unet_config cfg;
    std::vector<int> dims;        // [dim, dim*1, dim*2, dim*4, dim*8]
    std::vector<Param> params;
    std::map<std::string, int> pindex;
    std::vector<ConvDesc> convs;
    std::map<std::string, int> cindex;
    std::vector<std::string> resblocks;       // names in forward order
    std::map<std::string, int> ss_offset;     // resblock -> offset in the scale/shift row
    int ss_stride = 0;
    float* d_params = nullptr;
    bool owns_params = true;                  // false once the caller bound its own flat buffer
    size_t n_param_floats = 0;
    bf16_t* d_wbuf = nullptr;
    size_t n_wbuf = 0;
    MlpDesc* d_mlp = nullptr;
    bf16_t* d_labuf = nullptr;                // fused LinearAttention weights (C <= 128): wq | wkv | wout per block
    size_t n_labuf = 0;
    std::map<std::string, std::pair<size_t, int>> la_fused;   // block name -> (offset, C)
    bool prepared = false;
    // training (unet_train.hip): fp32 gradients laid out like d_params, tap-flipped transposed conv
    // weights for the data gradients, the tape of the last training forward
    float* d_grads = nullptr;            // bound by the caller (ofd_unet_bind_grad_buffer), not owned
    std::map<std::string, std::pair<size_t, size_t>> prange;   // op prefix -> [begin, end) floats of its parameters
    bf16_t* d_wtbuf = nullptr;
    ofd_weight_prep_desc* d_prep = nullptr;   // device tables of the batched weight preparation / transposition (built on first use)
    ofd_weight_prep_desc* d_tr = nullptr;
    int n_prep = 0, prep_blocks = 0, n_tr = 0, tr_blocks = 0;
    float* d_wacc = nullptr;      // fp32 weight-gradient accumulators of every conv (same offsets as d_wbuf): ONE memset per backward
    // deterministic backward (det.h): fixed-point shadows of d_grads | d_wacc | the time-embedding gradient rows, in this order
    bool deterministic = false;
    long long* d_fx = nullptr;
    size_t fx_cap = 0;            // elements
    unsigned* d_det_miss = nullptr;
    DetCtx det_host{};            // the context of the backward in flight (kept alive for the asynchronous upload)
    bool wt_prepared = false;
    std::vector<TapeRec> tape;
    TrainState ts;
    // hipGraph replay of the inference forward (ofd_unet_set_graph): one instantiated graph per (workspace, shape, stream);
    // inputs / output go through fixed staging buffers at the end of the workspace so every kernel argument is static
    struct GraphEntry {
        void* workspace; int B, H, W, Cx, Cc; hipStream_t stream;
        int state;                         // 0: seen once (eager run done), 2: instantiated
        hipGraph_t graph; hipGraphExec_t exec;
    };
    bool graph_enabled = false;
    hipStream_t cap_stream = nullptr;   // private stream the forward is captured on (the caller's may be the un-capturable null stream)
    std::vector<GraphEntry> graphs;
    // workspace plans of the training step, keyed by (B, H, W): {small, persist, scratch} bytes (a plan is a dry run of the
    // whole forward + backward: worth caching, three of them per step showed up as host time at small resolutions)
    std::map<std::tuple<int, int, int>, std::array<size_t, 3>> plans;
    // last forward: taps
    std::map<std::string, Tensor> taps;
    int last_B = 0;
    // two half-batch forwards on two streams (ofd_unet_set_split_streams; OFD_SPLIT_STREAMS / OFD_SPLIT_OFFSET give the defaults)
    // -1 (default): on for even batches of at least 2^21 pixels in all (the BASELINE sizes: same-box A/B 31.24 -> 30.82 ms per denoise step at
    // 16 x 440 x 1024, profiles/r03_split_streams_ab.jsonl; small problems are launch-bound and would pay the second launch sequence), 0: off, 1: on
    int split_streams = getenv("OFD_SPLIT_STREAMS") ? atoi(getenv("OFD_SPLIT_STREAMS")) : -1;
    int split_offset = getenv("OFD_SPLIT_OFFSET") ? atoi(getenv("OFD_SPLIT_OFFSET")) : 1;   // blocks half 1 starts behind half 0
    hipStream_t s2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_phase = nullptr;
    std::map<std::string, Tensor> taps_half0;
    bool last_split = false;
    bool debug_taps = false;       // ofd_unet_set_debug_taps: every tap of the inference forward is materialised (the fused final conv is off)
    // profiling
    bool profiling = false;
    std::string dump_path;                    // per-launch CSV (class,label,ms,flops,bytes) appended on resolve
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    double acc_ms[PC_COUNT] = {0}, acc_flops[PC_COUNT] = {0}, acc_bytes[PC_COUNT] = {0};
    long long acc_launch[PC_COUNT] = {0};

    const float* P(const std::string& n) const {
        auto it = pindex.find(n);
        return it == pindex.end() ? nullptr : d_params + params[it->second].offset;
    }
    const bf16_t* CW(const std::string& prefix) const { return d_wbuf + convs[cindex.at(prefix)].w_off; }
    float* G(const std::string& n) const {
        auto it = pindex.find(n);
        return (it == pindex.end() || !d_grads) ? nullptr : d_grads + params[it->second].offset;
    }
};


namespace ofd {

// ------------------------------------------------------------------------------- forward context
struct Ctx {
    ofd_unet* u;
    hipStream_t s;
    int B;
    char* persist;
    size_t persist_cap, persist_used = 0;
    char* scratch;
    size_t scratch_cap, scratch_used = 0;
    float* ss;
    int rc = OFD_OK;
    bool train = false;              // keep every intermediate and record the tape
    bool dry = false;                // size planning: allocate and count, launch nothing
    size_t grad_offset = 0;          // bytes from an activation to its gradient buffer (training)
    size_t scratch_high = 0;

    void* alloc(char* base, size_t& used, size_t cap, size_t bytes) {
        bytes = (bytes + 255) / 256 * 256;
        if (used + bytes > cap) {
            if (rc == OFD_OK) { set_error("unet_forward: workspace too small"); rc = OFD_ERR_WORKSPACE; }
            return nullptr;
        }
        void* p = base + used;
        used += bytes;
        if (base == scratch && used > scratch_high) scratch_high = used;
        return p;
    }
    Tensor keep(int C, int H, int W) {
        Tensor t;
        t.p = (bf16_t*)alloc(persist, persist_used, persist_cap, (size_t)B * H * W * C * 2);
        t.C = C; t.H = H; t.W = W;
        if (train && t.p) t.g = (bf16_t*)((char*)t.p + grad_offset);
        return t;
    }
    float* keepf(size_t n) { return (float*)alloc(persist, persist_used, persist_cap, n * 4); }
    Tensor tmp(int C, int H, int W) {
        if (train) return keep(C, H, W);
        Tensor t;
        t.p = (bf16_t*)alloc(scratch, scratch_used, scratch_cap, (size_t)B * H * W * C * 2);
        t.C = C; t.H = H; t.W = W;
        return t;
    }
    float* tmpf(size_t n) { return (float*)alloc(scratch, scratch_used, scratch_cap, n * 4); }
    // split forward: half 0 records `signal_ev` once it has enqueued `signal_at` blocks; half 1's stream waits on it
    int blocks_done = 0, signal_at = -1;
    hipEvent_t signal_ev = nullptr;
    bool signalled = false;
    void reset_scratch() {
        scratch_used = 0;
        if (signal_ev && !signalled && !dry && blocks_done++ >= signal_at) { (void)hipEventRecord(signal_ev, s); signalled = true; }
    }

    // profiling bracket.  Back-to-back regions share an event: the end of one is the start of the next unless something was
    // launched in between (RUN outside a region) -- half the event records, half their cost in the timed region.
    bool in_region = false, chain_ok = false;
    hipEvent_t last_end = nullptr;
    hipEvent_t next_event() {
        while (u->pool.size() < u->pool_used + 1) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            u->pool.push_back(e);
        }
        return u->pool[u->pool_used++];
    }
    void begin(int cls, double flops, double bytes, const std::string& label = std::string()) {
        if (!u->profiling || dry) return;
        hipEvent_t e0 = chain_ok ? last_end : next_event();
        hipEvent_t e1 = next_event();
        if (!e0 || !e1) return;
        ProfRec r{cls, e0, e1, flops, bytes, label};
        if (!chain_ok) (void)hipEventRecord(r.e0, s);
        u->recs.push_back(r);
        in_region = true;
    }
    void end() {
        if (!u->profiling || dry || u->recs.empty() || !in_region) return;
        (void)hipEventRecord(u->recs.back().e1, s);
        last_end = u->recs.back().e1;
        chain_ok = true;
        in_region = false;
    }
};

// switches of the training LinearAttention fusions, shared by the forward (unet.hip) and the backward (unet_train.hip)
static inline bool la_env_on(const char* name) { const char* e = getenv(name); return e ? atoi(e) != 0 : true; }
static inline bool la_fuse_to_out() { static const bool v = la_env_on("OFD_LA_FUSE_TO_OUT"); return v; }
static inline bool la_bwd_fuse_qkv() { static const bool v = la_env_on("OFD_LA_BWD_FUSE_QKV"); return v; }
static inline bool la_bwd_fuse_dao() { static const bool v = la_env_on("OFD_LA_BWD_FUSE_DAO"); return v; }
static inline bool la_recompute_q() { static const bool v = la_env_on("OFD_LA_RECOMPUTE_Q"); return v; }
// 64-channel block with every fusion on: the backward needs neither dout nor ao (the forward does not write ao then)
static inline bool la_train_no_ao(int C) { return C == 64 && la_fuse_to_out() && la_bwd_fuse_qkv() && la_bwd_fuse_dao(); }
// the training forward of a 64-channel block as the two fused passes (la_fused.hip TRAIN forms); OFD_LA_TRAIN_FUSED=0: LayerNorm + to_qkv conv + core
static inline bool la_train_fused(int C) {
    static const bool v = la_env_on("OFD_LA_TRAIN_FUSED");
    return v && la_train_no_ao(C) && la_recompute_q();
}

#define RUN(expr)                         \
    do {                                  \
        if (c.rc == OFD_OK && !c.dry) {   \
            if (!c.in_region) c.chain_ok = false; \
            int rc__ = (expr);            \
            if (rc__ != OFD_OK) c.rc = rc__; \
        }                                 \
    } while (0)



float site_eps(const ofd_unet* u, const std::string& site);
void conv(Ctx& c, const std::string& prefix, const std::vector<SrcSpec>& srcs, Tensor out, const float* in_scale, const float* in_shift,
          const bf16_t* residual, const bf16_t* res_act, const float* res_scale, const float* res_shift, float* gn_partial, int cout0 = 0,
          const FcFuse* fc = nullptr);
size_t persist_bytes(const ofd_unet* u, int B, int H, int W);
size_t scratch_bytes(const ofd_unet* u, int B, int H, int W);
size_t small_bytes(const ofd_unet* u, int B);
// the whole forward (DD:363-417) on an initialised context; fills u->taps (and u->tape / u->ts when c.train)
int run_forward(Ctx& c, const float* x, int Cx, const float* cond, int Cc, const int64_t* t, float* out, int H, int W,
                float* temb, float* temb_silu);

}  // namespace ofd
