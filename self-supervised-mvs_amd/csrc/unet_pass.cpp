// One C call per pass of a cost-volume regulariser.
//
// The reference runs CostRegNet (jdacs/models/mvsnet.py:37-74, jdacs-ms/models/network.py:44-74) as ~25 nn.Module calls forward
// and as many autograd nodes backward; this library's Python mirror (ops.UNetRegulariserFn) had already made that ONE autograd node,
// but the node still issued ~45 C-ABI calls forward and ~55 backward from Python: pointer extraction, ctypes marshalling, a torch
// allocation or three per call -- ~1.7 ms of launch-thread time per training step for kernels that take 3 ms.  These two entry
// points are the same launch sequence (same kernels, same order, same streams) driven from C: the caller hands over a table of
// blocks and tables of device pointers (its own allocations; nothing is allocated here), and gets the whole forward or backward
// pass enqueued.  The backward pass forks every weight gradient to `side` behind a HIP event recorded on `main` where the Python
// node forked it, and joins once at the end if asked to.
//
// Block program: block i reads the output of block src (-1: the volume x), runs conv / transposed conv (k3 p1, stride 1|2, bias-free)
// -> BatchNorm(train) -> ReLU and adds the output of block skip (-1: none) AFTER the ReLU; the closing `prob` layer is a stride-1
// convolution with bias.  Every tensor is channels-last-3d fp32 as everywhere in this library.
#include <string.h>
#include "mvs_rt.h"

// (include/mvs_hip.h is the C header of the boundary; it declares its own hipStream_t for callers without the HIP headers, so the
//  library's sources declare what they use of each other directly)
#define MVS_UNET_MAX_BLOCKS 32
extern "C" {
struct MvsUnetBlock {                 // == include/mvs_hip.h
    int transposed, stride, src, skip;
    float eps, momentum;
    int cin, cout, d, h, w;           // channels; (d, h, w) = spatial dims of the block's INPUT
};
int mvs_conv3d_fwd(const float* x, const float* w, float* y, float* ws, int B, int D, int H, int W, int Cin, int Cout, int stride,
                   const float* scale, const float* shift, const float* skip, int relu, double* stat_slots, int nslots, int ws_packed,
                   hipStream_t stream);
int mvs_convT3d_fwd(const float* x, const float* w, float* y, float* ws, int B, int D, int H, int W, int Cin, int Cout, int stride,
                    const float* scale, const float* shift, const float* skip, int relu, double* stat_slots, int nslots, int ws_packed,
                    hipStream_t stream);
int mvs_conv3d_dgrad(const float* gy, const float* w, const float* add, float* gx, float* ws, int B, int D, int H, int W, int Cin, int Cout,
                     int stride, const float* bn_raw, const float* bn_stats, double* bn_slots, int nslots, int ws_packed, hipStream_t stream);
int mvs_convT3d_dgrad(const float* gy, const float* w, const float* add, float* gx, float* ws, int B, int D, int H, int W, int Cin, int Cout,
                      int stride, const float* bn_raw, const float* bn_stats, double* bn_slots, int nslots, int ws_packed, hipStream_t stream);
int mvs_conv3d_wgrad(const float* x, const float* gy, float* gw, float* ws, int B, int D, int H, int W, int Cin, int Cout, int stride,
                     hipStream_t stream);
int mvs_convT3d_wgrad(const float* x, const float* gy, float* gw, float* ws, int B, int D, int H, int W, int Cin, int Cout, int stride,
                      hipStream_t stream);
int mvs_bn_relu_fwd_slots(const float* x, const double* slots, int nslots, int G, long long Vg, int C, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, const float* skip, int relu, float* stats,
                          float* y, hipStream_t stream);
int mvs_bn_bwd_reduce_slots(const float* dy, const float* x, const float* stats, int relu, int G, long long Vg, int C, double* slots,
                            int nslots, hipStream_t stream);
int mvs_bn_relu_bwd_slots(const float* dy, const float* x, const float* stats, const double* slots, int nslots, int relu, int G, long long Vg,
                          int C, float* dx, float* dgamma, float* dbeta, hipStream_t stream);
}

#if defined(MVS_CPU_EMUL)
// the emulation runs every launch synchronously: a fork / join is a no-op
struct MvsForkJoin {
    void fork(hipStream_t, hipStream_t) {}
    void join(hipStream_t, hipStream_t) {}
};
#else
// ONE reusable event per thread: hipStreamWaitEvent captures the record that is current when it is called, so the event can be
// recorded again for the next fork while an earlier wait is still queued
struct MvsForkJoin {
    static hipEvent_t event() {
        static thread_local hipEvent_t ev = nullptr;
        if (!ev) (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        return ev;
    }
    void fork(hipStream_t from, hipStream_t to) {
        hipEvent_t ev = event();
        (void)hipEventRecord(ev, from);
        (void)hipStreamWaitEvent(to, ev, 0);
    }
    void join(hipStream_t from, hipStream_t to) { fork(from, to); }
};
#endif

// ---- measurement hook (bench.py's roofline kernel is a weight gradient inside this call): HIP events around ONE block's weight
// gradient, on the stream it runs on.  Process-wide, like the mvs_set_tuning knobs; not part of the data path's contract. ----
static int g_time_block = -2;                  // block whose weight gradient is bracketed (n = the prob layer); < 0: off
#if defined(MVS_CPU_EMUL)
extern "C" int mvs_unet_time_wgrad(int block) { g_time_block = block; return MVS_OK; }
extern "C" int mvs_unet_time_read(float*, int) { return 0; }
struct MvsBracket {
    MvsBracket(int, hipStream_t) {}
    void stop() {}
};
#else
#define MVS_TIME_RING 1024
static hipEvent_t g_time_ev[MVS_TIME_RING][2];
static int g_time_n = 0, g_time_made = 0;
extern "C" int mvs_unet_time_wgrad(int block) {
    g_time_block = block;          // (brackets recorded so far stay until mvs_unet_time_read collects them)
    return MVS_OK;
}
// waits for the recorded brackets, writes their durations (ms, in launch order) and forgets them; returns how many
extern "C" int mvs_unet_time_read(float* ms, int max_n) {
    int k = 0;
    for (; k < g_time_n && k < max_n; ++k) {
        (void)hipEventSynchronize(g_time_ev[k][1]);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_time_ev[k][0], g_time_ev[k][1]);
        ms[k] = t;
    }
    g_time_n = 0;
    return k;
}
struct MvsBracket {
    int slot;
    hipStream_t st;
    MvsBracket(int block, hipStream_t s) : slot(-1), st(s) {
        if (block != g_time_block || g_time_n >= MVS_TIME_RING) return;
        slot = g_time_n++;
        if (slot >= g_time_made) {
            (void)hipEventCreate(&g_time_ev[slot][0]);
            (void)hipEventCreate(&g_time_ev[slot][1]);
            g_time_made = slot + 1;
        }
        (void)hipEventRecord(g_time_ev[slot][0], st);
    }
    void stop() {
        if (slot >= 0) (void)hipEventRecord(g_time_ev[slot][1], st);
    }
};
#endif

#define MVS_TRY(expr)            \
    do {                         \
        const int rc_ = (expr);  \
        if (rc_) return rc_;     \
    } while (0)

static void out_dims(const MvsUnetBlock& b, int& d, int& h, int& w) {
    if (b.transposed && b.stride == 2) { d = 2 * b.d; h = 2 * b.h; w = 2 * b.w; }
    else if (b.stride == 2) { d = (b.d - 1) / 2 + 1; h = (b.h - 1) / 2 + 1; w = (b.w - 1) / 2 + 1; }
    else { d = b.d; h = b.h; w = b.w; }
}

static int check_program(int n, const MvsUnetBlock* blk, const char* what) {
    MVS_REQUIRE(n >= 1 && n <= MVS_UNET_MAX_BLOCKS && blk, MVS_ERR_SHAPE, "%s: 1..%d blocks, got %d", what, MVS_UNET_MAX_BLOCKS, n);
    for (int i = 0; i < n; ++i) {
        MVS_REQUIRE(blk[i].src >= -1 && blk[i].src < i && blk[i].skip >= -1 && blk[i].skip < i, MVS_ERR_SHAPE,
                    "%s: block %d reads block %d / adds block %d: a block may only use earlier blocks", what, i, blk[i].src, blk[i].skip);
        MVS_REQUIRE(blk[i].stride == 1 || blk[i].stride == 2, MVS_ERR_UNSUPPORTED, "%s: block %d has stride %d", what, i, blk[i].stride);
    }
    return MVS_OK;
}

extern "C" int mvs_unet_fwd(int n, const MvsUnetBlock* blk, int B, const float* x, const float* const* w, const float* const* gamma,
                            const float* const* beta, float* const* running_mean, float* const* running_var, float* const* packed,
                            float* const* raw, float* const* y, float* const* stats, double* const* slots, const int* nslots,
                            const float* wprob, const float* bprob, int prob_cout, float* ws_prob, float* logits, hipStream_t stream) {
    MVS_TRY(check_program(n, blk, "mvs_unet_fwd"));
    MVS_REQUIRE(x && w && gamma && beta && running_mean && running_var && packed && raw && y && stats && slots && nslots && wprob &&
                ws_prob && logits, MVS_ERR_NULL, "mvs_unet_fwd: null pointer argument");
    for (int i = 0; i < n; ++i) {
        const MvsUnetBlock& b = blk[i];
        const float* xin = b.src < 0 ? x : y[b.src];
        if (b.transposed)
            MVS_TRY(mvs_convT3d_fwd(xin, w[i], raw[i], packed[i], B, b.d, b.h, b.w, b.cin, b.cout, b.stride, nullptr, nullptr, nullptr, 0,
                                    slots[i], nslots[i], 1, stream));
        else
            MVS_TRY(mvs_conv3d_fwd(xin, w[i], raw[i], packed[i], B, b.d, b.h, b.w, b.cin, b.cout, b.stride, nullptr, nullptr, nullptr, 0,
                                   slots[i], nslots[i], 1, stream));
        int od, oh, ow;
        out_dims(b, od, oh, ow);
        MVS_TRY(mvs_bn_relu_fwd_slots(raw[i], slots[i], nslots[i], 1, (long long)B * od * oh * ow, b.cout, gamma[i], beta[i], b.eps,
                                      b.momentum, running_mean[i], running_var[i], b.skip >= 0 ? y[b.skip] : nullptr, 1, stats[i], y[i],
                                      stream));
    }
    int od, oh, ow;
    out_dims(blk[n - 1], od, oh, ow);
    return mvs_conv3d_fwd(y[n - 1], wprob, logits, ws_prob, B, od, oh, ow, blk[n - 1].cout, prob_cout, 1, nullptr, bprob, nullptr, 0, nullptr,
                          0, 0, stream);
}

extern "C" int mvs_unet_bwd(int n, const MvsUnetBlock* blk, int B, const float* x, const float* const* w, const float* wprob,
                            int prob_cout, const float* const* y, const float* const* raw, const float* const* stats,
                            double* const* slots_b, const int* nslots, float* const* packed_dgrad, const float* glogits,
                            float* const* gbuf, float* const* draw, float* gx, float* const* gw, float* const* wgrad_ws,
                            float* const* dgamma, float* const* dbeta, hipStream_t main_stream, hipStream_t side_stream, int join,
                            int* side_stream_used) {
    MVS_TRY(check_program(n, blk, "mvs_unet_bwd"));
    MVS_REQUIRE(x && w && wprob && y && raw && stats && slots_b && nslots && packed_dgrad && glogits && gbuf && draw && gw && wgrad_ws &&
                dgamma && dbeta, MVS_ERR_NULL, "mvs_unet_bwd: null pointer argument");
    MvsForkJoin fj;
    const bool async = side_stream != nullptr && side_stream != main_stream;
    bool side_used = false;
    auto wgrad = [&](int i, const float* xin, const float* gout, const MvsUnetBlock* b, int d, int h, int wd, int cin, int cout, int stride,
                     int transposed) -> int {
        (void)b;
        if (!gw[i]) return MVS_OK;                       // this weight needs no gradient
        hipStream_t st = main_stream;
        if (async) {
            fj.fork(main_stream, side_stream);            // gout was produced on the main stream
            st = side_stream;
            side_used = true;
        }
        MvsBracket br(i, st);
        const int rc = transposed ? mvs_convT3d_wgrad(xin, gout, gw[i], wgrad_ws[i], B, d, h, wd, cin, cout, stride, st)
                                  : mvs_conv3d_wgrad(xin, gout, gw[i], wgrad_ws[i], B, d, h, wd, cin, cout, stride, st);
        br.stop();
        return rc;
    };
    // the consumer that contributes LAST to a block's output gradient (blocks run last to first; within a block the skip contribution
    // precedes the input gradient): if it does so through its input gradient, that kernel's epilogue also sums the block's BatchNorm
    // backward statistics (ops.UNetRegulariserFn.backward)
    int last[MVS_UNET_MAX_BLOCKS];
    for (int j = 0; j < n; ++j) last[j] = n;             // n = the prob layer (only block n-1 feeds it)
    for (int i = 0; i < n; ++i) {
        if (blk[i].src >= 0 && i < last[blk[i].src]) last[blk[i].src] = i;
        if (blk[i].skip >= 0 && i < last[blk[i].skip]) last[blk[i].skip] = i;
    }
    const float* gp[MVS_UNET_MAX_BLOCKS];               // current gradient w.r.t. each block's output (null: none yet)
    bool have[MVS_UNET_MAX_BLOCKS];                     // its backward statistics are already in slots_b[j]
    for (int j = 0; j < n; ++j) { gp[j] = nullptr; have[j] = false; }
    bool gx_written = false;                            // a second reader of the volume x ADDS to gx (the last write used to win)
    // a failure after the first fork must not leave work on the side stream unjoined (and must tell the caller the stream was used)
    struct Closer {
        MvsForkJoin& fj; bool& used; hipStream_t side, main; int* out; bool done = false;
        ~Closer() { if (!done) { if (used) fj.join(side, main); if (out) *out = used ? 1 : 0; } }
    } closer{fj, side_used, side_stream, main_stream, side_stream_used};

    // ---- prob layer ----
    int od, oh, ow;
    out_dims(blk[n - 1], od, oh, ow);
    {
        const bool bn = last[n - 1] == n;
        MVS_TRY(mvs_conv3d_dgrad(glogits, wprob, nullptr, gbuf[n - 1], packed_dgrad[n], B, od, oh, ow, blk[n - 1].cout, prob_cout, 1,
                                 bn ? raw[n - 1] : nullptr, bn ? stats[n - 1] : nullptr, bn ? slots_b[n - 1] : nullptr, bn ? nslots[n - 1] : 0,
                                 1, main_stream));
        gp[n - 1] = gbuf[n - 1];
        have[n - 1] = bn;
        MVS_TRY(wgrad(n, y[n - 1], glogits, nullptr, od, oh, ow, blk[n - 1].cout, prob_cout, 1, 0));
    }
    // ---- blocks, last to first ----
    for (int i = n - 1; i >= 0; --i) {
        const MvsUnetBlock& b = blk[i];
        MVS_REQUIRE(gp[i], MVS_ERR_SHAPE, "mvs_unet_bwd: block %d has no consumer", i);
        const float* gy = gp[i];
        out_dims(b, od, oh, ow);
        const long long vg = (long long)B * od * oh * ow;
        if (!have[i]) MVS_TRY(mvs_bn_bwd_reduce_slots(gy, raw[i], stats[i], 1, 1, vg, b.cout, slots_b[i], nslots[i], main_stream));
        MVS_TRY(mvs_bn_relu_bwd_slots(gy, raw[i], stats[i], slots_b[i], nslots[i], 1, 1, vg, b.cout, draw[i], dgamma[i], dbeta[i], main_stream));
        if (b.skip >= 0) {                               // y = relu(bn(raw)) + y_skip: the skip source receives gy as it is
            MVS_REQUIRE(!gp[b.skip], MVS_ERR_UNSUPPORTED,
                        "mvs_unet_bwd: block %d already has a gradient when the skip contribution of block %d arrives (needs an "
                        "explicit add: take the per-layer path)", b.skip, i);
            gp[b.skip] = gy;
        }
        const float* xin = b.src < 0 ? x : y[b.src];
        if (b.src >= 0) {
            const bool bn = last[b.src] == i;            // (i == consumer through its INPUT: prog[i].src == b.src by construction)
            const float* add = gp[b.src];
            MVS_REQUIRE(add != gbuf[b.src], MVS_ERR_UNSUPPORTED, "mvs_unet_bwd: block %d feeds two blocks as their input", b.src);
            if (b.transposed)
                MVS_TRY(mvs_convT3d_dgrad(draw[i], w[i], add, gbuf[b.src], packed_dgrad[i], B, b.d, b.h, b.w, b.cin, b.cout, b.stride,
                                          bn ? raw[b.src] : nullptr, bn ? stats[b.src] : nullptr, bn ? slots_b[b.src] : nullptr,
                                          bn ? nslots[b.src] : 0, 1, main_stream));
            else
                MVS_TRY(mvs_conv3d_dgrad(draw[i], w[i], add, gbuf[b.src], packed_dgrad[i], B, b.d, b.h, b.w, b.cin, b.cout, b.stride,
                                         bn ? raw[b.src] : nullptr, bn ? stats[b.src] : nullptr, bn ? slots_b[b.src] : nullptr,
                                         bn ? nslots[b.src] : 0, 1, main_stream));
            gp[b.src] = gbuf[b.src];
            have[b.src] = bn;
        } else if (gx) {
            MVS_REQUIRE(packed_dgrad[i], MVS_ERR_NULL, "mvs_unet_bwd: gx wanted but block %d has no input-gradient weight image", i);
            const float* add = gx_written ? gx : nullptr;
            if (b.transposed)
                MVS_TRY(mvs_convT3d_dgrad(draw[i], w[i], add, gx, packed_dgrad[i], B, b.d, b.h, b.w, b.cin, b.cout, b.stride, nullptr,
                                          nullptr, nullptr, 0, 1, main_stream));
            else
                MVS_TRY(mvs_conv3d_dgrad(draw[i], w[i], add, gx, packed_dgrad[i], B, b.d, b.h, b.w, b.cin, b.cout, b.stride, nullptr,
                                         nullptr, nullptr, 0, 1, main_stream));
            gx_written = true;
        }
        // the side stream forks where the weight gradient is enqueued: after the block's input gradient
        MVS_TRY(wgrad(i, xin, draw[i], &b, b.d, b.h, b.w, b.cin, b.cout, b.stride, b.transposed));
    }
    closer.done = true;
    if (side_used && join) fj.join(side_stream, main_stream);
    if (side_stream_used) *side_stream_used = side_used ? 1 : 0;     // 1 and no join: weight gradients are still running on the side stream
    return MVS_OK;
}
