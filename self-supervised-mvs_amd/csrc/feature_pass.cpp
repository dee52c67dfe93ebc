// One C call per pass of the TRAINING 2-D feature extractor.
//
// The reference runs FeatureNet (jdacs/models/mvsnet.py:17-34: seven ConvBnReLU blocks -- 3x3 stride 1 / 5x5 stride 2, module.py:15-22 --
// closed by a plain 3x3 convolution with bias) as 15 nn.Module calls forward and ~30 autograd nodes backward.  The Python mirror
// (ops.FeatureExtractorFn) had made that ONE autograd node in round 4, but the node still issued ~17 C-ABI calls forward and ~30 backward
// from Python -- pointer extraction, ctypes marshalling, a torch allocation or three per call: ~1 ms of launch-thread time per training
// step for ~1 ms of kernels.  These two entry points are the same launch sequence (same kernels, same order, same streams) driven from C,
// in the node's default configuration:
//   * every convolution through csrc/conv2d.hip (forward with BatchNorm's statistics in its epilogue; input gradients; the one-launch
//     weight gradient of all layers),
//   * consumer-side BatchNorm: block i's relu(bn(.)) is applied by block i+1's convolution and weight gradient while they stage their
//     input; only a finalize launch per block, the LAST block's output is materialised for the closing convolution,
//   * backward: the weight gradients of the layers from `early_from` up are forked to `side` as soon as their output gradients exist
//     (behind a HIP event recorded on `main`), the rest follow on `main` at the end; `main` waits for `side` before the call returns.
// The caller hands over a table of blocks and tables of device pointers (its own allocations; nothing is allocated here).
// Every tensor is channels-last fp32 ([N,H,W,C]) as everywhere in this library.
#include <string.h>
#include "mvs_rt.h"

#define MVS_FEAT_MAX_BLOCKS 7
extern "C" {
struct MvsFeatBlock {                 // == include/mvs_hip.h
    int cin, cout, ks, stride;
    float eps, momentum;
    int w_channels_last;              // the parameter tensor is [Cout][ks][ks][Cin] in memory (a channels_last nn.Conv2d weight)
    int h, w;                         // spatial dims of the block's INPUT
};
int mvs_conv2d_fwd_wl(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin, int Cout, int ks,
                      int stride, int w_channels_last, hipStream_t stream);
int mvs_conv2d_fwd_stats(const float* x, const float* w, float* y, float* ws, double* slots, int nslots, int G, int N, int H, int W, int Cin,
                         int Cout, int ks, int stride, int ws_packed, hipStream_t stream);
int mvs_conv2d_fwd_stats_xf(const float* x, const float* in_stats, const float* w, float* y, float* ws, double* slots, int nslots, int G, int N,
                            int H, int W, int Cin, int Cout, int ks, int stride, int ws_packed, hipStream_t stream);
int mvs_conv2d_pack_weights_batch(int n, const float* const* w, float* const* ws, const int* shapes, const int* w_channels_last,
                                  hipStream_t stream);
int mvs_conv2d_dgrad_wl(const float* gy, const float* w, float* gx, float* ws, int N, int H, int W, int Cin, int Cout, int ks, int stride,
                        int w_channels_last, hipStream_t stream);
int mvs_conv2d_wgrad_batch_xf(int n, const float* const* x, const float* const* x_stats, int imgs_per_group, const float* const* gy,
                              float* const* gw, float* ws, const int* shapes, hipStream_t stream);
int mvs_conv2d_wgrad_batch(int n, const float* const* x, const float* const* gy, float* const* gw, float* ws, const int* shapes,
                           hipStream_t stream);
int mvs_bn_finalize_slots(const double* slots, int nslots, int G, long long Vg, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* stats, hipStream_t stream);
int mvs_bn_relu_fwd_slots(const float* x, const double* slots, int nslots, int G, long long Vg, int C, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, const float* skip, int relu, float* stats,
                          float* y, hipStream_t stream);
int mvs_bn_bwd_reduce_slots(const float* dy, const float* x, const float* stats, int relu, int G, long long Vg, int C, double* slots,
                            int nslots, hipStream_t stream);
int mvs_bn_relu_bwd_slots(const float* dy, const float* x, const float* stats, const double* slots, int nslots, int relu, int G, long long Vg,
                          int C, float* dx, float* dgamma, float* dbeta, hipStream_t stream);
}

#define MVS_TRY(call)               \
    do {                            \
        const int rc_ = (call);     \
        if (rc_ != MVS_OK) return rc_; \
    } while (0)

static void out_hw(const MvsFeatBlock& b, int& ho, int& wo) {
    if (b.stride == 1) { ho = b.h; wo = b.w; }
    else { ho = (b.h - 1) / 2 + 1; wo = (b.w - 1) / 2 + 1; }
}

static int check_blocks(int n, const MvsFeatBlock* blk, int N, int G, const char* what) {
    MVS_REQUIRE(n >= 2 && n <= MVS_FEAT_MAX_BLOCKS && blk, MVS_ERR_SHAPE, "%s: 2..%d blocks, got %d", what, MVS_FEAT_MAX_BLOCKS, n);
    MVS_REQUIRE(N >= 1 && G >= 1 && N % G == 0, MVS_ERR_SHAPE, "%s: %d images do not split into %d statistics groups", what, N, G);
    for (int i = 0; i < n; ++i) {
        MVS_REQUIRE((blk[i].ks == 3 && blk[i].stride == 1) || (blk[i].ks == 5 && blk[i].stride == 2), MVS_ERR_UNSUPPORTED,
                    "%s: block %d is %dx%d stride %d (3x3 stride 1 or 5x5 stride 2)", what, i, blk[i].ks, blk[i].ks, blk[i].stride);
        if (i > 0) {
            int ho, wo;
            out_hw(blk[i - 1], ho, wo);
            MVS_REQUIRE(blk[i].cin == blk[i - 1].cout && blk[i].h == ho && blk[i].w == wo, MVS_ERR_SHAPE,
                        "%s: block %d does not read block %d's output", what, i, i - 1);
        }
    }
    return MVS_OK;
}

extern "C" int mvs_feature_fwd(int n, const MvsFeatBlock* blk, int N, int G, const float* x, const float* const* w, const float* const* gamma,
                               const float* const* beta, float* const* running_mean, float* const* running_var, float* const* packed,
                               float* const* raw, float* y_last, float* const* stats, double* const* slots, const int* nslots,
                               const float* wclose, const float* bclose, int close_cout, int close_w_channels_last, float* ws_close, float* out,
                               hipStream_t stream) {
    MVS_TRY(check_blocks(n, blk, N, G, "mvs_feature_fwd"));
    MVS_REQUIRE(x && w && gamma && beta && running_mean && running_var && packed && raw && y_last && stats && slots && nslots && wclose &&
                ws_close && out, MVS_ERR_NULL, "mvs_feature_fwd: null pointer argument");
    int shapes[4 * MVS_FEAT_MAX_BLOCKS], wcl[MVS_FEAT_MAX_BLOCKS];
    for (int i = 0; i < n; ++i) {
        shapes[4 * i] = blk[i].cin; shapes[4 * i + 1] = blk[i].cout; shapes[4 * i + 2] = blk[i].ks; shapes[4 * i + 3] = blk[i].stride;
        wcl[i] = blk[i].w_channels_last;
    }
    MVS_TRY(mvs_conv2d_pack_weights_batch(n, w, packed, shapes, wcl, stream));      // the forward weight images of all blocks: ONE launch
    int ho = 0, wo = 0;
    for (int i = 0; i < n; ++i) {
        const MvsFeatBlock& b = blk[i];
        if (i == 0)
            MVS_TRY(mvs_conv2d_fwd_stats(x, w[i], raw[i], packed[i], slots[i], nslots[i], G, N, b.h, b.w, b.cin, b.cout, b.ks, b.stride, 1, stream));
        else     // the block in front is applied while this convolution stages its input
            MVS_TRY(mvs_conv2d_fwd_stats_xf(raw[i - 1], stats[i - 1], w[i], raw[i], packed[i], slots[i], nslots[i], G, N, b.h, b.w, b.cin, b.cout,
                                            b.ks, b.stride, 1, stream));
        out_hw(b, ho, wo);
        const long long vg = (long long)(N / G) * ho * wo;
        if (i < n - 1)
            MVS_TRY(mvs_bn_finalize_slots(slots[i], nslots[i], G, vg, b.cout, gamma[i], beta[i], b.eps, b.momentum, running_mean[i],
                                          running_var[i], stats[i], stream));
        else
            MVS_TRY(mvs_bn_relu_fwd_slots(raw[i], slots[i], nslots[i], G, vg, b.cout, gamma[i], beta[i], b.eps, b.momentum, running_mean[i],
                                          running_var[i], nullptr, 1, stats[i], y_last, stream));
    }
    return mvs_conv2d_fwd_wl(y_last, wclose, bclose, out, ws_close, N, ho, wo, blk[n - 1].cout, close_cout, 3, 1, close_w_channels_last, stream);
}

#if defined(MVS_CPU_EMUL)
static void fork_event(hipStream_t, hipStream_t) {}
#else
static void fork_event(hipStream_t from, hipStream_t to) {
    static thread_local hipEvent_t ev = nullptr;
    if (!ev) (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    (void)hipEventRecord(ev, from);
    (void)hipStreamWaitEvent(to, ev, 0);
}
#endif

// layers lo .. hi-1 of the chain (layer n = the closing convolution) as one weight-gradient batch on `st`
static int wgrad_range(int n, const MvsFeatBlock* blk, int N, int G, int close_cout, int close_wcl, int lo, int hi, const float* x,
                       const float* const* raw, const float* y_last, const float* const* stats, float* const* draw, const float* gout,
                       float* const* gw, float* ws, hipStream_t st) {
    const float* xs[MVS_FEAT_MAX_BLOCKS + 1];
    const float* xst[MVS_FEAT_MAX_BLOCKS + 1];
    const float* gys[MVS_FEAT_MAX_BLOCKS + 1];
    float* gws[MVS_FEAT_MAX_BLOCKS + 1];
    int shapes[8 * (MVS_FEAT_MAX_BLOCKS + 1)];
    bool any_stats = false;
    int cnt = 0;
    for (int j = lo; j < hi; ++j, ++cnt) {
        // the input of layer j: the images (j = 0), the RAW output of block j-1 normalised while staged (0 < j < n), the materialised
        // output of the last block (the closing convolution)
        xs[cnt] = j == 0 ? x : (j == n ? y_last : raw[j - 1]);
        xst[cnt] = (j > 0 && j < n) ? stats[j - 1] : nullptr;
        any_stats = any_stats || xst[cnt] != nullptr;
        gys[cnt] = j == n ? gout : draw[j];
        gws[cnt] = gw[j];
        int* s = shapes + 8 * cnt;
        if (j < n) {
            s[0] = N; s[1] = blk[j].h; s[2] = blk[j].w; s[3] = blk[j].cin; s[4] = blk[j].cout; s[5] = blk[j].ks; s[6] = blk[j].stride;
            s[7] = blk[j].w_channels_last;
        } else {
            int ho, wo;
            out_hw(blk[n - 1], ho, wo);
            s[0] = N; s[1] = ho; s[2] = wo; s[3] = blk[n - 1].cout; s[4] = close_cout; s[5] = 3; s[6] = 1; s[7] = close_wcl;
        }
    }
    if (cnt == 0) return MVS_OK;
    if (any_stats) return mvs_conv2d_wgrad_batch_xf(cnt, xs, xst, N / G, gys, gws, ws, shapes, st);
    return mvs_conv2d_wgrad_batch(cnt, xs, gys, gws, ws, shapes, st);
}

extern "C" int mvs_feature_bwd(int n, const MvsFeatBlock* blk, int N, int G, const float* x, const float* const* w, const float* wclose,
                               int close_cout, int close_w_channels_last, const float* const* raw, const float* y_last,
                               const float* const* stats, double* const* slots_b, const int* nslots, const float* gout, float* const* gbuf,
                               float* const* draw, float* gx, float* dgrad_ws, float* const* gw, float* wgrad_ws_main, float* wgrad_ws_side,
                               float* const* dgamma, float* const* dbeta, int early_from, hipStream_t main_stream, hipStream_t side_stream,
                               int* side_stream_used) {
    MVS_TRY(check_blocks(n, blk, N, G, "mvs_feature_bwd"));
    MVS_REQUIRE(x && w && wclose && raw && y_last && stats && slots_b && nslots && gout && gbuf && draw && dgrad_ws && gw && wgrad_ws_main &&
                dgamma && dbeta, MVS_ERR_NULL, "mvs_feature_bwd: null pointer argument");
    for (int j = 0; j <= n; ++j) MVS_REQUIRE(gw[j], MVS_ERR_NULL, "mvs_feature_bwd: every layer's weight gradient is computed (gw[%d] is null)", j);
    const bool async = side_stream != nullptr && side_stream != main_stream && early_from > 0 && early_from < n && wgrad_ws_side != nullptr;
    bool side_used = false;
    struct Closer {       // a failure after the fork must not leave work on the side stream unjoined
        bool& used; hipStream_t side, main; int* out;
        ~Closer() { if (used) fork_event(side, main); if (out) *out = used ? 1 : 0; }
    } closer{side_used, side_stream, main_stream, side_stream_used};
    int ho, wo;
    out_hw(blk[n - 1], ho, wo);
    // the closing convolution's input gradient = the output gradient of the last block
    MVS_TRY(mvs_conv2d_dgrad_wl(gout, wclose, gbuf[n - 1], dgrad_ws, N, ho, wo, blk[n - 1].cout, close_cout, 3, 1, close_w_channels_last,
                                main_stream));
    for (int i = n - 1; i >= 0; --i) {
        const MvsFeatBlock& b = blk[i];
        out_hw(b, ho, wo);
        const long long vg = (long long)(N / G) * ho * wo;
        MVS_TRY(mvs_bn_bwd_reduce_slots(gbuf[i], raw[i], stats[i], 1, G, vg, b.cout, slots_b[i], nslots[i], main_stream));
        MVS_TRY(mvs_bn_relu_bwd_slots(gbuf[i], raw[i], stats[i], slots_b[i], nslots[i], 1, G, vg, b.cout, draw[i], dgamma[i], dbeta[i],
                                      main_stream));
        if (async && i == early_from) {
            fork_event(main_stream, side_stream);         // draw[early_from .. n-1] and gout exist on the main stream
            side_used = true;
            MVS_TRY(wgrad_range(n, blk, N, G, close_cout, close_w_channels_last, early_from, n + 1, x, raw, y_last, stats, draw, gout, gw,
                                wgrad_ws_side, side_stream));
        }
        if (i > 0)
            MVS_TRY(mvs_conv2d_dgrad_wl(draw[i], w[i], gbuf[i - 1], dgrad_ws, N, b.h, b.w, b.cin, b.cout, b.ks, b.stride, b.w_channels_last,
                                        main_stream));
        else if (gx)
            MVS_TRY(mvs_conv2d_dgrad_wl(draw[i], w[i], gx, dgrad_ws, N, b.h, b.w, b.cin, b.cout, b.ks, b.stride, b.w_channels_last, main_stream));
    }
    MVS_TRY(wgrad_range(n, blk, N, G, close_cout, close_w_channels_last, 0, async ? early_from : n + 1, x, raw, y_last, stats, draw, gout, gw,
                        wgrad_ws_main, main_stream));
    return MVS_OK;          // (the Closer joins side -> main and reports)
}
