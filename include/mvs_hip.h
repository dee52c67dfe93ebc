/* mvs_hip.h -- C ABI of libmvs_hip.so: the MI355X (gfx950) hot path of Self-Supervised-MVS.
 *
 * The reference (ToughStoneX/Self-Supervised-MVS) has NO native boundary on this path: the path is
 * a sequence of ATen calls inside two nn.Module.forward()s.  Each entry point below therefore cites
 * the reference Python it replaces (paths relative to the reference root); INTEGRATION.md shows the
 * ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller (e.g. PyTorch's caching allocator); the
 *    library never allocates, frees or retains them.  Outputs are fully overwritten unless stated.
 *  - Work is enqueued on `stream` only; no call synchronises the device.  Re-entrant: no global
 *    mutable state (the last-error string is thread local).
 *  - fp32 everywhere.  Feature maps are channels-last [B,H,W,C]; volumes channels-last [B,D,H,W,C]
 *    (== torch.channels_last / torch.channels_last_3d memory formats of NCHW / NCDHW tensors).
 *  - Return value: 0 on success, negative on error (MVS_ERR_*), message via mvs_last_error().
 */
#ifndef MVS_HIP_H
#define MVS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t; /* same opaque type as <hip/hip_runtime_api.h> */
#endif

#define MVS_OK 0
#define MVS_ERR_SHAPE (-1)
#define MVS_ERR_UNSUPPORTED (-2)
#define MVS_ERR_LAUNCH (-3)
#define MVS_ERR_NULL (-4)
#define MVS_MAX_SRC 10

int mvs_version(void);              /* 100 == 0.1.0 */
const char* mvs_last_error(void);   /* thread-local, valid until the next failing call */
int mvs_is_emulation(void);         /* 0 in the product library */
/* A/B knobs for measurements/tests (full-string keys; an unknown key is MVS_ERR_UNSUPPORTED): "sweep_fwd" 0 taps through
 * L1 | 1 LDS windows | 2,3 register-cached taps (default 3) | 6 quad-shared projection; "sweep_bwd" 0 per-wave windows |
 * 1 view pairs + LDS atomics; "fwd_dl" 0 | 1 per-plane depths staged in LDS (default) | 2 + in-block gather waits;
 * "bwd_gd" 2 (default: 2-plane gradient groups, 2 waves/SIMD) | 0 (1-plane groups, 3 waves/SIMD), "bwd_pf" 0 | 1 block
 * lookahead (1-2 views) | 2 one wave/SIMD (3-4 views), "bwd_dslab", "bwd_nowin", "bwd_cpt" (ignored); "nt", "tile_w", "dslab";
 * "conv_split", "conv_small", "conv_small_wgs", "tr2pw", "k8", "fs", "xcd"; "conv2d_s2_mfma", "wgrad2d_groups";
 * round 5: "conv_pers" / "conv_pers_min" / "conv_pers_groups" / "conv_pers_nw" (persistent LDS-DMA convolutions), "wgrad_pers",
 * "wgrad8_gs" 0 | 1 | 2 (conv0's weight gradient: 4x4x1-MFMA kernel | output-gradient-shifted form with eight | sixteen waves),
 * "wgrad8_groups" (its workgroups: 192 of 256 CUs on the side stream), "wgrad_groups".  The full table with ranges is in
 * csrc/plane_sweep.hip (mvs_set_tuning); the defaults are mirrored in _lib.DEFAULT_TUNING and checked by
 * tests/test_capi_symbols.py.  Process-wide, not part of the data path's contract. */
int mvs_set_tuning(const char* key, int value);
int mvs_get_tuning(const char* key, int* value);   /* the knob's current value (a freshly loaded library: its default) */

/* ---- K1/K2: homography warp + variance cost volume -------------------------------------------
 * Replaces homo_warping + the sum / sum-of-squares / variance chain:
 *   jdacs/models/module.py:105-140 (homo_warping), jdacs/models/mvsnet.py:120-136 (variance);
 *   jdacs-ms/models/modules.py:62-104 (homo_warping), :209-261 (proj_cost, per-pixel hypotheses),
 *   jdacs-ms/models/network.py:114-137 (coarse variance, in-place alias quirk => ms_alias=1).
 * ref, srcs[i]: [B,H,W,C] (N-1 source pointers, HOST array of device pointers); rot [B,N-1,9] and
 * trans [B,N-1,3] are rot=proj[:3,:3], trans=proj[:3,3] of src_proj @ inverse(ref_proj)
 * (module.py:116-118), computed by the caller.  depth: [B,D] or, if depth_is_per_pixel, [B,D,H,W].
 * align_corners: 0 = what F.grid_sample does on torch>=1.3 for the reference's call (default),
 * 1 = torch 1.1 behaviour.  C in {8,16,32}.  var_out: [B,D,H,W,C]. */
int mvs_plane_sweep_variance_fwd(const float* ref, const float* const* srcs, const float* rot, const float* trans,
                                 const float* depth, int depth_is_per_pixel, int B, int N, int C, int D, int H,
                                 int W, int align_corners, int ms_alias, float* var_out, hipStream_t stream);
/* The same volume stored in bf16 ([B,D,H,W,C], round to nearest even): the inference path of BASELINE configs[4] (MVSNet
 * N=7, 1600x1184, D=256: 1.94 GB instead of 3.9 GB).  C in {16,32}; 1, 2, 3, 4 or 6 source views.  The reference has no
 * reduced-precision path; this replaces the same lines as mvs_plane_sweep_variance_fwd. */
int mvs_plane_sweep_variance_fwd_bf16(const float* ref, const float* const* srcs, const float* rot, const float* trans,
                                      const float* depth, int depth_is_per_pixel, int B, int N, int C, int D, int H,
                                      int W, int align_corners, int ms_alias, void* var_out_bf16, hipStream_t stream);
/* rot [B,NS,9] / trans [B,NS,3] of src_proj[b,s] @ inverse(ref_proj[b]) for all NS source views in ONE launch (the caller-side
 * lines jdacs/models/module.py:116-118, jdacs-ms/models/modules.py:71-80 run once per view).  src_proj [B,NS,4,4], ref_proj
 * [B,4,4], row-major fp32; fp64 inside.  A singular ref_proj yields inf / nan (like torch.linalg.inv_ex), not an error. */
int mvs_relative_projection(const float* src_proj, const float* ref_proj, int B, int NS, float* rot, float* trans,
                            hipStream_t stream);
/* Backward of the above w.r.t. the feature maps (the reference builds the sampling grid under
 * no_grad, module.py:115).  grad_ref and grad_srcs[i] ([B,H,W,C]) must be ZERO-FILLED by the caller
 * (accumulated with atomics). */
int mvs_plane_sweep_variance_bwd(const float* grad_var, const float* ref, const float* const* srcs,
                                 const float* rot, const float* trans, const float* depth, int depth_is_per_pixel,
                                 int B, int N, int C, int D, int H, int W, int align_corners, int ms_alias,
                                 float* grad_ref, float* const* grad_srcs, hipStream_t stream);

/* homo_warping alone (jdacs/models/module.py:105-140): warped volume [B,D,H,W,C] of ONE source view
 * and its backward (grad_src zero-filled by the caller). */
int mvs_homo_warp_fwd(const float* src, const float* rot, const float* trans, const float* depth,
                      int depth_is_per_pixel, int B, int C, int D, int H, int W, int align_corners,
                      float* warped_out, hipStream_t stream);
int mvs_homo_warp_bwd(const float* grad_warped, const float* src, const float* rot, const float* trans,
                      const float* depth, int depth_is_per_pixel, int B, int C, int D, int H, int W,
                      int align_corners, float* grad_src, hipStream_t stream);

/* ---- K3-K8: 3-D convolutions of CostRegNet ------------------------------------------------------
 * Replace nn.Conv3d / nn.ConvTranspose3d (k=3, pad=1, bias=False; stride 1|2; transposed stride 2
 * has output_padding 1, stride 1 has 0) forward / input-gradient / weight-gradient:
 *   jdacs/models/module.py:35-42, jdacs/models/mvsnet.py:40-63; jdacs-ms/models/network.py:47-65.
 * (D,H,W) are ALWAYS the spatial dims of the forward op's input x.  Weights are the PyTorch
 * parameter tensors as they are: conv [Cout][Cin][3][3][3], transposed conv [Cin][Cout][3][3][3].
 * ws: workspace of mvs_conv3d_workspace_bytes(op,...) bytes, 16-byte aligned.
 * Forward epilogue (any subset): scale&&shift -> y*scale[c]+shift[c] (folded eval BatchNorm);
 * shift only -> y+shift[c] (bias of the prob layer, mvsnet.py:63); relu; + skip (added AFTER the
 * ReLU, mvsnet.py:70-72); stat_slots != NULL -> per-channel (sum, sum of squares) of the RAW conv output are
 * added (fp64 atomics) into row (workgroup mod nslots) of stat_slots [nslots][2][Cout] for train-mode BatchNorm
 * (module.py:39); the caller zeroes the rows, nslots is a power of two (mvs_bn_slots(Cout)); consumer: mvs_bn_relu_fwd_slots.
 * ws_packed = 1: ws already holds this op's weight image (mvs_conv3d_pack_weights[_batch] for the same op and shape). */
enum { MVS_OP_CONV_FWD = 0, MVS_OP_CONV_DGRAD = 1, MVS_OP_CONV_WGRAD = 2,
       MVS_OP_CONVT_FWD = 3, MVS_OP_CONVT_DGRAD = 4, MVS_OP_CONVT_WGRAD = 5 };
long long mvs_conv3d_workspace_bytes(int op, int B, int D, int H, int W, int Cin, int Cout, int stride);
/* Write the MFMA-fragment weight image of a forward / input-gradient op into ws ahead of time -- one launch for one op, or ONE
 * launch for a whole list (ops[n], w[n], ws[n], shapes[n][7] = B, D, H, W, Cin, Cout, stride): the regulariser packs all of its
 * layers once per training step instead of once in front of every convolution. */
int mvs_conv3d_pack_weights(int op, const float* w, float* ws, int B, int D, int H, int W, int Cin, int Cout, int stride,
                            hipStream_t stream);
int mvs_conv3d_pack_weights_batch(int n, const int* ops, const float* const* w, float* const* ws, const int* shapes,
                                  hipStream_t stream);
int mvs_conv3d_fwd(const float* x, const float* w, float* y, float* ws, int B, int D, int H, int W, int Cin, int Cout,
                   int stride, const float* scale, const float* shift, const float* skip, int relu,
                   double* stat_slots, int nslots, int ws_packed, hipStream_t stream);
/* Input gradients: gx = d conv / dx (+ add: [B,D,H,W,Cin] like gx, or NULL).  A tensor with two consumers -- the U-Net skip
 * connections, jdacs/models/mvsnet.py:70-72, jdacs-ms/models/network.py:71-72 -- receives its second gradient contribution in the
 * epilogue of the kernel that computes the first, instead of autograd's separate add pass over both.
 * bn_raw != NULL (like gx): x came out of a BatchNorm+ReLU block, x = relu(BatchNorm(bn_raw)) (+ skip), and gx is that block's
 * COMPLETE output gradient: the epilogue also adds the block's backward statistics (sum dyh, sum dyh*xhat per channel; dyh = gx where
 * the ReLU was active) into bn_slots [nslots][2][Cin] (fp64), from bn_stats [4][Cin] = mean, invstd, scale, shift of the block
 * (backward of module.py:35-42) -- no separate reduction pass over (gx, bn_raw); consumer: mvs_bn_relu_bwd_slots. */
int mvs_conv3d_dgrad(const float* gy, const float* w, const float* add, float* gx, float* ws, int B, int D, int H, int W, int Cin,
                     int Cout, int stride, const float* bn_raw, const float* bn_stats, double* bn_slots, int nslots, int ws_packed,
                     hipStream_t stream);
int mvs_conv3d_wgrad(const float* x, const float* gy, float* gw, float* ws, int B, int D, int H, int W, int Cin,
                     int Cout, int stride, hipStream_t stream);
int mvs_convT3d_fwd(const float* x, const float* w, float* y, float* ws, int B, int D, int H, int W, int Cin,
                    int Cout, int stride, const float* scale, const float* shift, const float* skip, int relu,
                    double* stat_slots, int nslots, int ws_packed, hipStream_t stream);
int mvs_convT3d_dgrad(const float* gy, const float* w, const float* add, float* gx, float* ws, int B, int D, int H, int W, int Cin,
                      int Cout, int stride, const float* bn_raw, const float* bn_stats, double* bn_slots, int nslots, int ws_packed,
                      hipStream_t stream);
int mvs_convT3d_wgrad(const float* x, const float* gy, float* gw, float* ws, int B, int D, int H, int W, int Cin,
                      int Cout, int stride, hipStream_t stream);

/* ---- One call per PASS of a cost-volume regulariser ------------------------------------------------------------------
 * Replace the whole of CostRegNet.forward (jdacs/models/mvsnet.py:66-74, jdacs-ms/models/network.py:67-74) in train mode and
 * its autograd backward: the same kernels as the per-layer entry points above, in the same order, enqueued from C (the
 * per-layer calls cost the launch thread ~1.7 ms per training step from Python).  Nothing is allocated: the caller supplies
 * every tensor (channels-last-3d fp32; sizes follow from the block table).
 * Block i: input = output of block src (-1: x), conv / transposed conv (k3 p1, bias-free; mvs_conv3d_fwd conventions, weight
 * image packed[i] already written by mvs_conv3d_pack_weights_batch) -> BatchNorm(train, statistic slots slots[i] zeroed by the
 * caller, running statistics updated) -> ReLU, + output of block skip (-1: none) AFTER the ReLU (mvsnet.py:70-72).
 * (cin, cout, d, h, w): channels and the spatial dims of the block's INPUT.  The closing prob layer (mvsnet.py:63): stride-1
 * convolution of block n-1's output with bias bprob, prob_cout output channels, workspace ws_prob (mvs_conv3d_workspace_bytes).
 * mvs_unet_fwd outputs: raw[i] (pre-BatchNorm), y[i] (block output), stats[i] [4][cout] (mean, invstd, scale, shift), logits.
 * mvs_unet_bwd: glogits -> gx (or NULL), gw[n+1] (weight gradients in the parameters' layouts; entry n = prob; a NULL entry
 *   is skipped), dgamma[i], dbeta[i]; work space: gbuf[i], draw[i] (each like y[i]), wgrad_ws[n+1] (mvs_conv3d_workspace_bytes
 *   of the WGRAD ops), slots_b[i] (zeroed backward statistic slots), packed_dgrad[n+1] (input-gradient weight images; entry i
 *   may be NULL when block i reads x and gx is NULL).  side_stream != NULL and != main_stream: every weight gradient is
 *   enqueued there behind a HIP event recorded on main_stream after the block's input gradient; join != 0: main_stream waits
 *   for side_stream at the end; *side_stream_used (may be NULL) reports whether anything went to side_stream.
 *   MVS_ERR_UNSUPPORTED for a program that needs an explicit gradient add (a block with two consumers through their INPUT, or
 *   a skip contribution that is not the first): neither reference network is one. */
#define MVS_UNET_MAX_BLOCKS 32
typedef struct MvsUnetBlock {
    int transposed, stride, src, skip;
    float eps, momentum;
    int cin, cout, d, h, w;
} MvsUnetBlock;
int mvs_unet_fwd(int n, const MvsUnetBlock* blocks, int B, const float* x, const float* const* w, const float* const* gamma,
                 const float* const* beta, float* const* running_mean, float* const* running_var, float* const* packed,
                 float* const* raw, float* const* y, float* const* stats, double* const* slots, const int* nslots,
                 const float* wprob, const float* bprob, int prob_cout, float* ws_prob, float* logits, hipStream_t stream);
int mvs_unet_bwd(int n, const MvsUnetBlock* blocks, int B, const float* x, const float* const* w, const float* wprob, int prob_cout,
                 const float* const* y, const float* const* raw, const float* const* stats, double* const* slots_b, const int* nslots,
                 float* const* packed_dgrad, const float* glogits, float* const* gbuf, float* const* draw, float* gx,
                 float* const* gw, float* const* wgrad_ws, float* const* dgamma, float* const* dbeta, hipStream_t main_stream,
                 hipStream_t side_stream, int join, int* side_stream_used);

/* Measurement hook of mvs_unet_bwd (process-wide, like mvs_set_tuning): HIP events around the weight gradient of ONE block
 * (0..n-1, n = the prob layer; < 0: off) on the stream it runs on; mvs_unet_time_read waits for them, writes the durations in
 * ms (launch order, at most max_n) and returns how many there were. */
int mvs_unet_time_wgrad(int block);
int mvs_unet_time_read(float* ms, int max_n);

/* ---- bf16-storage inference path of CostRegNet (BASELINE configs[4]) ------------------------------------------------
 * The same layers (jdacs/models/mvsnet.py:40-63) evaluated as in jdacs/eval.py:143 (eval mode, no_grad) with activations
 * stored in bf16 and fp32 accumulation (v_mfma_f32_16x16x32_bf16).  x, skip: bf16 [B,D,H,W,C]; w: the fp32 parameter;
 * y: bf16, or fp32 when out_is_f32 (the probability layer's logits).  transposed: stride 2 only.  Epilogue as
 * mvs_conv3d_fwd without statistics.  Supported (Cin -> Cout): the network's layer shapes -- stride 1: 32->8, 8->8, 16->8,
 * 16->16, 32->32, 64->64, 8->1, 16->1; stride 2: 8->16, 16->32, 32->64; transposed: 64->32, 32->16, 16->8.
 * ws: mvs_conv3d_bf16_workspace_bytes(...) bytes (packed bf16 weight image), 16-byte aligned. */
long long mvs_conv3d_bf16_workspace_bytes(int Cin, int Cout, int stride, int transposed);
int mvs_conv3d_bf16_fwd(const void* x_bf16, const float* w, void* y, void* ws, int B, int D, int H, int W, int Cin, int Cout,
                        int stride, int transposed, const float* scale, const float* shift, const void* skip_bf16, int relu,
                        int out_is_f32, hipStream_t stream);
int mvs_cast_f32_bf16(const float* x, void* y_bf16, long long n, hipStream_t stream);   /* n % 4 == 0 */

/* ---- BatchNorm3d / BatchNorm2d (+ReLU, + post-ReLU skip add) on channels-last [V][C] -----------
 * Replace nn.BatchNorm3d + F.relu of ConvBnReLU3D (module.py:35-42) and of the deconv blocks (mvsnet.py:48-61), and
 * nn.BatchNorm2d + F.relu of the 2-D ConvBnReLU (module.py:15-22).  C in {4,8,16,32,64}.
 * Train mode works on "statistic slots": nslots rows [2][C] of fp64 accumulators per statistics group, zeroed by the caller,
 * into which the PRODUCER of the tensor adds per-workgroup sums -- a convolution epilogue (mvs_conv3d_fwd / mvs_convT3d_fwd /
 * mvs_conv2d_fwd_stats; backward: mvs_conv3d_dgrad / mvs_convT3d_dgrad with bn_raw) or the stand-alone passes below -- and which
 * the APPLY kernel finishes in the prologue of every workgroup (no finalize launch in between).
 * G statistics groups of Vg contiguous rows each share the affine parameters; the running statistics are updated group after
 * group == G successive BatchNorm calls: the N views of a sample go through the shared-weight feature extractor as one batch
 * (jdacs/models/mvsnet.py:115) with the reference's per-view statistics.  slots [G][nslots][2][C]; stats [G][4][C] = mean, invstd
 * (biased variance, eps), scale = gamma*invstd, shift = beta - mean*scale (written by the forward, read by the backward). */
int mvs_bn_slots(int C);   /* recommended nslots (power of two; 16 KB of accumulators per group) */
int mvs_bn_stats_slots(const float* x, int G, long long Vg, int C, double* slots, int nslots, hipStream_t stream);
/* y = relu?(BatchNorm_train(x)) (+ skip); running_mean / running_var (both or neither NULL) updated with `momentum` and the
 * unbiased variance (PyTorch defaults) */
int mvs_bn_relu_fwd_slots(const float* x, const double* slots, int nslots, int G, long long Vg, int C, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                          const float* skip, int relu, float* stats, float* y, hipStream_t stream);
/* backward statistics (sum dyh, sum dyh*xhat) of dy = grad w.r.t. relu?(bn(x)) into slots, when no input-gradient epilogue did */
int mvs_bn_bwd_reduce_slots(const float* dy, const float* x, const float* stats, int relu, int G, long long Vg, int C,
                            double* slots, int nslots, hipStream_t stream);
/* dx [G*Vg][C] = BatchNorm+ReLU backward of dy given the slots; dgamma [C], dbeta [C] (summed over the groups; may be NULL) */
int mvs_bn_relu_bwd_slots(const float* dy, const float* x, const float* stats, const double* slots, int nslots, int relu, int G,
                          long long Vg, int C, float* dx, float* dgamma, float* dbeta, hipStream_t stream);
int mvs_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int C, float* scale, float* shift, hipStream_t stream);
/* y = relu?(x*scale+shift) (+ skip) */
int mvs_bn_relu_fwd(const float* x, const float* scale, const float* shift, const float* skip, int relu, long long V,
                    int C, float* y, hipStream_t stream);

/* ---- mvsnet_loss (jdacs/models/mvsnet.py:164-166): mean smooth-L1 (beta 1) of est - gt over the n pixels with mask > 0.5 ----
 * forward: out[0] = loss (nan for an empty mask, like the reference's mean over an empty selection), out[1] = pixel count;
 * backward: gest[i] = [mask > 0.5] * clamp(est - gt, -1, 1) * gloss[0] / out[1].  Two launches instead of ~14. */
int mvs_masked_smooth_l1_fwd(const float* est, const float* gt, const float* mask, long long n, float* out, hipStream_t stream);
int mvs_masked_smooth_l1_bwd(const float* est, const float* gt, const float* mask, const float* fwd_out, const float* gloss,
                             long long n, float* gest, hipStream_t stream);

/* ---- K9/K10: softmax over depth + soft-argmin regression + photometric confidence --------------
 * Replace F.softmax(dim=1) + depth_regression + the pad/avg_pool3d/gather confidence:
 *   jdacs/models/mvsnet.py:141-151, jdacs/models/module.py:145-148;
 *   jdacs-ms/models/network.py:147-149,173-189, jdacs-ms/models/modules.py:324-331.
 * logits [B,D,H,W]; depth [B,D] or [B,D,H,W]; outputs [B,H,W].  save_max/save_sum (may be NULL in
 * inference) are the per-pixel softmax max and denominator the backward needs. */
int mvs_softargmin_conf_fwd(const float* logits, const float* depth, int depth_is_per_pixel, int B, int D, int H,
                            int W, float* out_depth, float* out_conf, float* save_max, float* save_sum,
                            hipStream_t stream);
int mvs_softargmin_conf_bwd(const float* grad_depth, const float* logits, const float* depth, int depth_is_per_pixel,
                            const float* out_depth, const float* save_max, const float* save_sum, int B, int D, int H,
                            int W, float* grad_logits, hipStream_t stream);

/* ---- the self-supervised loss on the path's output ------------------------------------------------
 * Replaces UnSupLoss.forward and its autograd graph (jdacs/losses/unsup_loss.py:24-83; inverse_warping
 * jdacs/losses/homography.py:186-351; compute_reconstr_loss / SSIM / depth_smoothness jdacs/losses/modules.py:17-90).
 * ref, views[v]: quarter-resolution NHWC images [B,H,W,3] (the caller does F.interpolate(0.25, bilinear) + permute,
 * unsup_loss.py:36-37,53-54); kinv [B,9] = K_ref^-1; proj [B,V,12] = K_ref.[R_rel | t_rel] row major
 * (homography.py:200-236: R_rel = R_v R_ref^T, t_rel = t_v - R_rel t_ref; the REFERENCE intrinsics project, as there);
 * depth [B,H,W]; V = number of source views, 3 <= V <= 10 (the top-3 selection needs three).
 * out[4] (device): total = 12 reconstr + 6 ssim + 0.18 smooth, then the three terms.  ws: caller-owned scratch of
 * mvs_unsup_loss_workspace_floats() floats; the backward reads what the forward left there.  grad_out: device scalar.
 * Only the depth map receives a gradient. */
long long mvs_unsup_loss_workspace_floats(int B, int V, int H, int W);
int mvs_unsup_loss_fwd(const float* ref, const float* const* views, const float* kinv, const float* proj,
                       const float* depth, int B, int V, int H, int W, float smooth_lambda, float* ws, float* out,
                       hipStream_t stream);
int mvs_unsup_loss_bwd(const float* ref, const float* const* views, const float* kinv, const float* proj,
                       const float* depth, int B, int V, int H, int W, float smooth_lambda, float* ws,
                       const float* grad_out, float* grad_depth, hipStream_t stream);

/* ---- stage glue of CVP-MVSNet --------------------------------------------------------------------
 * Replaces calDepthHypo (jdacs-ms/models/modules.py:107-206): hypos[b,k] = ref_depths[b] + (k - 4) * interval_b, k = 0..7,
 * interval_b = mean over the pixels of |depth change that moves the projection into source view 0 by one pixel along the
 * epipolar line| (fp64 inside, like the reference).  mats [B,30] fp64: K_ref^-1 (9), K_src (E_src E_ref^-1)[:3,:] (12),
 * (K_ref R_ref)(K_src R_src)^-1 (9), prepared by the caller from the camera matrices.  ws: fp64 scratch. */
long long mvs_depth_hypo_workspace_doubles(int B, int H, int W);
int mvs_depth_hypo(const float* ref_depths, const double* mats, int B, int H, int W, double* ws, float* hypos,
                   hipStream_t stream);

/* ---- SURVEY.md 8(f)-3, first cut (not used by default): the feature extractors' 2-D convolutions ---------------------
 * nn.Conv2d of jdacs/models/module.py:15-22 as used by FeatureNet (jdacs/models/mvsnet.py:17-34): 3x3 stride 1 pad 1 and
 * 5x5 stride 2 pad 2, 1..32 channels, channels-last images x [N,H,W,Cin], w [Cout][Cin][ks][ks], y [N,Ho,Wo,Cout].
 * op for the workspace query: 0 forward, 1 input gradient, 2 weight gradient. */
long long mvs_conv2d_workspace_floats(int op, int N, int H, int W, int Cin, int Cout, int ks, int stride);
int mvs_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin,
                   int Cout, int ks, int stride, hipStream_t stream);
/* conv2d + bias + LeakyReLU(negative_slope) in one pass: the `conv` block of the CVP feature pyramid
 * (jdacs-ms/models/modules.py:15-19, network.py:16-41; widths 3/16/32/64).  Channels: 1..32 or exactly 64. */
/* conv2d forward (no bias) that also adds BatchNorm's statistics of its output into slots [G][nslots][2][Cout] (fp64, zeroed by
 * the caller): the N images are G statistics groups of N/G consecutive images; consumer: mvs_bn_relu_fwd_slots.
 * The convolution + statistics half of ConvBnReLU in training (jdacs/models/module.py:15-22). */
int mvs_conv2d_fwd_stats(const float* x, const float* w, float* y, float* ws, double* slots, int nslots, int G, int N, int H,
                         int W, int Cin, int Cout, int ks, int stride, int ws_packed, hipStream_t stream);
/* Forward weight images of n layers in ONE launch (then mvs_conv2d_fwd_stats(..., ws_packed = 1) skips its own packing launch):
 * w[n] parameter tensors [Cout][Cin][ks][ks] -- or channels-last in memory ([Cout][ks][ks][Cin]) where w_channels_last[i] --,
 * ws[n] workspaces of mvs_conv2d_workspace_floats(0, ...) floats, shapes[n][4] = Cin, Cout, ks, stride. */
int mvs_conv2d_pack_weights_batch(int n, const float* const* w, float* const* ws, const int* shapes, const int* w_channels_last,
                                  hipStream_t stream);
int mvs_conv2d_lrelu_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin,
                         int Cout, int ks, int stride, float negative_slope, hipStream_t stream);
int mvs_conv2d_dgrad(const float* gy, const float* w, float* gx, float* ws, int N, int H, int W, int Cin, int Cout, int ks,
                     int stride, hipStream_t stream);
/* mvs_conv2d_fwd with the parameter tensor channels-last in memory ([Cout][ks][ks][Cin]) when w_channels_last: read in place. */
int mvs_conv2d_fwd_wl(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin, int Cout,
                      int ks, int stride, int w_channels_last, hipStream_t stream);
/* mvs_conv2d_dgrad with the parameter tensor channels-last in memory ([Cout][ks][ks][Cin]) when w_channels_last: read in place. */
int mvs_conv2d_dgrad_wl(const float* gy, const float* w, float* gx, float* ws, int N, int H, int W, int Cin, int Cout, int ks,
                        int stride, int w_channels_last, hipStream_t stream);
int mvs_conv2d_wgrad(const float* x, const float* gy, float* gw, float* ws, int N, int H, int W, int Cin, int Cout, int ks,
                     int stride, hipStream_t stream);
/* Weight gradients of up to 8 layers in ONE launch + one reduction launch (the training extractor's backward pass: replaces the
 * weight half of the convolution_backward nodes autograd builds for jdacs/models/module.py:18 `self.conv` of every ConvBnReLU of
 * jdacs/models/mvsnet.py:21-31 and for mvsnet.py:32 `self.feature`).  shapes[n][8] = N, H, W, Cin, Cout, ks, stride,
 * w_channels_last; x[i] [N,H,W,Cin], gy[i] [N,Ho,Wo,Cout] (NHWC, pad ks/2); gw[i] is written as [Cout][Cin][ks][ks], or as
 * [Cout][ks][ks][Cin] when w_channels_last (the memory of a channels_last nn.Conv2d weight).  Served layers: 3x3 stride 1 with
 * 3 / 8 / 16 -> <= 16 or 32 -> <= 32 channels, 5x5 stride 2 with 8 -> <= 16 or 16 -> <= 32, Cout % 4 == 0.
 * mvs_conv2d_wgrad_batch_workspace_floats: size of ws for these shapes, or -1 if a layer is not served. */
long long mvs_conv2d_wgrad_batch_workspace_floats(int n, const int* shapes);
/* Consumer-side BatchNorm of the training extractor (the host path's default since round 4; MVS_FEATURE_FUSED_APPLY=0 restores the
 * apply passes): block i's `F.relu(self.bn(...))`
 * (jdacs/models/module.py:21-22) is applied by block i+1's convolution -- forward and weight gradient -- while it stages its
 * input, so block i has no apply pass and its normalised output exists nowhere in memory.
 * mvs_bn_finalize_slots: the statistic slots of a block -> stats [G][4][C] (mean, invstd, scale, shift) + running statistics
 *   (the prologue of mvs_bn_relu_fwd_slots without its elementwise pass).
 * mvs_conv2d_fwd_stats_xf: mvs_conv2d_fwd_stats with x = the RAW output of the block in front and in_stats = that block's stats.
 * mvs_conv2d_wgrad_batch_xf: mvs_conv2d_wgrad_batch where x[i] is raw when x_stats[i] is not null (groups of imgs_per_group images). */
int mvs_bn_finalize_slots(const double* slots, int nslots, int G, long long Vg, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* stats, hipStream_t stream);
int mvs_conv2d_fwd_stats_xf(const float* x, const float* in_stats, const float* w, float* y, float* ws, double* slots, int nslots,
                            int G, int N, int H, int W, int Cin, int Cout, int ks, int stride, int ws_packed, hipStream_t stream);
/* mvs_conv2d_dgrad_bnstats (opt-in, MVS_FEATURE_DGRAD_BNSTATS=1): mvs_conv2d_dgrad of a 3x3 stride-1 layer whose result is the
 * complete output gradient of the BatchNorm + ReLU block in front (raw output bn_raw [N,H,W,Cin], bn_stats [G][4][Cin]); the epilogue
 * adds that block's backward statistics into bn_slots [G][nslots][2][Cin], so mvs_bn_relu_bwd_slots needs no reduce pass. */
int mvs_conv2d_dgrad_bnstats(const float* gy, const float* w, float* gx, float* ws, int N, int H, int W, int Cin, int Cout, int ks,
                             const float* bn_raw, const float* bn_stats, double* bn_slots, int nslots, int G, hipStream_t stream);
int mvs_conv2d_wgrad_batch_xf(int n, const float* const* x, const float* const* x_stats, int imgs_per_group, const float* const* gy,
                              float* const* gw, float* ws, const int* shapes, hipStream_t stream);
int mvs_conv2d_wgrad_batch(int n, const float* const* x, const float* const* gy, float* const* gw, float* ws, const int* shapes,
                           hipStream_t stream);

/* ---- One C call per pass of the TRAINING feature extractor (round 6; csrc/feature_pass.cpp) ------------------------------------
 * FeatureNet (jdacs/models/mvsnet.py:17-34): n ConvBnReLU blocks (module.py:15-22; 3x3 stride 1 or 5x5 stride 2, bias-free) closed by a
 * 3x3 stride-1 convolution with bias (mvsnet.py:32), on N channels-last images that are G statistics groups of N/G consecutive images
 * (the views of a sample: mvsnet.py:115 calls the extractor once per view).  Replaces the 15 module calls of FeatureNet.forward and
 * the ~30 autograd nodes of its backward pass by two calls that enqueue the same kernels in the same order:
 * mvs_feature_fwd: one pack launch for all forward weight images (packed[i]: mvs_conv2d_workspace_floats(0, ...) floats each), then per
 *   block mvs_conv2d_fwd_stats(_xf) -> raw[i] [N,Ho,Wo,cout] + slots[i] (zeroed by the caller, [G][nslots[i]][2][cout] fp64), and
 *   mvs_bn_finalize_slots -> stats[i] [G][4][cout] (block i's BatchNorm + ReLU is applied by block i+1 while it stages its input);
 *   the LAST block gets mvs_bn_relu_fwd_slots -> y_last; then the closing convolution -> out [N,Ho,Wo,close_cout] (ws_close:
 *   mvs_conv2d_workspace_floats(0, ...)).  Running statistics are updated in place, group after group.
 * mvs_feature_bwd: gout -> gx (or NULL), gw[n+1] (every layer's weight gradient, in the parameter's own memory layout; entry n = the
 *   closing convolution), dgamma[i], dbeta[i].  Work space: gbuf[i] (gradient w.r.t. block i's output, like raw[i]), draw[i] (like
 *   raw[i]), slots_b[i] (zeroed), dgrad_ws (the largest mvs_conv2d_workspace_floats(1, ...) of the chain), wgrad_ws_main / _side
 *   (mvs_conv2d_wgrad_batch_workspace_floats of layers [0, early_from) / [early_from, n]).  0 < early_from < n and side_stream !=
 *   main_stream: the weight gradients of layers early_from .. n are enqueued on side_stream as soon as block early_from's raw-output
 *   gradient exists; main_stream waits for side_stream before the call returns (also on an error); *side_stream_used reports it. */
#define MVS_FEAT_MAX_BLOCKS 7
typedef struct MvsFeatBlock {
    int cin, cout, ks, stride;
    float eps, momentum;
    int w_channels_last;
    int h, w;                         /* spatial dims of the block's INPUT */
} MvsFeatBlock;
int mvs_feature_fwd(int n, const MvsFeatBlock* blocks, int N, int G, const float* x, const float* const* w, const float* const* gamma,
                    const float* const* beta, float* const* running_mean, float* const* running_var, float* const* packed,
                    float* const* raw, float* y_last, float* const* stats, double* const* slots, const int* nslots, const float* wclose,
                    const float* bclose, int close_cout, int close_w_channels_last, float* ws_close, float* out, hipStream_t stream);
int mvs_feature_bwd(int n, const MvsFeatBlock* blocks, int N, int G, const float* x, const float* const* w, const float* wclose,
                    int close_cout, int close_w_channels_last, const float* const* raw, const float* y_last, const float* const* stats,
                    double* const* slots_b, const int* nslots, const float* gout, float* const* gbuf, float* const* draw, float* gx,
                    float* dgrad_ws, float* const* gw, float* wgrad_ws_main, float* wgrad_ws_side, float* const* dgamma,
                    float* const* dbeta, int early_from, hipStream_t main_stream, hipStream_t side_stream, int* side_stream_used);

/* ---- geometric-consistency filter on the path's depth maps ---------------------------------------------
 * Replaces reproject_with_depth + check_geometric_consistency (jdacs/eval.py:169-224) for ALL source views of one
 * reference view, and the accumulation of filter_depth (eval.py:372-385): per pixel and source view, project with the
 * reference depth, look the source depth up (cv2.remap INTER_LINEAR semantics: 1/32-pixel fixed point, border 0), project
 * back, test |p' - p| < pix_thresh and |d' - d| / d < rel_thresh (1 and 0.01 in the reference).
 * depth_ref [H,W]; depth_srcs: HOST array of V device pointers [H,W]; mats: DEVICE array of 18 + 42 V doubles =
 * K_ref^-1 [9], K_ref [9], then per view E_src E_ref^-1 (rows 0-2) [12], K_src [9], K_src^-1 [9], E_ref E_src^-1 (rows 0-2)
 * [12], each formed in float32 like numpy does in the reference.  Outputs: count [H,W] int32 (geo_mask_sum), depth_sum
 * [H,W] fp32 (sum of the consistent reprojected depths); optional (NULL to skip) masks [V,H,W] uint8, reproj [V,H,W] fp32
 * (0 where inconsistent), xy_src [V,2,H,W] fp32 (x2d_src, y2d_src). */
int mvs_geo_consistency(const float* depth_ref, const float* const* depth_srcs, const double* mats, int V, int H, int W,
                        float pix_thresh, float rel_thresh, int* count, float* depth_sum, unsigned char* masks, float* reproj,
                        float* xy_src, hipStream_t stream);

/* ---- depth-map fusion (the `fusibile` CUDA program behind jdacs/fusion/depthfusion.py:366-386) ---------
 * Replaces the kernel `fusibile` (jdacs/fusion/fusibile/fusibile.cu:138-277) for ONE reference camera = one launch of the
 * host loop fusibile.cu:416-421: per pixel, back-project with its depth, project into every other view of `subset`,
 * accept the view when the disparities differ by less than depth_thresh and the normals by less than normal_thresh
 * (radians), average the accepted 3-D points / normals / colours, keep the point when at least num_consistent views agree.
 * normals_depths [V,H,W,4] (normal.xyz, depth): the texture of main.cpp:833-843; images [V,H,W,4] colour as float (or NULL);
 * cams [V,32] floats: P (3x4 row-major) [12], M_inv [9], P(:,3) [3], camera centre C [3], 5 unused  (what
 * cameraGeometryUtils.h:353-440 puts into Camera_cu); subset: DEVICE array of n_subset view ids; f = K(0,0).
 * Texture fetches restate CUDA's linear filtering (1.8 fixed-point weights, clamped addressing): csrc/fusibile.hip header.
 * out_points [H,W,12] = coord.xyz 0, normal.xyz 0, colour.xyz 0 -- all zeros where fewer views agree (fully overwritten). */
int mvs_fusibile_fuse(const float* normals_depths, const float* images, const float* cams, const int* subset, int n_subset,
                      int V, int H, int W, int ref_camera, float f, float depth_thresh, float normal_thresh,
                      int num_consistent, int save_texture, float* out_points, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVS_HIP_H */
