// K9 / K10: softmax over depth + soft-argmin depth regression + photometric confidence, one kernel.
//
// Replaces F.softmax(dim=1) + depth_regression + pad/avg_pool3d/gather confidence
// (jdacs/models/mvsnet.py:141-151, jdacs/models/module.py:145-148;
//  jdacs-ms/models/network.py:147-149,173-189, jdacs-ms/models/modules.py:324-331).
//
// logits [B,D,H,W] (the Cout=1 output of the regulariser).  A wavefront owns 64/DS consecutive
// pixels; the D planes of a pixel are split over DS lane groups (lane = slice*PX + pixel) and the
// max / sum / expectation reductions are finished with wavefront shuffles (xor PX, 2*PX).
// Reads of one plane by one lane group are contiguous (PX*4 bytes).
#include "mvs_rt.h"

struct SoftArgs {
    const float* logits;   // [B,D,H,W]
    const float* depth;    // [B,D] or [B,D,H,W]
    float* out_depth;      // [B,H,W]
    float* out_conf;       // [B,H,W]
    float* out_max;        // [B,H,W] saved for backward
    float* out_sum;        // [B,H,W] saved for backward
    const float* gdepth;   // bwd: [B,H,W]
    float* glogits;        // bwd: [B,D,H,W]
    int B, D, HW, per_pixel;
};

template <int DS>
__device__ __forceinline__ float wave_slices_max(float v) {
    constexpr int PX = 64 / DS;
#pragma unroll
    for (int m = PX; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
template <int DS>
__device__ __forceinline__ float wave_slices_sum(float v) {
    constexpr int PX = 64 / DS;
#pragma unroll
    for (int m = PX; m < 64; m <<= 1) v = v + __shfl_xor(v, m);
    return v;
}

template <int DS>
__global__ __launch_bounds__(256) void softargmin_conf_fwd_kernel(SoftArgs a) {
    constexpr int PX = 64 / DS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slice = lane / PX;
    const int b = blockIdx.y;
    const int pix_raw = (blockIdx.x * 4 + wave) * PX + (lane % PX);
    const bool valid = pix_raw < a.HW;
    const int pix = valid ? pix_raw : a.HW - 1;  // clamp: every lane takes part in the shuffles
    const float* __restrict__ lg = a.logits + (size_t)b * a.D * a.HW + pix;
    const float* __restrict__ dv = a.per_pixel ? a.depth + (size_t)b * a.D * a.HW + pix : a.depth + (size_t)b * a.D;
    const int dstride = a.per_pixel ? a.HW : 1;

    float m = -INFINITY, s = 0.f, dep = 0.f, eidx = 0.f;
    constexpr int NR = 64;       // logits a lane can hold: the three passes below then read registers instead of memory
    if (DS > 1 && a.D <= NR * DS) {
        // ONE pass over memory with every load of the lane in flight at once (the three dependent passes of the loop form are 3 x
        // D/DS memory round trips of 64-byte pieces: 32 us for a 16 MB volume at BASELINE config 2); the same operations in the same
        // order on the same values => bit-identical results
        float v[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int d = slice + k * DS;
            v[k] = d < a.D ? lg[(size_t)d * a.HW] : -INFINITY;
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) m = fmaxf(m, v[k]);          // (fmaxf with -inf leaves m unchanged)
        m = wave_slices_max<DS>(m);
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (slice + k * DS < a.D) s += expf(v[k] - m);
        s = wave_slices_sum<DS>(s);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int d = slice + k * DS;
            if (d < a.D) {
                float p = expf(v[k] - m) / s;
                dep += p * dv[(size_t)d * dstride];
                eidx += p * (float)d;
            }
        }
    } else {
        for (int d = slice; d < a.D; d += DS) m = fmaxf(m, lg[(size_t)d * a.HW]);
        m = wave_slices_max<DS>(m);
        for (int d = slice; d < a.D; d += DS) s += expf(lg[(size_t)d * a.HW] - m);
        s = wave_slices_sum<DS>(s);
        for (int d = slice; d < a.D; d += DS) {
            float p = expf(lg[(size_t)d * a.HW] - m) / s;
            dep += p * dv[(size_t)d * dstride];
            eidx += p * (float)d;
        }
    }
    dep = wave_slices_sum<DS>(dep);
    eidx = wave_slices_sum<DS>(eidx);
    if (slice == 0 && valid) {
        // confidence: sum of p over [idx-1, idx+2], idx = trunc(E[d]) (mvsnet.py:147-151, App. A Q7)
        int idx = (int)eidx;
        float c = 0.f;
        for (int k = idx - 1; k <= idx + 2; ++k)
            if (k >= 0 && k < a.D) c += expf(lg[(size_t)k * a.HW] - m) / s;
        size_t o = (size_t)b * a.HW + pix;
        a.out_depth[o] = dep;
        a.out_conf[o] = c;
        if (a.out_max) a.out_max[o] = m;
        if (a.out_sum) a.out_sum[o] = s;
    }
}

// dL/dlogit_d = g * p_d * (depth_d - depth)   (SURVEY.md App. C); confidence carries no gradient.
__global__ __launch_bounds__(256) void softargmin_bwd_kernel(SoftArgs a) {
    const size_t total = (size_t)a.D * a.HW;
    const int b = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i / a.HW), pix = (int)(i % a.HW);
        const size_t o = (size_t)b * a.HW + pix;
        const float p = expf(a.logits[(size_t)b * total + i] - a.out_max[o]) / a.out_sum[o];
        const float dvv = a.per_pixel ? a.depth[(size_t)b * total + i] : a.depth[b * a.D + d];
        a.glogits[(size_t)b * total + i] = a.gdepth[o] * p * (dvv - a.out_depth[o]);
    }
}

extern "C" int mvs_softargmin_conf_fwd(const float* logits, const float* depth, int depth_is_per_pixel, int B, int D,
                                       int H, int W, float* out_depth, float* out_conf, float* save_max,
                                       float* save_sum, hipStream_t stream) {
    MVS_REQUIRE(logits && depth && out_depth && out_conf, MVS_ERR_NULL, "softargmin fwd: null pointer argument");
    MVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, MVS_ERR_SHAPE, "softargmin fwd: bad shape");
    SoftArgs a = {};
    a.logits = logits; a.depth = depth; a.out_depth = out_depth; a.out_conf = out_conf;
    a.out_max = save_max; a.out_sum = save_sum;
    a.B = B; a.D = D; a.HW = H * W; a.per_pixel = depth_is_per_pixel;
    dim3 block(256);
    if (D >= 32) {
        dim3 grid(mvs_cdiv(a.HW, 4 * 16), B);
        MVS_LAUNCH((softargmin_conf_fwd_kernel<4>), grid, block, 0, stream, a);
    } else {
        dim3 grid(mvs_cdiv(a.HW, 4 * 64), B);
        MVS_LAUNCH((softargmin_conf_fwd_kernel<1>), grid, block, 0, stream, a);
    }
    return mvs_check_launch("softargmin_conf_fwd");
}

extern "C" int mvs_softargmin_conf_bwd(const float* grad_depth, const float* logits, const float* depth,
                                       int depth_is_per_pixel, const float* out_depth, const float* save_max,
                                       const float* save_sum, int B, int D, int H, int W, float* grad_logits,
                                       hipStream_t stream) {
    MVS_REQUIRE(grad_depth && logits && depth && out_depth && save_max && save_sum && grad_logits, MVS_ERR_NULL,
                "softargmin bwd: null pointer argument");
    SoftArgs a = {};
    a.logits = logits; a.depth = depth; a.gdepth = grad_depth; a.glogits = grad_logits;
    a.out_depth = const_cast<float*>(out_depth);
    a.out_max = const_cast<float*>(save_max);
    a.out_sum = const_cast<float*>(save_sum);
    a.B = B; a.D = D; a.HW = H * W; a.per_pixel = depth_is_per_pixel;
    size_t total = (size_t)D * a.HW;
    int gx = (int)((total + 255) / 256);
    if (gx > 8192) gx = 8192;
    MVS_LAUNCH(softargmin_bwd_kernel, dim3(gx, B), dim3(256), 0, stream, a);
    return mvs_check_launch("softargmin_conf_bwd");
}
