// the self-supervised loss on the path's output, fused.
//
// Replaces UnSupLoss.forward and its autograd graph (jdacs/losses/unsup_loss.py:24-83): per source view
// inverse_warping (losses/homography.py:186-351: back-project the reference pixels with the predicted depth,
// project with K_ref.[R_rel|t_rel], bilinear gather + validity mask), compute_reconstr_loss (losses/modules.py:80-90),
// SSIM (modules.py:17-52, views 1 and 2), depth_smoothness (modules.py:55-77) and the top-3 selection over the
// views.  The reference launches ~60 small indexing / elementwise kernels per view and an index_put backward; here
// the forward is three launches and the backward two, everything at quarter resolution, all reductions
// deterministic (per-workgroup partial rows, fixed-order finish).
//
// Layouts: images NHWC [B,H,W,3] at quarter resolution (the host does F.interpolate(0.25) + permute, no gradient),
// depth [B,H,W], kinv [B,9] = K_ref^-1, proj [B,V,12] = K_ref.[R_rel|t_rel] (3x4, row major).
// Only the depth map receives a gradient, like in the reference's training step (the images are inputs).
#include "mvs_rt.h"

#define UNSUP_MAXV 10

struct UnsupArgs {
    const float* ref;                 // [B,H,W,3]
    const float* view[UNSUP_MAXV];    // V x [B,H,W,3]
    const float* kinv;                // [B,9]
    const float* proj;                // [B,V,12]
    const float* depth;               // [B,H,W]
    const float* gout;                // bwd: device scalar d total
    float* warped;                    // ws: [V,B,H,W,3]
    float* mask;                      // ws: [V,B,H,W]
    float* part;                      // ws: [(4V+2)][nblk] partial sums
    float* saved;                     // ws: 64 floats (see unsup_finalize_kernel)
    float* coef;                      // ws: [2,B,H-2,W-2,3,3] SSIM derivative coefficients (bwd)
    float* out;                       // fwd: [4] total, reconstr, ssim, smooth
    float* gdepth;                    // bwd: [B,H,W]
    float lambda;
    int B, V, H, W, nblk;
};

__device__ __forceinline__ float sl1(float z) { const float a = fabsf(z); return a < 1.f ? 0.5f * z * z : a - 0.5f; }
__device__ __forceinline__ float sl1_grad(float z) { return fabsf(z) < 1.f ? z : (z > 0.f ? 1.f : -1.f); }
__device__ __forceinline__ float sgn(float z) { return z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f); }

// sample geometry of one (pixel, view): everything the forward and the backward need
struct UnsupSample {
    float fx, fy;        // x1c - x, y1c - y (weights formed with the CLAMPED upper indices, homography.py:328-331)
    int ia, ib, ic, id;  // element offsets of the taps a=(y0,x0) b=(y1,x0) c=(y0,x1) d=(y1,x1) within one image
    float valid;
    float dxdd, dydd;    // d x / d depth, d y / d depth
};

__device__ __forceinline__ UnsupSample unsup_sample(const float* __restrict__ kinv, const float* __restrict__ P,
                                                     float depth, int px, int py, int H, int W) {
    UnsupSample s;
    const float u = (float)px, v = (float)py;
    // ray = K^-1 (u, v, 1); cam = ray * depth (homography.py:245-247)
    const float r0 = fmaf(kinv[0], u, fmaf(kinv[1], v, kinv[2]));
    const float r1 = fmaf(kinv[3], u, fmaf(kinv[4], v, kinv[5]));
    const float r2 = fmaf(kinv[6], u, fmaf(kinv[7], v, kinv[8]));
    const float c0 = r0 * depth, c1 = r1 * depth, c2 = r2 * depth;
    const float X = fmaf(P[0], c0, fmaf(P[1], c1, fmaf(P[2], c2, P[3])));
    const float Y = fmaf(P[4], c0, fmaf(P[5], c1, fmaf(P[6], c2, P[7])));
    const float Z = fmaf(P[8], c0, fmaf(P[9], c1, fmaf(P[10], c2, P[11]))) + 1e-10f;
    float x = X / Z, y = Y / Z;
    // the reference normalises to [-1,1] and back (homography.py:270-272,292-293); kept for the rounding
    x = ((x / (float)(W - 1) * 2.0f - 1.0f) + 1.0f) * ((float)W - 1.0f) / 2.0f;
    y = ((y / (float)(H - 1) * 2.0f - 1.0f) + 1.0f) * ((float)H - 1.0f) / 2.0f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float xmax = (float)(W - 1), ymax = (float)(H - 1);
    // validity in floating point (no integer overflow for absurd coordinates; NaN -> invalid)
    s.valid = (x0 >= 0.f && x0 + 1.f <= xmax && y0 >= 0.f && y0 <= ymax) ? 1.f : 0.f;
    const float x0c = fminf(fmaxf(x0, 0.f), xmax), x1c = fminf(fmaxf(x0 + 1.f, 0.f), xmax);
    const float y0c = fminf(fmaxf(y0, 0.f), ymax), y1c = fminf(fmaxf(y0 + 1.f, 0.f), ymax);
    const bool finite = x == x && y == y;
    const int ix0 = finite ? (int)x0c : 0, ix1 = finite ? (int)x1c : 0, iy0 = finite ? (int)y0c : 0, iy1 = finite ? (int)y1c : 0;
    s.fx = x1c - x;
    s.fy = y1c - y;
    s.ia = (iy0 * W + ix0) * 3; s.ib = (iy1 * W + ix0) * 3; s.ic = (iy0 * W + ix1) * 3; s.id = (iy1 * W + ix1) * 3;
    // d(X/Z)/d depth with dX/dd = P[0..2].ray etc.
    const float dX = fmaf(P[0], r0, fmaf(P[1], r1, P[2] * r2));
    const float dY = fmaf(P[4], r0, fmaf(P[5], r1, P[6] * r2));
    const float dZ = fmaf(P[8], r0, fmaf(P[9], r1, P[10] * r2));
    const float iz = 1.0f / Z;
    s.dxdd = (dX - X * iz * dZ) * iz;
    s.dydd = (dY - Y * iz * dZ) * iz;
    return s;
}

// ---- forward 1: warp every view ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unsup_warp_kernel(UnsupArgs a) {
    const int HW = a.H * a.W, n = a.B * HW;
    const int i = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    if (i >= n) return;
    const int b = i / HW, p = i - b * HW, py = p / a.W, px = p - py * a.W;
    const UnsupSample s = unsup_sample(a.kinv + b * 9, a.proj + ((size_t)b * a.V + v) * 12, a.depth[i], px, py, a.H, a.W);
    const float* __restrict__ im = a.view[v] + (size_t)b * HW * 3;
    const float wa = s.fx * s.fy, wb = s.fx * (1.0f - s.fy), wc = (1.0f - s.fx) * s.fy, wd = (1.0f - s.fx) * (1.0f - s.fy);
    float* __restrict__ o = a.warped + ((size_t)v * n + i) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = wa * im[s.ia + c] + wb * im[s.ib + c] + wc * im[s.ic + c] + wd * im[s.id + c];
    a.mask[(size_t)v * n + i] = s.valid;
}

// deterministic workgroup sum of NV values per thread -> lane 0 of wave 0 returns the totals
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red /* [4][NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v[k] += __shfl_xor(v[k], m);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = (red[k] + red[NV + k]) + (red[2 * NV + k] + red[3 * NV + k]);
}

// SSIM dissimilarity of the 3x3 window whose top-left pixel is (px,py), channel c (modules.py:35-52); also returns
// the pieces the backward needs
struct SsimWin { float t, mp, A, Bn, Cd, Dd, mux, muy, s; };
__device__ __forceinline__ SsimWin ssim_window(const float* __restrict__ x, const float* __restrict__ y,
                                               const float* __restrict__ m, int W, int c) {
    // x, y point at pixel (px,py) channel 0 of NHWC images with 3 channels; m at the mask pixel
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f, sm = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float xv = x[(j * W + i) * 3 + c], yv = y[(j * W + i) * 3 + c];
            sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv; sm += m[j * W + i];
        }
    const float k9 = 1.0f / 9.0f;
    SsimWin w;
    w.mux = sx * k9; w.muy = sy * k9;
    const float vx = sxx * k9 - w.mux * w.mux, vy = syy * k9 - w.muy * w.muy, cxy = sxy * k9 - w.mux * w.muy;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    w.A = 2.f * w.mux * w.muy + C1; w.Bn = 2.f * cxy + C2;
    w.Cd = w.mux * w.mux + w.muy * w.muy + C1; w.Dd = vx + vy + C2;
    w.mp = sm * k9;
    w.s = (1.f - (w.A * w.Bn) / (w.Cd * w.Dd)) * 0.5f;
    w.t = w.mp * fminf(fmaxf(w.s, 0.f), 1.f);
    return w;
}

// ---- forward 2: per-view sums of the photometric / difference / SSIM terms, smoothness sums -----------------
// part[(4v + k) * nblk + block]: k = 0 photo, 1 x-differences, 2 y-differences, 3 SSIM; part[(4V + k) * nblk + block]: k = 0/1 smoothness x / y
__global__ __launch_bounds__(256) void unsup_terms_kernel(UnsupArgs a) {
    __shared__ float red[4 * 4];
    const int HW = a.H * a.W, n = a.B * HW;
    const int i = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    const bool in = i < n;
    const int ii = in ? i : 0;
    const int b = ii / HW, p = ii - b * HW, py = p / a.W, px = p - py * a.W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ rf = a.ref + (size_t)ii * 3;
    if (v < a.V) {
        const float* __restrict__ wp = a.warped + ((size_t)v * n + ii) * 3;
        const float* __restrict__ mk = a.mask + (size_t)v * n + ii;
        if (in) {
            const float m0 = mk[0];
            const bool hx = px + 1 < a.W, hy = py + 1 < a.H;
            const float mx = hx ? mk[1] : 0.f, my = hy ? mk[a.W] : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float w0 = wp[c] * m0, r0 = rf[c] * m0;
                acc[0] += sl1(w0 - r0);
                if (hx) acc[1] += sl1((wp[3 + c] * mx - w0) - (rf[3 + c] * mx - r0));
                if (hy) acc[2] += sl1((wp[a.W * 3 + c] * my - w0) - (rf[a.W * 3 + c] * my - r0));
            }
            if (v < 2 && px + 2 < a.W && py + 2 < a.H) {
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[3] += ssim_window(rf, wp, mk, a.W, c).t;
            }
        }
    } else if (in) {
        const float* __restrict__ d = a.depth + ii;
        if (px + 1 < a.W) {
            const float wgt = expf(-a.lambda * ((fabsf(rf[0] - rf[3]) + fabsf(rf[1] - rf[4]) + fabsf(rf[2] - rf[5])) / 3.0f));
            acc[0] = fabsf((d[0] - d[1]) * wgt);
        }
        if (py + 1 < a.H) {
            const float* r2 = rf + a.W * 3;
            const float wgt = expf(-a.lambda * ((fabsf(rf[0] - r2[0]) + fabsf(rf[1] - r2[1]) + fabsf(rf[2] - r2[2])) / 3.0f));
            acc[1] = fabsf((d[0] - d[a.W]) * wgt);
        }
    }
    block_sum<4>(acc, red);
    if (threadIdx.x < 4) {
        const int rows = v < a.V ? 4 : 2;
        if ((int)threadIdx.x < rows) a.part[((size_t)(4 * v + threadIdx.x)) * a.nblk + blockIdx.x] = acc[threadIdx.x];
    }
}

// ---- forward 3: finish the sums, per-pixel top-3 over the views, total -------------------------------------------
// saved[0..V) r_v (per-view reconstruction term), saved[16], saved[17] SSIM means of views 1/2, saved[18] smoothness,
// saved[19] reconstruction loss, saved[20] total, saved[32..32+V) number of pixels that selected view v
__global__ __launch_bounds__(256) void unsup_finalize_kernel(UnsupArgs a) {
    __shared__ float red[4 * (UNSUP_MAXV + 1)];
    __shared__ float rv[UNSUP_MAXV];
    const int HW = a.H * a.W, n = a.B * HW;
    const int tid = threadIdx.x;
    const int nrows = 4 * a.V + 2;
    // 1. partial rows -> scalars (one row per iteration, fixed order)
    for (int r = 0; r < nrows; ++r) {
        float s[1] = {0.f};
        for (int k = tid; k < a.nblk; k += 256) s[0] += a.part[(size_t)r * a.nblk + k];
        block_sum<1>(s, red);
        if (tid == 0) a.part[(size_t)r * a.nblk] = s[0];   // row total parked in its first slot
        __syncthreads();
    }
    if (tid < a.V) {
        const float photo = a.part[(size_t)(4 * tid) * a.nblk] / ((float)n * 3.f);
        const float gx = a.part[(size_t)(4 * tid + 1) * a.nblk] / ((float)a.B * a.H * (a.W - 1) * 3.f);
        const float gy = a.part[(size_t)(4 * tid + 2) * a.nblk] / ((float)a.B * (a.H - 1) * a.W * 3.f);
        const float r = 0.5f * photo + 0.5f * (gx + gy);
        rv[tid] = r;
        a.saved[tid] = r;
        if (tid < 2) a.saved[16 + tid] = a.part[(size_t)(4 * tid + 3) * a.nblk] / ((float)a.B * (a.H - 2) * (a.W - 2) * 3.f);
    }
    if (tid == 0)
        a.saved[18] = a.part[(size_t)(4 * a.V) * a.nblk] / ((float)a.B * a.H * (a.W - 1)) +
                      a.part[(size_t)(4 * a.V + 1) * a.nblk] / ((float)a.B * (a.H - 1) * a.W);
    __syncthreads();
    // 2. per pixel: the three smallest of r_v + 1e4 * (1 - mask_v); entries >= 1e4 contribute nothing (unsup_loss.py:72-81)
    float acc[UNSUP_MAXV + 1];
#pragma unroll
    for (int k = 0; k <= UNSUP_MAXV; ++k) acc[k] = 0.f;
    for (int i = tid; i < n; i += 256) {
        float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f;
        int i0 = -1, i1 = -1, i2 = -1;
        for (int v = 0; v < a.V; ++v) {
            const float val = rv[v] + 1e4f * (1.0f - a.mask[(size_t)v * n + i]);
            if (val < b0) { b2 = b1; i2 = i1; b1 = b0; i1 = i0; b0 = val; i0 = v; }
            else if (val < b1) { b2 = b1; i2 = i1; b1 = val; i1 = v; }
            else if (val < b2) { b2 = val; i2 = v; }
        }
        const float bv[3] = {b0, b1, b2};
        const int iv[3] = {i0, i1, i2};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (iv[k] >= 0 && bv[k] < 1e4f) {
                acc[UNSUP_MAXV] += bv[k];
#pragma unroll
                for (int v = 0; v < UNSUP_MAXV; ++v) acc[v] += (iv[k] == v) ? 1.f : 0.f;
            }
    }
    block_sum<UNSUP_MAXV + 1>(acc, red);
    if (tid == 0) {
        const float reconstr = acc[UNSUP_MAXV] / (float)n;
        const float ssim = a.saved[16] + (a.V > 1 ? a.saved[17] : 0.f);
        const float total = 12.f * reconstr + 6.f * ssim + 0.18f * a.saved[18];
        a.saved[19] = reconstr;
        a.saved[20] = total;
        for (int v = 0; v < a.V; ++v) a.saved[32 + v] = acc[v];
        a.out[0] = total; a.out[1] = reconstr; a.out[2] = ssim; a.out[3] = a.saved[18];
    }
}

// ---- backward 1: SSIM derivative coefficients per window ---------------------------------------------------------
// d t_w / d y_q = alpha + beta * x_q + gamma * y_q for every pixel q of window w (y = warped image, x = reference)
__global__ __launch_bounds__(256) void unsup_ssim_coef_kernel(UnsupArgs a) {
    const int WH = a.H - 2, WW = a.W - 2, nw = a.B * WH * WW;
    const int i = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    if (i >= nw) return;
    const int b = i / (WH * WW), r = i - b * WH * WW, py = r / WW, px = r - py * WW;
    const size_t pix = ((size_t)b * a.H + py) * a.W + px, n = (size_t)a.B * a.H * a.W;
    float* __restrict__ o = a.coef + ((size_t)v * nw + i) * 9;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const SsimWin w = ssim_window(a.ref + pix * 3, a.warped + ((size_t)v * n + pix) * 3, a.mask + (size_t)v * n + pix, a.W, c);
        float al = 0.f, be = 0.f, ga = 0.f;
        if (w.s >= 0.f && w.s <= 1.f) {
            const float d = w.Cd * w.Dd, nn = w.A * w.Bn;
            const float k1 = -0.5f / d, k2 = 0.5f * nn / (d * d);   // ds = k1 dn + k2 dd
            const float k9 = 2.0f / 9.0f;
            al = w.mp * (k1 * k9 * w.mux * (w.Bn - w.A) + k2 * k9 * w.muy * (w.Dd - w.Cd));
            be = w.mp * k1 * k9 * w.A;
            ga = w.mp * k2 * k9 * w.Cd;
        }
        o[c * 3] = al; o[c * 3 + 1] = be; o[c * 3 + 2] = ga;
    }
}

// ---- backward 2: everything else -> d total / d depth -------------------------------------------------------------
__global__ __launch_bounds__(256) void unsup_grad_depth_kernel(UnsupArgs a) {
    const int HW = a.H * a.W, n = a.B * HW;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = i / HW, p = i - b * HW, py = p / a.W, px = p - py * a.W;
    const float g = a.gout[0];
    const float* __restrict__ rf = a.ref + (size_t)i * 3;
    const bool hx = px + 1 < a.W, hy = py + 1 < a.H, lx = px > 0, ly = py > 0;
    float gd = 0.f;
    const float kssim = 6.f * g / ((float)a.B * (a.H - 2) * (a.W - 2) * 3.f);
    const int WH = a.H - 2, WW = a.W - 2;
    for (int v = 0; v < a.V; ++v) {
        const float sel = 12.f * g * a.saved[32 + v] / (float)n;                       // d total / d r_v
        const float kph = sel * 0.5f / ((float)n * 3.f);
        const float kgx = sel * 0.5f / ((float)a.B * a.H * (a.W - 1) * 3.f);
        const float kgy = sel * 0.5f / ((float)a.B * (a.H - 1) * a.W * 3.f);
        const float* __restrict__ wp = a.warped + ((size_t)v * n + i) * 3;
        const float* __restrict__ mk = a.mask + (size_t)v * n + i;
        const float m0 = mk[0];
        float gw[3] = {0.f, 0.f, 0.f};
        if (m0 != 0.f && sel != 0.f) {
            const float mxp = hx ? mk[1] : 0.f, mxm = lx ? mk[-1] : 0.f, myp = hy ? mk[a.W] : 0.f, mym = ly ? mk[-a.W] : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float w0 = wp[c] * m0, r0 = rf[c] * m0;
                float t = kph * sl1_grad(w0 - r0);
                if (hx) t -= kgx * sl1_grad((wp[3 + c] * mxp - w0) - (rf[3 + c] * mxp - r0));
                if (lx) t += kgx * sl1_grad((w0 - wp[c - 3] * mxm) - (r0 - rf[c - 3] * mxm));
                if (hy) t -= kgy * sl1_grad((wp[a.W * 3 + c] * myp - w0) - (rf[a.W * 3 + c] * myp - r0));
                if (ly) t += kgy * sl1_grad((w0 - wp[c - a.W * 3] * mym) - (r0 - rf[c - a.W * 3] * mym));
                gw[c] = t * m0;
            }
        }
        if (v < 2) {   // SSIM: the windows that contain this pixel (top-left corner (px-i, py-j))
            for (int j = 0; j < 3; ++j) {
                const int wy = py - j;
                if (wy < 0 || wy >= WH) continue;
                for (int ii = 0; ii < 3; ++ii) {
                    const int wx = px - ii;
                    if (wx < 0 || wx >= WW) continue;
                    const float* __restrict__ co = a.coef + ((size_t)v * a.B * WH * WW + ((size_t)b * WH + wy) * WW + wx) * 9;
#pragma unroll
                    for (int c = 0; c < 3; ++c) gw[c] += kssim * (co[c * 3] + co[c * 3 + 1] * rf[c] + co[c * 3 + 2] * wp[c]);
                }
            }
        }
        if (gw[0] != 0.f || gw[1] != 0.f || gw[2] != 0.f) {
            const UnsupSample s = unsup_sample(a.kinv + b * 9, a.proj + ((size_t)b * a.V + v) * 12, a.depth[i], px, py, a.H, a.W);
            const float* __restrict__ im = a.view[v] + (size_t)b * HW * 3;
            float dx = 0.f, dy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float pa = im[s.ia + c], pb = im[s.ib + c], pc = im[s.ic + c], pd = im[s.id + c];
                // out = fx fy a + fx (1-fy) b + (1-fx) fy c + (1-fx)(1-fy) d with fx = x1c - x, fy = y1c - y
                dx += gw[c] * (-(s.fy * pa) - (1.f - s.fy) * pb + s.fy * pc + (1.f - s.fy) * pd);
                dy += gw[c] * (-(s.fx * pa) + s.fx * pb - (1.f - s.fx) * pc + (1.f - s.fx) * pd);
            }
            gd += dx * s.dxdd + dy * s.dydd;
        }
    }
    // smoothness (modules.py:66-77)
    {
        const float ksx = 0.18f * g / ((float)a.B * a.H * (a.W - 1)), ksy = 0.18f * g / ((float)a.B * (a.H - 1) * a.W);
        const float* __restrict__ d = a.depth + i;
        if (hx) {
            const float wgt = expf(-a.lambda * ((fabsf(rf[0] - rf[3]) + fabsf(rf[1] - rf[4]) + fabsf(rf[2] - rf[5])) / 3.0f));
            gd += ksx * sgn((d[0] - d[1]) * wgt) * wgt;
        }
        if (lx) {
            const float wgt = expf(-a.lambda * ((fabsf(rf[-3] - rf[0]) + fabsf(rf[-2] - rf[1]) + fabsf(rf[-1] - rf[2])) / 3.0f));
            gd -= ksx * sgn((d[-1] - d[0]) * wgt) * wgt;
        }
        if (hy) {
            const float* r2 = rf + a.W * 3;
            const float wgt = expf(-a.lambda * ((fabsf(rf[0] - r2[0]) + fabsf(rf[1] - r2[1]) + fabsf(rf[2] - r2[2])) / 3.0f));
            gd += ksy * sgn((d[0] - d[a.W]) * wgt) * wgt;
        }
        if (ly) {
            const float* r2 = rf - a.W * 3;
            const float wgt = expf(-a.lambda * ((fabsf(r2[0] - rf[0]) + fabsf(r2[1] - rf[1]) + fabsf(r2[2] - rf[2])) / 3.0f));
            gd -= ksy * sgn((d[-a.W] - d[0]) * wgt) * wgt;
        }
    }
    a.gdepth[i] = gd;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int unsup_fill(UnsupArgs& a, const float* ref, const float* const* views, const float* kinv, const float* proj,
                      const float* depth, int B, int V, int H, int W, float lambda, float* ws) {
    MVS_REQUIRE(ref && views && kinv && proj && depth && ws, MVS_ERR_NULL, "unsup_loss: null pointer argument");
    MVS_REQUIRE(V >= 3 && V <= UNSUP_MAXV, MVS_ERR_UNSUPPORTED,
                "unsup_loss: needs 3..%d source views (top-3 selection, unsup_loss.py:76), got %d", UNSUP_MAXV, V);
    MVS_REQUIRE(B > 0 && H >= 3 && W >= 3, MVS_ERR_SHAPE, "unsup_loss: bad shape B=%d H=%d W=%d", B, H, W);
    a = UnsupArgs{};
    a.ref = ref; a.kinv = kinv; a.proj = proj; a.depth = depth; a.lambda = lambda;
    for (int v = 0; v < V; ++v) {
        MVS_REQUIRE(views[v], MVS_ERR_NULL, "unsup_loss: null view image %d", v);
        a.view[v] = views[v];
    }
    a.B = B; a.V = V; a.H = H; a.W = W;
    const size_t n = (size_t)B * H * W;
    a.nblk = (int)((n + 255) / 256);
    a.warped = ws;
    a.mask = a.warped + (size_t)V * n * 3;
    a.part = a.mask + (size_t)V * n;
    a.saved = a.part + (size_t)(4 * V + 2) * a.nblk;
    a.coef = a.saved + 64;
    return MVS_OK;
}

extern "C" long long mvs_unsup_loss_workspace_floats(int B, int V, int H, int W) {
    if (B <= 0 || V <= 0 || H < 3 || W < 3) return -1;
    const long long n = (long long)B * H * W, nblk = (n + 255) / 256;
    return (long long)V * n * 4 + (4LL * V + 2) * nblk + 64 + 2LL * B * (H - 2) * (W - 2) * 9;
}

// out[4] = total, reconstruction, SSIM, smoothness terms (device memory)
extern "C" int mvs_unsup_loss_fwd(const float* ref, const float* const* views, const float* kinv, const float* proj,
                                  const float* depth, int B, int V, int H, int W, float smooth_lambda, float* ws,
                                  float* out, hipStream_t stream) {
    UnsupArgs a;
    int rc = unsup_fill(a, ref, views, kinv, proj, depth, B, V, H, W, smooth_lambda, ws);
    if (rc) return rc;
    MVS_REQUIRE(out, MVS_ERR_NULL, "unsup_loss_fwd: null output");
    a.out = out;
    MVS_LAUNCH(unsup_warp_kernel, dim3(a.nblk, V), dim3(256), 0, stream, a);
    MVS_LAUNCH(unsup_terms_kernel, dim3(a.nblk, V + 1), dim3(256), 0, stream, a);
    MVS_LAUNCH(unsup_finalize_kernel, dim3(1), dim3(256), 0, stream, a);
    return mvs_check_launch("unsup_loss_fwd");
}

// ws: the workspace the forward filled (warped images, masks, per-view selection counts); grad_out: device scalar
extern "C" int mvs_unsup_loss_bwd(const float* ref, const float* const* views, const float* kinv, const float* proj,
                                  const float* depth, int B, int V, int H, int W, float smooth_lambda, float* ws,
                                  const float* grad_out, float* grad_depth, hipStream_t stream) {
    UnsupArgs a;
    int rc = unsup_fill(a, ref, views, kinv, proj, depth, B, V, H, W, smooth_lambda, ws);
    if (rc) return rc;
    MVS_REQUIRE(grad_out && grad_depth, MVS_ERR_NULL, "unsup_loss_bwd: null gradient pointer");
    a.gout = grad_out;
    a.gdepth = grad_depth;
    const int nw = B * (H - 2) * (W - 2);
    MVS_LAUNCH(unsup_ssim_coef_kernel, dim3((nw + 255) / 256, 2), dim3(256), 0, stream, a);
    MVS_LAUNCH(unsup_grad_depth_kernel, dim3(a.nblk), dim3(256), 0, stream, a);
    return mvs_check_launch("unsup_loss_bwd");
}
