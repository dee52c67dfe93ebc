// the stage glue of CVP-MVSNet, fused.
//
// Replaces calDepthHypo (jdacs-ms/models/modules.py:107-206): for every pixel, the depth change that moves its
// projection into source view 0 by ONE pixel along the epipolar line (project at d and d+1 for the direction, step one
// pixel, solve the reference's 2x2 system in closed form), the MEAN of its magnitude over the image = the level's depth
// interval, then the 8 per-pixel hypotheses depth + k * interval, k = -4..3.  The reference runs ~40 elementwise ops and
// an H*W-batched 2x2 torch.inverse in fp64 per batch item; here it is two launches, fp64 inside like the reference
// (App. A Q4), deterministic mean (per-workgroup partials, fixed-order finish).
//
// mats [B][30] (fp64, host-prepared from the 3x3 / 4x4 camera matrices): [0..9) K_ref^-1, [9..21) T = K_src (E_src E_ref^-1)[:3,:]
// (3x4 row major), [21..30) A = (K_ref R_ref)(K_src R_src)^-1.
#include <math.h>

#include "mvs_rt.h"

struct HypoArgs {
    const float* depth;    // [B,H,W]
    const double* mats;    // [B,30]
    double* part;          // [B][nblk]
    float* hypos;          // [B,8,H,W]
    int B, H, W, nblk;
};

__device__ __forceinline__ void hypo_project(const double* __restrict__ Ki, const double* __restrict__ T, double x, double y,
                                             double dz, double& u, double& v, double& z) {
    const double r0 = (Ki[0] * x + Ki[1] * y + Ki[2]) * dz, r1 = (Ki[3] * x + Ki[4] * y + Ki[5]) * dz, r2 = (Ki[6] * x + Ki[7] * y + Ki[8]) * dz;
    const double px = T[0] * r0 + T[1] * r1 + T[2] * r2 + T[3];
    const double py = T[4] * r0 + T[5] * r1 + T[6] * r2 + T[7];
    z = T[8] * r0 + T[9] * r1 + T[10] * r2 + T[11];
    u = px / z;
    v = py / z;
}

__global__ __launch_bounds__(256) void depth_hypo_interval_kernel(HypoArgs a) {
    __shared__ double red[4];
    const int HW = a.H * a.W, b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const double* __restrict__ m = a.mats + (size_t)b * 30;
    double val = 0.0;
    if (p < HW) {
        const int py = p / a.W, px = p - py * a.W;
        const double x = (double)px, y = (double)py, d1 = (double)a.depth[(size_t)b * HW + p];
        double u1, v1, z1, u2, v2, z2;
        hypo_project(m, m + 9, x, y, d1, u1, v1, z1);
        hypo_project(m, m + 9, x, y, d1 + 1.0, u2, v2, z2);
        const double theta = atan((v2 - v1) / (u2 - u1));
        const double u3 = u1 + cos(theta), v3 = v1 + sin(theta);          // one pixel along the epipolar line
        const double* __restrict__ A = m + 21;
        const double t1y = z1 * (A[3] * u1 + A[4] * v1 + A[5]), t1z = z1 * (A[6] * u1 + A[7] * v1 + A[8]);
        const double t2y = A[3] * u3 + A[4] * v3 + A[5], t2z = A[6] * u3 + A[7] * v3 + A[8];
        // [[y, t2.y], [1, t2.z]] (delta, .)^T = (t1.y, t1.z)^T  (modules.py:186-193)
        val = fabs((t2z * t1y - t2y * t1z) / (y * t2z - t2y));
    }
    // deterministic workgroup sum
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float hi = (float)val, lo = (float)(val - (double)hi);   // shuffles move 32-bit values: split the double
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const double o = (double)__shfl_xor(hi, s) + (double)__shfl_xor(lo, s);
        val += o;
        hi = (float)val;
        lo = (float)(val - (double)hi);
    }
    if (lane == 0) red[wave] = val;
    __syncthreads();
    if (threadIdx.x == 0) a.part[(size_t)b * a.nblk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void depth_hypo_write_kernel(HypoArgs a) {
    __shared__ double interval;
    const int HW = a.H * a.W, b = blockIdx.y;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < a.nblk; ++k) s += a.part[(size_t)b * a.nblk + k];   // same order in every workgroup
        interval = s / (double)HW;
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const double d1 = (double)a.depth[(size_t)b * HW + p];
#pragma unroll
    for (int k = 0; k < 8; ++k) a.hypos[((size_t)b * 8 + k) * HW + p] = (float)(d1 + (double)(k - 4) * interval);
}

extern "C" long long mvs_depth_hypo_workspace_doubles(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    return (long long)B * (((long long)H * W + 255) / 256);
}

// ref_depths [B,H,W] fp32 (the upsampled depth of the coarser level), mats [B,30] fp64 (see the file header), ws: fp64
// scratch of mvs_depth_hypo_workspace_doubles() values, hypos [B,8,H,W] fp32
extern "C" int mvs_depth_hypo(const float* ref_depths, const double* mats, int B, int H, int W, double* ws, float* hypos,
                              hipStream_t stream) {
    MVS_REQUIRE(ref_depths && mats && ws && hypos, MVS_ERR_NULL, "depth_hypo: null pointer argument");
    MVS_REQUIRE(B > 0 && H > 0 && W > 0, MVS_ERR_SHAPE, "depth_hypo: bad shape B=%d H=%d W=%d", B, H, W);
    HypoArgs a;
    a.depth = ref_depths; a.mats = mats; a.part = ws; a.hypos = hypos; a.B = B; a.H = H; a.W = W;
    a.nblk = (H * W + 255) / 256;
    MVS_LAUNCH(depth_hypo_interval_kernel, dim3(a.nblk, B), dim3(256), 0, stream, a);
    MVS_LAUNCH(depth_hypo_write_kernel, dim3(a.nblk, B), dim3(256), 0, stream, a);
    return mvs_check_launch("depth_hypo");
}
