// mvsnet_loss (jdacs/models/mvsnet.py:164-166): mean smooth-L1 (beta = 1) between the estimated and the ground-truth depth map over
// the pixels with mask > 0.5, forward and backward, as two launches.  The torch formulation of the same value is ~6 elementwise /
// reduction launches forward and ~8 backward on [B,H,W] maps of 20 k pixels -- pure launch latency on the step's critical path
// (profiles/r04_run3_trace_tail.csv: 14 kernels of 3-7 us between the soft-argmin forward and its backward).
#include "mvs_rt.h"

// One workgroup of 1024 threads walks all n pixels (n = B*H*W: 20 k per sample at BASELINE config 2): fp64 sums, fixed order.
// out[0] = sum / count (nan for an empty mask, like the reference's mean over an empty selection), out[1] = count.
__global__ __launch_bounds__(1024) void masked_smooth_l1_fwd_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                                   const float* __restrict__ mask, long long n, float* __restrict__ out) {
    __shared__ double sm[2 * 1024];
    const int tid = threadIdx.x;
    double s = 0.0, c = 0.0;
    for (long long i = tid; i < n; i += 1024) {
        if (mask[i] > 0.5f) {
            const float d = est[i] - gt[i], a = fabsf(d);
            s += (double)(a < 1.f ? 0.5f * d * d : a - 0.5f);
            c += 1.0;
        }
    }
    sm[tid] = s;
    sm[1024 + tid] = c;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if (tid < k) { sm[tid] += sm[tid + k]; sm[1024 + tid] += sm[1024 + tid + k]; }
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = (float)(sm[0] / sm[1024]);
        out[1] = (float)sm[1024];
    }
}

// d loss / d est[i] = [mask > 0.5] * clamp(est - gt, -1, 1) / count * gloss
__global__ __launch_bounds__(256) void masked_smooth_l1_bwd_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                                  const float* __restrict__ mask, const float* __restrict__ fwd_out,
                                                                  const float* __restrict__ gloss, long long n, float* __restrict__ gest) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float scale = gloss[0] / fwd_out[1];
    float g = 0.f;
    if (mask[i] > 0.5f) {
        const float d = est[i] - gt[i];
        g = fminf(fmaxf(d, -1.f), 1.f) * scale;
    }
    gest[i] = g;
}

extern "C" int mvs_masked_smooth_l1_fwd(const float* est, const float* gt, const float* mask, long long n, float* out,
                                        hipStream_t stream) {
    MVS_REQUIRE(est && gt && mask && out, MVS_ERR_NULL, "masked_smooth_l1_fwd: null pointer argument");
    MVS_REQUIRE(n > 0, MVS_ERR_SHAPE, "masked_smooth_l1_fwd: empty input");
    MVS_LAUNCH(masked_smooth_l1_fwd_kernel, dim3(1), dim3(1024), 0, stream, est, gt, mask, n, out);
    return mvs_check_launch("masked_smooth_l1_fwd");
}

extern "C" int mvs_masked_smooth_l1_bwd(const float* est, const float* gt, const float* mask, const float* fwd_out, const float* gloss,
                                        long long n, float* gest, hipStream_t stream) {
    MVS_REQUIRE(est && gt && mask && fwd_out && gloss && gest, MVS_ERR_NULL, "masked_smooth_l1_bwd: null pointer argument");
    MVS_REQUIRE(n > 0, MVS_ERR_SHAPE, "masked_smooth_l1_bwd: empty input");
    MVS_LAUNCH(masked_smooth_l1_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, est, gt, mask, fwd_out, gloss, n, gest);
    return mvs_check_launch("masked_smooth_l1_bwd");
}
