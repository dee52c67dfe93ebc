// BatchNorm3d (+ReLU, + skip add after the ReLU) on channels-last activations [V][C], V = B*D*H*W.
//
// Replaces nn.BatchNorm3d + F.relu(inplace) of ConvBnReLU3D (jdacs/models/module.py:35-42) and the
// BatchNorm3d/ReLU members of the deconvolution blocks followed by the post-ReLU skip add
// (jdacs/models/mvsnet.py:48-61,70-72; jdacs-ms/models/network.py:55-64,71-72; App. A Q12).
//
// Train mode: the convolution kernels emit per-workgroup partial sums (sum x, sum x^2) of their raw
// output; bn_finalize reduces them deterministically (fp64) into mean / invstd / scale / shift and
// updates the running statistics (momentum, unbiased variance: PyTorch defaults, module.py:39).
// Eval mode: scale/shift come from the running statistics (bn_eval_affine) and are applied inside
// the convolution epilogue; no separate pass.
#include "mvs_rt.h"

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

// block-wide reduction of two doubles; result valid on thread 0
__device__ __forceinline__ void block_reduce2(double& a, double& b, double* sm) {
    const int tid = threadIdx.x;
    sm[tid] = a;
    sm[256 + tid] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            sm[tid] += sm[tid + s];
            sm[256 + tid] += sm[256 + tid + s];
        }
        __syncthreads();
    }
    a = sm[0];
    b = sm[256];
}

// partials: [nparts][2][C] (sum, sum of squares).  One block per channel.
// groups > 1: independent statistics per group g (rows [g*count, (g+1)*count) of x), e.g. the N views pushed
// through a shared-weight feature extractor as one batch; running statistics are updated group after group,
// exactly like N successive BatchNorm calls.  partials [G][nparts][2][C]; *_out [G][gs] with the 4 arrays gs apart.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int nparts, int C,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          float* running_mean, float* running_var,
                                                          float* mean_out, float* invstd_out, float* scale_out,
                                                          float* shift_out, int groups, int gs) {
    __shared__ double sm[512];
    const int c = blockIdx.x;
    for (int g = 0; g < groups; ++g, partials += (size_t)nparts * 2 * C, mean_out += gs, invstd_out += gs, scale_out += gs,
             shift_out += gs) {
    double s1 = 0.0, s2 = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) {
        s1 += (double)partials[((size_t)p * 2 + 0) * C + c];
        s2 += (double)partials[((size_t)p * 2 + 1) * C + c];
    }
    __syncthreads();
    block_reduce2(s1, s2, sm);
    if (threadIdx.x == 0) {
        double mean = s1 / count;
        double var = s2 / count - mean * mean;  // biased (normalisation)
        if (var < 0.0) var = 0.0;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        mean_out[c] = (float)mean;
        invstd_out[c] = invstd;
        float sc = gamma[c] * invstd;
        scale_out[c] = sc;
        shift_out[c] = beta[c] - (float)mean * sc;
        if (running_mean) {
            double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        float sc = gamma[c] / sqrtf(rv[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

// y = relu(x*scale + shift) (+ skip).  n4 = V*C/4 float4 elements, C % 4 == 0.
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ skip, float* __restrict__ y,
                                                            size_t n4, int C, int relu, int gs) {
    const int cq = C / 4;
    x += (size_t)blockIdx.y * n4 * 4; y += (size_t)blockIdx.y * n4 * 4;
    if (skip) skip += (size_t)blockIdx.y * n4 * 4;
    scale += blockIdx.y * gs; shift += blockIdx.y * gs;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cq) * 4;
        float4 v = ld4g(x + i * 4);
        const float4 sc = ld4g(scale + c), sh = ld4g(shift + c);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (skip) {
            const float4 k = ld4g(skip + i * 4);
            v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w;
        }
        *reinterpret_cast<float4*>(y + i * 4) = v;
    }
}

// Backward, pass 1: per-channel partial sums of dyh = dy*[relu active] and dyh*xhat.
// Each thread always sees the same channel quad (grid stride is a multiple of C/4 because 256 % (C/4) == 0).
// partials [gridDim.x][2][C].
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, size_t n4, int C, int relu,
                                                            float* __restrict__ partials, int gs) {
    __shared__ float red[256 * 8];
    const int cq = C / 4;
    const int tid = threadIdx.x;
    const int c = (tid % cq) * 4;
    dy += (size_t)blockIdx.y * n4 * 4; x += (size_t)blockIdx.y * n4 * 4;
    partials += (size_t)blockIdx.y * gridDim.x * 2 * C;
    mean += blockIdx.y * gs; invstd += blockIdx.y * gs; scale += blockIdx.y * gs; shift += blockIdx.y * gs;
    const float4 mu = ld4g(mean + c), is = ld4g(invstd + c), sc = ld4g(scale + c), sh = ld4g(shift + c);
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 xv = ld4g(x + i * 4);
        float4 g = ld4g(dy + i * 4);
        if (relu) {
            if (!(xv.x * sc.x + sh.x > 0.f)) g.x = 0.f;
            if (!(xv.y * sc.y + sh.y > 0.f)) g.y = 0.f;
            if (!(xv.z * sc.z + sh.z > 0.f)) g.z = 0.f;
            if (!(xv.w * sc.w + sh.w > 0.f)) g.w = 0.f;
        }
        a0 += g.x; a1 += g.y; a2 += g.z; a3 += g.w;
        b0 += g.x * ((xv.x - mu.x) * is.x); b1 += g.y * ((xv.y - mu.y) * is.y);
        b2 += g.z * ((xv.z - mu.z) * is.z); b3 += g.w * ((xv.w - mu.w) * is.w);
    }
    float* r = red + tid * 8;
    r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3; r[4] = b0; r[5] = b1; r[6] = b2; r[7] = b3;
    __syncthreads();
    // threads [0, 2*C): stat = t / C, channel = t % C; sum over the 256/cq threads that own that quad
    if (tid < 2 * C) {
        const int stat = tid / C, ch = tid % C;
        const int q = ch / 4, k = ch % 4;
        float s = 0.f;
        for (int t = q; t < 256; t += cq) s += red[t * 8 + stat * 4 + k];
        partials[((size_t)blockIdx.x * 2 + stat) * C + ch] = s;
    }
}

// reduce [nparts][2][C] -> sums[2][C] (fp64 accumulate); also emits dgamma = sum dyh*xhat, dbeta = sum dyh
// sums [G][2][C]; dgamma / dbeta are summed over the groups (the affine parameters are shared)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nparts, int C,
                                                              float* sums, float* dgamma, float* dbeta, int groups) {
    __shared__ double sm[512];
    const int c = blockIdx.x;
    double t1 = 0.0, t2 = 0.0;
    for (int g = 0; g < groups; ++g, partials += (size_t)nparts * 2 * C, sums += 2 * C) {
        double s1 = 0.0, s2 = 0.0;
        for (int p = threadIdx.x; p < nparts; p += 256) {
            s1 += (double)partials[((size_t)p * 2 + 0) * C + c];
            s2 += (double)partials[((size_t)p * 2 + 1) * C + c];
        }
        __syncthreads();
        block_reduce2(s1, s2, sm);
        if (threadIdx.x == 0) {
            sums[c] = (float)s1;
            sums[C + c] = (float)s2;
            t1 += s1; t2 += s2;
        }
    }
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)t1;
        if (dgamma) dgamma[c] = (float)t2;
    }
}

// Backward, pass 2: dx = scale * (dyh - mean(dyh) - xhat * mean(dyh*xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ sums, float inv_count, size_t n4,
                                                           int C, int relu, float* __restrict__ dx, int gs) {
    const int cq = C / 4;
    dy += (size_t)blockIdx.y * n4 * 4; x += (size_t)blockIdx.y * n4 * 4; dx += (size_t)blockIdx.y * n4 * 4;
    sums += (size_t)blockIdx.y * 2 * C;
    mean += blockIdx.y * gs; invstd += blockIdx.y * gs; scale += blockIdx.y * gs; shift += blockIdx.y * gs;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cq) * 4;
        const float4 mu = ld4g(mean + c), is = ld4g(invstd + c), sc = ld4g(scale + c), sh = ld4g(shift + c);
        const float4 s1 = ld4g(sums + c), s2 = ld4g(sums + C + c);
        const float4 xv = ld4g(x + i * 4);
        float4 g = ld4g(dy + i * 4);
        if (relu) {
            if (!(xv.x * sc.x + sh.x > 0.f)) g.x = 0.f;
            if (!(xv.y * sc.y + sh.y > 0.f)) g.y = 0.f;
            if (!(xv.z * sc.z + sh.z > 0.f)) g.z = 0.f;
            if (!(xv.w * sc.w + sh.w > 0.f)) g.w = 0.f;
        }
        float4 o;
        o.x = sc.x * (g.x - s1.x * inv_count - ((xv.x - mu.x) * is.x) * (s2.x * inv_count));
        o.y = sc.y * (g.y - s1.y * inv_count - ((xv.y - mu.y) * is.y) * (s2.y * inv_count));
        o.z = sc.z * (g.z - s1.z * inv_count - ((xv.z - mu.z) * is.z) * (s2.z * inv_count));
        o.w = sc.w * (g.w - s1.w * inv_count - ((xv.w - mu.w) * is.w) * (s2.w * inv_count));
        *reinterpret_cast<float4*>(dx + i * 4) = o;
    }
}

// Per-channel partial sums of a raw tensor (used when statistics are not produced by a conv epilogue).
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, size_t n4, int C,
                                                       float* __restrict__ partials) {
    __shared__ float red[256 * 8];
    const int cq = C / 4;
    const int tid = threadIdx.x;
    x += (size_t)blockIdx.y * n4 * 4;
    partials += (size_t)blockIdx.y * gridDim.x * 2 * C;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = ld4g(x + i * 4);
        a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        b0 += v.x * v.x; b1 += v.y * v.y; b2 += v.z * v.z; b3 += v.w * v.w;
    }
    float* r = red + tid * 8;
    r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3; r[4] = b0; r[5] = b1; r[6] = b2; r[7] = b3;
    __syncthreads();
    if (tid < 2 * C) {
        const int stat = tid / C, ch = tid % C;
        const int q = ch / 4, k = ch % 4;
        float s = 0.f;
        for (int t = q; t < 256; t += cq) s += red[t * 8 + stat * 4 + k];
        partials[((size_t)blockIdx.x * 2 + stat) * C + ch] = s;
    }
}

static bool bn_c_ok(int C) { return C == 4 || C == 8 || C == 16 || C == 32 || C == 64; }
static int ew_grid(size_t n4) {
    size_t g = (n4 + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g == 0 ? 1 : g));
}

extern "C" int mvs_bn_reduce_blocks(void) { return 1024; }

extern "C" int mvs_bn_stats(const float* x, long long V, int C, float* partials, int* nparts_out, hipStream_t stream) {
    MVS_REQUIRE(x && partials && nparts_out, MVS_ERR_NULL, "bn_stats: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    size_t n4 = (size_t)V * C / 4;
    int g = ew_grid(n4);
    if (g > 1024) g = 1024;
    *nparts_out = g;
    MVS_LAUNCH(bn_stats_kernel, dim3(g), dim3(256), 0, stream, x, n4, C, partials);
    return mvs_check_launch("bn_stats");
}

extern "C" int mvs_bn_finalize(const float* partials, int nparts, int C, long long count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               float* mean, float* invstd, float* scale, float* shift, hipStream_t stream) {
    MVS_REQUIRE(partials && gamma && beta && mean && invstd && scale && shift, MVS_ERR_NULL,
                "bn_finalize: null pointer argument");
    MVS_REQUIRE(nparts > 0 && C > 0 && count > 0, MVS_ERR_SHAPE, "bn_finalize: bad sizes");
    MVS_LAUNCH(bn_finalize_kernel, dim3(C), dim3(256), 0, stream, partials, nparts, C, (double)count, gamma, beta, eps,
               momentum, running_mean, running_var, mean, invstd, scale, shift, 1, 0);
    return mvs_check_launch("bn_finalize");
}

extern "C" int mvs_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* scale, float* shift,
                                  hipStream_t stream) {
    MVS_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, MVS_ERR_NULL,
                "bn_eval_affine: null pointer argument");
    MVS_LAUNCH(bn_eval_affine_kernel, dim3(mvs_cdiv(C, 64)), dim3(64), 0, stream, gamma, beta, running_mean,
               running_var, eps, C, scale, shift);
    return mvs_check_launch("bn_eval_affine");
}

extern "C" int mvs_bn_relu_fwd(const float* x, const float* scale, const float* shift, const float* skip, int relu,
                               long long V, int C, float* y, hipStream_t stream) {
    MVS_REQUIRE(x && scale && shift && y, MVS_ERR_NULL, "bn_relu_fwd: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    size_t n4 = (size_t)V * C / 4;
    MVS_LAUNCH(bn_apply_relu_kernel, dim3(ew_grid(n4)), dim3(256), 0, stream, x, scale, shift, skip, y, n4, C, relu, 0);
    return mvs_check_launch("bn_relu_fwd");
}

// dy: grad wrt relu(bn(x)) (the skip branch receives dy unchanged, handled by the caller).
// ws: >= (1024*2*C + 2*C) floats.  Outputs dx [V][C], dgamma [C], dbeta [C].
extern "C" int mvs_bn_relu_bwd(const float* dy, const float* x, const float* mean, const float* invstd,
                               const float* scale, const float* shift, int relu, long long V, int C, float* ws,
                               float* dx, float* dgamma, float* dbeta, hipStream_t stream) {
    MVS_REQUIRE(dy && x && mean && invstd && scale && shift && ws && dx, MVS_ERR_NULL, "bn_relu_bwd: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    size_t n4 = (size_t)V * C / 4;
    int g = ew_grid(n4);
    if (g > 1024) g = 1024;
    float* partials = ws;
    float* sums = ws + (size_t)1024 * 2 * C;
    MVS_LAUNCH(bn_bwd_reduce_kernel, dim3(g), dim3(256), 0, stream, dy, x, mean, invstd, scale, shift, n4, C, relu, partials, 0);
    MVS_LAUNCH(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, (const float*)partials, g, C, sums, dgamma, dbeta, 1);
    MVS_LAUNCH(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, stream, dy, x, mean, invstd, scale, shift,
               (const float*)sums, 1.0f / (float)V, n4, C, relu, dx, 0);
    return mvs_check_launch("bn_relu_bwd");
}

// ---- grouped BatchNorm(+ReLU): G independent statistics groups of Vg rows each (rows of group g are contiguous) ----
// The N views of an MVS sample go through the shared-weight 2-D feature extractor as ONE batch while BatchNorm keeps
// the reference's per-view statistics and its view-after-view running-stat updates (jdacs/models/mvsnet.py:115).
// stats: [G][4][C] (mean, invstd, scale, shift) written by fwd, read by bwd.
// ws: fwd >= G*512*2*C floats; bwd >= G*512*2*C + G*2*C floats.
#define MVS_BN_GROUP_BLOCKS 512
extern "C" int mvs_bn_group_relu_fwd(const float* x, int G, long long Vg, int C, const float* gamma, const float* beta,
                                     float eps, float momentum, float* running_mean, float* running_var, int training,
                                     int relu, float* ws, float* stats, float* y, hipStream_t stream) {
    MVS_REQUIRE(x && gamma && beta && stats && y && ws, MVS_ERR_NULL, "bn_group_relu_fwd: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    MVS_REQUIRE(G >= 1 && G <= 64 && Vg > 0, MVS_ERR_SHAPE, "bn_group_relu_fwd: bad group shape G=%d", G);
    const size_t n4 = (size_t)Vg * C / 4;
    if (training) {
        int g = ew_grid(n4);
        if (g > MVS_BN_GROUP_BLOCKS) g = MVS_BN_GROUP_BLOCKS;
        MVS_LAUNCH(bn_stats_kernel, dim3(g, G), dim3(256), 0, stream, x, n4, C, ws);
        MVS_LAUNCH(bn_finalize_kernel, dim3(C), dim3(256), 0, stream, (const float*)ws, g, C, (double)Vg, gamma, beta, eps, momentum,
                   running_mean, running_var, stats, stats + C, stats + 2 * C, stats + 3 * C, G, 4 * C);
    } else {
        MVS_REQUIRE(running_mean && running_var, MVS_ERR_NULL, "bn_group_relu_fwd: eval mode needs running statistics");
        for (int gi = 0; gi < G; ++gi)
            MVS_LAUNCH(bn_eval_affine_kernel, dim3(mvs_cdiv(C, 64)), dim3(64), 0, stream, gamma, beta, (const float*)running_mean,
                       (const float*)running_var, eps, C, stats + (size_t)gi * 4 * C + 2 * C, stats + (size_t)gi * 4 * C + 3 * C);
    }
    MVS_LAUNCH(bn_apply_relu_kernel, dim3(ew_grid(n4), G), dim3(256), 0, stream, x, (const float*)(stats + 2 * C),
               (const float*)(stats + 3 * C), (const float*)nullptr, y, n4, C, relu, 4 * C);
    return mvs_check_launch("bn_group_relu_fwd");
}

// The same in train mode with the partial sums already there (a convolution epilogue wrote them: mvs_conv2d_fwd_stats):
// partials [G][nparts][2][C], nparts rows per statistics group.  No statistics pass over x.
extern "C" int mvs_bn_group_relu_fwd_parts(const float* x, const float* partials, int nparts, int G, long long Vg, int C,
                                           const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                           float* running_var, int relu, float* stats, float* y, hipStream_t stream) {
    MVS_REQUIRE(x && partials && gamma && beta && stats && y, MVS_ERR_NULL, "bn_group_relu_fwd_parts: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    MVS_REQUIRE(G >= 1 && G <= 64 && Vg > 0 && nparts > 0, MVS_ERR_SHAPE, "bn_group_relu_fwd_parts: bad group shape G=%d rows=%d", G, nparts);
    const size_t n4 = (size_t)Vg * C / 4;
    MVS_LAUNCH(bn_finalize_kernel, dim3(C), dim3(256), 0, stream, partials, nparts, C, (double)Vg, gamma, beta, eps, momentum,
               running_mean, running_var, stats, stats + C, stats + 2 * C, stats + 3 * C, G, 4 * C);
    MVS_LAUNCH(bn_apply_relu_kernel, dim3(ew_grid(n4), G), dim3(256), 0, stream, x, (const float*)(stats + 2 * C),
               (const float*)(stats + 3 * C), (const float*)nullptr, y, n4, C, relu, 4 * C);
    return mvs_check_launch("bn_group_relu_fwd_parts");
}

extern "C" int mvs_bn_group_relu_bwd(const float* dy, const float* x, const float* stats, int relu, int G, long long Vg,
                                     int C, float* ws, float* dx, float* dgamma, float* dbeta, hipStream_t stream) {
    MVS_REQUIRE(dy && x && stats && ws && dx, MVS_ERR_NULL, "bn_group_relu_bwd: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    MVS_REQUIRE(G >= 1 && G <= 64 && Vg > 0, MVS_ERR_SHAPE, "bn_group_relu_bwd: bad group shape G=%d", G);
    const size_t n4 = (size_t)Vg * C / 4;
    int g = ew_grid(n4);
    if (g > MVS_BN_GROUP_BLOCKS) g = MVS_BN_GROUP_BLOCKS;
    float* partials = ws;
    float* sums = ws + (size_t)G * MVS_BN_GROUP_BLOCKS * 2 * C;
    const float *mean = stats, *invstd = stats + C, *scale = stats + 2 * C, *shift = stats + 3 * C;
    MVS_LAUNCH(bn_bwd_reduce_kernel, dim3(g, G), dim3(256), 0, stream, dy, x, mean, invstd, scale, shift, n4, C, relu, partials, 4 * C);
    MVS_LAUNCH(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, (const float*)partials, g, C, sums, dgamma, dbeta, G);
    MVS_LAUNCH(bn_bwd_apply_kernel, dim3(ew_grid(n4), G), dim3(256), 0, stream, dy, x, mean, invstd, scale, shift,
               (const float*)sums, 1.0f / (float)Vg, n4, C, relu, dx, 4 * C);
    return mvs_check_launch("bn_group_relu_bwd");
}
