// BatchNorm3d / BatchNorm2d (+ReLU, + skip add after the ReLU) on channels-last activations [V][C].
//
// Replaces nn.BatchNorm3d + F.relu(inplace) of ConvBnReLU3D (jdacs/models/module.py:35-42), the BatchNorm3d/ReLU members of
// the deconvolution blocks followed by the post-ReLU skip add (jdacs/models/mvsnet.py:48-61,70-72;
// jdacs-ms/models/network.py:55-64,71-72; App. A Q12) and BatchNorm2d + ReLU of the 2-D ConvBnReLU (module.py:15-22).
//
// Train mode, round 4 form ("statistic slots"): whoever produces a tensor that BatchNorm will normalise -- a convolution
// epilogue (conv3d.hip / conv2d.hip) or bn_stats_slots_kernel -- adds its per-workgroup (sum x, sum x^2) per channel into one of
// `nslots` fp64 accumulator rows with global_atomic_add_f64 (row = workgroup index mod nslots; the caller zeroes the rows).
// The kernel that APPLIES the normalisation finishes the statistics in its own prologue: every workgroup sums the <= 16 KB of
// slot rows in a fixed order (fp64) and derives mean / invstd / scale / shift into LDS; workgroup (0, g) also writes them out
// for the backward pass and workgroup (0, 0) updates the running statistics (momentum, unbiased variance: PyTorch defaults,
// module.py:39).  No finalize launch between the producer and the apply pass (rounds 1-3: partial rows -> bn_finalize ->
// apply: 34 launches of 6-9 us per training step, profiles/r03_final_rocprofv3_kernel_stats.csv), and no cross-workgroup
// hand-off inside a launch.  The backward pass has the same shape: (sum dyh, sum dyh*xhat) arrive in slots -- from the epilogue
// of the input-gradient kernel that produced dy (conv3d.hip, "bn_raw") or from bn_bwd_reduce_slots_kernel -- and the apply
// kernel finishes them in its prologue (dgamma / dbeta written by workgroup (0, 0)).
// Eval mode: scale/shift come from the running statistics (bn_eval_affine) and are applied inside the convolution epilogue.
#include "mvs_rt.h"

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Sum the slot rows [nslots][2][C] of one statistics group in a fixed order (fp64): tot[stat*C + c].  All 256 threads call it.
// nslots * 2C <= 2048 doubles (mvs_bn_slots) -> <= 8 per thread, ALL requested before the first add: one memory round trip (a
// `for (k ...) s += slots[..]` loop made hipcc wait for every load before issuing the next: 8 serial round trips ~ 10 us in the
// prologue of every workgroup, profiles/r04_run1_*).  Thread t always sees element e = t % 2C (256 % 2C == 0).
__device__ __forceinline__ void bn_slot_totals(const double* __restrict__ slots, int nslots, int C, double* red, double* tot) {
    const int tid = threadIdx.x, E = 2 * C, nsub = 256 / E;
    const int total = nslots * E;
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j;
        v[j] = idx < total ? slots[idx] : 0.0;
    }
    double s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    for (int idx = tid + 2048; idx < total; idx += 256) s += slots[idx];   // (only with more slot rows than mvs_bn_slots recommends)
    __syncthreads();                 // a previous call's readers of red / tot are done
    red[tid] = s;
    __syncthreads();
    if (tid < E) {
        double t = 0.0;
        for (int k = 0; k < nsub; ++k) t += red[k * E + tid];
        tot[tid] = t;
    }
    __syncthreads();
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


// y = relu(x*scale + shift) (+ skip) with scale / shift finished from the statistic slots in the prologue.
// grid (blocks, G); slots [G][nslots][2][C]; stats [G][4][C] (mean, invstd, scale, shift) written for the backward pass.
__global__ __launch_bounds__(256) void bn_fwd_slots_kernel(const float* __restrict__ x, const double* __restrict__ slots, int nslots,
                                                           int C, double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* running_mean, float* running_var, const float* __restrict__ skip,
                                                           int relu, float* __restrict__ stats, float* __restrict__ y, size_t n4) {
    __shared__ double red[256];
    __shared__ double tot[128];
    __shared__ __attribute__((aligned(16))) float aff[128];   // scale[C], shift[C]
    const int tid = threadIdx.x, G = gridDim.y, g = blockIdx.y;
    const bool owner = blockIdx.x == 0 && g == 0 && running_mean;
    if (owner) {
        // running statistics, group after group like G successive BatchNorm calls (jdacs/models/mvsnet.py:115)
        for (int gg = 0; gg < G; ++gg) {
            bn_slot_totals(slots + (size_t)gg * nslots * 2 * C, nslots, C, red, tot);
            if (tid < C) {
                const double mean = tot[tid] / count;
                double var = tot[C + tid] / count - mean * mean;
                if (var < 0.0) var = 0.0;
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[tid] = (1.0f - momentum) * running_mean[tid] + momentum * (float)mean;
                running_var[tid] = (1.0f - momentum) * running_var[tid] + momentum * (float)unbiased;
            }
        }
    }
    if (!(owner && G == 1))          // (one group: tot already holds this workgroup's totals)
        bn_slot_totals(slots + (size_t)g * nslots * 2 * C, nslots, C, red, tot);
    if (tid < C) {
        const double mean = tot[tid] / count;
        double var = tot[C + tid] / count - mean * mean;   // biased (normalisation)
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[tid] * invstd, sh = beta[tid] - (float)mean * sc;
        aff[tid] = sc;
        aff[C + tid] = sh;
        if (blockIdx.x == 0) {
            float* st = stats + (size_t)g * 4 * C;
            st[tid] = (float)mean; st[C + tid] = invstd; st[2 * C + tid] = sc; st[3 * C + tid] = sh;
        }
    }
    __syncthreads();
    const int cq = C / 4;
    const int c = (tid % cq) * 4;                       // the same channel quad on every grid-stride step (256 % cq == 0)
    const float4 sc = *reinterpret_cast<const float4*>(aff + c), sh = *reinterpret_cast<const float4*>(aff + C + c);
    x += (size_t)g * n4 * 4; y += (size_t)g * n4 * 4;
    if (skip) skip += (size_t)g * n4 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ld4g(x + i * 4);
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

// 256 per-thread (quad of channels) x 8 values -> per-channel workgroup sums -> one slot row (fp64 atomics)
__device__ __forceinline__ void bn_block_to_slot(float (&v)[8], float* red, int C, double* __restrict__ slot_row) {
    const int tid = threadIdx.x, cq = C / 4;
    float* r = red + tid * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = v[k];
    __syncthreads();
    // threads [0, 2*C): stat = t / C, channel = t % C; sum over the 256/cq threads that own that quad
    if (tid < 2 * C) {
        const int stat = tid / C, ch = tid % C;
        const int q = ch / 4, k = ch % 4;
        float s = 0.f;
        for (int t = q; t < 256; t += cq) s += red[t * 8 + stat * 4 + k];
        MVS_GLOBAL_ATOMIC_ADD_F64(slot_row + tid, (double)s);
    }
}

// statistics of a tensor no convolution epilogue has summed: (sum x, sum x^2) per channel into the slots
__global__ __launch_bounds__(256) void bn_stats_slots_kernel(const float* __restrict__ x, size_t n4, int C,
                                                             double* __restrict__ slots, int nslots) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    x += (size_t)blockIdx.y * n4 * 4;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = ld4g(x + i * 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += a.x * a.x; v[5] += a.y * a.y; v[6] += a.z * a.z; v[7] += a.w * a.w;
    }
    bn_block_to_slot(v, red, C, slots + ((size_t)blockIdx.y * nslots + (blockIdx.x & (nslots - 1))) * 2 * C);
}

// Backward statistics when no input-gradient epilogue produced them: (sum dyh, sum dyh*xhat), dyh = dy*[relu active].
// stats [G][4][C].
__global__ __launch_bounds__(256) void bn_bwd_reduce_slots_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ stats, size_t n4, int C, int relu,
                                                                  double* __restrict__ slots, int nslots) {
    __shared__ float red[256 * 8];
    const int cq = C / 4;
    const int tid = threadIdx.x;
    const int c = (tid % cq) * 4;
    dy += (size_t)blockIdx.y * n4 * 4; x += (size_t)blockIdx.y * n4 * 4;
    const float* st = stats + (size_t)blockIdx.y * 4 * C;
    const float4 mu = ld4g(st + c), is = ld4g(st + C + c), sc = ld4g(st + 2 * C + c), sh = ld4g(st + 3 * C + c);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 xv = ld4g(x + i * 4);
        float4 g = ld4g(dy + i * 4);
        if (relu) {
            if (!(xv.x * sc.x + sh.x > 0.f)) g.x = 0.f;
            if (!(xv.y * sc.y + sh.y > 0.f)) g.y = 0.f;
            if (!(xv.z * sc.z + sh.z > 0.f)) g.z = 0.f;
            if (!(xv.w * sc.w + sh.w > 0.f)) g.w = 0.f;
        }
        v[0] += g.x; v[1] += g.y; v[2] += g.z; v[3] += g.w;
        v[4] += g.x * ((xv.x - mu.x) * is.x); v[5] += g.y * ((xv.y - mu.y) * is.y);
        v[6] += g.z * ((xv.z - mu.z) * is.z); v[7] += g.w * ((xv.w - mu.w) * is.w);
    }
    bn_block_to_slot(v, red, C, slots + ((size_t)blockIdx.y * nslots + (blockIdx.x & (nslots - 1))) * 2 * C);
}

// Backward apply: dx = scale * (dyh - mean(dyh) - xhat * mean(dyh*xhat)), the two means finished from the slots in the
// prologue; workgroup (0, 0) also writes dgamma = sum dyh*xhat, dbeta = sum dyh (summed over the groups: shared affine).
__global__ __launch_bounds__(256) void bn_bwd_slots_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ stats, const double* __restrict__ slots,
                                                           int nslots, float inv_count, size_t n4, int C, int relu,
                                                           float* __restrict__ dx, float* dgamma, float* dbeta) {
    __shared__ double red[256];
    __shared__ double tot[128];
    __shared__ __attribute__((aligned(16))) float sums[128];   // s1[C], s2[C]
    const int tid = threadIdx.x, G = gridDim.y, g = blockIdx.y;
    const bool owner = blockIdx.x == 0 && g == 0 && (dgamma || dbeta);
    if (owner) {
        double t1 = 0.0, t2 = 0.0;
        for (int gg = 0; gg < G; ++gg) {
            bn_slot_totals(slots + (size_t)gg * nslots * 2 * C, nslots, C, red, tot);
            if (tid < C) { t1 += tot[tid]; t2 += tot[C + tid]; }
        }
        if (tid < C) {
            if (dbeta) dbeta[tid] = (float)t1;
            if (dgamma) dgamma[tid] = (float)t2;
        }
    }
    if (!(owner && G == 1))
        bn_slot_totals(slots + (size_t)g * nslots * 2 * C, nslots, C, red, tot);
    if (tid < 2 * C) sums[tid] = (float)tot[tid];
    __syncthreads();
    const int cq = C / 4;
    const int c = (tid % cq) * 4;
    const float* st = stats + (size_t)g * 4 * C;
    const float4 mu = ld4g(st + c), is = ld4g(st + C + c), sc = ld4g(st + 2 * C + c), sh = ld4g(st + 3 * C + c);
    const float4 s1 = *reinterpret_cast<const float4*>(sums + c), s2 = *reinterpret_cast<const float4*>(sums + C + c);
    dy += (size_t)g * n4 * 4; x += (size_t)g * n4 * 4; dx += (size_t)g * n4 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 xv = ld4g(x + i * 4);
        float4 gv = ld4g(dy + i * 4);
        if (relu) {
            if (!(xv.x * sc.x + sh.x > 0.f)) gv.x = 0.f;
            if (!(xv.y * sc.y + sh.y > 0.f)) gv.y = 0.f;
            if (!(xv.z * sc.z + sh.z > 0.f)) gv.z = 0.f;
            if (!(xv.w * sc.w + sh.w > 0.f)) gv.w = 0.f;
        }
        float4 o;
        o.x = sc.x * (gv.x - s1.x * inv_count - ((xv.x - mu.x) * is.x) * (s2.x * inv_count));
        o.y = sc.y * (gv.y - s1.y * inv_count - ((xv.y - mu.y) * is.y) * (s2.y * inv_count));
        o.z = sc.z * (gv.z - s1.z * inv_count - ((xv.z - mu.z) * is.z) * (s2.z * inv_count));
        o.w = sc.w * (gv.w - s1.w * inv_count - ((xv.w - mu.w) * is.w) * (s2.w * inv_count));
        *reinterpret_cast<float4*>(dx + i * 4) = o;
    }
}

static bool bn_c_ok(int C) { return C == 4 || C == 8 || C == 16 || C == 32 || C == 64; }
static bool bn_slots_ok(int n) { return n >= 1 && n <= 256 && (n & (n - 1)) == 0; }
static int ew_grid(size_t n4) {
    size_t g = (n4 + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g == 0 ? 1 : g));
}

// slot rows for C channels: 16 KB of fp64 accumulators per statistics group (what every workgroup of an apply pass re-reads)
extern "C" int mvs_bn_slots(int C) {
    if (!bn_c_ok(C)) return -1;
    const int n = 1024 / C;
    return n > 128 ? 128 : n;
}

extern "C" int mvs_bn_stats_slots(const float* x, int G, long long Vg, int C, double* slots, int nslots, hipStream_t stream) {
    MVS_REQUIRE(x && slots, MVS_ERR_NULL, "bn_stats_slots: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    MVS_REQUIRE(G >= 1 && G <= 64 && Vg > 0 && bn_slots_ok(nslots), MVS_ERR_SHAPE, "bn_stats_slots: bad shape G=%d slots=%d", G, nslots);
    const size_t n4 = (size_t)Vg * C / 4;
    int g = ew_grid(n4);
    if (g > 512) g = 512;
    MVS_LAUNCH(bn_stats_slots_kernel, dim3(g, G), dim3(256), 0, stream, x, n4, C, slots, nslots);
    return mvs_check_launch("bn_stats_slots");
}

extern "C" int mvs_bn_relu_fwd_slots(const float* x, const double* slots, int nslots, int G, long long Vg, int C,
                                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, const float* skip, int relu, float* stats, float* y, hipStream_t stream) {
    MVS_REQUIRE(x && slots && gamma && beta && stats && y, MVS_ERR_NULL, "bn_relu_fwd_slots: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    MVS_REQUIRE(G >= 1 && G <= 64 && Vg > 0 && bn_slots_ok(nslots), MVS_ERR_SHAPE, "bn_relu_fwd_slots: bad shape G=%d slots=%d", G, nslots);
    MVS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), MVS_ERR_NULL, "bn_relu_fwd_slots: running_mean and running_var go together");
    const size_t n4 = (size_t)Vg * C / 4;
    MVS_LAUNCH(bn_fwd_slots_kernel, dim3(ew_grid(n4), G), dim3(256), 0, stream, x, slots, nslots, C, (double)Vg, gamma, beta, eps,
               momentum, running_mean, running_var, skip, relu, stats, y, n4);
    return mvs_check_launch("bn_relu_fwd_slots");
}

// The prologue of mvs_bn_relu_fwd_slots alone: mean / invstd / scale / shift of every group into stats [G][4][C] and the running
// statistics, group after group -- for a block whose normalisation is applied by its CONSUMER while it stages its input
// (mvs_conv2d_fwd_stats_xf, mvs_conv2d_wgrad_batch_xf), so that no apply pass and no normalised copy of the tensor exist.
extern "C" int mvs_bn_finalize_slots(const double* slots, int nslots, int G, long long Vg, int C, const float* gamma, const float* beta,
                                     float eps, float momentum, float* running_mean, float* running_var, float* stats,
                                     hipStream_t stream) {
    MVS_REQUIRE(slots && gamma && beta && stats, MVS_ERR_NULL, "bn_finalize_slots: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    MVS_REQUIRE(G >= 1 && G <= 64 && Vg > 0 && bn_slots_ok(nslots), MVS_ERR_SHAPE, "bn_finalize_slots: bad shape G=%d slots=%d", G, nslots);
    MVS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), MVS_ERR_NULL, "bn_finalize_slots: running_mean and running_var go together");
    MVS_LAUNCH(bn_fwd_slots_kernel, dim3(1, G), dim3(256), 0, stream, (const float*)nullptr, slots, nslots, C, (double)Vg, gamma, beta, eps,
               momentum, running_mean, running_var, (const float*)nullptr, 1, stats, (float*)nullptr, (size_t)0);
    return mvs_check_launch("bn_finalize_slots");
}

extern "C" int mvs_bn_bwd_reduce_slots(const float* dy, const float* x, const float* stats, int relu, int G, long long Vg, int C,
                                       double* slots, int nslots, hipStream_t stream) {
    MVS_REQUIRE(dy && x && stats && slots, MVS_ERR_NULL, "bn_bwd_reduce_slots: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    MVS_REQUIRE(G >= 1 && G <= 64 && Vg > 0 && bn_slots_ok(nslots), MVS_ERR_SHAPE, "bn_bwd_reduce_slots: bad shape G=%d slots=%d", G, nslots);
    const size_t n4 = (size_t)Vg * C / 4;
    int g = ew_grid(n4);
    if (g > 1024) g = 1024;
    MVS_LAUNCH(bn_bwd_reduce_slots_kernel, dim3(g, G), dim3(256), 0, stream, dy, x, stats, n4, C, relu, slots, nslots);
    return mvs_check_launch("bn_bwd_reduce_slots");
}

// dy: grad wrt relu(bn(x)) (the skip branch receives dy unchanged, handled by the caller).  Outputs dx [G*Vg][C], dgamma [C], dbeta [C].
extern "C" int mvs_bn_relu_bwd_slots(const float* dy, const float* x, const float* stats, const double* slots, int nslots, int relu,
                                     int G, long long Vg, int C, float* dx, float* dgamma, float* dbeta, hipStream_t stream) {
    MVS_REQUIRE(dy && x && stats && slots && dx, MVS_ERR_NULL, "bn_relu_bwd_slots: null pointer argument");
    MVS_REQUIRE(bn_c_ok(C), MVS_ERR_UNSUPPORTED, "bn: C must be 4/8/16/32/64, got %d", C);
    MVS_REQUIRE(G >= 1 && G <= 64 && Vg > 0 && bn_slots_ok(nslots), MVS_ERR_SHAPE, "bn_relu_bwd_slots: bad shape G=%d slots=%d", G, nslots);
    const size_t n4 = (size_t)Vg * C / 4;
    MVS_LAUNCH(bn_bwd_slots_kernel, dim3(ew_grid(n4), G), dim3(256), 0, stream, dy, x, stats, slots, nslots, 1.0f / (float)Vg, n4, C,
               relu, dx, dgamma, dbeta);
    return mvs_check_launch("bn_relu_bwd_slots");
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

