// K1 / K2: fused differentiable homography warp + variance aggregation (forward and backward).
//
// Replaces, in ONE pass over the cost volume, the reference's per-source-view chain
//   homo_warping (jdacs/models/module.py:105-140; jdacs-ms/models/modules.py:62-104,209-261)
//   + running sum / sum-of-squares + variance (jdacs/models/mvsnet.py:120-136;
//     jdacs-ms/models/network.py:114-137), which materialises ~32 volume-sized tensors.
//
// Layout (HBM): feature maps channels-last [B,H,W,C]; cost volume channels-last-3d [B,D,H,W,C].
// A bilinear tap is then one contiguous C*4-byte run (128 B at C=32 = one cache line) and a wave
// writes 1 KiB of contiguous volume per store instruction.
//
// Thread map: 256 threads = (256/(C/4)) consecutive ref pixels x (C/4) channel quads; each thread
// walks a slab of depth planes.  The only large HBM transaction is the single float4 store of the
// variance per (voxel, quad); the N feature maps (2.6 MB each at config 2) stay L2/MALL resident.
//
// Arithmetic mirrors the reference op by op in fp32 (built with -ffp-contract=off):
//   q = rot*(x,y,1)*d + t ; p = q.xy / q.z ; g = p/((size-1)/2) - 1 ;
//   ix = ((g+1)*size - 1)/2  (align_corners=False, what F.grid_sample does on torch>=1.3, App. A Q1)
//   bilinear weights as ATen: w = ix - floor(ix), e = 1 - w ; zero padding ; no z>0 guard (Q14).
#include "mvs_rt.h"

struct SweepArgs {
    const float* ref;                 // [B,H,W,C]
    const float* src[MVS_MAX_SRC];    // NS x [B,H,W,C]
    const float* rot;                 // [B,NS,9]
    const float* trans;               // [B,NS,3]
    const float* depth;               // [B,D] or [B,D,H,W]
    float* var;                       // fwd out [B,D,H,W,C]
    const float* gvar;                // bwd in  [B,D,H,W,C]
    float* gref;                      // bwd out [B,H,W,C]   (caller zero-fills)
    float* gsrc[MVS_MAX_SRC];         // bwd out NS x [B,H,W,C] (caller zero-fills)
    int B, H, W, D, NS;
    int per_pixel, align_corners, ms_alias;
    int dslab;
    int warp_only;   // 1: write / back-propagate the warped volume of source 0 itself (homo_warping)
};

struct Taps {
    float w00, w01, w10, w11;   // nw, ne, sw, se weights (0 where the tap is outside the image)
    int o00, o01, o10, o11;     // element offsets of the taps (pixel*C), valid only if weight used
    bool v00, v01, v10, v11;
};

__device__ __forceinline__ Taps make_taps(float rx, float ry, float rz, float tx, float ty, float tz,
                                          float dep, int H, int W, int C, int align_corners) {
    float X = rx * dep;
    X = X + tx;
    float Y = ry * dep;
    Y = Y + ty;
    float Z = rz * dep;
    Z = Z + tz;
    float px = X / Z;
    float py = Y / Z;
    float gx = px / ((float)(W - 1) / 2.0f) - 1.0f;
    float gy = py / ((float)(H - 1) / 2.0f) - 1.0f;
    float ix, iy;
    if (align_corners) {
        ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
        iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    } else {
        ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
        iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    }
    float x0 = floorf(ix), y0 = floorf(iy);
    float wx = ix - x0, wy = iy - y0;
    float ex = 1.0f - wx, ey = 1.0f - wy;
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    // bounds tests in float: robust for huge / non-finite coordinates (all taps then fall outside)
    bool xin0 = (x0 >= 0.0f) && (x0 <= (float)(W - 1));
    bool xin1 = (x1 >= 0.0f) && (x1 <= (float)(W - 1));
    bool yin0 = (y0 >= 0.0f) && (y0 <= (float)(H - 1));
    bool yin1 = (y1 >= 0.0f) && (y1 <= (float)(H - 1));
    int xi0 = xin0 ? (int)x0 : 0, xi1 = xin1 ? (int)x1 : 0;
    int yi0 = yin0 ? (int)y0 : 0, yi1 = yin1 ? (int)y1 : 0;
    Taps t;
    t.v00 = xin0 && yin0;
    t.v01 = xin1 && yin0;
    t.v10 = xin0 && yin1;
    t.v11 = xin1 && yin1;
    t.w00 = ey * ex;
    t.w01 = ey * wx;
    t.w10 = wy * ex;
    t.w11 = wy * wx;
    t.o00 = (yi0 * W + xi0) * C;
    t.o01 = (yi0 * W + xi1) * C;
    t.o10 = (yi1 * W + xi0) * C;
    t.o11 = (yi1 * W + xi1) * C;
    return t;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float4 sample4(const float* __restrict__ f, const Taps& t) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t.v00) {
        float4 a = ld4(f + t.o00);
        v.x = a.x * t.w00; v.y = a.y * t.w00; v.z = a.z * t.w00; v.w = a.w * t.w00;
    }
    if (t.v01) {
        float4 a = ld4(f + t.o01);
        v.x = v.x + a.x * t.w01; v.y = v.y + a.y * t.w01; v.z = v.z + a.z * t.w01; v.w = v.w + a.w * t.w01;
    }
    if (t.v10) {
        float4 a = ld4(f + t.o10);
        v.x = v.x + a.x * t.w10; v.y = v.y + a.y * t.w10; v.z = v.z + a.z * t.w10; v.w = v.w + a.w * t.w10;
    }
    if (t.v11) {
        float4 a = ld4(f + t.o11);
        v.x = v.x + a.x * t.w11; v.y = v.y + a.y * t.w11; v.z = v.z + a.z * t.w11; v.w = v.w + a.w * t.w11;
    }
    return v;
}

// NS_T > 0: number of source views known at compile time (loops unroll, gathers of all views overlap);
// NS_T == 0: runtime a.NS.
template <int C, int NS_T>
__global__ __launch_bounds__(256) void plane_sweep_variance_fwd_kernel(SweepArgs a) {
    constexpr int QUADS = C / 4;
    constexpr int PPB = 256 / QUADS;
    const int NS = NS_T > 0 ? NS_T : a.NS;
    const int tid = threadIdx.x;
    const int q = tid % QUADS;
    const int HW = a.H * a.W;
    const int pix = blockIdx.x * PPB + tid / QUADS;
    const int b = blockIdx.z;
    if (pix >= HW) return;  // no barriers / cross-lane ops below
    const int d0 = blockIdx.y * a.dslab;
    const int d1 = min(a.D, d0 + a.dslab);
    const float xf = (float)(pix % a.W), yf = (float)(pix / a.W);
    const size_t fbase = (size_t)b * HW * C + 4 * q;
    const float4 r = ld4(a.ref + fbase + (size_t)pix * C);
    const float4 r2 = make_float4(r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w);
    const float nviews = (float)(NS + 1);
    const float* __restrict__ rotb = a.rot + (size_t)b * NS * 9;
    const float* __restrict__ trb = a.trans + (size_t)b * NS * 3;

    for (int d = d0; d < d1; ++d) {
        const float dep = a.per_pixel ? a.depth[((size_t)b * a.D + d) * HW + pix] : a.depth[b * a.D + d];
        float4 S = a.ms_alias ? r2 : r;
        float4 Q = r2;
#pragma unroll(NS_T > 0 ? NS_T : 1)
        for (int s = 0; s < (NS_T > 0 ? NS_T : MVS_MAX_SRC); ++s) {
            if (NS_T == 0 && s >= NS) break;
            const float* R = rotb + s * 9;
            const float* T = trb + s * 3;
            float rx = R[0] * xf + R[1] * yf; rx = rx + R[2];
            float ry = R[3] * xf + R[4] * yf; ry = ry + R[5];
            float rz = R[6] * xf + R[7] * yf; rz = rz + R[8];
            Taps t = make_taps(rx, ry, rz, T[0], T[1], T[2], dep, a.H, a.W, C, a.align_corners);
            float4 v = sample4(a.src[s] + fbase, t);
            S.x = S.x + v.x; S.y = S.y + v.y; S.z = S.z + v.z; S.w = S.w + v.w;
            Q.x = Q.x + v.x * v.x; Q.y = Q.y + v.y * v.y; Q.z = Q.z + v.z * v.z; Q.w = Q.w + v.w * v.w;
        }
        float4 o;
        float m;
        if (a.warp_only) {  // S = r + v  =>  v = S - r is not exact; recompute the single source directly
            const float* R = rotb;
            const float* T = trb;
            float rx = R[0] * xf + R[1] * yf; rx = rx + R[2];
            float ry = R[3] * xf + R[4] * yf; ry = ry + R[5];
            float rz = R[6] * xf + R[7] * yf; rz = rz + R[8];
            Taps t = make_taps(rx, ry, rz, T[0], T[1], T[2], dep, a.H, a.W, C, a.align_corners);
            o = sample4(a.src[0] + fbase, t);
            *reinterpret_cast<float4*>(a.var + (((size_t)b * a.D + d) * HW + pix) * C + 4 * q) = o;
            continue;
        }
        m = S.x / nviews; o.x = Q.x / nviews - m * m;
        m = S.y / nviews; o.y = Q.y / nviews - m * m;
        m = S.z / nviews; o.z = Q.z / nviews - m * m;
        m = S.w / nviews; o.w = Q.w / nviews - m * m;
        *reinterpret_cast<float4*>(a.var + (((size_t)b * a.D + d) * HW + pix) * C + 4 * q) = o;
    }
}

__device__ __forceinline__ void scatter4(float* __restrict__ g, const Taps& t, const float4& gv) {
    if (t.v00) {
        float* p = g + t.o00;
        atomicAdd(p + 0, gv.x * t.w00); atomicAdd(p + 1, gv.y * t.w00);
        atomicAdd(p + 2, gv.z * t.w00); atomicAdd(p + 3, gv.w * t.w00);
    }
    if (t.v01) {
        float* p = g + t.o01;
        atomicAdd(p + 0, gv.x * t.w01); atomicAdd(p + 1, gv.y * t.w01);
        atomicAdd(p + 2, gv.z * t.w01); atomicAdd(p + 3, gv.w * t.w01);
    }
    if (t.v10) {
        float* p = g + t.o10;
        atomicAdd(p + 0, gv.x * t.w10); atomicAdd(p + 1, gv.y * t.w10);
        atomicAdd(p + 2, gv.z * t.w10); atomicAdd(p + 3, gv.w * t.w10);
    }
    if (t.v11) {
        float* p = g + t.o11;
        atomicAdd(p + 0, gv.x * t.w11); atomicAdd(p + 1, gv.y * t.w11);
        atomicAdd(p + 2, gv.z * t.w11); atomicAdd(p + 3, gv.w * t.w11);
    }
}

// Backward.  With Sm = S/N:  dL/dv_i = g*(2/N)*(v_i - Sm);
//   dL/dr = sum_d g*(2/N)*(r - Sm)                    (MVSNet)
//   dL/dr = sum_d g*(2r/N)*(1 - 2*Sm)                 (jdacs-ms alias quirk, S starts from r^2)
// No gradient to cameras / depths (the reference builds the grid under no_grad, module.py:115).
template <int C>
__global__ __launch_bounds__(256) void plane_sweep_variance_bwd_kernel(SweepArgs a) {
    constexpr int QUADS = C / 4;
    constexpr int PPB = 256 / QUADS;
    const int NS = a.NS;
    const int tid = threadIdx.x;
    const int q = tid % QUADS;
    const int HW = a.H * a.W;
    const int pix = blockIdx.x * PPB + tid / QUADS;
    const int b = blockIdx.z;
    if (pix >= HW) return;
    const int d0 = blockIdx.y * a.dslab;
    const int d1 = min(a.D, d0 + a.dslab);
    const float xf = (float)(pix % a.W), yf = (float)(pix / a.W);
    const size_t fbase = (size_t)b * HW * C + 4 * q;
    const float4 r = ld4(a.ref + fbase + (size_t)pix * C);
    const float nviews = (float)(NS + 1);
    const float two_n = 2.0f / nviews;
    const float* __restrict__ rotb = a.rot + (size_t)b * NS * 9;
    const float* __restrict__ trb = a.trans + (size_t)b * NS * 3;
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int d = d0; d < d1; ++d) {
        const float dep = a.per_pixel ? a.depth[((size_t)b * a.D + d) * HW + pix] : a.depth[b * a.D + d];
        const float4 g = ld4(a.gvar + (((size_t)b * a.D + d) * HW + pix) * C + 4 * q);
        float4 S = a.ms_alias ? make_float4(r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w) : r;
        for (int s = 0; s < NS; ++s) {
            const float* R = rotb + s * 9;
            const float* T = trb + s * 3;
            float rx = R[0] * xf + R[1] * yf; rx = rx + R[2];
            float ry = R[3] * xf + R[4] * yf; ry = ry + R[5];
            float rz = R[6] * xf + R[7] * yf; rz = rz + R[8];
            Taps t = make_taps(rx, ry, rz, T[0], T[1], T[2], dep, a.H, a.W, C, a.align_corners);
            float4 v = sample4(a.src[s] + fbase, t);
            S.x += v.x; S.y += v.y; S.z += v.z; S.w += v.w;
        }
        const float4 Sm = make_float4(S.x / nviews, S.y / nviews, S.z / nviews, S.w / nviews);
        for (int s = 0; s < NS; ++s) {
            const float* R = rotb + s * 9;
            const float* T = trb + s * 3;
            float rx = R[0] * xf + R[1] * yf; rx = rx + R[2];
            float ry = R[3] * xf + R[4] * yf; ry = ry + R[5];
            float rz = R[6] * xf + R[7] * yf; rz = rz + R[8];
            Taps t = make_taps(rx, ry, rz, T[0], T[1], T[2], dep, a.H, a.W, C, a.align_corners);
            float4 v = sample4(a.src[s] + fbase, t);
            float4 gv = make_float4(g.x * two_n * (v.x - Sm.x), g.y * two_n * (v.y - Sm.y),
                                    g.z * two_n * (v.z - Sm.z), g.w * two_n * (v.w - Sm.w));
            if (a.warp_only) gv = g;
            scatter4(a.gsrc[s] + fbase, t, gv);
        }
        if (a.ms_alias) {
            gr.x += g.x * two_n * r.x * (1.0f - 2.0f * Sm.x);
            gr.y += g.y * two_n * r.y * (1.0f - 2.0f * Sm.y);
            gr.z += g.z * two_n * r.z * (1.0f - 2.0f * Sm.z);
            gr.w += g.w * two_n * r.w * (1.0f - 2.0f * Sm.w);
        } else {
            gr.x += g.x * two_n * (r.x - Sm.x);
            gr.y += g.y * two_n * (r.y - Sm.y);
            gr.z += g.z * two_n * (r.z - Sm.z);
            gr.w += g.w * two_n * (r.w - Sm.w);
        }
    }
    if (a.warp_only) return;
    float* p = a.gref + fbase + (size_t)pix * C;
    atomicAdd(p + 0, gr.x); atomicAdd(p + 1, gr.y); atomicAdd(p + 2, gr.z); atomicAdd(p + 3, gr.w);
}

// ------------------------------------------------------------------------------------------------
template <int C>
static int launch_fwd(const SweepArgs& a, hipStream_t st) {
    constexpr int PPB = 256 / (C / 4);
    dim3 grid(mvs_cdiv(a.H * a.W, PPB), mvs_cdiv(a.D, a.dslab), a.B), block(256);
    switch (a.NS) {
        case 1: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 1>), grid, block, 0, st, a); break;
        case 2: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 2>), grid, block, 0, st, a); break;
        case 3: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 3>), grid, block, 0, st, a); break;
        case 4: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 4>), grid, block, 0, st, a); break;
        case 6: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 6>), grid, block, 0, st, a); break;
        default: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 0>), grid, block, 0, st, a); break;
    }
    return mvs_check_launch("plane_sweep_variance_fwd");
}

template <int C>
static int launch_bwd(const SweepArgs& a, hipStream_t st) {
    constexpr int PPB = 256 / (C / 4);
    dim3 grid(mvs_cdiv(a.H * a.W, PPB), mvs_cdiv(a.D, a.dslab), a.B), block(256);
    MVS_LAUNCH((plane_sweep_variance_bwd_kernel<C>), grid, block, 0, st, a);
    return mvs_check_launch("plane_sweep_variance_bwd");
}

static int pick_dslab(int B, int H, int W, int D, int C) {
    // enough workgroups to fill 256 CUs several times over, while keeping per-thread setup amortised
    long blocks_xy = (long)mvs_cdiv(H * W, 256 / (C / 4)) * B;
    int slab = D;
    while (slab > 8 && blocks_xy * mvs_cdiv(D, slab) < 4096) slab = (slab + 1) / 2;
    return slab;
}

static int fill_args(SweepArgs& a, const float* ref, const float* const* srcs, const float* rot, const float* trans,
                     const float* depth, int depth_is_per_pixel, int B, int N, int C, int D, int H, int W,
                     int align_corners, int ms_alias) {
    MVS_REQUIRE(ref && srcs && rot && trans && depth, MVS_ERR_NULL, "plane_sweep: null pointer argument");
    MVS_REQUIRE(N >= 2 && N - 1 <= MVS_MAX_SRC, MVS_ERR_SHAPE, "plane_sweep: need 2 <= N <= %d views, got %d",
                MVS_MAX_SRC + 1, N);
    MVS_REQUIRE(C == 8 || C == 16 || C == 32, MVS_ERR_UNSUPPORTED, "plane_sweep: C must be 8, 16 or 32, got %d", C);
    MVS_REQUIRE(B > 0 && D > 0 && H > 1 && W > 1, MVS_ERR_SHAPE, "plane_sweep: bad shape B=%d D=%d H=%d W=%d", B, D, H, W);
    a.ref = ref;
    for (int s = 0; s < N - 1; ++s) {
        MVS_REQUIRE(srcs[s], MVS_ERR_NULL, "plane_sweep: null source feature pointer %d", s);
        a.src[s] = srcs[s];
    }
    a.rot = rot; a.trans = trans; a.depth = depth;
    a.B = B; a.H = H; a.W = W; a.D = D; a.NS = N - 1;
    a.per_pixel = depth_is_per_pixel; a.align_corners = align_corners; a.ms_alias = ms_alias;
    a.dslab = pick_dslab(B, H, W, D, C);
    return MVS_OK;
}

extern "C" int mvs_plane_sweep_variance_fwd(const float* ref, const float* const* srcs, const float* rot,
                                            const float* trans, const float* depth, int depth_is_per_pixel,
                                            int B, int N, int C, int D, int H, int W, int align_corners,
                                            int ms_alias, float* var_out, hipStream_t stream) {
    SweepArgs a = {};
    int rc = fill_args(a, ref, srcs, rot, trans, depth, depth_is_per_pixel, B, N, C, D, H, W, align_corners, ms_alias);
    if (rc) return rc;
    MVS_REQUIRE(var_out, MVS_ERR_NULL, "plane_sweep fwd: null output");
    a.var = var_out;
    if (C == 32) return launch_fwd<32>(a, stream);
    if (C == 16) return launch_fwd<16>(a, stream);
    return launch_fwd<8>(a, stream);
}

extern "C" int mvs_plane_sweep_variance_bwd(const float* grad_var, const float* ref, const float* const* srcs,
                                            const float* rot, const float* trans, const float* depth,
                                            int depth_is_per_pixel, int B, int N, int C, int D, int H, int W,
                                            int align_corners, int ms_alias, float* grad_ref,
                                            float* const* grad_srcs, hipStream_t stream) {
    SweepArgs a = {};
    int rc = fill_args(a, ref, srcs, rot, trans, depth, depth_is_per_pixel, B, N, C, D, H, W, align_corners, ms_alias);
    if (rc) return rc;
    MVS_REQUIRE(grad_var && grad_ref && grad_srcs, MVS_ERR_NULL, "plane_sweep bwd: null pointer argument");
    a.gvar = grad_var; a.gref = grad_ref;
    for (int s = 0; s < N - 1; ++s) {
        MVS_REQUIRE(grad_srcs[s], MVS_ERR_NULL, "plane_sweep bwd: null grad pointer %d", s);
        a.gsrc[s] = grad_srcs[s];
    }
    if (C == 32) return launch_bwd<32>(a, stream);
    if (C == 16) return launch_bwd<16>(a, stream);
    return launch_bwd<8>(a, stream);
}

// ---- homo_warping alone (jdacs/models/module.py:105-140): warped volume of ONE source view ----
extern "C" int mvs_homo_warp_fwd(const float* src, const float* rot, const float* trans, const float* depth,
                                 int depth_is_per_pixel, int B, int C, int D, int H, int W, int align_corners,
                                 float* warped_out, hipStream_t stream) {
    SweepArgs a = {};
    const float* srcs[1] = {src};
    int rc = fill_args(a, src, srcs, rot, trans, depth, depth_is_per_pixel, B, 2, C, D, H, W, align_corners, 0);
    if (rc) return rc;
    MVS_REQUIRE(warped_out, MVS_ERR_NULL, "homo_warp fwd: null output");
    a.var = warped_out;
    a.warp_only = 1;
    if (C == 32) return launch_fwd<32>(a, stream);
    if (C == 16) return launch_fwd<16>(a, stream);
    return launch_fwd<8>(a, stream);
}

// grad_src [B,H,W,C] must be zero-filled by the caller
extern "C" int mvs_homo_warp_bwd(const float* grad_warped, const float* src, const float* rot, const float* trans,
                                 const float* depth, int depth_is_per_pixel, int B, int C, int D, int H, int W,
                                 int align_corners, float* grad_src, hipStream_t stream) {
    SweepArgs a = {};
    const float* srcs[1] = {src};
    int rc = fill_args(a, src, srcs, rot, trans, depth, depth_is_per_pixel, B, 2, C, D, H, W, align_corners, 0);
    if (rc) return rc;
    MVS_REQUIRE(grad_warped && grad_src, MVS_ERR_NULL, "homo_warp bwd: null pointer argument");
    a.gvar = grad_warped; a.gref = grad_src; a.gsrc[0] = grad_src;
    a.warp_only = 1;
    if (C == 32) return launch_bwd<32>(a, stream);
    if (C == 16) return launch_bwd<16>(a, stream);
    return launch_bwd<8>(a, stream);
}
