// the depth-map fusion kernel of `fusibile` (the one native GPU program of the reference) for gfx950.
//
// Restates  /root/reference/jdacs/fusion/fusibile/fusibile.cu:138-277  (kernel `fusibile`: one thread per pixel of the reference
// camera; back-project with the pixel's depth, project into every other view, compare disparities and normals, average the
// consistent 3-D points / normals / colours) with the helper arithmetic of fusibile.cu:41-133, config.h:147-188 (matvecmul4,
// matvecmul4P: plain float multiply-adds in this order, no FMA contraction: the build uses -ffp-contract=off) and
// vector_operations.h (float4 operators that drop w).  The host loop of fusibile.cu:322-440 (one launch per camera, compaction of
// the points whose three coordinates are all non-zero) is jdacs/fusion/depthfusion.py::run_fusibile.
//
// Texture fetches.  The reference reads normals+depth and colour images through CUDA texture objects created with
// cudaFilterModeLinear, unnormalised coordinates and cudaAddressModeWrap (main.cpp:491,541) -- with unnormalised coordinates the
// hardware clamps (wrap is only defined for normalised ones).  The CUDA programming guide's linear-filtering rule is restated
// here:  xB = x - 0.5, i = floor(xB), alpha = frac(xB) kept in 9-bit fixed point with 8 fractional bits (round to nearest, so 1.0
// is representable),  tex = (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1],  indices clamped to the image.
// A fetch at (p + 0.5) therefore returns texel p exactly.  PARITY UNPINNED for this leg: there is no CUDA device (nor the
// program's OpenCV / CUDA build chain) in the build container to generate vectors from; the oracle (oracle/fusibile_np.py) and
// this kernel share one reading of the published rule, everything around the fetch follows the reference line by line.
#include "mvs_rt.h"

struct FusibileArgs {
    const float* nd;        // [V,H,W,4] normal.xyz, depth
    const float* img;       // [V,H,W,4] colour (b, g, r, alpha as OpenCV loads it) or nullptr
    const float* cams;      // [V,32]: P[12] (3x4 row-major), M_inv[9], P_col34[3], C[3], pad
    const int* subset;      // [n_subset] view ids (camParams.viewSelectionSubset)
    int n_subset, V, H, W, ref;
    float f, depth_thresh, normal_thresh;
    int num_consistent, save_texture;
    float* out;             // [H,W,12]: coord.xyzw, normal.xyzw, texture4
};

struct F4 { float x, y, z, w; };

__device__ __forceinline__ F4 ld_f4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    F4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}

// tex2D<float4>(tex, x, y) with linear filtering, unnormalised coordinates, clamped addressing (see the header)
__device__ __forceinline__ F4 tex_linear(const float* __restrict__ t, int H, int W, float x, float y) {
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    // 1.8 fixed point, round to nearest
    const float a = floorf((xb - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const float b = floorf((yb - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const int i0 = min(max((int)fx, 0), W - 1), i1 = min(max((int)fx + 1, 0), W - 1);
    const int j0 = min(max((int)fy, 0), H - 1), j1 = min(max((int)fy + 1, 0), H - 1);
    const F4 t00 = ld_f4(t + ((size_t)j0 * W + i0) * 4), t10 = ld_f4(t + ((size_t)j0 * W + i1) * 4);
    const F4 t01 = ld_f4(t + ((size_t)j1 * W + i0) * 4), t11 = ld_f4(t + ((size_t)j1 * W + i1) * 4);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    F4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}

// get3Dpoint_cu (fusibile.cu:57-66): M_inv * (depth * (x, y, 1) - P_col34)
__device__ __forceinline__ void get3d(const float* __restrict__ cam, int px, int py, float depth, float& X, float& Y, float& Z) {
    const float* Mi = cam + 12;
    const float ptx = depth * (float)px - cam[21], pty = depth * (float)py - cam[22], ptz = depth - cam[23];
    X = Mi[0] * ptx + Mi[1] * pty + Mi[2] * ptz;
    Y = Mi[3] * ptx + Mi[4] * pty + Mi[5] * ptz;
    Z = Mi[6] * ptx + Mi[7] * pty + Mi[8] * ptz;
}

__global__ __launch_bounds__(256) void fusibile_fuse_kernel(FusibileArgs a) {
    const int px = blockIdx.x * 32 + (threadIdx.x & 31), py = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (px >= a.W || py >= a.H) return;
    const size_t center = (size_t)py * a.W + px;
    const size_t vstride = (size_t)a.H * a.W * 4;
    const float* camr = a.cams + (size_t)a.ref * 32;
    // fetch at (p + 0.5): texel p itself (fusibile.cu:158, 174)
    const F4 normal = ld_f4(a.nd + (size_t)a.ref * vstride + center * 4);
    float depth = normal.w;
    float Xx, Xy, Xz;
    get3d(camr, px, py, depth, Xx, Xy, Xz);
    float cXx = Xx, cXy = Xy, cXz = Xz;                       // consistent_X
    float cnx = normal.x, cny = normal.y, cnz = normal.z;     // consistent_normal (the float4 operators drop w)
    F4 ctex;
    ctex.x = ctex.y = ctex.z = ctex.w = 0.f;
    if (a.img) ctex = ld_f4(a.img + (size_t)a.ref * vstride + center * 4);
    bool tex_w_kept = true;                                   // operator+ / operator/ zero w: w survives only without any operation
    int number_consistent = 0;
    for (int i = 0; i < a.n_subset; ++i) {
        const int cur = a.subset[i];
        if (cur == a.ref) continue;
        const float* camc = a.cams + (size_t)cur * 32;
        // project_on_camera (fusibile.cu:127-133, matvecmul4P)
        const float tx = camc[0] * Xx + camc[1] * Xy + camc[2] * Xz + camc[3];
        const float ty = camc[4] * Xx + camc[5] * Xy + camc[6] * Xz + camc[7];
        const float tz = camc[8] * Xx + camc[9] * Xy + camc[10] * Xz + camc[11];
        const float ptx = tx / tz, pty = ty / tz;
        depth = tz;
        if (ptx >= 0 && ptx < (float)a.W && pty >= 0 && pty < (float)a.H) {
            const F4 nd = tex_linear(a.nd + (size_t)cur * vstride, a.H, a.W, ptx + 0.5f, pty + 0.5f);
            // disparityDepthConversion_cu2 (fusibile.cu:52-55): f * |C_ref - C_cur| / d
            const float dx = camr[24] - camc[24], dy = camr[25] - camc[25], dz = camr[26] - camc[26];
            const float baseline = sqrtf(dx * dx + dy * dy + dz * dz);
            const float depth_disp = a.f * baseline / depth;
            const float nd_disp = a.f * baseline / nd.w;
            if (fabsf(depth_disp - nd_disp) < a.depth_thresh) {
                float angle = acosf(nd.x * normal.x + nd.y * normal.y + nd.z * normal.z);   // getAngle_cu
                if (angle != angle) angle = 0.0f;
                if (angle < a.normal_thresh) {
                    float tXx, tXy, tXz;
                    get3d(camc, (int)ptx, (int)pty, nd.w, tXx, tXy, tXz);
                    cXx = cXx + tXx; cXy = cXy + tXy; cXz = cXz + tXz;
                    cnx = cnx + nd.x; cny = cny + nd.y; cnz = cnz + nd.z;
                    if (a.save_texture && a.img) {
                        const F4 tc = tex_linear(a.img + (size_t)cur * vstride, a.H, a.W, ptx + 0.5f, pty + 0.5f);
                        ctex.x = ctex.x + tc.x; ctex.y = ctex.y + tc.y; ctex.z = ctex.z + tc.z;
                    }
                    tex_w_kept = false;
                    ++number_consistent;
                }
            }
        }
    }
    const float div = (float)number_consistent + 1.0f;
    cXx = cXx / div; cXy = cXy / div; cXz = cXz / div;
    cnx = cnx / div; cny = cny / div; cnz = cnz / div;
    ctex.x = ctex.x / div; ctex.y = ctex.y / div; ctex.z = ctex.z / div;
    (void)tex_w_kept;
    float* o = a.out + center * 12;
    if (number_consistent >= a.num_consistent) {
        o[0] = cXx; o[1] = cXy; o[2] = cXz; o[3] = 0.f;
        o[4] = cnx; o[5] = cny; o[6] = cnz; o[7] = 0.f;
        o[8] = ctex.x; o[9] = ctex.y; o[10] = ctex.z; o[11] = 0.f;
    } else {
        // the reference leaves the (zero-initialised, reset after every camera) point untouched: fusibile.cu:252, 305
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = 0.f;
    }
}

// One launch = one reference camera (fusibile.cu:416-421).  out [H,W,12] is fully overwritten.
extern "C" int mvs_fusibile_fuse(const float* normals_depths, const float* images, const float* cams, const int* subset, int n_subset,
                                 int V, int H, int W, int ref_camera, float f, float depth_thresh, float normal_thresh,
                                 int num_consistent, int save_texture, float* out_points, hipStream_t stream) {
    MVS_REQUIRE(normals_depths && cams && subset && out_points, MVS_ERR_NULL, "fusibile: null pointer argument");
    MVS_REQUIRE(V >= 1 && H >= 1 && W >= 1 && n_subset >= 0 && n_subset <= V, MVS_ERR_SHAPE, "fusibile: bad shape V=%d H=%d W=%d subset=%d",
                V, H, W, n_subset);
    MVS_REQUIRE(ref_camera >= 0 && ref_camera < V, MVS_ERR_SHAPE, "fusibile: reference camera %d outside 0..%d", ref_camera, V - 1);
    FusibileArgs a = {};
    a.nd = normals_depths; a.img = images; a.cams = cams; a.subset = subset; a.n_subset = n_subset;
    a.V = V; a.H = H; a.W = W; a.ref = ref_camera; a.f = f; a.depth_thresh = depth_thresh; a.normal_thresh = normal_thresh;
    a.num_consistent = num_consistent; a.save_texture = save_texture; a.out = out_points;
    MVS_LAUNCH(fusibile_fuse_kernel, dim3(mvs_cdiv(W, 32), mvs_cdiv(H, 8)), dim3(256), 0, stream, a);
    return mvs_check_launch("fusibile_fuse");
}
