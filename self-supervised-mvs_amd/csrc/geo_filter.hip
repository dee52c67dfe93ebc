// the geometric-consistency filter that consumes the path's depth / confidence maps
// (jdacs/eval.py:169-224 `reproject_with_depth` + `check_geometric_consistency`, aggregated as in
// `filter_depth`, eval.py:372-396).  The reference runs it on the CPU with numpy (float64 projection chain on
// float32 camera matrices) and cv2.remap for the bilinear look-up of the source depth map; here ONE kernel does the
// whole reference view: per pixel, every source view is projected forth and back, the two tests
//   |p_reproj - p| < 1 px   and   |d_reproj - d| / d < 0.01
// are applied and the count of consistent views, the sum of the consistent reprojected depths and the per-view masks
// come out in one pass (the reference materialises ~25 [H,W] float64 temporaries per source view).
//
// Arithmetic follows the reference's dtypes: the 3x3 / 4x4 camera products are formed on the host in float32 exactly
// as numpy does (np.linalg.inv / np.matmul of float32 arrays) and handed over as doubles; the per-pixel chain runs in
// fp64; the map coordinates and the reprojected depth / coordinates are rounded to float32 where the reference casts
// (`.astype(np.float32)`); the depth comparison is a float32 comparison.  cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) is
// restated from OpenCV's remapBilinear: coordinates rounded to 1/32 pixel (INTER_BITS = 5, cvRound = round half to
// even), weights from the float32 table (1 - fy)(1 - fx) ..., taps outside the image read 0, float32 accumulation.
#include <math.h>
#include "mvs_rt.h"

struct GeoArgs {
    const float* depth_ref;          // [H,W]
    const float* depth_src[MVS_MAX_SRC];   // V x [H,W]
    const double* mats;              // ref: Kr_inv[9], Kr[9]; then per view Trs[12], Ks[9], Ks_inv[9], Tsr[12]
    int* count;                      // [H,W] number of consistent source views
    float* depth_sum;                // [H,W] float32 sequential sum of the consistent reprojected depths (0 elsewhere)
    unsigned char* masks;            // [V,H,W] per-view masks, or null
    float* reproj;                   // [V,H,W] per-view reprojected depth (0 where inconsistent), or null
    float* xy_src;                   // [V,2,H,W] float32 source-view coordinates (x2d_src, y2d_src), or null
    int H, W, V;
    float pix_thresh, rel_thresh;
};

// OpenCV remapBilinear on a float32 image with float32 maps, BORDER_CONSTANT value 0
__device__ __forceinline__ float cv_remap_linear(const float* __restrict__ img, int H, int W, float mx, float my) {
    // saturating float -> int like cvRound on the vector path; non-finite coordinates fall outside the image
    const float sxf = rintf(mx * 32.0f), syf = rintf(my * 32.0f);
    if (!(fabsf(sxf) < 1.0e9f) || !(fabsf(syf) < 1.0e9f)) return 0.0f;
    const int sx = (int)sxf, sy = (int)syf;
    const int x0 = sx >> 5, y0 = sy >> 5;
    const float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
    const float w0 = (1.0f - fy) * (1.0f - fx), w1 = (1.0f - fy) * fx, w2 = fy * (1.0f - fx), w3 = fy * fx;
    const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W, ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
    const float v0 = (xa && ya) ? img[(size_t)y0 * W + x0] : 0.0f, v1 = (xb && ya) ? img[(size_t)y0 * W + x0 + 1] : 0.0f;
    const float v2 = (xa && yb) ? img[(size_t)(y0 + 1) * W + x0] : 0.0f, v3 = (xb && yb) ? img[(size_t)(y0 + 1) * W + x0 + 1] : 0.0f;
    return v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
}

__global__ __launch_bounds__(256) void geo_consistency_kernel(GeoArgs a) {
    const int HW = a.H * a.W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / a.W, x = p - y * a.W;
    const double* __restrict__ Kri = a.mats;
    const double* __restrict__ Kr = a.mats + 9;
    const float dref = a.depth_ref[p];
    const double xd = (double)x, yd = (double)y, d = (double)dref;
    // reference camera frame: K_ref^-1 [x y 1]^T * depth   (eval.py:176-177: the pixel vector is scaled first)
    const double px = xd * d, py = yd * d, pz = d;
    const double X0 = Kri[0] * px + Kri[1] * py + Kri[2] * pz;
    const double X1 = Kri[3] * px + Kri[4] * py + Kri[5] * pz;
    const double X2 = Kri[6] * px + Kri[7] * py + Kri[8] * pz;
    int cnt = 0;
    float dsum = 0.0f;
    for (int v = 0; v < a.V; ++v) {
        const double* __restrict__ M = a.mats + 18 + 42 * v;
        const double* Trs = M; const double* Ks = M + 12; const double* Ksi = M + 21; const double* Tsr = M + 30;
        // source camera frame and pixel (eval.py:179-183)
        const double S0 = Trs[0] * X0 + Trs[1] * X1 + Trs[2] * X2 + Trs[3];
        const double S1 = Trs[4] * X0 + Trs[5] * X1 + Trs[6] * X2 + Trs[7];
        const double S2 = Trs[8] * X0 + Trs[9] * X1 + Trs[10] * X2 + Trs[11];
        const double k0 = Ks[0] * S0 + Ks[1] * S1 + Ks[2] * S2;
        const double k1 = Ks[3] * S0 + Ks[4] * S1 + Ks[5] * S2;
        const double k2 = Ks[6] * S0 + Ks[7] * S1 + Ks[8] * S2;
        const double us = k0 / k2, vs = k1 / k2;
        const float usf = (float)us, vsf = (float)vs;
        const float sampled = cv_remap_linear(a.depth_src[v], a.H, a.W, usf, vsf);       // eval.py:189
        // back to the reference view with the SAMPLED source depth (eval.py:194-205)
        const double sd = (double)sampled;
        const double qx = us * sd, qy = vs * sd, qz = sd;
        const double Y0 = Ksi[0] * qx + Ksi[1] * qy + Ksi[2] * qz;
        const double Y1 = Ksi[3] * qx + Ksi[4] * qy + Ksi[5] * qz;
        const double Y2 = Ksi[6] * qx + Ksi[7] * qy + Ksi[8] * qz;
        const double R0 = Tsr[0] * Y0 + Tsr[1] * Y1 + Tsr[2] * Y2 + Tsr[3];
        const double R1 = Tsr[4] * Y0 + Tsr[5] * Y1 + Tsr[6] * Y2 + Tsr[7];
        const double R2 = Tsr[8] * Y0 + Tsr[9] * Y1 + Tsr[10] * Y2 + Tsr[11];
        const float drep = (float)R2;
        const double r0 = Kr[0] * R0 + Kr[1] * R1 + Kr[2] * R2;
        const double r1 = Kr[3] * R0 + Kr[4] * R1 + Kr[5] * R2;
        const double r2 = Kr[6] * R0 + Kr[7] * R1 + Kr[8] * R2;
        const float xr = (float)(r0 / r2), yr = (float)(r1 / r2);
        // eval.py:213-220: pixel distance in float64 of the float32 coordinates, relative depth difference in float32
        const double ddx = (double)xr - xd, ddy = (double)yr - yd;
        const double dist = sqrt(ddx * ddx + ddy * ddy);
        const float rel = fabsf(drep - dref) / dref;
        const bool ok = dist < (double)a.pix_thresh && rel < a.rel_thresh;
        const float dmasked = ok ? drep : 0.0f;
        if (ok) ++cnt;
        dsum += dmasked;                                   // python sum(): view after view, float32
        if (a.masks) a.masks[(size_t)v * HW + p] = ok ? 1 : 0;
        if (a.reproj) a.reproj[(size_t)v * HW + p] = dmasked;
        if (a.xy_src) { a.xy_src[((size_t)v * 2 + 0) * HW + p] = usf; a.xy_src[((size_t)v * 2 + 1) * HW + p] = vsf; }
    }
    a.count[p] = cnt;
    a.depth_sum[p] = dsum;
}

// depth_ref [H,W] fp32; depth_srcs: HOST array of V device pointers to [H,W] fp32; mats: DEVICE array of 18 + 42 V doubles
// (layout in GeoArgs); outputs: count [H,W] int32, depth_sum [H,W] fp32; optional (may be NULL) masks [V,H,W] uint8,
// reproj [V,H,W] fp32, xy_src [V,2,H,W] fp32.  pix_thresh / rel_thresh: 1 and 0.01 in the reference (eval.py:220).
extern "C" int mvs_geo_consistency(const float* depth_ref, const float* const* depth_srcs, const double* mats, int V, int H, int W,
                                   float pix_thresh, float rel_thresh, int* count, float* depth_sum, unsigned char* masks,
                                   float* reproj, float* xy_src, hipStream_t stream) {
    MVS_REQUIRE(depth_ref && depth_srcs && mats && count && depth_sum, MVS_ERR_NULL, "geo_consistency: null pointer argument");
    MVS_REQUIRE(V >= 1 && V <= MVS_MAX_SRC, MVS_ERR_SHAPE, "geo_consistency: 1..%d source views, got %d", MVS_MAX_SRC, V);
    MVS_REQUIRE(H > 0 && W > 0, MVS_ERR_SHAPE, "geo_consistency: bad shape H=%d W=%d", H, W);
    GeoArgs a = {};
    a.depth_ref = depth_ref;
    for (int v = 0; v < V; ++v) {
        MVS_REQUIRE(depth_srcs[v], MVS_ERR_NULL, "geo_consistency: null source depth pointer %d", v);
        a.depth_src[v] = depth_srcs[v];
    }
    a.mats = mats; a.count = count; a.depth_sum = depth_sum; a.masks = masks; a.reproj = reproj; a.xy_src = xy_src;
    a.H = H; a.W = W; a.V = V; a.pix_thresh = pix_thresh; a.rel_thresh = rel_thresh;
    MVS_LAUNCH(geo_consistency_kernel, dim3((unsigned)((H * W + 255) / 256)), dim3(256), 0, stream, a);
    return mvs_check_launch("geo_consistency");
}
