// Shared by plane_sweep.hip (K1 + the round-1/2 backward forms) and plane_sweep_bwd.hip (K2, projection-table form):
// launch arguments, the projection of a reference pixel into a source view, footprint windows.
#pragma once
#include "mvs_rt.h"

struct SweepArgs {
    const float* ref;                 // [B,H,W,C]
    const float* src[MVS_MAX_SRC];    // NS x [B,H,W,C]
    const float* rot;                 // [B,NS,9]
    const float* trans;               // [B,NS,3]
    const float* depth;               // [B,D] or [B,D,H,W]
    float* var;                       // fwd out [B,D,H,W,C]
    const float* gvar;                // bwd in  [B,D,H,W,C]
    float* gref;                      // bwd out [B,H,W,C]
    float* gsrc[MVS_MAX_SRC];         // bwd out NS x [B,H,W,C] (caller zero-fills; accumulated atomically)
    int B, H, W, D, NS;
    int per_pixel, align_corners, ms_alias;
    int dslab;
    int warp_only;   // 1: write / back-propagate the warped volume of source 0 itself (homo_warping)
    float sx, ox, sy, oy;  // ix = px*sx + ox, iy = py*sy + oy
    int tiles_x, tiles_y;
    int tile_w;      // cached forward kernel: pixels per tile row (tile = tile_w x PPB/tile_w)
    int no_window;   // test knob "bwd_nowin": per-wave-window backward sends every flush down its global-atomic path
    int gpf_late;    // knob "bwd_gpf": the per-wave-window backward requests the next planes' upstream gradient after the plane's gathers
    int bf16_out;    // forward: the volume is stored in bf16 (inference path)
    int nt_store;    // stream the volume with non-temporal stores (written once, read by the next kernel from HBM anyway)
    int xcd;         // knob "sweep_xcd": XCD-compact workgroup order (sweep_wg below)
};

// Workgroup -> (pixel tile, depth slab, batch).  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each
// with its own 4 MB L2: with the plain blockIdx decoding every XCD walks the WHOLE image (every 8th tile) and its L2 has to hold
// the source footprints of a full-width band of all views.  With a.xcd the ids are re-dealt so that XCD k owns one contiguous
// eighth of the (batch, slab, tile) sequence: the workgroups resident on an XCD at any time are neighbouring tiles, whose taps
// share texels.  A bijection for any grid size (XCD x owns T/8 ids, the first T%8 XCDs one more).
struct SweepWg { int tile, slab, b; };
__device__ __forceinline__ SweepWg sweep_wg(const SweepArgs& a) {
    SweepWg w;
    if (!a.xcd) { w.tile = blockIdx.x; w.slab = blockIdx.y; w.b = blockIdx.z; return w; }
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    const unsigned lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned per = total >> 3, rem = total & 7u, x = lin & 7u, i = lin >> 3;
    const unsigned nw = (x < rem ? x * (per + 1) : rem * (per + 1) + (x - rem) * per) + i;
    w.tile = (int)(nw % nx); w.slab = (int)((nw / nx) % ny); w.b = (int)(nw / (nx * ny));
    return w;
}

struct Taps {
    float w00, w01, w10, w11;   // nw, ne, sw, se weights
    int x0, y0;                 // floor(ix), floor(iy), clamped to [-2, size] so int math is safe
    bool v00, v01, v10, v11;    // tap inside the image
};

__device__ __forceinline__ void source_index(const float* __restrict__ R, const float* __restrict__ T, float xf,
                                             float yf, float dep, const SweepArgs& a, float& ix, float& iy) {
    float rx = fmaf(R[0], xf, fmaf(R[1], yf, R[2]));
    float ry = fmaf(R[3], xf, fmaf(R[4], yf, R[5]));
    float rz = fmaf(R[6], xf, fmaf(R[7], yf, R[8]));
    float X = fmaf(rx, dep, T[0]);
    float Y = fmaf(ry, dep, T[1]);
    float Z = fmaf(rz, dep, T[2]);
    float iz = 1.0f / Z;
    ix = fmaf(X * iz, a.sx, a.ox);
    iy = fmaf(Y * iz, a.sy, a.oy);
}

__device__ __forceinline__ Taps make_taps(float ix, float iy, int H, int W) {
    float x0 = floorf(ix), y0 = floorf(iy);
    float wx = ix - x0, wy = iy - y0;
    float ex = 1.0f - wx, ey = 1.0f - wy;
    // clamp in float first: robust for huge / non-finite coordinates (all taps then fall outside)
    float x0c = fminf(fmaxf(x0, -2.0f), (float)W);
    float y0c = fminf(fmaxf(y0, -2.0f), (float)H);
    Taps t;
    t.x0 = (x0c == x0c) ? (int)x0c : -2;  // NaN -> outside
    t.y0 = (y0c == y0c) ? (int)y0c : -2;
    const bool xin0 = t.x0 >= 0 && t.x0 < W, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    const bool yin0 = t.y0 >= 0 && t.y0 < H, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    t.v00 = xin0 && yin0; t.v01 = xin1 && yin0; t.v10 = xin0 && yin1; t.v11 = xin1 && yin1;
    t.w00 = ey * ex; t.w01 = ey * wx; t.w10 = wy * ex; t.w11 = wy * wx;
    return t;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// f: base of this batch item's map + 4*quad
__device__ __forceinline__ float4 sample4(const float* __restrict__ f, const Taps& t, int W, int C) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int o = (t.y0 * W + t.x0) * C;
    if (t.v00) { float4 a = ld4(f + o); v.x = a.x * t.w00; v.y = a.y * t.w00; v.z = a.z * t.w00; v.w = a.w * t.w00; }
    if (t.v01) {
        float4 a = ld4(f + o + C);
        v.x = fmaf(a.x, t.w01, v.x); v.y = fmaf(a.y, t.w01, v.y); v.z = fmaf(a.z, t.w01, v.z); v.w = fmaf(a.w, t.w01, v.w);
    }
    if (t.v10) {
        float4 a = ld4(f + o + W * C);
        v.x = fmaf(a.x, t.w10, v.x); v.y = fmaf(a.y, t.w10, v.y); v.z = fmaf(a.z, t.w10, v.z); v.w = fmaf(a.w, t.w10, v.w);
    }
    if (t.v11) {
        float4 a = ld4(f + o + W * C + C);
        v.x = fmaf(a.x, t.w11, v.x); v.y = fmaf(a.y, t.w11, v.y); v.z = fmaf(a.z, t.w11, v.z); v.w = fmaf(a.w, t.w11, v.w);
    }
    return v;
}

// pixel tile of a workgroup: 256/(C/4) pixels as TW x TH (C=32: 8x4, C=16: 8x8, C=8: 16x8)
template <int C> struct Tile { static constexpr int TW = C == 8 ? 16 : 8, TH = (256 / (C / 4)) / TW; };

// ---- footprint windows (shared by the LDS-staged forward and the LDS-privatised backward) ----
template <int C> struct BwdCfg { static constexpr int WCAP = C == 32 ? 240 : (C == 16 ? 480 : 900), CP = C + 1; };

struct Win { int x0, y0, w, h; };

__device__ __forceinline__ void corner_bounds(const SweepArgs& a, const float* R, const float* T, float xa, float xb,
                                              float ya, float yb, float da, float db, float& lox, float& hix,
                                              float& loy, float& hiy) {
    lox = loy = 3.0e38f;
    hix = hiy = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float ix, iy;
        source_index(R, T, (k & 1) ? xb : xa, (k & 2) ? yb : ya, (k & 4) ? db : da, a, ix, iy);
        lox = fminf(lox, ix); hix = fmaxf(hix, ix); loy = fminf(loy, iy); hiy = fmaxf(hiy, iy);
    }
}

__device__ __forceinline__ Win make_window(const SweepArgs& a, float lox, float hix, float loy, float hiy) {
    // taps touch floor(lo) .. floor(hi)+1; clip to the image (outside taps are dropped anyway)
    float fx0 = fminf(fmaxf(floorf(lox), 0.0f), (float)(a.W - 1)), fx1 = fminf(fmaxf(floorf(hix) + 1.0f, 0.0f), (float)(a.W - 1));
    float fy0 = fminf(fmaxf(floorf(loy), 0.0f), (float)(a.H - 1)), fy1 = fminf(fmaxf(floorf(hiy) + 1.0f, 0.0f), (float)(a.H - 1));
    Win w;
    if (!(lox == lox) || !(hix == hix) || !(loy == loy) || !(hiy == hiy)) { w.x0 = w.y0 = 0; w.w = w.h = 1 << 14; return w; }
    w.x0 = (int)fx0; w.y0 = (int)fy0; w.w = (int)fx1 - w.x0 + 1; w.h = (int)fy1 - w.y0 + 1;
    return w;
}


