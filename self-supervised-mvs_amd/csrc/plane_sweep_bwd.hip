// K2, projection-table form (round 3; default for 1..4 source views): backward of the fused homography warp + variance.
//
// What bounded the per-wave-window kernel of round 2 (plane_sweep.hip: 195 vector + 93 scalar instructions per (wave, plane),
// profiles/r02_run19_pmc_sq_sweep_kernels.json): of its ~1560 lane-operations per (pixel, plane) only ~830 are channel
// arithmetic.  The rest is per-PIXEL work replicated over the 8 channel lanes of a pixel on every plane (homography, reciprocal,
// floor, bilinear weights: ~60 instructions for two views), change detection, and a flush that serialises the wave over its 8
// pixel groups with ~100 instructions per iteration; and 130-200 registers left 2 waves per SIMD to hide any of it.
//
// Here the per-pixel work is taken OUT of the plane loop:
//  * a wave walks PW = 64 / (C / CPL) pixels side by side (CPL = channels per lane: 1, 2 or 4 -> 2, 4 or 8 pixels at C = 32) and the
//    8 pixels of its 4x2 block as 8 / PW groups one after the other, all into the same per-wave LDS windows;
//  * before every batch of TB planes the wave computes a TABLE in LDS -- for every (pixel of the group, view, plane) the four
//    bilinear weights and the packed base texel -- with the 64 lanes spread over (row, plane): the projection arithmetic runs
//    ONCE per (pixel, view, plane) instead of once per channel lane (~2 instructions per (wave, plane) instead of ~60);
//  * in the plane loop a lane reads its pixel's table entry with two broadcast LDS reads per view (ds_read_b128 + ds_read_b32),
//    compares the packed texel with the one its register-resident 2x2 block belongs to (ONE v_cmp per view), and runs the channel
//    arithmetic: ~26 instructions per channel for two views, nothing else;
//  * a block change (every ~20 planes at DTU-like geometry) takes the slow path: re-gather (CPL floats per tap: a pixel's lanes
//    read one contiguous 128-byte texel) + flush of the accumulators into the wave's window, pixel groups one after the other
//    (plain read-add-write, no LDS atomics: plane_sweep.hip explains why);
//  * 1 channel per lane needs ~50 registers: the waves per SIMD are limited by the LDS windows, not by registers.
// Windows, depth segments, the summed write-out with coalesced global atomics and the fallbacks (tap outside the window ->
// global atomic; footprint larger than a window -> shorter segment) are those of the round-2 kernel.
//
// Math (reference: jdacs/models/module.py:105-140 backward + mvsnet.py:120-136):
//   Sm = S/N;  dL/dv_i = g*(2/N)*(v_i - Sm);  dL/dr = sum_d g*(2/N)*(r - Sm)  (alias quirk: g*(2r/N)*(1 - 2 Sm)).
#include <stdlib.h>
#include <string.h>
#include "plane_sweep_common.h"

template <int N> struct VecN { float v[N]; };

template <int N> __device__ __forceinline__ VecN<N> ldn(const float* p) {
    VecN<N> r;
    if constexpr (N == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else if constexpr (N == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
    else r.v[0] = *p;
    return r;
}
template <int N> __device__ __forceinline__ void stn(float* p, const VecN<N>& x) {
    if constexpr (N == 4) { float4 t; t.x = x.v[0]; t.y = x.v[1]; t.z = x.v[2]; t.w = x.v[3]; *reinterpret_cast<float4*>(p) = t; }
    else if constexpr (N == 2) { float2 t; t.x = x.v[0]; t.y = x.v[1]; *reinterpret_cast<float2*>(p) = t; }
    else *p = x.v[0];
}
template <int N> __device__ __forceinline__ VecN<N> zeron() {
    VecN<N> r;
#pragma unroll
    for (int k = 0; k < N; ++k) r.v[k] = 0.f;
    return r;
}
#if defined(MVS_CPU_EMUL)
#define MVS_PINN(x) ((void)0)
#else
template <int N> __device__ __forceinline__ void pinn(VecN<N>& x) {
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(x.v[k]));
}
#define MVS_PINN(x) pinn(x)
#endif

template <int C, int CPL> struct TbCfg {
    static constexpr int LPP = C / CPL;              // lanes per pixel
    static constexpr int PW = 64 / LPP;              // pixels a wave walks side by side
    static constexpr int BW = 4, BH = 2, PB = 8;     // the wave's pixel block ...
    static constexpr int NG = PB / PW;               // ... walked as NG groups one after the other
    static_assert(C % CPL == 0 && LPP >= 8 && LPP <= 32 && PB % PW == 0, "unsupported channels-per-lane for this channel count");
};

#define TB_NOBLOCK 0x7fff7fff                        // "no block held": not a packed base texel (the launcher keeps W, H < 32000)

__device__ __forceinline__ int tb_pack(int x0, int y0, int H, int W) {
    // every x0 <= -2 (>= W) puts both taps of the row outside the image: clamped so that the pair fits 2 x 16 bits
    x0 = max(-2, min(x0, W));
    y0 = max(-2, min(y0, H));
    return (x0 & 0xffff) | (int)((unsigned)y0 << 16);
}
__device__ __forceinline__ int tb_x(int p) { return (int)(short)(p & 0xffff); }
__device__ __forceinline__ int tb_y(int p) { return p >> 16; }

// The pixel groups (LPP consecutive lanes) whose `want` is set add their accumulators acc[tap] to the wave's window, one group per
// iteration of a wave-uniform loop (two pixels of a wave share texels: a plain read-add-write needs them one after the other;
// a wave's DS queue is in order).  cxy = packed base texel of the block being left.  win / gp include the lane's channel offset.
template <int C, int CPL, int LPP>
__device__ __forceinline__ void tb_flush_groups(bool want, int lane, int cxy, const VecN<CPL> (&acc)[4], int H, int W,
                                                float* __restrict__ win, const Win& w, bool use_win, float* __restrict__ gp) {
    unsigned long long m = MVS_BALLOT(want);
    while (m) {
        const int grp = (MVS_FFSLL(m) - 1) / LPP;
        if (lane / LPP == grp) {
            const int cx = tb_x(cxy), cy = tb_y(cxy);
            const int lx = cx - w.x0, ly = cy - w.y0;
            if (use_win && lx >= 0 && lx + 1 < w.w && ly >= 0 && ly + 1 < w.h && cx >= 0 && cx + 1 < W && cy >= 0 && cy + 1 < H) {
                // common case: the whole 2x2 block inside the image and the window; all reads before the first add
                float* p0 = win + (ly * w.w + lx) * C;
                float* p1 = p0 + w.w * C;
                VecN<CPL> a00 = ldn<CPL>(p0), a01 = ldn<CPL>(p0 + C), a10 = ldn<CPL>(p1), a11 = ldn<CPL>(p1 + C);
#pragma unroll
                for (int k = 0; k < CPL; ++k) { a00.v[k] += acc[0].v[k]; a01.v[k] += acc[1].v[k]; a10.v[k] += acc[2].v[k]; a11.v[k] += acc[3].v[k]; }
                stn<CPL>(p0, a00); stn<CPL>(p0 + C, a01); stn<CPL>(p1, a10); stn<CPL>(p1 + C, a11);
            } else {
                const bool xin0 = cx >= 0 && cx < W, xin1 = cx + 1 >= 0 && cx + 1 < W;
                const bool yin0 = cy >= 0 && cy < H, yin1 = cy + 1 >= 0 && cy + 1 < H;
                const bool wx0 = lx >= 0 && lx < w.w, wx1 = lx + 1 >= 0 && lx + 1 < w.w;
                const bool wy0 = ly >= 0 && ly < w.h, wy1 = ly + 1 >= 0 && ly + 1 < w.h;
                const bool img[4] = {xin0 && yin0, xin1 && yin0, xin0 && yin1, xin1 && yin1};
                const bool inw[4] = {img[0] && use_win && wx0 && wy0, img[1] && use_win && wx1 && wy0,
                                     img[2] && use_win && wx0 && wy1, img[3] && use_win && wx1 && wy1};
                const int off[4] = {(ly * w.w + lx) * C, (ly * w.w + lx + 1) * C, ((ly + 1) * w.w + lx) * C, ((ly + 1) * w.w + lx + 1) * C};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (inw[t]) {
                        VecN<CPL> v = ldn<CPL>(win + off[t]);
#pragma unroll
                        for (int k = 0; k < CPL; ++k) v.v[k] += acc[t].v[k];
                        stn<CPL>(win + off[t], v);
                    } else if (img[t]) {
                        // rare: footprint larger than the window allowance, or rounding at the hull of the projected box
                        float* p = gp + ((size_t)(cy + (t >> 1)) * W + cx + (t & 1)) * C;
#pragma unroll
                        for (int k = 0; k < CPL; ++k) MVS_GLOBAL_ATOMIC_ADD(p + k, acc[t].v[k]);
                    }
                }
            }
        }
        m &= ~((LPP == 64 ? ~0ull : ((1ull << (LPP & 63)) - 1ull)) << (grp * LPP));
        MVS_WAVE_SYNC();   // the next group may touch the same texels: keep the DS operations in program order
    }
}

// MODE: 0 variance (MVSNet), 1 variance with the jdacs-ms alias quirk (S starts from r^2), 2 plain homo_warping
// WF: floats of LDS window space per wave (all views together); WPS: waves per SIMD the register allocation is held to
template <int C, int NS_T, int CPL, int MODE, int WF>
__global__ __launch_bounds__(256) void plane_sweep_variance_bwd_tb_kernel(SweepArgs a) {
    constexpr bool WARP_ONLY = MODE == 2, MS_ALIAS = MODE == 1;
    using Cfg = TbCfg<C, CPL>;
    constexpr int LPP = Cfg::LPP, PW = Cfg::PW, NG = Cfg::NG, BW = Cfg::BW, BH = Cfg::BH, PB = Cfg::PB;
    constexpr int ROWS = PW * NS_T;                                  // table rows: (pixel of the group, view)
    constexpr int TB = ROWS <= 4 ? 32 : (ROWS <= 8 ? 16 : 8);        // planes per table batch (64 lanes fill 64 / TB rows per pass)
    constexpr int RSTRIDE = TB * 4 + 4;                              // floats per weight row (+4: the rows of two pixels on different banks)
    constexpr int VIEW_FLOATS = WF / NS_T / C * C, WCAP = VIEW_FLOATS / C;
    __shared__ __attribute__((aligned(16))) float lds[4 * NS_T * VIEW_FLOATS];      // [wave][view][texel][C]
    __shared__ __attribute__((aligned(16))) float s_tw[4][ROWS * RSTRIDE];          // [wave][row][plane][w00 w01 w10 w11]
    __shared__ int s_txy[4][ROWS * TB];                                             // [wave][row][plane] packed base texel
    __shared__ __attribute__((aligned(16))) float s_proj[4][PB * NS_T][8];          // [wave][pixel of the block, view][rx ry rz tx ty tz - -]
    __shared__ int s_win[4][NS_T][5];                                               // per wave and view: x0, y0, w, h, usable
    __shared__ int s_fit[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = lane % LPP, pl = lane / LPP;
    const int bx0 = (blockIdx.x % a.tiles_x) * (2 * BW) + (wv & 1) * BW, by0 = (blockIdx.x / a.tiles_x) * (2 * BH) + (wv >> 1) * BH;
    const int b = blockIdx.z;
    const int HW = a.H * a.W;
    const int cq = CPL * q;
    const float inv_n = 1.0f / (float)(NS_T + 1);
    const float* __restrict__ rotb = a.rot + (size_t)b * NS_T * 9;
    const float* __restrict__ trb = a.trans + (size_t)b * NS_T * 3;
    // per (pixel of the block, view): the homography rows applied to (x, y, 1) and the translation -- once per kernel
    if (lane < PB * NS_T) {
        const int p = lane / NS_T, s = lane % NS_T;
        const float xf = (float)min(bx0 + (p % BW), a.W - 1), yf = (float)min(by0 + (p / BW), a.H - 1);
        const float* R = rotb + s * 9;
        float* o = s_proj[wv][lane];
        o[0] = fmaf(R[0], xf, fmaf(R[1], yf, R[2]));
        o[1] = fmaf(R[3], xf, fmaf(R[4], yf, R[5]));
        o[2] = fmaf(R[6], xf, fmaf(R[7], yf, R[8]));
        o[3] = trb[s * 3]; o[4] = trb[s * 3 + 1]; o[5] = trb[s * 3 + 2];
    }
    // corners of the wave's pixel block (clipped to the image) for its footprint bound
    const float cxa = (float)min(bx0, a.W - 1), cxb = (float)min(bx0 + BW - 1, a.W - 1);
    const float cya = (float)min(by0, a.H - 1), cyb = (float)min(by0 + BH - 1, a.H - 1);
    float* const wwin = lds + (size_t)wv * NS_T * VIEW_FLOATS;     // this wave's windows

    int ds = blockIdx.y * a.dslab;
    const int dend = min(a.D, ds + a.dslab);
    while (ds < dend) {
        // ---- segment [ds, de): the longest one for which every wave's windows fit (workgroup-uniform) ----
        int de = dend;
        Win w[NS_T];
        bool use[NS_T];
        for (int it = 0; it < 16; ++it) {
            float da, db;
            if (a.per_pixel) {
                float lo = 3.0e38f, hi = -3.0e38f;
                if (lane < PB) {                       // one lane per pixel of the block
                    const int px = min(bx0 + (lane % BW), a.W - 1), py = min(by0 + (lane / BW), a.H - 1);
                    for (int d = ds; d < de; ++d) {
                        const float v = a.depth[((size_t)b * a.D + d) * HW + (size_t)py * a.W + px];
                        lo = fminf(lo, v); hi = fmaxf(hi, v);
                    }
                }
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { lo = fminf(lo, __shfl_xor(lo, m)); hi = fmaxf(hi, __shfl_xor(hi, m)); }
                da = lo; db = hi;                      // depth range of THIS wave's pixels: its windows only have to hold them
            } else {
                da = a.depth[b * a.D + ds];
                db = a.depth[b * a.D + de - 1];
            }
            bool fits = true;
#pragma unroll
            for (int s = 0; s < NS_T; ++s) {
                float lox, hix, loy, hiy;
                corner_bounds(a, rotb + s * 9, trb + s * 3, cxa, cxb, cya, cyb, da, db, lox, hix, loy, hiy);
                w[s] = make_window(a, lox, hix, loy, hiy);
                // wave-uniform, but computed on the vector ALU: move to scalar registers (they live through the plane loop)
                w[s].x0 = MVS_UNIFORM_I(w[s].x0); w[s].y0 = MVS_UNIFORM_I(w[s].y0);
                w[s].w = MVS_UNIFORM_I(w[s].w); w[s].h = MVS_UNIFORM_I(w[s].h);
                use[s] = (long)w[s].w * w[s].h <= WCAP;
                fits = fits && use[s];
                use[s] = use[s] && !a.no_window;
            }
            __syncthreads();                         // previous readers of s_fit are done
            if (lane == 0) s_fit[wv] = fits ? 1 : 0;
            __syncthreads();
            const bool all_fit = s_fit[0] && s_fit[1] && s_fit[2] && s_fit[3];
            if (all_fit || de - ds <= 1) break;
            de = ds + (de - ds + 1) / 2;
        }
        // ---- zero this wave's windows; publish their geometry for the write-out ----
#pragma unroll
        for (int s = 0; s < NS_T; ++s) {
            if (use[s]) {
                float* ws = wwin + s * VIEW_FLOATS;
                for (int i = lane * 4; i < w[s].w * w[s].h * C; i += 256) *reinterpret_cast<float4*>(ws + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (lane == 0) {
                s_win[wv][s][0] = w[s].x0; s_win[wv][s][1] = w[s].y0; s_win[wv][s][2] = w[s].w; s_win[wv][s][3] = w[s].h;
                s_win[wv][s][4] = use[s] ? 1 : 0;
            }
        }
        MVS_WAVE_SYNC();
        // ---- the wave's pixel groups, one after the other, over the planes of the segment (no workgroup barrier in here) ----
#pragma clang loop unroll(disable)
        for (int grp = 0; grp < NG; ++grp) {
            const int p = grp * PW + pl;             // this lane's pixel of the block
            const int xr = bx0 + (p % BW), yr = by0 + (p / BW);
            const bool live = xr < a.W && yr < a.H;  // lanes outside the image follow along (wave-wide votes) on a clamped pixel
            const int pix = min(yr, a.H - 1) * a.W + min(xr, a.W - 1);
            const unsigned voff = (unsigned)pix * C + cq;                    // this lane's channels inside one [H,W,C] plane
            const size_t fbase = (size_t)b * HW * C + cq;
            const VecN<CPL> r = ldn<CPL>(a.ref + (size_t)b * HW * C + voff);
            const float two_n = live ? 2.0f * inv_n : 0.0f;                  // dead lanes contribute exact zeros
            VecN<CPL> gr = zeron<CPL>();
            VecN<CPL> tap[NS_T][4], acc[NS_T][4];
            int cur[NS_T];
#pragma unroll
            for (int s = 0; s < NS_T; ++s) {
                cur[s] = TB_NOBLOCK;
#pragma unroll
                for (int t = 0; t < 4; ++t) { tap[s][t] = zeron<CPL>(); acc[s][t] = zeron<CPL>(); }
            }
            const float* __restrict__ twp = &s_tw[wv][(pl * NS_T) * RSTRIDE];   // this lane's pixel, view 0 (view s: + s * RSTRIDE)
            const int* __restrict__ txp = &s_txy[wv][(pl * NS_T) * TB];

            for (int b0 = ds; b0 < de; b0 += TB) {
                const int n = min(TB, de - b0);
                // ---- the table of this batch: lanes spread over (row, plane) ----
                MVS_WAVE_SYNC();                     // the previous batch's reads are done (a wave's DS queue is in order)
#pragma unroll
                for (int e0 = 0; e0 < ROWS * TB; e0 += 64) {
                    const int e = e0 + lane;
                    const int row = e / TB, i = e % TB;
                    if (e < ROWS * TB && i < n) {
                        const int pp = row / NS_T, s = row % NS_T;
                        const int pb = grp * PW + pp;
                        const float* pr = s_proj[wv][pb * NS_T + s];
                        float dep;
                        if (a.per_pixel) {
                            const int px = min(bx0 + (pb % BW), a.W - 1), py = min(by0 + (pb / BW), a.H - 1);
                            dep = a.depth[((size_t)b * a.D + b0 + i) * HW + (size_t)py * a.W + px];
                        } else {
                            dep = a.depth[b * a.D + b0 + i];
                        }
                        // the forward kernel's arithmetic (v_rcp_f32 + one Newton step)
                        const float zz = fmaf(pr[2], dep, pr[5]);
                        float iz = MVS_RCP(zz);
                        iz = fmaf(fmaf(-zz, iz, 1.0f), iz, iz);
                        const float ix = fmaf(fmaf(pr[0], dep, pr[3]) * iz, a.sx, a.ox);
                        const float iy = fmaf(fmaf(pr[1], dep, pr[4]) * iz, a.sy, a.oy);
                        const float fx = floorf(ix), fy = floorf(iy);
                        const float wx = ix - fx, wy = iy - fy;
                        const float ex = 1.0f - wx, ey = 1.0f - wy;
                        float4 wt;
                        wt.x = ey * ex; wt.y = ey * wx; wt.z = wy * ex; wt.w = wy * wx;
                        *reinterpret_cast<float4*>(&s_tw[wv][row * RSTRIDE + 4 * i]) = wt;
                        s_txy[wv][row * TB + i] = tb_pack(MVS_F2I(fx), MVS_F2I(fy), a.H, a.W);
                    }
                }
                MVS_WAVE_SYNC();
                // ---- the planes of the batch ----
                const float* __restrict__ gbase = a.gvar + ((size_t)b * a.D + b0) * HW * C;   // wave-uniform; + plane * HW*C + voff
                const size_t gstep = (size_t)HW * C;
                constexpr int PD = 4;                // upstream gradient requested PD planes ahead (PD registers per channel)
                VecN<CPL> gq[PD];
#pragma unroll
                for (int j = 0; j < PD; ++j) gq[j] = ldn<CPL>(gbase + (size_t)min(j, n - 1) * gstep + voff);

                // one plane: i = plane of the batch, g = its upstream gradient
                auto plane = [&](const int i, const VecN<CPL>& g) __attribute__((always_inline)) {
                    float4 wt[NS_T];
                    int xy[NS_T];
                    bool any = false;
#pragma unroll
                    for (int s = 0; s < NS_T; ++s) {
                        wt[s] = *reinterpret_cast<const float4*>(twp + s * RSTRIDE + 4 * i);
                        xy[s] = txp[s * TB + i];
                        any = any || xy[s] != cur[s];
                    }
                    if (MVS_ANY(any)) {
                        // ---- slow path: some pixel of the wave leaves its 2x2 block on this plane ----
#pragma unroll
                        for (int s = 0; s < NS_T; ++s) {
                            const bool chg = xy[s] != cur[s];
                            if (MVS_ANY(chg)) {
                                // request the new block first (the flush needs the accumulators and the OLD base texel, not the tap
                                // values): the L2 round trip of the gather overlaps the LDS round trips of the flush
                                if (chg) {
                                    const int x0 = tb_x(xy[s]), y0 = tb_y(xy[s]);
                                    const float* __restrict__ f = a.src[s] + fbase + ((long)y0 * a.W + x0) * C;
                                    if (x0 >= 0 && x0 + 1 < a.W && y0 >= 0 && y0 + 1 < a.H) {   // common case: all four taps inside
                                        tap[s][0] = ldn<CPL>(f); tap[s][1] = ldn<CPL>(f + C);
                                        tap[s][2] = ldn<CPL>(f + a.W * C); tap[s][3] = ldn<CPL>(f + a.W * C + C);
                                    } else {
                                        const bool xin0 = x0 >= 0 && x0 < a.W, xin1 = x0 + 1 >= 0 && x0 + 1 < a.W;
                                        const bool yin0 = y0 >= 0 && y0 < a.H, yin1 = y0 + 1 >= 0 && y0 + 1 < a.H;
                                        tap[s][0] = (xin0 && yin0) ? ldn<CPL>(f) : zeron<CPL>();
                                        tap[s][1] = (xin1 && yin0) ? ldn<CPL>(f + C) : zeron<CPL>();
                                        tap[s][2] = (xin0 && yin1) ? ldn<CPL>(f + a.W * C) : zeron<CPL>();
                                        tap[s][3] = (xin1 && yin1) ? ldn<CPL>(f + a.W * C + C) : zeron<CPL>();
                                    }
                                }
                                tb_flush_groups<C, CPL, LPP>(chg && live && cur[s] != TB_NOBLOCK, lane, cur[s], acc[s], a.H, a.W,
                                                             wwin + s * VIEW_FLOATS + cq, w[s], use[s], a.gsrc[s] + fbase);
                                if (chg) {
                                    cur[s] = xy[s];
#pragma unroll
                                    for (int t = 0; t < 4; ++t) acc[s][t] = zeron<CPL>();
                                }
                                // the gathered taps are waited for HERE, on the planes that re-gather
#pragma unroll
                                for (int t = 0; t < 4; ++t) MVS_PINN(tap[s][t]);
                            }
                        }
                    }
                    // ---- channel arithmetic: bilinear samples of all views, their mean, the gradients of the samples ----
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                        float S = WARP_ONLY ? 0.f : (MS_ALIAS ? r.v[k] * r.v[k] : r.v[k]);
                        float v[NS_T];
                        if (!WARP_ONLY) {
#pragma unroll
                            for (int s = 0; s < NS_T; ++s) {
                                v[s] = fmaf(tap[s][3].v[k], wt[s].w, fmaf(tap[s][2].v[k], wt[s].z, fmaf(tap[s][1].v[k], wt[s].y, tap[s][0].v[k] * wt[s].x)));
                                S += v[s];
                            }
                        }
                        float gs = g.v[k] * two_n;                       // g * 2/N (0 on dead lanes)
                        const float Sm = S * inv_n;
                        if (WARP_ONLY) gs = live ? g.v[k] : 0.f;         // plain homo_warping: the warped sample itself gets the gradient
                        else if (MS_ALIAS) gr.v[k] += gs * r.v[k] * (1.0f - 2.0f * Sm);
                        else gr.v[k] += gs * (r.v[k] - Sm);
#pragma unroll
                        for (int s = 0; s < NS_T; ++s) {
                            const float gv = WARP_ONLY ? gs : gs * (v[s] - Sm);
                            acc[s][0].v[k] = fmaf(gv, wt[s].x, acc[s][0].v[k]);
                            acc[s][1].v[k] = fmaf(gv, wt[s].y, acc[s][1].v[k]);
                            acc[s][2].v[k] = fmaf(gv, wt[s].z, acc[s][2].v[k]);
                            acc[s][3].v[k] = fmaf(gv, wt[s].w, acc[s][3].v[k]);
                        }
                    }
                };
#pragma clang loop unroll(disable)
                for (int i = 0; i < n; i += PD) {
#pragma unroll
                    for (int j = 0; j < PD; ++j) {
                        if (i + j < n) {             // wave-uniform
                            plane(i + j, gq[j]);
                            if (i + j + PD < n) gq[j] = ldn<CPL>(gbase + (size_t)(i + j + PD) * gstep + voff);
                        }
                    }
                }
            }
            // the blocks still held in registers, then this group's share of grad_ref (a pixel's lanes cover whole texels)
#pragma unroll
            for (int s = 0; s < NS_T; ++s)
                tb_flush_groups<C, CPL, LPP>(live && cur[s] != TB_NOBLOCK, lane, cur[s], acc[s], a.H, a.W, wwin + s * VIEW_FLOATS + cq,
                                             w[s], use[s], a.gsrc[s] + fbase);
            if (!WARP_ONLY && live) {
#pragma unroll
                for (int k = 0; k < CPL; ++k) MVS_GLOBAL_ATOMIC_ADD(a.gref + (size_t)b * HW * C + voff + k, gr.v[k]);
            }
        }
        // ---- write the segment out: the four waves' windows summed on the fly, coalesced global atomics ----
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NS_T; ++s) {
            int ux0 = 1 << 30, uy0 = 1 << 30, ux1 = -1, uy1 = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (s_win[j][s][4]) {
                    ux0 = min(ux0, s_win[j][s][0]); uy0 = min(uy0, s_win[j][s][1]);
                    ux1 = max(ux1, s_win[j][s][0] + s_win[j][s][2]); uy1 = max(uy1, s_win[j][s][1] + s_win[j][s][3]);
                }
            const int uw = ux1 - ux0, uh = uy1 - uy0;
            if (uw <= 0 || uh <= 0) continue;
            int jx0[4], jy0[4], jw[4], jh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                jx0[j] = s_win[j][s][0]; jy0[j] = s_win[j][s][1];
                jw[j] = s_win[j][s][4] ? s_win[j][s][2] : 0; jh[j] = s_win[j][s][3];
            }
            float* gp = a.gsrc[s] + (size_t)b * HW * C;
            for (int i = tid; i < uw * uh * C; i += 256) {
                const int c = i % C, t = i / C;
                const int txl = ux0 + t % uw, tyl = uy0 + t / uw;
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int lx = txl - jx0[j], ly = tyl - jy0[j];
                    if (lx >= 0 && lx < jw[j] && ly >= 0 && ly < jh[j])
                        sum += lds[(j * NS_T + s) * VIEW_FLOATS + (ly * jw[j] + lx) * C + c];
                }
                if (sum != 0.f) MVS_GLOBAL_ATOMIC_ADD(gp + ((size_t)tyl * a.W + txl) * C + c, sum);
            }
        }
        __syncthreads();
        ds = de;
    }
}

// ---- launcher ------------------------------------------------------------------------------------------------------------
int g_sweep_bwd_cpl = 1;     // knob "bwd_cpl": channels per lane of the table form (1, 2, 4; clipped to what the channel count allows)
int g_sweep_bwd_wf = 2048;   // knob "bwd_wf": floats of LDS window space per wave (1536, 2048 or 3200)
extern int g_sweep_bwd_dslab;   // plane_sweep.hip: knobs "bwd_dslab", "bwd_nowin"
extern int g_sweep_bwd_nowin;

template <int C, int NS_T, int CPL, int WF>
static int launch_tb_mode(SweepArgs& a, dim3 grid, hipStream_t st) {
    if (a.warp_only) {
        if constexpr (NS_T == 1) MVS_LAUNCH((plane_sweep_variance_bwd_tb_kernel<C, 1, CPL, 2, WF>), grid, dim3(256), 0, st, a);
    } else if (a.ms_alias) MVS_LAUNCH((plane_sweep_variance_bwd_tb_kernel<C, NS_T, CPL, 1, WF>), grid, dim3(256), 0, st, a);
    else MVS_LAUNCH((plane_sweep_variance_bwd_tb_kernel<C, NS_T, CPL, 0, WF>), grid, dim3(256), 0, st, a);
    return mvs_check_launch("plane_sweep_variance_bwd_tb");
}

template <int C, int NS_T, int CPL>
static int launch_tb_wf(SweepArgs& a, dim3 grid, hipStream_t st) {
    // the A/B window sizes are instantiated for the benchmarked channel count only (C = 32); 3200 floats elsewhere (4 views need them)
    if constexpr (C == 32) {
        if (g_sweep_bwd_wf <= 1536 && NS_T <= 2) return launch_tb_mode<C, NS_T, CPL, 1536>(a, grid, st);
        if (g_sweep_bwd_wf <= 2048 && NS_T <= 2) return launch_tb_mode<C, NS_T, CPL, 2048>(a, grid, st);
    }
    return launch_tb_mode<C, NS_T, CPL, 3200>(a, grid, st);
}

template <int C, int NS_T>
static int launch_tb_cpl(SweepArgs& a, dim3 grid, hipStream_t st) {
    if constexpr (C == 32) {
        if (g_sweep_bwd_cpl >= 4) return launch_tb_wf<C, NS_T, 4>(a, grid, st);
        if (g_sweep_bwd_cpl == 2) return launch_tb_wf<C, NS_T, 2>(a, grid, st);
    } else if constexpr (C == 16) {
        if (g_sweep_bwd_cpl >= 2) return launch_tb_wf<C, NS_T, 2>(a, grid, st);
    }
    return launch_tb_wf<C, NS_T, 1>(a, grid, st);
}

template <int C>
static int launch_tb_c(SweepArgs& a, hipStream_t st) {
    a.tiles_x = mvs_cdiv(a.W, 8);
    a.tiles_y = mvs_cdiv(a.H, 4);
    // depth slabs: >= ~2048 workgroups, each >= 16 planes: every extra slab re-gathers the blocks and writes its windows out once more
    const int tiles = a.tiles_x * a.tiles_y * a.B;
    int nslab = mvs_cdiv(2048, tiles);
    if (nslab > a.D / 16) nslab = a.D / 16;
    if (nslab < 1) nslab = 1;
    a.dslab = g_sweep_bwd_dslab > 0 ? g_sweep_bwd_dslab : mvs_cdiv(a.D, nslab);
    a.no_window = g_sweep_bwd_nowin;
    dim3 grid(a.tiles_x * a.tiles_y, mvs_cdiv(a.D, a.dslab), a.B);
    switch (a.NS) {
        case 1: return launch_tb_cpl<C, 1>(a, grid, st);
        case 2: return launch_tb_cpl<C, 2>(a, grid, st);
        case 3: return launch_tb_cpl<C, 3>(a, grid, st);
        case 4: return launch_tb_cpl<C, 4>(a, grid, st);
        default: break;
    }
    mvs_set_error("plane_sweep backward (table form): 1..4 source views, got %d", a.NS);
    return MVS_ERR_UNSUPPORTED;
}

// 1..4 source views, image sides < 32000 (the packed base texel); the caller falls back to the round-1 kernel otherwise
bool sweep_bwd_tb_supports(const SweepArgs& a) { return a.NS >= 1 && a.NS <= 4 && a.W < 32000 && a.H < 32000; }

int launch_sweep_bwd_tb(SweepArgs& a, int C, hipStream_t st) {
    if (C == 32) return launch_tb_c<32>(a, st);
    if (C == 16) return launch_tb_c<16>(a, st);
    return launch_tb_c<8>(a, st);
}
