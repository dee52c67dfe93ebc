// K2, projection-table form with an LDS-DMA ring (round 3).  EXPERIMENTAL: behind mvs_set_tuning("sweep_bwd", 2); the default
// stays the round-2 per-wave-window kernel (plane_sweep.hip), which measured faster (0.32 ms vs 0.70 ms at BASELINE config 2:
// profiles/r03_run4_k2_table_form_v3_lds_dma_ab.log, counters profiles/r03_run5_pmc_sq_k2_table_form_v3.json).  Parity-green on
// the GPU and in the emulation; kept as the test bed of two ideas and of what they cost.
//
// Idea 1 -- take the per-PIXEL work out of the plane loop.  The round-2 kernel spends ~195 vector instructions per (wave of 8
// pixels, plane), of which ~104 are channel arithmetic; the rest is the projection replicated over the 8 channel lanes of a pixel
// (homography, reciprocal, floor, bilinear weights), change detection and the flush.  Here a wave = PW = 64 / C pixels side by side x
// C channel lanes (ONE channel per lane; the 8 (4) pixels of its block as groups one after the other, all into the same per-wave LDS
// windows), and before every batch of 16 planes the wave computes a TABLE in LDS -- for every (pixel of the group, view, plane) the
// four bilinear weights and the packed base texel, the 64 lanes spread over (row, plane) -- plus a uniform bit mask of the planes on
// which SOME lane enters a new 2x2 block.  Planes without a change take a fast step: three LDS reads (gradient, two weight quads),
// ~24 vector instructions for two views.
// Idea 2 -- nothing in the plane loop waits on a register load.  The upstream gradient arrives by LDS-DMA (global_load_lds_dword,
// inline assembly: mvs_rt.h says why) into a ring of R planes, exactly one request per step, so `s_waitcnt vmcnt(R-1)` at the top
// of step i means plane i has landed; a lane that will enter a new block on plane i + R requests it on plane i, 4 taps by LDS-DMA
// into its staging slots, BEFORE that step's ring request -- whose arrival step i + R waits for anyway (loads complete in order).
//
// What the measurement says (counters of the C = 32, 2-view launch: 184 M vector + 142 M scalar instructions, i.e. 93 + 72 per
// (wave of TWO pixels, plane) -- twice the round-2 kernel's count per pixel; waves parked at s_waitcnt 56 % of their cycles):
//  * at DTU-like geometry the sample point moves 0.03-0.055 texel per plane in x and 0.012-0.02 in y: a (pixel, view) enters a new
//    block every 13-22 planes, a wave of 2 pixels x 2 views has such an event every ~4 planes, and every event makes TWO planes slow
//    (the request R planes ahead and the take-over + flush): half of all planes run the ~100-instruction slow step, not the fast one;
//  * every walk (group x depth segment, 22-48 planes with 24 texels of window per view) starts with two exposed memory round trips:
//    the ring's first requests, and a register gather of the first block (nothing could be staged for plane 0);
//  * hipcc keeps ~90 loop-invariant addresses in registers across the plane loop (161-205 VGPRs for what needs ~60): 3 waves per SIMD
//    at best, with the LDS (52 KB per workgroup) as the second limit.
// Channel arithmetic alone is 13 wave-instructions per (pixel, plane) in either layout; the round-2 kernel spends 24, and a version
// of this one with the slow step slimmed to ~40 instructions and the start-up bubbles removed would land at ~20-26: not the factor
// the 40 % target needs (DESIGN.md section 4 has the arithmetic).  The flush of register accumulators into LDS windows on every
// block change is what both designs pay for; it is the thing to replace, not the bookkeeping around it.
//
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
#if defined(MVS_CPU_EMUL)
#define MVS_PIN1(x) ((void)0)
#else
#define MVS_PIN1(x) asm volatile("" : "+v"(x))
#endif

// A wave = PW pixels side by side x C channel lanes (ONE channel per lane); the PB pixels of its block (BW x 2) as NG groups.
// 3-4 source views take a 2x2 block: the windows of four views must share the same LDS budget.
template <int C, int NS_T> struct TbCfg {
    static constexpr int PW = 64 / C;                                   // pixels a wave walks side by side
    static constexpr int PB0 = NS_T <= 2 ? 8 : 4;
    static constexpr int PB = PB0 < PW ? PW : PB0;                      // pixels of the wave's block ...
    static constexpr int BH = 2, BW = PB / BH;
    static constexpr int NG = PB / PW;                                  // ... walked as NG groups one after the other
    static_assert(C == 8 || C == 16 || C == 32, "one channel per lane: 8, 16 or 32 channels");
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
// WF: floats of LDS window space per wave (all views together)
// R:  depth of the LDS-DMA ring = look-ahead distance in planes.
//
// Memory pipeline of a wave (everything below is per wave; no workgroup barrier inside a depth segment):
//  * the upstream gradient arrives by LDS-DMA (global_load_lds_dword: 64 lanes x 4 bytes = the wave's PW pixels x C channels of one
//    plane, no register destination) into a ring of R planes; step i issues the request of plane i + R, ALWAYS exactly one per step
//    (past the end of the segment a valid plane is re-requested), so that at the top of step i "at most R - 1 vector-memory
//    operations outstanding" (s_waitcnt vmcnt(R-1)) means plane i has landed: loads complete in order;
//  * the table (computed once per batch) also says on which planes SOME lane of the wave enters a new 2x2 block (a uniform bit
//    mask): on plane i, R planes before such a plane, the lanes concerned request the new block -- 4 taps, LDS-DMA into the lane's
//    staging slots -- BEFORE the ring request of step i; that ring request is the one whose arrival step i + R waits for, so the
//    staged block has arrived too: taking it over costs no wait of its own, and nothing in the loop waits on a register load.
//    (A gather issued on the plane that needs it must be waited for together with everything requested before it: the first cut
//    of this kernel drained its prefetch ring on every block change -- profiles/r03_run1_k2_table_form_v1_ab.log.)
//  * planes without a block change and without a look-ahead request take the FAST step: three LDS reads (gradient, two weight
//    quads), ~24 vector instructions for two views, one DMA request.
template <int C, int NS_T, int MODE, int WF, int R>
__global__ __launch_bounds__(256) MVS_MIN_WAVES_PER_SIMD((NS_T <= 2 ? 3 : 2)) void plane_sweep_variance_bwd_tb_kernel(SweepArgs a) {
    constexpr bool WARP_ONLY = MODE == 2, MS_ALIAS = MODE == 1;
    using Cfg = TbCfg<C, NS_T>;
    constexpr int PW = Cfg::PW, NG = Cfg::NG, BW = Cfg::BW, BH = Cfg::BH, PB = Cfg::PB;
    constexpr int ROWS = PW * NS_T;                                  // table rows: (pixel of the group, view)
    constexpr int TB = 16;                                           // planes per table batch
    constexpr int TS = TB + R;                                       // entries per row: the batch + the look-ahead into the next one
    constexpr int TSX = TS + 1;                                      // packed texels: one more in front (the plane before the batch)
    constexpr int RSTRIDE = TS * 4 + 4;                              // floats per weight row (+4: the rows of two pixels on different banks)
    constexpr int VIEW_FLOATS = WF / NS_T / C * C, WCAP = VIEW_FLOATS / C;
    constexpr int DEPMAX = 256;                                      // planes per workgroup (the launcher keeps the slab below)
    static_assert(TS < 64 && R >= 2 && R <= 32, "the per-batch change mask is one 64-bit word");
    __shared__ __attribute__((aligned(16))) float lds[4 * NS_T * VIEW_FLOATS];      // [wave][view][texel][C]
    __shared__ __attribute__((aligned(16))) float s_tw[4][ROWS * RSTRIDE];          // [wave][row][plane][w00 w01 w10 w11]
    __shared__ int s_txy[4][ROWS * TSX];                                            // [wave][row][1 + plane] packed base texel
    __shared__ __attribute__((aligned(16))) float s_proj[4][PB * NS_T][8];          // [wave][pixel of the block, view][rx ry rz tx ty tz - -]
    __shared__ float s_ring[4][R][64];                                              // [wave][slot][lane] upstream gradient (LDS-DMA)
    __shared__ float s_stage[4][NS_T][4][64];                                       // [wave][view][tap][lane] staged blocks (LDS-DMA)
    __shared__ float s_dep[DEPMAX];                                                 // the slab's per-plane depth hypotheses
    __shared__ int s_win[4][NS_T][5];                                               // per wave and view: x0, y0, w, h, usable
    __shared__ int s_fit[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cq = lane % C, pl = lane / C;                          // channel, pixel of the group
    const int bx0 = (blockIdx.x % a.tiles_x) * (2 * BW) + (wv & 1) * BW, by0 = (blockIdx.x / a.tiles_x) * (2 * BH) + (wv >> 1) * BH;
    const int b = blockIdx.z;
    const int HW = a.H * a.W;
    const float inv_n = 1.0f / (float)(NS_T + 1);
    const float* __restrict__ rotb = a.rot + (size_t)b * NS_T * 9;
    const float* __restrict__ trb = a.trans + (size_t)b * NS_T * 3;
    const int ds0 = blockIdx.y * a.dslab;
    const int dend = min(a.D, ds0 + a.dslab);
    // per (pixel of the block, view): the homography rows applied to (x, y, 1) and the translation -- once per kernel
    if (lane < PB * NS_T) {
        const int p = lane / NS_T, s = lane % NS_T;
        const float xf = (float)min(bx0 + (p % BW), a.W - 1), yf = (float)min(by0 + (p / BW), a.H - 1);
        const float* Rm = rotb + s * 9;
        float* o = s_proj[wv][lane];
        o[0] = fmaf(Rm[0], xf, fmaf(Rm[1], yf, Rm[2]));
        o[1] = fmaf(Rm[3], xf, fmaf(Rm[4], yf, Rm[5]));
        o[2] = fmaf(Rm[6], xf, fmaf(Rm[7], yf, Rm[8]));
        o[3] = trb[s * 3]; o[4] = trb[s * 3 + 1]; o[5] = trb[s * 3 + 2];
    }
    if (!a.per_pixel) {
        for (int i = tid; i < dend - ds0; i += 256) s_dep[i] = a.depth[b * a.D + ds0 + i];
    }
    __syncthreads();
    // corners of the wave's pixel block (clipped to the image) for its footprint bound
    const float cxa = (float)min(bx0, a.W - 1), cxb = (float)min(bx0 + BW - 1, a.W - 1);
    const float cya = (float)min(by0, a.H - 1), cyb = (float)min(by0 + BH - 1, a.H - 1);
    float* const wwin = lds + (size_t)wv * NS_T * VIEW_FLOATS;     // this wave's windows

    int ds = ds0;
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
                da = s_dep[ds - ds0];
                db = s_dep[de - 1 - ds0];
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
        const int L = de - ds;                       // planes of the segment
        // ---- the wave's pixel groups, one after the other, over the planes of the segment (no workgroup barrier in here) ----
#pragma clang loop unroll(disable)
        for (int grp = 0; grp < NG; ++grp) {
            const int p = grp * PW + pl;             // this lane's pixel of the block
            const int xr = bx0 + (p % BW), yr = by0 + (p / BW);
            const bool live = xr < a.W && yr < a.H;  // lanes outside the image follow along (wave-wide votes) on a clamped pixel
            const int pix = min(yr, a.H - 1) * a.W + min(xr, a.W - 1);
            const unsigned voff = (unsigned)pix * C + cq;                    // this lane's channel inside one [H,W,C] plane
            const size_t fbase = (size_t)b * HW * C + cq;
            const float r = a.ref[(size_t)b * HW * C + voff];
            const float two_n = live ? 2.0f * inv_n : 0.0f;                  // dead lanes contribute exact zeros
            float gr = 0.f;
            VecN<1> tap[NS_T][4], acc[NS_T][4];
            int cur[NS_T], stage_xy[NS_T];
#pragma unroll
            for (int s = 0; s < NS_T; ++s) {
                cur[s] = TB_NOBLOCK; stage_xy[s] = TB_NOBLOCK;
#pragma unroll
                for (int t = 0; t < 4; ++t) { tap[s][t].v[0] = 0.f; acc[s][t].v[0] = 0.f; }
            }
            const float* __restrict__ twp = &s_tw[wv][(pl * NS_T) * RSTRIDE];   // this lane's pixel, view 0 (view s: + s * RSTRIDE)
            const int* __restrict__ txp = &s_txy[wv][(pl * NS_T) * TSX + 1];     // [-1] = the plane before the batch
            const float* __restrict__ gseg = a.gvar + ((size_t)b * a.D + ds) * HW * C;    // wave-uniform; plane i of the segment: + i * HW*C
            const size_t gstep = (size_t)HW * C;
            float* const ring = &s_ring[wv][0][0];
            // the ring's first R planes (re-requesting the last plane when the segment is shorter keeps the count invariant)
#pragma unroll
            for (int j = 0; j < R; ++j) MVS_DMA4(ring + j * 64, gseg + (size_t)min(j, L - 1) * gstep, voff * 4u);

            int i = 0;                               // plane of the segment being processed
            int tb0 = 0, next_fill = 0;              // first plane of the table batch in LDS, first plane of the next one
            unsigned long long chgmask = 0ull;       // bit ii: some lane of the wave enters a new block on plane tb0 + ii

            // channel arithmetic of one plane: bilinear samples of all views, their mean, the gradients of the samples
            auto arith = [&](const float g, const float4 (&wt)[NS_T]) __attribute__((always_inline)) {
                float S = WARP_ONLY ? 0.f : (MS_ALIAS ? r * r : r);
                float v[NS_T];
                if (!WARP_ONLY) {
#pragma unroll
                    for (int s = 0; s < NS_T; ++s) {
                        v[s] = fmaf(tap[s][3].v[0], wt[s].w, fmaf(tap[s][2].v[0], wt[s].z, fmaf(tap[s][1].v[0], wt[s].y, tap[s][0].v[0] * wt[s].x)));
                        S += v[s];
                    }
                }
                float gs = g * two_n;                            // g * 2/N (0 on dead lanes)
                const float Sm = S * inv_n;
                if (WARP_ONLY) gs = live ? g : 0.f;              // plain homo_warping: the warped sample itself gets the gradient
                else if (MS_ALIAS) gr += gs * r * (1.0f - 2.0f * Sm);
                else gr += gs * (r - Sm);
#pragma unroll
                for (int s = 0; s < NS_T; ++s) {
                    const float gv = WARP_ONLY ? gs : gs * (v[s] - Sm);
                    acc[s][0].v[0] = fmaf(gv, wt[s].x, acc[s][0].v[0]);
                    acc[s][1].v[0] = fmaf(gv, wt[s].y, acc[s][1].v[0]);
                    acc[s][2].v[0] = fmaf(gv, wt[s].z, acc[s][2].v[0]);
                    acc[s][3].v[0] = fmaf(gv, wt[s].w, acc[s][3].v[0]);
                }
            };

#pragma clang loop unroll(disable)
            while (i < L) {
                // ---- table refill at a batch boundary (+ R planes of look-ahead): lanes spread over (row, plane) ----
                if (i == next_fill) {
                    int keep = TB_NOBLOCK;               // the packed texel of plane i - 1 (lane = row), before the rows are overwritten
                    if (i > 0 && lane < ROWS) keep = s_txy[wv][lane * TSX + 1 + (i - 1 - tb0)];
                    MVS_WAVE_SYNC();                     // the previous batch's reads are done (a wave's DS queue is in order)
                    const int nfill = min(TS, L - i);
#pragma unroll 1
                    for (int e0 = 0; e0 < ROWS * TS; e0 += 64) {
                        const int e = e0 + lane;
                        const int row = e / TS, ii = e % TS;
                        if (e < ROWS * TS && ii < nfill) {
                            const int pp = row / NS_T, s = row % NS_T;
                            const int pb = grp * PW + pp;
                            const float* pr = s_proj[wv][pb * NS_T + s];
                            float dep;
                            if (a.per_pixel) {
                                const int px = min(bx0 + (pb % BW), a.W - 1), py = min(by0 + (pb / BW), a.H - 1);
                                dep = a.depth[((size_t)b * a.D + ds + i + ii) * HW + (size_t)py * a.W + px];
                            } else {
                                dep = s_dep[ds - ds0 + i + ii];
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
                            float4 wt4;
                            wt4.x = ey * ex; wt4.y = ey * wx; wt4.z = wy * ex; wt4.w = wy * wx;
                            *reinterpret_cast<float4*>(&s_tw[wv][row * RSTRIDE + 4 * ii]) = wt4;
                            s_txy[wv][row * TSX + 1 + ii] = tb_pack(MVS_F2I(fx), MVS_F2I(fy), a.H, a.W);
                        }
                    }
                    if (lane < ROWS) s_txy[wv][lane * TSX] = keep;
                    MVS_WAVE_SYNC();
                    // which planes of the batch see a block change in SOME lane of the wave: entry != the entry of the plane before
                    chgmask = 0ull;
#pragma unroll
                    for (int e0 = 0; e0 < ROWS * TS; e0 += 64) {
                        const int e = e0 + lane;
                        const int row = e / TS, ii = e % TS;
                        bool c = false;
                        if (e < ROWS * TS && ii < nfill) c = s_txy[wv][row * TSX + 1 + ii] != s_txy[wv][row * TSX + ii];
                        const unsigned long long m = MVS_BALLOT(c);
#pragma unroll
                        for (int rr = e0 / TS; rr <= (e0 + 63) / TS && rr < ROWS; ++rr) {
                            const int sh = rr * TS - e0;             // lane of entry (rr, 0)
                            const unsigned long long part = sh >= 0 ? (m >> (sh & 63)) : (m << ((-sh) & 63));
                            chgmask |= part & ((1ull << TS) - 1ull);
                        }
                    }
                    tb0 = i;
                    next_fill = i + TB;
                }
                const int ti = i - tb0;
                const bool c_plane = ((chgmask >> ti) & 1ull) != 0ull;                        // scalar: a block change on this plane
                const bool l_plane = i + R < L && ((chgmask >> (ti + R)) & 1ull) != 0ull;     // scalar: a block change R planes ahead
                // plane i has landed in the ring, and so has every block staged for it (see the template comment)
                MVS_WAIT_VMCNT(R - 1);
                const float g = ring[(i % R) * 64 + lane];
                float4 wt[NS_T];
#pragma unroll
                for (int s = 0; s < NS_T; ++s) wt[s] = *reinterpret_cast<const float4*>(twp + s * RSTRIDE + 4 * ti);
                if (c_plane) {
                    // ---- lanes whose sample point leaves its 2x2 block on this plane ----
#pragma unroll
                    for (int s = 0; s < NS_T; ++s) {
                        const int xy = txp[s * TSX + ti];
                        const bool chg = xy != cur[s];
                        if (MVS_ANY(chg)) {
                            if (chg) {
                                const int x0 = tb_x(xy), y0 = tb_y(xy);
                                const bool xin0 = x0 >= 0 && x0 < a.W, xin1 = x0 + 1 >= 0 && x0 + 1 < a.W;
                                const bool yin0 = y0 >= 0 && y0 < a.H, yin1 = y0 + 1 >= 0 && y0 + 1 < a.H;
                                if (stage_xy[s] == xy) {         // requested R planes ago: in the lane's staging slots by now
                                    const float* st = &s_stage[wv][s][0][lane];
                                    const float t0 = st[0], t1 = st[64], t2 = st[128], t3 = st[192];
                                    tap[s][0].v[0] = (xin0 && yin0) ? t0 : 0.f; tap[s][1].v[0] = (xin1 && yin0) ? t1 : 0.f;
                                    tap[s][2].v[0] = (xin0 && yin1) ? t2 : 0.f; tap[s][3].v[0] = (xin1 && yin1) ? t3 : 0.f;
                                    stage_xy[s] = TB_NOBLOCK;
                                } else {
                                    // first plane of a walk, or a second change inside the look-ahead distance: a register gather
                                    // (the compiler's wait for it drains the ring: rare)
                                    const float* __restrict__ f = a.src[s] + fbase + ((long)y0 * a.W + x0) * C;
                                    tap[s][0].v[0] = (xin0 && yin0) ? f[0] : 0.f;
                                    tap[s][1].v[0] = (xin1 && yin0) ? f[C] : 0.f;
                                    tap[s][2].v[0] = (xin0 && yin1) ? f[(size_t)a.W * C] : 0.f;
                                    tap[s][3].v[0] = (xin1 && yin1) ? f[(size_t)a.W * C + C] : 0.f;
                                }
                            }
                            tb_flush_groups<C, 1, C>(chg && live && cur[s] != TB_NOBLOCK, lane, cur[s], acc[s], a.H, a.W,
                                                     wwin + s * VIEW_FLOATS + cq, w[s], use[s], a.gsrc[s] + fbase);
                            if (chg) {
                                cur[s] = xy;
#pragma unroll
                                for (int t = 0; t < 4; ++t) acc[s][t].v[0] = 0.f;
                            }
#pragma unroll
                            for (int t = 0; t < 4; ++t) MVS_PIN1(tap[s][t].v[0]);
                        }
                    }
                }
                if (l_plane) {
                    // ---- look-ahead: blocks entered on plane i + R are requested now, BEFORE this step's ring request ----
#pragma unroll
                    for (int s = 0; s < NS_T; ++s) {
                        const int xl = txp[s * TSX + ti + R];
                        const bool c = xl != txp[s * TSX + ti + R - 1];
                        if (c && stage_xy[s] == TB_NOBLOCK) {
                            const int x0 = tb_x(xl), y0 = tb_y(xl);
                            const bool xin0 = x0 >= 0 && x0 < a.W, xin1 = x0 + 1 >= 0 && x0 + 1 < a.W;
                            const bool yin0 = y0 >= 0 && y0 < a.H, yin1 = y0 + 1 >= 0 && y0 + 1 < a.H;
                            const float* sb = a.src[s] + (size_t)b * HW * C;                    // wave-uniform
                            const unsigned o = (unsigned)(((long)y0 * a.W + x0) * C + cq) * 4u;  // byte offset of tap 00 (used only where inside)
                            float* st = &s_stage[wv][s][0][0];
                            if (xin0 && yin0) MVS_DMA4(st, sb, o);
                            if (xin1 && yin0) MVS_DMA4(st + 64, sb, o + (unsigned)C * 4u);
                            if (xin0 && yin1) MVS_DMA4(st + 128, sb, o + (unsigned)(a.W * C) * 4u);
                            if (xin1 && yin1) MVS_DMA4(st + 192, sb, o + (unsigned)(a.W * C + C) * 4u);
                            stage_xy[s] = xl;
                        }
                    }
                }
                arith(g, wt);
                // this step's ring request: plane i + R into the slot just read (past the segment: the last plane again)
                MVS_DMA4(ring + (i % R) * 64, gseg + (size_t)min(i + R, L - 1) * gstep, voff * 4u);
                ++i;
            }
            // the blocks still held in registers, then this group's share of grad_ref (a pixel's lanes cover whole texels)
#pragma unroll
            for (int s = 0; s < NS_T; ++s)
                tb_flush_groups<C, 1, C>(live && cur[s] != TB_NOBLOCK, lane, cur[s], acc[s], a.H, a.W, wwin + s * VIEW_FLOATS + cq,
                                         w[s], use[s], a.gsrc[s] + fbase);
            if (!WARP_ONLY && live) MVS_GLOBAL_ATOMIC_ADD(a.gref + (size_t)b * HW * C + voff, gr);
            // the next group's (segment's) ring prologue must not overtake this walk's last requests into the same slots: they are all
            // older, and loads complete in order -- nothing to do
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
int g_sweep_bwd_cpl = 1;     // knob "bwd_cpl": accepted and ignored (the table form settled on one channel per lane)
int g_sweep_bwd_wf = 1536;   // knob "bwd_wf": floats of LDS window space per wave (1536, 2048 or 3200), 1-2 source views at C = 32
int g_sweep_bwd_pd = 8;      // knob "bwd_pd": depth of the LDS-DMA ring = look-ahead distance in planes (8 or 16)
extern int g_sweep_bwd_dslab;   // plane_sweep.hip: knobs "bwd_dslab", "bwd_nowin"
extern int g_sweep_bwd_nowin;

template <int C, int NS_T, int WF, int R>
static int launch_tb_mode(SweepArgs& a, dim3 grid, hipStream_t st) {
    if (a.warp_only) {
        if constexpr (NS_T == 1) MVS_LAUNCH((plane_sweep_variance_bwd_tb_kernel<C, 1, 2, WF, R>), grid, dim3(256), 0, st, a);
    } else if (a.ms_alias) MVS_LAUNCH((plane_sweep_variance_bwd_tb_kernel<C, NS_T, 1, WF, R>), grid, dim3(256), 0, st, a);
    else MVS_LAUNCH((plane_sweep_variance_bwd_tb_kernel<C, NS_T, 0, WF, R>), grid, dim3(256), 0, st, a);
    return mvs_check_launch("plane_sweep_variance_bwd_tb");
}

template <int C, int NS_T, int WF>
static int launch_tb_r(SweepArgs& a, dim3 grid, hipStream_t st) {
    // the deeper ring is an A/B instantiation for the benchmarked shapes only (C = 32, 2 or 4 source views)
    if constexpr (C == 32 && (NS_T == 2 || NS_T == 4)) {
        if (g_sweep_bwd_pd >= 16) return launch_tb_mode<C, NS_T, WF, 16>(a, grid, st);
    }
    return launch_tb_mode<C, NS_T, WF, 8>(a, grid, st);
}

template <int C, int NS_T>
static int launch_tb_wf(SweepArgs& a, dim3 grid, hipStream_t st) {
    // 3-4 views: 2x2-pixel blocks, 2048 floats (16 texels per view at C = 32).  1-2 views: 4x2 blocks, 1536 floats by default
    // (24 texels per view at C = 32; three workgroups per CU); the other sizes are A/B instantiations for C = 32, 2 views
    if constexpr (NS_T >= 3) return launch_tb_r<C, NS_T, 2048>(a, grid, st);
    if constexpr (C == 32 && NS_T == 2) {
        if (g_sweep_bwd_wf >= 3200) return launch_tb_r<C, NS_T, 3200>(a, grid, st);
        if (g_sweep_bwd_wf >= 2048) return launch_tb_r<C, NS_T, 2048>(a, grid, st);
    }
    return launch_tb_r<C, NS_T, 1536>(a, grid, st);
}

template <int C>
static int launch_tb_c(SweepArgs& a, hipStream_t st) {
    const int bw = a.NS <= 2 ? TbCfg<C, 1>::BW : TbCfg<C, 4>::BW;
    a.tiles_x = mvs_cdiv(a.W, 2 * bw);
    a.tiles_y = mvs_cdiv(a.H, 4);
    // depth slabs: >= ~2048 workgroups, each >= 16 planes: every extra slab re-gathers the blocks and writes its windows out once more
    const int tiles = a.tiles_x * a.tiles_y * a.B;
    int nslab = mvs_cdiv(2048, tiles);
    if (nslab > a.D / 16) nslab = a.D / 16;
    if (nslab < mvs_cdiv(a.D, 256)) nslab = mvs_cdiv(a.D, 256);     // the kernel stages a slab's per-plane depths in LDS (256 planes)
    if (nslab < 1) nslab = 1;
    a.dslab = g_sweep_bwd_dslab > 0 ? g_sweep_bwd_dslab : mvs_cdiv(a.D, nslab);
    if (a.dslab > 256) a.dslab = 256;
    a.no_window = g_sweep_bwd_nowin;
    dim3 grid(a.tiles_x * a.tiles_y, mvs_cdiv(a.D, a.dslab), a.B);
    switch (a.NS) {
        case 1: return launch_tb_wf<C, 1>(a, grid, st);
        case 2: return launch_tb_wf<C, 2>(a, grid, st);
        case 3: return launch_tb_wf<C, 3>(a, grid, st);
        case 4: return launch_tb_wf<C, 4>(a, grid, st);
        default: break;
    }
    mvs_set_error("plane_sweep backward (table form): 1..4 source views, got %d", a.NS);
    return MVS_ERR_UNSUPPORTED;
}

// 1..4 source views, image sides < 32000 (the packed base texel), a feature map below 4 GiB (32-bit byte offsets of the DMA);
// the caller falls back to the round-2 / round-1 kernels otherwise
bool sweep_bwd_tb_supports(const SweepArgs& a, int C) {
    return a.NS >= 1 && a.NS <= 4 && a.W < 32000 && a.H < 32000 && (double)a.H * a.W * C * 4.0 < 4.0e9;
}

int launch_sweep_bwd_tb(SweepArgs& a, int C, hipStream_t st) {
    if (C == 32) return launch_tb_c<32>(a, st);
    if (C == 16) return launch_tb_c<16>(a, st);
    return launch_tb_c<8>(a, st);
}
