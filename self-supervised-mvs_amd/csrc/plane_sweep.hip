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
// Thread map (both directions): 256 threads = (256/(C/4)) ref pixels of a small 2-D tile x (C/4)
// channel quads; each thread walks depth planes.
//
// Geometry (fp32):  q = rot*(x,y,1)*d + t ; p = q.xy / q.z ; sample index ix = p.x*W/(W-1) - 0.5
// (align_corners=False: what the reference's F.grid_sample call does on torch>=1.3, App. A Q1; =p.x for
// align_corners=True), bilinear weights w = ix - floor(ix), e = 1 - w, zero padding, no z>0 guard (Q14).
// The reference reaches the same index through  g = p/((W-1)/2) - 1 ; ix = ((g+1)*W - 1)/2  -- the same
// value up to fp32 rounding of the chain (tests bound both against an fp64 evaluation).
#include <stdlib.h>
#include <string.h>
#include "mvs_rt.h"

#include <type_traits>
#include "plane_sweep_common.h"

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// NS_T > 0: number of source views known at compile time (loops unroll, gathers of all views overlap);
// NS_T == 0: runtime a.NS.
template <int C, int NS_T>
__global__ __launch_bounds__(256) void plane_sweep_variance_fwd_kernel(SweepArgs a) {
    constexpr int QUADS = C / 4;
    constexpr int TW = Tile<C>::TW, TH = Tile<C>::TH;
    const int NS = NS_T > 0 ? NS_T : a.NS;
    const int tid = threadIdx.x;
    const int q = tid % QUADS, pl = tid / QUADS;
    const int x = (blockIdx.x % a.tiles_x) * TW + pl % TW, y = (blockIdx.x / a.tiles_x) * TH + pl / TW;
    const int b = blockIdx.z;
    if (x >= a.W || y >= a.H) return;  // no barriers / cross-lane ops below
    const int HW = a.H * a.W, pix = y * a.W + x;
    const int d0 = blockIdx.y * a.dslab;
    const int d1 = min(a.D, d0 + a.dslab);
    const float xf = (float)x, yf = (float)y;
    const size_t fbase = (size_t)b * HW * C + 4 * q;
    const float4 r = ld4(a.ref + fbase + (size_t)pix * C);
    const float4 r2 = make_float4(r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w);
    const float inv_n = 1.0f / (float)(NS + 1);
    const float* __restrict__ rotb = a.rot + (size_t)b * NS * 9;
    const float* __restrict__ trb = a.trans + (size_t)b * NS * 3;

    for (int d = d0; d < d1; ++d) {
        const float dep = a.per_pixel ? a.depth[((size_t)b * a.D + d) * HW + pix] : a.depth[b * a.D + d];
        float4 o;
        if (a.warp_only) {
            float ix, iy;
            source_index(rotb, trb, xf, yf, dep, a, ix, iy);
            o = sample4(a.src[0] + fbase, make_taps(ix, iy, a.H, a.W), a.W, C);
        } else {
            float4 S = a.ms_alias ? r2 : r;
            float4 Q = r2;
#pragma unroll
            for (int s = 0; s < (NS_T > 0 ? NS_T : 1); ++s) {
                for (int s2 = s; s2 < (NS_T > 0 ? s + 1 : NS); ++s2) {  // runtime loop only when NS_T == 0
                    float ix, iy;
                    source_index(rotb + s2 * 9, trb + s2 * 3, xf, yf, dep, a, ix, iy);
                    float4 v = sample4(a.src[s2] + fbase, make_taps(ix, iy, a.H, a.W), a.W, C);
                    S.x += v.x; S.y += v.y; S.z += v.z; S.w += v.w;
                    Q.x = fmaf(v.x, v.x, Q.x); Q.y = fmaf(v.y, v.y, Q.y); Q.z = fmaf(v.z, v.z, Q.z); Q.w = fmaf(v.w, v.w, Q.w);
                }
            }
            float m;
            m = S.x * inv_n; o.x = Q.x * inv_n - m * m;
            m = S.y * inv_n; o.y = Q.y * inv_n - m * m;
            m = S.z * inv_n; o.z = Q.z * inv_n - m * m;
            m = S.w * inv_n; o.w = Q.w * inv_n - m * m;
        }
        *reinterpret_cast<float4*>(a.var + (((size_t)b * a.D + d) * HW + pix) * C + 4 * q) = o;
    }
}


// Forward, register-cached variant (default).  Along a depth sweep the sample point of a pixel moves by a
// fraction of a source texel per plane (0.05-0.5 px at DTU-like geometry), so the 2x2 texel block under
// it changes only every few planes.  Each thread keeps that block (4 taps x CPT channels) in registers and
// re-gathers it only when floor(ix) or floor(iy) changes: vector-L1 traffic drops by the average run
// length (~3-12x), the kernel is left with the index arithmetic and the one streaming store.
// CPT = channels per thread (4 or 8): 8 halves the redundancy of the per-pixel index chain.
template <int C, int CPT> struct TileC {
    static constexpr int LPP = C / CPT, PPB = 256 / LPP, TW = PPB >= 128 ? 16 : 8, TH = PPB / TW;
};

// (Round 3's "shared projection" form -- one lane of a pixel's lane group computes a view's projection, the others fetch it by DPP /
// ds_bpermute -- was bit-identical and slower everywhere (N=3 0.1023 -> 0.1064 ms, N=7 bf16 2.57 -> 2.87) and was removed in round 4.)
// BF (inference path, BASELINE configs[4]): the volume is stored in bf16 (round to nearest even) -- a lane then owns CPT
// CONSECUTIVE channels so that its values are one 16- (8-) byte store and the lanes of a pixel write one 64-byte segment.
// DL (per-plane hypotheses only): the slab's depths are staged in LDS once and read back one plane ahead.  DL == 2 is the MERGED
// form: a plane first walks all views (projection, block test, re-gather loads issued), then samples all views -- the taps of
// every view that left its block are in flight together, one memory round trip per plane instead of one per such view (the
// plain form consumes view s's taps right after requesting them: up to NS_T sequential round trips).  Same arithmetic, same order.
// (An earlier DL == 2, waiting for the taps INSIDE the re-gather block, measured slower and is gone.)  Without DL the plane loop
// starts with a vmcnt(0) (the depth is a vector load) that also waits for the previous plane's stores -- on gfx9 stores and loads
// share the counter.
template <int C, int NS_T, int CPT, bool BF = false, int DL = 0>
__global__ __launch_bounds__(256) void plane_sweep_variance_fwd_cached_kernel(SweepArgs a) {
    static_assert(!BF || CPT == 8 || CPT == 4, "bf16 store: 4 or 8 consecutive channels per thread");
    constexpr int V = CPT / 4;                     // float4s per tap per thread
    constexpr int LPP = TileC<C, CPT>::LPP, PPB = TileC<C, CPT>::PPB;
    const int TW = a.tile_w, TH = PPB / TW;
    const int tid = threadIdx.x;
    const int q = tid % LPP, pl = tid / LPP;
    const SweepWg wg = sweep_wg(a);
    const int xr = (wg.tile % a.tiles_x) * TW + pl % TW, yr = (wg.tile / a.tiles_x) * TH + pl / TW;
    const int b = wg.b;
    const int d0 = wg.slab * a.dslab;
    const int d1 = min(a.D, d0 + a.dslab);
    __shared__ float s_dep[DL ? 512 : 1];            // DL: the launcher keeps a slab <= 512 planes
    if constexpr (DL != 0) {
        for (int i = tid; i < d1 - d0; i += 256) s_dep[i] = a.depth[b * a.D + d0 + i];
        __syncthreads();
    }
    if (xr >= a.W || yr >= a.H) return;  // no barriers / cross-lane ops below
    const int x = xr, y = yr;
    const int HW = a.H * a.W, pix = y * a.W + x;
    const float xf = (float)x, yf = (float)y;
    // channel of float4 k of lane q: 4q + 4*LPP*k, so that ONE store instruction writes whole 64-byte segments
    // (with CPT*q + 4k every store instruction wrote 16 of each 32 bytes: 5 % slower, profiles/r01_run19_kernels.log)
    const int cq = BF ? CPT * q : 4 * q;
    constexpr int ck = BF ? 4 : 4 * LPP;
    const size_t fbase = (size_t)b * HW * C + cq;
    float4 r[V], r2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        r[k] = ld4(a.ref + fbase + (size_t)pix * C + ck * k);
        r2[k] = make_float4(r[k].x * r[k].x, r[k].y * r[k].y, r[k].z * r[k].z, r[k].w * r[k].w);
    }
    const float inv_n = 1.0f / (float)(NS_T + 1);
    const float* __restrict__ rotb = a.rot + (size_t)b * NS_T * 9;
    const float* __restrict__ trb = a.trans + (size_t)b * NS_T * 3;
    // Measured and rejected (profiles/README.md): a branch-free re-gather (clamped addresses, padding folded into the
    // weights) 5 % slower; keeping the two texels that stay in the block on a one-texel step (parity bits, half the
    // gathers) no faster; caching the block as bilinear coefficients (3 FMAs per sample instead of 4) 10 % slower --
    // hipcc then packs the arithmetic into v_pk_fma_f32, which issues at 4.6 cycles against 2 x 2.8.
    // per view: homography rows applied to (x,y,1) once, cached base texel and its 2x2 block
    float rx[NS_T], ry[NS_T], rz[NS_T], tx[NS_T], ty[NS_T], tz[NS_T];
    int cx[NS_T], cy[NS_T];
    float4 t00[NS_T][V], t01[NS_T][V], t10[NS_T][V], t11[NS_T][V];
#pragma unroll
    for (int s = 0; s < NS_T; ++s) {
        const int sv = s;
        const float* R = rotb + sv * 9;
        rx[s] = fmaf(R[0], xf, fmaf(R[1], yf, R[2]));
        ry[s] = fmaf(R[3], xf, fmaf(R[4], yf, R[5]));
        rz[s] = fmaf(R[6], xf, fmaf(R[7], yf, R[8]));
        tx[s] = trb[sv * 3]; ty[s] = trb[sv * 3 + 1]; tz[s] = trb[sv * 3 + 2];
        cx[s] = -0x40000000; cy[s] = -0x40000000;
#pragma unroll
        for (int k = 0; k < V; ++k) t00[s][k] = t01[s][k] = t10[s][k] = t11[s][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // the plane's depth is requested one plane ahead (a load issued at the top of the iteration is consumed by its very next
    // instruction: one memory round trip per plane; the same change took the backward from 0.41 to 0.34 ms)
    const float* __restrict__ dptr = a.depth + (a.per_pixel ? ((size_t)b * a.D + d0) * HW + pix : (size_t)b * a.D + d0);
    const size_t dstep = a.per_pixel ? (size_t)HW : 1;
    float dep_next;
    if constexpr (DL != 0) dep_next = s_dep[0];
    else dep_next = dptr[0];
    for (int d = d0; d < d1; ++d) {
        const float dep = dep_next;
        if (d + 1 < d1) {
            if constexpr (DL != 0) dep_next = s_dep[d + 1 - d0];
            else dep_next = dptr[(size_t)(d + 1 - d0) * dstep];
        }
        float4 S[V], Q[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { S[k] = a.ms_alias ? r2[k] : r[k]; Q[k] = r2[k]; }
        float wxs[DL == 2 ? NS_T : 1], wys[DL == 2 ? NS_T : 1];
#pragma unroll
        for (int s = 0; s < NS_T; ++s) {
            float wx, wy;
            int x0, y0;
            {
                const float zz = fmaf(rz[s], dep, tz[s]);
                float iz = MVS_RCP(zz);                 // v_rcp_f32 (1 ulp) + one Newton step: < 1 ulp, 3 instructions
                iz = fmaf(fmaf(-zz, iz, 1.0f), iz, iz); // instead of the ~10 of an IEEE division
                const float ix = fmaf(fmaf(rx[s], dep, tx[s]) * iz, a.sx, a.ox);
                const float iy = fmaf(fmaf(ry[s], dep, ty[s]) * iz, a.sy, a.oy);
                const float fx = floorf(ix), fy = floorf(iy);
                wx = ix - fx; wy = iy - fy;
                // v_cvt_i32_f32 saturates (huge -> INT_MAX/INT_MIN: every tap outside the image; NaN -> 0 with NaN
                // weights, i.e. NaN out like ATen), so no float clamp is needed before the conversion
                x0 = MVS_F2I(fx); y0 = MVS_F2I(fy);
            }
            if constexpr (DL == 2) { wxs[s] = wx; wys[s] = wy; }
            const float ex = 1.0f - wx, ey = 1.0f - wy;
            if (x0 != cx[s] || y0 != cy[s]) {
                cx[s] = x0; cy[s] = y0;
                const bool xin0 = x0 >= 0 && x0 < a.W, xin1 = x0 + 1 >= 0 && x0 + 1 < a.W;
                const bool yin0 = y0 >= 0 && y0 < a.H, yin1 = y0 + 1 >= 0 && y0 + 1 < a.H;
                const float* __restrict__ f = a.src[s] + fbase + ((long)y0 * a.W + x0) * C;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    t00[s][k] = (xin0 && yin0) ? ld4(f + ck * k) : z4;
                    t01[s][k] = (xin1 && yin0) ? ld4(f + C + ck * k) : z4;
                    t10[s][k] = (xin0 && yin1) ? ld4(f + a.W * C + ck * k) : z4;
                    t11[s][k] = (xin1 && yin1) ? ld4(f + a.W * C + C + ck * k) : z4;
                }
            }
            if constexpr (DL == 2) continue;   // merged form: every view's re-gather is in flight before the first sample (below)
            const float w00 = ey * ex, w01 = ey * wx, w10 = wy * ex, w11 = wy * wx;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float4 v;
                v.x = fmaf(t11[s][k].x, w11, fmaf(t10[s][k].x, w10, fmaf(t01[s][k].x, w01, t00[s][k].x * w00)));
                v.y = fmaf(t11[s][k].y, w11, fmaf(t10[s][k].y, w10, fmaf(t01[s][k].y, w01, t00[s][k].y * w00)));
                v.z = fmaf(t11[s][k].z, w11, fmaf(t10[s][k].z, w10, fmaf(t01[s][k].z, w01, t00[s][k].z * w00)));
                v.w = fmaf(t11[s][k].w, w11, fmaf(t10[s][k].w, w10, fmaf(t01[s][k].w, w01, t00[s][k].w * w00)));
                S[k].x += v.x; S[k].y += v.y; S[k].z += v.z; S[k].w += v.w;
                Q[k].x = fmaf(v.x, v.x, Q[k].x); Q[k].y = fmaf(v.y, v.y, Q[k].y);
                Q[k].z = fmaf(v.z, v.z, Q[k].z); Q[k].w = fmaf(v.w, v.w, Q[k].w);
            }
        }
        if constexpr (DL == 2) {
            // second phase of the merged form: the samples, in the same order and with the same arithmetic as above
#pragma unroll
            for (int s = 0; s < NS_T; ++s) {
                const float wx = wxs[s], wy = wys[s];
                const float ex = 1.0f - wx, ey = 1.0f - wy;
                const float w00 = ey * ex, w01 = ey * wx, w10 = wy * ex, w11 = wy * wx;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float4 v;
                    v.x = fmaf(t11[s][k].x, w11, fmaf(t10[s][k].x, w10, fmaf(t01[s][k].x, w01, t00[s][k].x * w00)));
                    v.y = fmaf(t11[s][k].y, w11, fmaf(t10[s][k].y, w10, fmaf(t01[s][k].y, w01, t00[s][k].y * w00)));
                    v.z = fmaf(t11[s][k].z, w11, fmaf(t10[s][k].z, w10, fmaf(t01[s][k].z, w01, t00[s][k].z * w00)));
                    v.w = fmaf(t11[s][k].w, w11, fmaf(t10[s][k].w, w10, fmaf(t01[s][k].w, w01, t00[s][k].w * w00)));
                    S[k].x += v.x; S[k].y += v.y; S[k].z += v.z; S[k].w += v.w;
                    Q[k].x = fmaf(v.x, v.x, Q[k].x); Q[k].y = fmaf(v.y, v.y, Q[k].y);
                    Q[k].z = fmaf(v.z, v.z, Q[k].z); Q[k].w = fmaf(v.w, v.w, Q[k].w);
                }
            }
        }
        const size_t oidx = (((size_t)b * a.D + d) * HW + pix) * C + cq;
        float* __restrict__ outp = a.var + oidx;
        unsigned packed[2 * V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float4 o;
            float m;
            m = S[k].x * inv_n; o.x = Q[k].x * inv_n - m * m;
            m = S[k].y * inv_n; o.y = Q[k].y * inv_n - m * m;
            m = S[k].z * inv_n; o.z = Q[k].z * inv_n - m * m;
            m = S[k].w * inv_n; o.w = Q[k].w * inv_n - m * m;
            if (BF) {
                packed[2 * k] = mvs_cvt_pk_bf16(o.x, o.y);
                packed[2 * k + 1] = mvs_cvt_pk_bf16(o.z, o.w);
                continue;
            }
            if (a.nt_store) MVS_NT_STORE4(outp + ck * k, o);
            else *reinterpret_cast<float4*>(outp + ck * k) = o;
        }
        if (BF) {
            if (V == 2) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.var) + oidx) =
                            make_uint4(packed[0], packed[1], packed[2 % (2 * V)], packed[3 % (2 * V)]);
            else { uint2 o2; o2.x = packed[0]; o2.y = packed[1]; *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.var) + oidx) = o2; }
        }
    }
}

// Forward, register-cached taps + PROJECTION TABLE (round 6; per-plane hypotheses).  In the kernel above every one of the LPP lanes of a
// pixel repeats the pixel's projective arithmetic for every view and plane: 16 instructions of projection + 6 of weights per (view,
// plane) -- at 7 views that is 52 % of a VALU-bound kernel (84 % issue: DESIGN.md section 4).  Here the 64 / LPP pixels of a wave and the
// NS_T views form <= 64 (pixel, view) PAIRS and lane l of the wave computes pair (pixel l % PPW, view l / PPW) ONCE per plane -- base texel
// and the four bilinear weights -- into a per-wave LDS table that the lanes of the pixel read back (two broadcast reads per view).  The
// table is double-buffered by plane parity and filled one plane ahead, so a plane's reads never wait for its own writes; a wave's DS
// operations execute in order: no barrier in the plane loop.  Same arithmetic on the same values as the kernel above: bit-identical.
// Round 3's "shared projection" (DPP / ds_bpermute exchange between the lanes of a pixel, each still walking all views) was slower;
// this form divides the work instead of exchanging it -- and is slower too (knob "fwd_pt", off: N=3 0.108 -> 0.117 ms, N=7 bf16 2.56 ->
// 2.95 with 26 % fewer vector instructions per plane): the table's LDS round trip and the wave-wide dependency on the slowest pair weigh
// more than the instructions saved.  K1's "84 % VALU issue" is a symptom of how its waves overlap, not a budget that buys time back.
template <int C, int NS_T, int CPT, bool BF = false>
__global__ __launch_bounds__(256) void plane_sweep_variance_fwd_pt_kernel(SweepArgs a) {
    static_assert(!BF || CPT == 8 || CPT == 4, "bf16 store: 4 or 8 consecutive channels per thread");
    constexpr int V = CPT / 4;
    constexpr int LPP = TileC<C, CPT>::LPP, PPB = TileC<C, CPT>::PPB, PPW = 64 / LPP;
    static_assert(PPW * NS_T <= 64, "the (pixel, view) pairs of a wave must fit its 64 lanes");
    const int TW = a.tile_w, TH = PPB / TW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = tid % LPP, pl = tid / LPP;
    const SweepWg wg = sweep_wg(a);
    const int tx0 = (wg.tile % a.tiles_x) * TW, ty0 = (wg.tile / a.tiles_x) * TH;
    const int xr = tx0 + pl % TW, yr = ty0 + pl / TW;
    const int b = wg.b;
    const int d0 = wg.slab * a.dslab;
    const int d1 = min(a.D, d0 + a.dslab);
    __shared__ float s_dep[512];                                         // the launcher keeps a slab <= 512 planes
    __shared__ __attribute__((aligned(16))) float s_w[4][2][NS_T][PPW][4];   // [wave][plane parity][view][pixel]: w00, w01, w10, w11
    __shared__ __attribute__((aligned(8))) int s_xy[4][2][NS_T][PPW][2];    //                                   : x0, y0
    for (int i = tid; i < d1 - d0; i += 256) s_dep[i] = a.depth[b * a.D + d0 + i];
    __syncthreads();
    // a lane outside the image stays (the wave shares the table); it works on a clamped pixel and does not store
    const bool live = xr < a.W && yr < a.H;
    const int x = min(xr, a.W - 1), y = min(yr, a.H - 1);
    const int HW = a.H * a.W, pix = y * a.W + x;
    const int cq = BF ? CPT * q : 4 * q;
    constexpr int ck = BF ? 4 : 4 * LPP;
    const size_t fbase = (size_t)b * HW * C + cq;
    float4 r[V], r2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        r[k] = ld4(a.ref + fbase + (size_t)pix * C + ck * k);
        r2[k] = make_float4(r[k].x * r[k].x, r[k].y * r[k].y, r[k].z * r[k].z, r[k].w * r[k].w);
    }
    const float inv_n = 1.0f / (float)(NS_T + 1);
    // ---- this lane as the owner of ONE (pixel, view) pair of its wave ----
    const int pp = lane % PPW, ps = lane / PPW;                          // pair: pixel pp of the wave, view ps (< NS_T: a real pair)
    const bool pair = ps < NS_T;
    const int ppl = wave * PPW + pp;                                      // the pair's pixel within the workgroup's tile
    const float pxf = (float)min(tx0 + ppl % TW, a.W - 1), pyf = (float)min(ty0 + ppl / TW, a.H - 1);
    const float* __restrict__ R = a.rot + ((size_t)b * NS_T + (pair ? ps : 0)) * 9;
    const float* __restrict__ T = a.trans + ((size_t)b * NS_T + (pair ? ps : 0)) * 3;
    const float prx = fmaf(R[0], pxf, fmaf(R[1], pyf, R[2]));
    const float pry = fmaf(R[3], pxf, fmaf(R[4], pyf, R[5]));
    const float prz = fmaf(R[6], pxf, fmaf(R[7], pyf, R[8]));
    const float ptx = T[0], pty = T[1], ptz = T[2];
    auto fill = [&](int d, int par) __attribute__((always_inline)) {
        if (!pair) return;
        const float dep = s_dep[d - d0];
        const float zz = fmaf(prz, dep, ptz);
        float iz = MVS_RCP(zz);
        iz = fmaf(fmaf(-zz, iz, 1.0f), iz, iz);
        const float ix = fmaf(fmaf(prx, dep, ptx) * iz, a.sx, a.ox);
        const float iy = fmaf(fmaf(pry, dep, pty) * iz, a.sy, a.oy);
        const float fx = floorf(ix), fy = floorf(iy);
        const float wx = ix - fx, wy = iy - fy;
        const float ex = 1.0f - wx, ey = 1.0f - wy;
        *reinterpret_cast<float4*>(&s_w[wave][par][ps][pp][0]) = make_float4(ey * ex, ey * wx, wy * ex, wy * wx);
        s_xy[wave][par][ps][pp][0] = MVS_F2I(fx);
        s_xy[wave][par][ps][pp][1] = MVS_F2I(fy);
    };
    int cx[NS_T], cy[NS_T];
    float4 t00[NS_T][V], t01[NS_T][V], t10[NS_T][V], t11[NS_T][V];
#pragma unroll
    for (int s = 0; s < NS_T; ++s) {
        cx[s] = -0x40000000; cy[s] = -0x40000000;
#pragma unroll
        for (int k = 0; k < V; ++k) t00[s][k] = t01[s][k] = t10[s][k] = t11[s][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int mp = pl % PPW;                                              // this lane's pixel within its wave's table
    fill(d0, 0);
    for (int d = d0; d < d1; ++d) {
        const int par = (d - d0) & 1;
        MVS_WAVE_SYNC();                                                 // every lane has finished plane d - 1's reads of the other buffer
        if (d + 1 < d1) fill(d + 1, par ^ 1);                            // one plane ahead, into the buffer plane d - 1 has finished with
        MVS_WAVE_SYNC();
        float4 S[V], Q[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { S[k] = a.ms_alias ? r2[k] : r[k]; Q[k] = r2[k]; }
        // every view's table entry first (2 x NS_T reads in flight, ONE wait): read inside the view loop, each view's block test waited
        // for its own LDS round trip
        float4 wts[NS_T];
        int x0s[NS_T], y0s[NS_T];
#pragma unroll
        for (int s = 0; s < NS_T; ++s) {
            wts[s] = *reinterpret_cast<const float4*>(&s_w[wave][par][s][mp][0]);
            x0s[s] = s_xy[wave][par][s][mp][0]; y0s[s] = s_xy[wave][par][s][mp][1];
        }
        MVS_SCHED_FENCE();
#pragma unroll
        for (int s = 0; s < NS_T; ++s) {
            const float4 wt = wts[s];
            const int x0 = x0s[s], y0 = y0s[s];
            if (x0 != cx[s] || y0 != cy[s]) {
                cx[s] = x0; cy[s] = y0;
                const bool xin0 = x0 >= 0 && x0 < a.W, xin1 = x0 + 1 >= 0 && x0 + 1 < a.W;
                const bool yin0 = y0 >= 0 && y0 < a.H, yin1 = y0 + 1 >= 0 && y0 + 1 < a.H;
                const float* __restrict__ f = a.src[s] + fbase + ((long)y0 * a.W + x0) * C;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    t00[s][k] = (xin0 && yin0) ? ld4(f + ck * k) : z4;
                    t01[s][k] = (xin1 && yin0) ? ld4(f + C + ck * k) : z4;
                    t10[s][k] = (xin0 && yin1) ? ld4(f + a.W * C + ck * k) : z4;
                    t11[s][k] = (xin1 && yin1) ? ld4(f + a.W * C + C + ck * k) : z4;
                }
            }
            const float w00 = wt.x, w01 = wt.y, w10 = wt.z, w11 = wt.w;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float4 v;
                v.x = fmaf(t11[s][k].x, w11, fmaf(t10[s][k].x, w10, fmaf(t01[s][k].x, w01, t00[s][k].x * w00)));
                v.y = fmaf(t11[s][k].y, w11, fmaf(t10[s][k].y, w10, fmaf(t01[s][k].y, w01, t00[s][k].y * w00)));
                v.z = fmaf(t11[s][k].z, w11, fmaf(t10[s][k].z, w10, fmaf(t01[s][k].z, w01, t00[s][k].z * w00)));
                v.w = fmaf(t11[s][k].w, w11, fmaf(t10[s][k].w, w10, fmaf(t01[s][k].w, w01, t00[s][k].w * w00)));
                S[k].x += v.x; S[k].y += v.y; S[k].z += v.z; S[k].w += v.w;
                Q[k].x = fmaf(v.x, v.x, Q[k].x); Q[k].y = fmaf(v.y, v.y, Q[k].y);
                Q[k].z = fmaf(v.z, v.z, Q[k].z); Q[k].w = fmaf(v.w, v.w, Q[k].w);
            }
        }
        const size_t oidx = (((size_t)b * a.D + d) * HW + pix) * C + cq;
        float* __restrict__ outp = a.var + oidx;
        unsigned packed[2 * V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float4 o;
            float m;
            m = S[k].x * inv_n; o.x = Q[k].x * inv_n - m * m;
            m = S[k].y * inv_n; o.y = Q[k].y * inv_n - m * m;
            m = S[k].z * inv_n; o.z = Q[k].z * inv_n - m * m;
            m = S[k].w * inv_n; o.w = Q[k].w * inv_n - m * m;
            if (BF) {
                packed[2 * k] = mvs_cvt_pk_bf16(o.x, o.y);
                packed[2 * k + 1] = mvs_cvt_pk_bf16(o.z, o.w);
                continue;
            }
            if (!live) continue;
            if (a.nt_store) MVS_NT_STORE4(outp + ck * k, o);
            else *reinterpret_cast<float4*>(outp + ck * k) = o;
        }
        if (BF && live) {
            if (V == 2) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.var) + oidx) =
                            make_uint4(packed[0], packed[1], packed[2 % (2 * V)], packed[3 % (2 * V)]);
            else { uint2 o2; o2.x = packed[0]; o2.y = packed[1]; *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.var) + oidx) = o2; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward.  With Sm = S/N:  dL/dv_i = g*(2/N)*(v_i - Sm);
//   dL/dr = sum_d g*(2/N)*(r - Sm)            (MVSNet)
//   dL/dr = sum_d g*(2r/N)*(1 - 2*Sm)         (jdacs-ms alias quirk, S starts from r^2)
// No gradient to cameras / depths (the reference builds the grid under no_grad, module.py:115).
//
// The bilinear scatter of dL/dv_i is privatised in LDS: a workgroup owns a pixel tile and up to two
// source views; depth planes are walked in segments chosen so that the tile's projected footprint
// (bounding box of the 8 corners of the pixel-tile x depth-range box; the warp is projective so the
// box maps into their hull) fits an LDS window per view.  Taps inside the window use LDS float
// atomics, the window is flushed once per segment with global atomics (non-zero entries only); a tap
// outside the window (rounding at the hull, extreme zoom) falls back to a global atomic, so the
// result is always complete.  Device-scope atomics drop from 4 per (voxel, view, channel) to about
// one per (window texel, channel) per segment (~30x fewer at BASELINE config 2).
// ------------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void scatter_tap(float* __restrict__ win, const Win& w, bool use_win, float* __restrict__ g,
                                            int xi, int yi, int W, int q, const float4& gv, float wt) {
    const int wx = xi - w.x0, wy = yi - w.y0;
    if (use_win && wx >= 0 && wx < w.w && wy >= 0 && wy < w.h) {
        float* p = win + (wy * w.w + wx) * BwdCfg<C>::CP + 4 * q;
        MVS_LDS_ATOMIC_ADD(p + 0, gv.x * wt); MVS_LDS_ATOMIC_ADD(p + 1, gv.y * wt);
        MVS_LDS_ATOMIC_ADD(p + 2, gv.z * wt); MVS_LDS_ATOMIC_ADD(p + 3, gv.w * wt);
    } else {
        float* p = g + ((size_t)yi * W + xi) * C;
        MVS_GLOBAL_ATOMIC_ADD(p + 0, gv.x * wt); MVS_GLOBAL_ATOMIC_ADD(p + 1, gv.y * wt);
        MVS_GLOBAL_ATOMIC_ADD(p + 2, gv.z * wt); MVS_GLOBAL_ATOMIC_ADD(p + 3, gv.w * wt);
    }
}


// Register-resident 2x2 source texel block of one (pixel, channel quad, view): tap values + grad accumulators.
// (Measured and rejected, run 27: flushing only the texel pair that leaves the block on a one-texel step -- half the LDS
//  atomics and gathers -- 0.63 -> 0.66 ms: 241 instead of 207 VGPRs and more VALU outweigh it.)
struct ViewCache {
    int cx, cy;
    float4 t00, t01, t10, t11;   // tap values (zero where the tap is outside the image)
    float4 g00, g01, g10, g11;   // gradient accumulated since the block was entered
    __device__ __forceinline__ void reset() {
        cx = cy = -0x40000000;
        t00 = t01 = t10 = t11 = g00 = g01 = g10 = g11 = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    template <int C>
    __device__ __forceinline__ void flush(int H, int W, float* __restrict__ win, const Win& w, bool use_win,
                                          float* __restrict__ gp, int q) {
        if (cx == -0x40000000) return;
        const bool xin0 = cx >= 0 && cx < W, xin1 = cx + 1 >= 0 && cx + 1 < W;
        const bool yin0 = cy >= 0 && cy < H, yin1 = cy + 1 >= 0 && cy + 1 < H;
        if (xin0 && yin0) scatter_tap<C>(win, w, use_win, gp, cx, cy, W, q, g00, 1.0f);
        if (xin1 && yin0) scatter_tap<C>(win, w, use_win, gp, cx + 1, cy, W, q, g01, 1.0f);
        if (xin0 && yin1) scatter_tap<C>(win, w, use_win, gp, cx, cy + 1, W, q, g10, 1.0f);
        if (xin1 && yin1) scatter_tap<C>(win, w, use_win, gp, cx + 1, cy + 1, W, q, g11, 1.0f);
        g00 = g01 = g10 = g11 = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // move to the texel block of (ix, iy) (flush + re-gather if it changed); returns the bilinear sample, wt[4] = weights
    template <int C>
    __device__ __forceinline__ float4 advance(float ix, float iy, const float* __restrict__ f, int H, int W,
                                              float* __restrict__ win, const Win& w, bool use_win,
                                              float* __restrict__ gp, int q, float (&wt)[4]) {
        const float fx = floorf(ix), fy = floorf(iy);
        const float wx = ix - fx, wy = iy - fy;
        const float ex = 1.0f - wx, ey = 1.0f - wy;
        const float fxc = fminf(fmaxf(fx, -2.0f), (float)W), fyc = fminf(fmaxf(fy, -2.0f), (float)H);
        const int x0 = (fxc == fxc) ? (int)fxc : -2, y0 = (fyc == fyc) ? (int)fyc : -2;
        if (x0 != cx || y0 != cy) {
            flush<C>(H, W, win, w, use_win, gp, q);
            cx = x0; cy = y0;
            const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
            const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
            const float* __restrict__ p = f + ((long)y0 * W + x0) * C;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            t00 = (xin0 && yin0) ? ld4(p) : z4;
            t01 = (xin1 && yin0) ? ld4(p + C) : z4;
            t10 = (xin0 && yin1) ? ld4(p + W * C) : z4;
            t11 = (xin1 && yin1) ? ld4(p + W * C + C) : z4;
        }
        wt[0] = ey * ex; wt[1] = ey * wx; wt[2] = wy * ex; wt[3] = wy * wx;
        float4 v;
        v.x = fmaf(t11.x, wt[3], fmaf(t10.x, wt[2], fmaf(t01.x, wt[1], t00.x * wt[0])));
        v.y = fmaf(t11.y, wt[3], fmaf(t10.y, wt[2], fmaf(t01.y, wt[1], t00.y * wt[0])));
        v.z = fmaf(t11.z, wt[3], fmaf(t10.z, wt[2], fmaf(t01.z, wt[1], t00.z * wt[0])));
        v.w = fmaf(t11.w, wt[3], fmaf(t10.w, wt[2], fmaf(t01.w, wt[1], t00.w * wt[0])));
        return v;
    }
    __device__ __forceinline__ void accumulate(const float4& gv, const float (&wt)[4]) {
        g00.x = fmaf(gv.x, wt[0], g00.x); g00.y = fmaf(gv.y, wt[0], g00.y); g00.z = fmaf(gv.z, wt[0], g00.z); g00.w = fmaf(gv.w, wt[0], g00.w);
        g01.x = fmaf(gv.x, wt[1], g01.x); g01.y = fmaf(gv.y, wt[1], g01.y); g01.z = fmaf(gv.z, wt[1], g01.z); g01.w = fmaf(gv.w, wt[1], g01.w);
        g10.x = fmaf(gv.x, wt[2], g10.x); g10.y = fmaf(gv.y, wt[2], g10.y); g10.z = fmaf(gv.z, wt[2], g10.z); g10.w = fmaf(gv.w, wt[2], g10.w);
        g11.x = fmaf(gv.x, wt[3], g11.x); g11.y = fmaf(gv.y, wt[3], g11.y); g11.z = fmaf(gv.z, wt[3], g11.z); g11.w = fmaf(gv.w, wt[3], g11.w);
    }
};

template <int C>
__global__ __launch_bounds__(256) void plane_sweep_variance_bwd_kernel(SweepArgs a) {
    constexpr int QUADS = C / 4;
    constexpr int TW = Tile<C>::TW, TH = Tile<C>::TH;
    constexpr int WCAP = BwdCfg<C>::WCAP, CP = BwdCfg<C>::CP;
    __shared__ float win[2 * WCAP * CP];
    __shared__ float red[8];
    const int NS = a.NS;
    const int tid = threadIdx.x;
    const int q = tid % QUADS, pl = tid / QUADS;
    const int tx0 = (blockIdx.x % a.tiles_x) * TW, ty0 = (blockIdx.x / a.tiles_x) * TH;
    const int x = tx0 + pl % TW, y = ty0 + pl / TW;
    const int ngroups = (NS + 1) / 2;
    const int vg = blockIdx.y % ngroups;  // view group: sources 2*vg, 2*vg+1
    const int slab = blockIdx.y / ngroups;  // depth slab [slab*dslab, +dslab): more workgroups than tiles alone
    const int sA = 2 * vg, sB = (2 * vg + 1 < NS) ? 2 * vg + 1 : -1;
    const int b = blockIdx.z;
    const bool valid = x < a.W && y < a.H;
    const int HW = a.H * a.W, pix = valid ? y * a.W + x : 0;
    const float xf = (float)x, yf = (float)y;
    const size_t fbase = (size_t)b * HW * C + 4 * q;
    const float4 r = ld4(a.ref + fbase + (size_t)pix * C);
    const float inv_n = 1.0f / (float)(NS + 1);
    const float two_n = 2.0f * inv_n;
    const float* __restrict__ rotb = a.rot + (size_t)b * NS * 9;
    const float* __restrict__ trb = a.trans + (size_t)b * NS * 3;
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
    // tile corners (clipped to the image) for the footprint bound
    const float cxa = (float)tx0, cxb = (float)min(tx0 + TW - 1, a.W - 1);
    const float cya = (float)ty0, cyb = (float)min(ty0 + TH - 1, a.H - 1);

    int ds = slab * a.dslab;
    const int dend = min(a.D, ds + a.dslab);
    while (ds < dend) {
        // ---- choose the segment [ds, de) and the windows (block-uniform arithmetic) ----
        int de = dend;
        Win wA_, wB_;
        bool fits = false;
        for (int it = 0; it < 12; ++it) {
            float da, db;
            if (a.per_pixel) {
                // depth range of the tile over the segment: block min/max reduction
                float lo = 3.0e38f, hi = -3.0e38f;
                if (valid && q == 0)
                    for (int d = ds; d < de; ++d) {
                        float v = a.depth[((size_t)b * a.D + d) * HW + pix];
                        lo = fminf(lo, v); hi = fmaxf(hi, v);
                    }
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { lo = fminf(lo, __shfl_xor(lo, m)); hi = fmaxf(hi, __shfl_xor(hi, m)); }
                __syncthreads();
                if ((tid & 63) == 0) { red[(tid >> 6) * 2] = lo; red[(tid >> 6) * 2 + 1] = hi; }
                __syncthreads();
                da = fminf(fminf(red[0], red[2]), fminf(red[4], red[6]));
                db = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
            } else {
                da = a.depth[b * a.D + ds];
                db = a.depth[b * a.D + de - 1];
            }
            float lox, hix, loy, hiy;
            corner_bounds(a, rotb + sA * 9, trb + sA * 3, cxa, cxb, cya, cyb, da, db, lox, hix, loy, hiy);
            wA_ = make_window(a, lox, hix, loy, hiy);
            fits = (long)wA_.w * wA_.h <= WCAP;
            if (sB >= 0) {
                corner_bounds(a, rotb + sB * 9, trb + sB * 3, cxa, cxb, cya, cyb, da, db, lox, hix, loy, hiy);
                wB_ = make_window(a, lox, hix, loy, hiy);
                fits = fits && (long)wB_.w * wB_.h <= WCAP;
            } else {
                wB_ = wA_;
            }
            if (fits || de - ds <= 1) break;
            de = ds + (de - ds + 1) / 2;
        }
        const bool useA = (long)wA_.w * wA_.h <= WCAP, useB = sB >= 0 && (long)wB_.w * wB_.h <= WCAP;
        // ---- zero the windows ----
        __syncthreads();
        if (useA) for (int i = tid; i < wA_.w * wA_.h * CP; i += 256) win[i] = 0.f;
        if (useB) for (int i = tid; i < wB_.w * wB_.h * CP; i += 256) win[WCAP * CP + i] = 0.f;
        __syncthreads();
        // ---- walk the planes of the segment ----
        // Per scatter view the thread keeps the 2x2 source texel block under its sample point in
        // registers -- both the tap VALUES (for v_i) and the gradient ACCUMULATORS -- and touches memory
        // only when floor(ix)/floor(iy) changes (every ~3-12 planes): LDS float atomics run at about one
        // lane per clock per CU, so issuing 4 taps x 4 channels of them per plane and view was the bottleneck.
        if (valid) {
            ViewCache cA, cB;
            cA.reset();
            cB.reset();
            float* const gpA = a.gsrc[sA] + fbase;
            float* const gpB = sB >= 0 ? a.gsrc[sB] + fbase : gpA;
            float* const winB = win + WCAP * CP;
            // the upstream gradient streams from HBM (one float4 per plane and thread): fetch plane d+1 while plane d
            // is processed, otherwise every iteration eats a full HBM round trip at 2 waves/SIMD
            float4 g_next = ld4(a.gvar + (((size_t)b * a.D + ds) * HW + pix) * C + 4 * q);
            for (int d = ds; d < de; ++d) {
                const float dep = a.per_pixel ? a.depth[((size_t)b * a.D + d) * HW + pix] : a.depth[b * a.D + d];
                const float4 g = g_next;
                if (d + 1 < de) g_next = ld4(a.gvar + (((size_t)b * a.D + d + 1) * HW + pix) * C + 4 * q);
                float4 S = a.warp_only ? make_float4(0.f, 0.f, 0.f, 0.f)
                                       : (a.ms_alias ? make_float4(r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w) : r);
                if (!a.warp_only)
                    for (int s = 0; s < NS; ++s) {
                        if (s == sA || s == sB) continue;
                        float ix, iy;
                        source_index(rotb + s * 9, trb + s * 3, xf, yf, dep, a, ix, iy);
                        float4 v = sample4(a.src[s] + fbase, make_taps(ix, iy, a.H, a.W), a.W, C);
                        S.x += v.x; S.y += v.y; S.z += v.z; S.w += v.w;
                    }
                float4 vA, vB = make_float4(0.f, 0.f, 0.f, 0.f);
                float wtA[4], wtB[4] = {0.f, 0.f, 0.f, 0.f};
                {
                    float ix, iy;
                    source_index(rotb + sA * 9, trb + sA * 3, xf, yf, dep, a, ix, iy);
                    vA = cA.template advance<C>(ix, iy, a.src[sA] + fbase, a.H, a.W, win, wA_, useA, gpA, q, wtA);
                    S.x += vA.x; S.y += vA.y; S.z += vA.z; S.w += vA.w;
                }
                if (sB >= 0) {
                    float ix, iy;
                    source_index(rotb + sB * 9, trb + sB * 3, xf, yf, dep, a, ix, iy);
                    vB = cB.template advance<C>(ix, iy, a.src[sB] + fbase, a.H, a.W, winB, wB_, useB, gpB, q, wtB);
                    S.x += vB.x; S.y += vB.y; S.z += vB.z; S.w += vB.w;
                }
                float4 gA, gB;
                if (a.warp_only) {
                    gA = g;
                    gB = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    const float4 Sm = make_float4(S.x * inv_n, S.y * inv_n, S.z * inv_n, S.w * inv_n);
                    if (vg == 0) {
                        if (a.ms_alias) {
                            gr.x += g.x * two_n * r.x * (1.0f - 2.0f * Sm.x); gr.y += g.y * two_n * r.y * (1.0f - 2.0f * Sm.y);
                            gr.z += g.z * two_n * r.z * (1.0f - 2.0f * Sm.z); gr.w += g.w * two_n * r.w * (1.0f - 2.0f * Sm.w);
                        } else {
                            gr.x += g.x * two_n * (r.x - Sm.x); gr.y += g.y * two_n * (r.y - Sm.y);
                            gr.z += g.z * two_n * (r.z - Sm.z); gr.w += g.w * two_n * (r.w - Sm.w);
                        }
                    }
                    gA = make_float4(g.x * two_n * (vA.x - Sm.x), g.y * two_n * (vA.y - Sm.y), g.z * two_n * (vA.z - Sm.z),
                                     g.w * two_n * (vA.w - Sm.w));
                    gB = make_float4(g.x * two_n * (vB.x - Sm.x), g.y * two_n * (vB.y - Sm.y), g.z * two_n * (vB.z - Sm.z),
                                     g.w * two_n * (vB.w - Sm.w));
                }
                cA.accumulate(gA, wtA);
                if (sB >= 0) cB.accumulate(gB, wtB);
            }
            cA.template flush<C>(a.H, a.W, win, wA_, useA, gpA, q);
            if (sB >= 0) cB.template flush<C>(a.H, a.W, winB, wB_, useB, gpB, q);
        }
        // ---- flush the windows ----
        __syncthreads();
        if (useA) {
            float* gp = a.gsrc[sA] + (size_t)b * HW * C;
            for (int i = tid; i < wA_.w * wA_.h * C; i += 256) {
                const int c = i % C, t = i / C;
                const float v = win[t * CP + c];
                if (v != 0.f) MVS_GLOBAL_ATOMIC_ADD(gp + ((size_t)(wA_.y0 + t / wA_.w) * a.W + wA_.x0 + t % wA_.w) * C + c, v);
            }
        }
        if (useB) {
            float* gp = a.gsrc[sB] + (size_t)b * HW * C;
            for (int i = tid; i < wB_.w * wB_.h * C; i += 256) {
                const int c = i % C, t = i / C;
                const float v = win[(WCAP + t) * CP + c];
                if (v != 0.f) MVS_GLOBAL_ATOMIC_ADD(gp + ((size_t)(wB_.y0 + t / wB_.w) * a.W + wB_.x0 + t % wB_.w) * C + c, v);
            }
        }
        ds = de;
    }
    if (valid && vg == 0 && !a.warp_only) {
        // one atomic per (pixel, channel, depth slab): grad_ref is zero-filled by the caller
        float* p = a.gref + fbase + (size_t)pix * C;
        MVS_GLOBAL_ATOMIC_ADD(p + 0, gr.x); MVS_GLOBAL_ATOMIC_ADD(p + 1, gr.y);
        MVS_GLOBAL_ATOMIC_ADD(p + 2, gr.z); MVS_GLOBAL_ATOMIC_ADD(p + 3, gr.w);
    }
}


// ------------------------------------------------------------------------------------------------
// Backward, per-wave windows (default for 1..4 source views).
//
// What bounded the kernel above (profiles/r01_run19_pmc_sq_summary.json, profiles/r02_run1_atomic_rate.log): an LDS fp32
// atomic costs ~3 LDS cycles PER ACTIVE LANE (ds_add_f32: 25 cycles with 8 lanes, 193 with 64, whatever the addresses),
// against ~9 cycles for a whole ds_read_b128 / ds_write_b128 and 5.5 for a dense 64-lane ds_read_b32 + ds_write_b32 pair;
// the flush of a 2x2 block was 16 atomics with a handful of lanes active each, i.e. the LDS array was busy 42 cycles
// per instruction, 52 % of the kernel.  A plain read-add-write needs exclusive ownership of the window, so here
//  * every WAVE owns its own LDS windows (one per source view: the footprint of the wave's small pixel block over the
//    depth segment), so there is no inter-wave race and no barrier inside the plane loop;
//  * the pixel groups of a wave that leave their block on the same plane flush ONE AFTER THE OTHER (a wave-uniform loop
//    over the ballot; the DS queue of a wave is in order), so there is no intra-instruction race either: a flush is
//    4 taps x (CPT/4) x {ds_read_b128, 4 adds, ds_write_b128}, no atomics;
//  * a thread carries CPT = 8 channels for <= 2 source views (4 lanes per pixel: the per-pixel projection arithmetic is
//    replicated 4x instead of 8x) and 4 channels for 3-4 views (register budget), ALL views in one workgroup: the
//    upstream gradient is read exactly once and no view is ever re-sampled through L1;
//  * at the end of a depth segment the four waves' windows are summed on the fly and written out with coalesced global
//    atomics (64 consecutive floats per wave instruction: 330 G/s against 79 G/s for the 4-float pattern of a
//    per-thread flush), grad_ref likewise goes through LDS so that a wave instruction covers whole 128-byte texels.
// ------------------------------------------------------------------------------------------------
// component c (a constant after unrolling) of a float4 held in registers
__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }
__device__ __forceinline__ float& f4r(float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

template <int C, int CPT> struct PwCfg {
    static constexpr int LPP = C / CPT;             // lanes per pixel
    static constexpr int PPW = 64 / LPP;            // pixels per wave, as a BW x BH block
#ifndef MVS_PW_BW
#define MVS_PW_BW 4            // width of a wave's pixel block at 8 pixels per wave (4 x 2); build-time A/B: -DMVS_PW_BW=2 (2 x 4), =8 (8 x 1)
#endif
    static constexpr int BW = PPW >= 32 ? 8 : MVS_PW_BW, BH = PPW / BW;
    static constexpr int V = CPT / 4;               // float4s per tap per thread
};

// One source view's register-resident 2x2 texel block of a thread: tap values + gradient accumulators (V float4 each).
template <int V> struct PwBlock {
    int cx, cy;
    float4 t00[V], t01[V], t10[V], t11[V];
    float4 g00[V], g01[V], g10[V], g11[V];
};

// The pixel groups (LPP consecutive lanes) whose `want` is set add their accumulators to the wave's window, one group
// per iteration of a wave-uniform loop.  win already includes the lane's channel offset (gbase / foff: see below).  Within an iteration all
// LDS reads of the block (4 taps x V float4) are issued before the first add: ONE LDS round trip per flush (taps handled
// one after the other cost eight dependent round trips, ~1000 cycles).  A tap outside the window (or all of them when
// the window is unusable) goes to global memory with atomics; a tap outside the image is dropped (zero padding).
template <int C, int V, int CK, int LPP>
__device__ __forceinline__ void pw_flush_groups(bool want, int lane, const PwBlock<V>& blk, int H, int W,
                                                float* __restrict__ win, const Win& w, bool use_win, float* __restrict__ gbase, unsigned foff) {
    // (gbase: the view's gradient map, wave-uniform; foff: the lane's batch + channel offset.  The per-lane 64-bit pointer gbase + foff
    //  is only formed on the rare global-atomic path: as an argument it lived in two VGPRs per view across the whole plane loop)
    unsigned long long m = MVS_BALLOT(want);
    while (m) {
        const int grp = (MVS_FFSLL(m) - 1) / LPP;
        if (lane / LPP == grp) {
            const int cx = blk.cx, cy = blk.cy;
            const int lx = cx - w.x0, ly = cy - w.y0;
            if (use_win && lx >= 0 && lx + 1 < w.w && ly >= 0 && ly + 1 < w.h && cx >= 0 && cx + 1 < W && cy >= 0 && cy + 1 < H) {
                // common case: the whole 2x2 block lies inside the image and the window -> no per-tap tests
                float* p0 = win + (ly * w.w + lx) * C;
                float* p1 = p0 + w.w * C;
                float4 a00[V], a01[V], a10[V], a11[V];
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    a00[k] = *reinterpret_cast<const float4*>(p0 + CK * k); a01[k] = *reinterpret_cast<const float4*>(p0 + C + CK * k);
                    a10[k] = *reinterpret_cast<const float4*>(p1 + CK * k); a11[k] = *reinterpret_cast<const float4*>(p1 + C + CK * k);
                }
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    a00[k].x += blk.g00[k].x; a00[k].y += blk.g00[k].y; a00[k].z += blk.g00[k].z; a00[k].w += blk.g00[k].w;
                    a01[k].x += blk.g01[k].x; a01[k].y += blk.g01[k].y; a01[k].z += blk.g01[k].z; a01[k].w += blk.g01[k].w;
                    a10[k].x += blk.g10[k].x; a10[k].y += blk.g10[k].y; a10[k].z += blk.g10[k].z; a10[k].w += blk.g10[k].w;
                    a11[k].x += blk.g11[k].x; a11[k].y += blk.g11[k].y; a11[k].z += blk.g11[k].z; a11[k].w += blk.g11[k].w;
                }
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    *reinterpret_cast<float4*>(p0 + CK * k) = a00[k]; *reinterpret_cast<float4*>(p0 + C + CK * k) = a01[k];
                    *reinterpret_cast<float4*>(p1 + CK * k) = a10[k]; *reinterpret_cast<float4*>(p1 + C + CK * k) = a11[k];
                }
            } else {
            const bool xin0 = cx >= 0 && cx < W, xin1 = cx + 1 >= 0 && cx + 1 < W;
            const bool yin0 = cy >= 0 && cy < H, yin1 = cy + 1 >= 0 && cy + 1 < H;
            const bool wx0 = lx >= 0 && lx < w.w, wx1 = lx + 1 >= 0 && lx + 1 < w.w;
            const bool wy0 = ly >= 0 && ly < w.h, wy1 = ly + 1 >= 0 && ly + 1 < w.h;
            const bool img[4] = {xin0 && yin0, xin1 && yin0, xin0 && yin1, xin1 && yin1};
            const bool inw[4] = {img[0] && use_win && wx0 && wy0, img[1] && use_win && wx1 && wy0,
                                 img[2] && use_win && wx0 && wy1, img[3] && use_win && wx1 && wy1};
            const int off[4] = {(ly * w.w + lx) * C, (ly * w.w + lx + 1) * C, ((ly + 1) * w.w + lx) * C, ((ly + 1) * w.w + lx + 1) * C};
            float4 acc[4][V];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* p = win + (inw[t] ? off[t] : 0);      // a tap that does not go to the window reads texel 0 (discarded)
#pragma unroll
                for (int k = 0; k < V; ++k) acc[t][k] = *reinterpret_cast<const float4*>(p + CK * k);
            }
#pragma unroll
            for (int k = 0; k < V; ++k) {
                acc[0][k].x += blk.g00[k].x; acc[0][k].y += blk.g00[k].y; acc[0][k].z += blk.g00[k].z; acc[0][k].w += blk.g00[k].w;
                acc[1][k].x += blk.g01[k].x; acc[1][k].y += blk.g01[k].y; acc[1][k].z += blk.g01[k].z; acc[1][k].w += blk.g01[k].w;
                acc[2][k].x += blk.g10[k].x; acc[2][k].y += blk.g10[k].y; acc[2][k].z += blk.g10[k].z; acc[2][k].w += blk.g10[k].w;
                acc[3][k].x += blk.g11[k].x; acc[3][k].y += blk.g11[k].y; acc[3][k].z += blk.g11[k].z; acc[3][k].w += blk.g11[k].w;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (inw[t]) {
#pragma unroll
                    for (int k = 0; k < V; ++k) *reinterpret_cast<float4*>(win + off[t] + CK * k) = acc[t][k];
                }
            if ((img[0] && !inw[0]) || (img[1] && !inw[1]) || (img[2] && !inw[2]) || (img[3] && !inw[3])) {
                // rare: footprint larger than the window allowance, or rounding at the hull of the projected box
                const float4* gsrc4[4] = {blk.g00, blk.g01, blk.g10, blk.g11};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (img[t] && !inw[t]) {
                        unsigned fo = foff;
                        MVS_OPAQUE_U(fo);
                        float* p = gbase + (size_t)fo + ((size_t)(cy + (t >> 1)) * W + cx + (t & 1)) * C;
#pragma unroll
                        for (int k = 0; k < V; ++k) {
                            MVS_GLOBAL_ATOMIC_ADD(p + CK * k + 0, gsrc4[t][k].x); MVS_GLOBAL_ATOMIC_ADD(p + CK * k + 1, gsrc4[t][k].y);
                            MVS_GLOBAL_ATOMIC_ADD(p + CK * k + 2, gsrc4[t][k].z); MVS_GLOBAL_ATOMIC_ADD(p + CK * k + 3, gsrc4[t][k].w);
                        }
                    }
            }
            }
        }
        m &= ~(((1ull << LPP) - 1ull) << (grp * LPP));
        MVS_WAVE_SYNC();   // next group may touch the same texels: keep the DS operations in program order
    }
}

// MODE: 0 variance (MVSNet), 1 variance with the jdacs-ms alias quirk (S starts from r^2), 2 plain homo_warping
// GD: 0 = the upstream gradient of the next plane in ONE rotating register set (fits 3 waves/SIMD), 2 = requested two planes ahead into
// three sets (2 waves/SIMD).  PPD: per-plane depth hypotheses only (scalar depth loads; the generic form decides at run time)
// PFL: block lookahead -- the 2x2 block a lane enters on the NEXT plane is requested one plane ahead into staging registers
template <int C, int NS_T, int CPT, int MODE, int GD, int WPS, bool PPD, bool PFL = false>
__global__ __launch_bounds__(256) MVS_WAVES_PER_SIMD(WPS) void plane_sweep_variance_bwd_pw_kernel(SweepArgs a) {
    constexpr bool WARP_ONLY = MODE == 2, MS_ALIAS = MODE == 1;
    using Cfg = PwCfg<C, CPT>;
    constexpr int LPP = Cfg::LPP, BW = Cfg::BW, BH = Cfg::BH, V = Cfg::V;
    constexpr int CK = 4 * LPP;                      // channel of float4 k of lane q: 4q + CK*k (as in the forward)
    // LDS per wave: 12.5 KiB when 3 workgroups share a CU (4 channels per thread, <= 2 views: 3 waves/SIMD by registers),
    // 19.5 KiB when registers allow only 2 waves/SIMD anyway (8 channels per thread, or 3-4 views: more room per view -> longer
    // depth segments before a window overflows)
    // (WPS = 1 -- knob "bwd_pf" = 2, 3-4 source views -- has the CU's LDS to itself: 37.5 KiB per wave, twice the window per view,
    //  i.e. depth segments twice as long before a window overflows and half the window write-outs)
    constexpr int WAVE_FLOATS = WPS >= 3 ? 3200 : (WPS == 1 ? 9600 : 4992);
    constexpr int VIEW_FLOATS = WAVE_FLOATS / NS_T / C * C, WCAP = VIEW_FLOATS / C;
    __shared__ __attribute__((aligned(16))) float lds[4 * NS_T * VIEW_FLOATS];   // [wave][view][texel][C]
    __shared__ int s_win[4][NS_T][5];                // per wave and view: x0, y0, w, h, usable
    __shared__ int s_fit[4];
    __shared__ float red[8];
    __shared__ float s_dep[4][64];                   // per wave: the per-plane depth hypotheses of 64 planes of the segment
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = lane % LPP, pl = lane / LPP;
    const SweepWg wg = sweep_wg(a);
    const int bx0 = (wg.tile % a.tiles_x) * (2 * BW) + (wv & 1) * BW, by0 = (wg.tile / a.tiles_x) * (2 * BH) + (wv >> 1) * BH;
    const int xr = bx0 + pl % BW, yr = by0 + pl / BW;
    const int b = wg.b;
    const bool live = xr < a.W && yr < a.H;          // lanes outside the image follow along (wave-wide exchanges) on a clamped pixel
    const int x = min(xr, a.W - 1), y = min(yr, a.H - 1);
    const int HW = a.H * a.W, pix = y * a.W + x;
    const float xf = (float)x, yf = (float)y;
    const int cq = 4 * q;
    const size_t fbase = (size_t)b * HW * C + cq;
    const unsigned fb32 = (unsigned)fbase;           // the host checks B * H * W * C < 2^31 for this kernel
    float4 r[V];
#pragma unroll
    for (int k = 0; k < V; ++k) r[k] = ld4(a.ref + fbase + (size_t)pix * C + CK * k);
    const float inv_n = 1.0f / (float)(NS_T + 1);
    const float two_n = live ? 2.0f * inv_n : 0.0f;  // dead lanes contribute exact zeros
    const float* __restrict__ rotb = a.rot + (size_t)b * NS_T * 9;
    const float* __restrict__ trb = a.trans + (size_t)b * NS_T * 3;
    // REMAT (3-4 source views at 2 waves per SIMD, round 6): the per-lane ray coefficients rx / ry / rz (12 VGPRs at 4 views) are NOT kept
    // across the plane loop but recomputed per plane from the view's rotation, read through the scalar cache -- the same two fused
    // multiply-adds, bit for bit.  Kept in VGPRs they were spilled, and the reload of every view's pair sat behind an `s_waitcnt
    // vmcnt(0)` at the top of its `locate`: four scratch round trips per plane, each also draining the upstream-gradient prefetch
    // (N = 5: 62 % of wave time parked, 1.09 ms; profiles/r06_run3_bench_c3.json).
    constexpr bool REMAT = NS_T >= 4 && WPS >= 2 && GD == 2;
    float rx[REMAT ? 1 : NS_T], ry[REMAT ? 1 : NS_T], rz[REMAT ? 1 : NS_T], tx[NS_T], ty[NS_T], tz[NS_T];
#pragma unroll
    for (int s = 0; s < NS_T; ++s) {
        const float* R = rotb + s * 9;
        if constexpr (!REMAT) {
            rx[s] = fmaf(R[0], xf, fmaf(R[1], yf, R[2]));
            ry[s] = fmaf(R[3], xf, fmaf(R[4], yf, R[5]));
            rz[s] = fmaf(R[6], xf, fmaf(R[7], yf, R[8]));
        }
        tx[s] = trb[s * 3]; ty[s] = trb[s * 3 + 1]; tz[s] = trb[s * 3 + 2];
    }
    float4 gr[V];
#pragma unroll
    for (int k = 0; k < V; ++k) gr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // corners of the wave's pixel block (clipped to the image) for its footprint bound
    const float cxa = (float)min(bx0, a.W - 1), cxb = (float)min(bx0 + BW - 1, a.W - 1);
    const float cya = (float)min(by0, a.H - 1), cyb = (float)min(by0 + BH - 1, a.H - 1);
    float* const wwin = lds + (size_t)wv * NS_T * VIEW_FLOATS;     // this wave's windows
#define z4 (make_float4(0.f, 0.f, 0.f, 0.f))   /* a literal: a const object captured by the lambdas below lives in scratch */

    int ds = wg.slab * a.dslab;
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
                if (q == 0)
                    for (int d = ds; d < de; ++d) {
                        float v = a.depth[((size_t)b * a.D + d) * HW + pix];
                        lo = fminf(lo, v); hi = fmaxf(hi, v);
                    }
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { lo = fminf(lo, __shfl_xor(lo, m)); hi = fmaxf(hi, __shfl_xor(hi, m)); }
                da = lo; db = hi;                    // depth range of THIS wave's pixels: its windows only have to hold them
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
                if constexpr (NS_T >= 3 || WPS >= 3) {   // (1-2 views at 2 waves/SIMD have vector registers to spare and no scalar ones: there the geometry stays where it was computed)
                    w[s].x0 = MVS_UNIFORM_I(w[s].x0); w[s].y0 = MVS_UNIFORM_I(w[s].y0);
                    w[s].w = MVS_UNIFORM_I(w[s].w); w[s].h = MVS_UNIFORM_I(w[s].h);
                }
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
                for (int i = lane * 4; i < w[s].w * w[s].h * C; i += 256) *reinterpret_cast<float4*>(ws + i) = z4;
            }
            if (lane == 0) {
                s_win[wv][s][0] = w[s].x0; s_win[wv][s][1] = w[s].y0; s_win[wv][s][2] = w[s].w; s_win[wv][s][3] = w[s].h;
                s_win[wv][s][4] = use[s] ? 1 : 0;
            }
        }
        MVS_WAVE_SYNC();
        // ---- walk the planes of the segment (no workgroup barrier in here) ----
        {
            PwBlock<V> blk[NS_T];
#pragma unroll
            for (int s = 0; s < NS_T; ++s) {
                blk[s].cx = blk[s].cy = -0x40000000;
#pragma unroll
                for (int k = 0; k < V; ++k)
                    blk[s].t00[k] = blk[s].t01[k] = blk[s].t10[k] = blk[s].t11[k] = blk[s].g00[k] = blk[s].g01[k] = blk[s].g10[k] =
                        blk[s].g11[k] = z4;
            }
            // where view s samples at depth `dep`: base texel + fractions (the forward kernel's arithmetic)
            auto locate = [&](int s, float dep, int& x0, int& y0, float& wx, float& wy) {
                float rxs, rys, rzs;
                if constexpr (REMAT) {
                    const float* R = rotb + s * 9;
                    float xo = xf, yo = yf;          // "produced here": without this the six multiply-adds are hoisted back out of the loop
                    MVS_OPAQUE_U(xo); MVS_OPAQUE_U(yo);
                    rxs = fmaf(MVS_SCALAR_LD(R, 0), xo, fmaf(MVS_SCALAR_LD(R, 1), yo, MVS_SCALAR_LD(R, 2)));
                    rys = fmaf(MVS_SCALAR_LD(R, 3), xo, fmaf(MVS_SCALAR_LD(R, 4), yo, MVS_SCALAR_LD(R, 5)));
                    rzs = fmaf(MVS_SCALAR_LD(R, 6), xo, fmaf(MVS_SCALAR_LD(R, 7), yo, MVS_SCALAR_LD(R, 8)));
                } else {
                    rxs = rx[s]; rys = ry[s]; rzs = rz[s];
                }
                const float zz = fmaf(rzs, dep, tz[s]);
                float iz = MVS_RCP(zz);
                iz = fmaf(fmaf(-zz, iz, 1.0f), iz, iz);
                const float ix = fmaf(fmaf(rxs, dep, tx[s]) * iz, a.sx, a.ox);
                const float iy = fmaf(fmaf(rys, dep, ty[s]) * iz, a.sy, a.oy);
                const float fx = floorf(ix), fy = floorf(iy);
                wx = ix - fx; wy = iy - fy;
                x0 = MVS_F2I(fx); y0 = MVS_F2I(fy);
            };
            // the 2x2 block with base texel (x0, y0) of view s: zero where a tap is outside the image
            auto gather = [&](int s, int x0, int y0, float4 (&o00)[V], float4 (&o01)[V], float4 (&o10)[V], float4 (&o11)[V]) {
                // (wave-uniform base + a 32-bit lane offset: the loads take the SGPR-base addressing form and no 64-bit per-lane pointer
                //  per view stays alive across the plane loop -- at 4 views those were spilled and reloaded in front of every gather)
                const float* __restrict__ sb = a.src[s];
                const int o = (int)fb32 + (y0 * a.W + x0) * C;
                if (x0 >= 0 && x0 + 1 < a.W && y0 >= 0 && y0 + 1 < a.H) {   // common case: all four taps inside the image
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        o00[k] = ld4(sb + (unsigned)(o + CK * k)); o01[k] = ld4(sb + (unsigned)(o + C + CK * k));
                        o10[k] = ld4(sb + (unsigned)(o + a.W * C + CK * k)); o11[k] = ld4(sb + (unsigned)(o + a.W * C + C + CK * k));
                    }
                    return;
                }
                const float* __restrict__ f = sb + (long)o;
                const bool xin0 = x0 >= 0 && x0 < a.W, xin1 = x0 + 1 >= 0 && x0 + 1 < a.W;
                const bool yin0 = y0 >= 0 && y0 < a.H, yin1 = y0 + 1 >= 0 && y0 + 1 < a.H;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    o00[k] = (xin0 && yin0) ? ld4(f + CK * k) : z4;
                    o01[k] = (xin1 && yin0) ? ld4(f + C + CK * k) : z4;
                    o10[k] = (xin0 && yin1) ? ld4(f + a.W * C + CK * k) : z4;
                    o11[k] = (xin1 && yin1) ? ld4(f + a.W * C + C + CK * k) : z4;
                }
            };
            const float* __restrict__ gptr = a.gvar + (((size_t)b * a.D + ds) * HW + pix) * C + cq;
            const size_t gstep = (size_t)HW * C;
            // upstream gradient: the planes of the NEXT group (GD = 0: one plane, 3 waves/SIMD; GD = 2: two planes, 2 waves/SIMD) are
            // requested at the TOP of the current group and taken over at its end, so their HBM round trip overlaps a whole group
            // of arithmetic.  (Requested in the middle of a plane and copied at its end they had ~65 instructions of cover: most of
            // a round trip exposed per plane.  Rotating three register sets through an unrolled loop does not work with hipcc: the
            // loop-carried sets are copied anyway and every copy waits for the request just issued.)
            constexpr int G = GD == 2 ? 2 : 1;
            float4 gc[G][V], gn[G][V];
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < V; ++k) gc[j][k] = ld4(gptr + (size_t)min(j, de - 1 - ds) * gstep + CK * k);
            // depth of plane d.  Per-plane hypotheses (one value per sample and plane) are staged 64 planes at a time in the wave's
            // own LDS row and read back one plane ahead: a global load here (hipcc emits a VECTOR load, the kernel also stores) sits
            // in the same in-order queue as the upstream-gradient requests, and waiting for it drains them.  Per-pixel hypotheses:
            // a vector load per plane.
            // (the wave index as a SCALAR: with the row address derived from threadIdx the compiler kept it in a vector register, spilled
            //  it at 4 views and reloaded it from scratch in front of every plane's read -- behind an s_waitcnt vmcnt(0))
            const int wvu = MVS_UNIFORM_I(wv);
            auto depth_of = [&](int d) __attribute__((always_inline)) -> float {
                if constexpr (!PPD) {
                    if (a.per_pixel) return a.depth[((size_t)b * a.D + d) * HW + pix];
                }
                const int i = d - ds;
                if ((i & 63) == 0) {                 // wave-uniform; a wave's DS operations execute in order
                    MVS_WAVE_SYNC();
                    if (d + lane < de) s_dep[wvu][lane] = a.depth[b * a.D + d + lane];
                    MVS_WAVE_SYNC();
                }
                return s_dep[wvu][i & 63];
            };
            float dep_next = depth_of(ds);
            // PFL: sample position of the plane about to be processed + the staged block of the lanes that enter a new one there
            int nx[PFL ? NS_T : 1], ny[PFL ? NS_T : 1];
            float nwx[PFL ? NS_T : 1], nwy[PFL ? NS_T : 1];
            float4 s00[PFL ? NS_T : 1][V], s01[PFL ? NS_T : 1][V], s10[PFL ? NS_T : 1][V], s11[PFL ? NS_T : 1][V];
            if constexpr (PFL) {
#pragma unroll
                for (int s = 0; s < NS_T; ++s) {
                    locate(s, dep_next, nx[s], ny[s], nwx[s], nwy[s]);
                    gather(s, nx[s], ny[s], s00[s], s01[s], s10[s], s11[s]);
                }
                if (ds + 1 < de) dep_next = depth_of(ds + 1);
            }
            // one plane: gu = the plane's upstream gradient
            // the upstream gradient of the NEXT group of planes.  Where it is requested matters more than how far ahead: vector loads return
            // IN ORDER, so the wait for a re-gathered block (most planes have one) also waits for every request in front of it.  Requested
            // at the top of the group (knob "bwd_gpf" = 0, the default) the prefetch sits in front of this plane's gathers; requested AFTER
            // the gathers have been waited for (1, round 6) it has the arithmetic half of the plane as cover before the next plane's gathers
            // queue up behind it.  Measured (profiles/r06_run6_*): the late form is SLOWER -- config-2 step 4.811 -> 4.843 ms, config 3
            // 5.928 -> 5.988 -- so the exposed round trip is not the prefetch's.  Kept as a knob.
            auto request_next_group = [&](const int d) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const float* __restrict__ gnx = gptr + (size_t)(min(d + G + j, de - 1) - ds) * gstep;   // clamped: always a valid plane
#pragma unroll
                    for (int k = 0; k < V; ++k) gn[j][k] = ld4(gnx + CK * k);
                }
            };
            auto plane = [&](const int d, const float4 (&gu)[V], auto first) __attribute__((always_inline)) {
                float fwx[NS_T], fwy[NS_T];
                if constexpr (PFL) {
                    // (a) lanes whose sample point left their block: flush the accumulators, take over the staged block (requested
                    // a whole plane ago: the wait is short)
                    bool chg[NS_T], any = false;
#pragma unroll
                    for (int s = 0; s < NS_T; ++s) {
                        chg[s] = nx[s] != blk[s].cx || ny[s] != blk[s].cy;
                        any = any || chg[s];
                        fwx[s] = nwx[s]; fwy[s] = nwy[s];
                    }
                    if (MVS_ANY(any)) {
#pragma unroll
                        for (int s = 0; s < NS_T; ++s)
                            if (MVS_ANY(chg[s]))
                                pw_flush_groups<C, V, CK, LPP>(chg[s] && live && blk[s].cx != -0x40000000, lane, blk[s], a.H, a.W,
                                                               wwin + s * VIEW_FLOATS + cq, w[s], use[s], a.gsrc[s], fb32);
#pragma unroll
                        for (int s = 0; s < NS_T; ++s) {
#pragma unroll
                            for (int k = 0; k < V; ++k) { MVS_PIN4(s00[s][k]); MVS_PIN4(s01[s][k]); MVS_PIN4(s10[s][k]); MVS_PIN4(s11[s][k]); }
                            if (chg[s]) {
                                blk[s].cx = nx[s]; blk[s].cy = ny[s];
#pragma unroll
                                for (int k = 0; k < V; ++k) {
                                    blk[s].t00[k] = s00[s][k]; blk[s].t01[k] = s01[s][k]; blk[s].t10[k] = s10[s][k]; blk[s].t11[k] = s11[s][k];
                                    blk[s].g00[k] = blk[s].g01[k] = blk[s].g10[k] = blk[s].g11[k] = z4;
                                }
                            }
                        }
                    }
                    // (b) one plane ahead: request the blocks that will be entered there
                    if (d + 1 < de) {
                        const float dep1 = dep_next;
                        if (d + 2 < de) dep_next = depth_of(d + 2);
#pragma unroll
                        for (int s = 0; s < NS_T; ++s) {
                            locate(s, dep1, nx[s], ny[s], nwx[s], nwy[s]);
                            if (nx[s] != blk[s].cx || ny[s] != blk[s].cy) gather(s, nx[s], ny[s], s00[s], s01[s], s10[s], s11[s]);
                        }
                    }
                } else {
                    // the plane's depth was requested one plane ahead: a load issued here is consumed by the very next instruction,
                    // i.e. every plane would start with a full memory round trip
                    const float dep = dep_next;
                    if (d + 1 < de) dep_next = depth_of(d + 1);
                    if constexpr (NS_T >= 3) {
                        // 3-4 views: ALL views' new blocks are requested first, then the flushes (they need the accumulators and the
                        // OLD base texel, not the tap values), then ONE wait (N = 5: 1.01 -> 0.95 ms; for 1-2 views a round per
                        // view measured faster: 0.322 vs 0.333 ms)
                        int x0[NS_T], y0[NS_T];
                        bool chg[NS_T], any = false;
#pragma unroll
                        for (int s = 0; s < NS_T; ++s) {
                            locate(s, dep, x0[s], y0[s], fwx[s], fwy[s]);
                            chg[s] = x0[s] != blk[s].cx || y0[s] != blk[s].cy;
                            any = any || chg[s];
                        }
                        if (MVS_ANY(any)) {
#pragma unroll
                            for (int s = 0; s < NS_T; ++s)
                                if (chg[s]) gather(s, x0[s], y0[s], blk[s].t00, blk[s].t01, blk[s].t10, blk[s].t11);
#pragma unroll
                            for (int s = 0; s < NS_T; ++s)
                                if (MVS_ANY(chg[s]))
                                    pw_flush_groups<C, V, CK, LPP>(chg[s] && live && blk[s].cx != -0x40000000, lane, blk[s], a.H, a.W,
                                                                   wwin + s * VIEW_FLOATS + cq, w[s], use[s], a.gsrc[s], fb32);
#pragma unroll
                            for (int s = 0; s < NS_T; ++s) {
                                if (chg[s]) {
                                    blk[s].cx = x0[s]; blk[s].cy = y0[s];
#pragma unroll
                                    for (int k = 0; k < V; ++k) blk[s].g00[k] = blk[s].g01[k] = blk[s].g10[k] = blk[s].g11[k] = z4;
                                }
#pragma unroll
                                for (int k = 0; k < V; ++k) {
                                    MVS_PIN4(blk[s].t00[k]); MVS_PIN4(blk[s].t01[k]); MVS_PIN4(blk[s].t10[k]); MVS_PIN4(blk[s].t11[k]);
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < NS_T; ++s) {
                            int x0, y0;
                            locate(s, dep, x0, y0, fwx[s], fwy[s]);
                            const bool chg = x0 != blk[s].cx || y0 != blk[s].cy;
                            if (MVS_ANY(chg)) {
                                // request the new block first (the flush below needs the accumulators and the OLD base texel, not
                                // the tap values): the L2 round trip of the gather overlaps the LDS round trips of the flush
                                if (chg) gather(s, x0, y0, blk[s].t00, blk[s].t01, blk[s].t10, blk[s].t11);
                                pw_flush_groups<C, V, CK, LPP>(chg && live && blk[s].cx != -0x40000000, lane, blk[s], a.H, a.W,
                                                               wwin + s * VIEW_FLOATS + cq, w[s], use[s], a.gsrc[s], fb32);
                                if (chg) {
                                    blk[s].cx = x0; blk[s].cy = y0;
#pragma unroll
                                    for (int k = 0; k < V; ++k) blk[s].g00[k] = blk[s].g01[k] = blk[s].g10[k] = blk[s].g11[k] = z4;
                                }
                                // the gathered taps are waited for HERE, on the planes that re-gather: left to the join below, the
                                // wait would be a vmcnt(0) on every plane and would also drain the upstream-gradient requests in flight
#pragma unroll
                                for (int k = 0; k < V; ++k) {
                                    MVS_PIN4(blk[s].t00[k]); MVS_PIN4(blk[s].t01[k]); MVS_PIN4(blk[s].t10[k]); MVS_PIN4(blk[s].t11[k]);
                                }
                            }
                        }
                    }
                }
                MVS_SCHED_FENCE();
                if constexpr (decltype(first)::value) {
                    if (a.gpf_late) {
                        request_next_group(d);
                        MVS_SCHED_FENCE();
                    }
                }
                // phase 2, one float4 of channels at a time (keeps the live temporaries to one chunk): bilinear samples of all
                // views, their mean, then the gradients of the samples into the register accumulators.  The upstream gradient of
                // the NEXT plane is requested as soon as this plane's chunk has been consumed (no second buffer).
                float wt[NS_T][4];
#pragma unroll
                for (int s = 0; s < NS_T; ++s) {
                    const float wx = fwx[s], wy = fwy[s];
                    const float ex = 1.0f - wx, ey = 1.0f - wy;
                    wt[s][0] = ey * ex; wt[s][1] = ey * wx; wt[s][2] = wy * ex; wt[s][3] = wy * wx;
                }
                if constexpr (NS_T >= 3) {
                    static_assert(NS_T < 3 || !WARP_ONLY, "plain homo_warping has one source view");
                    // 3-4 views: the same arithmetic TWO channels at a time (round 6) -- per component nothing changes (same operations in the
                    // same order: bit-identical), but only 2 x NS_T sample values are alive between the sampling and the gradient half
                    // instead of 4 x NS_T: the 8 registers that decide whether the plane loop of the 4-view kernel spills
#pragma unroll
                    for (int k = 0; k < V; ++k) {
#pragma unroll
                        for (int h0 = 0; h0 < 4; h0 += 2) {
                            float S[2], v[NS_T][2], gsv[2], Smv[2];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float rc = f4c(r[k], h0 + j);
                                S[j] = MS_ALIAS ? rc * rc : rc;
                            }
#pragma unroll
                            for (int s = 0; s < NS_T; ++s) {
                                const PwBlock<V>& B = blk[s];
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const int c = h0 + j;
                                    v[s][j] = fmaf(f4c(B.t11[k], c), wt[s][3], fmaf(f4c(B.t10[k], c), wt[s][2], fmaf(f4c(B.t01[k], c), wt[s][1], f4c(B.t00[k], c) * wt[s][0])));
                                    S[j] += v[s][j];
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int c = h0 + j;
                                const float rc = f4c(r[k], c);
                                gsv[j] = f4c(gu[k], c) * two_n;          // g * 2/N (0 on dead lanes)
                                Smv[j] = S[j] * inv_n;
                                if (MS_ALIAS) f4r(gr[k], c) += gsv[j] * rc * (1.0f - 2.0f * Smv[j]);
                                else f4r(gr[k], c) += gsv[j] * (rc - Smv[j]);
                            }
#pragma unroll
                            for (int s = 0; s < NS_T; ++s) {
                                PwBlock<V>& B = blk[s];
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const int c = h0 + j;
                                    const float gv = gsv[j] * (v[s][j] - Smv[j]);
                                    f4r(B.g00[k], c) = fmaf(gv, wt[s][0], f4c(B.g00[k], c));
                                    f4r(B.g01[k], c) = fmaf(gv, wt[s][1], f4c(B.g01[k], c));
                                    f4r(B.g10[k], c) = fmaf(gv, wt[s][2], f4c(B.g10[k], c));
                                    f4r(B.g11[k], c) = fmaf(gv, wt[s][3], f4c(B.g11[k], c));
                                }
                            }
                            MVS_SCHED_FENCE();
                        }
                    }
                } else {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float4 S = WARP_ONLY ? z4 : (MS_ALIAS ? make_float4(r[k].x * r[k].x, r[k].y * r[k].y, r[k].z * r[k].z, r[k].w * r[k].w) : r[k]);
                    float4 v[NS_T];
                    if (!WARP_ONLY) {
#pragma unroll
                        for (int s = 0; s < NS_T; ++s) {
                            const PwBlock<V>& B = blk[s];
                            v[s].x = fmaf(B.t11[k].x, wt[s][3], fmaf(B.t10[k].x, wt[s][2], fmaf(B.t01[k].x, wt[s][1], B.t00[k].x * wt[s][0])));
                            v[s].y = fmaf(B.t11[k].y, wt[s][3], fmaf(B.t10[k].y, wt[s][2], fmaf(B.t01[k].y, wt[s][1], B.t00[k].y * wt[s][0])));
                            v[s].z = fmaf(B.t11[k].z, wt[s][3], fmaf(B.t10[k].z, wt[s][2], fmaf(B.t01[k].z, wt[s][1], B.t00[k].z * wt[s][0])));
                            v[s].w = fmaf(B.t11[k].w, wt[s][3], fmaf(B.t10[k].w, wt[s][2], fmaf(B.t01[k].w, wt[s][1], B.t00[k].w * wt[s][0])));
                            S.x += v[s].x; S.y += v[s].y; S.z += v[s].z; S.w += v[s].w;
                        }
                    }
                    const float4 g = gu[k];
                    float4 gs = make_float4(g.x * two_n, g.y * two_n, g.z * two_n, g.w * two_n);   // g * 2/N (0 on dead lanes)
                    float4 Sm = make_float4(S.x * inv_n, S.y * inv_n, S.z * inv_n, S.w * inv_n);
                    if (WARP_ONLY) {
                        gs = live ? g : z4;          // plain homo_warping: the warped sample itself gets the gradient
                    } else if (MS_ALIAS) {
                        gr[k].x += gs.x * r[k].x * (1.0f - 2.0f * Sm.x); gr[k].y += gs.y * r[k].y * (1.0f - 2.0f * Sm.y);
                        gr[k].z += gs.z * r[k].z * (1.0f - 2.0f * Sm.z); gr[k].w += gs.w * r[k].w * (1.0f - 2.0f * Sm.w);
                    } else {
                        gr[k].x += gs.x * (r[k].x - Sm.x); gr[k].y += gs.y * (r[k].y - Sm.y);
                        gr[k].z += gs.z * (r[k].z - Sm.z); gr[k].w += gs.w * (r[k].w - Sm.w);
                    }
#pragma unroll
                    for (int s = 0; s < NS_T; ++s) {
                        float4 gv;
                        if (WARP_ONLY) gv = gs;
                        else gv = make_float4(gs.x * (v[s].x - Sm.x), gs.y * (v[s].y - Sm.y), gs.z * (v[s].z - Sm.z), gs.w * (v[s].w - Sm.w));
                        PwBlock<V>& B = blk[s];
                        B.g00[k].x = fmaf(gv.x, wt[s][0], B.g00[k].x); B.g00[k].y = fmaf(gv.y, wt[s][0], B.g00[k].y);
                        B.g00[k].z = fmaf(gv.z, wt[s][0], B.g00[k].z); B.g00[k].w = fmaf(gv.w, wt[s][0], B.g00[k].w);
                        B.g01[k].x = fmaf(gv.x, wt[s][1], B.g01[k].x); B.g01[k].y = fmaf(gv.y, wt[s][1], B.g01[k].y);
                        B.g01[k].z = fmaf(gv.z, wt[s][1], B.g01[k].z); B.g01[k].w = fmaf(gv.w, wt[s][1], B.g01[k].w);
                        B.g10[k].x = fmaf(gv.x, wt[s][2], B.g10[k].x); B.g10[k].y = fmaf(gv.y, wt[s][2], B.g10[k].y);
                        B.g10[k].z = fmaf(gv.z, wt[s][2], B.g10[k].z); B.g10[k].w = fmaf(gv.w, wt[s][2], B.g10[k].w);
                        B.g11[k].x = fmaf(gv.x, wt[s][3], B.g11[k].x); B.g11[k].y = fmaf(gv.y, wt[s][3], B.g11[k].y);
                        B.g11[k].z = fmaf(gv.z, wt[s][3], B.g11[k].z); B.g11[k].w = fmaf(gv.w, wt[s][3], B.g11[k].w);
                    }
                    MVS_SCHED_FENCE();
                }
                }
            };
#pragma clang loop unroll(disable)
            for (int d = ds; d < de; d += G) {
                if (!a.gpf_late) request_next_group(d);
                MVS_SCHED_FENCE();
                plane(d, gc[0], std::true_type());
                if constexpr (G == 2) {
                    if (d + 1 < de) plane(d + 1, gc[1], std::false_type());
                }
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int k = 0; k < V; ++k) gc[j][k] = gn[j][k];
            }
            // the blocks still held in registers
#pragma unroll
            for (int s = 0; s < NS_T; ++s)
                pw_flush_groups<C, V, CK, LPP>(live && blk[s].cx != -0x40000000, lane, blk[s], a.H, a.W, wwin + s * VIEW_FLOATS + cq,
                                               w[s], use[s], a.gsrc[s], fb32);
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
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int lx = txl - jx0[j], ly = tyl - jy0[j];
                    if (lx >= 0 && lx < jw[j] && ly >= 0 && ly < jh[j])
                        acc += lds[(j * NS_T + s) * VIEW_FLOATS + (ly * jw[j] + lx) * C + c];
                }
                if (acc != 0.f) MVS_GLOBAL_ATOMIC_ADD(gp + ((size_t)tyl * a.W + txl) * C + c, acc);
            }
        }
        __syncthreads();
        ds = de;
    }
    if (!WARP_ONLY) {
        // grad_ref: one atomic per (pixel, channel, depth slab); through LDS so that a wave instruction covers whole texels
        float* stage = wwin;                          // PPW x C floats of this wave's (now idle) window space
#pragma unroll
        for (int k = 0; k < V; ++k) *reinterpret_cast<float4*>(stage + pl * C + cq + CK * k) = gr[k];
        MVS_WAVE_SYNC();
        constexpr int PPW = Cfg::PPW;
        for (int i = lane; i < PPW * C; i += 64) {
            const int p = i / C, c = i % C;
            const int px = bx0 + p % BW, py = by0 + p / BW;
            const float val = stage[i];
            if (px < a.W && py < a.H && val != 0.f) MVS_GLOBAL_ATOMIC_ADD(a.gref + ((size_t)b * HW + (size_t)py * a.W + px) * C + c, val);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MVS_SWEEP_FWD=direct selects the tap-through-L1 kernel (kept for A/B measurements); default: LDS-staged
// forward variants: 0 taps through L1 every plane, 1 LDS-staged windows, 2 register-cached taps (4 ch/thread),
// 3 register-cached taps (8 ch/thread).  Default 3; MVS_SWEEP_FWD=<n> or mvs_set_tuning("sweep_fwd", n) for A/B.
static int g_sweep_fwd_variant = -1;
static int sweep_fwd_variant() {
    if (g_sweep_fwd_variant < 0) {
        const char* e = getenv("MVS_SWEEP_FWD");
        g_sweep_fwd_variant = (e && e[0] >= '0' && e[0] <= '4') ? e[0] - '0' : 3;
    }
    return g_sweep_fwd_variant;
}
static int g_sweep_nt = 0;
static int g_sweep_tile_w = 0;   // knob "tile_w": 0 = default square-ish tile
static int g_sweep_dslab = 0;    // knob "dslab": planes per workgroup of the forward kernels, 0 = auto
extern int g_conv_split;
extern int g_conv_small;
extern int g_conv_small_wgs;
extern int g_conv_tr2pw;
extern int g_conv_cc_wide;
extern int g_conv_cin1_vpt;
extern int g_conv_c8;
extern int g_conv_xcd;
extern int g_conv_x3;
extern int g_conv_side_pre;
extern int g_conv_pers, g_conv_pers_min_wgs, g_conv_pers_groups, g_conv_pers_nw, g_conv_wgrad_pers;
extern int g_conv_wgrad_small;
extern int g_conv_wgrad_groups;
extern int g_conv_wgrad8_groups;
extern int g_conv_wgrad8_gs;
extern int g_conv_wgrad8_nch;
extern int g_conv_cout1_d4;
extern int g_conv_bf16_dp;
extern int g_conv2d_s2_mfma;
extern int g_conv2d_pp;
extern int g_conv2d_wgrad_groups;
extern int g_conv2d_wgrad_batch_groups;
extern int g_conv_cout1_h4;
static int g_sweep_bwd_variant = 0;   // knob "sweep_bwd": 0 = per-wave windows (<= 4 source views; the default), 1 = view-pair kernel with LDS atomics (what > 4 source views run).  (Round 3's projection-table form with an LDS-DMA ring measured 2.2x slower and was removed in round 4: DESIGN.md section 4, git tag r3-rejected-variants.)
static int g_sweep_bwd_cpt = 4;       // knob "bwd_cpt": accepted and ignored (the 8-channels-per-thread form was measured slower and removed)
static int g_sweep_bwd_pf = 0;        // knob "bwd_pf": 1 = block lookahead for 1-2 source views, 2 = ONE wave per SIMD for 3-4 source views
static int g_sweep_xcd = 0;           // knob "sweep_xcd": XCD-compact workgroup order of the cached forward and the per-wave-window backward
static int g_sweep_fwd_dl = 2;        // knob "fwd_dl": forward with LDS-staged per-plane depths (1), + all views' re-gathers in flight before the first sample (2, default since round 6: with the 4-wide tile -5.6 % at N = 3, -4.4 % at N = 5); 0: the round-1 loop
static int g_sweep_bwd_gd = 2;        // knob "bwd_gd": 2 = upstream gradient requested two planes ahead at 2 waves/SIMD (1-2 source views), 0 = rotating set at 3 waves/SIMD
static int g_sweep_fwd_pt = 0;        // knob "fwd_pt": the cached forward with the per-wave projection table (plane_sweep_variance_fwd_pt_kernel).  MEASURED AND REJECTED (round 6, profiles/r06_run15_*, r06_run16_*): 9-26 % fewer vector instructions per plane, bit-identical, and SLOWER at every view count -- N=3 0.108 -> 0.117 ms, N=5 0.208 -> 0.226, N=7 bf16 2.56 -> 2.95
static int g_sweep_bwd_gd34 = 0;      // knob "bwd_gd34": 3-4 source views with the upstream gradient requested two planes ahead (as 1-2 views run), 2 waves/SIMD
static int g_sweep_bwd_gpf = 0;       // knob "bwd_gpf": where the per-wave-window backward requests the next planes' upstream gradient: 0 top of the group, 1 after the plane's gathers
int g_sweep_bwd_nowin = 0;            // knob "bwd_nowin" (tests): 1 = no LDS windows, every flush through global atomics
int g_sweep_bwd_dslab = 0;            // knob "bwd_dslab": planes per workgroup of the per-wave-window backward, 0 = auto
// Measurement knobs (A/B runs of tools/bench_kernels.py and the tests).  Full-string keys: an unknown or misspelt key
// is an error, never a silent hit on another knob.  Process-wide; not part of the data path's contract.
struct MvsKnob { const char* name; int* var; int lo, hi; };
static const MvsKnob* mvs_find_knob(const char* key) {
    static const MvsKnob knobs[] = {
        {"nt", &g_sweep_nt, 0, 1},           {"tile_w", &g_sweep_tile_w, 0, 256},   {"dslab", &g_sweep_dslab, 0, 1 << 20},
        {"conv_split", &g_conv_split, 0, 1}, {"conv_small", &g_conv_small, 0, 2}, {"conv_small_wgs", &g_conv_small_wgs, 0, 1 << 20}, {"tr2pw", &g_conv_tr2pw, 0, 1}, {"cc_wide", &g_conv_cc_wide, 0, 1}, {"cin1_vpt", &g_conv_cin1_vpt, 1, 5}, {"k8", &g_conv_c8, 0, 15},             {"cout1_d4", &g_conv_cout1_d4, 0, 3}, {"bf16_dp", &g_conv_bf16_dp, 0, 1}, {"conv2d_pp", &g_conv2d_pp, 0, 1},
        {"wgrad2d_groups", &g_conv2d_wgrad_groups, 0, 1 << 20}, {"wgrad2d_batch", &g_conv2d_wgrad_batch_groups, 1, 4096},                     {"conv2d_s2_mfma", &g_conv2d_s2_mfma, 0, 2},
        {"xcd", &g_conv_xcd, 0, 1}, {"side_pre", &g_conv_side_pre, 0, 1}, {"conv_pers", &g_conv_pers, 0, 1}, {"conv_pers_min", &g_conv_pers_min_wgs, 0, 1 << 30}, {"conv_pers_groups", &g_conv_pers_groups, 0, 4096}, {"conv_pers_nw", &g_conv_pers_nw, 4, 8}, {"wgrad_pers", &g_conv_wgrad_pers, 0, 1}, {"wgrad_small", &g_conv_wgrad_small, 0, 3}, {"wgrad_groups", &g_conv_wgrad_groups, 1, 768}, {"wgrad8_groups", &g_conv_wgrad8_groups, 1, 512}, {"wgrad8_gs", &g_conv_wgrad8_gs, 0, 2}, {"wgrad8_nch", &g_conv_wgrad8_nch, 1, 2}, {"cout1_h4", &g_conv_cout1_h4, 0, 1},          {"sweep_fwd", &g_sweep_fwd_variant, 0, 4}, {"sweep_bwd", &g_sweep_bwd_variant, 0, 1},
        {"bwd_dslab", &g_sweep_bwd_dslab, 0, 1 << 20}, {"bwd_nowin", &g_sweep_bwd_nowin, 0, 1}, {"bwd_cpt", &g_sweep_bwd_cpt, 4, 8}, {"bwd_pf", &g_sweep_bwd_pf, 0, 2}, {"bwd_gd", &g_sweep_bwd_gd, 0, 2}, {"bwd_gd34", &g_sweep_bwd_gd34, 0, 1}, {"fwd_pt", &g_sweep_fwd_pt, 0, 1}, {"bwd_gpf", &g_sweep_bwd_gpf, 0, 1}, {"fwd_dl", &g_sweep_fwd_dl, 0, 2}, {"sweep_xcd", &g_sweep_xcd, 0, 1}, {"conv0_x3", &g_conv_x3, 0, 3},
    };
    for (const MvsKnob& k : knobs)
        if (strcmp(key, k.name) == 0) return &k;
    return nullptr;
}
extern "C" int mvs_set_tuning(const char* key, int value) {
    MVS_REQUIRE(key, MVS_ERR_NULL, "mvs_set_tuning: null key");
    const MvsKnob* k = mvs_find_knob(key);
    if (!k) {
        mvs_set_error("mvs_set_tuning: unknown key '%s'", key);
        return MVS_ERR_UNSUPPORTED;
    }
    *k->var = value < k->lo ? k->lo : (value > k->hi ? k->hi : value);
    return MVS_OK;
}
// the knob's current value (bench.py --ab restores the LIBRARY's defaults after a toggle: tests/test_capi_symbols.py holds
// _lib.DEFAULT_TUNING against the values a freshly loaded library reports)
extern "C" int mvs_get_tuning(const char* key, int* value) {
    MVS_REQUIRE(key && value, MVS_ERR_NULL, "mvs_get_tuning: null argument");
    const MvsKnob* k = mvs_find_knob(key);
    if (!k) {
        mvs_set_error("mvs_get_tuning: unknown key '%s'", key);
        return MVS_ERR_UNSUPPORTED;
    }
    *value = *k->var;
    return MVS_OK;
}

// the projection-table forward serves a (C, views, channels per thread) combination when the (pixel, view) pairs of a wave fit its 64 lanes
template <int C, int N, int CPT, bool BF>
static bool launch_fwd_pt(const SweepArgs& a, dim3 grid, hipStream_t st) {
    if constexpr ((64 / (C / CPT)) * N <= 64) {
        MVS_LAUNCH((plane_sweep_variance_fwd_pt_kernel<C, N, CPT, BF>), grid, dim3(256), 0, st, a);
        return true;
    } else {
        return false;
    }
}

template <int C>
static int launch_fwd(SweepArgs& a, hipStream_t st) {
    a.tiles_x = mvs_cdiv(a.W, Tile<C>::TW);
    a.tiles_y = mvs_cdiv(a.H, Tile<C>::TH);
    dim3 grid(a.tiles_x * a.tiles_y, mvs_cdiv(a.D, a.dslab), a.B), block(256);
    const int variant = sweep_fwd_variant();
    a.nt_store = g_sweep_nt;
    a.xcd = g_sweep_xcd;
    if (g_sweep_dslab > 0) a.dslab = g_sweep_dslab;
    if (!a.warp_only && variant >= 2 && (a.NS <= 4 || a.NS == 6) && !(variant == 4 && a.NS > 2)) {
        constexpr int CPT8 = C >= 16 ? 8 : 4;
        constexpr int CPT16 = C >= 32 ? 16 : CPT8;
        const bool c16 = variant == 4 && CPT16 == 16;
        // 8 channels per thread up to 4 source views; 6 views' blocks at 8 channels leave one wave per SIMD (256 VGPRs) and
        // measured 3.99 ms against 2.98 ms with 4 channels at config 5 (N = 7, 1600x1184, D = 256; profiles/r02_run11_*)
        const bool c8 = !c16 && variant >= 3 && CPT8 == 8 && a.NS <= 4;
        const int ppb = c16 ? TileC<C, CPT16>::PPB : (c8 ? TileC<C, CPT8>::PPB : TileC<C, 4>::PPB);
        // (round 6, 8 channels per thread: a 4-wide x 16-high pixel tile instead of 8 x 8 -- with the merged re-gather form below K1 at
        //  config 2 0.1141 -> 0.1077 ms, at N = 5 0.2034 -> 0.1945, six interleaved rounds each: profiles/r06_run18_k1_knobs.log)
        //  Per-plane hypotheses at 32 channels only: with 16 channels (a 4 x 32 tile) and per-pixel hypotheses -- CVP's refine sweep at
        //  1152 x 864 -- the narrow tile costs 0.39 -> 0.43 ms and 1.02 -> 1.50 GB of traffic (profiles/r06_final_bench_c4.json vs r06_final5_*).
        int tw = g_sweep_tile_w > 0 ? g_sweep_tile_w : (c16 ? TileC<C, CPT16>::TW : (c8 ? ((C == 32 && !a.per_pixel) ? 4 : TileC<C, CPT8>::TW) : TileC<C, 4>::TW));
        if (tw > ppb) tw = ppb;
        while (ppb % tw) --tw;
        a.tile_w = tw;
        a.tiles_x = mvs_cdiv(a.W, tw);
        a.tiles_y = mvs_cdiv(a.H, ppb / tw);
        if (g_sweep_dslab <= 0) {
            // >= ~2500 workgroups, >= 8 planes each (rounds 1-5, 8 x 8 tiles: ~5000 workgroups, 12-16 planes:
            // profiles/r01_run16_k1_depth_slab_sweep.log); fewer, longer workgroups lose to load imbalance
            const long tiles = (long)a.tiles_x * a.tiles_y * a.B;
            int slab = a.D;
            // (round 6, with the 4-wide tile: 24 planes per workgroup at config 2's 320 tiles -- N = 3 0.1092 -> 0.1076 ms, N = 5 0.1944 -> 0.1880 against 12 planes)
            while (slab > 8 && tiles * mvs_cdiv(a.D, slab) < 2500) slab = (slab + 1) / 2;
            a.dslab = slab;
        }
        const int dl = a.per_pixel ? 0 : g_sweep_fwd_dl;   // knob "fwd_dl": 1 = LDS-staged depths, 2 = + in-block gather waits (per-plane hypotheses)
        if (dl && a.dslab > 512) a.dslab = 512;
        dim3 gridc(a.tiles_x * a.tiles_y, mvs_cdiv(a.D, a.dslab), a.B);
        if (a.bf16_out) {
            MVS_REQUIRE(CPT8 == 8, MVS_ERR_UNSUPPORTED, "plane_sweep bf16 volume: needs >= 16 feature channels");
            constexpr int CB = CPT8 == 8 ? C : 16;       // (C = 8 never gets here; keeps the template instantiable)
            const bool b8 = a.NS <= 4;
            const int ppbb = b8 ? TileC<CB, 8>::PPB : TileC<CB, 4>::PPB;
            a.tile_w = b8 ? TileC<CB, 8>::TW : TileC<CB, 4>::TW;
            a.tiles_x = mvs_cdiv(a.W, a.tile_w);
            a.tiles_y = mvs_cdiv(a.H, ppbb / a.tile_w);
            const long tilesb = (long)a.tiles_x * a.tiles_y * a.B;
            int slab = a.D;
            while (slab > 8 && tilesb * mvs_cdiv(a.D, slab) < 5000) slab = (slab + 1) / 2;
            a.dslab = g_sweep_dslab > 0 ? g_sweep_dslab : slab;
            if (dl && a.dslab > 512) a.dslab = 512;
            dim3 gridb(a.tiles_x * a.tiles_y, mvs_cdiv(a.D, a.dslab), a.B);
#define MVS_BF_CASE(N, CPTN)                                                                                                   \
    case N:                                                                                                                    \
        if (dl && g_sweep_fwd_pt && launch_fwd_pt<CB, N, CPTN, true>(a, gridb, st)) {}                                         \
        else if (dl == 2) MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<CB, N, CPTN, true, 2>), gridb, block, 0, st, a);  \
        else if (dl) MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<CB, N, CPTN, true, 1>), gridb, block, 0, st, a);       \
        else MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<CB, N, CPTN, true, 0>), gridb, block, 0, st, a);               \
        break;
            switch (a.NS) { MVS_BF_CASE(1, 8) MVS_BF_CASE(2, 8) MVS_BF_CASE(3, 8) MVS_BF_CASE(4, 8) MVS_BF_CASE(6, 4) }
#undef MVS_BF_CASE
            return mvs_check_launch("plane_sweep_variance_fwd_cached (bf16 volume)");
        }
#define MVS_CACHED_CASE(N)                                                                                      \
    case N:                                                                                                     \
        if (c16) MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<C, N, CPT16>), gridc, block, 0, st, a);    \
        else if (c8 && dl && g_sweep_fwd_pt && launch_fwd_pt<C, N, CPT8, false>(a, gridc, st)) {}              \
        else if (!c8 && dl && g_sweep_fwd_pt && launch_fwd_pt<C, N, 4, false>(a, gridc, st)) {}                \
        else if (c8 && dl == 2) MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<C, N, CPT8, false, 2>), gridc, block, 0, st, a); \
        else if (c8 && dl) MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<C, N, CPT8, false, 1>), gridc, block, 0, st, a); \
        else if (c8) MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<C, N, CPT8>), gridc, block, 0, st, a); \
        else if (dl == 2) MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<C, N, 4, false, 2>), gridc, block, 0, st, a); \
        else if (dl) MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<C, N, 4, false, 1>), gridc, block, 0, st, a); \
        else MVS_LAUNCH((plane_sweep_variance_fwd_cached_kernel<C, N, 4>), gridc, block, 0, st, a);            \
        break;
        switch (a.NS) {
            MVS_CACHED_CASE(1) MVS_CACHED_CASE(2) MVS_CACHED_CASE(3) MVS_CACHED_CASE(4) MVS_CACHED_CASE(6)
        }
#undef MVS_CACHED_CASE
        return mvs_check_launch("plane_sweep_variance_fwd_cached");
    }
    MVS_REQUIRE(!a.bf16_out, MVS_ERR_UNSUPPORTED, "plane_sweep bf16 volume: only the register-cached forward (knob sweep_fwd >= 2) "
                "with 1, 2, 3, 4 or 6 source views stores bf16");
    switch (a.warp_only ? 1 : a.NS) {
        case 1: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 1>), grid, block, 0, st, a); break;
        case 2: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 2>), grid, block, 0, st, a); break;
        case 3: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 3>), grid, block, 0, st, a); break;
        case 4: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 4>), grid, block, 0, st, a); break;
        case 6: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 6>), grid, block, 0, st, a); break;
        default: MVS_LAUNCH((plane_sweep_variance_fwd_kernel<C, 0>), grid, block, 0, st, a); break;
    }
    return mvs_check_launch("plane_sweep_variance_fwd");
}

#undef z4

template <int C, int NS_T, int GD, int WPS, bool PFL = false>
static int launch_bwd_pw(SweepArgs& a, hipStream_t st) {
    constexpr int CPT = 4;   // 8 channels per thread measured slower at every occupancy (0.46-0.52 vs 0.41 ms, round 2 run 7)
    using Cfg = PwCfg<C, CPT>;
    a.tiles_x = mvs_cdiv(a.W, 2 * Cfg::BW);
    a.tiles_y = mvs_cdiv(a.H, 2 * Cfg::BH);
    // depth slabs: >= ~1280 workgroups (2 resident per CU, 2.5 rounds), each >= 16 planes: every extra slab
    // re-gathers the blocks, writes its windows out once more and adds one grad_ref atomic per pixel and channel
    // (round 6, config 2 / 3 = 640 tiles: 2 slabs of 96 planes instead of 4 of 48 -- step 4.782 -> 4.761 ms, 6.32 -> 6.27 ms; ONE slab of 192
    //  planes is slower again: 4.777 -> 4.789, 6.20 -> 6.32: profiles/r06_run3_bench_ab.json)
    const int tiles = a.tiles_x * a.tiles_y * a.B;
    int nslab = mvs_cdiv(1280, tiles);
    if (nslab > a.D / 16) nslab = a.D / 16;
    if (nslab < 1) nslab = 1;
    a.dslab = g_sweep_bwd_dslab > 0 ? g_sweep_bwd_dslab : mvs_cdiv(a.D, nslab);
    a.no_window = g_sweep_bwd_nowin;
    a.gpf_late = g_sweep_bwd_gpf;
    a.xcd = g_sweep_xcd;
    dim3 grid(a.tiles_x * a.tiles_y, mvs_cdiv(a.D, a.dslab), a.B), block(256);
    if (a.warp_only) {
        if constexpr (NS_T == 1) MVS_LAUNCH((plane_sweep_variance_bwd_pw_kernel<C, 1, CPT, 2, GD, WPS, false, PFL>), grid, block, 0, st, a);
    } else if (a.ms_alias) MVS_LAUNCH((plane_sweep_variance_bwd_pw_kernel<C, NS_T, CPT, 1, GD, WPS, false, PFL>), grid, block, 0, st, a);
    else if (a.per_pixel) MVS_LAUNCH((plane_sweep_variance_bwd_pw_kernel<C, NS_T, CPT, 0, GD, WPS, false, PFL>), grid, block, 0, st, a);
    else MVS_LAUNCH((plane_sweep_variance_bwd_pw_kernel<C, NS_T, CPT, 0, GD, WPS, true, PFL>), grid, block, 0, st, a);
    return mvs_check_launch("plane_sweep_variance_bwd_pw");
}

template <int C>
static int launch_bwd(SweepArgs& a, hipStream_t st) {
    if (g_sweep_bwd_variant != 1 && a.NS <= 4 && (long long)a.B * a.H * a.W * C < (1LL << 31)) {   // (32-bit feature-map offsets in the per-wave-window kernel)
        // 1-2 views: 2 waves per SIMD with the upstream gradient requested two planes ahead (knob "bwd_gd" = 0: 3 waves per SIMD, one
        // rotating register set); 3-4 views: 2 waves per SIMD (knob "bwd_pf" = 2: ONE wave per SIMD, 512 registers, nothing spills)
        const bool gd2 = g_sweep_bwd_gd == 2;
        if (g_sweep_bwd_pf == 1 && a.NS <= 2) {   // knob "bwd_pf" = 1: block lookahead (2 waves/SIMD, one-plane groups)
            if (a.NS == 1) return launch_bwd_pw<C, 1, 0, 2, true>(a, st);
            return launch_bwd_pw<C, 2, 0, 2, true>(a, st);
        }
        if (a.NS == 1) return gd2 ? launch_bwd_pw<C, 1, 2, 2>(a, st) : launch_bwd_pw<C, 1, 0, 3>(a, st);
        if (a.NS == 2) return gd2 ? launch_bwd_pw<C, 2, 2, 2>(a, st) : launch_bwd_pw<C, 2, 0, 3>(a, st);
        if (a.NS == 3) return g_sweep_bwd_pf == 2 ? launch_bwd_pw<C, 3, 2, 1>(a, st) : (g_sweep_bwd_gd34 ? launch_bwd_pw<C, 3, 2, 2>(a, st) : launch_bwd_pw<C, 3, 0, 2>(a, st));
        return g_sweep_bwd_pf == 2 ? launch_bwd_pw<C, 4, 2, 1>(a, st) : (g_sweep_bwd_gd34 ? launch_bwd_pw<C, 4, 2, 2>(a, st) : launch_bwd_pw<C, 4, 0, 2>(a, st));
    }
    // more than four source views (or knob "sweep_bwd" = 1): view pairs per workgroup, LDS-atomic windows
    a.tiles_x = mvs_cdiv(a.W, Tile<C>::TW);
    a.tiles_y = mvs_cdiv(a.H, Tile<C>::TH);
    const int ngroups = mvs_cdiv(a.NS, 2);
    // depth slabs so that the launch has >= ~2048 workgroups (2 resident per CU), each >= 16 planes
    int nslab = mvs_cdiv(2048, a.tiles_x * a.tiles_y * a.B * ngroups);
    if (nslab > a.D / 16) nslab = a.D / 16;
    if (nslab < 1) nslab = 1;
    a.dslab = mvs_cdiv(a.D, nslab);
    nslab = mvs_cdiv(a.D, a.dslab);
    dim3 grid(a.tiles_x * a.tiles_y, ngroups * nslab, a.B), block(256);
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
    if (align_corners) { a.sx = 1.0f; a.ox = 0.0f; a.sy = 1.0f; a.oy = 0.0f; }
    else {
        a.sx = (float)((double)W / (double)(W - 1)); a.ox = -0.5f;
        a.sy = (float)((double)H / (double)(H - 1)); a.oy = -0.5f;
    }
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

// Same with the volume stored in bf16 [B,D,H,W,C] (inference path, BASELINE configs[4]); 1, 2, 3, 4 or 6 source views,
// C = 16 or 32, per-plane or per-pixel hypotheses.
extern "C" int mvs_plane_sweep_variance_fwd_bf16(const float* ref, const float* const* srcs, const float* rot,
                                                 const float* trans, const float* depth, int depth_is_per_pixel,
                                                 int B, int N, int C, int D, int H, int W, int align_corners,
                                                 int ms_alias, void* var_out_bf16, hipStream_t stream) {
    SweepArgs a = {};
    int rc = fill_args(a, ref, srcs, rot, trans, depth, depth_is_per_pixel, B, N, C, D, H, W, align_corners, ms_alias);
    if (rc) return rc;
    MVS_REQUIRE(var_out_bf16, MVS_ERR_NULL, "plane_sweep fwd bf16: null output");
    MVS_REQUIRE(C == 16 || C == 32, MVS_ERR_UNSUPPORTED, "plane_sweep fwd bf16: C must be 16 or 32, got %d", C);
    MVS_REQUIRE(a.NS <= 4 || a.NS == 6, MVS_ERR_UNSUPPORTED, "plane_sweep fwd bf16: 1, 2, 3, 4 or 6 source views, got %d", a.NS);
    a.var = (float*)var_out_bf16;
    a.bf16_out = 1;
    if (C == 32) return launch_fwd<32>(a, stream);
    return launch_fwd<16>(a, stream);
}

// grad_ref and grad_srcs[i] must be ZERO-FILLED by the caller (accumulated atomically)
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

// ---- rot / trans of src_proj @ inverse(ref_proj) for all source views: ONE launch -----------------------------------
// The reference's host lines (jdacs/models/module.py:116-118: torch.matmul(src_proj, torch.inverse(ref_proj)), once per
// source view) cost ~10 tiny library launches per step on the GPU (LU factorisation + solve + matmul + slices per view:
// 91 us of the 6.3-ms training step).  One thread per (sample, view): Gauss-Jordan with partial pivoting and the 3x4 product in
// fp64, rounded to fp32 at the end (closer to the exact result than the fp32 LU; a singular ref_proj gives inf / nan like
// torch.linalg.inv_ex, no exception).
__global__ void relative_projection_kernel(const float* __restrict__ src, const float* __restrict__ ref, int B, int NS,
                                           float* __restrict__ rot, float* __restrict__ trans) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NS) return;
    const int b = i / NS;
    double m[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { m[r][c] = (double)ref[b * 16 + r * 4 + c]; m[r][4 + c] = r == c ? 1.0 : 0.0; }
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        double best = fabs(m[col][col]);
#pragma unroll
        for (int r = col + 1; r < 4; ++r) {
            const double v = fabs(m[r][col]);
            if (r > col && v > best) { best = v; piv = r; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r == piv && piv != col) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { const double t = m[col][c]; m[col][c] = m[r][c]; m[r][c] = t; }
            }
        const double inv = 1.0 / m[col][col];
#pragma unroll
        for (int c = 0; c < 8; ++c) m[col][c] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = m[r][col];
#pragma unroll
            for (int c = 0; c < 8; ++c) m[r][c] -= f * m[col][c];
        }
    }
    const float* __restrict__ sp = src + (size_t)i * 16;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += (double)sp[r * 4 + k] * m[k][4 + c];
            if (c < 3) rot[(size_t)i * 9 + r * 3 + c] = (float)acc;
            else trans[(size_t)i * 3 + r] = (float)acc;
        }
    }
}

extern "C" int mvs_relative_projection(const float* src_proj, const float* ref_proj, int B, int NS, float* rot, float* trans,
                                       hipStream_t stream) {
    MVS_REQUIRE(src_proj && ref_proj && rot && trans, MVS_ERR_NULL, "relative_projection: null pointer argument");
    MVS_REQUIRE(B > 0 && NS > 0 && NS <= MVS_MAX_SRC, MVS_ERR_SHAPE, "relative_projection: bad shape B=%d source views=%d", B, NS);
    MVS_LAUNCH(relative_projection_kernel, dim3(mvs_cdiv(B * NS, 64)), dim3(64), 0, stream, src_proj, ref_proj, B, NS, rot, trans);
    return mvs_check_launch("relative_projection");
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
