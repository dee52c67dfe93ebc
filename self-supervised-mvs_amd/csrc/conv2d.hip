// SURVEY.md 8(f)-3 (first cut, OFF by default -- see jdacs/models/module.py::ConvBnReLU.hip_conv): the 2-D convolutions
// of the feature extractors (jdacs/models/mvsnet.py:17-34: 3x3 stride 1 pad 1 and 5x5 stride 2 pad 2, 3/8/16/32
// channels; jdacs-ms/models/network.py:16-41 uses the same 3x3 shapes) on channels-last images, as fp32 MFMA implicit
// GEMMs in the style of conv3d.hip.  Not yet measured on the GPU: kernel logic is covered by the CPU emulation tests,
// tile shapes and staging are the untuned first version.
//
//   forward      y[n,oy,ox,co] = sum_{ty,tx,ci} x[n, oy*S + ty - P, ox*S + tx - P, ci] * w[co][ci][ty][tx]   (+ bias)
//   input grad   stride 1: the same kernel on gy with the weights transposed and flipped;
//                stride 2: four output-parity classes, each a 3x3 stride-1 pass over gy on the coarse grid with its
//                own (partly empty) weight image; a direct VALU form is kept behind mvs_set_tuning("conv2d_s2_mfma", 0)
//   weight grad  dW[(ty,tx,ci)][co] = sum_positions x[...] * gy[...]: rows = (tap, ci), columns = co, K = positions;
//                persistent workgroups, one partial image each, deterministic finish
#include "mvs_rt.h"
#include "conv_map.h"   // MVS_HD

struct Conv2dArgs {
    const float* x;      // [N,Hi,Wi,Cin]
    const float* wp;     // packed weights [kstep][nb][64][4]
    const float* bias;   // [Cout] or null
    float* y;            // [N,Ho,Wo,Cout]
    int N, Hi, Wi, Ho, Wo, Cin, Cout;   // Ho x Wo: the grid the tiles walk (== the output map for a forward pass)
    int nth, ntw, nb_total;
    int os, py, px, YH, YW;             // output pixel of grid point (oy,ox) is (oy*os + py, ox*os + px) of a YH x YW map
    int act;                            // 1: LeakyReLU(slope) after the bias (jdacs-ms/models/modules.py:15-19)
    float slope;
    double* slots;                      // STATS kernels: BatchNorm statistic slots [group][nslots][2][Cout] (fp64 atomics, bn.hip):
    int nslots, imgs_per_group;         //   (sum, sum of squares) of the workgroup's outputs -> slot row (workgroup mod nslots) of image n's group
    const float* bn_raw;                // BST kernels (an input gradient that completes the output gradient of the BatchNorm + ReLU block in
    const float* bn_stats;              //   front): that block's raw output [N,YH,YW,Cout] and stats [group][4][Cout]; `slots` then receive the
                                        //   block's BACKWARD statistics (sum dyh, sum dyh * xhat) instead of (sum y, sum y^2)
    const float* in_stats;              // XF kernels: x is the RAW output of the BatchNorm block in front; [group][4][Cin] (mean, invstd, scale,
                                        //   shift: mvs_bn_finalize_slots) -- relu(x * scale + shift) is applied while the halo is staged
};

template <int KS, int S>
struct Geo2 {
    static constexpr int TH = 8, TW = 32, P = KS / 2;
    static constexpr int RH = (TH - 1) * S + KS, RW = (TW - 1) * S + KS, NT = KS * KS;
};

MVS_HD inline int c2_ksteps(int ntaps, int cc) { return (ntaps * cc + 15) / 16; }

// packed image: lane l, element j of k-step ks, column tile nb holds W[flattened k = 16 ks + 4 (l>>4) + j][co = 16 nb + (l&15)],
// flattened k -> (tap = k / CC, ci = chunk * CC + k % CC); zero beyond the taps / channels.
// layout 0: w[co][ci][tap]; layout 1 (input gradient of a stride-1 layer): w[ci'][co'][tap] read as co = ci', ci = co', tap flipped
// cls >= 0 (input gradient of the 5x5 stride-2 layer, parity class cls = 2 py + px): a 3x3 stride-1 kernel over gy whose tap
// (ty', tx') is the original tap (ty, tx) with ty = 2 (2 - ty') + py - ... see c2_s2_tap; w is [co'][ci'][5][5] read transposed.
MVS_HD inline int c2_s2_tap(int tp, int par) {   // tp in 0..2 = offset -1, 0, +1 on the coarse grid; -> original tap or -1
    // iy = 2 q + par, oy = (iy + 2 - t) / 2 = q + off  =>  t = par + 2 - 2 off, off = tp - 1
    const int t = par + 2 - 2 * (tp - 1);
    return (t >= 0 && t < 5) ? t : -1;
}
// pp (pixel pairs, 3x3 stride 1, Cout <= 8, one column tile): column n = p*8 + co is output channel co of the pixel with X parity p
// of the row's pixel pair; K walks the 3 x 4 input offsets under the pair: tap' = ty*4 + tx', the weight of column (p, co) at tap'
// is W[ty][tx' - p] (zero where tx' - p is outside 0..2).  All 16 columns carry channels (a plain Cout = 8 layer fills 8 of them),
// 12 taps per pixel PAIR instead of 9 per pixel, and the 16 lanes of a row store 64 consecutive bytes.
// wcl (forward images only): the parameter tensor is channels-last in memory ([Cout][ky][kx][Cin], what
// module.to(memory_format=torch.channels_last) makes of a Conv2d weight) instead of [Cout][Cin][ky][kx].
// parameter element [o][i][tap] of a layer with I input channels and T taps: [O][I][T] contiguous, or [O][T][I] when the tensor is channels-last in memory
__device__ __forceinline__ float c2_wt(const float* __restrict__ w, int o, int i, int I, int T, int tap, int wcl) {
    return wcl ? w[((size_t)o * T + tap) * I + i] : w[((size_t)o * I + i) * T + tap];
}
__device__ __forceinline__ void conv2d_pack_item(const float* __restrict__ w, float* __restrict__ wp, int NT, int CC, int Cin, int Cout,
                                                 int NB, int transposed, int cls, int pp, int wcl, int idx) {
    const int j = idx & 3, lane = (idx >> 2) & 63, nb = (idx >> 8) % NB, kk = (idx >> 8) / NB;
    const int KS = c2_ksteps(pp ? 12 : NT, CC), chunk = kk / KS, ks = kk % KS;
    const int k = 16 * ks + 4 * (lane >> 4) + j, tap = k / CC, ci = chunk * CC + k % CC, co = pp ? (lane & 7) : nb * 16 + (lane & 15);
    float v = 0.f;
    // (input-gradient images read the parameter transposed: element [co_layer = ci][ci_layer = co][tap'] -- c2_wt below -- in either memory layout)
    if (pp) {
        const int p = (lane & 15) >> 3, ty = tap / 4, tx = tap % 4 - p;
        if (tap < 12 && tx >= 0 && tx <= 2 && ci < Cin && co < Cout) {
            const int t9 = ty * 3 + tx;
            v = transposed ? c2_wt(w, ci, co, Cout, 9, 8 - t9, wcl)
                           : (wcl ? w[((size_t)co * 9 + t9) * Cin + ci] : w[((size_t)co * Cin + ci) * 9 + t9]);
        }
    } else if (tap < NT && ci < Cin && co < Cout) {
        if (cls >= 0) {
            const int ty = c2_s2_tap(tap / 3, cls >> 1), tx = c2_s2_tap(tap % 3, cls & 1);
            if (ty >= 0 && tx >= 0) v = c2_wt(w, ci, co, Cout, 25, ty * 5 + tx, wcl);
        } else {
            v = transposed ? c2_wt(w, ci, co, Cout, NT, NT - 1 - tap, wcl)
                           : (wcl ? w[((size_t)co * NT + tap) * Cin + ci] : w[((size_t)co * Cin + ci) * NT + tap]);
        }
    }
    wp[idx] = v;
}
__global__ __launch_bounds__(256) void conv2d_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int NT, int CC,
                                                          int Cin, int Cout, int NB, int transposed, int total, int cls, int pp, int wcl) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    conv2d_pack_item(w, wp, NT, CC, Cin, Cout, NB, transposed, cls, pp, wcl, idx);
}
// Round 6: the four parity-class images of a 5x5 stride-2 layer's input gradient in ONE launch (blockIdx.y = class), taps COMPACTED:
// class (py, px) only has (3 - py) x (3 - px) non-empty taps of the 3x3 coarse-grid kernel (c2_s2_tap: offset -1 of an odd parity has
// no original tap), so its K dimension walks 9 / 6 / 6 / 4 taps instead of 9 each (25 instead of 36 tap-slices of MFMA work).
// Image of class c at wp + c2_s2d_prefix(c): [chunk][ksteps_c][NB][64][4]; flattened k = 16 ks + 4 (l>>4) + j -> (compact tap i = k / CC,
// ci = chunk*CC + k % CC), compact tap i -> coarse offsets (ty', tx') = (py + i / ntx, px + i % ntx).
MVS_HD inline int c2_s2d_ntaps(int cls) { return (3 - (cls >> 1)) * (3 - (cls & 1)); }
MVS_HD inline int c2_s2d_floats(int cls, int CC, int nch, int NB) { return nch * c2_ksteps(c2_s2d_ntaps(cls), CC) * NB * 256; }
MVS_HD inline int c2_s2d_prefix(int cls, int CC, int nch, int NB) {
    int s = 0;
    for (int c = 0; c < cls; ++c) s += c2_s2d_floats(c, CC, nch, NB);
    return s;
}
__global__ __launch_bounds__(256) void conv2d_pack_s2d_kernel(const float* __restrict__ w, float* __restrict__ wp, int CC, int Cin, int Cout,
                                                              int NB, int nch, int wcl) {
    const int cls = blockIdx.y, py = cls >> 1, px = cls & 1, ntx = 3 - px, ntk = c2_s2d_ntaps(cls);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= c2_s2d_floats(cls, CC, nch, NB)) return;
    const int j = idx & 3, lane = (idx >> 2) & 63, nb = (idx >> 8) % NB, kk = (idx >> 8) / NB;
    const int KS = c2_ksteps(ntk, CC), chunk = kk / KS, ks = kk % KS;
    const int k = 16 * ks + 4 * (lane >> 4) + j, tap = k / CC, ci = chunk * CC + k % CC, co = nb * 16 + (lane & 15);
    float v = 0.f;
    if (tap < ntk && ci < Cin && co < Cout) {
        const int ty = c2_s2_tap(py + tap / ntx, py), tx = c2_s2_tap(px + tap % ntx, px);
        v = c2_wt(w, ci, co, Cout, 25, ty * 5 + tx, wcl);       // w[co_layer = ci][ci_layer = co][ty][tx]
    }
    wp[c2_s2d_prefix(cls, CC, nch, NB) + idx] = v;
}
// the forward images of a list of layers in one launch (blockIdx.y = list entry): the 2-D extractor packs its layers once per
// step (round 3: one pack launch, plus one layout copy of a channels-last weight, in front of every convolution)
struct Pack2dItem {
    const float* w;
    float* wp;
    int NT, CC, Cin, Cout, NB, pp, wcl, total;
};
#define MVS_PACK2D_BATCH_MAX 16
struct Pack2dBatch {
    Pack2dItem it[MVS_PACK2D_BATCH_MAX];
};
__global__ __launch_bounds__(256) void conv2d_pack_batch_kernel(Pack2dBatch pb) {
    const Pack2dItem& p = pb.it[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.total) return;
    conv2d_pack_item(p.w, p.wp, p.NT, p.CC, p.Cin, p.Cout, p.NB, 0, -1, p.pp, p.wcl, idx);
}

// STATS: the workgroup also adds the per-channel sum and sum of squares of its outputs into a BatchNorm statistic slot row of its
// image's statistics group -- BatchNorm's statistics pass folded into the convolution that produces its input, like the 3-D
// kernels' epilogue.  Separate instantiations: the plain kernels' code and register allocation do not change.
// XF: the input is normalised on the way into LDS (consumer-side BatchNorm + ReLU: the producing block has no apply pass).  Zero
// padding stays zero: only elements inside the image are transformed.
// BST (with STATS): the output is the complete gradient w.r.t. relu(bn(raw)) of the block in front; what goes to the slots is that
// block's backward statistics, like the 3-D input-gradient epilogues (conv3d.hip) -- its reduce pass (mvs_bn_bwd_reduce_slots) goes.
// S2D (round 6): the input gradient of a 5x5 stride-2 layer, all four output-parity classes in ONE launch (class = blockIdx.z): a 3x3
// stride-1 pass over gy on the coarse grid per class with the class's COMPACTED tap list (conv2d_pack_s2d_kernel); a.py / a.px / a.wp are
// replaced by the class's.  Rounds 2-5 issued four pack launches + four passes of nine taps each.
template <int KS, int S, int CC, int NB, bool PP = false, bool STATS = false, bool XF = false, bool BST = false, bool S2D = false>
__global__ __launch_bounds__(256) void conv2d_igemm_kernel(Conv2dArgs a) {
    using G = Geo2<KS, S>;
    static_assert(!PP || (KS == 3 && S == 1 && NB == 1), "pixel pairs: 3x3 stride 1, one column tile");
    static_assert(!S2D || (KS == 3 && S == 1 && !PP && !STATS && !XF && !BST), "S2D: the plain 3x3 stride-1 pass");
    constexpr int NTK = PP ? 12 : G::NT;                 // taps the K dimension walks
    constexpr int MBW = PP ? 2 : 4;                      // m-blocks per wave: a block is 16 pixel PAIRS of a row with PP
    constexpr int CCP = CC + 4, CQ = CC / 4, NR = G::RH * G::RW, KSTEPS = (NTK * CC + 15) / 16;
    __shared__ __attribute__((aligned(16))) float tile[NR * CCP];
    __shared__ int tapoff[(NTK + 4 + 3) & ~3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
    int t = blockIdx.x;
    const int tw = t % a.ntw; t /= a.ntw;
    const int th = t % a.nth; t /= a.nth;
    const int n = t;
    const int oy0 = th * G::TH, ox0 = tw * G::TW;
    const int nb0 = blockIdx.y * NB;
    int ksteps = KSTEPS, py = a.py, px = a.px;
    const float* __restrict__ wp = a.wp;
    if constexpr (S2D) {
        const int cls = blockIdx.z, nch = (a.Cin + CC - 1) / CC;
        py = cls >> 1; px = cls & 1;
        if (oy0 * 2 + py >= a.YH || ox0 * 2 + px >= a.YW) return;           // the odd classes have one grid row / column fewer
        const int ntx = 3 - px, ntk = c2_s2d_ntaps(cls);
        ksteps = c2_ksteps(ntk, CC);
        wp += c2_s2d_prefix(cls, CC, nch, a.nb_total);
        for (int i = tid; i < (int)(sizeof(tapoff) / sizeof(int)); i += 256)
            tapoff[i] = i < ntk ? ((py + i / ntx) * G::RW + px + i % ntx) * CCP : 0;
    } else
    for (int i = tid; i < (int)(sizeof(tapoff) / sizeof(int)); i += 256)   // padded k-steps read a valid location (zero weights)
        tapoff[i] = i < NTK ? (PP ? ((i / 4) * G::RW + i % 4) * CCP : ((i / KS) * G::RW + i % KS) * CCP) : 0;
    // wave -> output rows 2w, 2w+1; m-block mb: row 2w + (mb >> 1), columns 16 (mb & 1) + l15
    // (PP: m-block mb = row 2w + mb, pixel pair l15 = columns 2 l15, 2 l15 + 1)
    int baseA[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
        baseA[mb] = PP ? ((2 * wave + mb) * G::RW + 2 * l15) * CCP : (((2 * wave + (mb >> 1)) * S) * G::RW + (16 * (mb & 1) + l15) * S) * CCP;
    f32x4 acc[MBW][NB];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nchunks = (a.Cin + CC - 1) / CC;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();
        // halo tile of channels [chunk*CC, +CC): zero outside the image and beyond Cin.  All loads of a batch are issued
        // before the first LDS write (a load -> store loop serialises one memory round trip per iteration).
        constexpr int NIT = (NR * CQ + 255) / 256, BATCH = NIT < 12 ? NIT : 12;
        float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (XF) {   // a thread stages the same channel quad in every batch (256 % CQ == 0); Cin % 4 == 0 for a BatchNorm output
            const int c0 = chunk * CC + 4 * (tid % CQ);
            const float* __restrict__ st = a.in_stats + (size_t)(n / a.imgs_per_group) * 4 * a.Cin;
            if (c0 + 3 < a.Cin) {
                xsc = *reinterpret_cast<const float4*>(st + 2 * a.Cin + c0);
                xsh = *reinterpret_cast<const float4*>(st + 3 * a.Cin + c0);
            }
        }
#pragma unroll
        for (int k0 = 0; k0 < NIT; k0 += BATCH) {
            float4 v[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int i = tid + 256 * (k0 + k);
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + k < NIT && i < NR * CQ) {
                    const int px = i / CQ, cq = i % CQ;
                    const int rx = px % G::RW, ry = px / G::RW;
                    const int iy = oy0 * S + ry - G::P, ix = ox0 * S + rx - G::P, c0 = chunk * CC + 4 * cq;
                    if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
                        const float* __restrict__ src = a.x + (((size_t)n * a.Hi + iy) * a.Wi + ix) * a.Cin + c0;
                        if ((a.Cin & 3) == 0 && c0 + 3 < a.Cin) v[k] = *reinterpret_cast<const float4*>(src);
                        else {
                            if (c0 < a.Cin) v[k].x = src[0];
                            if (c0 + 1 < a.Cin) v[k].y = src[1];
                            if (c0 + 2 < a.Cin) v[k].z = src[2];
                            if (c0 + 3 < a.Cin) v[k].w = src[3];
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int i = tid + 256 * (k0 + k);
                if (k0 + k < NIT && i < NR * CQ) {
                    float4 o = v[k];
                    if (XF) {
                        const int px = i / CQ, rx = px % G::RW, ry = px / G::RW;
                        const int iy = oy0 * S + ry - G::P, ix = ox0 * S + rx - G::P;
                        if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
                            // (multiply, then add: the apply kernel's arithmetic, csrc/bn.hip, to the bit)
                            o.x = fmaxf(o.x * xsc.x + xsh.x, 0.f); o.y = fmaxf(o.y * xsc.y + xsh.y, 0.f);
                            o.z = fmaxf(o.z * xsc.z + xsh.z, 0.f); o.w = fmaxf(o.w * xsc.w + xsh.w, 0.f);
                        }
                    }
                    *reinterpret_cast<float4*>(&tile[(i / CQ) * CCP + 4 * (i % CQ)]) = o;
                }
            }
        }
        __syncthreads();
        // weight fragments (global / L2) run two k-steps ahead of the MFMAs that consume them
        float4 bq[2][NB];
        auto load_b = [&](int ks, float4 (&dst)[NB]) {
            const int kc = ks < ksteps ? ks : ksteps - 1;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                dst[nb] = *reinterpret_cast<const float4*>(wp + (((size_t)(chunk * ksteps + kc) * a.nb_total + nb0 + nb) * 64 + lane) * 4);
        };
        load_b(0, bq[0]);
        load_b(1, bq[1]);
        for (int ks0 = 0; ks0 < ksteps; ks0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ks = ks0 + u;
                if (ks < ksteps) {
                    const int kflat = 16 * ks + 4 * g;
                    const int aoff = tapoff[kflat / CC] + kflat % CC;
                    float4 af[MBW];
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) af[mb] = *reinterpret_cast<const float4*>(&tile[baseA[mb] + aoff]);
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            // (the WEIGHT fragment is the A operand -- the packed image serves as either: lane (l15, g) holds
                            //  W[k = 4g..][column l15] -- so the result tile is transposed: see the epilogue)
                            acc[mb][nb] = MVS_MFMA_16x16x4(bq[u][nb].x, af[mb].x, acc[mb][nb]);
                            acc[mb][nb] = MVS_MFMA_16x16x4(bq[u][nb].y, af[mb].y, acc[mb][nb]);
                            acc[mb][nb] = MVS_MFMA_16x16x4(bq[u][nb].z, af[mb].z, acc[mb][nb]);
                            acc[mb][nb] = MVS_MFMA_16x16x4(bq[u][nb].w, af[mb].w, acc[mb][nb]);
                        }
                    load_b(ks + 2, bq[u]);
                }
            }
        }
    }
    // D layout (weights as the A operand): row = 4 (lane >> 4) + r -> output column n of the weight tile, col = lane & 15 -> position
    // within the m-block.  A lane ends with FOUR CONSECUTIVE CHANNELS of one pixel: the raw values of the backward statistics are read
    // and the result is written as float4 -- a wave instruction covers 16 pixels x 64 bytes (16 pixel pairs x 64 bytes with PP) of
    // contiguous memory; with the positions in the lane's registers (rounds 1-4) the same bytes took four 4-byte instructions.
    // (PP: column n = p*8 + co -> lanes g = 0, 1 hold the two channel quads of pixel 2 l15, lanes g = 2, 3 those of pixel 2 l15 + 1)
    float st1[NB][4], st2[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) st1[nb][r] = st2[nb][r] = 0.f;
    float bmean[BST ? NB : 1][4], binv[BST ? NB : 1][4], bsc[BST ? NB : 1][4], bsh[BST ? NB : 1][4];
    if (BST) {
        const float* __restrict__ bs = a.bn_stats + (size_t)(n / a.imgs_per_group) * 4 * a.Cout;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (PP ? 4 * (g & 1) : (nb0 + nb) * 16 + 4 * g) + r, cc = co < a.Cout ? co : 0;
                bmean[nb][r] = bs[cc]; binv[nb][r] = bs[a.Cout + cc]; bsc[nb][r] = bs[2 * a.Cout + cc]; bsh[nb][r] = bs[3 * a.Cout + cc];
            }
    }
    const bool vec = (a.Cout & 3) == 0;            // (a 1- or 3-channel output layer takes the scalar stores)
    float4 rawv[BST ? MBW : 1][BST ? NB : 1];
    if (BST) {      // ALL requested before the first store (a load written after a store waits for it: the stores may alias as far as hipcc knows)
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int oy = (oy0 + 2 * wave + (PP ? mb : (mb >> 1))) * a.os + py;
                const int ox = (ox0 + (PP ? 2 * l15 + (g >> 1) : 16 * (mb & 1) + l15)) * a.os + px;
                const int co0 = PP ? 4 * (g & 1) : (nb0 + nb) * 16 + 4 * g;
                float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (oy < a.YH && ox < a.YW && co0 < a.Cout) {
                    const float* __restrict__ rp = a.bn_raw + (((size_t)n * a.YH + oy) * a.YW + ox) * a.Cout + co0;
                    if (vec) rv = *reinterpret_cast<const float4*>(rp);
                    else { rv.x = rp[0]; if (co0 + 1 < a.Cout) rv.y = rp[1]; if (co0 + 2 < a.Cout) rv.z = rp[2]; if (co0 + 3 < a.Cout) rv.w = rp[3]; }
                }
                rawv[mb][nb] = rv;
            }
    }
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        const int oy = (oy0 + 2 * wave + (PP ? mb : (mb >> 1))) * a.os + py;
        const int ox = (ox0 + (PP ? 2 * l15 + (g >> 1) : 16 * (mb & 1) + l15)) * a.os + px;
        if (oy >= a.YH || ox >= a.YW) continue;
        float* __restrict__ o = a.y + (((size_t)n * a.YH + oy) * a.YW + ox) * a.Cout;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int co0 = PP ? 4 * (g & 1) : (nb0 + nb) * 16 + 4 * g;
            if (co0 >= a.Cout) continue;
            float ov[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[mb][nb][r] + ((a.bias && co0 + r < a.Cout) ? a.bias[co0 + r] : 0.f);
                if (a.act) v = v > 0.f ? v : v * a.slope;
                ov[r] = v;
                if (co0 + r < a.Cout) {
                    if (BST) {
                        const float rw = r == 0 ? rawv[BST ? mb : 0][BST ? nb : 0].x : (r == 1 ? rawv[BST ? mb : 0][BST ? nb : 0].y : (r == 2 ? rawv[BST ? mb : 0][BST ? nb : 0].z : rawv[BST ? mb : 0][BST ? nb : 0].w));
                        const float d1 = (rw * bsc[BST ? nb : 0][r] + bsh[BST ? nb : 0][r] > 0.f) ? v : 0.f;
                        st1[nb][r] += d1; st2[nb][r] = fmaf(d1, (rw - bmean[BST ? nb : 0][r]) * binv[BST ? nb : 0][r], st2[nb][r]);
                    } else if (STATS) { st1[nb][r] += v; st2[nb][r] = fmaf(v, v, st2[nb][r]); }
                }
            }
            if (vec) *reinterpret_cast<float4*>(o + co0) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < a.Cout) o[co0 + r] = ov[r];
            }
        }
    }
    if constexpr (STATS) {
        // the 16 lanes of a group hold 16 positions of the same four columns: butterfly over them (fixed order), one partial per
        // (wave, column), summed over the four waves in a fixed order; PP: columns c and c + 8 are the same channel
        __shared__ float red[4 * NB * 16 * 2];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = st1[nb][r], s2 = st2[nb][r];
                s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4); s1 += __shfl_xor(s1, 8);
                s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4); s2 += __shfl_xor(s2, 8);
                if (l15 == 0) {
                    red[((wave * NB + nb) * 16 + 4 * g + r) * 2] = s1;
                    red[((wave * NB + nb) * 16 + 4 * g + r) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        const int ncol = PP ? 8 : NB * 16;
        if (tid < 2 * ncol) {
            const int stat = tid / ncol, col = tid % ncol;
            float t = 0.f;
            for (int k = 0; k < 4; ++k) {
                t += red[(k * NB * 16 + col) * 2 + stat];
                if (PP) t += red[(k * NB * 16 + col + 8) * 2 + stat];
            }
            const int co = PP ? col : nb0 * 16 + col;
            if (co < a.Cout)
                MVS_GLOBAL_ATOMIC_ADD_F64(a.slots + (((size_t)(n / a.imgs_per_group) * a.nslots + (blockIdx.x & (a.nslots - 1))) * 2 + stat) * a.Cout + co,
                                          (double)t);
        }
    }
}

// input gradient of a stride-2 layer, direct form: gx[n,iy,ix,ci] = sum over the taps (ty,tx) with iy + P - ty and ix + P - tx
// even and the output position inside the map, and over co, of gy[n,(iy+P-ty)/2,(ix+P-tx)/2,co] * w[co][ci][ty][tx]
template <int KS>
__global__ __launch_bounds__(256) void conv2d_dgrad_s2_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                              float* __restrict__ gx, int N, int Hi, int Wi, int Ho, int Wo,
                                                              int Cin, int Cout) {
    constexpr int P = KS / 2;
    const size_t total = (size_t)N * Hi * Wi * Cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        size_t p = i / Cin;
        const int ix = (int)(p % Wi); p /= Wi;
        const int iy = (int)(p % Hi);
        const int n = (int)(p / Hi);
        float s = 0.f;
        for (int ty = (iy + P) & 1; ty < KS; ty += 2) {
            const int oy = (iy + P - ty) / 2;
            if (iy + P - ty < 0 || oy >= Ho) continue;
            for (int tx = (ix + P) & 1; tx < KS; tx += 2) {
                const int ox = (ix + P - tx) / 2;
                if (ix + P - tx < 0 || ox >= Wo) continue;
                const float* __restrict__ g = gy + (((size_t)n * Ho + oy) * Wo + ox) * Cout;
                const float* __restrict__ wt = w + (size_t)ci * KS * KS + ty * KS + tx;
                for (int co = 0; co < Cout; ++co) s = fmaf(g[co], wt[(size_t)co * Cin * KS * KS], s);
            }
        }
        gx[i] = s;
    }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
struct Wgrad2dArgs {
    const float* x;     // [N,Hi,Wi,CX]
    const float* g;     // [N,Ho,Wo,CG]
    float* part;        // [groups][rows = NT*CXP][CGP]
    int N, Hi, Wi, Ho, Wo, CX, CG, nth, ntw;   // CX / CG: channels of THIS pass (a slice of <= 32 when the layer has more)
    int xs, x0, gs, g0;                         // channel stride and first channel of the slice in x and g
};

// CXP: CX rounded up to a multiple of 4 (LDS image), MT: m-tiles (16 rows of (tap, cx)) per wave, NB: 16-wide CG tiles
template <int KS, int S, int CXP, int MT, int NB>
__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(Wgrad2dArgs a) {
    using G = Geo2<KS, S>;
    constexpr int NR = G::RH * G::RW, NPOS = G::TH * G::TW, ROWS = G::NT * CXP, CGP = NB * 16;
    constexpr int XP = CXP + 1;                       // odd pixel stride: the A gather walks (tap, cx) rows
    __shared__ float xt[NR * XP];
    __shared__ __attribute__((aligned(16))) float gt[NPOS * CGP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, l15 = lane & 15;
    // m-tile m of this wave covers rows 16 (wave + 4 m) .. +15; row -> (tap = row / CXP, cx = row % CXP)
    int rowoff[MT];
    bool rowok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = 16 * (wave + 4 * m) + l15;
        rowok[m] = row < ROWS;
        const int tap = rowok[m] ? row / CXP : 0, cx = row % CXP;
        rowoff[m] = ((tap / KS) * G::RW + tap % KS) * XP + cx;
    }
    f32x4 acc[MT][NB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ntiles = a.N * a.nth * a.ntw;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tw = t % a.ntw; t /= a.ntw;
        const int th = t % a.nth; t /= a.nth;
        const int n = t, oy0 = th * G::TH, ox0 = tw * G::TW;
        __syncthreads();
        {   // X halo tile: float4 loads (CX % 4 == 0) issued in batches before the LDS writes; odd pixel stride -> scalar writes
            constexpr int XQ = CXP / 4, NIT = (NR * XQ + 255) / 256, BATCH = NIT < 10 ? NIT : 10;
            const bool vec = (a.CX & 3) == 0 && (a.xs & 3) == 0 && (a.x0 & 3) == 0;
#pragma unroll 1
            for (int k0 = 0; k0 < NIT; k0 += BATCH) {
                float4 v[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int i = tid + 256 * (k0 + k);
                    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k0 + k < NIT && i < NR * XQ) {
                        const int px = i / XQ, c0 = 4 * (i % XQ), rx = px % G::RW, ry = px / G::RW;
                        const int iy = oy0 * S + ry - G::P, ix = ox0 * S + rx - G::P;
                        if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
                            const float* __restrict__ src = a.x + (((size_t)n * a.Hi + iy) * a.Wi + ix) * a.xs + a.x0 + c0;
                            if (vec) v[k] = *reinterpret_cast<const float4*>(src);
                            else {
                                if (c0 < a.CX) v[k].x = src[0];
                                if (c0 + 1 < a.CX) v[k].y = src[1];
                                if (c0 + 2 < a.CX) v[k].z = src[2];
                                if (c0 + 3 < a.CX) v[k].w = src[3];
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int i = tid + 256 * (k0 + k);
                    if (k0 + k < NIT && i < NR * XQ) {
                        float* __restrict__ d = &xt[(i / XQ) * XP + 4 * (i % XQ)];
                        d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
                    }
                }
            }
        }
        if ((a.CG & 3) == 0 && (a.gs & 3) == 0 && (a.g0 & 3) == 0) {
            // 16-byte loads of the output-gradient tile (channel quads beyond CG and positions outside the image are zero)
            for (int i = tid; i < NPOS * (CGP / 4); i += 256) {
                const int p = i / (CGP / 4), co = 4 * (i % (CGP / 4)), oy = oy0 + p / G::TW, ox = ox0 + p % G::TW;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (co < a.CG && oy < a.Ho && ox < a.Wo)
                    v = *reinterpret_cast<const float4*>(a.g + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.gs + a.g0 + co);
                *reinterpret_cast<float4*>(&gt[p * CGP + co]) = v;
            }
        } else {
            for (int i = tid; i < NPOS * CGP; i += 256) {
                const int p = i / CGP, co = i % CGP, oy = oy0 + p / G::TW, ox = ox0 + p % G::TW;
                gt[i] = (co < a.CG && oy < a.Ho && ox < a.Wo) ? a.g[(((size_t)n * a.Ho + oy) * a.Wo + ox) * a.gs + a.g0 + co] : 0.f;
            }
        }
        __syncthreads();
        for (int ks = 0; ks < NPOS / 4; ++ks) {
            const int p = 4 * ks + kq;                                    // this lane's position of the k-step
            const int posoff = ((p / G::TW) * S * G::RW + (p % G::TW) * S) * XP;
            float bv[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = gt[p * CGP + nb * 16 + l15];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float av = rowok[m] ? xt[posoff + rowoff[m]] : 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[m][nb] = MVS_MFMA_16x16x4(av, bv[nb], acc[m][nb]);
            }
        }
    }
    // D: column = lane & 15 (co), row = 4 (lane >> 4) + r of the m-tile
    float* __restrict__ out = a.part + (size_t)blockIdx.x * ROWS * CGP;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * (wave + 4 * m) + 4 * kq + r;
            if (row >= ROWS) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) out[(size_t)row * CGP + nb * 16 + l15] = acc[m][nb][r];
        }
}

// gw[co][ci][tap] = sum over the partial images part[p][(tap*CXP + ci)][co]  (fixed order)
// (CX, CG: the slice; the result goes to gw[(co0 + co)][(ci0 + ci)][tap] of a layer with CXT input channels)
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int NT, int CX, int CXP,
                                                                  int CG, int CGP, float* __restrict__ gw, int CXT, int ci0, int co0) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= CG * CX * NT) return;
    const int tap = e % NT, ci = (e / NT) % CX, co = e / (NT * CX);
    const size_t stride = (size_t)NT * CXP * CGP, off = (size_t)(tap * CXP + ci) * CGP + co;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 3 < nparts; p += 4) {
        s0 += part[(size_t)p * stride + off]; s1 += part[(size_t)(p + 1) * stride + off];
        s2 += part[(size_t)(p + 2) * stride + off]; s3 += part[(size_t)(p + 3) * stride + off];
    }
    for (; p < nparts; ++p) s0 += part[(size_t)p * stride + off];
    gw[((size_t)(co0 + co) * CXT + ci0 + ci) * NT + tap] = (s0 + s1) + (s2 + s3);
}

// The same in one WIDE launch (round 4): the kernel above gives every output element ONE thread that walks all partial images --
// 64 dependent rounds of strided loads at 256 images, a ~90 us floor under every layer's weight gradient
// (profiles/r04_run9_conv2d_layers.log: 0.09-0.17 ms against the library's 0.03-0.085).  Here a workgroup owns 16 consecutive
// elements of the partial-image layout (coalesced 64-byte rows); its 16 x 16 threads = (element, slice) walk the images slice,
// slice + 16, ... four loads in flight, then the slices are summed through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_wide_kernel(const float* __restrict__ part, int nparts, int NT, int CX, int CXP,
                                                                       int CG, int CGP, float* __restrict__ gw, int CXT, int ci0, int co0) {
    __shared__ float red[16][17];
    const int n = NT * CXP * CGP;                       // elements of one partial image, padding included
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const size_t stride = (size_t)n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int p = slice;
        for (; p + 48 < nparts; p += 64) {
            const float a0 = part[(size_t)p * stride + e], a1 = part[(size_t)(p + 16) * stride + e];
            const float a2 = part[(size_t)(p + 32) * stride + e], a3 = part[(size_t)(p + 48) * stride + e];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; p < nparts; p += 16) s0 += part[(size_t)p * stride + e];
    }
    red[slice][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && e < n) {
        const int co = e % CGP, row = e / CGP, ci = row % CXP, tap = row / CXP;
        if (co < CG && ci < CX) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][el];
            gw[((size_t)(co0 + co) * CXT + ci0 + ci) * NT + tap] = t;
        }
    }
}

// ---- weight gradients of ALL layers of an extractor in one launch (round 4) ---------------------------------------------------
// The per-layer kernel above (and the library's) pays a launch, a fill of the accumulation buffer and an under-filled GPU for every
// one of FeatureNet's eight small layers: 0.40 ms of library kernels + 13 zero fills per config-2 step for 9.2 GFLOP / 0.25 GB
// (profiles/r04_final_rocprofv3_kernel_stats_c2.csv: igemm_wrw_* + SubTensorOpWithScalar1d).  Here every layer is a range of
// workgroups of ONE launch (the layers' activations and output gradients are all alive at the end of the extractor's backward
// pass), followed by ONE reduction launch.  GEMM view per layer: M = (tap, ci) rows, N = co, K = output positions; a workgroup
// owns `tpw` consecutive 32-wide position tiles, stages the input halo and the output-gradient tile in LDS (dense copies: the LDS
// pixel stride IS the channel count, so a halo row is one contiguous run of global memory) with the next tile's loads in flight
// during the MFMA loop; the four waves split the tile's POSITIONS (and, for the two widest layers, the rows in two halves), so a
// k-step costs MTW + NB LDS reads for MTW * NB MFMAs; partial sums of the waves meet in LDS, one partial image per workgroup.
struct Wg2Layer {
    const float* x;      // [N,Hi,Wi,CX]
    const float* g;      // [N,Ho,Wo,CG]
    float* part;         // [nwg][ROWSP][CGP]
    float* gw;           // [CG][CX][ks][ks], or [CG][ks][ks][CX] when wcl
    int N, Hi, Wi, Ho, Wo, CX, CG;
    int cfg, nth, ntw, ntiles, tpw;   // instantiation, tiles, tiles per workgroup
    int wg0, nwg;                     // this layer's workgroups in the main launch
    int rb0, nrb;                     // this layer's blocks in the reduction launch
    int rowsp, cgp, nt, wcl;
    const float* xstats; // XF kernels, or null: x is the RAW output of the BatchNorm block in front, [group][4][CX] (mean, invstd, scale, shift);
    int ipg;             //   relu(x * scale + shift) of image n's group (n / ipg) is applied when the halo goes to LDS
};
constexpr int WG2_MAX_LAYERS = 8;
struct Wg2Batch {
    Wg2Layer l[WG2_MAX_LAYERS];
    int n;
};

template <int KS_, int S_, int CX_, int NB_, int TH_, int MSPLIT_>
struct Wg2Cfg {
    static constexpr int KS = KS_, S = S_, CX = CX_, NB = NB_, TH = TH_, MSPLIT = MSPLIT_;
    static constexpr int TW = 32, P = KS / 2, RH = (TH - 1) * S + KS, RW = (TW - 1) * S + KS, NT = KS * KS;
    static constexpr int ROWS = NT * CX, MTT = (ROWS + 15) / 16, MTW = (MTT + MSPLIT - 1) / MSPLIT, KW = 4 / MSPLIT;
    // LDS strides: a half-wave's A read covers 16 channels of 2 consecutive positions, its B read 16 output channels of 2
    // positions -> a position stride of 32 floats would put both positions on the same banks: 48 there
    static constexpr int NPOS = TH * TW, XP = CX == 32 ? 48 : CX, CGP = NB * 16, GP = NB == 2 ? 48 : 16;
    static constexpr int XT = (RH * RW * XP + 3) / 4 * 4, GT = NPOS * GP, ROWSP = MTT * 16;
    static constexpr int LDS = XT + GT > ROWSP * CGP ? XT + GT : ROWSP * CGP;
    static constexpr int COST = (NPOS / 4) * MTT * NB + 96;    // MFMAs of a tile + its staging, in MFMA units (host: work split)
};
using Wg2A = Wg2Cfg<3, 1, 3, 1, 8, 1>;     // 3 -> <= 16   (27 rows)
using Wg2B = Wg2Cfg<3, 1, 8, 1, 8, 1>;     // 8 -> <= 16
using Wg2C = Wg2Cfg<5, 2, 8, 1, 4, 2>;     // 8 -> <= 16, 5x5 stride 2 (13 m-tiles: rows in two halves)
using Wg2D = Wg2Cfg<3, 1, 16, 1, 8, 1>;    // 16 -> <= 16
using Wg2E = Wg2Cfg<5, 2, 16, 2, 4, 2>;    // 16 -> <= 32, 5x5 stride 2
using Wg2F = Wg2Cfg<3, 1, 32, 2, 4, 2>;    // 32 -> <= 32
// Two launch classes: one kernel for all six would give every workgroup the widest layers' 250 registers and 72 KB of LDS, i.e. 2
// waves per SIMD, while a tile's MFMA loop (1-2 us) is shorter than the latency of the next tile's loads: the narrow layers --
// which hold 5 of FeatureNet's 8 layers and most of its positions -- ran 3x above their MFMA time (profiles/r04_run28_*).  Class 0
// (A-D: <= 16 input channels) needs ~110 registers and 38 KB.
constexpr int wg2_max(int a, int b) { return a > b ? a : b; }
constexpr int WG2_LDS0 = wg2_max(wg2_max(Wg2A::LDS, Wg2B::LDS), wg2_max(Wg2C::LDS, Wg2D::LDS));
constexpr int WG2_LDS1 = wg2_max(Wg2E::LDS, Wg2F::LDS);
constexpr int WG2_AFF = 512;     // XF kernels: (scale, shift) of every statistics group behind the tile buffers: groups * 2 * CX <= 512 floats
MVS_HD inline int wg2_class(int cfg) { return cfg >= 4 ? 1 : 0; }

template <class C, bool XF>
__device__ __forceinline__ void wg2_body(const Wg2Layer& L, int wgl, float* __restrict__ lds, float* __restrict__ aff) {
    constexpr int KS = C::KS, S = C::S, CX = C::CX, NB = C::NB, TW = C::TW, RW = C::RW, XP = C::XP, GP = C::GP, MTW = C::MTW;
    constexpr bool VEC = CX % 4 == 0;
    constexpr int CQ = VEC ? CX / 4 : 1;
    constexpr int XN = VEC ? C::RH * RW * CQ : C::RH * RW * CX;     // staging items of the halo: float4 / float
    constexpr int XIT = (XN + 255) / 256;
    constexpr int GQ = C::CGP / 4, GN = C::NPOS * GQ, GIT = (GN + 255) / 256;
    float* __restrict__ xt = lds;
    float* __restrict__ gt = lds + C::XT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, l15 = lane & 15;
    const int mg = wave % C::MSPLIT, kg = wave / C::MSPLIT;
    // this wave's m-tiles mg * MTW .. + MTW - 1; row -> (tap = row / CX, ci = row % CX); rows past the layer's read offset 0 and
    // are never written out
    int rowoff[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int row = 16 * (mg * MTW + m) + l15;
        const bool ok = row < C::ROWS;
        const int tap = ok ? row / CX : 0, ci = ok ? row % CX : 0;
        rowoff[m] = ((tap / KS) * RW + tap % KS) * XP + ci;
    }
    f32x4 acc[MTW][NB];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 xv[VEC ? XIT : 1];
    float xs[VEC ? 1 : XIT];
    float4 gv[GIT];
    // the layer's scalars once, in registers: read through the kernel-argument reference they were re-fetched (a scalar load and
    // a wait) in front of every bounds test of the staging loads.  The loads stay CONDITIONAL (zero-initialised register, load
    // under the bounds test): unconditional loads from clamped addresses with the zeroing as a select -- after the load or at the
    // LDS store -- made hipcc keep the prefetched tile in scratch memory and wait for every load ahead of the MFMA loop.
    const float* __restrict__ gx = L.x;
    const float* __restrict__ gg = L.g;
    const int Hi = L.Hi, Wi = L.Wi, Ho = L.Ho, Wo = L.Wo, CG = L.CG, ntw = L.ntw, nth = L.nth;
    const float* __restrict__ xst = XF ? L.xstats : nullptr;
    const int ipg = XF ? L.ipg : 1;
    if (XF && VEC && xst) {
        // (scale, shift) of every group into LDS once: aff[g][2][CX] -- read at the LDS store of a tile (holding the tile's
        // eight values in registers across the MFMA loop put the 32-channel instantiation at 272 registers: one wave per SIMD)
        const int ng = L.N / ipg;
        for (int i = tid; i < ng * 2 * CX; i += 256) aff[i] = xst[(size_t)(i / (2 * CX)) * 4 * CX + 2 * CX + i % (2 * CX)];
    }
    auto load_tile = [&](int tile) {
        int t = tile;
        const int tw = t % ntw; t /= ntw;
        const int th = t % nth;
        const int n = t / nth;
        const int oy0 = th * C::TH, ox0 = tw * TW, iy0 = oy0 * S - C::P, ix0 = ox0 * S - C::P;
        const float* __restrict__ xn = gx + (size_t)n * Hi * Wi * CX;
        const float* __restrict__ gn = gg + (size_t)n * Ho * Wo * CG;
#pragma unroll
        for (int k = 0; k < XIT; ++k) {
            const int i = tid + 256 * k;
            if (VEC) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < XN) {
                    const int px = i / CQ, cq = i % CQ, iy = iy0 + px / RW, ix = ix0 + px % RW;
                    if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi)
                        v = *reinterpret_cast<const float4*>(xn + ((size_t)iy * Wi + ix) * CX + 4 * cq);
                }
                xv[k] = v;
            } else {
                float v = 0.f;
                if (i < XN) {
                    const int ry = i / (RW * CX), rem = i % (RW * CX), iy = iy0 + ry, ix = ix0 + rem / CX;
                    if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) v = xn[((size_t)iy * Wi + ix0) * CX + rem];
                }
                xs[k] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < GIT; ++k) {
            const int i = tid + 256 * k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < GN) {
                const int p = i / GQ, c4 = 4 * (i % GQ), oy = oy0 + p / TW, ox = ox0 + p % TW;
                if (c4 < CG && oy < Ho && ox < Wo) v = *reinterpret_cast<const float4*>(gn + ((size_t)oy * Wo + ox) * CG + c4);
            }
            gv[k] = v;
        }
    };
    auto store_tile = [&](int tile) {
        int t = tile;
        const int tw = t % ntw; t /= ntw;
        const int iy0 = (t % nth) * C::TH * S - C::P, ix0 = tw * TW * S - C::P;
        float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (XF && VEC && xst) {   // this thread's channel quad is the same for all its items (256 % CQ == 0)
            const float* __restrict__ a2 = aff + ((t / nth) / ipg) * 2 * CX + 4 * (tid % CQ);
            xsc = *reinterpret_cast<const float4*>(a2);
            xsh = *reinterpret_cast<const float4*>(a2 + CX);
        }
#pragma unroll
        for (int k = 0; k < XIT; ++k) {
            const int i = tid + 256 * k;
            if (i < XN) {
                if (VEC) {
                    float4 o = xv[k];
                    if (XF && xst) {   // zero padding stays zero: only positions inside the image are normalised
                        const int px = i / CQ, iy = iy0 + px / RW, ix = ix0 + px % RW;
                        if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) {
                            o.x = fmaxf(o.x * xsc.x + xsh.x, 0.f); o.y = fmaxf(o.y * xsc.y + xsh.y, 0.f);
                            o.z = fmaxf(o.z * xsc.z + xsh.z, 0.f); o.w = fmaxf(o.w * xsc.w + xsh.w, 0.f);
                        }
                    }
                    *reinterpret_cast<float4*>(&xt[(i / CQ) * XP + 4 * (i % CQ)]) = o;
                } else xt[i] = xs[k];
            }
        }
#pragma unroll
        for (int k = 0; k < GIT; ++k) {
            const int i = tid + 256 * k;
            if (i < GN) *reinterpret_cast<float4*>(&gt[(i / GQ) * GP + 4 * (i % GQ)]) = gv[k];
        }
    };
    const int tpw = L.tpw, ntiles = L.ntiles;
    const int t0 = wgl * tpw, t1 = t0 + tpw < ntiles ? t0 + tpw : ntiles;
    if (t0 < t1) load_tile(t0);
    for (int tile = t0; tile < t1; ++tile) {
        __syncthreads();            // the previous tile's MFMA loop has read the LDS images
        store_tile(tile);
        __syncthreads();
        if (tile + 1 < t1) load_tile(tile + 1);     // in flight during the MFMA loop
#pragma unroll 2
        for (int j = 0; j < C::NPOS / 4 / C::KW; ++j) {
            const int ks = kg + j * C::KW;
            const int p = 4 * ks + kq;
            const int posoff = ((p / TW) * S * RW + (p % TW) * S) * XP;
            float bv[NB], av[MTW];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = gt[p * GP + nb * 16 + l15];
#pragma unroll
            for (int m = 0; m < MTW; ++m) av[m] = xt[posoff + rowoff[m]];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[m][nb] = MVS_MFMA_16x16x4(av[m], bv[nb], acc[m][nb]);
        }
    }
    // the waves' partial sums meet in LDS (k-group after k-group: fixed order), then one coalesced partial image per workgroup
    // D: column = lane & 15 (co), row = 4 (lane >> 4) + r of the m-tile
    float* __restrict__ red = lds;
    for (int k = 0; k < C::KW; ++k) {
        __syncthreads();
        if (kg == k) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int mt = mg * MTW + m;
                if (mt >= C::MTT) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* __restrict__ d = &red[(16 * mt + 4 * kq + r) * C::CGP + nb * 16 + l15];
                        *d = k == 0 ? acc[m][nb][r] : *d + acc[m][nb][r];
                    }
            }
        }
    }
    __syncthreads();
    float4* __restrict__ out = reinterpret_cast<float4*>(L.part + (size_t)wgl * C::ROWSP * C::CGP);
    for (int i = tid; i < C::ROWSP * C::CGP / 4; i += 256) out[i] = reinterpret_cast<const float4*>(red)[i];
}

template <int CLS, bool XF>
__device__ __forceinline__ void wg2_dispatch(const Wg2Batch& b, float* __restrict__ lds) {
    int li = -1;
    for (int i = 0; i < b.n; ++i)
        if (wg2_class(b.l[i].cfg) == CLS && (int)blockIdx.x >= b.l[i].wg0 && (int)blockIdx.x < b.l[i].wg0 + b.l[i].nwg) li = i;
    if (li < 0) return;
    const Wg2Layer& L = b.l[li];
    const int wgl = blockIdx.x - L.wg0;
    float* __restrict__ aff = lds + (CLS == 0 ? WG2_LDS0 : WG2_LDS1);
    if (CLS == 0) {
        switch (L.cfg) {
            case 0: wg2_body<Wg2A, false>(L, wgl, lds, aff); break;     // 3 input channels: an image, never a normalised tensor
            case 1: wg2_body<Wg2B, XF>(L, wgl, lds, aff); break;
            case 2: wg2_body<Wg2C, XF>(L, wgl, lds, aff); break;
            default: wg2_body<Wg2D, XF>(L, wgl, lds, aff); break;
        }
    } else {
        if (L.cfg == 4) wg2_body<Wg2E, XF>(L, wgl, lds, aff);
        else wg2_body<Wg2F, XF>(L, wgl, lds, aff);
    }
}

template <int CLS>
__global__ __launch_bounds__(256) void conv2d_wgrad_batch_kernel(Wg2Batch b) {
    __shared__ __attribute__((aligned(16))) float lds[CLS == 0 ? WG2_LDS0 : WG2_LDS1];
    wg2_dispatch<CLS, false>(b, lds);
}
// the normalising (XF) forms: their own kernels, so that the default ones keep their code and registers; the wide class is capped
// at 256 registers (two waves per SIMD: left alone it takes 260)
__global__ __launch_bounds__(256) void conv2d_wgrad_batch_xf0_kernel(Wg2Batch b) {
    __shared__ __attribute__((aligned(16))) float lds[WG2_LDS0 + WG2_AFF];
    wg2_dispatch<0, true>(b, lds);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv2d_wgrad_batch_xf1_kernel(Wg2Batch b) {
    __shared__ __attribute__((aligned(16))) float lds[WG2_LDS1 + WG2_AFF];
    wg2_dispatch<1, true>(b, lds);
}

// gw = sum over the layer's partial images, fixed order: a workgroup owns 16 consecutive elements of the partial-image layout, its
// 16 x 16 threads = (element, slice) walk the images slice, slice + 16, ... (as conv2d_wgrad_reduce_wide_kernel), all layers in
// one launch
__global__ __launch_bounds__(256) void conv2d_wgrad_batch_reduce_kernel(Wg2Batch b) {
    __shared__ float red[16][17];
    int li = 0;
    while (li + 1 < b.n && (int)blockIdx.x >= b.l[li].rb0 + b.l[li].nrb) ++li;
    const Wg2Layer& L = b.l[li];
    const int n = L.rowsp * L.cgp;
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = ((int)blockIdx.x - L.rb0) * 16 + el;
    const size_t stride = (size_t)n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        const float* __restrict__ part = L.part;
        int p = slice;
        for (; p + 48 < L.nwg; p += 64) {
            const float a0 = part[(size_t)p * stride + e], a1 = part[(size_t)(p + 16) * stride + e];
            const float a2 = part[(size_t)(p + 32) * stride + e], a3 = part[(size_t)(p + 48) * stride + e];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; p < L.nwg; p += 16) s0 += part[(size_t)p * stride + e];
    }
    red[slice][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && e < n) {
        const int co = e % L.cgp, row = e / L.cgp, ci = row % L.CX, tap = row / L.CX;
        if (co < L.CG && tap < L.nt) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][el];
            L.gw[L.wcl ? ((size_t)co * L.nt + tap) * L.CX + ci : ((size_t)co * L.CX + ci) * L.nt + tap] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static const int C2_WGRAD_GROUPS = 1024;  // partial images the workspace holds
int g_conv2d_wgrad_groups = 256;          // tuning knob "wgrad2d_groups" (<= 1024): persistent workgroups of the weight gradient (256 = one per CU;
                                          // more let a CU overlap one workgroup's tile staging with another's MFMA loop -- not yet measured)
int g_conv2d_pp = 1;        // tuning knob "conv2d_pp": 3x3 stride-1 layers with <= 8 output channels as pixel-pair GEMMs (conv2d_igemm_kernel<.., PP>)
int g_conv2d_s2_mfma = 2;   // tuning knob "conv2d_s2_mfma": stride-2 input gradient as ONE four-class MFMA pass with compacted taps (2, round 6), four parity-class passes (1), direct VALU form (0)

static bool c2_shape_ok(int ks, int stride) { return (ks == 3 && stride == 1) || (ks == 5 && stride == 2); }
static int c2_cc(int ks, int cin) { return ks == 5 ? 8 : (cin <= 4 ? 4 : (cin <= 8 ? 8 : (cin <= 16 ? 16 : 32))); }

extern "C" long long mvs_conv2d_workspace_floats(int op, int N, int H, int W, int Cin, int Cout, int ks, int stride) {
    if (!c2_shape_ok(ks, stride) || Cin < 1 || Cin > 64 || Cout < 1 || Cout > 64) return -1;
    const int nt = ks * ks;
    if (op == 2) {   // one <= 32 x <= 32 channel slice at a time
        const int cxs = Cin > 32 ? 32 : (Cin + 3) / 4 * 4, cgs = Cout > 32 ? 32 : (Cout + 15) / 16 * 16;
        // one partial image per persistent workgroup the launch will actually use (knob "wgrad2d_groups", evaluated now: the
        // caller queries the size right before the call), not the 1024-image ceiling (37.7 MB instead of 9.4 MB per 32x32 layer)
        const int gq = g_conv2d_wgrad_groups < 1 ? 1 : (g_conv2d_wgrad_groups > C2_WGRAD_GROUPS ? C2_WGRAD_GROUPS : g_conv2d_wgrad_groups);
        return (long long)gq * nt * cxs * cgs;
    }
    const int ci = op == 1 ? Cout : Cin, co = op == 1 ? Cin : Cout;   // an input gradient is a forward-style pass on gy
    if (op == 1 && stride == 2) {                                      // four parity-class 3x3 weight images
        const int cc3 = c2_cc(3, ci);
        return 4LL * ((ci + cc3 - 1) / cc3) * c2_ksteps(9, cc3) * ((co + 15) / 16) * 256;
    }
    const int cc = c2_cc(ks, ci), nch = (ci + cc - 1) / cc;
    const int ntk = (ks == 3 && co <= 8) ? 12 : nt;   // the pixel-pair form of narrow layers walks 12 taps
    return (long long)nch * c2_ksteps(ntk, cc) * ((co + 15) / 16) * 256;
}

template <int KS, int S, int CC>
static void c2_launch(const Conv2dArgs& a, int nb, dim3 grid, hipStream_t st) {
    if (a.slots && a.bn_raw) {
        if (nb == 1) MVS_LAUNCH((conv2d_igemm_kernel<KS, S, CC, 1, false, true, false, true>), grid, dim3(256), 0, st, a);
        else MVS_LAUNCH((conv2d_igemm_kernel<KS, S, CC, 2, false, true, false, true>), grid, dim3(256), 0, st, a);
        return;
    }
    if (a.slots && a.in_stats) {
        if (nb == 1) MVS_LAUNCH((conv2d_igemm_kernel<KS, S, CC, 1, false, true, true>), grid, dim3(256), 0, st, a);
        else MVS_LAUNCH((conv2d_igemm_kernel<KS, S, CC, 2, false, true, true>), grid, dim3(256), 0, st, a);
        return;
    }
    if (a.slots) {
        if (nb == 1) MVS_LAUNCH((conv2d_igemm_kernel<KS, S, CC, 1, false, true>), grid, dim3(256), 0, st, a);
        else MVS_LAUNCH((conv2d_igemm_kernel<KS, S, CC, 2, false, true>), grid, dim3(256), 0, st, a);
        return;
    }
    if (nb == 1) MVS_LAUNCH((conv2d_igemm_kernel<KS, S, CC, 1>), grid, dim3(256), 0, st, a);
    else MVS_LAUNCH((conv2d_igemm_kernel<KS, S, CC, 2>), grid, dim3(256), 0, st, a);
}

// the forward weight image of a layer: which form (pixel pairs or plain) and how large
static void c2_fwd_pack_plan(int Cin, int Cout, int ks, int stride, Pack2dItem& it) {
    const int cc = c2_cc(ks, Cin), nt = ks * ks, nch = mvs_cdiv(Cin, cc);
    it.NT = nt; it.CC = cc; it.Cin = Cin; it.Cout = Cout;
    if (ks == 3 && stride == 1 && Cout <= 8 && (cc == 4 || cc == 8) && g_conv2d_pp) {
        it.pp = 1; it.NB = 1; it.total = nch * c2_ksteps(12, cc) * 256;
    } else {
        it.pp = 0; it.NB = mvs_cdiv(Cout, 16); it.total = nch * c2_ksteps(nt, cc) * it.NB * 256;
    }
}

static int c2_run_igemm(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int Hi, int Wi, int Cin,
                        int Cout, int ks, int stride, int transposed, hipStream_t st, int act = 0, float slope = 0.f,
                        double* slots = nullptr, int nslots = 0, int imgs_per_group = 1, int ws_packed = 0,
                        const float* in_stats = nullptr, const float* bn_raw = nullptr, const float* bn_stats = nullptr, int wcl = 0) {
    Conv2dArgs a = {};
    a.bn_raw = bn_raw; a.bn_stats = bn_stats;
    a.act = act; a.slope = slope; a.slots = slots; a.nslots = nslots; a.imgs_per_group = imgs_per_group; a.in_stats = in_stats;
    a.x = x; a.bias = bias; a.y = y; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout;
    a.Ho = stride == 1 ? Hi : (Hi - 1) / 2 + 1; a.Wo = stride == 1 ? Wi : (Wi - 1) / 2 + 1;
    a.os = 1; a.py = 0; a.px = 0; a.YH = a.Ho; a.YW = a.Wo;
    a.nth = mvs_cdiv(a.Ho, 8); a.ntw = mvs_cdiv(a.Wo, 32);
    const int cc = c2_cc(ks, Cin), nt = ks * ks, nch = mvs_cdiv(Cin, cc);
    a.nb_total = mvs_cdiv(Cout, 16);
    if (ks == 3 && stride == 1 && Cout <= 8 && (cc == 4 || cc == 8) && g_conv2d_pp) {
        // narrow layers (3 -> 8, 8 -> 8 of FeatureNet): pixel pairs fill the MFMA's 16 columns (knob "conv2d_pp")
        const int totalp = nch * c2_ksteps(12, cc) * 256;
        if (!ws_packed)
            MVS_LAUNCH(conv2d_pack_kernel, dim3(mvs_cdiv(totalp, 256)), dim3(256), 0, st, w, ws, nt, cc, Cin, Cout, 1, transposed, totalp, -1, 1, wcl);
        a.wp = ws;
        dim3 gridp(N * a.nth * a.ntw, 1);
        if (slots && bn_raw) {
            if (cc == 4) MVS_LAUNCH((conv2d_igemm_kernel<3, 1, 4, 1, true, true, false, true>), gridp, dim3(256), 0, st, a);
            else MVS_LAUNCH((conv2d_igemm_kernel<3, 1, 8, 1, true, true, false, true>), gridp, dim3(256), 0, st, a);
        } else if (slots && in_stats) {
            MVS_REQUIRE(cc == 8, MVS_ERR_UNSUPPORTED, "conv2d: a normalised input has a multiple of 4 channels, got %d", Cin);
            MVS_LAUNCH((conv2d_igemm_kernel<3, 1, 8, 1, true, true, true>), gridp, dim3(256), 0, st, a);
        } else if (slots) {
            if (cc == 4) MVS_LAUNCH((conv2d_igemm_kernel<3, 1, 4, 1, true, true>), gridp, dim3(256), 0, st, a);
            else MVS_LAUNCH((conv2d_igemm_kernel<3, 1, 8, 1, true, true>), gridp, dim3(256), 0, st, a);
        } else if (cc == 4) MVS_LAUNCH((conv2d_igemm_kernel<3, 1, 4, 1, true>), gridp, dim3(256), 0, st, a);
        else MVS_LAUNCH((conv2d_igemm_kernel<3, 1, 8, 1, true>), gridp, dim3(256), 0, st, a);
        return mvs_check_launch("conv2d_igemm (pixel pairs)");
    }
    const int total = nch * c2_ksteps(nt, cc) * a.nb_total * 256;
    if (!ws_packed)
        MVS_LAUNCH(conv2d_pack_kernel, dim3(mvs_cdiv(total, 256)), dim3(256), 0, st, w, ws, nt, cc, Cin, Cout, a.nb_total, transposed, total, -1, 0, wcl);
    a.wp = ws;
    const int nb = a.nb_total <= 2 ? a.nb_total : 2;   // 16-wide Cout tiles per workgroup; the rest over blockIdx.y
    dim3 grid(N * a.nth * a.ntw, mvs_cdiv(a.nb_total, nb));
    MVS_REQUIRE(a.nb_total % nb == 0, MVS_ERR_UNSUPPORTED, "conv2d: %d output channels not supported (1..32, 33..64 in steps of 32)", Cout);
    if (ks == 5) c2_launch<5, 2, 8>(a, nb, grid, st);
    else if (cc == 4) c2_launch<3, 1, 4>(a, nb, grid, st);
    else if (cc == 8) c2_launch<3, 1, 8>(a, nb, grid, st);
    else if (cc == 16) c2_launch<3, 1, 16>(a, nb, grid, st);
    else c2_launch<3, 1, 32>(a, nb, grid, st);
    return mvs_check_launch("conv2d_igemm");
}

static int c2_check(const char* what, int N, int H, int W, int Cin, int Cout, int ks, int stride) {
    MVS_REQUIRE(c2_shape_ok(ks, stride), MVS_ERR_UNSUPPORTED, "%s: supports 3x3 stride 1 and 5x5 stride 2, got %dx%d stride %d", what, ks, ks, stride);
    MVS_REQUIRE(N > 0 && H > 0 && W > 0, MVS_ERR_SHAPE, "%s: bad shape N=%d H=%d W=%d", what, N, H, W);
    MVS_REQUIRE(Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 64, MVS_ERR_UNSUPPORTED, "%s: channels must be 1..64, got %d -> %d", what, Cin, Cout);
    MVS_REQUIRE((Cin <= 32 || Cin == 64) && (Cout <= 32 || Cout == 64), MVS_ERR_UNSUPPORTED,
                "%s: more than 32 channels means exactly 64 (the feature pyramid's widths), got %d -> %d", what, Cin, Cout);
    return MVS_OK;
}

// x [N,H,W,Cin] channels-last, w [Cout][Cin][ks][ks], y [N,Ho,Wo,Cout]; pad = ks/2
extern "C" int mvs_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W,
                              int Cin, int Cout, int ks, int stride, hipStream_t stream) {
    int rc = c2_check("conv2d_fwd", N, H, W, Cin, Cout, ks, stride);
    if (rc) return rc;
    MVS_REQUIRE(x && w && y && ws, MVS_ERR_NULL, "conv2d_fwd: null pointer argument");
    return c2_run_igemm(x, w, bias, y, ws, N, H, W, Cin, Cout, ks, stride, 0, stream);
}
// the same with the parameter tensor channels-last in memory ([Cout][ks][ks][Cin]) when w_channels_last: read in place
extern "C" int mvs_conv2d_fwd_wl(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin, int Cout,
                                 int ks, int stride, int w_channels_last, hipStream_t stream) {
    int rc = c2_check("conv2d_fwd", N, H, W, Cin, Cout, ks, stride);
    if (rc) return rc;
    MVS_REQUIRE(x && w && y && ws, MVS_ERR_NULL, "conv2d_fwd: null pointer argument");
    return c2_run_igemm(x, w, bias, y, ws, N, H, W, Cin, Cout, ks, stride, 0, stream, 0, 0.f, nullptr, 0, 1, 0, nullptr, nullptr, nullptr,
                        w_channels_last ? 1 : 0);
}

// Forward without bias that also adds BatchNorm's statistics of its output into slots [G][nslots][2][Cout] (fp64, zeroed by the
// caller; nslots a power of two, e.g. mvs_bn_slots(Cout)): the N images are G statistics groups of N/G consecutive images (the
// views of a sample through the shared-weight extractor).  ConvBnReLU of the 2-D extractor in training (module.py:15-22);
// consumer: mvs_bn_relu_fwd_slots.
// ws_packed = 1: ws already holds the layer's forward weight image (mvs_conv2d_pack_weights_batch); w is not read then.
extern "C" int mvs_conv2d_fwd_stats(const float* x, const float* w, float* y, float* ws, double* slots, int nslots, int G, int N,
                                    int H, int W, int Cin, int Cout, int ks, int stride, int ws_packed, hipStream_t stream) {
    int rc = c2_check("conv2d_fwd_stats", N, H, W, Cin, Cout, ks, stride);
    if (rc) return rc;
    MVS_REQUIRE(x && (w || ws_packed) && y && ws && slots, MVS_ERR_NULL, "conv2d_fwd_stats: null pointer argument");
    MVS_REQUIRE(G >= 1 && N % G == 0 && nslots >= 1 && nslots <= 256 && (nslots & (nslots - 1)) == 0, MVS_ERR_SHAPE,
                "conv2d_fwd_stats: %d images do not split into %d groups, or bad slot count %d", N, G, nslots);
    return c2_run_igemm(x, w, nullptr, y, ws, N, H, W, Cin, Cout, ks, stride, 0, stream, 0, 0.f, slots, nslots, N / G, ws_packed);
}

// The same with x = the RAW output of the BatchNorm + ReLU block in front (its statistics finished by mvs_bn_finalize_slots into
// in_stats [G][4][Cin]): relu(x * scale + shift) of x's own group is applied while the halo tile is staged -- the producing block
// needs no apply pass (jdacs/models/module.py:21-22 `F.relu(self.bn(self.conv(x)))` of block i fused into block i+1's `self.conv`).
extern "C" int mvs_conv2d_fwd_stats_xf(const float* x, const float* in_stats, const float* w, float* y, float* ws, double* slots,
                                       int nslots, int G, int N, int H, int W, int Cin, int Cout, int ks, int stride, int ws_packed,
                                       hipStream_t stream) {
    int rc = c2_check("conv2d_fwd_stats_xf", N, H, W, Cin, Cout, ks, stride);
    if (rc) return rc;
    MVS_REQUIRE(x && in_stats && (w || ws_packed) && y && ws && slots, MVS_ERR_NULL, "conv2d_fwd_stats_xf: null pointer argument");
    MVS_REQUIRE(G >= 1 && N % G == 0 && nslots >= 1 && nslots <= 256 && (nslots & (nslots - 1)) == 0, MVS_ERR_SHAPE,
                "conv2d_fwd_stats_xf: %d images do not split into %d groups, or bad slot count %d", N, G, nslots);
    MVS_REQUIRE((Cin & 3) == 0, MVS_ERR_UNSUPPORTED, "conv2d_fwd_stats_xf: input channels must be a multiple of 4, got %d", Cin);
    return c2_run_igemm(x, w, nullptr, y, ws, N, H, W, Cin, Cout, ks, stride, 0, stream, 0, 0.f, slots, nslots, N / G, ws_packed, in_stats);
}

// Forward weight images of n layers in ONE launch: w[n] (parameter tensors [Cout][Cin][ks][ks], or channels-last in memory when
// w_channels_last[i]), ws[n] (each >= mvs_conv2d_workspace_floats(0, ...)), shapes[n][4] = Cin, Cout, ks, stride.
extern "C" int mvs_conv2d_pack_weights_batch(int n, const float* const* w, float* const* ws, const int* shapes, const int* w_channels_last,
                                             hipStream_t stream) {
    MVS_REQUIRE(w && ws && shapes && w_channels_last, MVS_ERR_NULL, "conv2d_pack_weights_batch: null pointer argument");
    MVS_REQUIRE(n >= 0 && n <= 1024, MVS_ERR_SHAPE, "conv2d_pack_weights_batch: bad count %d", n);
    for (int i0 = 0; i0 < n; i0 += MVS_PACK2D_BATCH_MAX) {
        Pack2dBatch pb = {};
        const int m = n - i0 < MVS_PACK2D_BATCH_MAX ? n - i0 : MVS_PACK2D_BATCH_MAX;
        int maxtotal = 0;
        for (int i = 0; i < m; ++i) {
            const int* sh = shapes + (size_t)(i0 + i) * 4;
            int rc = c2_check("conv2d_pack_weights_batch", 1, 1, 1, sh[0], sh[1], sh[2], sh[3]);
            if (rc) return rc;
            MVS_REQUIRE(w[i0 + i] && ws[i0 + i], MVS_ERR_NULL, "conv2d_pack_weights_batch: null pointer in entry %d", i0 + i);
            Pack2dItem& it = pb.it[i];
            it.w = w[i0 + i]; it.wp = ws[i0 + i]; it.wcl = w_channels_last[i0 + i] ? 1 : 0;
            c2_fwd_pack_plan(sh[0], sh[1], sh[2], sh[3], it);
            if (it.total > maxtotal) maxtotal = it.total;
        }
        if (m > 0) MVS_LAUNCH(conv2d_pack_batch_kernel, dim3(mvs_cdiv(maxtotal, 256), m), dim3(256), 0, stream, pb);
    }
    return mvs_check_launch("conv2d_pack_weights_batch");
}

// the same followed by LeakyReLU(negative_slope): the `conv` block of the feature pyramid (jdacs-ms/models/modules.py:15-19,
// jdacs-ms/models/network.py:16-41: 3 -> 64 -> 64 -> 64 -> 32 -> 32 -> 32 -> 16 -> 16 -> 16, bias, slope 0.1) in one pass
extern "C" int mvs_conv2d_lrelu_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W,
                                    int Cin, int Cout, int ks, int stride, float negative_slope, hipStream_t stream) {
    int rc = c2_check("conv2d_lrelu_fwd", N, H, W, Cin, Cout, ks, stride);
    if (rc) return rc;
    MVS_REQUIRE(x && w && y && ws, MVS_ERR_NULL, "conv2d_lrelu_fwd: null pointer argument");
    return c2_run_igemm(x, w, bias, y, ws, N, H, W, Cin, Cout, ks, stride, 0, stream, 1, negative_slope);
}

// gx [N,H,W,Cin] from gy [N,Ho,Wo,Cout]
// Input gradient of a 3x3 stride-1 layer whose gx is the COMPLETE output gradient of the BatchNorm + ReLU block in front of it
// (jdacs/models/module.py:21-22; bn_raw = that block's raw output [N,H,W,Cin], bn_stats [G][4][Cin]): the epilogue also adds the
// block's backward statistics (sum dyh, sum dyh * xhat per channel and group of N / G images) into bn_slots [G][nslots][2][Cin]
// (fp64 atomics) -- mvs_bn_relu_bwd_slots then runs without its reduce pass.
extern "C" int mvs_conv2d_dgrad_bnstats(const float* gy, const float* w, float* gx, float* ws, int N, int H, int W, int Cin, int Cout,
                                        int ks, const float* bn_raw, const float* bn_stats, double* bn_slots, int nslots, int G,
                                        hipStream_t stream) {
    int rc = c2_check("conv2d_dgrad_bnstats", N, H, W, Cin, Cout, ks, 1);
    if (rc) return rc;
    MVS_REQUIRE(gy && w && gx && ws && bn_raw && bn_stats && bn_slots, MVS_ERR_NULL, "conv2d_dgrad_bnstats: null pointer argument");
    MVS_REQUIRE(ks == 3, MVS_ERR_UNSUPPORTED, "conv2d_dgrad_bnstats: 3x3 stride-1 layers only");
    MVS_REQUIRE(G >= 1 && N % G == 0 && nslots >= 1 && nslots <= 256 && (nslots & (nslots - 1)) == 0, MVS_ERR_SHAPE,
                "conv2d_dgrad_bnstats: %d images do not split into %d groups, or bad slot count %d", N, G, nslots);
    return c2_run_igemm(gy, w, nullptr, gx, ws, N, H, W, Cout, Cin, ks, 1, 1, stream, 0, 0.f, bn_slots, nslots, N / G, 0, nullptr, bn_raw,
                        bn_stats);
}

extern "C" int mvs_conv2d_dgrad_wl(const float* gy, const float* w, float* gx, float* ws, int N, int H, int W, int Cin, int Cout, int ks,
                                   int stride, int w_channels_last, hipStream_t stream);
extern "C" int mvs_conv2d_dgrad(const float* gy, const float* w, float* gx, float* ws, int N, int H, int W, int Cin, int Cout,
                                int ks, int stride, hipStream_t stream) {
    return mvs_conv2d_dgrad_wl(gy, w, gx, ws, N, H, W, Cin, Cout, ks, stride, 0, stream);
}
// the same with the parameter tensor channels-last in memory ([Cout][ks][ks][Cin]: what module.to(memory_format=torch.channels_last) makes of
// an nn.Conv2d weight) when w_channels_last -- read in place by the weight-image pack (rounds 2-5 needed a contiguous copy per layer and step)
extern "C" int mvs_conv2d_dgrad_wl(const float* gy, const float* w, float* gx, float* ws, int N, int H, int W, int Cin, int Cout, int ks,
                                   int stride, int w_channels_last, hipStream_t stream) {
    int rc = c2_check("conv2d_dgrad", N, H, W, Cin, Cout, ks, stride);
    if (rc) return rc;
    MVS_REQUIRE(gy && w && gx && ws, MVS_ERR_NULL, "conv2d_dgrad: null pointer argument");
    w_channels_last = w_channels_last ? 1 : 0;
    if (stride == 1) return c2_run_igemm(gy, w, nullptr, gx, ws, N, H, W, Cout, Cin, ks, 1, 1, stream, 0, 0.f, nullptr, 0, 1, 0, nullptr, nullptr, nullptr, w_channels_last);
    if (g_conv2d_s2_mfma == 2) {
        // round 6: the four parity classes in ONE pass (class = blockIdx.z) behind ONE pack launch, compacted taps (25 tap slices, not 36)
        const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
        const int cc = c2_cc(3, Cout), nch = mvs_cdiv(Cout, cc), nbt = mvs_cdiv(Cin, 16);
        MVS_REQUIRE(nbt <= 2, MVS_ERR_UNSUPPORTED, "conv2d_dgrad: the 5x5 stride-2 layers have <= 32 input channels");
        Conv2dArgs a = {};
        a.x = gy; a.bias = nullptr; a.y = gx; a.N = N; a.Hi = Ho; a.Wi = Wo; a.Cin = Cout; a.Cout = Cin;
        a.os = 2; a.py = 0; a.px = 0; a.YH = H; a.YW = W;
        a.Ho = mvs_cdiv(H, 2); a.Wo = mvs_cdiv(W, 2);                               // grid points of the even class (the largest)
        a.nth = mvs_cdiv(a.Ho, 8); a.ntw = mvs_cdiv(a.Wo, 32); a.nb_total = nbt;
        a.wp = ws;
        MVS_LAUNCH(conv2d_pack_s2d_kernel, dim3(mvs_cdiv(c2_s2d_floats(0, cc, nch, nbt), 256), 4), dim3(256), 0, stream, w, ws, cc, Cout, Cin, nbt, nch, w_channels_last);
        dim3 grid(N * a.nth * a.ntw, 1, 4);
#define MVS_S2D_CASE(CCV)                                                                                                           \
    if (nbt == 1) MVS_LAUNCH((conv2d_igemm_kernel<3, 1, CCV, 1, false, false, false, false, true>), grid, dim3(256), 0, stream, a);  \
    else MVS_LAUNCH((conv2d_igemm_kernel<3, 1, CCV, 2, false, false, false, false, true>), grid, dim3(256), 0, stream, a);
        if (cc == 8) { MVS_S2D_CASE(8) } else if (cc == 16) { MVS_S2D_CASE(16) } else if (cc == 32) { MVS_S2D_CASE(32) } else { MVS_S2D_CASE(4) }
#undef MVS_S2D_CASE
        return mvs_check_launch("conv2d_dgrad_s2_one_pass");
    }
    if (g_conv2d_s2_mfma) {
        // four parity classes, each a 3x3 stride-1 pass over gy on the coarse grid with its own (partly empty) weight image
        const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
        const int cc = c2_cc(3, Cout), nch = mvs_cdiv(Cout, cc), nbt = mvs_cdiv(Cin, 16);
        const int total_w = nch * c2_ksteps(9, cc) * nbt * 256;
        for (int cls = 0; cls < 4; ++cls) {
            Conv2dArgs a = {};
            a.x = gy; a.bias = nullptr; a.y = gx; a.N = N; a.Hi = Ho; a.Wi = Wo; a.Cin = Cout; a.Cout = Cin;
            a.os = 2; a.py = cls >> 1; a.px = cls & 1; a.YH = H; a.YW = W;
            a.Ho = mvs_cdiv(H - a.py, 2); a.Wo = mvs_cdiv(W - a.px, 2);          // grid points of this class
            if (a.Ho <= 0 || a.Wo <= 0) continue;
            a.nth = mvs_cdiv(a.Ho, 8); a.ntw = mvs_cdiv(a.Wo, 32); a.nb_total = nbt;
            float* wpc = ws + (size_t)cls * total_w;
            MVS_LAUNCH(conv2d_pack_kernel, dim3(mvs_cdiv(total_w, 256)), dim3(256), 0, stream, w, wpc, 9, cc, Cout, Cin, nbt, 0, total_w, cls, 0, w_channels_last);
            a.wp = wpc;
            MVS_REQUIRE(nbt <= 2, MVS_ERR_UNSUPPORTED, "conv2d_dgrad: the 5x5 stride-2 layers have <= 32 input channels");
            dim3 grid(N * a.nth * a.ntw, 1);
            if (cc == 8) c2_launch<3, 1, 8>(a, nbt, grid, stream);
            else if (cc == 16) c2_launch<3, 1, 16>(a, nbt, grid, stream);
            else if (cc == 32) c2_launch<3, 1, 32>(a, nbt, grid, stream);
            else c2_launch<3, 1, 4>(a, nbt, grid, stream);
        }
        return mvs_check_launch("conv2d_dgrad_s2_classes");
    }
    MVS_REQUIRE(!w_channels_last, MVS_ERR_UNSUPPORTED, "conv2d_dgrad: the direct stride-2 form (knob conv2d_s2_mfma = 0) reads a contiguous weight");
    const size_t total = (size_t)N * H * W * Cin;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    MVS_LAUNCH((conv2d_dgrad_s2_kernel<5>), dim3(blocks), dim3(256), 0, stream, gy, w, gx, N, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, Cin, Cout);
    return mvs_check_launch("conv2d_dgrad_s2");
}

template <int KS, int S, int CXP, int NB>
static void c2_wgrad_launch(const Wgrad2dArgs& a, int groups, hipStream_t st) {
    constexpr int MT = (KS * KS * CXP + 63) / 64;
    MVS_LAUNCH((conv2d_wgrad_kernel<KS, S, CXP, MT, NB>), dim3(groups), dim3(256), 0, st, a);
}

// gw [Cout][Cin][ks][ks]
extern "C" int mvs_conv2d_wgrad(const float* x, const float* gy, float* gw, float* ws, int N, int H, int W, int Cin, int Cout,
                                int ks, int stride, hipStream_t stream) {
    int rc = c2_check("conv2d_wgrad", N, H, W, Cin, Cout, ks, stride);
    if (rc) return rc;
    MVS_REQUIRE(x && gy && gw && ws, MVS_ERR_NULL, "conv2d_wgrad: null pointer argument");
    const int Ho = stride == 1 ? H : (H - 1) / 2 + 1, Wo = stride == 1 ? W : (W - 1) / 2 + 1;
    const int gmax = g_conv2d_wgrad_groups < 1 ? 1 : (g_conv2d_wgrad_groups > C2_WGRAD_GROUPS ? C2_WGRAD_GROUPS : g_conv2d_wgrad_groups);
    const int nt = ks * ks;
    // layers wider than 32 channels (the feature pyramid's 64) run as <= 32 x <= 32 channel slices, one after the other
    for (int ci0 = 0; ci0 < Cin; ci0 += 32)
        for (int co0 = 0; co0 < Cout; co0 += 32) {
            Wgrad2dArgs a = {};
            a.x = x; a.g = gy; a.part = ws; a.N = N; a.Hi = H; a.Wi = W; a.Ho = Ho; a.Wo = Wo;
            a.CX = Cin - ci0 < 32 ? Cin - ci0 : 32; a.CG = Cout - co0 < 32 ? Cout - co0 : 32;
            a.xs = Cin; a.x0 = ci0; a.gs = Cout; a.g0 = co0;
            a.nth = mvs_cdiv(a.Ho, 8); a.ntw = mvs_cdiv(a.Wo, 32);
            const int ntiles = N * a.nth * a.ntw, groups = ntiles < gmax ? ntiles : gmax;
            const int cxp = (a.CX + 3) / 4 * 4, nb = mvs_cdiv(a.CG, 16);
            bool ok = true;
            if (ks == 3) {
                if (cxp == 4) { if (nb == 1) c2_wgrad_launch<3, 1, 4, 1>(a, groups, stream); else c2_wgrad_launch<3, 1, 4, 2>(a, groups, stream); }
                else if (cxp == 8) { if (nb == 1) c2_wgrad_launch<3, 1, 8, 1>(a, groups, stream); else c2_wgrad_launch<3, 1, 8, 2>(a, groups, stream); }
                else if (cxp == 16) { if (nb == 1) c2_wgrad_launch<3, 1, 16, 1>(a, groups, stream); else c2_wgrad_launch<3, 1, 16, 2>(a, groups, stream); }
                else if (cxp == 32) { if (nb == 1) c2_wgrad_launch<3, 1, 32, 1>(a, groups, stream); else c2_wgrad_launch<3, 1, 32, 2>(a, groups, stream); }
                else ok = false;
            } else {
                if (cxp == 8) { if (nb == 1) c2_wgrad_launch<5, 2, 8, 1>(a, groups, stream); else c2_wgrad_launch<5, 2, 8, 2>(a, groups, stream); }
                else if (cxp == 16) { if (nb == 1) c2_wgrad_launch<5, 2, 16, 1>(a, groups, stream); else c2_wgrad_launch<5, 2, 16, 2>(a, groups, stream); }
                else ok = false;
            }
            MVS_REQUIRE(ok, MVS_ERR_UNSUPPORTED, "conv2d_wgrad: input channels %d not supported for the %dx%d layer", Cin, ks, ks);
            rc = mvs_check_launch("conv2d_wgrad");
            if (rc) return rc;
            const int n = a.CG * a.CX * nt;
            if (groups > 16)
                MVS_LAUNCH(conv2d_wgrad_reduce_wide_kernel, dim3(mvs_cdiv(nt * cxp * nb * 16, 16)), dim3(256), 0, stream, (const float*)ws, groups, nt,
                           a.CX, cxp, a.CG, nb * 16, gw, Cin, ci0, co0);
            else
                MVS_LAUNCH(conv2d_wgrad_reduce_kernel, dim3(mvs_cdiv(n, 256)), dim3(256), 0, stream, (const float*)ws, groups, nt, a.CX, cxp, a.CG,
                           nb * 16, gw, Cin, ci0, co0);
        }
    return mvs_check_launch("conv2d_wgrad_reduce");
}

// ---- all layers' weight gradients in one launch -------------------------------------------------------------------------------
int g_conv2d_wgrad_batch_groups = 2048;   // tuning knob "wgrad2d_batch": workgroups of the batched weight gradient (both launches together), shared out by work; FeatureNet at config 2: 0.185 / 0.180 / 0.176 / 0.174 / 0.176 ms at 1024 / 1536 / 2048 / 3072 / 4096 (profiles/r04_run29_*)

struct Wg2Static { int ks, stride, cx, cgmax, th, rowsp, cgp, cost; };
template <class C>
static Wg2Static wg2_static() { return {C::KS, C::S, C::CX, C::CGP, C::TH, C::ROWSP, C::CGP, C::COST}; }
static const Wg2Static* wg2_table() {
    static const Wg2Static t[6] = {wg2_static<Wg2A>(), wg2_static<Wg2B>(), wg2_static<Wg2C>(), wg2_static<Wg2D>(), wg2_static<Wg2E>(),
                                   wg2_static<Wg2F>()};
    return t;
}
// shapes[n][8] = N, H, W, Cin, Cout, ks, stride, w_channels_last.  Fills everything of the batch except the pointers; -> workspace
// floats, or -1 when a layer has no instantiation (3x3 stride 1 with 3 / 8 / 16 -> <= 16 or 32 -> <= 32 channels, 5x5 stride 2
// with 8 -> <= 16 or 16 -> <= 32; output channels a multiple of 4)
static long long wg2_plan(int n, const int* shapes, Wg2Batch& b) {
    if (n < 1 || n > WG2_MAX_LAYERS || !shapes) return -1;
    const Wg2Static* tab = wg2_table();
    double cost[WG2_MAX_LAYERS], total = 0.0;
    b.n = n;
    for (int i = 0; i < n; ++i) {
        const int* s = shapes + 8 * i;
        Wg2Layer& L = b.l[i];
        L = Wg2Layer{};
        L.N = s[0]; L.Hi = s[1]; L.Wi = s[2]; L.CX = s[3]; L.CG = s[4];
        const int ks = s[5], stride = s[6];
        L.wcl = s[7] ? 1 : 0;
        if (L.N < 1 || L.Hi < 1 || L.Wi < 1 || (L.CG & 3) || L.CG < 4) return -1;
        if ((long long)L.N * L.Hi * L.Wi * (L.CX > L.CG ? L.CX : L.CG) >= (1LL << 31)) return -1;
        L.cfg = -1;
        for (int c = 0; c < 6; ++c)
            if (tab[c].ks == ks && tab[c].stride == stride && tab[c].cx == L.CX && L.CG <= tab[c].cgmax) { L.cfg = c; break; }
        if (L.cfg < 0) return -1;
        const Wg2Static& T = tab[L.cfg];
        L.Ho = stride == 1 ? L.Hi : (L.Hi - 1) / 2 + 1; L.Wo = stride == 1 ? L.Wi : (L.Wi - 1) / 2 + 1;
        L.nth = mvs_cdiv(L.Ho, T.th); L.ntw = mvs_cdiv(L.Wo, 32); L.ntiles = L.N * L.nth * L.ntw;
        L.rowsp = T.rowsp; L.cgp = T.cgp; L.nt = ks * ks;
        cost[i] = (double)L.ntiles * T.cost;
        total += cost[i];
    }
    const int budget = g_conv2d_wgrad_batch_groups < n ? n : (g_conv2d_wgrad_batch_groups > 4096 ? 4096 : g_conv2d_wgrad_batch_groups);
    long long floats = 0;
    int wgc[2] = {0, 0}, rb = 0;       // workgroup ranges per launch class (conv2d_wgrad_batch_kernel<0 / 1>)
    for (int i = 0; i < n; ++i) {
        Wg2Layer& L = b.l[i];
        int want = (int)(budget * cost[i] / total + 0.5);
        if (want < 1) want = 1;
        if (want > L.ntiles) want = L.ntiles;
        L.tpw = mvs_cdiv(L.ntiles, want);
        L.nwg = mvs_cdiv(L.ntiles, L.tpw);
        L.wg0 = wgc[wg2_class(L.cfg)]; wgc[wg2_class(L.cfg)] += L.nwg;
        L.nrb = mvs_cdiv(L.rowsp * L.cgp, 16);
        L.rb0 = rb; rb += L.nrb;
        floats += (long long)L.nwg * L.rowsp * L.cgp;
    }
    return floats;
}

extern "C" long long mvs_conv2d_wgrad_batch_workspace_floats(int n, const int* shapes) {
    Wg2Batch b;
    return wg2_plan(n, shapes, b);
}

// gw[i] = weight gradient of layer i (x[i] [N,H,W,Cin], gy[i] [N,Ho,Wo,Cout], pad ks/2), written in the layout shapes[i][7] names.
// x_stats (or null) / imgs_per_group: see mvs_conv2d_wgrad_batch_xf.
static int wg2_run(int n, const float* const* x, const float* const* x_stats, int imgs_per_group, const float* const* gy,
                   float* const* gw, float* ws, const int* shapes, hipStream_t stream) {
    MVS_REQUIRE(x && gy && gw && ws && shapes, MVS_ERR_NULL, "conv2d_wgrad_batch: null pointer argument");
    Wg2Batch b;
    const long long floats = wg2_plan(n, shapes, b);
    MVS_REQUIRE(floats >= 0, MVS_ERR_UNSUPPORTED, "conv2d_wgrad_batch: %d layers, or a layer shape without an instantiation", n);
    float* part = ws;
    bool any_xf = false;
    for (int i = 0; i < n; ++i) {
        MVS_REQUIRE(x[i] && gy[i] && gw[i], MVS_ERR_NULL, "conv2d_wgrad_batch: null pointer for layer %d", i);
        Wg2Layer& L = b.l[i];
        L.x = x[i]; L.g = gy[i]; L.gw = gw[i]; L.part = part;
        L.xstats = x_stats ? x_stats[i] : nullptr; L.ipg = imgs_per_group;
        if (L.xstats) {
            MVS_REQUIRE((L.CX & 3) == 0 && imgs_per_group >= 1 && L.N % imgs_per_group == 0 && (L.N / imgs_per_group) * 2 * L.CX <= WG2_AFF,
                        MVS_ERR_SHAPE, "conv2d_wgrad_batch_xf: layer %d: %d channels / %d images in groups of %d (groups x channels <= 256)", i,
                        L.CX, L.N, imgs_per_group);
            any_xf = true;
        }
        part += (size_t)L.nwg * L.rowsp * L.cgp;
    }
    int wgc[2] = {0, 0};
    for (int i = 0; i < n; ++i) wgc[wg2_class(b.l[i].cfg)] += b.l[i].nwg;
    // the wide layers first: few, long workgroups; the narrow layers' many short ones fill the GPU behind them
    if (any_xf) {
        if (wgc[1]) MVS_LAUNCH(conv2d_wgrad_batch_xf1_kernel, dim3(wgc[1]), dim3(256), 0, stream, b);
        if (wgc[0]) MVS_LAUNCH(conv2d_wgrad_batch_xf0_kernel, dim3(wgc[0]), dim3(256), 0, stream, b);
    } else {
        if (wgc[1]) MVS_LAUNCH((conv2d_wgrad_batch_kernel<1>), dim3(wgc[1]), dim3(256), 0, stream, b);
        if (wgc[0]) MVS_LAUNCH((conv2d_wgrad_batch_kernel<0>), dim3(wgc[0]), dim3(256), 0, stream, b);
    }
    int rc = mvs_check_launch("conv2d_wgrad_batch");
    if (rc) return rc;
    const Wg2Layer& last = b.l[n - 1];
    MVS_LAUNCH(conv2d_wgrad_batch_reduce_kernel, dim3(last.rb0 + last.nrb), dim3(256), 0, stream, b);
    return mvs_check_launch("conv2d_wgrad_batch_reduce");
}

extern "C" int mvs_conv2d_wgrad_batch(int n, const float* const* x, const float* const* gy, float* const* gw, float* ws,
                                      const int* shapes, hipStream_t stream) {
    return wg2_run(n, x, nullptr, 1, gy, gw, ws, shapes, stream);
}

// The same where x[i] may be the RAW output of the BatchNorm + ReLU block in front of layer i: x_stats[i] [G][4][Cin] (mean, invstd,
// scale, shift of that block, mvs_bn_finalize_slots; null = x[i] is used as it is), groups of imgs_per_group images -- the layer's
// input relu(x * scale + shift) is formed while the halo is staged (the forward pass did the same: mvs_conv2d_fwd_stats_xf), so
// the normalised activation exists nowhere in memory.
extern "C" int mvs_conv2d_wgrad_batch_xf(int n, const float* const* x, const float* const* x_stats, int imgs_per_group,
                                         const float* const* gy, float* const* gw, float* ws, const int* shapes, hipStream_t stream) {
    MVS_REQUIRE(x_stats, MVS_ERR_NULL, "conv2d_wgrad_batch_xf: null pointer argument");
    return wg2_run(n, x, x_stats, imgs_per_group, gy, gw, ws, shapes, stream);
}
