// bf16-storage inference path of the cost-volume regulariser (BASELINE configs[4]: MVSNet N=7, 1600x1184, D=256).
//
// Replaces nn.Conv3d / nn.ConvTranspose3d + folded BatchNorm3d + ReLU (+ skip) of CostRegNet in eval mode
// (jdacs/models/mvsnet.py:37-74, evaluated under no_grad in jdacs/eval.py:143) with activations STORED in bf16
// (the 1600x1184x256 volume is 1.94 GB instead of 3.9 GB) and fp32 accumulation: v_mfma_f32_16x16x32_bf16.
// The reference has no reduced-precision path: the oracle for this path is the fp32 path on the
// same inputs, the tolerance is stated in tests/test_gpu_parity.py::test_bf16_inference_path.
//
// GEMM view per workgroup tile (the fp32 kernels' geometry, conv_map.h): M = Cout (the WEIGHTS are the MFMA's A operand,
// so a lane ends up with 4 consecutive output channels of ONE voxel -> 8-byte bf16x4 stores, no transpose), N = 16
// voxels along W, K = (tap, ci) with ci fastest.  A lane's 8 consecutive k are 8 consecutive input channels of one tap:
// one ds_read_b128 from the bf16 halo tile in LDS (channels-last, 16-byte padded pitch: conflict-free for the 16 voxels
// of a lane group).  Weight fragments come from a pre-packed bf16 image (L1/L2 resident, 16 bytes per lane and k-step).
// These layers are HBM / LDS bound at bf16 MFMA rates (615 GFLOP forward at config 5 = 0.25 ms of MFMA time against
// >= 1 ms of activation traffic), so the kernel is one straightforward form for every layer shape instead of the fp32
// family's specialisations; Cout = 8 and Cout = 1 simply leave MFMA rows unused.
#include <string.h>
#include "mvs_rt.h"
#include "conv_map.h"

typedef unsigned short bf16_t;
extern int g_conv_cout1_d4;   // conv3d.hip: knob "cout1_d4"
extern int g_conv_tr2pw;       // conv3d.hip: knob "tr2pw"
int g_conv_bf16_dp = 1;         // knob "bf16_dp": conv0 of the bf16 path (32 -> 8) with two output depth slices per MFMA (GEOM_S1_DP)

struct Bf16ConvArgs {
    const bf16_t* x;        // [B,Di,Hi,Wi,CIN] bf16
    const bf16_t* wp;       // packed weights
    void* y;                // [B,Do,Ho,Wo,COUT] bf16, or fp32 when out_f32
    const float* scale;     // [COUT] or null
    const float* shift;     // [COUT] or null (bias when scale == null)
    const bf16_t* skip;     // like y (bf16), added after the ReLU, or null
    int relu, out_f32;
    int B, Di, Hi, Wi, Do, Ho, Wo;
    int QD, QH, QW;         // coarse-grid extents
    int ntd, nth, ntw;      // tiles per dim
};

// fp32 -> bf16, round to nearest even (NaN stays NaN)
__host__ __device__ static inline bf16_t f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ __device__ static inline float bf2f(bf16_t h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// taps of a class: S1 / S2 have one class of 27 taps; TR2 has 8 parity classes of 1..8 taps
template <int GEOM> struct Bf16Geom {
    using G = ConvGeom<GEOM>;
    static constexpr int NBW = (GEOM == GEOM_S2 || GEOM == GEOM_S1_DP) ? 2 : 4;          // 16-voxel rows per wave
};
// GEOM_TR2_PW (transposed, Cout == 8): the two output parities along W share one MFMA -- row m = pw*8 + co, class = (pd, ph), taps
// = the (d, h) taps of the class times the two input offsets along W (conv_map.h: tr2p_*): all 16 rows carry channels, 9 k-steps
// instead of 14 at Cin = 16, and the lanes of a voxel pair store 32 consecutive bytes
MVS_HD inline int bf16_ncls(int geom) { return geom == GEOM_TR2 ? 8 : (geom == GEOM_TR2_PW ? 4 : 1); }
// GEOM_S1_DP (stride 1, Cout == 8): two output depth slices share one MFMA -- row m = pd*8 + co, 36 taps = the 4 input slices under
// the pair x 3 x 3 (tap (kd', kh, kw) carries W[kd' - pd] for row parity pd, zero where kd' - pd is outside 0..2): all 16 rows carry
// channels, 36 k-steps per slice pair instead of 54, and a wave reads each input row once for both slices
MVS_HD inline int bf16_ntaps(int geom, int cls) {
    return geom == GEOM_TR2 ? tr2_ntaps(cls) : (geom == GEOM_TR2_PW ? tr2p_ntaps(cls) : (geom == GEOM_S1_DP ? 36 : 27));
}
MVS_HD inline int bf16_ksteps(int geom, int cls, int cin) { return (bf16_ntaps(geom, cls) * cin + 31) / 32; }
MVS_HD inline int bf16_kstep_prefix(int geom, int cls, int cin) {
    int s = 0;
    for (int c = 0; c < cls; ++c) s += bf16_ksteps(geom, c, cin);
    return s;
}

// Packed image: [global k-step][mb][64 lanes][8] bf16.  Lane l = (m = l & 15, kg = l >> 4), element j:
// flattened k = 32*ks + 8*kg + j -> (tap = k / CIN, ci = k % CIN); value W[co = 16*mb + m][ci][tap's kernel index]
// (zero beyond the class's taps / beyond COUT).
__global__ __launch_bounds__(256) void conv_bf16_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp, int geom,
                                                             int cin, int cout, int MB, int layout, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    const int mb = (idx >> 9) % MB;
    int kk = (idx >> 9) / MB;
    int cls = 0;
    const int ncls = bf16_ncls(geom);
    for (; cls < ncls; ++cls) {
        const int n = bf16_ksteps(geom, cls, cin);
        if (kk < n) break;
        kk -= n;
    }
    const int kflat = 32 * kk + 8 * (lane >> 4) + j;
    const int tap = kflat / cin, ci = kflat % cin;
    int co = 16 * mb + (lane & 15);
    float v = 0.f;
    if (geom == GEOM_TR2_PW) {   // row = pw*8 + co (cout == 8, one m-block)
        const int pw = (lane & 15) >> 3;
        co = lane & 7;
        if (tap < tr2p_ntaps(cls)) {
            int dd, dh, dw, kd, kh;
            tr2p_tap(cls, tap, dd, dh, dw, kd, kh);
            const int kw = tr2p_kw(pw, dw);
            if (kw >= 0) {
                const int kidx = kd * 9 + kh * 3 + kw;
                v = layout == WL_OIK ? w[((size_t)co * cin + ci) * 27 + kidx] : w[((size_t)ci * cout + co) * 27 + kidx];
            }
        }
    } else if (geom == GEOM_S1_DP) {   // row = pd*8 + co (cout == 8, one m-block); tap = (kd', kh, kw), kd' in 0..3
        const int pd = (lane & 15) >> 3;
        co = lane & 7;
        const int kd = tap / 9 - pd, kh = (tap / 3) % 3, kw = tap % 3;
        if (tap < 36 && kd >= 0 && kd <= 2) {
            const int kidx = kd * 9 + kh * 3 + kw;
            v = layout == WL_OIK ? w[((size_t)co * cin + ci) * 27 + kidx] : w[((size_t)ci * cout + co) * 27 + kidx];
        }
    } else if (co < cout && tap < bf16_ntaps(geom, cls)) {
        int kd, kh, kw;
        if (geom == GEOM_TR2) {
            int dd, dh, dw;
            tr2_tap(cls, tap, dd, dh, dw, kd, kh, kw);
        } else {
            kd = tap / 9; kh = (tap / 3) % 3; kw = tap % 3;
        }
        const int kidx = kd * 9 + kh * 3 + kw;
        v = layout == WL_OIK ? w[((size_t)co * cin + ci) * 27 + kidx] : w[((size_t)ci * cout + co) * 27 + kidx];
    }
    wp[idx] = f2bf(v);
}

template <int GEOM, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_bf16_kernel(Bf16ConvArgs a) {
    using G = ConvGeom<GEOM>;
    constexpr int NBW = Bf16Geom<GEOM>::NBW;
    constexpr int MB = (COUT + 15) / 16;
    constexpr int PITCH = CIN + 8;                       // bf16 elements per halo voxel (16 bytes of padding)
    constexpr int NR = G::RD * G::RH * G::RW;
    constexpr int CH = CIN / 8;                          // 16-byte chunks per voxel
    constexpr int NCLS = G::NCLS;
    __shared__ __attribute__((aligned(16))) bf16_t halo[NR * PITCH];
    constexpr bool DP = GEOM == GEOM_S1_DP;
    constexpr int TAPS = DP ? 64 : 32;                   // table slots per class
    __shared__ int s_tapoff[NCLS * TAPS];                // halo offset (bf16 elements) of tap t of class c; clamped beyond the class
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int t = blockIdx.x;
    const int tw = t % a.ntw; t /= a.ntw;
    const int th = t % a.nth; t /= a.nth;
    const int td = t % a.ntd; t /= a.ntd;
    const int b = t;
    const int q0d = td * G::TQD, q0h = th * G::TQH, q0w = tw * G::TQW;
    // ---- tap table ----
    if (tid < NCLS * TAPS) {
        const int cls = tid / TAPS;
        int tap = tid % TAPS;
        const int nt = bf16_ntaps(GEOM, cls);
        if (tap >= nt) tap = nt - 1;                     // padded k: zero weights, any valid address
        int dz, dy, dx;
        if (GEOM == GEOM_TR2) {
            int kd, kh, kw;
            tr2_tap(cls, tap, dz, dy, dx, kd, kh, kw);
        } else if (GEOM == GEOM_TR2_PW) {
            int kd, kh;
            tr2p_tap(cls, tap, dz, dy, dx, kd, kh);
        } else {   // S1 / S2: dz in 0..2; S1_DP: dz in 0..3 (the input slices under an output slice pair)
            dz = tap / 9; dy = (tap / 3) % 3; dx = tap % 3;
        }
        s_tapoff[tid] = ((dz * G::RH + dy) * G::RW + dx) * PITCH;
    }
    // Weight fragments live in global memory (L2): a k-step that waits for its own fragment pays the full L2 round trip
    // (first version: 25 k cycles per conv0 tile, 10x the LDS + MFMA time).  Small images (<= 28 fragments: conv0, the
    // 8- and 16-channel layers) are therefore fetched into registers ONCE, here, so that their round trip overlaps the halo
    // staging; larger ones run through a ring of PD fragments requested PD k-steps ahead.
    constexpr int KS1 = (27 * CIN + 31) / 32;            // k-steps of the single class of S1 / S2
    // (measured, run 9: preloading conv0's 27 fragments cost a wave of occupancy and was 35 % SLOWER than fetching them in the
    //  loop, 2.39 vs 1.77 ms at config 5 -- only images of <= 8 fragments are preloaded)
    constexpr bool PRELOAD = G::BASE != GEOM_TR2 && KS1 * MB <= 8;
    constexpr int PD = 4;
    mvs_bf16x8 areg[PRELOAD ? KS1 : 1][MB];
    if constexpr (PRELOAD) {
        const bf16_t* __restrict__ wk0 = a.wp + (size_t)(tid & 63) * 8;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int m = 0; m < MB; ++m) areg[ks][m] = *reinterpret_cast<const mvs_bf16x8*>(wk0 + ((size_t)ks * MB + m) * 512);
        MVS_SCHED_FENCE();   // keep ALL the fragment loads up here (the scheduler otherwise sinks each next to its MFMA)
    }
    // ---- stage the halo region (zero outside the volume): all loads first, then the LDS writes ----
    {
        const int g0d = q0d * G::IS - G::PAD, g0h = q0h * G::IS - G::PAD, g0w = q0w * G::IS - G::PAD;
        const bf16_t* __restrict__ xb = a.x + (size_t)b * a.Di * a.Hi * a.Wi * CIN;
        constexpr int NITEMS = NR * CH;
        constexpr int NIT = (NITEMS + 255) / 256;
        constexpr int BMAX = (PRELOAD && KS1 * MB > 16) ? 6 : 12;   // register room next to the preloaded weight fragments
        constexpr int BATCH = NIT < BMAX ? NIT : BMAX;
#pragma unroll 1
        for (int k0 = 0; k0 < NIT; k0 += BATCH) {
            uint4 v[BATCH];
            int off[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int i = tid + 256 * (k0 + k);
                off[k] = -1;
                v[k] = make_uint4(0u, 0u, 0u, 0u);
                if (i < NITEMS) {
                    const int pos = i / CH, ch = i % CH;
                    const int rw = pos % G::RW, rh = (pos / G::RW) % G::RH, rd = pos / (G::RW * G::RH);
                    const int gd = g0d + rd, gh = g0h + rh, gw = g0w + rw;
                    off[k] = pos * PITCH + 8 * ch;
                    if (gd >= 0 && gd < a.Di && gh >= 0 && gh < a.Hi && gw >= 0 && gw < a.Wi)
                        v[k] = *reinterpret_cast<const uint4*>(xb + (((size_t)gd * a.Hi + gh) * a.Wi + gw) * CIN + 8 * ch);
                }
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k)
                if (off[k] >= 0) *reinterpret_cast<uint4*>(halo + off[k]) = v[k];
        }
    }
    __syncthreads();
    // ---- implicit GEMM ----
    const int n = lane & 15, kg = lane >> 4;
    // the wave's 16-voxel rows: S1 / TR2 tile 4 x 4 x 16 -> wave = d slice, rows = h; S2 tile 2 x 4 x 16 -> wave = (d, h pair)
    int rowbase[NBW], rqd[NBW], rqh[NBW];
#pragma unroll
    for (int r = 0; r < NBW; ++r) {
        rqd[r] = (GEOM == GEOM_S2 || DP) ? (wv >> 1) : wv;        // S1_DP: the slice PAIR
        rqh[r] = (GEOM == GEOM_S2 || DP) ? 2 * (wv & 1) + r : r;
        rowbase[r] = ((rqd[r] * (DP ? 2 : G::IS) * G::RH + rqh[r] * G::IS) * G::RW + n * G::IS) * PITCH;
    }
    const int Do = a.Do, Ho = a.Ho, Wo = a.Wo;
    int kbase = 0;                                       // global k-step index of the class's first step
#pragma unroll 1
    for (int cls = 0; cls < NCLS; ++cls) {
        const int nks = bf16_ksteps(GEOM, cls, CIN);
        f32x4 acc[NBW][MB];
#pragma unroll
        for (int r = 0; r < NBW; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bf16_t* __restrict__ wk = a.wp + ((size_t)kbase * MB * 64 + lane) * 8;
        if constexpr (PRELOAD) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int kflat = 32 * ks + 8 * kg;
                const int boff = s_tapoff[kflat / CIN] + kflat % CIN;
                mvs_bf16x8 bfrag[NBW];
#pragma unroll
                for (int r = 0; r < NBW; ++r) bfrag[r] = *reinterpret_cast<const mvs_bf16x8*>(halo + rowbase[r] + boff);
#pragma unroll
                for (int r = 0; r < NBW; ++r)
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[r][m] = MVS_MFMA_16x16x32_BF16(areg[ks][m], bfrag[r], acc[r][m]);
            }
        } else {
            mvs_bf16x8 ring[PD][MB];
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int kc = u < nks ? u : nks - 1;
#pragma unroll
                for (int m = 0; m < MB; ++m) ring[u][m] = *reinterpret_cast<const mvs_bf16x8*>(wk + ((size_t)kc * MB + m) * 512);
            }
            for (int ks0 = 0; ks0 < nks; ks0 += PD) {
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int ks = ks0 + u;
                    if (ks < nks) {
                        const int kflat = 32 * ks + 8 * kg;
                        const int boff = s_tapoff[cls * TAPS + kflat / CIN] + kflat % CIN;
                        mvs_bf16x8 bfrag[NBW];
#pragma unroll
                        for (int r = 0; r < NBW; ++r) bfrag[r] = *reinterpret_cast<const mvs_bf16x8*>(halo + rowbase[r] + boff);
#pragma unroll
                        for (int r = 0; r < NBW; ++r)
#pragma unroll
                            for (int m = 0; m < MB; ++m) acc[r][m] = MVS_MFMA_16x16x32_BF16(ring[u][m], bfrag[r], acc[r][m]);
                        const int kn = ks + PD < nks ? ks + PD : nks - 1;   // past the end: re-read the last fragment (harmless)
#pragma unroll
                        for (int m = 0; m < MB; ++m) ring[u][m] = *reinterpret_cast<const mvs_bf16x8*>(wk + ((size_t)kn * MB + m) * 512);
                    }
                }
            }
        }
        kbase += nks;
        // ---- epilogue: folded BatchNorm (or bias) + ReLU + skip, 4 consecutive channels of one voxel per lane ----
        // TR2_PW: class = (pd, ph); the lane's four rows are channels 4*(kg & 1).. of the voxel with W parity kg >> 1
        const int pd = G::PW ? (cls >> 1) & 1 : (cls >> 2) & 1, ph = G::PW ? cls & 1 : (cls >> 1) & 1, pw = G::PW ? kg >> 1 : cls & 1;
#pragma unroll
        for (int r = 0; r < NBW; ++r) {
            const int qd = q0d + (DP ? 2 * rqd[r] + (kg >> 1) : rqd[r]), qh = q0h + rqh[r], qw = q0w + n;
            if (qd >= a.QD || qh >= a.QH || qw >= a.QW) continue;
            const int od = qd * G::OS + (G::BASE == GEOM_TR2 ? pd : 0), oh = qh * G::OS + (G::BASE == GEOM_TR2 ? ph : 0),
                      ow = qw * G::OS + (G::BASE == GEOM_TR2 ? pw : 0);
            const size_t vox = (((size_t)b * Do + od) * Ho + oh) * Wo + ow;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int c0 = (G::PW || DP) ? 4 * (kg & 1) : 16 * m + 4 * kg;
                if (c0 >= COUT) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[r][m][e];
                    if (c0 + e < COUT) {
                        if (a.scale) v[e] = fmaf(v[e], a.scale[c0 + e], a.shift[c0 + e]);
                        else if (a.shift) v[e] += a.shift[c0 + e];
                    }
                    if (a.relu) v[e] = fmaxf(v[e], 0.f);
                }
                if (COUT % 4 == 0) {
                    if (a.skip) {
                        const uint2 s = *reinterpret_cast<const uint2*>(a.skip + vox * COUT + c0);
                        v[0] += bf2f((bf16_t)(s.x & 0xffffu)); v[1] += bf2f((bf16_t)(s.x >> 16));
                        v[2] += bf2f((bf16_t)(s.y & 0xffffu)); v[3] += bf2f((bf16_t)(s.y >> 16));
                    }
                    if (a.out_f32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + vox * COUT + c0) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 o;
                        o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                        o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.y) + vox * COUT + c0) = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c0 + e < COUT) {
                            float val = v[e];
                            if (a.skip) val += bf2f(a.skip[vox * COUT + c0 + e]);
                            if (a.out_f32) reinterpret_cast<float*>(a.y)[vox * COUT + c0 + e] = val;
                            else reinterpret_cast<bf16_t*>(a.y)[vox * COUT + c0 + e] = f2bf(val);
                        }
                }
            }
        }
    }
}

// Cout == 1 (the probability layer, mvsnet.py:63) on bf16 activations as a direct convolution: the MFMA form above uses one of its 16
// rows here (0.43 ms at config 5 against 0.12 ms of HBM time).  Same scheme as conv_cout1_d4_kernel of conv3d.hip: a thread owns a
// column of four depth slices of one (h, w), the halo tile is one 16-byte word (8 bf16 channels) per voxel and channel octet, the
// weights are rounded to bf16 like the packed image of the MFMA form (bf16 operands, fp32 accumulation).  Tile 4 x 8 x 16, 128 threads.
template <int CIN>
__global__ __launch_bounds__(128) void conv_bf16_cout1_d4_kernel(Bf16ConvArgs a, const float* __restrict__ w) {
    constexpr int TD = 4, TH = 8, TW = 16, RD = TD + 2, RH = TH + 2, RW = TW + 2, NR = RD * RH * RW, CO = CIN / 8;
    __shared__ uint4 tile[CO * NR];       // [octet][voxel]
    __shared__ float4 wl[27 * CIN / 4];   // [tap][ci]
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tw = t % a.ntw; t /= a.ntw;
    const int th = t % a.nth; t /= a.nth;
    const int td = t % a.ntd; t /= a.ntd;
    const int b = t;
    const int qd0 = td * TD, qh0 = th * TH, qw0 = tw * TW;
    for (int i = tid; i < 27 * CIN; i += 128) {
        const int tap = i / CIN, ci = i % CIN;   // W[0][ci][tap]
        reinterpret_cast<float*>(wl)[i] = bf2f(f2bf(w[(size_t)ci * 27 + tap]));
    }
    constexpr int NITEMS = NR * CO, NIT = (NITEMS + 127) / 128, BATCH = NIT < 12 ? NIT : 12;
    const bf16_t* __restrict__ xb = a.x + (size_t)b * a.Di * a.Hi * a.Wi * CIN;
#pragma unroll
    for (int k0 = 0; k0 < NIT; k0 += BATCH) {
        uint4 v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int i = tid + 128 * (k0 + k);
            v[k] = make_uint4(0u, 0u, 0u, 0u);
            if (k0 + k < NIT && i < NITEMS) {
                const int vox = i / CO, oc = i % CO;
                const int rw = vox % RW, rh = (vox / RW) % RH, rd = vox / (RW * RH);
                const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
                if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                    v[k] = *reinterpret_cast<const uint4*>(xb + (((size_t)id * a.Hi + ih) * a.Wi + iw) * CIN + 8 * oc);
            }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int i = tid + 128 * (k0 + k);
            if (k0 + k < NIT && i < NITEMS) tile[(i % CO) * NR + i / CO] = v[k];
        }
    }
    __syncthreads();
    const int pw = tid % TW, ph = tid / TW;
    float acc[TD];
#pragma unroll
    for (int pd = 0; pd < TD; ++pd) acc[pd] = 0.f;
    // not unrolled over (kh, kw) / the channel octets: see conv_cout1_d4_kernel (register pressure)
#pragma unroll 1
    for (int khw = 0; khw < 9; ++khw) {
        const int kh = khw / 3, kw = khw % 3;
        const int col = (ph + kh) * RW + pw + kw;
#pragma unroll 1
        for (int oc = 0; oc < CO; ++oc) {
            float4 wv[3][2];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                wv[kd][0] = wl[(((kd * 3 + kh) * 3 + kw) * CIN + 8 * oc) / 4];
                wv[kd][1] = wl[(((kd * 3 + kh) * 3 + kw) * CIN + 8 * oc) / 4 + 1];
            }
#pragma unroll
            for (int din = 0; din < RD; ++din) {
                const uint4 u = tile[oc * NR + din * RH * RW + col];
                float xv[8];
                xv[0] = bf2f((bf16_t)(u.x & 0xffffu)); xv[1] = bf2f((bf16_t)(u.x >> 16));
                xv[2] = bf2f((bf16_t)(u.y & 0xffffu)); xv[3] = bf2f((bf16_t)(u.y >> 16));
                xv[4] = bf2f((bf16_t)(u.z & 0xffffu)); xv[5] = bf2f((bf16_t)(u.z >> 16));
                xv[6] = bf2f((bf16_t)(u.w & 0xffffu)); xv[7] = bf2f((bf16_t)(u.w >> 16));
#pragma unroll
                for (int pd = 0; pd < TD; ++pd) {
                    const int kd = din - pd;
                    if (kd < 0 || kd > 2) continue;
                    acc[pd] = fmaf(xv[0], wv[kd][0].x, acc[pd]); acc[pd] = fmaf(xv[1], wv[kd][0].y, acc[pd]);
                    acc[pd] = fmaf(xv[2], wv[kd][0].z, acc[pd]); acc[pd] = fmaf(xv[3], wv[kd][0].w, acc[pd]);
                    acc[pd] = fmaf(xv[4], wv[kd][1].x, acc[pd]); acc[pd] = fmaf(xv[5], wv[kd][1].y, acc[pd]);
                    acc[pd] = fmaf(xv[6], wv[kd][1].z, acc[pd]); acc[pd] = fmaf(xv[7], wv[kd][1].w, acc[pd]);
                }
            }
        }
    }
    const int qh = qh0 + ph, qw = qw0 + pw;
    if (qh < a.QH && qw < a.QW) {
#pragma unroll
        for (int pd = 0; pd < TD; ++pd) {
            const int qd = qd0 + pd;
            if (qd >= a.QD) continue;
            const size_t o = (((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qw;
            float v = acc[pd];
            if (a.scale) v = fmaf(v, a.scale[0], a.shift[0]);
            else if (a.shift) v += a.shift[0];
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.skip) v += bf2f(a.skip[o]);
            if (a.out_f32) reinterpret_cast<float*>(a.y)[o] = v;
            else reinterpret_cast<bf16_t*>(a.y)[o] = f2bf(v);
        }
    }
}

// fp32 -> bf16 (feature maps are fp32; the volume is produced in bf16 by the sweep kernel directly)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    uint2 o;
    o.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    o.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    reinterpret_cast<uint2*>(y)[i] = o;
}

// ================================================================================================
// host side
// ================================================================================================
// The layer shapes of CostRegNet (jdacs/models/mvsnet.py:40-63) plus their neighbours; anything else is an error, not a
// silent fallback.  (Stride-2 tiles with 64 input channels would not fit LDS; the network has none.)
#define MVS_BF16_CASE(G, CI, CO) \
    if (geom == G && cin == CI && cout == CO) { MVS_LAUNCH((conv_bf16_kernel<G, CI, CO>), grid, block, 0, st, a); return mvs_check_launch("conv_bf16"); }
static int launch_bf16(const Bf16ConvArgs& a, int geom, int cin, int cout, int nblocks, hipStream_t st) {
    dim3 grid(nblocks), block(256);
    MVS_BF16_CASE(GEOM_S1, 32, 8) MVS_BF16_CASE(GEOM_S1, 8, 8) MVS_BF16_CASE(GEOM_S1, 16, 16) MVS_BF16_CASE(GEOM_S1, 32, 32)
    MVS_BF16_CASE(GEOM_S1, 64, 64) MVS_BF16_CASE(GEOM_S1, 8, 1) MVS_BF16_CASE(GEOM_S1, 16, 1) MVS_BF16_CASE(GEOM_S1, 16, 8)
    MVS_BF16_CASE(GEOM_S2, 8, 16) MVS_BF16_CASE(GEOM_S2, 16, 32) MVS_BF16_CASE(GEOM_S2, 32, 64)
    MVS_BF16_CASE(GEOM_TR2, 64, 32) MVS_BF16_CASE(GEOM_TR2, 32, 16) MVS_BF16_CASE(GEOM_TR2, 16, 8) MVS_BF16_CASE(GEOM_TR2_PW, 16, 8) MVS_BF16_CASE(GEOM_S1_DP, 32, 8)
    mvs_set_error("conv3d bf16: no kernel for %s %d -> %d channels (supported: the CostRegNet layer shapes)",
                  geom == GEOM_S1 ? "stride-1 conv" : (geom == GEOM_S2 ? "stride-2 conv" : "transposed stride-2 conv"), cin, cout);
    return MVS_ERR_UNSUPPORTED;
}
#undef MVS_BF16_CASE

static int bf16_total_ksteps(int geom, int cin) { return bf16_kstep_prefix(geom, bf16_ncls(geom), cin); }

// bytes of workspace (the packed bf16 weight image) a call needs
extern "C" long long mvs_conv3d_bf16_workspace_bytes(int Cin, int Cout, int stride, int transposed) {
    if (!(Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64) || Cout < 1 || Cout > 64) return -1;
    const int geom = transposed ? GEOM_TR2 : (stride == 2 ? GEOM_S2 : GEOM_S1);
    int ks = bf16_total_ksteps(geom, Cin);
    if (geom == GEOM_S1 && Cout == 8) {   // whichever form the knob picks
        const int dp = bf16_total_ksteps(GEOM_S1_DP, Cin);
        ks = ks > dp ? ks : dp;
    }
    return (long long)ks * mvs_cdiv(Cout, 16) * 512 * 2;
}

// y = conv3d(x, w, stride 1|2, pad 1)  |  conv_transpose3d(x, w, stride 2, pad 1, output_padding 1), then
// y*scale[c]+shift[c] (folded BatchNorm) or +shift[c] (bias), ReLU, + skip.   x, skip: bf16 channels-last-3d;
// w: the fp32 parameter ([Cout][Cin][27] conv, [Cin][Cout][27] transposed); y: bf16, or fp32 when out_is_f32.
// (D,H,W) are the spatial dims of x.  Inference only (no statistics, no gradient).
extern "C" int mvs_conv3d_bf16_fwd(const void* x, const float* w, void* y, void* ws, int B, int D, int H, int W, int Cin, int Cout,
                                   int stride, int transposed, const float* scale, const float* shift, const void* skip,
                                   int relu, int out_is_f32, hipStream_t stream) {
    MVS_REQUIRE(x && w && y && ws, MVS_ERR_NULL, "conv3d bf16: null pointer argument");
    MVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, MVS_ERR_SHAPE, "conv3d bf16: bad shape B=%d D=%d H=%d W=%d", B, D, H, W);
    MVS_REQUIRE(stride == 1 || stride == 2, MVS_ERR_UNSUPPORTED, "conv3d bf16: stride must be 1 or 2, got %d", stride);
    MVS_REQUIRE(!(transposed && stride != 2), MVS_ERR_UNSUPPORTED, "conv3d bf16: the transposed form is stride 2 only");
    MVS_REQUIRE(!(scale && !shift), MVS_ERR_NULL, "conv3d bf16: scale without shift");
    MVS_REQUIRE(mvs_conv3d_bf16_workspace_bytes(Cin, Cout, stride, transposed) > 0, MVS_ERR_UNSUPPORTED,
                "conv3d bf16: channels must be Cin 8/16/32/64, Cout <= 64 (got %d -> %d)", Cin, Cout);
    const int geom = transposed ? GEOM_TR2 : (stride == 2 ? GEOM_S2 : GEOM_S1);
    Bf16ConvArgs a = {};
    a.x = (const bf16_t*)x; a.y = y; a.scale = scale; a.shift = shift; a.skip = (const bf16_t*)skip;
    a.relu = relu; a.out_f32 = out_is_f32;
    a.B = B; a.Di = D; a.Hi = H; a.Wi = W;
    if (geom == GEOM_S1) { a.Do = D; a.Ho = H; a.Wo = W; a.QD = D; a.QH = H; a.QW = W; }
    else if (geom == GEOM_S2) {
        a.Do = (D - 1) / 2 + 1; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
        a.QD = a.Do; a.QH = a.Ho; a.QW = a.Wo;
    } else { a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W; a.QD = D; a.QH = H; a.QW = W; }
    a.ntd = mvs_cdiv(a.QD, geom == GEOM_S2 ? 2 : 4); a.nth = mvs_cdiv(a.QH, 4); a.ntw = mvs_cdiv(a.QW, 16);
    if (geom == GEOM_S1 && Cout == 1 && (Cin == 8 || Cin == 16) && (g_conv_cout1_d4 & 2)) {   // direct form, four outputs per thread
        a.nth = mvs_cdiv(a.QH, 8);
        const long long nb4 = (long long)B * a.ntd * a.nth * a.ntw;
        MVS_REQUIRE(nb4 < (1ll << 31), MVS_ERR_SHAPE, "conv3d bf16: too many tiles");
        if (Cin == 8) MVS_LAUNCH((conv_bf16_cout1_d4_kernel<8>), dim3((unsigned)nb4), dim3(128), 0, stream, a, w);
        else MVS_LAUNCH((conv_bf16_cout1_d4_kernel<16>), dim3((unsigned)nb4), dim3(128), 0, stream, a, w);
        return mvs_check_launch("conv_bf16_cout1_d4");
    }
    const int MB = mvs_cdiv(Cout, 16);
    // transposed with 8 output channels (16 -> 8, the last decoder block): both W parities in one MFMA (knob "tr2pw", shared with
    // the fp32 kernels' GEOM_TR2_PW)
    const int kgeom = (geom == GEOM_TR2 && Cout == 8 && Cin == 16 && g_conv_tr2pw) ? GEOM_TR2_PW
                    : ((geom == GEOM_S1 && Cout == 8 && Cin == 32 && g_conv_bf16_dp) ? GEOM_S1_DP : geom);   // knob "bf16_dp"
    const int total = bf16_total_ksteps(kgeom, Cin) * MB * 512;
    MVS_LAUNCH(conv_bf16_pack_kernel, dim3(mvs_cdiv(total, 256)), dim3(256), 0, stream, w, (bf16_t*)ws, kgeom, Cin, Cout, MB,
               transposed ? WL_IOK : WL_OIK, total);
    a.wp = (const bf16_t*)ws;
    const long long nblocks = (long long)B * a.ntd * a.nth * a.ntw;
    MVS_REQUIRE(nblocks < (1ll << 31), MVS_ERR_SHAPE, "conv3d bf16: too many tiles");
    return launch_bf16(a, kgeom, Cin, Cout, (int)nblocks, stream);
}

// elementwise fp32 -> bf16 (n % 4 == 0)
extern "C" int mvs_cast_f32_bf16(const float* x, void* y, long long n, hipStream_t stream) {
    MVS_REQUIRE(x && y, MVS_ERR_NULL, "cast: null pointer argument");
    MVS_REQUIRE(n > 0 && n % 4 == 0, MVS_ERR_SHAPE, "cast: element count must be a positive multiple of 4, got %lld", n);
    const size_t n4 = (size_t)n / 4;
    MVS_LAUNCH(cast_f32_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, (bf16_t*)y, n4);
    return mvs_check_launch("cast_f32_bf16");
}
