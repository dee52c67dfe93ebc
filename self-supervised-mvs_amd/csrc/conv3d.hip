// K3-K8: 3-D convolution family of the cost-volume regulariser as fp32 MFMA implicit GEMMs.
//
// Replaces nn.Conv3d / nn.ConvTranspose3d forward, input-gradient and weight-gradient of
// CostRegNet (jdacs/models/mvsnet.py:37-74) and its CVP twin (jdacs-ms/models/network.py:44-74).
// All kernels k=3, pad=1; stride 1 or 2; transposed stride 2 has output_padding 1, stride 1 has 0.
//
// Layout: activations channels-last [B,D,H,W,C] fp32.  GEMM view: M = voxels, N = Cout, K = taps*Cin.
// A workgroup (4 wavefronts) owns a TQDxTQHx16 block of "coarse grid" positions; the input halo
// region of that block is staged once into LDS (zero filled outside the volume, so the inner loop
// has no bounds checks), and every wavefront walks K in steps of 16 with
// v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain).  One ds_read_b128 per lane feeds the
// A operand of 4 MFMAs; B operands come from a pre-packed weight image (conv_map.h) that is
// L1/L2 resident.  Epilogue: raw store + per-workgroup BatchNorm partial sums (train), or
// scale/shift/ReLU/skip fused (eval), or bias (prob layer).
#include "mvs_rt.h"
#include "conv_map.h"

// tuning knob "cout1_d4": the Cout = 1 layer with four outputs per thread -- bit 0: the fp32 forward and input gradient
// (measured, profiles/r03_run16_*: forward 0.089 -> 0.091 ms, 16 channels 0.331 -> 0.473, input gradient 0.083 -> 0.107: the
// smaller LDS traffic does not pay for 1-2 waves per SIMD -- off), bit 1: the bf16 inference forward (0.426 -> 0.368 ms -- on)
int g_conv_cout1_d4 = 2;

extern int g_conv_split, g_conv_small, g_conv_small_wgs, g_conv_tr2pw;

struct ConvArgs {
    const float* x;         // [B,Di,Hi,Wi,Cin]
    const float* wp;        // packed weights
    float* y;               // [B,Do,Ho,Wo,Cout]
    const float* scale;     // [Cout] or null
    const float* shift;     // [Cout] or null (bias when scale == null)
    const float* skip;      // like y, or null (added after the ReLU)
    float* partials;        // [numWG][2][Cout] or null
    int relu;
    int B, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout;
    int QD, QH, QW;         // coarse-grid extents
    int ntd, nth, ntw;      // tiles per dim
    int nb_total;           // 16-wide Cout tiles in the packed weight image; a workgroup handles NB of them from blockIdx.y*NB
};

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                                int geom, int CC, int Cin, int Cout, int NB,
                                                                int layout, int flip, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    const int nb = (idx >> 8) % NB, kk = (idx >> 8) / NB;
    const int co = nb * 16 + (lane & 15);
    int ks, chunk = 0, cls = 0;
    if (geom == GEOM_TR2_PW) {
        // columns n = pw*8 + co (Cout == 8, NB == 1); classes (pd, ph); taps incl. the input offset dw
        int k0 = 0;
        for (cls = 0; cls < 4; ++cls) {
            int n = tr2p_ntaps(cls) * CC / 16;
            if (kk < k0 + n) break;
            k0 += n;
        }
        ks = kk - k0;
        const int kf = 16 * ks + 4 * (lane >> 4) + j;
        const int tp = kf / CC, cip = kf % CC, n = lane & 15, pw = n >> 3, cop = n & 7;
        int dd, dh, dw, kd, kh;
        tr2p_tap(cls, tp, dd, dh, dw, kd, kh);
        int kw = tr2p_kw(pw, dw);
        float v = 0.f;
        if (kw >= 0 && nb == 0 && cop < Cout) {
            if (flip) { kd = 2 - kd; kh = 2 - kh; kw = 2 - kw; }
            const int kidx = kd * 9 + kh * 3 + kw;
            v = layout == WL_OIK ? w[((size_t)cop * Cin + cip) * 27 + kidx] : w[((size_t)cip * Cout + cop) * 27 + kidx];
        }
        wp[idx] = v;
        return;
    }
    if (geom == GEOM_TR2) {
        int k0 = 0;
        for (cls = 0; cls < 8; ++cls) {
            int n = tr2_ntaps(cls) * CC / 16;
            if (kk < k0 + n) break;
            k0 += n;
        }
        ks = kk - k0;
    } else {
        const int KS = ksteps_for(27, CC);
        chunk = kk / KS;
        ks = kk % KS;
    }
    const int kflat = 16 * ks + 4 * (lane >> 4) + j;
    const int tap = kflat / CC, ci = chunk * CC + kflat % CC;
    int kd, kh, kw;
    bool valid = co < Cout;
    if (geom == GEOM_TR2) {
        int dd, dh, dw;
        tr2_tap(cls, tap, dd, dh, dw, kd, kh, kw);
    } else {
        valid = valid && tap < 27;
        kd = tap / 9; kh = (tap / 3) % 3; kw = tap % 3;
    }
    if (flip) { kd = 2 - kd; kh = 2 - kh; kw = 2 - kw; }
    float v = 0.f;
    if (valid) {
        const int kidx = kd * 9 + kh * 3 + kw;
        v = layout == WL_OIK ? w[((size_t)co * Cin + ci) * 27 + kidx] : w[((size_t)ci * Cout + co) * 27 + kidx];
    }
    wp[idx] = v;
}


// ------------------------------------------------------------------------------------------------
// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so
// xcd_block() gives every XCD one contiguous range of the tile order, and brick_tile() makes that order bricks
// of (all W tiles) x (4 H tiles) marching along D: tiles resident together on an XCD share their halos in L2.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_block(int bid, int nb) {
    const int per = nb >> 3, rem = nb & 7, x = bid & 7, idx = bid >> 3;
    return x * per + (x < rem ? x : rem) + idx;
}
__device__ __forceinline__ void brick_tile(int t, int ntw, int nth, int ntd, int& b, int& td, int& th, int& tw) {
    const int per_b = ntw * nth * ntd;
    b = t / per_b;
    int r = t - b * per_b;
    const int full = ntw * 4 * ntd;
    int g = r / full, gh = 4;
    if (g >= (nth >> 2)) { g = nth >> 2; gh = nth & 3; }
    r -= g * full;
    td = r / (ntw * gh);
    r -= td * (ntw * gh);
    th = g * 4 + r / ntw;
    tw = r % ntw;
}
__device__ __forceinline__ void linear_tile(int t, int ntw, int nth, int ntd, int& b, int& td, int& th, int& tw) {
    tw = t % ntw; t /= ntw;
    th = t % nth; t /= nth;
    td = t % ntd; t /= ntd;
    b = t;
}

// ------------------------------------------------------------------------------------------------
// LDS staging helpers.  A plain `for (i = tid; i < N; i += 256) lds[..] = global[..]` loop makes hipcc wait for each
// load before the next one (load -> s_waitcnt -> ds_write per iteration): at ~1-2 us of HBM latency per round
// trip and ~10 round trips per tile that serialised latency was the largest term of every conv kernel here.
// These helpers issue ALL loads of a batch first (registers), and write LDS afterwards; the split load / store
// form lets a kernel keep the next tile's loads in flight while the MFMAs of the current tile run.
// ------------------------------------------------------------------------------------------------
template <int NIT, class Map>
__device__ __forceinline__ void stage_load(float4 (&v)[NIT], int (&off)[NIT], int tid, int nitems, Map map) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = tid + 256 * k;
        off[k] = -1;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nitems) {
            const float* src = nullptr;
            map(i, src, off[k]);
            if (src) v[k] = *reinterpret_cast<const float4*>(src);
        }
    }
}
template <int NIT>
__device__ __forceinline__ void stage_store(float* __restrict__ lds, const float4 (&v)[NIT], const int (&off)[NIT]) {
#pragma unroll
    for (int k = 0; k < NIT; ++k)
        if (off[k] >= 0) *reinterpret_cast<float4*>(lds + off[k]) = v[k];
}
// one-shot: batches of <= 12 float4 per thread
template <int NITEMS, class Map>
__device__ __forceinline__ void stage_batched(float* __restrict__ lds, int tid, Map map) {
    constexpr int NIT = (NITEMS + 255) / 256;
    constexpr int BATCH = NIT < 12 ? NIT : 12;
#pragma unroll
    for (int k0 = 0; k0 < NIT; k0 += BATCH) {
        float4 v[BATCH];
        int off[BATCH];
        stage_load<BATCH>(v, off, tid, NITEMS - 256 * k0, [&](int i, const float*& src, int& o) { map(i + 256 * k0, src, o); });
        stage_store<BATCH>(lds, v, off);
    }
}

// ------------------------------------------------------------------------------------------------
// implicit-GEMM forward-style kernel (conv s1/s2, transposed s2; dgrads map onto these)
// ------------------------------------------------------------------------------------------------
// FS (fast staging, tuning knob "fs", not the default -- written after the Cout = 8 kernels gained 15 % from the same recipe,
// not yet measured here): tiles whose halo lies inside the volume skip the six bounds compares and the zero fill per float4
// and address the halo relative to one tile base pointer.
template <int GEOM, int CC, int NB, bool FS = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    using G = ConvGeom<GEOM>;
    constexpr int CCP = CC + 4;
    constexpr int NR = G::RD * G::RH * G::RW;
    constexpr int MB = G::MB;
    constexpr int CQ = CC / 4;
    __shared__ __attribute__((aligned(16))) float tile[NR * CCP];
    __shared__ int tapoff[32];
    __shared__ float red[4 * NB * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int nb0 = blockIdx.y * NB;   // first Cout tile of this workgroup (small layers split Cout over blockIdx.y)

    int t = blockIdx.x;
    const int tw = t % a.ntw; t /= a.ntw;
    const int th = t % a.nth; t /= a.nth;
    const int td = t % a.ntd; t /= a.ntd;
    const int b = t;
    const int qd0 = td * G::TQD, qh0 = th * G::TQH, qw0 = tw * G::TQW;

    if (tid < 32) {
        int off = 0;
        if (G::PW) {
            if (tid < 18) {
                int cls = 0, k0 = 0;
                for (; cls < 4; ++cls) {
                    int n = tr2p_ntaps(cls);
                    if (tid < k0 + n) break;
                    k0 += n;
                }
                int dd, dh, dw, kd, kh;
                tr2p_tap(cls, tid - k0, dd, dh, dw, kd, kh);
                off = ((dd * G::RH + dh) * G::RW + dw) * CCP;
            }
        } else if (G::BASE == GEOM_TR2) {
            if (tid < 27) {
                int cls = 0, k0 = 0;
                for (; cls < 8; ++cls) {
                    int n = tr2_ntaps(cls);
                    if (tid < k0 + n) break;
                    k0 += n;
                }
                int dd, dh, dw, kd, kh, kw;
                tr2_tap(cls, tid - k0, dd, dh, dw, kd, kh, kw);
                off = ((dd * G::RH + dh) * G::RW + dw) * CCP;
            }
        } else if (tid < 27) {
            off = (((tid / 9) * G::RH + (tid / 3) % 3) * G::RW + tid % 3) * CCP;
        }
        tapoff[tid] = off;
    }

    int baseA[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int f = wave * MB + mb;
        const int qd_l = f / G::TQH, qh_l = f % G::TQH;
        baseA[mb] = (((qd_l * G::IS) * G::RH + qh_l * G::IS) * G::RW + l15 * G::IS) * CCP;
    }

    f32x4 acc[MB][NB];
    float st1[NB], st2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { st1[nb] = 0.f; st2[nb] = 0.f; }

    const int nchunks = G::BASE == GEOM_TR2 ? 1 : a.Cin / CC;
    const int KSF = ksteps_for(27, CC);

    for (int cls = 0; cls < G::NCLS; ++cls) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int chunk = 0; chunk < nchunks; ++chunk) {
            if (cls == 0) {
                // ---- stage the input halo region (channels [chunk*CC, +CC)) into LDS ----
                __syncthreads();
                const int id0 = qd0 * G::IS - G::PAD, ih0 = qh0 * G::IS - G::PAD, iw0 = qw0 * G::IS - G::PAD;
                if (FS && id0 >= 0 && id0 + G::RD <= a.Di && ih0 >= 0 && ih0 + G::RH <= a.Hi && iw0 >= 0 && iw0 + G::RW <= a.Wi) {
                    const float* __restrict__ base = a.x + ((((size_t)b * a.Di + id0) * a.Hi + ih0) * a.Wi + iw0) * a.Cin + chunk * CC;
                    stage_batched<NR * CQ>(tile, tid, [&](int i, const float*& src, int& o) {
                        const int vox = i / CQ, cq = i % CQ;
                        const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
                        o = vox * CCP + 4 * cq;
                        src = base + ((rd * a.Hi + rh) * a.Wi + rw) * a.Cin + 4 * cq;
                    });
                } else
                stage_batched<NR * CQ>(tile, tid, [&](int i, const float*& src, int& o) {
                    const int vox = i / CQ, cq = i % CQ;
                    const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
                    const int id = qd0 * G::IS + rd - G::PAD, ih = qh0 * G::IS + rh - G::PAD,
                              iw = qw0 * G::IS + rw - G::PAD;
                    o = vox * CCP + 4 * cq;
                    if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                        src = a.x + ((((size_t)b * a.Di + id) * a.Hi + ih) * a.Wi + iw) * a.Cin + chunk * CC + 4 * cq;
                });
                __syncthreads();
            }
            int KS, kk0, tapbase;
            if (G::PW) {
                KS = tr2p_ntaps(cls) * CC / 16;
                tapbase = tr2p_tap_prefix(cls);
                kk0 = tapbase * CC / 16;
            } else if (G::BASE == GEOM_TR2) {
                KS = tr2_ntaps(cls) * CC / 16;
                tapbase = tr2_tap_prefix(cls);
                kk0 = tapbase * CC / 16;
            } else {
                KS = KSF;
                tapbase = 0;
                kk0 = chunk * KSF;
            }
            // software pipeline: the weight operands (global memory / L2, ~1 us away) run PD k-steps ahead of the MFMAs
            // that consume them, the LDS operands one k-step.  A one-wave-per-SIMD launch (the deep U-Net levels: fewer
            // workgroups than CUs) has nothing else to hide that latency: with PD = 1 those layers spent two thirds of
            // their time waiting on the next 256-byte weight fragment (profiles/r01_run17_bench_kernel_stats.csv).
            // (measured: 64>64 at 24x16x20 0.073 -> 0.061 ms with PD = 4; conv0's dgrad, NB = 2: 0.556 ms with PD = 2 against 0.60
            //  with 1; the transposed geometry got slower with PD > 1 and keeps 1)
            constexpr int PD = G::BASE == GEOM_TR2 ? 1 : (NB == 1 ? 4 : 2);
            float4 bq[PD][NB], af[MB];
            auto load_b = [&](int ks, float4 (&dst)[NB]) {
                const int kc = ks < KS ? ks : KS - 1;   // past the end: re-read the last fragment (harmless)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    dst[nb] = *reinterpret_cast<const float4*>(a.wp + (((size_t)(kk0 + kc) * a.nb_total + nb0 + nb) * 64 + lane) * 4);
            };
            auto load_a = [&](int ks, float4 (&dst)[MB]) {
                const int kc = ks < KS ? ks : KS - 1;
                const int kflat = 16 * kc + 4 * g;
                const int aoff = tapoff[tapbase + kflat / CC] + kflat % CC;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) dst[mb] = *reinterpret_cast<const float4*>(&tile[baseA[mb] + aoff]);
            };
#pragma unroll
            for (int u = 0; u < PD; ++u) load_b(u, bq[u]);
            load_a(0, af);
            for (int ks0 = 0; ks0 < KS; ks0 += PD) {
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int ks = ks0 + u;
                    if (ks < KS) {
                        float4 an[MB];
                        load_a(ks + 1, an);
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                acc[mb][nb] = MVS_MFMA_16x16x4(af[mb].x, bq[u][nb].x, acc[mb][nb]);
                                acc[mb][nb] = MVS_MFMA_16x16x4(af[mb].y, bq[u][nb].y, acc[mb][nb]);
                                acc[mb][nb] = MVS_MFMA_16x16x4(af[mb].z, bq[u][nb].z, acc[mb][nb]);
                                acc[mb][nb] = MVS_MFMA_16x16x4(af[mb].w, bq[u][nb].w, acc[mb][nb]);
                            }
                        load_b(ks + PD, bq[u]);
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) af[mb] = an[mb];
                    }
                }
            }
        }

        // ---- epilogue for this class: D layout col = lane&15 (co), row = 4*(lane>>4)+r (position along qw) ----
        // (PW: class = (pd, ph); the column index l15 = pw*8 + co carries the W parity -> 16 consecutive floats per voxel pair)
        const int pd = G::PW ? (cls >> 1) & 1 : (cls >> 2) & 1, ph = G::PW ? cls & 1 : (cls >> 1) & 1, pw = G::PW ? (l15 >> 3) : cls & 1;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int f = wave * MB + mb;
            const int qd = qd0 + f / G::TQH, qh = qh0 + f % G::TQH;
            if (qd >= a.QD || qh >= a.QH) continue;
            const int od = qd * G::OS + pd, oh = qh * G::OS + ph;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qw = qw0 + 4 * g + r;
                if (qw >= a.QW) continue;
                const int ow = qw * G::OS + pw;
                const size_t obase = ((((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.Cout;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int co = G::PW ? (l15 & 7) : (nb0 + nb) * 16 + l15;
                    if (co >= a.Cout) continue;
                    float v = acc[mb][nb][r];
                    st1[nb] += v;
                    st2[nb] += v * v;
                    if (a.scale) v = v * a.scale[co] + a.shift[co];
                    else if (a.shift) v = v + a.shift[co];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (a.skip) v += a.skip[obase + co];
                    a.y[obase + co] = v;
                }
            }
        }
    }

    if (a.partials) {
        // reduce the 4 lane groups of the wave, then the 4 waves, -> partials[block][2][Cout]
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float s1 = st1[nb], s2 = st2[nb];
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            if (G::PW) { s1 += __shfl_xor(s1, 8); s2 += __shfl_xor(s2, 8); }   // columns n and n ^ 8 are the same channel (pw = 0 / 1)
            if (lane < 16) {
                red[((wave * NB + nb) * 16 + lane) * 2 + 0] = s1;
                red[((wave * NB + nb) * 16 + lane) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < 2 * NB * 16) {
            const int stat = tid / (NB * 16), n = tid % (NB * 16);
            if (nb0 * 16 + n < a.Cout) {
                float s = 0.f;
                for (int w = 0; w < 4; ++w) s += red[(w * NB * 16 + n) * 2 + stat];
                a.partials[((size_t)blockIdx.x * 2 + stat) * a.Cout + nb0 * 16 + n] = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Stride-1 implicit GEMM with PERSISTENT workgroups (round 3; layers with many tiles: conv0's input gradient, the 16->16 layers at
// L1, CVP's full-resolution layers).  The one-tile-per-workgroup kernel above exposes every tile's halo staging (global -> registers
// -> LDS, ~2 us of latency) to the MFMA pipe unless another resident workgroup happens to be in its k-loop, and the last round of a
// launch runs half empty (1920 tiles on 768 slots).  Here a workgroup walks tiles t, t + G, t + 2G, ... in the XCD-aware brick order
// and keeps the NEXT chunk's / tile's halo in flight in registers while the MFMAs of the current chunk run (the recipe of the
// Cout = 8 kernels below: 15 % there).  Same arithmetic and the same k-order as conv_igemm_kernel<GEOM_S1>: bit-identical output,
// one BatchNorm partial row per TILE (not per workgroup) so that the statistics buffer is the same.
// ------------------------------------------------------------------------------------------------
template <int CC, int NB>
__global__ __launch_bounds__(256) MVS_MIN_WAVES_PER_SIMD((NB == 4 ? 2 : 3)) void conv_igemm_s1p_kernel(ConvArgs a, int xcd) {
    using G = ConvGeom<GEOM_S1>;
    constexpr int CCP = CC + 4, CQ = CC / 4, MB = G::MB;
    constexpr int NR = G::RD * G::RH * G::RW;
    __shared__ __attribute__((aligned(16))) float tile[NR * CCP];
    __shared__ int tapoff[32];
    __shared__ float red[4 * NB * 16 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    const int vb = xcd ? xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    if (tid < 32) tapoff[tid] = tid < 27 ? (((tid / 9) * G::RH + (tid / 3) % 3) * G::RW + tid % 3) * CCP : 0;
    int baseA[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int f = wave * MB + mb;
        baseA[mb] = (((f / G::TQH) * G::RH + f % G::TQH) * G::RW + l15) * CCP;
    }
    auto tile_origin = [&](int t, int& b, int& qd0, int& qh0, int& qw0) {
        int td, th, tw;
        if (xcd) brick_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        else linear_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        qd0 = td * G::TQD; qh0 = th * G::TQH; qw0 = tw * G::TQW;
    };
    // halo tile -> registers (all loads issued back to back; zero outside the volume), registers -> LDS; the offset of item k relative
    // to the tile's origin voxel does not depend on the tile; interior tiles skip the per-item bounds checks
    constexpr int XIT = (NR * CQ + 255) / 256;
    float4 xv[XIT];
    // (the offsets are recomputed per chunk: a dozen integer operations per item against 11 registers held across the MFMA loop)
    auto rel_of = [&](int i) {
        const int vox = i / CQ, cq = i % CQ;
        const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
        return (((rd - 1) * a.Hi + (rh - 1)) * a.Wi + (rw - 1)) * a.Cin + 4 * cq;
    };
    auto load_chunk = [&](int b, int qd0, int qh0, int qw0, int chunk) {
        const float* __restrict__ base = a.x + ((((size_t)b * a.Di + qd0) * a.Hi + qh0) * a.Wi + qw0) * a.Cin + chunk * CC;
        const bool interior = qd0 >= 1 && qd0 + G::TQD + 1 <= a.Di && qh0 >= 1 && qh0 + G::TQH + 1 <= a.Hi &&
                              qw0 >= 1 && qw0 + G::TQW + 1 <= a.Wi;
        if (interior) {
#pragma unroll
            for (int k = 0; k < XIT; ++k)
                if (tid + 256 * k < NR * CQ) xv[k] = *reinterpret_cast<const float4*>(base + rel_of(tid + 256 * k));
        } else {
#pragma unroll
            for (int k = 0; k < XIT; ++k) {
                const int i = tid + 256 * k;
                xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < NR * CQ) {
                    const int vox = i / CQ;
                    const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
                    const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
                    if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                        xv[k] = *reinterpret_cast<const float4*>(base + rel_of(i));
                }
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < XIT; ++k) {
            const int i = tid + 256 * k;
            if (i < NR * CQ) *reinterpret_cast<float4*>(&tile[(i / CQ) * CCP + 4 * (i % CQ)]) = xv[k];
        }
    };
    const int nchunks = a.Cin / CC;
    const int KS = ksteps_for(27, CC);
    int b, qd0, qh0, qw0;
    if (vb < ntiles) {
        tile_origin(vb, b, qd0, qh0, qw0);
        load_chunk(b, qd0, qh0, qw0, 0);
    }
    for (int t = vb; t < ntiles; t += gridDim.x) {
        tile_origin(t, b, qd0, qh0, qw0);
        f32x4 acc[MB][NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            __syncthreads();                                   // the previous chunk's / tile's reads of the LDS image are done
            store_chunk();
            __syncthreads();
            if (chunk + 1 < nchunks) load_chunk(b, qd0, qh0, qw0, chunk + 1);      // in flight while this chunk's MFMAs run
            else if (t + (int)gridDim.x < ntiles) {
                int b2, d2, h2, w2;
                tile_origin(t + gridDim.x, b2, d2, h2, w2);
                load_chunk(b2, d2, h2, w2, 0);
            }
            const int kk0 = chunk * KS;
            constexpr int PD = 2;                              // weight fragments PD k-steps ahead (see conv_igemm_kernel)
            float4 bq[PD][NB], af[MB];
            auto load_b = [&](int ks, float4 (&dst)[NB]) {
                const int kc = ks < KS ? ks : KS - 1;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    dst[nb] = *reinterpret_cast<const float4*>(a.wp + (((size_t)(kk0 + kc) * a.nb_total + nb) * 64 + lane) * 4);
            };
            auto load_a = [&](int ks, float4 (&dst)[MB]) {
                const int kc = ks < KS ? ks : KS - 1;
                const int kflat = 16 * kc + 4 * g;
                const int aoff = tapoff[kflat / CC] + kflat % CC;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) dst[mb] = *reinterpret_cast<const float4*>(&tile[baseA[mb] + aoff]);
            };
#pragma unroll
            for (int u = 0; u < PD; ++u) load_b(u, bq[u]);
            load_a(0, af);
            for (int ks0 = 0; ks0 < KS; ks0 += PD) {
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int ks = ks0 + u;
                    if (ks < KS) {
                        float4 an[MB];
                        load_a(ks + 1, an);
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                acc[mb][nb] = MVS_MFMA_16x16x4(af[mb].x, bq[u][nb].x, acc[mb][nb]);
                                acc[mb][nb] = MVS_MFMA_16x16x4(af[mb].y, bq[u][nb].y, acc[mb][nb]);
                                acc[mb][nb] = MVS_MFMA_16x16x4(af[mb].z, bq[u][nb].z, acc[mb][nb]);
                                acc[mb][nb] = MVS_MFMA_16x16x4(af[mb].w, bq[u][nb].w, acc[mb][nb]);
                            }
                        load_b(ks + PD, bq[u]);
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) af[mb] = an[mb];
                    }
                }
            }
        }
        // ---- epilogue: D layout col = lane&15 (co), row = 4*(lane>>4)+r (position along qw) ----
        float st1[NB], st2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { st1[nb] = 0.f; st2[nb] = 0.f; }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int f = wave * MB + mb;
            const int qd = qd0 + f / G::TQH, qh = qh0 + f % G::TQH;
            if (qd >= a.QD || qh >= a.QH) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qw = qw0 + 4 * g + r;
                if (qw >= a.QW) continue;
                const size_t obase = ((((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qw) * a.Cout;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int co = nb * 16 + l15;
                    if (co >= a.Cout) continue;
                    float v = acc[mb][nb][r];
                    st1[nb] += v;
                    st2[nb] += v * v;
                    if (a.scale) v = v * a.scale[co] + a.shift[co];
                    else if (a.shift) v = v + a.shift[co];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (a.skip) v += a.skip[obase + co];
                    a.y[obase + co] = v;
                }
            }
        }
        if (a.partials) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float s1 = st1[nb], s2 = st2[nb];
                s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
                if (lane < 16) {
                    red[((wave * NB + nb) * 16 + lane) * 2 + 0] = s1;
                    red[((wave * NB + nb) * 16 + lane) * 2 + 1] = s2;
                }
            }
            __syncthreads();
            if (tid < 2 * NB * 16) {
                const int stat = tid / (NB * 16), n = tid % (NB * 16);
                if (n < a.Cout) {
                    float sm = 0.f;
                    for (int w = 0; w < 4; ++w) sm += red[(w * NB * 16 + n) * 2 + stat];
                    a.partials[((size_t)t * 2 + stat) * a.Cout + n] = sm;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Cin == 1 (input gradient of the Cout=1 probability layer): direct form, one thread per voxel.
// y[v][co] = sum_t x[v + t - 1] * wt[t][co]   (wt already flipped / transposed by the caller-side packer)
// ------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv_cin1_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                        float* __restrict__ y, int B, int D, int H, int W) {
    __shared__ __attribute__((aligned(16))) float ws[27 * COUT];
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) ws[i] = wt[i];
    __syncthreads();
    const size_t total = (size_t)B * D * H * W;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= total) return;
    const int w_ = (int)(v % W), h_ = (int)((v / W) % H), d_ = (int)((v / ((size_t)W * H)) % D);
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
    for (int t = 0; t < 27; ++t) {
        const int dd = d_ + t / 9 - 1, hh = h_ + (t / 3) % 3 - 1, ww = w_ + t % 3 - 1;
        if (dd < 0 || dd >= D || hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
        const float xv = x[v + ((long long)(t / 9 - 1) * H + ((t / 3) % 3 - 1)) * W + (t % 3 - 1)];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = fmaf(xv, ws[t * COUT + c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < COUT; c += 4)
        *reinterpret_cast<float4*>(y + v * COUT + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
}

// The same with a column of FOUR depth slices per thread (knob "cout1_d4"): each of the 6 x 9 input values under the column is
// loaded once and feeds up to three outputs; the 27 taps are unrolled, the bounds tests are three small per-axis tables instead of
// six compares per tap, and the weights are read at compile-time offsets of a kernel-argument pointer (scalar loads, SGPR operands
// of the FMAs) instead of two LDS reads per tap.
template <int COUT>
__global__ __launch_bounds__(256) void conv_cin1_d4_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                           float* __restrict__ y, int B, int D, int H, int W) {
    const int D4 = (D + 3) / 4;
    const size_t total = (size_t)B * D4 * H * W;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= total) return;
    const int w_ = (int)(v % W), h_ = (int)((v / W) % H), dq = (int)((v / ((size_t)W * H)) % D4), b = (int)(v / ((size_t)W * H * D4));
    const int d0 = 4 * dq;
    // per-axis clamped coordinates and validity of the 6 / 3 / 3 input positions
    int dc[6], hc[3], wc[3];
    bool vd[6], vh[3], vw[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) { const int d = d0 + i - 1; vd[i] = d >= 0 && d < D; dc[i] = min(max(d, 0), D - 1) * H * W; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int h = h_ + i - 1, w = w_ + i - 1;
        vh[i] = h >= 0 && h < H; hc[i] = min(max(h, 0), H - 1) * W;
        vw[i] = w >= 0 && w < W; wc[i] = min(max(w, 0), W - 1);
    }
    const float* __restrict__ xb = x + (size_t)b * D * H * W;
    float acc[4][COUT];
#pragma unroll
    for (int pd = 0; pd < 4; ++pd)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[pd][c] = 0.f;
#pragma unroll
    for (int khw = 0; khw < 9; ++khw) {
        const int kh = khw / 3, kw = khw % 3;
        const bool vhw = vh[kh] && vw[kw];
        const int col = hc[kh] + wc[kw];
#pragma unroll
        for (int din = 0; din < 6; ++din) {
            float xv = xb[dc[din] + col];
            xv = (vhw && vd[din]) ? xv : 0.f;
#pragma unroll
            for (int pd = 0; pd < 4; ++pd) {
                const int kd = din - pd;
                if (kd < 0 || kd > 2) continue;
                const int t = (kd * 3 + kh) * 3 + kw;
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[pd][c] = fmaf(xv, wt[t * COUT + c], acc[pd][c]);
            }
        }
    }
#pragma unroll
    for (int pd = 0; pd < 4; ++pd) {
        if (d0 + pd >= D) continue;
        float* __restrict__ o = y + ((((size_t)b * D + d0 + pd) * H + h_) * W + w_) * COUT;
#pragma unroll
        for (int c = 0; c < COUT; c += 4) *reinterpret_cast<float4*>(o + c) = make_float4(acc[pd][c], acc[pd][c + 1], acc[pd][c + 2], acc[pd][c + 3]);
    }
}

// wt[t][co] = W[0][co][2-kd][2-kh][2-kw]   (W is [1][Cin][3][3][3], OIK with O == 1)
__global__ void conv_cin1_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 27 * C) return;
    const int t = i / C, c = i % C;
    wt[i] = w[(size_t)c * 27 + (26 - t)];
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  dW[tap][ci][co] = sum_{b,o} X[b, o*S + tap - 1][ci] * G[b,o][co]
// GEMM view: M = ci (CC=16) or (tap pair, ci) (CC=8), N = co, K = positions.  Persistent workgroups
// walk tiles, keep dW for their (ci chunk, co chunk) in MFMA accumulators, and emit one partial
// image each; wgrad_reduce sums the partial images deterministically.
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;      // [B,Di,Hi,Wi,CX]
    const float* g;      // [B,QD,QH,QW,CG]
    float* part;         // [gridDim.x][27][CX][CG]
    int B, Di, Hi, Wi, CX, CG;
    int QD, QH, QW, ntd, nth, ntw;
    int xcd;             // XCD-aware tile order (conv_c8_wgrad_kernel)
};

template <int GEOM, int CC, int NBW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    using G = ConvGeom<GEOM>;
    // voxel stride in LDS: a half-wave of the A read (ds_read_b32) covers two consecutive positions x 16 channels;
    // the two 16-bank windows are disjoint iff the position stride is 16 mod 32 floats: 16 for stride 1 (stride 2
    // would need 24, which does not fit the 160 KB LDS next to the G tile, so it keeps the padded 20)
    constexpr int CCP = (CC == 16 && G::IS == 1) ? 16 : CC + 4;
    constexpr int NR = G::RD * G::RH * G::RW;
    constexpr int NPOS = G::TQD * G::TQH * G::TQW;
    constexpr int COP = NBW * 16 + ((NBW % 2 == 0) ? 16 : 0);
    constexpr int CQ = CC / 4;
    constexpr int NSLOT = CC == 16 ? 27 : 14;   // CC==8: a slot is a pair of taps (2s, 2s+1)
    constexpr int SPW = (NSLOT + 3) / 4;        // slots per wave
    __shared__ __attribute__((aligned(16))) float xt[NR * CCP];
    __shared__ __attribute__((aligned(16))) float gt[NPOS * COP];
    __shared__ int tapoff[32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kpos = lane >> 4, l15 = lane & 15;
    const int chunk = blockIdx.y, cobase = blockIdx.z * NBW * 16;

    if (tid < 32) tapoff[tid] = tid < 27 ? (((tid / 9) * G::RH + (tid / 3) % 3) * G::RW + tid % 3) * CCP : 0;

    f32x4 acc[SPW][NBW];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[s][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    __syncthreads();
    int toff[SPW];   // LDS offset of this wave's tap slots (per lane for CC == 8: two taps share a fragment)
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int slot = wave + 4 * s;
        toff[s] = slot < NSLOT ? (CC == 16 ? tapoff[slot] + l15 : tapoff[2 * slot + (l15 >> 3)] + (l15 & 7)) : 0;
    }

    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    // next tile's X region and G tile are fetched into registers while this tile's MFMAs run (when they fit: the
    // stride-2 / 16-channel case needs 24 float4 per thread and stages without the overlap)
    constexpr int XIT = (NR * CQ + 255) / 256;
    constexpr int NG4 = NPOS * NBW * 4;                 // float4 of the G tile (CG % 4 == 0 path)
    constexpr int GIT = (NG4 + 255) / 256;
    constexpr bool PREFETCH = XIT <= 12;
    constexpr int XR = PREFETCH ? XIT : 1, GR = PREFETCH ? GIT : 1;
    float4 xv[XR], gv[GR];
    int xo[XR], go[GR];
    const bool g_vec = (a.CG % 4) == 0;
    auto tile_origin = [&](int tile, int& b, int& qd0, int& qh0, int& qw0) {
        int t = tile;
        const int tw = t % a.ntw; t /= a.ntw;
        const int th = t % a.nth; t /= a.nth;
        const int td = t % a.ntd; t /= a.ntd;
        b = t; qd0 = td * G::TQD; qh0 = th * G::TQH; qw0 = tw * G::TQW;
    };
    auto xmap = [&](int b, int qd0, int qh0, int qw0) {
        return [=, &a](int i, const float*& src, int& o) {
            const int vox = i / CQ, cq = i % CQ;
            const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
            const int id = qd0 * G::IS + rd - 1, ih = qh0 * G::IS + rh - 1, iw = qw0 * G::IS + rw - 1;
            o = vox * CCP + 4 * cq;
            if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                src = a.x + ((((size_t)b * a.Di + id) * a.Hi + ih) * a.Wi + iw) * a.CX + chunk * CC + 4 * cq;
        };
    };
    auto gmap = [&](int b, int qd0, int qh0, int qw0) {
        return [=, &a](int i, const float*& src, int& o) {
            const int p = i / (NBW * 4), n4 = i % (NBW * 4);
            const int qw = qw0 + p % G::TQW, qh = qh0 + (p / G::TQW) % G::TQH, qd = qd0 + p / (G::TQW * G::TQH);
            const int co = cobase + 4 * n4;
            o = p * COP + 4 * n4;
            if (qd < a.QD && qh < a.QH && qw < a.QW && co < a.CG)
                src = a.g + ((((size_t)b * a.QD + qd) * a.QH + qh) * a.QW + qw) * a.CG + co;
        };
    };
    auto stage_g_scalar = [&](int b, int qd0, int qh0, int qw0) {   // CG not a multiple of 4
        for (int i = tid; i < NPOS * NBW * 16; i += 256) {
            const int p = i / (NBW * 16), n = i % (NBW * 16);
            const int qw = qw0 + p % G::TQW, qh = qh0 + (p / G::TQW) % G::TQH, qd = qd0 + p / (G::TQW * G::TQH);
            const int co = cobase + n;
            float v = 0.f;
            if (qd < a.QD && qh < a.QH && qw < a.QW && co < a.CG)
                v = a.g[((((size_t)b * a.QD + qd) * a.QH + qh) * a.QW + qw) * a.CG + co];
            gt[p * COP + n] = v;
        }
    };
    if (PREFETCH && g_vec && (int)blockIdx.x < ntiles) {
        int b, qd0, qh0, qw0;
        tile_origin(blockIdx.x, b, qd0, qh0, qw0);
        stage_load<XR>(xv, xo, tid, NR * CQ, xmap(b, qd0, qh0, qw0));
        stage_load<GR>(gv, go, tid, NG4, gmap(b, qd0, qh0, qw0));
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b, qd0, qh0, qw0;
        tile_origin(tile, b, qd0, qh0, qw0);
        __syncthreads();
        if (PREFETCH && g_vec) {
            stage_store<XR>(xt, xv, xo);
            stage_store<GR>(gt, gv, go);
            __syncthreads();
            if (tile + (int)gridDim.x < ntiles) {
                int b2, d2, h2, w2;
                tile_origin(tile + gridDim.x, b2, d2, h2, w2);
                stage_load<XR>(xv, xo, tid, NR * CQ, xmap(b2, d2, h2, w2));
                stage_load<GR>(gv, go, tid, NG4, gmap(b2, d2, h2, w2));
            }
        } else {
            stage_batched<NR * CQ>(xt, tid, xmap(b, qd0, qh0, qw0));
            if (g_vec) stage_batched<NG4>(gt, tid, gmap(b, qd0, qh0, qw0));
            else stage_g_scalar(b, qd0, qh0, qw0);
            __syncthreads();
        }
        for (int ks = 0; ks < NPOS / 4; ++ks) {
            const int p = 4 * ks + kpos;
            const int pw_ = p % G::TQW, ph_ = (p / G::TQW) % G::TQH, pd_ = p / (G::TQW * G::TQH);
            const int xoff = (((pd_ * G::IS) * G::RH + ph_ * G::IS) * G::RW + pw_ * G::IS) * CCP;
            float bfr[NBW];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) bfr[nb] = gt[p * COP + nb * 16 + l15];
            // straight-line: all A reads first, then all MFMAs (a slot past NSLOT computes unused values from
            // offset 0 instead of branching, which would serialise read -> MFMA)
            float av[SPW];
#pragma unroll
            for (int s = 0; s < SPW; ++s) av[s] = xt[xoff + toff[s]];
#pragma unroll
            for (int s = 0; s < SPW; ++s)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[s][nb] = MVS_MFMA_16x16x4(av[s], bfr[nb], acc[s][nb]);
        }
    }
    // D layout: col = lane&15 -> co, row = 4*(lane>>4)+r -> M index
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int slot = wave + 4 * s;
        if (slot >= NSLOT) continue;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int co = cobase + nb * 16 + l15;
            if (co >= a.CG) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * kpos + r;
                int tap, ci;
                if (CC == 16) { tap = slot; ci = chunk * 16 + i; }
                else { tap = 2 * slot + (i >> 3); ci = i & 7; }
                if (tap < 27) a.part[(((size_t)blockIdx.x * 27 + tap) * a.CX + ci) * a.CG + co] = acc[s][nb][r];
            }
        }
    }
}

// out (OIK: [CG][CX][27]) = sum over partial images;  part index [p][tap][cx][cg]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int CX,
                                                                int CG, float* __restrict__ gw) {
    const int n = 27 * CX * CG;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const size_t stride = (size_t)n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 3 < nparts; p += 4) {
        s0 += part[(size_t)p * stride + e];
        s1 += part[(size_t)(p + 1) * stride + e];
        s2 += part[(size_t)(p + 2) * stride + e];
        s3 += part[(size_t)(p + 3) * stride + e];
    }
    for (; p < nparts; ++p) s0 += part[(size_t)p * stride + e];
    const int cg = e % CG, cx = (e / CG) % CX, tap = e / (CG * CX);
    gw[((size_t)cg * CX + cx) * 27 + tap] = (s0 + s1) + (s2 + s3);
}

// ------------------------------------------------------------------------------------------------
// Cout == 1 (the probability layer, mvsnet.py:63 / network.py:65): a 16-wide MFMA N tile would be
// 1/16 used, so this layer runs as a direct VALU convolution: one thread per output voxel, input halo
// tile in LDS, weights broadcast from LDS.  HBM bound (reads the 8/16-channel activation once).
// ------------------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvArgs a, const float* __restrict__ w) {
    using G = ConvGeom<GEOM_S1>;
    constexpr int CCP = CIN + 4;
    constexpr int NR = G::RD * G::RH * G::RW;
    constexpr int CQ = CIN / 4;
    __shared__ __attribute__((aligned(16))) float tile[NR * CCP];
    __shared__ __attribute__((aligned(16))) float wl[27 * CIN];   // [tap][ci]
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tw = t % a.ntw; t /= a.ntw;
    const int th = t % a.nth; t /= a.nth;
    const int td = t % a.ntd; t /= a.ntd;
    const int b = t;
    const int qd0 = td * G::TQD, qh0 = th * G::TQH, qw0 = tw * G::TQW;
    for (int i = tid; i < 27 * CIN; i += 256) wl[i] = w[(size_t)(i % CIN) * 27 + i / CIN];   // W[0][ci][tap]
    stage_batched<NR * CQ>(tile, tid, [&](int i, const float*& src, int& o) {
        const int vox = i / CQ, cq = i % CQ;
        const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
        const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
        o = vox * CCP + 4 * cq;
        if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
            src = a.x + ((((size_t)b * a.Di + id) * a.Hi + ih) * a.Wi + iw) * CIN + 4 * cq;
    });
    __syncthreads();
    const int pw = tid % G::TQW, ph = (tid / G::TQW) % G::TQH, pd = tid / (G::TQW * G::TQH);
    const int base = ((pd * G::RH + ph) * G::RW + pw) * CCP;
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int off = base + (((tap / 9) * G::RH + (tap / 3) % 3) * G::RW + tap % 3) * CCP;
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
            const float4 xv = *reinterpret_cast<const float4*>(&tile[off + 4 * cq]);
            const float4 wv = *reinterpret_cast<const float4*>(&wl[tap * CIN + 4 * cq]);
            acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
        }
    }
    const int qd = qd0 + pd, qh = qh0 + ph, qw = qw0 + pw;
    if (qd < a.QD && qh < a.QH && qw < a.QW) {
        const size_t o = (((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qw;
        float v = acc;
        if (a.scale) v = v * a.scale[0] + a.shift[0];
        else if (a.shift) v = v + a.shift[0];
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.skip) v += a.skip[o];
        a.y[o] = v;
    }
}

// The same layer with FOUR outputs per thread (a column of 4 depth slices): the kernel above is LDS-bandwidth bound -- every output
// reads its 27 x CIN inputs and the 27 x CIN weights from LDS (108 ds_read_b128 per output at CIN = 8: 0.086 ms of LDS time for the
// 3.9 M voxels of MVSNet's probability layer, measured 0.095).  A thread that owns outputs d .. d+3 of one (h, w) reads each of the 6
// input slices under a (kh, kw) offset once and each weight once per four outputs: 40 reads per output instead of 108 (measured: no
// gain, see the knob).  The halo is
// kept as one plane per channel quad ([cq][voxel] float4), so the 16 lanes along W read consecutive 16-byte words: no padding, no
// bank conflicts.  Tile 4 x 8 x 16 outputs, 128 threads, halo 6 x 10 x 18.  Knob "cout1_d4" (1 = this form).
template <int CIN>
__global__ __launch_bounds__(128) void conv_cout1_d4_kernel(ConvArgs a, const float* __restrict__ w) {
    constexpr int TD = 4, TH = 8, TW = 16, RD = TD + 2, RH = TH + 2, RW = TW + 2, NR = RD * RH * RW, CQ = CIN / 4;
    __shared__ float4 tile[CQ * NR];      // [cq][voxel]
    __shared__ float4 wl[27 * CQ];        // [tap][cq]
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tw = t % a.ntw; t /= a.ntw;
    const int th = t % a.nth; t /= a.nth;
    const int td = t % a.ntd; t /= a.ntd;
    const int b = t;
    const int qd0 = td * TD, qh0 = th * TH, qw0 = tw * TW;
    if (tid < 27 * CQ) {
        const int tap = tid / CQ, cq = tid % CQ;   // W[0][ci][tap]
        wl[tid] = make_float4(w[(size_t)(4 * cq) * 27 + tap], w[(size_t)(4 * cq + 1) * 27 + tap], w[(size_t)(4 * cq + 2) * 27 + tap],
                              w[(size_t)(4 * cq + 3) * 27 + tap]);
    }
    // halo: all loads of a batch first, then the LDS writes (zero outside the volume)
    constexpr int NITEMS = NR * CQ, NIT = (NITEMS + 127) / 128, BATCH = NIT < 12 ? NIT : 12;
#pragma unroll
    for (int k0 = 0; k0 < NIT; k0 += BATCH) {
        float4 v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int i = tid + 128 * (k0 + k);
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + k < NIT && i < NITEMS) {
                const int vox = i / CQ, cq = i % CQ;
                const int rw = vox % RW, rh = (vox / RW) % RH, rd = vox / (RW * RH);
                const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
                if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                    v[k] = *reinterpret_cast<const float4*>(a.x + ((((size_t)b * a.Di + id) * a.Hi + ih) * a.Wi + iw) * CIN + 4 * cq);
            }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int i = tid + 128 * (k0 + k);
            if (k0 + k < NIT && i < NITEMS) tile[(i % CQ) * NR + i / CQ] = v[k];
        }
    }
    __syncthreads();
    const int pw = tid % TW, ph = tid / TW;
    float acc[TD];
#pragma unroll
    for (int pd = 0; pd < TD; ++pd) acc[pd] = 0.f;
    // NOT unrolled over (kh, kw) and the channel-quad pairs: fully unrolled, hipcc hoists all 54 x CQ tile reads to the top and
    // runs out of registers (256 VGPRs + scratch); one iteration = 12 reads + 6 weight reads + 96 FMAs
#pragma unroll 1
    for (int khw = 0; khw < 9; ++khw) {
        const int kh = khw / 3, kw = khw % 3;
        const int col = (ph + kh) * RW + pw + kw;
#pragma unroll 1
        for (int cq0 = 0; cq0 < CQ; cq0 += 2) {
            float4 wv[3][2];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int c = 0; c < 2; ++c) wv[kd][c] = wl[((kd * 3 + kh) * 3 + kw) * CQ + cq0 + c];
            float4 xv[RD][2];
#pragma unroll
            for (int din = 0; din < RD; ++din)
#pragma unroll
                for (int c = 0; c < 2; ++c) xv[din][c] = tile[(cq0 + c) * NR + din * RH * RW + col];
#pragma unroll
            for (int din = 0; din < RD; ++din)
#pragma unroll
                for (int pd = 0; pd < TD; ++pd) {
                    const int kd = din - pd;
                    if (kd < 0 || kd > 2) continue;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        acc[pd] = fmaf(xv[din][c].x, wv[kd][c].x, acc[pd]); acc[pd] = fmaf(xv[din][c].y, wv[kd][c].y, acc[pd]);
                        acc[pd] = fmaf(xv[din][c].z, wv[kd][c].z, acc[pd]); acc[pd] = fmaf(xv[din][c].w, wv[kd][c].w, acc[pd]);
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
            if (a.scale) v = v * a.scale[0] + a.shift[0];
            else if (a.shift) v = v + a.shift[0];
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.skip) v += a.skip[o];
            a.y[o] = v;
        }
    }
}

// dW[tap][cx] = sum_pos X[pos + tap - 1][cx] * g[pos]   (CG == 1, stride 1): thread = one (tap, cx) output
// (two for CX == 16), persistent over tiles; X halo tile and g tile in LDS.  Partial image per workgroup
// in the generic layout [group][27][CX][1] so conv_wgrad_reduce_kernel finishes it.
template <int CX>
__global__ __launch_bounds__(256) void conv_wgrad_cg1_kernel(WgradArgs a) {
    using G = ConvGeom<GEOM_S1>;
    constexpr int CCP = CX + 1;
    constexpr int NR = G::RD * G::RH * G::RW;
    constexpr int NPOS = G::TQD * G::TQH * G::TQW;
    constexpr int NOUT = 27 * CX, OPT = (NOUT + 255) / 256;
    __shared__ float xt[NR * CCP];
    __shared__ float gt[NPOS];
    const int tid = threadIdx.x;
    float acc[OPT];
    int ooff[OPT];
#pragma unroll
    for (int k = 0; k < OPT; ++k) {
        acc[k] = 0.f;
        const int o = tid + 256 * k;   // o = tap * CX + cx
        const int tap = o < NOUT ? o / CX : 0, cx = o % CX;
        ooff[k] = (((tap / 9) * G::RH + (tap / 3) % 3) * G::RW + tap % 3) * CCP + cx;
    }
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tw = t % a.ntw; t /= a.ntw;
        const int th = t % a.nth; t /= a.nth;
        const int td = t % a.ntd; t /= a.ntd;
        const int b = t;
        const int qd0 = td * G::TQD, qh0 = th * G::TQH, qw0 = tw * G::TQW;
        __syncthreads();
        {
            // all loads of the tile first (float4), then the LDS writes (odd voxel stride -> 4 scalar writes each)
            constexpr int XIT = (NR * (CX / 4) + 255) / 256;
            float4 xv[XIT];
            int xo[XIT];
            stage_load<XIT>(xv, xo, tid, NR * (CX / 4), [&](int i, const float*& src, int& o) {
                const int vox = i / (CX / 4), cq = i % (CX / 4);
                const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
                const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
                o = vox * CCP + 4 * cq;
                if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                    src = a.x + ((((size_t)b * a.Di + id) * a.Hi + ih) * a.Wi + iw) * CX + 4 * cq;
            });
            const int pw = tid % G::TQW, ph = (tid / G::TQW) % G::TQH, pd = tid / (G::TQW * G::TQH);
            const int qd = qd0 + pd, qh = qh0 + ph, qw = qw0 + pw;
            const float gval = (qd < a.QD && qh < a.QH && qw < a.QW) ? a.g[(((size_t)b * a.QD + qd) * a.QH + qh) * a.QW + qw] : 0.f;
#pragma unroll
            for (int k = 0; k < XIT; ++k)
                if (xo[k] >= 0) { xt[xo[k]] = xv[k].x; xt[xo[k] + 1] = xv[k].y; xt[xo[k] + 2] = xv[k].z; xt[xo[k] + 3] = xv[k].w; }
            gt[tid] = gval;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < NPOS; ++p) {
            const int pw = p % G::TQW, ph = (p / G::TQW) % G::TQH, pd = p / (G::TQW * G::TQH);
            const int xoff = ((pd * G::RH + ph) * G::RW + pw) * CCP;
            const float gv = gt[p];
#pragma unroll
            for (int k = 0; k < OPT; ++k) acc[k] = fmaf(xt[xoff + ooff[k]], gv, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < OPT; ++k) {
        const int o = tid + 256 * k;
        if (o < NOUT) a.part[(size_t)blockIdx.x * NOUT + o] = acc[k];
    }
}

// ------------------------------------------------------------------------------------------------
// CG == 1, CX == 8 (the `prob` layer, mvsnet.py:63 / network.py:65) on the 16-block 4x4x1 MFMA:
//   dW[tap][cx] = sum_q X[q][cx] * g[q - (tap - 1)]      (q over the X positions, g zero outside its volume)
// is one MFMA per position q: block (tg, cg) accumulates the outer product of 4 taps (A: the Toeplitz gather
// g[q - tap + 1] from a 6x6x18 g halo tile in LDS) and 4 channels (B: X[q][4cg + j]); 8 tap groups x 2 channel
// groups = 16 blocks, 27 of 32 taps used.  X is read exactly once and needs no halo; the VALU form above did two
// LDS reads per FMA and ran 6x over its HBM floor (profiles/r01_run17_bench_kernel_table.txt: 0.24 ms).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_wgrad_cg1_mfma_kernel(WgradArgs a) {
    constexpr int TQD = 4, TQH = 4, TQW = 16, NPOS = TQD * TQH * TQW;
    constexpr int GD = TQD + 2, GH = TQH + 2, GW = TQW + 2, NG = GD * GH * GW;   // 648
    __shared__ __attribute__((aligned(16))) float xt[NPOS * 8];
    __shared__ float gt[NG + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    // A-operand lane (block b = lane>>2, row i = lane&3): tap 4*(b>>1) + i; B-operand lane (b, j): channel 4*(b&1) + j
    const int tapA = 4 * (lane >> 3) + (lane & 3);
    const int tA = tapA < 27 ? tapA : 13;                  // rows 27..31 accumulate the centre tap again; never written
    const int goffA = -((((tA / 9) - 1) * GH + ((tA / 3) % 3 - 1)) * GW + (tA % 3 - 1));
    const int cxB = 4 * ((lane >> 2) & 1) + (lane & 3);

    constexpr int GIT = (NG + 255) / 256;
    float4 xv[2];
    float gv[GIT];
    auto load_tile = [&](int tile) {
        int b, td, th, tw;
        linear_tile(tile, a.ntw, a.nth, a.ntd, b, td, th, tw);
        const int qd0 = td * TQD, qh0 = th * TQH, qw0 = tw * TQW;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k, q = i >> 1, hq = i & 1;
            const int qw = qw0 + q % TQW, qh = qh0 + (q / TQW) % TQH, qd = qd0 + q / (TQW * TQH);
            xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qd < a.Di && qh < a.Hi && qw < a.Wi)
                xv[k] = *reinterpret_cast<const float4*>(a.x + ((((size_t)b * a.Di + qd) * a.Hi + qh) * a.Wi + qw) * 8 + 4 * hq);
        }
#pragma unroll
        for (int k = 0; k < GIT; ++k) {
            const int i = tid + 256 * k;
            const int rw = i % GW, rh = (i / GW) % GH, rd = i / (GW * GH);
            const int gd = qd0 + rd - 1, gh = qh0 + rh - 1, gw = qw0 + rw - 1;
            gv[k] = 0.f;
            if (i < NG && gd >= 0 && gd < a.QD && gh >= 0 && gh < a.QH && gw >= 0 && gw < a.QW)
                gv[k] = a.g[(((size_t)b * a.QD + gd) * a.QH + gh) * a.QW + gw];
        }
    };
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                   // previous tile's MFMAs have read the LDS images
#pragma unroll
        for (int k = 0; k < 2; ++k) *reinterpret_cast<float4*>(&xt[(tid + 256 * k) * 4]) = xv[k];
#pragma unroll
        for (int k = 0; k < GIT; ++k)
            if (tid + 256 * k < NG) gt[tid + 256 * k] = gv[k];
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);   // in flight during this tile's MFMAs
        // wave = plane of the tile; 4 rows x 16 positions
#pragma unroll
        for (int ph = 0; ph < TQH; ++ph) {
            const int grow = ((wave + 1) * GH + ph + 1) * GW + 1 + goffA;
            const int xrow = ((wave * TQH + ph) * TQW) * 8 + cxB;
#pragma unroll
            for (int pw = 0; pw < TQW; ++pw)
                acc[pw & 3] = MVS_MFMA_4x4x1(gt[grow + pw], xt[xrow + pw * 8], acc[pw & 3]);
        }
    }
    // D: lane (b, j) register r = dW[tap 4*(b>>1) + r][cx 4*(b&1) + j]; sum the 4 accumulators, then the 4 waves
    __syncthreads();
    float* red = xt;                                        // [wave][32 taps][8]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tap = 4 * (lane >> 3) + r;
        red[(wave * 32 + tap) * 8 + cxB] = acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r];
    }
    __syncthreads();
    if (tid < 27 * 8)
        a.part[(size_t)blockIdx.x * (27 * 8) + tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
}


// ------------------------------------------------------------------------------------------------
// Cout == 8 (conv0 32->8: 68 % of the regulariser's FLOPs; mvsnet.py:40): a 16-wide MFMA N tile is half
// empty, so this layer uses the 16-block form v_mfma_f32_4x4x1_16b_f32 instead: every MFMA is 16
// independent (4 positions) x (4 output channels) outer products over ONE (tap, ci), i.e. 64 positions x 4
// channels with no padding anywhere; two MFMAs (h = 0, 1) cover the 8 channels.  Measured issue rate on
// gfx950 is 81 % of the 16x16x4 form (profiles/r01_mfma_rate.log) vs 50 % useful work there.
// Lane l = position l of a 64-position set for the A operand (block l>>2, row l&3); the B operand is the
// weight of channel (l&3)+4h broadcast to all blocks; D: lane holds channel (l&3)+4h of the 4 positions
// 4*(l>>2)+r.  One ds_read_b128 per lane feeds the A operands of 4 consecutive ci (8 MFMAs per set).
// ------------------------------------------------------------------------------------------------
template <int CC, int NV>
__global__ __launch_bounds__(256) void conv_c8_fwd_kernel(ConvArgs a, const float* __restrict__ w, int wlayout, int flip) {
    constexpr int TQD = 4, TQH = 4 * NV, TQW = 16;
    constexpr int RD = TQD + 2, RH = TQH + 2, RW = TQW + 2;
    constexpr int CCP = CC + 4, CQ = CC / 4, NR = RD * RH * RW;
    __shared__ __attribute__((aligned(16))) float tile[NR * CCP];
    __shared__ __attribute__((aligned(16))) float wl[27 * CQ * 2 * 4 * 4];   // [tap][cq][h][j][kk]
    __shared__ float red[4 * 8 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tw = t % a.ntw; t /= a.ntw;
    const int th = t % a.nth; t /= a.nth;
    const int td = t % a.ntd; t /= a.ntd;
    const int b = t;
    const int qd0 = td * TQD, qh0 = th * TQH, qw0 = tw * TQW;
    int baseA[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int row = (wave * NV + v) * 4 + (lane >> 4);      // row of the TQD x TQH plane of rows
        baseA[v] = (((row / TQH) * RH + row % TQH) * RW + (lane & 15)) * CCP;
    }
    f32x4 acc[NV][2];
#pragma unroll
    for (int v = 0; v < NV; ++v) { acc[v][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[v][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int nchunks = a.Cin / CC;
    constexpr int XIT = (NR * CQ + 255) / 256;          // float4 of the input halo tile per thread
    constexpr int WIT = (27 * CQ * 32 + 255) / 256;     // weight floats per thread
    float4 xv[XIT];
    int xo[XIT];
    float wv[WIT];
    auto load_chunk = [&](int chunk) {
        stage_load<XIT>(xv, xo, tid, NR * CQ, [&](int i, const float*& src, int& o) {
            const int vox = i / CQ, cq = i % CQ;
            const int rw = vox % RW, rh = (vox / RW) % RH, rd = vox / (RW * RH);
            const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
            o = vox * CCP + 4 * cq;
            if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                src = a.x + ((((size_t)b * a.Di + id) * a.Hi + ih) * a.Wi + iw) * a.Cin + chunk * CC + 4 * cq;
        });
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + 256 * k;
            float v = 0.f;
            if (i < 27 * CQ * 32) {
                const int kk = i & 3, j = (i >> 2) & 3, h = (i >> 4) & 1, cq = (i >> 5) % CQ, tap = (i >> 5) / CQ;
                const int co = j + 4 * h, ci = chunk * CC + 4 * cq + kk;
                const int kidx = flip ? 26 - tap : tap;
                if (co < a.Cout) v = wlayout == WL_OIK ? w[((size_t)co * a.Cin + ci) * 27 + kidx] : w[((size_t)ci * a.Cout + co) * 27 + kidx];
            }
            wv[k] = v;
        }
    };
    load_chunk(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();                                   // previous chunk's MFMAs have read the LDS image
        stage_store<XIT>(tile, xv, xo);
#pragma unroll
        for (int k = 0; k < WIT; ++k) { const int i = tid + 256 * k; if (i < 27 * CQ * 32) wl[i] = wv[k]; }
        __syncthreads();
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);    // in flight while this chunk's MFMAs run
        for (int tap = 0; tap < 27; ++tap) {
            const int toff = (((tap / 9) * RH + (tap / 3) % 3) * RW + tap % 3) * CCP;
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                const float4 b0 = *reinterpret_cast<const float4*>(&wl[((tap * CQ + cq) * 2 + 0) * 16 + (lane & 3) * 4]);
                const float4 b1 = *reinterpret_cast<const float4*>(&wl[((tap * CQ + cq) * 2 + 1) * 16 + (lane & 3) * 4]);
                float4 av[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) av[v] = *reinterpret_cast<const float4*>(&tile[baseA[v] + toff + 4 * cq]);
                // (measured: this v-outer order is 8 % faster than k-outer, profiles/r01_run10_kernels.log)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    acc[v][0] = MVS_MFMA_4x4x1(av[v].x, b0.x, acc[v][0]); acc[v][1] = MVS_MFMA_4x4x1(av[v].x, b1.x, acc[v][1]);
                    acc[v][0] = MVS_MFMA_4x4x1(av[v].y, b0.y, acc[v][0]); acc[v][1] = MVS_MFMA_4x4x1(av[v].y, b1.y, acc[v][1]);
                    acc[v][0] = MVS_MFMA_4x4x1(av[v].z, b0.z, acc[v][0]); acc[v][1] = MVS_MFMA_4x4x1(av[v].z, b1.z, acc[v][1]);
                    acc[v][0] = MVS_MFMA_4x4x1(av[v].w, b0.w, acc[v][0]); acc[v][1] = MVS_MFMA_4x4x1(av[v].w, b1.w, acc[v][1]);
                }
            }
        }
    }
    // epilogue: lane (block bl = lane>>2, j = lane&3) holds channel j+4h of positions 4*bl + r of its set
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int bl = lane >> 2, j = lane & 3;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int row = (wave * NV + v) * 4 + (bl >> 2);
        const int qd = qd0 + row / TQH, qh = qh0 + row % TQH;
        if (qd >= a.QD || qh >= a.QH) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qw = qw0 + 4 * (bl & 3) + r;
            if (qw >= a.QW) continue;
            const size_t obase = ((((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qw) * a.Cout;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int co = j + 4 * h;
                if (co >= a.Cout) continue;
                float val = acc[v][h][r];
                s1[h] += val;
                s2[h] += val * val;
                if (a.scale) val = val * a.scale[co] + a.shift[co];
                else if (a.shift) val = val + a.shift[co];
                if (a.relu) val = fmaxf(val, 0.f);
                if (a.skip) val += a.skip[obase + co];
                a.y[obase + co] = val;
            }
        }
    }
    if (a.partials) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float x1 = s1[h], x2 = s2[h];
#pragma unroll
            for (int m = 4; m < 64; m <<= 1) { x1 += __shfl_xor(x1, m); x2 += __shfl_xor(x2, m); }
            if (lane < 4) { red[(wave * 8 + lane + 4 * h) * 2] = x1; red[(wave * 8 + lane + 4 * h) * 2 + 1] = x2; }
        }
        __syncthreads();
        if (tid < 16) {
            const int stat = tid >> 3, co = tid & 7;
            if (co < a.Cout) {
                float sm = 0.f;
                for (int wv = 0; wv < 4; ++wv) sm += red[(wv * 8 + co) * 2 + stat];
                a.partials[((size_t)blockIdx.x * 2 + stat) * a.Cout + co] = sm;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Cout == 8 forward, second form: the WEIGHTS are the broadcast operand and never touch LDS.
// v_mfma_f32_4x4x1_16b_f32 has a block-broadcast control on its A operand (cbsz = 4, abid = k: all 16 blocks
// use the A values of block k).  One VGPR per (tap, channel half h) therefore carries the weights of 16 input
// channels: lane (k = l>>2, i = l&3) holds w[ci = k][co = 4h + i], and `abid` selects the input channel.
// The B operand is the voxel value of the lane's own position, so D leaves each lane with 4 (x2) consecutive
// output channels of ITS position -> float4 stores.  LDS carries only the halo tile (one ds_read_b128 per 8
// MFMAs); the channel chunk is 16 floats = one 64-byte segment per voxel (the 8-channel chunks of the first
// form fetched every 128-byte line four times, 32 bytes at a time: profiles/r01_run17_pmc_summary.json).
// Tile 4 x 4 x 16 positions, 4 waves, one row group per wave.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma_4x4x1_bc(float a, float b, f32x4 c, int k) {
    switch (k) {   // folded after unrolling: abid must be an immediate
        case 0: return MVS_MFMA_4x4x1_BC(a, b, c, 0);
        case 1: return MVS_MFMA_4x4x1_BC(a, b, c, 1);
        case 2: return MVS_MFMA_4x4x1_BC(a, b, c, 2);
        case 3: return MVS_MFMA_4x4x1_BC(a, b, c, 3);
        case 4: return MVS_MFMA_4x4x1_BC(a, b, c, 4);
        case 5: return MVS_MFMA_4x4x1_BC(a, b, c, 5);
        case 6: return MVS_MFMA_4x4x1_BC(a, b, c, 6);
        case 7: return MVS_MFMA_4x4x1_BC(a, b, c, 7);
        case 8: return MVS_MFMA_4x4x1_BC(a, b, c, 8);
        case 9: return MVS_MFMA_4x4x1_BC(a, b, c, 9);
        case 10: return MVS_MFMA_4x4x1_BC(a, b, c, 10);
        case 11: return MVS_MFMA_4x4x1_BC(a, b, c, 11);
        case 12: return MVS_MFMA_4x4x1_BC(a, b, c, 12);
        case 13: return MVS_MFMA_4x4x1_BC(a, b, c, 13);
        case 14: return MVS_MFMA_4x4x1_BC(a, b, c, 14);
        default: return MVS_MFMA_4x4x1_BC(a, b, c, 15);
    }
}

template <int CC, int NCH>
__global__ __launch_bounds__(256) void conv_c8_fwd_bc_kernel(ConvArgs a, const float* __restrict__ w, int wlayout, int flip, int xcd) {
    constexpr int TQD = 4, TQH = 4, TQW = 16;
    constexpr int RD = TQD + 2, RH = TQH + 2, RW = TQW + 2;
    constexpr int CCP = CC + 4, CQ = CC / 4, NR = RD * RH * RW;
    __shared__ __attribute__((aligned(16))) float tile[NR * CCP];
    __shared__ float wl[NCH * 27 * 2 * 64];   // the broadcast-operand image: [chunk][tap][h][lane], read back as one ds_read_b32 per (tap, h)
    __shared__ float red[4 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // persistent workgroups (2 per CU): the weight image is built once, and the first chunk of the next tile is
    // fetched while the last chunk of the current one is in the MFMAs
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    const int vb = xcd ? xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    // this lane's position: plane `wave` of the tile, row lane>>4, column lane&15
    const int baseB = ((wave * RH + (lane >> 4)) * RW + (lane & 15)) * CCP;

    constexpr int XIT = (NR * CQ + 255) / 256;
    float4 xv[XIT];
    auto tile_origin = [&](int t, int& b, int& qd0, int& qh0, int& qw0) {
        int td, th, tw;
        if (xcd) brick_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        else linear_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        qd0 = td * TQD; qh0 = th * TQH; qw0 = tw * TQW;
    };
    // halo tile -> registers (all loads issued back to back; zero outside the volume), registers -> LDS.
    // The offset of item k relative to the tile's origin voxel does not depend on the tile, so it is computed once
    // per (persistent) workgroup; tiles that do not touch the volume boundary skip the per-item bounds checks.
    int rel[XIT];
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
        const int i = tid + 256 * k, vox = i / CQ, cq = i % CQ;
        const int rw = vox % RW, rh = (vox / RW) % RH, rd = vox / (RW * RH);
        rel[k] = (((rd - 1) * a.Hi + (rh - 1)) * a.Wi + (rw - 1)) * a.Cin + 4 * cq;
    }
    auto load_chunk = [&](int b, int qd0, int qh0, int qw0, int chunk) {
        const float* __restrict__ base = a.x + ((((size_t)b * a.Di + qd0) * a.Hi + qh0) * a.Wi + qw0) * a.Cin + chunk * CC;
        const bool interior = qd0 >= 1 && qd0 + TQD + 1 <= a.Di && qh0 >= 1 && qh0 + TQH + 1 <= a.Hi &&
                              qw0 >= 1 && qw0 + TQW + 1 <= a.Wi;
        if (interior) {
#pragma unroll
            for (int k = 0; k < XIT; ++k)
                if (tid + 256 * k < NR * CQ) xv[k] = *reinterpret_cast<const float4*>(base + rel[k]);
        } else {
#pragma unroll
            for (int k = 0; k < XIT; ++k) {
                const int i = tid + 256 * k;
                xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < NR * CQ) {
                    const int vox = i / CQ;
                    const int rw = vox % RW, rh = (vox / RW) % RH, rd = vox / (RW * RH);
                    const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
                    if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                        xv[k] = *reinterpret_cast<const float4*>(base + rel[k]);
                }
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < XIT; ++k) {
            const int i = tid + 256 * k;
            if (i < NR * CQ) *reinterpret_cast<float4*>(&tile[(i / CQ) * CCP + 4 * (i % CQ)]) = xv[k];
        }
    };
    int b, qd0, qh0, qw0;
    if (vb < ntiles) {
        tile_origin(vb, b, qd0, qh0, qw0);
        load_chunk(b, qd0, qh0, qw0, 0);
    }
    // wl[ch][tap][h][l] = w[ci = ch*CC + (l>>2)][co = 4h + (l&3)][tap]
    for (int i = tid; i < NCH * 27 * 2 * 64; i += 256) {
        const int l = i & 63, h = (i >> 6) & 1, tap = (i >> 7) % 27, ch = (i >> 7) / 27;
        const int k = l >> 2, ci = ch * CC + k, co = 4 * h + (l & 3);
        const int kidx = flip ? 26 - tap : tap;
        float v = 0.f;
        if (k < CC && co < a.Cout) v = wlayout == WL_OIK ? w[((size_t)co * a.Cin + ci) * 27 + kidx] : w[((size_t)ci * a.Cout + co) * 27 + kidx];
        wl[i] = v;
    }
    for (int t = vb; t < ntiles; t += gridDim.x) {
    tile_origin(t, b, qd0, qh0, qw0);
    f32x4 acc[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) { acc[p][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[p][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        __syncthreads();                                   // previous chunk's / tile's reads of the LDS image are done
        store_chunk();
        __syncthreads();
        if (ch + 1 < NCH) load_chunk(b, qd0, qh0, qw0, ch + 1);   // in flight while this chunk's MFMAs run
        else if (t + (int)gridDim.x < ntiles) {
            int b2, d2, h2, w2;
            tile_origin(t + gridDim.x, b2, d2, h2, w2);
            load_chunk(b2, d2, h2, w2, 0);
        }
        // the LDS reads run one tap ahead of the MFMAs that consume them
        float4 xq[2][CQ];
        float wq[2][2];
        auto read_tap = [&](int tap, float4 (&dst)[CQ], float (&wd)[2]) {
            const int toff = (((tap / 9) * RH + (tap / 3) % 3) * RW + tap % 3) * CCP;
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) dst[cq] = *reinterpret_cast<const float4*>(&tile[baseB + toff + 4 * cq]);
            wd[0] = wl[((ch * 27 + tap) * 2 + 0) * 64 + lane];
            wd[1] = wl[((ch * 27 + tap) * 2 + 1) * 64 + lane];
        };
        read_tap(0, xq[0], wq[0]);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 1 < 27) read_tap(tap + 1, xq[(tap + 1) & 1], wq[(tap + 1) & 1]);
            MVS_SCHED_FENCE();   // (hipcc otherwise sinks the reads to just before their first use)
            const float w0 = wq[tap & 1][0], w1 = wq[tap & 1][1];
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                const float4 x4 = xq[tap & 1][cq];
                const int p = cq & 1;
                acc[p][0] = mfma_4x4x1_bc(w0, x4.x, acc[p][0], 4 * cq + 0);
                acc[p][1] = mfma_4x4x1_bc(w1, x4.x, acc[p][1], 4 * cq + 0);
                acc[p][0] = mfma_4x4x1_bc(w0, x4.y, acc[p][0], 4 * cq + 1);
                acc[p][1] = mfma_4x4x1_bc(w1, x4.y, acc[p][1], 4 * cq + 1);
                acc[p][0] = mfma_4x4x1_bc(w0, x4.z, acc[p][0], 4 * cq + 2);
                acc[p][1] = mfma_4x4x1_bc(w1, x4.z, acc[p][1], 4 * cq + 2);
                acc[p][0] = mfma_4x4x1_bc(w0, x4.w, acc[p][0], 4 * cq + 3);
                acc[p][1] = mfma_4x4x1_bc(w1, x4.w, acc[p][1], 4 * cq + 3);
            }
        }
    }
    // epilogue: the lane owns position (qd0 + wave, qh0 + lane>>4, qw0 + lane&15) and channels 4h + r
    const int qd = qd0 + wave, qh = qh0 + (lane >> 4), qw = qw0 + (lane & 15);
    const bool inside = qd < a.QD && qh < a.QH && qw < a.QW;
    float s1[8], s2[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * h + r;
            float val = acc[0][h][r] + acc[1][h][r];
            if (!inside || co >= a.Cout) val = 0.f;
            s1[co] = val;
            s2[co] = val * val;
            if (co < a.Cout) {
                if (a.scale) val = val * a.scale[co] + a.shift[co];
                else if (a.shift) val = val + a.shift[co];
                if (a.relu) val = fmaxf(val, 0.f);
            }
            o[r] = val;
        }
        if (inside) {
            const size_t obase = ((((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qw) * a.Cout + 4 * h;
            if (a.Cout == 8) {
                float4 ov = make_float4(o[0], o[1], o[2], o[3]);
                if (a.skip) {
                    const float4 sk = *reinterpret_cast<const float4*>(a.skip + obase);
                    ov.x += sk.x; ov.y += sk.y; ov.z += sk.z; ov.w += sk.w;
                }
                *reinterpret_cast<float4*>(a.y + obase) = ov;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * h + r < a.Cout) a.y[obase + r] = o[r] + (a.skip ? a.skip[obase + r] : 0.f);
            }
        }
    }
    if (a.partials) {
        // 16 per-lane values -> wave sums by a halving butterfly: after the step with mask m a lane keeps the
        // half of its values selected by its bit m, so the exchange count is 8+4+2+1 (+2 full steps) instead of 16*6
        float v[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c] = s1[c]; v[8 + c] = s2[c]; }
        // (written out per step: with a loop over n the array index is not a compile-time constant and hipcc
        //  falls back to 16-way v_cndmask selection chains, ~900 VALU instructions per tile)
#define MVS_BFLY_STEP(N, M)                                              \
        {                                                                \
            const bool up = (lane & (M)) != 0;                           \
            _Pragma("unroll") for (int q = 0; q < (N); ++q) {            \
                const float keep = up ? v[(N) + q] : v[q];               \
                const float send = up ? v[q] : v[(N) + q];               \
                v[q] = keep + __shfl_xor(send, (M));                     \
            }                                                            \
        }
        MVS_BFLY_STEP(8, 32)
        MVS_BFLY_STEP(4, 16)
        MVS_BFLY_STEP(2, 8)
        MVS_BFLY_STEP(1, 4)
#undef MVS_BFLY_STEP
        // lane now holds value index (bit5,bit4,bit3,bit2 of lane) summed over the lanes that share those bits
        float r = v[0];
        r += __shfl_xor(r, 2);
        r += __shfl_xor(r, 1);
        if ((lane & 3) == 0) {
            const int idx = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            red[wave * 16 + idx] = r;
        }
        __syncthreads();
        if (tid < 16) {
            const int stat = tid >> 3, co = tid & 7;
            if (co < a.Cout)
                a.partials[((size_t)t * 2 + stat) * a.Cout + co] =
                    red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid];
        }
    }
    }   // tile loop
}


// Weight gradient for CG == 8 (conv0, mvsnet.py:40) with the 16-block 4x4x1 MFMA: per position ONE MFMA
// accumulates (4 taps x 16 input channels) x (4 output channels); 7 tap groups x 2 channel halves = 14 MFMAs
// per position, 96 % useful (the 16x16x4 form pads N 8 -> 16 and was 50 % useful).  Each wave takes a
// quarter of the tile's positions; persistent over tiles, dW kept in accumulators, waves summed through LDS.
// BC = true: the output gradient is the MFMA's broadcast operand instead (cbsz = 4: one VGPR pair carries g of 16
// consecutive W positions, abid selects the position), so g costs 2 LDS reads per 16 positions instead of 2 per
// position and each lane ends up owning its (tap, input channel) for 4 output channels.
template <bool BC>
__global__ __launch_bounds__(256) void conv_c8_wgrad_kernel(WgradArgs a) {
    using G = ConvGeom<GEOM_S1>;
    // LDS image of the X halo region: 16 floats per voxel, ODD row / plane strides (RHP x RWP = 7 x 19), so the
    // bank window (16 of 32 banks) of a voxel is selected by the parity of rd+rh+rw.  Each MFMA's four lane groups
    // read four different taps; pairing an even-parity tap with an odd-parity tap in each half-wave makes every
    // ds_read_b32 of the A operand conflict-free (with the natural tap order 1/3 of the reads were 2-way).
    constexpr int CC = 16, CCP = 16;
    constexpr int RHP = 7, RWP = 19;
    constexpr int NRP = G::RD * RHP * RWP;
    constexpr int NPOS = G::TQD * G::TQH * G::TQW;   // 256
    constexpr int NTG = 7;                           // tap groups of 4 (27 -> 28)
    __shared__ __attribute__((aligned(16))) float xt[NRP * CCP];
    __shared__ __attribute__((aligned(16))) float gt[NPOS * 8];
    __shared__ int tapmap[32];   // slot (tg, lane group) -> tap id (27 = dummy)
    __shared__ int tapoff[32];   // slot -> LDS offset of that tap
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.y;
    if (tid == 0) {
        int ne = 0, no = 0;
        int ev[16], od[16];
        for (int t = 0; t < 27; ++t) {
            if (((t / 9) + (t / 3) % 3 + t % 3) & 1) od[no++] = t; else ev[ne++] = t;   // 14 even, 13 odd
        }
        od[no++] = 27;   // dummy partner of the 14th even tap
        for (int tg = 0; tg < NTG; ++tg)
            for (int grp = 0; grp < 4; ++grp) {
                const int t = (grp & 1) ? od[2 * tg + (grp >> 1)] : ev[2 * tg + (grp >> 1)];
                const int tt = t < 27 ? t : od[0];   // dummy reads a valid odd-parity location, result discarded
                tapmap[4 * tg + grp] = t;
                tapoff[4 * tg + grp] = (((tt / 9) * RHP + (tt / 3) % 3) * RWP + tt % 3) * CCP;
            }
    }
    __syncthreads();
    int toff[NTG];
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg) toff[tg] = tapoff[4 * tg + (lane >> 4)] + (lane & 15);
    f32x4 acc[NTG][2];
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg) { acc[tg][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[tg][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    constexpr int NXI = G::RD * G::RH * G::RW * (CC / 4);
    constexpr int XIT = (NXI + 255) / 256, GIT = (NPOS * 2 + 255) / 256;
    float4 xv[XIT], gv[GIT];
    // per-item offsets relative to the tile's origin voxel / position and their LDS slots: tile-invariant, computed
    // once per persistent workgroup; tiles that do not touch the volume boundary skip the per-item bounds checks
    int xrel[XIT], xlo[XIT], grel[GIT], glo[GIT];
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
        const int i = tid + 256 * k, vox = i / (CC / 4), cq = i % (CC / 4);
        const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
        xrel[k] = (((rd - 1) * a.Hi + (rh - 1)) * a.Wi + (rw - 1)) * a.CX + 4 * cq;
        xlo[k] = ((rd * RHP + rh) * RWP + rw) * CCP + 4 * cq;
    }
#pragma unroll
    for (int k = 0; k < GIT; ++k) {
        const int i = tid + 256 * k, p = i >> 1, hq = i & 1;
        const int pw = p % G::TQW, ph = (p / G::TQW) % G::TQH, pd = p / (G::TQW * G::TQH);
        grel[k] = ((pd * a.QH + ph) * a.QW + pw) * 8 + 4 * hq;
        glo[k] = p * 8 + 4 * hq;
    }
    auto load_tile = [&](int tile) {
        int b, td, th, tw;
        if (a.xcd) brick_tile(tile, a.ntw, a.nth, a.ntd, b, td, th, tw);
        else linear_tile(tile, a.ntw, a.nth, a.ntd, b, td, th, tw);
        const int qd0 = td * G::TQD, qh0 = th * G::TQH, qw0 = tw * G::TQW;
        const float* __restrict__ xb = a.x + ((((size_t)b * a.Di + qd0) * a.Hi + qh0) * a.Wi + qw0) * a.CX + chunk * CC;
        const float* __restrict__ gb = a.g + ((((size_t)b * a.QD + qd0) * a.QH + qh0) * a.QW + qw0) * 8;
        const bool interior = qd0 >= 1 && qd0 + G::TQD + 1 <= a.Di && qh0 >= 1 && qh0 + G::TQH + 1 <= a.Hi &&
                              qw0 >= 1 && qw0 + G::TQW + 1 <= a.Wi;
        if (interior) {
#pragma unroll
            for (int k = 0; k < XIT; ++k)
                if (tid + 256 * k < NXI) xv[k] = *reinterpret_cast<const float4*>(xb + xrel[k]);
#pragma unroll
            for (int k = 0; k < GIT; ++k)
                if (tid + 256 * k < NPOS * 2) gv[k] = *reinterpret_cast<const float4*>(gb + grel[k]);
        } else {
#pragma unroll
            for (int k = 0; k < XIT; ++k) {
                const int i = tid + 256 * k, vox = i / (CC / 4);
                const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
                const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
                xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < NXI && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                    xv[k] = *reinterpret_cast<const float4*>(xb + xrel[k]);
            }
#pragma unroll
            for (int k = 0; k < GIT; ++k) {
                const int i = tid + 256 * k, p = i >> 1;
                const int qw = qw0 + p % G::TQW, qh = qh0 + (p / G::TQW) % G::TQH, qd = qd0 + p / (G::TQW * G::TQH);
                gv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < NPOS * 2 && qd < a.QD && qh < a.QH && qw < a.QW)
                    gv[k] = *reinterpret_cast<const float4*>(gb + grel[k]);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < XIT; ++k)
            if (tid + 256 * k < NXI) *reinterpret_cast<float4*>(xt + xlo[k]) = xv[k];
#pragma unroll
        for (int k = 0; k < GIT; ++k)
            if (tid + 256 * k < NPOS * 2) *reinterpret_cast<float4*>(gt + glo[k]) = gv[k];
    };
    const int vb = a.xcd ? xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // co-resident workgroups of an XCD take neighbouring tiles
    if (vb < ntiles) load_tile(vb);
    for (int tile = vb; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                     // previous tile's MFMAs have read the LDS images
        store_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);   // in flight during this tile's MFMAs
        if (BC) {
            static_assert(G::TQW == 16 && G::TQH == 4, "one group of 16 positions = one W row of the tile");
#pragma unroll 1
            for (int g16 = 0; g16 < 4; ++g16) {                 // wave = plane of the tile, g16 = row
                const int p0 = wave * 64 + g16 * 16;
                const int xoff0 = ((wave * RHP + g16) * RWP) * CCP;
                const float ga0 = gt[(p0 + (lane >> 2)) * 8 + (lane & 3)], ga1 = gt[(p0 + (lane >> 2)) * 8 + 4 + (lane & 3)];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float av[NTG];
#pragma unroll
                    for (int tg = 0; tg < NTG; ++tg) av[tg] = xt[xoff0 + k * CCP + toff[tg]];
#pragma unroll
                    for (int tg = 0; tg < NTG; ++tg) {
                        acc[tg][0] = mfma_4x4x1_bc(ga0, av[tg], acc[tg][0], k);
                        acc[tg][1] = mfma_4x4x1_bc(ga1, av[tg], acc[tg][1], k);
                    }
                }
            }
        } else {
#pragma unroll 2
        for (int k = 0; k < NPOS / 4; ++k) {
            const int p = wave * (NPOS / 4) + k;
            const int pw_ = p % G::TQW, ph_ = (p / G::TQW) % G::TQH, pd_ = p / (G::TQW * G::TQH);
            const int xoff = ((pd_ * RHP + ph_) * RWP + pw_) * CCP;
            const float b0 = gt[p * 8 + (lane & 3)], b1 = gt[p * 8 + 4 + (lane & 3)];
            float av[NTG];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) av[tg] = xt[xoff + toff[tg]];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                acc[tg][0] = MVS_MFMA_4x4x1(av[tg], b0, acc[tg][0]);
                acc[tg][1] = MVS_MFMA_4x4x1(av[tg], b1, acc[tg][1]);
            }
        }
        }
    }
    // sum the 4 waves through LDS (reusing xt: 28 slots x 16 cx x 8 co = 3584 floats), then one partial image
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // lane: block bl = lane>>2 -> slot 4*tg + (bl>>2), cx group bl&3; j = lane&3 -> co = j + 4h; reg r -> cx = 4*(bl&3) + r
                        // (BC: the lane is (slot, cx) = (lane>>4, lane&15) and reg r -> co = 4h + r)
                        const int bl = lane >> 2;
                        const int idx = BC ? ((4 * tg + (lane >> 4)) * 16 + (lane & 15)) * 8 + 4 * h + r
                                           : ((4 * tg + (bl >> 2)) * 16 + 4 * (bl & 3) + r) * 8 + (lane & 3) + 4 * h;
                        if (wv == 0) xt[idx] = acc[tg][h][r];
                        else xt[idx] += acc[tg][h][r];
                    }
        }
        __syncthreads();
    }
    for (int i = tid; i < 28 * 16 * 8; i += 256) {
        const int co = i & 7, cx = (i >> 3) & 15, tap = tapmap[i >> 7];
        if (tap < 27) a.part[(((size_t)blockIdx.x * 27 + tap) * a.CX + chunk * 16 + cx) * 8 + co] = xt[i];
    }
}

// first level of a two-level reduction: [nparts][n] -> [nout][n], block y sums parts y, y+nout, ...
__global__ __launch_bounds__(256) void conv_wgrad_prereduce_kernel(const float* __restrict__ part, int nparts, int n,
                                                                   int nout, float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s0 = 0.f, s1 = 0.f;
    int p = blockIdx.y;
    for (; p + nout < nparts; p += 2 * nout) {
        s0 += part[(size_t)p * n + e];
        s1 += part[(size_t)(p + nout) * n + e];
    }
    if (p < nparts) s0 += part[(size_t)p * n + e];
    out[(size_t)blockIdx.y * n + e] = s0 + s1;
}

// ================================================================================================
// host side
// ================================================================================================
static int pick_cc(int geom, int cin) {
    if (geom == GEOM_TR2) return cin;
    if (geom == GEOM_S2) return 8;
    return (cin % 16 == 0) ? 16 : 8;
}

static size_t packed_floats(int geom, int cin, int cout) {
    const int cc = pick_cc(geom, cin);
    int nb = mvs_cdiv(cout, 16);
    if (nb == 3) nb = 4;
    return (size_t)total_ksteps(geom, cin, cc) * nb * 256;
}

int g_conv_split = 1;
int g_conv_tr2pw = 1;       // tuning knob "tr2pw": transposed stride-2 conv with Cout == 8 as W-parity-merged GEMMs (GEOM_TR2_PW)
int g_conv_small_wgs = 384;   // tuning knob "conv_small_wgs": quarter-size tiles below this many workgroups (~1.5 per CU)
int g_conv_small = 1;   // tuning knob "conv_small": quarter-size workgroup tiles for under-filled launches (0 never, 1 auto, 2 always)
int g_conv_c8 = 7;      // tuning knob "k8", bit mask: 1 = Cout==8 stride-1 layers use the 4x4x1 MFMA kernels, +2 = forward with the weights as the broadcast operand, +4 = weight gradient with g as the broadcast operand
int g_conv_persist = 0;  // tuning knob "conv_persist": 0 = never (default: measured SLOWER than five small workgroups per CU, profiles/r03_run8_*), 1 = stride-1 layers with many tiles run the persistent implicit-GEMM kernel (next tile's halo in flight during the MFMAs), n > 1 = with exactly n workgroups (tests)
int g_conv_fs = 0;      // tuning knob "fs": fast halo staging of interior tiles in the generic implicit-GEMM kernels (unmeasured)
int g_conv_xcd = 1;     // tuning knob "xcd": XCD-aware tile order in the broadcast-operand forward   // tuning knob "conv_split" (mvs_set_tuning): 0 keeps all Cout tiles in one workgroup

template <int GEOM, int CC>
static int launch_igemm_nb(const ConvArgs& a, int NB, int nblocks, hipStream_t st) {
    dim3 grid(nblocks, a.nb_total / NB), block(256);
    if (g_conv_fs) {
        switch (NB) {
            case 1: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 1, true>), grid, block, 0, st, a); break;
            case 2: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 2, true>), grid, block, 0, st, a); break;
            case 4: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 4, true>), grid, block, 0, st, a); break;
            default: mvs_set_error("conv igemm: Cout tile count %d unsupported", NB); return MVS_ERR_UNSUPPORTED;
        }
        return mvs_check_launch("conv_igemm_fs");
    }
    switch (NB) {
        case 1: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 1>), grid, block, 0, st, a); break;
        case 2: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 2>), grid, block, 0, st, a); break;
        case 4: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 4>), grid, block, 0, st, a); break;
        default: mvs_set_error("conv igemm: Cout tile count %d unsupported", NB); return MVS_ERR_UNSUPPORTED;
    }
    return mvs_check_launch("conv_igemm");
}

// Which tiling the generic kernel runs with (shared by run_igemm and mvs_conv3d_stat_rows: the BatchNorm partial-sum buffer has
// exactly one row per workgroup tile).  Small volumes (deep U-Net levels) have too few tiles to fill 256 CUs: first one 16-wide
// Cout tile per workgroup, and if that still leaves fewer than ~1.5 workgroups per CU, quarter-size tiles (knob "conv_small":
// 0 never, 1 auto, 2 always).
static void igemm_tiling(int geom, int B, int QD, int QH, int QW, int cout, int& kgeom, int& NB, int& nblocks) {
    const int nb_total = mvs_cdiv(cout, 16) == 3 ? 4 : mvs_cdiv(cout, 16);
    nblocks = B * mvs_cdiv(QD, geom_tqd(geom)) * mvs_cdiv(QH, geom_tqh(geom)) * mvs_cdiv(QW, 16);
    NB = nb_total;
    if (g_conv_split && nblocks < 512 && NB > 1) NB = 1;
    kgeom = geom;
    // measured per geometry (profiles/r02_run14_*): stride-2 layers gain ~10 % from quarter tiles at every size of the network
    // (5x9x33-voxel halo in LDS -> 3x9x33: one more workgroup per CU), stride-1 layers up to ~1000 workgroups, the transposed
    // geometry only when the chip is under-filled (its 8 parity classes re-walk the tile: smaller tiles lose 10 % at L0)
    const long thr = (long)g_conv_small_wgs * (geom == GEOM_S2 ? 24 : (geom == GEOM_S1 ? 3 : 1));
    if (g_conv_small == 2 || (g_conv_small == 1 && (long)nblocks * (nb_total / NB) < thr)) {
        kgeom = geom + GEOM_S1_SMALL;
        nblocks = B * mvs_cdiv(QD, geom_tqd(kgeom)) * mvs_cdiv(QH, geom_tqh(kgeom)) * mvs_cdiv(QW, 16);
    }
}

struct Epilogue {
    const float* scale; const float* shift; const float* skip; int relu; float* partials;
};

// in: [B,Di,Hi,Wi,cin] ; out: [B,Do,Ho,Wo,cout] ; coarse grid: S1/S2 -> output dims, TR2 -> input dims
static int run_igemm(int geom, const float* in, const float* wsrc, int wlayout, int flip, float* out, float* ws,
                     int B, int Di, int Hi, int Wi, int cin, int cout, const Epilogue& ep, hipStream_t st) {
    MVS_REQUIRE(in && wsrc && out && ws, MVS_ERR_NULL, "conv: null pointer argument");
    MVS_REQUIRE(cin == 8 || cin == 16 || cin == 32 || cin == 64, MVS_ERR_UNSUPPORTED,
                "conv igemm: input channels must be 8/16/32/64, got %d", cin);
    MVS_REQUIRE(cout >= 1 && cout <= 64, MVS_ERR_UNSUPPORTED, "conv igemm: output channels must be <= 64, got %d", cout);
    MVS_REQUIRE(!(geom == GEOM_TR2 && cin < 16), MVS_ERR_UNSUPPORTED, "transposed stride-2 conv needs >= 16 input channels");
    ConvArgs a = {};
    a.x = in; a.y = out; a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = cin; a.Cout = cout;
    a.scale = ep.scale; a.shift = ep.shift; a.skip = ep.skip; a.relu = ep.relu; a.partials = ep.partials;
    if (geom == GEOM_S1) { a.Do = Di; a.Ho = Hi; a.Wo = Wi; a.QD = Di; a.QH = Hi; a.QW = Wi; }
    else if (geom == GEOM_S2) {
        a.Do = (Di - 1) / 2 + 1; a.Ho = (Hi - 1) / 2 + 1; a.Wo = (Wi - 1) / 2 + 1;
        a.QD = a.Do; a.QH = a.Ho; a.QW = a.Wo;
    } else { a.Do = 2 * Di; a.Ho = 2 * Hi; a.Wo = 2 * Wi; a.QD = Di; a.QH = Hi; a.QW = Wi; }
    a.ntd = mvs_cdiv(a.QD, geom_tqd(geom)); a.nth = mvs_cdiv(a.QH, geom_tqh(geom)); a.ntw = mvs_cdiv(a.QW, 16);
    if ((g_conv_c8 & 2) && geom == GEOM_S1 && cout == 8 && (cin == 8 || cin == 16 || cin == 32)) {
        // 4x4x1 MFMA with the weights as the broadcast operand, tile 4 x 4 x 16 positions
        const int ntl = B * a.ntd * a.nth * a.ntw;
        const int nbc = ntl < 512 ? ntl : 512;            // 80 KB of LDS -> 2 resident workgroups per CU
        if (cin == 32) MVS_LAUNCH((conv_c8_fwd_bc_kernel<16, 2>), dim3(nbc), dim3(256), 0, st, a, wsrc, wlayout, flip, g_conv_xcd);
        else if (cin == 16) MVS_LAUNCH((conv_c8_fwd_bc_kernel<16, 1>), dim3(nbc), dim3(256), 0, st, a, wsrc, wlayout, flip, g_conv_xcd);
        else MVS_LAUNCH((conv_c8_fwd_bc_kernel<8, 1>), dim3(nbc), dim3(256), 0, st, a, wsrc, wlayout, flip, g_conv_xcd);
        return mvs_check_launch("conv_c8_fwd_bc");
    }
    if (g_conv_c8 && geom == GEOM_S1 && cout == 8 && cin % 8 == 0) {
        // 4x4x1 MFMA path, tile 4 x 8 x 16 positions
        a.nth = mvs_cdiv(a.QH, 8);
        const int nb8 = B * a.ntd * a.nth * a.ntw;
        MVS_LAUNCH((conv_c8_fwd_kernel<8, 2>), dim3(nb8), dim3(256), 0, st, a, wsrc, wlayout, flip);
        return mvs_check_launch("conv_c8_fwd");
    }
    int nblocks = B * a.ntd * a.nth * a.ntw;
    if (geom == GEOM_S1 && cout == 1 && wlayout == WL_OIK && !flip && !ep.partials && (cin == 8 || cin == 16)) {
        if (g_conv_cout1_d4 & 1) {   // four outputs per thread, tile 4 x 8 x 16
            a.nth = mvs_cdiv(a.QH, 8);
            const int nb4 = B * a.ntd * a.nth * a.ntw;
            if (cin == 8) MVS_LAUNCH((conv_cout1_d4_kernel<8>), dim3(nb4), dim3(128), 0, st, a, wsrc);
            else MVS_LAUNCH((conv_cout1_d4_kernel<16>), dim3(nb4), dim3(128), 0, st, a, wsrc);
            return mvs_check_launch("conv_cout1_d4");
        }
        if (cin == 8) MVS_LAUNCH((conv_cout1_kernel<8>), dim3(nblocks), dim3(256), 0, st, a, wsrc);
        else MVS_LAUNCH((conv_cout1_kernel<16>), dim3(nblocks), dim3(256), 0, st, a, wsrc);
        return mvs_check_launch("conv_cout1");
    }
    const int cc = pick_cc(geom, cin);
    int NB, kgeom;
    a.nb_total = mvs_cdiv(cout, 16) == 3 ? 4 : mvs_cdiv(cout, 16);
    igemm_tiling(geom, B, a.QD, a.QH, a.QW, cout, kgeom, NB, nblocks);
    a.ntd = mvs_cdiv(a.QD, geom_tqd(kgeom)); a.nth = mvs_cdiv(a.QH, geom_tqh(kgeom));
    // pack weights into ws
    {
        const int pgeom = (kgeom == GEOM_TR2 && cc == 16 && cout == 8 && g_conv_tr2pw) ? GEOM_TR2_PW : geom;
        const int total = (int)((size_t)total_ksteps(pgeom, cin, cc) * a.nb_total * 256);
        MVS_LAUNCH(conv_pack_weights_kernel, dim3(mvs_cdiv(total, 256)), dim3(256), 0, st, wsrc, ws, pgeom, cc, cin, cout,
                   a.nb_total, wlayout, flip, total);
    }
    a.wp = ws;
    if (kgeom == GEOM_S1 && g_conv_persist && NB == a.nb_total && !g_conv_fs) {
        // persistent workgroups when every slot gets >= 2 tiles: 3 workgroups per CU (52 KB of LDS with 16-channel chunks; the
        // register allocation is held to 3 waves per SIMD), 2 for the 64-wide layers (register budget)
        const int slots = g_conv_persist > 1 ? g_conv_persist : 256 * (NB == 4 ? 2 : 3);   // > 1: that many workgroups (tests)
        if (nblocks >= 2 * slots && (NB == 1 || NB == 2 || NB == 4)) {
            dim3 pgrid(slots), block(256);
#define MVS_S1P(CCV, NBV) MVS_LAUNCH((conv_igemm_s1p_kernel<CCV, NBV>), pgrid, block, 0, st, a, g_conv_xcd)
            if (cc == 16) { if (NB == 1) MVS_S1P(16, 1); else if (NB == 2) MVS_S1P(16, 2); else MVS_S1P(16, 4); }
            else { if (NB == 1) MVS_S1P(8, 1); else if (NB == 2) MVS_S1P(8, 2); else MVS_S1P(8, 4); }
#undef MVS_S1P
            return mvs_check_launch("conv_igemm_s1p");
        }
    }
    if (kgeom == GEOM_S1) return cc == 16 ? launch_igemm_nb<GEOM_S1, 16>(a, NB, nblocks, st)
                                          : launch_igemm_nb<GEOM_S1, 8>(a, NB, nblocks, st);
    if (kgeom == GEOM_S2) return launch_igemm_nb<GEOM_S2, 8>(a, NB, nblocks, st);
    if (kgeom == GEOM_S1_SMALL) return cc == 16 ? launch_igemm_nb<GEOM_S1_SMALL, 16>(a, NB, nblocks, st)
                                                : launch_igemm_nb<GEOM_S1_SMALL, 8>(a, NB, nblocks, st);
    if (kgeom == GEOM_S2_SMALL) return launch_igemm_nb<GEOM_S2_SMALL, 8>(a, NB, nblocks, st);
    if (kgeom == GEOM_TR2_SMALL) {
        if (cc == 16) return launch_igemm_nb<GEOM_TR2_SMALL, 16>(a, NB, nblocks, st);
        if (cc == 32) return launch_igemm_nb<GEOM_TR2_SMALL, 32>(a, NB, nblocks, st);
        return launch_igemm_nb<GEOM_TR2_SMALL, 64>(a, NB, nblocks, st);
    }
    if (cc == 16 && cout == 8 && g_conv_tr2pw) return launch_igemm_nb<GEOM_TR2_PW, 16>(a, NB, nblocks, st);
    if (cc == 16) return launch_igemm_nb<GEOM_TR2, 16>(a, NB, nblocks, st);
    if (cc == 32) return launch_igemm_nb<GEOM_TR2, 32>(a, NB, nblocks, st);
    return launch_igemm_nb<GEOM_TR2, 64>(a, NB, nblocks, st);
}

// rows of the BatchNorm partial-sum buffer of a forward call: one per workgroup tile of the tiling the call will use
static int igemm_blocks(int geom, int B, int Di, int Hi, int Wi, int cout = 16, bool generic = false) {
    int QD = Di, QH = Hi, QW = Wi;
    if (geom == GEOM_S2) { QD = (Di - 1) / 2 + 1; QH = (Hi - 1) / 2 + 1; QW = (Wi - 1) / 2 + 1; }
    if (generic) {
        int kgeom, NB, nblocks;
        igemm_tiling(geom, B, QD, QH, QW, cout, kgeom, NB, nblocks);
        return nblocks;
    }
    return B * mvs_cdiv(QD, geom_tqd(geom)) * mvs_cdiv(QH, geom_tqh(geom)) * mvs_cdiv(QW, 16);
}

static const int WGRAD_MAX_GROUPS = 768;
static int wgrad_finish(float* ws, int nparts, int CX, int CG, float* gw, hipStream_t st);   // persistent workgroups per (ci chunk, co chunk): ~3 per CU

template <int GEOM, int CC>
static void launch_wgrad(const WgradArgs& a, int nbw, dim3 grid, hipStream_t st) {
    if (nbw == 1) MVS_LAUNCH((conv_wgrad_kernel<GEOM, CC, 1>), grid, dim3(256), 0, st, a);
    else MVS_LAUNCH((conv_wgrad_kernel<GEOM, CC, 2>), grid, dim3(256), 0, st, a);
}

// X: [B,Di,Hi,Wi,CX] (the tensor indexed at o*S + tap - 1), G: [B,QD,QH,QW,CG]; out OIK [CG][CX][27]
static int run_wgrad(int geom, const float* X, const float* Gt, float* gw, float* ws, int B, int Di, int Hi, int Wi,
                     int CX, int CG, hipStream_t st) {
    MVS_REQUIRE(X && Gt && gw && ws, MVS_ERR_NULL, "conv wgrad: null pointer argument");
    MVS_REQUIRE(CX == 8 || CX == 16 || CX == 32 || CX == 64, MVS_ERR_UNSUPPORTED,
                "conv wgrad: X channels must be 8/16/32/64, got %d", CX);
    MVS_REQUIRE(CG >= 1 && CG <= 64, MVS_ERR_UNSUPPORTED, "conv wgrad: G channels must be <= 64, got %d", CG);
    WgradArgs a = {};
    a.x = X; a.g = Gt; a.part = ws; a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.CX = CX; a.CG = CG; a.xcd = g_conv_xcd;
    if (geom == GEOM_S1) { a.QD = Di; a.QH = Hi; a.QW = Wi; }
    else { a.QD = (Di - 1) / 2 + 1; a.QH = (Hi - 1) / 2 + 1; a.QW = (Wi - 1) / 2 + 1; }
    a.ntd = mvs_cdiv(a.QD, geom == GEOM_S2 ? 2 : 4); a.nth = mvs_cdiv(a.QH, 4); a.ntw = mvs_cdiv(a.QW, 16);
    const int ntiles = B * a.ntd * a.nth * a.ntw;
    const int cc = CX % 16 == 0 ? 16 : 8;
    const int nbw = CG > 16 ? 2 : 1;
    const int groups = ntiles < WGRAD_MAX_GROUPS ? ntiles : WGRAD_MAX_GROUPS;
    if (g_conv_c8 && geom == GEOM_S1 && CG == 8 && CX % 16 == 0) {
        const int g8 = ntiles < 512 ? ntiles : 512;   // 60 KB LDS -> 2 resident workgroups per CU
        if (g_conv_c8 & 4) MVS_LAUNCH(conv_c8_wgrad_kernel<true>, dim3(g8, CX / 16), dim3(256), 0, st, a);
        else MVS_LAUNCH(conv_c8_wgrad_kernel<false>, dim3(g8, CX / 16), dim3(256), 0, st, a);
        int rc8 = mvs_check_launch("conv_c8_wgrad");
        if (rc8) return rc8;
        return wgrad_finish(ws, g8, CX, CG, gw, st);
    }
    if (geom == GEOM_S1 && CG == 1 && (CX == 8 || CX == 16)) {
        if (CX == 8 && g_conv_c8) MVS_LAUNCH(conv_wgrad_cg1_mfma_kernel, dim3(groups), dim3(256), 0, st, a);
        else if (CX == 8) MVS_LAUNCH((conv_wgrad_cg1_kernel<8>), dim3(groups), dim3(256), 0, st, a);
        else MVS_LAUNCH((conv_wgrad_cg1_kernel<16>), dim3(groups), dim3(256), 0, st, a);
        int rc1 = mvs_check_launch("conv_wgrad_cg1");
        if (rc1) return rc1;
        return wgrad_finish(ws, groups, CX, CG, gw, st);
    }
    dim3 grid(groups, CX / cc, mvs_cdiv(CG, nbw * 16));
    if (geom == GEOM_S1) { if (cc == 16) launch_wgrad<GEOM_S1, 16>(a, nbw, grid, st); else launch_wgrad<GEOM_S1, 8>(a, nbw, grid, st); }
    else { if (cc == 16) launch_wgrad<GEOM_S2, 16>(a, nbw, grid, st); else launch_wgrad<GEOM_S2, 8>(a, nbw, grid, st); }
    int rc = mvs_check_launch("conv_wgrad");
    if (rc) return rc;
    return wgrad_finish(ws, groups, CX, CG, gw, st);
}

static size_t wgrad_ws_floats(int CX, int CG) { return (size_t)(WGRAD_MAX_GROUPS + 16) * 27 * CX * CG; }

// deterministic reduction of the per-workgroup partial images: > 32 images go through 16 intermediate rows first
// (a single pass has only 27*CX*CG threads, each walking all images serially -> latency bound)
static int wgrad_finish(float* ws, int nparts, int CX, int CG, float* gw, hipStream_t st) {
    const int n = 27 * CX * CG;
    const float* src = ws;
    if (nparts > 32) {
        float* mid = ws + (size_t)WGRAD_MAX_GROUPS * n;
        MVS_LAUNCH(conv_wgrad_prereduce_kernel, dim3(mvs_cdiv(n, 256), 16), dim3(256), 0, st, (const float*)ws, nparts, n, 16, mid);
        src = mid;
        nparts = 16;
    }
    MVS_LAUNCH(conv_wgrad_reduce_kernel, dim3(mvs_cdiv(n, 256)), dim3(256), 0, st, src, nparts, CX, CG, gw);
    return mvs_check_launch("conv_wgrad_reduce");
}

static int check_stride(int stride, int D, int H, int W, const char* what) {
    MVS_REQUIRE(stride == 1 || stride == 2, MVS_ERR_UNSUPPORTED, "%s: stride must be 1 or 2, got %d", what, stride);
    MVS_REQUIRE(D > 0 && H > 0 && W > 0, MVS_ERR_SHAPE, "%s: bad spatial shape %dx%dx%d", what, D, H, W);
    return MVS_OK;
}

// ---- C ABI ---------------------------------------------------------------------------------------
// (D,H,W) are always the spatial dims of the forward op's INPUT x.
enum { MVS_OP_CONV_FWD = 0, MVS_OP_CONV_DGRAD = 1, MVS_OP_CONV_WGRAD = 2,
       MVS_OP_CONVT_FWD = 3, MVS_OP_CONVT_DGRAD = 4, MVS_OP_CONVT_WGRAD = 5 };

extern "C" long long mvs_conv3d_workspace_bytes(int op, int B, int D, int H, int W, int Cin, int Cout, int stride) {
    (void)B; (void)D; (void)H; (void)W;
    size_t fl = 0;
    switch (op) {
        case MVS_OP_CONV_FWD: fl = packed_floats(stride == 2 ? GEOM_S2 : GEOM_S1, Cin, Cout); break;
        case MVS_OP_CONV_DGRAD: fl = Cout == 1 ? (size_t)27 * Cin : packed_floats(stride == 2 ? GEOM_TR2 : GEOM_S1, Cout, Cin); break;
        case MVS_OP_CONVT_FWD: fl = packed_floats(stride == 2 ? GEOM_TR2 : GEOM_S1, Cin, Cout); break;
        case MVS_OP_CONVT_DGRAD: fl = packed_floats(stride == 2 ? GEOM_S2 : GEOM_S1, Cout, Cin); break;
        case MVS_OP_CONV_WGRAD: fl = wgrad_ws_floats(Cin, Cout); break;
        case MVS_OP_CONVT_WGRAD: fl = wgrad_ws_floats(Cout, Cin); break;
        default: return -1;
    }
    return (long long)(fl * sizeof(float) + 256);
}

// rows of the [rows][2][Cout] BatchNorm partial-sum buffer a forward call writes
extern "C" int mvs_conv3d_stat_rows(int op, int B, int D, int H, int W, int Cin, int Cout, int stride) {
    if ((op == MVS_OP_CONV_FWD || op == MVS_OP_CONVT_FWD) && stride == 1 && (g_conv_c8 & 2) && Cout == 8 &&
        (Cin == 8 || Cin == 16 || Cin == 32))
        return igemm_blocks(GEOM_S1, B, D, H, W);                        // conv_c8_fwd_bc_kernel tiling
    if ((op == MVS_OP_CONV_FWD || op == MVS_OP_CONVT_FWD) && stride == 1 && g_conv_c8 && Cout == 8 && Cin % 8 == 0)
        return B * mvs_cdiv(D, 4) * mvs_cdiv(H, 8) * mvs_cdiv(W, 16);   // conv_c8_fwd_kernel tiling
    if (op == MVS_OP_CONV_FWD) return igemm_blocks(stride == 2 ? GEOM_S2 : GEOM_S1, B, D, H, W, Cout, true);
    if (op == MVS_OP_CONVT_FWD) return igemm_blocks(stride == 2 ? GEOM_TR2 : GEOM_S1, B, D, H, W, Cout, true);
    return -1;
}

// y = conv3d(x, w[Cout][Cin][3][3][3], stride, pad 1) then optional epilogue:
//   scale&&shift: y*scale[c]+shift[c] ; only shift: y+shift[c] (bias) ; relu ; + skip ; stat partials of raw y
extern "C" int mvs_conv3d_fwd(const float* x, const float* w, float* y, float* ws, int B, int D, int H, int W,
                              int Cin, int Cout, int stride, const float* scale, const float* shift,
                              const float* skip, int relu, float* stat_partials, hipStream_t stream) {
    int rc = check_stride(stride, D, H, W, "conv3d_fwd");
    if (rc) return rc;
    Epilogue ep = {scale, shift, skip, relu, stat_partials};
    return run_igemm(stride == 2 ? GEOM_S2 : GEOM_S1, x, w, WL_OIK, 0, y, ws, B, D, H, W, Cin, Cout, ep, stream);
}

// gx[B,D,H,W,Cin] = d conv3d / dx applied to gy[B,Do,Ho,Wo,Cout]
extern "C" int mvs_conv3d_dgrad(const float* gy, const float* w, float* gx, float* ws, int B, int D, int H, int W,
                                int Cin, int Cout, int stride, hipStream_t stream) {
    int rc = check_stride(stride, D, H, W, "conv3d_dgrad");
    if (rc) return rc;
    Epilogue ep = {nullptr, nullptr, nullptr, 0, nullptr};
    if (stride == 1) {
        if (Cout == 1) {
            MVS_REQUIRE(gy && w && gx && ws, MVS_ERR_NULL, "conv3d_dgrad: null pointer argument");
            MVS_REQUIRE(Cin == 8 || Cin == 16, MVS_ERR_UNSUPPORTED, "conv3d_dgrad(Cout=1): Cin must be 8 or 16, got %d", Cin);
            MVS_LAUNCH(conv_cin1_pack_kernel, dim3(mvs_cdiv(27 * Cin, 256)), dim3(256), 0, stream, w, ws, Cin);
            if (g_conv_cout1_d4 & 1) {
                const size_t total4 = (size_t)B * ((D + 3) / 4) * H * W;
                dim3 grid4((unsigned)((total4 + 255) / 256));
                if (Cin == 8) MVS_LAUNCH((conv_cin1_d4_kernel<8>), grid4, dim3(256), 0, stream, gy, (const float*)ws, gx, B, D, H, W);
                else MVS_LAUNCH((conv_cin1_d4_kernel<16>), grid4, dim3(256), 0, stream, gy, (const float*)ws, gx, B, D, H, W);
                return mvs_check_launch("conv_cin1_d4");
            }
            const size_t total = (size_t)B * D * H * W;
            dim3 grid((unsigned)((total + 255) / 256));
            if (Cin == 8) MVS_LAUNCH((conv_cin1_kernel<8>), grid, dim3(256), 0, stream, gy, (const float*)ws, gx, B, D, H, W);
            else MVS_LAUNCH((conv_cin1_kernel<16>), grid, dim3(256), 0, stream, gy, (const float*)ws, gx, B, D, H, W);
            return mvs_check_launch("conv_cin1");
        }
        return run_igemm(GEOM_S1, gy, w, WL_IOK, 1, gx, ws, B, D, H, W, Cout, Cin, ep, stream);
    }
    MVS_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, MVS_ERR_SHAPE, "conv3d_dgrad stride 2: D,H,W must be even");
    return run_igemm(GEOM_TR2, gy, w, WL_IOK, 0, gx, ws, B, D / 2, H / 2, W / 2, Cout, Cin, ep, stream);
}

// The same with `add` [B,D,H,W,Cin] (or NULL) summed into the result in the epilogue: gx = d conv3d / dx + add.  A tensor with two
// consumers (the U-Net's skip connections, mvsnet.py:70-72) gets its second gradient contribution without a separate pass over it.
extern "C" int mvs_conv3d_dgrad_acc(const float* gy, const float* w, const float* add, float* gx, float* ws, int B, int D, int H, int W,
                                    int Cin, int Cout, int stride, hipStream_t stream) {
    if (!add || (stride == 1 && Cout == 1)) {
        MVS_REQUIRE(!add, MVS_ERR_UNSUPPORTED, "conv3d_dgrad_acc: the Cout = 1 input gradient takes no summand");
        return mvs_conv3d_dgrad(gy, w, gx, ws, B, D, H, W, Cin, Cout, stride, stream);
    }
    int rc = check_stride(stride, D, H, W, "conv3d_dgrad_acc");
    if (rc) return rc;
    Epilogue ep = {nullptr, nullptr, add, 0, nullptr};
    if (stride == 1) return run_igemm(GEOM_S1, gy, w, WL_IOK, 1, gx, ws, B, D, H, W, Cout, Cin, ep, stream);
    MVS_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, MVS_ERR_SHAPE, "conv3d_dgrad_acc stride 2: D,H,W must be even");
    return run_igemm(GEOM_TR2, gy, w, WL_IOK, 0, gx, ws, B, D / 2, H / 2, W / 2, Cout, Cin, ep, stream);
}

// gw[Cout][Cin][27] = d conv3d / dw
extern "C" int mvs_conv3d_wgrad(const float* x, const float* gy, float* gw, float* ws, int B, int D, int H, int W,
                                int Cin, int Cout, int stride, hipStream_t stream) {
    int rc = check_stride(stride, D, H, W, "conv3d_wgrad");
    if (rc) return rc;
    return run_wgrad(stride == 2 ? GEOM_S2 : GEOM_S1, x, gy, gw, ws, B, D, H, W, Cin, Cout, stream);
}

// y = conv_transpose3d(x, w[Cin][Cout][3][3][3], stride, pad 1, output_padding stride-1)
extern "C" int mvs_convT3d_fwd(const float* x, const float* w, float* y, float* ws, int B, int D, int H, int W,
                               int Cin, int Cout, int stride, const float* scale, const float* shift,
                               const float* skip, int relu, float* stat_partials, hipStream_t stream) {
    int rc = check_stride(stride, D, H, W, "convT3d_fwd");
    if (rc) return rc;
    Epilogue ep = {scale, shift, skip, relu, stat_partials};
    if (stride == 1) return run_igemm(GEOM_S1, x, w, WL_IOK, 1, y, ws, B, D, H, W, Cin, Cout, ep, stream);
    return run_igemm(GEOM_TR2, x, w, WL_IOK, 0, y, ws, B, D, H, W, Cin, Cout, ep, stream);
}

// gx[B,D,H,W,Cin] from gy[B,sD,sH,sW,Cout]
extern "C" int mvs_convT3d_dgrad(const float* gy, const float* w, float* gx, float* ws, int B, int D, int H, int W,
                                 int Cin, int Cout, int stride, hipStream_t stream) {
    int rc = check_stride(stride, D, H, W, "convT3d_dgrad");
    if (rc) return rc;
    Epilogue ep = {nullptr, nullptr, nullptr, 0, nullptr};
    if (stride == 1) return run_igemm(GEOM_S1, gy, w, WL_OIK, 0, gx, ws, B, D, H, W, Cout, Cin, ep, stream);
    return run_igemm(GEOM_S2, gy, w, WL_OIK, 0, gx, ws, B, 2 * D, 2 * H, 2 * W, Cout, Cin, ep, stream);
}

extern "C" int mvs_convT3d_dgrad_acc(const float* gy, const float* w, const float* add, float* gx, float* ws, int B, int D, int H,
                                     int W, int Cin, int Cout, int stride, hipStream_t stream) {
    int rc = check_stride(stride, D, H, W, "convT3d_dgrad_acc");
    if (rc) return rc;
    Epilogue ep = {nullptr, nullptr, add, 0, nullptr};
    if (stride == 1) return run_igemm(GEOM_S1, gy, w, WL_OIK, 0, gx, ws, B, D, H, W, Cout, Cin, ep, stream);
    return run_igemm(GEOM_S2, gy, w, WL_OIK, 0, gx, ws, B, 2 * D, 2 * H, 2 * W, Cout, Cin, ep, stream);
}

// gw[Cin][Cout][27]
extern "C" int mvs_convT3d_wgrad(const float* x, const float* gy, float* gw, float* ws, int B, int D, int H, int W,
                                 int Cin, int Cout, int stride, hipStream_t stream) {
    int rc = check_stride(stride, D, H, W, "convT3d_wgrad");
    if (rc) return rc;
    // roles swap: the tensor indexed at o*S + tap - 1 is gy (fine grid), the "G" operand is x (coarse grid)
    if (stride == 1) return run_wgrad(GEOM_S1, gy, x, gw, ws, B, D, H, W, Cout, Cin, stream);
    return run_wgrad(GEOM_S2, gy, x, gw, ws, B, 2 * D, 2 * H, 2 * W, Cout, Cin, stream);
}
