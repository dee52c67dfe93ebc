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

// tuning knob "cout1_d4", bit 1: the Cout = 1 layer of the bf16 inference path with four outputs per thread (conv3d_bf16.hip:
// 0.426 -> 0.368 ms -- on).  The fp32 forms of the same idea (bit 0 in round 3) measured slower (profiles/r03_run16_*: forward
// 0.089 -> 0.091 ms, 16 channels 0.331 -> 0.473, input gradient 0.083 -> 0.107) and were removed in round 4.
int g_conv_cout1_d4 = 2;

extern int g_conv_split, g_conv_small, g_conv_small_wgs, g_conv_tr2pw;

#include "conv_args.h"

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void conv_pack_weights_item(const float* __restrict__ w, float* __restrict__ wp, int geom, int CC,
                                                       int Cin, int Cout, int NB, int layout, int flip, int idx) {
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



// One launch packs the weight images of a whole list of layers (blockIdx.y = list entry): the regulariser packs the images of
// its 10 forward and 10 input-gradient convolutions ONCE per training step (rounds 1-3: one 5-us launch in front of every
// convolution, 19 per step -- profiles/r03_final_rocprofv3_kernel_stats.csv).  kind 0: implicit-GEMM image, kind 1: the
// [tap][co] table of the Cin == 1 direct kernel (wt[t][co] = W[0][co][26 - t]; W is [1][C][3][3][3]).
struct PackItem {
    const float* w;
    float* wp;
    int kind, geom, CC, Cin, Cout, NB, layout, flip, total;
};
#define MVS_PACK_BATCH_MAX 24
struct PackBatch {
    PackItem it[MVS_PACK_BATCH_MAX];
};
__global__ __launch_bounds__(256) void conv_pack_batch_kernel(PackBatch pb) {
    const PackItem& p = pb.it[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.total) return;
    if (p.kind == 1) {
        const int t = idx / p.Cout, c = idx % p.Cout;
        p.wp[idx] = p.w[(size_t)c * 27 + (26 - t)];
    } else {
        conv_pack_weights_item(p.w, p.wp, p.geom, p.CC, p.Cin, p.Cout, p.NB, p.layout, p.flip, idx);
    }
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
// SIDE: the epilogue reads side inputs (a.skip and / or a.bn_raw).  A separate instantiation, so that the plain kernels (training
// forward, conv0's input gradient) do not pay its registers: <S1, 16, 1> 116 -> 188 VGPRs with everything in one kernel.
// (SIDE 2: the side inputs of a one-Cout-tile kernel are requested before the class's k-loop -- knob "side_pre".)
template <int GEOM, int CC, int NB, int SIDE>
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
    float bmu[NB], bis[NB], bsc[NB], bsh[NB];   // bn_raw: mean, invstd, scale, shift of the lane's output channel(s)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        st1[nb] = 0.f; st2[nb] = 0.f;
        bmu[nb] = bis[nb] = bsc[nb] = bsh[nb] = 0.f;
        if (SIDE && a.bn_raw) {
            const int co = G::PW ? (l15 & 7) : (nb0 + nb) * 16 + l15;
            if (co < a.Cout) { bmu[nb] = a.bn_stats[co]; bis[nb] = a.bn_stats[a.Cout + co]; bsc[nb] = a.bn_stats[2 * a.Cout + co]; bsh[nb] = a.bn_stats[3 * a.Cout + co]; }
        }
    }

    const int nchunks = G::BASE == GEOM_TR2 ? 1 : a.Cin / CC;
    const int KSF = ksteps_for(27, CC);

    // side inputs of the epilogue (see there): m-blocks per load group, and -- one 16-wide Cout tile, i.e. the narrow HBM-bound
    // layers -- requested BEFORE the class's k-loop, so that they arrive under its MFMAs
    constexpr int EG = (MB * NB * 8 <= 64) ? MB : 1;
    constexpr bool SIDE_PRE = SIDE == 2 && NB == 1 && EG == MB;
    auto pd_of = [&](int cls) { return G::PW ? (cls >> 1) & 1 : (cls >> 2) & 1; };
    auto ph_of = [&](int cls) { return G::PW ? cls & 1 : (cls >> 1) & 1; };
    auto pw_of = [&](int cls) { return G::PW ? (l15 >> 3) : cls & 1; };
    auto load_side = [&](int cls, int mb0, float (&sk)[EG][4][NB], float (&rwv)[EG][4][NB]) {
#pragma unroll
        for (int e = 0; e < EG; ++e) {
            const int f = wave * MB + mb0 + e;
            const int qd = qd0 + f / G::TQH, qh = qh0 + f % G::TQH;
            const int od = qd * G::OS + pd_of(cls), oh = qh * G::OS + ph_of(cls);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qw = qw0 + 4 * g + r;
                const int ow = qw * G::OS + pw_of(cls);
                const size_t obase = ((((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.Cout;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int co = G::PW ? (l15 & 7) : (nb0 + nb) * 16 + l15;
                    const bool ok = qd < a.QD && qh < a.QH && qw < a.QW && co < a.Cout;
                    sk[e][r][nb] = (ok && a.skip) ? a.skip[obase + co] : 0.f;
                    rwv[e][r][nb] = (ok && a.bn_raw) ? a.bn_raw[obase + co] : 0.f;
                }
            }
        }
    };
    float skp[SIDE_PRE ? MB : 1][4][NB], rwp[SIDE_PRE ? MB : 1][4][NB];

    for (int cls = 0; cls < G::NCLS; ++cls) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (SIDE_PRE) {
            if (a.skip || a.bn_raw) load_side(cls, 0, skp, rwp);
        }

        for (int chunk = 0; chunk < nchunks; ++chunk) {
            if (cls == 0) {
                // ---- stage the input halo region (channels [chunk*CC, +CC)) into LDS ----
                __syncthreads();
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
        // The epilogue's side inputs (skip summand, raw tensor of the backward statistics) are read in their own phase, ALL loads
        // of a group of m-blocks before the group's first store: a.y may alias them as far as the compiler knows, so a load written
        // after a store waits for it, and the per-element form paid one memory round trip per output element (conv1's input
        // gradient 0.084 -> 0.141 ms with the summand, 0.201 ms with the statistics: profiles/r04_run1_kernels.log).
#pragma unroll
        for (int mb0 = 0; mb0 < MB; mb0 += EG) {
            float sk[EG][4][NB], rwv[EG][4][NB];
            if (SIDE && !SIDE_PRE && (a.skip || a.bn_raw)) load_side(cls, mb0, sk, rwv);
#pragma unroll
            for (int e = 0; e < EG; ++e) {
                const int mb = mb0 + e;
                const int f = wave * MB + mb;
                const int qd = qd0 + f / G::TQH, qh = qh0 + f % G::TQH;
                if (qd >= a.QD || qh >= a.QH) continue;
                const int od = qd * G::OS + pd_of(cls), oh = qh * G::OS + ph_of(cls);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qw = qw0 + 4 * g + r;
                    if (qw >= a.QW) continue;
                    const int ow = qw * G::OS + pw_of(cls);
                    const size_t obase = ((((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.Cout;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int co = G::PW ? (l15 & 7) : (nb0 + nb) * 16 + l15;
                        if (co >= a.Cout) continue;
                        float v = acc[mb][nb][r];
                        float sv1 = v, sv2 = v * v;
                        if (a.scale) v = v * a.scale[co] + a.shift[co];
                        else if (a.shift) v = v + a.shift[co];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (SIDE && a.skip) v += SIDE_PRE ? skp[SIDE_PRE ? mb : 0][r][nb] : sk[e][r][nb];
                        if (SIDE && a.bn_raw) {
                            // v is the complete output gradient of a BatchNorm+ReLU block at this voxel: its backward statistics
                            const float rw = SIDE_PRE ? rwp[SIDE_PRE ? mb : 0][r][nb] : rwv[e][r][nb];
                            sv1 = (rw * bsc[nb] + bsh[nb] > 0.f) ? v : 0.f;
                            sv2 = sv1 * ((rw - bmu[nb]) * bis[nb]);
                        }
                        st1[nb] += sv1;
                        st2[nb] += sv2;
                        a.y[obase + co] = v;
                    }
                }
            }
        }
    }

    if (a.slots) {
        // reduce the 4 lane groups of the wave, then the 4 waves, -> one slot row [2][Cout] (fp64 atomics)
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
                MVS_GLOBAL_ATOMIC_ADD_F64(a.slots + ((size_t)(blockIdx.x & (a.nslots - 1)) * 2 + stat) * a.Cout + nb0 * 16 + n, (double)s);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Cin == 1 (input gradient of the Cout=1 probability layer): direct form, one thread per voxel.
// y[v][co] = sum_t x[v + t - 1] * wt[t][co]   (wt already flipped / transposed by the caller-side packer)
// ------------------------------------------------------------------------------------------------
// VPT (round 6): voxels per thread -- the workgroup walks VPT consecutive runs of 256 voxels and reduces the backward statistics ONCE
// (the wave reduction is 16 values x 6 shuffle steps per thread: a quarter of the kernel's vector instructions at one voxel per thread).
template <int COUT, int VPT>
__global__ __launch_bounds__(256) void conv_cin1_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                        float* __restrict__ y, int B, int D, int H, int W,
                                                        const float* __restrict__ bn_raw, const float* __restrict__ bn_stats,
                                                        double* __restrict__ slots, int nslots) {
    // Round 4 form: branch-free.  The 27 input values sit at per-axis CLAMPED coordinates, so all 27 loads are unconditional and
    // issued back to back (the round 1-3 loop tested the bounds of every tap and `continue`d: one load -> wait -> FMAs round trip
    // per tap, 0.083 ms for a 15.7 MB read + 126 MB write); out-of-volume taps are zeroed by three per-axis flags; the weights are
    // read at compile-time offsets of the read-only table (uniform: scalar loads, SGPR operands of the FMAs), not from LDS.
    __shared__ float red[4 * 2 * COUT];
    const size_t total = (size_t)B * D * H * W;
    float sv[2 * COUT];
#pragma unroll
    for (int k = 0; k < 2 * COUT; ++k) sv[k] = 0.f;
#pragma unroll 1
    for (int it = 0; it < VPT; ++it) {
        const size_t v0 = ((size_t)blockIdx.x * VPT + it) * 256 + threadIdx.x;
        const bool live = v0 < total;
        const size_t v = live ? v0 : total - 1;
        const int w_ = (int)(v % W), h_ = (int)((v / W) % H), d_ = (int)((v / ((size_t)W * H)) % D);
        // the raw tensor of the backward statistics: requested first (it arrives under the taps' loads and FMAs)
        float4 rwq[COUT / 4];
#pragma unroll
        for (int q = 0; q < COUT / 4; ++q)
            rwq[q] = (slots && live) ? *reinterpret_cast<const float4*>(bn_raw + v * COUT + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        int dofs[3], hofs[3], wofs[3];
        bool vd[3], vh[3], vw[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int d = d_ + i - 1, h = h_ + i - 1, w = w_ + i - 1;
            vd[i] = d >= 0 && d < D; vh[i] = h >= 0 && h < H; vw[i] = w >= 0 && w < W;
            dofs[i] = (min(max(d, 0), D - 1) - d_) * H * W;
            hofs[i] = (min(max(h, 0), H - 1) - h_) * W;
            wofs[i] = min(max(w, 0), W - 1) - w_;
        }
        const float* __restrict__ xc = x + v;
        float xv[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) xv[t] = xc[dofs[t / 9] + hofs[(t / 3) % 3] + wofs[t % 3]];
        float acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const float xt = (vd[t / 9] && vh[(t / 3) % 3] && vw[t % 3]) ? xv[t] : 0.f;
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = fmaf(xt, wt[t * COUT + c], acc[c]);
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < COUT; c += 4)
                *reinterpret_cast<float4*>(y + v * COUT + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
        }
        if (slots && live) {
            // y is the complete output gradient of the BatchNorm+ReLU block in front of this layer (bn_raw = that block's raw
            // output): its backward statistics (sum dyh, sum dyh*xhat) per channel, summed over the thread's voxels
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                const float4 rq = rwq[c / 4];
                const float rw = (c & 3) == 0 ? rq.x : ((c & 3) == 1 ? rq.y : ((c & 3) == 2 ? rq.z : rq.w));
                const float d1 = (rw * bn_stats[2 * COUT + c] + bn_stats[3 * COUT + c] > 0.f) ? acc[c] : 0.f;
                sv[c] += d1;
                sv[COUT + c] += d1 * ((rw - bn_stats[c]) * bn_stats[COUT + c]);
            }
        }
    }
    if (slots) {
        // -> wave sums -> workgroup sums -> one slot row
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 2 * COUT; ++k) {
            float t = sv[k];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) t += __shfl_xor(t, m);
            if (lane == 0) red[wave * 2 * COUT + k] = t;
        }
        __syncthreads();
        if (threadIdx.x < 2 * COUT) {
            const int k = threadIdx.x;
            const float t = (red[k] + red[2 * COUT + k]) + (red[4 * COUT + k] + red[6 * COUT + k]);
            MVS_GLOBAL_ATOMIC_ADD_F64(slots + (size_t)(blockIdx.x & (nslots - 1)) * 2 * COUT + k, (double)t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  dW[tap][ci][co] = sum_{b,o} X[b, o*S + tap - 1][ci] * G[b,o][co]
// GEMM view: M = ci (CC=16) or (tap pair, ci) (CC=8), N = co, K = positions.  Persistent workgroups
// walk tiles, keep dW for their (ci chunk, co chunk) in MFMA accumulators, and emit one partial
// image each; wgrad_reduce sums the partial images deterministically.
// ------------------------------------------------------------------------------------------------

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
    // (measured and rejected in round 4, profiles/r04_run4_*: the weights as scalar loads at compile-time offsets of the parameter
    //  instead of this LDS image -- 0.089 -> 0.173 ms: scalar loads and LDS reads share the lgkmcnt counter, so every wait for a
    //  weight also drained the tile reads in flight)
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

// The same layer with FOUR outputs per thread (round 4, 8 input channels: the probability layer of MVSNet at full resolution).
// The kernel above is LDS-bandwidth bound: 54 tile reads + 54 weight reads of 16 bytes per output voxel = 1.7 KB of LDS traffic per
// voxel, 86 us at config 2 against 28 us of HBM time (142 MB).  Here a thread owns 4 outputs that are consecutive in H: the 6 input
// rows under them are read once per (kd, kw) instead of 3 rows per output (108 reads per 4 outputs), and a weight quad is read once
// per thread and used for all rows (54 reads per 4 outputs): 0.65 KB per voxel.  Tile 4 x 8 x 32 outputs (a wave = one W row of 32
// x 2 H groups), halo 6 x 10 x 34 voxels kept as two 4-channel planes so the 16 lanes of a b128 read phase, consecutive in W, hit
// 16 different bank quads (a voxel stride of 8 floats would give 2-way conflicts): 65 KB, two workgroups per CU.
constexpr int CO1_TD = 4, CO1_TH = 8, CO1_TW = 32, CO1_RD = CO1_TD + 2, CO1_RH = CO1_TH + 2, CO1_RW = CO1_TW + 2;
__global__ __launch_bounds__(256) void conv_cout1_h4_kernel(ConvArgs a, const float* __restrict__ w) {
    constexpr int NR = CO1_RD * CO1_RH * CO1_RW;
    __shared__ __attribute__((aligned(16))) float tile[2 * NR * 4];   // [channel quad][voxel][4]
    __shared__ __attribute__((aligned(16))) float wl[27 * 8];         // [tap][ci]
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tw = t % a.ntw; t /= a.ntw;
    const int th = t % a.nth; t /= a.nth;
    const int td = t % a.ntd; t /= a.ntd;
    const int b = t;
    const int qd0 = td * CO1_TD, qh0 = th * CO1_TH, qw0 = tw * CO1_TW;
    for (int i = tid; i < 27 * 8; i += 256) wl[i] = w[(size_t)(i % 8) * 27 + i / 8];   // W[0][ci][tap]
    stage_batched<NR * 2>(tile, tid, [&](int i, const float*& src, int& o) {
        const int vox = i >> 1, cq = i & 1;
        const int rw = vox % CO1_RW, rh = (vox / CO1_RW) % CO1_RH, rd = vox / (CO1_RW * CO1_RH);
        const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
        o = (cq * NR + vox) * 4;
        if (id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
            src = a.x + ((((size_t)b * a.Di + id) * a.Hi + ih) * a.Wi + iw) * 8 + 4 * cq;
    });
    __syncthreads();
    const int pw = tid & 31, hg = (tid >> 5) & 1, pd = tid >> 6;
    const int base = ((pd * CO1_RH + 4 * hg) * CO1_RW + pw) * 4;      // halo voxel under output (pd, 4 hg, pw), tap (0, 0, 0)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // (kd, kw) as a real loop of 9 trips: fully unrolled, hipcc hoists all 162 LDS reads ahead of the arithmetic (512 VGPRs + scratch)
#pragma unroll 1
    for (int kdw = 0; kdw < 9; ++kdw) {
        const int kd = kdw / 3, kw = kdw % 3;
#pragma unroll
            for (int cq = 0; cq < 2; ++cq) {
                float4 wv[3];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) wv[kh] = *reinterpret_cast<const float4*>(&wl[((kd * 3 + kh) * 3 + kw) * 8 + 4 * cq]);
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const float4 xv = *reinterpret_cast<const float4*>(&tile[cq * NR * 4 + base + ((kd * CO1_RH + r) * CO1_RW + kw) * 4]);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int o = r - kh;
                        if (o >= 0 && o < 4) {
                            acc[o] = fmaf(xv.x, wv[kh].x, acc[o]); acc[o] = fmaf(xv.y, wv[kh].y, acc[o]);
                            acc[o] = fmaf(xv.z, wv[kh].z, acc[o]); acc[o] = fmaf(xv.w, wv[kh].w, acc[o]);
                        }
                    }
                }
            }
    }
    const int qd = qd0 + pd, qw = qw0 + pw;
#pragma unroll
    for (int o4 = 0; o4 < 4; ++o4) {
        const int qh = qh0 + 4 * hg + o4;
        if (qd < a.QD && qh < a.QH && qw < a.QW) {
            const size_t o = (((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qw;
            float v = acc[o4];
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
    // BatchNorm statistics of the raw output: per-lane sums over ALL tiles of this persistent workgroup, one butterfly and one
    // slot row per workgroup at the end (rounds 1-3: one butterfly and one partial row per tile)
    float s1[8], s2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
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
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * h + r;
            float val = acc[0][h][r] + acc[1][h][r];
            if (!inside || co >= a.Cout) val = 0.f;
            s1[co] += val;
            s2[co] += val * val;
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
    }   // tile loop
    if (a.slots) {
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
                MVS_GLOBAL_ATOMIC_ADD_F64(a.slots + ((size_t)(blockIdx.x & (a.nslots - 1)) * 2 + stat) * a.Cout + co,
                                          (double)(red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid]));
        }
    }
}


// Weight gradient for CG == 8 (conv0, mvsnet.py:40) with the 16-block 4x4x1 MFMA: per position ONE MFMA
// accumulates (4 taps x 16 input channels) x (4 output channels); 7 tap groups x 2 channel halves = 14 MFMAs
// per position, 96 % useful (the 16x16x4 form pads N 8 -> 16 and was 50 % useful).  Each wave takes a
// quarter of the tile's positions; persistent over tiles, dW kept in accumulators, waves summed through LDS.
// BC = true: the output gradient is the MFMA's broadcast operand instead (cbsz = 4: one VGPR pair carries g of 16
// consecutive W positions, abid selects the position), so g costs 2 LDS reads per 16 positions instead of 2 per
// position and each lane ends up owning its (tap, input channel) for 4 output channels.
// NCH: 16-channel chunks of X one workgroup handles (blockIdx.y counts groups of NCH chunks).  NCH = 2 at conv0 (32 input channels):
// the X halo tile of BOTH chunks -- the whole 128-byte line of every voxel -- is staged by the workgroup that needs it, the output
// gradient tile once for both (NCH = 1 fetched every X line from two workgroups and the g tile twice: 2.12 GB of HBM traffic for
// 0.63 GB algorithmic); 110 KB of LDS = one workgroup per CU, which is all this kernel is given on the side stream.
template <bool BC, int NCH>
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
    __shared__ __attribute__((aligned(16))) float xt[NCH * NRP * CCP];
    __shared__ __attribute__((aligned(16))) float gt[NPOS * 8];
    __shared__ int tapmap[32];   // slot (tg, lane group) -> tap id (27 = dummy)
    __shared__ int tapoff[32];   // slot -> LDS offset of that tap
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk0 = blockIdx.y * NCH;
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
    f32x4 acc[NCH][NTG][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int tg = 0; tg < NTG; ++tg) { acc[ch][tg][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[ch][tg][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    constexpr int NXI = G::RD * G::RH * G::RW * (CC / 4);
    constexpr int XIT = (NXI + 255) / 256, GIT = (NPOS * 2 + 255) / 256;
    float4 xv[NCH][XIT], gv[GIT];
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
        const float* __restrict__ xb = a.x + ((((size_t)b * a.Di + qd0) * a.Hi + qh0) * a.Wi + qw0) * a.CX + chunk0 * CC;
        const float* __restrict__ gb = a.g + ((((size_t)b * a.QD + qd0) * a.QH + qh0) * a.QW + qw0) * 8;
        const bool interior = qd0 >= 1 && qd0 + G::TQD + 1 <= a.Di && qh0 >= 1 && qh0 + G::TQH + 1 <= a.Hi &&
                              qw0 >= 1 && qw0 + G::TQW + 1 <= a.Wi;
        if (interior) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int k = 0; k < XIT; ++k)
                    if (tid + 256 * k < NXI) xv[ch][k] = *reinterpret_cast<const float4*>(xb + ch * CC + xrel[k]);
#pragma unroll
            for (int k = 0; k < GIT; ++k)
                if (tid + 256 * k < NPOS * 2) gv[k] = *reinterpret_cast<const float4*>(gb + grel[k]);
        } else {
#pragma unroll
            for (int k = 0; k < XIT; ++k) {
                const int i = tid + 256 * k, vox = i / (CC / 4);
                const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
                const int id = qd0 + rd - 1, ih = qh0 + rh - 1, iw = qw0 + rw - 1;
                const bool in = i < NXI && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch)
                    xv[ch][k] = in ? *reinterpret_cast<const float4*>(xb + ch * CC + xrel[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int k = 0; k < XIT; ++k)
                if (tid + 256 * k < NXI) *reinterpret_cast<float4*>(xt + ch * NRP * CCP + xlo[k]) = xv[ch][k];
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
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        float av[NTG];
#pragma unroll
                        for (int tg = 0; tg < NTG; ++tg) av[tg] = xt[ch * NRP * CCP + xoff0 + k * CCP + toff[tg]];
#pragma unroll
                        for (int tg = 0; tg < NTG; ++tg) {
                            acc[ch][tg][0] = mfma_4x4x1_bc(ga0, av[tg], acc[ch][tg][0], k);
                            acc[ch][tg][1] = mfma_4x4x1_bc(ga1, av[tg], acc[ch][tg][1], k);
                        }
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
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                float av[NTG];
#pragma unroll
                for (int tg = 0; tg < NTG; ++tg) av[tg] = xt[ch * NRP * CCP + xoff + toff[tg]];
#pragma unroll
                for (int tg = 0; tg < NTG; ++tg) {
                    acc[ch][tg][0] = MVS_MFMA_4x4x1(av[tg], b0, acc[ch][tg][0]);
                    acc[ch][tg][1] = MVS_MFMA_4x4x1(av[tg], b1, acc[ch][tg][1]);
                }
            }
        }
        }
    }
    // sum the 4 waves through LDS (reusing xt: 28 slots x 16 cx x 8 co = 3584 floats), then one partial image -- chunk after chunk
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
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
                            if (wv == 0) xt[idx] = acc[ch][tg][h][r];
                            else xt[idx] += acc[ch][tg][h][r];
                        }
            }
            __syncthreads();
        }
        for (int i = tid; i < 28 * 16 * 8; i += 256) {
            const int co = i & 7, cx = (i >> 3) & 15, tap = tapmap[i >> 7];
            if (tap < 27) a.part[(((size_t)blockIdx.x * 27 + tap) * a.CX + (chunk0 + ch) * 16 + cx) * 8 + co] = xt[i];
        }
    }
}

// Many partial images in ONE launch (round 4; rounds 1-3: a 16-row pre-reduction launch + the kernel above, 22 launches per
// step): a workgroup owns 16 output elements; its 16 x 16 threads = (element, slice) walk the images slice, slice + 16, ... with
// all loads of a batch of four in flight, then the 16 slices are summed through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void conv_wgrad_reduce_wide_kernel(const float* __restrict__ part, int nparts, int CX, int CG,
                                                                     float* __restrict__ gw) {
    __shared__ float red[16][17];
    const int n = 27 * CX * CG;
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
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][el];
        const int cg = e % CG, cx = (e / CG) % CX, tap = e / (CG * CX);
        gw[((size_t)cg * CX + cx) * 27 + tap] = t;
    }
}

// ================================================================================================
// host side
// ================================================================================================
static int pick_cc(int geom, int cin) {
    if (geom == GEOM_TR2) return cin;
    if (geom == GEOM_S2) return 8;
    return (cin % 16 == 0) ? 16 : 8;
}

// Round 6, knob "cc_wide" (MEASURED AND REJECTED, default off): the quarter-tile kernels of the deep U-Net levels with ALL input channels
// as ONE chunk (32 / 64 for stride 1, 16 / 32 for stride 2) instead of 16- / 8-channel chunks -- one exposed staging round trip and two
// barriers per workgroup instead of 2-4 and 4-8.  It is SLOWER: 64 -> 64 at 24x16x20 0.040 -> 0.077 ms forward, 0.042 -> 0.078 input
// gradient; 32 -> 32 at 48x32x40 0.052 -> 0.062; 32 -> 64 stride 2 0.034 -> 0.058; step 4.699 -> 4.871 ms (profiles/r06_run11_*).  The
// small chunks ARE the pipeline of these launches: 24 KB of LDS per workgroup lets several workgroups share a CU and overlap each other's
// stage / MFMA phases, while a 78-128 KB tile leaves one workgroup per CU alone with a 28-load staging phase and nothing to hide it.
int g_conv_cc_wide = 0;
int g_conv_cin1_vpt = 1;   // tuning knob "cin1_vpt": voxels per thread of the Cout == 1 layer's input gradient (1 = default; 4 = four from 1 M voxels on, 5 = always four).  MEASURED AND REJECTED (round 6, profiles/r06_run13_*): four voxels per thread amortise the wave reduction of the backward statistics but serialise the loads of a quarter as many workgroups: 0.090 -> 0.123 ms, step 4.726 -> 4.749
static int pick_cc_k(int geom, int kgeom, int cin) {
    if (g_conv_cc_wide) {
        if (kgeom == GEOM_S1_SMALL && (cin == 32 || cin == 64)) return cin;
        if (kgeom == GEOM_S2_SMALL && (cin == 16 || cin == 32)) return cin;
    }
    return pick_cc(geom, cin);
}

static size_t packed_floats(int geom, int cin, int cout) {
    const int cc = pick_cc(geom, cin);
    int nb = mvs_cdiv(cout, 16);
    if (nb == 3) nb = 4;
    return (size_t)total_ksteps(geom, cin, cc) * nb * 256;
}

int g_conv_split = 1;       // tuning knob "conv_split" (mvs_set_tuning): 0 keeps all Cout tiles in one workgroup
int g_conv_tr2pw = 1;       // tuning knob "tr2pw": transposed stride-2 conv with Cout == 8 as W-parity-merged GEMMs (GEOM_TR2_PW)
int g_conv_small_wgs = 384;   // tuning knob "conv_small_wgs": quarter-size tiles below this many workgroups (~1.5 per CU)
int g_conv_small = 1;   // tuning knob "conv_small": quarter-size workgroup tiles for under-filled launches (0 never, 1 auto, 2 always)
int g_conv_c8 = 7;      // tuning knob "k8", bit mask: 1|2 = Cout==8 stride-1 layers run the 4x4x1 MFMA forward with the weights as the broadcast operand (0: generic kernel), +4 = weight gradient with g as the broadcast operand
int g_conv_wgrad_groups = 768;   // tuning knob "wgrad_groups": persistent workgroups of the generic weight-gradient kernels (<= 768)
int g_conv_cout1_h4 = 1;         // tuning knob "cout1_h4": the 8 -> 1 layer with four outputs per thread (conv_cout1_h4_kernel); 0: one output per thread
int g_conv_wgrad8_groups = 192;  // tuning knob "wgrad8_groups": ... of the CG == 8 kernel (conv0; <= 512).  On the side stream it runs under the plane-sweep backward and the 2-D extractor's backward.  While the main stream was the longer one the step was faster the less this kernel took from it (5.440 ms at 512, 5.416 at 384, 5.413 at 256, 5.400 at 128: profiles/r04_run9_*); since the extractor's weight gradients became one launch the side stream ends last (+0.05 ms at the join) and 192 is the best of 128 / 192 / 256 / 384 / 512 (5.28 / 5.23 / 5.25 / 5.24 / 5.28: profiles/r04_run25_*); round 5, conv_c8_wgrad_gs_kernel (sixteen waves, 106 KB of LDS per workgroup): 4.77 ms/step at 192, 4.78 at 176 / 160, 4.79 at 224, 4.83 at 256, 4.85 at 128 (profiles/r05_final_session.log, r05_run33_*)
int g_conv_wgrad8_nch = 2;       // tuning knob "wgrad8_nch": 2 = the CG == 8 weight gradient stages both 16-channel chunks of a 32-channel X in one workgroup.  Default since the end of round 4: at 192 workgroups the one-chunk form draws 2.31 GB from HBM per launch (0.63 GB algorithmic; 1.14 GB at 128 workgroups, 2.04 GB at 256: which halo lines neighbouring workgroups find in their XCD's L2 depends on the count), the two-chunk form 1.08 GB, at the same step time (5.313 vs 5.298 ms, profiles/r04_run31_*); alone on the GPU it is the slower kernel (0.85 vs 0.76 ms at 192 workgroups)
int g_conv_wgrad_small = 0;   // tuning knob "wgrad_small": 1 = quarter-size tiles in the generic weight-gradient kernel for 8-channel / stride-2 layers with many tiles, 2 = for every layer with many tiles, 3 = always (tests)
int g_conv_side_pre = 1;   // tuning knob "side_pre": one-Cout-tile kernels with epilogue side inputs (skip / bn_raw) request them before the k-loop (1) or at the top of the epilogue (0)
int g_conv_pers = 1;    // tuning knob "conv_pers": 1 = one-chunk layers (16 -> <= 16, 8 -> 32 stride 1; 8 -> <= 16 stride 2) run conv3d_pers.hip
int g_conv_pers_min_wgs = 1024;   // tuning knob "conv_pers_min": ... when the one-tile kernel would launch at least this many workgroups (a persistent grid needs several tiles per workgroup)
int g_conv_wgrad_pers = 1;   // tuning knob "wgrad_pers"
bool conv_wgrad_pers_serves(int geom, int CX, int CG);
int run_conv_wgrad_pers(int geom, const WgradArgs& a, int max_groups, hipStream_t st);
bool conv_c8_wgrad_gs_serves(int geom, int CX, int CG);
int run_conv_c8_wgrad_gs(const WgradArgs& a, int max_groups, int waves, hipStream_t st);
int g_conv_wgrad8_gs = 2;   // tuning knob "wgrad8_gs": conv0's weight gradient in the output-gradient-shifted form on the 16x16x4 MFMA (conv3d_pers.hip: conv_c8_wgrad_gs_kernel); 1: eight waves per workgroup, 2: sixteen (0.463 / 0.447 ms alone); 0: conv_c8_wgrad_kernel (4x4x1 MFMA, X shifted)
bool conv_pers_serves(int geom, int cin, int cout);
int run_conv_pers(int geom, const ConvArgs& a, hipStream_t st);
bool conv_x3_serves(int geom, const ConvArgs& a);       // conv3d_x3.hip (knob "conv0_x3", opt-in)
int run_conv_x3(const ConvArgs& a, const float* w, int wlayout, int flip, hipStream_t st);
bool conv_x3_fwd_serves(int geom, const ConvArgs& a);
int run_conv_x3_fwd(const ConvArgs& a, const float* w, int wlayout, int flip, float* ws, hipStream_t st);
extern int g_conv_x3;
int g_conv_xcd = 1;     // tuning knob "xcd": XCD-aware tile order in the broadcast-operand forward and the Cout == 8 weight gradient

template <int GEOM, int CC>
static int launch_igemm_nb(const ConvArgs& a, int NB, int nblocks, hipStream_t st) {
    dim3 grid(nblocks, a.nb_total / NB), block(256);
    if (a.skip || a.bn_raw) {
        switch (NB) {
            case 1:
                if (g_conv_side_pre) MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 1, 2>), grid, block, 0, st, a);
                else MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 1, 1>), grid, block, 0, st, a);
                break;
            case 2: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 2, 1>), grid, block, 0, st, a); break;
            case 4: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 4, 1>), grid, block, 0, st, a); break;
            default: mvs_set_error("conv igemm: Cout tile count %d unsupported", NB); return MVS_ERR_UNSUPPORTED;
        }
        return mvs_check_launch("conv_igemm");
    }
    switch (NB) {
        case 1: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 1, 0>), grid, block, 0, st, a); break;
        case 2: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 2, 0>), grid, block, 0, st, a); break;
        case 4: MVS_LAUNCH((conv_igemm_kernel<GEOM, CC, 4, 0>), grid, block, 0, st, a); break;
        default: mvs_set_error("conv igemm: Cout tile count %d unsupported", NB); return MVS_ERR_UNSUPPORTED;
    }
    return mvs_check_launch("conv_igemm");
}

// Which tiling the generic kernel runs with.  Small volumes (deep U-Net levels) have too few tiles to fill 256 CUs: first one
// 16-wide Cout tile per workgroup, and if that still leaves fewer than ~1.5 workgroups per CU, quarter-size tiles (knob
// "conv_small": 0 never, 1 auto, 2 always).
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
    const float* scale; const float* shift; const float* skip; int relu;
    double* slots; int nslots;                       // BatchNorm statistic slots (see ConvArgs)
    const float* bn_raw; const float* bn_stats;      // backward statistics of the block whose output gradient is written
};

// One convolution-shaped op as the implicit-GEMM kernels see it: in [B,Di,Hi,Wi,cin] -> out [B,Do,Ho,Wo,cout];
// coarse grid: S1/S2 -> output dims, TR2 -> input dims; weights read with (wlayout, flip) from the parameter tensor.
struct IgemmPlan {
    int geom, wlayout, flip, B, Di, Hi, Wi, cin, cout;
    bool cin1;                                        // Cout == 1 stride-1 input gradient: the direct Cin == 1 kernel
};
enum { MVS_OP_CONV_FWD = 0, MVS_OP_CONV_DGRAD = 1, MVS_OP_CONV_WGRAD = 2,
       MVS_OP_CONVT_FWD = 3, MVS_OP_CONVT_DGRAD = 4, MVS_OP_CONVT_WGRAD = 5 };

static int check_stride(int stride, int D, int H, int W, const char* what) {
    MVS_REQUIRE(stride == 1 || stride == 2, MVS_ERR_UNSUPPORTED, "%s: stride must be 1 or 2, got %d", what, stride);
    MVS_REQUIRE(D > 0 && H > 0 && W > 0, MVS_ERR_SHAPE, "%s: bad spatial shape %dx%dx%d", what, D, H, W);
    return MVS_OK;
}

// (D,H,W) are always the spatial dims of the forward op's INPUT x; Cin / Cout those of the forward op.
static int plan_for(int op, int B, int D, int H, int W, int Cin, int Cout, int stride, IgemmPlan& p, const char* what) {
    int rc = check_stride(stride, D, H, W, what);
    if (rc) return rc;
    // element offsets are size_t everywhere (a batch-3 config-5 volume has 2.9 G elements: tests/test_gpu_parity.py); VOXEL counts,
    // tile counts and in-tile offsets are 32-bit ints, so the number of (fine-grid) voxels has to fit one
    MVS_REQUIRE(B > 0 && (long long)B * D * H * W * (stride == 2 ? 8 : 1) < (1LL << 31), MVS_ERR_SHAPE,
                "%s: %d x %d x %d x %d voxels do not fit the kernels' 32-bit voxel indices", what, B, D, H, W);
    p = {};
    p.B = B; p.cin1 = false;
    switch (op) {
        case MVS_OP_CONV_FWD:
            p.geom = stride == 2 ? GEOM_S2 : GEOM_S1; p.wlayout = WL_OIK; p.flip = 0;
            p.Di = D; p.Hi = H; p.Wi = W; p.cin = Cin; p.cout = Cout;
            return MVS_OK;
        case MVS_OP_CONV_DGRAD:
            p.cin = Cout; p.cout = Cin; p.wlayout = WL_IOK;
            if (stride == 1) {
                p.geom = GEOM_S1; p.flip = 1; p.Di = D; p.Hi = H; p.Wi = W;
                p.cin1 = Cout == 1;
            } else {
                MVS_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, MVS_ERR_SHAPE, "%s stride 2: D,H,W must be even", what);
                p.geom = GEOM_TR2; p.flip = 0; p.Di = D / 2; p.Hi = H / 2; p.Wi = W / 2;
            }
            return MVS_OK;
        case MVS_OP_CONVT_FWD:
            p.wlayout = WL_IOK; p.Di = D; p.Hi = H; p.Wi = W; p.cin = Cin; p.cout = Cout;
            if (stride == 1) { p.geom = GEOM_S1; p.flip = 1; } else { p.geom = GEOM_TR2; p.flip = 0; }
            return MVS_OK;
        case MVS_OP_CONVT_DGRAD:
            p.wlayout = WL_OIK; p.flip = 0; p.cin = Cout; p.cout = Cin;
            if (stride == 1) { p.geom = GEOM_S1; p.Di = D; p.Hi = H; p.Wi = W; }
            else { p.geom = GEOM_S2; p.Di = 2 * D; p.Hi = 2 * H; p.Wi = 2 * W; }
            return MVS_OK;
        default:
            mvs_set_error("%s: op %d has no weight image", what, op);
            return MVS_ERR_UNSUPPORTED;
    }
}

static void plan_grid(const IgemmPlan& p, int& QD, int& QH, int& QW) {
    if (p.geom == GEOM_S2) { QD = (p.Di - 1) / 2 + 1; QH = (p.Hi - 1) / 2 + 1; QW = (p.Wi - 1) / 2 + 1; }
    else { QD = p.Di; QH = p.Hi; QW = p.Wi; }
}
static bool plan_is_c8(const IgemmPlan& p, const Epilogue* ep) {
    return (g_conv_c8 & 3) && p.geom == GEOM_S1 && p.cout == 8 && (p.cin == 8 || p.cin == 16 || p.cin == 32) && !(ep && ep->bn_raw);
}
static bool plan_is_cout1(const IgemmPlan& p, const Epilogue* ep) {
    return p.geom == GEOM_S1 && p.cout == 1 && p.wlayout == WL_OIK && !p.flip && (p.cin == 8 || p.cin == 16) &&
           !(ep && (ep->slots || ep->bn_raw));
}

// What mvs_conv3d_pack_weights has to write for this op (kind -1: nothing -- the kernel reads the parameter tensor itself).
static int plan_pack(const IgemmPlan& p, const float* w, float* ws, PackItem& it) {
    it = {};
    it.w = w; it.wp = ws; it.kind = -1;
    if (p.cin1) {
        MVS_REQUIRE(p.cout == 8 || p.cout == 16, MVS_ERR_UNSUPPORTED, "conv3d_dgrad(Cout=1): Cin must be 8 or 16, got %d", p.cout);
        it.kind = 1; it.Cout = p.cout; it.total = 27 * p.cout;
        return MVS_OK;
    }
    MVS_REQUIRE(p.cin == 8 || p.cin == 16 || p.cin == 32 || p.cin == 64, MVS_ERR_UNSUPPORTED,
                "conv igemm: input channels must be 8/16/32/64, got %d", p.cin);
    MVS_REQUIRE(p.cout >= 1 && p.cout <= 64, MVS_ERR_UNSUPPORTED, "conv igemm: output channels must be <= 64, got %d", p.cout);
    MVS_REQUIRE(!(p.geom == GEOM_TR2 && p.cin < 16), MVS_ERR_UNSUPPORTED, "transposed stride-2 conv needs >= 16 input channels");
    // (an op served by the Cout == 8 / Cout == 1 kernels needs no image -- unless its epilogue carries backward statistics, which
    //  only the generic kernel implements; the image is cheap, so it is written whenever the generic kernel COULD run)
    int QD, QH, QW, kgeom, NB, nblocks;
    plan_grid(p, QD, QH, QW);
    igemm_tiling(p.geom, p.B, QD, QH, QW, p.cout, kgeom, NB, nblocks);
    const int cc = pick_cc_k(p.geom, kgeom, p.cin);
    const int nb_total = mvs_cdiv(p.cout, 16) == 3 ? 4 : mvs_cdiv(p.cout, 16);
    const int pgeom = (kgeom == GEOM_TR2 && cc == 16 && p.cout == 8 && g_conv_tr2pw) ? GEOM_TR2_PW : p.geom;
    it.kind = 0; it.geom = pgeom; it.CC = cc; it.Cin = p.cin; it.Cout = p.cout; it.NB = nb_total; it.layout = p.wlayout; it.flip = p.flip;
    it.total = (int)((size_t)total_ksteps(pgeom, p.cin, cc) * nb_total * 256);
    return MVS_OK;
}

static int launch_pack(const PackItem* items, int n, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += MVS_PACK_BATCH_MAX) {
        PackBatch pb = {};
        int m = 0, maxtotal = 0;
        for (int i = i0; i < n && m < MVS_PACK_BATCH_MAX; ++i) {
            if (items[i].kind < 0) continue;
            pb.it[m++] = items[i];
            if (items[i].total > maxtotal) maxtotal = items[i].total;
        }
        if (m == 0) continue;
        MVS_LAUNCH(conv_pack_batch_kernel, dim3(mvs_cdiv(maxtotal, 256), m), dim3(256), 0, st, pb);
    }
    return mvs_check_launch("conv_pack_weights");
}

static int run_igemm(const IgemmPlan& p, const float* in, const float* wsrc, float* out, float* ws, const Epilogue& ep,
                     int ws_packed, hipStream_t st) {
    MVS_REQUIRE(in && wsrc && out && ws, MVS_ERR_NULL, "conv: null pointer argument");
    MVS_REQUIRE(!ep.slots || (ep.nslots >= 1 && ep.nslots <= 256 && (ep.nslots & (ep.nslots - 1)) == 0), MVS_ERR_SHAPE,
                "conv: statistic slot count must be a power of two <= 256, got %d", ep.nslots);
    MVS_REQUIRE(!ep.bn_raw || (ep.bn_stats && ep.slots), MVS_ERR_NULL, "conv: bn_raw needs bn_stats and slots");
    const int geom = p.geom, B = p.B, cin = p.cin, cout = p.cout;
    ConvArgs a = {};
    a.x = in; a.y = out; a.B = B; a.Di = p.Di; a.Hi = p.Hi; a.Wi = p.Wi; a.Cin = cin; a.Cout = cout;
    a.scale = ep.scale; a.shift = ep.shift; a.skip = ep.skip; a.relu = ep.relu;
    a.slots = ep.slots; a.nslots = ep.nslots; a.bn_raw = ep.bn_raw; a.bn_stats = ep.bn_stats;
    plan_grid(p, a.QD, a.QH, a.QW);
    if (geom == GEOM_S1) { a.Do = p.Di; a.Ho = p.Hi; a.Wo = p.Wi; }
    else if (geom == GEOM_S2) { a.Do = a.QD; a.Ho = a.QH; a.Wo = a.QW; }
    else { a.Do = 2 * p.Di; a.Ho = 2 * p.Hi; a.Wo = 2 * p.Wi; }
    a.ntd = mvs_cdiv(a.QD, geom_tqd(geom)); a.nth = mvs_cdiv(a.QH, geom_tqh(geom)); a.ntw = mvs_cdiv(a.QW, 16);
    if ((g_conv_x3 & 1) && conv_x3_serves(geom, a)) return run_conv_x3(a, wsrc, p.wlayout, p.flip, st);   // opt-in: split-bf16 products
    if ((g_conv_x3 & 2) && conv_x3_fwd_serves(geom, a)) return run_conv_x3_fwd(a, wsrc, p.wlayout, p.flip, ws, st);
    if (plan_is_c8(p, &ep)) {
        // 4x4x1 MFMA with the weights as the broadcast operand, tile 4 x 4 x 16 positions (reads the parameter tensor itself)
        const int ntl = B * a.ntd * a.nth * a.ntw;
        const int nbc = ntl < 512 ? ntl : 512;            // 80 KB of LDS -> 2 resident workgroups per CU
        if (cin == 32) MVS_LAUNCH((conv_c8_fwd_bc_kernel<16, 2>), dim3(nbc), dim3(256), 0, st, a, wsrc, p.wlayout, p.flip, g_conv_xcd);
        else if (cin == 16) MVS_LAUNCH((conv_c8_fwd_bc_kernel<16, 1>), dim3(nbc), dim3(256), 0, st, a, wsrc, p.wlayout, p.flip, g_conv_xcd);
        else MVS_LAUNCH((conv_c8_fwd_bc_kernel<8, 1>), dim3(nbc), dim3(256), 0, st, a, wsrc, p.wlayout, p.flip, g_conv_xcd);
        return mvs_check_launch("conv_c8_fwd_bc");
    }
    int nblocks = B * a.ntd * a.nth * a.ntw;
    if (plan_is_cout1(p, &ep)) {
        if (cin == 8 && g_conv_cout1_h4) {
            a.ntd = mvs_cdiv(a.QD, CO1_TD); a.nth = mvs_cdiv(a.QH, CO1_TH); a.ntw = mvs_cdiv(a.QW, CO1_TW);
            MVS_LAUNCH(conv_cout1_h4_kernel, dim3(B * a.ntd * a.nth * a.ntw), dim3(256), 0, st, a, wsrc);
        } else if (cin == 8) MVS_LAUNCH((conv_cout1_kernel<8>), dim3(nblocks), dim3(256), 0, st, a, wsrc);
        else MVS_LAUNCH((conv_cout1_kernel<16>), dim3(nblocks), dim3(256), 0, st, a, wsrc);
        return mvs_check_launch("conv_cout1");
    }
    int NB, kgeom;
    a.nb_total = mvs_cdiv(cout, 16) == 3 ? 4 : mvs_cdiv(cout, 16);
    igemm_tiling(geom, B, a.QD, a.QH, a.QW, cout, kgeom, NB, nblocks);
    const int cc = pick_cc_k(geom, kgeom, cin);
    a.ntd = mvs_cdiv(a.QD, geom_tqd(kgeom)); a.nth = mvs_cdiv(a.QH, geom_tqh(kgeom));
    if (!ws_packed) {
        PackItem it;
        int rc = plan_pack(p, wsrc, ws, it);
        if (rc) return rc;
        rc = launch_pack(&it, 1, st);
        if (rc) return rc;
    } else {
        MVS_REQUIRE(cin == 8 || cin == 16 || cin == 32 || cin == 64, MVS_ERR_UNSUPPORTED, "conv igemm: input channels must be 8/16/32/64, got %d", cin);
        MVS_REQUIRE(cout >= 1 && cout <= 64, MVS_ERR_UNSUPPORTED, "conv igemm: output channels must be <= 64, got %d", cout);
        MVS_REQUIRE(!(geom == GEOM_TR2 && cin < 16), MVS_ERR_UNSUPPORTED, "transposed stride-2 conv needs >= 16 input channels");
    }
    a.wp = ws;
    // knob "conv_pers": single-chunk layers with a small weight image through the persistent LDS-DMA kernel (conv3d_pers.hip)
    if (g_conv_pers && conv_pers_serves(geom, cin, cout) && (long)nblocks * (a.nb_total / NB) >= g_conv_pers_min_wgs) {
        a.ntd = mvs_cdiv(a.QD, geom_tqd(geom)); a.nth = mvs_cdiv(a.QH, geom_tqh(geom));
        return run_conv_pers(geom, a, st);
    }
    // (the transposed 16 -> 8 layers: conv11 forward, conv1's input gradient -- when the W-parity-merged image was packed)
    if ((g_conv_pers & 1) && kgeom == GEOM_TR2 && cc == 16 && cout == 8 && g_conv_tr2pw && conv_pers_serves(GEOM_TR2_PW, cin, cout) &&
        nblocks >= g_conv_pers_min_wgs)
        return run_conv_pers(GEOM_TR2_PW, a, st);
    if (kgeom == GEOM_S1) return cc == 16 ? launch_igemm_nb<GEOM_S1, 16>(a, NB, nblocks, st)
                                          : launch_igemm_nb<GEOM_S1, 8>(a, NB, nblocks, st);
    if (kgeom == GEOM_S2) return launch_igemm_nb<GEOM_S2, 8>(a, NB, nblocks, st);
    if (kgeom == GEOM_S1_SMALL) {
        if (cc == 64) return launch_igemm_nb<GEOM_S1_SMALL, 64>(a, NB, nblocks, st);
        if (cc == 32) return launch_igemm_nb<GEOM_S1_SMALL, 32>(a, NB, nblocks, st);
        return cc == 16 ? launch_igemm_nb<GEOM_S1_SMALL, 16>(a, NB, nblocks, st) : launch_igemm_nb<GEOM_S1_SMALL, 8>(a, NB, nblocks, st);
    }
    if (kgeom == GEOM_S2_SMALL) {
        if (cc == 32) return launch_igemm_nb<GEOM_S2_SMALL, 32>(a, NB, nblocks, st);
        if (cc == 16) return launch_igemm_nb<GEOM_S2_SMALL, 16>(a, NB, nblocks, st);
        return launch_igemm_nb<GEOM_S2_SMALL, 8>(a, NB, nblocks, st);
    }
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

// input gradient of the Cout == 1 layer (direct kernel over the [tap][co] table in ws)
static int run_cin1(const IgemmPlan& p, const float* gy, const float* w, float* gx, float* ws, const Epilogue& ep, int ws_packed,
                    hipStream_t st) {
    MVS_REQUIRE(gy && w && gx && ws, MVS_ERR_NULL, "conv3d_dgrad: null pointer argument");
    MVS_REQUIRE(!ep.skip, MVS_ERR_UNSUPPORTED, "conv3d_dgrad: the Cout = 1 input gradient takes no summand");
    MVS_REQUIRE(!ep.bn_raw || (ep.bn_stats && ep.slots), MVS_ERR_NULL, "conv3d_dgrad: bn_raw needs bn_stats and slots");
    MVS_REQUIRE(!ep.slots || (ep.bn_raw && ep.nslots >= 1 && ep.nslots <= 256 && (ep.nslots & (ep.nslots - 1)) == 0), MVS_ERR_SHAPE,
                "conv3d_dgrad: bad statistic slots");
    const int C = p.cout;
    if (!ws_packed) {
        PackItem it;
        int rc = plan_pack(p, w, ws, it);
        if (rc) return rc;
        rc = launch_pack(&it, 1, st);
        if (rc) return rc;
    } else {
        MVS_REQUIRE(C == 8 || C == 16, MVS_ERR_UNSUPPORTED, "conv3d_dgrad(Cout=1): Cin must be 8 or 16, got %d", C);
    }
    const size_t total = (size_t)p.B * p.Di * p.Hi * p.Wi;
    // knob "cin1_vpt": 4 = four voxels per thread from 1 M voxels on (small volumes keep one: they need the workgroups), 5 = always (tests), 1 = never
    const int vpt = (g_conv_cin1_vpt == 5 || (g_conv_cin1_vpt == 4 && total >= (size_t)1 << 20)) ? 4 : 1;
    dim3 grid((unsigned)((total + 256 * vpt - 1) / (256 * vpt)));
    if (C == 8 && vpt == 4) MVS_LAUNCH((conv_cin1_kernel<8, 4>), grid, dim3(256), 0, st, gy, (const float*)ws, gx, p.B, p.Di, p.Hi, p.Wi, ep.bn_raw, ep.bn_stats, ep.slots, ep.nslots);
    else if (C == 8) MVS_LAUNCH((conv_cin1_kernel<8, 1>), grid, dim3(256), 0, st, gy, (const float*)ws, gx, p.B, p.Di, p.Hi, p.Wi, ep.bn_raw, ep.bn_stats, ep.slots, ep.nslots);
    else if (vpt == 4) MVS_LAUNCH((conv_cin1_kernel<16, 4>), grid, dim3(256), 0, st, gy, (const float*)ws, gx, p.B, p.Di, p.Hi, p.Wi, ep.bn_raw, ep.bn_stats, ep.slots, ep.nslots);
    else MVS_LAUNCH((conv_cin1_kernel<16, 1>), grid, dim3(256), 0, st, gy, (const float*)ws, gx, p.B, p.Di, p.Hi, p.Wi, ep.bn_raw, ep.bn_stats, ep.slots, ep.nslots);
    return mvs_check_launch("conv_cin1");
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
    int ntiles = B * a.ntd * a.nth * a.ntw;
    const int cc = CX % 16 == 0 ? 16 : 8;
    const int nbw = CG > 16 ? 2 : 1;
    // knobs "wgrad_groups" / "wgrad8_groups": persistent workgroups of a launch.  The weight gradients run on a side stream next
    // to the backward pass's critical path; fewer workgroups take longer but leave more of the chip to the main stream.
    const int maxg = g_conv_wgrad_groups < 1 ? 1 : (g_conv_wgrad_groups > WGRAD_MAX_GROUPS ? WGRAD_MAX_GROUPS : g_conv_wgrad_groups);
    int groups = ntiles < maxg ? ntiles : maxg;
    if (g_conv_c8 && geom == GEOM_S1 && CG == 8 && CX % 16 == 0) {
        const int max8 = g_conv_wgrad8_groups < 1 ? 1 : (g_conv_wgrad8_groups > 512 ? 512 : g_conv_wgrad8_groups);
        const int g8 = ntiles < max8 ? ntiles : max8;   // 60 KB LDS -> 2 resident workgroups per CU
        if (g_conv_wgrad8_gs && conv_c8_wgrad_gs_serves(geom, CX, CG)) {
            const int np = run_conv_c8_wgrad_gs(a, g8, g_conv_wgrad8_gs == 1 ? 8 : 16, st);
            if (np < 0) return np;
            return wgrad_finish(ws, np, CX, CG, gw, st);
        }
        if (CX % 32 == 0 && g_conv_wgrad8_nch == 2) {     // both 16-channel chunks of a 32-channel line in one workgroup (knob "wgrad8_nch")
            const int g2 = g8 > 256 ? 256 : g8;              // 110 KB of LDS: one workgroup per CU
            if (g_conv_c8 & 4) MVS_LAUNCH((conv_c8_wgrad_kernel<true, 2>), dim3(g2, CX / 32), dim3(256), 0, st, a);
            else MVS_LAUNCH((conv_c8_wgrad_kernel<false, 2>), dim3(g2, CX / 32), dim3(256), 0, st, a);
            int rc2 = mvs_check_launch("conv_c8_wgrad");
            if (rc2) return rc2;
            return wgrad_finish(ws, g2, CX, CG, gw, st);
        }
        if (g_conv_c8 & 4) MVS_LAUNCH((conv_c8_wgrad_kernel<true, 1>), dim3(g8, CX / 16), dim3(256), 0, st, a);
        else MVS_LAUNCH((conv_c8_wgrad_kernel<false, 1>), dim3(g8, CX / 16), dim3(256), 0, st, a);
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
    // knob "wgrad_pers": one-chunk layers with <= 16 gradient channels and many tiles through the persistent LDS-DMA kernel
    // (conv3d_pers.hip): the level-0 / level-1 layers conv1, conv11 (stride 2, 8 X channels) and conv2 (stride 1, 16)
    if (g_conv_wgrad_pers && conv_wgrad_pers_serves(geom, CX, CG) && ntiles >= g_conv_pers_min_wgs) {
        const int np = run_conv_wgrad_pers(geom, a, WGRAD_MAX_GROUPS, st);
        if (np < 0) return np;
        return wgrad_finish(ws, np, CX, CG, gw, st);
    }
    // knob "wgrad_small": quarter-size tiles (the *_SMALL geometries) for the generic kernel when the launch has many tiles anyway:
    // a stride-2 layer with 8 X channels (the L0 layers) holds a 5x9x33-voxel halo + the G tile = 87 KB of LDS per workgroup, ONE
    // workgroup per CU; at 3x9x33 it is 51 KB and three fit (latency bound: 126 MB + 31 MB read in 99 us)
    const bool small = g_conv_wgrad_small == 3 ||      // (3: always -- tests)
                       (g_conv_wgrad_small && ntiles >= 2 * WGRAD_MAX_GROUPS && (g_conv_wgrad_small == 2 || cc == 8 || geom == GEOM_S2));
    if (small) {
        const int kg = geom + GEOM_S1_SMALL;
        a.ntd = mvs_cdiv(a.QD, geom_tqd(kg)); a.nth = mvs_cdiv(a.QH, geom_tqh(kg));
        ntiles = B * a.ntd * a.nth * a.ntw;
        groups = ntiles < maxg ? ntiles : maxg;
        dim3 grids(groups, CX / cc, mvs_cdiv(CG, nbw * 16));
        if (geom == GEOM_S1) { if (cc == 16) launch_wgrad<GEOM_S1_SMALL, 16>(a, nbw, grids, st); else launch_wgrad<GEOM_S1_SMALL, 8>(a, nbw, grids, st); }
        else { if (cc == 16) launch_wgrad<GEOM_S2_SMALL, 16>(a, nbw, grids, st); else launch_wgrad<GEOM_S2_SMALL, 8>(a, nbw, grids, st); }
        int rcs = mvs_check_launch("conv_wgrad (small tiles)");
        if (rcs) return rcs;
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

// deterministic reduction of the per-workgroup partial images: > 32 images take the wide kernel (16 slices per output element)
// (one thread per element walking all images serially is latency bound)
static int wgrad_finish(float* ws, int nparts, int CX, int CG, float* gw, hipStream_t st) {
    const int n = 27 * CX * CG;
    if (nparts > 32)
        MVS_LAUNCH(conv_wgrad_reduce_wide_kernel, dim3(mvs_cdiv(n, 16)), dim3(256), 0, st, (const float*)ws, nparts, CX, CG, gw);
    else
        MVS_LAUNCH(conv_wgrad_reduce_kernel, dim3(mvs_cdiv(n, 256)), dim3(256), 0, st, (const float*)ws, nparts, CX, CG, gw);
    return mvs_check_launch("conv_wgrad_reduce");
}

// ---- C ABI ---------------------------------------------------------------------------------------
// (D,H,W) are always the spatial dims of the forward op's INPUT x.
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

// Write into ws (>= mvs_conv3d_workspace_bytes) what the kernels of `op` (a forward or input-gradient op) derive from the
// parameter tensor w; a later call of that op with the same shape and ws_packed = 1 skips its own packing launch.
extern "C" int mvs_conv3d_pack_weights(int op, const float* w, float* ws, int B, int D, int H, int W, int Cin, int Cout, int stride,
                                       hipStream_t stream) {
    MVS_REQUIRE(w && ws, MVS_ERR_NULL, "conv3d_pack_weights: null pointer argument");
    IgemmPlan p;
    int rc = plan_for(op, B, D, H, W, Cin, Cout, stride, p, "conv3d_pack_weights");
    if (rc) return rc;
    PackItem it;
    rc = plan_pack(p, w, ws, it);
    if (rc) return rc;
    return launch_pack(&it, 1, stream);
}

// The same for n ops in ONE launch: ops[n], w[n], ws[n], shapes[n][7] = (B, D, H, W, Cin, Cout, stride) of each op.
extern "C" int mvs_conv3d_pack_weights_batch(int n, const int* ops, const float* const* w, float* const* ws, const int* shapes,
                                             hipStream_t stream) {
    MVS_REQUIRE(ops && w && ws && shapes, MVS_ERR_NULL, "conv3d_pack_weights_batch: null pointer argument");
    MVS_REQUIRE(n >= 0 && n <= 1024, MVS_ERR_SHAPE, "conv3d_pack_weights_batch: bad count %d", n);
    PackItem items[64];
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int m = n - i0 < 64 ? n - i0 : 64;
        for (int i = 0; i < m; ++i) {
            const int* sh = shapes + (size_t)(i0 + i) * 7;
            MVS_REQUIRE(w[i0 + i] && ws[i0 + i], MVS_ERR_NULL, "conv3d_pack_weights_batch: null pointer in entry %d", i0 + i);
            IgemmPlan p;
            int rc = plan_for(ops[i0 + i], sh[0], sh[1], sh[2], sh[3], sh[4], sh[5], sh[6], p, "conv3d_pack_weights_batch");
            if (rc) return rc;
            rc = plan_pack(p, w[i0 + i], ws[i0 + i], items[i]);
            if (rc) return rc;
        }
        int rc = launch_pack(items, m, stream);
        if (rc) return rc;
    }
    return MVS_OK;
}

// y = conv3d(x, w[Cout][Cin][3][3][3], stride, pad 1) then optional epilogue:
//   scale&&shift: y*scale[c]+shift[c] ; only shift: y+shift[c] (bias) ; relu ; + skip ;
//   stat_slots [nslots][2][Cout] fp64: per-channel (sum, sum of squares) of the RAW y added into slot (workgroup mod nslots)
extern "C" int mvs_conv3d_fwd(const float* x, const float* w, float* y, float* ws, int B, int D, int H, int W,
                              int Cin, int Cout, int stride, const float* scale, const float* shift,
                              const float* skip, int relu, double* stat_slots, int nslots, int ws_packed, hipStream_t stream) {
    IgemmPlan p;
    int rc = plan_for(MVS_OP_CONV_FWD, B, D, H, W, Cin, Cout, stride, p, "conv3d_fwd");
    if (rc) return rc;
    Epilogue ep = {scale, shift, skip, relu, stat_slots, nslots, nullptr, nullptr};
    return run_igemm(p, x, w, y, ws, ep, ws_packed, stream);
}

// gx[B,D,H,W,Cin] = d conv3d / dx applied to gy[B,Do,Ho,Wo,Cout]  (+ add, like gx, or NULL: the second gradient contribution of a
// tensor with two consumers -- the U-Net's skip connections, mvsnet.py:70-72 -- summed in the epilogue).
// bn_raw (like gx, or NULL): x = relu(BatchNorm(bn_raw)) (+ skip) came out of a BatchNorm+ReLU block and gx is that block's COMPLETE
// output gradient; the epilogue then also adds the block's backward statistics (sum dyh, sum dyh*xhat per channel, dyh = gx where
// the ReLU was active) into bn_slots [nslots][2][Cin] (fp64), using bn_stats [4][Cin] = the block's mean, invstd, scale, shift
// (module.py:35-42 backward): no separate reduction pass over (gx, raw).
extern "C" int mvs_conv3d_dgrad(const float* gy, const float* w, const float* add, float* gx, float* ws, int B, int D, int H, int W,
                                int Cin, int Cout, int stride, const float* bn_raw, const float* bn_stats, double* bn_slots,
                                int nslots, int ws_packed, hipStream_t stream) {
    IgemmPlan p;
    int rc = plan_for(MVS_OP_CONV_DGRAD, B, D, H, W, Cin, Cout, stride, p, "conv3d_dgrad");
    if (rc) return rc;
    Epilogue ep = {nullptr, nullptr, add, 0, bn_slots, nslots, bn_raw, bn_stats};
    MVS_REQUIRE(!bn_slots || bn_raw, MVS_ERR_NULL, "conv3d_dgrad: bn_slots without bn_raw");
    if (p.cin1) return run_cin1(p, gy, w, gx, ws, ep, ws_packed, stream);
    return run_igemm(p, gy, w, gx, ws, ep, ws_packed, stream);
}

// gw[Cout][Cin][27] = d conv3d / dw
extern "C" int mvs_conv3d_wgrad(const float* x, const float* gy, float* gw, float* ws, int B, int D, int H, int W,
                                int Cin, int Cout, int stride, hipStream_t stream) {
    int rc = check_stride(stride, D, H, W, "conv3d_wgrad");
    if (rc) return rc;
    return run_wgrad(stride == 2 ? GEOM_S2 : GEOM_S1, x, gy, gw, ws, B, D, H, W, Cin, Cout, stream);
}

// y = conv_transpose3d(x, w[Cin][Cout][3][3][3], stride, pad 1, output_padding stride-1); epilogue like mvs_conv3d_fwd
extern "C" int mvs_convT3d_fwd(const float* x, const float* w, float* y, float* ws, int B, int D, int H, int W,
                               int Cin, int Cout, int stride, const float* scale, const float* shift,
                               const float* skip, int relu, double* stat_slots, int nslots, int ws_packed, hipStream_t stream) {
    IgemmPlan p;
    int rc = plan_for(MVS_OP_CONVT_FWD, B, D, H, W, Cin, Cout, stride, p, "convT3d_fwd");
    if (rc) return rc;
    Epilogue ep = {scale, shift, skip, relu, stat_slots, nslots, nullptr, nullptr};
    return run_igemm(p, x, w, y, ws, ep, ws_packed, stream);
}

// gx[B,D,H,W,Cin] from gy[B,sD,sH,sW,Cout]; add / bn_raw / bn_stats / bn_slots as in mvs_conv3d_dgrad
extern "C" int mvs_convT3d_dgrad(const float* gy, const float* w, const float* add, float* gx, float* ws, int B, int D, int H,
                                 int W, int Cin, int Cout, int stride, const float* bn_raw, const float* bn_stats, double* bn_slots,
                                 int nslots, int ws_packed, hipStream_t stream) {
    IgemmPlan p;
    int rc = plan_for(MVS_OP_CONVT_DGRAD, B, D, H, W, Cin, Cout, stride, p, "convT3d_dgrad");
    if (rc) return rc;
    MVS_REQUIRE(!bn_slots || bn_raw, MVS_ERR_NULL, "convT3d_dgrad: bn_slots without bn_raw");
    Epilogue ep = {nullptr, nullptr, add, 0, bn_slots, nslots, bn_raw, bn_stats};
    return run_igemm(p, gy, w, gx, ws, ep, ws_packed, stream);
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
