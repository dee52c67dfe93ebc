// Narrow level-0 layers of the regulariser without LDS staging.
//
// conv1 (8 -> 16, stride 2; jdacs/models/mvsnet.py:41) reads the 126 MB level-0 activation once and does 3.4 GFLOP: 20 us of HBM
// time, 22 us of MFMA time -- and took 80 us in the LDS-staged implicit-GEMM kernel (conv3d.hip: load halo -> barrier -> 56
// MFMAs -> store, three workgroups per CU at 43 KB of LDS each: the loads of a workgroup are in flight only while it is not
// computing, ~40 KB per CU on average where 8 TB/s x ~2 us of loaded latency need ~64 KB per CU all the time).
// Here the MFMA A fragments come straight from global memory: with 8 input channels a k-step of 16 is two taps x 8 channels,
// i.e. lane (position l15, k-quarter g) loads ONE float4 = channels 4(g&1).. of tap 2*ks + (g>>1) of its position.  No LDS, no
// barrier in the loop, ~100 registers -> 4-5 waves per SIMD with every wave's next k-steps in flight; the 3.4x re-use of an
// input voxel by neighbouring taps / rows is served by L1 / L2 (430 MB through the texture path for 126 MB of HBM reads).
// Same packed weight image, same MFMA order over k as conv_igemm_kernel<GEOM_S2, 8, 1, *> => bit-identical results.
#include "mvs_rt.h"
#include "conv_map.h"
#include "conv_args.h"

// workgroup tile: 2 x 8 x 16 coarse positions; wave w owns depth slice (w >> 1) and the 4 rows 4*(w & 1) .. +3 (MB = 4 m-blocks)
#define DIR_TQD 2
#define DIR_TQH 8
#define DIR_MB 4

struct DirCheck { static constexpr bool value = true; };
struct DirNoCheck { static constexpr bool value = false; };

template <bool CHECK>
__device__ __forceinline__ float4 dir_load(const float* __restrict__ p, bool ok) {
    if (CHECK) return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(p);
}

__global__ __launch_bounds__(256) void conv_s2c8_direct_kernel(ConvArgs a) {
    constexpr int MB = DIR_MB, KS = 14;           // ksteps_for(27, 8)
    __shared__ float red[4 * 16 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int par = g >> 1, ci4 = 4 * (g & 1);

    // tile order: every XCD (workgroups are dealt to the 8 XCDs round-robin) walks ONE contiguous range of bricks of (all W tiles) x
    // (4 H tiles) marching along D, so the input rows / slices that neighbouring tiles share are found in that XCD's own L2
    int b, td, th, tw;
    if (a.relu & 2) brick_tile(xcd_block(blockIdx.x, gridDim.x), a.ntw, a.nth, a.ntd, b, td, th, tw);
    else linear_tile(blockIdx.x, a.ntw, a.nth, a.ntd, b, td, th, tw);
    const int qd = td * DIR_TQD + (wave >> 1);
    const int qh0 = th * DIR_TQH + 4 * (wave & 1);
    const int qw = tw * 16 + l15;

    // element offset (within the batch item) of tap (0,0,0) of the lane's position in row mb, + the lane's channel quarter
    const float* __restrict__ xb = a.x + (size_t)b * a.Di * a.Hi * a.Wi * 8;
    int base[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
        base[mb] = (((2 * qd - 1) * a.Hi + (2 * (qh0 + mb) - 1)) * a.Wi + (2 * qw - 1)) * 8 + ci4;
    // the lane's tap of k-step ks is 2*ks + par: its element offset and (border tiles) its validity per row
    const int sH = a.Wi * 8, sD = a.Hi * a.Wi * 8;
    // wave-uniform: is every tap of every position of this wave's rows inside the volume (and every position inside the grid)?
    const int qw_lo = tw * 16, qw_hi = tw * 16 + 15;
    const bool interior = qd >= 1 && 2 * qd + 1 < a.Di && qd < a.QD && qh0 >= 1 && 2 * (qh0 + MB - 1) + 1 < a.Hi && qh0 + MB - 1 < a.QH &&
                          qw_lo >= 1 && 2 * qw_hi + 1 < a.Wi && qw_hi < a.QW;
    unsigned vm[MB];          // bit ks: the lane's tap of k-step ks is inside the volume (row mb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) vm[mb] = 0x3fffu;
    if (!interior) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            unsigned m = 0;
            const int qh = qh0 + mb;
            const bool pos_ok = qd < a.QD && qh < a.QH && qw < a.QW;
            for (int ks = 0; ks < KS; ++ks) {
                const int tap = 2 * ks + par;
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const int id = 2 * qd - 1 + kd, ih = 2 * qh - 1 + kh, iw = 2 * qw - 1 + kw;
                if (pos_ok && tap < 27 && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi) m |= 1u << ks;
            }
            vm[mb] = m;
        }
    }

    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto toff = [&](int ks) {      // element offset of the lane's tap of k-step ks (ks is a constant after unrolling)
        const int t0 = 2 * ks, t1 = 2 * ks + 1 < 27 ? 2 * ks + 1 : 26;      // (tap 27 does not exist: the lane re-reads tap 26 and is zeroed)
        const int o0 = (t0 / 9) * sD + ((t0 / 3) % 3) * sH + (t0 % 3) * 8;
        const int o1 = (t1 / 9) * sD + ((t1 / 3) % 3) * sH + (t1 % 3) * 8;
        return par ? o1 : o0;
    };
    auto run = [&](auto check_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        // ring of PF + 1 k-steps: the loads of k-step ks + PF are ISSUED before the MFMAs of k-step ks (the scheduler fences keep
        // hipcc from sinking them back next to their use, which it does otherwise: load -> wait -> MFMA per k-step)
        constexpr int PF = 2;
        float4 af[PF + 1][MB], bq[PF + 1];
        auto load = [&](int ks, float4 (&av)[MB], float4& bv) {
            const int o = toff(ks);
            bv = *reinterpret_cast<const float4*>(a.wp + ((size_t)ks * a.nb_total * 64 + lane) * 4);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bool ok = (vm[mb] >> ks) & 1u;
                av[mb] = dir_load<CHECK>(xb + base[mb] + o, ok);
                if (!CHECK && ks == KS - 1 && par) av[mb] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
#pragma unroll
        for (int u = 0; u < PF; ++u) load(u, af[u], bq[u]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PF < KS) load(ks + PF, af[(ks + PF) % (PF + 1)], bq[(ks + PF) % (PF + 1)]);
            MVS_SCHED_FENCE();
            const float4 bv = bq[ks % (PF + 1)];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const float4 av = af[ks % (PF + 1)][mb];
                acc[mb] = MVS_MFMA_16x16x4(av.x, bv.x, acc[mb]);
                acc[mb] = MVS_MFMA_16x16x4(av.y, bv.y, acc[mb]);
                acc[mb] = MVS_MFMA_16x16x4(av.z, bv.z, acc[mb]);
                acc[mb] = MVS_MFMA_16x16x4(av.w, bv.w, acc[mb]);
            }
            MVS_SCHED_FENCE();
        }
    };
    if (interior) run(DirNoCheck{});
    else run(DirCheck{});

    // epilogue: D layout col = lane&15 (co), row = 4*(lane>>4)+r (position along qw); raw store + BatchNorm statistic slots
    float st1 = 0.f, st2 = 0.f;
    const int co = l15;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int qh = qh0 + mb;
        if (qd >= a.QD || qh >= a.QH || co >= a.Cout) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qwr = tw * 16 + 4 * g + r;
            if (qwr >= a.QW) continue;
            const float v = acc[mb][r];
            st1 += v;
            st2 += v * v;
            a.y[((((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qwr) * a.Cout + co] = v;
        }
    }
    if (a.slots) {
        st1 += __shfl_xor(st1, 16); st1 += __shfl_xor(st1, 32);
        st2 += __shfl_xor(st2, 16); st2 += __shfl_xor(st2, 32);
        if (lane < 16) {
            red[(wave * 16 + lane) * 2 + 0] = st1;
            red[(wave * 16 + lane) * 2 + 1] = st2;
        }
        __syncthreads();
        if (tid < 32) {
            const int stat = tid / 16, n = tid % 16;
            if (n < a.Cout) {
                float s = 0.f;
                for (int w = 0; w < 4; ++w) s += red[(w * 16 + n) * 2 + stat];
                MVS_GLOBAL_ATOMIC_ADD_F64(a.slots + ((size_t)(blockIdx.x & (a.nslots - 1)) * 2 + stat) * a.Cout + n, (double)s);
            }
        }
    }
}

// a: as run_igemm fills it for GEOM_S2 (coarse grid = output grid), a.wp = the packed image of conv_pack_weights_item(GEOM_S2, CC 8)
extern int g_conv_direct;
int run_s2c8_direct(ConvArgs a, hipStream_t st) {
    a.relu = (g_conv_direct & 2) ? 2 : 0;     // (the layer has no ReLU epilogue: the field carries the tile-order choice, knob bit 2)
    a.ntd = mvs_cdiv(a.QD, DIR_TQD); a.nth = mvs_cdiv(a.QH, DIR_TQH); a.ntw = mvs_cdiv(a.QW, 16);
    MVS_LAUNCH(conv_s2c8_direct_kernel, dim3(a.B * a.ntd * a.nth * a.ntw), dim3(256), 0, st, a);
    return mvs_check_launch("conv_s2c8_direct");
}
