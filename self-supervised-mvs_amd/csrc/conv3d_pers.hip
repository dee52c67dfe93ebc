// Persistent implicit-GEMM convolution with double-buffered LDS-DMA staging (gfx950 global_load_lds_dwordx4).
//
// conv_igemm_kernel (conv3d.hip) is one tile per workgroup: load halo -> barrier -> k-loop -> store.  Its loads are in flight only
// while the workgroup is not computing, every wave pays prologue / epilogue / tail for 100-450 MFMAs, and its weight fragments come
// from L2 inside the k-loop.  Measured where that hurts most (profiles/r05_run5..9: the narrow level-0 layers run at 22-26 % of the
// HBM peak AND 30 % of the MFMA rate; a direct-from-global variant removed the barrier and landed at the same time -- memory side
// alone 72 us, MFMA side alone 55 us for 26 us of MFMA work).  This kernel is the other decomposition:
//   * persistent workgroups (one or two per CU) walk tiles blockIdx.x, blockIdx.x + gridDim.x, ...;
//   * the halo tile of the NEXT tile is requested by LDS-DMA (no registers, no ds_write pass) into the other half of a
//     double buffer when the current tile's k-loop starts, and is waited for (s_waitcnt vmcnt(0), by then long landed) after it;
//   * the layer's whole packed weight image lives in LDS for the lifetime of the workgroup -- no global load in the k-loop at all,
//     so nothing of hipcc's own s_waitcnt bookkeeping ever sits between an LDS-DMA request and its completion;
//   * one barrier per tile, without the release fence of __syncthreads() (the epilogue's stores stay in flight across it);
//   * BatchNorm statistics are accumulated over all tiles of a workgroup and added to the slots once.
// LDS layout of a halo tile: voxel-major, CC floats per voxel, NO padding (an LDS-DMA instruction writes 64 lanes x 16 bytes
// linearly); ds_read_b128 of 16 positions at a 32- / 64-byte pitch is conflict-free / two-way (the hardware's b128 lane groups are
// {0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md), 36 LDS cycles per wave and k-step against 512 MFMA cycles.
// Halo voxels outside the volume are DMA'd from a zero page.  Same packed weight image, same MFMA order over k as
// conv_igemm_kernel => bit-identical outputs (the statistics are summed in another order).
// Serves: one channel chunk (Cin == CC in {8, 16}), stride 1 or 2, up to two 16-wide Cout tiles per workgroup.
#include "mvs_rt.h"
#include "conv_map.h"
#include "conv_args.h"

__device__ __attribute__((aligned(16))) float g_conv_zero_page[64];

__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// NW: waves per workgroup (4 or 8).  The LDS footprint (two halo buffers + the weight image) allows one or two workgroups per
// CU; with four waves each that is one wave per SIMD, and a lone wave exposes every MFMA dependency and LDS latency of its own
// instruction stream (measured: 16 -> 16 at level 1 0.094 -> 0.134 ms).  Eight waves share the SAME tile (two rows of the 4 x 4 x 16
// block each instead of four): twice the waves per SIMD for the same LDS.
template <int GEOM, int CC, int NB, int SIDE, int NW>
__global__ __launch_bounds__(NW * 64) void conv_pers_kernel(ConvArgs a) {
    using G = ConvGeom<GEOM>;
    constexpr int NT = NW * 64;
    constexpr int CQ = CC / 4, MB = G::MB * 4 / NW;
    static_assert(G::MB * 4 % NW == 0 && MB >= 1, "rows of the tile must split evenly over the waves");
    constexpr int NR = G::RD * G::RH * G::RW;
    constexpr int NITEMS = NR * CQ;              // 16-byte items of a halo tile
    constexpr int NDMA = (NITEMS + 63) / 64;     // wave-level DMA instructions per tile
    constexpr int DPW = (NDMA + NW - 1) / NW;    // ... per wave
    constexpr int TILEF = NDMA * 256;            // floats per buffer (whole DMA instructions)
    constexpr int KS = G::PW ? 18 * CC / 16 : (27 * CC + 15) / 16;      // k-steps of the whole weight image (conv_map.h: total_ksteps)
    __shared__ __attribute__((aligned(16))) float tile[2 * TILEF];
    __shared__ __attribute__((aligned(16))) float wl[KS * NB * 256];
    __shared__ float red[NW * NB * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int nb0 = blockIdx.y * NB;
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;

    // ---- the layer's weight image -> LDS (once) ----
    for (int i = tid; i < KS * NB * 64; i += NT) {
        const int l = i & 63, nb = (i >> 6) % NB, ks = (i >> 6) / NB;
        *reinterpret_cast<float4*>(&wl[(size_t)i * 4]) =
            *reinterpret_cast<const float4*>(a.wp + (((size_t)ks * a.nb_total + nb0 + nb) * 64 + l) * 4);
    }

    // ---- per-lane DMA items: tile-invariant source offsets + halo coordinates ----
    int rel[DPW], crd[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int d = NW * j + wave;
        const int i = 64 * d + lane;
        const int vox = i / CQ, cq = i % CQ;
        const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
        rel[j] = ((rd * a.Hi + rh) * a.Wi + rw) * CC + 4 * cq;
        crd[j] = (d < NDMA && i < NITEMS) ? (rd | (rh << 8) | (rw << 16)) : -1;
    }
    const float* __restrict__ zero = g_conv_zero_page;
    auto issue = [&](int t, int buf) {
        int b, td, th, tw;
        linear_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        const int id0 = td * G::TQD * G::IS - G::PAD, ih0 = th * G::TQH * G::IS - G::PAD, iw0 = tw * G::TQW * G::IS - G::PAD;
        const long long org = ((((long long)b * a.Di + id0) * a.Hi + ih0) * a.Wi + iw0) * CC;
        const float* __restrict__ xb = a.x + org;    // (may lie in front of the tensor for border tiles: only in-volume items use it)
        const bool interior = id0 >= 0 && id0 + G::RD <= a.Di && ih0 >= 0 && ih0 + G::RH <= a.Hi && iw0 >= 0 && iw0 + G::RW <= a.Wi;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int d = NW * j + wave;
            if (d >= NDMA) break;                    // wave-uniform
            const float* src = xb + rel[j];
            bool ok = crd[j] >= 0;
            if (!interior) {
                const int rd = crd[j] & 255, rh = (crd[j] >> 8) & 255, rw = (crd[j] >> 16) & 255;
                ok = ok && id0 + rd >= 0 && id0 + rd < a.Di && ih0 + rh >= 0 && ih0 + rh < a.Hi && iw0 + rw >= 0 && iw0 + rw < a.Wi;
            }
            if (!ok) src = zero + 4 * (lane & 15);
            MVS_DMA16(&tile[buf * TILEF + d * 256], src);
        }
    };

    // ---- per-lane A offsets (floats, inside a buffer) of the lane's position in each m-block, at tap (0,0,0) ----
    int aoff[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int f = wave * MB + mb;
        const int qd_l = f / G::TQH, qh_l = f % G::TQH;
        const int kq = CC == 16 ? 4 * g : 4 * (g & 1);
        aoff[mb] = (((qd_l * G::IS) * G::RH + qh_l * G::IS) * G::RW + l15 * G::IS) * CC + kq;
    }
    const int tsel = CC == 8 ? (g >> 1) : 0;     // CC == 8: a k-step of 16 is two taps; lanes g = 2, 3 take the second

    // BatchNorm statistics (and, with bn_raw, the normalisation constants) of the lane's four channels 4g .. 4g+3 of each Cout tile
    float st1[NB][4], st2[NB][4], bmu[NB][4], bis[NB][4], bsc[NB][4], bsh[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st1[nb][r] = st2[nb][r] = 0.f;
            bmu[nb][r] = bis[nb][r] = bsc[nb][r] = bsh[nb][r] = 0.f;
            if (SIDE && a.bn_raw) {
                const int co = (G::PW ? 4 * (g & 1) : (nb0 + nb) * 16 + 4 * g) + r;
                if (co < a.Cout) { bmu[nb][r] = a.bn_stats[co]; bis[nb][r] = a.bn_stats[a.Cout + co]; bsc[nb][r] = a.bn_stats[2 * a.Cout + co]; bsh[nb][r] = a.bn_stats[3 * a.Cout + co]; }
            }
        }

    if ((int)blockIdx.x < ntiles) issue(blockIdx.x, 0);
    MVS_WAIT_VMCNT(0);
    __syncthreads();             // weights + first tile in LDS

    int it = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
        const int buf = it & 1;
        if (t + (int)gridDim.x < ntiles) issue(t + gridDim.x, buf ^ 1);     // lands under this tile's MFMAs
        const int tb = buf * TILEF;        // (indexing the __shared__ arrays themselves keeps the accesses ds_read_b128, not flat loads)

        int b, td, th, tw;
        linear_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        const int qd0 = td * G::TQD, qh0 = th * G::TQH, qw0 = tw * G::TQW;
        // GEOM_TR2_PW: four (pd, ph) parity classes, each its own k-loop (2 / 4 / 4 / 8 taps) and epilogue; the 8-tap class runs
        // FIRST so that the next tile's DMA has landed when the first epilogue's loads queue up behind it
#pragma unroll
        for (int cix = 0; cix < G::NCLS; ++cix) {
            const int cls = G::PW ? G::NCLS - 1 - cix : 0;
            const int KSC = G::PW ? tr2p_ntaps(cls) * CC / 16 : KS;          // k-steps of this class
            const int kk0 = G::PW ? tr2p_tap_prefix(cls) * CC / 16 : 0;      // ... and where they start in the weight image
            f32x4 acc[MB][NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // operands of k-step ks + 1 are requested from LDS before the MFMAs of k-step ks; the MFMAs of a k-step walk the accumulators
            // round-robin (x of every (mb, nb), then y, ...), so back-to-back MFMAs never wait for each other's result
            auto toff_of = [&](int ks) {      // the lane's tap of k-step ks -> float offset inside the halo tile (ks: constant after unrolling)
                if (G::PW) {                  // CC == 16: one tap per k-step
                    int dd, dh, dw, kd, kh;
                    tr2p_tap(cls, ks, dd, dh, dw, kd, kh);
                    return ((dd * G::RH + dh) * G::RW + dw) * CC;
                }
                if (CC == 16) return (((ks / 9) * G::RH + (ks / 3) % 3) * G::RW + ks % 3) * CC;
                const int t0 = 2 * ks, t1 = 2 * ks + 1 < 27 ? 2 * ks + 1 : 26;      // (tap 27: its weights are zero; the lane re-reads tap 26)
                const int o0 = (((t0 / 9) * G::RH + (t0 / 3) % 3) * G::RW + t0 % 3) * CC;
                const int o1 = (((t1 / 9) * G::RH + (t1 / 3) % 3) * G::RW + t1 % 3) * CC;
                return tsel ? o1 : o0;
            };
            float4 bq[2][NB], af[2][MB];
            auto fetch = [&](int ks, float4 (&bv)[NB], float4 (&av)[MB]) {
                const int toff = toff_of(ks);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[nb] = *reinterpret_cast<const float4*>(&wl[(((kk0 + ks) * NB + nb) * 64 + lane) * 4]);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[mb] = *reinterpret_cast<const float4*>(&tile[tb + aoff[mb] + toff]);
            };
            fetch(0, bq[0], af[0]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks >= KSC) break;         // (compile-time after unrolling)
                if (ks + 1 < KSC) fetch(ks + 1, bq[(ks + 1) & 1], af[(ks + 1) & 1]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = MVS_MFMA_16x16x4(f4c(bq[ks & 1][nb], c), f4c(af[ks & 1][mb], c), acc[mb][nb]);   // D^T: rows = channels
            }
            // the next tile's DMA (requested a whole k-loop ago) has landed; so have the previous epilogue's stores
            if (cix == 0) MVS_WAIT_VMCNT(0);

            // ---- epilogue of the class.  The WEIGHT fragment is the MFMA's A operand (the packed image serves as either operand:
            // lane (l15, g) holds W[k = 4g..][co = l15]), so D = (W^T X^T): row = 4*(lane>>4)+r -> output channel, col = lane&15 ->
            // position along qw.  A lane ends with FOUR CONSECUTIVE CHANNELS of one voxel: the side inputs are read and the result is
            // written as float4, and a wave instruction covers 16 voxels x 64 bytes = 1 KiB of contiguous memory (the one-tile kernel's
            // D has the positions in the lane's registers: 4-byte accesses, four instructions for the same bytes).
            // PW: row n = pw*8 + co -> lanes g = 0, 1 are the two channel quads of output voxel 2*qw, g = 2, 3 of voxel 2*qw + 1 ----
            const int pd = G::PW ? (cls >> 1) & 1 : 0, ph = G::PW ? cls & 1 : 0, pw = G::PW ? (g >> 1) : 0;
            const int qw = qw0 + l15;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int f = wave * MB + mb;
                const int qd = qd0 + f / G::TQH, qh = qh0 + f % G::TQH;
                const int od = qd * G::OS + pd, oh = qh * G::OS + ph;
                const size_t obase = ((((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + qw * G::OS + pw) * a.Cout;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int co0 = G::PW ? 4 * (g & 1) : (nb0 + nb) * 16 + 4 * g;
                    if (!(qd < a.QD && qh < a.QH && qw < a.QW && co0 < a.Cout)) continue;
                    float4 sk = make_float4(0.f, 0.f, 0.f, 0.f), rw = sk;
                    if (SIDE && a.skip) sk = *reinterpret_cast<const float4*>(a.skip + obase + co0);
                    if (SIDE && a.bn_raw) rw = *reinterpret_cast<const float4*>(a.bn_raw + obase + co0);
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[mb][nb][r];
                        float sv1 = v, sv2 = v * v;
                        if (a.scale) v = v * a.scale[co0 + r] + a.shift[co0 + r];
                        else if (a.shift) v = v + a.shift[co0 + r];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (SIDE && a.skip) v += f4c(sk, r);
                        if (SIDE && a.bn_raw) {
                            sv1 = (f4c(rw, r) * bsc[nb][r] + bsh[nb][r] > 0.f) ? v : 0.f;
                            sv2 = sv1 * ((f4c(rw, r) - bmu[nb][r]) * bis[nb][r]);
                        }
                        st1[nb][r] += sv1;
                        st2[nb][r] += sv2;
                        o[r] = v;
                    }
                    *reinterpret_cast<float4*>(a.y + obase + co0) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        // every wave has read its A fragments of `buf` (they fed MFMAs that have been issued) and has waited for its own share of the
        // next tile's DMA: after the barrier `buf` may be overwritten and `buf ^ 1` read
        MVS_LDS_BARRIER();
    }

    if (a.slots) {
        // the 16 lanes of a group hold 16 positions of the same four channels: sum over them, then over the waves
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = st1[nb][r], s2 = st2[nb][r];
                s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4); s1 += __shfl_xor(s1, 8);
                s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4); s2 += __shfl_xor(s2, 8);
                if (G::PW) { s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); }   // rows n and n ^ 8 are the same channel (pw = 0 / 1)
                if (l15 == 0 && (!G::PW || g < 2)) {
                    red[((wave * NB + nb) * 16 + 4 * g + r) * 2 + 0] = s1;
                    red[((wave * NB + nb) * 16 + 4 * g + r) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        if (tid < 2 * NB * 16) {
            const int stat = tid / (NB * 16), n = tid % (NB * 16);
            if (nb0 * 16 + n < a.Cout) {
                float s = 0.f;
                for (int w = 0; w < NW; ++w) s += red[(w * NB * 16 + n) * 2 + stat];
                MVS_GLOBAL_ATOMIC_ADD_F64(a.slots + ((size_t)(blockIdx.x & (a.nslots - 1)) * 2 + stat) * a.Cout + nb0 * 16 + n, (double)s);
            }
        }
    }
}

int g_conv_pers_groups = 0;
int g_conv_pers_nw = 8;
// LDS bytes of an instantiation (host side: how many workgroups fit a CU)
template <int GEOM, int CC, int NB>
static int pers_lds_bytes() {
    using G = ConvGeom<GEOM>;
    constexpr int NDMA = (G::RD * G::RH * G::RW * (CC / 4) + 63) / 64;
    return (2 * NDMA * 256 + (G::PW ? 18 * CC / 16 : (27 * CC + 15) / 16) * NB * 256 + 8 * NB * 16 * 2) * 4;
}

template <int GEOM, int CC, int NB, int NW>
static int launch_pers(const ConvArgs& a, hipStream_t st) {
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    const int per_cu = (160 * 1024) / pers_lds_bytes<GEOM, CC, NB>() >= 2 ? 2 : 1;
    int groups = g_conv_pers_groups > 0 ? g_conv_pers_groups : 256 * per_cu;     // knob "conv_pers_groups" (tests: a few workgroups walk many tiles)
    if (groups > ntiles) groups = ntiles;
    dim3 grid(groups, a.nb_total / NB);
    if (a.skip || a.bn_raw) MVS_LAUNCH((conv_pers_kernel<GEOM, CC, NB, 1, NW>), grid, dim3(NW * 64), 0, st, a);
    else MVS_LAUNCH((conv_pers_kernel<GEOM, CC, NB, 0, NW>), grid, dim3(NW * 64), 0, st, a);
    return mvs_check_launch("conv_pers");
}

// Does the persistent kernel serve this op?  (one channel chunk; a.* filled as run_igemm does for the full-size tiles of `geom`)
bool conv_pers_serves(int geom, int cin, int cout) {
    if (cout % 4) return false;           // (a lane writes four consecutive channels)
    if (geom == GEOM_S1) return (cin == 16 && cout <= 16) || (cin == 8 && cout > 16 && cout <= 32);
    if (geom == GEOM_S2) return cin == 8 && cout <= 16;
    if (geom == GEOM_TR2_PW) return cin == 16 && cout == 8;      // (the caller has packed the W-parity-merged image)
    return false;
}

int run_conv_pers(int geom, const ConvArgs& a, hipStream_t st) {
    // knob "conv_pers_nw": waves per workgroup
    if (g_conv_pers_nw == 8) {
        if (geom == GEOM_S1 && a.Cin == 16) return launch_pers<GEOM_S1, 16, 1, 8>(a, st);
        if (geom == GEOM_S1 && a.Cin == 8) return launch_pers<GEOM_S1, 8, 2, 8>(a, st);
        if (geom == GEOM_S2 && a.Cin == 8) return launch_pers<GEOM_S2, 8, 1, 8>(a, st);
    }
    if (geom == GEOM_TR2_PW && a.Cin == 16) return launch_pers<GEOM_TR2_PW, 16, 1, 8>(a, st);
    if (geom == GEOM_S1 && a.Cin == 16) return launch_pers<GEOM_S1, 16, 1, 4>(a, st);
    if (geom == GEOM_S1 && a.Cin == 8) return launch_pers<GEOM_S1, 8, 2, 4>(a, st);
    if (geom == GEOM_S2 && a.Cin == 8) return launch_pers<GEOM_S2, 8, 1, 4>(a, st);
    mvs_set_error("conv_pers: geometry %d with %d input channels is not served", geom, a.Cin);
    return MVS_ERR_UNSUPPORTED;
}

// ================================================================================================
// Weight gradient, same staging:  dW[tap][ci][co] = sum_{b,o} X[b, o*S + tap - 1][ci] * G[b,o][co]
// GEMM view as in conv_wgrad_kernel (conv3d.hip): M = ci (CC = 16) or (tap pair, ci) (CC = 8), N = co (<= 16), K = positions.
// conv_wgrad_kernel already walks tiles with persistent workgroups, but stages through registers (24-44 of them held across the
// MFMA loop) into PADDED LDS tiles of 71-87 KB: ONE workgroup of four waves per CU, one wave per SIMD, nothing to cover its
// ds_write pass and two barriers per tile (the level-0 layers: 0.108 ms for 22 us of MFMA work and 20 us of HBM time).  Here:
//   * the X halo AND the G tile of the next tile arrive by LDS-DMA into the other half of a double buffer (unpadded: the A reads
//     are ds_read_b32 of 16 consecutive floats per lane group -- a 16-float position pitch keeps the four groups of a wave on
//     64 consecutive floats);
//   * eight waves share a tile: waves 0-3 take the even k-steps (four positions each), waves 4-7 the odd ones, every wave
//     with the full set of tap slots of its rank (7 or 4 MFMAs per k-step on independent accumulators); the two halves meet in LDS
//     once, at the end of the kernel;
//   * one fence-free barrier per tile.
// Each workgroup writes ONE partial image; wgrad_finish (conv3d.hip) sums them in a fixed order.  The products are the same as
// conv_wgrad_kernel's, the order of the K sum differs (another split of the positions), so results agree to fp32 rounding.
// ================================================================================================
template <int GEOM, int CC>
__global__ __launch_bounds__(512) void conv_wgrad_pers_kernel(WgradArgs a) {
    using G = ConvGeom<GEOM>;
    constexpr int NW = 8;
    constexpr int CQ = CC / 4;
    constexpr int NR = G::RD * G::RH * G::RW;
    constexpr int NPOS = G::TQD * G::TQH * G::TQW;
    constexpr int XITEMS = NR * CQ, GITEMS = NPOS * 4;                     // 16-byte items: X halo, G tile (16 floats per position)
    constexpr int XDMA = (XITEMS + 63) / 64, GDMA = GITEMS / 64;           // wave-level DMA instructions
    constexpr int NDMA = XDMA + GDMA;
    constexpr int DPW = (NDMA + NW - 1) / NW;
    constexpr int XF = XDMA * 256, BUFF = XF + GITEMS * 4;                 // floats: X part, whole buffer
    constexpr int NSLOT = CC == 16 ? 27 : 14;                              // CC == 8: a slot is a pair of taps (2s, 2s+1)
    constexpr int SPW = (NSLOT + 3) / 4;                                   // slots per wave (waves w and w + 4 share a slot set)
    constexpr int NKS = NPOS / 4;                                          // k-steps of a tile (4 positions each)
    static_assert(G::TQW == 16 && NKS % 2 == 0, "a k-step is four consecutive positions of one row");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUFF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int w4 = wave & 3, half = wave >> 2;
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;

    // ---- per-lane DMA items ----
    int rel[DPW], crd[DPW];      // crd: X items (rd | rh << 8 | rw << 16), G items (pd | ph << 8 | pw << 16 | 1 << 30), -1: none
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int d = NW * j + wave;
        rel[j] = 0; crd[j] = -1;
        if (d < XDMA) {
            const int i = 64 * d + lane, vox = i / CQ, cq = i % CQ;
            const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
            rel[j] = ((rd * a.Hi + rh) * a.Wi + rw) * CC + 4 * cq;
            if (i < XITEMS) crd[j] = rd | (rh << 8) | (rw << 16);
        } else if (d < NDMA) {
            const int i = 64 * (d - XDMA) + lane, p = i >> 2, n4 = i & 3;
            const int pw = p % G::TQW, ph = (p / G::TQW) % G::TQH, pd = p / (G::TQW * G::TQH);
            rel[j] = ((pd * a.QH + ph) * a.QW + pw) * a.CG + 4 * n4;
            if (4 * n4 < a.CG) crd[j] = pd | (ph << 8) | (pw << 16) | (1 << 30);
        }
    }
    const float* __restrict__ zero = g_conv_zero_page;
    auto issue = [&](int t, int buf) {
        int b, td, th, tw;
        linear_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        const int qd0 = td * G::TQD, qh0 = th * G::TQH, qw0 = tw * G::TQW;
        const int id0 = qd0 * G::IS - 1, ih0 = qh0 * G::IS - 1, iw0 = qw0 * G::IS - 1;
        const float* __restrict__ xb = a.x + ((((long long)b * a.Di + id0) * a.Hi + ih0) * a.Wi + iw0) * CC;
        const float* __restrict__ gb = a.g + ((((long long)b * a.QD + qd0) * a.QH + qh0) * a.QW + qw0) * a.CG;
        const bool interior = id0 >= 0 && id0 + G::RD <= a.Di && ih0 >= 0 && ih0 + G::RH <= a.Hi && iw0 >= 0 && iw0 + G::RW <= a.Wi &&
                              qd0 + G::TQD <= a.QD && qh0 + G::TQH <= a.QH && qw0 + G::TQW <= a.QW;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int d = NW * j + wave;
            if (d >= NDMA) break;                    // wave-uniform
            const bool is_g = d >= XDMA;             // wave-uniform
            const float* src = (is_g ? gb : xb) + rel[j];
            bool ok = crd[j] >= 0;
            if (!interior) {
                const int c0 = crd[j] & 255, c1 = (crd[j] >> 8) & 255, c2 = (crd[j] >> 16) & 255;
                if (is_g) ok = ok && qd0 + c0 < a.QD && qh0 + c1 < a.QH && qw0 + c2 < a.QW;
                else ok = ok && id0 + c0 >= 0 && id0 + c0 < a.Di && ih0 + c1 >= 0 && ih0 + c1 < a.Hi && iw0 + c2 >= 0 && iw0 + c2 < a.Wi;
            }
            if (!ok) src = zero + 4 * (lane & 15);
            MVS_DMA16(&lds[buf * BUFF + (is_g ? XF + (d - XDMA) * 256 : d * 256)], src);
        }
    };

    // ---- per-lane A offsets of the wave's tap slots (floats inside the X part), incl. the lane group's position g of a k-step ----
    int toff[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int slot = w4 + 4 * s;
        int tap = CC == 16 ? slot : 2 * slot + (l15 >> 3);
        if (slot >= NSLOT || tap > 26) tap = 0;      // a slot past the end computes unused values from tap 0 (no branch in the loop)
        const int ci = CC == 16 ? l15 : (l15 & 7);
        toff[s] = (((tap / 9) * G::RH + (tap / 3) % 3) * G::RW + tap % 3 + g * G::IS) * CC + ci;
    }
    f32x4 acc[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if ((int)blockIdx.x < ntiles) issue(blockIdx.x, 0);
    MVS_WAIT_VMCNT(0);
    __syncthreads();

    int it = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
        const int buf = it & 1;
        if (t + (int)gridDim.x < ntiles) issue(t + gridDim.x, buf ^ 1);
        const int xb0 = buf * BUFF, gb0 = buf * BUFF + XF;
#pragma unroll
        for (int k2 = 0; k2 < NKS / 2; ++k2) {
            // this wave's k-step: 2*k2 + half (half is wave-uniform; both candidates are constants after unrolling)
            const int ks0 = 2 * k2, ks1 = 2 * k2 + 1;
            auto rowoff = [](int ks) {
                const int p = 4 * ks, pw = p % G::TQW, ph = (p / G::TQW) % G::TQH, pd = p / (G::TQW * G::TQH);
                return (((pd * G::IS) * G::RH + ph * G::IS) * G::RW + pw * G::IS) * CC;
            };
            const int xo = xb0 + (half ? rowoff(ks1) : rowoff(ks0));
            const int go = gb0 + 64 * (half ? ks1 : ks0) + lane;
            const float bv = lds[go];
            float av[SPW];
#pragma unroll
            for (int s = 0; s < SPW; ++s) av[s] = lds[xo + toff[s]];
#pragma unroll
            for (int s = 0; s < SPW; ++s) acc[s] = MVS_MFMA_16x16x4(av[s], bv, acc[s]);
        }
        MVS_WAIT_VMCNT(0);       // the next tile's DMA has landed (it was requested a whole tile of MFMAs ago)
        MVS_LDS_BARRIER();
    }

    // ---- the two halves of the K split meet in LDS; half 0 writes the workgroup's partial image ----
    __syncthreads();
    if (half == 1) {
#pragma unroll
        for (int s = 0; s < SPW; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[((w4 * SPW + s) * 4 + r) * 64 + lane] = acc[s][r];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int slot = w4 + 4 * s;
            if (slot >= NSLOT) continue;
            const int co = l15;
            if (co >= a.CG) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[s][r] + lds[((w4 * SPW + s) * 4 + r) * 64 + lane];
                const int i = 4 * g + r;            // D layout: row = 4*(lane>>4)+r -> M index, col = lane&15 -> co
                int tap, ci;
                if (CC == 16) { tap = slot; ci = i; }
                else { tap = 2 * slot + (i >> 3); ci = i & 7; }
                if (tap < 27) a.part[(((size_t)blockIdx.x * 27 + tap) * a.CX + ci) * a.CG + co] = v;
            }
        }
    }
}

bool conv_wgrad_pers_serves(int geom, int CX, int CG) {
    if (CG > 16 || CG % 4) return false;
    return (geom == GEOM_S1 && CX == 16) || (geom == GEOM_S2 && CX == 8);
}

// a: as run_wgrad fills it for the full-size tiles of `geom`; returns the number of partial images written (one per workgroup), < 0 on error
int run_conv_wgrad_pers(int geom, const WgradArgs& a, int max_groups, hipStream_t st) {
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    int groups = g_conv_pers_groups > 0 ? g_conv_pers_groups : 256;       // 111-115 KB of LDS: one workgroup (8 waves) per CU
    if (groups > ntiles) groups = ntiles;
    if (groups > max_groups) groups = max_groups;
    if (geom == GEOM_S1) MVS_LAUNCH((conv_wgrad_pers_kernel<GEOM_S1, 16>), dim3(groups), dim3(512), 0, st, a);
    else MVS_LAUNCH((conv_wgrad_pers_kernel<GEOM_S2, 8>), dim3(groups), dim3(512), 0, st, a);
    const int rc = mvs_check_launch("conv_wgrad_pers");
    return rc ? rc : groups;
}

// ================================================================================================
// conv0's weight gradient (mvsnet.py:40: 32 -> 8, stride 1), output-gradient-shifted form.
//   dW[t][ci][co] = sum_p X[p + t - 1][ci] * G[p][co] = sum_q X[q][ci] * G[q + 1 - t][co]        (q = p + t - 1)
// conv_c8_wgrad_kernel (conv3d.hip) walks p, shifts X and runs on the 16-block 4x4x1 MFMA (Cout = 8 half-fills a 16-wide N tile);
// that instruction issues every 11 cycles instead of its nominal 8 (profiles/r01_mfma_rate.log), and the kernel sits at 85 % of
// THAT ceiling = 53 % of the fp32 MFMA peak with one wave per SIMD (110 KB of LDS: a 6 x 6 x 18 halo of 32-channel voxels).
// Walking q instead makes the SHIFTED operand the narrow one:
//   * GEMM view: M = ci (two 16-row tiles), K = 4 consecutive W positions q, N = 16 = (a PAIR of taps, co): A = X[q][ci] is read
//     unshifted, B = G[q + 1 - t][co] for the two taps of the pair -- the full-rate v_mfma_f32_16x16x4_f32 with every N column
//     useful (14 pairs for 27 taps: 96 %), 28 MFMAs of 32 cycles per four positions instead of 112 of 11;
//   * X needs NO halo (each voxel is read from HBM exactly once: 503 MB), the 8-channel G takes the 6 x 6 x 18 halo instead
//     (20 KB per tile, out of L2 for the most part);
//   * X tile + G halo = 53 KB -> double-buffered LDS-DMA staging, eight waves on one tile (two per SIMD): wave rank
//     (wave & 3) = (M tile, parity of the tap pair) owns seven accumulator tiles, waves w and w + 4 split the k-steps of a tile
//     and meet in LDS once at the end of the kernel;
//   * tap pairs are chosen so that the two halves of a B read sit on different banks: (wt = 0, wt = 2) of a (dt, ht) are 16 floats
//     apart, (ht, ht + 1) at wt = 1 are 18 voxels = 16 floats (mod 32) apart.
// Same products as conv_c8_wgrad_kernel, another order of the K sum: results agree to fp32 rounding.  One partial image per
// workgroup, summed by wgrad_finish (conv3d.hip).
// ================================================================================================
__device__ __forceinline__ int gs_pair_tap(int nt, int t2) {      // tap (dt*9 + ht*3 + wt) of column half t2 of pair nt; 27: none
    if (nt < 9) return 3 * nt + 2 * t2;                            // (dt, ht) = nt: wt = 0 / 2
    if (nt < 12) return 9 * (nt - 9) + 1 + 3 * t2;                 // dt = nt - 9: (ht = 0, wt = 1) / (ht = 1, wt = 1)
    if (nt == 12) return t2 ? 16 : 7;                              // (0, 2, 1) / (1, 2, 1)
    return t2 ? 27 : 25;                                           // (2, 2, 1) / none
}

// NW: waves per workgroup, 8 or 16 (two or four per SIMD: ~100 registers per wave leave room for either); NW / 4 wave groups split a
// tile's k-steps.  Measured alone at config 2 (profiles/r05_run28..31; conv_c8_wgrad_kernel: 0.650 ms): first form (hipcc's own
// schedule, DMA requests in front of the tile's first MFMA) 0.553, software pipeline 0.463, 16 waves + requests under the MFMAs
// 0.447 ms = 121 TFLOP/s = 77 % of the fp32 MFMA peak.  Diagnostic builds: without any DMA after the first tile 0.403 (the MFMA loop
// itself: 27.5 M MFMAs x 32 cycles / 1024 SIMDs at 2.2-2.3 GHz = 0.38-0.39), X requests only 0.449, G requests only 0.445 -- the
// staging traffic, not the schedule, is what is left.  The loop is written as an explicit software pipeline (operands of k-step j + 1 requested before the MFMAs of
// k-step j, scheduling fences around the MFMA block): left to itself hipcc issues the eight reads of a k-step, waits, issues its
// seven MFMAs and only then the next reads.  All k-step-dependent parts of the LDS addresses are instruction immediates.
template <int NW>
__global__ __launch_bounds__(NW * 64) void conv_c8_wgrad_gs_kernel(WgradArgs a) {
    constexpr int CX = 32, CG = 8, KSPLIT = NW / 4;
    constexpr int TD = 4, TH = 4, TW = 16, NPOS = TD * TH * TW;           // X tile (no halo)
    constexpr int GD = TD + 2, GH = TH + 2, GW = TW + 2, NGV = GD * GH * GW;   // G halo tile
    constexpr int XITEMS = NPOS * (CX / 4), GITEMS = NGV * (CG / 4);      // 16-byte items
    constexpr int XDMA = XITEMS / 64, GDMA = (GITEMS + 63) / 64;          // wave-level DMA instructions
    constexpr int NDMA = XDMA + GDMA, DPW = (NDMA + NW - 1) / NW;
    constexpr int XF = XDMA * 256, BUFF = XF + GDMA * 256;                // floats
    constexpr int NS = 7;                                                 // tap pairs (accumulator tiles) per wave
    constexpr int NJ = NPOS / 4 / KSPLIT;                                 // k-steps of a tile per wave
    static_assert(NW == 8 || NW == 16, "the k-steps of one row (four) go to one or two wave groups each");
    static_assert((KSPLIT - 1) * 4 * NS * 256 <= 2 * BUFF, "the K split's exchange fits the staging buffers");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUFF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int w4 = wave & 3, kq = wave >> 2;
    const int mt = w4 >> 1, par = w4 & 1;
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;

    // ---- per-lane DMA items ----
    int rel[DPW], crd[DPW];      // crd: X items (pd | ph << 8 | pw << 16), G items (rd | rh << 8 | rw << 16), -1: none
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int d = NW * j + wave;
        rel[j] = 0; crd[j] = -1;
        if (d < XDMA) {
            const int i = 64 * d + lane, vox = i >> 3, cq = i & 7;
            const int pw = vox % TW, ph = (vox / TW) % TH, pd = vox / (TW * TH);
            rel[j] = ((pd * a.Hi + ph) * a.Wi + pw) * CX + 4 * cq;
            crd[j] = pd | (ph << 8) | (pw << 16);
        } else if (d < NDMA) {
            const int i = 64 * (d - XDMA) + lane, gv = i >> 1, hq = i & 1;
            const int rw = gv % GW, rh = (gv / GW) % GH, rd = gv / (GW * GH);
            rel[j] = ((rd * a.QH + rh) * a.QW + rw) * CG + 4 * hq;
            if (i < GITEMS) crd[j] = rd | (rh << 8) | (rw << 16);
        }
    }
    const float* __restrict__ zero = g_conv_zero_page;
    // tile order: linear, workgroup i takes tiles i, i + gridDim.x, ... (X has no halo to share; the brick order of conv_c8_wgrad_kernel
    // measured 1-2 % slower here: profiles/r05_run29..31)
    const int vb = blockIdx.x;
    auto issue = [&](int t, int buf) {
        int b, td, th, tw;
        linear_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        const int qd0 = td * TD, qh0 = th * TH, qw0 = tw * TW;
        const float* __restrict__ xb = a.x + ((((long long)b * a.Di + qd0) * a.Hi + qh0) * a.Wi + qw0) * CX;
        const float* __restrict__ gb = a.g + ((((long long)b * a.QD + (qd0 - 1)) * a.QH + (qh0 - 1)) * a.QW + (qw0 - 1)) * CG;
        const bool interior = qd0 >= 1 && qd0 + TD + 1 <= a.Di && qh0 >= 1 && qh0 + TH + 1 <= a.Hi && qw0 >= 1 && qw0 + TW + 1 <= a.Wi;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int d = NW * j + wave;
            if (d >= NDMA) break;                    // wave-uniform
            const bool is_g = d >= XDMA;             // wave-uniform
            const float* src = (is_g ? gb : xb) + rel[j];
            bool ok = crd[j] >= 0;
            if (!interior) {
                const int c0 = crd[j] & 255, c1 = (crd[j] >> 8) & 255, c2 = (crd[j] >> 16) & 255;
                if (is_g) ok = ok && qd0 - 1 + c0 >= 0 && qd0 - 1 + c0 < a.QD && qh0 - 1 + c1 >= 0 && qh0 - 1 + c1 < a.QH &&
                               qw0 - 1 + c2 >= 0 && qw0 - 1 + c2 < a.QW;
                else ok = ok && qd0 + c0 < a.Di && qh0 + c1 < a.Hi && qw0 + c2 < a.Wi;
            }
            if (!ok) src = zero + 4 * (lane & 15);
            MVS_DMA16(&lds[buf * BUFF + d * 256], src);      // X instructions fill [0, XF), G instructions follow
        }
    };

    // ---- per-lane operand offsets (floats): A inside the X part, B inside the G part, both incl. the lane group's position g and
    //      the wave group's k-step inside a row (k-step ks = KSPLIT * j + kq starts at position 4 * ks: row j / (4 / KSPLIT), pw = ...) ----
    const int aoff = (g + 4 * kq) * CX + 16 * mt + l15;
    int goff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int tap = gs_pair_tap(2 * s + par, l15 >> 3);
        if (tap > 26) tap = 13;                      // the column half without a tap computes unused values (no branch in the loop)
        const int dt = tap / 9, ht = (tap / 3) % 3, wt = tap % 3;
        goff[s] = XF + ((((2 - dt) * GH + (2 - ht)) * GW + (2 - wt) + g + 4 * kq) * CG) + (l15 & 7);
    }
    f32x4 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (vb < ntiles) issue(vb, 0);
    MVS_WAIT_VMCNT(0);
    __syncthreads();

    int it = 0;
    for (int t = vb; t < ntiles; t += gridDim.x, ++it) {
        const int buf = it & 1;
        const float* __restrict__ xs = lds + buf * BUFF + aoff;
        const float* __restrict__ gs[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) gs[s] = lds + buf * BUFF + goff[s];
        // wave-independent (compile-time) part of k-step j's offsets: positions 4 * KSPLIT * j ...
        auto xrow = [](int j) { return 4 * KSPLIT * j * CX; };
        auto grow = [](int j) {
            const int p = 4 * KSPLIT * j, pw = p % TW, ph = (p / TW) % TH, pd = p / (TW * TH);
            return ((pd * GH + ph) * GW + pw) * CG;
        };
        float av = xs[xrow(0)], bv[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) bv[s] = gs[s][grow(0)];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float an = 0.f, bn[NS];
            if (j + 1 < NJ) {
                an = xs[xrow(j + 1)];
#pragma unroll
                for (int s = 0; s < NS; ++s) bn[s] = gs[s][grow(j + 1)];
            }
            MVS_SCHED_FENCE();
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = MVS_MFMA_16x16x4(av, bv[s], acc[s]);
            MVS_SCHED_FENCE();
            // the next tile's requests (tile coordinates, addresses, bounds, NDMA / NW instructions) go out under the execution of
            // the MFMAs just issued instead of in front of the tile's first one
            if (j == 1 && t + (int)gridDim.x < ntiles) issue(t + gridDim.x, buf ^ 1);
            if (j + 1 < NJ) {
                av = an;
#pragma unroll
                for (int s = 0; s < NS; ++s) bv[s] = bn[s];
            }
        }
        MVS_WAIT_VMCNT(0);       // the next tile's DMA has landed (it was requested a whole tile of MFMAs ago)
        MVS_LDS_BARRIER();
    }

    // ---- the wave groups of the K split meet in LDS; group 0 writes the workgroup's partial image ----
    __syncthreads();
    if (kq > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[((((kq - 1) * 4 + w4) * NS + s) * 4 + r) * 64 + lane] = acc[s][r];
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int tap = gs_pair_tap(2 * s + par, l15 >> 3);
            if (tap > 26) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[s][r];
#pragma unroll
                for (int q = 1; q < KSPLIT; ++q) v += lds[((((q - 1) * 4 + w4) * NS + s) * 4 + r) * 64 + lane];
                const int ci = 16 * mt + 4 * g + r;          // D layout: row = 4*(lane>>4)+r -> M index, col = lane&15 -> (tap of the pair, co)
                a.part[(((size_t)blockIdx.x * 27 + tap) * CX + ci) * CG + (l15 & 7)] = v;
            }
        }
    }
}

bool conv_c8_wgrad_gs_serves(int geom, int CX, int CG) { return geom == GEOM_S1 && CX == 32 && CG == 8; }

// a: as run_wgrad fills it (4 x 4 x 16 tiles over the volume); waves: 8 or 16 per workgroup; returns the number of partial images written, < 0 on error
int run_conv_c8_wgrad_gs(const WgradArgs& a, int max_groups, int waves, hipStream_t st) {
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    int groups = max_groups < 256 ? max_groups : 256;                     // 106 KB of LDS: one workgroup (8 waves) per CU
    if (groups > ntiles) groups = ntiles;
    if (groups < 1) groups = 1;
    if (waves == 16) MVS_LAUNCH((conv_c8_wgrad_gs_kernel<16>), dim3(groups), dim3(1024), 0, st, a);
    else MVS_LAUNCH((conv_c8_wgrad_gs_kernel<8>), dim3(groups), dim3(512), 0, st, a);
    const int rc = mvs_check_launch("conv_c8_wgrad_gs");
    return rc ? rc : groups;
}
