// OPT-IN (knob "conv0_x3", default 0; bit 0 = input gradient, bit 1 = forward): conv0 of the regulariser (mvsnet.py:40: 32 -> 8
// channels, stride 1, 192x128x160 at BASELINE config 2) on the bf16 MFMA with fp32 operands SPLIT into three bf16 terms each
// ("bf16x3"), fp32 accumulation.  This header describes the arithmetic and the INPUT-GRADIENT kernel (8 -> 32 channels); the forward
// kernel (a D-marching kernel, further down) has its own.  Step at config 2: 4.69 ms default, 4.48 with either kernel, 4.28 with both.
//
// Why: the fp32 MFMA rate of gfx950 (157 TFLOP/s) is what bounds conv0's three kernels (1.7 of the step's 4.7 ms, 56-65 % of that
// peak); v_mfma_f32_16x16x32_bf16 runs at 16x that rate.  An fp32 number is EXACTLY the sum of three bf16 numbers
// (x = h + m + l: 8 + 8 + 8 significant bits, same exponent range -- truncation split, every subtraction exact), a bf16 x bf16
// product is exact in fp32, so  x * w = sum over the nine term products; this kernel issues the six largest
// (h h, h m, m h, h l, l h, m m: the three dropped ones are <= 2^-23 |x w| together, half an ulp of the fp32 product the
// fp32 kernels round to) and accumulates in the MFMA's fp32 accumulator like the fp32 kernels do.  Same arithmetic class as
// "BF16x9 / 3xTF32 fp32 emulation" in vendor BLAS libraries; it is NOT the product's default because it changes which hardware
// unit computes the path's dominant layer -- the default step stays on the fp32 MFMA, bench.py reports this mode beside it.
// Inf / NaN inputs give NaN (inf - inf in the split), where the fp32 kernels would give inf.
//
// GEMM view (conv_map.h GEOM_S1, tile 4 x 4 x 16 output voxels per workgroup, wave = depth slice, 4 rows of 16 voxels per wave):
//   M = 32 output channels (two 16-row blocks; the WEIGHTS are the A operand, so a lane ends up with 4 consecutive output
//   channels of one voxel -> 16-byte stores), N = 16 voxels along W, K = (tap, ci) = 27 x 8 = 216 -> 7 k-steps of 32 (one k-step =
//   four taps; lane group kg = lane >> 4 takes tap 4 ks + kg, its 8 k = the 8 input channels: one ds_read_b128 per term).
//   * the weight-term fragments (3 terms x 7 k-steps x 2 channel blocks) are built ONCE per persistent workgroup from the parameter
//     tensor itself -- no packed image, no extra launch; the high term (half of the products) stays in registers (56 VGPRs), the
//     middle and low terms are read back from LDS once per k-step and serve the wave's four rows (first version: all 168 VGPRs in
//     registers, one wave per SIMD -- hipcc keeps MFMA inputs in the architectural half of the file, 16 registers were left for
//     the B fragments and every k-step waited for its own ds_reads);
//   * the halo tile is split ONCE when it is staged: global fp32 -> registers (requested one tile ahead, under the MFMAs of the
//     current tile) -> three bf16 planes in LDS (16 bytes per voxel and plane, 31 KB);
//   * per k-step: 4 + 12 ds_read_b128 and 48 MFMAs (4 rows x 2 channel blocks x 6 products) on eight independent accumulators;
//     two workgroups per CU (60 KB of LDS each, <= 256 registers): one wave's LDS latency is the other's MFMA time.
// MFMA time at config 2: 245 760 rows x 84 MFMAs x 16 cycles / 1024 SIMDs = 0.134 ms (fp32 form: conv_pers_kernel, 0.49-0.52 ms).
// Measured (profiles/r06_x3_*): 0.29-0.30 ms alone, step 4.68 -> 4.48 ms; with one k-step of seven (staging + stores + 1/7 of the
// MFMAs) 0.175 ms, without staging after the first tile (all MFMAs + stores) 0.245 ms: neither side is at its floor and they
// overlap imperfectly -- two workgroups per CU is all the 238 registers allow.  Error against an fp64 reference: relative L1
// 1.06e-7 against the fp32-MFMA kernel's 2.26e-7 (fewer rounding steps: 7 x 32-wide exact-product sums per accumulator).
#include <string.h>
#include "mvs_rt.h"
#include "conv_map.h"
#include "conv_args.h"

__device__ __attribute__((aligned(16))) float g_x3_zero_page[4];   // what an out-of-volume halo item loads
int g_conv_x3 = 0;   // tuning knob "conv0_x3": bit 0 = conv0's input gradient, bit 1 = conv0's forward through this file

// Which tap lane group kg of k-step ks multiplies (27 = none: zero weights).  ds_read_b128 is served in four groups of 16 lanes
// that MIX two lane groups of the MFMA layout ({0-3, 12-15} of kg 0 with {4-11} of kg 1, ...: MI355X_MICROARCH.md, LDS table), and a
// voxel is 16 bytes = four banks: the two taps of a (kg 0, kg 1) or (kg 2, kg 3) pair are conflict-free when their halo offsets
// differ by a multiple of 16 voxels.  With the depth-plane stride padded from 108 to 112 voxels that holds for (dz 0, dz 1) of one
// (dy, dx) -- nine pairs -- and for (dy, dx 2) with (dy + 1, dx 0); the first version's order (tap = 4 ks + kg: neighbours along W
// in one pair) made every read two-way conflicted.  Two pairs of the fourteen keep a one-voxel shift.
//   ks 0: 0 9 1 10 | 1: 2 11 3 12 | 2: 4 13 5 14 | 3: 6 15 7 16 | 4: 8 17 20 21 | 5: 23 24 18 19 | 6: 25 26 22 27   (tap = dz*9 + dy*3 + dx)
__device__ __forceinline__ int x3_tap(int ks, int kg) {
    const unsigned packed = ks == 0 ? 0x0A010900u : ks == 1 ? 0x0C030B02u : ks == 2 ? 0x0E050D04u : ks == 3 ? 0x10070F06u :
                            ks == 4 ? 0x15141108u : ks == 5 ? 0x13121817u : 0x1B161A19u;      // byte kg
    return (int)((packed >> (8 * kg)) & 0xffu);
}

__device__ __forceinline__ unsigned x3_f2u(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
__device__ __forceinline__ float x3_u2f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
// x == u2f(h) + u2f(m) + u2f(l) exactly; each term has zero low 16 bits (a bf16 value held in an fp32 pattern)
__device__ __forceinline__ void x3_split(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = x3_f2u(x) & 0xffff0000u;
    const float r = x - x3_u2f(h);          // the low 16 significand bits: exact
    m = x3_f2u(r) & 0xffff0000u;
    l = x3_f2u(r - x3_u2f(m));              // <= 8 significant bits: exactly a bf16 value
}
// two such patterns -> one dword of packed bf16 (element 0 in bits 0..15)
__device__ __forceinline__ unsigned x3_pk(unsigned e0, unsigned e1) { return (e1 & 0xffff0000u) | (e0 >> 16); }
__device__ __forceinline__ mvs_bf16x8 x3_frag(const unsigned (&e)[8]) {
    uint4 q = make_uint4(x3_pk(e[0], e[1]), x3_pk(e[2], e[3]), x3_pk(e[4], e[5]), x3_pk(e[6], e[7]));
    mvs_bf16x8 f;
    memcpy(&f, &q, 16);
    return f;
}

template <int MINW>
__global__ __launch_bounds__(256, MINW) void conv_x3_s1_8_32_kernel(ConvArgs a, const float* __restrict__ w, int wlayout, int flip) {
    using G = ConvGeom<GEOM_S1>;
    constexpr int NR = G::RD * G::RH * G::RW;            // 648 halo voxels
    constexpr int KS = 7;                                // k-steps: 27 taps x 8 channels = 216 -> 7 x 32 (tap 27 = zero weights)
    constexpr int NITEMS = NR * 2, NIT = (NITEMS + 255) / 256;   // 16-byte items of the fp32 halo (4 channels each)
    constexpr int PS = 112;                              // depth-plane stride of the halo in LDS (voxels): 6 x 18 = 108, padded (x3_tap)
    constexpr int NRP = G::RD * PS;
    __shared__ __attribute__((aligned(16))) uint4 halo[3][NRP];  // term planes: 8 bf16 per voxel
    __shared__ __attribute__((aligned(16))) uint4 wlds[2][KS][2][64];   // middle / low weight terms: [term][k-step][channel block][lane]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, kg = lane >> 4;
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;

    // ---- weight fragments, three terms: A[row = output channel 16 mb + (lane & 15)][k = 8 kg + j] = Wk[tap 4 ks + kg][ci j][row] ----
    mvs_bf16x8 areg[KS][2];      // the high term
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int tap = x3_tap(ks, kg);
        const int kidx = flip ? 26 - tap : tap;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int co = 16 * mb + n;
            unsigned h[8], m[8], l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = 0.f;
                if (tap < 27) v = wlayout == WL_OIK ? w[((size_t)co * 8 + j) * 27 + kidx] : w[((size_t)j * 32 + co) * 27 + kidx];
                x3_split(v, h[j], m[j], l[j]);
            }
            areg[ks][mb] = x3_frag(h);
            if (wv == 0) {       // (every wave computes the same fragments; one writes the shared terms)
                const mvs_bf16x8 fm = x3_frag(m), fl = x3_frag(l);
                memcpy(&wlds[0][ks][mb][lane], &fm, 16);
                memcpy(&wlds[1][ks][mb][lane], &fl, 16);
            }
        }
    }
    // ---- per-lane B offsets (halo voxels): the tap of each k-step, the wave's four rows ----
    int toff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int tap = x3_tap(ks, kg) < 27 ? x3_tap(ks, kg) : x3_tap(ks, kg - 1);   // no tap: zero weights, its partner's address
        toff[ks] = (tap / 9) * PS + ((tap / 3) % 3) * G::RW + tap % 3;
    }
    // ---- per-thread staging items: tile-invariant source offsets + halo coordinates ----
    int rel[NIT], crd[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = tid + 256 * k, vox = i >> 1, half = i & 1;
        const int rw = vox % G::RW, rh = (vox / G::RW) % G::RH, rd = vox / (G::RW * G::RH);
        rel[k] = ((rd * a.Hi + rh) * a.Wi + rw) * 8 + 4 * half;
        crd[k] = i < NITEMS ? (rd | (rh << 8) | (rw << 16)) : -1;
    }
    auto request = [&](int t, float4 (&pf)[NIT]) {
        int b, td, th, tw;
        linear_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        const int id0 = td * G::TQD - 1, ih0 = th * G::TQH - 1, iw0 = tw * G::TQW - 1;
        const long long org = ((((long long)b * a.Di + id0) * a.Hi + ih0) * a.Wi + iw0) * 8;
        const float* __restrict__ xb = a.x + org;    // (may lie in front of the tensor for border tiles: only in-volume items use it)
        const bool interior = id0 >= 0 && id0 + G::RD <= a.Di && ih0 >= 0 && ih0 + G::RH <= a.Hi && iw0 >= 0 && iw0 + G::RW <= a.Wi;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            bool ok = crd[k] >= 0;
            if (!interior && ok) {
                const int rd = crd[k] & 255, rh = (crd[k] >> 8) & 255, rw = (crd[k] >> 16) & 255;
                ok = id0 + rd >= 0 && id0 + rd < a.Di && ih0 + rh >= 0 && ih0 + rh < a.Hi && iw0 + rw >= 0 && iw0 + rw < a.Wi;
            }
            // BRANCH-FREE on purpose: a load under `if (ok)` is a phi with the zero, and hipcc waits for it right there -- the first
            // version's "prefetch" sat in front of an s_waitcnt vmcnt(0) ahead of the MFMAs, one exposed memory round trip per tile
            const float* src = ok ? xb + rel[k] : g_x3_zero_page;
            pf[k] = *reinterpret_cast<const float4*>(src);
        }
    };
    auto deposit = [&](const float4 (&pf)[NIT]) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (crd[k] < 0) continue;
            const int i = tid + 256 * k, half = i & 1;
            const int vox = (crd[k] & 255) * PS + ((crd[k] >> 8) & 255) * G::RW + ((crd[k] >> 16) & 255);
            unsigned h[4], m[4], l[4];
            x3_split(pf[k].x, h[0], m[0], l[0]); x3_split(pf[k].y, h[1], m[1], l[1]);
            x3_split(pf[k].z, h[2], m[2], l[2]); x3_split(pf[k].w, h[3], m[3], l[3]);
            uint2 qh, qm, ql;
            qh.x = x3_pk(h[0], h[1]); qh.y = x3_pk(h[2], h[3]);
            qm.x = x3_pk(m[0], m[1]); qm.y = x3_pk(m[2], m[3]);
            ql.x = x3_pk(l[0], l[1]); ql.y = x3_pk(l[2], l[3]);
            reinterpret_cast<uint2*>(&halo[0][vox])[half] = qh;
            reinterpret_cast<uint2*>(&halo[1][vox])[half] = qm;
            reinterpret_cast<uint2*>(&halo[2][vox])[half] = ql;
        }
    };

    float4 pf[NIT];
    if ((int)blockIdx.x < ntiles) request(blockIdx.x, pf);
    // (measured: deposit AFTER the MFMAs with the output stores behind it -- so that the wait in front of the deposit does not
    //  include this tile's stores -- is slower, 0.335 vs 0.30 ms: the accumulators stay live across the deposit)
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        deposit(pf);
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) request(t + gridDim.x, pf);     // in flight under this tile's MFMAs
        int b, td, th, tw;
        linear_tile(t, a.ntw, a.nth, a.ntd, b, td, th, tw);
        const int qd = td * G::TQD + wv, qw = tw * G::TQW + n;
        // two accumulators per output fragment: the h h products (the magnitude of the result) and the five small products
        // (2^-8 of it and below, so their rounding steps are 2^-8 of the former's): the sum has the rounding steps of ONE chain
        f32x4 acc[4][2], acs[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[r][mb] = acs[r][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            mvs_bf16x8 at[3][2], bt[4][3];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                at[0][mb] = areg[ks][mb];
                at[1][mb] = *reinterpret_cast<const mvs_bf16x8*>(&wlds[0][ks][mb][lane]);
                at[2][mb] = *reinterpret_cast<const mvs_bf16x8*>(&wlds[1][ks][mb][lane]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = wv * PS + r * G::RW + n + toff[ks];
#pragma unroll
                for (int p = 0; p < 3; ++p) bt[r][p] = *reinterpret_cast<const mvs_bf16x8*>(&halo[p][v]);
            }
            // six products, smallest first; independent accumulators take turns
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        if (q < 5) acs[r][mb] = MVS_MFMA_16x16x32_BF16(at[PA[q]][mb], bt[r][PB[q]], acs[r][mb]);
                        else acc[r][mb] = MVS_MFMA_16x16x32_BF16(at[PA[q]][mb], bt[r][PB[q]], acc[r][mb]);
                    }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qh = th * G::TQH + r;
            if (qd >= a.QD || qh >= a.QH || qw >= a.QW) continue;
            float* __restrict__ yo = a.y + ((((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qw) * 32 + 4 * kg;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                *reinterpret_cast<float4*>(yo + 16 * mb) = make_float4(acc[r][mb][0] + acs[r][mb][0], acc[r][mb][1] + acs[r][mb][1],
                                                                     acc[r][mb][2] + acs[r][mb][2], acc[r][mb][3] + acs[r][mb][3]);
        }
        __syncthreads();     // every wave is done with the planes before the next tile's deposit
    }
}

// ------------------------------------------------------------------------------------------------
// conv0's FORWARD (mvsnet.py:40: 32 -> 8 channels, stride 1) in the same arithmetic (knob bit 1): a D-MARCHING kernel.
// Cout = 8 fills half of the MFMA's 16 rows, so TWO output depth slices share one MFMA (the bf16 inference path's GEOM_S1_DP,
// conv3d_bf16.hip): row m = pd*8 + co; the pair (2j, 2j+1) reads the four input slices 2j-1 .. 2j+2 = "kd' 0..3", and tap
// (kd', kh, kw) carries W[kd' - pd] for row parity pd, zero where kd' - pd is outside 0..2 (75 % of the MFMA work is real).
// A k-step = ONE tap x all 32 input channels (lane group kg = channels 8 kg ..), so a phase of the kernel needs ONE input depth slice:
//   * a workgroup owns an 8 x 16 (H x W) column and a segment of D and marches along D: per phase it stages one input slice of the
//     column (10 x 18 voxels x 32 channels, every 128-byte voxel record read ONCE, halo overhead 1.4 x 1.17) as three bf16 term planes
//     (34.5 KB), and every slice pair that has this slice among its four multiplies it by the nine (kh, kw) taps of its kd';
//   * waves 0-1 take the EVEN pairs, waves 2-3 the ODD ones (four H rows each): pair j is live for input slices 2j .. 2j+3, the next
//     pair of the same parity starts right after it -- every wave works in every phase and its accumulators follow one pair;
//   * the whole three-term weight image [tap][co][32 ci] (43 KB, packed once per call by conv_x3_pack_fwd_kernel) is resident in LDS;
//     per k-step a wave reads 3 A fragments (shared by its four rows) and 12 B fragments for 24 MFMAs;
//   * both LDS layouts keep a 64-byte record per voxel / per (tap, co) with the four 16-byte channel-group slots XOR-swizzled by
//     bit 2 of the record index: every ds_read_b128 lane group ({0-3, 12-15} of one kg with {4-11} of the next) is conflict-free
//     for every alignment (checked exhaustively when the layout was chosen).
// (The first form of this kernel staged a 4 x 4 x 16 tile in four 8-channel chunks: each chunk phase read 32 of the 128 bytes of every
//  voxel record, its staging side alone took 0.40 ms and the kernel 0.55 ms against the fp32 form's 0.58: profiles/r06_x3_forward_attempt_*.)
// BatchNorm statistics (sum y, sum y^2 per channel) are accumulated per lane and added to the fp64 slots once per workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int X3F_TH = 8, X3F_TW = 16, X3F_RH = X3F_TH + 2, X3F_RW = X3F_TW + 2, X3F_PV = X3F_RH * X3F_RW;   // 180 voxels per input slice
constexpr int X3F_WENT = 28 * 8;                                         // (tap, co) records; tap 27 = zeros

__device__ __forceinline__ int x3_slot(int kg, int rec) { return kg ^ ((rec >> 1) & 2); }     // 16-byte slot of channel group kg in record rec

// [term 3][tap 28][co 8][slot 4] x 16 bytes, swizzled as the kernel reads it
__global__ __launch_bounds__(256) void conv_x3_pack_fwd_kernel(const float* __restrict__ w, int wlayout, int flip, uint4* __restrict__ img) {
    const int i = blockIdx.x * 256 + threadIdx.x;         // (record, channel group)
    if (i >= X3F_WENT * 4) return;
    const int kg = i & 3, rec = i >> 2, co = rec & 7, tap = rec >> 3;
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = 8 * kg + j, kidx = flip ? 26 - tap : tap;
        float v = 0.f;
        if (tap < 27) v = wlayout == WL_OIK ? w[((size_t)co * 32 + ci) * 27 + kidx] : w[((size_t)ci * 8 + co) * 27 + kidx];
        x3_split(v, h[j], m[j], l[j]);
    }
    uint4* dst = img + rec * 4 + x3_slot(kg, rec);
    dst[0] = make_uint4(x3_pk(h[0], h[1]), x3_pk(h[2], h[3]), x3_pk(h[4], h[5]), x3_pk(h[6], h[7]));
    dst[X3F_WENT * 4] = make_uint4(x3_pk(m[0], m[1]), x3_pk(m[2], m[3]), x3_pk(m[4], m[5]), x3_pk(m[6], m[7]));
    dst[2 * X3F_WENT * 4] = make_uint4(x3_pk(l[0], l[1]), x3_pk(l[2], l[3]), x3_pk(l[4], l[5]), x3_pk(l[6], l[7]));
}

// grid: (column, segment) = blockIdx.x; SD = depth slices per segment (even)
__global__ __launch_bounds__(256, 2) void conv_x3_fwd_march_kernel(ConvArgs a, const uint4* __restrict__ wimg, int nch, int ncw, int SD) {
    constexpr int RW = X3F_RW, PV = X3F_PV;
    constexpr int NITEMS = PV * 8, NIT = (NITEMS + 255) / 256;          // 16-byte items of one fp32 input slice of the column
    __shared__ __attribute__((aligned(16))) uint4 plane[3][PV * 4];
    __shared__ __attribute__((aligned(16))) uint4 wl[3][X3F_WENT * 4];
    __shared__ float red[4 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, kg = lane >> 4;
    const int role = wv >> 1, hh = wv & 1;                // pair parity of the wave, its four rows 4 hh ..
    int bid = blockIdx.x;
    const int cw = bid % ncw; bid /= ncw;
    const int ch = bid % nch; bid /= nch;
    const int nseg = (a.Di + SD - 1) / SD;
    const int seg = bid % nseg, b = bid / nseg;
    const int d0 = seg * SD, d1 = d0 + SD < a.Di ? d0 + SD : a.Di;
    const int npairs = (d1 - d0 + 1) >> 1;
    const int h0 = ch * X3F_TH, w0 = cw * X3F_TW;

    // ---- the weight image -> LDS (once) ----
    for (int i = tid; i < 3 * X3F_WENT * 4; i += 256) (&wl[0][0])[i] = wimg[i];

    // A: row m = lane & 15 = (pd, co)
    const int pdA = n >> 3, coA = n & 7;
    const int slotA = x3_slot(kg, coA);                  // record (tap*8 + co): bit 2 of it is bit 2 of co
    // B: voxel (row 4 hh + r + kh, column n + kw)
    const int vb0 = (4 * hh) * RW + n;

    // ---- staging items: (voxel, 16-byte part) -- 8 consecutive lanes read one voxel's whole 128-byte record ----
    int rel[NIT], dst[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = tid + 256 * k, vox = i >> 3, part = i & 7;
        const int rw = vox % RW, rh = vox / RW;
        const bool in = i < NITEMS && h0 - 1 + rh >= 0 && h0 - 1 + rh < a.Hi && w0 - 1 + rw >= 0 && w0 - 1 + rw < a.Wi;
        rel[k] = in ? (((h0 - 1 + rh) * a.Wi + (w0 - 1 + rw)) * 32 + 4 * part) : -1;
        dst[k] = i < NITEMS ? (vox * 4 + x3_slot(part >> 1, vox)) * 2 + (part & 1) : -1;      // uint2 index inside a term plane
    }
    float4 pf[NIT];
    auto request = [&](int p) {                           // input slice d0 - 1 + p
        const int din = d0 - 1 + p;
        const bool ind = din >= 0 && din < a.Di;
        const float* __restrict__ xb = a.x + ((size_t)b * a.Di + (ind ? din : 0)) * a.Hi * a.Wi * 32;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const float* src = (ind && rel[k] >= 0) ? xb + rel[k] : g_x3_zero_page;       // branch-free
            pf[k] = *reinterpret_cast<const float4*>(src);
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (dst[k] < 0) continue;
            unsigned h[4], m[4], l[4];
            x3_split(pf[k].x, h[0], m[0], l[0]); x3_split(pf[k].y, h[1], m[1], l[1]);
            x3_split(pf[k].z, h[2], m[2], l[2]); x3_split(pf[k].w, h[3], m[3], l[3]);
            uint2 qh, qm, ql;
            qh.x = x3_pk(h[0], h[1]); qh.y = x3_pk(h[2], h[3]);
            qm.x = x3_pk(m[0], m[1]); qm.y = x3_pk(m[2], m[3]);
            ql.x = x3_pk(l[0], l[1]); ql.y = x3_pk(l[2], l[3]);
            reinterpret_cast<uint2*>(&plane[0][0])[dst[k]] = qh;
            reinterpret_cast<uint2*>(&plane[1][0])[dst[k]] = qm;
            reinterpret_cast<uint2*>(&plane[2][0])[dst[k]] = ql;
        }
    };

    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};     // statistics of channels 4 (kg & 1) + e
    f32x4 acc[4], acs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acs[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nplanes = 2 * npairs + 2;
    request(0);
#pragma unroll 1
    for (int p = 0; p < nplanes; ++p) {
        deposit();
        __syncthreads();
        if (p + 1 < nplanes) request(p + 1);              // in flight under this phase's MFMAs
        // the wave's pair and which of its four input slices this is
        const int q = p - 2 * role;
        const int j = 2 * (q >> 2) + role, kdp = q & 3;
        if (q >= 0 && j < npairs) {                       // wave-uniform
            const int kd = kdp - pdA;
            const bool av = kd >= 0 && kd <= 2;
            const int abase = ((av ? kd * 9 : 27) * 8 + coA) * 4 + slotA, astep = av ? 32 : 0;      // uint4 units; + ks * astep
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                mvs_bf16x8 at[3], bt[4][3];
#pragma unroll
                for (int t = 0; t < 3; ++t) at[t] = *reinterpret_cast<const mvs_bf16x8*>(&wl[t][abase + ks * astep]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int v = vb0 + (r + ks / 3) * RW + ks % 3;
                    const int o = v * 4 + x3_slot(kg, v);
#pragma unroll
                    for (int t = 0; t < 3; ++t) bt[r][t] = *reinterpret_cast<const mvs_bf16x8*>(&plane[t][o]);
                }
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q6 = 0; q6 < 6; ++q6)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (q6 < 5) acs[r] = MVS_MFMA_16x16x32_BF16(at[PA[q6]], bt[r][PB[q6]], acs[r]);
                        else acc[r] = MVS_MFMA_16x16x32_BF16(at[PA[q6]], bt[r][PB[q6]], acc[r]);
                    }
            }
            if (kdp == 3) {
                // ---- the pair is complete: lane (n, kg) holds channels 4 (kg & 1) .. + 3 of slice d0 + 2 j + (kg >> 1), rows 4 hh + r ----
                const int qd = d0 + 2 * j + (kg >> 1), qw = w0 + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qh = h0 + 4 * hh + r;
                    const float4 o = make_float4(acc[r][0] + acs[r][0], acc[r][1] + acs[r][1], acc[r][2] + acs[r][2], acc[r][3] + acs[r][3]);
                    acc[r] = acs[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (qd >= d1 || qh >= a.Ho || qw >= a.Wo) continue;
                    *reinterpret_cast<float4*>(a.y + ((((size_t)b * a.Do + qd) * a.Ho + qh) * a.Wo + qw) * 8 + 4 * (kg & 1)) = o;
                    s1[0] += o.x; s1[1] += o.y; s1[2] += o.z; s1[3] += o.w;
                    s2[0] += o.x * o.x; s2[1] += o.y * o.y; s2[2] += o.z * o.z; s2[3] += o.w * o.w;
                }
            }
        }
        __syncthreads();     // every wave is done with the slice before the next deposit
    }
    if (a.slots) {
        // lanes that share (kg & 1) hold the same four channels: sum over n (16 lanes) and over the slice parity (kg >> 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int msk = 1; msk <= 8; msk <<= 1) { s1[e] += __shfl_xor(s1[e], msk); s2[e] += __shfl_xor(s2[e], msk); }
            s1[e] += __shfl_xor(s1[e], 32); s2[e] += __shfl_xor(s2[e], 32);
        }
        if (n == 0 && kg < 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[wv * 16 + 4 * kg + e] = s1[e]; red[wv * 16 + 8 + 4 * kg + e] = s2[e]; }
        }
        __syncthreads();
        if (tid < 16) {
            const int stat = tid >> 3, co = tid & 7;
            MVS_GLOBAL_ATOMIC_ADD_F64(a.slots + ((size_t)(blockIdx.x & (a.nslots - 1)) * 2 + stat) * a.Cout + co,
                                      (double)(red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid]));
        }
    }
}

bool conv_x3_fwd_serves(int geom, const ConvArgs& a) {
    return geom == GEOM_S1 && a.Cin == 32 && a.Cout == 8 && !a.scale && !a.shift && !a.skip && !a.bn_raw && !a.relu;
}

// ws: the op's weight workspace (mvs_conv3d_workspace_bytes: 55 KB for this layer, unused by the default 4x4x1-MFMA forward)
int run_conv_x3_fwd(const ConvArgs& a, const float* w, int wlayout, int flip, float* ws, hipStream_t st) {
    uint4* img = reinterpret_cast<uint4*>(ws);
    MVS_LAUNCH(conv_x3_pack_fwd_kernel, dim3(mvs_cdiv(X3F_WENT * 4, 256)), dim3(256), 0, st, w, wlayout, flip, img);
    const int nch = mvs_cdiv(a.Hi, X3F_TH), ncw = mvs_cdiv(a.Wi, X3F_TW), ncol = a.B * nch * ncw;
    int nseg = mvs_cdiv(2560, ncol);                      // ~5 rounds of 2 x 256 resident workgroups
    if (nseg > a.Di / 4) nseg = a.Di / 4;
    if (nseg < 1) nseg = 1;
    int SD = mvs_cdiv(a.Di, nseg);
    SD += SD & 1;
    nseg = mvs_cdiv(a.Di, SD);
    MVS_LAUNCH(conv_x3_fwd_march_kernel, dim3(ncol * nseg), dim3(256), 0, st, a, (const uint4*)img, nch, ncw, SD);
    return mvs_check_launch("conv_x3_fwd_march");
}

bool conv_x3_serves(int geom, const ConvArgs& a) {
    return geom == GEOM_S1 && a.Cin == 8 && a.Cout == 32 && !a.scale && !a.shift && !a.skip && !a.slots && !a.bn_raw && !a.relu;
}

// a: as run_igemm fills it for the full-size GEOM_S1 tiles; w: the parameter tensor (layout / flip as in conv_pack_weights_item)
int run_conv_x3(const ConvArgs& a, const float* w, int wlayout, int flip, hipStream_t st) {
    const int ntiles = a.B * a.ntd * a.nth * a.ntw;
    const int groups = ntiles < 512 ? ntiles : 512;      // 60 KB of LDS, <= 256 registers: two workgroups per CU
    MVS_LAUNCH((conv_x3_s1_8_32_kernel<2>), dim3(groups), dim3(256), 0, st, a, w, wlayout, flip);
    return mvs_check_launch("conv_x3_s1_8_32");
}
