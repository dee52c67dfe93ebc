// Index maps shared by the implicit-GEMM convolution kernels, the weight packer and the host code.
//
// Geometry (all k=3, pad=1 as in the reference: jdacs/models/mvsnet.py:40-63,
// jdacs-ms/models/network.py:47-65):
//   GEOM_S1 : y[o] = sum_t x[o + t - 1] W[t]        "coarse grid" q = o      (stride-1 conv, and the
//             stride-1 transposed conv / stride-1 dgrad after flipping taps in the packer)
//   GEOM_S2 : y[o] = sum_t x[2o + t - 1] W[t]       q = o                    (stride-2 conv)
//   GEOM_TR2: y[2q+p] = sum_{t in class p} x[q + delta_t] W[k_t]             (stride-2 transposed conv
//             with output_padding 1 == dgrad of the stride-2 conv); 8 parity classes p=(pd,ph,pw);
//             per dim: p=0 -> {(delta 0, k 1)}, p=1 -> {(delta 1, k 0), (delta 0, k 2)}.
#pragma once

#if defined(__HIPCC__) || defined(MVS_CPU_EMUL)
#define MVS_HD __host__ __device__
#else
#define MVS_HD
#endif

enum { GEOM_S1 = 0, GEOM_S2 = 1, GEOM_TR2 = 2,
       // the same index maps on quarter-size workgroup tiles (one 16-voxel row per wave): 4x the workgroups for the deep U-Net
       // levels, whose volumes (24x16x20 at config 2) give the full-size tiles fewer workgroups than the chip has CUs
       GEOM_S1_SMALL = 3, GEOM_S2_SMALL = 4, GEOM_TR2_SMALL = 5,
       // transposed stride-2 conv with Cout == 8: the two W-parity classes of a (pd, ph) pair as ONE GEMM with N = 16 = (pw, co)
       GEOM_TR2_PW = 6,
       // stride-1 conv with Cout == 8 on the bf16 path (conv3d_bf16.hip only): two output depth slices per MFMA, row = pd*8 + co,
       // K = the 4 x 3 x 3 input offsets under the slice pair
       GEOM_S1_DP = 7 };
// source weight tensor layout: OIK = [out'][in'][3][3][3], IOK = [in'][out'][3][3][3]
enum { WL_OIK = 0, WL_IOK = 1 };

template <int GEOM>
struct ConvGeom;
template <>
struct ConvGeom<GEOM_S1> {
    static constexpr bool PW = false;
    static constexpr int BASE = GEOM_S1;
    static constexpr int TQD = 4, TQH = 4, TQW = 16, MB = 4;
    static constexpr int IS = 1, OS = 1, NCLS = 1, PAD = 1;
    static constexpr int RD = TQD + 2, RH = TQH + 2, RW = TQW + 2;
};
template <>
struct ConvGeom<GEOM_S1_DP> {
    static constexpr bool PW = false;
    static constexpr int BASE = GEOM_S1;
    static constexpr int TQD = 4, TQH = 4, TQW = 16, MB = 4;
    static constexpr int IS = 1, OS = 1, NCLS = 1, PAD = 1;
    static constexpr int RD = TQD + 2, RH = TQH + 2, RW = TQW + 2;
};
template <>
struct ConvGeom<GEOM_S2> {
    static constexpr bool PW = false;
    static constexpr int BASE = GEOM_S2;
    static constexpr int TQD = 2, TQH = 4, TQW = 16, MB = 2;
    static constexpr int IS = 2, OS = 1, NCLS = 1, PAD = 1;
    static constexpr int RD = 2 * TQD + 1, RH = 2 * TQH + 1, RW = 2 * TQW + 1;
};
template <>
struct ConvGeom<GEOM_TR2_PW> {
    static constexpr int BASE = GEOM_TR2;
    static constexpr bool PW = true;
    static constexpr int TQD = 4, TQH = 4, TQW = 16, MB = 4;
    static constexpr int IS = 1, OS = 2, NCLS = 4, PAD = 0;
    static constexpr int RD = TQD + 1, RH = TQH + 1, RW = TQW + 1;
};
template <>
struct ConvGeom<GEOM_TR2> {
    static constexpr bool PW = false;
    static constexpr int BASE = GEOM_TR2;
    static constexpr int TQD = 4, TQH = 4, TQW = 16, MB = 4;
    static constexpr int IS = 1, OS = 2, NCLS = 8, PAD = 0;
    static constexpr int RD = TQD + 1, RH = TQH + 1, RW = TQW + 1;
};
template <>
struct ConvGeom<GEOM_S1_SMALL> {
    static constexpr bool PW = false;
    static constexpr int BASE = GEOM_S1;
    static constexpr int TQD = 2, TQH = 2, TQW = 16, MB = 1;
    static constexpr int IS = 1, OS = 1, NCLS = 1, PAD = 1;
    static constexpr int RD = TQD + 2, RH = TQH + 2, RW = TQW + 2;
};
template <>
struct ConvGeom<GEOM_S2_SMALL> {
    static constexpr bool PW = false;
    static constexpr int BASE = GEOM_S2;
    static constexpr int TQD = 1, TQH = 4, TQW = 16, MB = 1;
    static constexpr int IS = 2, OS = 1, NCLS = 1, PAD = 1;
    static constexpr int RD = 2 * TQD + 1, RH = 2 * TQH + 1, RW = 2 * TQW + 1;
};
template <>
struct ConvGeom<GEOM_TR2_SMALL> {
    static constexpr bool PW = false;
    static constexpr int BASE = GEOM_TR2;
    static constexpr int TQD = 2, TQH = 2, TQW = 16, MB = 1;
    static constexpr int IS = 1, OS = 2, NCLS = 8, PAD = 0;
    static constexpr int RD = TQD + 1, RH = TQH + 1, RW = TQW + 1;
};
// tile extents of a geometry id as plain functions (host code)
MVS_HD inline int geom_base(int geom) { return geom == GEOM_TR2_PW ? GEOM_TR2 : (geom >= GEOM_S1_SMALL ? geom - GEOM_S1_SMALL : geom); }
MVS_HD inline int geom_tqd(int geom) { return geom == GEOM_TR2_PW ? 4 : (geom == GEOM_S2 ? 2 : (geom == GEOM_S2_SMALL ? 1 : (geom >= GEOM_S1_SMALL ? 2 : 4))); }
MVS_HD inline int geom_tqh(int geom) { return (geom == GEOM_S1_SMALL || geom == GEOM_TR2_SMALL) ? 2 : 4; }

// number of taps of a TR2 parity class and of the classes before it (class id = pd*4 + ph*2 + pw)
MVS_HD inline int tr2_ntaps(int cls) { return (1 + ((cls >> 2) & 1)) * (1 + ((cls >> 1) & 1)) * (1 + (cls & 1)); }
MVS_HD inline int tr2_tap_prefix(int cls) {
    int s = 0;
    for (int c = 0; c < cls; ++c) s += tr2_ntaps(c);
    return s;
}
// tap t of class cls -> region offset (dd,dh,dw) in {0,1} and kernel index (kd,kh,kw)
MVS_HD inline void tr2_tap(int cls, int t, int& dd, int& dh, int& dw, int& kd, int& kh, int& kw) {
    const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
    const int nh = 1 + ph, nw = 1 + pw;
    const int tw = t % nw, th = (t / nw) % nh, td = t / (nw * nh);
    dd = (pd && td == 0) ? 1 : 0; kd = pd ? (td == 0 ? 0 : 2) : 1;
    dh = (ph && th == 0) ? 1 : 0; kh = ph ? (th == 0 ? 0 : 2) : 1;
    dw = (pw && tw == 0) ? 1 : 0; kw = pw ? (tw == 0 ? 0 : 2) : 1;
}

// GEOM_TR2_PW (Cout == 8): class = (pd, ph) (id = pd*2 + ph); its taps are the (d, h) taps of the parity pair times the TWO input
// offsets dw in {0, 1} along W; GEMM column n = pw*8 + co.  Column pw = 0 uses (dw 0, kw 1) only (its dw = 1 weights are zero),
// column pw = 1 uses (dw 1, kw 0) and (dw 0, kw 2): 18 k-steps of 16 instead of 27 with a half-empty N tile, and a lane group
// writes 16 consecutive floats = the 8 channels of two adjacent output voxels.
MVS_HD inline int tr2p_ntaps(int cls) { return (1 + ((cls >> 1) & 1)) * (1 + (cls & 1)) * 2; }
MVS_HD inline int tr2p_tap_prefix(int cls) {
    int s = 0;
    for (int c = 0; c < cls; ++c) s += tr2p_ntaps(c);
    return s;
}
// tap t of class cls -> region offset (dd,dh,dw) and kernel index (kd,kh); kw depends on the column's pw: tr2p_kw
MVS_HD inline void tr2p_tap(int cls, int t, int& dd, int& dh, int& dw, int& kd, int& kh) {
    const int pd = (cls >> 1) & 1, ph = cls & 1;
    const int nh = 1 + ph;
    dw = t % 2;
    const int th = (t / 2) % nh, td = t / (2 * nh);
    dd = (pd && td == 0) ? 1 : 0; kd = pd ? (td == 0 ? 0 : 2) : 1;
    dh = (ph && th == 0) ? 1 : 0; kh = ph ? (th == 0 ? 0 : 2) : 1;
}
MVS_HD inline int tr2p_kw(int pw, int dw) { return pw == 0 ? (dw == 0 ? 1 : -1) : (dw == 1 ? 0 : 2); }   // -1: no such tap

// K-steps (16 flattened (tap, ci) indices each) of one class for a CC-channel chunk
MVS_HD inline int ksteps_for(int ntaps, int CC) { return (ntaps * CC + 15) / 16; }

// Packed weight buffer: [kk][nb][64 lanes][4] floats, kk = global k-step index:
//   S1/S2 : kk = chunk * ksteps_for(27, CC) + ks
//   TR2   : kk = tr2_tap_prefix(cls) * CC / 16 + ks        (single chunk, CC == Cin, CC % 16 == 0)
// lane l, element j holds W[flattened k = 16*ks + 4*(l>>4) + j][co = 16*nb + (l & 15)],
// flattened k -> (tap = k / CC, ci = chunk*CC + k % CC).
MVS_HD inline int total_ksteps(int geom, int Cin, int CC) {
    if (geom == GEOM_TR2_PW) return 18 * CC / 16;
    if (geom == GEOM_TR2) return 27 * CC / 16;
    return (Cin / CC) * ksteps_for(27, CC);
}
