// Kernel-argument block of the implicit-GEMM convolution kernels (conv3d.hip, conv3d_direct.hip).
#pragma once
struct ConvArgs {
    const float* x;         // [B,Di,Hi,Wi,Cin]
    const float* wp;        // packed weights
    float* y;               // [B,Do,Ho,Wo,Cout]
    const float* scale;     // [Cout] or null
    const float* shift;     // [Cout] or null (bias when scale == null)
    const float* skip;      // like y, or null (added after the ReLU)
    double* slots;          // BatchNorm statistic slots [nslots][2][Cout] (fp64 atomics, bn.hip) or null: per channel (sum y, sum y^2)
                            // of the RAW output -- or, with bn_raw, (sum dyh, sum dyh*xhat) of the BatchNorm+ReLU block whose output
                            // gradient this kernel writes (y = an input gradient incl. the `skip` summand)
    int nslots;             // power of two
    const float* bn_raw;    // like y: that block's raw (pre-BatchNorm) output, or null
    const float* bn_stats;  // [4][Cout]: its mean, invstd, scale, shift
    int relu;
    int B, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout;
    int QD, QH, QW;         // coarse-grid extents
    int ntd, nth, ntw;      // tiles per dim
    int nb_total;           // 16-wide Cout tiles in the packed weight image; a workgroup handles NB of them from blockIdx.y*NB
};

// weight-gradient kernels (conv3d.hip, conv3d_pers.hip)
struct WgradArgs {
    const float* x;      // [B,Di,Hi,Wi,CX]
    const float* g;      // [B,QD,QH,QW,CG]
    float* part;         // [gridDim.x][27][CX][CG]
    int B, Di, Hi, Wi, CX, CG;
    int QD, QH, QW, ntd, nth, ntw;
    int xcd;             // XCD-aware tile order (conv_c8_wgrad_kernel)
};

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

