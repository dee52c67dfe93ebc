// Micro-benchmark: issue rate of the fp32-input MFMA shapes on gfx950 (DESIGN.md section 4 cites it).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0 = {0}, d1 = d0;
    for (int i = 0; i < iters; ++i) {
        if (SHAPE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        } else if (SHAPE == 1) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
        }
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE>
void run(const char* name, double macs_per_inst, int inst_per_iter) {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE><<<1024, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SHAPE><<<1024, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double insts = 1024.0 * 4 * iters * inst_per_iter;  // wave-instructions
    double tflops = insts * macs_per_inst * 2 / (ms * 1e-3) / 1e12;
    // per SIMD: 1024 SIMDs; waves per SIMD = 4 (1024 blocks * 4 waves / 1024 SIMDs)
    double cyc = ms * 1e-3 * 2.4e9 / (insts / 1024.0);
    printf("%-28s %8.3f ms  %7.1f TFLOP/s  ~%.1f cycles/inst/SIMD @2.4GHz\n", name, ms, tflops, cyc);
    hipFree(out);
}

int main() {
    run<0>("v_mfma_f32_16x16x4_f32", 16 * 16 * 4, 4);
    run<1>("v_mfma_f32_4x4x1_16b_f32", 16 * 4 * 4 * 1, 4);
    run<2>("v_mfma_f32_32x32x2_f32", 32 * 32 * 2, 2);
    return 0;
}
