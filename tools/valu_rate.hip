// Micro-benchmark: v_fma_f32 vs v_pk_fma_f32 issue rate on gfx950 (decides whether SLP-packing the
// plane-sweep arithmetic helps).  hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a = seed + threadIdx.x * 1e-6f, b = 0.999f;
    float s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    f32x2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8};
    f32x2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#define F(s) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(b), "v"(a))
            F(s0); F(s1); F(s2); F(s3); F(s4); F(s5); F(s6); F(s7);
        } else {
            p0 = __builtin_elementwise_fma(p0, pb, pa); p1 = __builtin_elementwise_fma(p1, pb, pa);
            p2 = __builtin_elementwise_fma(p2, pb, pa); p3 = __builtin_elementwise_fma(p3, pb, pa);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 2048 * 256 * 4);
    const int iters = 100000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<2048, 256>>>(out, 100, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<2048, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double fma_lanes = 2048.0 * 256 * iters * 8;   // 8 scalar FMAs per lane per iteration in both modes
    double winst = 2048.0 * 4 * iters * (MODE == 0 ? 8 : 4);
    printf("%-16s %8.3f ms  %7.1f TFLOP/s  %.2f cycles per wave-instruction per SIMD @2.4GHz (8 waves/SIMD)\n", name, ms,
           fma_lanes * 2 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (winst / 1024.0));
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_pk_fma_f32");
    return 0;
}
