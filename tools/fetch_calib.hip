// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes the conv kernels use
// (MI355X_MICROARCH.md: the counter reports 1/2 of a wide streaming read; other shapes are uncalibrated).
// Each kernel reads the first SEG bytes of every 128-byte line of a 1 GiB buffer (> the 256 MiB Infinity Cache)
// with 16-byte loads, so the useful bytes are N * SEG / 128.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int SEG>
__global__ __launch_bounds__(256) void fetch_calib_kernel(const float4* __restrict__ buf, float* out, size_t nlines) {
    constexpr int L = SEG / 16;                       // lanes per 128-byte line
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float s = 0.f;
    for (size_t i = g; i < nlines * L; i += stride) {
        const size_t line = i / L, part = i % L;
        const float4 v = buf[line * 8 + part];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}

template <int SEG>
static void run(const float4* buf, float* out, size_t nlines) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    fetch_calib_kernel<SEG><<<4096, 256>>>(buf, out, nlines);
    hipEventRecord(e0);
    fetch_calib_kernel<SEG><<<4096, 256>>>(buf, out, nlines);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double useful = (double)nlines * SEG;
    printf("SEG %3d B of each 128 B line: useful %.1f MB, %.3f ms, %.2f TB/s useful\n", SEG, useful / 1e6, ms, useful / ms / 1e9);
}

int main() {
    const size_t bytes = (size_t)1 << 30, nlines = bytes / 128;
    float4* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 256);
    hipMemset(buf, 0, bytes);
    run<128>(buf, out, nlines);
    run<64>(buf, out, nlines);
    run<32>(buf, out, nlines);
    run<16>(buf, out, nlines);
    hipDeviceSynchronize();
    return 0;
}
