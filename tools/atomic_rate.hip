// Micro-benchmark: what an LDS / device-scope fp32 atomic costs on gfx950 as a function of the active lanes and the
// address pattern, next to a non-atomic ds_read/add/ds_write of the same data (decides the flush scheme of K2,
// DESIGN.md section 4).   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_rate.hip -o tools/atomic_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(1))) float glb_float;
#define LDS_ADD(p, v) ((void)__hip_atomic_fetch_add((lds_float*)(p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define GLB_ADD(p, v) ((void)__hip_atomic_fetch_add((glb_float*)(p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))

// MODE: 0 all 64 lanes, consecutive floats      1 lanes 0..7, stride 4 floats (K2 round-1 flush: 4q + j)
//       2 lanes 0..7 consecutive               3 lanes 0..31 consecutive
//       4 all lanes one address                 5 64 lanes, 2 lanes per address
//       6 lanes 0..7: ds_read_b128 + add + ds_write_b128 (non-atomic, 4 floats per lane)
//       7 64 lanes: ds_read_b32 + add + ds_write_b32 (non-atomic)
//       8 lanes 0..3: 2 x (ds_read_b128 + add + ds_write_b128)   (8 channels per lane)
//       9 lanes 0..15 consecutive               10 64 lanes stride 33 floats (conflict-free scatter)
template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[4 * 64 * 40];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 4 * 64 * 40; i += 256) buf[i] = 0.f;
    __syncthreads();
    float* w = buf + wv * 64 * 40;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int o = (j & 3) * 64 * 4 + (j >> 2);   // 16 distinct targets per iteration, like 4 taps x 4 channels
            if (MODE == 0) LDS_ADD(w + ((lane + o) & 2047), v);
            if (MODE == 1) { if (lane < 8) LDS_ADD(w + 4 * lane + o, v); }
            if (MODE == 2) { if (lane < 8) LDS_ADD(w + lane + o, v); }
            if (MODE == 3) { if (lane < 32) LDS_ADD(w + lane + o, v); }
            if (MODE == 4) LDS_ADD(w + o, v);
            if (MODE == 5) LDS_ADD(w + (lane >> 1) + o, v);
            if (MODE == 9) { if (lane < 16) LDS_ADD(w + lane + o, v); }
            if (MODE == 10) LDS_ADD(w + ((lane * 33 + o) & 2047), v);
            if (MODE == 6) {
                if (lane < 8 && (j & 3) == 0) {
                    float4* p = reinterpret_cast<float4*>(w + (j >> 2) * 256 + 4 * lane);
                    float4 a = *p;
                    a.x += v; a.y += v; a.z += v; a.w += v;
                    *p = a;
                }
            }
            if (MODE == 8) {
                if (lane < 4 && (j & 3) == 0) {
                    float4* p = reinterpret_cast<float4*>(w + (j >> 2) * 256 + 4 * lane);
                    float4 a = p[0], b = p[4];
                    a.x += v; a.y += v; a.z += v; a.w += v; b.x += v; b.y += v; b.z += v; b.w += v;
                    p[0] = a; p[4] = b;
                }
            }
            if (MODE == 7) {
                float* p = w + ((lane + o) & 2047);
                *p = *p + v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = buf[tid];
}

template <int MODE>
void run_lds(const char* name, double inst_per_iter) {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    lds_kernel<MODE><<<1024, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    lds_kernel<MODE><<<1024, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 1024 workgroups x 4 waves over 256 CUs: 16 waves per CU share one LDS
    const double inst_per_cu = 16.0 * iters * inst_per_iter;
    printf("LDS %-58s %8.3f ms  %7.1f cycles per wave-instruction per CU @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / inst_per_cu);
    hipFree(out);
}

// MODE: 0 each wave adds to 64 consecutive floats, wave-private rotating region inside a 4 MB buffer
//       1 lanes 0..7 stride 4 floats (+j)        2 scattered: every lane its own 128-byte line
//       3 64 consecutive floats at a pseudo-random texel of a shared 2.6 MB map (contention between workgroups)
//       4 as 3 but float4-granular pattern of K2's window write-out: lanes = 2 texels x 32 channels
template <int MODE>
__global__ __launch_bounds__(256) void glb_kernel(float* buf, int iters, unsigned nfloats) {
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned gw = blockIdx.x * 4 + (tid >> 6);
    unsigned s = gw * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        if (MODE == 0) GLB_ADD(buf + ((gw * 4096u + (it & 63) * 64u + lane) % nfloats), 1.0f);
        if (MODE == 1) { if (lane < 8) { for (int j = 0; j < 4; ++j) GLB_ADD(buf + ((gw * 4096u + (it & 63) * 64u + 4 * lane + j) % nfloats), 1.0f); } }
        if (MODE == 2) GLB_ADD(buf + (((s >> 8) % (nfloats / 32u)) * 32u + (lane * 32u * 997u) % nfloats) % nfloats, 1.0f);
        if (MODE == 3) GLB_ADD(buf + (((s >> 8) % (nfloats / 64u)) * 64u + lane), 1.0f);
    }
}

template <int MODE>
void run_glb(const char* name, double lanes_per_iter) {
    const unsigned nfloats = 160 * 128 * 32;   // one config-2 feature map
    float* buf;
    hipMalloc(&buf, 4u << 20 << 2);
    hipMemset(buf, 0, 4u << 20 << 2);
    const int iters = 2000, blocks = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    glb_kernel<MODE><<<blocks, 256>>>(buf, 10, nfloats);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    glb_kernel<MODE><<<blocks, 256>>>(buf, iters, nfloats);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double atomics = (double)blocks * 4 * iters * lanes_per_iter;
    printf("GLB %-58s %8.3f ms  %8.1f G float-atomics/s\n", name, ms, atomics / (ms * 1e-3) / 1e9);
    hipFree(buf);
}

int main() {
    run_lds<0>("ds_add_f32, 64 lanes consecutive", 16);
    run_lds<3>("ds_add_f32, 32 lanes consecutive", 16);
    run_lds<9>("ds_add_f32, 16 lanes consecutive", 16);
    run_lds<2>("ds_add_f32, 8 lanes consecutive", 16);
    run_lds<1>("ds_add_f32, 8 lanes stride 4 floats (round-1 flush)", 16);
    run_lds<10>("ds_add_f32, 64 lanes stride 33 floats", 16);
    run_lds<5>("ds_add_f32, 64 lanes, 2 lanes per address", 16);
    run_lds<4>("ds_add_f32, 64 lanes one address", 16);
    run_lds<6>("b128 read+add+write, 8 lanes (per 4-float group)", 4);
    run_lds<8>("2 x b128 read+add+write, 4 lanes (per 8-float group)", 4);
    run_lds<7>("b32 read+add+write, 64 lanes consecutive", 16);
    run_glb<0>("global_atomic_add_f32, 64 consecutive floats per wave", 64);
    run_glb<1>("global_atomic_add_f32, 8 lanes x 4 (stride-4 pattern)", 32);
    run_glb<2>("global_atomic_add_f32, scattered lines", 64);
    run_glb<3>("global_atomic_add_f32, 64 consecutive at random texels", 64);
    return 0;
}
