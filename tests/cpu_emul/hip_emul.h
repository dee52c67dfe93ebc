// TEST INFRASTRUCTURE ONLY: minimal emulation of the HIP execution model on host threads.
// One OS thread per work-item of ONE workgroup at a time (workgroups run sequentially), pthread
// barriers for __syncthreads and for wave-level exchanges (shuffles, MFMA).  Slow; for tiny shapes.
#pragma once
#include <pthread.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
typedef float f32x4 __attribute__((vector_size(16)));
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emul"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace emul {
struct Wave {
    pthread_barrier_t bar;
    float fa[2][64];
    float fb[2][64];
    float fa8[2][64][8];
    float fb8[2][64][8];
};
struct Block {
    pthread_barrier_t bar;
    std::vector<Wave> waves;
};
inline thread_local Block* cur_block = nullptr;
inline thread_local Wave* cur_wave = nullptr;
inline thread_local int lane = 0;
inline thread_local unsigned xcnt = 0;

template <class F>
void launch(dim3 grid, dim3 block, F body);
}  // namespace emul

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline void __syncthreads() { pthread_barrier_wait(&emul::cur_block->bar); }

static inline float __shfl_xor(float v, int mask, int width = 64) {
    (void)width;
    emul::Wave* w = emul::cur_wave;
    unsigned s = emul::xcnt++ & 1u;
    w->fa[s][emul::lane] = v;
    pthread_barrier_wait(&w->bar);
    return w->fa[s][emul::lane ^ mask];
}
static inline float __shfl(float v, int src, int width = 64) {
    (void)width;
    emul::Wave* w = emul::cur_wave;
    unsigned s = emul::xcnt++ & 1u;
    w->fa[s][emul::lane] = v;
    pthread_barrier_wait(&w->bar);
    return w->fa[s][src & 63];
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col=l&15, row=4*(l>>4)+r,
// k-ordered fmaf chain (cdna_hip_programming.md section 3).
static inline f32x4 emul_mfma_16x16x4(float a, float b, f32x4 c) {
    emul::Wave* w = emul::cur_wave;
    unsigned s = emul::xcnt++ & 1u;
    int l = emul::lane;
    w->fa[s][l] = a;
    w->fb[s][l] = b;
    pthread_barrier_wait(&w->bar);
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w->fa[s][k * 16 + row], w->fb[s][k * 16 + col], acc);
        c[r] = acc;
    }
    return c;
}
#define MVS_MFMA_16x16x4(a, b, c) emul_mfma_16x16x4((a), (b), (c))

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products.  Lane l: block = l>>2; A row i = l&3;
// B col j = l&3; D: lane (block, col j = l&3), register r = row i.
static inline f32x4 emul_mfma_4x4x1(float a, float b, f32x4 c) {
    emul::Wave* w = emul::cur_wave;
    unsigned s = emul::xcnt++ & 1u;
    int l = emul::lane;
    w->fa[s][l] = a;
    w->fb[s][l] = b;
    pthread_barrier_wait(&w->bar);
    int blk = l >> 2;
    for (int r = 0; r < 4; ++r) c[r] = fmaf(w->fa[s][blk * 4 + r], w->fb[s][l], c[r]);
    return c;
}
#define MVS_MFMA_4x4x1(a, b, c) emul_mfma_4x4x1((a), (b), (c))
// same with cbsz = 4 / abid: every block takes its A rows from block `abid`
static inline f32x4 emul_mfma_4x4x1_bc(float a, float b, f32x4 c, int abid) {
    emul::Wave* w = emul::cur_wave;
    unsigned s = emul::xcnt++ & 1u;
    int l = emul::lane;
    w->fa[s][l] = a;
    w->fb[s][l] = b;
    pthread_barrier_wait(&w->bar);
    for (int r = 0; r < 4; ++r) c[r] = fmaf(w->fa[s][abid * 4 + r], w->fb[s][l], c[r]);
    return c;
}
static inline int __shfl(int v, int src, int width = 64) {
    float f;
    memcpy(&f, &v, 4);
    f = __shfl(f, src, width);
    memcpy(&v, &f, 4);
    return v;
}
// v_mfma_f32_16x16x32_bf16: A[i = l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][n = l&15]; D: col = l&15, row = 4*(l>>4)+r.
// fp32 accumulation of exact bf16 products (the hardware's internal summation order is not specified; tests use a tolerance).
struct __attribute__((aligned(16))) mvs_bf16x8 { unsigned short v[8]; };
static inline float emul_bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline f32x4 emul_mfma_16x16x32_bf16(mvs_bf16x8 a, mvs_bf16x8 b, f32x4 c) {
    emul::Wave* w = emul::cur_wave;
    unsigned s = emul::xcnt++ & 1u;
    int l = emul::lane;
    for (int j = 0; j < 8; ++j) { w->fa8[s][l][j] = emul_bf2f(a.v[j]); w->fb8[s][l][j] = emul_bf2f(b.v[j]); }
    pthread_barrier_wait(&w->bar);
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg)
            for (int j = 0; j < 8; ++j) acc = fmaf(w->fa8[s][kg * 16 + row][j], w->fb8[s][kg * 16 + col][j], acc);
        c[r] = acc;
    }
    return c;
}
#define MVS_MFMA_16x16x32_BF16(a, b, c) emul_mfma_16x16x32_bf16((a), (b), (c))
static inline unsigned emul_f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
static inline unsigned mvs_cvt_pk_bf16(float lo, float hi) { return emul_f2bf(lo) | (emul_f2bf(hi) << 16); }

// wave-wide votes: every lane of the wave calls them (the kernels keep out-of-range lanes alive for that)
static inline unsigned long long emul_ballot(bool p) {
    emul::Wave* w = emul::cur_wave;
    unsigned s = emul::xcnt++ & 1u;
    w->fa[s][emul::lane] = p ? 1.0f : 0.0f;
    pthread_barrier_wait(&w->bar);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (w->fa[s][l] != 0.0f) m |= 1ull << l;
    return m;
}
static inline void emul_wave_sync() {
    emul::xcnt += 2;   // keeps the double-buffer parity of the exchange slots
    pthread_barrier_wait(&emul::cur_wave->bar);
}
#define MVS_BALLOT(p) emul_ballot(p)
#define MVS_ANY(p) (emul_ballot(p) != 0ull)
#define MVS_FFSLL(m) __builtin_ffsll((long long)(m))
#define MVS_WAVE_SYNC() emul_wave_sync()
#define MVS_UNIFORM_I(x) (x)
#define MVS_SCALAR_LD(ptr, i) ((ptr)[(i)])
#define MVS_OPAQUE_U(x) ((void)0)
#define MVS_QUAD_BCAST_I(v, s) __shfl((int)(v), (emul::lane & ~3) | (s))
#define MVS_QUAD_BCAST_F(v, s) __shfl((float)(v), (emul::lane & ~3) | (s))
#define MVS_SCHED_FENCE() ((void)0)
#define MVS_PIN4(v) ((void)0)
#define MVS_WAVES_PER_SIMD(n)
#define MVS_MIN_WAVES_PER_SIMD(n)
#define MVS_MFMA_4x4x1_BC(a, b, c, abid) emul_mfma_4x4x1_bc((a), (b), (c), (abid))

static inline float atomicAdd(float* addr, float v) {
    unsigned* p = (unsigned*)addr;
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED), nw;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&nw, &f, 4);
    } while (!__atomic_compare_exchange_n(p, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}

namespace emul {
template <class F>
void launch(dim3 grid, dim3 block, F body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    Block blk;
    blk.waves = std::vector<Wave>(nwaves);
    pthread_barrier_init(&blk.bar, nullptr, nthreads);
    for (int w = 0; w < nwaves; ++w) {
        int cnt = nthreads - w * 64;
        if (cnt > 64) cnt = 64;
        pthread_barrier_init(&blk.waves[w].bar, nullptr, cnt);
    }
    std::vector<std::thread> ts;
    ts.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t) {
        ts.emplace_back([&, t]() {
            cur_block = &blk;
            cur_wave = &blk.waves[t / 64];
            lane = t % 64;
            xcnt = 0;
            blockDim = block;
            gridDim = grid;
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = dim3(bx, by, bz);
                        body();
                        pthread_barrier_wait(&blk.bar);
                    }
        });
    }
    for (auto& th : ts) th.join();
    pthread_barrier_destroy(&blk.bar);
    for (int w = 0; w < nwaves; ++w) pthread_barrier_destroy(&blk.waves[w].bar);
}
}  // namespace emul

#define MVS_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emul::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

// LDS-DMA: the emulation copies synchronously (one lane = one thread), so the vmcnt waits are no-ops
#define MVS_DMA4(lds_dst, gbase, voff_bytes) ((void)((lds_dst)[emul::lane] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(gbase) + (voff_bytes))))
#define MVS_WAIT_VMCNT(n) ((void)0)
#define MVS_DMA16(lds_dst, gsrc) memcpy(reinterpret_cast<char*>(lds_dst) + 16 * emul::lane, (gsrc), 16)
#define MVS_LDS_BARRIER() __syncthreads()
#define MVS_LDS_ATOMIC_ADD(ptr, v) ((void)atomicAdd((ptr), (v)))
#define MVS_GLOBAL_ATOMIC_ADD(ptr, v) ((void)atomicAdd((ptr), (v)))
static inline void emul_atomic_add_f64(double* addr, double v) {
    unsigned long long* p = (unsigned long long*)addr;
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED), nw;
    double f;
    do {
        memcpy(&f, &old, 8);
        f += v;
        memcpy(&nw, &f, 8);
    } while (!__atomic_compare_exchange_n(p, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}
#define MVS_GLOBAL_ATOMIC_ADD_F64(ptr, v) emul_atomic_add_f64((ptr), (v))
#define MVS_NT_STORE4(ptr, o) (*reinterpret_cast<float4*>(ptr) = (o))
#define MVS_RCP(x) (1.0f / (x))
static inline int emul_f2i(float x) { return x != x ? 0 : (x >= 2147483647.0f ? 2147483647 : (x <= -2147483648.0f ? (-2147483647 - 1) : (int)x)); }
#define MVS_F2I(x) emul_f2i(x)
