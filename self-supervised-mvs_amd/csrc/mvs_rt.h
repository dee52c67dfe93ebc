// Runtime glue shared by every kernel file.
//  * hipcc build (the product): plain HIP for gfx950.
//  * -DMVS_CPU_EMUL (tests/cpu_emul only): the same kernel sources are compiled with g++ against a
//    tiny pthread-based emulation of the HIP execution model so kernel *logic* (index maps, MFMA
//    fragment layouts, reductions) can be checked in the GPU-less build container.  The emulation
//    library is test infrastructure; the product never loads it.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(MVS_CPU_EMUL)
#include "hip_emul.h"
#else
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MVS_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
#define MVS_MFMA_16x16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MVS_RCP(x) __builtin_amdgcn_rcpf(x)
// value of quad lane s (0..3) broadcast to the 4 lanes of its quad: one DPP move (quad_perm [s,s,s,s]); the control word is
// an immediate, so s must be known after unrolling (the switch folds)
static __device__ __forceinline__ int mvs_quad_bcast_i(int v, int s) {
    switch (s) {
        case 0: return __builtin_amdgcn_update_dpp(0, v, 0x00, 0xf, 0xf, false);
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x55, 0xf, 0xf, false);
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xf, 0xf, false);
        default: return __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xf, 0xf, false);
    }
}
#define MVS_QUAD_BCAST_I(v, s) mvs_quad_bcast_i((v), (s))
#define MVS_QUAD_BCAST_F(v, s) __int_as_float(mvs_quad_bcast_i(__float_as_int(v), (s)))
#define MVS_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))   // register budget 512 / n per wave
#define MVS_MIN_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n)))   // at least n waves per SIMD: caps the register allocation
#define MVS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   // the instruction scheduler moves nothing across this point
#define MVS_F2I(x) __float2int_rz(x)   // v_cvt_i32_f32: saturating
// "these four registers are needed now": the compiler places the s_waitcnt for the loads that produce them here instead of at their
// first arithmetic use (used to keep a conditional gather's wait inside the conditional block)
#define MVS_PIN4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
// wave-wide votes (every lane of the wave must reach them) and a compiler-level ordering point for a wave's own LDS
// traffic (the DS queue of a wave is in order in hardware; this only stops the compiler from moving accesses across it)
#define MVS_BALLOT(p) ((unsigned long long)__ballot(p))
#define MVS_ANY(p) (__any(p) != 0)
#define MVS_FFSLL(m) __ffsll((unsigned long long)(m))
#define MVS_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#define MVS_UNIFORM_I(x) __builtin_amdgcn_readfirstlane(x)   // a value known to be the same in every lane -> SGPR
// element i of a wave-uniform read-only float table through the SCALAR cache (s_load_dword into an SGPR, lgkmcnt): a load through a plain
// pointer is a VECTOR load in every kernel that also stores (nothing is provably read-only for hipcc) and sits in the in-order vmcnt queue
typedef const __attribute__((address_space(4))) float mvs_const_float;
#define MVS_SCALAR_LD(ptr, i) (((mvs_const_float*)(ptr))[(i)])
// "this value is produced HERE": stops loop-invariant code motion from hoisting what is computed from it (a 64-bit address needed on
// a rare path would otherwise be formed once in front of the loop and live -- or be spilled -- across it)
#define MVS_OPAQUE_U(x) asm volatile("" : "+v"(x))
#define MVS_MFMA_4x4x1(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
// cbsz = 4: the A operand of block `abid` (lanes 4*abid .. 4*abid+3) is broadcast to all 16 blocks
#define MVS_MFMA_4x4x1_BC(a, b, c, abid) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, (abid), 0)
// bf16 MFMA (inference path): 8 bf16 per lane for A and B, k = 8*(lane>>4)+j for both; D col = lane&15, row = 4*(lane>>4)+r
typedef __bf16 mvs_bf16x8 __attribute__((ext_vector_type(8)));
#define MVS_MFMA_16x16x32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// two fp32 -> packed bf16 pair (lo in bits 0..15), round to nearest even: v_cvt_pk_bf16_f32
typedef __bf16 mvs_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mvs_f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ unsigned mvs_cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((mvs_f32x2){lo, hi}, mvs_bf16x2));
}
#define MVS_NT_STORE4(ptr, o) \
    __builtin_nontemporal_store((f32x4){(o).x, (o).y, (o).z, (o).w}, reinterpret_cast<f32x4*>(ptr))
// Address-space-explicit float atomics: a generic (flat) pointer makes hipcc emit flat_atomic_add_f32 even
// for LDS, which is an order of magnitude slower than ds_add_f32.
typedef __attribute__((address_space(3))) float mvs_lds_float;
typedef __attribute__((address_space(1))) float mvs_global_float;
// LDS-DMA (gfx950 global_load_lds): every lane's 4 bytes at gbase + voff_bytes land in LDS at lds_dst + 4 * lane -- no register
// destination; completion is counted by vmcnt like any vector load, in order.  Issued as inline assembly ON PURPOSE: for the
// builtin, hipcc puts an s_waitcnt vmcnt(0) in front of every later LDS read (it cannot tell which reads alias the DMA target),
// which would drain a prefetch ring on every plane.  The kernels that use it place the waits themselves (MVS_WAIT_VMCNT) and
// state the invariant that makes the count sufficient.  lds_dst must be wave-uniform (it travels in M0).
static __device__ __forceinline__ void mvs_dma4(float* lds_dst, const float* gbase, unsigned voff_bytes) {
    // wave-uniform by contract; the compiler cannot always prove it (an address derived from the wave index): readfirstlane
    // moves the values into scalar registers (it folds away when they already are)
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(mvs_lds_float*)lds_dst);
    const unsigned long long ga = (unsigned long long)(size_t)gbase;
    const unsigned ga_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32));   // the builtin returns int:
    const unsigned ga_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga);           // widen as UNSIGNED halves
    gbase = (const float*)(size_t)(((unsigned long long)ga_hi << 32) | (unsigned long long)ga_lo);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(m0v), "v"(voff_bytes), "s"(gbase) : "memory", "m0");
}
#define MVS_DMA4(lds_dst, gbase, voff_bytes) mvs_dma4((lds_dst), (gbase), (voff_bytes))
// 16 bytes per lane (global_load_lds_dwordx4, gfx950): lane l's 16 bytes at its own pointer gsrc land in LDS at lds_dst + 16 * l --
// one wave instruction moves 1 KiB.  lds_dst wave-uniform (it travels in M0; M0 is saved and restored inside the statement, the
// compiler keeps values of its own there).  Not counted by hipcc's s_waitcnt bookkeeping: the caller waits (MVS_WAIT_VMCNT).
static __device__ __forceinline__ void mvs_dma16(float* lds_dst, const float* gsrc) {
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(mvs_lds_float*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(m0v)
                 : "memory");
}
#define MVS_DMA16(lds_dst, gsrc) mvs_dma16((lds_dst), (gsrc))
// workgroup barrier WITHOUT the release fence of __syncthreads() (which drains every outstanding global store: s_waitcnt vmcnt(0)):
// orders LDS traffic only -- the caller has waited for its own LDS operations / LDS-DMA where that matters
#define MVS_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// wait until at most n vector-memory operations of this wave are outstanding (n: compile-time constant <= 63)
#define MVS_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define MVS_LDS_ATOMIC_ADD(ptr, v) \
    ((void)__hip_atomic_fetch_add((mvs_lds_float*)(ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define MVS_GLOBAL_ATOMIC_ADD(ptr, v) \
    ((void)__hip_atomic_fetch_add((mvs_global_float*)(ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
// fp64 accumulation slots of the BatchNorm statistics (global_atomic_add_f64, no return value): the order in which workgroups
// arrive changes the sum by ~1e-16 relative, far below the fp32 values derived from it
typedef __attribute__((address_space(1))) double mvs_global_double;
#define MVS_GLOBAL_ATOMIC_ADD_F64(ptr, v) \
    ((void)__hip_atomic_fetch_add((mvs_global_double*)(ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#endif

#define MVS_WAVE 64
#define MVS_MAX_SRC 10

// ---- error reporting (no exceptions across the C boundary) ---------------------------------
#define MVS_OK 0
#define MVS_ERR_SHAPE (-1)
#define MVS_ERR_UNSUPPORTED (-2)
#define MVS_ERR_LAUNCH (-3)
#define MVS_ERR_NULL (-4)

void mvs_set_error(const char* fmt, ...);
int mvs_check_launch(const char* what);

#define MVS_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            mvs_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

static inline int mvs_cdiv(int a, int b) { return (a + b - 1) / b; }
