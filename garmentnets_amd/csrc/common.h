// common.h -- shared helpers for the gfx950 kernels of libgarmentnets_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/garmentnets_hip.h"

#define GN_WAVE 64

// This library is written for ONE target: gfx950 (MI355X, CDNA4) with the ROCm 7 toolchain (clang >= 20).  The kernels use its instructions directly
// (v_mfma_f32_32x32x16_f16, global_load_lds_dwordx4, v_fma_mix_f32, v_maximum3_f32 through __builtin_elementwise_maximum) and there is no other code
// path: the Makefile's HIPCC / ARCH overrides exist for a differently installed ROCm, not for another architecture -- say so at compile time.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libgarmentnets_hip.so is gfx950-only (MI355X): build with --offload-arch=gfx950"
#endif
#if !defined(__has_builtin) || !__has_builtin(__builtin_elementwise_maximum)
#error "libgarmentnets_hip.so needs the ROCm 7 hipcc (clang >= 20: __builtin_elementwise_maximum)"
#endif

void gn_set_error(const char *fmt, ...);

#define GN_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            gn_set_error(__VA_ARGS__);   \
            return GN_EINVAL;            \
        }                                \
    } while (0)

#define GN_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            gn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return GN_ELAUNCH;                                                       \
        }                                                                            \
    } while (0)

#define GN_HIP(call, name)                                                           \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            gn_set_error("%s: %s", name, hipGetErrorString(e__));                    \
            return GN_ELAUNCH;                                                       \
        }                                                                            \
    } while (0)

// the launch stream of a C-ABI call.  A non-null stream also selects the device: kernels, memsets and function attributes of this
// call go to the device the stream lives on (hipSetDevice when it differs from the thread's current one), so a caller working on
// cuda:N only has to pass a stream of that device (ops._stream takes it from the tensors).  The null stream means "current device".
hipStream_t gn_stream(void *s);
// name of the kernel variant the last gn_conv3d_* call of this thread launched (bench.py labels its roofline with it)
void gn_note_kernel(const char *name);
static inline int64_t gn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// zero a pair of fp64 statistics buffers (sum, sumsq: n values each) that a kernel accumulates into with atomics: ONE fill when the caller laid them out back
// to back (ops._stats_buffers does) -- a fill is ~5 us of stream time, and a 32^3 UNet layer is not much longer than that
static inline hipError_t gn_zero_stats(double *sum, double *sumsq, size_t n, hipStream_t st) {
    if (sumsq == sum + n) return hipMemsetAsync(sum, 0, 2 * n * sizeof(double), st);
    hipError_t e = hipMemsetAsync(sum, 0, n * sizeof(double), st);
    return e != hipSuccess ? e : hipMemsetAsync(sumsq, 0, n * sizeof(double), st);
}

// squared distance in the pinned operation order ((dx*dx + dy*dy) + dz*dz), fp32, no FMA contraction
// (the library is compiled with -ffp-contract=off; the intrinsics make it explicit).
__device__ __forceinline__ float gn_sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    float s = __fmul_rn(dx, dx);
    s = __fadd_rn(s, __fmul_rn(dy, dy));
    s = __fadd_rn(s, __fmul_rn(dz, dz));
    return s;
}

// 16 bytes per lane global -> LDS DMA (wave-uniform LDS destination `lds_addr` + 16 * lane), issued from inline asm ON PURPOSE:
// hipcc (ROCm 7.2) guards every ds_read that follows a __builtin_amdgcn_global_load_lds with s_waitcnt vmcnt(0) (it cannot prove
// the read does not alias the DMA's destination), which drains a multi-stage ring at every step.  Users make slot reuse safe by
// hand (counted s_waitcnt vmcnt(N) + s_barrier) and must not use m0 otherwise.
__device__ __forceinline__ void gn_glds16(const void *g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}
// s_waitcnt vmcnt(N) lgkmcnt(0)   (gfx9 immediate: vmcnt[3:0] | expcnt[6:4] = 7 (no wait) | lgkmcnt[11:8] | vmcnt_hi[15:14])
// s_waitcnt vmcnt(N) alone (lgkmcnt field = 15: no wait)
#define GN_WAIT_VM_ONLY(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | 0x70 | 0xF00 | (((N) >> 4) << 14))
#define GN_WAIT_VM_LGKM0(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | 0x70 | (((N) >> 4) << 14))

// ReLU with torch's NaN behaviour (relu(NaN) = NaN).  fmaxf(v, 0) would return 0 for a NaN and turn an upstream overflow
// (e.g. an activation beyond the fp16 range in the split-operand kernels) into a silently wrong finite result.
// IEEE 754-2019 maximum(v, 0) -- NaN-propagating -- is ONE gfx950 instruction (v_maximum3_f32 v, v, 0, 0); the C form `v < 0 ? 0 : v`
// compiled to v_cmp_ngt_f32 + 2 hazard wait states on vcc + v_cndmask_b32 per value (the largest VALU item of the decoder MLPs'
// layer hand-over).  Same values; the only bit that can differ is the sign of a zero (maximum(-0, +0) = +0, the C form kept -0).
__device__ __forceinline__ float gn_relu(float v) { return __builtin_elementwise_maximum(v, 0.f); }

// Exact residual of an fp32 value r against one half of a packed fp16 pair h2 in ONE instruction: v_fma_mix_f32 reads the fp16 half in place,
// fma(f32(h), -1, r) = r - f32(h) with one rounding -- of a value that IS representable when h = fp16_rn(r) or any fp16 within the split's range (the
// difference has at most 13 significant bits), so the result is bit-identical to v_cvt_f32_f16 + v_sub_f32 (hipcc's selection for the C
// expression: two instructions per value; the plane split is the largest VALU item of every f16x2 kernel).  Not volatile: schedulable, removable.
__device__ __forceinline__ float gn_resid_lo(unsigned h2, float r) {
    float o;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(o) : "v"(h2), "v"(r));
    return o;
}
__device__ __forceinline__ float gn_resid_hi(unsigned h2, float r) {
    float o;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(o) : "v"(h2), "v"(r));
    return o;
}

__device__ __forceinline__ int gn_lane() { return threadIdx.x & 63; }
