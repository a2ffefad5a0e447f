// dev / measurement tool (not product code): what each ingredient of the split-operand conv kernels' main loop costs the matrix pipe, one at a time.
// Every conv / decoder kernel of this library sits at 0.55 - 0.70 MFMA-pipe utilisation with the same counter profile (PMC: ~0.2 of wave cycles issuing,
// ~0.3 parked, ~0.5 issue-stalled); this burn rebuilds the Winograd kernel's group loop (18 x v_mfma_f32_32x32x16_f16 per wave and group, 512 threads =
// 2 waves per SIMD, one workgroup per CU) from nothing and adds the ingredients back:
//   V0  MFMAs only, operands in registers, ACC accumulators alternating
//   V1  + one ds_read_b128 per MFMA (fresh A / B fragments, conflict-free, two register sets as in unet_wino.hip)
//   V2  + one s_barrier per 18 MFMAs
//   V3  + three global_load_lds_dwordx4 pieces per 18 MFMAs from an L2-resident buffer, counted s_waitcnt vmcnt(2) before the barrier
//   V4  + 48 v_add_f32 per 18 MFMAs (the output transform's share)
// build: hipcc -O3 --offload-arch=gfx950 tools/dev/mfma_lds_burn.hip -o tools/dev/_build/mfma_lds_burn ; run: prints one line per (variant, ACC)
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mf(const uint4 &a, const uint4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int V, int ACC>
__global__ __launch_bounds__(512, 1) void burn(const unsigned char *__restrict__ wsrc, float *__restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[144 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 144 * 1024 / 16; i += 512) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x34003400u);
    __syncthreads();
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + 72 * 1024;
    const unsigned char *rd = smem + lane * 16 + (wave & 3) * 4096;
    const unsigned char *src = wsrc + (size_t)(blockIdx.x & 7) * (3 << 20);     // 3 MB per XCD-ish slice: L2-resident
    const unsigned voff = wave * 1024 + lane * 16;
    f32x16 acc[ACC];
    float tot[48];
#pragma unroll
    for (int a = 0; a < ACC; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
#pragma unroll
    for (int q = 0; q < 48; ++q) tot[q] = 0.f;
    uint4 fa[2][2], fb[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[s][i] = *reinterpret_cast<const uint4 *>(rd + (s * 2 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[s][i] = *reinterpret_cast<const uint4 *>(rd + 16384 + (s * 4 + i) * 1024);
    }
    unsigned off = 0;
    for (int it = 0; it < iters; ++it) {
        if (V >= 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (V >= 2) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int X = st & 1;
            // six MFMAs of one step from register set X
            acc[0 % ACC] = mf(fa[X][1], fb[X][0], acc[0 % ACC]); acc[1 % ACC] = mf(fa[X][1], fb[X][2], acc[1 % ACC]);
            if (V >= 3) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff + off), "s"(src), "s"(lds_ring + st * 8192 + wave * 1024) : "memory");
                off = (off + 8192u) & ((2u << 20) - 1u);
            }
            acc[2 % ACC] = mf(fa[X][0], fb[X][1], acc[2 % ACC]); acc[3 % ACC] = mf(fa[X][0], fb[X][3], acc[3 % ACC]);
            acc[0 % ACC] = mf(fa[X][0], fb[X][0], acc[0 % ACC]); acc[1 % ACC] = mf(fa[X][0], fb[X][2], acc[1 % ACC]);
            if (V >= 1) {                            // the fragments this set is needed for next (two steps ahead), 6 reads = 1 per MFMA
                const int o = ((it * 3 + st) & 7) * 6144;
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[X][i] = *reinterpret_cast<const uint4 *>(rd + o + i * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) fb[X][i] = *reinterpret_cast<const uint4 *>(rd + 16384 + o + i * 1024);
            }
        }
        if (V >= 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { tot[q] += acc[0][q]; tot[16 + q] += acc[1 % ACC][q]; tot[32 + q] -= acc[1 % ACC][q]; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < ACC; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[a][q];
#pragma unroll
    for (int q = 0; q < 48; ++q) s += tot[q];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s == 123.456f) out[blockIdx.x * 512 + tid] = s;
}

template <int V, int ACC>
int run(const unsigned char *w, float *o, int cus, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 4000, grid = cus * 4;
    hipLaunchKernelGGL((burn<V, ACC>), dim3(grid), dim3(512), 0, 0, w, o, 10);
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((burn<V, ACC>), dim3(grid), dim3(512), 0, 0, w, o, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)grid * 8 * iters * 18 * 2.0 * 32 * 32 * 16;
    printf("V%d ACC=%d: %.3f ms  %.0f TFLOP/s executed = %.3f of 2500\n", V, ACC, ms, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 2.5e15);
    fflush(stdout);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    unsigned char *w; float *o;
    CHECK(hipMalloc(&w, 32 << 20)); CHECK(hipMemset(w, 0, 32 << 20)); CHECK(hipMalloc(&o, 64 << 20));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int cus = prop.multiProcessorCount;
    run<0, 2>(w, o, cus, e0, e1); run<0, 4>(w, o, cus, e0, e1);
    run<1, 2>(w, o, cus, e0, e1); run<1, 4>(w, o, cus, e0, e1);
    run<2, 2>(w, o, cus, e0, e1); run<2, 4>(w, o, cus, e0, e1);
    run<3, 2>(w, o, cus, e0, e1); run<3, 4>(w, o, cus, e0, e1);
    run<4, 2>(w, o, cus, e0, e1);
    return 0;
}
