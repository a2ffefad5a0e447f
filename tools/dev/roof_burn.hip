// dev / measurement tool (not product code): measured ceilings for the members of the path that are bound by neither HBM nor the matrix
// cores -- the fp64 vector rate (Gaussian gradient magnitude and marching cubes compute in fp64, scipy's / scikit-image's arithmetic) and the
// LDS read rate.  bench.py runs it once (hbm_members.roofs) so that "not HBM bound" comes with a number measured on the same box.
//   build: hipcc -O3 --offload-arch=gfx950 tools/dev/roof_burn.hip -o tools/dev/_build/roof_burn     (__graft_entry__.build() does it)
//   run:   tools/dev/_build/roof_burn  ->  one JSON line {"fp64_fma_tflops": ..., "lds_read_TBs": ..., "sclk_note": ...}
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

// 8 independent fp64 FMA chains per thread: enough ILP to cover the pipe latency at 4+ waves per SIMD
__global__ __launch_bounds__(256) void fp64_burn(double *out, int iters, double a, double b) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (double)(threadIdx.x + i) * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __fma_rn(v[i], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456) out[blockIdx.x * blockDim.x + threadIdx.x] = s;      // never true: keeps the chains alive
}

// conflict-free ds_read_b128 stream: every lane reads its own 16-byte column of a 16 KB window, 8 reads in flight
__global__ __launch_bounds__(256) void lds_burn(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float4 buf[4096];               // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    float4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    int base = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = buf[(base + i * 256) & 4095];
            acc[i].x += t.x; acc[i].y += t.y; acc[i].z += t.z; acc[i].w += t.w;
        }
        base += 64;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *d64; float *d32;
    CHECK(hipMalloc(&d64, 1 << 20)); CHECK(hipMalloc(&d32, 1 << 20));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = cus * 8;                                                // 8 workgroups x 4 waves per CU = 8 waves per SIMD
    float ms;
    // fp64
    const int it64 = 4000;
    hipLaunchKernelGGL(fp64_burn, dim3(grid), dim3(256), 0, 0, d64, 10, 1.0000001, 1e-9);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); 
    hipLaunchKernelGGL(fp64_burn, dim3(grid), dim3(256), 0, 0, d64, it64, 1.0000001, 1e-9);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double fma64 = (double)grid * 256 * it64 * 64.0;                   // FMAs
    const double tf64 = 2.0 * fma64 / (ms * 1e-3) / 1e12;
    // LDS (2 workgroups per CU: 64 KB each)
    const int itl = 20000, gridl = cus * 2;
    hipLaunchKernelGGL(lds_burn, dim3(gridl), dim3(256), 0, 0, d32, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(lds_burn, dim3(gridl), dim3(256), 0, 0, d32, itl);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ldsb = (double)gridl * 256 * itl * 8 * 16.0;
    const double tbs = ldsb / (ms * 1e-3) / 1e12;
    printf("{\"fp64_fma_tflops\": %.2f, \"lds_read_TBs\": %.2f, \"compute_units\": %d, \"what\": \"fp64: 8 independent v_fma_f64 chains per thread, 8 waves per SIMD; LDS: conflict-free ds_read_b128, 8 in flight per lane, 2 waves per SIMD\"}\n",
           tf64, tbs, cus);
    return 0;
}
