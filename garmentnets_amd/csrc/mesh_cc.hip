// mesh_cc.hip -- gn_mesh_largest_component: the connected-component filter of the reference's hole removal.
//
// Reference: eval.py:497-503 and :536-545 -- after delete_invalid_verts,
//     adj_mat = igl.adjacency_matrix(faces); num_cc, cc_idxs, cc_sizes = igl.connected_components(adj_mat)
//     is_cc_vert = (cc_idxs == np.argmax(cc_sizes))
// followed by delete_invalid_verts(verts, faces, is_cc_vert) (gn_mesh_compact).  libigl numbers the components in the order of their
// lowest vertex index (breadth-first from vertex 0 upwards) and np.argmax takes the first maximum: of several largest components the one
// that contains the lowest vertex index wins.  Here: lock-free union-find over the 3 F mesh edges, always hooking the LARGER root under the
// smaller one -- whatever the interleaving, a component's final root is its lowest vertex index, so labels, sizes and the winner are
// deterministic.  HBM-bound integer work: 3 passes over the faces / vertices.
#include "common.h"

__device__ __forceinline__ int cc_find(int *parent, int x) {
    // path halving; concurrent hooks only ever replace a root by a smaller index, so the walk terminates at a (momentary) root
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const int g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != p) __hip_atomic_store(&parent[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = g;
    }
    return x;
}

__device__ __forceinline__ void cc_union(int *parent, int a, int b) {
    while (true) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }                    // a < b: hook b under a
        const int old = atomicCAS(&parent[b], b, a);
        if (old == b) return;
        // b stopped being a root meanwhile: retry from the new state
    }
}

__global__ void cc_init_kernel(int *__restrict__ parent, int *__restrict__ count, int64_t V, unsigned long long *__restrict__ best, int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) { parent[i] = (int)i; count[i] = 0; }
    if (i == 0) { *best = 0ull; *bad = 0; }
}

__global__ void cc_hook_kernel(const int32_t *__restrict__ faces, int64_t F, int64_t V, int *__restrict__ parent, int *__restrict__ bad) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
    if (a < 0 || b < 0 || c < 0 || a >= V || b >= V || c >= V) { *bad = 1; return; }
    cc_union(parent, a, b);
    cc_union(parent, b, c);                                                // (c, a) is implied
}

__global__ void cc_label_kernel(int *__restrict__ parent, int *__restrict__ count, int64_t V) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const int r = cc_find(parent, (int)i);
    atomicAdd(&count[r], 1);
}

__global__ void cc_best_kernel(const int *__restrict__ parent, const int *__restrict__ count, int64_t V, unsigned long long *__restrict__ best, int64_t *__restrict__ ncomp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long key = 0ull;
    int is_root = 0;
    if (i < V && parent[i] == (int)i) {                                    // roots only (count is zero elsewhere)
        key = ((unsigned long long)(unsigned)count[i] << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);   // size first, then the LOWEST index
        is_root = 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
        is_root += __shfl_xor(is_root, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (key) atomicMax(best, key);
        if (is_root) atomicAdd(reinterpret_cast<unsigned long long *>(ncomp), (unsigned long long)is_root);
    }
}

__global__ void cc_mask_kernel(int *__restrict__ parent, int64_t V, const unsigned long long *__restrict__ best, int32_t *__restrict__ label,
                               unsigned char *__restrict__ mask, int64_t *__restrict__ info) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long k = *best;
    const int win = (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));
    if (i == 0) { info[1] = (int64_t)(k >> 32); info[2] = k ? win : -1; }
    if (i >= V) return;
    int x = (int)i, p = parent[x];                                         // (read-only walk: path halving leaves chains longer than one hop)
    while (p != x) { x = p; p = parent[x]; }
    if (label) label[i] = x;
    mask[i] = (x == win) ? 1 : 0;
}

extern "C" size_t gn_mesh_largest_component_workspace_bytes(int64_t V) {
    return V < 0 ? 0 : (size_t)V * 8 + 64;                                 // parent + count, best key, bad flag
}

extern "C" int gn_mesh_largest_component(const int32_t *faces, int64_t F, int64_t V, void *ws, size_t ws_bytes, unsigned char *mask, int32_t *label,
                                         int64_t *info, void *stream) {
    GN_REQUIRE(F >= 0 && V >= 0 && V < ((int64_t)1 << 31), "gn_mesh_largest_component: bad sizes");
    GN_REQUIRE(info != nullptr, "gn_mesh_largest_component: info (device int64[4]) is required");
    hipStream_t st = gn_stream(stream);
    GN_HIP(hipMemsetAsync(info, 0, 4 * sizeof(int64_t), st), "gn_mesh_largest_component");
    if (V == 0) {
        if (F > 0) {   // faces over an empty vertex set: every index is out of range (gn_mesh_compact reports the same case the same way)
            const int64_t one = 1;
            GN_HIP(hipMemcpyAsync(info + 3, &one, sizeof(one), hipMemcpyHostToDevice, st), "gn_mesh_largest_component");
            GN_HIP(hipStreamSynchronize(st), "gn_mesh_largest_component");   // `one` lives on this frame
        }
        return GN_OK;
    }
    GN_REQUIRE((faces || F == 0) && mask && ws, "gn_mesh_largest_component: null pointer");
    GN_REQUIRE(ws_bytes >= gn_mesh_largest_component_workspace_bytes(V), "gn_mesh_largest_component: workspace too small");
    int *parent = (int *)ws, *count = parent + V;
    unsigned long long *best = (unsigned long long *)(((uintptr_t)(count + V) + 15) & ~(uintptr_t)15);
    int *bad = (int *)(best + 1);
    const unsigned gv = (unsigned)gn_cdiv(V, 256), gf = (unsigned)gn_cdiv(F, 256);
    hipLaunchKernelGGL(cc_init_kernel, dim3(gv), dim3(256), 0, st, parent, count, V, best, bad);
    if (F > 0) hipLaunchKernelGGL(cc_hook_kernel, dim3(gf), dim3(256), 0, st, faces, F, V, parent, bad);
    hipLaunchKernelGGL(cc_label_kernel, dim3(gv), dim3(256), 0, st, parent, count, V);
    hipLaunchKernelGGL(cc_best_kernel, dim3(gv), dim3(256), 0, st, parent, count, V, best, info);
    hipLaunchKernelGGL(cc_mask_kernel, dim3(gv), dim3(256), 0, st, parent, V, best, label, mask, info);
    GN_HIP(hipMemcpyAsync(info + 3, bad, sizeof(int), hipMemcpyDeviceToDevice, st), "gn_mesh_largest_component");
    GN_LAUNCH_CHECK("gn_mesh_largest_component");
    return GN_OK;
}
