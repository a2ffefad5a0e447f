s=open('upconv.hip').read()
# 4 classes per workgroup (256 threads), two workgroups per tile
s=s.replace('''template <int NT, bool F16>
__global__ __launch_bounds__(512, 1) void upconv_partial_kernel(UpArgs p) {
    constexpr int TZC = NT == 1 ? 2 : 1, NF = 2 * TZC, HZ = TZC + 2, HVOX = HZ * 100;
    using HL = UpHalo<HZ>;
    constexpr int NIT = (HVOX * 4 + 511) / 512;     // float4 row loads per thread per slice (4 / 3)''','''template <int NT, bool F16>
__global__ __launch_bounds__(256, 2) void upconv_partial_kernel(UpArgs p) {
    constexpr int TZC = NT == 1 ? 2 : 1, NF = 2 * TZC, HZ = TZC + 2, HVOX = HZ * 100;
    using HL = UpHalo<HZ>;
    constexpr int NTHR = 256;
    constexpr int NIT = (HVOX * 4 + NTHR - 1) / NTHR;     // float4 row loads per thread per slice (7 / 5)''')
s=s.replace('''    const int cls = __builtin_amdgcn_readfirstlane(tid >> 6), pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;''','''    const int cls = __builtin_amdgcn_readfirstlane(tid >> 6) + 4 * (int)((gridDim.x & 15u) == 0 ? (blockIdx.x >> 3) & 1u : blockIdx.x & 1u), pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;''')
s=s.replace('''    const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3, qx = nblk >> 3, rx = nblk & 7u;''','''    const unsigned nblk = gridDim.x >> 1, bid = (gridDim.x & 15u) == 0 ? ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7u) : blockIdx.x >> 1, xcd = bid & 7u, jx = bid >> 3, qx = nblk >> 3, rx = nblk & 7u;''')
s=s.replace('''    for (int i = tid; i < p.C1; i += 512) {''','''    for (int i = tid; i < p.C1; i += NTHR) {''')
s=s.replace('''            const int idx = tid + it * 512;
            const int hv = (idx < HVOX * 4 ? idx : HVOX * 4 - 1) >> 2;''','''            const int idx = tid + it * NTHR;
            const int hv = (idx < HVOX * 4 ? idx : HVOX * 4 - 1) >> 2;''')
s=s.replace('''        const int idx = tid + it * 512;
        if (idx < HVOX * 4) {''','''        const int idx = tid + it * NTHR;
        if (idx < HVOX * 4) {''')
s=s.replace('''            if (more && tap >= 2 && tap < 2 + NIT) convert_row(tap - 2, s + 1, (s + 1) & 1);''','''            if (more && tap >= 1 && tap < 1 + NIT) convert_row(tap - 1, s + 1, (s + 1) & 1);''')
s=s.replace('''        const unsigned g = (unsigned)(gn_cdiv(Dc, 1) * p.tiles_y * p.tiles_x * (Cout / 64));
        if (f16) hipLaunchKernelGGL((upconv_partial_kernel<2, true>), dim3(g, B), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((upconv_partial_kernel<2, false>), dim3(g, B), dim3(512), 0, st, p);''','''        const unsigned g = 2 * (unsigned)(gn_cdiv(Dc, 1) * p.tiles_y * p.tiles_x * (Cout / 64));
        if (f16) hipLaunchKernelGGL((upconv_partial_kernel<2, true>), dim3(g, B), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((upconv_partial_kernel<2, false>), dim3(g, B), dim3(256), 0, st, p);''')
s=s.replace('''        const unsigned g = (unsigned)(gn_cdiv(Dc, 2) * p.tiles_y * p.tiles_x * (Cout / 32));
        if (f16) hipLaunchKernelGGL((upconv_partial_kernel<1, true>), dim3(g, B), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((upconv_partial_kernel<1, false>), dim3(g, B), dim3(512), 0, st, p);''','''        const unsigned g = 2 * (unsigned)(gn_cdiv(Dc, 2) * p.tiles_y * p.tiles_x * (Cout / 32));
        if (f16) hipLaunchKernelGGL((upconv_partial_kernel<1, true>), dim3(g, B), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((upconv_partial_kernel<1, false>), dim3(g, B), dim3(256), 0, st, p);''')
assert s.count('NTHR') >= 5 and 'dim3(512)' not in s
open('upconv.hip','w').write(s)
