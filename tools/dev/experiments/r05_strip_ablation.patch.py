"""dev-only patch script for tools/dev/build_variant.sh: TIMING-ONLY ablations of conv3d_split_strip_kernel (wrong results on purpose), to read off what the
x-strip kernel's time is made of.  ST_ABL = noload (the slice's 16 row loads per thread replaced by zeros: conversion + LDS stores stay) | nostage (slices > 0 reuse
slice 0's halo: no loads, no conversion) | noepi (no output stores, no statistics) | nobarrier (no per-group hand-over barrier)"""
import os
which = os.environ["ST_ABL"]
p = "unet_split.hip"
s = open(p).read()
a = s.index("void conv3d_split_strip_kernel(SplitArgs p) {")
b = s.index("// ------------------------------------------------------------------------------------------------ 128-wide variant")
k = s[a:b]
if which == "noload":
    k = k.replace("                    raw[it] = *reinterpret_cast<const float4 *>(base0 + (int64_t)((gz * p.H + gy) * p.W + gx) * p.C0 + c0);",
                  "                    raw[it] = make_float4((float)it, 1.f, 2.f, (float)c0);")
elif which == "nostage":
    k = k.replace("        const int c0 = s * SP_KS;\n        {", "        const int c0 = s * SP_KS;\n        if (s == 0) {", 1)
elif which == "noepi":
    k = k.replace("                ob[(q >> 2) * zs_ + (q & 3) * rs] = v;", "                if (v == 123.456f) ob[(q >> 2) * zs_ + (q & 3) * rs] = v;")
    k = k.replace("    if (p.osum) {", "    if (p.osum && n0 < 0) {")
elif which == "nobarrier":
    k = k.replace("            GN_WAIT_VM_LGKM0(0);\n            __builtin_amdgcn_s_barrier();\n            if (G + 1 < ngroups) issue_group(G + 1);",
                  "            GN_WAIT_VM_LGKM0(0);\n            if (G + 1 < ngroups) issue_group(G + 1);")
else:
    raise SystemExit("unknown ST_ABL " + which)
assert k != s[a:b], "pattern not found"
open(p, "w").write(s[:a] + k + s[b:])
