"""timing-only ablations of csrc/unet_wino32.hip (results are WRONG by construction; tools/dev/build_variant.sh <name> this-file with ABL=<which>):
noconv  the next slice's transform / split / halo stores removed (the halo keeps the prologue's values)
norows  the row loads removed (and with them the affine)
nostore the epilogue's output stores and statistics removed
nodma   the weight DMA of the steady state removed (the ring keeps the prologue's three groups)
nobar / noread / noflush   the hand-over barrier / the steady state's fragment reads / three of four output transforms removed"""
import os
abl = os.environ["ABL"].split(",")
s = open("unet_wino32.hip").read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
if "noconv" in abl:
    for g, j, k in ((3, 0, "0, 2"), (4, 0, "2, 4"), (5, 1, "0, 2"), (6, 1, "2, 4"), (7, 2, "0, 2"), (8, 2, "2, 4"), (9, 3, "0, 2"), (10, 3, "2, 4")):
        if g == 3:
            rep("if (g == 3) { affine_load(sn); affine_math(0, NIT); convert(0, nslo[0], 0, 2); }", "")
        else:
            rep(f"                if (g == {g}) convert({j}, nslo[{j}], {k});\n", "")
if "norows" in abl:
    rep("                if (g == 0) issue_rows(sn);             // always (uniform wait counts)\n", "")
    rep("if ((G) == 1 || (G) == 2) GN_WAIT_VM_LGKM0(3 + NIT);", "if ((G) == 1 || (G) == 2) GN_WAIT_VM_LGKM0(3);")
if "nostore" in abl:
    rep('                    asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(vo[q & 3]), "v"(v), "s"(ob) : "memory");\n', '                    asm volatile("" :: "v"(v), "v"(vo[q & 3]), "s"(ob));\n')
if "nodma" in abl:
    rep("                W32_ISSUE_GROUP((g + 3) % RING);                                   // group g + 3 -> the slot group g - 1 vacated\n", "")
if "rowsl2" in abl:      # every tile stages the sample-0 corner tile's rows: the same loads, all of them L2 hits
    rep("        const int gz = z0_ + hz - 1, gy = y0_ + hy - 1;\n        const bool rowin", "        z0_ = 8; y0_ = 8; x0_ = 8;\n        const int gz = z0_ + hz - 1, gy = y0_ + hy - 1;\n        const bool rowin")
    rep("            base0 = p.src0 + (int64_t)b * p.D * p.H * p.W * p.C0;", "            base0 = p.src0;")
if "samehalf" in abl:    # odd slices re-read the even slice's 64 bytes of every voxel: does the L2 keep a line from one slice to the next?
    rep("const unsigned cb4 = (unsigned)sl * (SP_KS * 4u), vs", "const unsigned cb4 = (unsigned)(sl & ~1) * (SP_KS * 4u), vs")
if "contig" in abl:      # the loads of a channel-BLOCKED input [slice][z][y][x][16]: same count, same rows, contiguous 64-byte pieces along x
    rep("const unsigned cb4 = (unsigned)sl * (SP_KS * 4u), vs = (unsigned)p.C0 * 4u;", "const unsigned vs = 64u, cb4 = (unsigned)sl * (unsigned)(p.D * p.H * p.W) * 64u;")
    rep("        const unsigned vs = (unsigned)p.C0 * 4u;\n        voff1 = rowin ? ((unsigned)((gz * p.H + gy) * p.W + x0_) * (unsigned)p.C0 + (unsigned)cq) * 4u : (unsigned)cq * 4u;",
        "        const unsigned vs = 64u;\n        voff1 = rowin ? ((unsigned)((gz * p.H + gy) * p.W + x0_) * 16u + (unsigned)cq) * 4u : (unsigned)cq * 4u;")
for hint in ("nt", "sc1", "sc0 sc1 nt"):     # cache-policy hints on the row loads (real variants: results stay correct)
    if "rows_" + hint.replace(" ", "_") in abl:
        rep('asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw[k])', 'asm volatile("global_load_dwordx4 %0, %1, %2 ' + hint + '" : "=v"(raw[k])')
if "acc2" in abl:        # a REAL variant: the cross products (1,0), (0,1) and the main product (0,0) in accumulators of their own (no MFMA waits for the one before it)
    import re
    rep("    f32x16s acc, tot[2];", "    f32x16s acc, acch, tot[2];")
    s, n1 = re.subn(r"acc = mfma16<F16>\(fa\[SET\]\[0\], fb\[SET\]\[0\], acc\);  ", "acch = mfma16<F16>(fa[SET][0], fb[SET][0], acch);", s)
    s, n2 = re.subn(r"const float m = acc\[q\];                                     ", "const float m = __fadd_rn(acc[q], acch[q]); acch[q] = 0.f;", s)
    assert n1 == 1 and n2 == 1, (n1, n2)
    rep("        for (int q = 0; q < 16; ++q) { acc[q] = 0.f; tot[0][q] = 0.f; tot[1][q] = 0.f; }", "        for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acch[q] = 0.f; tot[0][q] = 0.f; tot[1][q] = 0.f; }")
if "pairload" in abl:    # timing-only: both 64-byte halves of every 128-byte line requested together, every second slice (the other slice requests nothing)
    rep("    f32x4m raw[NIT];", "    f32x4m raw[NIT], rawd[NIT];")
    rep("""            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw[k]) : "v"(vo), "s"(base0) : "memory");""",
        """            if ((sl & 1) == 0) {
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw[k]) : "v"(vo), "s"(base0) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(rawd[k]) : "v"(vo), "s"(base0) : "memory");
            }""")
    rep("                if (g == 3) { affine_load(sn);", "                if (g == 3) { _Pragma(\"unroll\") for (int k_ = 0; k_ < NIT; ++k_) asm volatile(\"\" :: \"v\"(rawd[k_])); affine_load(sn);")
    rep("if (dma_wave) { if ((G) == 1 || (G) == 2) GN_WAIT_VM_LGKM0(3 + NIT); else GN_WAIT_VM_LGKM0(3); }", "if (dma_wave) GN_WAIT_VM_LGKM0(3);")
if "nobar" in abl:
    rep("                if (g > 0 || s > 0) __builtin_amdgcn_s_barrier();\n", "")
if "noread" in abl:      # the fragment reads of the steady state removed (registers keep the prologue's first fragments)
    rep("                W32_READ(Y, slo[j], dz * WL::HY + 1, (g % RING) * GB + STEPB);\n", "")
    rep("                W32_READ(X, slo[j], dz * WL::HY + 2, (g % RING) * GB + 2 * STEPB);\n", "")
    rep("                    W32_READ(Y, so, (g1 % 3) * WL::HY, ((g + 1) % RING) * GB);\n", '')
if "noflush" in abl:
    rep("                if (dz == 2) W32_FLUSH(j);", "                if (dz == 2 && j == 3) W32_FLUSH(j);")
open("unet_wino32.hip", "w").write(s)
