"""dev-only patch script for tools/dev/build_variant.sh: TIMING-ONLY ablations of csrc/unet_wino.hip (results are wrong on purpose; each removes one
cost item of the Winograd kernel's group loop so that its share of the group time can be read off an A/B run).  WN_ABL = noflush (NOT a measurement: removes every MFMA) | flush1of4 | noconvert |
nodma | nobarrier | norows"""
import os
which = os.environ["WN_ABL"]
p = "unet_wino.hip"
s = open(p).read()
if which == "noflush":          # the output transform's VALU work (tot += acc) -> nothing; accumulators keep running
    a = s.index("            if (dz == 2) {                          // this transform position is complete")
    b = s.index("        sbase = nbase;")
    s = s[:a] + "        }\n" + s[b:]
elif which == "flush1of4":      # ONE output transform per slice (64 of the 192 v_add; every MFMA stays alive): what two thirds of the transform's VALU work cost
    s = s.replace("            if (dz == 2) {                          // this transform position is complete", "            if (g == 11) {                          //")
    s = s.replace("                WN_FLUSH(j, acc);", "                WN_FLUSH(1, acc);")
elif which == "direct03":       # a REAL variant (correct results): transform positions 0 and 3 accumulate straight into the even / odd totals (position 3 on
    # the negated input transform d3 - d1), only positions 1 and 2 go through the accumulator + output transform: 128 instead of 192 v_add per slice,
    # at the price of single-level summation for two of the four chains
    s = s.replace("        else { va = raw[1] - raw[3]; vb = raw[3] - raw[5]; }", "        else { va = raw[3] - raw[1]; vb = raw[5] - raw[3]; }")
    s = s.replace("            const int j = g / 3, dz = g % 3, X = g & 1, Y = X ^ 1;",
                  "            const int j = g / 3, dz = g % 3, X = g & 1, Y = X ^ 1;\n            f32x16s (&ACJ)[NT] = j == 0 ? tot[0] : j == 3 ? tot[1] : acc;")
    for m in ("WN_PROD_A(X, acc)", "WN_PROD_B(X, acc)", "WN_PROD_A(Y, acc)", "WN_PROD_B(Y, acc)"):
        s = s.replace(m, m.replace("acc", "ACJ"))
    s = s.replace("            if (dz == 2) {                          // this transform position is complete", "            if (dz == 2 && (j == 1 || j == 2)) {    //")
elif which == "dephase":        # a REAL variant (correct results): the two waves of a SIMD (column groups 0 / 1) convert their halo rows in DIFFERENT groups (2, 4, 7, 10 vs
    # 3, 5, 8, 11), so that one wave's VALU block runs beside the other's MFMAs instead of beside its VALU block
    s = s.replace("            if (g == 2) { affine_rows(sn); convert(0, nslo[0]); }\n", "            if (g == 2 + cg_odd) { affine_rows(sn); convert(0, nslo[0]); }\n")
    for g, j in ((4, 1), (7, 2), (10, 3)):
        s = s.replace(f"            if (g == {g}) convert({j}, nslo[{j}]);\n", f"            if (g == {g} + cg_odd) convert({j}, nslo[{j}]);\n")
    s = s.replace("            else if (g == 3 || g == 5 || g == 8 || g == 11) GN_WAIT_VM_LGKM0(2);", "            else if (g != 2 && g != 7 && g != 10) GN_WAIT_VM_LGKM0(2);")
    s = s.replace("    int sbase = 0;                                  // (4 s) % 5: slot of this slice's j = 0", "    const int cg_odd = cg;\n    int sbase = 0;                                  //")
elif which == "noconvert":      # no staging conversions inside the loop (slice 0's halo is reused for every slice)
    for g, j in ((2, 0), (4, 1), (7, 2), (10, 3)):
        s = s.replace(f"            if (g == {g}) {{ affine_rows(sn); convert(0, nslo[0]); }}\n", "")
        s = s.replace(f"            if (g == {g}) convert({j}, nslo[{j}]);\n", "")
    s = s.replace("            WN_READ(X, slo[j], dz * WL::HY + 2,", "            WN_READ(X, j * WL::SLOT, dz * WL::HY + 2,").replace(
        "            WN_READ(Y, slo[j], dz * WL::HY + 1,", "            WN_READ(Y, j * WL::SLOT, dz * WL::HY + 1,").replace(
        "const int so = g + 1 < 12 ? slo[g1 / 3] : nslo[0] * WL::SLOT;", "const int so = (g1 / 3) * WL::SLOT;")
elif which == "nodma":          # no weight DMA inside the loop (the ring keeps groups 0 / 1 forever)
    s = s.replace("            WN_ISSUE_PIECE((g + 2) % RING, 0);      // group g+2 -> the slot group g-1 vacated\n", "")
    s = s.replace("            WN_ISSUE_PIECE((g + 2) % RING, 1);\n", "").replace("            WN_ISSUE_PIECE((g + 2) % RING, 2);\n", "")
    s = s.replace("if (g == 1) GN_WAIT_VM_ONLY(2 + NIT);", "if (g == 1) GN_WAIT_VM_ONLY(NIT);").replace(
        "GN_WAIT_VM_LGKM0(2);", "GN_WAIT_VM_LGKM0(0);").replace("else GN_WAIT_VM_ONLY(2);", "else GN_WAIT_VM_ONLY(0);")
elif which == "nobarrier":      # no hand-over barrier inside the loop
    a = s.index("#pragma unroll\n        for (int g = 0; g < 12; ++g) {")
    b = s.index("        sbase = nbase;")
    s = s[:a] + s[a:b].replace("            __builtin_amdgcn_s_barrier();\n", "") + s[b:]
elif which == "norows":         # no row loads / conversions at all
    s = s.replace("            if (g == 0) issue_rows(sn);             // always (uniform wait counts); unused after the last slice\n", "")
    s = s.replace("if (g == 1) GN_WAIT_VM_ONLY(2 + NIT);", "if (g == 1) GN_WAIT_VM_ONLY(2);")
    for g, j in ((2, 0), (4, 1), (7, 2), (10, 3)):
        s = s.replace(f"            if (g == {g}) {{ affine_rows(sn); convert(0, nslo[0]); }}\n", "")
        s = s.replace(f"            if (g == {g}) convert({j}, nslo[{j}]);\n", "")
else:
    raise SystemExit("unknown WN_ABL " + which)
open(p, "w").write(s)
