"""dev-only patch script for tools/dev/build_variant.sh (round 5): the weight-DMA pieces of (a) the x-strip conv kernel and (b) the decoder-MLP kernel issued
BETWEEN the MFMAs of the group / stage that follows the hand-over instead of right behind its barrier -- what moved the Winograd kernel from 655 to 717
TF-eq.  R5_DMA = strip | decoder | both"""
import os
which = os.environ.get("R5_DMA", "both")
if which in ("strip", "both"):
    p = "unet_split.hip"
    s = open(p).read()
    old_issue = "            if (G + 1 < ngroups) issue_group(G + 1);\n"
    assert old_issue in s
    s = s.replace(old_issue, "")
    old = """                for (int dx = 0; dx < 3; ++dx) {     // smallest terms first, as conv3d_split_kernel
                    acc[xo] = mfma16<F16>(af[xo + dx][1], bf[dx][0], acc[xo]);
                    acc[xo] = mfma16<F16>(af[xo + dx][0], bf[dx][1], acc[xo]);
                    acc[xo] = mfma16<F16>(af[xo + dx][0], bf[dx][0], acc[xo]);
                }
"""
    new = """                for (int dx = 0; dx < 3; ++dx) {     // smallest terms first, as conv3d_split_kernel
                    acc[xo] = mfma16<F16>(af[xo + dx][1], bf[dx][0], acc[xo]);
                    acc[xo] = mfma16<F16>(af[xo + dx][0], bf[dx][1], acc[xo]);
                    acc[xo] = mfma16<F16>(af[xo + dx][0], bf[dx][0], acc[xo]);
                    // group G+1's fragments, requested in the matrix pipe's shadow (always: the pack ends in zero steps)
                    if (xo == 0 && dx == 1) issue_group(G + 1);
                }
"""
    assert old in s
    s = s.replace(old, new)
    open(p, "w").write(s)
if which in ("decoder", "both"):
    p = "decode_split.hip"
    s = open(p).read()
    # prologue: three stages (stage 3 is requested during stage 0's steps, like every later stage t+3 during stage t)
    old = "    if (!LAT) { DS_ISSUE(3, 3) }\n"
    assert old in s
    s = s.replace(old, "")
    old = "                DS_ISSUE((t + RING) % NSTAGE, (t + DS_SB) % RING)\n"
    assert old in s
    s = s.replace(old, "                if (LAT) { DS_ISSUE((t + RING) % NSTAGE, (t + DS_SB) % RING) }\n")
    # VM queue at the hand-over of stage t, oldest first: stage t+1 (4 pieces), t+2 (4), t+3 (the 3 pieces of steps 0-2) -> 7 may remain; the next
    # tile's rows are requested inside hand-over RAW_STAGE, i.e. between pieces 2 and 3 of stage RAW_STAGE+3: one hand-over later they are still
    # younger than everything waited for (7 + NRAW), two hand-overs later they are older than piece 3 of the stage that must have landed
    old = """                else if (t > RAW_STAGE && t <= RAW_STAGE + 3) DS_WAIT_VM_LGKM0(8 + NRAW);
                else DS_WAIT_VM_LGKM0(8);
"""
    assert old in s
    s = s.replace(old, """                else if (t == RAW_STAGE + 1) DS_WAIT_VM_LGKM0(7 + NRAW);
                else DS_WAIT_VM_LGKM0(7);
""")
    old = """            acc[set][0] = ds_mfma(A[1], b1, acc[set][0]);
            acc[set][1] = ds_mfma(A[3], b1, acc[set][1]);
"""
    new = """            acc[set][0] = ds_mfma(A[1], b1, acc[set][0]);
            acc[set][1] = ds_mfma(A[3], b1, acc[set][1]);
            if (!LAT) {                 // piece kg of stage t+3 -> the slot stage t-1 vacated at the last hand-over, in the matrix pipe's shadow
                ds_glds16_s(wsrc + (size_t)((t + RING - 1) % NSTAGE) * DS_STAGE_BYTES + kg * 1024, lane16,
                            lds_base + ((t + RING - 1 + DS_SB) % RING) * DS_STAGE_BYTES + (wave * 4 + kg) * 1024);
            }
"""
    assert old in s
    s = s.replace(old, new)
    open(p, "w").write(s)
