import os
which=os.environ['ABL']
p='decode_split.hip'
s=open(p).read()
def rep(old,new,count=1):
    global s
    assert s.count(old)>=1,(old)
    s=s.replace(old,new) if count==0 else s.replace(old,new,count)
if which=='nobar':
    rep('''                else DS_WAIT_VM_LGKM0(8);
                __builtin_amdgcn_s_barrier();''','''                else DS_WAIT_VM_LGKM0(8);''')
elif which=='noepi':
    # cheap epilogue: keep a data dependence on every accumulator quad, no bias / relu / split / output layer
    rep('''        auto epilogue = [&](int P, int qd) {
            const int set = P & (NSETS - 1);''','''        auto epilogue = [&](int P, int qd) {
            const int set = P & (NSETS - 1);
            if (true) {
                for (int blk = 0; blk < 2; ++blk) {
                    if (P < 4) { const int nb = 2 * P + blk, g2 = 2 * nb + (qd >> 1);
                        if (qd & 1) { h1[0][g2].z = __float_as_uint(acc[set][blk][4 * qd]); h1[1][g2].z = __float_as_uint(acc[set][blk][4 * qd + 1]); h1[0][g2].w = __float_as_uint(acc[set][blk][4 * qd + 2]); h1[1][g2].w = __float_as_uint(acc[set][blk][4 * qd + 3]); }
                        else { h1[0][g2].x = __float_as_uint(acc[set][blk][4 * qd]); h1[1][g2].x = __float_as_uint(acc[set][blk][4 * qd + 1]); h1[0][g2].y = __float_as_uint(acc[set][blk][4 * qd + 2]); h1[1][g2].y = __float_as_uint(acc[set][blk][4 * qd + 3]); }
                    } else psum[0] += (acc[set][blk][4 * qd] + acc[set][blk][4 * qd + 1]) + (acc[set][blk][4 * qd + 2] + acc[set][blk][4 * qd + 3]);
                }
                return;
            }''')
elif which=='nofrag':
    rep('''                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((t + DS_SB) % RING) * DS_STAGE_BYTES + ((kg + 1) * 4 + f) * 1024);''',
        '''                for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(nA[f].x), "+v"(nA[f].y), "+v"(nA[f].z), "+v"(nA[f].w));''')
    rep('''                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((t + 1 + DS_SB) % RING) * DS_STAGE_BYTES + f * 1024);''',
        '''                for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(nA[f].x), "+v"(nA[f].y), "+v"(nA[f].z), "+v"(nA[f].w));''')
elif which=='nodmaloop':
    rep('''                DS_ISSUE((t + RING) % NSTAGE, (t + DS_SB) % RING)''','''''')
elif which=='occ1':
    rep('''    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * DS_STAGE_BYTES + TAB_BYTES + LATB];''',
        '''    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * DS_STAGE_BYTES + TAB_BYTES + LATB + (LAT ? 0 : 40 * 1024)];''')
elif which=='nobar_nofrag':
    os.environ['ABL']='nobar'; 
    rep('''                else DS_WAIT_VM_LGKM0(8);
                __builtin_amdgcn_s_barrier();''','''                else DS_WAIT_VM_LGKM0(8);''')
    rep('''                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((t + DS_SB) % RING) * DS_STAGE_BYTES + ((kg + 1) * 4 + f) * 1024);''',
        '''                for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(nA[f].x), "+v"(nA[f].y), "+v"(nA[f].z), "+v"(nA[f].w));''')
    rep('''                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((t + 1 + DS_SB) % RING) * DS_STAGE_BYTES + f * 1024);''',
        '''                for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(nA[f].x), "+v"(nA[f].y), "+v"(nA[f].z), "+v"(nA[f].w));''')


if which=='all':
    for w in ('nobar','noepi','nofrag','nodmaloop'):
        pass
open(p,'w').write(s)
