"""dev-only: per-kernel ISA statistics of a hipcc -S --cuda-device-only listing (spill traffic inside the MFMA loop, waits before barriers)"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else r'_Z\w+'
funcs = re.split(r'\n(?=' + pat + r':)', txt)
for f in funcs[1:]:
    name = f.split(':')[0]
    lines = f.split('\n')
    end = [i for i, l in enumerate(lines) if 's_endpgm' in l][0]
    lines = lines[:end]
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    if not mf: continue
    sc = [i for i, l in enumerate(lines) if 'scratch_' in l]
    gl = [i for i, l in enumerate(lines) if 'global_load_lds' in l]
    bar = [i for i, l in enumerate(lines) if 's_barrier' in l]
    print(name, 'lines', len(lines), 'mfma', len(mf), 'range', mf[0], mf[-1], 'scratch', len(sc), 'inside mfma range',
          sum(1 for i in sc if mf[0] < i < mf[-1]), 'glds', len(gl), 'barriers', len(bar))
    for i in bar[2:5]:
        print('   ', [l.strip() for l in lines[i - 3:i + 1]])
