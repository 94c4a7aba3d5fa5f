"""Instruction mix of the MFMA loops of one kernel in a -save-temps assembly file:
    python tools/isa_mix.py build_dev/nn-hip-amdgcn-amd-amdhsa-gfx950.s <mangled-name prefix>"""
import collections
import sys
lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]))
end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i])
body = lines[start:end]
for l in body:
    if any(k in l for k in ['.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_private_segment_fixed_size']):
        print(l.strip())
blocks, cur, name = [], [], 'entry'
for l in body:
    t = l.strip()
    if t.startswith('.LBB') and ':' in t:
        blocks.append((name, cur))
        cur, name = [], t.split(':')[0]
    else:
        cur.append(t)
blocks.append((name, cur))
for name, b in blocks:
    n = sum(1 for x in b if x.startswith('v_mfma'))
    if n > 8:
        cnt = collections.Counter(x.split()[0] for x in b if x and not x.startswith((';', '.')))
        print(name, 'mfma', n, 'total', sum(cnt.values()))
        print('  ', dict(cnt.most_common(30)))
        print('  ', [x for x in b if x.startswith('v_mfma')][:2])
