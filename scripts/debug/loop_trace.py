"""Instruction-class trace of the hottest basic block (most MFMAs) of a kernel in a hipcc -S listing:
M = matrix instruction, v = vector ALU, d = ds_read, w = ds_write, L = LDS-DMA / buffer load, G = global/buffer other, s = scalar ALU,
S = scalar memory, W = s_waitcnt, B = s_barrier, n = s_nop, x = scratch.     python loop_trace.py file.s kernel_substring"""
import re, sys
s = open(sys.argv[1]).read()
names = re.findall(r'^(_Z\S+):', s, re.M)
for name in names:
    if len(sys.argv) > 2 and sys.argv[2] not in name: continue
    i = s.index(name + ':'); j = s.index('s_endpgm', i)
    blocks = re.split(r'\n(?=\.LBB)', s[i:j])
    loop = max(blocks, key=lambda x: len(re.findall('v_mfma', x))).split('\n')
    out = []
    for l in loop:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'): continue
        op = l.split()[0]
        if op.startswith('v_mfma'): c = 'M'
        elif op.startswith('v_'): c = 'v'
        elif op.startswith('ds_read') or op.startswith('ds_load'): c = 'd'
        elif op.startswith('ds_'): c = 'w'
        elif 'lds' in l and op.startswith('buffer_load'): c = 'L'
        elif op.startswith('buffer_') or op.startswith('global_'): c = 'G'
        elif op.startswith('scratch_'): c = 'x'
        elif op == 's_waitcnt': c = 'W'
        elif op == 's_barrier': c = 'B'
        elif op == 's_nop': c = 'n'
        elif op.startswith('s_load') or op.startswith('s_buffer'): c = 'S'
        elif op.startswith('s_'): c = 's'
        else: c = '?'
        out.append(c)
    t = ''.join(out)
    print(name, len(t))
    for k in range(0, len(t), 120): print('   ', t[k:k + 120])
