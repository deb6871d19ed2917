"""Condensed instruction stream of one kernel from a hipcc -S listing: M = MFMA, r/w = ds_read/ds_write, g = global load,
s = global store, D = LDS-DMA, [..] = s_waitcnt, |B| = barrier, . = other VALU.
    python tools/isa_seq.py file.s mangled-substring [maxchars]"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M)
lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith(';')]
seq = []
for l in lines:
    op = l.split()[0]
    if op.endswith(':'): seq.append('\n' + op + ' ')
    elif op.startswith('v_mfma'): seq.append('M')
    elif op.startswith('ds_read'): seq.append('r')
    elif op.startswith('ds_write'): seq.append('w')
    elif op.startswith('s_waitcnt'): seq.append('[' + l.split(None, 1)[1].replace('lgkmcnt', 'L').replace('vmcnt', 'V') + ']')
    elif op.startswith('s_barrier'): seq.append('|B|')
    elif 'load_lds' in op or (op.startswith('buffer_load') and ' lds' in l): seq.append('D')
    elif op.startswith('global_load') or op.startswith('buffer_load'): seq.append('g')
    elif op.startswith('global_store') or op.startswith('buffer_store'): seq.append('s')
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): seq.append('<br>')
    elif op.startswith('v_'): seq.append('.')
out = ''.join(seq)
print(out[:int(sys.argv[3]) if len(sys.argv) > 3 else 8000])
