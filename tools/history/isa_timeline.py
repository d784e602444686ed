"""Compress a kernel's gfx950 assembly into a one-line schedule: M=mfma, G=global_load, S=global_store,
d=ds_read, w=ds_write, [vN lN]=s_waitcnt, |B|=barrier, ^=branch.  usage: isa_timeline.py file.s substr [maxchars]"""
import re, sys
s = open(sys.argv[1]).read()
sub = sys.argv[2]
maxc = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
lines = s.split('\n')
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and sub in l]
for st in start:
    seq = []
    for l in lines[st + 1:]:
        l = l.strip()
        if l.startswith('s_endpgm'):
            break
        if l.startswith('v_mfma'): seq.append('M')
        elif l.startswith('global_load') or l.startswith('buffer_load'): seq.append('G')
        elif l.startswith('global_store') or l.startswith('buffer_store'): seq.append('S')
        elif l.startswith('ds_read') or l.startswith('ds_load'): seq.append('d')
        elif l.startswith('ds_write') or l.startswith('ds_store'): seq.append('w')
        elif l.startswith('scratch_'): seq.append('X')
        elif l.startswith('s_waitcnt'):
            m = re.search(r'vmcnt\((\d+)\)', l); n = re.search(r'lgkmcnt\((\d+)\)', l)
            seq.append('[' + ('v%s' % m.group(1) if m else '') + ('l%s' % n.group(1) if n else '') + ']')
        elif l.startswith('s_barrier'): seq.append('|B|')
        elif l.startswith('s_cbranch'): seq.append('^')
        elif l.startswith('v_exp') or l.startswith('v_rcp'): seq.append('e')
    out = ''.join(seq)
    for ch in 'MdGSwXe':
        out = re.sub('(%s+)' % ch, lambda m: '%s%d ' % (ch, len(m.group(1))), out)
    print(lines[st][:80]); print(out[:maxc]); print()
