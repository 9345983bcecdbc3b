"""Static instruction mix of the longest loop of each kernel in a hipcc -S listing (development aid)."""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name: continue
    lines = body.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'\s*(\.LBB\d+_\d+):', l)
        if mm: labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r'\s*s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]: best = span
    if best is None: continue
    cnt = collections.Counter()
    for l in lines[best[0]:best[1]]:
        t = l.strip().split(' ')[0]
        if not t or t.startswith(('.', ';')): continue
        key = 'other:' + t[:10]
        if t.startswith('v_mfma'): key = 'mfma'
        elif t.startswith('v_accvgpr'): key = 'accmov'
        elif t.startswith('v_'): key = 'valu'
        elif t.startswith('ds_'): key = t[:14]
        elif t.startswith('s_waitcnt'): key = 'waitcnt'
        elif t.startswith('s_'): key = 'salu'
        elif t.startswith('buffer'): key = 'buffer'
        elif t.startswith('scratch'): key = 'scratch'
        cnt[key] += 1
    print(name[:60], 'loop', best, 'instrs', sum(cnt.values()))
    print('   ', sorted(cnt.items(), key=lambda x: -x[1]))
