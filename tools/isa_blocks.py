"""Per-basic-block instruction statistics of one kernel in a hipcc -S listing (development aid).
usage: python tools/isa_blocks.py file.s mangled_kernel_name_substring"""
import collections, re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
i = m.start(); j = s.index('.end_amdhsa_kernel', i)
blocks = []; cur = ['<entry>', []]
for l in s[i:j].splitlines():
    t = l.strip()
    mm = re.match(r'^(\.LBB\d+_\d+):', t)
    if mm:
        blocks.append(cur); cur = [mm.group(1) + (' LOOP' if 'Loop' in t else ''), []]
    elif t and not t.startswith((';', '.')):
        cur[1].append(t)
blocks.append(cur)
for name, ins in blocks:
    c = collections.Counter(x.split()[0] for x in ins)
    print(f"{name:18s} n={len(ins):4d} mfma={c['v_mfma_f32_32x32x16_bf16'] + c['v_mfma_f32_16x16x32_bf16']:3d} accw={c['v_accvgpr_write_b32']:3d} "
          f"accr={c['v_accvgpr_read_b32']:3d} exp={c['v_exp_f32_e32']:2d} dma={c['buffer_load_dwordx4']:2d} ds_r={sum(v for k, v in c.items() if k.startswith('ds_read')):3d} "
          f"ds_w={sum(v for k, v in c.items() if k.startswith('ds_write')):2d} scratch={sum('scratch' in x for x in ins):2d} bar={c['s_barrier']} "
          f"waitcnt={c['s_waitcnt']:3d} valu={sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k and 'accvgpr' not in k):3d}")
if len(sys.argv) > 3:
    for name, ins in blocks:
        if name.split()[0] == sys.argv[3]:
            print('\n'.join(ins))
