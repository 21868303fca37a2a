#!/usr/bin/env python3
"""Static instruction mix per kernel of a hipcc -S listing: tools/isa_mix.py file.s [name-filter ...]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2:]
cur, cnt = None, None
def cls(op):
    if op.startswith('v_fmac_f64_dpp') or op.startswith('v_fmac_f32_dpp'): return 'fmac_dpp'
    if op.startswith('v_accvgpr'): return 'accvgpr'
    if '_dpp' in op: return 'other_dpp'
    if op.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64')): return 'valu_f64'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_load', 'buffer_load')): return 'gload'
    if op.startswith(('global_store', 'buffer_store')): return 'gstore'
    if op.startswith('scratch'): return 'scratch'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_'): return 'salu'
    return 'other'
for line in txt:
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur, cnt = m.group(1), collections.Counter()
        continue
    if cur is None: continue
    s = line.strip()
    if s.startswith('s_endpgm'):
        if not flt or all(f in cur for f in flt): print(cur[:60], dict(sorted(cnt.items())))
        cur = None
        continue
    if not s or s[0] in '.;/' or s.endswith(':'): continue
    cnt[cls(s.split()[0])] += 1
