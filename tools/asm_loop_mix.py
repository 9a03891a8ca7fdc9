"""Instruction mix of the main (MFMA-carrying) loop of a kernel in a gfx950 assembly listing:
   hipcc -S --cuda-device-only -o x.s file.hip ; python tools/asm_loop_mix.py x.s <mangled-name-substring>"""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):', s, re.M)
    i0 = m.end()
    i1 = s.index('s_endpgm', i0)
    body = s[i0:i1].split('\n')
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(body):
        mm = re.match(r'\s*s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            a, b = labels[mm.group(1)], i
            n = sum(1 for x in mf if a <= x <= b)
            if n and (best is None or n > best[2]):
                best = (a, b, n)
    a, b, _ = best
    c = collections.Counter()
    for l in body[a:b + 1]:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'):
            continue
        op = l.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith('scratch_'):
            c[op] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith('ds_'):
            c['ds'] += 1
        elif op.startswith('global_') or op.startswith('buffer_'):
            c[op] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
            if 'vmcnt' in l:
                print('   ', l)
        elif op.startswith('s_barrier'):
            c['barrier'] += 1
        elif op.startswith('s_nop'):
            c['nop'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
    print(m.group(1)[:60], 'lines', a, b, dict(c))


main()
