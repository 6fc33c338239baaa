"""DEVELOPER-ONLY: per-kernel instruction mix from `hipcc -S --cuda-device-only` output.  Usage: isa_mix.py file.s [name-substring ...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2:]
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\w+):', s, re.M)]
for i, (pos, name) in enumerate(starts):
    if want and not any(w in name for w in want):
        continue
    end = starts[i + 1][0] if i + 1 < len(starts) else len(s)
    body = s[pos:end].split('.end_amdhsa_kernel')[0]
    ins = [l.split()[0] for l in body.split('\n') if l.strip() and l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
    c = collections.Counter(ins)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print("%s\n  total %d  valu %d (pk %d)  ds %d  global/buffer %d  scratch %d  salu %d  waitcnt %d" % (
        name[:70], len(ins), grp('v_'), grp('v_pk'), grp('ds_'), grp('global_') + grp('buffer_'), grp('scratch_'), grp('s_') - c['s_waitcnt'] - c['s_nop'], c['s_waitcnt']))
    print("  ", ", ".join("%s %d" % kv for kv in c.most_common(24)))
