"""Dev aid: per-basic-block instruction mix of one kernel in a hipcc -S listing."""
import re, sys
from collections import Counter
path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
names = re.findall(r'^(_Z\w+):', s, flags=re.M)
name = [n for n in names if re.search(pat, n)][0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
blocks = []; cur = ['entry', Counter(), []]
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith(';'): continue
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur); cur = [m.group(1), Counter(), []]
    elif t.startswith('.'): continue
    else:
        cur[1][t.split()[0]] += 1; cur[2].append(t)
blocks.append(cur)
print(name)
for nm, c, ins in blocks:
    tot = sum(c.values())
    fp = sum(v for k, v in c.items() if 'f64' in k or ('f32' in k and k.startswith('v_')))
    rl = c['v_readlane_b32'] + c['v_writelane_b32']
    br = [x.split()[-1] for x in ins if x.startswith(('s_cbranch', 's_branch'))]
    print(f"{nm:12s} total={tot:4d} fp={fp:4d} lane={rl:3d} st={sum(v for k,v in c.items() if 'store' in k)} ld={sum(v for k,v in c.items() if 'global_load' in k or 'scratch' in k)} smov={c['s_mov_b32']} vmov={c['v_mov_b64_e32']+c['v_mov_b32_e32']} br={br}")
