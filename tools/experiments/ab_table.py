import sys, re, collections
tab = collections.OrderedDict(); cur = None; seen = []
for l in sys.stdin:
    l = l.rstrip()
    if l.startswith('=='):
        cur = l[3:]
        while cur in seen: cur += "'"
        seen.append(cur); continue
    m = re.match(r'(.{24}) t\d+:\s*([\d.]+)', l)
    if m and cur: tab.setdefault(m.group(1).strip(), collections.OrderedDict())[cur] = m.group(2)
    elif l.startswith('[gpurun]') or l.startswith('STEP'): print(l)
if tab:
    cols = list(next(iter(tab.values())).keys())
    print('%-26s' % 'shape' + ''.join('%9s' % c[:8] for c in cols))
    for k, v in tab.items(): print('%-26s' % k + ''.join('%9s' % v.get(c, '-') for c in cols))
