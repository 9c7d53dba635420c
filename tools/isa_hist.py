"""Instruction histogram of a kernel (whole function and its largest loop) from hipcc -S output.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/v.s vslam_hip.hip
    python tools/isa_hist.py /tmp/v.s <substring of the mangled name> [top]"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
for f in re.split(r'\n(?=_Z\w+:)', s):
    name = f.split(':', 1)[0]
    if pat not in name:
        continue
    lines = f.splitlines()
    def hist(ls):
        c = collections.Counter()
        for l in ls:
            m = re.match(r'\s+([vsdg]\w+|buffer\w+|flat\w+)\s', l)
            if m:
                c[m.group(1)] += 1
        return c
    c = hist(lines)
    blk = s[s.find('.name:           ' + name):][:3000]
    vg = re.search(r'\.vgpr_count:\s+(\d+)', blk)
    lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', blk)
    print(name[:90])
    print(' whole: total', sum(c.values()), 'valu', sum(n for k, n in c.items() if k.startswith('v_')), 'vgpr', vg.group(1) if vg else '?', 'static lds', lds.group(1) if lds else '?')
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    for span, a, b in sorted(loops, reverse=True)[:3]:
        c = hist(lines[a:b])
        print('  loop of', sum(c.values()), 'instr, valu', sum(n for k, n in c.items() if k.startswith('v_')), ':',
              ', '.join(f'{k} {n}' for k, n in c.most_common(top)))
