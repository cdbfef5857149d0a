#!/usr/bin/env python3
"""tools/tidy_kernel_stats.py <rocprofv3 kernel_stats.csv> <out.csv> "<header comment>" -- shorten the demangled kernel names to the function name."""
import csv, re, sys
src, dst, note = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.reader(open(src)))
out = []
for r in rows:
    n = re.sub(r'^void ', '', r[0]).replace('(anonymous namespace)::', '')
    if 'distribution_elementwise' in n:
        n = 'at::native::distribution_elementwise_grid_stride_kernel<...random_from_to...> (torch.randint: synthetic input generation, outside the timed region)'
    elif r[0] != 'Name':
        m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', n)
        n = m.group(1) if m else n
    out.append([n] + r[1:])
with open(dst, 'w') as f:
    f.write('"# %s"\n' % note)
    csv.writer(f).writerows(out)
