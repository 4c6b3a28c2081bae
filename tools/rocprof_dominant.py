"""Write profiles/<tag>_dominant_kernel_rocprof.json from a committed `<tag>_chunk_kernel_stats.md` (tools/run_profile.sh): the rocprofv3 average duration of the
dominant kernel (chain_gu_kernel) -- bench.py reports it next to the in-chain figure and quotes `roofline.frac` on the larger of the two.
usage: python tools/rocprof_dominant.py profiles/r06x_chunk_kernel_stats.md"""
import json
import os
import re
import sys


def main():
    path = sys.argv[1]
    rows = []
    tail = False
    for ln in open(path):
        if 'last' in ln and 'ms of the trace' in ln:
            tail = True
        m = re.match(r'\| `(.*chain_gu_kernel[^`]*)` \| (\d+) \| (\d+) \| ([\d.]+) \|', ln)
        if m:
            rows.append((tail, m.group(1), int(m.group(2)), float(m.group(4))))
    if not rows:
        sys.exit('no chain_gu_kernel row in ' + path)
    whole = [r for r in rows if not r[0]]
    timed = [r for r in rows if r[0]]
    pick = max(whole or rows, key=lambda r: r[2])
    rec = {'kernel': pick[1][:120], 'calls': pick[2], 'avg_us': pick[3], 'avg_us_timed_tail': max(timed, key=lambda r: r[2])[3] if timed else None,
           'from': os.path.relpath(path, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}
    tag = os.path.basename(path).split('_')[0]
    out = os.path.join(os.path.dirname(path), f'{tag}_dominant_kernel_rocprof.json')
    json.dump(rec, open(out, 'w'), indent=1)
    print(out, rec)


if __name__ == '__main__':
    main()
