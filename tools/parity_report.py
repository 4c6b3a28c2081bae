"""gpurun_out/parity_numbers.jsonl (appended by tests/parity.py during `pytest -m gpu`) -> a markdown table of every measured parity figure next to its
asserted bound (profiles/rNN_parity_numbers.md).  Usage: python tools/parity_report.py [jsonl] > profiles/r05_parity_numbers.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'parity_numbers.jsonl')
    rows = {}
    for ln in open(path):
        r = json.loads(ln)
        k = r['name']
        worst = max if r['kind'] == '<=' else min
        if k in rows:
            rows[k]['measured'] = worst(rows[k]['measured'], r['measured'])
            rows[k]['runs'] += 1
        else:
            rows[k] = dict(r, runs=1)
    print('# Measured parity figures (MI355X, `pytest -m gpu`), worst value over the runs of the log, next to the asserted bound\n')
    print('| check | measured | bound | bound / measured |')
    print('|---|---|---|---|')
    for k, r in rows.items():
        m, b = r['measured'], r['bound']
        ratio = (b / m if r['kind'] == '<=' else (1 - b) / max(1 - m, 1e-12)) if m else float('inf')
        print(f"| {k} | {m:.4g} | {r['kind']} {b:.4g} | {ratio:.2f} |")


if __name__ == '__main__':
    main()
