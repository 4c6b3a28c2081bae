"""Measured-error bookkeeping of the GPU parity tests (VERDICT r04 #6a/b).

`parity(name, measured, bound)` asserts `measured <= bound` (or >= for lower bounds), prints the pair (visible with `pytest -s`) and appends
it to gpurun_out/parity_numbers.jsonl, which travels back from the GPU box; `tools/parity_report.py` turns that file into the committed table
profiles/rNN_parity_numbers.md.  Bounds are kept at <= 2 x the value measured in the round that set them, so a 2 x regression fails.
"""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.environ.get('VLASER_PARITY_LOG') or os.path.join(ROOT, 'gpurun_out', 'parity_numbers.jsonl')


def parity(name, measured, bound, lower=False, note=''):
    measured = float(measured)
    rec = {'name': name, 'measured': measured, 'bound': float(bound), 'kind': '>=' if lower else '<=', 'note': note}
    print('PARITY', json.dumps(rec))
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
    ok = measured >= bound if lower else measured <= bound
    assert ok, f'{name}: measured {measured:.4g} violates the bound {bound:.4g} ({"lower" if lower else "upper"}) {note}'
    return measured


def relmax(a, b):
    """max|a - b| / max|b|: the max-normalised figure (an outright wrong SMALL element passes it: pair it with `elementwise`)."""
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


def rel_l2(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm()).item()


def elementwise(a, b, rtol, atol):
    """Worst element of |a - b| / (atol + rtol |b|): <= 1 means every element is within atol + rtol |b| (torch.allclose's criterion, as a number)."""
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).abs() / (atol + rtol * b.abs())).max().item()
