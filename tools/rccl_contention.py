"""What the RCCL gradient exchange costs the backward it overlaps with -- measured where it can be measured today: ONE GPU, `VLASER_FORCE_DP=1` (the SFT step runs its
ZeRO-1 exchange through RCCL with a world of one rank: every bucket's in-place reduce_scatter_tensor / all_gather_into_tensor is a real RCCL launch on the comm stream).

    python tools/rccl_contention.py sweep                 # ms / step for NCCL_MAX_NCHANNELS in {default, 8, 16, 32} (one child bench per setting, same box)
    python tools/rccl_contention.py analyze <trace dir>   # over a `rocprofv3 --kernel-trace --output-format csv` of `VLASER_FORCE_DP=1 python bench.py --workload sft`:
                                                           # the RCCL kernels (grid, workgroup size, duration) and the stretch of the compute kernels they overlap

The child processes are started with subprocess (never an exec from a process that touched the GPU).  Output is markdown (profiles/r05_rccl_contention.md)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sweep():
    print('## NCCL_MAX_NCHANNELS sweep, `VLASER_FORCE_DP=1 python bench.py --workload sft --sft-steps 8` (world 1, same box, one process per setting)\n')
    print('| NCCL_MAX_NCHANNELS | ms / step | forward + backward ms | first collective ms |')
    print('|---|---|---|---|')
    for ch in ('default', '8', '16', '32'):
        env = dict(os.environ, VLASER_FORCE_DP='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
        env.pop('NCCL_MAX_NCHANNELS', None)
        if ch != 'default':
            env['NCCL_MAX_NCHANNELS'] = ch
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'sft', '--sft-steps', '8', '--no-cpu-baseline'], env=env, capture_output=True, text=True,
                           timeout=900)
        line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith('{')), None)
        if line is None:
            print(f'| {ch} | failed (rc {r.returncode}): {r.stderr.strip().splitlines()[-1][:120] if r.stderr.strip() else ""} | | |')
            continue
        d = json.loads(line)
        print(f"| {ch} | {d['ms_per_step']} | {d['fwd_bwd_ms']} | {d['exchange'].get('first_collective_ms')} |")
    sys.stdout.flush()


def analyze(d):
    tr = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))
    if not tr:
        sys.exit(f'no *kernel_trace.csv under {d}')
    rows = list(csv.DictReader(open(tr[-1])))
    ev = []
    for r in rows:
        ev.append(dict(name=r['Kernel_Name'], s=int(r['Start_Timestamp']), e=int(r['End_Timestamp']), grid=int(r.get('Grid_Size', 0) or 0), wg=int(r.get('Workgroup_Size', 0) or 0),
                       q=r.get('Queue_Id', '')))
    is_rccl = lambda n: any(t in n.lower() for t in ('nccl', 'rccl', 'comm_shadow'))     # (comm_shadow: tools/micro/rccl_shadow_lab.py's stand-in)
    rc = [x for x in ev if is_rccl(x['name'])]
    co = [x for x in ev if not is_rccl(x['name'])]
    print(f'## RCCL kernels in the trace ({len(rc)} launches, {len(co)} compute launches)\n')
    agg = defaultdict(list)
    for x in rc:
        agg[(x['name'][:90], x['grid'], x['wg'])].append((x['e'] - x['s']) / 1e3)
    print('| kernel | grid (work-items) | workgroup | workgroups | calls | avg us | max us |')
    print('|---|---|---|---|---|---|---|')
    for (n, g, w), ds in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f'| `{n}` | {g} | {w} | {g // max(w, 1)} | {len(ds)} | {sum(ds) / len(ds):.1f} | {max(ds):.1f} |')
    # stretch of the compute kernels that run while an RCCL kernel is resident
    rc_iv = sorted((x['s'], x['e']) for x in rc)
    def overlaps(x):
        return any(s < x['e'] and x['s'] < e for s, e in rc_iv)
    by = defaultdict(lambda: ([], []))
    for x in co:
        by[x['name'][:90]][1 if overlaps(x) else 0].append((x['e'] - x['s']) / 1e3)
    print('\n## Compute kernels beside an RCCL kernel vs alone (same trace)\n')
    print('| kernel | alone: calls, avg us | beside RCCL: calls, avg us | stretch |')
    print('|---|---|---|---|')
    out = []
    for n, (a, b) in by.items():
        if len(a) >= 2 and len(b) >= 2:
            out.append((sum(b), n, a, b))
    for _, n, a, b in sorted(out, reverse=True)[:25]:
        ma, mb = sum(a) / len(a), sum(b) / len(b)
        print(f'| `{n}` | {len(a)}, {ma:.1f} | {len(b)}, {mb:.1f} | x{mb / ma:.2f} |')


if __name__ == '__main__':
    if len(sys.argv) >= 2 and sys.argv[1] == 'sweep':
        sweep()
    elif len(sys.argv) >= 3 and sys.argv[1] == 'analyze':
        analyze(sys.argv[2])
    else:
        sys.exit(__doc__)
