"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` output directory into the markdown table committed under
profiles/ (top kernels by total time) -- and, with --timeline, a per-stream occupancy summary of the kernel trace (how much of
the side-stream work overlaps the main stream); with --tail-ms T, the same figures plus the top kernels and the idle time
inside the LAST T milliseconds of the trace (the timed steps of a bench run, without model construction and warm-up).
usage: prof_summary.py <dir> "<title>" [--timeline] [--tail-ms T]"""
import csv
import glob
import os
import sys


def main():
    d, title = sys.argv[1], sys.argv[2]
    stats = sorted(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True))
    if not stats:
        sys.exit(f'no *kernel_stats.csv under {d}')
    rows = list(csv.DictReader(open(stats[-1])))
    print(f'# {title}\n')
    print('rocprofv3 --kernel-trace --stats; durations in microseconds.\n')
    print('| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|')
    for r in rows[:45]:
        name = r['Name'][:100]
        print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e3:.0f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
    if '--timeline' in sys.argv:
        tr = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))
        if tr:
            ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(tr[-1]))]
            ev.sort()
            busy = 0
            cur_s, cur_e = ev[0][0], ev[0][1]
            tot = 0
            for s, e, _ in ev:
                tot += e - s
                if s > cur_e:
                    busy += cur_e - cur_s
                    cur_s, cur_e = s, e
                else:
                    cur_e = max(cur_e, e)
            busy += cur_e - cur_s
            print(f'\nkernel trace: {len(ev)} dispatches, sum of durations {tot / 1e6:.2f} ms, union (wall with >= 1 kernel running) {busy / 1e6:.2f} ms '
                  f'-> {100.0 * (tot - busy) / tot:.1f} % of kernel time overlapped with another kernel')
            if '--tail-ms' in sys.argv:
                T = float(sys.argv[sys.argv.index('--tail-ms') + 1]) * 1e6
                end = max(e for _, e, _ in ev)
                win = [(max(s, end - T), e, n) for s, e, n in ev if e > end - T]
                agg = {}
                for s, e, n in win:
                    a = agg.setdefault(n[:100], [0, 0]); a[0] += 1; a[1] += e - s
                busy = 0; cur_s, cur_e = win[0][0], win[0][1]
                for s, e, _ in win:
                    if s > cur_e:
                        busy += cur_e - cur_s; cur_s, cur_e = s, e
                    else:
                        cur_e = max(cur_e, e)
                busy += cur_e - cur_s
                print(f'\n## last {T / 1e6:.0f} ms of the trace: {len(win)} dispatches, >= 1 kernel running for {busy / 1e6:.2f} ms, idle {(T - busy) / 1e6:.2f} ms\n')
                print('| kernel | calls | total_us | avg_us | % of window |\n|---|---|---|---|---|')
                for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
                    print(f'| `{n}` | {c} | {t / 1e3:.0f} | {t / 1e3 / c:.2f} | {100.0 * t / T:.2f} |')


if __name__ == '__main__':
    main()
