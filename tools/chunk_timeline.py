"""Per-dispatch timeline of the LAST chunks of a `rocprofv3 --kernel-trace --output-format csv` run of
`bench.py --workload vla_chunk`: which kernels run between the last kernel of chunk i and the first of chunk i+1 (the part of
`ms_per_step` that no phase graph contains), and the in-chain duration + gap-before of every kernel class inside a chunk.
usage: chunk_timeline.py <dir> [n_chunks]"""
import csv
import glob
import os
import sys


def short(n):
    n = n.replace('void ', '')
    for a, b in (('at::native::', 'at::'), ('(anonymous namespace)::', '')):
        n = n.replace(a, b)
    return n[:72]


def main():
    d = sys.argv[1]
    nch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tr = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(tr[-1]))]
    ev.sort()
    # a chunk starts at im2col_kernel (first kernel of the captured graph)
    starts = [i for i, e in enumerate(ev) if e[2].startswith('im2col_kernel')]
    if len(starts) < nch + 1:
        sys.exit('not enough chunks in the trace')
    starts = starts[-(nch + 1):]
    print(f'# chunk timeline: last {nch} chunks of the trace ({len(ev)} dispatches)\n')
    print('| chunk | graph kernels | first->last kernel ms | sum of kernel durations ms | idle inside ms | gap to next chunk us | kernels in the gap |')
    print('|---|---|---|---|---|---|---|')
    # the graph's last kernel: vla_euler_kernel
    for c in range(nch):
        lo, hi = starts[c], starts[c + 1]
        last = max(i for i in range(lo, hi) if ev[i][2].startswith('void vla_euler_kernel'))
        span = (ev[last][1] - ev[lo][0]) / 1e6
        busy = sum(ev[i][1] - ev[i][0] for i in range(lo, last + 1)) / 1e6
        gap = (ev[hi][0] - ev[last][1]) / 1e3
        names = [short(ev[i][2]) + f' {((ev[i][1] - ev[i][0]) / 1e3):.1f}us' for i in range(last + 1, hi)]
        print(f'| {c} | {last + 1 - lo} | {span:.3f} | {busy:.3f} | {span - busy:.3f} | {gap:.1f} | {len(names)} |')
    lo, hi = starts[nch - 2], starts[nch - 1]
    last = max(i for i in range(lo, hi) if ev[i][2].startswith('void vla_euler_kernel'))
    print('\n## what runs between two chunks (chunk %d -> %d), us relative to the end of the last graph kernel\n' % (nch - 2, nch - 1))
    print('| kernel | start | dur |\n|---|---|---|')
    t0 = ev[last][1]
    for i in range(last + 1, hi + 1):
        print(f'| `{short(ev[i][2])}` | {(ev[i][0] - t0) / 1e3:.1f} | {(ev[i][1] - ev[i][0]) / 1e3:.1f} |')
    # per-class in-chain stats over the chunk
    agg = {}
    for i in range(lo, last + 1):
        n = short(ev[i][2])
        a = agg.setdefault(n, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += (ev[i][1] - ev[i][0]) / 1e3
        if i > lo:
            a[2] += (ev[i][0] - ev[i - 1][1]) / 1e3
    print('\n## one chunk by kernel class: calls, mean duration, mean gap BEFORE the kernel (us), total incl. gaps (ms)\n')
    print('| kernel | calls | dur us | gap-before us | total ms |\n|---|---|---|---|---|')
    for n, (c, t, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f'| `{n}` | {c} | {t / c:.2f} | {g / c:.2f} | {(t + g) / 1e3:.3f} |')
    # sequence of the first ViT layer, one prefill layer, and one euler layer-step
    def dump(title, a, b):
        print(f'\n## {title}\n\n| kernel | start us | dur us | gap-before us |\n|---|---|---|---|')
        t0 = ev[a][0]
        for i in range(a, b):
            print(f'| `{short(ev[i][2])}` | {(ev[i][0] - t0) / 1e3:.2f} | {(ev[i][1] - ev[i][0]) / 1e3:.2f} | {(ev[i][0] - ev[i - 1][1]) / 1e3:.2f} |')
    dump('chunk head: first 24 dispatches (patch embed + 2 ViT layers)', lo, lo + 24)
    # find first LLM prefill kernel: rmsnorm norm_kernel<true>? use gemm<6 (QKV_ROPE)
    q = [i for i in range(lo, last) if 'gemm_glds_kernel<6' in ev[i][2]]
    if len(q) > 3:
        dump('joint prefill: layers 2-3', q[2], q[4])
    s = [i for i in range(lo, last) if 'attn_skinny_kernel' in ev[i][2]]
    if len(s) > 60:
        dump('Euler phase: layer-steps 57-59', s[56] - 1, s[59] - 1)


if __name__ == '__main__':
    main()
