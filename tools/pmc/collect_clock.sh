#!/bin/bash
# GPU-box helper: effective shader clock of a torch-free lab binary's kernels = GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md, DVFS give-back), per kernel name.
#   tools/pmc/collect_clock.sh <tag> <binary> [args...]   -> gpurun_out/<tag>_clock.md
set -e
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
tag="$1"; shift
out="$REPO/gpurun_out/${tag}_clock.md"
cd /tmp && export TMPDIR=/tmp
d="/tmp/pmc_clock"; rm -rf "$d"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$d" -- "$@" > /tmp/pmc_clock.log 2>&1 || { tail -3 /tmp/pmc_clock.log; exit 1; }
python3 - "$d" "$*" > "$out" <<'PY'
import csv, sys, glob, collections, os
d, cmd = sys.argv[1], sys.argv[2]
cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) if 'End_Timestamp' in r else None
    rows[(r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Dispatch_Id'])].append((float(r['Counter_Value']), dur))
per = collections.defaultdict(list)
for (k, did), v in rows.items():
    cyc = max(x[0] for x in v)           # the counter is reported per XCD / SE instance: each counts the same wall cycles
    dur = v[0][1]
    if dur: per[k].append((cyc, dur))
print(f'# effective shader clock per kernel = GRBM_GUI_ACTIVE (max over instances) / duration, under `rocprofv3 --pmc GRBM_GUI_ACTIVE`: {os.path.basename(cmd.split()[0])} {" ".join(cmd.split()[1:])}\n')
print('| kernel | dispatches | mean duration us | effective clock GHz |\n|---|---|---|---|')
for k, v in sorted(per.items()):
    v = v[len(v) // 4:]
    print(f'| `{k[:110]}` | {len(v)} | {sum(x[1] for x in v) / len(v) / 1e3:.2f} | {sum(x[0] for x in v) / sum(x[1] for x in v):.3f} |')
PY
cat "$out"
