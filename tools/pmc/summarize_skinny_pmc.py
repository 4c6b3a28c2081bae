"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc/collect_skinny_pmc.sh into <tag>_pmc_dominant_kernel.{md,json} (bench.py reads the json)."""
import csv, json, os, sys

raw, tag, out = sys.argv[1:4]


def col(name):
    vals = []
    with open(os.path.join(raw, name)) as f:
        for r in csv.DictReader(f):
            if 'chain_gu_kernel' in r['Kernel_Name'] or 'skinny_kernel' in r['Kernel_Name']:
                vals.append(float(r['Counter_Value']))
    return vals[len(vals) // 3:]          # drop the warm-up round (first third: HBM-cold allocations, code fetch)


fetch, write = col('fetch_size_counter_collection.csv'), col('write_size_counter_collection.csv')
fm, wm = sum(fetch) / len(fetch), sum(write) / len(write)
M, K, N, NP = 4, 768, 17920, 3
alg = N * K * 2 + M * K * 2 + NP * M * K * 4 + M * (N // 2) * 2
fetch_b = fm * 1024 * 2                   # gfx950: FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md, HBM)
write_b = wm * 1024
unprof = open(os.path.join(raw, 'unprofiled.log')).read().strip().splitlines()[-1] if os.path.exists(os.path.join(raw, 'unprofiled.log')) else ''
js = {'kernel': 'chain_gu_kernel<NS=3, UE=5, SP=3, CPT=1, TPU=1> (csrc/chain.hip: action-expert gate/up GEMV, N=17920 K=768 M=4, 3 split-K slabs of o_proj)',
      'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- tools/pmc/skinny_pmc 2 (separate passes; tools/pmc/collect_skinny_pmc.sh)',
      'fetch_size_kb_mean': round(fm, 2), 'write_size_kb_mean': round(wm, 2), 'fetch_bytes_corrected': fetch_b, 'write_bytes': write_b,
      'traffic_bytes_per_launch': int(fetch_b + write_b), 'algorithmic_bytes_per_launch': alg, 'traffic_over_algorithmic': round((fetch_b + write_b) / alg, 4)}
json.dump(js, open(os.path.join(out, f'{tag}_pmc_dominant_kernel.json'), 'w'), indent=1)
with open(os.path.join(out, f'{tag}_pmc_dominant_kernel.md'), 'w') as f:
    f.write(f'# {tag} -- HBM traffic of the dominant kernel from hardware counters\n\n')
    f.write('`chain_gu_kernel<3,5,3,1,1>` (r05; r01-r04: `skinny_kernel<NORM,SWIGLU,2,3,SP=3>`), N=17920, K=768, M=4, 3 slabs; harness `tools/pmc/skinny_pmc.cpp` (torch-free), 28 distinct 27.5 MB weight buffers cycled;\n')
    f.write('separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes with `--kernel-trace` only (`tools/pmc/collect_skinny_pmc.sh`); raw CSVs next to this file.\n\n')
    f.write('| counter | dispatches used | mean (KB) | min | max |\n|---|---|---|---|---|\n')
    f.write(f'| FETCH_SIZE | {len(fetch)} | {fm:.1f} | {min(fetch):.1f} | {max(fetch):.1f} |\n| WRITE_SIZE | {len(write)} | {wm:.1f} | {min(write):.1f} | {max(write):.1f} |\n\n')
    f.write('gfx950 correction (guide, HBM section): FETCH_SIZE counts the 128-B requests of wide coalesced reads at 64 B -> x2.\n')
    f.write(f'HBM read = {fm:.1f} KB x 1024 x 2 = {fetch_b / 1e6:.2f} MB, write = {write_b / 1e3:.1f} KB -> traffic **{(fetch_b + write_b) / 1e6:.2f} MB** per launch vs '
            f'{alg / 1e6:.2f} MB algorithmic = {(fetch_b + write_b) / alg:.3f}x.\n')
    if unprof:
        f.write(f'Un-profiled run of the same harness: {unprof}\n')
print(open(os.path.join(out, f'{tag}_pmc_dominant_kernel.md')).read())
