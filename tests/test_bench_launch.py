"""bench.py launch contract (CPU, no GPU): `python bench.py --gpus N` must itself start N ranks and
`torch.distributed.run ... bench.py --gpus N` must see world == N; a mismatch fails loudly."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    return e


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert lines, out
    return json.loads(lines[-1])


def test_gpus_2_spawns_two_ranks():
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run'], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2
    assert p.stdout.count('"metric"') == 1          # exactly one JSON line, from rank 0


def test_gpus_1_runs_in_process():
    p = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--dry-run'], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _last_json(p.stdout)['n_gpus'] == 1


def test_world_mismatch_fails_loudly():
    e = _env()
    e.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, BENCH, '--gpus', '4', '--dry-run'], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and 'WORLD_SIZE=1' in (p.stderr + p.stdout)


def test_failed_rank_after_the_line_is_a_failed_run():
    """a rank that dies (or the SFT watchdog firing) after rank 0 printed its line must not be reported as success."""
    e = _env()
    e['VLASER_BENCH_DRYRUN_EXIT'] = '3'
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run'], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert _last_json(p.stdout)['n_gpus'] == 2          # the line is still relayed
