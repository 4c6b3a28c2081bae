"""The ZeRO-1 exchange at world size 8 without eight GPUs: the real Vlaser-2B bucket / shard plan, and dp.py's collectives run by
eight THREADS against a stand-in for torch.distributed that enforces what NCCL / RCCL enforce -- sizes (`input == world x output`) and the IN-PLACE
contract (`reduce_scatter`: recv == send + rank * count; `all_gather`: send == recv + rank * count; any other overlap is an error) -- and computes the
collectives' results.  Both branches of both collectives: buckets that are multiples of world * 128 (in place, no staging) and a ragged one (padded)."""
import threading

import pytest
import torch

from vlaser_amd import config as C, dp
from vlaser_amd.sft import plan_flat_layout

WORLD = 8


class _Group:
    def __init__(self, rank):
        self.rank = rank


class _FakeDist:
    """torch.distributed for `world` threads of one process; `group` carries the calling thread's rank."""

    class ReduceOp:
        AVG = 'avg'

    def __init__(self, world):
        self.world, self.bar, self.slots, self.calls = world, threading.Barrier(world), [None] * world, []

    def get_world_size(self, group=None):
        return self.world

    def get_backend(self, group=None):
        return 'nccl'

    @staticmethod
    def _span(t):
        return t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()

    def _in_place_ok(self, small, big, rank, what):
        (a0, a1), (b0, b1) = self._span(small), self._span(big)
        if a1 <= b0 or b1 <= a0:
            return 'out of place'
        count = small.numel() * small.element_size()
        assert a0 == b0 + rank * count, f'{what}: overlapping buffers must obey the in-place contract (offset {a0 - b0} != rank {rank} x {count} bytes)'
        return 'in place'

    def reduce_scatter_tensor(self, out, inp, op=None, group=None):
        r = group.rank
        assert op == self.ReduceOp.AVG and out.is_contiguous() and inp.is_contiguous() and out.dtype == inp.dtype
        assert inp.numel() == self.world * out.numel(), 'reduce_scatter_tensor: input must be world x output elements'
        mode = self._in_place_ok(out, inp, r, 'reduce_scatter_tensor')
        self.slots[r] = inp.float().clone()
        self.bar.wait()
        acc = self.slots[0].clone()
        for k in range(1, self.world):
            acc += self.slots[k]
        acc /= self.world
        n = out.numel()
        res = acc[r * n:(r + 1) * n].to(out.dtype)
        self.bar.wait()
        out.copy_(res)
        if r == 0:
            self.calls.append(('reduce_scatter', mode, n))

    def all_gather_into_tensor(self, out, inp, group=None):
        r = group.rank
        assert out.is_contiguous() and inp.is_contiguous() and out.numel() == self.world * inp.numel() and out.dtype == inp.dtype
        mode = self._in_place_ok(inp, out, r, 'all_gather_into_tensor')
        self.slots[r] = inp.clone()
        self.bar.wait()
        full = torch.cat(self.slots)
        self.bar.wait()
        out.copy_(full)
        if r == 0:
            self.calls.append(('all_gather', mode, inp.numel()))


def _run_ranks(fn):
    errs = []

    def work(r):
        try:
            fn(r)
        except BaseException as e:      # noqa: BLE001 -- reported below, with the rank
            errs.append((r, e))
            fake_abort()

    def fake_abort():
        try:
            _run_ranks.bar.abort()
        except Exception:               # noqa: BLE001
            pass

    th = [threading.Thread(target=work, args=(r,)) for r in range(WORLD)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errs, errs[0]


def test_vlaser_2b_shard_plan_at_world_8():
    """Every bucket of the real 2B layout is a multiple of 8 x 128 elements, its 8 shards are equal, 128-aligned and tile it exactly; the
    buckets tile the flat buffer; the per-rank shard sizes add up to 1/8 of the 1.78 G parameters (ZeRO-1's 1/N of the AdamW state)."""
    cfg = C.vlaser_2b()
    fp, buckets = plan_flat_layout(cfg, WORLD, bucket_layers=4)
    assert buckets[0][0] == 0 and all(a[1] == b[0] for a, b in zip(buckets, buckets[1:])) and buckets[-1][1] >= fp.n
    assert len(buckets) == 9
    per_rank = [0] * WORLD
    for lo, hi in buckets:
        assert (hi - lo) % (128 * WORLD) == 0
        shards = [dp.plan_shards([(lo, hi)], WORLD, r)[0] for r in range(WORLD)]
        assert shards[0][0] == lo and shards[-1][1] == hi
        for r, (s_lo, s_hi, per) in enumerate(shards):
            assert per * WORLD == hi - lo and s_hi - s_lo == per and s_lo == lo + r * per and s_lo % 128 == 0
            per_rank[r] += s_hi - s_lo
    assert len(set(per_rank)) == 1 and per_rank[0] * WORLD == buckets[-1][1]
    # every tensor lies inside exactly one bucket (a bucket boundary never cuts a gradient tensor)
    for name, shape, off in fp.specs:
        numel = 1
        for s_ in shape:
            numel *= s_
        assert sum(lo <= off and off + numel <= hi for lo, hi in buckets) == 1, name
    # world 1 .. 8: the same tensors in the same order, only the padding differs
    for w in (1, 2, 4):
        fw, bw = plan_flat_layout(cfg, w, bucket_layers=4)
        assert [s_[:2] for s_ in fw.specs] == [s_[:2] for s_ in fp.specs] and len(bw) == 9 and all((hi - lo) % (128 * w) == 0 for lo, hi in bw)


@pytest.mark.parametrize('ragged', [False, True])
def test_exchange_at_fake_world_8(monkeypatch, ragged):
    """dp.reduce_scatter_mean + dp.all_gather_params on 8 threads: aligned buckets take the in-place forms (checked against NCCL's in-place contract),
    a ragged bucket the padded forms; afterwards every rank holds the mean on its shard and, after the all-gather, identical parameter buffers."""
    fake = _FakeDist(WORLD)
    _run_ranks.bar = fake.bar
    monkeypatch.setattr(dp, 'dist', fake)
    if ragged:
        buckets = [(0, 128 * WORLD * 3), (128 * WORLD * 3, 128 * WORLD * 3 + 128 * 13 + 40)]       # the last bucket: 13.3 x 128 elements
    else:
        buckets = [(0, 128 * WORLD * 5), (128 * WORLD * 5, 128 * WORLD * 7), (128 * WORLD * 7, 128 * WORLD * 16)]
    n = buckets[-1][1]
    g0 = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)).to(torch.bfloat16) for r in range(WORLD)]
    flat_g = [g.clone() for g in g0]
    flat_p = [torch.full((n,), -1.0, dtype=torch.bfloat16) for _ in range(WORLD)]
    shards = [dp.plan_shards(buckets, WORLD, r) for r in range(WORLD)]

    def rank_fn(r):
        grp = _Group(r)
        for b, sh in zip(buckets, shards[r]):
            dp.reduce_scatter_mean(flat_g[r], b, sh, grp)
        for b, (s_lo, s_hi, per) in zip(buckets, shards[r]):
            flat_p[r][s_lo:s_hi] = flat_g[r][s_lo:s_hi] * 2 + r           # "the update": distinct per owner
        for b, sh in zip(buckets, shards[r]):
            dp.all_gather_params(flat_p[r], b, sh, grp)

    _run_ranks(rank_fn)
    mean = (sum(g.float() for g in g0) / WORLD).to(torch.bfloat16)
    owner = torch.empty(n, dtype=torch.long)
    for r in range(WORLD):
        for (s_lo, s_hi, per) in shards[r]:
            owner[s_lo:s_hi] = r
            assert torch.equal(flat_g[r][s_lo:s_hi], mean[s_lo:s_hi])
    for r in range(WORLD):
        mine = torch.zeros(n, dtype=torch.bool)
        for (s_lo, s_hi, per) in shards[r]:
            mine[s_lo:s_hi] = True
        assert torch.equal(flat_g[r][~mine], g0[r][~mine]), 'reduce_scatter_mean wrote outside its shard'
    want = ((mean * 2) + owner.to(torch.bfloat16)).to(torch.bfloat16)
    for r in range(WORLD):
        assert torch.equal(flat_p[r], flat_p[0])
    assert torch.equal(flat_p[0], want)
    modes = {(k, m) for k, m, _ in fake.calls}
    if ragged:
        assert ('reduce_scatter', 'out of place') in modes and ('all_gather', 'out of place') in modes and ('reduce_scatter', 'in place') in modes
    else:
        assert modes == {('reduce_scatter', 'in place'), ('all_gather', 'in place')}
