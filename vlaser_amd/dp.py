"""Data-parallel gradient exchange of the SFT step: bucketed mean reduce-scatter of the flat bf16
gradient buffer + all-gather of the updated bf16 parameters (ZeRO-1 ownership: rank r owns slice r of every bucket).

One process per GPU; `backend="nccl"` is RCCL over xGMI on ROCm.  The same code runs over `gloo` on CPU tensors for the
world_size-2 tests (gloo has no reduce_scatter: emulated with all_reduce there).  No arithmetic besides the collective's
own reduction happens here -- the kernels own the math.
"""
import torch
import torch.distributed as dist


def plan_shards(buckets, world, rank, align=128):
    """For each bucket [lo, hi): (shard_lo, shard_hi, per) with per = aligned ceil(len / world); rank r owns
    [lo + r*per, min(lo + (r+1)*per, hi))."""
    out = []
    for lo, hi in buckets:
        n = hi - lo
        per = (n + world - 1) // world
        per = (per + align - 1) // align * align
        s_lo = min(lo + rank * per, hi)
        s_hi = min(s_lo + per, hi)
        out.append((s_lo, s_hi, per))
    return out


def reduce_scatter_mean(flat_g, bucket, shard, group=None, capi=None):
    """Average bucket `flat_g[lo:hi]` over the group; this rank's slice of the average is written back in place
    (other positions of the bucket are left untouched).  `capi`: a `rccl_capi.CapiExchange` -- the same collective through RCCL's C API on that
    object's own stream (the CALLER orders that stream against the gradients; nothing is switched here)."""
    lo, hi = bucket
    s_lo, s_hi, per = shard
    world = dist.get_world_size(group)
    n = hi - lo
    inp = flat_g[lo:hi]
    if capi is not None:
        if n == per * world:
            capi.reduce_scatter_avg(flat_g[s_lo:s_hi], inp)
            return
        with torch.cuda.stream(capi.stream):                    # ragged tail: pad the collective's input (allocations follow the stream they are used on)
            pad = torch.zeros(per * world, dtype=flat_g.dtype, device=flat_g.device)
            pad[:n] = inp
            out = torch.empty(per, dtype=flat_g.dtype, device=flat_g.device)
            capi.reduce_scatter_avg(out, pad)
            if s_hi > s_lo:
                flat_g[s_lo:s_hi].copy_(out[:s_hi - s_lo])
        return
    if dist.get_backend(group) == 'gloo':                       # CPU test path: gloo lacks reduce_scatter (and bf16 reductions)
        tmp = inp.float()
        dist.all_reduce(tmp, group=group)
        tmp /= world
        if s_hi > s_lo:
            flat_g[s_lo:s_hi] = tmp[s_lo - lo:s_hi - lo].to(flat_g.dtype)
        return
    if n == per * world:                                        # the SFT buckets are padded to world*128: in place, no copies
        dist.reduce_scatter_tensor(flat_g[s_lo:s_hi], inp, op=dist.ReduceOp.AVG, group=group)
        return
    pad = torch.zeros(per * world, dtype=flat_g.dtype, device=flat_g.device)      # ragged tail: pad the collective's input
    pad[:n] = inp
    out = torch.empty(per, dtype=flat_g.dtype, device=flat_g.device)
    dist.reduce_scatter_tensor(out, pad, op=dist.ReduceOp.AVG, group=group)
    if s_hi > s_lo:
        flat_g[s_lo:s_hi].copy_(out[:s_hi - s_lo])


def all_gather_params(flat_p, bucket, shard, group=None, capi=None):
    """Every rank contributes its updated slice of the bucket; afterwards flat_p[lo:hi] is identical on all ranks."""
    lo, hi = bucket
    s_lo, s_hi, per = shard
    world = dist.get_world_size(group)
    if capi is not None:
        if hi - lo == per * world:
            capi.all_gather(flat_p[lo:hi], flat_p[s_lo:s_hi])
            return
        with torch.cuda.stream(capi.stream):
            mine = torch.zeros(per, dtype=flat_p.dtype, device=flat_p.device)
            if s_hi > s_lo:
                mine[:s_hi - s_lo] = flat_p[s_lo:s_hi]
            full = torch.empty(per * world, dtype=flat_p.dtype, device=flat_p.device)
            capi.all_gather(full, mine)
            flat_p[lo:hi].copy_(full[:hi - lo])
        return
    if dist.get_backend(group) != 'gloo' and hi - lo == per * world:          # in place: slice r of the bucket is rank r's input
        dist.all_gather_into_tensor(flat_p[lo:hi], flat_p[s_lo:s_hi], group=group)
        return
    mine = torch.zeros(per, dtype=flat_p.dtype, device=flat_p.device)
    if s_hi > s_lo:
        mine[:s_hi - s_lo] = flat_p[s_lo:s_hi]
    if dist.get_backend(group) == 'gloo':
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        full = torch.cat(parts)
    else:
        full = torch.empty(per * world, dtype=flat_p.dtype, device=flat_p.device)
        dist.all_gather_into_tensor(full, mine, group=group)
    flat_p[lo:hi].copy_(full[:hi - lo])
