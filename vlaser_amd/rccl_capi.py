"""RCCL through its C API on a stream this package owns (`VLASER_DP_EXCHANGE=capi`; default stays torch's ProcessGroupNCCL).

Why: the data-parallel SFT step (DeepSpeed ZeRO-1 behind HF Trainer in the reference: internvl_chat_finetune.py:1041-1057,
zero_stage1_config.json `overlap_comm: true`, `reduce_scatter: true`) overlaps its bucketed reduce-scatter / all-gather with the backward /
next forward.  On MI355X a GEMM workgroup that shares its CU with a resident streaming workgroup -- which is what RCCL's channel kernels are --
stretches the forward + backward x1.23-1.34; with the two sides on DISJOINT CU sets the stretch is bounded at x1.13 whatever the channel count
(one-GPU stand-in: profiles/r05_rccl_contention.md, r05j_rccl_shadow_masks.md).  ProcessGroupNCCL issues its collectives on a stream of its own
that this package cannot mask; the C API takes any stream.  So: one extra communicator (`ncclCommInitRank`, the unique id broadcast over the
existing torch.distributed group), `ncclReduceScatter` / `ncclAllGather` / `ncclAllReduce` issued through ctypes on a stream created with
`hipExtStreamCreateWithCUMask` (vlaser_stream_create_cumask, ABI 7), the compute streams masked to the complement (sft.py).

The library loaded is the `librccl.so` torch itself ships and uses for backend "nccl" (same code, same version as the default path).
Nothing here computes: the collectives' own reduction is the only arithmetic.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L

# rccl.h (ROCm 7.x): ncclDataType_t / ncclRedOp_t values
NCCL_FLOAT32, NCCL_BFLOAT16 = 7, 9
NCCL_SUM, NCCL_AVG = 0, 4
_DT = {torch.float32: NCCL_FLOAT32, torch.bfloat16: NCCL_BFLOAT16}
UNIQUE_ID_BYTES = 128

_rccl = None


class RcclError(RuntimeError):
    pass


def _find_rccl():
    """torch's own copy first (the one ProcessGroupNCCL runs on), then the ROCm install."""
    cands = [os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), '/opt/rocm/lib/librccl.so', 'librccl.so']
    for c in cands:
        if os.path.sep not in c or os.path.exists(c):
            return c
    raise RcclError('librccl.so not found next to torch or under /opt/rocm/lib')


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_ubyte * UNIQUE_ID_BYTES)]       # (c_ubyte, not c_char: a c_char array field reads back truncated at its first NUL)


def rccl():
    global _rccl
    if _rccl is None:
        l = C.CDLL(_find_rccl())
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        l.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        l.ncclCommInitRank.argtypes = [C.POINTER(vp), i32, _UniqueId, i32]          # (comm*, nranks, id BY VALUE, rank)
        l.ncclCommDestroy.argtypes = [vp]
        l.ncclReduceScatter.argtypes = [vp, vp, sz, i32, i32, vp, vp]               # (send, recv, recvcount, dtype, op, comm, stream)
        l.ncclAllGather.argtypes = [vp, vp, sz, i32, vp, vp]                        # (send, recv, sendcount, dtype, comm, stream)
        l.ncclAllReduce.argtypes = [vp, vp, sz, i32, i32, vp, vp]                   # (send, recv, count, dtype, op, comm, stream)
        l.ncclGetErrorString.argtypes = [i32]
        l.ncclGetErrorString.restype = C.c_char_p
        l.ncclGetVersion.argtypes = [C.POINTER(i32)]
        for f in ('ncclGetUniqueId', 'ncclCommInitRank', 'ncclCommDestroy', 'ncclReduceScatter', 'ncclAllGather', 'ncclAllReduce', 'ncclGetVersion'):
            getattr(l, f).restype = i32
        _rccl = l
    return _rccl


def _ck(rc, what):
    if rc != 0:
        raise RcclError(f'{what}: {rccl().ncclGetErrorString(rc).decode()} ({rc})')


def masked_stream(first_cu, n_cus):
    """torch.cuda.ExternalStream over a HIP stream restricted to CUs [first_cu, first_cu + n_cus) (kept alive by the returned object)."""
    out = C.c_void_p()
    L.check(L.lib().vlaser_stream_create_cumask(int(first_cu), int(n_cus), C.byref(out)), 'vlaser_stream_create_cumask')
    s = torch.cuda.ExternalStream(out.value)
    s._vl_raw = out.value
    return s


class CapiExchange:
    """The four collectives of the ZeRO-1 step on an own communicator + an own (optionally CU-masked) stream.  `group` is the existing torch.distributed
    group: it only carries the unique id (one broadcast_object_list at construction).  In-place contracts as NCCL's: reduce-scatter's receive buffer is slice
    `rank` of its send buffer, all-gather's send buffer is slice `rank` of its receive buffer."""

    def __init__(self, group, device, comm_cus=0):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device)
        lib = rccl()
        uid = _UniqueId()
        if self.rank == 0:
            _ck(lib.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
        box = [C.string_at(C.byref(uid), UNIQUE_ID_BYTES) if self.rank == 0 else None]
        # the id travels over the group that already exists (rank 0 of the GROUP is the source)
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group, device=self.device if dist.get_backend(group) == 'nccl' else None)
        assert len(box[0]) == UNIQUE_ID_BYTES
        C.memmove(C.byref(uid), box[0], UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _ck(lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), 'ncclCommInitRank')
            total = torch.cuda.get_device_properties(self.device).multi_processor_count
            self.comm_cus = int(comm_cus)
            if self.comm_cus > 0:
                # RCCL's channel workgroups on the LAST comm_cus CUs; everything else of the step on the first total - comm_cus (sft.py masks its streams with compute_mask())
                self.stream = masked_stream(total - self.comm_cus, self.comm_cus)
            else:
                self.stream = torch.cuda.Stream(device=self.device)
            self.total_cus = total
        v = C.c_int()
        lib.ncclGetVersion(C.byref(v))
        self.version = v.value

    def compute_mask(self):
        """(first_cu, n_cus) the compute streams should be masked to, or None without masks."""
        return (0, self.total_cus - self.comm_cus) if self.comm_cus > 0 else None

    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    def reduce_scatter_avg(self, recv, send):
        assert send.numel() == recv.numel() * self.world and send.dtype == recv.dtype and send.is_contiguous() and recv.is_contiguous()
        _ck(rccl().ncclReduceScatter(send.data_ptr(), recv.data_ptr(), recv.numel(), _DT[send.dtype], NCCL_AVG, self.comm, self._s()), 'ncclReduceScatter')

    def all_gather(self, recv, send):
        assert recv.numel() == send.numel() * self.world and send.dtype == recv.dtype and send.is_contiguous() and recv.is_contiguous()
        _ck(rccl().ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), _DT[send.dtype], self.comm, self._s()), 'ncclAllGather')

    def all_reduce_sum(self, t):
        _ck(rccl().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DT[t.dtype], NCCL_SUM, self.comm, self._s()), 'ncclAllReduce')

    def destroy(self):
        if self.comm:
            self.stream.synchronize()
            rccl().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
