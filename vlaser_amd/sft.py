"""SFT-step pieces (modeling_internvl_chat.py:204-243 + the data-parallel step of SURVEY.md 8 a15).
Round 1 holds the loss head only; the trainable step (backward kernels, fused AdamW, bucketed RCCL gradient
reduction) is the next row of the scope table."""
import torch

from . import _lib as L
from .ops import _stream


def ce_loss(logits, labels, ignore_index=-100):
    """Mean cross entropy over labels != ignore_index of fp32 logits [R, V] (already shifted)."""
    assert logits.dtype == torch.float32 and logits.is_cuda
    logits = logits if logits.is_contiguous() else logits.contiguous()
    labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
    R, V = logits.shape
    rows = torch.empty(R, dtype=torch.float32, device=logits.device)
    L.check(L.lib().vlaser_ce_rows(logits.data_ptr(), labels.data_ptr(), R, V, logits.stride(0), rows.data_ptr(), None,
                                   ignore_index, _stream()), 'vlaser_ce_rows')
    n = (labels != ignore_index).sum().clamp(min=1)
    return rows.sum() / n
