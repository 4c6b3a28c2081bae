"""SFT step parity (GPU): loss and gradients of the HIP backward vs torch autograd through the fp32 CPU oracle on the same
inputs / weights (truncated true-width model); fused AdamW vs torch.optim.AdamW; op-level checks of the backward kernels.

Tolerances (bf16 params / activations / grads vs an fp32 reference): per-tensor relative Frobenius error <= 4e-2 and cosine
similarity >= 0.999; loss |err| <= 5e-3."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item(), F.cosine_similarity(a, b, dim=0).item()


@pytest.fixture(scope='module')
def ops():
    from vlaser_amd import ops as o
    return o


@pytest.fixture(scope='module')
def setup(golden_model, golden_dir):
    from vlaser_amd.sft import SFTModel
    cfg, _, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g5g6_vlm.npz'))
    ids = torch.from_numpy(d['input_ids'])
    pv = torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(0))
    labels = torch.full_like(ids, -100)
    labels[0, -16:] = ids[0, -16:]
    m = SFTModel(cfg, max_seq_len=ids.shape[1], lr=1e-3, weight_decay=0.05)
    m.load_state_dict(sd)
    return cfg, sd, m, pv, ids, labels, float(d['sft_loss'])


def test_loss_and_grads_vs_oracle_autograd(setup):
    from oracle import vlm as ovlm
    cfg, sd, m, pv, ids, labels, golden_loss = setup
    loss = m.forward_backward(pv, ids, labels)
    assert abs(loss.item() - golden_loss) < 5e-3            # reference's own loss value (golden G5)
    grads = m.named_grads()
    # fp32 autograd through the CPU oracle; the ViT is frozen (freeze_backbone) -> leaves without grad
    torch.set_grad_enabled(True)
    try:
        sdg = {}
        for k, v in sd.items():
            if k.startswith(('language_model.', 'mlp1.')):
                sdg[k] = v.clone().requires_grad_(True)
            elif k.startswith('vision_model.'):
                sdg[k] = v
        logits = ovlm.forward_logits(sdg, cfg, pv, ids)
        ref_loss = ovlm.sft_loss(logits, labels)
        ref_loss.backward()
    finally:
        torch.set_grad_enabled(False)
    assert abs(loss.item() - ref_loss.item()) < 5e-3
    checked = 0
    worst = (0, '')
    for k, g in grads.items():
        ref = sdg[k].grad
        if ref is None:
            continue
        if k == 'language_model.model.embed_tokens.weight':
            rows = ids[0][ids[0] != cfg.img_context_token_id].unique()
            rel, cos = _rel(g[rows.cuda()], ref[rows])
            assert g.float().abs().sum().item() == pytest.approx(g[rows.cuda()].float().abs().sum().item(), rel=1e-6)   # nothing outside the text rows
        else:
            rel, cos = _rel(g, ref)
        worst = max(worst, (rel, k))
        assert cos > 0.999 and rel < 4e-2, (k, rel, cos)
        checked += 1
    assert checked >= 2 * 12 + 9
    print('worst relative gradient error', worst)


def test_grads_vs_reference_golden(setup, golden_dir):
    """HIP backward against the gradients of the reference model itself (tests/golden/g8_sft_grads.npz, reference forward + torch
    autograd in fp32): per-tensor norm within 3 %, sampled entries within 4e-2 of the tensor's scale."""
    cfg, sd, m, pv, ids, labels, _ = setup
    d = np.load(os.path.join(golden_dir, 'g8_sft_grads.npz'))
    loss = m.forward_backward(pv, ids, labels)
    assert abs(loss.item() - float(d['loss'])) < 5e-3
    grads = m.named_grads()
    names = [str(n) for n in d['names']]
    assert set(names) == set(grads)
    for n in names:
        g = grads[n].double().flatten().cpu()
        ref_norm = float(d[f'norm::{n}'])
        assert abs(g.norm().item() - ref_norm) <= 3e-2 * ref_norm, (n, g.norm().item(), ref_norm)
        idx, val = torch.from_numpy(d[f'idx::{n}']), torch.from_numpy(d[f'val::{n}'])
        scale = max(val.abs().max().item(), ref_norm / g.numel() ** 0.5)
        assert (g[idx] - val).abs().max().item() <= 4e-2 * scale + 1e-12, n


def test_recompute_mode_gives_identical_grads(setup):
    """Per-layer recompute (the reference's grad_checkpoint) and the default keep-activations backward run the same
    kernels on the same values: gradients agree bit for bit."""
    from vlaser_amd.sft import SFTModel
    cfg, sd, m, pv, ids, labels, _ = setup
    loss = m.forward_backward(pv, ids, labels)
    g_keep = {k: v.clone() for k, v in m.named_grads().items()}
    m2 = SFTModel(cfg, max_seq_len=ids.shape[1], lr=1e-3, weight_decay=0.05, recompute=True)
    m2.load_state_dict(sd)
    assert m2.x1.shape[0] == 1 and m.x1.shape[0] == cfg.llm.num_hidden_layers
    loss2 = m2.forward_backward(pv, ids, labels)
    assert loss.item() == loss2.item()
    for k, g in m2.named_grads().items():
        assert torch.equal(g, g_keep[k]), k


def test_adamw_matches_torch(setup):
    from vlaser_amd import ops
    n = 100_000
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g) * 0.05
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    p = p0.to(BF).cuda(); master = p0.cuda().clone(); m = torch.zeros(n).cuda(); v = torch.zeros(n).cuda()
    for step in range(1, 4):
        gr = (torch.randn(n, generator=g) * 0.01).to(BF)
        ref.grad = gr.float()
        opt.step()
        ops.adamw(p, master, m, v, gr.cuda(), 1e-3, 0.9, 0.999, 1e-8, 0.05, 1.0, step)
        torch.testing.assert_close(master.cpu(), ref.detach(), rtol=2e-5, atol=2e-7)
    assert torch.equal(p.cpu(), master.cpu().to(BF))


def test_step_reduces_loss_and_exports_hf_names(setup):
    cfg, sd, m, pv, ids, labels, _ = setup
    l0 = m.forward_backward(pv, ids, labels).item()
    out = m.step(pv, ids, labels)
    assert out.grad_norm > 0
    l1 = m.forward_backward(pv, ids, labels).item()
    assert l1 < l0, (l0, l1)
    exported = m.state_dict()
    trainable = {k for k in sd if k.startswith(('language_model.', 'mlp1.'))}
    assert set(exported) == trainable
    for k in trainable:
        assert exported[k].shape == sd[k].shape, k


def test_backward_kernels_oplevel():
    from vlaser_amd import ops, _lib as L
    S, C = 77, 1536
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(S, C, generator=g) * 2).to(BF).cuda(); dy = torch.randn(S, C, generator=g).to(BF).cuda()
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(BF).cuda(); dres = torch.randn(S, C, generator=g).to(BF).cuda()
    xr = x.float().requires_grad_(True); wr = w.float().requires_grad_(True)
    with torch.enable_grad():
        y = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6) * wr
        y.backward(dy.float())
    dx = torch.zeros_like(x)
    ops.rmsnorm_bwd(dy, x, w, dres, dx, S, C, 1e-6)
    rel, cos = _rel(dx, xr.grad + dres.float())
    assert rel < 1e-2 and cos > 0.9999
    bsum = torch.full((C,), 7.0, dtype=BF, device='cuda')
    ops.colsum_bf16(dy, bsum, S, C)                                  # bias gradient of a Linear: column sum in one launch
    assert _rel(bsum, dy.float().sum(0))[0] < 5e-3
    # same call with the weight gradient fused into the pass
    dx2 = torch.zeros_like(x); dw = torch.full((C,), 7.0, dtype=BF, device='cuda'); dw_ws = torch.zeros((S + 3) // 4 * C, device='cuda')
    ops.rmsnorm_bwd(dy, x, w, dres, dx2, S, C, 1e-6, dw_out=dw, dw_ws=dw_ws)
    assert torch.equal(dx2, dx)
    rel, cos = _rel(dw, wr.grad)
    assert rel < 1e-2 and cos > 0.9999
    col = torch.zeros(C, device='cuda'); ws = torch.zeros(2 * S + 16 * C, device='cuda')
    ops.colsum_mul(dy, x, col, S, C, 2, 1e-6, ws)
    rel, cos = _rel(col, wr.grad)
    assert rel < 1e-2
    # transpose with padding
    out = torch.full((C, 128), 7, dtype=BF, device='cuda')
    ops.transpose(x, out, S, C, C, 128)
    assert torch.equal(out[:, :S], x.t()) and (out[:, S:] == 0).all()
    # two-level batch (the grouped Q^T / dO^T of the attention backward): [S2, (kvh, g, d)] -> [kvh][d][g*Sp + s]
    S2, nkv, G_, hd, Sp = 40, 2, 3, 64, 64
    qx = torch.randn(S2, nkv * G_ * hd, generator=g).to(BF).cuda()
    qt = torch.full((nkv, hd, G_ * Sp), 7, dtype=BF, device='cuda')
    ops.transpose(qx, qt, S2, hd, nkv * G_ * hd, G_ * Sp, Sp, nkv, G_ * hd, hd * G_ * Sp, inner=G_, in_is=hd, out_is=Sp)
    ref = torch.zeros(nkv, hd, G_, Sp, dtype=BF, device='cuda')
    ref[:, :, :, :S2] = qx.view(S2, nkv, G_, hd).permute(1, 3, 2, 0)
    assert torch.equal(qt.view(nkv, hd, G_, Sp), ref)
    # swiglu fwd / bwd on the packed layout
    I = 64
    gu_nat = torch.randn(S, 2, I, generator=g)
    gu = torch.stack([gu_nat[:, 0].view(S, I // 16, 16), gu_nat[:, 1].view(S, I // 16, 16)], 2).reshape(S, 2 * I).to(BF).cuda()
    act = torch.zeros(S, I, dtype=BF, device='cuda')
    ops.swiglu(gu, act, S, I)
    gg, uu = gu_nat[:, 0].to(BF).float(), gu_nat[:, 1].to(BF).float()
    rel, _ = _rel(act, F.silu(gg) * uu)
    assert rel < 1e-2
    dact = torch.randn(S, I, generator=g).to(BF).cuda(); dgu = torch.zeros_like(gu)
    ops.swiglu_bwd(gu, dact, dgu, S, I)
    ggr, uur = gg.clone().requires_grad_(True), uu.clone().requires_grad_(True)
    with torch.enable_grad():
        (F.silu(ggr) * uur).backward(dact.float().cpu())
    d = dgu.float().cpu().view(S, I // 16, 2, 16)
    assert _rel(d[:, :, 0].reshape(S, I), ggr.grad)[0] < 1e-2 and _rel(d[:, :, 1].reshape(S, I), uur.grad)[0] < 1e-2
    # batched GEMM
    A = torch.randn(3, 40, 128, generator=g).to(BF).cuda(); W = torch.randn(3, 50, 128, generator=g).to(BF).cuda()
    o = torch.zeros(3, 40, 64, dtype=torch.float32, device='cuda')
    ops.gemm_raw(L.EPI_F32, A, W, o, 40, 50, 128, 128, 128, 64, batch=3, a_bs=40 * 128, w_bs=50 * 128, o_bs=40 * 64, w_group=1)
    rel, _ = _rel(o[:, :, :50], A.float() @ W.float().transpose(1, 2))
    assert rel < 5e-3


@pytest.mark.parametrize('K,M,N', [(560, 1536, 2048), (313, 256, 1536), (64, 8, 136), (130, 17920, 1536), (48, 1002, 520)])
def test_gemm_tn_weight_gradient(K, M, N):
    """out = At^T @ Wt with the contraction along the rows of both operands (transposing LDS reads): ragged K (zero-filled
    tail rows), edge tiles in M and N, strided operand views."""
    from vlaser_amd import ops
    g = torch.Generator().manual_seed(K + M)
    big_a = torch.randn(K, (M + 7) // 8 * 8 + 16, generator=g).to(BF).cuda()
    big_w = torch.randn(K + 3, N, generator=g).to(BF).cuda()
    At, Wt = big_a[:, 8:8 + M], big_w[:K]                     # column-offset view (16-byte aligned) and a row-truncated view
    out = torch.full((M, N), 7.0, dtype=BF, device='cuda')
    ops.gemm_tn(At, Wt, out)
    ref = At.float().t() @ Wt.float()
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item() + 1e-3, err
    rel, cos = _rel(out, ref)
    assert rel < 5e-3 and cos > 0.9999


@pytest.mark.parametrize('K,M,N,cfg', [(560, 1536, 2048, 0), (560, 17920, 1536, 0), (560, 1536, 8960, 0), (313, 256, 1536, 1100), (64, 136, 264, 1105),
                                       (130, 2048, 1536, 1200), (200, 1000, 520, 1300), (560, 17920, 1536, 1340), (200, 1000, 520, 1340), (313, 264, 1536, 1240), (64, 136, 264, 1140),
                                       (96, 1536, 2048, 1140), (32, 304, 264, 1340), (313, 256, 1536, 1110), (130, 2048, 1536, 1210)])
def test_gemm_tn_lds_padded_contraction(K, M, N, cfg):
    """vlaser_gemm_tn_lds (the TN weight-gradient product on the LDS-DMA pipeline, both operands k-major): contraction axis padded to 64-row
    tiles with ZERO pad rows in At and arbitrary finite pad rows in Wt; edge tiles in M and N; every tile configuration.  Against fp32 and
    against the register-staged vlaser_gemm_tn on the unpadded views (same products, different summation order: bf16-rounding tolerance)."""
    from vlaser_amd import ops
    g = torch.Generator().manual_seed(K + M + N)
    Kp = (K + 63) // 64 * 64
    At = torch.zeros(Kp, M, dtype=BF, device='cuda'); Wt = torch.randn(Kp, N, generator=g).to(BF).cuda()       # Wt pad rows: finite garbage
    At[:K] = torch.randn(K, M, generator=g).to(BF).cuda()
    out = torch.full((M, N), 7.0, dtype=BF, device='cuda'); out_old = torch.zeros_like(out)
    ops.gemm_tn_lds(At, Wt, out, Kp, force_cfg=cfg)
    ops.gemm_tn(At[:K], Wt[:K], out_old)
    ref = At[:K].float().t() @ Wt[:K].float()
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item() + 1e-3, err
    rel, cos = _rel(out, ref)
    assert rel < 5e-3 and cos > 0.9999
    assert (out.float() - out_old.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()


def test_embedding_gradient_row_clear_equals_full_clear(setup, monkeypatch):
    """r04: inside train_step's one-sample path only the embedding-gradient rows the previous step touched are cleared (not the 466 MB table).  Steps over
    samples with DIFFERENT token ids (A, B, A, B): parameters, losses and gradient norms bit-identical to the same steps with the full clear."""
    from vlaser_amd.sft import SFTModel
    cfg, sd, _, pv, ids, labels, _ = setup
    ids_b = ids.clone()
    txt = ids_b != cfg.img_context_token_id
    ids_b[txt] = (ids_b[txt] * 7 + 13) % 150000 + 1            # other text tokens, same image positions
    lab_b = labels.clone()
    lab_b[labels != -100] = ids_b[labels != -100]

    def run(full):
        monkeypatch.setenv('VLASER_SFT_EMBED_FULL_CLEAR', '1' if full else '0')
        m = SFTModel(cfg, max_seq_len=ids.shape[1], lr=1e-3, weight_decay=0.05, max_grad_norm=1.0)
        m.load_state_dict(sd)
        outs = [m.step(pv, i_, l_) for i_, l_ in ((ids, labels), (ids_b, lab_b), (ids, labels), (ids_b, lab_b))]
        m.wait_optimizer()
        return [o.loss.item() for o in outs], [o.grad_norm.item() for o in outs], m.state_dict()['language_model.model.embed_tokens.weight'].clone(), m.fp.gview['embed'].clone()

    la, na, ea, ga = run(False)
    lb, nb, eb, gb = run(True)
    assert la == lb and na == nb
    assert torch.equal(ea, eb) and torch.equal(ga, gb)
    assert int((ga.float().abs().sum(1) > 0).sum()) <= int(torch.unique(ids_b).numel())       # only the last step's rows hold gradients


@pytest.mark.parametrize('cfg', [1340, 1240, 1140])
def test_gemm_tn_staggered_race_screen(cfg):
    """The staggered two-wave-group TN kernel has its own synchronisation structure (refill two phases after the last read, read one phase after the counted vmcnt):
    30 runs on the same operands, with an unrelated GEMM stream running beside them to move the timing around, must be bit-identical to the first run and within
    bf16 rounding of fp32 -- an early LDS read would show as a run that differs."""
    from vlaser_amd import ops, _lib as L
    g = torch.Generator().manual_seed(cfg)
    K, M, N = 576, 2048 + 24, 1536 + 8
    At = torch.zeros(K, M, dtype=BF, device='cuda'); Wt = torch.randn(K, N, generator=g).to(BF).cuda()
    At[:560] = torch.randn(560, M, generator=g).to(BF).cuda()
    ref = At.float().t() @ Wt.float()
    side = torch.cuda.Stream()
    a2, w2, o2 = torch.randn(1024, 1024, device='cuda').to(BF), torch.randn(4096, 1024, device='cuda').to(BF), torch.empty(1024, 4096, dtype=BF, device='cuda')
    first = None
    for it in range(30):
        out = torch.full((M, N), 3.0, dtype=BF, device='cuda')
        if it % 2:
            with torch.cuda.stream(side):
                for _ in range(1 + it % 5):
                    ops.gemm(L.EPI_NONE, a2, w2, out=o2)
        ops.gemm_tn_lds(At, Wt, out, K, force_cfg=cfg)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            assert (out.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-3
        else:
            assert torch.equal(out, first), it


def test_rmsnorm_bwd_from_split_k_slabs():
    """r04: vlaser_rmsnorm_bwd fed the fp32 split-K slabs of the dgrad before it == vlaser_reduce_norm into bf16 followed by vlaser_rmsnorm_bwd, bit for bit
    (dx and the weight gradient); ragged row count, 1 .. 8 slabs."""
    from vlaser_amd import ops
    g = torch.Generator().manual_seed(21)
    for S, C, n in ((561, 1536, 8), (70, 768, 3), (5, 1536, 1)):
        slabs = (torch.randn(n, S, C, generator=g) * 0.3).cuda()
        x = torch.randn(S, C, generator=g).to(BF).cuda(); w = (1 + 0.1 * torch.randn(C, generator=g)).to(BF).cuda(); dres = torch.randn(S, C, generator=g).to(BF).cuda()
        dy = torch.empty(S, C, dtype=BF, device='cuda')
        ops.reduce_norm(None, slabs, n, S, C, dy)
        ws = torch.zeros(((S + 3) // 4) * C, device='cuda')
        dx_a, dx_b = torch.empty_like(dy), torch.empty_like(dy)
        dw_a, dw_b = torch.empty(C, dtype=BF, device='cuda'), torch.empty(C, dtype=BF, device='cuda')
        ops.rmsnorm_bwd(dy, x, w, dres, dx_a, S, C, 1e-6, dw_out=dw_a, dw_ws=ws)
        ops.rmsnorm_bwd(None, x, w, dres, dx_b, S, C, 1e-6, dw_out=dw_b, dw_ws=ws, dy_partials=slabs, n_partials=n)
        assert torch.equal(dx_a, dx_b) and torch.equal(dw_a, dw_b), (S, C, n)
        ops.rmsnorm_bwd(None, x, w, None, dx_b, S, C, 1e-6, dy_partials=slabs, n_partials=n)              # no residual, no weight gradient
        ops.rmsnorm_bwd(dy, x, w, None, dx_a, S, C, 1e-6)
        assert torch.equal(dx_a, dx_b)


_RB_SCRIPT = """
import hashlib, sys, torch
sys.path.insert(0, %r)
from vlaser_amd import ops
BF = torch.bfloat16
g = torch.Generator().manual_seed(33)
h = hashlib.sha256()
for S, C, n in ((561, 1536, 3), (70, 4096, 0), (130, 1024, 2), (9, 3584, 0), (33, 2048, 5), (64, 512, 1)):
    slabs = (torch.randn(max(n, 1), S, C, generator=g) * 0.3).cuda()
    x = torch.randn(S, C, generator=g).to(BF).cuda(); w = (1 + 0.1 * torch.randn(C, generator=g)).to(BF).cuda(); dres = torch.randn(S, C, generator=g).to(BF).cuda()
    dy = torch.randn(S, C, generator=g).to(BF).cuda()
    ws = torch.zeros(((S + 3) // 4) * C, device='cuda'); dx = torch.empty_like(dy); dw = torch.empty(C, dtype=BF, device='cuda')
    if n:
        ops.rmsnorm_bwd(None, x, w, dres, dx, S, C, 1e-6, dw_out=dw, dw_ws=ws, dy_partials=slabs, n_partials=n)
    else:
        ops.rmsnorm_bwd(dy, x, w, dres, dx, S, C, 1e-6, dw_out=dw, dw_ws=ws)
    h.update(dx.view(torch.int16).cpu().numpy().tobytes()); h.update(dw.view(torch.int16).cpu().numpy().tobytes())
print('SHA', h.hexdigest())
"""


def test_rmsnorm_bwd_register_kernel_bit_identical_to_loops():
    """r04: the register-resident vlaser_rmsnorm_bwd (C = 512 * n, one load round trip) writes the same bits as the chunk-loop kernel it replaces
    (VLASER_RMSNORM_BWD_LOOPS=1, read once per process: two child processes): dx and the norm-weight gradient, bf16 dy and split-K slabs, C = 512 .. 4096."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for loops in ('0', '1'):
        env = dict(os.environ, VLASER_RMSNORM_BWD_LOOPS=loops)
        r = subprocess.run([sys.executable, '-c', _RB_SCRIPT % root], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith('SHA')][0])
    assert outs[0] == outs[1], outs


@pytest.mark.parametrize('K,M,N,cfg,lds', [(576, 1536, 2048, 0, True), (576, 17920, 1536, 0, True), (320, 1000, 520, 1300, True), (64, 136, 264, 1105, True),
                                           (128, 2048, 1536, 1200, True), (576, 17920, 1536, 1340, True), (320, 1000, 520, 1240, True), (100, 304, 200, 0, False), (200, 1000, 1528, 0, False)])
def test_gemm_tn_sumsq_slots(K, M, N, cfg, lds):
    """`sumsq_part` of the weight-gradient GEMMs (r04): the slots the launch writes add up to the sum of the squares of the bf16 values it stored (the
    gradient norm's share of the tensor, without reading it back) -- every tile configuration, ragged edges, both kernels; untouched slots stay as
    they were; too few slots is refused."""
    from vlaser_amd import ops, _lib as L
    g = torch.Generator().manual_seed(K + M + N)
    At = torch.randn(K, M, generator=g).to(BF).cuda(); Wt = torch.randn(K, N, generator=g).to(BF).cuda()
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    cap = ops.tn_sumsq_slots(M, N)
    part = torch.full((cap,), -1.0, device='cuda')
    if lds:
        ops.gemm_tn_lds(At, Wt, out, K, force_cfg=cfg, sumsq_part=part)
    else:
        L.check(L.lib().vlaser_gemm_tn(At.data_ptr(), Wt.data_ptr(), out.data_ptr(), M, N, K, At.stride(0), Wt.stride(0), out.stride(0), part.data_ptr(), cap, None), 'tn')
    written = part >= 0
    assert 0 < int(written.sum()) <= cap
    want = out.double().pow(2).sum().item()
    got = part[written].double().sum().item()
    assert abs(got - want) <= 1e-5 * want, (got, want)
    tot = torch.zeros(1, device='cuda')
    part[~written] = 0
    ops.sum_partials(part, tot, accumulate=False)
    tot2 = torch.full((1,), 3.0, device='cuda')
    ops.sum_partials(part, tot2)
    assert abs(tot.item() - want) <= 1e-5 * want and tot2.item() == (torch.tensor(3.0) + torch.tensor(tot.item())).item()     # out[0] + sum, in fp32
    with pytest.raises(L.VlaserHipError, match='slots'):
        if lds:
            ops.gemm_tn_lds(At, Wt, out, K, force_cfg=cfg, sumsq_part=part[:3])
        else:
            L.check(L.lib().vlaser_gemm_tn(At.data_ptr(), Wt.data_ptr(), out.data_ptr(), M, N, K, At.stride(0), Wt.stride(0), out.stride(0), part.data_ptr(), 3, None), 'tn')


def test_sumsq_chunks_and_rows():
    """The small-tensor and embedding-row legs of the producer-side gradient norm: chunk table with unaligned / odd-length chunks; rows of a table
    listed through the id-sorted order (duplicates counted once, out-of-range ids skipped, the slots past n zeroed)."""
    from vlaser_amd import ops, _lib as L
    g = torch.Generator().manual_seed(11)
    x = torch.randn(100_000, generator=g).to(BF).cuda()
    chunks = [(0, 8192), (8192, 100), (8293, 4097), (20000, 1), (20008, 8192), (99990, 10)]
    tab = torch.tensor(chunks, dtype=torch.int64, device='cuda')
    part = torch.full((len(chunks),), -1.0, device='cuda')
    ops.sumsq_chunks(x, tab, part)
    for (o, n), p in zip(chunks, part.tolist()):
        want = x[o:o + n].double().pow(2).sum().item()
        assert abs(p - want) <= 1e-5 * want + 1e-12, (o, n, p, want)
    V, H, n = 500, 64, 40
    table = torch.randn(V, H, generator=g).to(BF).cuda()
    ids = torch.randint(0, V, (n,), generator=g)
    ids[5] = ids[17] = ids[3]; ids[9] = 10 ** 6
    ids = ids.cuda()
    order = torch.sort(ids, stable=True).indices.to(torch.int32)
    rp = torch.full((48,), -1.0, device='cuda')
    L.check(L.lib().vlaser_sumsq_rows(ids.data_ptr(), order.data_ptr(), table.data_ptr(), n, H, V, rp.data_ptr(), 48, None), 'rows')
    uniq = [i for i in torch.unique(ids).tolist() if 0 <= i < V]
    want = table[uniq].double().pow(2).sum().item()
    assert (rp[n:] == 0).all() and int((rp[:n] > 0).sum()) == len(uniq)
    assert abs(rp.double().sum().item() - want) <= 1e-5 * want


def test_step_norm_from_producers_matches_buffer_norm(setup, monkeypatch):
    """r04: on one rank with one sample per step the gradient norm is assembled from the weight-gradient GEMM epilogues, the touched embedding rows and
    a chunk pass over the small tensors instead of re-reading the gradient buffer.  Same step with VLASER_SFT_NO_FUSED_NORM=1 (the buffer pass): the
    norms agree to fp32 summation order -- checked at every step against the buffer norm of the SAME gradients, and across the two runs on the first step --, the
    updated parameters to what the clip factor's last bit can move; the producer path is deterministic."""
    from vlaser_amd.sft import SFTModel
    cfg, sd, _, pv, ids, labels, _ = setup

    def run(fused):
        monkeypatch.setenv('VLASER_SFT_NO_FUSED_NORM', '0' if fused else '1')
        m = SFTModel(cfg, max_seq_len=ids.shape[1], lr=1e-3, weight_decay=0.05, max_grad_norm=1.0)
        m.load_state_dict(sd)
        outs = []
        for _ in range(3):
            outs.append(m.step(pv, ids, labels))
            # the same gradients, summed from the buffer: the producers' slots must account for every element (any trajectory, every step)
            buf = torch.linalg.vector_norm(m.fp.g.float()).item()
            assert abs(outs[-1].grad_norm.item() - buf) <= 2e-5 * buf, (fused, outs[-1].grad_norm.item(), buf)
        m.wait_optimizer()
        assert (getattr(m, 'norm_parts', None) is not None) == fused
        return [o.grad_norm.item() for o in outs], [o.loss.item() for o in outs], {k: v.clone() for k, v in m.state_dict().items()}

    na, la, pa = run(True)
    nb, lb, pb = run(False)
    nc, lc, pc = run(True)
    assert na == nc and la == lc and all(torch.equal(pa[k], pc[k]) for k in pa), 'the producer-side norm is not deterministic'
    assert la[0] == lb[0]
    assert abs(na[0] - nb[0]) <= 2e-5 * nb[0], (na, nb)
    # later steps: the two runs' clip factors may differ in their last bit, which the lr = 1e-3 steps of this test amplify (a few hundred bf16 parameters round the
    # other way; measured 3e-4 on the second step's norm) -- the per-step buffer check above is the exact statement, this one only bounds the drift
    for a, b in zip(na[1:], nb[1:]):
        assert abs(a - b) <= 5e-3 * b, (na, nb)
    for k in pa:
        d = (pa[k].float() - pb[k].float()).abs().max().item()
        # steps 2 and 3 of the two trajectories may move an element with a near-zero gradient in opposite directions: <= 2 lr per step
        assert d <= 2 * 2 * 1e-3 + 2e-2 * pb[k].float().abs().max().item(), (k, d)


@pytest.mark.parametrize('pipe,plain', [(1110, 1100), (1210, 1200)])
def test_gemm_tn_lds_pipelined_reads_bit_identical(pipe, plain):
    """r06 lab: TN weight-gradient tiles with their fragment reads pipelined across the K-step's barrier (csrc/gemm.hip PIPE, codes 1110 / 1210; not the default): same accumulation order,
    bit-identical outputs -- contraction of 1 .. 9 tiles, ragged M / N."""
    from vlaser_amd import ops
    g = torch.Generator().manual_seed(5)
    for (K, M, N) in [(64, 136, 264), (128, 1000, 520), (576, 2048 + 24, 1536 + 8), (320, 17920 // 8, 1536)]:
        At = torch.randn(K, M, generator=g).to(BF).cuda(); Wt = torch.randn(K, N, generator=g).to(BF).cuda()
        o1, o2 = torch.zeros(M, N, dtype=BF, device='cuda'), torch.zeros(M, N, dtype=BF, device='cuda')
        ops.gemm_tn_lds(At, Wt, o1, K, force_cfg=pipe)
        ops.gemm_tn_lds(At, Wt, o2, K, force_cfg=plain)
        assert torch.equal(o1, o2), (K, M, N)


def test_gemm_tn_lds_ragged_output_rows():
    """The lm_head weight gradient's shape class: M (= vocabulary rows of dW) not a multiple of 8, the operand a column-view of a wider buffer whose
    rows cover M rounded up to 8 (16-byte pieces are read whole, rows >= M of the result are dropped); through ops.gemm_tn's routing."""
    from vlaser_amd import ops
    g = torch.Generator().manual_seed(5)
    K, M, Mp, N = 128, 1002, 1024, 520
    big = torch.randn(K, Mp, generator=g).to(BF).cuda()
    At, Wt = big[:, :M], torch.randn(K, N, generator=g).to(BF).cuda()
    out = torch.full((M + 3, N), 7.0, dtype=BF, device='cuda')
    ops.gemm_tn(At, Wt, out[:M])
    ref = At.float().t() @ Wt.float()
    rel, cos = _rel(out[:M], ref)
    assert rel < 5e-3 and cos > 0.9999
    assert (out[M:] == 7.0).all()                          # nothing written past row M


def test_attention_backward_fused_pds_and_grouped_tn():
    """vlaser_attn_bwd_pds_masked (softmax + dS in one pass, causal) and the grouped TN GEMM (dK / dV summed over the q heads of a kv group)
    against the closed forms, on a ragged S (not a multiple of 64)."""
    from vlaser_amd import ops
    g = torch.Generator().manual_seed(5)
    H, nkv, S, hd = 4, 2, 83, 128
    G, Sp = H // nkv, 128
    scale = hd ** -0.5
    sc = torch.zeros(H, S, Sp, device='cuda'); dP = torch.zeros(H, S, Sp, device='cuda')
    sc[:, :, :S] = (torch.randn(H, S, S, generator=g) * 4).cuda(); dP[:, :, :S] = torch.randn(H, S, S, generator=g).cuda()
    dO = torch.randn(S, H * hd, generator=g).to(BF).cuda(); O = torch.randn(S, H * hd, generator=g).to(BF).cuda()
    P = torch.full((H, S, Sp), 7, dtype=BF, device='cuda'); dS = torch.full((H, S, Sp), 7, dtype=BF, device='cuda')
    ops.attn_bwd_pds_masked(sc, dP, dO, O, P, dS, H, S, Sp, hd, scale, True, Sp, 0)
    mask = torch.tril(torch.ones(S, S, dtype=torch.bool, device='cuda'))
    Pref = (sc[:, :, :S] * scale).masked_fill(~mask, float('-inf')).softmax(-1)
    D = (dO.float() * O.float()).view(S, H, hd).sum(-1).t()                      # [H, S]
    dSref = Pref.to(BF).float() * (dP[:, :, :S] - D[:, :, None]) * scale
    assert (P[:, :, :S].float() - Pref).abs().max().item() < 4e-3 and (P[:, :, S:] == 0).all()
    assert (dS[:, :, :S].float() - dSref).abs().max().item() < 2e-2 * dSref.abs().max().item() and (dS[:, :, S:] == 0).all()
    assert (dS[:, :, :S].float().masked_select(~mask) == 0).all()
    # dK[kvh] = sum_g dS[kvh*G+g]^T Q_g
    q = torch.randn(S, H * hd, generator=g).to(BF).cuda()
    dk = torch.full((S, nkv * hd), 7, dtype=BF, device='cuda')
    ops.gemm_tn_grouped(dS, q, dk, S, hd, S, Sp, H * hd, nkv * hd, G, S * Sp, hd, nkv, G * S * Sp, G * hd, hd)
    ref = torch.einsum('hqk,qhd->khd', dS[:, :, :S].float(), q.float().view(S, H, hd)).view(S, nkv, G, hd).sum(2).reshape(S, nkv * hd)
    rel, cos = _rel(dk, ref)
    assert rel < 6e-3 and cos > 0.9999, (rel, cos)


def _four_samples(cfg, seed=11):
    """Four ragged samples as the collator batches them: ids right-padded with 0, labels with -100, tiles concatenated."""
    g = torch.Generator().manual_seed(seed)
    rows, labs, pvs = [], [], []
    for n_text, n_lab in ((20, 6), (33, 12), (9, 4), (27, 9)):
        ids = torch.cat([torch.randint(1, 151643, (12,), generator=g), torch.full((256,), cfg.img_context_token_id),
                         torch.randint(1, 151643, (n_text,), generator=g)])
        lab = torch.full_like(ids, -100); lab[-n_lab:] = ids[-n_lab:]
        rows.append(ids); labs.append(lab); pvs.append(torch.randn(1, 3, 448, 448, generator=g))
    S = max(len(r) for r in rows)
    ids = torch.zeros(4, S, dtype=torch.long); lab = torch.full((4, S), -100)
    for b, (r, l) in enumerate(zip(rows, labs)):
        ids[b, :len(r)] = r; lab[b, :len(r)] = l
    return torch.cat(pvs), ids, lab, rows, labs, pvs


def test_batch4_equals_accumulation_of_4_and_oracle_batch_gradient(golden_model):
    """per-device batch > 1 and gradient accumulation (…2nd_finetune_full.sh:5-6,49-50).  (a) one per-device batch
    of 4 padded samples: loss and gradients vs torch autograd of the fp32 oracle's batch loss (mean over ALL supervised tokens);
    (b) the same 4 samples as 4 accumulation micro-batches of 1 with HF's loss / GA scaling: every sample weighs 1/4 instead of
    R_b / R -- checked against the oracle with those weights; both bit-reproducible."""
    from oracle import vlm as ovlm
    from vlaser_amd.sft import SFTModel
    cfg, _, sd = golden_model
    pv, ids, lab, rows, labs, pvs = _four_samples(cfg)
    keys = ['language_model.model.layers.1.mlp.down_proj.weight', 'language_model.model.layers.0.self_attn.q_proj.weight', 'mlp1.3.weight',
            'language_model.model.norm.weight', 'language_model.lm_head.weight']

    def oracle(weights):
        torch.set_grad_enabled(True)
        try:
            sdg = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
            tot = 0
            for b in range(4):
                lg = ovlm.forward_logits(sdg, cfg, pvs[b], rows[b][None])
                tot = tot + weights[b] * ovlm.sft_loss(lg, labs[b][None])
            tot.backward()
            return tot.item(), {k: sdg[k].grad for k in keys}
        finally:
            torch.set_grad_enabled(False)

    R = torch.tensor([float((l[1:] != -100).sum()) for l in labs])
    for mode, weights in (('batch', (R / R.sum()).tolist()), ('accum', [0.25] * 4)):
        m = SFTModel(cfg, max_seq_len=ids.shape[1], max_grad_norm=0.0, lr=0.0, weight_decay=0.0)      # lr 0: the step leaves the weights alone
        m.load_state_dict(sd)
        if mode == 'batch':
            out = m.train_step([(pv, ids, lab, torch.ones(4, 1, dtype=torch.long))])
        else:
            out = m.train_step([(pvs[b], rows[b][None], labs[b][None], None) for b in range(4)])
        ref_loss, ref_g = oracle(weights)
        assert abs(out.loss.item() - ref_loss) < 5e-3, (mode, out.loss.item(), ref_loss)
        grads = m.named_grads()
        for k in keys:
            rel, cos = _rel(grads[k], ref_g[k])
            assert rel < 4e-2 and cos > 0.999, (mode, k, rel, cos)
        m2 = SFTModel(cfg, max_seq_len=ids.shape[1], max_grad_norm=0.0, lr=0.0, weight_decay=0.0); m2.load_state_dict(sd)
        (m2.train_step([(pv, ids, lab, torch.ones(4, 1, dtype=torch.long))]) if mode == 'batch' else
         m2.train_step([(pvs[b], rows[b][None], labs[b][None], None) for b in range(4)]))
        assert torch.equal(m2.fp.g, m.fp.g)                                    # deterministic accumulation order


def test_embedding_gradient_sums_duplicates_in_fp32(ops):
    """a token that occurs hundreds of times must not lose its later contributions to bf16 re-rounding; ids out of range
    are skipped, <IMG_CONTEXT> positions (rank >= 0) do not touch the table."""
    n, H, V = 3000, 256, 500
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (n,), generator=g)
    ids[::3] = 7                                                               # one id on a third of the positions
    rank = torch.full((n,), -1, dtype=torch.int32); rank[100:140] = torch.arange(40, dtype=torch.int32)
    dh = (torch.randn(n, H, generator=g) * 0.01 + 0.01).to(BF)
    base = (torch.randn(V, H, generator=g) * 0.01).to(BF)
    demb = base.clone().cuda()
    ops.embed_scatter_add(ids.cuda(), rank.cuda(), dh.cuda(), demb, n, H)
    ref = base.float().clone()
    keep = rank < 0
    ref.index_add_(0, ids[keep], dh[keep].float())
    ref = ref.to(BF).float()
    got = demb.float().cpu()
    assert (got - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()      # one bf16 rounding of the fp32 sum
    assert (got[7] - base[7].float()).abs().mean() > 5.0                                # ~1000 contributions of ~0.01 really arrived


def test_checkpoint_resume_is_bit_identical(golden_model, tmp_path):
    """Resumable training state (HF Trainer checkpoints / VLA step{N}.pt): 3 uninterrupted steps == 2 steps, save_checkpoint, a fresh
    model's load_checkpoint, 1 more step -- weights, fp32 masters and AdamW moments bit for bit."""
    from vlaser_amd.sft import SFTModel
    cfg, _, sd = golden_model
    pv, ids, lab, rows, labs, pvs = _four_samples(cfg, seed=21)
    mk = lambda: SFTModel(cfg, max_seq_len=ids.shape[1], lr=1e-3, weight_decay=0.05)
    a = mk(); a.load_state_dict(sd)
    for s in range(3):
        a.step(pvs[s], rows[s][None], labs[s][None], total_steps=10)
    b = mk(); b.load_state_dict(sd)
    for s in range(2):
        b.step(pvs[s], rows[s][None], labs[s][None], total_steps=10)
    b.save_checkpoint(str(tmp_path / 'ckpt'))
    c = mk(); c.load_checkpoint(str(tmp_path / 'ckpt'))
    assert c.step_count == 2
    c.step(pvs[2], rows[2][None], labs[2][None], total_steps=10)
    a.wait_optimizer(); c.wait_optimizer()          # the last AdamW is still in flight on the optimizer stream
    assert torch.equal(a.fp.p, c.fp.p) and torch.equal(a.master, c.master) and torch.equal(a.m, c.m) and torch.equal(a.v, c.v)


@pytest.mark.parametrize('S,C,mode', [(256, 4096, 3), (37, 1024, 3), (50, 100, 3), (20, 8192, 3), (256, 4096, 2), (33, 100, 2)])
def test_colsum_mul_norm_weight_grad(ops, S, C, mode):
    """vlaser_colsum_mul modes 2 / 3 (RMSNorm / LayerNorm weight gradient: sum_s dy * normalised x, row statistics by `rowstat_kernel`) against fp32 autograd -- the
    projector's LayerNorm(4096) of modeling_internvl_chat.py:72-77 and widths that take the kernel's register-resident (C <= 4096, C % 8 == 0) and fallback paths"""
    g = torch.Generator().manual_seed(S * C + mode)
    x = (torch.randn(S, C, generator=g) * 1.5 + 0.3).to(BF).cuda(); dy = torch.randn(S, C, generator=g).to(BF).cuda()
    col = torch.zeros(C, device='cuda'); ws = torch.zeros(2 * S + 16 * C, device='cuda')
    ops.colsum_mul(dy, x, col, S, C, mode, 1e-6, ws)
    xf = x.float()
    if mode == 3:
        xn = (xf - xf.mean(-1, keepdim=True)) * torch.rsqrt(xf.var(-1, unbiased=False, keepdim=True) + 1e-6)
    else:
        xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
    ref = (dy.float() * xn).sum(0)
    assert float((col - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-4


@pytest.mark.parametrize('V,ld,off', [(151674, 151674, 0), (1001, 1001, 0), (1000, 1003, 1), (8195, 8200, 0), (7, 7, 0)])
def test_ce_rows_matches_logsumexp(ops, V, ld, off):
    """vlaser_ce_rows (CrossEntropyLoss rows of modeling_internvl_chat.py:231-243) against torch.logsumexp in fp64: full vocabulary width, odd widths, rows that are
    only 4-byte aligned (the 8-byte loads of the kernel must fall back), ignore_index rows"""
    R = 9
    g = torch.Generator(device='cuda').manual_seed(V + off)
    buf = torch.randn(R * ld + off + 8, device='cuda', generator=g) * 6.0
    logits = buf[off:off + R * ld].view(R, ld)[:, :V]
    labels = torch.randint(0, V, (R,), device='cuda', generator=g)
    labels[2] = -100
    loss = torch.full((R,), 7.0, device='cuda'); lse = torch.empty(R, device='cuda')
    ops.ce_rows(logits, labels, loss, lse)
    ref_lse = torch.logsumexp(logits.double(), dim=1)
    ref = ref_lse - logits.double().gather(1, labels.clamp(min=0)[:, None])[:, 0]
    ref[2] = 0.0
    assert torch.allclose(lse.double(), ref_lse, rtol=0, atol=2e-5 * max(1.0, float(ref_lse.abs().max())))
    assert torch.allclose(loss.double(), ref, rtol=0, atol=4e-5 * max(1.0, float(ref_lse.abs().max())))


def test_device_side_clip_matches_host_formula(ops):
    n = 50_000
    g = torch.Generator().manual_seed(8)
    gr = (torch.randn(n, generator=g) * 0.05).to(BF).cuda()
    p0 = (torch.randn(n, generator=g) * 0.05)
    gn2 = gr.float().pow(2).sum().reshape(1)
    for max_norm in (0.0, 1.0, 100.0):
        a = [p0.to(BF).cuda(), p0.cuda().clone(), torch.zeros(n).cuda(), torch.zeros(n).cuda()]
        b = [p0.to(BF).cuda(), p0.cuda().clone(), torch.zeros(n).cuda(), torch.zeros(n).cuda()]
        gnorm = gn2.sqrt().item()
        scale = max_norm / (gnorm + 1e-6) if (max_norm and gnorm > max_norm) else 1.0
        ops.adamw(*a, gr, 1e-3, 0.9, 0.999, 1e-8, 0.05, scale, 1)
        ops.adamw_clipped(*b, gr, 1e-3, 0.9, 0.999, 1e-8, 0.05, 1.0, gn2, max_norm, 1)
        torch.testing.assert_close(a[1], b[1], rtol=1e-6, atol=1e-8)


def _packed_inputs(golden_dir):
    d = np.load(os.path.join(golden_dir, 'g11_packed.npz'))
    g = torch.Generator().manual_seed(int(d['seed']))
    for n_img, n_text, n_lab in ((256, 22, 7), (0, 31, 9), (256, 15, 5)):
        torch.randint(1, 151643, (9,), generator=g); torch.randint(1, 151643, (n_text,), generator=g)
    pv = torch.randn(3, 3, 448, 448, generator=g)
    return d, pv, torch.from_numpy(d['input_ids']), torch.from_numpy(d['labels']), torch.from_numpy(d['loss_weight']), torch.from_numpy(d['cu_seqlens'])[None], \
        torch.from_numpy(d['image_flags'])


def test_packed_sequence_forward_and_sft_step(golden_model, golden_dir):
    """SURVEY 8f-4 / VERDICT r01 missing #5: packed-sequence SFT.  (a) InternVLChatModel.forward with loss_weight + cu_seqlens vs the
    reference's own packed loss (golden G11); (b) SFTModel.train_step_packed: loss vs G11 and every gradient norm vs the reference
    model's autograd on the packed row."""
    from vlaser_amd.internvl_chat import InternVLChatModel
    from vlaser_amd.sft import SFTModel
    cfg, _, sd = golden_model
    d, pv, ids, labels, w, cu, flags = _packed_inputs(golden_dir)
    vsd = {k: v for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))}
    chat = InternVLChatModel(cfg, max_seq_len=640, max_tiles=3); chat.load_state_dict(vsd)
    chat.img_context_token_id = cfg.img_context_token_id
    out = chat.forward(pv, ids, attention_mask=cu, image_flags=flags, labels=labels, loss_weight=w.tolist())
    assert abs(out.loss.item() - float(d['loss'])) < 1e-2, (out.loss.item(), float(d['loss']))
    assert (out.logits[0, -1].topk(8).values.cpu() - torch.from_numpy(d['last_logits'])).abs().max() < 3e-2 * abs(d['last_logits']).max()
    del chat
    m = SFTModel(cfg, max_seq_len=320, max_tiles=1, lr=0.0, weight_decay=0.0, max_grad_norm=0.0); m.load_state_dict(sd)
    res = m.train_step_packed(pv, ids, labels, w, cu, image_flags=flags)
    assert abs(res.loss.item() - float(d['loss'])) < 1e-2
    grads = m.named_grads()
    names = [str(n) for n in d['names']]
    assert set(names) == set(grads)
    for n in names:
        ref = float(d[f'norm::{n}'])
        assert abs(grads[n].double().norm().item() - ref) < 4e-2 * ref + 1e-9, (n, grads[n].double().norm().item(), ref)
