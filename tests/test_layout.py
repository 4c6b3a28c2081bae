"""Host-side contract checks that run without a GPU: the C-ABI library loads and exports every symbol declared in
include/vlaser_hip.h, ctypes structs mirror the header, the product never imports the oracle, and the product fails
loudly (no CPU fallback) when there is no GPU."""
import ast
import ctypes
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header(name='vlaser_hip.h'):
    return open(os.path.join(ROOT, 'include', name)).read()


_PROTO = r'\b(?:int|const char\*)\s+(vlaser_\w+)\s*\('


def test_library_exports_every_declared_symbol():
    from vlaser_amd import _lib
    lib = _lib.lib()
    names = set(re.findall(_PROTO, _header()))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), f'{n} declared in include/vlaser_hip.h but not exported by libvlaser_hip.so'
    assert lib.vlaser_abi_version() == 8
    # __graft_entry__.build() asserts the same number (the driver's build check): the two must move together
    assert f'vlaser_abi_version() == {lib.vlaser_abi_version()}' in open(os.path.join(ROOT, '__graft_entry__.py')).read()
    # ONE public header (the experimental one and its two off-by-default kernels left in r05); every bound signature refers to a declared symbol, and the
    # library exports nothing that no header declares
    assert sorted(os.listdir(os.path.join(ROOT, 'include'))) == ['vlaser_hip.h']
    assert set(_lib._SIGS) <= names
    import subprocess
    out = subprocess.run(['/usr/bin/nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r'\bT (vlaser_\w+)', out))
    assert exported and exported - {'vlaser_set_error'} <= names, sorted(exported - names)


def _struct_fields(name):
    h = _header()
    end = h.index('} ' + name + ';')
    body = h[h.rindex('typedef struct {', 0, end) + len('typedef struct {'):end]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for stmt in body.split(';'):
        stmt = stmt.strip()
        if not stmt:
            continue
        for part in stmt.split(','):
            part = re.sub(r'\[\d+\]', '', part)        # float mean[3]
            ident = re.findall(r'(\w+)\s*$', part.strip().replace('*', ' '))
            if ident:
                fields.append(ident[0])
    return fields


@pytest.mark.parametrize('cname,pyname', [('VlaserGemmArgs', 'GemmArgs'), ('VlaserAttnArgs', 'AttnArgs'), ('VlaserSkinnyArgs', 'SkinnyArgs'),
                                          ('VlaserVlaStageArgs', 'VlaStageArgs')])
def test_ctypes_structs_mirror_header(cname, pyname):
    from vlaser_amd import _lib
    assert [f[0] for f in getattr(_lib, pyname)._fields_] == _struct_fields(cname)


def test_no_kernel_uses_scratch(tmp_path):
    """No kernel of the shipped library may spill: every reload of a spilled register is a `s_waitcnt vmcnt(0)` in the middle of a load burst (lesson from
    csrc/attn_o.hip; r04 shipped three such variants, none of them on a hot path).  Read from the code objects' metadata notes -- what the loader sees."""
    import shutil
    import subprocess
    from vlaser_amd import _lib
    llvm = '/opt/rocm/lib/llvm/bin'
    so = shutil.copy(_lib.LIB_PATH, tmp_path / 'lib.so')
    subprocess.run([f'{llvm}/llvm-objdump', '--offloading', str(so)], check=True, capture_output=True, cwd=tmp_path)
    cos = sorted(p for p in tmp_path.iterdir() if 'gfx950' in p.name)
    assert cos, 'no gfx950 code object found in the library'
    n_kernels, spills = 0, []
    for co in cos:
        notes = subprocess.run([f'{llvm}/llvm-readelf', '--notes', str(co)], check=True, capture_output=True, text=True).stdout
        for name, size in re.findall(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)', notes):
            n_kernels += 1
            if int(size):
                spills.append((name, int(size)))
    assert n_kernels > 100, n_kernels
    assert not spills, f'kernels with scratch (private segment) bytes: {spills}'


def test_enum_values_mirror_header():
    from vlaser_amd import _lib
    h = _header()
    for name, val in re.findall(r'\b(VL_\w+)\s*=\s*(\d+)', h):
        py = name[3:]
        if hasattr(_lib, py):
            assert getattr(_lib, py) == int(val), name


def test_product_never_imports_oracle_or_reference():
    pkg = os.path.join(ROOT, 'vlaser_amd')
    for fn in os.listdir(pkg):
        if not fn.endswith('.py'):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or '']
            for m in mods:
                assert not m.startswith('oracle') and 'ref_import' not in m, f'{fn} imports {m}'
        assert '/root/reference' not in open(os.path.join(pkg, fn)).read()
    for fn in ('bench.py', '__graft_entry__.py'):
        assert '/root/reference' not in open(os.path.join(ROOT, fn)).read()


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_no_cpu_fallback():
    from vlaser_amd import _lib, config as C
    from vlaser_amd.internvl_chat import InternVLChatModel
    from vlaser_amd.pizero import PiZeroInference
    with pytest.raises(_lib.VlaserHipError):
        InternVLChatModel(C.vlaser_2b())
    with pytest.raises(_lib.VlaserHipError):
        PiZeroInference(C.VLAConfig())


def test_bad_arguments_return_errors_not_crashes():
    """Argument validation happens on the host before any launch, so it can be exercised without a GPU."""
    from vlaser_amd import _lib
    lib = _lib.lib()
    a = _lib.GemmArgs()
    assert lib.vlaser_gemm(0, ctypes.byref(a), None) != 0
    assert b'null' in lib.vlaser_last_error()
    a.A, a.W, a.out = 16, 16, 16
    a.M, a.N, a.K, a.lda, a.ldw, a.ldo = 4, 4, 100, 104, 104, 4
    assert lib.vlaser_gemm(0, ctypes.byref(a), None) != 0
    assert b'multiple of 64' in lib.vlaser_last_error()
    s = _lib.SkinnyArgs()
    s.x, s.W, s.M, s.N, s.K, s.k_splits = 16, 16, 17, 32, 256, 1
    assert lib.vlaser_skinny(0, 0, ctypes.byref(s), None) != 0
    assert b'1..16' in lib.vlaser_last_error()
    # round-3 entry points: contraction not padded / rows too short for 16-byte pieces
    P = 4096
    assert lib.vlaser_gemm_tn_lds(P, P, P, 1536, 2048, 100, 1536, 2048, 2048, 0, None, 0, None) != 0
    assert b'multiple of 64' in lib.vlaser_last_error()
    assert lib.vlaser_gemm_tn_lds(P, P, P, 1002, 2048, 128, 1002, 2048, 2048, 0, None, 0, None) != 0
    assert b'rounded up to 8' in lib.vlaser_last_error()
    # fused attention backward: cache row length not a multiple of 64 / more valid keys than cache rows
    assert lib.vlaser_attn_bwd(P, P, P, P, P, P, P, P, P, P, 70, 12, 2, 100, 0.1, 1, 70, 128, None) != 0
    assert b'bad geometry' in lib.vlaser_last_error()
    assert lib.vlaser_attn_bwd(P, P, P, P, P, P, P, P, P, P, 64, 12, 2, 128, 0.1, 1, 64, 64, None) != 0           # head_dim other than 128 is refused
    assert b'head_dim 64' in lib.vlaser_last_error()
    # one launch between two passes through the expert: width not a multiple of 256; finishing in place
    assert lib.vlaser_vla_step(None, None, 0, 4, 0, None, 1e-6, None, None, P, P, None, 0.1, 0, P, P, P, P, P, 4, 700, 7, 0, None) != 0
    assert b'multiple of 256' in lib.vlaser_last_error()
    assert lib.vlaser_vla_step(P, None, 0, 4, 0, P, 1e-6, P, P, P, P, None, 0.1, 1, P, P, P, P, P, 4, 768, 7, 0, None) != 0
    assert b'distinct action buffers' in lib.vlaser_last_error()
    assert lib.vlaser_vla_step(None, None, 0, 4, 0, None, 1e-6, None, None, P, P, None, 0.1, 0, P, P, P, P, P, 4, 768, 7, 3, None) != 0      # (ABI 7) integration method 0 | 1 | 2
    assert b'method' in lib.vlaser_last_error()


def test_chain_gu_predicate_mirrors_the_dispatch_table():
    """`vlaser_chain_gu_supported` answered 1 for K = 768 / 2 producer slabs / 2 units per workgroup, which `vlaser_chain_gu` has no instantiation for.
    Both are generated from one variant list now: the predicate (host-only, no GPU) must say yes exactly on the list's rows."""
    from vlaser_amd import _lib
    lib = _lib.lib()
    built = {(3, 3, 3, 2), (3, 3, 2, 2), (6, 3, 2, 2), (3, 2, 3, 2), (6, 2, 2, 2), (3, 5, 3, 1), (6, 5, 2, 1)}     # {K / 256, units per workgroup, slabs, tiles per unit}
    for ns in (2, 3, 4, 6):
        for longest in (1, 2, 3, 4, 5, 6):
            for sp in (1, 2, 3, 4):
                for tpu in (1, 2):
                    N = 16 * tpu * (256 * (longest - 1) + 100)
                    got = lib.vlaser_chain_gu_supported(4, N, ns * 256, sp, tpu)
                    assert got == int((ns, longest, sp, tpu) in built), (ns, longest, sp, tpu, got)
    assert lib.vlaser_chain_gu_supported(17, 17920, 768, 3, 2) == 0 and lib.vlaser_chain_gu_supported(16, 17920, 1536, 2, 2) == 0     # rows / chunks per thread


def test_weight_packing_roundtrip():
    """pack_qkv / pack_gate_up / pack_skinny are pure permutations (CPU check of the index maps)."""
    from vlaser_amd import ops
    perm = ops.head_perm(128)
    assert sorted(perm.tolist()) == list(range(128))
    # RoPE pair (d, d+64) lands 16 packed rows apart inside the same 32-row group
    inv = torch.empty(128, dtype=torch.long); inv[perm] = torch.arange(128)
    for d in range(64):
        assert inv[d + 64] - inv[d] == 16 and inv[d] // 32 == inv[d + 64] // 32
    g = torch.arange(64 * 8, dtype=torch.float32).view(64, 8); u = -g
    gu = ops.pack_gate_up(g, u)
    assert torch.equal(gu[0:16], g[0:16]) and torch.equal(gu[16:32], u[0:16]) and torch.equal(gu[32:48], g[16:32])
    W = torch.arange(70 * 512, dtype=torch.float32).view(70, 512).to(torch.bfloat16)
    for ks in (1, 2):
        pw = ops.pack_skinny(W, ks)
        assert pw.N == 96 and pw.n_valid == 70 and pw.t.numel() == 96 * 512
        ns = 512 // (ks * 8 * 32)
        v = pw.t.view(ks, 3, 8, ns, 2, 4, 16, 8)        # [ks, u, w, s, t, g, r, e]
        k0 = 1 * (512 // ks) * 0 + 3 * ns * 32 + 0 * 32 + 2 * 8      # ks=0, wave 3, step 0, g=2
        assert torch.equal(v[0, 1, 3, 0, 1, 2, 5], W[32 + 16 + 5, k0:k0 + 8])


def test_vla_checkpoint_keys_of_the_reference_canonicalize():
    """Golden G9 = the key names / shapes of the reference's own PiZero module tree (built on the meta device by
    tools/gen_golden_vla_keys.py).  Every key must canonicalise onto a tensor the loader consumes (or onto the short list of
    heads the inference path never reads), aliases of one module must agree in shape, and nothing the loader needs may be
    missing -- the reference loads `strict=False` but asserts no missing keys (eval.py:196-212)."""
    import json
    import re
    import torch
    from vlaser_amd import config as C, synth
    from vlaser_amd.pizero import canonicalize_vla_state_dict
    keys = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g9_vla_state_keys.json')))['keys']
    assert len(keys) == 1713
    # a checkpoint saved from a torch.compile'd module carries the `_orig_mod.` prefix on every key (eval.py:200-210)
    sd = {('_orig_mod.' + k): torch.empty(shape, device='meta') for k, shape in keys.items()}
    canon = canonicalize_vla_state_dict(sd)
    # what the loader consumes: the synthetic checkpoint's key set, generalised from a 1+1-layer model to the full depth
    full = C.VLAConfig(base=C.vlaser_2b())
    small = synth.vla_state_dict(C.VLAConfig(base=C.truncated(C.vlaser_2b(), 1, 1)), with_head=True)
    need = {}
    for k, v in small.items():
        if '.layers.0.' in k:
            n = full.base.vision.num_hidden_layers if k.startswith('vision_model.') else full.base.llm.num_hidden_layers
            for i in range(n):
                need[k.replace('.layers.0.', f'.layers.{i}.')] = tuple(v.shape)
        else:
            need[k] = tuple(v.shape)
    vocab_rows = {'language_model.model.embed_tokens.weight', 'language_model.lm_head.weight'}    # + 256 '<a i>' tokens (pizero_internvl.py:45-48,85)
    for k, shape in need.items():
        assert k in canon, f'the reference checkpoint has no tensor for {k}'
        got = tuple(canon[k].shape)
        if k in vocab_rows:
            assert got[1:] == shape[1:] and got[0] == shape[0] + 256, (k, got, shape)
        else:
            assert got == shape, (k, got, shape)
    extra = set(canon) - set(need)
    assert extra == {'internvl_model.action_expert.lm_head.weight'}, sorted(extra)[:10]           # the expert's unused vocabulary head
    # the proprio mixture IS the action mixture after tie_action_proprio_weights (pizero_internvl.py:508-510)
    assert all(keys[k] == keys[k.replace('.proprio.', '.action.')] for k in keys if '.mixtures.proprio.' in k)
    assert not any(re.search(r'mixtures\.(vlm|action|proprio)\.layers\.\d+\.(self_attn\.o_proj\.bias|mlp\..*bias)', k) for k in keys)


def test_vla_checkpoint_writer_emits_every_reference_key():
    """`reference_vla_state_dict` (the VLA `.pt` writer) must produce exactly the key set of the reference's PiZero.state_dict()
    (golden G9, aliases included) so that the reference's loader finds no missing key; round trip through the canonicaliser is the
    identity on the canonical tensors."""
    import json
    import torch
    from vlaser_amd import config as C, synth
    from vlaser_amd.pizero import canonicalize_vla_state_dict, reference_vla_state_dict
    keys = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g9_vla_state_keys.json')))['keys']
    small = synth.vla_state_dict(C.VLAConfig(base=C.truncated(C.vlaser_2b(), 1, 1)), with_head=True)
    ref = reference_vla_state_dict(small)
    want = {k for k in keys if '.layers.' not in k or '.layers.0.' in k}          # the 1 + 1-layer slice of the full key list
    assert set(ref) == want, (sorted(set(ref) - want)[:5], sorted(want - set(ref))[:5])
    back = canonicalize_vla_state_dict(ref)
    for k, v in small.items():
        assert torch.equal(back[k], v), k


def test_internvla_processor_matches_reference_contract():
    """InternVLAProcessor mirror: prompt ids equal golden G1's VLA prompt (the reference's processor output), pixel normalisation is
    the ImageNet affine map on uint8 input, padding to max_seq_len on the right."""
    import json
    import torch
    from vlaser_amd import prep

    class Tok:                                     # records the query string; the real Qwen2 tokenizer is not on the GPU box
        model_max_length = 0

        def convert_tokens_to_ids(self, t):
            return 151665

        def __call__(self, query, **kw):
            self.query, self.kw = query, kw
            return {'input_ids': torch.zeros(len(query), kw['max_length'], dtype=torch.long), 'attention_mask': torch.ones(len(query), kw['max_length'], dtype=torch.long)}
    tok = Tok()
    proc = prep.InternVLAProcessor(tok, num_image_tokens=256, max_seq_len=384, tokenizer_padding='max_length')
    img = torch.randint(0, 256, (2, 1, 3, 448, 448), dtype=torch.uint8)          # [B, n_images, 3, H, W] as simpler.py:82-92 builds it
    out = proc(['put the spoon on the towel', 'pick up the carrot'], img)
    assert tok.kw == {'return_tensors': 'pt', 'max_length': 384, 'padding': 'max_length', 'truncation': True} and tok.model_max_length == 384
    assert tok.query[0] == prep.build_vla_query('put the spoon on the towel', 256) and tok.query[0].count('<IMG_CONTEXT>') == 256
    assert tok.query[0].startswith('<|im_start|>system\nNone<|im_end|>\n<|im_start|>user\n<img>')
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    ref = (img[:, 0].float() / 255.0 - mean[None, :, None, None]) / std[None, :, None, None]
    assert torch.allclose(out['pixel_values'].float(), ref, atol=1e-6) and out['input_ids'].shape == (2, 384)
    with pytest.raises(AssertionError):
        proc(['x'], img.float())


def test_split_k_choice_respects_kernel_constraints():
    """ops.gemm_splits: K/splits must stay a multiple of 64 and >= 256 (csrc/gemm.hip checks both), splits never exceed what the consumers'
    slab buffers hold, and an output that already fills the chip is not split."""
    from vlaser_amd import ops
    shapes = [(1025, 1024, 1024), (1025, 1024, 4096), (272, 1536, 1536), (272, 1536, 8960), (560, 1536, 8960), (560, 1536, 17920),
              (3400, 3584, 18944), (5, 1536, 1536), (5, 1536, 8960), (64, 128, 256), (33, 4096, 320)]
    for M, N, K in shapes:
        s = ops.gemm_splits(M, N, K)
        assert 1 <= s <= 8 and K % (s * 64) == 0 and (K // s >= 256 or s == 1), (M, N, K, s)
        budget = ops.split_slab_elems(M, N)          # a workspace sized for M rows always holds the factor chosen for <= M rows
        for m in (M, max(1, M // 2), max(1, M // 7)):
            assert ops.gemm_splits(m, N, K, budget) * m * N <= budget
    assert ops.gemm_splits(4096, 1536, 8960, ops.split_slab_elems(4096, 1536)) <= 2
    assert ops.gemm_splits(4096, 4096, 4096) == 1
    assert ops.gemm_tile_config(1025, 4096)[0] == 1440 and ops.gemm_tile_config(16, 4096)[0] == 32


def test_golden_manifest_matches_directory():
    """`python tools/gen_golden.py` (no flags) rebuilds every fixture listed in tools/golden_manifest.py and asserts the list equals
    tests/golden/ (the documented command used to skip G10b); this is the same assertion without the reference."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        from golden_manifest import FIXTURES
    finally:
        sys.path.pop(0)
    assert sorted(os.listdir(os.path.join(ROOT, 'tests', 'golden'))) == sorted(FIXTURES)
    for f, script in FIXTURES.items():
        assert os.path.exists(os.path.join(ROOT, script)), (f, script)
    src = open(os.path.join(ROOT, 'tools', 'gen_golden.py')).read()
    main = src[src.index('def main():'):]
    default_path = main[main.index("t_start = time.time()"):]
    for fn in ('g1_prompts(', 'g2_tiling(', 'g3_g4(', 'g5_g6(', 'g6b_ragged(', 'g7_vla(', 'g7b_trace(', 'g7c_integrators(', 'g7d_general_masks(', 'g10_flow_matching(', 'g10b_flow_matching_vlm(',
               'g8_sft_grads(', 'g11_packed('):
        assert fn in default_path, f'{fn} missing from the default path of tools/gen_golden.py'


def test_committed_bench_line_keeps_the_contract():
    """The bench line committed with the round (profiles/r06*_bench_line.json, produced by `python bench.py` on the GPU box) carries what the contract asks of it:
    the headline metric of BASELINE.json with `roofline` (measured traffic) and `cpu_baseline`, the SFT side line with its own CPU baseline, a consistent value."""
    import glob
    import json
    lines = sorted(p for p in glob.glob(os.path.join(ROOT, 'profiles', 'r06*_bench_line.json')) if 'chunk' not in os.path.basename(p) and 'sft' not in os.path.basename(p))
    assert lines
    d = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'action_chunks_per_sec' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'bf16'
    assert 'workload' in d['config'] and 'model' not in d['config'] and 'reference_mode' in d['config']
    assert abs(d['value'] - d['n_gpus'] * 1e3 / d['ms_per_step']) < 0.02 * d['value']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['traffic'] is not None and 0.95 < r['traffic'] / r['bytes_per_launch'] < 1.2       # counter traffic ~ algorithmic bytes: no wasted re-reads
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    s = d['sft']
    assert s['metric'] == 'sft_tokens_per_sec' and s['fwd_bwd_ms'] < s['ms_per_step'] and 'cpu_baseline' in s
    # r05: the headline is the reference's own eight-tensor call, the phases are also timed behind their real predecessors, the forced-DP step is on the line
    assert d['reference_signature']['is_headline'] is True and abs(d['reference_signature']['ms_per_chunk'] - d['ms_per_step']) < 1e-6
    ic = d['phases']['in_chain']
    assert abs(ic['sum_ms'] - d['phases']['chunk_graph_ms']) < 0.1
    assert 'NCCL_MAX_NCHANNELS' in s['exchange']
    # r06: both clocks of the dominant kernel with `frac` on the larger one; the SFT side numbers; the forced-DP step in BOTH exchange modes
    assert r['us_per_launch'] >= max(r['us_per_launch_in_chain'], r['us_per_launch_rocprof'] or 0.0) - 1e-9 and 'rocprof_source' in r
    assert s['recompute_ms_per_step'] > s['ms_per_step'] and s['micro_batch4']['tokens_per_step'] == 4 * s['tokens_per_rank_step'] and s['exchange']['mode'] == 'none'
    f = s['forced_dp_world1_ms']
    assert f['pg']['exchange']['mode'] == 'pg' and f['capi']['exchange'] == {'mode': 'capi', 'comm_cus': 32, 'compute_cus': 224, 'cu_masks': True}
    assert f['pg']['ms_per_step'] > s['ms_per_step'] and f['capi']['ms_per_step'] > s['ms_per_step']


def test_cu_budget_moves_the_tile_choice_on_both_sides():
    """vlaser_set_cu_budget (ABI 6): the C heuristics and their Python mirror count single-round grids against the same number; out-of-range values only read it."""
    from vlaser_amd import ops, _lib
    assert ops.set_cu_budget(256) == 256
    try:
        full = ops.gemm_tile_config(560, 17920)              # 3 x 70 = 210 tiles of 192x256: one round on 256 CUs
        assert full[0] == 1900
        assert ops.set_cu_budget(200) == 256 and ops._CU_BUDGET == 200
        assert ops.gemm_tile_config(560, 17920)[0] != 1900   # 210 > 200: no longer a single round
        assert _lib.lib().vlaser_set_cu_budget(7) == 200 and _lib.lib().vlaser_set_cu_budget(1000) == 200      # refused, unchanged
    finally:
        ops.set_cu_budget(256)
    assert ops._CU_BUDGET == 256 and ops.gemm_tile_config(560, 17920) == full
