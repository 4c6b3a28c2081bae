"""Host-side integer / string logic pinned bit-exactly against reference-generated goldens (G1, G2, G4)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from vlaser_amd import prep


def test_chat_prompt_strings(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'g1_prompts.json')))
    sysmsg = g['system_message']
    assert prep.get_conv_template('internvl2_5').system_message == sysmsg
    for name in ('chat_1tile', 'chat_13tiles'):
        c = g[name]
        q, _, t = prep.build_chat_query('internvl2_5', sysmsg, c['question'], c['num_patches_list'], 256)
        assert hashlib.sha256(q.encode('utf-8')).hexdigest() == c['prompt_sha']
        assert q[:120] == c['prompt_head'] and q[-80:] == c['prompt_tail']
        assert q.count('<IMG_CONTEXT>') == c['img_count'] == 256 * sum(c['num_patches_list'])
        assert t.sep.strip() == '<|im_end|>'
    assert g['special'] == {'<IMG_CONTEXT>': 151667, '<img>': 151665, '</img>': 151666, '<|im_end|>': 151645,
                            '<|endoftext|>': 151643, '<|im_start|>': 151644}
    assert g['vocab_len'] == 151674
    assert g['chat_1tile']['img_first'] == 41 and g['chat_1tile']['img_contiguous']


def test_vla_prompt_layout(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'g1_prompts.json')))
    for name in ('vla_spoon', 'vla_carrot'):
        c = g[name]
        q = prep.build_vla_query(c['text'])
        assert q.count('<IMG_CONTEXT>') == 256 and q.startswith('<|im_start|>system\nNone<|im_end|>\n<|im_start|>user\n<img>')
        assert c['img_first'] == 10 and c['img_count'] == 256
        # ids: 10 prefix + 256 image + text, right-padded with <|endoftext|> to 384
        non_img = c['ids_nonimg']
        assert len(non_img) == 384 - 256 and non_img.count(151643) == 384 - c['n_valid']
    img = torch.zeros(1, 1, 3, 448, 448, dtype=torch.uint8)
    img[0, 0, 0] = 255; img[0, 0, 1] = 128; img[0, 0, 2, :100] = 7
    pv = prep.vla_normalize_images(img)
    probe = [float(pv[0, c, 0, 0]) for c in range(3)] + [float(pv[0, 2, 50, 0])]
    np.testing.assert_allclose(probe, g['vla_spoon']['pixel_probe'], rtol=0, atol=1e-6)
    # all 256 byte values x 3 channels through the reference's InternVLAProcessor: the host normalisation must reproduce the fp32 values BIT FOR BIT
    # (the device kernel vlaser_normalize_u8 is tested bit-exact against this host function over every byte value, tests/test_ops_gpu.py)
    ramp = torch.zeros(1, 1, 3, 448, 448, dtype=torch.uint8)
    ramp[0, 0, :, 0, :256] = torch.arange(256, dtype=torch.uint8)
    pr = prep.vla_normalize_images(ramp)[0, :, 0, :256]
    assert np.array_equal(pr.numpy(), np.array(g['pixel_ramp256'], dtype=np.float32))


def test_dynamic_grid(golden_dir):
    rows = json.load(open(os.path.join(golden_dir, 'g2_tiling.json')))
    for r in rows:
        cols_rows = prep.dynamic_grid(r['w'], r['h'], 1, r['max_num'], 448)
        assert list(cols_rows) == r['grid'], r
        n = cols_rows[0] * cols_rows[1]
        assert (n + (1 if n != 1 else 0)) == r['n_tiles'], r


def test_dynamic_preprocess_tiles():
    from PIL import Image
    im = Image.new('RGB', (640, 480))
    tiles = prep.dynamic_preprocess(im, max_num=12, use_thumbnail=True)
    assert len(tiles) == 13 and all(t.size == (448, 448) for t in tiles)
    assert prep.load_image(im).shape == (13, 3, 448, 448)


def test_vla_masks(golden_dir):
    d = np.load(os.path.join(golden_dir, 'g3g4_shuffle_masks.npz'))
    for n_valid in (277, 384, 1, 300):
        am = torch.zeros(2, 384, dtype=torch.long)
        am[0, :n_valid] = 1
        am[1, :max(1, n_valid - 17)] = 1
        m, vp, pp, ap = prep.build_causal_mask_and_position_ids(am, torch.float32)
        assert np.array_equal((m == 0).numpy().astype(np.uint8), d[f'mask_{n_valid}_zero'])
        assert set(m.unique().tolist()) <= {0.0, torch.finfo(torch.float32).min}
        m1, m2 = prep.split_full_mask_into_submasks(m)
        assert np.array_equal((m1 == 0).numpy().astype(np.uint8), d[f'mask_{n_valid}_sub1_zero'])
        assert np.array_equal((m2 == 0).numpy().astype(np.uint8), d[f'mask_{n_valid}_sub2_zero'])
        assert np.array_equal(vp.numpy(), d[f'pos_{n_valid}_vlm'])
        assert np.array_equal(pp.numpy(), d[f'pos_{n_valid}_pro'])
        assert np.array_equal(ap.numpy(), d[f'pos_{n_valid}_act'])
        assert prep.mask_to_descriptor(m1).tolist() == [n_valid, max(1, n_valid - 17)]


def test_normalize_bound_roundtrip():
    lo, hi = torch.tensor([-0.1, 0.2]), torch.tensor([0.3, 0.9])
    x = torch.tensor([[0.0, 0.5], [0.3, 0.2]])
    y = prep.normalize_bound(x, lo, hi)
    assert y.min() >= -1 and y.max() <= 1
    torch.testing.assert_close(prep.denormalize_bound(y, lo, hi), x, rtol=0, atol=1e-6)


def test_hf_config_and_checkpoint_round_trip(tmp_path):
    """config.json in the layout of the reference's vendored InternVL3 config (keys as found there; sizes of Vlaser-2B) ->
    VlaserConfig; sharded safetensors + index written and read back with the key names unchanged."""
    from vlaser_amd import config as C
    hf = {'architectures': ['InternVLChatModel'], 'downsample_ratio': 0.5, 'dynamic_image_size': True, 'force_image_size': 448,
          'max_dynamic_patch': 12, 'min_dynamic_patch': 1, 'model_type': 'internvl_chat', 'ps_version': 'v2', 'select_layer': -1,
          'system_message': None, 'template': 'internvl2_5', 'tie_word_embeddings': False, 'use_thumbnail': True,
          'llm_config': {'architectures': ['Qwen2ForCausalLM'], 'hidden_size': 1536, 'intermediate_size': 8960, 'max_position_embeddings': 32768,
                         'num_attention_heads': 12, 'num_hidden_layers': 28, 'num_key_value_heads': 2, 'rms_norm_eps': 1e-06,
                         'rope_theta': 1000000.0, 'tie_word_embeddings': False, 'vocab_size': 151674},
          'vision_config': {'hidden_act': 'gelu', 'hidden_size': 1024, 'image_size': 448, 'initializer_factor': 0.1, 'intermediate_size': 4096,
                            'layer_norm_eps': 1e-06, 'norm_type': 'layer_norm', 'num_attention_heads': 16, 'num_hidden_layers': 24,
                            'patch_size': 14, 'qk_normalization': False, 'qkv_bias': True}}
    cfg = C.from_hf_config(hf)
    assert cfg == C.vlaser_2b()
    assert C.from_hf_config(C.to_hf_config(C.vlaser_8b())) == C.vlaser_8b()
    bad = json.loads(json.dumps(hf)); bad['llm_config'].update(hidden_size=896, num_attention_heads=14)        # InternVL3-1B: head_dim 64
    with pytest.raises(ValueError):
        C.from_hf_config(bad)
    bad = json.loads(json.dumps(hf)); bad['llm_config']['architectures'] = ['InternLM2ForCausalLM']
    with pytest.raises(ValueError):
        C.from_hf_config(bad)
    sd = {'language_model.model.norm.weight': torch.arange(8, dtype=torch.bfloat16), 'mlp1.0.bias': torch.ones(5),
          'vision_model.embeddings.class_embedding': torch.zeros(1, 1, 4, dtype=torch.bfloat16)}
    C.save_hf_checkpoint(str(tmp_path), cfg, sd, max_shard_bytes=20)                 # forces several shards
    assert len([f for f in os.listdir(tmp_path) if f.endswith('.safetensors')]) >= 2
    hf2, sd2 = C.load_hf_checkpoint(str(tmp_path))
    assert C.from_hf_config(hf2) == cfg and set(sd2) == set(sd) and all(torch.equal(sd2[k], sd[k]) for k in sd)


def test_env_adapter_vs_reference(golden_dir):
    """Bridge / WidowX adapter (proprio in, env actions out) and its rotation helpers against the reference's own functions
    (tools/gen_golden_adapter.py), incl. the identity rotation and a gimbal-lock pitch."""
    from vlaser_amd import adapter as A
    d = np.load(os.path.join(golden_dir, 'g8_adapter.npz'))
    for q, m, e in zip(d['quat'], d['quat2mat'], d['mat2euler']):
        np.testing.assert_allclose(A.quat2mat(q), m, rtol=0, atol=1e-14)
        np.testing.assert_allclose(A.mat2euler(m), e, rtol=0, atol=1e-12)
    for e, ax, ang in zip(d['euler'], d['axangle_axis'], d['axangle_angle']):
        axis, angle = A.euler2axangle(*e)
        np.testing.assert_allclose(axis, ax, rtol=0, atol=1e-12)
        np.testing.assert_allclose(angle, ang, rtol=0, atol=1e-12)
    stats = {k1: {k2: d[f'stats_{k1}_{k2}'].tolist() for k2 in ('p01', 'p99', 'mean', 'std')} for k1 in ('proprio', 'action')}
    for kind in ('bound', 'gaussian'):
        ad = A.BridgeSimplerAdapter(stats, action_normalization_type=kind, proprio_normalization_type=kind)
        for eef, raw, nb in zip(d['eef_pos'], d['raw_proprio'], d[f'proprio_{kind}']):
            r = ad.preprocess_proprio({'agent': {'eef_pos': eef}})
            np.testing.assert_allclose(r, raw, rtol=0, atol=1e-12)
            np.testing.assert_allclose(ad.normalize_proprio(r), nb, rtol=0, atol=1e-12)
        for a, ref in zip(d['actions'], d[f'post_{kind}']):
            out = ad.postprocess(a)
            np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12)
            assert set(np.unique(out[:, -1])) <= {-1.0, 1.0}


def test_check_block_mask_accepts_only_the_descriptor_pattern():
    """VLATrainer.forward_backward validates a dense causal_mask instead of ignoring it."""
    from vlaser_amd import prep
    am = torch.zeros(2, 384, dtype=torch.long)
    am[0, :277] = 1
    am[1, :] = 1
    m, _, _, _ = prep.build_causal_mask_and_position_ids(am, torch.float32)
    prep.check_block_mask(m, [277, 384])
    bad = m.clone()
    bad[0, 0, 385, 100] = torch.finfo(torch.float32).min          # an action row that cannot see a valid prefix key
    with pytest.raises(ValueError):
        prep.check_block_mask(bad, [277, 384])
    with pytest.raises(ValueError):
        prep.check_block_mask(m, [276, 384])                      # disagrees with the pad count
    dc = m.clone()
    dc[0, 0, 300, 5] = 0                                          # rows of padded positions are "don't care"
    prep.check_block_mask(dc, [277, 384])


def test_split_batch_length_from_labels_and_mask():
    """id 0 is a real Qwen token; a sample that ENDS in it keeps its trailing supervised positions."""
    from vlaser_amd import config as C
    from vlaser_amd.sft import SFTModel
    m = object.__new__(SFTModel)                                  # host-side logic only: no GPU needed
    m.cfg = C.truncated(C.vlaser_2b(), 1, 1)
    m.img_context_token_id = m.cfg.img_context_token_id
    ids = torch.tensor([[5, 6, 7, 0], [5, 6, 0, 0]])
    lab = torch.tensor([[-100, 6, 7, 0], [-100, 6, -100, -100]])
    pv = torch.zeros(2, 3, 448, 448)
    flags = torch.zeros(2, 1, dtype=torch.long)                   # text-only samples: one dummy tile each
    out = m._split_batch(pv, ids, lab, flags)
    assert [o[1].shape[1] for o in out] == [4, 2] and [o[4] for o in out] == [3, 1]
    am = torch.tensor([[1, 1, 1, 1], [1, 1, 1, 0]])               # the collator's mask wins when given
    out = m._split_batch(pv, ids, lab, flags, am)
    assert [o[1].shape[1] for o in out] == [4, 3]


def test_fold_action_encoder_equals_the_unfolded_module():
    """ops.fold_action_encoder (host constants of vlaser_vla_step): W21 a + C[s] must be linear_2([temb(t_s) || linear_1(a)]) of the oracle's
    ActionEncoder (modules.py:25-56) for every Euler step, up to the bf16 rounding of the time embedding the fold applies (fp32 module here)."""
    from oracle import vla as ovla
    from vlaser_amd import ops
    torch.manual_seed(3)
    W, ad, n, mp = 256, 7, 10, 10000.0
    sd = {'action_encoder.linear_1.weight': torch.randn(W, ad) * 0.3, 'action_encoder.linear_1.bias': torch.randn(W) * 0.1,
          'action_encoder.linear_2.weight': torch.randn(W, 2 * W) * 0.05, 'action_encoder.linear_2.bias': torch.randn(W) * 0.1,
          'action_encoder.linear_3.weight': torch.randn(W, W) * 0.05, 'action_encoder.linear_3.bias': torch.randn(W) * 0.1}
    w21, cs = ops.fold_action_encoder(sd['action_encoder.linear_1.weight'], sd['action_encoder.linear_1.bias'], sd['action_encoder.linear_2.weight'],
                                      sd['action_encoder.linear_2.bias'], W, ad, n, mp)
    assert w21.shape == (W, ad) and cs.shape == (n, W)
    a = torch.randn(1, 4, ad)
    for s in (0, 3, 9):
        temb = ovla.sinusoidal_pos_emb(torch.full((1,), s / n), W, mp)
        ref = ovla.action_encoder(sd, a, temb)
        pre = a[0] @ w21.t() + cs[s][None]
        got = torch.nn.functional.linear(torch.nn.functional.silu(pre), sd['action_encoder.linear_3.weight'], sd['action_encoder.linear_3.bias'])
        assert (got - ref[0]).abs().max().item() < 5e-3 * max(1.0, ref.abs().max().item()), s      # the only difference: temb rounded to bf16 inside the fold
