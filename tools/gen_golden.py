"""Generate golden vectors under tests/golden/ by running the REFERENCE's own code (build container only).

    python tools/gen_golden.py            # writes tests/golden/*.npz|json

The reference is imported from /root/reference through tools/ref_import.py (stubs for packages that are not
installed).  Weights come from vlaser_amd.synth (deterministic integer hash, identical on CPU and GPU) and are
loaded into the reference modules with strict key matching, which also pins the checkpoint key names.
Models use the true Vlaser-2B widths with truncated depth so that fixtures stay small and the CPU oracle
replays them in seconds.  Fixtures hold inputs' seeds and output slices only -- never reference source.
"""
import copy
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402

ref_import.install()
from internvl.model.internvl_chat import InternVLChatModel, InternVLChatConfig  # noqa: E402
from internvl.conversation import get_conv_template  # noqa: E402
from internvl.train.dataset import preprocess_internvl2_5, dynamic_preprocess, find_closest_aspect_ratio  # noqa: E402
from src.model.vla import pizero_internvl as RP  # noqa: E402
from src.model.vla.joint_model import JointModel  # noqa: E402
from src.model.vla.modules import ActionEncoder, SinusoidalPosEmb  # noqa: E402
from src.model.kv_cache import KVCache  # noqa: E402
from transformers import Qwen2ForCausalLM  # noqa: E402

from vlaser_amd import config as C, synth  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)

VIT_L, LLM_L = 2, 2


def ref_config(cfg: C.VlaserConfig):
    raw = json.load(open(ref_import.TOK_DIR + '/config.json'))
    for k in ('architectures', 'auto_map', 'model_type', '_commit_hash', '_name_or_path',
              'transformers_version', 'torch_dtype'):
        raw.pop(k, None)
    l = raw['llm_config']
    l.update(hidden_size=cfg.llm.hidden_size, intermediate_size=cfg.llm.intermediate_size,
             num_hidden_layers=cfg.llm.num_hidden_layers, num_attention_heads=cfg.llm.num_attention_heads,
             num_key_value_heads=cfg.llm.num_key_value_heads, vocab_size=cfg.llm.vocab_size)
    raw['vision_config']['drop_path_rate'] = 0.0
    raw['vision_config']['num_hidden_layers'] = cfg.vision.num_hidden_layers
    return InternVLChatConfig(**raw)


def build_ref_vlm(cfg, sd):
    rc = ref_config(cfg)
    m = InternVLChatModel(rc, use_flash_attn=False).eval()
    m.language_model.config._attn_implementation = 'eager'
    missing = m.load_state_dict(sd, strict=True)
    print('ref VLM loaded', missing)
    return m


def subsample(t, n=4096):
    f = t.detach().float().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return idx.numpy().astype(np.int64), f[idx].numpy()


def stats(t):
    f = t.detach().double()
    return np.array([f.mean().item(), f.abs().mean().item(), f.norm().item()])


def pack(prefix, t, d):
    idx, val = subsample(t)
    d[prefix + '_idx'] = idx
    d[prefix + '_val'] = val
    d[prefix + '_stats'] = stats(t)
    d[prefix + '_shape'] = np.array(t.shape)


# ----------------------------------------------------------------------------------------------- G1: prompts
def g1_prompts(tok):
    out = {}
    m = types.SimpleNamespace(template='internvl2_5', num_image_token=256)
    sysmsg = get_conv_template('internvl2_5').system_message

    def chat_query(question, num_patches_list):
        # the reference's own string assembly: modeling_internvl_chat.py:347-375
        if '<image>' not in question:
            question = '<image>\n' + question
        t = get_conv_template('internvl2_5')
        t.system_message = sysmsg
        t.append_message(t.roles[0], question)
        t.append_message(t.roles[1], None)
        q = t.get_prompt()
        for n in num_patches_list:
            q = q.replace('<image>', '<img>' + '<IMG_CONTEXT>' * 256 * n + '</img>', 1)
        return q
    cases = {'chat_1tile': ('What objects are on the table and where should the robot move next?', [1]),
             'chat_13tiles': ('Point to the red cup.', [13])}
    for name, (q, npl) in cases.items():
        s = chat_query(q, npl)
        ids = tok(s, return_tensors='pt')['input_ids'][0]
        out[name] = dict(question=q, num_patches_list=npl, prompt_sha=_sha(s), prompt_head=s[:120],
                         prompt_tail=s[-80:], n_tokens=int(ids.numel()),
                         ids_nonimg=[int(x) for x in ids[ids != 151667]],
                         img_first=int((ids == 151667).nonzero()[0]), img_count=int((ids == 151667).sum()),
                         img_contiguous=bool(((ids == 151667).nonzero().flatten().diff() == 1).all()))
    # VLA prompt: processing.py:358-363 (system "None", right-padded to 384 with <|endoftext|>)
    from src.model.vla.processing import InternVLAProcessor
    os.environ['IMAGE_448'] = '1'
    proc = InternVLAProcessor(tok, num_image_tokens=256, max_seq_len=384, tokenizer_padding='max_length')
    tok.padding_side = 'right'
    img = torch.zeros(1, 1, 3, 448, 448, dtype=torch.uint8)     # [B, n_img, 3, H, W] (processing.py:40)
    img[0, 0, 0] = 255; img[0, 0, 1] = 128; img[0, 0, 2, :100] = 7
    for name, text in {'vla_spoon': 'put the spoon on the towel', 'vla_carrot': 'put carrot on plate'}.items():
        r = proc([text], img)
        ids = r['input_ids'][0]
        out[name] = dict(text=text, ids_nonimg=[int(x) for x in ids[ids != 151667]],
                         n_valid=int(r['attention_mask'][0].sum()), img_first=int((ids == 151667).nonzero()[0]),
                         img_count=int((ids == 151667).sum()),
                         pixel_probe=[float(r['pixel_values'][0, c, 0, 0]) for c in range(3)] +
                                     [float(r['pixel_values'][0, 2, 50, 0])])
    # every byte value in every channel through the reference's processor (the 4-value probe above pins too little)
    ramp = torch.zeros(1, 1, 3, 448, 448, dtype=torch.uint8)
    ramp[0, 0, :, 0, :256] = torch.arange(256, dtype=torch.uint8)
    r = proc(['x'], ramp)
    out['pixel_ramp256'] = [[float(v) for v in r['pixel_values'][0, c, 0, :256]] for c in range(3)]
    out['system_message'] = sysmsg
    out['special'] = {t: int(tok.convert_tokens_to_ids(t)) for t in
                      ['<IMG_CONTEXT>', '<img>', '</img>', '<|im_end|>', '<|endoftext|>', '<|im_start|>']}
    out['vocab_len'] = len(tok)
    # SFT label masking (preprocess_internvl2_5, dataset.py:711-810)
    src = [[{'from': 'human', 'value': '<image>\nWhat is this?'}, {'from': 'gpt', 'value': 'A cat.'}]]
    tok.model_max_length = 4096
    r = preprocess_internvl2_5('internvl2_5', copy.deepcopy(src), tok, [256], group_by_length=True, ds_name='g', num_image=1)
    ids, lab = r['input_ids'][0], r['labels'][0]
    out['sft_sample'] = dict(n_tokens=int(ids.numel()), ids_nonimg=[int(x) for x in ids[ids != 151667]],
                             supervised_pos=[int(i) for i in (lab != -100).nonzero().flatten()],
                             supervised_ids=[int(x) for x in lab[lab != -100]])
    json.dump(out, open(os.path.join(OUT, 'g1_prompts.json'), 'w'), ensure_ascii=False, indent=1)
    print('G1 ok', {k: v.get('n_tokens', v.get('n_valid')) for k, v in out.items() if isinstance(v, dict) and 'ids_nonimg' in v})


def _sha(s):
    import hashlib
    return hashlib.sha256(s.encode('utf-8')).hexdigest()


# ----------------------------------------------------------------------------------------------- G2: tiling
def g2_tiling():
    from PIL import Image
    rows = []
    for (w, h) in [(640, 480), (1920, 1080), (1000, 300), (448, 448), (300, 1000), (224, 224), (800, 800),
                   (1280, 720), (720, 1280), (2000, 500), (500, 2000), (449, 448), (1344, 896), (33, 4000)]:
        for max_num in (6, 12):
            img = Image.new('RGB', (w, h))
            tiles = dynamic_preprocess(img, min_num=1, max_num=max_num, image_size=448, use_thumbnail=True)
            target_ratios = sorted(set((i, j) for n in range(1, max_num + 1) for i in range(1, n + 1)
                                       for j in range(1, n + 1) if 1 <= i * j <= max_num),
                                   key=lambda x: x[0] * x[1])
            grid = find_closest_aspect_ratio(w / h, target_ratios, w, h, 448)
            rows.append(dict(w=w, h=h, max_num=max_num, n_tiles=len(tiles), grid=list(grid)))
    json.dump(rows, open(os.path.join(OUT, 'g2_tiling.json'), 'w'), indent=0)
    print('G2 ok', rows[:3])


# ----------------------------------------------------------------------------------------------- G3 / G4
def g3_g4(ref_vlm):
    x = torch.arange(2 * 32 * 32 * 8, dtype=torch.float32).reshape(2, 32, 32, 8)
    y = ref_vlm.pixel_shuffle(x, scale_factor=0.5)
    d = {'ps_in_shape': np.array(x.shape), 'ps_out': y.numpy().astype(np.int32)}
    fake = types.SimpleNamespace(max_image_text_tokens=384, num_proprio_tokens=1, num_action_tokens=4,
                                 debug_causal=False)
    for n_valid in (277, 384, 1, 300):
        am = torch.zeros(2, 384, dtype=torch.long)
        am[0, :n_valid] = 1
        am[1, :max(1, n_valid - 17)] = 1
        mask, vp, pp, ap = RP.PiZero.build_causal_mask_and_position_ids(fake, am, torch.float32)
        m1, m2 = RP.PiZero.split_full_mask_into_submasks(fake, mask)
        d[f'mask_{n_valid}_zero'] = (mask == 0).numpy().astype(np.uint8)       # structure; values are 0 or finfo.min
        assert set(mask.unique().tolist()) <= {0.0, torch.finfo(torch.float32).min}
        d[f'mask_{n_valid}_sub_shapes'] = np.array(list(m1.shape) + list(m2.shape))
        d[f'mask_{n_valid}_sub1_zero'] = (m1 == 0).numpy().astype(np.uint8)
        d[f'mask_{n_valid}_sub2_zero'] = (m2 == 0).numpy().astype(np.uint8)
        d[f'pos_{n_valid}_vlm'] = vp.numpy(); d[f'pos_{n_valid}_pro'] = pp.numpy(); d[f'pos_{n_valid}_act'] = ap.numpy()
    np.savez_compressed(os.path.join(OUT, 'g3g4_shuffle_masks.npz'), **d)
    print('G3/G4 ok')


# ----------------------------------------------------------------------------------------------- G5 / G6
def make_inputs(cfg, seed, n_tiles, n_text, tok_specials=True):
    g = torch.Generator().manual_seed(seed)
    pv = torch.randn(n_tiles, 3, 448, 448, generator=g)
    # 48 "template" ids + 256*T image + n_text random text ids (SURVEY §8d synthetic prompt)
    pre = torch.randint(0, 151643, (41,), generator=g)
    post = torch.randint(0, 151643, (7 + n_text,), generator=g)
    ids = torch.cat([pre, torch.full((256 * n_tiles,), 151667), post])[None]
    return pv, ids


def g5_g6(cfg, sd, ref_vlm):
    d = {}
    pv, ids = make_inputs(cfg, seed=0, n_tiles=1, n_text=32)
    d['seed'] = np.array(0); d['input_ids'] = ids.numpy()
    # G5a: ViT embeddings / per-layer / extract_feature
    emb = ref_vlm.vision_model.embeddings(pv)
    pack('vit_emb', emb, d)
    h = emb
    for i, layer in enumerate(ref_vlm.vision_model.encoder.layers):
        h = layer(h)
        pack(f'vit_l{i}', h, d)
    feat = ref_vlm.extract_feature(pv)
    pack('vit_feat', feat, d)
    # G5b: LLM logits for the full prompt (forward with labels=None path) -- via language_model
    ref_vlm.img_context_token_id = 151667
    out = ref_vlm(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids),
                  image_flags=torch.ones(1, 1, dtype=torch.long), return_dict=True)
    logits = out.logits
    pack('logits', logits[:, -4:], d)
    top = logits[0, -1].topk(8)
    d['last_top_vals'] = top.values.numpy(); d['last_top_ids'] = top.indices.numpy()
    # SFT loss (labels on the last 16 positions)
    labels = torch.full_like(ids, -100); labels[0, -16:] = ids[0, -16:]
    out2 = ref_vlm(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids),
                   image_flags=torch.ones(1, 1, dtype=torch.long), labels=labels, return_dict=True)
    d['sft_loss'] = np.array(out2.loss.item())
    # G6: greedy ids (reference generate -> HF GenerationMixin)
    gen = ref_vlm.generate(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids),
                           max_new_tokens=8, min_new_tokens=8, do_sample=False, eos_token_id=151645,
                           pad_token_id=151643, output_scores=True, return_dict_in_generate=True)
    d['greedy_ids'] = gen.sequences.numpy()
    sc = torch.stack(gen.scores, dim=1)[0]           # [8, V]
    t2 = sc.topk(2, dim=-1).values
    d['greedy_margin'] = (t2[:, 0] - t2[:, 1]).numpy()
    d['greedy_top_vals'] = sc.topk(4, dim=-1).values.numpy()
    np.savez_compressed(os.path.join(OUT, 'g5g6_vlm.npz'), **d)
    print('G5/G6 ok greedy', d['greedy_ids'], 'margins', d['greedy_margin'], 'loss', d['sft_loss'])


def g6b_ragged(cfg, ref_vlm):
    """batch_chat's generate call (modeling_internvl_chat.py:326-341): two prompts of different length, LEFT padded
    (tokenizer.padding_side = 'left' :318) with an attention mask; greedy ids from the reference's own generate."""
    ref_vlm.img_context_token_id = 151667
    pad = 151643
    for sa in range(11, 400, 2):                      # first seed pair whose greedy margins clear bf16 logit noise
        pv_a, ids_a = make_inputs(cfg, seed=sa, n_tiles=1, n_text=32)
        pv_b, ids_b = make_inputs(cfg, seed=sa + 1, n_tiles=1, n_text=9)
        S = max(ids_a.shape[1], ids_b.shape[1])
        ids = torch.full((2, S), pad, dtype=torch.long)
        am = torch.zeros(2, S, dtype=torch.long)
        for b, x in enumerate((ids_a, ids_b)):
            ids[b, S - x.shape[1]:] = x[0]
            am[b, S - x.shape[1]:] = 1
        pv = torch.cat([pv_a, pv_b])
        gen = ref_vlm.generate(pixel_values=pv, input_ids=ids, attention_mask=am, max_new_tokens=6, min_new_tokens=6,
                               do_sample=False, eos_token_id=151645, pad_token_id=pad, output_scores=True,
                               return_dict_in_generate=True)
        sc = torch.stack(gen.scores, dim=1)              # [2, 8, V]
        t2 = sc.topk(2, dim=-1).values
        margin = t2[..., 0] - t2[..., 1]
        print('seeds', sa, sa + 1, 'min margin', float(margin.min()))
        if float(margin.min()) > 0.03:
            break
    else:
        raise SystemExit('no seed pair with a clear greedy margin')
    d = dict(seeds=np.array([sa, sa + 1]), n_text=np.array([32, 9]), input_ids=ids.numpy(), attention_mask=am.numpy(),
             greedy_ids=gen.sequences.numpy(), greedy_margin=margin.numpy(),
             greedy_top_vals=sc.topk(4, dim=-1).values.numpy())
    np.savez_compressed(os.path.join(OUT, 'g6b_ragged.npz'), **d)
    print('G6b ok greedy', d['greedy_ids'], 'margins', d['greedy_margin'])


# ----------------------------------------------------------------------------------------------- G7: infer_action
class _FakeJoint:
    """Carries exactly the attributes JointModel.forward / build_mixture_caches read (joint_model.py:702-814)."""

    def __init__(self, mixtures, n_layers):
        self.mixtures = mixtures
        self.num_hidden_layers = n_layers
        self.backbone_type = 'INTERNVL'
        self.use_flash_attention = False
        self.stop_grad_to_vlm = False
        self.training = False
        self.cache_names = ['vlm', 'proprio']

    def build_mixture_caches(self):
        return JointModel.build_mixture_caches(self)

    def __call__(self, **kw):
        return JointModel.forward(self, **kw)


class _Mix(torch.nn.Module):
    def __init__(self, layers, norm):
        super().__init__()
        self.layers, self.norm = layers, norm

    def forward_norm(self, x, cond=None):
        return self.norm(x)


def g7_vla(vla, sd, ref_vlm):
    ecfg = copy.deepcopy(ref_vlm.config.llm_config)
    ecfg.hidden_size = vla.action_hidden_size; ecfg.intermediate_size = vla.action_intermediate_size
    ecfg.head_dim = 128
    expert = Qwen2ForCausalLM(ecfg).eval()
    expert.config._attn_implementation = 'eager'
    esd = {k[len('action_expert.'):]: v for k, v in sd.items() if k.startswith('action_expert.')}
    r = expert.load_state_dict(esd, strict=False)
    assert set(r.missing_keys) <= {'model.embed_tokens.weight', 'lm_head.weight'} and not r.unexpected_keys, r
    W = vla.action_hidden_size
    aenc = ActionEncoder(7, W, time_cond=True)
    aenc.load_state_dict({k[len('action_encoder.'):]: v for k, v in sd.items() if k.startswith('action_encoder.')})
    pro = torch.nn.Linear(7, W); pro.load_state_dict({'weight': sd['proprio_encoder.weight'], 'bias': sd['proprio_encoder.bias']})
    dec = torch.nn.Linear(W, 7); dec.load_state_dict({'weight': sd['action_decoder.weight'], 'bias': sd['action_decoder.bias']})
    vlm_mix = _Mix(ref_vlm.language_model.model.layers, ref_vlm.language_model.model.norm)
    act_mix = _Mix(expert.model.layers, expert.model.norm)
    mixtures = torch.nn.ModuleDict({'vlm': vlm_mix, 'proprio': act_mix, 'action': act_mix})
    fake = types.SimpleNamespace(
        num_images=1, imgfeat=False, image_448=True, no_img=False, image_token_index=151667, pad_token_id=151643,
        joint_model=_FakeJoint(mixtures, vla.base.llm.num_hidden_layers),
        embed_tokens=ref_vlm.language_model.model.embed_tokens,
        vision_tower=lambda pv: ref_vlm.vision_model(pixel_values=pv, return_dict=True),
        multi_modal_projector=ref_vlm.mlp1, proprio_encoder=pro,
        internvl_model=types.SimpleNamespace(language_model=ref_vlm.language_model, action_expert=expert),
        horizon_steps=vla.num_action_tokens, action_dim=7, num_inference_steps=vla.num_inference_steps,
        time_embedding=SinusoidalPosEmb(W, vla.time_max_period), action_expert_adaptive_mode=None,
        action_encoder=aenc, action_decoder=dec, integration_method='euler',
        final_action_clip_value=vla.final_action_clip_value, cfg=types.SimpleNamespace(horizon_steps=vla.horizon_steps),
        max_image_text_tokens=384, num_proprio_tokens=1, num_action_tokens=4, debug_causal=False)
    fake.pixel_shuffle = types.MethodType(RP.PiZero.pixel_shuffle, fake)
    fake._forward_siglip_and_text_embedding = types.MethodType(RP.PiZero._forward_siglip_and_text_embedding, fake)

    d = {}
    for case, (seed, n_valid) in {'a': (0, 277), 'b': (1, 300)}.items():
        g = torch.Generator().manual_seed(seed)
        pv = torch.randn(1, 3, 448, 448, generator=g)
        ids = torch.full((1, 384), 151643)
        n_text = n_valid - 10 - 256
        ids[0, :10] = torch.randint(0, 151643, (10,), generator=g)
        ids[0, 10:266] = 151667
        ids[0, 266:266 + n_text] = torch.randint(0, 151643, (n_text,), generator=g)
        am = (ids != 151643).long()
        proprio = torch.rand(1, 1, 7, generator=g) * 2 - 1
        mask, vp, pp, ap = RP.PiZero.build_causal_mask_and_position_ids(fake, am, torch.float32)
        m1, m2 = RP.PiZero.split_full_mask_into_submasks(fake, mask)
        torch.manual_seed(1234 + seed)
        noise = torch.randn(1, 4, 7)
        torch.manual_seed(1234 + seed)
        act = RP.PiZero.infer_action(fake, ids, pv, m1, m2, vp, pp, ap, proprio)
        d[f'{case}_seed'] = np.array(seed); d[f'{case}_n_valid'] = np.array(n_valid)
        d[f'{case}_input_ids'] = ids.numpy(); d[f'{case}_proprio'] = proprio.numpy()
        d[f'{case}_noise'] = noise.numpy(); d[f'{case}_action'] = act.numpy()
        print('G7', case, act)
    np.savez_compressed(os.path.join(OUT, 'g7_vla.npz'), **d)
    fake._ref_vlm = ref_vlm
    return fake


def g7c_integrators(fake, vla):
    """G7c: `integration_method` other than "euler" (pizero_internvl.py:164,910-922,1309-1331).  The reference's `model_step` closure ignores its arguments and returns
    the decoder output of THIS step's joint pass, so "heun" and "rk4" re-combine one velocity: heun == euler bit for bit, rk4 == euler up to the rounding of
    (dt / 6) * (k1 + 2 k2 + 2 k3 + k4).  The fixture pins exactly that: the reference's own chunks for the two G7 cases under each method."""
    d = {}
    for case, (seed, n_valid) in {'a': (0, 277), 'b': (1, 300)}.items():
        pv, ids, am, proprio, mask, vp, pp, ap = _g7_case(fake, seed, n_valid)
        m1, m2 = RP.PiZero.split_full_mask_into_submasks(fake, mask)
        for method in ('euler', 'heun', 'rk4'):
            fake.integration_method = method
            torch.manual_seed(1234 + seed)
            act = RP.PiZero.infer_action(fake, ids, pv, m1, m2, vp, pp, ap, proprio)
            d[f'{case}_{method}_action'] = act.numpy()
        fake.integration_method = 'euler'
        print('G7c', case, 'heun == euler:', bool((d[f'{case}_heun_action'] == d[f'{case}_euler_action']).all()), ' max|rk4 - euler|',
              float(np.abs(d[f'{case}_rk4_action'] - d[f'{case}_euler_action']).max()))
    np.savez_compressed(os.path.join(OUT, 'g7c_integrators.npz'), **d)


def g7d_general_masks(fake, vla):
    """G7d: the reference's `infer_action` under masks its own builder never produces -- the additive [B,1,Sq,Skv] tensors go straight into `eager_attention_forward`
    (joint_model.py:636-656), so any pattern / finite bias is legal input.  Three cases: the prompt LEFT-padded by 50 positions (the valid tokens are not a prefix; position ids
    as the builder returns them), the same with the text after the image attending causally, and the right-padded prompt with finite biases on every visible entry + a hole in
    one action row.  Inputs that cannot be rebuilt from a seed (the masks) are stored."""
    FMIN = torch.finfo(torch.float32).min
    T, na = 384, vla.num_action_tokens
    Lt = T + 1 + na
    d = {}
    for case, (seed, left, kind) in {'a': (0, 50, 'plain'), 'b': (1, 50, 'causal_text'), 'c': (2, 0, 'bias')}.items():
        g = torch.Generator().manual_seed(100 + seed)
        pv = torch.randn(1, 3, 448, 448, generator=g)
        ids = torch.full((1, T), 151643)
        ids[0, left:left + 10] = torch.randint(0, 151643, (10,), generator=g)
        ids[0, left + 10:left + 266] = 151667
        ids[0, left + 266:left + 277] = torch.randint(0, 151643, (11,), generator=g)
        am = (ids != 151643).long()
        proprio = torch.rand(1, 1, 7, generator=g) * 2 - 1
        noise = torch.randn(1, na, 7, generator=g)
        _, vp, pp, ap = RP.PiZero.build_causal_mask_and_position_ids(fake, am, torch.float32)
        m = torch.full((1, Lt, Lt), FMIN)
        vis = am[0].bool()
        m[0, :T, :T][vis[:, None] & vis[None, :]] = 0.0
        m[0, T:, :T][:, vis] = 0.0
        m[:, T, T] = 0.0
        m[:, T + 1:, T:] = 0.0
        if kind == 'causal_text':
            idx = am[0].nonzero().flatten()[-11:]
            for a_, i in enumerate(idx):
                m[0, i, idx[a_ + 1:]] = FMIN
        if kind == 'bias':
            m = torch.where(m < -1e30, m, torch.randn(1, Lt, Lt, generator=g))
            m[:, T + 2, 20:60] = FMIN
        mask = m[:, None]
        m1, m2 = RP.PiZero.split_full_mask_into_submasks(fake, mask)
        real_randn = torch.randn
        torch.randn = lambda *a, **k: noise.clone()          # the reference draws its noise inside infer_action (:879-881): hand it this case's
        try:
            act = RP.PiZero.infer_action(fake, ids, pv, m1, m2, vp, pp, ap, proprio)
        finally:
            torch.randn = real_randn
        d[f'{case}_input_ids'], d[f'{case}_pixel_seed'], d[f'{case}_proprio'], d[f'{case}_noise'] = ids.numpy(), np.array(100 + seed), proprio.numpy(), noise.numpy()
        d[f'{case}_mask_bits'] = np.packbits((mask[0, 0] > -1e30).numpy())                      # visibility as bits ...
        d[f'{case}_mask_bias'] = (torch.where(mask[0, 0] < -1e30, torch.zeros(()), mask[0, 0]).to(torch.float16).numpy() if kind == 'bias' else np.zeros(0, np.float16))
        if kind == 'bias':       # (... and the biases as the fp16 values the mask is REBUILT from: the reference ran on exactly those)
            mask_r = torch.where(mask[0, 0] < -1e30, mask[0, 0], torch.from_numpy(d[f'{case}_mask_bias']).float())[None, None]
            m1, m2 = RP.PiZero.split_full_mask_into_submasks(fake, mask_r)
            torch.randn = lambda *a, **k: noise.clone()
            try:
                act = RP.PiZero.infer_action(fake, ids, pv, m1, m2, vp, pp, ap, proprio)
            finally:
                torch.randn = real_randn
        d[f'{case}_action'] = act.numpy()
        print('G7d', case, kind, 'left pad', left, 'action', act.flatten()[:4].tolist())
    np.savez_compressed(os.path.join(OUT, 'g7d_general_masks.npz'), **d)


def _g7_case(fake, seed, n_valid):
    g = torch.Generator().manual_seed(seed)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    ids = torch.full((1, 384), 151643)
    n_text = n_valid - 10 - 256
    ids[0, :10] = torch.randint(0, 151643, (10,), generator=g)
    ids[0, 10:266] = 151667
    ids[0, 266:266 + n_text] = torch.randint(0, 151643, (n_text,), generator=g)
    am = (ids != 151643).long()
    proprio = torch.rand(1, 1, 7, generator=g) * 2 - 1
    mask, vp, pp, ap = RP.PiZero.build_causal_mask_and_position_ids(fake, am, torch.float32)
    return pv, ids, am, proprio, mask, vp, pp, ap


def _infer_action_naive(self, input_ids, pixel_values, causal_mask, vlm_position_ids, proprio_position_ids, action_position_ids, proprios):
    """The reference's `infer_action_naive` (pizero_internvl.py:938-1003) cannot run on the InternVL path as shipped: it never
    passes `position_embeddings_all`, which `forward_mixture_attn_internvl` indexes unconditionally (joint_model.py:549, KeyError
    'vlm').  This driver makes the SAME calls in the same order -- the reference's own JointModel.forward with all three mixtures
    active and cache_mode="no_append" every Euler step -- with that one argument supplied the way `infer_action` builds it
    (:854-855,895)."""
    bsz = pixel_values.size(0)
    kv_caches = self.joint_model.build_mixture_caches()
    inputs_embeds = self._forward_siglip_and_text_embedding(input_ids, pixel_values)
    proprio_embeds = self.proprio_encoder(proprios)
    rot_v = self.internvl_model.language_model.model.rotary_emb
    rot_a = self.internvl_model.action_expert.model.rotary_emb
    action = torch.randn((bsz, self.horizon_steps, self.action_dim))
    delta_t = 1.0 / self.num_inference_steps
    t = torch.zeros(bsz)
    for _ in range(self.num_inference_steps):
        time_cond = self.time_embedding(t)
        action_embeds = self.action_encoder(action, time_cond)
        pe = {'vlm': rot_v(inputs_embeds, vlm_position_ids), 'proprio': rot_a(proprio_embeds, proprio_position_ids),
              'action': rot_a(action_embeds, action_position_ids)}
        action_embeds = self.joint_model(
            attention_mask=causal_mask,
            position_ids_all={'vlm': vlm_position_ids, 'proprio': proprio_position_ids, 'action': action_position_ids},
            embeds_all={'vlm': inputs_embeds.clone(), 'proprio': proprio_embeds.clone(), 'action': action_embeds},
            time_cond=time_cond, kv_caches=kv_caches, position_embeddings_all=pe, cache_mode='no_append')['action']
        action = action + delta_t * self.action_decoder(action_embeds)
        t = t + delta_t
    if self.final_action_clip_value is not None:
        action = torch.clamp(action, -self.final_action_clip_value, self.final_action_clip_value)
    return action


def g7b_trace(fake, vla):
    """G7b: what happens INSIDE the reference's infer_action for the two G7 cases -- the velocity the action decoder returns at every
    Euler step (forward hook on `action_decoder`), slices of the K / V caches the joint prefill leaves behind (kv_cache.py:6-46;
    first and last layer, a few VLM positions + the proprio token), and the cache-free path (`_infer_action_naive` above: the
    reference's JointModel.forward with all three mixtures in one joint pass per step, pizero_internvl.py:938-1003)."""
    d = {}
    vels = []
    hook = fake.action_decoder.register_forward_hook(lambda m, i, o: vels.append(o.detach().clone()))
    orig_build = fake.joint_model.build_mixture_caches
    for case, (seed, n_valid) in {'a': (0, 277), 'b': (1, 300)}.items():
        pv, ids, am, proprio, mask, vp, pp, ap = _g7_case(fake, seed, n_valid)
        m1, m2 = RP.PiZero.split_full_mask_into_submasks(fake, mask)
        grabbed = {}

        def build():
            grabbed['c'] = orig_build()
            return grabbed['c']
        fake.joint_model.build_mixture_caches = build
        vels.clear()
        torch.manual_seed(1234 + seed)
        act = RP.PiZero.infer_action(fake, ids, pv, m1, m2, vp, pp, ap, proprio)
        d[f'{case}_action'] = act.numpy()
        d[f'{case}_vel'] = torch.stack(vels, 0)[:, 0].numpy()                    # [n_steps, 4, 7]
        kv = grabbed['c']
        nl = len(kv['vlm'].key_cache)
        pos = np.array([0, 9, 10, 137, 265, n_valid - 1])
        d[f'{case}_kv_pos'] = pos
        for li in (0, nl - 1):
            kc, vc = kv['vlm'].get(li)                                            # [1, n_kv, 384, 128], K after RoPE
            d[f'{case}_k_vlm_L{li}'] = kc[0][:, pos].numpy(); d[f'{case}_v_vlm_L{li}'] = vc[0][:, pos].numpy()
            kp, vp_ = kv['proprio'].get(li)                                       # [1, n_kv, 1, 128]
            d[f'{case}_k_pro_L{li}'] = kp[0, :, 0].numpy(); d[f'{case}_v_pro_L{li}'] = vp_[0, :, 0].numpy()
        d[f'{case}_n_layers'] = np.array(nl)
        # the reference's cache-free path on the same noise
        fake.joint_model.build_mixture_caches = orig_build
        vels.clear()
        torch.manual_seed(1234 + seed)
        naive = _infer_action_naive(fake, ids, pv, mask, vp, pp, ap, proprio)
        d[f'{case}_action_naive'] = naive.numpy()
        d[f'{case}_vel_naive'] = torch.stack(vels, 0)[:, 0].numpy()
        print('G7b', case, 'cached vs naive max diff', (act - naive).abs().max().item())
    hook.remove()
    fake.joint_model.build_mixture_caches = orig_build
    np.savez_compressed(os.path.join(OUT, 'g7b_vla_trace.npz'), **d)


def g10_flow_matching(fake, vla):
    """G10: the flow-matching TRAINING step of the VLA from the reference's own `PiZero.forward` (pizero_internvl.py:1064-1197) +
    torch autograd: loss and the gradients of the action-expert parameter group (`action_expert_parameters`, :358-374 -- action /
    proprio encoders, action decoder, the expert's decoder layers and final norm; the reference's default `train_vlm=False`
    optimises exactly this group, train.py:246-255).  x0 is drawn inside the method (`randn_like`): recorded by re-seeding."""
    import types as _t
    fake.flow_sig_min = 0.001
    fake.psi_t = _t.MethodType(RP.PiZero.psi_t, fake)
    expert = fake.internvl_model.action_expert
    mods = {'action_expert.model.': expert.model, 'action_encoder.': fake.action_encoder, 'proprio_encoder.': fake.proprio_encoder,
            'action_decoder.': fake.action_decoder}
    params = {}
    for pre, m in mods.items():
        for n, p_ in m.named_parameters():
            p_.requires_grad_(True); p_.grad = None
            params[pre + n] = p_
    for p_ in list(fake.internvl_model.language_model.parameters()) + list(fake.embed_tokens.parameters()) + list(fake.multi_modal_projector.parameters()):
        p_.requires_grad_(False)
    d = {}
    for case, (seed, n_valid, tval) in {'a': (0, 277, 0.35), 'b': (1, 300, 0.8)}.items():
        pv, ids, am, proprio, mask, vp, pp, ap = _g7_case(fake, seed, n_valid)
        g = torch.Generator().manual_seed(100 + seed)
        actions = torch.rand(1, 4, 7, generator=g) * 2 - 1
        t = torch.tensor([tval])
        torch.manual_seed(4321 + seed)
        x0 = torch.randn_like(actions)
        torch.manual_seed(4321 + seed)
        for p_ in params.values():
            p_.grad = None
        with torch.enable_grad():
            loss = RP.PiZero.forward(fake, ids, pv, mask, vp, pp, ap, proprio, actions, t)
            loss.backward()
        d[f'{case}_actions'] = actions.numpy(); d[f'{case}_t'] = t.numpy(); d[f'{case}_x0'] = x0.numpy(); d[f'{case}_loss'] = np.array(loss.item())
        names = []
        for n, p_ in params.items():
            if p_.grad is None:
                continue
            gr = p_.grad.detach().double().flatten()
            names.append(n)
            d[f'{case}_norm::{n}'] = np.array(gr.norm().item())
            k = min(32, gr.numel())
            idx = (torch.arange(k, dtype=torch.int64) * (gr.numel() - 1)) // max(1, k - 1)
            d[f'{case}_idx::{n}'] = idx.numpy(); d[f'{case}_val::{n}'] = gr[idx].numpy()
        d[f'{case}_names'] = np.array(names)
        print('G10', case, 'loss', loss.item(), len(names), 'gradient tensors')
    for p_ in params.values():
        p_.requires_grad_(False); p_.grad = None
    np.savez_compressed(os.path.join(OUT, 'g10_flow_matching.npz'), **d)


def g10b_flow_matching_vlm(fake, vla, ref_vlm):
    """G10b: the same flow-matching training loss with the reference's SECOND parameter group unfrozen --
    `trainable_vlm_parameters` (pizero_internvl.py:405-411) = vision_tower + multi_modal_projector + joint_model.mixtures["vlm"] (the LLM's
    decoder layers and final norm; embed_tokens is NOT in the group) -- on top of `action_expert_parameters`, i.e. `train_vlm: True`
    (train.py:270-295).  Loss (identical to G10's) and every gradient `PiZero.forward` + autograd produces: the gradient reaches the VLM
    only through the K / V the image/text rows hand to the proprio / action rows (block mask, :517-587); the last layer's post-attention
    parameters and the final norm of the VLM mixture get none (`final_layer_post_attn_skip_names`)."""
    import types as _t
    fake.flow_sig_min = 0.001
    fake.psi_t = _t.MethodType(RP.PiZero.psi_t, fake)
    expert = fake.internvl_model.action_expert
    mods = {'action_expert.model.': expert.model, 'action_encoder.': fake.action_encoder, 'proprio_encoder.': fake.proprio_encoder,
            'action_decoder.': fake.action_decoder, 'vision_model.': ref_vlm.vision_model, 'mlp1.': ref_vlm.mlp1,
            'language_model.model.layers.': ref_vlm.language_model.model.layers, 'language_model.model.norm.': ref_vlm.language_model.model.norm}
    params = {}
    for pre, m in mods.items():
        for n, p_ in m.named_parameters():
            p_.requires_grad_(True); p_.grad = None
            params[pre + n] = p_
    for p_ in list(fake.embed_tokens.parameters()) + list(ref_vlm.language_model.lm_head.parameters()):
        p_.requires_grad_(False)
    d = {}
    for case, (seed, n_valid, tval) in {'a': (0, 277, 0.35), 'b': (1, 300, 0.8)}.items():
        pv, ids, am, proprio, mask, vp, pp, ap = _g7_case(fake, seed, n_valid)
        g = torch.Generator().manual_seed(100 + seed)
        actions = torch.rand(1, 4, 7, generator=g) * 2 - 1
        t = torch.tensor([tval])
        torch.manual_seed(4321 + seed)
        x0 = torch.randn_like(actions)
        torch.manual_seed(4321 + seed)
        for p_ in params.values():
            p_.grad = None
        with torch.enable_grad():
            loss = RP.PiZero.forward(fake, ids, pv, mask, vp, pp, ap, proprio, actions, t)
            loss.backward()
        d[f'{case}_actions'] = actions.numpy(); d[f'{case}_t'] = t.numpy(); d[f'{case}_x0'] = x0.numpy(); d[f'{case}_loss'] = np.array(loss.item())
        names, nograd = [], []
        for n, p_ in params.items():
            if p_.grad is None or float(p_.grad.abs().max()) == 0.0:
                nograd.append(n)
                continue
            gr = p_.grad.detach().double().flatten()
            names.append(n)
            d[f'{case}_norm::{n}'] = np.array(gr.norm().item())
            k = min(32, gr.numel())
            idx = (torch.arange(k, dtype=torch.int64) * (gr.numel() - 1)) // max(1, k - 1)
            d[f'{case}_idx::{n}'] = idx.numpy(); d[f'{case}_val::{n}'] = gr[idx].numpy()
        d[f'{case}_names'] = np.array(names); d[f'{case}_nograd'] = np.array(nograd)
        print('G10b', case, 'loss', loss.item(), len(names), 'gradient tensors;', len(nograd), 'without gradient:', nograd[:6])
    for p_ in params.values():
        p_.requires_grad_(False); p_.grad = None
    np.savez_compressed(os.path.join(OUT, 'g10b_flow_matching_vlm.npz'), **d)


def g11_packed(cfg, ref_vlm):
    """G11: packed-sequence SFT loss (`--use_packed_ds`) from the reference's own `InternVLChatModel.forward` with `loss_weight`
    (modeling_internvl_chat.py:207-230).  The reference realises the block-diagonal causal attention with flash_attn_varlen_func
    behind a CUDA-only monkey patch (qwen2_packed_training_patch.py:14-101); here the SAME visibility is handed to HF's eager
    attention as an explicit 4-D additive mask, with the packed collator's restarting position ids (dataset_packed.py:517-545).
    One row = [1-tile sample | text-only sample (dummy tile, image_flags 0) | 1-tile sample]; loss_weight = 1/sqrt(n_eff) per
    sub-sequence ('square' reduction), 0 on ignored labels (:622)."""
    ref_vlm.img_context_token_id = 151667
    g = torch.Generator().manual_seed(77)
    subs, labs = [], []
    for n_img, n_text, n_lab in ((256, 22, 7), (0, 31, 9), (256, 15, 5)):
        ids = torch.cat([torch.randint(1, 151643, (9,), generator=g), torch.full((n_img,), 151667), torch.randint(1, 151643, (n_text,), generator=g)])
        lab = torch.full_like(ids, -100); lab[-n_lab:] = ids[-n_lab:]
        subs.append(ids); labs.append(lab)
    ids = torch.cat(subs)[None]; labels = torch.cat(labs)[None]
    cu = torch.tensor([0] + list(torch.tensor([len(x) for x in subs]).cumsum(0)))
    S = ids.shape[1]
    pos = torch.cat([torch.arange(len(x)) for x in subs])[None]
    w = torch.cat([torch.full((len(x),), 1.0 / math.sqrt(float((l != -100).sum()))) for x, l in zip(subs, labs)])[None]
    w = torch.where(labels == -100, torch.zeros_like(w), w)
    mask = torch.full((S, S), torch.finfo(torch.float32).min)
    for lo, hi in zip(cu[:-1].tolist(), cu[1:].tolist()):
        mask[lo:hi, lo:hi] = torch.full((hi - lo, hi - lo), torch.finfo(torch.float32).min).triu(1)
    pv = torch.randn(3, 3, 448, 448, generator=g)
    flags = torch.tensor([[1], [0], [1]])
    for n, p_ in ref_vlm.named_parameters():
        p_.requires_grad_(not n.startswith('vision_model.')); p_.grad = None
    with torch.enable_grad():
        out = ref_vlm(pixel_values=pv, input_ids=ids, attention_mask=mask[None, None], position_ids=pos, image_flags=flags, labels=labels,
                      loss_weight=w.tolist(), return_dict=True)
        out.loss.backward()
    d = {'seed': np.array(77), 'input_ids': ids.numpy(), 'labels': labels.numpy(), 'cu_seqlens': cu.numpy(), 'loss_weight': w.numpy(),
         'image_flags': flags.numpy(), 'loss': np.array(out.loss.item()), 'last_logits': out.logits[0, -1].detach().topk(8).values.numpy()}
    names = []
    for n, p_ in ref_vlm.named_parameters():
        if p_.grad is not None:
            names.append(n); d[f'norm::{n}'] = np.array(p_.grad.double().norm().item()); p_.grad = None
        p_.requires_grad_(False)
    d['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'g11_packed.npz'), **d)
    print('G11 packed loss', out.loss.item(), len(names), 'gradient tensors')


def g8_sft_grads(cfg, ref_vlm):
    """SFT step gradients from the REFERENCE's own forward + torch autograd (modeling_internvl_chat.py:143-255 with labels): the
    same sample as G5's sft_loss (seed 0, labels on the last 16 positions), vision tower frozen (freeze_backbone), every LLM /
    projector gradient summarised by its norm and a few sampled entries."""
    ref_vlm.img_context_token_id = 151667
    pv, ids = make_inputs(cfg, seed=0, n_tiles=1, n_text=32)
    labels = torch.full_like(ids, -100); labels[0, -16:] = ids[0, -16:]
    for n, p_ in ref_vlm.named_parameters():
        p_.requires_grad_(not n.startswith('vision_model.'))
    with torch.enable_grad():
        out = ref_vlm(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids), image_flags=torch.ones(1, 1, dtype=torch.long),
                      labels=labels, return_dict=True)
        out.loss.backward()
    d = {'loss': np.array(out.loss.item())}
    names = []
    for n, p_ in ref_vlm.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad.detach().double().flatten()
        names.append(n)
        d[f'norm::{n}'] = np.array(g.norm().item())
        k = min(64, g.numel())
        idx = (torch.arange(k, dtype=torch.int64) * (g.numel() - 1)) // max(1, k - 1)
        d[f'idx::{n}'] = idx.numpy(); d[f'val::{n}'] = g[idx].numpy()
        p_.grad = None
    d['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'g8_sft_grads.npz'), **d)
    print('G8 ok', len(names), 'gradient tensors, loss', d['loss'])


def main():
    if '--only-g8grad' in sys.argv:
        cfg = C.truncated(C.vlaser_2b(), VIT_L, LLM_L)
        sd = synth.vla_state_dict(C.VLAConfig(base=cfg), with_head=True)
        vlm_sd = {k: v for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))}
        return g8_sft_grads(cfg, build_ref_vlm(cfg, vlm_sd))
    if '--only-g11' in sys.argv:
        cfg = C.truncated(C.vlaser_2b(), VIT_L, LLM_L)
        sd = synth.vla_state_dict(C.VLAConfig(base=cfg), with_head=True)
        vlm_sd = {k: v for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))}
        return g11_packed(cfg, build_ref_vlm(cfg, vlm_sd))
    if '--only-g7' in sys.argv:
        cfg = C.truncated(C.vlaser_2b(), VIT_L, LLM_L)
        vla = C.VLAConfig(base=cfg)
        sd = synth.vla_state_dict(vla, with_head=True)
        vlm_sd = {k: v for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))}
        fake = g7_vla(vla, sd, build_ref_vlm(cfg, vlm_sd))
        if '--only-g10b' in sys.argv:
            return g10b_flow_matching_vlm(fake, vla, fake._ref_vlm)
        if '--only-g7c' in sys.argv:
            return g7c_integrators(fake, vla)
        if '--only-g7d' in sys.argv:
            return g7d_general_masks(fake, vla)
        if '--skip-g7b' not in sys.argv:
            g7b_trace(fake, vla)
        g7c_integrators(fake, vla)
        g7d_general_masks(fake, vla)
        g10_flow_matching(fake, vla)
        return g10b_flow_matching_vlm(fake, vla, fake._ref_vlm)
    import subprocess
    import time
    from golden_manifest import FIXTURES
    t_start = time.time() - 1.0
    if '--only-g6b' not in sys.argv:
        tok = ref_import.tokenizer()
        g1_prompts(tok)
        g2_tiling()
    cfg = C.truncated(C.vlaser_2b(), VIT_L, LLM_L)
    vla = C.VLAConfig(base=cfg)
    sd = synth.vla_state_dict(vla, with_head=True)
    vlm_sd = {k: v for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))}
    ref_vlm = build_ref_vlm(cfg, vlm_sd)
    if '--only-g6b' in sys.argv:
        return g6b_ragged(cfg, ref_vlm)
    g6b_ragged(cfg, ref_vlm)
    g3_g4(ref_vlm)
    g5_g6(cfg, sd, ref_vlm)
    fake = g7_vla(vla, sd, ref_vlm)
    g7b_trace(fake, vla)
    g7c_integrators(fake, vla)
    g7d_general_masks(fake, vla)
    g10_flow_matching(fake, vla)
    g10b_flow_matching_vlm(fake, vla, fake._ref_vlm)
    g8_sft_grads(cfg, build_ref_vlm(cfg, vlm_sd))
    g11_packed(cfg, build_ref_vlm(cfg, vlm_sd))
    meta = dict(vit_layers=VIT_L, llm_layers=LLM_L, widths='vlaser-2b', weights='vlaser_amd.synth seed 0',
                torch=torch.__version__, transformers=__import__('transformers').__version__,
                generated_by='tools/gen_golden.py (imports /root/reference)')
    json.dump(meta, open(os.path.join(OUT, 'META.json'), 'w'), indent=1)
    # the two sibling generators (their own import stubs: separate processes), then the manifest check: ONE command rebuilds every fixture
    for script in sorted({v for v in FIXTURES.values() if not v.endswith('gen_golden.py')}):
        subprocess.check_call([sys.executable, os.path.join(ROOT, script)])
    have = sorted(os.listdir(OUT))
    assert have == sorted(FIXTURES), f'tests/golden/ != tools/golden_manifest.py: {sorted(set(have) ^ set(FIXTURES))}'
    stale = [f for f in FIXTURES if os.path.getmtime(os.path.join(OUT, f)) < t_start]
    assert not stale, f'fixtures not rewritten by this run: {stale}'
    print(f'all {len(FIXTURES)} fixtures rebuilt')


if __name__ == '__main__':
    main()
