"""Host-side preparation that defines the inputs of the hot path: chat prompt assembly,
<IMG_CONTEXT> expansion, dynamic tiling grid, ImageNet normalisation, VLA prompt / masks / position ids, WidowX
proprio / action (de)normalisation.  Pure Python / torch-CPU integer and string logic, pinned bit-exactly by the
golden fixtures in tests/golden (generated from the reference by tools/gen_golden.py)."""
from dataclasses import dataclass, field
from typing import List

import torch

IMG_START_TOKEN = '<img>'
IMG_END_TOKEN = '</img>'
IMG_CONTEXT_TOKEN = '<IMG_CONTEXT>'
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
VLA_MEAN, VLA_STD = (0.4850, 0.4560, 0.4060), (0.2290, 0.2240, 0.2250)      # InternVLAProcessor's constants (processing.py:303-304)


# ------------------------------------------------------------------------------------------------ conversation
@dataclass
class Conversation:
    """MPT-style template (reference conversation.py:238-247): system + sep, then role + message + sep per turn,
    a trailing bare role for the pending assistant turn."""
    name: str
    system_template: str
    system_message: str
    roles: tuple
    sep: str
    messages: List[list] = field(default_factory=list)

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_prompt(self):
        ret = self.system_template.format(system_message=self.system_message) + self.sep
        for role, message in self.messages:
            ret += (role + message + self.sep) if message else role
        return ret


_TEMPLATES = {
    # reference conversation.py:393-402
    'internvl2_5': dict(
        system_template='<|im_start|>system\n{system_message}',
        system_message='你是书生·万象，英文名是InternVL，是由上海人工智能实验室、清华大学及多家合作单位联合开发的多模态大语言模型。',
        roles=('<|im_start|>user\n', '<|im_start|>assistant\n'), sep='<|im_end|>\n'),
}


def get_conv_template(name) -> Conversation:
    t = _TEMPLATES[name]
    return Conversation(name=name, system_template=t['system_template'], system_message=t['system_message'],
                        roles=t['roles'], sep=t['sep'], messages=[])


def build_chat_query(template_name, system_message, question, num_patches_list, num_image_token, history=None,
                     has_pixels=True):
    """String assembly of InternVLChatModel.chat (modeling_internvl_chat.py:347-375)."""
    if history is None and has_pixels and '<image>' not in question:
        question = '<image>\n' + question
    t = get_conv_template(template_name)
    t.system_message = system_message
    for (old_q, old_a) in (history or []):
        t.append_message(t.roles[0], old_q)
        t.append_message(t.roles[1], old_a)
    t.append_message(t.roles[0], question)
    t.append_message(t.roles[1], None)
    query = t.get_prompt()
    for n in num_patches_list:
        query = query.replace('<image>', IMG_START_TOKEN + IMG_CONTEXT_TOKEN * num_image_token * n + IMG_END_TOKEN, 1)
    return query, question, t


def build_vla_query(text, num_image_token=256):
    """Hard-coded VLA prompt (Vlaser_VLA/Simpler/src/model/vla/processing.py:355-358): system message "None"."""
    img = IMG_CONTEXT_TOKEN * num_image_token
    return ('<|im_start|>system\nNone<|im_end|>\n<|im_start|>user\n<img>{}</img>\n{}<|im_end|>\n'
            '<|im_start|>assistant\n').format(img, text)


# ------------------------------------------------------------------------------------------------ dynamic tiling
def find_closest_aspect_ratio(aspect_ratio, target_ratios, width, height, image_size):
    """dataset.py:813-827: closest grid aspect ratio; ties go to the larger grid only if the image is big enough."""
    best_diff, best = float('inf'), (1, 1)
    area = width * height
    for r in target_ratios:
        diff = abs(aspect_ratio - r[0] / r[1])
        if diff < best_diff:
            best_diff, best = diff, r
        elif diff == best_diff:
            if area > 0.5 * image_size * image_size * r[0] * r[1]:
                best = r
    return best


def dynamic_grid(width, height, min_num=1, max_num=12, image_size=448):
    """Grid (cols, rows) chosen by dynamic_preprocess (dataset.py:830-866)."""
    ratios = set((i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1)
                 if min_num <= i * j <= max_num)
    ratios = sorted(ratios, key=lambda x: x[0] * x[1])
    return find_closest_aspect_ratio(width / height, ratios, width, height, image_size)


def dynamic_preprocess(image, min_num=1, max_num=12, image_size=448, use_thumbnail=False):
    """PIL image -> list of image_size x image_size tiles (+ thumbnail iff more than one tile)."""
    w, h = image.size
    cols, rows = dynamic_grid(w, h, min_num, max_num, image_size)
    tw, th = image_size * cols, image_size * rows
    resized = image.resize((tw, th))
    tiles = []
    for i in range(cols * rows):
        box = ((i % cols) * image_size, (i // cols) * image_size, ((i % cols) + 1) * image_size, ((i // cols) + 1) * image_size)
        tiles.append(resized.crop(box))
    if use_thumbnail and len(tiles) != 1:
        tiles.append(image.resize((image_size, image_size)))
    return tiles


def normalize_tiles(tiles, image_size=448):
    """Eval branch of build_transform (dataset.py:276-310): RGB -> bicubic resize -> /255 -> ImageNet normalise."""
    import numpy as np
    from PIL import Image
    out = []
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    for im in tiles:
        im = im.convert('RGB').resize((image_size, image_size), Image.BICUBIC)
        t = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        out.append((t - mean) / std)
    return torch.stack(out)


def load_image(image, input_size=448, max_num=12):
    """eval_example.py:76-82."""
    return normalize_tiles(dynamic_preprocess(image, image_size=input_size, use_thumbnail=True, max_num=max_num), input_size)


def vla_normalize_images(images_u8):
    """InternVLAProcessor image path (processing.py:303-311): uint8 [B,n,3,H,W] -> fp32 [B*n,3,H,W].
    NB the processor's constants are MEAN (0.4850, 0.4560, 0.4060), STD (0.2290, 0.2240, 0.2250)."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 5
    mean = torch.tensor([0.4850, 0.4560, 0.4060])[None, None, :, None, None]
    std = torch.tensor([0.2290, 0.2240, 0.2250])[None, None, :, None, None]
    x = (images_u8 * (1 / 255.0) - mean) / std
    return x.flatten(0, 1)


# ------------------------------------------------------------------------------------------------ VLA masks
class InternVLAProcessor:
    """Mirror of the reference's `InternVLAProcessor` (Vlaser_VLA/Simpler/src/model/vla/processing.py:250-366): same constructor,
    same `__call__(text, images uint8 [B,3,H,W]) -> {pixel_values, input_ids, attention_mask}` contract (ImageNet normalisation
    :303-311, the hard-coded chat string with system message "None" :358, right padding to `max_seq_len` :360-363).  The number of
    <IMG_CONTEXT> tokens comes from the constructor (`num_image_tokens`), not from the IMAGE_448 environment variable."""
    IMAGE_TOKEN = '<image>'

    def __init__(self, tokenizer, num_image_tokens, max_seq_len, actions=None, tokenizer_padding='max_length', num_images=1):
        self.image_seq_length = num_image_tokens
        self.max_seq_len = max_seq_len
        self.tokenizer_padding = tokenizer_padding
        self.image_token_id = tokenizer.convert_tokens_to_ids(self.IMAGE_TOKEN)
        self.tokenizer = tokenizer
        self.num_images = num_images

    def __call__(self, text, images, truncation=True, actions=None):
        assert len(images) == len(text), f'Received {len(images)} images for {len(text)} prompts.'
        assert images.dtype == torch.uint8, f'Expected uint8 tensor for images, got {images.dtype}.'
        pixel_values = vla_normalize_images(images)
        self.tokenizer.model_max_length = self.max_seq_len
        query = [build_vla_query(prompt, self.image_seq_length * self.num_images) for prompt in text]
        inputs = self.tokenizer(query, return_tensors='pt', max_length=self.max_seq_len, padding=self.tokenizer_padding, truncation=truncation)
        return {'pixel_values': pixel_values, **inputs}


def build_causal_mask_and_position_ids(attention_mask, dtype, max_image_text_tokens=384, num_proprio_tokens=1,
                                       num_action_tokens=4):
    """Dense block mask + position ids of PiZero.build_causal_mask_and_position_ids (pizero_internvl.py:517-587).
    Kept for API parity; the kernels take (valid_len, blk_start) descriptors instead (mask_to_descriptor)."""
    bsz, T = attention_mask.shape
    ps, pe = max_image_text_tokens, max_image_text_tokens + num_proprio_tokens
    Lt = T + num_action_tokens + 1
    m = torch.full((bsz, Lt, Lt), torch.finfo(dtype).min, dtype=dtype)
    for i, c in enumerate(attention_mask.sum(dim=1).tolist()):
        m[i, :c, :c] = 0
        m[i, ps:, :c] = 0
    m[:, ps:pe, ps:pe] = 0
    m[:, pe:, ps:] = 0
    vlm = torch.arange(1, max_image_text_tokens + 1).repeat(bsz, 1)
    pro = torch.arange(1, num_proprio_tokens + 1).repeat(bsz, 1)
    act = torch.arange(num_proprio_tokens + 1, num_proprio_tokens + num_action_tokens + 1).repeat(bsz, 1)
    return m.unsqueeze(1), vlm, pro, act


def split_full_mask_into_submasks(mask, max_image_text_tokens=384, num_proprio_tokens=1, num_action_tokens=4):
    """pizero_internvl.py:589-603."""
    n = max_image_text_tokens + num_proprio_tokens
    return mask[..., :n, :n], mask[..., -num_action_tokens:, :]


def mask_to_descriptor(image_text_proprio_mask, max_image_text_tokens=384):
    """Recover valid_len[b] from the dense mask (row of the proprio token: zeros over the valid prefix) and check the
    mask has the block structure the kernels assume."""
    row = image_text_proprio_mask[:, 0, max_image_text_tokens, :max_image_text_tokens]
    valid = (row == 0).sum(-1).to(torch.int32)
    idx = torch.arange(max_image_text_tokens, device=row.device)[None]
    if not bool(((row == 0) == (idx < valid[:, None])).all()):
        raise ValueError('image/text mask is not a contiguous valid prefix')
    return valid


def check_block_mask(causal_mask, valid_len, max_image_text_tokens=384, num_proprio_tokens=1, num_action_tokens=4):
    """Raise unless the dense mask [B,1,L,L] (L = T + proprio + action) has EXACTLY the visibility pattern that
    `build_causal_mask_and_position_ids` derives from `valid_len` -- the only pattern the (valid_len, blk_start) descriptors of the
    kernels can express (pizero_internvl.py:517-587).  Rows of padded image/text positions are "don't care" (nobody attends to them)."""
    B = causal_mask.shape[0]
    vl = [int(x) for x in torch.as_tensor(valid_len).reshape(-1).tolist()]
    am = torch.zeros(B, max_image_text_tokens, dtype=torch.long)
    for b, c in enumerate(vl):
        am[b, :c] = 1
    want, _, _, _ = build_causal_mask_and_position_ids(am, torch.float32, max_image_text_tokens, num_proprio_tokens, num_action_tokens)
    got = causal_mask.detach().to('cpu')
    if got.shape != want.shape:
        raise ValueError(f'causal_mask has shape {tuple(got.shape)}, expected {tuple(want.shape)}')
    for b, c in enumerate(vl):
        rows = torch.cat([torch.arange(c), torch.arange(max_image_text_tokens, want.shape[-1])])      # valid prefix rows + proprio / action rows
        if not torch.equal(got[b, 0, rows] == 0, want[b, 0, rows] == 0):
            raise ValueError('causal_mask is not the prefix + trailing-block mask of build_causal_mask_and_position_ids for this pad count: '
                             'the kernels express visibility as (valid_len, blk_start) descriptors and cannot honour it')


# ------------------------------------------------------------------------------------------------ WidowX adapter
def normalize_bound(x, lo, hi, clip_min=-1.0, clip_max=1.0, eps=1e-8):
    """env_adapter/base.py:8-31: map [p01, p99] -> [-1, 1] and clip."""
    y = 2 * (x - lo) / (hi - lo + eps) - 1
    return y.clip(clip_min, clip_max)


def denormalize_bound(x, lo, hi, clip_min=-1.0, clip_max=1.0, eps=1e-8):
    """env_adapter/base.py:33-49."""
    x = x.clip(clip_min, clip_max)
    return (x - clip_min) / (clip_max - clip_min) * (hi - lo + eps) + lo
