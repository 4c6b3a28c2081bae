"""Pin the key names (and shapes) of the reference's VLA `.pt` checkpoints (`torch.save({"model": model.state_dict(), ...})`,
Vlaser_VLA/Simpler/src/agent/train.py:639-672) by building the REFERENCE's own `PiZero.__init__` module tree on the `meta`
device (no weights needed) -- build container only.  Writes tests/golden/g9_vla_state_keys.json.

What is stubbed (nothing that registers a parameter): hydra / omegaconf (a dict-with-attributes config built from the reference's
own eval YAML, `${...}` interpolation resolved), `from_pretrained` of the tokenizer / config / InternVLChatModel (constructed
from the vendored InternVL3 config patched to the 2B widths), flash-attention selection, gradient-checkpointing toggles.
"""
import json
import os
import re
import sys
import importlib

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402

ref_import.install()
os.environ['IMAGE_448'] = '1'


class Cfg(dict):
    """Minimal DictConfig: attribute access, .get, nested."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def get(self, k, default=None):
        return self[k] if k in self and self[k] is not None else default


def to_cfg(x):
    if isinstance(x, dict):
        return Cfg({k: to_cfg(v) for k, v in x.items()})
    if isinstance(x, list):
        return [to_cfg(v) for v in x]
    return x


def resolve(root):
    pat = re.compile(r'^\$\{([A-Za-z0-9_.]+)\}$')

    def look(path):
        cur = root
        for p in path.split('.'):
            cur = cur[p]
        return cur

    def walk(node):
        for k, v in list(node.items()):
            if isinstance(v, dict):
                walk(v)
            elif isinstance(v, str):
                m = pat.match(v)
                seen = 0
                while m and seen < 8:
                    v = look(m.group(1))
                    m = pat.match(v) if isinstance(v, str) else None
                    seen += 1
                node[k] = v
    for _ in range(3):
        walk(root)
    return root


def merge(a, b):
    out = Cfg(a)
    for k, v in b.items():
        out[k] = merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def instantiate(node, *a, **kw):
    mod, cls = node['_target_'].rsplit('.', 1)
    return getattr(importlib.import_module(mod), cls)(node['config'])


def main():
    import omegaconf
    import hydra
    omegaconf.OmegaConf.merge = staticmethod(merge)
    hydra.utils.instantiate = instantiate
    raw = yaml.safe_load(open(ref_import.REF + '/Vlaser_VLA/Simpler/config/eval/bridge_internvl_448.yaml'))
    raw.pop('hydra', None); raw.pop('log_dir', None)
    cfg = to_cfg(resolve(raw))
    from internvl.model.internvl_chat import InternVLChatModel, InternVLChatConfig
    import transformers
    from src.model.vla import pizero_internvl as RP
    import src.model.vla.joint_model as JM
    JM.OmegaConf.merge = staticmethod(merge)
    RP.hydra.utils.instantiate = instantiate

    tok = ref_import.tokenizer()
    transformers.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: tok)

    def cfg_from_pretrained(path, **kw):
        rawc = json.load(open(ref_import.TOK_DIR + '/config.json'))
        for k in ('architectures', 'auto_map', 'model_type', '_commit_hash', '_name_or_path', 'transformers_version', 'torch_dtype'):
            rawc.pop(k, None)
        rawc['llm_config'].update(hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12, num_key_value_heads=2)
        rawc['vision_config']['drop_path_rate'] = 0.0
        return InternVLChatConfig(**rawc)
    InternVLChatConfig.from_pretrained = staticmethod(cfg_from_pretrained)

    def model_from_pretrained(path, torch_dtype=None, config=None, **kw):
        config.llm_config._attn_implementation = 'eager'
        m = InternVLChatModel(config, use_flash_attn=False)
        m.language_model._set_gradient_checkpointing = lambda *a, **k: None

        def resize(n, *a, **k):          # HF's mean-resizing calls .item(): same module shapes, built directly (pizero_internvl.py:85)
            lm = m.language_model
            hs = lm.config.hidden_size
            lm.model.embed_tokens = torch.nn.Embedding(n, hs, device='meta')
            lm.lm_head = torch.nn.Linear(hs, n, bias=False, device='meta')
        m.language_model.resize_token_embeddings = resize
        return m
    InternVLChatModel.from_pretrained = staticmethod(model_from_pretrained)
    orig_linspace = torch.linspace
    torch.linspace = lambda *a, **k: orig_linspace(*a, **{**k, 'device': 'cpu'})       # InternVisionEncoder calls .item() on it (drop-path schedule)
    try:
        with torch.device('meta'):
            model = RP.PiZero(cfg)
    finally:
        torch.linspace = orig_linspace
    model.tie_action_proprio_weights()
    sd = model.state_dict()
    out = {'source': 'reference PiZero.__init__ + tie_action_proprio_weights on the meta device (tools/gen_golden_vla_keys.py)',
           'config': 'Vlaser_VLA/Simpler/config/eval/bridge_internvl_448.yaml, InternVL3-2B widths',
           'keys': {k: list(v.shape) for k, v in sd.items()}}
    path = os.path.join(ROOT, 'tests', 'golden', 'g9_vla_state_keys.json')
    json.dump(out, open(path, 'w'), indent=0, separators=(',', ':'))
    print(len(sd), 'keys ->', path, os.path.getsize(path), 'bytes')
    import collections
    pre = collections.Counter(re.sub(r'\.\d+\.', '.N.', k).rsplit('.', 2)[0] if k.count('.') > 3 else k for k in sd)
    for k, v in sorted(pre.items()):
        print(v, k)


if __name__ == '__main__':
    main()
