"""Import shim for the *reference* (read-only, /root/reference). Build-container only.

Used by tools/gen_golden.py and tools/check_oracle_vs_ref.py to run the reference's own
Python so that golden vectors under tests/golden/ can be generated.  Nothing under tests/,
bench.py or the package imports this file: the reference does not exist on the GPU box.

Recipe follows SURVEY.md Appendix A (stub timm / peft / cv2 / ... which are not installed).
"""
import importlib.machinery as mach
import os
import sys
import types

import torch  # noqa: F401
from torch import nn
import transformers  # noqa: F401
from transformers import Qwen2ForCausalLM  # noqa: F401  (force the lazy import before stubbing)

REF = '/root/reference'
TOK_DIR = (REF + '/Vlaser_VLA/RoboTwin/policy/internvla_2B_parallel_decoding/'
           'internvl/pretrained/InternVL3-1B')


def _mk(name):
    m = types.ModuleType(name)
    m.__spec__ = mach.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def install():
    if 'timm' in sys.modules and getattr(sys.modules['timm'], '_vlaser_stub', False):
        return
    t = _mk('timm'); t._vlaser_stub = True
    _mk('timm.models'); tl = _mk('timm.models.layers')

    class DropPath(nn.Module):
        def __init__(self, p=0.):
            super().__init__(); self.p = p

        def forward(self, x):
            return x
    tl.DropPath = DropPath
    peft = _mk('peft'); peft.LoraConfig = lambda *a, **k: None
    peft.get_peft_model = lambda m, c: (_ for _ in ()).throw(RuntimeError('peft stub'))
    for n in ('cv2', 'imageio'):
        _mk(n)
    _mk('decord').VideoReader = object
    tv, tt, tf = _mk('torchvision'), _mk('torchvision.transforms'), _mk('torchvision.transforms.functional')
    tf.InterpolationMode = type('InterpolationMode', (), {'BICUBIC': 'bicubic'})
    tv.transforms = tt; tt.functional = tf
    _mk('omegaconf').OmegaConf = type('OmegaConf', (), {'merge': staticmethod(lambda a, b: a)})
    _mk('hydra').utils = _mk('hydra.utils')
    bnb = _mk('bitsandbytes'); bnb.nn = _mk('bitsandbytes.nn')
    bnb.nn.Params4bit = type('Params4bit', (nn.Parameter,), {})
    bnb.nn.Linear4bit = type('Linear4bit', (nn.Linear,), {})
    sys.path.insert(0, REF + '/Vlaser_VLM/internvl_chat')
    sys.path.insert(0, REF + '/Vlaser_VLA/Simpler')
    os.environ['INTERNVL'] = '1'


def tokenizer():
    return transformers.AutoTokenizer.from_pretrained(TOK_DIR, trust_remote_code=False, use_fast=False)
