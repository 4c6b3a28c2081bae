"""Golden vectors for the WidowX/Bridge environment adapter (host-side pre/post-processing), produced by the
REFERENCE's own code (build container only): geometry helpers (quat2mat, mat2euler, euler2axangle; utils/geometry.py) and
BridgeSimplerAdapter.preprocess_proprio / SimplerAdapter.postprocess / BaseEnvAdapter normalisation (env_adapter/simpler.py,
base.py), driven on random inputs with made-up dataset statistics.  Writes tests/golden/g8_adapter.npz (inputs + outputs only)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference/Vlaser_VLA/Simpler'
sys.path.insert(0, REF)
import importlib.machinery  # noqa: E402
import transformers  # noqa: E402,F401   (before the stubs: it probes optional packages by their __spec__)
for name in ('cv2', 'tensorflow', 'simpler_env', 'simpler_env.utils', 'simpler_env.utils.env', 'simpler_env.utils.env.observation_utils'):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
sys.modules['simpler_env.utils.env.observation_utils'].get_image_from_maniskill2_obs_dict = lambda env, obs: None
sys.modules['cv2'].INTER_LANCZOS4 = 4
sys.modules['cv2'].resize = None

from src.utils.geometry import quat2mat, mat2euler, euler2axangle  # noqa: E402


def main():
    # the adapter module also imports the processors (tokenizer stack): not needed for the methods driven here
    m = types.ModuleType('src.model.vla.processing')
    m.VLAProcessor = m.InternVLAProcessor = m.InternVLAProcessor_old = object
    sys.modules['src.model.vla.processing'] = m
    from src.agent.env_adapter.simpler import BridgeSimplerAdapter, SimplerAdapter
    rng = np.random.default_rng(0)
    d = {}
    q = rng.normal(size=(16, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    d['quat'] = q
    d['quat2mat'] = np.stack([quat2mat(x) for x in q])
    d['mat2euler'] = np.stack([np.array(mat2euler(m)) for m in d['quat2mat']])
    eul = rng.uniform(-3.0, 3.0, size=(16, 3)); eul[0] = 0.0; eul[1] = [0.0, np.pi / 2, 0.3]          # identity and a gimbal-lock pitch
    d['euler'] = eul
    ax = [euler2axangle(*e) for e in eul]
    d['axangle_axis'] = np.stack([a[0] for a in ax]); d['axangle_angle'] = np.array([a[1] for a in ax])
    # adapter with made-up statistics
    stats = {'proprio': {'p01': rng.uniform(-1, 0, 7).tolist(), 'p99': rng.uniform(0.5, 1.5, 7).tolist(),
                         'mean': rng.normal(size=7).tolist(), 'std': rng.uniform(0.1, 1, 7).tolist()},
             'action': {'p01': rng.uniform(-0.05, 0, 7).tolist(), 'p99': rng.uniform(0.01, 0.06, 7).tolist(),
                        'mean': rng.normal(scale=0.01, size=7).tolist(), 'std': rng.uniform(0.005, 0.02, 7).tolist()}}
    for k1 in stats:
        for k2 in stats[k1]:
            d[f'stats_{k1}_{k2}'] = np.array(stats[k1][k2])
    fake = types.SimpleNamespace(dataset_statistics=stats, default_rot=np.array([[0, 0, 1.0], [0, 1.0, 0], [-1.0, 0, 0]]))
    for nm in ('normalize_bound', 'denormalize_bound', 'normalize_gaussian', 'denormalize_gaussian'):
        setattr(fake, nm, types.MethodType(getattr(SimplerAdapter, nm), fake))
    fake.postprocess_gripper = types.MethodType(BridgeSimplerAdapter.postprocess_gripper, fake)
    eef = np.concatenate([rng.uniform(-0.5, 0.5, (8, 3)), q[:8], rng.uniform(0, 1, (8, 1))], axis=1)     # pos, quat (wxyz), gripper
    d['eef_pos'] = eef
    d['raw_proprio'] = np.stack([BridgeSimplerAdapter.preprocess_proprio(fake, {'agent': {'eef_pos': e}}) for e in eef])
    d['proprio_bound'] = np.stack([fake.normalize_bound(p, np.array(stats['proprio']['p01']), np.array(stats['proprio']['p99']), clip_min=-1, clip_max=1)
                                   for p in d['raw_proprio']])
    d['proprio_gaussian'] = np.stack([fake.normalize_gaussian(p, np.array(stats['proprio']['mean']), np.array(stats['proprio']['std']))
                                      for p in d['raw_proprio']])
    acts = rng.uniform(-1, 1, size=(3, 4, 7)); acts[..., -1] = rng.uniform(0, 1, size=(3, 4))
    d['actions'] = acts
    for kind in ('bound', 'gaussian'):
        fake.action_normalization_type = kind
        d[f'post_{kind}'] = np.stack([SimplerAdapter.postprocess(fake, a) for a in acts])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'g8_adapter.npz'), **d)
    print('G8 ok', {k: v.shape for k, v in d.items() if k.startswith(('post', 'raw', 'axangle'))})


if __name__ == '__main__':
    main()
