"""Host-side environment adapter for the WidowX / Bridge SimplerEnv observations: proprioception in, env actions
out.  Mirrors `BridgeSimplerAdapter` (Vlaser_VLA/Simpler/src/agent/env_adapter/simpler.py:65-221) and `BaseEnvAdapter`
(base.py:8-49); the rotation helpers restate the 'sxyz' (static x-y-z) branches of utils/geometry.py (:49 mat2euler, :118 quat2mat,
:261 euler2axangle) from the formulas.  Pure numpy, float64 like the reference; pinned by tests/golden/g8_adapter.npz."""
import math

import numpy as np

_EPS4 = np.finfo(np.float64).eps * 4.0


def quat2mat(q):
    """Rotation matrix of a quaternion (w, x, y, z); near-zero quaternions give the identity."""
    w, x, y, z = (float(v) for v in q)
    n = w * w + x * x + y * y + z * z
    if n < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / n
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy],
                     [xy + wz, 1.0 - (xx + zz), yz - wx],
                     [xz - wy, yz + wx, 1.0 - (xx + yy)]])


def mat2euler(mat):
    """(roll, pitch, yaw) of a rotation matrix, static-frame x-y-z convention ('sxyz')."""
    m = np.asarray(mat, dtype=np.float64)[:3, :3]
    cy = math.sqrt(m[0, 0] * m[0, 0] + m[1, 0] * m[1, 0])
    if cy > _EPS4:
        ax = math.atan2(m[2, 1], m[2, 2])
        ay = math.atan2(-m[2, 0], cy)
        az = math.atan2(m[1, 0], m[0, 0])
    else:                                   # gimbal lock: yaw is folded into roll
        ax = math.atan2(-m[1, 2], m[1, 1])
        ay = math.atan2(-m[2, 0], cy)
        az = 0.0
    return ax, ay, az


def euler2quat(ai, aj, ak):
    """Quaternion (w, x, y, z) of static-frame x-y-z Euler angles."""
    ai, aj, ak = ai / 2.0, aj / 2.0, ak / 2.0
    ci, si, cj, sj, ck, sk = math.cos(ai), math.sin(ai), math.cos(aj), math.sin(aj), math.cos(ak), math.sin(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([cj * cc + sj * ss, cj * sc - sj * cs, cj * ss + sj * cc, cj * cs - sj * sc])


def quat2axangle(quat, identity_thresh=None):
    """(unit axis, angle) of a quaternion; the identity maps to axis (1, 0, 0), angle 0."""
    w, x, y, z = quat
    n = w * w + x * x + y * y + z * z
    if not np.isfinite(n):
        return np.array([1.0, 0, 0]), float('nan')
    if identity_thresh is None:
        identity_thresh = np.finfo(np.float64).eps * 3
    if n < np.finfo(np.float64).eps ** 2:
        return np.array([1.0, 0, 0]), 0.0
    if n != 1:
        s = math.sqrt(n)
        w, x, y, z = w / s, x / s, y / s, z / s
    len2 = x * x + y * y + z * z
    if len2 < identity_thresh ** 2:
        return np.array([1.0, 0, 0]), 0.0
    theta = 2 * math.acos(max(min(w, 1), -1))
    return np.array([x, y, z]) / math.sqrt(len2), theta


def euler2axangle(ai, aj, ak):
    return quat2axangle(euler2quat(ai, aj, ak))


class BridgeSimplerAdapter:
    """Proprio pre-processing and action post-processing of the Bridge / WidowX evaluation (no simulator, image or tokenizer
    plumbing here: prompts and pixels go through vlaser_amd.prep)."""

    default_rot = np.array([[0, 0, 1.0], [0, 1.0, 0], [-1.0, 0, 0]])      # Bridge EE poses are relative to a top-down pose

    def __init__(self, dataset_statistics, action_normalization_type='bound', proprio_normalization_type='bound'):
        self.dataset_statistics = dataset_statistics
        self.action_normalization_type = action_normalization_type
        self.proprio_normalization_type = proprio_normalization_type

    # ---- BaseEnvAdapter
    @staticmethod
    def normalize_bound(data, data_min, data_max, clip_min=-1, clip_max=1, eps=1e-8):
        return np.clip(2 * (data - data_min) / (data_max - data_min + eps) - 1, clip_min, clip_max)

    @staticmethod
    def denormalize_bound(data, data_min, data_max, clip_min=-1, clip_max=1, eps=1e-8):
        return (data - clip_min) / (clip_max - clip_min) * (data_max - data_min) + data_min

    @staticmethod
    def normalize_gaussian(data, mean, std, eps=1e-8):
        return (data - mean) / (std + eps)

    @staticmethod
    def denormalize_gaussian(data, mean, std, eps=1e-8):
        return data * (std + eps) + mean

    # ---- observation -> model proprio [7] = xyz, rpy in the top-down frame, gripper openness
    def preprocess_proprio(self, obs):
        p = np.asarray(obs['agent']['eef_pos'], dtype=np.float64)
        rpy = mat2euler(quat2mat(p[3:7]) @ self.default_rot.T)
        return np.concatenate([p[:3], rpy, [p[7]]])

    def normalize_proprio(self, raw):
        st = self.dataset_statistics['proprio']
        if self.proprio_normalization_type == 'bound':
            return self.normalize_bound(raw, np.array(st['p01']), np.array(st['p99']), clip_min=-1, clip_max=1)
        return self.normalize_gaussian(raw, np.array(st['mean']), np.array(st['std']))

    # ---- model chunk [n, 7] -> env actions [n, 7] = xyz delta, axis*angle rotation, binarised gripper
    @staticmethod
    def postprocess_gripper(action):
        return 2.0 * (action > 0.5) - 1.0

    def postprocess(self, actions):
        actions = np.asarray(actions, dtype=np.float64)
        st = self.dataset_statistics['action']
        if self.action_normalization_type == 'bound':       # the gripper channel is not normalised in the training data
            raw = self.denormalize_bound(actions[:, :-1], np.array(st['p01'])[:-1], np.array(st['p99'])[:-1], clip_min=-1, clip_max=1)
        else:
            raw = self.denormalize_gaussian(actions[:, :-1], np.array(st['mean'])[:-1], np.array(st['std'])[:-1])
        out = np.zeros((len(actions), 7))
        for i, (r, grip) in enumerate(zip(raw, actions[:, -1])):
            axis, angle = euler2axangle(r[3], r[4], r[5])
            out[i] = np.concatenate([r[:3], axis * angle, [self.postprocess_gripper(grip)]])
        return out
