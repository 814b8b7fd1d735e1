"""TEST INFRASTRUCTURE -- CPU restatement (numpy, sequential fp32) of the reference's post-sampling transform, the
oracle for csrc/motion_recover.h.  Follows, line by line:
  * inv_transform                data_loaders/humanml/data/dataset.py:132-133      data * std + mean
  * recover_root_rot_pos         data_loaders/humanml/scripts/motion_process.py:366-385
  * recover_from_ric             data_loaders/humanml/scripts/motion_process.py:437-452
  * qinv / qrot                  data_loaders/humanml/common/quaternion.py:16-20, :56-75
  * the reshape/permute of       sample/generate.py:163-166
Pinned against the reference itself: oracle/make_golden.py writes tests/golden/recover_B3_T196.npz from the upstream
functions and records the max-abs difference to this restatement in PIN_REPORT.json."""
import numpy as np


def _qrot(q, v):
    """quaternion.py:56-75 (q [...,4] as (w, x, y, z), v [...,3])."""
    qvec = q[..., 1:]
    uv = np.cross(qvec, v)
    uuv = np.cross(qvec, uv)
    return v + 2 * (q[..., :1] * uv + uuv)


def _qinv(q):
    return q * np.array([1, -1, -1, -1], dtype=q.dtype)


def recover_from_ric(sample, mean, std, joints_num):
    """sample [B, JF, 1, T] normalised -> [B, joints_num, 3, T] (float32 throughout, cumsum in time order)."""
    data = sample.astype(np.float32).transpose(0, 2, 3, 1) * std.astype(np.float32) + mean.astype(np.float32)  # [B,1,T,JF]
    rot_vel = data[..., 0]
    ang = np.zeros_like(rot_vel)
    ang[..., 1:] = rot_vel[..., :-1]
    ang = np.cumsum(ang, axis=-1, dtype=np.float32)
    q = np.zeros(data.shape[:-1] + (4,), np.float32)
    q[..., 0] = np.cos(ang)
    q[..., 2] = np.sin(ang)
    r_pos = np.zeros(data.shape[:-1] + (3,), np.float32)
    r_pos[..., 1:, [0, 2]] = data[..., :-1, 1:3]
    r_pos = _qrot(_qinv(q), r_pos).astype(np.float32)
    r_pos = np.cumsum(r_pos, axis=-2, dtype=np.float32)
    r_pos[..., 1] = data[..., 3]
    pos = data[..., 4:(joints_num - 1) * 3 + 4]
    pos = pos.reshape(pos.shape[:-1] + (-1, 3))
    qi = np.broadcast_to(_qinv(q)[..., None, :], pos.shape[:-1] + (4,))
    pos = _qrot(qi, pos).astype(np.float32)
    pos[..., 0] += r_pos[..., 0:1]
    pos[..., 2] += r_pos[..., 2:3]
    pos = np.concatenate([r_pos[..., None, :], pos], axis=-2)          # [B,1,T,J,3]
    return pos.reshape((-1,) + pos.shape[2:]).transpose(0, 2, 3, 1)    # generate.py:166 -> [B,J,3,T]
