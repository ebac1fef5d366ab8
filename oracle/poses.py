"""Oracle (TEST INFRASTRUCTURE ONLY, parity unpinned -- see oracle/__init__.py):
restatement of matryodshka/utils.py:55-74 `interpolate_pose`, the pose half-way between the reference and
the source camera at which train.py:118-121 builds the perspective plane-sweep volume (input_type PP,
BASELINE configs[4]).

The reference calls three functions of a THIRD-PARTY dependency that is not vendored under /root/reference:
tensorflow-graphics 1.0.0 (pinned in matryodshka-gpu.yml:44).  Their published algorithms are restated here
[TF-knowledge: tfg 1.0.0 sources, not executable in this container]:
  * `quaternion.from_rotation_matrix` (utils.py:60-61): quaternions are (x, y, z, w); four cases selected by the
    trace and the largest diagonal entry (the numerically safe form of Shepperd's method);
  * `slerp.interpolate(q0, q1, 0.5)` (utils.py:63), method QUATERNION: the dot product is taken to the shortest
    arc (q1 -> -q1 when it is negative), theta = acos(dot), weights sin((1 - t) theta) / sin(theta) and
    sin(t theta) / sin(theta), falling back to (1 - t, t) when sin(theta) vanishes;
  * `rotation_matrix_3d.from_quaternion` (utils.py:64): the usual 1 - 2 (y^2 + z^2) ... form.
Translation: 0.5 ref_t + 0.5 src_t (utils.py:67-69); last row: the reference pose's (utils.py:74).

Vectorised over the batch, float64 throughout, returned as float32 -- written independently of
matryodshka_amd/poses.py (which loops per sample and normalises differently); tests/test_poses.py and the
configs[4] fixture compare the two to 1e-6.
"""
import numpy as np


def quaternion_from_rotation_matrix(rot):
    """[B,3,3] -> [B,4] unit quaternions (x, y, z, w)."""
    m = np.asarray(rot, dtype=np.float64).reshape(-1, 3, 3)
    m00, m01, m02 = m[:, 0, 0], m[:, 0, 1], m[:, 0, 2]
    m10, m11, m12 = m[:, 1, 0], m[:, 1, 1], m[:, 1, 2]
    m20, m21, m22 = m[:, 2, 0], m[:, 2, 1], m[:, 2, 2]
    trace = m00 + m11 + m22
    eps = 1e-300   # (keeps the unused branches finite; np.where picks one per sample)

    def case_trace():
        sq = np.sqrt(np.maximum(trace + 1.0, eps)) * 2.0
        return np.stack([(m21 - m12) / sq, (m02 - m20) / sq, (m10 - m01) / sq, 0.25 * sq], axis=1)

    def case_x():
        sq = np.sqrt(np.maximum(1.0 + m00 - m11 - m22, eps)) * 2.0
        return np.stack([0.25 * sq, (m01 + m10) / sq, (m02 + m20) / sq, (m21 - m12) / sq], axis=1)

    def case_y():
        sq = np.sqrt(np.maximum(1.0 + m11 - m00 - m22, eps)) * 2.0
        return np.stack([(m01 + m10) / sq, 0.25 * sq, (m12 + m21) / sq, (m02 - m20) / sq], axis=1)

    def case_z():
        sq = np.sqrt(np.maximum(1.0 + m22 - m00 - m11, eps)) * 2.0
        return np.stack([(m02 + m20) / sq, (m12 + m21) / sq, 0.25 * sq, (m10 - m01) / sq], axis=1)

    cond_x = ((m00 > m11) & (m00 > m22))[:, None]
    cond_y = (m11 > m22)[:, None]
    q = np.where((trace > 0.0)[:, None], case_trace(), np.where(cond_x, case_x(), np.where(cond_y, case_y(), case_z())))
    return q / np.sqrt((q * q).sum(axis=1, keepdims=True))


def slerp(q0, q1, percent):
    """Spherical interpolation on the shortest arc, [B,4] x [B,4] -> [B,4] (normalised)."""
    dot = (q0 * q1).sum(axis=1, keepdims=True)
    q1 = np.where(dot < 0.0, -q1, q1)
    dot = np.clip(np.abs(dot), -1.0, 1.0)
    theta = np.arccos(dot)
    sin_theta = np.sin(theta)
    safe = np.where(sin_theta > 1e-9, sin_theta, 1.0)
    w0 = np.where(sin_theta > 1e-9, np.sin((1.0 - percent) * theta) / safe, 1.0 - percent)
    w1 = np.where(sin_theta > 1e-9, np.sin(percent * theta) / safe, percent)
    q = w0 * q0 + w1 * q1
    return q / np.sqrt((q * q).sum(axis=1, keepdims=True))


def rotation_matrix_from_quaternion(q):
    """[B,4] (x, y, z, w) -> [B,3,3]."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    rows = [np.stack([1.0 - (tyy + tzz), txy - twz, txz + twy], axis=1),
            np.stack([txy + twz, 1.0 - (txx + tzz), tyz - twx], axis=1),
            np.stack([txz - twy, tyz + twx, 1.0 - (txx + tyy)], axis=1)]
    return np.stack(rows, axis=1)


def interpolate_pose(ref_pose, src_pose):
    """utils.py:55-74.  [B,4,4] x [B,4,4] -> [B,4,4] float32."""
    ref = np.asarray(ref_pose, dtype=np.float64).reshape(-1, 4, 4)
    src = np.asarray(src_pose, dtype=np.float64).reshape(-1, 4, 4)
    out_quat = slerp(quaternion_from_rotation_matrix(ref[:, :3, :3]), quaternion_from_rotation_matrix(src[:, :3, :3]), 0.5)
    out_rot = rotation_matrix_from_quaternion(out_quat)
    out_t = 0.5 * ref[:, :3, 3:] + 0.5 * src[:, :3, 3:]
    combined = np.concatenate([out_rot, out_t], axis=2)
    return np.concatenate([combined, ref[:, 3:, :]], axis=1).astype(np.float32)
