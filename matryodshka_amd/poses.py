"""Host-side pose helpers of the perspective (PP) path.

`interpolate_pose` is matryodshka/utils.py:55-74: the pose half-way between the reference and the source
camera -- rotation by quaternion slerp at 0.5 (tensorflow_graphics `quaternion.from_rotation_matrix` ->
`slerp.interpolate(.., 0.5)` -> `rotation_matrix_3d.from_quaternion`; shortest arc), translation by the
mean -- at which train.py:118-121 builds the plane-sweep volume: `interp_pose_inv = inverse(interp_pose)`
is what `MSI(input_type='PP').format_network_input(..., ref_pose_inv=interp_pose_inv)` takes, and the render
pose is `tgt_pose @ interp_pose_inv` (msi.py:644-646).  4x4 host math in float64, returned as float32.
"""
import numpy as np


def _quat_from_matrix(r):
    """Rotation matrix -> unit quaternion (x, y, z, w), numerically safe branch on the largest diagonal term."""
    t = np.trace(r)
    if t > 0.0:
        s = np.sqrt(t + 1.0) * 2.0
        q = np.array([(r[2, 1] - r[1, 2]) / s, (r[0, 2] - r[2, 0]) / s, (r[1, 0] - r[0, 1]) / s, 0.25 * s])
    elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
        s = np.sqrt(1.0 + r[0, 0] - r[1, 1] - r[2, 2]) * 2.0
        q = np.array([0.25 * s, (r[0, 1] + r[1, 0]) / s, (r[0, 2] + r[2, 0]) / s, (r[2, 1] - r[1, 2]) / s])
    elif r[1, 1] > r[2, 2]:
        s = np.sqrt(1.0 + r[1, 1] - r[0, 0] - r[2, 2]) * 2.0
        q = np.array([(r[0, 1] + r[1, 0]) / s, 0.25 * s, (r[1, 2] + r[2, 1]) / s, (r[0, 2] - r[2, 0]) / s])
    else:
        s = np.sqrt(1.0 + r[2, 2] - r[0, 0] - r[1, 1]) * 2.0
        q = np.array([(r[0, 2] + r[2, 0]) / s, (r[1, 2] + r[2, 1]) / s, 0.25 * s, (r[1, 0] - r[0, 1]) / s])
    return q / np.linalg.norm(q)


def _matrix_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _slerp(q0, q1, t):
    d = float(np.dot(q0, q1))
    if d < 0.0:                       # shortest arc
        q1, d = -q1, -d
    if d > 1.0 - 1e-12:               # (nearly) identical: linear interpolation is exact to rounding
        q = (1.0 - t) * q0 + t * q1
    else:
        th = np.arccos(d)
        q = (np.sin((1.0 - t) * th) * q0 + np.sin(t * th) * q1) / np.sin(th)
    return q / np.linalg.norm(q)


def interpolate_pose(ref_pose, src_pose):
    """[B,4,4] x [B,4,4] -> [B,4,4] float32 (utils.py:55-74); the last row is the reference pose's."""
    ref = np.asarray(ref_pose, dtype=np.float64).reshape(-1, 4, 4)
    src = np.asarray(src_pose, dtype=np.float64).reshape(-1, 4, 4)
    out = np.empty_like(ref)
    for b in range(ref.shape[0]):
        q = _slerp(_quat_from_matrix(ref[b, :3, :3]), _quat_from_matrix(src[b, :3, :3]), 0.5)
        out[b, :3, :3] = _matrix_from_quat(q)
        out[b, :3, 3] = 0.5 * ref[b, :3, 3] + 0.5 * src[b, :3, 3]
        out[b, 3, :] = ref[b, 3, :]
    return out.astype(np.float32)


def rotation_from_euler(angles):
    """tensorflow_graphics rotation_matrix_3d.from_euler [TF-knowledge, tfg 1.0.0]: angles (x, y, z) in radians,
    R = Rz(z) @ Ry(y) @ Rx(x) (the x rotation is applied first).  float64 in, float64 out."""
    ax, ay, az = (float(a) for a in angles)
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def random_rotation(rc, tc, rng, angle_range=(-0.03, 0.03), offset_range=(-0.01, 0.01)):
    """geometry/spherical.py:21-40 tf_random_rotation: the jitter pose of the transform-inverse test path
    (test.py:112): Euler angles uniform in angle_range * rc, translation uniform in offset_range * tc.
    rng: numpy RandomState (the reference draws from the TF graph seed).  Returns [1,4,4] float32."""
    ang = rng.uniform(angle_range[0] * rc, angle_range[1] * rc, size=3)
    tr = rng.uniform(offset_range[0] * tc, offset_range[1] * tc, size=3)
    m = np.eye(4)
    m[:3, :3] = rotation_from_euler(ang)
    m[:3, 3] = tr
    return m[None].astype(np.float32)
