"""CPU known-answer tests of matryodshka_amd.poses.interpolate_pose (reference utils.py:55-74)."""
import numpy as np

from matryodshka_amd.poses import interpolate_pose


def _rot(axis, ang):
    axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    k = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * k + (1 - np.cos(ang)) * (k @ k)


def _pose(r, t):
    p = np.eye(4)
    p[:3, :3], p[:3, 3] = r, t
    return p[None]


def test_halfway_rotation_and_mean_translation():
    axis = [0.3, -0.5, 0.8]
    a, b = _pose(_rot(axis, 0.2), [0.0, 0.1, 0.2]), _pose(_rot(axis, 1.0), [-0.064, 0.3, 0.0])
    m = interpolate_pose(a, b)[0].astype(np.float64)
    assert np.allclose(m[:3, :3], _rot(axis, 0.6), atol=1e-6)          # same axis: the angle is the mean
    assert np.allclose(m[:3, 3], [-0.032, 0.2, 0.1], atol=1e-7) and np.allclose(m[3], [0, 0, 0, 1])
    assert np.allclose(interpolate_pose(b, a)[0], interpolate_pose(a, b)[0], atol=1e-6)   # symmetric at t = 0.5


def test_identity_large_angle_and_orthonormality():
    eye = np.eye(4, dtype=np.float32)[None]
    assert np.array_equal(interpolate_pose(eye, eye), eye)
    # data_loader.py:215-216 (PP): pure x-translation of the source camera
    src = eye.copy(); src[0, 0, 3] = -0.064
    assert np.allclose(interpolate_pose(eye, src)[0, :3, 3], [-0.032, 0, 0]) and np.allclose(interpolate_pose(eye, src)[0, :3, :3], np.eye(3))
    rng = np.random.RandomState(0)
    for _ in range(20):
        r0, r1 = _rot(rng.normal(size=3), rng.uniform(-3.1, 3.1)), _rot(rng.normal(size=3), rng.uniform(-3.1, 3.1))
        m = interpolate_pose(_pose(r0, [0, 0, 0]), _pose(r1, [0, 0, 0]))[0, :3, :3].astype(np.float64)
        assert np.allclose(m @ m.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(m) - 1) < 1e-6
        # half-way on the shortest arc: the relative rotations ref->mid and mid->src are equal
        assert np.allclose(r0.T @ m, m.T @ r1, atol=1e-5)


def test_random_rotation_is_a_rigid_jitter_pose():
    from matryodshka_amd import poses
    m = poses.random_rotation(1.0, 1.0, np.random.RandomState(3))
    assert m.shape == (1, 4, 4) and m.dtype == np.float32
    r = m[0, :3, :3].astype(np.float64)
    assert np.allclose(r @ r.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(r) - 1.0) < 1e-6
    assert np.all(np.abs(m[0, :3, 3]) <= 0.01) and np.array_equal(m[0, 3], [0, 0, 0, 1])
    assert np.array_equal(poses.random_rotation(0.0, 0.0, np.random.RandomState(3))[0], np.eye(4, dtype=np.float32))
    # y-only Euler angle = the crop rotation of projector.py:78-86 ([[c,0,s],[0,1,0],[-s,0,c]])
    ry = poses.rotation_from_euler([0.0, 0.3, 0.0])
    assert np.allclose(ry, [[np.cos(0.3), 0, np.sin(0.3)], [0, 1, 0], [-np.sin(0.3), 0, np.cos(0.3)]])
