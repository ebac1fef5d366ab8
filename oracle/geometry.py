"""Oracle (TEST INFRASTRUCTURE ONLY, parity unpinned -- see oracle/__init__.py):
numpy fp32 op-by-op restatement of the reference's spherical geometry, bilinear
wrap-around gather and over-composite.

Every TensorFlow elementwise op of the reference is one numpy fp32 op here, in
the association Python's operator precedence gives the reference expression, so
each intermediate rounds to fp32 exactly once (no FMA contraction).  Python
double constants (np.pi / width ...) are rounded to fp32 at the point TF would
convert them to a tensor constant.

Reference files followed (paths inside the reference checkout):
  geometry/spherical.py  :42-44, 54-68, 116-129, 170-246, 268-326
  geometry/projector.py  :34-62, 129-170, 209-211, 225-291
  geometry/sampling.py   :135-201

One deliberate definition (documented in DESIGN.md): cos/sin of the lat-long
grid are the CORRECTLY ROUNDED fp32 values of the fp32 grid angles (computed in
fp64, rounded once).  TF 1.14's Eigen sin/cos have unknown last-ulp behaviour
and TF cannot be run here, so the library-independent definition is used on
both sides (oracle and HIP path) to make every branch of project_ods
reproducible.
"""
import numpy as np

F = np.float32
PI = np.pi


def _f(x):
    return np.asarray(x, dtype=F)


# ----------------------------------------------------------------------------
# grids
# ----------------------------------------------------------------------------
def linspace_f32(start, stop, num):
    """tf.linspace in TF 1.14 [TF-knowledge]: step = (stop-start)/(num-1) in
    fp32, value[i] = start + step*i in fp32 (spherical.py:43-44)."""
    start = F(start)
    stop = F(stop)
    if num == 1:
        return np.array([start], dtype=F)
    step = F(F(stop - start) / F(num - 1))
    i = np.arange(num, dtype=F)
    return (start + step * i).astype(F)


def lat_long_axes(height, width):
    """The two 1-D axes of spherical.lat_long_grid (spherical.py:42-44):
    S varies along W, T along H (tf.meshgrid default indexing='xy')."""
    s = linspace_f32(-PI + PI / width, PI - PI / width, width)
    t = linspace_f32(-PI / 2.0 + PI / (2 * height), PI / 2.0 - PI / (2 * height), height)
    return s, t


def lat_long_grid(shape):
    """spherical.lat_long_grid (spherical.py:42-44). shape = (H, W). Returns S, T [H,W]."""
    s, t = lat_long_axes(shape[0], shape[1])
    S, T = np.meshgrid(s, t)
    return S.astype(F), T.astype(F)


def cos_f32(x):
    """Correctly rounded fp32 cosine of an fp32 angle (see module docstring)."""
    return np.cos(np.asarray(x, dtype=np.float64)).astype(F)


def sin_f32(x):
    return np.sin(np.asarray(x, dtype=np.float64)).astype(F)


def trig_tables(height, width):
    """cos S, sin S [W]; cos T, sin T [H] -- the separable factors every
    geometry function below uses."""
    s, t = lat_long_axes(height, width)
    return cos_f32(s), sin_f32(s), cos_f32(t), sin_f32(t)


def theta_phi_to_pixels(theta, phi, width, height):
    """spherical.theta_phi_to_pixels (spherical.py:54-68)."""
    u = theta + F(PI)
    u = u - F(PI / width)
    u = u / F(2 * PI - (2 * PI / width))
    u = u * F(width - 1)
    v = ((phi + F(0.5 * PI)) - F(0.5 * PI / height)) / F(PI - PI / height)
    v = v * F(height - 1)
    return np.stack([u, v], axis=-1).astype(F)


# ----------------------------------------------------------------------------
# sphere sweep (PSV construction)
# ----------------------------------------------------------------------------
def backproject_spherical(S, T, depth):
    """spherical.backproject_spherical (spherical.py:116-129).
    S,T [H,W]; depth [D] -> x,y,z [D,H,W]."""
    depth = _f(depth).reshape(-1, 1, 1)
    cosT = cos_f32(T)
    x = depth * (cos_f32(S) * cosT)[None]
    y = depth * sin_f32(T)[None]
    z = depth * (sin_f32(S) * cosT)[None]
    return x.astype(F), y.astype(F), z.astype(F)


def apply_pose(points, pose):
    """projector.apply_pose (projector.py:275-291): pose[4,4] @ [x,y,z,1].
    The 4-term dot products are summed left to right in fp32 (the Eigen matmul
    order is not observable from the reference; identity poses -- the only ones
    the ODS test path uses, data_loader.py:146-157 -- are exact either way)."""
    x, y, z = points
    p = _f(pose)
    one = F(1.0)

    def row(r):
        return ((p[r, 0] * x + p[r, 1] * y) + p[r, 2] * z) + p[r, 3] * one

    return row(0).astype(F), row(1).astype(F), row(2).astype(F)


def project_ods(points, order, baseline, width, height):
    """spherical.project_ods (spherical.py:170-233), tuple-input branch
    (y is NOT negated, :176-177).  `baseline` = intrinsics[0][0][0] (:181).
    Returns uv [D,H,W,2] and the masks (z_larger_x, valid) for branch tests."""
    x, y, z = points
    r = F(baseline)
    with np.errstate(all="ignore"):
        f = r * r - (x * x + z * z)
        z_larger_x = np.abs(z) > np.abs(x)
        px = np.where(z_larger_x, x, z)
        pz = np.where(z_larger_x, z, x)

        pz_square = pz * pz
        a = F(1) + (px * px) / pz_square
        b = ((F(-2) * f) * px) / pz_square
        c = f + (f * f) / pz_square
        disc = b * b - (F(4) * a) * c

        s = (F(-order) * np.sign(pz)) * np.sqrt(disc)
        s = np.where(z_larger_x, s, -s)

        dx = (-b + s) / (F(2) * a)
        dz = (f - px * dx) / pz

        dx_final = np.where(z_larger_x, -dx, -dz)
        dz_final = np.where(z_larger_x, -dz, -dx)
        dx = dx_final
        dz = dz_final
        dy = y

        theta = -np.arctan2(dz, dx)
        phi = np.arctan2(dy, np.sqrt(dx * dx + dz * dz))
        phi = np.where(np.isnan(phi), F(1), phi)
        half_pi = F(PI / 2)
        phi = np.where(phi <= half_pi, phi, half_pi)
        phi = np.where(phi >= -half_pi, phi, -half_pi)

        u = (((theta + F(PI)) - F(PI / width)) / F(2 * PI - 2 * PI / width)) * F(width - 1)
        v = (((phi + F(0.5 * PI)) - F(0.5 * PI / height)) / F(PI - PI / height)) * F(height - 1)

        valid = disc >= F(0)
        u = np.where(valid, u, F(1))
        v = np.where(valid, v, F(1))
    uv = np.stack([u, v], axis=-1).astype(F)
    return uv, z_larger_x, valid


def resample(image, pixels):
    """sampling.resample (sampling.py:135-197): bilinear gather with wrap-around
    in BOTH axes; weights from the unwrapped corners; sum in the order a,b,c,d.
    image [N,H,W,C]; pixels [N,Ht,Wt,2] (last dim = (x, y)) -> [N,Ht,Wt,C]."""
    image = _f(image)
    pixels = _f(pixels)
    n, ht, wt, _ = pixels.shape
    _, height, width, ch = image.shape
    x = pixels[..., 0].reshape(n, -1)
    y = pixels[..., 1].reshape(n, -1)

    x0 = np.floor(x).astype(np.int32)
    x1 = x0 + 1
    y0 = np.floor(y).astype(np.int32)
    y1 = y0 + 1

    diff_x0 = x - x0.astype(F)
    diff_y0 = y - y0.astype(F)
    diff_x1 = x1.astype(F) - x
    diff_y1 = y1.astype(F) - y

    x0 = np.mod(x0 + width, width)      # numpy mod on ints is floor-mod, as tf.mod
    y0 = np.mod(y0 + height, height)
    x1 = np.mod(x1 + width, width)
    y1 = np.mod(y1 + height, height)

    b = np.arange(n)[:, None]
    va = image[b, y0, x0]
    vb = image[b, y0, x1]
    vc = image[b, y1, x0]
    vd = image[b, y1, x1]

    area_a = (diff_y1 * diff_x1)[..., None]
    area_b = (diff_y1 * diff_x0)[..., None]
    area_c = (diff_y0 * diff_x1)[..., None]
    area_d = (diff_y0 * diff_x0)[..., None]

    res = ((area_a * va + area_b * vb) + area_c * vc) + area_d * vd
    return res.reshape(n, ht, wt, ch).astype(F)


def ods_sphere_sweep(image, order, depths, pose, intrinsics, return_masks=False):
    """projector.ods_sphere_sweep -> sweep_one (projector.py:209-211, 129-170).
    image [B,H,W,C]; depths list/array [D]; pose [B,4,4]; intrinsics [B,3,3]
    (baseline in [b,0,0]).  Returns [B,H,W,C*D], channel index = d*C + c.
    Each batch element uses its own pose / baseline (the reference slices
    element i, projector.py:145-151)."""
    image = _f(image)
    pose = _f(pose)
    intrinsics = _f(intrinsics)
    depths = _f(depths)
    batch, height, width, ch = image.shape
    num_planes = depths.shape[0]
    S, T = lat_long_grid((height, width))
    out = np.empty((batch, height, width, ch * num_planes), dtype=F)
    masks = []
    for i in range(batch):
        points = backproject_spherical(S, T, depths)
        points = apply_pose(points, pose[i])
        uv, zlx, valid = project_ods(points, order, intrinsics[i, 0, 0], width, height)
        masks.append((zlx, valid, uv))
        image_tiled = np.broadcast_to(image[i:i + 1], (num_planes, height, width, ch))
        res = resample(image_tiled, uv)                  # [D,H,W,C]
        res = np.transpose(res, (1, 2, 0, 3))             # [H,W,D,C]
        out[i] = res.reshape(height, width, ch * num_planes)
    if return_masks:
        return out, masks
    return out


# ----------------------------------------------------------------------------
# target-view reprojection + compositing
# ----------------------------------------------------------------------------
def intersect_sphere(pos, center, radius, width, height):
    """spherical.intersect_sphere (spherical.py:268-326) -> project_spherical
    (:235-246) -> theta_phi_to_pixels (:54-68).
    pos [4,4]; center [3] (note the x<->z swap, :286-288); radius [D].
    Returns pixel coords [D,H,W,2]."""
    pos = _f(pos)
    center = _f(center).reshape(-1)
    radius = _f(radius).reshape(-1, 1, 1)
    S, T = lat_long_grid((height, width))
    cosT = cos_f32(T)
    rx = (cos_f32(S) * cosT)[None]
    ry = sin_f32(T)[None]
    rz = (sin_f32(S) * cosT)[None]

    cx = center[2]
    cy = center[1]
    cz = center[0]

    rot = pos[:3, :3]

    def rrow(r):
        return (rot[r, 0] * rx + rot[r, 1] * ry) + rot[r, 2] * rz

    rx, ry, rz = rrow(0).astype(F), rrow(1).astype(F), rrow(2).astype(F)

    one = F(1)

    def prow(r):
        return F(F(F(pos[r, 0] * cx + pos[r, 1] * cy) + pos[r, 2] * cz) + pos[r, 3] * one)

    cx, cy, cz = prow(0), prow(1), prow(2)

    with np.errstate(all="ignore"):
        a = (rx * rx + ry * ry) + rz * rz
        b = F(2) * ((rx * cx + ry * cy) + rz * cz)
        c = F(F(F(cx * cx + cy * cy) + cz * cz)) - radius * radius
        disc = b * b - (F(4) * a) * c
        t = (-b + np.sqrt(disc)) / (F(2) * a)
        x = cx + t * rx
        y = cy + t * ry
        z = cz + t * rz
        theta = -np.arctan2(z, x)
        phi = np.arctan2(y, np.sqrt(x * x + z * z))
    return theta_phi_to_pixels(theta.astype(F), phi.astype(F), width, height)


def _transform_ray(r, c, pos):
    """spherical.transform_ray (spherical.py:70-94): rotate the direction by pos[:3,:3], move the
    origin by the full 4x4; dot products summed left to right in fp32."""
    rx, ry, rz = r
    cx, cy, cz = c
    pos = _f(pos)
    rot = pos[:3, :3]
    nr = [((rot[k, 0] * rx + rot[k, 1] * ry) + rot[k, 2] * rz).astype(F) for k in range(3)]
    one = F(1)
    nc = [(((pos[k, 0] * cx + pos[k, 1] * cy) + pos[k, 2] * cz) + pos[k, 3] * one).astype(F) for k in range(3)]
    return nr, nc


def _sphere_hit_pixels(r, c, radius, width, height):
    """spherical.get_sphere_intersections (:96-111) + project_spherical (:235-246)."""
    rx, ry, rz = r
    cx, cy, cz = c
    with np.errstate(all="ignore"):
        a = (rx * rx + ry * ry) + rz * rz
        b = F(2) * ((rx * cx + ry * cy) + rz * cz)
        cc = ((cx * cx + cy * cy) + cz * cz) - radius * radius
        disc = b * b - (F(4) * a) * cc
        t = (-b + np.sqrt(disc)) / (F(2) * a)
        x = cx + t * rx
        y = cy + t * ry
        z = cz + t * rz
        theta = -np.arctan2(z, x)
        phi = np.arctan2(y, np.sqrt(x * x + z * z))
    return theta_phi_to_pixels(theta.astype(F), phi.astype(F), width, height)


def intersect_ods(pose, order, baseline, radius, width, height):
    """spherical.intersect_ods (spherical.py:328-365): rays of the left (order=+1) / right (-1)
    ODS eye, tangent to the viewing circle of radius `baseline` (= intrinsics[0][0][0], :346)."""
    radius = _f(radius).reshape(-1, 1, 1)
    S, T = lat_long_grid((height, width))
    cosT = cos_f32(T)
    bl = F(baseline)
    od = F(order)
    rx = (cos_f32(S) * cosT)[None]
    ry = sin_f32(T)[None]
    rz = ((-sin_f32(S)) * cosT)[None]
    cx = (((-sin_f32(S)) * bl) * od)[None]
    cy = np.zeros_like(cx)
    cz = (((-cos_f32(S)) * bl) * od)[None]
    r, c = _transform_ray((rx, ry, rz), (cx, cy, cz), pose)
    return _sphere_hit_pixels(r, c, radius, width, height)


def uv_axes(height, width):
    """spherical.uv_grid (spherical.py:46-48)."""
    return (linspace_f32(-1. + 1. / width, 1. - 1. / width, width),
            linspace_f32(-1. + 1. / height, 1. - 1. / height, height))


def intersect_perspective(pos, center, radius, width, height, tgt_width, tgt_height):
    """spherical.intersect_perspective (spherical.py:367-401): hard-coded intrinsics
    (rx = S*0.1, ry = T*0.05, rz = -0.05), centre (c0, c1, -c2)."""
    radius = _f(radius).reshape(-1, 1, 1)
    center = _f(center).reshape(-1)
    s, t = uv_axes(tgt_height, tgt_width)
    S, T = np.meshgrid(s, t)
    rx = (S.astype(F) * F(0.1))[None]
    ry = (T.astype(F) * F(0.05))[None]
    rz = (-np.ones_like(S, dtype=F) * F(0.05))[None]
    cx, cy, cz = center[0], center[1], -center[2]
    r, c = _transform_ray((rx, ry, rz), (cx, cy, cz), pos)
    return _sphere_hit_pixels(r, c, radius, width, height)


def euler_y_pose(viewing_window):
    """The crop rotation of projector.py:78-86: tfgt.rotation_matrix_3d.from_euler([0, vw*pi/2, 0])
    = Ry(theta) [tensorflow-graphics 1.0.0, TF-knowledge], zero translation.  cos/sin are the
    correctly rounded fp32 values of the fp32 angle (as for the trig tables)."""
    ang = F(viewing_window * np.pi / 2.)
    c, s = cos_f32(ang), sin_f32(ang)
    m = np.eye(4, dtype=F)
    m[0, 0] = c; m[0, 2] = s; m[2, 0] = -s; m[2, 2] = c
    return m


def _warp_layers(src_images, coords_per_batch):
    coords = np.transpose(np.stack(coords_per_batch, axis=0), (1, 0, 2, 3, 4))   # [D,B,Ht,Wt,2]
    return np.stack([resample(src_images[i], coords[i]) for i in range(src_images.shape[0])], axis=0)


def projective_forward_ods(src_images, order, intrinsics, jitter_pose, depths):
    """projector.projective_forward_ods (projector.py:100-127).  Batch element i uses pose
    jitter_pose[i] and its own baseline (the reference reads intrinsics[0][0][0] for every
    element, spherical.py:346 -- identical for the B=1 it supports)."""
    src_images = _f(src_images)
    n_layers, n_batch, height, width, _ = src_images.shape
    depths = _f(depths)
    coords = [intersect_ods(jitter_pose[i], order, _f(intrinsics)[i, 0, 0], depths[:, i], width, height)
              for i in range(n_batch)]
    return _warp_layers(src_images, coords)


def projective_forward_sphere_to_perspective(src_images, tgt_pos, depths, viewing_window=3,
                                             tgt_height=320, tgt_width=640):
    """projector.projective_forward_sphere_to_perspective (projector.py:64-98); the caller's
    tgt_pose_rt is overwritten by the crop rotation there (:78-86)."""
    src_images = _f(src_images)
    n_layers, n_batch, height, width, _ = src_images.shape
    depths = _f(depths)
    tgt_pos = _f(tgt_pos).reshape(n_batch, 3)
    pose = euler_y_pose(viewing_window)
    coords = [intersect_perspective(pose, tgt_pos[i], depths[:, i], width, height, tgt_width, tgt_height)
              for i in range(n_batch)]
    return _warp_layers(src_images, coords)


def projective_forward_sphere(src_images, tgt_pose_rt, tgt_pos, depths):
    """projector.projective_forward_sphere (projector.py:34-62).
    src_images [D,B,H,W,C]; tgt_pose_rt [B,4,4]; tgt_pos [B,3]; depths [D,B].
    Returns warped layers [D,B,H,W,C]."""
    src_images = _f(src_images)
    n_layers, n_batch, height, width, ch = src_images.shape
    depths = _f(depths)
    tgt_pos = _f(tgt_pos).reshape(n_batch, 3)
    coords = []
    for i in range(n_batch):
        coords.append(intersect_sphere(tgt_pose_rt[i], tgt_pos[i], depths[:, i], width, height))
    coords = np.stack(coords, axis=0)                       # [B,D,H,W,2]
    coords = np.transpose(coords, (1, 0, 2, 3, 4))          # [D,B,H,W,2]
    out = np.empty_like(src_images)
    for i in range(n_layers):
        out[i] = resample(src_images[i], coords[i])
    return out


def over_composite(rgbas):
    """projector.over_composite (projector.py:246-265). List of [B,H,W,4], back to front."""
    output = None
    for i, rgba in enumerate(rgbas):
        rgb = rgba[..., 0:3]
        alpha = rgba[..., 3:]
        if i == 0:
            output = rgb
        else:
            rgb_by_alpha = rgb * alpha
            output = rgb_by_alpha + output * (F(1.0) - alpha)
    return output.astype(F)


def over_composite_depth(rgbas):
    """projector.over_composite_depth (projector.py:225-244).  `i / len(rgbas)`
    is Python true division (from __future__ import division, projector.py:23),
    converted to an fp32 constant."""
    n = len(rgbas)
    output = None
    for i, rgba in enumerate(rgbas):
        alpha = np.repeat(rgba[..., 3:], 3, axis=-1)
        if i == 0:
            output = np.zeros_like(alpha)
        else:
            output = F(i / n) * alpha + output * (F(1.0) - alpha)
    return output.astype(F)


# ----------------------------------------------------------------------------
# PP (perspective cube-face) path -- BASELINE configs[4]
# ----------------------------------------------------------------------------
def perspective_plane_sweep(image, depths, pose, intrinsics):
    """projector.perspective_plane_sweep (projector.py:221-223) = sweep_one with uv_grid
    (spherical.py:46-48), backproject_planar (:131-149), apply_pose, project_perspective (:248-266:
    matmul(intrinsics4x4, pose) applied to the ALREADY posed points -- the pose enters twice, as in
    the reference) and the wrap-around sampler.  image [B,H,W,C]; returns [B,H,W,C*D]."""
    image = _f(image)
    pose = _f(pose)
    intrinsics = _f(intrinsics)
    depths = _f(depths)
    batch, height, width, ch = image.shape
    nd = depths.shape[0]
    s, t = uv_axes(height, width)
    S, T = np.meshgrid(s, t)
    S = S.astype(F)[None]
    T = T.astype(F)[None]
    dep = depths.reshape(-1, 1, 1)
    out = np.empty((batch, height, width, ch * nd), dtype=F)
    one = F(1)
    for i in range(batch):
        K = intrinsics[i]
        fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        with np.errstate(all="ignore"):
            x = ((dep * S) * cx) / fx
            y = ((dep * T) * cy) / fy
            z = dep * np.ones_like(x)
            x, y, z = apply_pose((x.astype(F), y.astype(F), z.astype(F)), pose[i])
            P = pose[i]
            pr = []
            for r in range(3):
                m = [F(F(F(K[r, 0] * P[0, c] + K[r, 1] * P[1, c]) + K[r, 2] * P[2, c]) + F(0) * P[3, c]) for c in range(4)]
                pr.append(((m[0] * x + m[1] * y) + m[2] * z) + m[3] * one)
            u = pr[0] / pr[2]
            v = pr[1] / pr[2]
        uv = np.stack([u, v], axis=-1).astype(F)
        tiled = np.broadcast_to(image[i:i + 1], (nd, height, width, ch))
        res = np.transpose(resample(tiled, uv), (1, 2, 0, 3))
        out[i] = res.reshape(height, width, ch * nd)
    return out


def resampler_zero_pad(image, coords):
    """tf.contrib.resampler.resampler (sampling.py:51) [TF-knowledge]: bilinear with zero padding;
    zero outside (-1, W) x (-1, H); sum order fxfy + cxcy + fxcy + cxfy."""
    image = _f(image)
    coords = _f(coords)
    n, height, width, ch = image.shape
    x = coords[..., 0]
    y = coords[..., 1]
    with np.errstate(all="ignore"):
        inside = (x > F(-1)) & (y > F(-1)) & (x < F(width)) & (y < F(height))
        xs = np.where(inside, x, F(0))
        ys = np.where(inside, y, F(0))
        fx = np.floor(xs).astype(np.int64)
        fy = np.floor(ys).astype(np.int64)
        cx, cy = fx + 1, fy + 1
        dx = cx.astype(F) - xs
        dy = cy.astype(F) - ys
    b = np.arange(n).reshape(n, 1, 1)

    def get(ix, iy):
        ok = (ix >= 0) & (iy >= 0) & (ix < width) & (iy < height)
        v = image[b, np.clip(iy, 0, height - 1), np.clip(ix, 0, width - 1)]
        return np.where(ok[..., None], v, F(0))

    one = F(1)
    out = (((dx * dy)[..., None] * get(fx, fy) + ((one - dx) * (one - dy))[..., None] * get(cx, cy))
           + (dx * (one - dy))[..., None] * get(fx, cy)) + ((one - dx) * dy)[..., None] * get(cx, fy)
    return np.where(inside[..., None], out, F(0)).astype(F)


def _mm3(a, b):
    """3x3 fp32 matmul, dot products summed left to right."""
    a = _f(a)
    b = _f(b)
    out = np.empty((3, 3), dtype=F)
    for r in range(3):
        for c in range(3):
            out[r, c] = F(F(a[r, 0] * b[0, c] + a[r, 1] * b[1, c]) + a[r, 2] * b[2, c])
    return out


def inv_homography(k_s, k_t_inv, rot, t, a):
    """homography.inv_homography (homography.py:35-58) with n_hat = [0,0,1]
    (projector.py:365-366); k_t_inv is the hidden graph input `intrinsics_inv:0`."""
    rot_t = _f(rot).T.copy()
    t = _f(t).reshape(3)
    nrt = rot_t[2]                                            # n_hat @ rot_t
    den = F(a) - F(F(nrt[0] * t[0] + nrt[1] * t[1]) + nrt[2] * t[2])
    den = F(den + F(1e-8) * F(1.0 if den == 0 else 0.0))      # divide_safe (homography.py:30-33)
    q = np.array([F(F(rot_t[r, 0] * t[0] + rot_t[r, 1] * t[1]) + rot_t[r, 2] * t[2]) for r in range(3)], dtype=F)
    m1 = np.empty((3, 3), dtype=F)
    for r in range(3):
        for c in range(3):
            m1[r, c] = F(rot_t[r, c] + F(F(q[r] * rot_t[2, c]) / den))
    return _mm3(_mm3(k_s, m1), k_t_inv)


def projective_forward_homography(src_images, intrinsics, intrinsics_inv, pose, depths):
    """projector.projective_forward_homography (projector.py:343-373) -> homography.planar_transform
    (homography.py:96-157).  src_images [D,B,H,W,C]; returns the warped layers."""
    src_images = _f(src_images)
    n_layers, n_batch, height, width, _ = src_images.shape
    pose = _f(pose)
    depths = _f(depths)
    xs, ys = np.meshgrid(np.arange(width, dtype=F), np.arange(height, dtype=F))   # meshgrid_abs
    out = np.empty_like(src_images)
    one = F(1)
    for l in range(n_layers):
        coords = []
        for bb in range(n_batch):
            h = inv_homography(intrinsics[bb], intrinsics_inv[bb], pose[bb, :3, :3], pose[bb, :3, 3], -depths[l, bb])
            with np.errstate(all="ignore"):
                px = (xs * h[0, 0] + ys * h[0, 1]) + one * h[0, 2]
                py = (xs * h[1, 0] + ys * h[1, 1]) + one * h[1, 2]
                pw = (xs * h[2, 0] + ys * h[2, 1]) + one * h[2, 2]
                pw = pw + F(1e-8) * (pw == 0).astype(F)
                coords.append(np.stack([px / pw, py / pw], axis=-1).astype(F))
        out[l] = resampler_zero_pad(src_images[l], np.stack(coords, axis=0))
    return out
