"""Fixed ray batches for intersection parity: camera rays, random interior rays, rays from surfaces,
axis-parallel rays (zero direction components), and rays aimed at primitive vertices and edges."""
import numpy as np


def ray_batch(oracle_scene, W, H, n_random=4096, seed=0, max_corners=4000):
    rng = np.random.default_rng(seed)
    bb = oracle_scene.shape_bboxes()
    lo = bb[:, :3].min(axis=0)
    hi = bb[:, 3:].max(axis=0)
    span = np.maximum(hi - lo, 1e-3)
    # scenes with one huge ground sphere: also sample near the camera
    cam = oracle_scene.camera()
    origins, dirs = [], []
    # 1. camera rays
    n = n_random
    px = rng.integers(0, W, n)
    py = rng.integers(0, H, n)
    sm = rng.integers(0, 64, n)
    o, d = oracle_scene.camera_rays(W, H, seed, px, py, sm)
    origins.append(o)
    dirs.append(d)
    # 2. random rays inside the bounds (clipped to a region around the camera for huge scenes)
    c = cam[:3]
    rad = min(float(np.linalg.norm(span)), 50.0 * float(np.linalg.norm(cam[6:9])) + 10.0)
    o = (c + rng.uniform(-1, 1, (n, 3)) * rad).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    origins.append(o)
    dirs.append(d)
    # 3. secondary rays: start on surfaces found by batch 1
    p, t = oracle_scene.intersect(origins[0], dirs[0])
    hit = p >= 0
    so = (origins[0][hit] + dirs[0][hit] * t[hit, None]).astype(np.float32)
    sd = rng.normal(size=so.shape).astype(np.float32)
    origins.append(so)
    dirs.append(sd)
    # 4. axis-parallel rays (one or two zero direction components, +0 and -0)
    k = 512
    o = (c + rng.uniform(-1, 1, (k, 3)) * rad).astype(np.float32)
    d = np.zeros((k, 3), dtype=np.float32)
    ax = rng.integers(0, 3, k)
    d[np.arange(k), ax] = rng.choice([-1.0, 1.0], k)
    d[np.arange(k), (ax + 1) % 3] = rng.choice([0.0, -0.0, 0.5], k)
    origins.append(o)
    dirs.append(d)
    # 5. rays aimed exactly at box corners / centres of every primitive (vertex- and edge-grazing)
    corners = []
    if len(bb) > 100000:  # huge meshes: aim at a sample of the primitives
        bb = bb[rng.choice(len(bb), max_corners, replace=False)]
    for b in bb:
        for cx in (b[0], b[3]):
            for cy in (b[1], b[4]):
                for cz in (b[2], b[5]):
                    corners.append([cx, cy, cz])
        corners.append(0.5 * (b[:3] + b[3:]))
    corners = np.asarray(corners, dtype=np.float32)
    if len(corners) > max_corners:
        corners = corners[rng.choice(len(corners), max_corners, replace=False)]
    o = np.repeat(c[None, :], len(corners), axis=0).astype(np.float32)
    d = (corners - o).astype(np.float32)
    origins.append(o)
    dirs.append(d)
    o2 = (c + rng.uniform(-1, 1, (len(corners), 3)) * rad).astype(np.float32)
    origins.append(o2)
    dirs.append((corners - o2).astype(np.float32))
    ro = np.ascontiguousarray(np.concatenate(origins), dtype=np.float32)
    rd = np.ascontiguousarray(np.concatenate(dirs), dtype=np.float32)
    return ro, rd


GRAZER_TILTS = (0.0, 1e-7, 1e-6, 1e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 0.1, 1.0)


def triangle_outline_grazers(slots, extent, seed=0, max_triangles=48, per_edge=6, tilts=GRAZER_TILTS):
    """Rays built to fall between the exact triangle test and a padded box around the triangle: each passes a point on (or a
    few 1e-7..1e-4 of the scene's extent beside) a triangle's outline - vertices included - in a direction that leaves the
    triangle's plane by an angle between 0 and 1 rad, from either side.  Near the plane the exact test's own rounding accepts
    points slightly outside the triangle, which is where a box shortcut could lose a hit the exact test alone would find.

    slots: the structured array HostScene.slots() returns; extent: the scene's largest coordinate.
    Returns (origins, directions, slot of the triangle aimed at, angle off its plane in rad) per ray."""
    rng = np.random.default_rng(seed)
    tri = np.nonzero((slots["info"] >> 30) == 0)[0]
    if len(tri) > max_triangles:
        tri = np.sort(rng.choice(tri, max_triangles, replace=False))
    offs = np.array([0.0, 1e-7, -1e-7, 1e-6, -1e-6, 3e-6, -3e-6, 1e-5, -1e-5, 3e-5, -3e-5, 1e-4], dtype=np.float64) * extent
    tilts = np.asarray(tilts, dtype=np.float64)
    origins, dirs, which, angle = [], [], [], []
    for s in tri:
        v0 = slots["v0"][s].astype(np.float64)
        e1 = slots["e1"][s].astype(np.float64)
        e2 = slots["e2"][s].astype(np.float64)
        n = np.cross(e1, e2)
        if not np.linalg.norm(n) > 0.0:
            continue
        n /= np.linalg.norm(n)
        centre = v0 + (e1 + e2) / 3.0
        for start, along in ((v0, e1), (v0, e2), (v0 + e1, e2 - e1)):
            a = np.concatenate([[0.0, 1.0], rng.uniform(0.0, 1.0, per_edge - 2)])
            unit = along / np.linalg.norm(along)
            out = np.cross(unit, n)
            if np.dot(out, start + 0.5 * along - centre) < 0.0:
                out = -out
            pts = start[None, :] + a[:, None] * along[None, :]
            A, D, T = np.meshgrid(np.arange(len(a)), np.arange(len(offs)), np.arange(len(tilts)), indexing="ij")
            A, D, T = A.ravel(), D.ravel(), T.ravel()
            k = len(A)
            target = pts[A] + offs[D][:, None] * out[None, :]
            phi = rng.uniform(0.0, 2.0 * np.pi, k)
            phi = np.where(rng.uniform(0.0, 1.0, k) < 0.3, rng.choice([0.0, np.pi], k), phi)  # along the edge: past the triangle, never across it
            w = np.cos(phi)[:, None] * unit[None, :] + np.sin(phi)[:, None] * out[None, :]  # in the plane
            side = rng.choice([-1.0, 1.0], k)
            d = np.cos(tilts[T])[:, None] * w + (side * np.sin(tilts[T]))[:, None] * n[None, :]
            d *= rng.choice([1.0, 0.37, 2.9], k)[:, None]  # the reference's rays are not all unit length
            o = target - rng.uniform(0.01, 1.0, k)[:, None] * extent * d
            origins.append(o)
            dirs.append(d)
            which.append(np.full(k, s, dtype=np.int32))
            angle.append(tilts[T])
    return (np.ascontiguousarray(np.concatenate(origins), dtype=np.float32), np.ascontiguousarray(np.concatenate(dirs), dtype=np.float32),
            np.concatenate(which), np.concatenate(angle))
