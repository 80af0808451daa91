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
