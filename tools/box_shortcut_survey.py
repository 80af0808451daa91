"""Where the padded-box shortcuts of the traversal and the exhaustive mode part ways: rays along triangle outlines, by the angle
between ray and triangle plane.  Prints, per scene and angle, how many rays each mode answers differently from the CPU oracle,
and how many of the differing rays miss (float64) the grown box of the triangle the oracle reports.

    python tools/box_shortcut_survey.py > gpurun_out/box_survey.txt        (needs a GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import rays as raygen  # noqa: E402
import rust_raytracer_amd as rrt  # noqa: E402
import test_gpu_box_shortcuts as cases  # noqa: E402

TILTS = (0.0, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 1e-2, 0.1, 1.0)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def main():
    for case in cases.CASES:
        o, g = cases.scene_pair(case)
        slots = g.host.slots()
        bb = o.shape_bboxes()
        extent = float(np.abs(bb[np.isfinite(bb).all(axis=1)]).max())
        ro, rd, which, angle = raygen.triangle_outline_grazers(slots, extent, seed=5, tilts=TILTS)
        p_ref, t_ref = o.intersect(ro, rd)
        slot_of_prim = {int(pr): i for i, pr in enumerate(slots["prim"])}
        print("%s: %d rays, extent %.4g, pad %.3g" % (case, len(ro), extent, 1e-5 * extent))
        for mode, name in ((rrt.AABB_TIGHT, "tight"), (rrt.AABB_REFERENCE, "reference"), (rrt.AABB_EXHAUSTIVE, "exhaustive")):
            p, t = g.hit(ro, rd, aabb_mode=mode)
            bad = (p != p_ref) | (bits(t) != bits(t_ref))
            row = ["%g:%d" % (a, int((bad & (angle == a)).sum())) for a in TILTS]
            print("  %-10s %4d differ | by angle %s" % (name, int(bad.sum()), " ".join(row)))
            if mode == rrt.AABB_TIGHT:
                i = np.nonzero(bad)[0]
                s = np.array([slot_of_prim.get(int(pr), -1) for pr in p_ref[i]])
                print("      of these, rays that (float64) do not touch the box, grown by 0.99 pads, of the triangle the oracle reports: %d" % (
                    int(cases.misses_grown_box(slots, s, ro[i], rd[i], 0.99e-5 * extent).sum()) if len(i) and (s >= 0).all() else 0))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
