"""Command line front end and PNG output (main.rs:38-107, 166-197; progress.rs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLI = os.path.join(ROOT, "rust_raytracer_amd", "rust-raytracer")
ASSETS = os.path.join(ROOT, "tests", "golden", "assets")


def run(*args):
    r = subprocess.run([CLI, *args], capture_output=True, timeout=300)
    r.stdout, r.stderr = r.stdout.decode(), r.stderr.decode()  # no newline translation: the progress bar redraws with \r
    return r


def test_cli_requires_both_positionals():
    r = run()
    assert r.returncode == 2 and "IN_SCENE_FILE" in r.stderr and "OUT_FILEPATH" in r.stderr
    r = run(os.path.join(ASSETS, "Cornell_Box.json"))
    assert r.returncode == 2


def test_cli_thread_count_validation(tmp_path):
    # main.rs:68-75
    r = run("-t", "0", os.path.join(ASSETS, "Cornell_Box.json"), str(tmp_path / "o.png"))
    assert r.returncode != 0 and "Need a thread count greater than zero" in r.stderr
    r = run("--thread-count", "many", os.path.join(ASSETS, "Cornell_Box.json"), str(tmp_path / "o.png"))
    assert r.returncode != 0 and "thread-count requires a valid positive integer" in r.stderr
    assert not (tmp_path / "o.png").exists()


def test_cli_missing_scene_file(tmp_path):
    r = run(str(tmp_path / "nope.json"), str(tmp_path / "o.png"))
    assert r.returncode != 0 and "Failed to read scene spec file." in r.stderr
    assert not (tmp_path / "o.png").exists()


def test_cli_never_overwrites_an_existing_file(tmp_path):
    # main.rs:98-106: create_new(true), checked before any tracing
    out = tmp_path / "taken.png"
    out.write_bytes(b"precious")
    r = run(os.path.join(ASSETS, "Cornell_Box.json"), str(out))
    assert r.returncode != 0 and "Failed to create new file" in r.stderr
    assert out.read_bytes() == b"precious"


def test_png_writer_round_trip(tmp_path):
    import rust_raytracer_amd as rrt
    from PIL import Image

    rng = np.random.default_rng(5)
    for shape in [(1, 1, 3), (7, 13, 3), (64, 48, 3)]:
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        path = tmp_path / ("img_%dx%d.png" % (shape[1], shape[0]))
        rrt.write_png(str(path), img)
        back = np.asarray(Image.open(path).convert("RGB"))
        assert back.shape == img.shape and (back == img).all()
        with pytest.raises(rrt.RrtError, match="Failed to create new file"):
            rrt.write_png(str(path), img)
        rrt.write_png(str(path), img[::-1].copy(), create_new=False)
        assert (np.asarray(Image.open(path).convert("RGB")) == img[::-1]).all()


@pytest.mark.gpu
def test_cli_renders_the_same_png_as_the_library(gpu, tmp_path):
    import rust_raytracer_amd as rrt
    from PIL import Image

    scene = os.path.join(ASSETS, "Cornell_Box_With_Sphere.json")
    out = tmp_path / "cornell.png"
    r = run("--width", "96", "--height", "80", "--samples", "200", "--seed", "3", scene, str(out))
    assert r.returncode == 0, r.stderr
    assert "\rProgress: <" in r.stdout and "(100%)" in r.stdout and "Success! Took " in r.stdout
    sc = rrt.Scene.load(scene, width=96, height=80, samples=200)
    sums, _ = sc.render(seed=3)
    want = rrt.resolve_rgb8(sums, 200)
    got = np.asarray(Image.open(out).convert("RGB"))
    assert got.shape == (80, 96, 3) and (got == want).all()


@pytest.mark.gpu
def test_accumulate_flag_continues_the_sample_sums_bit_exactly(gpu):
    import rust_raytracer_amd as rrt

    sc = rrt.Scene.load(os.path.join(ASSETS, "Cornell_Box.json"), width=64, height=48, samples=9)
    whole, _ = sc.render(seed=11)
    part, _ = sc.render(samples=4, first_sample=0, seed=11)
    part, _ = sc.render(samples=5, first_sample=4, seed=11, out=part, accumulate=True)
    assert part.tobytes() == whole.tobytes()
    fresh, _ = sc.render(samples=5, first_sample=4, seed=11, out=part.copy())  # without the flag the call starts from zero
    assert fresh.tobytes() != whole.tobytes()


@pytest.mark.gpu
def test_cli_gpus_and_stats(gpu, tmp_path):
    """--gpus N shares the frame between GPUs (rrt_render_multi); --stats prints the phases the reference's one number
    (main.rs:64, 191) lumps together.  With tiles the picture is the single-GPU picture whatever N is."""
    import rust_raytracer_amd as rrt
    from PIL import Image

    scene = os.path.join(ASSETS, "Cornell_Box.json")
    n = min(rrt.device_count(), 2)
    out = tmp_path / "a.png"
    r = run("--width", "64", "--height", "64", "--samples", "130", "--seed", "5", "--gpus", str(n), "--shard", "tiles", "--stats", scene, str(out))
    assert r.returncode == 0, r.stderr
    for word in ("load ", "bvh + flatten ", "upload ", "render ", "reduce ", "resolve ", "png ", "Mrays/s"):
        assert word in r.stdout, word
    sc = rrt.Scene.load(scene, width=64, height=64, samples=130)
    sums, _ = sc.render(seed=5)
    assert (np.asarray(Image.open(out).convert("RGB")) == rrt.resolve_rgb8(sums, 130)).all()
    r = run("--gpus", "99", scene, str(tmp_path / "b.png"))
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and not (tmp_path / "b.png").exists()
