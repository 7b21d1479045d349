"""Host-only check of the hierarchy builder (ray_tracing_amd/csrc/bvh_build.h): the binary SAH tree over the
members of a BVHNode entry (aabb.rs:63-100 builds the reference's; any conservative tree returns the same closest
hit) and its collapse into the four-wide nodes the kernels walk. tests/native/bvh4_check.cpp holds the checks
(containment, every object in exactly one leaf, packed children, exact stack need); this test builds it with g++
and runs it — no GPU, no library."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_four_wide_hierarchy_structure(tmp_path):
    exe = tmp_path / "bvh4_check"
    build = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "bvh4_check.cpp"), "-o", str(exe)],
                           capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stdout[-3000:] + run.stderr[-1000:]
    # the 485-sphere scene of scene.rs:46-52 is the size of the "cloud 485" case: its walks fit a 16-entry stack
    line = [l for l in run.stdout.splitlines() if l.startswith("cloud 485")][0]
    assert int(line.split()[-1]) <= 24


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_scene_mirror_under_sanitizers(tmp_path):
    """The C++ mirror of the reference's scene builders (scene.rs:19-275) and its flattening, built with
    AddressSanitizer and UBSan on the host: every scene, the scaled random scene, the PNG writer
    (tests/native/host_scenes_check.cpp checks the index consistency of the flat tables)."""
    exe = tmp_path / "host_scenes_check"
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            os.path.join(ROOT, "tests", "native", "host_scenes_check.cpp"), "-lz", "-o", str(exe)],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    png = tmp_path / "t.png"
    run = subprocess.run([str(exe), str(png)], capture_output=True, text=True, timeout=600)
    if run.returncode != 0 and "ASan runtime does not come first" in run.stderr:
        # a preloaded library took the first place in the link order: AddressSanitizer cannot start there, UBSan can
        build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                                os.path.join(ROOT, "tests", "native", "host_scenes_check.cpp"), "-lz", "-o", str(exe)],
                               capture_output=True, text=True, timeout=600)
        assert build.returncode == 0, build.stderr[-2000:]
        run = subprocess.run([str(exe), str(png)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stdout[-3000:] + run.stderr[-3000:]
    assert png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
