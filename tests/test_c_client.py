"""The C ABI used from C: tests/native/c_client.c includes include/mirt.h and include/mirt_host.h, is compiled
with gcc -std=c99 -pedantic (the headers have to be valid C, not only C++) and linked against the library the way a
cgo or Rust build script would. Without a GPU it exercises the host-side entry points; on the GPU it renders the
Cornell box through the call sequence that stands for Renderer::new + draw (renderer.rs:41-123) and the frame must
equal the one the Python binding gets for the same parameters."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "ray_tracing_amd")


def _build(tmp_path):
    exe = tmp_path / "c_client"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "c_client.c"), "-o", str(exe), "-L", LIBDIR, "-lmirt",
                        "-Wl,-rpath," + LIBDIR], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_client_builds_and_uses_the_host_entry_points(rt, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("host ok: 13 primitives, camera at (278, 278, -800)"), r.stdout + r.stderr


@pytest.mark.gpu
def test_c_client_renders_the_same_frame_as_the_binding(rt, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe), "run"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    words = r.stdout.split()
    got = dict(zip(words[3::2], words[4::2]))            # "frame digest <hex> lit <n> samples <n> segments <n> faults <n> alpha <n>"
    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene("cornell", earth_rgba=None))
    ctx.camera_set(rt.Camera.select_camera(64 / 36, "cornell"))
    ctx.frame_begin(64, 36)
    ctx.render(8, 50, seed=3, jitter=True)
    frame = ctx.resolve_rgba8()
    st = ctx.stats()
    digest = 1469598103934665603
    for b in frame.reshape(-1).tolist():
        digest = ((digest ^ b) * 1099511628211) & 0xffffffffffffffff
    assert words[0] == "frame" and int(words[2], 16) == digest
    assert int(got["samples"]) == 64 * 36 * 8 == st["samples"] and int(got["segments"]) == st["segments"]
    assert int(got["faults"]) == 0 and int(got["alpha"]) == 255 and int(got["lit"]) > 1000
