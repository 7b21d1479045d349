"""Sweep of the shared-walk thresholds (MIRT_WALK_BUDGET / MIRT_WALK_LANES / MIRT_NODE_LANES) on the hierarchy scenes:
one child process per setting (the library reads the variables once), Msamples/s of the persistent kernel from the
HIP events, best of two timed renders. usage: python scripts/walk_sweep.py [scene ...]"""
import itertools, os, subprocess, sys

CHILD = r"""
import sys
sys.path.insert(0, %r)
import ray_tracing_amd as rt
CASES = {"random": ("random", 1920, 1080, 32, 50, {}), "final": ("final", 1920, 1080, 16, 50, {}),
         "million": ("random", 3840, 2160, 1, 10, {"random_num": 500})}
scene, W, H, spp, depth, kw = CASES[sys.argv[1]]
ctx = rt.Context(0)
ctx.scene_upload(rt.select_scene(scene, **kw))
ctx.camera_set(rt.Camera.select_camera(W / H, scene))
best = 0.0
for rep in range(3):
    ctx.frame_begin(W, H)
    ctx.render(spp, depth, seed=rep, jitter=spp != 1, kernel=rt.KERNEL_QUEUE)
    ctx.synchronize()
    st = ctx.stats()
    if rep:
        best = max(best, st["samples"] / st["kernel_ms"] / 1e3)
print(f"{best:.0f}")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    scenes = sys.argv[1:] or ["random", "final", "million"]
    settings = [(None, None, None)] + list(itertools.product((16, 32, 48, 96), (8, 24, 40), (8, 16, 32)))
    for scene in scenes:
        for budget, lanes, node in settings:
            env = dict(os.environ)
            if budget is not None:
                env.update(MIRT_WALK_BUDGET=str(budget), MIRT_WALK_LANES=str(lanes), MIRT_NODE_LANES=str(node))
            out = subprocess.run([sys.executable, "-c", CHILD, scene], env=env, capture_output=True, text=True, timeout=120)
            print(scene, "default" if budget is None else f"budget={budget} lanes={lanes} node={node}", out.stdout.strip() or out.stderr[-300:], flush=True)


if __name__ == "__main__":
    main()
