"""First hits (primitive id, t) of a seeded set of rays in a scene, saved to argv[1] — for comparing two builds
of the library (MIRT_LIB) on large scenes the CPU oracle cannot scan. usage: first_hit_dump.py out.npz [random_num]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt
num = int(sys.argv[2]) if len(sys.argv) > 2 else 500
sc = rt.select_scene("random", random_num=num)
ctx = rt.Context(0); ctx.scene_upload(sc)
rng = np.random.default_rng(5)
n = 400000
o = np.concatenate([rng.uniform(-num, num, (n, 1)), rng.uniform(0.3, 30, (n, 1)), rng.uniform(-num, num, (n, 1))], axis=1)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d, rng.uniform(0, 1, (n, 1))], axis=1).astype(np.float32)
ids, t = ctx.first_hit(rays)[:2]
np.savez(sys.argv[1], ids=ids, t=t)
print(os.path.basename(rt.LIB_PATH), "hits", int((ids >= 0).sum()), "of", n, "checksum", int(ids.astype(np.int64).sum()), float(t[np.isfinite(t)].sum()))
