import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt, oracle as O
W, H = 1920, 1080
orc = O.OracleRenderer("cornell", W, H)
ctx = rt.Context(0); sc = rt.select_scene("cornell"); ctx.scene_upload(sc)
rays = orc.camera_rays(0.25)
oi, ot = orc.first_hit(rays); gi, gt = ctx.first_hit(rays)
bad = np.nonzero(oi != gi)[0]
print("mismatches", len(bad))
for b in bad[:20]:
    print(b, b % W, b // W, oi[b], gi[b], ot[b], gt[b], rays[b])
hit = (oi == gi) & (oi >= 0)
rel = np.abs(ot[hit] - gt[hit]) / ot[hit]
print("max rel t", rel.max(), "n t differing", (ot[hit] != gt[hit]).sum(), "by id", np.unique(oi[hit][ot[hit] != gt[hit]], return_counts=True))
