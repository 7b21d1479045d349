"""Generates tests/golden/cornell_first_hit_kat.json: analytic first-hit answers for the Cornell box.

The reference ships no tests or fixtures, so these are derived, in float64 and independently of any
renderer code, from the constants the reference declares: the camera preset of src/camera.rs:107-112
run through Camera::new (src/camera.rs:40-52) and the geometry of cornell_box (src/scene.rs:148-180).
Leaf ids follow world order: 0 green x=555, 1 red x=0, 2 floor, 3 ceiling, 4 back wall, 5-10 the six
sides of the rotated box (XY kmin, XY kmax, XZ kmin, XZ kmax, YZ kmin, YZ kmax), 11 light, 12 glass sphere.
"""
import json
import math
import os

import numpy as np

ASPECT = 1920.0 / 1080.0
theta = math.radians(40.0)
vh = 2.0 * math.tan(theta / 2.0)
vw = vh * ASPECT
origin = np.array([278.0, 278.0, -800.0])
w = np.array([0.0, 0.0, -1.0])          # -direction
u = np.cross([0.0, 1.0, 0.0], w)
v = np.cross(w, u)
horizontal = 10.0 * vw * u
vertical = 10.0 * vh * v
llc = origin - horizontal / 2 - vertical / 2 - 10.0 * w


def ray_dir(s, t):
    d = llc + s * horizontal + t * vertical - origin
    return d / np.linalg.norm(d)


def rect(axis, k, lo, hi):
    idx = {"XY": (0, 1, 2), "XZ": (0, 2, 1), "YZ": (1, 2, 0)}[axis]
    return ("rect", idx, k, lo, hi)


L = 555.0
prims = [
    (rect("YZ", L, (0, 0), (L, L)), None),
    (rect("YZ", 0.0, (0, 0), (L, L)), None),
    (rect("XZ", 0.0, (0, 0), (L, L)), None),
    (rect("XZ", L, (0, 0), (L, L)), None),
    (rect("XY", L, (0, 0), (L, L)), None),
]
box = (165.0, 330.0, 165.0)
inst = (math.radians(15.0), np.array([265.0, 0.0, 295.0]))
for axis, idx in (("XY", (0, 1, 2)), ("XZ", (0, 2, 1)), ("YZ", (1, 2, 0))):
    for k in (0.0, box[idx[2]]):
        prims.append((rect(axis, k, (0.0, 0.0), (box[idx[0]], box[idx[1]])), inst))
prims.append((rect("XZ", 554.0, (213.0, 227.0), (343.0, 332.0)), None))
prims.append((("sphere", np.array([190.0, 90.0, 190.0]), 90.0), None))


def to_local(o, d, inst):
    if inst is None:
        return o, d
    ang, off = inst
    c, s = math.cos(ang), math.sin(ang)
    def inv(p):
        return np.array([c * p[0] - s * p[2], p[1], s * p[0] + c * p[2]])
    return inv(o - off), inv(d)


def first_hit(o, d):
    best_t, best_id = math.inf, -1
    for i, (p, inst) in enumerate(prims):
        lo_, ld = to_local(o, d, inst)
        if p[0] == "rect":
            _, (xi, yi, zi), k, lo, hi = p
            if ld[zi] == 0:
                continue
            t = (k - lo_[zi]) / ld[zi]
            if t < 1e-3 or t > best_t:
                continue
            x, y = lo_[xi] + t * ld[xi], lo_[yi] + t * ld[yi]
            if x < lo[0] or x > hi[0] or y < lo[1] or y > hi[1]:
                continue
        else:
            _, c, r = p
            oc = lo_ - c
            a, hb, cc = ld @ ld, oc @ ld, oc @ oc - r * r
            disc = hb * hb - a * cc
            if disc < 0:
                continue
            t = (-hb - math.sqrt(disc)) / a
            if t < 1e-3 or t > best_t:
                t = (-hb + math.sqrt(disc)) / a
                if t < 1e-3 or t > best_t:
                    continue
        best_t, best_id = t, i
    return best_id, best_t


points = [(0.5, 0.5), (0.5, 0.9), (0.5, 0.85108), (0.56869, 0.23913), (0.02, 0.5), (0.35, 0.5), (0.3, 0.5),
          (0.7, 0.5), (0.5, 0.1), (0.45, 0.3), (0.62, 0.4), (0.4, 0.75), (0.55, 0.2), (0.25, 0.25), (0.75, 0.8)]
rows = []
for s, t in points:
    d = ray_dir(s, t)
    pid, tt = first_hit(origin, d)
    rows.append({"s": s, "t": t, "dir": [float(x) for x in d], "prim_id": pid, "hit_t": (None if pid < 0 else float(tt))})
out = {
    "source": "analytic, float64; camera src/camera.rs:40-52,107-112; scene src/scene.rs:148-180",
    "aspect": ASPECT,
    "camera": {"origin": origin.tolist(), "lower_left_corner": llc.tolist(), "horizontal": horizontal.tolist(),
               "vertical": vertical.tolist()},
    "sampler": {"cosine_z_min": math.sqrt(0.1), "cosine_Ez2": 0.55, "cosine_pdf_below_horizon": 1e-4,
                "light_rect_area": 130.0 * 105.0,
                "glass_sphere_pdf_from_floor_190_0_100": 1.0 / (2 * math.pi * (1 - math.sqrt(1 - 90.0 ** 2 / (90.0 ** 2 + 90.0 ** 2))))},
    "rays": rows,
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_first_hit_kat.json")
json.dump(out, open(path, "w"), indent=1)
print(path)
for r in rows:
    print(r["s"], r["t"], r["prim_id"], r["hit_t"])
