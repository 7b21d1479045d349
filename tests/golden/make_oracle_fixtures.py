"""Generates the oracle-made golden vectors under tests/golden/ (run from the repository root:
`python tests/golden/make_oracle_fixtures.py`; `... converged final` remakes one converged image only, `... first_hits`
only the first-hit grids). The reference ships no fixtures and cannot be built in
this environment, so these vectors come from the CPU oracle (oracle/), which is itself pinned only by
the analytic Cornell known answers, the Philox known answers and the sampler properties — they freeze the
oracle's behaviour so that a later change of the oracle or of the device code shows up against fixed
data.

  first_hit_<scene>.npz   64x36 grid of fixed camera rays (lens offset 0, time 0.5), free-path seed 11:
                          leaf primitive id (int16, -1 = miss) and t (float32)
  converged_<scene>.npz   96x54 mean radiance (float32, row 0 = image bottom) of SPP samples with seed 1000,
                          depth 50, and `half_rmse` = per-channel RMSE between the means of the first and
                          the second half of the samples (= 2 sigma / sqrt(SPP) for per-sample sigma)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SCENES = ["cornell", "smoke", "random", "final", "2sp", "2psp", "earth", "simplelight"]
CONVERGED = {"cornell": 16384, "smoke": 8192, "random": 4096, "final": 4096}
FW, FH = 64, 36
CW, CH = 96, 54


def earth():
    from PIL import Image
    im = Image.open(os.path.join(ROOT, "ray_tracing_amd", "assets", "earthmap.jpg")).convert("RGBA")
    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))  # top row first; the scene builder applies texture.rs:84


def main():
    e = earth()
    only = sys.argv[2:] if len(sys.argv) > 2 and sys.argv[1] == "converged" else None
    for name in ([] if only else SCENES):
        orc = O.OracleRenderer(name, FW, FH, earth_rgba=e if name in ("earth", "final") else None)
        ids, t = orc.first_hit(orc.camera_rays(0.5), seed=11)
        assert ids.max() < 32767
        np.savez_compressed(os.path.join(OUT, f"first_hit_{name}.npz"), prim_id=ids.astype(np.int16).reshape(FH, FW),
                            t=t.reshape(FH, FW))
        print(name, "first hits:", int((ids >= 0).sum()), "of", ids.size, flush=True)
    for name, spp in CONVERGED.items():
        if len(sys.argv) > 1 and sys.argv[1] == "first_hits":
            break
        if only and name not in only:
            continue
        t0 = time.time()
        orc = O.OracleRenderer(name, CW, CH, earth_rgba=e if name == "final" else None)
        a, _ = orc.render_sums(spp // 2, 50, seed=1000, spp_begin=0)
        b, _ = orc.render_sums(spp // 2, 50, seed=1000, spp_begin=spp // 2)
        mean = ((a.astype(np.float64) + b) / spp).astype(np.float32)
        half = np.sqrt((((a - b) / (spp // 2)) ** 2).mean(axis=(0, 1))).astype(np.float32)
        np.savez_compressed(os.path.join(OUT, f"converged_{name}.npz"), mean=mean, half_rmse=half,
                            spp=np.int32(spp), seed=np.int32(1000), depth=np.int32(50))
        print(name, "converged", spp, "spp in", round(time.time() - t0, 1), "s; half rmse", half, flush=True)


if __name__ == "__main__":
    main()
