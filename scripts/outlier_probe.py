"""How far the pixels that are not 'close' in the image parity tests lie from the oracle (to set the outlier cap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, ray_tracing_amd as rt
W, H = 64, 36
earth = rt.load_earthmap()
for name, spp in [("cornell", 64), ("smoke", 32), ("random", 32), ("final", 16), ("2psp", 32), ("earth", 32), ("simplelight", 32), ("2sp", 32)]:
    e = earth if name in ("earth", "final") else None
    orc = oracle.OracleRenderer(name, W, H, earth_rgba=e)
    ctx = rt.Context(0); ctx.scene_upload(rt.select_scene(name, earth_rgba=e)); ctx.camera_set(rt.Camera.select_camera(W / H, name))
    worst = []
    for seed in (3, 4, 5, 6):
        want, _ = orc.render_sums(spp, 50, seed=seed)
        ctx.frame_begin(W, H); ctx.render(spp, 50, seed=seed)
        got = ctx.read_accum()[..., :3]
        close = np.isclose(got, want, rtol=5e-3, atol=5e-3 * spp * 0.05)
        bad = ~close.all(axis=2)
        d = np.abs(got - want).max(axis=2)
        worst.append((bad.mean(), d[bad].max() if bad.any() else 0.0, want.max(), (d[bad] / np.maximum(np.abs(want).max(axis=2)[bad], 1e-3)).max() if bad.any() else 0.0))
    print(name, spp, " ".join(f"[bad {b:.4f} maxdiff {m:.3f} wantmax {wm:.2f} rel {r:.3f}]" for b, m, wm, r in worst), flush=True)
