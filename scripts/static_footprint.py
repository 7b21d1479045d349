"""Device memory the library holds before anything is rendered: free memory of device 0 (hipMemGetInfo) in a fresh
process after each step — HIP runtime up, libmirt.so loaded, a context created (the library looks up its static
scratch then, which makes the runtime load the code object and with it the file-scope device arrays), a scene
uploaded, a frame begun, one render. usage: python scripts/static_footprint.py [library]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["MIRT_LIB"] = sys.argv[1]
hip = C.CDLL("libamdhip64.so")


def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 2**20


assert hip.hipSetDevice(0) == 0
steps = [("HIP runtime initialised", free_mb())]
import ray_tracing_amd as rt
rt.lib()
steps.append(("libmirt.so loaded", free_mb()))
ctx = rt.Context(0)
steps.append(("mirt_create", free_mb()))
ctx.scene_upload(rt.select_scene("cornell"))
ctx.camera_set(rt.Camera.select_camera(16 / 9, "cornell"))
steps.append(("scene uploaded", free_mb()))
ctx.frame_begin(1920, 1080)
steps.append(("mirt_frame_begin 1920x1080 (33.2 MB accumulation buffer)", free_mb()))
ctx.render(64, 50)
ctx.synchronize()
steps.append(("mirt_render 64 spp (16 chunk images: 531 MB)", free_mb()))
print("library:", rt.LIB_PATH)
for (name, f), (_, prev) in zip(steps, [steps[0]] + steps[:-1]):
    print(f"  {name:62s} free {f:10.1f} MiB   held by this step {prev - f:8.1f} MiB")
