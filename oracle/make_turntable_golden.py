"""Golden turntable cameras from the reference's own `get_turntable_cameras` (gs_core.py:50-85): the function's source is
sliced out of the file (the module itself imports the CUDA rasterizer) and executed verbatim.  Build container only."""
import os
import re

import numpy as np

SRC = open("/root/reference/diffusionGS/models/gsrenderer/gs_core.py").read()
m = re.search(r"^def get_turntable_cameras\(.*?\n(?=^def )", SRC, flags=re.S | re.M)
ns = {"np": np}
exec(m.group(0), ns)
out = {}
for tag, kw in (("default", {}), ("v150_512", dict(num_views=150, w=512, h=512)), ("elev20", dict(num_views=5, elevation=20, radius=3.0, w=256, h=192))):
    w, h, v, k, c2w = ns["get_turntable_cameras"](**kw)
    out[f"{tag}_whv"] = np.array([w, h, v])
    out[f"{tag}_fxfycxcy"] = k
    out[f"{tag}_c2w"] = c2w
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "turntable_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path)
