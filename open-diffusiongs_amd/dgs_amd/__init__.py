"""dgs_amd -- MI355X-native implementation package behind the reference's operator surface.

Layout (everything the hot path needs, nothing else):
  csrc/ (sibling dir)      hand-written gfx950 HIP kernels + the C ABI (include/*.h)
  _native.py               ctypes loader of lib/libdgs_hip.so (fails loudly when missing)
  raster.py                torch-tensor front end of the rasterizer C ABI
  synth.py / cameras.py    synthetic inputs + camera math shared by tests and bench.py
"""
