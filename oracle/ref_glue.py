"""The REFERENCE's own render glue, unchanged, on top of the drop-in package (checker; TEST INFRASTRUCTURE ONLY).

`north_star`: the drop-in must sit under diffusionGS/models unchanged.  oracle/build_ref.py copies the reference's
diffusionGS/models/gsrenderer/{gs_core.py, renderer.py} VERBATIM into the git-ignored oracle/_ref/py/ (like the translated
rasterizer sources next to it: nothing of the reference is committed, but the directory travels to the GPU box with the snapshot);
`load()` imports them as the reference's package path `diffusionGS.models.gsrenderer.*` with

  * `diff_gaussian_rasterization` resolving to open-diffusiongs_amd/diff_gaussian_rasterization (the drop-in under test), and
  * the third-party modules gs_core.py imports at module level but the render path never calls (cv2, plyfile, imageio, kiui, trimesh,
    easydict, diffusionGS.utils.mesh_utils: video / PLY / mesh export) replaced by empty stand-ins when they are not installed.

What runs is the reference's `Camera` (gs_core.py:277-316), `GaussianModel.set_data / get_*` (:321-575), `render_opencv_cam`
(:874-945), `DeferredGaussianRender` + `deferred_gaussian_render` (:949-1064) and `Renderer.forward` (renderer.py:20-92).
"""
import importlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
PY = os.path.join(HERE, "_ref", "py")
FILES = ("gs_core.py", "renderer.py")


def available():
    return all(os.path.exists(os.path.join(PY, "diffusionGS", "models", "gsrenderer", f)) for f in FILES)


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except Exception:                     # noqa: BLE001 -- not installed (or not importable here): an empty stand-in
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__stub__ = True
        sys.modules[name] = m
        return m


class _EasyDict(dict):
    """easydict.EasyDict as far as the reference's config objects use it (attribute access)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


_loaded = None


def load():
    """-> the reference's `diffusionGS.models.gsrenderer.renderer` module (its gs_core is `.gs_core` next to it)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("oracle/_ref/py is missing: run oracle/build_ref.py where /root/reference exists")
    pkg = os.path.join(os.path.dirname(HERE), "open-diffusiongs_amd")
    for p in (PY, pkg):
        if p not in sys.path:
            sys.path.insert(0, p)
    missing = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub of a module the render path does not use"))
    _stub("cv2")
    _stub("plyfile", PlyData=missing, PlyElement=missing)
    _stub("imageio")
    _stub("kiui")
    _stub("trimesh")
    _stub("easydict", EasyDict=_EasyDict)
    if "diffusionGS.utils" not in sys.modules:      # the package skeleton under oracle/_ref/py has no utils/: mesh export only
        u = types.ModuleType("diffusionGS.utils")
        u.__path__ = []
        sys.modules["diffusionGS.utils"] = u
        mu = types.ModuleType("diffusionGS.utils.mesh_utils")
        mu.decimate_mesh = mu.clean_mesh = missing
        sys.modules["diffusionGS.utils.mesh_utils"] = mu
    import diff_gaussian_rasterization as dgr
    assert os.path.realpath(os.path.dirname(dgr.__file__)).startswith(os.path.realpath(pkg)), "diff_gaussian_rasterization is not the drop-in"
    _loaded = importlib.import_module("diffusionGS.models.gsrenderer.renderer")
    return _loaded


def diffusion_available():
    return os.path.exists(os.path.join(PY, "diffusionGS", "models", "diffusion", "gaussian_diffusion.py")) and os.path.exists(os.path.join(PY, "ref_callers.py"))


def load_diffusion():
    """-> the reference's `diffusionGS.models.diffusion` package (create_diffusion, SpacedDiffusion, GaussianDiffusion), verbatim."""
    if not diffusion_available():
        raise RuntimeError("oracle/_ref/py/diffusionGS/models/diffusion is missing: run oracle/build_ref.py where /root/reference exists")
    if PY not in sys.path:
        sys.path.insert(0, PY)
    for name in ("diffusionGS", "diffusionGS.models"):          # namespace skeleton (the gsrenderer loader may have made it already)
        importlib.import_module(name)
    return importlib.import_module("diffusionGS.models.diffusion")


def load_callers():
    """-> module with the reference's `TransformInput` and `PointDiffusionSystem.forward` (as a plain function of `self`), verbatim
    source ranges (oracle/build_ref.py)."""
    if not diffusion_available():
        raise RuntimeError("oracle/_ref/py/ref_callers.py is missing: run oracle/build_ref.py where /root/reference exists")
    if PY not in sys.path:
        sys.path.insert(0, PY)
    return importlib.import_module("ref_callers")
