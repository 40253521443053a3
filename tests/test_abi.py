"""The product library (hipcc, gfx950) loads on a GPU-less host and exports every symbol include/*.h declares; the ctypes
structures mirror the C structs (size check through a tiny C program).  No compute call is made here."""
import ctypes
import os
import re
import subprocess
import tempfile

from dgs_amd import _native
from dgs_amd import build as build_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def _declared(header):
    txt = open(os.path.join(INC, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dgs_[a-z0-9_]+)\s*\(", txt)) - {"dgs_alloc_fn"})


def test_every_declared_symbol_is_exported():
    build_mod.build_hip()
    lib = _native.lib()
    for header, table in (("dgs_raster.h", _native.RASTER_SYMBOLS), ("dgs_dit.h", _native.DIT_SYMBOLS), ("dgs_sampler.h", _native.SAMPLER_SYMBOLS), ("dgs_loss.h", _native.LOSS_SYMBOLS),
                          ("dgs_optim.h", _native.OPTIM_SYMBOLS)):
        names = _declared(header)
        assert names, header
        for n in names:
            assert hasattr(lib, n), f"{n} declared in {header} but not exported"
        assert sorted(table) == names, (header, sorted(set(names) ^ set(table)))
    assert lib.dgs_abi_version() == _native.ABI_VERSION == int(re.search(r"#define DGS_ABI_VERSION (\d+)", open(os.path.join(INC, "dgs_raster.h")).read()).group(1))
    assert lib.dgs_dit_lpad(4098) == 4352
    assert lib.dgs_status_string(-1).decode().startswith("invalid argument")


def test_ctypes_structs_match_c_layout():
    structs = ["DgsRasterForwardArgs", "DgsRasterBackwardArgs", "DgsDitGemmArgs", "DgsDitAttentionArgs", "DgsDitLayerNormArgs",
               "DgsDitRowLinearArgs", "DgsDitLayerWeights", "DgsDitModel", "DgsDitForwardArgs", "DgsSamplerStepArgs", "DgsMseArgs",
               "DgsDitLayerNormBackwardArgs", "DgsDitRowLinearBackwardArgs", "DgsDitGateMulArgs", "DgsDitAttentionBackwardArgs",
               "DgsDitBackwardArgs", "DgsDitRunBlocksArgs", "DgsResizeArgs", "DgsAdamWTensor", "DgsAdamWArgs"]
    src = '#include <stdio.h>\n#include "dgs_dit.h"\n#include "dgs_sampler.h"\n#include "dgs_loss.h"\n#include "dgs_optim.h"\nint main(){' + "".join(
        f'printf("%zu\\n", sizeof({s}));' for s in structs) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I" + INC, c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    for s, n in zip(structs, sizes):
        assert ctypes.sizeof(getattr(_native, s)) == n, (s, ctypes.sizeof(getattr(_native, s)), n)
