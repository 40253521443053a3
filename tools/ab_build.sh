#!/bin/bash
# Builds open-diffusiongs_amd/lib/libdgs_hip_base.so = the product library with ONE csrc file taken from a git revision (default
# HEAD) instead of the working tree: the "A" of an A/B run on the GPU box (tools/attn_ab.py takes both libraries in one process;
# *.so is git-ignored but travels with gpurun).
#   tools/ab_build.sh dit_attention.hip [rev]      then   gpurun -- 'python tools/attn_ab.py open-diffusiongs_amd/lib/libdgs_hip_base.so'
set -eu
f=$1; rev=${2:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/open-diffusiongs_amd/lib
PYTHONPATH=$R/open-diffusiongs_amd python -m dgs_amd.build > /dev/null          # objects of the working tree
tmp=$(mktemp -d)
git -C "$R" show "$rev:open-diffusiongs_amd/csrc/$f" > "$tmp/$f"
extra=$(python - "$f" <<'PY'
import sys
sys.path.insert(0, "open-diffusiongs_amd")
from dgs_amd import build
print(" ".join(build.FLAGS.get(sys.argv[1], [])))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$R/include" -I"$R/open-diffusiongs_amd/csrc" -Wno-unused-value -Wno-unused-result \
    $extra -c "$tmp/$f" -o "$tmp/base.o" 2> /dev/null
objs=$(ls "$L"/*.o | grep -v "/${f%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$L/libdgs_hip_base.so" $objs "$tmp/base.o"
rm -rf "$tmp"
echo "$L/libdgs_hip_base.so  ($f from $rev)"
