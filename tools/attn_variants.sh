#!/bin/bash
# Builds open-diffusiongs_amd/lib/libdgs_hip_<tag>.so = the product library with dit_attention.hip compiled under extra -D flags
# (the experiment switches of the attention kernel): the "B" sides of A/B runs with tools/attn_ab.py on the GPU box.
#   tools/attn_variants.sh tag1 "-DDGS_EXP_X" tag2 "-DDGS_EXP_Y -DDGS_EXP_Z" ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/open-diffusiongs_amd/lib
PYTHONPATH=$R/open-diffusiongs_amd python -m dgs_amd.build > /dev/null
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$R/include" -I"$R/open-diffusiongs_amd/csrc" -Wno-unused-value -Wno-unused-result \
      -fno-honor-nans -fno-slp-vectorize $flags -c "$R/open-diffusiongs_amd/csrc/dit_attention.hip" -o "/tmp/attn_$tag.o" 2> /dev/null
  objs=$(ls "$L"/*.o | grep -v "/dit_attention.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$L/libdgs_hip_$tag.so" $objs "/tmp/attn_$tag.o"
  echo "$L/libdgs_hip_$tag.so  ($flags)"
done
