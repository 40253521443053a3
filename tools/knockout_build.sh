#!/bin/bash
# Knock-out builds of the attention backward (dit_attention_backward.hip, -DDGS_INSTRUMENT -DDGS_KNOCK=n): lib/libdgs_hip_k<n>.so =
# the product objects with that one file replaced.  Timing only (tools/attn_bwd_ab.py); see kKnock in the source for the list.
set -e
cd "$(dirname "$0")/../open-diffusiongs_amd"
python -m dgs_amd.build > /dev/null
objs=$(ls lib/*.o | grep -v dit_attention_backward.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -Wno-unused-value -Wno-unused-result -fno-slp-vectorize \
      -DDGS_INSTRUMENT -DDGS_KNOCK=$n -c csrc/dit_attention_backward.hip -o /tmp/attn_bwd_k$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libdgs_hip_k$n.so $objs /tmp/attn_bwd_k$n.o
  echo lib/libdgs_hip_k$n.so
done
