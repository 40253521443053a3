#!/bin/bash
# Round-3 GPU call 3: GPU suite with the hardware-exp blend default, the reference's own glue over the drop-in, edge cases on the gfx950
# build; rasterizer timings exact vs hardware exp; FETCH_SIZE calibration patterns; the contract bench with the new raster object.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03c
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_ref_glue_gpu.py tests/test_raster_forward_gpu.py tests/test_raster_backward_gpu.py tests/test_raster_ref_gpu.py -m gpu -q -x 2>&1 | tail -12 > $out/pytest_raster.txt
cat $out/pytest_raster.txt
for mode in 1 0; do
  for regime in trained init; do
    DGS_RASTER_EXACT_EXP=$mode timeout 120 python tools/raster_microbench.py --res 256 --regime $regime 2>&1 | grep -E "forward|views/s" | sed "s/^/exact_exp=$mode $regime: /" >> $out/raster_exp_ab.txt
  done
done
cat $out/raster_exp_ab.txt
(cd tools/ubench && timeout 120 ./fetch_calib_bench) > $out/fetch_calib_plain.txt 2>&1
cat $out/fetch_calib_plain.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_ref_glue_gpu.py --deselect tests/test_raster_forward_gpu.py --deselect tests/test_raster_backward_gpu.py --deselect tests/test_raster_ref_gpu.py 2>&1 | tail -6 > $out/pytest_rest.txt
cat $out/pytest_rest.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
cut -c1-700 $out/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03c/bench.json"))
print(json.dumps(d.get("raster"), indent=1)[:3000])
PY
