#!/bin/bash
# conventional-encoder intermediate CTC on the HIP path + the box calibration with data operands
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_interctc.py tests/test_gpu_av.py -x -q -m gpu 2>&1 | tail -5
python - <<'PY' 2>&1 | tail -5
import json, sys, torch
sys.path.insert(0, "tailored-avsr_amd"); sys.path.insert(0, ".")
import bench
for _ in range(2):
    print(json.dumps(bench.box_calibration(torch.device("cuda:0"))))
PY
