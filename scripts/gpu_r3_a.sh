#!/bin/bash
# round 3, first GPU call: parity of the streaming FFN kernel, its timing under several plans, the 12-layer forward
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ffn2.py -x -q 2>&1 | tail -15 > gpurun_out/r3a_ffn2_tests.txt
cat gpurun_out/r3a_ffn2_tests.txt
python scripts/ffn2_bench.py > gpurun_out/r3a_ffn2_bench.txt 2>&1
cat gpurun_out/r3a_ffn2_bench.txt
python bench.py --mode fwd-encoder > gpurun_out/r3a_fwd_encoder_ffn2.json 2>gpurun_out/r3a_fwd_encoder_ffn2.err
TAVSR_FFN2=0 python bench.py --mode fwd-encoder > gpurun_out/r3a_fwd_encoder_base.json 2>gpurun_out/r3a_fwd_encoder_base.err
python - <<'PY'
import json
for n in ("ffn2", "base"):
    try:
        d = json.loads(open(f"gpurun_out/r3a_fwd_encoder_{n}.json").read().strip().splitlines()[-1])["fwd_encoder"]
        print(n, {k: v for k, v in d.items() if k.startswith("layers12")})
    except Exception as e:
        print(n, "failed", e)
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffn.py -x -q 2>&1 | tail -5
