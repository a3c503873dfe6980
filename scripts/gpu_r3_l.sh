python -m pytest tests/test_gpu_attn.py -x -q 2>&1 | tail -3




