# the N > 1 step's exchange path on ONE GPU with RCCL itself (one-rank communicator): two-graph step, eager hook-driven loop
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dp_gloo.py -q -m gpu -k "world1 or rccl" 2>&1 | tail -4
timeout 900 python bench.py --gpus 1 --split-backward --force-rccl --no-fwd-encoder --no-cpu-baseline --no-asr > gpurun_out/r04_dp_rehearsal_rccl_world1.json 2> gpurun_out/r04_dp_rehearsal_rccl_world1.err; echo "rc=$?"; tail -3 gpurun_out/r04_dp_rehearsal_rccl_world1.err; cut -c1-1500 gpurun_out/r04_dp_rehearsal_rccl_world1.json
timeout 900 python bench.py --gpus 1 --no-graph --force-rccl --no-fwd-encoder --no-cpu-baseline --no-asr --no-roofline > gpurun_out/r04_dp_rehearsal_rccl_world1_eager.json 2>/dev/null; echo "rc=$?"; cut -c1-900 gpurun_out/r04_dp_rehearsal_rccl_world1_eager.json
