#!/bin/bash
# in-call A/B of the round-3 switches on the headline step (interleaved, two rounds)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
F="--steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0"
for r in 1 2; do
  for cfg in "default:" "ffn2_off:TAVSR_FFN2=0" "joint_off:TAVSR_AV_JOINT_FFN=0"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    v=$(env $envs python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "round $r av  $name $v"
    v=$(env $envs python bench.py $F --workload asr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "round $r asr $name $v"
  done
done | tee gpurun_out/r3h_ab.txt
