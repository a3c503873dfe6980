# how the number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES, default 4) moves the search:
# batch 64 (utt/s) and batch 1 (p50 RTF)
mkdir -p gpurun_out
for q in "" 1 2 3 6 8; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  timeout 600 python bench_decode.py --utterances 128 --batch 64 --no-cpu-baseline > gpurun_out/hwq.json 2> gpurun_out/hwq.err || tail -3 gpurun_out/hwq.err
  python - "${q:-default}" 64 <<'PY'
import json, sys
d = json.loads(open("gpurun_out/hwq.json").read().strip().splitlines()[-1])
print(f"GPU_MAX_HW_QUEUES={sys.argv[1]:8s} batch {sys.argv[2]}: {d.get('value')} {d.get('unit')}  p50 RTF {d.get('rtf_p50')}", flush=True)
PY
  timeout 600 python bench_decode.py --utterances 12 --batch 1 --no-cpu-baseline > gpurun_out/hwq.json 2> gpurun_out/hwq.err || tail -3 gpurun_out/hwq.err
  python - "${q:-default}" 1 <<'PY'
import json, sys
d = json.loads(open("gpurun_out/hwq.json").read().strip().splitlines()[-1])
print(f"GPU_MAX_HW_QUEUES={sys.argv[1]:8s} batch {sys.argv[2]}: {d.get('value')} {d.get('unit')}  p50 RTF {d.get('rtf_p50')}", flush=True)
PY
done
