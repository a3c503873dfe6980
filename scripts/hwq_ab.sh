# how the number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES, default 4) moves the search
# (driver-record protocol: batch-1 p50 RTF over 8 utterances, batch-64 utt/s over 128), alternating inside one call
mkdir -p gpurun_out
DR='import bench_decode as B, torch, json; d = B.driver_record(torch.device("cuda:0"), cpu=False); print(json.dumps({"b1_p50": d["batch1"]["rtf_p50"], "us_per_token": d["batch1"]["search_us_per_token"], "b64": d["batch64"]["utterances_per_s"]}))'
for rep in 1 2; do
for q in "" 2 3 8; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  echo "GPU_MAX_HW_QUEUES=${q:-default(4)}: $(timeout 300 python -c "$DR" 2>/dev/null | tail -1)"
done; done
