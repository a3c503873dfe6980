# HIP / ROCr runtime knobs against the batch-1 search (driver-record protocol, no CPU leg), alternating inside one call
mkdir -p gpurun_out
DR='import bench_decode as B, torch, json; d = B.driver_record(torch.device("cuda:0"), cpu=False); print(json.dumps({"b1_p50": d["batch1"]["rtf_p50"], "us_per_token": d["batch1"]["search_us_per_token"], "b64": d["batch64"]["utterances_per_s"]}))'
for rep in 1 2; do
for cfg in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=2 HIP_FORCE_DEV_KERNARG=1"; do
  echo "$cfg: $(env $cfg timeout 300 python -c "$DR" 2>/dev/null | tail -1)"
done; done
