# why the `decode` object of bench.py's default line is slower than a stand-alone bench_decode.py on the same box:
#  A stand-alone, idle GPU   B right behind 30 s of training steps (clocks / temperature)   C beside an idle parent process that holds a HIP context
mkdir -p gpurun_out
DR='import bench_decode as B, torch, json; d = B.driver_record(torch.device("cuda:0"), cpu=False); print(json.dumps({"b1_p50": d["batch1"]["rtf_p50"], "us_per_token": d["batch1"]["search_us_per_token"], "b64": d["batch64"]["utterances_per_s"]}))'
echo "A stand-alone:"; timeout 300 python -c "$DR" 2>/dev/null | tail -1
echo "B behind 30 s of audio-only training steps:"; timeout 300 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-decode --no-fwd-encoder --no-box --no-eager --sustain-s 30 > /dev/null 2>&1; timeout 300 python -c "$DR" 2>/dev/null | tail -1
echo "A again (60 s later):"; sleep 45; timeout 300 python -c "$DR" 2>/dev/null | tail -1
echo "C beside an idle process with a HIP context, 8 streams and 4 GB:"
python - <<'PY' &
import torch, time
x = torch.empty(1 << 30, device="cuda"); s = [torch.cuda.Stream() for _ in range(8)]
for q in s:
    with torch.cuda.stream(q):
        x[:1024].zero_()
torch.cuda.synchronize(); time.sleep(100)
PY
PID=$!
sleep 15; timeout 300 python -c "$DR" 2>/dev/null | tail -1
kill $PID 2>/dev/null; wait $PID 2>/dev/null
echo "A again:"; timeout 300 python -c "$DR" 2>/dev/null | tail -1
