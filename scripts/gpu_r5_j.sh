mkdir -p gpurun_out
i=0
for env in "TAVSR_DECODE_RECORD_QUEUE=1 TAVSR_DECODE_CTC_BESIDE=1" "TAVSR_DECODE_RECORD_QUEUE=0 TAVSR_DECODE_CTC_BESIDE=1" "TAVSR_DECODE_RECORD_QUEUE=1 TAVSR_DECODE_CTC_BESIDE=0" "TAVSR_DECODE_RECORD_QUEUE=0 TAVSR_DECODE_CTC_BESIDE=0"; do
i=$((i+1))
env $env timeout 300 python scripts/decode_stress.py 15 > gpurun_out/stress_$i.log 2>&1; echo "stress [$env] rc=$?"; grep -E "decode stress|differs|error|Error|HSA" gpurun_out/stress_$i.log | head -5
done
