# A/B of the two-stream scorer step under the kernel trace: batch 1, parallel on / off
bash scripts/gpu_decode_prof.sh 1 | head -3
export TAVSR_DECODE_PARALLEL=0
bash scripts/gpu_decode_prof.sh 1 | head -3
