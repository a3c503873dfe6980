bash scripts/gpu_ab_lib.sh "lib_b/libtavsr_hip.so tailored-avsr_amd/tavsr/lib/libtavsr_hip.so" --steps 10 --warmup 3 --sustain-s 0
bash scripts/gpu_ab_lib.sh "lib_b/libtavsr_hip.so tailored-avsr_amd/tavsr/lib/libtavsr_hip.so" --workload asr --sustain-s 0
