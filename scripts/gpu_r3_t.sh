python -m pytest tests/test_gpu_av.py -x -q -k "stage or resnet or frontend" 2>&1 | tail -2
for v in 0 -1 1; do
  if [ $v = -1 ]; then unset TAVSR_CONV_COLMAJOR; else export TAVSR_CONV_COLMAJOR=$v; fi
  timeout 600 python profiles/gemm_shapes.py --workload avsr > gpurun_out/shapes_cm$v.txt 2>&1
  echo "== TAVSR_CONV_COLMAJOR=$v"; grep "gemm_kernel<NT> M=\(387200\|115200\|28800\) N=\(128\|256\|512\) K=\(1152\|2304\|4608\)" gpurun_out/shapes_cm$v.txt | cut -c1-110
done
unset TAVSR_CONV_COLMAJOR
for r in 1 2; do
for v in 0 -1; do
  if [ $v = -1 ]; then unset TAVSR_CONV_COLMAJOR; else export TAVSR_CONV_COLMAJOR=$v; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('av colmajor=$v', d['value'], d['ms_per_step'])"
done
done
