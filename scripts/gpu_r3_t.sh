for v in 1 0; do
  TAVSR_CONV_TILE=$v timeout 600 python profiles/gemm_shapes.py --workload avsr > gpurun_out/shapes_tile$v.txt 2>&1
  echo "== TAVSR_CONV_TILE=$v"; grep "gemm_kernel<NT> M=\(1548800\|387200\|115200\|28800\) \|gemm_kernel<NN> M=\(1548800\|387200\|115200\|28800\) " gpurun_out/shapes_tile$v.txt | cut -c1-110
done
