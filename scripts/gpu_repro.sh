for f in "X=0" "TAVSR_BRANCH_STREAM=0" "TAVSR_FRONT_PAIR=0" "TAVSR_STEM_IMPLICIT=0" "TAVSR_CONV_ZMAP=0"; do
  env $f REPRO_COLD=1 timeout 800 python scripts/av_repro_check.py 32 6 2>&1 | grep -v Warning | tail -22
done
