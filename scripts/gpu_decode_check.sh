# decode path: parity tests, then BASELINE config 5 at batch 1 / 64 with the A/B switches given as arguments ("VAR=0 VAR2=1" per run)
mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || timeout 1200 python -m pytest tests/test_gpu_rowlin.py tests/test_beam_search.py -m gpu -x -q 2>&1 | tail -3
run() {
  echo "== $1 | batch $2"
  env $1 timeout 600 python bench_decode.py --utterances $3 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:j[k] for k in ('value','utterances_per_s','encoder_s','search_s','tokens_decoded')})"
}
for cfg in "$@"; do
  run "$cfg" 1 8
  run "$cfg" 64 256
done
