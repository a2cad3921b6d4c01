for p in 16 32 64; do
  echo "KC_PPT=$p"; KC_PPT=$p python bench.py --steps 2 --warmup 1 --no-cli --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['phases_ms'])"
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "global or superstring or compute_S or cfg1" 2>&1 | tail -3
