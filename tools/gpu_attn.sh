for qt in 2 3 4; do
  echo "== QT=$qt"; SS_ATTN_QT=$qt timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['phase_ms'], j['roofline']['achieved'])"
done
SS_ATTN_QT=4 timeout 300 python -m pytest tests -q -m gpu -x -k "encoder or greedy" 2>&1 | tail -2
