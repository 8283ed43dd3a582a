for w in ident pond droste; do
  for g in 0 1; do
    MMHIP_PAIR_GENERIC=$g python bench.py --workload $w --no-extras --steps 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w generic=$g kernel_ms', round(j['roofline']['kernel_ms'],4))"
  done
done
MMHIP_PAIR_GENERIC=1 python bench.py --workload droste -D NoTransparency=1 --no-extras --steps 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('droste NT generic=1 kernel_ms', round(j['roofline']['kernel_ms'],4))"
