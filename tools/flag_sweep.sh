for fl in "" "-O2" "-fno-unroll-loops" "-mllvm -amdgpu-early-inline-all=true" "-ffast-math-dummy"; do
  if [ "$fl" = "-ffast-math-dummy" ]; then continue; fi
  echo "FLAGS=[$fl]"
  for w in mandelbrot ident pond droste; do
    MMHIP_HIPRTC_FLAGS="$fl" python bench.py --workload $w --no-cpu-baseline --no-generic 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['config']['workload'][:12], round(d['roofline']['kernel_ms'],4))
"
  done
done
