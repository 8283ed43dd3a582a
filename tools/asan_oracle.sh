# The oracle's C (runtime + generated filters) under AddressSanitizer + UBSan on the CPU: renders the native-flow
# fuzzer's filters (both modes) and the CPU test-suite's oracle cases.  usage: bash tools/asan_oracle.sh [seeds]
set -e
cd "$(dirname "$0")/.."
export MM_ORACLE_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
python3 - "${1:-120}" <<'PY'
import importlib.util, sys
import numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fz", "tools/fuzz_native_flow.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import mathmap_amd as mm
from oracle.ccgen import CpuFilter
from tests import filters as F
n = int(sys.argv[1])
ran = 0
for rich in (False, True):
    for seed in range(n):
        w, h = [(96, 64), (64, 96), (80, 80), (112, 48)][seed % 4] if rich else (96, 64)
        iw, ih = [(w, h), (50, 70), (w, h), (131, 40)][(seed // 4) % 4] if rich else (w, h)
        g = m.Gen(seed, rich)
        src = g.filter()
        if g.calls_in_loops > 16:
            continue
        try:
            flt = mm.Filter(src)
        except mm.MathMapError:
            continue
        img = np.ascontiguousarray(F.synthetic_image(iw, ih, seed=3))
        cf = CpuFilter(flt.ir_json_raw)
        for mode, nn, t in ((1, 2, 0.25), (2, 3, 0.75), (0, 1, 0.6)):
            cf.render(w, h, uservals={"mode": mode, "n": nn}, images={"in": img}, t=t)
        ran += 1
print("oracle renders under ASan + UBSan:", ran, "filters x 3 settings, clean")
PY
rm -rf oracle/_build_san      # (224 MB of instrumented objects: not something to ship to the GPU box)
