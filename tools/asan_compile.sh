# Builds the compiler's host sources with AddressSanitizer + UBSan (g++, CPU only) and compiles the fuzz generators'
# filters and the reference's example filters through them.  usage: bash tools/asan_compile.sh
set -e
cd "$(dirname "$0")/.."
B=/tmp/mm_asan
mkdir -p $B
S=mathmap_amd/csrc
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$S -Iinclude -o $B/driver tools/asan_compile_driver.cpp \
    $S/ir.cpp $S/parser.cpp $S/gen.cpp $S/lower.cpp $S/builtins.cpp $S/passes.cpp $S/specialize.cpp $S/hipgen.cpp \
    $S/prelude_blob.cpp $S/noise_prelude_blob.cpp $S/noise_table_blob.cpp $S/fastmath_blob.cpp
python3 - > $B/filters.txt <<'PY'
import importlib.util, sys, glob
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fz", "tools/fuzz_native_flow.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
out = []
for rich in (False, True):
    for seed in range(400):
        out.append(m.Gen(seed, rich).filter())
sys.path.insert(0, "tests")
from fuzz_filters import make_filter, make_filter_ex
for seed in range(300):
    out.append(make_filter(seed)[0]); out.append(make_filter_ex(seed)[0])
from tests import filters as F
for k, v in F.SOURCES.items():
    out.append(v)
for fn in sorted(glob.glob("tests/test_gpu_*.py")):
    import re
    for mm_ in re.finditer(r'"""(\s*filter .*?)"""', open(fn).read(), re.S):
        out.append(mm_.group(1))
print("\n====\n".join(out))
PY
ASAN_OPTIONS=detect_leaks=0 $B/driver < $B/filters.txt
