#!/bin/bash
# CPU ORACLE support: builds the pieces of the reference that compile from their own
# sources (nothing else of the reference builds here: clisp, bison and the glib/GSL/GIMP
# headers are missing).  Sources are compiled where they lie under /root/reference (the
# vendored libnoise zip is unpacked into oracle/_ref/, which is git-ignored); outputs go
# only to oracle/_ref/ (shipped to the GPU box as built files).
#   * builtins/spec_func.c            -> _ref/libspec_func.so  (cgamma)
#   * libnoisesrc-1.0.0.zip + libnoise-bestest.diff + oracle/noise_wrap.cpp
#                                     -> _ref/libmmnoise.so    (noise builtins; -fwrapv: libnoise relies on
#                                        wrapping signed overflow in IntValueNoise3D, which the 2009 build that made
#                                        the goldens had; with it render_voronoi_cells.png is reproduced exactly)
set -e
REF=/root/reference
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF" ] || { echo "no reference tree"; exit 0; }
mkdir -p "$OUT"
gcc -O2 -fPIC -shared -std=gnu99 -o "$OUT/libspec_func.so" "$REF/builtins/spec_func.c" -lm
echo "built $OUT/libspec_func.so"
if [ ! -f "$OUT/libmmnoise.so" ] || [ "$HERE/noise_wrap.cpp" -nt "$OUT/libmmnoise.so" ]; then
    rm -rf "$OUT/libnoise_src" && mkdir -p "$OUT/libnoise_src"
    (cd "$OUT/libnoise_src" && unzip -q -o "$REF/libnoisesrc-1.0.0.zip" && patch -p1 -s < "$REF/libnoise-bestest.diff")
    S="$OUT/libnoise_src/noise/src"
    g++ -O2 -fwrapv -fPIC -shared -w -I"$S" -o "$OUT/libmmnoise.so" "$HERE/noise_wrap.cpp" "$S/noisegen.cpp" \
        "$S/module/perlin.cpp" "$S/module/billow.cpp" "$S/module/ridgedmulti.cpp" "$S/module/voronoi.cpp" \
        "$S/module/modulebase.cpp"
    rm -rf "$OUT/libnoise_src"
fi
echo "built $OUT/libmmnoise.so"
