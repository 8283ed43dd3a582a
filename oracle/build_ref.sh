#!/bin/bash
# CPU ORACLE support: builds the pieces of the reference that compile from their own
# sources (nothing else of the reference builds here: clisp, bison and the glib/GSL/GIMP
# headers are missing).  Sources are compiled where they lie under /root/reference;
# outputs go only to oracle/_ref/ (git-ignored, shipped to the GPU box).
#   * builtins/spec_func.c  -> _ref/libspec_func.so   (cgamma; validates oracle + device cgamma)
set -e
REF=/root/reference
OUT="$(dirname "$0")/_ref"
[ -d "$REF" ] || { echo "no reference tree"; exit 0; }
mkdir -p "$OUT"
gcc -O2 -fPIC -shared -std=gnu99 -o "$OUT/libspec_func.so" "$REF/builtins/spec_func.c" -lm
echo "built $OUT/libspec_func.so"
