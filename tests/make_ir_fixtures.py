#!/usr/bin/env python3
"""Generates tests/golden/ir/*.json.gz: the IR (mmhip_filter_ir_json) of every filter of the
reference's test-suite (tests/run_tests.sh) that this front-end compiles, plus a manifest with
the user values and the golden PNG each case is compared with.  Run where /root/reference is
available; the GPU box, which has no reference tree, replays the fixtures through
mmhip_compile_ir_json (tests/test_gpu_parity.py::test_reference_suite_on_gpu).

The fixtures are compiler output (structured SSA in this project's own JSON form), i.e.
derived data like an object file -- not the text of the reference's scripts.
"""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mathmap_amd as mm  # noqa: E402
from tests.test_cpu_suite import _run_tests_cases  # noqa: E402

REF = "/root/reference/tests"
OUT = os.path.join(ROOT, "tests", "golden", "ir")


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = []
    for script, golden, uv, needs in _run_tests_cases():
        stem = os.path.splitext(golden)[0]
        try:
            flt = mm.Filter(open(os.path.join(REF, script)).read())
            flt.jit(load=False)      # must at least compile for gfx950
        except mm.MathMapError as e:
            print("skip %-40s %s" % (stem, str(e).splitlines()[0][:70]))
            continue
        with open(os.path.join(OUT, stem + ".json.gz"), "wb") as raw:      # mtime=0: reproducible bytes
            with gzip.GzipFile(fileobj=raw, mode="wb", compresslevel=9, mtime=0, filename="") as f:
                f.write(flt.ir_json_raw.encode())
        manifest.append({"ir": stem + ".json.gz", "golden": golden, "uservals": uv, "needs_image": needs})
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print("%d fixtures" % len(manifest))
    make_example_fixtures()


def make_example_fixtures():
    """The same for every filter under the reference's examples/ (189): IR only, default user
    values; tests/test_gpu_parity.py::test_reference_examples_on_gpu renders them on the GPU and
    with the oracle."""
    import glob
    out = os.path.join(ROOT, "tests", "golden", "ir_examples")
    os.makedirs(out, exist_ok=True)
    names = []
    for path in sorted(glob.glob("/root/reference/examples/**/*.mm", recursive=True)):
        rel = os.path.relpath(path, "/root/reference/examples")
        stem = rel[:-3].replace("/", "__").replace(" ", "_")
        flt = mm.Filter(open(path, errors="replace").read())
        with open(os.path.join(out, stem + ".json.gz"), "wb") as raw:
            with gzip.GzipFile(fileobj=raw, mode="wb", compresslevel=9, mtime=0, filename="") as f:
                f.write(flt.ir_json_raw.encode())
        names.append(stem)
    json.dump(names, open(os.path.join(out, "manifest.json"), "w"), indent=1)
    print("%d example fixtures" % len(names))


if __name__ == "__main__":
    main()
