"""mathmap_amd -- MI355X-native per-pixel expression evaluator for MathMap filters.

Package layout: ``csrc/`` (C++ compiler front-end, IR, HIP code generator, hiprtc
runtime, hand-written HIP kernels behind the C ABI of include/mmhip.h),
``api.py`` (host-side mirror of the reference session interface), ``striping.py``
(row stripes across GPUs).  The benchmark and test filters live outside the package
(``tests/filters.py``: the reference's filters as compiled IR fixtures, project-written probes as text).
"""
from .api import (EDGE_COLOR, EDGE_REFLECT, EDGE_ROTATE, EDGE_WRAP, Filter, Invocation, MathMapError,  # noqa: F401
                  device_count, set_device)
