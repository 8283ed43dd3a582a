// IR passes; see passes.cpp.
#pragma once
#include "ir.h"

namespace mm {
bool copy_propagate(FilterCode &code);
bool eliminate_dead_code(FilterCode &code);
void optimize(FilterCode &code);
void analyze_frame_constants(FilterCode &code);
void specialize_constants(FilterCode &code);   // specialize.cpp
}  // namespace mm
