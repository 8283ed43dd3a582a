// IR passes; see passes.cpp.
#pragma once
#include <vector>

#include "ir.h"

namespace mm {
bool copy_propagate(FilterCode &code);
bool eliminate_dead_code(FilterCode &code);
bool eliminate_dead_cycles(FilterCode &code);   // what only feeds itself around a loop (closure render code: passes.cpp)
bool loop_carried_cse(FilterCode &code);
bool common_subexpressions(FilterCode &code);
void optimize(FilterCode &code);
void analyze_frame_constants(FilterCode &code);
void specialize_constants(FilterCode &code);   // specialize.cpp
// Folds `op` applied to literal arguments with the C semantics of its macro; false when the
// op is not one of the foldable arithmetic / comparison ops.  specialize.cpp
bool fold_constant_op(const OpInfo *op, const std::vector<Primary> &args, Primary &out);
}  // namespace mm
