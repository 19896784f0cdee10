// Mode N (fused normal equations + per-frame Schur complement) and solver-loop workspaces.
#pragma once
#include "ccal_internal.hpp"

namespace ccal {
void normal_ws_destroy(ccal_problem* p);
}  // namespace ccal
