// placeholder until mode N lands (next commit)
#include "ccal_normal.hpp"
namespace ccal { void normal_ws_destroy(ccal_problem*) {} }
extern "C" {
int ccal_build_normal(ccal_problem*, const double*, const double*, const double*, double, double*, double*, double*) { return CCAL_ERR_UNSUPPORTED; }
int ccal_build_normal_dev(ccal_problem*, double) { return CCAL_ERR_UNSUPPORTED; }
int ccal_solve(ccal_problem*, const ccal_solver_opts*, double*, double*, double*, ccal_report*) { return CCAL_ERR_UNSUPPORTED; }
}
