// ykh_handles.hpp -- definitions of the opaque C handles (include/yask_hip_c_api.h).
#pragma once
#include <memory>
#include <string>
#include "ykh_runtime.hpp"

struct yk_env_s { std::shared_ptr<ykh::Env> env; };
struct yk_solution_s {
    std::shared_ptr<ykh::Solution> soln;
    std::string help, values, variant;
};
// yk_var_h is a borrowed ykh::Var*.
