// hiprtc specialisation of the quotient kernel (jit.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "air.hpp"

namespace ts {

struct JitKernel {
    void* module = nullptr;
    void* fn = nullptr;
};

// HIP source of the specialised kernel (exposed for tests / inspection)
std::string jit_quotient_source(const AirProgram& air);
// false (with a reason in `log`) if hiprtc is missing, disabled (TS_NO_JIT) or compilation fails
bool jit_compile_quotient(const AirProgram& air, const char* arch, JitKernel& out, std::string& log);
void jit_release(JitKernel& k);

}  // namespace ts
