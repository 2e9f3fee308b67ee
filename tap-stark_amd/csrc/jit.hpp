// hiprtc specialisation of the quotient kernel (jit.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "air.hpp"

namespace ts {

struct JitKernel {
    void* module = nullptr;
    void* fn = nullptr;
};

// HIP source of the specialised kernel (exposed for tests / inspection)
std::string jit_quotient_source(const AirProgram& air);
// the gfx code object of that source (hiprtc; needs no GPU): false with the reason / compiler output in `log`
bool jit_compile_code(const AirProgram& air, const char* arch, std::vector<char>& code, std::string& log);
bool jit_compile_source(const std::string& src, const char* arch, std::vector<char>& code, std::string& log);
// TS_JIT_CACHE_DIR: "" when unset; load checks the ELF magic; store writes beside and renames
std::string jit_cache_path(const std::string& src, const char* arch);
bool jit_cache_load(const std::string& path, std::vector<char>& code);
void jit_cache_store(const std::string& path, const std::vector<char>& code);
// loads a code object on the current device
bool jit_load_code(const std::vector<char>& code, JitKernel& out, std::string& log);
// false (with a reason in `log`) if hiprtc is missing, disabled (TS_NO_JIT) or compilation fails
bool jit_compile_quotient(const AirProgram& air, const char* arch, JitKernel& out, std::string& log);
void jit_release(JitKernel& k);

}  // namespace ts
