// Run-time specialisation of the quotient kernel for one AIR: the register program produced by
// compile_air (air.cpp) is translated 1:1 into straight-line HIP source and compiled with hiprtc
// for the local GPU, so the constraint evaluation runs out of VGPRs with no instruction decode.
// The AIR is user code in the reference too (a monomorphised `Air::eval`, uni-stark/src/prover.rs:180);
// this is the GPU analogue.  If hiprtc is unavailable the interpreter in quotient.hip is used
// (also a GPU path).  libhiprtc is loaded with dlopen so that the library itself has no hard
// dependency on it.
#include "jit.hpp"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <sstream>

namespace ts {

namespace {

typedef void* rtcProgram;
struct Rtc {
    void* lib = nullptr;
    int (*create)(rtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*compile)(rtcProgram, int, const char**) = nullptr;
    int (*log_size)(rtcProgram, size_t*) = nullptr;
    int (*get_log)(rtcProgram, char*) = nullptr;
    int (*code_size)(rtcProgram, size_t*) = nullptr;
    int (*get_code)(rtcProgram, char*) = nullptr;
    int (*destroy)(rtcProgram*) = nullptr;
    bool ok = false;
};

Rtc load_rtc() {
    Rtc r;
    for (const char* name : {"libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return r;
    r.create = (decltype(r.create))dlsym(r.lib, "hiprtcCreateProgram");
    r.compile = (decltype(r.compile))dlsym(r.lib, "hiprtcCompileProgram");
    r.log_size = (decltype(r.log_size))dlsym(r.lib, "hiprtcGetProgramLogSize");
    r.get_log = (decltype(r.get_log))dlsym(r.lib, "hiprtcGetProgramLog");
    r.code_size = (decltype(r.code_size))dlsym(r.lib, "hiprtcGetCodeSize");
    r.get_code = (decltype(r.get_code))dlsym(r.lib, "hiprtcGetCode");
    r.destroy = (decltype(r.destroy))dlsym(r.lib, "hiprtcDestroyProgram");
    r.ok = r.create && r.compile && r.log_size && r.get_log && r.code_size && r.get_code && r.destroy;
    return r;
}

Rtc& rtc() {
    static Rtc r = load_rtc();  // function-local static: initialised once, thread-safe
    return r;
}

const char* kPrelude = R"SRC(
typedef unsigned int u32;
typedef unsigned long long u64;
#define P 0x78000001u
__device__ __forceinline__ u32 umin32(u32 a, u32 b) { return a < b ? a : b; }
__device__ __forceinline__ u32 add(u32 a, u32 b) { u32 s = a + b; return umin32(s, s - P); }
__device__ __forceinline__ u32 sub(u32 a, u32 b) { u32 d = a - b; return umin32(d, d + P); }
__device__ __forceinline__ u32 neg(u32 a) { return a ? P - a : 0u; }
// additive form, as bb.hpp: m = -t p^-1 mod 2^32, (t + m p) / 2^32 < 2p  (mul_lo, mad_u64, sub, min)
__device__ __forceinline__ u32 mont_reduce(u64 t) {
    u32 m = (u32)t * 0x77ffffffu;
    u32 r = (u32)((t + (u64)m * P) >> 32);
    return umin32(r, r - P);
}
__device__ __forceinline__ u32 mont_mul(u32 a, u32 b) { return mont_reduce((u64)a * b); }
__device__ __forceinline__ u32 to_mont(u32 a) { return mont_mul(a, 0x45dddde3u); }
typedef u32 v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64 lazy_fix(u64 acc) {  // high word corrected in place (bb.hpp lazy_fix)
    v2u u = __builtin_bit_cast(v2u, acc);
    u.y = umin32(u.y, u.y - P);
    return __builtin_bit_cast(u64, u);
}
__device__ __forceinline__ u32 lazy_finish(u64 acc) { return mont_reduce(lazy_fix(acc)); }
struct QC { u32 inv_zh[64]; };  // = QuotConsts
struct QO { u32* chunk[64]; };  // = QuotOut, MAX_QUOTIENT_CHUNKS
extern "C" __global__ void __launch_bounds__(256)
k_quotient_jit(const u32* __restrict__ lde, u64 col_stride, unsigned log_n, unsigned log_qd,
               const u32* __restrict__ C, const u32* __restrict__ AP, const u32* __restrict__ isf,
               const u32* __restrict__ isl, const u32* __restrict__ ist, QC qc, QO out,
               u32 row_begin, u32 row_end) {
    const unsigned L = log_n + log_qd;
    const u32 total = 1u << L;
    const u32 r = row_begin + blockIdx.x * 256u + threadIdx.x;
    if (r >= row_end) return;
    const u32 i = L ? (__brev(r) >> (32 - L)) : 0u;
    const u32 i_next = (i + (1u << log_qd)) & (total - 1u);
    const u32 r_next = L ? (__brev(i_next) >> (32 - L)) : 0u;
    const u32* __restrict__ row0 = lde + r;
    const u32* __restrict__ row1 = lde + r_next;
    const u32 sel0 = isf[r], sel1 = isl[r], sel2 = ist[r];
    u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
)SRC";

const char* kEpilogue = R"SRC(
    a0 = lazy_fix(a0); a1 = lazy_fix(a1); a2 = lazy_fix(a2); a3 = lazy_fix(a3);
    const u32 c = log_qd ? (__brev(r >> log_n) >> (32 - log_qd)) : 0u;
    const u32 iz = qc.inv_zh[c];
    const u64 n = 1ull << log_n;
    u32* o = out.chunk[c] + (r & (n - 1));
    o[0] = mont_mul(lazy_finish(a0), iz);
    o[n] = mont_mul(lazy_finish(a1), iz);
    o[2 * n] = mont_mul(lazy_finish(a2), iz);
    o[3 * n] = mont_mul(lazy_finish(a3), iz);
}
)SRC";

}  // namespace

std::string jit_quotient_source(const AirProgram& air) {
    std::ostringstream s;
    s << kPrelude;
    for (uint32_t r = 0; r < air.n_regs; r++) s << "    u32 r" << r << " = 0;\n";
    const size_t n_instr = air.code.size() / 4;
    uint32_t n_assert = 0;
    for (size_t pc = 0; pc < n_instr; pc++) {
        const uint32_t op = air.code[4 * pc], dst = air.code[4 * pc + 1], a = air.code[4 * pc + 2],
                       b = air.code[4 * pc + 3];
        switch (op) {
            case D_LOAD:
                s << "    r" << dst << " = to_mont(row" << a << "[" << b << "ull * col_stride]);\n";
                break;
            case D_CONST: s << "    r" << dst << " = C[" << a << "];\n"; break;
            case D_SEL: s << "    r" << dst << " = sel" << a << ";\n"; break;
            case D_ADD: s << "    r" << dst << " = add(r" << a << ", r" << b << ");\n"; break;
            case D_SUB: s << "    r" << dst << " = sub(r" << a << ", r" << b << ");\n"; break;
            case D_NEG: s << "    r" << dst << " = neg(r" << a << ");\n"; break;
            case D_MUL: s << "    r" << dst << " = mont_mul(r" << a << ", r" << b << ");\n"; break;
            default:  // D_ASSERT: acc += reg[a] * alpha_pow[b]
                s << "    a0 += (u64)r" << a << " * AP[" << 4 * b << "]; a1 += (u64)r" << a << " * AP["
                  << 4 * b + 1 << "]; a2 += (u64)r" << a << " * AP[" << 4 * b + 2 << "]; a3 += (u64)r"
                  << a << " * AP[" << 4 * b + 3 << "];\n";
                if (++n_assert % 2 == 0)
                    s << "    a0 = lazy_fix(a0); a1 = lazy_fix(a1); a2 = lazy_fix(a2); a3 = lazy_fix(a3);\n";
                break;
        }
    }
    s << kEpilogue;
    return s.str();
}

// ---- optional on-disk cache of code objects (TS_JIT_CACHE_DIR): the reference pays for `Air::eval` once, at
// build time; a prover process that restarts should not pay hiprtc again for an AIR it has compiled before.
// Key: 128 bits of FNV-1a over (generator version, hiprtc version, arch, source).
static const char* kGeneratorVersion = "tapstark-jit-1";

std::string jit_cache_path(const std::string& src, const char* arch) {
    const char* dir = getenv("TS_JIT_CACHE_DIR");
    if (!dir || !*dir) return "";
    int major = 0, minor = 0;
    if (void* lib = rtc().lib)
        if (auto ver = (int (*)(int*, int*))dlsym(lib, "hiprtcVersion")) (void)ver(&major, &minor);
    uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i < n; i++) {
            h1 = (h1 ^ b[i]) * 0x100000001b3ull;
            h2 = (h2 ^ (b[i] + 0x9e)) * 0x100000001b3ull;
            h2 ^= h2 >> 29;
        }
    };
    mix(kGeneratorVersion, strlen(kGeneratorVersion));
    mix(&major, sizeof major);
    mix(&minor, sizeof minor);
    mix(arch, strlen(arch));
    mix(src.data(), src.size());
    char name[96];
    snprintf(name, sizeof name, "/q_%016llx%016llx_%s.co", (unsigned long long)h1, (unsigned long long)h2, arch);
    return std::string(dir) + name;
}
bool jit_cache_load(const std::string& path, std::vector<char>& code) {
    if (path.empty()) return false;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    code.clear();
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + n);
    fclose(f);
    return code.size() > 64 && memcmp(code.data(), "\177ELF", 4) == 0;
}
void jit_cache_store(const std::string& path, const std::vector<char>& code) {
    if (path.empty() || code.empty()) return;
    const std::string tmp = path + ".part" + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;  // the directory must exist; a cache that cannot be written is simply not used
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}

bool jit_compile_code(const AirProgram& air, const char* arch, std::vector<char>& code, std::string& log) {
    return jit_compile_source(jit_quotient_source(air), arch, code, log);
}

bool jit_compile_source(const std::string& src, const char* arch, std::vector<char>& code, std::string& log) {
    Rtc& r = rtc();
    if (!r.ok) {
        log = "libhiprtc not available";
        return false;
    }
    if (const char* dump = getenv("TS_JIT_DUMP")) {  // the generated source, for offline inspection (hipcc -S)
        if (FILE* f = fopen(dump, "w")) {
            fwrite(src.data(), 1, src.size(), f);
            fclose(f);
        }
    }
    const std::string cached = jit_cache_path(src, arch);
    if (jit_cache_load(cached, code)) {
        log = "code object from " + cached;
        return true;
    }
    rtcProgram prog = nullptr;
    if (r.create(&prog, src.c_str(), "quotient_jit.hip", 0, nullptr, nullptr) != 0) {
        log = "hiprtcCreateProgram failed";
        return false;
    }
    std::string arch_opt = std::string("--offload-arch=") + arch;
    const char* opts[] = {arch_opt.c_str(), "-O3"};
    const int rc = r.compile(prog, 2, opts);
    size_t sz = 0;
    r.log_size(prog, &sz);
    if (sz > 1) {
        log.resize(sz);
        r.get_log(prog, &log[0]);
    }
    if (rc != 0) {
        r.destroy(&prog);
        return false;
    }
    r.code_size(prog, &sz);
    code.resize(sz);
    r.get_code(prog, code.data());
    r.destroy(&prog);
    jit_cache_store(cached, code);
    return true;
}

bool jit_compile_quotient(const AirProgram& air, const char* arch, JitKernel& out, std::string& log) {
    if (getenv("TS_NO_JIT")) {
        log = "disabled by TS_NO_JIT";
        return false;
    }
    std::vector<char> code;
    if (!jit_compile_code(air, arch, code, log)) return false;
    return jit_load_code(code, out, log);
}

bool jit_load_code(const std::vector<char>& code, JitKernel& out, std::string& log) {
    hipModule_t mod = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
        log += " hipModuleLoadData failed";
        return false;
    }
    hipFunction_t fn = nullptr;
    if (hipModuleGetFunction(&fn, mod, "k_quotient_jit") != hipSuccess) {
        (void)hipModuleUnload(mod);
        log += " hipModuleGetFunction failed";
        return false;
    }
    out.module = mod;
    out.fn = fn;
    return true;
}

void jit_release(JitKernel& k) {
    if (k.module) (void)hipModuleUnload((hipModule_t)k.module);
    k.module = nullptr;
    k.fn = nullptr;
}

}  // namespace ts
