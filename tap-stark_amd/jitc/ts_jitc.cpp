// ts_jitc -- out-of-process hiprtc compilation for libtapstark_hip.so's background specialisation of
// large AIRs (csrc/abi.cpp).  hiprtc serialises compilations inside one process and cannot be
// interrupted; a child process compiles beside the prover (and beside other children), can be killed
// when its AIR is freed, and takes the compiler's global state with it when it ends.  It never touches
// the GPU (hiprtc only).
//
//     ts_jitc <arch> <source.hip> <out.co> <log.txt>        exit 0 = code object written
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

typedef void* rtcProgram;

static bool read_file(const char* path, std::string& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}
static void write_file(const char* path, const void* p, size_t n) {
    if (FILE* f = fopen(path, "wb")) {
        fwrite(p, 1, n, f);
        fclose(f);
    }
}

int main(int argc, char** argv) {
    if (argc != 5) {
        fprintf(stderr, "usage: ts_jitc <arch> <source.hip> <out.co> <log.txt>\n");
        return 2;
    }
    const char *arch = argv[1], *src_path = argv[2], *out_path = argv[3], *log_path = argv[4];
    std::string src;
    if (!read_file(src_path, src)) {
        write_file(log_path, "cannot read the source", 22);
        return 1;
    }
    void* lib = nullptr;
    for (const char* name : {"libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"})
        if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) {
        write_file(log_path, "libhiprtc not available", 23);
        return 1;
    }
    auto create = (int (*)(rtcProgram*, const char*, const char*, int, const char**, const char**))dlsym(lib, "hiprtcCreateProgram");
    auto compile = (int (*)(rtcProgram, int, const char**))dlsym(lib, "hiprtcCompileProgram");
    auto log_size = (int (*)(rtcProgram, size_t*))dlsym(lib, "hiprtcGetProgramLogSize");
    auto get_log = (int (*)(rtcProgram, char*))dlsym(lib, "hiprtcGetProgramLog");
    auto code_size = (int (*)(rtcProgram, size_t*))dlsym(lib, "hiprtcGetCodeSize");
    auto get_code = (int (*)(rtcProgram, char*))dlsym(lib, "hiprtcGetCode");
    if (!create || !compile || !log_size || !get_log || !code_size || !get_code) {
        write_file(log_path, "libhiprtc lacks an entry point", 30);
        return 1;
    }
    rtcProgram prog = nullptr;
    if (create(&prog, src.c_str(), "quotient_jit.hip", 0, nullptr, nullptr) != 0) {
        write_file(log_path, "hiprtcCreateProgram failed", 26);
        return 1;
    }
    const std::string arch_opt = std::string("--offload-arch=") + arch;
    const char* opts[] = {arch_opt.c_str(), "-O3"};
    const int rc = compile(prog, 2, opts);
    size_t sz = 0;
    log_size(prog, &sz);
    if (sz > 1) {
        std::vector<char> log(sz);
        get_log(prog, log.data());
        write_file(log_path, log.data(), sz - 1);
    }
    if (rc != 0) return 1;
    code_size(prog, &sz);
    std::vector<char> code(sz);
    get_code(prog, code.data());
    // write beside, then rename: the reader never sees a partial file
    const std::string tmp = std::string(out_path) + ".part";
    write_file(tmp.c_str(), code.data(), sz);
    return rename(tmp.c_str(), out_path) == 0 ? 0 : 1;
    // (no hiprtcDestroyProgram / dlclose: the process ends here)
}
