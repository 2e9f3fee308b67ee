// extern "C" boundary (include/tapstark.h): plain pointers and sizes, status codes, no exceptions
// across the ABI.
#include <dlfcn.h>
#include <signal.h>
#include <spawn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "abi_types.hpp"
#include "jit.hpp"

// hiprtc's time grows faster than the program (a 4.6k-instruction program compiles in 14 s, an
// 18.8k one in 146 s: profiles/r06_air_jit_compile.txt), so the specialisation has a budget:
//   <= TS_JIT_SYNC_INSTR (default 2048, ~3 s)   compiled inside ts_air_compile, as before;
//   <= TS_JIT_MAX_INSTR  (default 32768)        compiled by a child process (ts_jitc): proofs run on the
//                                               interpreter (a GPU path too) until the code object is
//                                               ready, the next use loads it; ts_air_jit_wait joins;
//   larger                                      interpreter only.
// Both kernels compute the same words, so which one ran never shows in a proof.
// A background compilation runs in a CHILD PROCESS (tap-stark_amd/jitc/ts_jitc.cpp, built beside the
// library): hiprtc serialises compilations inside one process, cannot be interrupted, and a thread still
// inside it when the host exits meets the compiler's static destructors.  The child compiles beside the
// prover and beside other children, is killed when its AIR is freed, and leaves nothing behind but a
// code object in a private temporary directory.
static std::string jitc_path() {
    if (const char* e = getenv("TS_JITC_PATH")) return e;
    Dl_info info;
    if (dladdr((void*)&jitc_path, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.rfind('/');
        return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/ts_jitc";
    }
    return "ts_jitc";
}

struct JitChild {
    pid_t pid = -1;
    std::string dir, src, out, log, cache;
    std::chrono::steady_clock::time_point t0;
    void cleanup() {
        for (const std::string& f : {src, out, out + ".part", log})
            if (!f.empty()) (void)unlink(f.c_str());
        if (!dir.empty()) (void)rmdir(dir.c_str());
        dir.clear();
    }
    ~JitChild() {
        if (pid > 0) {  // still compiling for an AIR nobody wants any more
            (void)kill(pid, SIGKILL);
            int st = 0;
            (void)waitpid(pid, &st, 0);
        }
        cleanup();
    }
};

struct ts_air {
    ts::AirProgram prog;
    ts::DevBuf<uint32_t> code;
    std::string jit_log;
    int device = -1;  // the device the jit module was loaded on (-1: host-only AIR)
    enum { JIT_NONE = 0, JIT_COMPILING = 1, JIT_LOADED = 3, JIT_FAILED = 4 };
    int jit_state = JIT_NONE;
    std::unique_ptr<JitChild> job;
    double jit_seconds = 0;
    std::string arch;
    std::mutex poll_m;  // two threads proving with one ts_air: the adoption happens once

    // a code object of this very source left by an earlier process (TS_JIT_CACHE_DIR): no compilation at all
    bool adopt_cached() {
        std::vector<char> code_obj;
        if (!ts::jit_cache_load(ts::jit_cache_path(ts::jit_quotient_source(prog), arch.c_str()), code_obj)) return false;
        ts::JitKernel jk;
        std::string log;
        if (!ts::jit_load_code(code_obj, jk, log)) return false;
        prog.jit_module = jk.module;
        prog.jit_fn = jk.fn;
        jit_state = JIT_LOADED;
        return true;
    }
    void start_background_jit() {
        const std::string helper = jitc_path();
        if (access(helper.c_str(), X_OK) != 0) {
            jit_log = "background specialisation needs the helper " + helper + " (not found): interpreter only";
            return;
        }
        auto j = std::make_unique<JitChild>();
        const char* tmp = getenv("TMPDIR");
        std::string tmpl = std::string(tmp && *tmp ? tmp : "/tmp") + "/ts_jit_XXXXXX";
        std::vector<char> buf(tmpl.begin(), tmpl.end());
        buf.push_back(0);
        if (!mkdtemp(buf.data())) {
            jit_log = "mkdtemp failed: interpreter only";
            return;
        }
        j->dir = buf.data();
        j->src = j->dir + "/quotient_jit.hip";
        j->out = j->dir + "/quotient_jit.co";
        j->log = j->dir + "/log.txt";
        const std::string src = ts::jit_quotient_source(prog);
        j->cache = ts::jit_cache_path(src, arch.c_str());
        FILE* f = fopen(j->src.c_str(), "wb");
        if (!f || fwrite(src.data(), 1, src.size(), f) != src.size()) {
            if (f) fclose(f);
            jit_log = "cannot write the kernel source: interpreter only";
            return;
        }
        fclose(f);
        char* argv[] = {const_cast<char*>(helper.c_str()), const_cast<char*>(arch.c_str()),
                        const_cast<char*>(j->src.c_str()), const_cast<char*>(j->out.c_str()),
                        const_cast<char*>(j->log.c_str()), nullptr};
        // the child is a plain compiler run: nothing preloaded into the host (profilers, sanitizer runtimes)
        // belongs in it
        std::vector<char*> envp;
        for (char** e = environ; e && *e; e++)
            if (strncmp(*e, "LD_PRELOAD=", 11) != 0 && strncmp(*e, "HSA_TOOLS_LIB=", 14) != 0 &&
                strncmp(*e, "ROCP_TOOL_", 10) != 0)
                envp.push_back(*e);
        envp.push_back(nullptr);
        j->t0 = std::chrono::steady_clock::now();
        pid_t pid = -1;
        if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv, envp.data()) != 0) {
            jit_log = "posix_spawn of " + helper + " failed: interpreter only";
            return;
        }
        j->pid = pid;
        job = std::move(j);
        jit_state = JIT_COMPILING;
    }
    // called on the thread that drives the context: adopt a finished compilation
    void poll_jit(bool wait) {
        std::lock_guard<std::mutex> pg(poll_m);
        if (!job) return;
        int st = 0;
        const pid_t r = waitpid(job->pid, &st, wait ? 0 : WNOHANG);
        if (r == 0) return;  // still compiling
        job->pid = -1;
        jit_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - job->t0).count();
        std::vector<char> code_obj;
        bool ok = r > 0 && WIFEXITED(st) && WEXITSTATUS(st) == 0;
        if (ok) {
            if (FILE* f = fopen(job->out.c_str(), "rb")) {
                char buf[1 << 16];
                size_t n;
                while ((n = fread(buf, 1, sizeof buf, f)) > 0) code_obj.insert(code_obj.end(), buf, buf + n);
                fclose(f);
            }
            ok = !code_obj.empty();
            if (ok) ts::jit_cache_store(job->cache, code_obj);
        }
        if (!ok) {
            if (FILE* f = fopen(job->log.c_str(), "rb")) {
                char buf[4096];
                const size_t n = fread(buf, 1, sizeof buf, f);
                jit_log.assign(buf, n);
                fclose(f);
            }
            if (jit_log.empty()) jit_log = "the compiler child ended without a code object";
        }
        ts::JitKernel jk;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (device >= 0 && cur != device) (void)hipSetDevice(device);  // the module belongs to the AIR's device
        const bool loaded = ok && ts::jit_load_code(code_obj, jk, jit_log);
        if (device >= 0 && cur >= 0 && cur != device) (void)hipSetDevice(cur);
        if (loaded) {
            prog.jit_module = jk.module;
            prog.jit_fn = jk.fn;
            jit_state = JIT_LOADED;
        } else {
            jit_state = JIT_FAILED;
        }
        job.reset();
    }
    ~ts_air() {
        job.reset();
        if (device >= 0 && prog.jit_module) (void)hipSetDevice(device);
        ts::JitKernel jk;
        jk.module = prog.jit_module;
        jk.fn = prog.jit_fn;
        ts::jit_release(jk);
    }
};
// the program as the prover sees it, with a background specialisation adopted if it has finished
static const ts::AirProgram& ready_prog(const ts_air* air) {
    const_cast<ts_air*>(air)->poll_jit(false);
    return air->prog;
}
extern char** environ;

struct ts_challenger {
    ts::BfChallenger c;
    ts_challenger(int perm, bool ext) : c(perm, ext) {}
};

namespace {

thread_local std::string g_create_error;

template <class F>
ts_status guard(ts_ctx* ctx, F&& f) {
    try {
        if (ctx) TS_HIP(hipSetDevice(ctx->ctx.device));
        f();
        return TS_OK;
    } catch (const ts::Error& e) {
        if (ctx) ctx->ctx.last_error = e.what();
        else g_create_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        if (ctx) ctx->ctx.last_error = "host allocation failed";
        return TS_ERR_OOM;
    } catch (const std::exception& e) {
        if (ctx) ctx->ctx.last_error = e.what();
        return TS_ERR_INVALID;
    }
}

ts::Ef load_ef(const uint32_t w[4]) {
    for (int i = 0; i < 4; i++)
        TS_REQUIRE(w[i] < ts::P, ts::TS_ERR_INVALID, "non-canonical extension-field element");
    return ts::Ef{{w[0], w[1], w[2], w[3]}};
}

ts::FriConfig load_cfg(const ts_fri_config* cfg) {
    TS_REQUIRE(cfg != nullptr, ts::TS_ERR_INVALID, "null FriConfig");
    TS_REQUIRE(cfg->log_blowup >= 1 && cfg->log_blowup <= 8, ts::TS_ERR_INVALID,
               "FriConfig.log_blowup must be in [1, 8]");
    TS_REQUIRE(cfg->num_queries >= 1 && cfg->num_queries <= 4096, ts::TS_ERR_INVALID,
               "FriConfig.num_queries must be in [1, 4096]");
    TS_REQUIRE(cfg->proof_of_work_bits <= 31, ts::TS_ERR_INVALID, "FriConfig.proof_of_work_bits > 31");
    ts::FriConfig f;
    f.log_blowup = cfg->log_blowup;
    f.num_queries = cfg->num_queries;
    f.proof_of_work_bits = cfg->proof_of_work_bits;
    return f;
}

}  // namespace

extern "C" {

ts_status ts_pcs_verify(const ts_fri_config* cfg, ts_challenger* chal, uint32_t n_rounds,
                        const uint32_t* commitments, const uint32_t* mats_per_round,
                        const uint32_t* log_degrees, const uint32_t* widths, const uint32_t* n_points,
                        const uint32_t* points, const uint32_t* opened_values,
                        const uint32_t* fri_proof, size_t n_words, int* verdict) {
    if (!chal || !commitments || !mats_per_round || !log_degrees || !widths || !n_points || !fri_proof ||
        !verdict || n_rounds == 0)
        return TS_ERR_INVALID;
    *verdict = 9;
    return guard(nullptr, [&] {
        const ts::FriConfig fri = load_cfg(cfg);
        std::vector<ts::PcsRoundClaim> rounds(n_rounds);
        size_t k = 0, pw = 0, ow = 0;
        auto load_checked = [&](const uint32_t* p) {
            for (int j = 0; j < 4; j++) TS_REQUIRE(p[j] < ts::P, ts::TS_ERR_INVALID, "non-canonical element");
            return load_ef(p);
        };
        for (uint32_t r = 0; r < n_rounds; r++) {
            rounds[r].root = commitments + 8 * (size_t)r;
            TS_REQUIRE(mats_per_round[r] >= 1 && mats_per_round[r] <= 16, ts::TS_ERR_INVALID, "mats per round");
            for (uint32_t i = 0; i < mats_per_round[r]; i++, k++) {
                ts::PcsMatClaim m;
                TS_REQUIRE(log_degrees[k] + fri.log_blowup <= 27, ts::TS_ERR_INVALID, "matrix too tall");
                m.log_height = log_degrees[k] + fri.log_blowup;
                m.width = widths[k];
                TS_REQUIRE(n_points[k] == 0 || (points && opened_values), ts::TS_ERR_INVALID, "null points");
                for (uint32_t p = 0; p < n_points[k]; p++, pw += 4) {
                    m.points.push_back(load_checked(points + pw));
                    std::vector<ts::Ef> vals(m.width);
                    for (uint32_t c = 0; c < m.width; c++, ow += 4) vals[c] = load_checked(opened_values + ow);
                    m.values.push_back(std::move(vals));
                }
                rounds[r].mats.push_back(std::move(m));
            }
        }
        *verdict = ts::pcs_verify(fri, chal->c, rounds, fri_proof, n_words);
    });
}

ts_status ts_proof_to_postcard(const uint32_t* proof, size_t n_words, uint8_t* out, size_t cap_bytes,
                               size_t* n_bytes_out) {
    if (!proof || !out || !n_bytes_out) return TS_ERR_INVALID;
    *n_bytes_out = 0;
    return guard(nullptr, [&] {
        std::vector<uint8_t> b;
        TS_REQUIRE(ts::tspf_to_postcard(proof, n_words, b), ts::TS_ERR_INVALID, "not a TSPF v1 proof");
        *n_bytes_out = b.size();
        TS_REQUIRE(b.size() <= cap_bytes, ts::TS_ERR_BUFFER, "postcard buffer too small");
        memcpy(out, b.data(), b.size());
    });
}
ts_status ts_proof_from_postcard(const uint8_t* bytes, size_t n_bytes, uint32_t* proof_out,
                                 size_t cap_words, size_t* n_words_out) {
    return ts_proof_from_postcard_v(bytes, n_bytes, 0, proof_out, cap_words, n_words_out);
}
ts_status ts_proof_from_postcard_v(const uint8_t* bytes, size_t n_bytes, int tspf_version,
                                   uint32_t* proof_out, size_t cap_words, size_t* n_words_out) {
    if (!bytes || !proof_out || !n_words_out || tspf_version < 0 || tspf_version > 2) return TS_ERR_INVALID;
    *n_words_out = 0;
    return guard(nullptr, [&] {
        std::vector<uint32_t> w;
        TS_REQUIRE(ts::postcard_to_tspf(bytes, n_bytes, w, tspf_version), ts::TS_ERR_INVALID,
                   "malformed postcard proof (or not of the TSPF version asked for)");
        *n_words_out = w.size();
        TS_REQUIRE(w.size() <= cap_words, ts::TS_ERR_BUFFER, "proof buffer too small");
        memcpy(proof_out, w.data(), w.size() * 4);
    });
}

uint32_t ts_abi_version(void) { return 5; }  // 5: ts_shard_options (struct_size first, the dead column_sharded_inverse gone), ts_air_program / _jit_*

int ts_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

ts_status ts_ctx_create(int device, ts_ctx** out) {
    if (!out) return TS_ERR_INVALID;
    *out = nullptr;
    return guard(nullptr, [&] { *out = new ts_ctx(device); });
}
void ts_ctx_destroy(ts_ctx* ctx) { delete ctx; }
const char* ts_last_error(const ts_ctx* ctx) {
    return ctx ? ctx->ctx.last_error.c_str() : g_create_error.c_str();
}
ts_status ts_ctx_synchronize(ts_ctx* ctx) {
    if (!ctx) return TS_ERR_INVALID;
    return guard(ctx, [&] { ctx->ctx.sync(); });
}
void* ts_ctx_stream(ts_ctx* ctx) { return ctx ? (void*)ctx->ctx.stream : nullptr; }
ts_status ts_ctx_set_timing(ts_ctx* ctx, int enabled) {
    if (!ctx) return TS_ERR_INVALID;
    ctx->ctx.timing = enabled != 0;
    ctx->ctx.stage_ms.clear();
    return TS_OK;
}
ts_status ts_ctx_take_timings(ts_ctx* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || cap == 0) return TS_ERR_INVALID;
    std::string s;
    for (auto& kv : ctx->ctx.stage_ms) {
        char tmp[160];
        snprintf(tmp, sizeof tmp, "%s=%.6f;", kv.first.c_str(), kv.second);
        s += tmp;
    }
    ctx->ctx.stage_ms.clear();
    if (s.size() + 1 > cap) return TS_ERR_BUFFER;
    memcpy(buf, s.c_str(), s.size() + 1);
    return TS_OK;
}

ts_status ts_ctx_set_replay(ts_ctx* ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        ctx->ctx.sync();
        if (mode == 1) ctx->ctx.replay_log.clear();
        ctx->ctx.replay_pos = 0;
        ctx->ctx.replay_mode = mode;
    });
}
ts_status ts_ctx_set_kernel_timing(ts_ctx* ctx, int enabled) {
    if (!ctx) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        ctx->ctx.take_kernel_timings();  // drop anything pending
        ctx->ctx.kernel_timing = enabled != 0;
    });
}
ts_status ts_ctx_take_kernel_timings(ts_ctx* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || cap == 0) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        std::string s;
        for (auto& kv : ctx->ctx.take_kernel_timings()) {
            char tmp[200];
            snprintf(tmp, sizeof tmp, "%s=%llu:%.6f;", kv.first.c_str(),
                     (unsigned long long)kv.second.first, kv.second.second);
            s += tmp;
        }
        TS_REQUIRE(s.size() + 1 <= cap, ts::TS_ERR_BUFFER, "timing buffer too small");
        memcpy(buf, s.c_str(), s.size() + 1);
    });
}

ts_status ts_ctx_graph_stats(ts_ctx* ctx, uint64_t out[4]) {
    if (!ctx || !out) return TS_ERR_INVALID;
    out[0] = ctx->ctx.fri_graph_replays;
    out[1] = ctx->ctx.fri_graph_fallbacks;
    out[2] = ctx->ctx.fri_graph_sizes.size();
    out[3] = ctx->ctx.bytes_reserved;
    return TS_OK;
}

ts_status ts_ctx_stat(ts_ctx* ctx, int which, uint64_t* out) {
    if (!ctx || !out) return TS_ERR_INVALID;
    switch (which) {
    case 0: *out = ctx->ctx.fri_graph_replays; break;
    case 1: *out = ctx->ctx.fri_graph_fallbacks; break;
    case 2: *out = ctx->ctx.fri_graph_sizes.size(); break;
    case 3: *out = ctx->ctx.bytes_reserved; break;
    case 4: *out = ctx->ctx.fri_graph_reserve_failures; break;
    case 5: *out = ctx->ctx.local_quotient_fallbacks; break;
    case 6: *out = ctx->ctx.pow_hints_accepted; break;
    case 7: *out = ctx->ctx.pow_hints_rejected; break;
    case 8: *out = ctx->ctx.pow_host_grinds; break;
    default: return TS_ERR_INVALID;
    }
    return TS_OK;
}

ts_status ts_bench_alu(ts_ctx* ctx, int kind, double* units_per_second) {
    if (!ctx || !units_per_second) return TS_ERR_INVALID;
    return guard(ctx, [&] { *units_per_second = ts::alu_ceiling(ctx->ctx, kind); });
}

// One stage of the path in a sustained loop on resident, arbitrary data (measurement aid; the values
// are whatever the previous repetition left -- valid lazy-range field elements, never checked):
//   stage 0: coset_lde of a 2^log_n x width matrix (all three NTT passes, every coset)
//   stage 1: BFMmcs::commit's hashing of a 2^(log_n + log_blowup) x width matrix (leaves + tree)
// Returns the mean time of a repetition from HIP events on the context's stream.
ts_status ts_bench_stage(ts_ctx* ctx, int stage, unsigned log_n, uint32_t width, unsigned log_blowup,
                         uint32_t reps, double* ms_per_rep) {
    if (!ctx || !ms_per_rep) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        ts::Context& c = ctx->ctx;
        TS_REQUIRE(stage >= 0 && stage <= 4 && width >= 1 && width <= 256 && reps >= 1 && log_n >= 1 &&
                       log_n + log_blowup <= 27,
                   ts::TS_ERR_INVALID, "bench_stage: stage 0 .. 4, width 1..256, log_n + log_blowup <= 27");
        // stages 2, 3, 4: ONE pass of the LDE alone (inverse contiguous / strided middle / forward
        // contiguous; two-pass shapes only, log_n > 12), on whatever the buffers hold
        struct MaskGuard {
            ts::Context& c;
            ~MaskGuard() { c.lde_pass_mask = 7; }
        } guard_mask{c};
        if (stage >= 2) {
            TS_REQUIRE(log_n > 12, ts::TS_ERR_INVALID, "bench_stage: single LDE passes exist for log_n > 12 only");
            c.lde_pass_mask = 1u << (stage - 2);
        }
        const bool is_lde = stage != 1;
        const uint64_t n = 1ull << log_n, N = n << log_blowup;
        c.ensure_twiddles(log_n + log_blowup);
        ts::DevBuf<uint32_t> lde(&c, (size_t)width * N), in(&c, is_lde ? (size_t)width * n : 1);
        TS_HIP(hipMemsetAsync(lde.p, 0x11, (size_t)width * N * 4, c.stream));  // 0x11111111 < p
        if (is_lde) TS_HIP(hipMemsetAsync(in.p, 0x11, (size_t)width * n * 4, c.stream));
        ts::DevBuf<uint32_t> tree(&c, stage == 1 ? ts::merkle_total_digests(log_n + log_blowup) * 8 : 1);
        std::vector<const uint32_t*> cols(width);
        for (uint32_t k = 0; k < width; k++) cols[k] = lde.p + (uint64_t)k * N;
        ts::DevBuf<const uint32_t*> d_cols(&c, width);
        TS_HIP(hipMemcpyAsync(d_cols.p, cols.data(), width * sizeof(const uint32_t*), hipMemcpyHostToDevice, c.stream));
        c.sync();
        ts::LeafMats lm;
        memset(&lm, 0, sizeof lm);
        lm.n_mats = 1;
        lm.d[0] = lde.p;
        lm.col_stride[0] = N;
        lm.width[0] = width;
        lm.total_width = width;
        lm.cols = d_cols.p;
        auto once = [&] {
            if (is_lde) ts::coset_lde(c, in.p, n, width, log_n, log_blowup, ts::GENERATOR, lde.p, N);
            else ts::launch_commit_tree(c, lm, log_n + log_blowup, tree.p);
        };
        once();  // tables, first-touch
        hipEvent_t e0, e1;
        TS_HIP(hipEventCreate(&e0));
        TS_HIP(hipEventCreate(&e1));
        TS_HIP(hipEventRecord(e0, c.stream));
        for (uint32_t r = 0; r < reps; r++) once();
        TS_HIP(hipEventRecord(e1, c.stream));
        c.sync();
        float ms = 0;
        TS_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_rep = (double)ms / reps;
    });
}

// pinned host memory for traces handed over as host buffers (PCIe at full rate, truly async copies)
ts_status ts_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return TS_ERR_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? TS_OK : TS_ERR_OOM;
}
void ts_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// ------------------------------------------------------------------ matrices
static ts_status matrix_from(ts_ctx* ctx, const uint32_t* src, uint64_t height, uint32_t width,
                             hipMemcpyKind kind, ts_matrix** out, bool sync = true) {
    if (!ctx || !out) return TS_ERR_INVALID;
    *out = nullptr;
    return guard(ctx, [&] {
        TS_REQUIRE(src && height >= 1 && width >= 1, ts::TS_ERR_INVALID, "matrix: empty");
        TS_REQUIRE((height & (height - 1)) == 0, ts::TS_ERR_INVALID, "matrix: height must be a power of two");
        TS_REQUIRE(height <= (1ull << 27), ts::TS_ERR_INVALID, "matrix: height > 2^27");
        auto m = std::make_unique<ts_matrix>();
        m->m.buf = ts::DevBuf<uint32_t>(&ctx->ctx, (size_t)height * width);
        m->m.height = height;
        m->m.width = width;
        m->m.layout = ts::DeviceMatrix::ROW_MAJOR;
        TS_HIP(hipMemcpyAsync(m->m.buf.p, src, (size_t)height * width * 4, kind, ctx->ctx.stream));
        if (sync) ctx->ctx.sync();
        *out = m.release();
    });
}
ts_status ts_matrix_upload(ts_ctx* ctx, const uint32_t* host, uint64_t height, uint32_t width,
                           ts_matrix** out) {
    return matrix_from(ctx, host, height, width, hipMemcpyHostToDevice, out);
}
ts_status ts_matrix_upload_async(ts_ctx* ctx, const uint32_t* host_pinned, uint64_t height,
                                 uint32_t width, ts_matrix** out) {
    return matrix_from(ctx, host_pinned, height, width, hipMemcpyHostToDevice, out, false);
}
ts_status ts_matrix_from_device(ts_ctx* ctx, const uint32_t* dev, uint64_t height, uint32_t width,
                                ts_matrix** out) {
    return matrix_from(ctx, dev, height, width, hipMemcpyDeviceToDevice, out);
}
static ts_status generated_matrix(ts_ctx* ctx, uint64_t height, uint32_t width, ts_matrix** out,
                                  const std::function<void(uint32_t*)>& fill) {
    if (!ctx || !out || height == 0 || width == 0) return TS_ERR_INVALID;
    *out = nullptr;
    return guard(ctx, [&] {
        TS_REQUIRE((height & (height - 1)) == 0, ts::TS_ERR_INVALID, "trace height must be a power of two");
        auto m = std::make_unique<ts_matrix>();
        m->m.buf = ts::DevBuf<uint32_t>(&ctx->ctx, (size_t)height * width);
        m->m.height = height;
        m->m.width = width;
        m->m.layout = ts::DeviceMatrix::ROW_MAJOR;
        fill(m->m.buf.p);
        *out = m.release();
    });
}
ts_status ts_trace_fibonacci(ts_ctx* ctx, uint32_t a, uint32_t b, uint64_t n, ts_matrix** out) {
    return generated_matrix(ctx, n, 2, out,
                            [&](uint32_t* p) { ts::launch_trace_fibonacci(ctx->ctx, p, a, b, n); });
}
ts_status ts_trace_synth_mul(ts_ctx* ctx, uint64_t n, uint32_t width, uint64_t seed, ts_matrix** out) {
    return generated_matrix(ctx, n, width, out,
                            [&](uint32_t* p) { ts::launch_trace_synth_mul(ctx->ctx, p, n, width, seed); });
}
ts_status ts_trace_synth_ext(ts_ctx* ctx, uint64_t n, uint32_t width, uint64_t seed, ts_matrix** out) {
    return generated_matrix(ctx, n, width, out,
                            [&](uint32_t* p) { ts::launch_trace_synth_ext(ctx->ctx, p, n, width, seed); });
}
ts_status ts_matrix_dims(const ts_matrix* m, uint64_t* height, uint32_t* width) {
    if (!m) return TS_ERR_INVALID;
    if (height) *height = m->m.height;
    if (width) *width = m->m.width;
    return TS_OK;
}
ts_status ts_matrix_download(ts_ctx* ctx, const ts_matrix* m, uint32_t* host) {
    if (!ctx || !m || !host) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        TS_REQUIRE(m->m.buf.p, ts::TS_ERR_INVALID, "matrix was consumed");
        const size_t words = (size_t)m->m.height * m->m.width;
        if (m->m.layout == ts::DeviceMatrix::ROW_MAJOR) {
            TS_HIP(hipMemcpyAsync(host, m->m.buf.p, words * 4, hipMemcpyDeviceToHost, ctx->ctx.stream));
            ctx->ctx.sync();
        } else {
            // column-major with bit-reversed rows -> row-major natural
            ts::DevBuf<uint32_t> rm(&ctx->ctx, words);
            ts::launch_transpose_to_row_major(ctx->ctx, m->m.buf.p, m->m.height, rm.p, m->m.height,
                                              m->m.width);
            std::vector<uint32_t> tmp(words);
            TS_HIP(hipMemcpyAsync(tmp.data(), rm.p, words * 4, hipMemcpyDeviceToHost, ctx->ctx.stream));
            ctx->ctx.sync();
            unsigned bits = 0;
            while ((1ull << bits) < m->m.height) bits++;
            for (uint64_t r = 0; r < m->m.height; r++) {
                uint64_t nat = ts::bitrev32((uint32_t)r, bits);
                memcpy(host + nat * m->m.width, tmp.data() + r * m->m.width, (size_t)m->m.width * 4);
            }
        }
    });
}
void ts_matrix_free(ts_ctx* ctx, ts_matrix* m) {
    (void)ctx;
    delete m;
}

// ------------------------------------------------------------------ AIR
ts_status ts_air_compile(ts_ctx* ctx, const uint32_t* tape, size_t n_words, ts_air** out) {
    if (!out) return TS_ERR_INVALID;
    *out = nullptr;
    if (!ctx) {  // host-only AIR (no GPU needed): usable by ts_verify
        return guard(nullptr, [&] {
            auto a = std::make_unique<ts_air>();
            a->prog = ts::compile_air(tape, n_words);
            *out = a.release();
        });
    }
    return guard(ctx, [&] {
        auto a = std::make_unique<ts_air>();
        a->device = ctx->ctx.device;
        a->prog = ts::compile_air(tape, n_words);
        a->code = ts::DevBuf<uint32_t>(&ctx->ctx, std::max<size_t>(a->prog.code.size(), 4));
        if (!a->prog.code.empty())
            TS_HIP(hipMemcpyAsync(a->code.p, a->prog.code.data(), a->prog.code.size() * 4,
                                  hipMemcpyHostToDevice, ctx->ctx.stream));
        ctx->ctx.sync();
        a->prog.d_code = a->code.p;
        // specialise the quotient kernel for this AIR (the interpreter runs it otherwise)
        a->arch = ctx->ctx.arch_name;
        const size_t n_instr = a->prog.code.size() / 4;
        auto env_or = [](const char* name, size_t dflt) {
            const char* v = getenv(name);
            return v && *v ? (size_t)strtoull(v, nullptr, 10) : dflt;
        };
        if (getenv("TS_NO_JIT")) {
            a->jit_log = "disabled by TS_NO_JIT";
        } else if (n_instr <= env_or("TS_JIT_SYNC_INSTR", 2048)) {
            ts::JitKernel jk;
            const auto t0 = std::chrono::steady_clock::now();
            if (ts::jit_compile_quotient(a->prog, a->arch.c_str(), jk, a->jit_log)) {
                a->prog.jit_module = jk.module;
                a->prog.jit_fn = jk.fn;
                a->jit_state = ts_air::JIT_LOADED;
            } else {
                a->jit_state = ts_air::JIT_FAILED;
            }
            a->jit_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } else if (n_instr <= env_or("TS_JIT_MAX_INSTR", 32768)) {
            if (!a->adopt_cached()) a->start_background_jit();
        } else {
            a->jit_log = "program above TS_JIT_MAX_INSTR: interpreter only";
        }
        *out = a.release();
    });
}
int ts_air_is_jit(const ts_air* air) {
    if (!air) return 0;
    if (air->device >= 0 && air->job) {
        (void)hipSetDevice(air->device);
        const_cast<ts_air*>(air)->poll_jit(false);
    }
    return air->prog.jit_fn ? 1 : 0;
}
ts_status ts_air_jit_wait(ts_ctx* ctx, ts_air* air, int* state, double* compile_seconds) {
    if (!ctx || !air) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        air->poll_jit(true);
        if (state) *state = air->jit_state;
        if (compile_seconds) *compile_seconds = air->jit_seconds;
    });
}
ts_status ts_air_info(const ts_air* air, uint32_t* width, uint32_t* n_public,
                      uint32_t* max_constraint_degree, uint32_t* log_quotient_degree) {
    if (!air) return TS_ERR_INVALID;
    if (width) *width = air->prog.width;
    if (n_public) *n_public = air->prog.n_public;
    if (max_constraint_degree) *max_constraint_degree = air->prog.max_degree;
    if (log_quotient_degree) *log_quotient_degree = air->prog.log_quotient_degree;
    return TS_OK;
}
void ts_air_free(ts_ctx* ctx, ts_air* air) {
    (void)ctx;
    delete air;
}
ts_status ts_air_program(const ts_air* air, uint32_t* out, size_t cap_words, size_t* n_words) {
    if (!air || !n_words) return TS_ERR_INVALID;
    const ts::AirProgram& p = air->prog;
    const size_t nc = p.const_canonical.size();
    *n_words = 3 + p.code.size() + 2 * nc;
    if (!out || cap_words < *n_words) return TS_ERR_BUFFER;
    out[0] = p.n_regs;
    out[1] = (uint32_t)(p.code.size() / 4);
    out[2] = (uint32_t)nc;
    std::copy(p.code.begin(), p.code.end(), out + 3);
    std::copy(p.const_canonical.begin(), p.const_canonical.end(), out + 3 + p.code.size());
    std::copy(p.const_public_idx.begin(), p.const_public_idx.end(), out + 3 + p.code.size() + nc);
    return TS_OK;
}
ts_status ts_air_jit_source(const ts_air* air, char* buf, size_t cap, size_t* n_bytes) {
    if (!air || !n_bytes) return TS_ERR_INVALID;
    return guard(nullptr, [&] {
        const std::string src = ts::jit_quotient_source(air->prog);
        *n_bytes = src.size();
        TS_REQUIRE(buf && cap >= src.size(), ts::TS_ERR_BUFFER, "jit source buffer too small");
        memcpy(buf, src.data(), src.size());
    });
}
ts_status ts_air_jit_compile(const ts_air* air, const char* arch, void* code_out, size_t cap, size_t* n_bytes,
                             double* seconds) {
    if (!air || !arch || !n_bytes) return TS_ERR_INVALID;
    return guard(nullptr, [&] {
        std::vector<char> code;
        std::string log;
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = ts::jit_compile_code(air->prog, arch, code, log);
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        TS_REQUIRE(ok, ts::TS_ERR_UNSUPPORTED, ("hiprtc: " + log).c_str());
        *n_bytes = code.size();
        TS_REQUIRE(code_out && cap >= code.size(), ts::TS_ERR_BUFFER, "code object buffer too small");
        memcpy(code_out, code.data(), code.size());
    });
}

// ------------------------------------------------------------------ PCS
ts_status ts_pcs_commit(ts_ctx* ctx, const ts_fri_config* cfg, uint32_t n_mats, ts_matrix* const* evals,
                        const uint32_t* domain_shifts, uint32_t root_out[8], ts_pcs_data** out) {
    if (!ctx || !out || !evals || !domain_shifts) return TS_ERR_INVALID;
    *out = nullptr;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
        TS_REQUIRE(n_mats >= 1 && n_mats <= (uint32_t)ts::MAX_BATCH_MATS, ts::TS_ERR_INVALID,
                   "commit: between 1 and 64 matrices");
        std::vector<ts::DeviceMatrix> ms;
        for (uint32_t i = 0; i < n_mats; i++) {
            TS_REQUIRE(evals[i] && evals[i]->m.buf.p, ts::TS_ERR_INVALID, "commit: null or consumed matrix");
            ms.push_back(std::move(evals[i]->m));
        }
        std::vector<uint32_t> shifts(domain_shifts, domain_shifts + n_mats);
        auto d = std::make_unique<ts_pcs_data>();
        d->d = pcs.commit(ms, shifts);
        if (root_out) memcpy(root_out, d->d->root, 32);
        *out = d.release();
    });
}
ts_status ts_mmcs_commit(ts_ctx* ctx, uint32_t n_mats, ts_matrix* const* mats, uint32_t root_out[8],
                         ts_pcs_data** out) {
    if (!ctx || !out || !mats) return TS_ERR_INVALID;
    *out = nullptr;
    return guard(ctx, [&] {
        TS_REQUIRE(n_mats >= 1 && n_mats <= (uint32_t)ts::MAX_BATCH_MATS, ts::TS_ERR_INVALID,
                   "mmcs commit: between 1 and 64 matrices");
        auto d = std::make_unique<ts_pcs_data>();
        d->d = std::make_unique<ts::PcsData>();
        uint64_t max_h = 0;
        for (uint32_t i = 0; i < n_mats; i++) {
            TS_REQUIRE(mats[i] && mats[i]->m.buf.p, ts::TS_ERR_INVALID, "mmcs commit: null or consumed matrix");
            TS_REQUIRE(mats[i]->m.layout == ts::DeviceMatrix::ROW_MAJOR, ts::TS_ERR_INVALID,
                       "mmcs commit: row-major (uploaded) matrices expected");
            max_h = std::max(max_h, mats[i]->m.height);
        }
        d->d->log_height = ts::log2_strict(max_h);
        for (uint32_t i = 0; i < n_mats; i++) {
            ts::DeviceMatrix& m = mats[i]->m;
            ts::DevBuf<uint32_t> cmaj(&ctx->ctx, (size_t)m.height * m.width);
            ts::launch_transpose_plain(ctx->ctx, m.buf.p, cmaj.p, m.height, m.width, m.height);
            ts::ColMat cm;
            cm.d = cmaj.p;
            cm.height = m.height;
            cm.width = m.width;
            cm.col_stride = m.height;
            d->d->ldes.push_back(cm);
            d->d->lde_storage.push_back(std::move(cmaj));
            m.buf.reset();  // consumed, like the moved RowMajorMatrix arguments
        }
        ts::mmcs_commit(ctx->ctx, *d->d);
        if (root_out) memcpy(root_out, d->d->root, 32);
        *out = d.release();
    });
}
ts_status ts_pcs_data_info(const ts_pcs_data* d, uint32_t* n_mats, uint32_t* log_height) {
    if (!d || !d->d) return TS_ERR_INVALID;
    if (n_mats) *n_mats = (uint32_t)d->d->ldes.size();
    if (log_height) *log_height = d->d->log_height;
    return TS_OK;
}
ts_status ts_pcs_data_lde(ts_ctx* ctx, const ts_pcs_data* d, uint32_t idx, uint32_t* host) {
    if (!ctx || !d || !d->d || !host) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        TS_REQUIRE(idx < d->d->ldes.size(), ts::TS_ERR_INVALID, "lde index out of range");
        const ts::ColMat& cm = d->d->ldes[idx];
        const size_t words = (size_t)cm.height * cm.width;
        ts::DevBuf<uint32_t> rm(&ctx->ctx, words);
        ts::launch_transpose_to_row_major(ctx->ctx, cm.d, cm.col_stride, rm.p, cm.height, cm.width);
        TS_HIP(hipMemcpyAsync(host, rm.p, words * 4, hipMemcpyDeviceToHost, ctx->ctx.stream));
        ctx->ctx.sync();
    });
}
ts_status ts_pcs_data_matrix_info(const ts_pcs_data* d, uint32_t idx, uint64_t* height, uint32_t* width) {
    if (!d || !d->d || idx >= d->d->ldes.size()) return TS_ERR_INVALID;
    if (height) *height = d->d->ldes[idx].height;
    if (width) *width = d->d->ldes[idx].width;
    return TS_OK;
}
ts_status ts_pcs_data_digests(ts_ctx* ctx, const ts_pcs_data* d, uint32_t level, uint32_t* host) {
    if (!ctx || !d || !d->d || !host) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        TS_REQUIRE(level <= d->d->log_height, ts::TS_ERR_INVALID, "digest level out of range");
        const uint64_t off = ts::merkle_level_offset(d->d->log_height, level);
        const uint64_t cnt = 1ull << (d->d->log_height - level);
        TS_HIP(hipMemcpyAsync(host, d->d->tree.p + 8 * off, cnt * 32, hipMemcpyDeviceToHost, ctx->ctx.stream));
        ctx->ctx.sync();
    });
}
ts_status ts_pcs_open_batch(ts_ctx* ctx, const ts_pcs_data* d, uint64_t index, uint32_t* rows_out,
                            uint32_t* path_out) {
    if (!ctx || !d || !d->d || !rows_out || !path_out) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, ts::FriConfig{});
        std::vector<uint32_t> rows, path;
        pcs.open_batch(*d->d, index, rows, path);
        memcpy(rows_out, rows.data(), rows.size() * 4);
        memcpy(path_out, path.data(), path.size() * 4);
    });
}
void ts_pcs_data_free(ts_ctx* ctx, ts_pcs_data* d) {
    (void)ctx;
    delete d;
}

ts_status ts_quotient_chunks(ts_ctx* ctx, const ts_pcs_data* trace_data, uint32_t log_blowup,
                             const ts_air* air, const uint32_t* public_values, uint32_t n_public,
                             const uint32_t alpha[4], ts_matrix** chunks_out) {
    if (!ctx || !trace_data || !trace_data->d || !air || !alpha || !chunks_out) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        ts_fri_config raw{log_blowup, 1, 0};
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(&raw));  // same [1, 8] bound as everywhere else
        std::vector<uint32_t> pis;
        if (n_public) {
            TS_REQUIRE(public_values, ts::TS_ERR_INVALID, "null public values");
            pis.assign(public_values, public_values + n_public);
        }
        auto chunks = pcs.quotient_chunks(*trace_data->d, ready_prog(air), pis, load_ef(alpha));
        for (size_t c = 0; c < chunks.size(); c++) {
            auto m = std::make_unique<ts_matrix>();
            m->m = std::move(chunks[c]);
            chunks_out[c] = m.release();
        }
    });
}

ts_status ts_pcs_open_reduce(ts_ctx* ctx, const ts_fri_config* cfg, const ts_pcs_data* trace_data,
                             const ts_pcs_data* quotient_data, const uint32_t zeta[4],
                             const uint32_t batch_alpha[4], uint32_t* opened_out, uint32_t* reduced_out) {
    if (!ctx || !trace_data || !trace_data->d || !quotient_data || !quotient_data->d || !zeta ||
        !batch_alpha || !opened_out)
        return TS_ERR_INVALID;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
        std::vector<ts::Ef> opened;
        ts::DevBuf<ts::Ef> ro =
            pcs.open_reduce(*trace_data->d, *quotient_data->d, load_ef(zeta), load_ef(batch_alpha), opened);
        memcpy(opened_out, opened.data(), opened.size() * sizeof(ts::Ef));
        if (reduced_out) {
            TS_HIP(hipMemcpyAsync(reduced_out, ro.p, ro.n * sizeof(ts::Ef), hipMemcpyDeviceToHost,
                                  ctx->ctx.stream));
            ctx->ctx.sync();
        }
    });
}

ts_status ts_pcs_open(ts_ctx* ctx, const ts_fri_config* cfg, ts_challenger* chal, uint32_t n_rounds,
                      const ts_pcs_data* const* rounds, const uint32_t* n_points,
                      const uint32_t* points, uint32_t* opened_out, size_t opened_cap_words,
                      size_t* n_opened_words, uint32_t* proof_out, size_t proof_cap_words,
                      size_t* n_proof_words) {
    if (!ctx || !chal || !rounds || !n_points || !opened_out || !n_opened_words || !proof_out ||
        !n_proof_words || n_rounds == 0)
        return TS_ERR_INVALID;
    *n_opened_words = *n_proof_words = 0;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
        std::vector<ts::TwoAdicFriPcs::OpenRound> rs(n_rounds);
        size_t k = 0, pw = 0;
        for (uint32_t r = 0; r < n_rounds; r++) {
            TS_REQUIRE(rounds[r] && rounds[r]->d, ts::TS_ERR_INVALID, "open: null round data");
            rs[r].data = rounds[r]->d.get();
            rs[r].points.resize(rs[r].data->ldes.size());
            for (auto& pl : rs[r].points) {
                const uint32_t np = n_points[k++];
                TS_REQUIRE(np == 0 || points, ts::TS_ERR_INVALID, "open: null points");
                for (uint32_t p = 0; p < np; p++, pw += 4) {
                    for (int j = 0; j < 4; j++)
                        TS_REQUIRE(points[pw + j] < ts::P, ts::TS_ERR_INVALID, "open: non-canonical point");
                    pl.push_back(load_ef(points + pw));
                }
            }
        }
        std::vector<ts::Ef> opened;
        std::vector<uint32_t> proof = pcs.open(rs, chal->c, opened);
        *n_opened_words = opened.size() * 4;
        *n_proof_words = proof.size();
        TS_REQUIRE(opened.size() * 4 <= opened_cap_words, ts::TS_ERR_BUFFER, "opened-values buffer too small");
        TS_REQUIRE(proof.size() <= proof_cap_words, ts::TS_ERR_BUFFER, "proof buffer too small");
        memcpy(opened_out, opened.data(), opened.size() * sizeof(ts::Ef));
        memcpy(proof_out, proof.data(), proof.size() * 4);
    });
}

ts_status ts_fri_prove(ts_ctx* ctx, const ts_fri_config* cfg, ts_challenger* chal, uint32_t n_inputs,
                       const uint32_t* log_lens, const uint32_t* const* inputs, uint32_t* proof_out,
                       size_t cap_words, size_t* n_words_out) {
    if (!ctx || !chal || !log_lens || !inputs || !proof_out || !n_words_out || n_inputs == 0 || n_inputs > 32)
        return TS_ERR_INVALID;
    *n_words_out = 0;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
        std::vector<ts::DevBuf<ts::Ef>> in;
        std::vector<unsigned> logs;
        for (uint32_t k = 0; k < n_inputs; k++) {
            TS_REQUIRE(inputs[k] && log_lens[k] <= 27, ts::TS_ERR_INVALID, "fri_prove: bad input");
            const size_t len = (size_t)1 << log_lens[k];
            for (size_t i = 0; i < 4 * len; i++)
                TS_REQUIRE(inputs[k][i] < ts::P, ts::TS_ERR_INVALID, "fri_prove: non-canonical element");
            ts::DevBuf<ts::Ef> d(&ctx->ctx, len);
            TS_HIP(hipMemcpyAsync(d.p, inputs[k], len * sizeof(ts::Ef), hipMemcpyHostToDevice, ctx->ctx.stream));
            in.push_back(std::move(d));
            logs.push_back(log_lens[k]);
        }
        ctx->ctx.sync();
        std::vector<uint32_t> pf;
        pcs.fri_prove(in, logs, chal->c, {}, pf, /*pass_through=*/true);
        *n_words_out = pf.size();
        TS_REQUIRE(pf.size() <= cap_words, ts::TS_ERR_BUFFER, "proof buffer too small");
        memcpy(proof_out, pf.data(), pf.size() * 4);
    });
}
ts_status ts_fri_verify(const ts_fri_config* cfg, ts_challenger* chal, const uint32_t* proof,
                        size_t n_words, int* verdict) {
    if (!chal || !proof || !verdict) return TS_ERR_INVALID;
    *verdict = 9;
    return guard(nullptr, [&] {
        *verdict = ts::fri_verify_pass_through(load_cfg(cfg), chal->c, proof, n_words);
    });
}

ts_status ts_fri_fold(ts_ctx* ctx, const uint32_t* in, uint64_t h, const uint32_t beta[4], uint32_t* out) {
    if (!ctx || !in || !beta || !out || h == 0) return TS_ERR_INVALID;
    return guard(ctx, [&] {
        ts::DevBuf<ts::Ef> d_in(&ctx->ctx, 2 * h), d_out(&ctx->ctx, h);
        TS_HIP(hipMemcpyAsync(d_in.p, in, 2 * h * 16, hipMemcpyHostToDevice, ctx->ctx.stream));
        ts::launch_fri_fold(ctx->ctx, d_in.p, h, load_ef(beta), d_out.p, nullptr);
        TS_HIP(hipMemcpyAsync(out, d_out.p, h * 16, hipMemcpyDeviceToHost, ctx->ctx.stream));
        ctx->ctx.sync();
    });
}

ts_status ts_fri_fold_device(ts_ctx* ctx, const uint32_t* in_dev, uint64_t h, const uint32_t beta[4],
                             uint32_t* out_dev) {
    if (!ctx || !in_dev || !beta || !out_dev || h == 0) return TS_ERR_INVALID;
    if (((uintptr_t)in_dev | (uintptr_t)out_dev) & 15) return TS_ERR_INVALID;  // EF4 = 16-byte accesses
    return guard(ctx, [&] {
        ts::launch_fri_fold(ctx->ctx, reinterpret_cast<const ts::Ef*>(in_dev), h, load_ef(beta),
                            reinterpret_cast<ts::Ef*>(out_dev), nullptr);
    });
}

// ------------------------------------------------------------------ challenger
ts_status ts_chal_new(int permutation, int sample_ext, ts_challenger** out) {
    if (!out || (permutation != 0 && permutation != 1)) return TS_ERR_INVALID;
    *out = new (std::nothrow) ts_challenger(permutation, sample_ext != 0);
    return *out ? TS_OK : TS_ERR_OOM;
}
ts_status ts_chal_clone(const ts_challenger* c, ts_challenger** out) {
    if (!c || !out) return TS_ERR_INVALID;
    *out = new (std::nothrow) ts_challenger(*c);
    return *out ? TS_OK : TS_ERR_OOM;
}
void ts_chal_free(ts_challenger* c) { delete c; }
// void/value returns leave no room for a status: a null handle (or out pointer) is a no-op
void ts_chal_observe(ts_challenger* c, uint32_t word) {
    if (c) c->c.observe(word);
}
void ts_chal_observe_commitment(ts_challenger* c, const uint32_t d[8]) {
    if (c && d) c->c.observe_commitment(d);
}
void ts_chal_sample(ts_challenger* c, uint32_t out[4]) {
    if (!c || !out) return;
    ts::Ef e = c->c.sample();
    memcpy(out, e.c, 16);
}
uint64_t ts_chal_sample_bits(ts_challenger* c, uint32_t bits) {
    return c && bits <= 32 ? c->c.sample_bits(bits) : 0;
}
int ts_chal_check_witness(ts_challenger* c, uint32_t bits, uint32_t witness) {
    return c && bits <= 32 && c->c.check_witness(bits, witness) ? 1 : 0;
}
ts_status ts_chal_grind(ts_challenger* c, uint32_t bits, uint32_t* witness) {
    if (!c || !witness || bits > 31) return TS_ERR_INVALID;
    try {
        *witness = c->c.grind(bits);
        return TS_OK;
    } catch (const ts::Error& e) {
        return e.code;
    }
}
void ts_chal_state(const ts_challenger* c, uint32_t out[34]) {
    if (c && out) c->c.export_state(out);
}

// ------------------------------------------------------------------ prove
ts_status ts_prove(ts_ctx* ctx, const ts_fri_config* cfg, const ts_air* air, ts_challenger* chal,
                   ts_matrix* trace, const uint32_t* public_values, uint32_t n_public,
                   uint32_t* proof_out, size_t cap_words, size_t* n_words_out) {
    if (!ctx || !air || !chal || !trace || !proof_out || !n_words_out) return TS_ERR_INVALID;
    *n_words_out = 0;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
        TS_REQUIRE(trace->m.buf.p, ts::TS_ERR_INVALID, "prove: trace matrix was already consumed");
        std::vector<uint32_t> pis;
        if (n_public) {
            TS_REQUIRE(public_values, ts::TS_ERR_INVALID, "null public values");
            pis.assign(public_values, public_values + n_public);
        }
        ts::StageTimer t(&ctx->ctx, "prove");
        std::vector<uint32_t> proof = ts::prove(pcs, ready_prog(air), chal->c, std::move(trace->m), pis);
        *n_words_out = proof.size();
        TS_REQUIRE(proof.size() <= cap_words, ts::TS_ERR_BUFFER, "proof buffer too small");
        memcpy(proof_out, proof.data(), proof.size() * 4);
    });
}

// ------------------------------------------------------------------ prove_stream
// Throughput mode as ONE call: n_proofs independent proofs of one AIR on `n_lanes` contexts of a device, one
// host thread per lane INSIDE the call (examples/prove_stream.cpp as an entry point).  Proofs are independent
// objects (uni-stark/src/prover.rs:25-35 takes one trace), so several are kept in flight: one proof's kernels
// fill the gaps the serial transcript of another leaves.  A host whose own threads are cheap (Rust, C++)
// does this itself; a host behind an interpreter lock (the Python binding: bench.py) gets the same loop
// without taking its lock once per proof.
ts_status ts_prove_stream(ts_ctx* const* ctxs, const ts_air* const* airs, uint32_t n_lanes,
                          const ts_fri_config* cfg, ts_matrix* const* traces, const uint32_t* lane_of,
                          uint32_t n_proofs, const uint32_t* public_values, uint32_t n_public, double gate_ms,
                          uint32_t* last_proof_out, size_t cap_words, size_t* n_words_out,
                          double* start_ms_out, double* wall_ms_out) {
    if (!ctxs || !airs || !traces || !lane_of || !n_words_out || n_lanes == 0 || n_lanes > 64) return TS_ERR_INVALID;
    *n_words_out = 0;
    for (uint32_t l = 0; l < n_lanes; l++)
        if (!ctxs[l] || !airs[l]) return TS_ERR_INVALID;
    for (uint32_t i = 0; i < n_proofs; i++)
        if (!traces[i] || lane_of[i] >= n_lanes) return TS_ERR_INVALID;
    if (n_public && !public_values) return TS_ERR_INVALID;
    const std::vector<uint32_t> pis(public_values, public_values + n_public);
    struct Gate {
        std::mutex m;
        double last = -1e300, gap = 0;
    } gate;
    gate.gap = gate_ms > 0 ? gate_ms : 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::vector<ts_status> status(n_lanes, TS_OK);
    std::vector<std::vector<uint32_t>> last_proof(n_lanes);
    std::vector<uint32_t> last_index(n_lanes, 0);
    std::atomic<bool> stop{false};
    auto lane_main = [&](uint32_t l) {
        ts_ctx* ctx = ctxs[l];
        status[l] = guard(ctx, [&] {
            ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
            for (uint32_t i = 0; i < n_proofs && !stop.load(); i++) {
                if (lane_of[i] != l) continue;
                TS_REQUIRE(traces[i]->m.buf.p, ts::TS_ERR_INVALID, "prove_stream: trace matrix was already consumed");
                ts::BfChallenger chal(0, true);  // a fresh challenger per proof, as prove() is handed
                if (gate.gap > 0) {  // no two proofs start within gate_ms of each other
                    std::lock_guard<std::mutex> g(gate.m);
                    for (;;) {
                        const double wait = gate.last + gate.gap - now_ms();
                        if (wait <= 0) break;
                        std::this_thread::sleep_for(std::chrono::microseconds((long)std::min(wait * 1e3, 200.0)));
                    }
                    gate.last = now_ms();
                }
                const double t0 = now_ms();
                std::vector<uint32_t> proof = ts::prove(pcs, ready_prog(airs[l]), chal, std::move(traces[i]->m), pis);
                if (start_ms_out) start_ms_out[i] = t0;
                if (wall_ms_out) wall_ms_out[i] = now_ms() - t0;
                last_proof[l] = std::move(proof);
                last_index[l] = i;
            }
        });
        if (status[l] != TS_OK) stop = true;
    };
    std::vector<std::thread> threads;
    for (uint32_t l = 1; l < n_lanes; l++) threads.emplace_back(lane_main, l);
    lane_main(0);
    for (auto& t : threads) t.join();
    for (uint32_t l = 0; l < n_lanes; l++)
        if (status[l] != TS_OK) return status[l];
    // the proof of the highest index goes back (every proof of a run is checked by the caller's tests, not here)
    uint32_t best = 0;
    bool any = false;
    for (uint32_t l = 0; l < n_lanes; l++)
        if (!last_proof[l].empty() && (!any || last_index[l] > last_index[best])) best = l, any = true;
    if (any && last_proof_out) {
        *n_words_out = last_proof[best].size();
        if (last_proof[best].size() > cap_words) return TS_ERR_BUFFER;
        memcpy(last_proof_out, last_proof[best].data(), last_proof[best].size() * 4);
    }
    return TS_OK;
}

static ts::Comm wrap_comm(const ts_comm& cb) {
    ts::Comm c;
    c.rank = cb.rank;
    c.world = cb.world;
    c.all_gather = [cb](const void* send, void* recv, size_t bytes, hipStream_t stream) {
        if (cb.all_gather(cb.user, send, recv, bytes, (void*)stream) != 0)
            throw ts::Error(ts::TS_ERR_COMM, "all_gather callback failed");
    };
    c.broadcast = [cb](void* buf, size_t bytes, int root, hipStream_t stream) {
        if (cb.broadcast(cb.user, buf, bytes, root, (void*)stream) != 0)
            throw ts::Error(ts::TS_ERR_COMM, "broadcast callback failed");
    };
    return c;
}

ts_status ts_prove_sharded(ts_ctx* ctx, const ts_fri_config* cfg, const ts_comm* comm,
                           const ts_air* air, ts_challenger* chal, ts_matrix* trace_rows,
                           const uint32_t* public_values, uint32_t n_public,
                           const ts_shard_options* options, uint32_t* proof_out, size_t cap_words,
                           size_t* n_words_out) {
    if (!ctx || !comm || !air || !chal || !trace_rows || !proof_out || !n_words_out ||
        !comm->all_gather || !comm->broadcast)
        return TS_ERR_INVALID;
    *n_words_out = 0;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
        TS_REQUIRE(trace_rows->m.buf.p, ts::TS_ERR_INVALID, "prove: trace matrix was already consumed");
        std::vector<uint32_t> pis;
        if (n_public) {
            TS_REQUIRE(public_values, ts::TS_ERR_INVALID, "null public values");
            pis.assign(public_values, public_values + n_public);
        }
        const ts_comm cb = *comm;
        ts::Comm c = wrap_comm(cb);
        ts::ShardOptions opt;
        if (options)
            TS_REQUIRE(options->struct_size == sizeof(ts_shard_options), ts::TS_ERR_INVALID,
                       "ts_shard_options.struct_size != sizeof(ts_shard_options): caller built against another ABI");
        if (options && options->min_local_log) {
            TS_REQUIRE(options->min_local_log <= 27, ts::TS_ERR_INVALID, "min_local_log > 27");
            opt.min_local_log = options->min_local_log;
        }
        if (options) opt.trace_replicated = options->trace_replicated != 0;
        if (options) opt.local_quotient = options->local_quotient != 0;
        ts::StageTimer t(&ctx->ctx, "prove");
        std::vector<uint32_t> proof;
        try {
            proof = ts::prove_sharded(pcs, c, ready_prog(air), chal->c, std::move(trace_rows->m), pis, opt);
        } catch (...) {
            // this rank is leaving the protocol: make the peers' pending collectives fail, not hang
            if (cb.abort) cb.abort(cb.user);
            throw;
        }
        *n_words_out = proof.size();
        TS_REQUIRE(proof.size() <= cap_words, ts::TS_ERR_BUFFER, "proof buffer too small");
        memcpy(proof_out, proof.data(), proof.size() * 4);
    });
}

// ------------------------------------------------------------------ prove / verify over taptrees
ts_status ts_prove_tap(ts_ctx* ctx, const ts_fri_config* cfg, const ts_air* air, ts_challenger* chal,
                       ts_matrix* trace, const uint32_t* public_values, uint32_t n_public,
                       const uint8_t* lock_scripts, const uint64_t* lock_offsets, size_t n_scripts,
                       uint32_t* proof_out, size_t cap_words, size_t* n_words_out) {
    if (!ctx || !air || !chal || !trace || !proof_out || !n_words_out || !lock_scripts || !lock_offsets)
        return TS_ERR_INVALID;
    *n_words_out = 0;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
        TS_REQUIRE(trace->m.buf.p, ts::TS_ERR_INVALID, "prove: trace matrix was already consumed");
        std::vector<uint32_t> pis;
        if (n_public) {
            TS_REQUIRE(public_values, ts::TS_ERR_INVALID, "null public values");
            pis.assign(public_values, public_values + n_public);
        }
        ts::TapLocks locks;
        locks.bytes = lock_scripts;
        locks.offsets = lock_offsets;
        locks.n_scripts = n_scripts;
        ts::StageTimer t(&ctx->ctx, "prove");
        std::vector<uint32_t> proof = ts::prove_tap(pcs, ready_prog(air), chal->c, std::move(trace->m), pis, locks);
        *n_words_out = proof.size();
        TS_REQUIRE(proof.size() <= cap_words, ts::TS_ERR_BUFFER, "proof buffer too small");
        memcpy(proof_out, proof.data(), proof.size() * 4);
    });
}

ts_status ts_prove_tap_sharded(ts_ctx* ctx, const ts_fri_config* cfg, const ts_comm* comm, const ts_air* air,
                               ts_challenger* chal, ts_matrix* trace, const uint32_t* public_values,
                               uint32_t n_public, const uint8_t* lock_scripts, const uint64_t* lock_offsets,
                               size_t n_scripts, uint32_t* proof_out, size_t cap_words, size_t* n_words_out) {
    if (!ctx || !comm || !air || !chal || !trace || !proof_out || !n_words_out || !lock_scripts ||
        !lock_offsets || !comm->all_gather)
        return TS_ERR_INVALID;
    *n_words_out = 0;
    return guard(ctx, [&] {
        ts::TwoAdicFriPcs pcs(ctx->ctx, load_cfg(cfg));
        TS_REQUIRE(trace->m.buf.p, ts::TS_ERR_INVALID, "prove: trace matrix was already consumed");
        std::vector<uint32_t> pis;
        if (n_public) {
            TS_REQUIRE(public_values, ts::TS_ERR_INVALID, "null public values");
            pis.assign(public_values, public_values + n_public);
        }
        ts::TapLocks locks;
        locks.bytes = lock_scripts;
        locks.offsets = lock_offsets;
        locks.n_scripts = n_scripts;
        const ts_comm cb = *comm;
        ts::Comm c = wrap_comm(cb);
        ts::StageTimer t(&ctx->ctx, "prove");
        std::vector<uint32_t> proof;
        try {
            proof = ts::prove_tap(pcs, ready_prog(air), chal->c, std::move(trace->m), pis, locks, &c);
        } catch (...) {
            if (cb.abort) cb.abort(cb.user);  // the peers' pending collectives fail instead of waiting
            throw;
        }
        *n_words_out = proof.size();
        TS_REQUIRE(proof.size() <= cap_words, ts::TS_ERR_BUFFER, "proof buffer too small");
        memcpy(proof_out, proof.data(), proof.size() * 4);
    });
}

ts_status ts_verify_tap(const ts_fri_config* cfg, const ts_air* air, ts_challenger* chal,
                        const uint32_t* proof, size_t n_words, const uint32_t* public_values,
                        uint32_t n_public, const uint8_t* lock_scripts, const uint64_t* lock_offsets,
                        size_t n_scripts, int* verdict) {
    if (!air || !chal || !proof || !verdict || !lock_scripts || !lock_offsets) return TS_ERR_INVALID;
    *verdict = -1;
    return guard(nullptr, [&] {
        ts::FriConfig f = load_cfg(cfg);
        std::vector<uint32_t> pis;
        if (n_public) {
            TS_REQUIRE(public_values, ts::TS_ERR_INVALID, "null public values");
            pis.assign(public_values, public_values + n_public);
        }
        for (size_t i = 0; i < n_scripts; i++)
            TS_REQUIRE(lock_offsets[i + 1] >= lock_offsets[i], ts::TS_ERR_INVALID, "bad lock script offsets");
        ts::TapLocks locks;
        locks.bytes = lock_scripts;
        locks.offsets = lock_offsets;
        locks.n_scripts = n_scripts;
        *verdict = ts::verify_tap(f, air->prog, chal->c, proof, n_words, pis, locks);
    });
}

// ------------------------------------------------------------------ check_constraints
ts_status ts_check_constraints(ts_ctx* ctx, const ts_air* air, const ts_matrix* trace,
                               const uint32_t* public_values, uint32_t n_public,
                               int64_t* first_violation) {
    if (!ctx || !air || !trace || !first_violation) return TS_ERR_INVALID;
    *first_violation = -1;
    return guard(ctx, [&] {
        const ts::AirProgram& p = air->prog;
        TS_REQUIRE(trace->m.buf.p && trace->m.layout == ts::DeviceMatrix::ROW_MAJOR, ts::TS_ERR_INVALID,
                   "check_constraints: needs an uploaded (row-major, unconsumed) trace");
        TS_REQUIRE(trace->m.width == p.width, ts::TS_ERR_INVALID, "check_constraints: width != AIR width");
        TS_REQUIRE(n_public == p.n_public, ts::TS_ERR_INVALID, "check_constraints: public value count");
        TS_REQUIRE(n_public == 0 || public_values, ts::TS_ERR_INVALID, "null public values");
        std::vector<uint32_t> consts(std::max<size_t>(p.const_canonical.size(), 1), 0);
        for (size_t k = 0; k < p.const_canonical.size(); k++) {
            uint32_t v = p.const_public_idx[k] != ~0u ? public_values[p.const_public_idx[k]]
                                                      : p.const_canonical[k];
            TS_REQUIRE(v < ts::P, ts::TS_ERR_INVALID, "non-canonical public value");
            consts[k] = ts::to_mont(v);
        }
        ts::DevBuf<uint32_t> d_consts(&ctx->ctx, consts.size());
        ts::DevBuf<unsigned long long> d_v(&ctx->ctx, 1);
        TS_HIP(hipMemcpyAsync(d_consts.p, consts.data(), consts.size() * 4, hipMemcpyHostToDevice,
                              ctx->ctx.stream));
        TS_HIP(hipMemsetAsync(d_v.p, 0xff, 8, ctx->ctx.stream));
        ts::launch_check_constraints(ctx->ctx, p, trace->m.buf.p, trace->m.height, d_consts.p, d_v.p);
        unsigned long long v = 0;
        TS_HIP(hipMemcpyAsync(&v, d_v.p, 8, hipMemcpyDeviceToHost, ctx->ctx.stream));
        ctx->ctx.sync();
        *first_violation = v == ~0ull ? -1 : (int64_t)v;
    });
}

// ------------------------------------------------------------------ verify
ts_status ts_verify(const ts_fri_config* cfg, const ts_air* air, ts_challenger* chal,
                    const uint32_t* proof, size_t n_words, const uint32_t* public_values,
                    uint32_t n_public, int* verdict) {
    if (!air || !chal || !proof || !verdict) return TS_ERR_INVALID;
    *verdict = -1;
    return guard(nullptr, [&] {
        ts::FriConfig f = load_cfg(cfg);
        std::vector<uint32_t> pis;
        if (n_public) {
            TS_REQUIRE(public_values, ts::TS_ERR_INVALID, "null public values");
            pis.assign(public_values, public_values + n_public);
        }
        *verdict = ts::verify(f, air->prog, chal->c, proof, n_words, pis);
    });
}

}  // extern "C"
