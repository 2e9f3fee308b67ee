// Per-GPU context of the tap-stark HIP library: stream, caching device allocator, twiddle tables,
// error reporting.  One context is driven by one host thread (SURVEY.md section 8(b) Threading).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "bb.hpp"

namespace ts {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// fri/src/prover.rs:129-134 `assert_eq!(x, final_poly)`: its own type so that a caller that can take
// another route (prove_sharded's local quotient) can tell it from every other invariant
struct FinalPolyNotConstant : Error {
    explicit FinalPolyNotConstant(const std::string& m) : Error(5 /* TS_ERR_INVARIANT */, m) {}
};

// status codes of the C ABI (include/tapstark.h)
enum : int {
    TS_OK = 0,
    TS_ERR_INVALID = 1,    // bad argument / shape
    TS_ERR_HIP = 2,        // HIP runtime failure
    TS_ERR_OOM = 3,        // device allocation failed
    TS_ERR_UNSUPPORTED = 4,
    TS_ERR_INVARIANT = 5,  // the reference would have panicked (assert/expect)
    TS_ERR_BUFFER = 6,     // output buffer too small
    TS_ERR_COMM = 7,       // a communicator callback failed
};

#define TS_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            throw ts::Error(ts::TS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define TS_REQUIRE(cond, code, msg)                   \
    do {                                              \
        if (!(cond)) throw ts::Error((code), (msg));  \
    } while (0)

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    int num_cus = 256;
    size_t max_lds_per_block = 64 * 1024;
    std::string arch_name;  // "gfx950"

    // caching allocator: blocks are rounded up and recycled by exact rounded size
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live_blocks;
    size_t bytes_reserved = 0;

    // twiddle tables in bit-reversed block order (kernels_ntt.hip): W[m + i] = w_{2m}^bitrev(i),
    // Montgomery form; index 0 unused.  Grown on demand, never shrunk.
    uint32_t* d_twiddle_fwd = nullptr;
    uint32_t* d_twiddle_inv = nullptr;
    unsigned twiddle_log = 0;

    // coset scale tables of the LDE (ntt_lde.hip): T[beta][k] = s_beta^k / n for one (log_n,
    // log_blowup, shift); a trace commit asks for the same one every proof, so a few are kept
    struct ScaleTable {
        unsigned log_n, log_blowup;
        uint32_t shift;
        uint32_t* d;
        size_t words;
        uint64_t last_use;
    };
    std::vector<ScaleTable> scale_tables;
    uint64_t scale_clock = 0;
    // the last selector table (is_first | is_last | is_transition on the quotient domain,
    // quotient.hip): a function of (log_n, log_qd) only, so repeated proofs of one shape reuse it
    // selector tables (quotient.hip) by (log_n, log_qd, domain shift): two entries, so that a context
    // that alternates between the reference shift (prove, the broadcast path) and a rank's own
    // (local quotient) rebuilds neither -- a rebuild is a stream sync, a hipFree and a hipMalloc
    struct SelTable {
        uint32_t* d = nullptr;
        unsigned log_n = ~0u, log_qd = ~0u;
        uint32_t shift = 0;
        uint64_t last_use = 0;
    };
    SelTable sel_tables[2];
    uint64_t sel_clock = 0;

    // zero-initialised words (17 lines of 64 bytes) the "last workgroup done" kernels count in
    // (merkle.hip); each kernel leaves them at zero, and launches on the one stream run in order
    uint32_t* d_ticket = nullptr;
    uint32_t* ticket();

    // measurement only (ts_bench_stage): which of the three LDE passes coset_lde launches (bit 0 inverse
    // contiguous, 1 strided middle, 2 forward contiguous); 7 everywhere else
    unsigned lde_pass_mask = 7;

    // Mailbox: a page of page-locked host memory that kernels write small results into directly (a
    // Merkle root, the opened-value sums): the host reads them after a stream sync, with no copy
    // kernel in between (a D2H hipMemcpyAsync of 32 bytes is a launch of its own: ~7 us of a proof's
    // critical path each time).  Slots are handed out round-robin; a slot is only reused after the
    // sync that read it.
    uint32_t* h_mailbox = nullptr;
    size_t mailbox_off = 0;
    uint32_t* h_mailbox_big = nullptr;  // messages above a quarter of the page: one at a time
    size_t h_mailbox_big_words = 0;
    uint32_t* mailbox(size_t words);

    // pinned host staging
    void* h_pinned = nullptr;
    size_t h_pinned_bytes = 0;
    // page-locked bump arena for small host->device uploads (challenge powers, index lists,
    // pointer tables): the bytes are copied here first, so the async copy never reads a caller's
    // stack or vector and no sync is needed to protect them.  When full: one sync, then reuse.
    char* h_arena = nullptr;
    size_t h_arena_bytes = 0, h_arena_off = 0;
    const void* stage(const void* src, size_t bytes);

    // optional per-stage timing (bench.py): name -> accumulated ms, measured with HIP events on
    // `stream`
    bool timing = false;
    std::vector<std::pair<std::string, float>> stage_ms;

    // optional per-kernel timing: HIP events recorded on `stream` around every launch, resolved
    // lazily (no sync per kernel) by take_kernel_timings()
    struct KernelEvent {
        const char* name;
        hipEvent_t e0, e1;
    };
    bool kernel_timing = false;
    std::vector<KernelEvent> kernel_events;
    // name -> (launch count, total ms)
    std::map<std::string, std::pair<uint64_t, double>> take_kernel_timings();

    // TS_FRI_GRAPH (prover.cpp): the instantiated graph of the FRI commit phase, updated per proof.
    // A stream capture must not reach hipMalloc / hipFree / a stream sync, so while `capturing`:
    // alloc() only serves from the free list and throws CaptureMiss otherwise (no HIP call made);
    // free() parks the block in `deferred_free` (it stays live: nothing captured has run yet, and a
    // fallback to the eager path may need the buffer's contents); stage() throws CaptureMiss instead
    // of wrapping its arena.  `alloc_log`, when set, records every rounded size alloc() hands out:
    // the first (eager) proof of a shape records what the commit phase needs, reserve() then makes
    // the free list hold all of it at once before a capture starts.
    struct CaptureMiss {};
    hipGraphExec_t fri_graph_exec = nullptr;
    bool capturing = false;
    std::vector<void*> deferred_free;
    std::vector<size_t>* alloc_log = nullptr;
    std::map<std::vector<uint32_t>, std::vector<size_t>> fri_graph_sizes;  // shape key -> block sizes
    uint64_t fri_graph_replays = 0, fri_graph_fallbacks = 0;
    uint64_t fri_graph_reserve_failures = 0;  // Context::reserve refused: the phase ran eagerly
    // fri_pow_witness (prover.cpp): witnesses taken from the device search after the host's one-step check,
    // device candidates the host refused (never expected), and searches run on the host
    uint64_t pow_hints_accepted = 0, pow_hints_rejected = 0, pow_host_grinds = 0;
    uint64_t local_quotient_fallbacks = 0;    // prove_sharded: local quotient -> broadcast path (invalid trace)
    bool reserve(const std::vector<size_t>& sizes);
    // ends the deferral: blocks in `revive` stay live (returned: which of them had been parked),
    // every other parked block goes back to the free list
    std::vector<void*> flush_deferred(const std::vector<void*>& revive);

    explicit Context(int dev);
    ~Context();
    Context(const Context&) = delete;

    void* alloc(size_t bytes);
    void free(void* p);
    template <class T>
    T* alloc_n(size_t n) { return static_cast<T*>(alloc(n * sizeof(T))); }
    void release_cache();

    // MEASUREMENT AID (ts_ctx_set_replay, tools/latency_replay.py): what would a proof cost if the host never
    // had to wait for the device -- the ceiling of a "zero-host-sync" design?  Mode 1 records, at every point
    // where prove() synchronises to read a small device result (two commitment roots, the opened-value sums,
    // the FRI block), the bytes it read; mode 2 proves the SAME trace again without those synchronisations,
    // handing the host the recorded bytes at once (the device still computes and writes the same values).
    // The proof must come out identical; only the final gather still synchronises.
    int replay_mode = 0;
    size_t replay_pos = 0;
    std::vector<std::vector<char>> replay_log;
    void sync_point(void* host, size_t bytes);                        // host memory a kernel wrote (mailbox)
    void d2h_point(void* host_dst, const void* dev_src, size_t bytes);  // an async copy + sync
    void* pinned(size_t bytes);
    void ensure_twiddles(unsigned log_size);
    // The host's wait for the stream: hipStreamSynchronize puts the thread to sleep on an interrupt after a
    // short spin, and the wake-up is the OS scheduler's business -- tens of microseconds normally, milliseconds
    // now and then.  TS_SYNC_SPIN=1 polls hipStreamQuery instead (the thread has nothing else to do during
    // the 0.1-0.5 ms it waits for; one core per context is busy while a proof runs).
    void sync() {
        static const int spin = [] { const char* e = getenv("TS_SYNC_SPIN"); return e ? atoi(e) : 0; }();
        if (!spin) {
            TS_HIP(hipStreamSynchronize(stream));
            return;
        }
        hipError_t e;
        while ((e = hipStreamQuery(stream)) == hipErrorNotReady) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        TS_HIP(e);
    }
};

// RAII device buffer from the context's pool
template <class T>
struct DevBuf {
    Context* ctx = nullptr;
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(Context* c, size_t count) : ctx(c), p(c->alloc_n<T>(count)), n(count) {}
    // takes over a block that is still live in the context's pool (Context::flush_deferred)
    static DevBuf adopt(Context* c, T* ptr, size_t count) {
        DevBuf b;
        b.ctx = c; b.p = ptr; b.n = count;
        return b;
    }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : ctx(o.ctx), p(o.p), n(o.n) { o.p = nullptr; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            reset();
            ctx = o.ctx; p = o.p; n = o.n; o.p = nullptr;
        }
        return *this;
    }
    ~DevBuf() { reset(); }
    void reset() {
        if (p) ctx->free(p);
        p = nullptr;
    }
};

// Brackets one kernel launch with events when per-kernel timing is on (bench.py's roofline leg).
struct KernelTimer {
    Context* ctx;
    Context::KernelEvent ev;
    bool on;
    KernelTimer(Context* c, const char* name) : ctx(c), on(c->kernel_timing) {
        if (on) {
            ev.name = name;
            (void)hipEventCreate(&ev.e0);
            (void)hipEventCreate(&ev.e1);
            (void)hipEventRecord(ev.e0, ctx->stream);
        }
    }
    ~KernelTimer() {
        if (on) {
            (void)hipEventRecord(ev.e1, ctx->stream);
            ctx->kernel_events.push_back(ev);
        }
    }
};

#define TS_LAUNCH(ctx, kernel, grid, block, lds, ...)                               \
    do {                                                                            \
        ts::KernelTimer _kt(&(ctx), #kernel);                                       \
        hipLaunchKernelGGL(kernel, grid, block, lds, (ctx).stream, __VA_ARGS__);    \
    } while (0)

struct StageTimer {
    Context* ctx;
    const char* name;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    StageTimer(Context* c, const char* n) : ctx(c), name(n) {
        if (ctx->timing) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, ctx->stream);
        }
    }
    ~StageTimer() {
        if (ctx->timing && e0) {
            (void)hipEventRecord(e1, ctx->stream);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ctx->stage_ms.emplace_back(name, ms);
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
    }
};

}  // namespace ts
