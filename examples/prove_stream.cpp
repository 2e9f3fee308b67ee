// examples/prove_stream.cpp -- a stream of proofs from a compiled-language host over the public C ABI
// only (include/tapstark.h + the header-only AIR capture): what the Rust `prove_gpu_stream`
// (bindings/rust/tapstark-gpu/src/prove.rs, unbuilt: no Rust toolchain here) does, in C++ so that the
// test suite can build and run it.
//
//   * S lanes = S host threads, one ts_ctx each (own HIP stream, own device pool, own cached tables):
//     the library's threading contract -- one context is driven by one thread;
//   * a start gate: two proofs never start within `gate_ms` of each other, so that the lanes run in
//     complementary phases instead of lockstep (bench.py does the same);
//   * mode "device": every trace is generated in HBM (ts_trace_synth_mul);
//     mode "pinned": every trace comes from page-locked host memory (ts_host_alloc), uploaded with
//     ts_matrix_upload_async on the lane's stream -- the upload of one lane overlaps the proofs of the
//     others: the PCIe-inclusive rate of INTEGRATION.md section 3c;
//   * every proof is checked: proofs of the same trace must be byte-identical (the prover is
//     deterministic) and the first one is verified with ts_verify.
//
//   g++ -std=c++17 -pthread -I include examples/prove_stream.cpp -L tap-stark_amd/lib -ltapstark_hip -o prove_stream
//   ./prove_stream [log_n=20] [n_proofs=40] [lanes=4] [device|pinned] [proof0.bin]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tapstark.h"
#include "tapstark_air.hpp"

namespace {

// the build-defined SynthMulAir-64 of BASELINE configs 3/4 (tap-stark_amd/airs.py; constraint shape
// from the reference's commented mul_air.rs:29-116): per triple (a, b, c): a*a*b - c = 0,
// first row a*a + 1 = b, transition a + reps = a'
struct SynthMulAir {
    uint32_t w;
    uint32_t width() const { return w; }
    void eval(ts::air::Builder& builder) const {
        const auto &local = builder.local(), &next = builder.next();
        const uint32_t reps = w / 3;
        for (uint32_t i = 0; i < reps; i++) {
            const auto a = local[3 * i], b = local[3 * i + 1], c = local[3 * i + 2];
            builder.assert_zero(a * a * b - c);
            builder.when_first_row().assert_eq(a * a + 1, b);
            builder.when_transition().assert_eq(a + (uint64_t)reps, next[3 * i]);
        }
    }
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct Gate {  // minimum spacing between the starts of two proofs
    std::mutex mu;
    double last = -1e18, gap_ms = 0;
    void pass() {
        if (gap_ms <= 0) return;
        std::lock_guard<std::mutex> lk(mu);
        for (;;) {
            const double wait = last + gap_ms - now_ms();
            if (wait <= 0) break;
            std::this_thread::sleep_for(std::chrono::microseconds((long)(1000 * (wait < 0.2 ? wait : 0.2))));
        }
        last = now_ms();
    }
};

}  // namespace

int main(int argc, char** argv) {
    const unsigned log_n = argc > 1 ? (unsigned)atoi(argv[1]) : 20;
    const int n_proofs = argc > 2 ? atoi(argv[2]) : 40;
    const int lanes = argc > 3 ? atoi(argv[3]) : 4;
    const bool pinned = argc > 4 && std::string(argv[4]) == "pinned";
    const char* dump = argc > 5 ? argv[5] : nullptr;
    const uint64_t n = 1ull << log_n;
    const uint32_t w = 64;
    const ts_fri_config fri = {2, 28, 8};
    const uint64_t seed = 0x7A957A12ull;
    if (lanes < 1 || lanes > 16 || n_proofs < lanes) {
        fprintf(stderr, "usage: prove_stream [log_n] [n_proofs >= lanes] [lanes 1..16] [device|pinned]\n");
        return 2;
    }

    SynthMulAir air_def{w};
    ts::air::Builder builder(air_def.width(), 0);
    air_def.eval(builder);
    const std::vector<uint32_t> tape = builder.tape();

    // one host copy of the trace for the pinned mode (made on the device, downloaded once)
    std::vector<uint32_t*> pin(lanes, nullptr);
    std::vector<uint32_t> host_trace;
    {
        ts_ctx* c0 = nullptr;
        if (ts_ctx_create(0, &c0) != TS_OK) {
            fprintf(stderr, "no MI355X context: %s\n", ts_last_error(nullptr));
            return 2;  // no fallback path exists
        }
        if (pinned) {
            ts_matrix* m = nullptr;
            if (ts_trace_synth_mul(c0, n, w, seed, &m) != TS_OK) return 1;
            host_trace.resize(n * w);
            if (ts_matrix_download(c0, m, host_trace.data()) != TS_OK) return 1;
            ts_matrix_free(c0, m);
            for (int l = 0; l < lanes; l++) {
                void* p = nullptr;
                if (ts_host_alloc(n * w * 4, &p) != TS_OK) return 1;
                pin[l] = (uint32_t*)p;
            }
        }
        ts_ctx_destroy(c0);
    }

    std::vector<std::vector<uint32_t>> proofs(n_proofs);
    std::atomic<int> failed{0};
    Gate gate;
    std::vector<double> solo_ms(lanes, 0.0), lane_done_ms(lanes, 0.0);
    // TS_STREAM_LATENCIES=1: every timed proof's (start, wall time), to look for stalls without any Python in the process
    const bool want_lat = getenv("TS_STREAM_LATENCIES") != nullptr;
    std::vector<std::vector<std::pair<double, double>>> lat(lanes);
    double t_begin = 0, t_end = 0;
    std::mutex mu;
    int primed = 0;
    std::vector<std::thread> threads;
    std::atomic<bool> go{false};

    for (int l = 0; l < lanes; l++) {
        threads.emplace_back([&, l] {
            ts_ctx* ctx = nullptr;
            ts_air* air = nullptr;
            auto fail = [&](const char* what) {
                fprintf(stderr, "lane %d: %s: %s\n", l, what, ctx ? ts_last_error(ctx) : "?");
                failed++;
            };
            if (ts_ctx_create(0, &ctx) != TS_OK) return fail("ts_ctx_create");
            if (ts_air_compile(ctx, tape.data(), tape.size(), &air) != TS_OK) return fail("ts_air_compile");
            std::vector<uint32_t> out(1u << 20);
            if (pinned) memcpy(pin[l], host_trace.data(), n * w * 4);  // stands for the caller's trace generator
            // `m` is consumed.  Pinned mode: the lane's buffer is only read (by the asynchronous copy)
            // until ts_prove returns, and ts_prove blocks until the proof is on the host.
            auto make_trace = [&]() -> ts_matrix* {
                ts_matrix* m = nullptr;
                const ts_status s = pinned ? ts_matrix_upload_async(ctx, pin[l], n, w, &m)
                                           : ts_trace_synth_mul(ctx, n, w, seed, &m);
                return s == TS_OK ? m : nullptr;
            };
            auto prove_one = [&](ts_matrix* m, std::vector<uint32_t>* keep) -> bool {
                if (!m) return false;
                ts_challenger* ch = nullptr;
                if (ts_chal_new(0, 1, &ch) != TS_OK) return false;
                size_t nw = 0;
                gate.pass();
                const ts_status s = ts_prove(ctx, &fri, air, ch, m, nullptr, 0, out.data(), out.size(), &nw);
                ts_chal_free(ch);
                ts_matrix_free(ctx, m);
                if (s != TS_OK) return false;
                if (keep) keep->assign(out.begin(), out.begin() + nw);
                return true;
            };
            // device mode: this lane's traces (proofs l, l + S, ...) are resident in HBM before the clock
            // starts, as in bench.py; pinned mode: the upload is part of every step.  They are made BEFORE
            // the priming proof: the context's device pool recycles blocks by size, and a trace made after
            // priming would take the block the next proof's transposed copy is about to ask for -- the
            // first timed proof of every lane would then hipMalloc 268 MB (~10 ms, device-wide).
            std::vector<ts_matrix*> mats;
            if (!pinned)
                for (int i = l; i < n_proofs; i += lanes) mats.push_back(make_trace());
            // prime the lane (tables, pools, code objects) and time one proof alone on lane 0
            if (!prove_one(make_trace(), nullptr)) return fail("ts_prove (prime)");
            if (l == 0) {
                ts_matrix* m = make_trace();
                ts_ctx_synchronize(ctx);
                const double t0 = now_ms();
                if (!prove_one(m, nullptr)) return fail("ts_prove");
                solo_ms[0] = now_ms() - t0;
            }
            ts_ctx_synchronize(ctx);
            {
                std::lock_guard<std::mutex> lk(mu);
                primed++;
            }
            while (!go.load()) std::this_thread::yield();
            for (int i = l, k = 0; i < n_proofs; i += lanes, k++) {
                const double t0 = now_ms();
                if (!prove_one(pinned ? make_trace() : mats[k], &proofs[i])) return fail("ts_prove");
                if (want_lat) lat[l].push_back({t0, now_ms() - t0});
            }
            lane_done_ms[l] = now_ms();  // before the teardown: ts_ctx_destroy gives a ~3.5 GB pool back (tens of ms)
            ts_air_free(ctx, air);
            ts_ctx_destroy(ctx);
        });
    }
    for (;;) {  // every lane primed (lane 0 has also measured a proof alone): set the gate, start the clock
        {
            std::lock_guard<std::mutex> lk(mu);
            if (primed + failed.load() >= lanes) break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (failed.load()) {
        go = true;
        for (auto& t : threads) t.join();
        return 1;
    }
    gate.gap_ms = lanes > 1 ? 0.25 * solo_ms[0] : 0.0;
    t_begin = now_ms();
    go = true;
    for (auto& t : threads) t.join();
    for (double d : lane_done_ms) t_end = d > t_end ? d : t_end;
    if (failed.load()) return 1;

    // every proof of the same trace is the same proof; the first one verifies
    for (int i = 1; i < n_proofs; i++)
        if (proofs[i] != proofs[0]) {
            fprintf(stderr, "proof %d differs from proof 0\n", i);
            return 1;
        }
    ts_air* vair = nullptr;  // verification needs no GPU: compile the AIR without a context
    ts_challenger* fresh = nullptr;
    int verdict = -1;
    if (ts_air_compile(nullptr, tape.data(), tape.size(), &vair) != TS_OK || ts_chal_new(0, 1, &fresh) != TS_OK ||
        ts_verify(&fri, vair, fresh, proofs[0].data(), proofs[0].size(), nullptr, 0, &verdict) != TS_OK) {
        fprintf(stderr, "ts_verify failed to run\n");
        return 1;
    }
    if (dump) {
        FILE* f = fopen(dump, "wb");
        if (!f || fwrite(proofs[0].data(), 4, proofs[0].size(), f) != proofs[0].size()) return 1;
        fclose(f);
    }
    for (int l = 0; l < lanes; l++)
        if (pin[l]) ts_host_free(pin[l]);
    const double ms = (t_end - t_begin) / n_proofs;
    printf("prove_stream: 2^%u x %u, %d proofs on %d lanes, traces %s: %.3f ms per proof (%.1f proofs/s, %.3g cells/s); "
           "one proof alone %.3f ms, start gate %.2f ms; all proofs identical (%zu words), verify -> %d\n",
           log_n, w, n_proofs, lanes, pinned ? "from pinned host memory (ts_matrix_upload_async)" : "generated on the device",
           ms, 1e3 / ms, (double)n * w * 1e3 / ms, solo_ms[0], gate.gap_ms, proofs[0].size(), verdict);
    if (want_lat) {
        std::vector<double> all;
        for (auto& v : lat)
            for (auto& e : v) all.push_back(e.second);
        std::sort(all.begin(), all.end());
        const double med = all[all.size() / 2];
        printf("  proof wall times: median %.2f ms, p99 %.2f, max %.2f; proofs above 1.4 x median (lane: start since the clock started, ms -> wall time):\n",
               med, all[(size_t)(0.99 * all.size())], all.back());
        int n_slow = 0;
        for (int l = 0; l < lanes; l++)
            for (auto& e : lat[l])
                if (e.second > 1.4 * med) {
                    printf("    lane %d: %9.1f -> %.2f\n", l, e.first - t_begin, e.second);
                    n_slow++;
                }
        printf("  %d of %zu proofs\n", n_slow, all.size());
    }
    return verdict == 0 ? 0 : 1;
}
