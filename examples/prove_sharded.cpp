// examples/prove_sharded.cpp -- ONE proof over G ranks from a compiled-language host over the public C
// ABI only: what the Rust `prove_gpu_sharded` (bindings/rust/tapstark-gpu/src/prove.rs, unbuilt: no Rust
// toolchain here) does, in C++ so that the test suite can build and run it.  BASELINE config 4's split:
// SynthMulAir-64, log_blowup 4, 16 queries, rank g owns cosets [2g, 2g + 2).
//
//   * G ranks = G host threads, one ts_ctx each.  Communicator "local": the library's in-process group
//     (ts_comm_local_*), every rank on device 0 -- what a one-GPU box can run.  Communicator "rccl": one
//     rank per device, the library's native RCCL communicator (ts_comm_rccl_create: ncclAllGather /
//     ncclBroadcast on each context's stream); the 128-byte unique id is made by rank 0 and handed to
//     the other threads through memory -- a multi-process host would send it over its own channel.
//   * every rank generates the whole trace on its device (ts_trace_synth_mul: trace_replicated), so the
//     only exchanges are sub-roots, the FRI tail vector, the answered queries -- and, unless
//     `localq` is given, the broadcast of the quotient chunks.
//   * checked: every rank's proof equals the single-GPU ts_prove proof of the same trace word for
//     word, and ts_verify accepts it.
//
//   g++ -std=c++17 -pthread -I include examples/prove_sharded.cpp -L tap-stark_amd/lib -ltapstark_hip -o prove_sharded
//   ./prove_sharded [log_n=12] [G=8] [local|rccl] [bcast|localq] [proof.bin]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "tapstark.h"
#include "tapstark_air.hpp"

namespace {

// the build-defined SynthMulAir-64 (tap-stark_amd/airs.py; shape from the reference's commented
// mul_air.rs:29-116): per triple (a, b, c): a*a*b - c = 0, first row a*a + 1 = b, transition a + reps = a'
struct SynthMulAir {
    uint32_t w;
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

}  // namespace

int main(int argc, char** argv) {
    const unsigned log_n = argc > 1 ? (unsigned)atoi(argv[1]) : 12;
    const int G = argc > 2 ? atoi(argv[2]) : 8;
    const bool use_rccl = argc > 3 && std::string(argv[3]) == "rccl";
    const bool localq = argc > 4 && std::string(argv[4]) == "localq";
    const char* dump = argc > 5 ? argv[5] : nullptr;
    const uint64_t n = 1ull << log_n;
    const uint32_t w = 64;
    const ts_fri_config fri = {4, 16, 8};
    const uint64_t seed = 0x7A957A12ull;
    if (G < 1 || G > 16 || (G & (G - 1))) {
        fprintf(stderr, "usage: prove_sharded [log_n] [G = 1, 2, 4, 8, 16] [local|rccl] [bcast|localq] [proof.bin]\n");
        return 2;
    }
    SynthMulAir air_def{w};
    ts::air::Builder builder(w, 0);
    air_def.eval(builder);
    const std::vector<uint32_t> tape = builder.tape();

    // ---- the proof to match: ts_prove on one GPU
    std::vector<uint32_t> want(1u << 22);
    {
        ts_ctx* c0 = nullptr;
        if (ts_ctx_create(0, &c0) != TS_OK) {
            fprintf(stderr, "no MI355X context: %s\n", ts_last_error(nullptr));
            return 2;  // no fallback path exists
        }
        ts_air* air = nullptr;
        ts_matrix* m = nullptr;
        ts_challenger* ch = nullptr;
        size_t nw = 0;
        if (ts_air_compile(c0, tape.data(), tape.size(), &air) != TS_OK || ts_trace_synth_mul(c0, n, w, seed, &m) != TS_OK ||
            ts_chal_new(0, 1, &ch) != TS_OK ||
            ts_prove(c0, &fri, air, ch, m, nullptr, 0, want.data(), want.size(), &nw) != TS_OK) {
            fprintf(stderr, "ts_prove: %s\n", ts_last_error(c0));
            return 1;
        }
        want.resize(nw);
        ts_chal_free(ch);
        ts_matrix_free(c0, m);
        ts_air_free(c0, air);
        ts_ctx_destroy(c0);
    }

    // ---- G ranks
    ts_comm_group* group = nullptr;
    uint8_t uid[128] = {0};
    if (use_rccl) {
        if (!ts_rccl_available() || ts_rccl_unique_id(uid) != TS_OK) {
            fprintf(stderr, "librccl is not available\n");
            return 2;
        }
        // a rank that cannot create its context would leave its peers inside ncclCommInitRank for good:
        // check what every rank needs BEFORE any of them is started
        const int n_dev = ts_device_count();
        if (n_dev < G) {
            fprintf(stderr, "rccl mode needs one device per rank: %d rank(s), %d device(s)\n", G, n_dev);
            return 2;
        }
    } else if (ts_comm_local_group_create(G, &group) != TS_OK) {
        return 1;
    }
    std::vector<std::vector<uint32_t>> proofs(G);
    std::vector<double> ms(G, 0.0);
    std::atomic<int> failed{0};
    std::vector<std::thread> threads;
    for (int r = 0; r < G; r++) {
        threads.emplace_back([&, r] {
            ts_ctx* ctx = nullptr;
            ts_air* air = nullptr;
            ts_rccl_comm* rc = nullptr;
            ts_comm comm;
            memset(&comm, 0, sizeof comm);
            auto fail = [&](const char* what) {
                fprintf(stderr, "rank %d: %s: %s\n", r, what, ctx ? ts_last_error(ctx) : ts_last_error(nullptr));
                failed++;
                if (comm.abort) comm.abort(comm.user);  // the peers must fail, not wait
            };
            // the in-process communicator first: a rank that fails in any later step can then abort the
            // group, and its peers fail at their first rendezvous instead of waiting out the timeout
            if (!use_rccl && ts_comm_local_get(group, r, &comm) != TS_OK) return fail("communicator");
            if (ts_ctx_create(use_rccl ? r : 0, &ctx) != TS_OK) return fail("ts_ctx_create");
            if (ts_air_compile(ctx, tape.data(), tape.size(), &air) != TS_OK) return fail("ts_air_compile");
            if (use_rccl && ts_comm_rccl_create(ctx, uid, r, G, &comm, &rc) != TS_OK) return fail("communicator");
            ts_shard_options opt;
            memset(&opt, 0, sizeof opt);
            opt.struct_size = sizeof opt;
            opt.trace_replicated = 1;
            opt.local_quotient = localq ? 1 : 0;
            std::vector<uint32_t> out(1u << 22);
            for (int pass = 0; pass < 2; pass++) {  // pass 0 primes tables, pools and code objects
                ts_matrix* m = nullptr;
                ts_challenger* ch = nullptr;
                size_t nw = 0;
                if (ts_trace_synth_mul(ctx, n, w, seed, &m) != TS_OK || ts_chal_new(0, 1, &ch) != TS_OK)
                    return fail("inputs");
                ts_ctx_synchronize(ctx);
                const double t0 = now_ms();
                const ts_status s =
                    ts_prove_sharded(ctx, &fri, &comm, air, ch, m, nullptr, 0, &opt, out.data(), out.size(), &nw);
                ms[r] = now_ms() - t0;
                ts_chal_free(ch);
                ts_matrix_free(ctx, m);
                if (s != TS_OK) return fail("ts_prove_sharded");
                proofs[r].assign(out.begin(), out.begin() + nw);
            }
            if (rc) ts_comm_rccl_destroy(rc);
            ts_air_free(ctx, air);
            ts_ctx_destroy(ctx);
        });
    }
    for (auto& t : threads) t.join();
    if (group) ts_comm_local_group_destroy(group);
    if (failed.load()) return 1;
    for (int r = 0; r < G; r++)
        if (proofs[r] != want) {
            fprintf(stderr, "rank %d: the sharded proof differs from ts_prove's\n", r);
            return 1;
        }
    ts_air* vair = nullptr;  // verification needs no GPU
    ts_challenger* fresh = nullptr;
    int verdict = -1;
    if (ts_air_compile(nullptr, tape.data(), tape.size(), &vair) != TS_OK || ts_chal_new(0, 1, &fresh) != TS_OK ||
        ts_verify(&fri, vair, fresh, want.data(), want.size(), nullptr, 0, &verdict) != TS_OK) {
        fprintf(stderr, "ts_verify failed to run\n");
        return 1;
    }
    if (dump) {
        FILE* f = fopen(dump, "wb");
        if (!f || fwrite(proofs[0].data(), 4, proofs[0].size(), f) != proofs[0].size()) return 1;
        fclose(f);
    }
    double slowest = 0;
    for (double d : ms) slowest = d > slowest ? d : slowest;
    printf("prove_sharded: 2^%u x %u, log_blowup 4, one proof over %d rank(s) (%s communicator, %s): %.3f ms on the "
           "slowest rank; every rank's proof equals ts_prove's (%zu words), verify -> %d\n",
           log_n, w, G, use_rccl ? "native RCCL, one device per rank" : "in-process, all ranks on device 0",
           localq ? "local quotient" : "quotient chunks broadcast", slowest, want.size(), verdict);
    return verdict == 0 ? 0 : 1;
}
