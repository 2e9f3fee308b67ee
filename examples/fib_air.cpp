// examples/fib_air.cpp -- the reference's live end-to-end test (uni-stark/tests/fib_air.rs:21-149)
// written against the public C ABI only (include/tapstark.h + the header-only AIR capture), the way
// a compiled-language host (the reference is Rust) would drive the MI355X prover:
//
//   FibonacciAir::eval            fib_air.rs:29-57
//   generate_trace_rows(0, 1, n)  fib_air.rs:59-78     (computed in HBM: ts_trace_fibonacci)
//   test_public_value             fib_air.rs:117-149   prove, then verify with a fresh challenger
//
//   g++ -std=c++17 -I include examples/fib_air.cpp -L tap-stark_amd/lib -ltapstark_hip -o fib_air
//   ./fib_air [log_n] [proof.bin]  |  ./fib_air --tape      (default log_n = 3: the reference's n = 8, pis = [0, 1, 21])
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "tapstark.h"
#include "tapstark_air.hpp"

namespace {

constexpr uint32_t P = 0x78000001u;

struct FibonacciAir {  // fib_air.rs:21-27: two columns, left and right
    static constexpr uint32_t NUM_FIBONACCI_COLS = 2;
    uint32_t width() const { return NUM_FIBONACCI_COLS; }
    void eval(ts::air::Builder& builder) const {
        const auto& pis = builder.public_values();
        const auto a = pis[0], b = pis[1], x = pis[2];
        const auto &local = builder.local(), &next = builder.next();
        const int left = 0, right = 1;

        auto when_first_row = builder.when_first_row();
        when_first_row.assert_eq(local[left], a);
        when_first_row.assert_eq(local[right], b);

        auto when_transition = builder.when_transition();
        // a' <- b
        when_transition.assert_eq(local[right], next[left]);
        // b' <- a + b
        when_transition.assert_eq(local[left] + local[right], next[right]);

        builder.when_last_row().assert_eq(local[right], x);
    }
};

#define CHECK(call)                                                                      \
    do {                                                                                 \
        ts_status _s = (call);                                                           \
        if (_s != TS_OK) {                                                               \
            fprintf(stderr, "%s -> status %d: %s\n", #call, (int)_s, ts_last_error(ctx)); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

}  // namespace

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--tape") {  // the captured AIR only (needs no GPU)
        FibonacciAir fib;
        ts::air::Builder builder(fib.width(), 3);
        fib.eval(builder);
        for (uint32_t w : builder.tape()) printf("%u ", w);
        printf("\n%u %u\n", builder.max_constraint_degree(), builder.log_quotient_degree());
        return 0;
    }
    const unsigned log_n = argc > 1 ? (unsigned)atoi(argv[1]) : 3;
    const uint64_t n = 1ull << log_n;
    ts_ctx* ctx = nullptr;
    if (ts_ctx_create(0, &ctx) != TS_OK) {
        fprintf(stderr, "no MI355X context: %s\n", ts_last_error(nullptr));
        return 2;  // no fallback path exists
    }
    // public values [a, b, x]: x = the last row's right column (fib_air.rs:133-139)
    uint32_t l = 0, r = 1;
    for (uint64_t i = 1; i < n; i++) {
        const uint32_t nx = (uint32_t)(((uint64_t)l + r) % P);
        l = r;
        r = nx;
    }
    const std::vector<uint32_t> pis = {0, 1, r};

    // the AIR, captured symbolically like get_symbolic_constraints (symbolic_builder.rs:52-64)
    FibonacciAir fib;
    ts::air::Builder builder(fib.width(), (uint32_t)pis.size());
    fib.eval(builder);
    const std::vector<uint32_t> tape = builder.tape();
    ts_air* air = nullptr;
    CHECK(ts_air_compile(ctx, tape.data(), tape.size(), &air));

    // fib_air.rs:119-129: log_blowup 2, 28 queries, 8 proof-of-work bits
    const ts_fri_config fri = {2, 28, 8};
    ts_matrix* trace = nullptr;
    CHECK(ts_trace_fibonacci(ctx, 0, 1, n, &trace));
    int64_t violation = 0;
    CHECK(ts_check_constraints(ctx, air, trace, pis.data(), (uint32_t)pis.size(), &violation));
    if (violation != -1) {
        fprintf(stderr, "constraint %lld violated\n", (long long)violation);
        return 1;
    }

    ts_challenger* challenger = nullptr;
    CHECK(ts_chal_new(0, 1, &challenger));
    std::vector<uint32_t> proof(1u << 22);
    size_t n_words = 0;
    CHECK(ts_prove(ctx, &fri, air, challenger, trace, pis.data(), (uint32_t)pis.size(), proof.data(),
                   proof.size(), &n_words));
    proof.resize(n_words);

    ts_challenger* fresh = nullptr;  // fib_air.rs:145-148: verify with a new challenger
    CHECK(ts_chal_new(0, 1, &fresh));
    int verdict = -1;
    CHECK(ts_verify(&fri, air, fresh, proof.data(), proof.size(), pis.data(), (uint32_t)pis.size(), &verdict));
    // and a wrong public value must be refused (OodEvaluationMismatch)
    std::vector<uint32_t> wrong = pis;
    wrong[2] = (wrong[2] + 1) % P;
    ts_challenger* fresh2 = nullptr;
    CHECK(ts_chal_new(0, 1, &fresh2));
    int verdict_wrong = -1;
    CHECK(ts_verify(&fri, air, fresh2, proof.data(), proof.size(), wrong.data(), (uint32_t)wrong.size(),
                    &verdict_wrong));

    std::vector<uint8_t> wire(5 * proof.size() + 16);
    size_t n_bytes = 0;
    CHECK(ts_proof_to_postcard(proof.data(), proof.size(), wire.data(), wire.size(), &n_bytes));
    if (argc > 2) {
        FILE* f = fopen(argv[2], "wb");
        if (!f) return 1;
        fwrite(proof.data(), 4, proof.size(), f);
        fclose(f);
    }
    printf("fib_air: n = 2^%u, public values [0, 1, %u], proof %zu words (%zu postcard bytes), "
           "verify -> %d, with a wrong public value -> %d\n",
           log_n, r, n_words, n_bytes, verdict, verdict_wrong);
    ts_chal_free(challenger);
    ts_chal_free(fresh);
    ts_chal_free(fresh2);
    ts_matrix_free(ctx, trace);
    ts_air_free(ctx, air);
    ts_ctx_destroy(ctx);
    return verdict == 0 && verdict_wrong == 7 ? 0 : 1;
}
