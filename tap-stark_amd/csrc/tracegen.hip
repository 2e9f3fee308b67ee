// Trace generation on the device: the trace is born in HBM, so
// prove() needs no H2D of it.
//   Fibonacci   reference uni-stark/tests/fib_air.rs:59-78 generate_trace_rows(a, b, n): row 0 =
//               (a, b), then (l, r) -> (r, l + r).  A scan, done here as [l_i, r_i] = M^i [a, b],
//               M = [[0,1],[1,1]]: each thread jumps to the start of its 64-row block with a 2x2
//               matrix power and walks the block with additions.
//   SynthMul    build-defined "SynthMulAir-w" trace (tap-stark_amd/airs.py generate_synth_mul_trace;
//               SURVEY.md section 8(d) config 3): a = reps*row + k, b = SplitMix64 stream value
//               (row 0: a*a + 1), c = a*a*b, free columns from the stream.  Element (row, j) of the
//               stream has index row*(reps+free) + j: every row is independent.
#include "kernels.hpp"

namespace ts {

namespace {

struct M2 {  // 2x2 matrix, Montgomery entries
    uint32_t a, b, c, d;
};
__device__ __forceinline__ M2 m2_mul(M2 x, M2 y) {
    return M2{add(mont_mul(x.a, y.a), mont_mul(x.b, y.c)), add(mont_mul(x.a, y.b), mont_mul(x.b, y.d)),
              add(mont_mul(x.c, y.a), mont_mul(x.d, y.c)), add(mont_mul(x.c, y.b), mont_mul(x.d, y.d))};
}

constexpr int FIB_BLOCK = 64;

__global__ void __launch_bounds__(256)
k_trace_fibonacci(uint32_t* __restrict__ out, uint32_t a, uint32_t b, uint64_t n) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t row0 = t * FIB_BLOCK;
    if (row0 >= n) return;
    // M^row0 by square-and-multiply; M = [[0,1],[1,1]]
    M2 r{R_MOD_P, 0, 0, R_MOD_P}, base{0, R_MOD_P, R_MOD_P, R_MOD_P};
    for (uint64_t e = row0; e; e >>= 1) {
        if (e & 1) r = m2_mul(r, base);
        base = m2_mul(base, base);
    }
    // (l, r) = M^row0 (a, b): Montgomery matrix x canonical vector -> canonical
    uint32_t l = add(mont_mul(r.a, a), mont_mul(r.b, b));
    uint32_t rr = add(mont_mul(r.c, a), mont_mul(r.d, b));
    const uint64_t end = row0 + FIB_BLOCK < n ? row0 + FIB_BLOCK : n;
    for (uint64_t i = row0; i < end; i++) {
        *reinterpret_cast<uint2*>(out + 2 * i) = make_uint2(l, rr);
        const uint32_t nx = add(l, rr);
        l = rr;
        rr = nx;
    }
}

__device__ __forceinline__ uint32_t splitmix_mod_p(uint64_t seed, uint64_t index) {
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z % P);
}

// one thread per (row, stream column j): j < reps writes the triple k = j, j >= reps a free column
__global__ void __launch_bounds__(256)
k_trace_synth_mul(uint32_t* __restrict__ out, uint64_t n, uint32_t width, uint32_t reps, uint64_t seed) {
    const uint32_t per_row = width - 2 * reps;  // reps + free
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * per_row) return;
    const uint64_t row = t / per_row;
    const uint32_t j = (uint32_t)(t % per_row);
    const uint32_t rnd = splitmix_mod_p(seed, t);
    uint32_t* o = out + row * width;
    if (j < reps) {
        const uint32_t a = (uint32_t)(((uint64_t)reps * row + j) % P);
        const uint32_t am = to_mont(a);
        const uint32_t a2 = mont_mul(a, am);  // a*a, canonical
        const uint32_t bv = row == 0 ? add(a2, 1u) : rnd;
        o[3 * j] = a;
        o[3 * j + 1] = bv;
        o[3 * j + 2] = mont_mul(a2, to_mont(bv));
    } else {
        o[3 * reps + (j - reps)] = rnd;
    }
}

// SynthExt-w (airs.py generate_synth_ext_trace): every element
// (row, c) is stream value row*width + c, except z = x*y in F[X]/(X^4 - 11) for each group of 12
// columns (x, y, z) and the last column = 7 + row.  One thread per (row, group): it draws x and y,
// multiplies, and writes the group's 12 columns; the columns after the last group are filled by
// the threads with g == groups.
__global__ void __launch_bounds__(256)
k_trace_synth_ext(uint32_t* __restrict__ out, uint64_t n, uint32_t width, uint32_t groups, uint64_t seed) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * (groups + 1)) return;
    const uint64_t row = t / (groups + 1);
    const uint32_t g = (uint32_t)(t % (groups + 1));
    uint32_t* o = out + row * width;
    const uint64_t s0 = row * width;
    if (g == groups) {
        for (uint32_t c = 12 * groups; c + 1 < width; c++) o[c] = splitmix_mod_p(seed, s0 + c);
        o[width - 1] = (uint32_t)((7 + row) % P);
        return;
    }
    const uint32_t base = 12 * g;
    uint32_t x[4], y[4];
    for (int i = 0; i < 4; i++) {
        x[i] = splitmix_mod_p(seed, s0 + base + i);
        y[i] = splitmix_mod_p(seed, s0 + base + 4 + i);
        o[base + i] = x[i];
        o[base + 4 + i] = y[i];
    }
    const uint32_t w11 = to_mont(11u);
    for (int k = 0; k < 4; k++) {
        uint32_t acc = 0;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                if (((i + j) & 3) != k) continue;
                uint32_t term = mont_mul(x[i], to_mont(y[j]));  // canonical product
                if (i + j >= 4) term = mont_mul(term, w11);
                acc = add(acc, term);
            }
        o[base + 8 + k] = acc;
    }
}

}  // namespace

void launch_trace_synth_ext(Context& ctx, uint32_t* out, uint64_t n, uint32_t width, uint64_t seed) {
    TS_REQUIRE(width >= 1 && width <= 4096, TS_ERR_INVALID, "synth_ext trace: bad width");
    const uint32_t groups = (width - 1) / 12;
    const uint64_t total = n * (groups + 1);
    TS_REQUIRE(total < (1ull << 40), TS_ERR_INVALID, "synth_ext trace: too large");
    TS_LAUNCH(ctx, k_trace_synth_ext, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, out, n, width,
              groups, seed);
    TS_HIP(hipGetLastError());
}

void launch_trace_fibonacci(Context& ctx, uint32_t* out, uint32_t a, uint32_t b, uint64_t n) {
    const uint64_t threads = (n + FIB_BLOCK - 1) / FIB_BLOCK;
    TS_LAUNCH(ctx, k_trace_fibonacci, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, out, a % P,
              b % P, n);
    TS_HIP(hipGetLastError());
}

void launch_trace_synth_mul(Context& ctx, uint32_t* out, uint64_t n, uint32_t width, uint64_t seed) {
    TS_REQUIRE(width >= 1 && width <= 4096, TS_ERR_INVALID, "synth_mul trace: bad width");
    const uint32_t reps = width / 3;
    const uint64_t total = n * (width - 2 * reps);
    TS_REQUIRE(total < (1ull << 40), TS_ERR_INVALID, "synth_mul trace: too large");
    TS_LAUNCH(ctx, k_trace_synth_mul, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, out, n, width,
              reps, seed);
    TS_HIP(hipGetLastError());
}

}  // namespace ts
