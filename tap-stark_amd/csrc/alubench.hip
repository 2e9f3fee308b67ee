// The integer-ALU ceilings of the two instruction mixes the prover is made of, measured with the
// library's own device functions (bb.hpp butterflies in the lazy [0, 2p) form the NTT kernels use,
// blake3.hpp compressions), no memory traffic: bench.py prints them next to the achieved
// butterflies/s and compressions/s of a proof (its `alu_ceiling` block), so that "VALU-bound, not
// HBM-bound" is a measured statement of the same run rather than a derivation.
#include "blake3.hpp"
#include "kernels.hpp"
#include "sha256.hpp"

namespace ts {

namespace {

constexpr int BF_ITER = 1024, BF_ILP = 8;
constexpr int B3_ITER = 64;

// forward butterflies exactly as radix_butterflies (ntt_lde.hip): a = red2p(a), t = mont_mul(b, w),
// a' = a + t, b' = a - t + p
__global__ void __launch_bounds__(256) k_alu_butterflies(uint32_t* __restrict__ out, uint32_t seed) {
    uint32_t a[BF_ILP], b[BF_ILP];
#pragma unroll
    for (int i = 0; i < BF_ILP; i++) {
        a[i] = (seed + threadIdx.x * 7 + i) % P;
        b[i] = (seed * 3 + threadIdx.x + i) % P;
    }
    const uint32_t w = seed % P;
    for (int it = 0; it < BF_ITER; it++) {
#pragma unroll
        for (int i = 0; i < BF_ILP; i++) {
            const uint32_t x = red2p(a[i]);
            const uint32_t t = mont_mul(b[i], w);
            a[i] = x + t;
            b[i] = x - t + P;
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < BF_ILP; i++) s ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_alu_blake3(uint32_t* __restrict__ out, uint32_t seed) {
    uint32_t m[16], cv[8];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = seed * (i + 1) + threadIdx.x + blockIdx.x * 977;
    for (int it = 0; it < B3_ITER; it++) {
        b3::hash64(m, cv);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            m[i] = cv[i];
            m[8 + i] ^= cv[i];
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= cv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

constexpr int SHA_ITER = 32;
__global__ void __launch_bounds__(256) k_alu_sha256(uint32_t* __restrict__ out, uint32_t seed) {
    uint32_t m[16], h[8];
    sha::iv(h);
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = seed * (i + 1) + threadIdx.x + blockIdx.x * 977;
    for (int it = 0; it < SHA_ITER; it++) {
        sha::compress(h, m);
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] ^= h[i];
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= h[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

// kind 0: NTT butterflies per second; kind 1: Blake3 compressions per second; kind 2: SHA-256
// compressions per second (whole chip)
double alu_ceiling(Context& ctx, int kind) {
    TS_REQUIRE(kind >= 0 && kind <= 2, TS_ERR_INVALID, "alu_ceiling: kind is 0 (butterflies), 1 (blake3) or 2 (sha256)");
    const int blocks = ctx.num_cus * 16, threads = 256, reps = 5;
    DevBuf<uint32_t> out(&ctx, (size_t)blocks * threads);
    hipEvent_t e0, e1;
    TS_HIP(hipEventCreate(&e0));
    TS_HIP(hipEventCreate(&e1));
    auto launch = [&] {
        if (kind == 0)
            hipLaunchKernelGGL(k_alu_butterflies, dim3(blocks), dim3(threads), 0, ctx.stream, out.p, 12345u);
        else if (kind == 1)
            hipLaunchKernelGGL(k_alu_blake3, dim3(blocks), dim3(threads), 0, ctx.stream, out.p, 12345u);
        else
            hipLaunchKernelGGL(k_alu_sha256, dim3(blocks), dim3(threads), 0, ctx.stream, out.p, 12345u);
    };
    launch();  // warm-up (code object load, clocks)
    TS_HIP(hipEventRecord(e0, ctx.stream));
    for (int r = 0; r < reps; r++) launch();
    TS_HIP(hipEventRecord(e1, ctx.stream));
    TS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    TS_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double per = (double)ms / reps * 1e-3;
    const double units = kind == 0 ? (double)BF_ITER * BF_ILP : kind == 1 ? (double)B3_ITER : (double)SHA_ITER;
    return units * blocks * threads / per;
}

}  // namespace ts
