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

// Measurement kinds 3 / 4 (round 5: where does a contiguous NTT pass's power go?): the same butterflies
// with the real kernels' LDS traffic -- a thread's 16 values leave for LDS and come back after every
// radix-16 round (4 stages x 8 butterflies): ONE 4-byte LDS access per butterfly, as in
// k_lde_fwd_contig / k_intt_contig (three round trips per 12 stages).  Kind 3 uses ds_write_b32 /
// ds_read_b32 on the padded image (element i at word i + (i >> 5)); kind 4 moves the same bytes as
// 16-byte accesses.  No global traffic either way.
constexpr int LDSB_ROUNDS = 256;
template <bool WIDE>
__global__ void __launch_bounds__(256) k_alu_butterflies_lds(uint32_t* __restrict__ out, uint32_t seed) {
    __shared__ uint32_t img[4096 + 128 + 16];
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = (seed + threadIdx.x * 7 + i) % P;
    const uint32_t w = seed % P;
    for (int it = 0; it < LDSB_ROUNDS; it++) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int half = 8 >> d;
#pragma unroll
            for (int q = 0; q < 16; q++)
                if ((q & half) == 0) {
                    const uint32_t x = red2p(v[q]);
                    const uint32_t t = mont_mul(v[q + half], w);
                    v[q] = x + t;
                    v[q + half] = x - t + P;
                }
        }
        if (WIDE) {
            uint4* p4 = reinterpret_cast<uint4*>(img) + threadIdx.x * 4 + (threadIdx.x >> 3);
#pragma unroll
            for (int k = 0; k < 4; k++) p4[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
            __syncthreads();
            const uint4* r4 = reinterpret_cast<const uint4*>(img) + ((threadIdx.x + 64) & 255) * 4 + (((threadIdx.x + 64) & 255) >> 3);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 u = r4[k];
                v[4 * k] = u.x; v[4 * k + 1] = u.y; v[4 * k + 2] = u.z; v[4 * k + 3] = u.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const uint32_t i = threadIdx.x + 256u * q;  // distance-256 round: lanes consecutive
                img[i + (i >> 5)] = v[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const uint32_t i = 16u * threadIdx.x + q;  // distance-1 round
                v[q] = img[i + (i >> 5)];
            }
        }
        __syncthreads();
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= v[i];
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
    TS_REQUIRE(kind >= 0 && kind <= 4, TS_ERR_INVALID,
               "alu_ceiling: kind is 0 (butterflies), 1 (blake3), 2 (sha256), 3 / 4 (butterflies + LDS round trips)");
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
        else if (kind == 3)
            hipLaunchKernelGGL(k_alu_butterflies_lds<false>, dim3(blocks), dim3(threads), 0, ctx.stream, out.p, 12345u);
        else if (kind == 4)
            hipLaunchKernelGGL(k_alu_butterflies_lds<true>, dim3(blocks), dim3(threads), 0, ctx.stream, out.p, 12345u);
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
    const double units = kind == 0 ? (double)BF_ITER * BF_ILP : kind == 1 ? (double)B3_ITER
                         : kind == 2 ? (double)SHA_ITER : (double)LDSB_ROUNDS * 32;
    return units * blocks * threads / per;
}

}  // namespace ts
