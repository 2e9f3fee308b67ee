// tools/microbench.hip -- integer-ALU ceilings on gfx950 for the prover's two inner operations
// (Montgomery product, Blake3 compression).  Measurement tool only; not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -I tap-stark_amd/csrc tools/microbench.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#include "bb.hpp"
#include "blake3.hpp"

using namespace ts;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int ILP = 8;

__global__ void k_add(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < ILP; i++) x[i] = (x[i] + seed) ^ (x[i] >> 3);
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < ILP; i++) x[i] = x[i] * (seed | 1);
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < ILP; i++) x[i] = __umulhi(x[i], seed) + 0x9e3779b9u;
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul64(uint32_t* out, uint32_t seed) {
    uint64_t x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < ILP; i++) x[i] = (uint64_t)(uint32_t)x[i] * seed + (x[i] >> 32);
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ void k_mul24(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < ILP; i++) x[i] = __umul24(x[i], seed) + 7u;
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mont(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = (seed + threadIdx.x + i) % P;
    const uint32_t w = seed % P;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < ILP; i++) x[i] = mont_mul(x[i], w);
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// butterfly: (a, b) -> (a + w b, a - w b)
__global__ void k_bfly(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = (seed + threadIdx.x + i) % P;
    const uint32_t w = seed % P;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < ILP; i += 2) {
            uint32_t v = mont_mul(x[i + 1], w);
            uint32_t a = x[i];
            x[i] = add(a, v);
            x[i + 1] = sub(a, v);
        }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_blake3(uint32_t* out, uint32_t seed) {
    uint32_t m[16], cv[8];
    for (int i = 0; i < 16; i++) m[i] = seed + threadIdx.x * 16 + i;
    b3::iv(cv);
    for (int it = 0; it < ITER / 16; it++) {
        b3::compress(cv, m, 64, 11);
        m[it & 15] ^= cv[0];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = cv[0] ^ cv[7];
}

template <class K>
double run(K k, uint32_t* d, const char* name, double ops_per_thread) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double per = ms / 5 * 1e-3;
    double rate = ops_per_thread * blocks * threads / per;
    printf("%-10s %8.3f ms  %9.2f Gop/s  (%.2f ops/clk/CU at 2.4 GHz)\n", name, per * 1e3, rate / 1e9,
           rate / 256 / 2.4e9);
    return rate;
}

int main() {
    uint32_t* d;
    CHK(hipMalloc(&d, 256 * 8 * 256 * 4));
    double n = (double)ITER * ILP;
    run(k_add, d, "add+xor+shr", n * 3);
    run(k_mullo, d, "mul_lo", n);
    run(k_mulhi, d, "mul_hi+add", n);
    run(k_mul64, d, "mad_u64", n);
    run(k_mul24, d, "mul24+add", n);
    run(k_mont, d, "mont_mul", n);
    run(k_bfly, d, "butterfly", n / 2);
    run(k_blake3, d, "blake3", (double)(ITER / 16));
    hipFree(d);
    return 0;
}
