// tools/microbench2.hip -- cost of 32-bit rotate forms on gfx950 (Blake3 uses 4 rotates per G).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int ITER = 4096, ILP = 8;
#define BODY(NAME, EXPR) __global__ void NAME(uint32_t* out, uint32_t seed) { \
    uint32_t x[ILP]; for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x + i; \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < ILP; i++) { uint32_t v = x[i]; x[i] = (EXPR) ^ seed; } } \
    uint32_t s = 0; for (int i = 0; i < ILP; i++) s ^= x[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = s; }
__device__ __forceinline__ uint32_t lshl_or(uint32_t a, uint32_t sh, uint32_t c) { uint32_t r; asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(sh), "v"(c)); return r; }
__device__ __forceinline__ uint32_t perm(uint32_t a, uint32_t b, uint32_t sel) { uint32_t r; asm("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(sel)); return r; }
BODY(k_xor, v)
BODY(k_alignbit, __builtin_amdgcn_alignbit(v, v, 12))
BODY(k_shift2, lshl_or(v, 20, v >> 12))
BODY(k_perm16, perm(v, v, 0x01000302u))
BODY(k_alignbyte, __builtin_amdgcn_alignbyte(v, v, 2))
BODY(k_add3, v + seed + (uint32_t)it)
template <class K> void run(K k, uint32_t* d, const char* name) {
    const int blocks = 256 * 8, threads = 256; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double per = ms / 5 * 1e-3; double rate = (double)ITER * ILP * blocks * threads / per;
    printf("%-12s %8.3f ms  %.2f iter/clk/CU at 2.4 GHz\n", name, per * 1e3, rate / 256 / 2.4e9);
}
int main() { uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run(k_xor, d, "xor"); run(k_alignbit, d, "alignbit+xor"); run(k_shift2, d, "shr+lshl_or+xor"); run(k_perm16, d, "perm+xor");
    run(k_alignbyte, d, "alignbyte+xor"); run(k_add3, d, "add3+xor"); return 0; }
