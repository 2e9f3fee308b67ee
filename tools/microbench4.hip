// tools/microbench4.hip -- Blake3 compression variants on gfx950: throughput of chains of hash64.
//   V0: as the compiler likes it (v_add3_u32 for a+b+m, v_alignbit_b32 for every rotate)
//   V1: a+b+m as two v_add_u32 (an empty asm between them stops the add3 fusion)
//   V2: V1 + rotr(x ^ y, 16) as two v_xor_b32_sdwa (word-swapped halves), no alignbit
//   V3: V0 + the SDWA rot16 only
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int V> __device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t m) {
    if (V == 1 || V == 2) { uint32_t t = a + b; asm volatile("" : "+v"(t)); return t + m; }
    return a + b + m;
}
__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
template <int V> __device__ __forceinline__ uint32_t xrot16(uint32_t d, uint32_t a) {
    if (V >= 2) {
        uint32_t r;
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(r) : "v"(d), "v"(a));
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(r) : "v"(d), "v"(a));
        return r;
    }
    return rotr(d ^ a, 16);
}
#define G(a, b, c, d, mx, my) a = add3<V>(a, b, mx); d = xrot16<V>(d, a); c = c + d; b = rotr(b ^ c, 12); \
    a = add3<V>(a, b, my); d = rotr(d ^ a, 8); c = c + d; b = rotr(b ^ c, 7);
#define ROUND(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15) \
    G(s0, s4, s8, s12, m0, m1) G(s1, s5, s9, s13, m2, m3) G(s2, s6, s10, s14, m4, m5) G(s3, s7, s11, s15, m6, m7) \
    G(s0, s5, s10, s15, m8, m9) G(s1, s6, s11, s12, m10, m11) G(s2, s7, s8, s13, m12, m13) G(s3, s4, s9, s14, m14, m15)
template <int V> __device__ __forceinline__ void hash64(const uint32_t m[16], uint32_t cv[8]) {
    uint32_t s0 = 0x6A09E667u, s1 = 0xBB67AE85u, s2 = 0x3C6EF372u, s3 = 0xA54FF53Au, s4 = 0x510E527Fu, s5 = 0x9B05688Cu,
             s6 = 0x1F83D9ABu, s7 = 0x5BE0CD19u;
    uint32_t s8 = 0x6A09E667u, s9 = 0xBB67AE85u, s10 = 0x3C6EF372u, s11 = 0xA54FF53Au, s12 = 0, s13 = 0, s14 = 64, s15 = 11;
    ROUND(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15])
    ROUND(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8])
    ROUND(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1])
    ROUND(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6])
    ROUND(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4])
    ROUND(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7])
    ROUND(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13])
    cv[0] = s0 ^ s8; cv[1] = s1 ^ s9; cv[2] = s2 ^ s10; cv[3] = s3 ^ s11; cv[4] = s4 ^ s12; cv[5] = s5 ^ s13; cv[6] = s6 ^ s14; cv[7] = s7 ^ s15;
}
constexpr int ITER = 64;
template <int V> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t m[16], cv[8];
    for (int i = 0; i < 16; i++) m[i] = seed * (i + 1) + threadIdx.x + blockIdx.x * 977;
    for (int it = 0; it < ITER; it++) {
        hash64<V>(m, cv);
        for (int i = 0; i < 8; i++) { m[i] = cv[i]; m[8 + i] ^= cv[i]; }
    }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= cv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K> uint32_t run(K kk, uint32_t* d, const char* name) {
    const int blocks = 256 * 16, threads = 256; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kk, dim3(blocks), dim3(threads), 0, 0, d, 12345u); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kk, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per = ms / 5 * 1e-3; double rate = (double)ITER * blocks * threads / per;
    uint32_t h; (void)hipMemcpy(&h, d + 777, 4, hipMemcpyDeviceToHost);
    printf("%-34s %8.3f ms  %.2f G compress/s   check %08x\n", name, per * 1e3, rate / 1e9, h);
    return h;
}
int main() { uint32_t* d; (void)hipMalloc(&d, 256 * 16 * 256 * 4);
    uint32_t a = run(k<0>, d, "V0 add3 + alignbit");
    uint32_t b = run(k<1>, d, "V1 two adds + alignbit");
    uint32_t c = run(k<2>, d, "V2 two adds + sdwa rot16");
    uint32_t e = run(k<3>, d, "V3 add3 + sdwa rot16");
    printf(a == b && b == c && c == e ? "all variants agree\n" : "MISMATCH\n"); return 0; }
