// tools/microbench3.hip -- is an FP64-FMA modular butterfly cheaper than the int32 Montgomery one on
// gfx950?  Chains of dependent butterflies, ILP independent chains per thread.
//   int : t = mont_mul(b, w); a' = add(a, t); b' = sub(a, t)                 (canonical in/out)
//   f64 : t = b*w mod p via h = b*w, l = fma(b,w,-h), q = rint(h/p), t = fma(-q,p,h)+l  (|t| <= p/2+),
//         a' = a + t, b' = a - t with NO range correction (doubles have the headroom)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr uint32_t P = 0x78000001u, P_NEG_INV = 0x77ffffffu;
constexpr int ITER = 2048, ILP = 8;
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t mont_mul(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t m = (uint32_t)t * P_NEG_INV;
    uint32_t r = (uint32_t)((t + (uint64_t)m * P) >> 32);
    return umin32(r, r - P);
}
__device__ __forceinline__ uint32_t addp(uint32_t a, uint32_t b) { uint32_t s = a + b; return umin32(s, s - P); }
__device__ __forceinline__ uint32_t subp(uint32_t a, uint32_t b) { uint32_t d = a - b; return umin32(d, d + P); }

__global__ void k_int(uint32_t* out, uint32_t seed) {
    uint32_t a[ILP], b[ILP];
    for (int i = 0; i < ILP; i++) { a[i] = (seed + threadIdx.x * 7 + i) % P; b[i] = (seed * 3 + threadIdx.x + i) % P; }
    uint32_t w = seed % P;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t t = mont_mul(b[i], w);
            uint32_t na = addp(a[i], t), nb = subp(a[i], t);
            a[i] = na; b[i] = nb;
        }
    }
    uint32_t s = 0; for (int i = 0; i < ILP; i++) s ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__device__ __forceinline__ double mulmod(double b, double w, double p, double pinv) {
    double h = b * w;
    double l = fma(b, w, -h);
    double q = rint(h * pinv);
    double r = fma(-q, p, h);
    return r + l;
}
__global__ void k_f64(uint32_t* out, uint32_t seed) {
    double a[ILP], b[ILP];
    const double p = (double)P, pinv = 1.0 / (double)P;
    for (int i = 0; i < ILP; i++) { a[i] = (double)((seed + threadIdx.x * 7 + i) % P); b[i] = (double)((seed * 3 + threadIdx.x + i) % P); }
    double w = (double)(seed % P);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            double t = mulmod(b[i], w, p, pinv);
            double na = a[i] + t, nb = a[i] - t;
            a[i] = na; b[i] = nb;
        }
        if ((it & 15) == 15) {  // keep |a|, |b| bounded (a real kernel normalises once per LDS round)
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                a[i] = fma(-rint(a[i] * pinv), p, a[i]);
                b[i] = fma(-rint(b[i] * pinv), p, b[i]);
            }
        }
    }
    double s = 0; for (int i = 0; i < ILP; i++) s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(long long)s;
}
// conversion + normalisation cost: u32 -> f64, reduce, f64 -> canonical u32
__global__ void k_cvt(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP];
    const double p = (double)P, pinv = 1.0 / (double)P;
    for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            double d = (double)x[i] * 3.0;
            double r = fma(-rint(d * pinv), p, d);
            r = r < 0 ? r + p : r;
            x[i] = (uint32_t)r ^ seed;
        }
    }
    uint32_t s = 0; for (int i = 0; i < ILP; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K> void run(K k, uint32_t* d, const char* name, int threads) {
    const int blocks = 256 * 8; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double per = ms / 5 * 1e-3; double rate = (double)ITER * ILP * blocks * threads / per;
    printf("%-28s %8.3f ms  %.3f G butterflies/s  (%.2f per clk per CU at 2.4 GHz)\n", name, per * 1e3, rate / 1e9, rate / 256 / 2.4e9);
}
int main() { uint32_t* d; hipMalloc(&d, 256 * 8 * 512 * 4);
    run(k_int, d, "int32 montgomery butterfly", 256); run(k_f64, d, "f64 fma butterfly (lazy)", 256);
    run(k_cvt, d, "u32->f64, reduce, ->u32", 256);
    uint32_t h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("%u\n", h[0]); return 0; }
