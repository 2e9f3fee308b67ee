// tools/microbench5.hip -- what a plain streaming kernel reaches on this GPU (the practical HBM ceiling
// against which the 8 TB/s spec number should be read): copy, read-only, write-only, 1 GiB each.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ a, uint32_t* __restrict__ out, size_t n) {
    uint32_t s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = a[i]; s ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345678u) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    uint4 *a, *b; uint32_t* o;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 4);
    (void)hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
        float ms;
        hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("grid %6d  copy  %.2f TB/s (read+write)", grid, 2.0 * bytes * 5 / (ms * 1e-3) / 1e12);
        (void)hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, o, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("   read %.2f TB/s", 1.0 * bytes * 5 / (ms * 1e-3) / 1e12);
        (void)hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("   write %.2f TB/s\n", 1.0 * bytes * 5 / (ms * 1e-3) / 1e12);
    }
    (void)hipEventRecord(e0); for (int r = 0; r < 5; r++) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    float ms; (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy D2D %.2f TB/s (read+write)\n", 2.0 * bytes * 5 / (ms * 1e-3) / 1e12);
    return 0;
}
