// LDS fp32 atomic-add throughput: conflict-free (lane-linear), 2-way, random, same-address.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) probe(float* out, const int* idx, int iters, int mode) {
    __shared__ float acc[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) acc[i] = 0.0f;
    __syncthreads();
    int a = idx[blockIdx.x % 4 * 256 + tid + mode * 1024];
    const float v = 1.0f + tid * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&acc[(a + k * 1031 + it * 17) & 8191], v);
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = acc[tid];
}
__global__ void __launch_bounds__(256) probe_u32(unsigned* out, const int* idx, int iters, int mode) {
    __shared__ unsigned acc[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) acc[i] = 0;
    __syncthreads();
    int a = idx[blockIdx.x % 4 * 256 + tid + mode * 1024];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&acc[(a + k * 1031 + it * 17) & 8191], 3u);
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = acc[tid];
}
__global__ void __launch_bounds__(256) probe_u64(unsigned* out, const int* idx, int iters, int mode) {
    __shared__ unsigned long long acc[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) acc[i] = 0;
    __syncthreads();
    int a = idx[blockIdx.x % 4 * 256 + tid + mode * 1024];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&acc[(a + k * 1031 + it * 17) & 8191], 0x100000003ull);
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = (unsigned)(acc[tid] >> 7);
}
int main() {
    int h[4 * 1024];
    for (int m = 0; m < 4; ++m)
        for (int i = 0; i < 1024; ++i) {
            int t = i & 255;
            if (m == 0) h[m * 1024 + i] = t;                      // lane-linear
            if (m == 1) h[m * 1024 + i] = (t & 63) * 32 + (t >> 6); // all lanes of a wave in one bank
            if (m == 2) h[m * 1024 + i] = (rand() & 8191);        // random
            if (m == 3) h[m * 1024 + i] = (t >> 6);               // same address per wave
        }
    int* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    float* o; hipMalloc(&o, 512 * 256 * 4);
    const char* names[4] = {"lane-linear", "one bank", "random", "same address"};
    for (int u = 0; u < 3; ++u)
        for (int m = 0; m < 4; ++m) {
            const int iters = u == 0 ? 200 : 2000, grid = 512;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            if (u == 0) probe<<<grid, 256>>>(o, d, 10, m); else if (u == 1) probe_u32<<<grid, 256>>>((unsigned*)o, d, 10, m); else probe_u64<<<grid, 256>>>((unsigned*)o, d, 10, m);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (u == 0) probe<<<grid, 256>>>(o, d, iters, m); else if (u == 1) probe_u32<<<grid, 256>>>((unsigned*)o, d, iters, m); else probe_u64<<<grid, 256>>>((unsigned*)o, d, iters, m);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double lane_atomics = (double)grid * 256 * iters * 8;
            printf("%s %-12s: %.3f ms  %.2f lane-atomics/clk/CU (2.1 GHz)\n", u == 0 ? "f32" : (u == 1 ? "u32" : "u64"), names[m], ms, lane_atomics / (ms * 1e-3) / 256 / 2.1e9);
        }
    return 0;
}
