// global (L2) fp32 atomic-add throughput: lane-linear vs locally-random addresses over a 115 MB output (16x4x440x1024 floats)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) probe(float* out, size_t n, int mode, int per_thread) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned r = (unsigned)tid * 2654435761u;
    for (int k = 0; k < per_thread; ++k) {
        size_t a;
        if (mode == 0) a = (tid + (size_t)k * gridDim.x * 256) % n;                       // streaming, coalesced
        else { r = r * 1664525u + 1013904223u; a = (tid * 4 + (r >> 8) % 20000) % n; }   // within ~20 rows of the pixel
        atomicAdd(out + a, 1.0f);
    }
}
int main() {
    const size_t n = (size_t)16 * 4 * 440 * 1024;
    float* o; hipMalloc(&o, n * 4); hipMemset(o, 0, n * 4);
    for (int m = 0; m < 2; ++m) {
        const int per = 16, grid = (int)(n / 4 / 256 / 4);      // 16 atomics per "pixel-thread", n/16 threads -> n atomics in total
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<<<grid, 256>>>(o, n, m, 1); hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<<<grid, 256>>>(o, n, m, per);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double atoms = (double)grid * 256 * per;
        printf("mode %d: %.3f ms for %.1f M atomics -> %.1f G atomics/s\n", m, ms, atoms / 1e6, atoms / ms / 1e6);
    }
    return 0;
}
