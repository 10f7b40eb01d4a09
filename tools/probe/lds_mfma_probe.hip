// Micro-benchmark: how many 16-byte LDS operand-fragment reads per MFMA 32x32x16 bf16 can a CU sustain?
// Each wave runs ITER x { R ds_read_b128 (conflict-free, lane-linear) ; M MFMAs using those fragments }.
// Prints MFMA TFLOP/s (chip) and LDS bytes/clk/CU for several (R, M) ratios and waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int R, int M>
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) ((float*)lds)[i] = (float)(i & 7) * 1e-3f;
    __syncthreads();
    f32x16 acc[M];
    for (int m = 0; m < M; ++m)
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    const unsigned char* base = lds + wave * 16384 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        bf16x8 f[R];
#pragma unroll
        for (int r = 0; r < R; ++r) f[r] = *(const bf16x8*)(base + ((it * R + r) & 15) * 1024);
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m % R], f[(m + 1) % R], acc[m], 0, 0, 0);
    }
    float s = 0.0f;
    for (int m = 0; m < M; ++m)
        for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int R, int M>
void run(float* out, int blocks_per_cu) {
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<R, M><<<grid, 256, 65536, 0>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<R, M><<<grid, 256, 65536, 0>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * M, flops = mfma * 32768.0;
    const double bytes = (double)grid * 4 * iters * R * 1024.0;
    printf("R=%d M=%d blocks/CU=%d: %.3f ms  %.0f TFLOP/s  LDS %.1f B/clk/CU (at 2.4 GHz)  frags/MFMA %.2f\n", R, M, blocks_per_cu, ms,
           flops / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, (double)R / M);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 2 * 256 * 4);
    hipFuncSetAttribute((const void*)probe<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<6, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<1, 8>(out, bpc);
        run<2, 8>(out, bpc);
        run<4, 8>(out, bpc);
        run<6, 8>(out, bpc);
        run<8, 8>(out, bpc);
        run<8, 4>(out, bpc);
        run<8, 1>(out, bpc);
    }
    return 0;
}
