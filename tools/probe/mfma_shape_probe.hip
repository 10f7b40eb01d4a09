// Which bf16 MFMA shape does this chip run faster under the load pattern of conv_wp (operands: A from registers, B fragments re-read
// from LDS at ~0.42 ds_read_b128 per 32x32x16-equivalent, 128 accumulator registers per wave, 2 workgroups of 4 waves per CU,
// RANDOM operand data)?  Same FLOPs and the same LDS bytes per wave in both kernels.  hipcc --offload-arch=gfx950 -O3 mfma_shape_probe.hip -o p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void __launch_bounds__(256, 2) k32(const uint4* __restrict__ rnd, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 256) ((uint4*)lds)[i] = rnd[i];
    __syncthreads();
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    bf16x8 a[3];
    for (int i = 0; i < 3; ++i) a[i] = __builtin_bit_cast(bf16x8, rnd[4096 + (wave * 3 + i) * 64 + lane]);
    const unsigned char* base = lds + wave * 16384 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        bf16x8 x[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) x[j] = *(const bf16x8*)(base + ((it * 10 + j) & 15) * 1024);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ky], x[r + ky], acc[r], 0, 0, 0);
    }
    float s = 0.0f;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + tid] = s;
}

__global__ void __launch_bounds__(256, 2) k16(const uint4* __restrict__ rnd, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 256) ((uint4*)lds)[i] = rnd[i];
    __syncthreads();
    f32x4 acc[32];
    for (int m = 0; m < 32; ++m) for (int r = 0; r < 4; ++r) acc[m][r] = 0.0f;
    bf16x8 a[6];
    for (int i = 0; i < 6; ++i) a[i] = __builtin_bit_cast(bf16x8, rnd[4096 + (wave * 6 + i) * 64 + lane]);
    const unsigned char* base = lds + wave * 16384 + lane * 16;
    for (int it = 0; it < iters; ++it) {      // = two iterations of k32: 96 MFMAs 16x16x32, 20 fragment reads
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            bf16x8 x[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) x[j] = *(const bf16x8*)(base + ((it * 20 + p * 10 + j) & 15) * 1024);
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        acc[(r * 2 + p) * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ky * 2 + h], x[r + ky], acc[(r * 2 + p) * 2 + h], 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int m = 0; m < 32; ++m) for (int r = 0; r < 4; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main() {
    const int n16 = 4096 + 4 * 6 * 64;
    std::vector<unsigned short> h((size_t)n16 * 8);
    srand(1);
    for (auto& v : h) { float f = ((rand() & 0xffff) / 32768.0f - 1.0f); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    uint4* rnd; float* out;
    hipMalloc(&rnd, (size_t)n16 * 16); hipMalloc(&out, 512 * 256 * 4);
    hipMemcpy(rnd, h.data(), (size_t)n16 * 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 512;
    for (int round = 0; round < 4; ++round) {
        for (int which = 0; which < 2; ++which) {
            const int iters = which ? 20000 : 40000;      // same FLOPs
            for (int w = 0; w < 2; ++w) {      // warm launch, then timed
                hipEventRecord(e0);
                if (which) k16<<<grid, 256, 65536, 0>>>(rnd, out, iters); else k32<<<grid, 256, 65536, 0>>>(rnd, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 * 40000.0 * 24 * 32768.0;
            printf("%s: %.2f ms  %.0f TFLOP/s\n", which ? "16x16x32" : "32x32x16", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
