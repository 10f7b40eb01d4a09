#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base_px0, int row_stride_bytes, int lane) {
    const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
    const unsigned char* a = base_px0 + (size_t)(8 * h + q) * row_stride_bytes + (cb * 16 + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * row_stride_bytes));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}
__global__ void k(float* o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* t = (__bf16*)smem;   // [16 pixels][64 ch]
  for (int i = threadIdx.x; i < 16 * 64; i += 64) t[i] = (__bf16)(float)((i / 64) * 100 + (i % 64));   // pixel*100 + ch (exact in bf16? up to 1563: not exact) 
  __syncthreads();
  bf16x8 f = tr_frag(smem + 32 * 2, 128, threadIdx.x & 63);   // channel tile 1 (ch 32..63)
  for (int j = 0; j < 8; ++j) o[threadIdx.x * 8 + j] = (float)f[j];
}
int main() {
  float* d; (void)hipMalloc(&d, 64 * 8 * 4); k<<<1, 64, 4096>>>(d); float h[512]; (void)hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
  for (int lane : {0, 1, 5, 17, 33, 50}) { printf("lane %2d:", lane); for (int j = 0; j < 8; ++j) printf(" %6.0f", h[lane * 8 + j]); printf("\n"); }
  return 0;
}
