// probe of ds_read_b64_tr_b16 lane mapping (used by the wgrad kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(float* o) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int lane = threadIdx.x & 63;
  int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const short* addr = lds + q * 64 + grp * 16 + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int j = 0; j < 4; ++j) o[lane * 4 + j] = (float)v[j];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
    int grp = lane >> 4, li = lane & 15; float exp = j * 64 + grp * 16 + li;
    if (h[lane * 4 + j] != exp) { if (bad < 8) printf("lane %d j %d got %g exp %g\n", lane, j, h[lane*4+j], exp); ++bad; }
  }
  printf("tr_probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK lane i gets column i of rows 0..3", bad);
  return 0;
}
