// Write-bandwidth of the conv epilogue's store pattern vs a fully coalesced one.  Output [npix][C] bf16.
//  mode 0: conv-epilogue pattern: a wave covers 32 pixels; each store instruction writes, per pixel, 32 contiguous bytes
//          (two half-waves x 16 B); C*2/32 instructions fill a pixel row
//  mode 1: coalesced: consecutive lanes write consecutive 16-byte units of a pixel row
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(uint4* out, int C, size_t npix, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int upr = C / 8;                        // 16-byte units per pixel row
    const uint4 v = make_uint4(tid, lane, wave, 7);
    for (size_t p0 = ((size_t)blockIdx.x * 4 + wave) * 32; p0 < npix; p0 += (size_t)gridDim.x * 128) {
        if (mode == 0) {
            const size_t pix = p0 + l31;
            for (int u = 0; u < upr; u += 2) out[pix * upr + u + half] = v;
        } else {
            for (int i = lane; i < 32 * upr; i += 64) out[p0 * upr + i] = v;
        }
    }
}
int main() {
    const size_t npix = (size_t)16 * 440 * 1024;
    uint4* o; hipMalloc(&o, npix * 384 * 2);
    for (int C : {64, 128, 384})
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<2048, 256>>>(o, C, npix, mode); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) k<<<2048, 256>>>(o, C, npix, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("C=%3d mode %d (%s): %.3f ms  %.2f TB/s\n", C, mode, mode ? "coalesced" : "epilogue pattern", ms, npix * C * 2.0 / ms / 1e9);
        }
    return 0;
}
