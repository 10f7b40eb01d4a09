// Memory-only forms of the grid_sample warp (WP:95-119) at B x 3 x H x W fp32: what the access PATTERN of a decomposition streams, with no
// arithmetic and no waits between a tile's loads and its stores (an upper bound for any kernel of that decomposition).
//   per output pixel: flow 8 B read, 3 image planes read through a staged window (LDS-DMA, 16 B per lane), 3 + 3 planes written (out, mask).
//   form T (tiles):  a persistent workgroup walks TH x TW output tiles; per tile it fetches the (TH + 43) x (TW + 48) window of every channel
//   form S (slide):  a workgroup owns a TW-column band of one sample and a run of rows; after a 43-row warm-up it fetches TH NEW window
//                    rows per step (the sliding ring of the band form), the flow and the stores as in T
//   form L (linear): grid-stride float4 copy of the same planes (5 read, 6 written)
// hipcc -O3 --offload-arch=gfx950 tools/probe/gw_stream_probe.hip -o tools/probe/gw_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int RX = 24, RY = 21;

struct Args {
    const float* img; const float* flow; float* out; float* mask;
    int B, H, W, TH, TW, tiles_x, tiles_y, seg_rows, nseg, xcd_order, per_channel;
};

__device__ __forceinline__ void dma16(const float* src, float* lds) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// window rows [wy0, wy0 + rows) x columns [wx0, wx0 + cols) of one plane -> LDS (all pieces land in the same region: contents are irrelevant here)
__device__ __forceinline__ void fetch_rows(const float* plane, int wy0, int rows, int wx0, int cols, int H, int W, float* lds, int lds_floats) {
    const int vpr = cols / 4, nv = rows * vpr;
    const int nthreads = blockDim.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int base = wave * 64; base < nv; base += nthreads) {
        const int vid = min(base + lane, nv - 1), row = vid / vpr, col = vid - row * vpr;
        const int gy = min(max(wy0 + row, 0), H - 1), gx = min(max(wx0 + col * 4, 0), W - 4);
        dma16(plane + (size_t)gy * W + gx, lds + ((base * 4) % lds_floats));
    }
}

__device__ __forceinline__ void tile_flow_and_stores(const Args& a, int n, int oy0, int ox0, int th, int c0 = 0, int c1 = 3) {
    const size_t plane = (size_t)a.H * a.W;
    const int vpr = a.TW / 4, nv = th * vpr;
    for (int v = threadIdx.x; v < nv; v += blockDim.x) {
        const int r = v / vpr, c4 = (v - r * vpr) * 4;
        const int y = oy0 + r, x = ox0 + c4;
        if (y < a.H && x < a.W) {
            const size_t pix = (size_t)y * a.W + x;
            const float4 f0 = *(const float4*)(a.flow + (size_t)n * 2 * plane + pix);
            const float4 f1 = *(const float4*)(a.flow + (size_t)n * 2 * plane + plane + pix);
            const float4 o = make_float4(f0.x + f1.x, f0.y + f1.y, f0.z + f1.z, f0.w + f1.w);
            for (int c = c0; c < c1; ++c) {
                *(float4*)(a.out + ((size_t)n * 3 + c) * plane + pix) = o;
                *(float4*)(a.mask + ((size_t)n * 3 + c) * plane + pix) = o;
            }
        }
    }
}

__global__ void __launch_bounds__(1024) form_tiles(const Args a, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const size_t plane = (size_t)a.H * a.W;
    const int tpi = a.tiles_x * a.tiles_y, ntiles = tpi * a.B, g = gridDim.x;
    for (int l = blockIdx.x; l < ntiles; l += g) {
        int t = l;
        if (a.xcd_order) {           // in every round of g tiles the workgroups of one XCD (block % 8) take a contiguous run of tiles
            const int k = l / g, b = l - k * g;
            if ((g & 7) == 0 && (k + 1) * g <= ntiles) t = k * g + (b & 7) * (g >> 3) + (b >> 3);
        }
        const int n = t / tpi, t_in = t - n * tpi;
        const int oy0 = (t_in / a.tiles_x) * a.TH, ox0 = (t_in % a.tiles_x) * a.TW;
        for (int c = 0; c < 3; ++c)
            fetch_rows(a.img + ((size_t)n * 3 + c) * plane, oy0 - RY, a.TH + 2 * RY + 1, ox0 - RX, a.TW + 2 * RX, a.H, a.W, lds, lds_floats);
        tile_flow_and_stores(a, n, oy0, ox0, a.TH);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void __launch_bounds__(1024) form_slide(const Args a, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const size_t plane = (size_t)a.H * a.W;
    const int bands = a.W / a.TW;
    int j = blockIdx.x;
    if (a.xcd_order) {               // the segments of a band, and neighbouring bands, on one XCD: block b -> unit (b % 8) * (G / 8) + b / 8
        const int g = gridDim.x;
        if ((g & 7) == 0) j = (j & 7) * (g >> 3) + (j >> 3);
    }
    // per_channel: one workgroup per (sample, band, segment, channel), the three channel workgroups of a segment adjacent in the order
    const int nch = a.per_channel ? 3 : 1;
    const int ch = j % nch; j /= nch;
    const int seg = j % a.nseg, band = (j / a.nseg) % bands, n = j / (a.nseg * bands);
    if (n >= a.B) return;
    const int c0 = a.per_channel ? ch : 0, c1 = a.per_channel ? ch + 1 : 3;
    const int r0 = seg * a.seg_rows, r1 = min(r0 + a.seg_rows, a.H), ox0 = band * a.TW;
    // warm-up: rows r0 - RY .. r0 + RY of the planes
    for (int c = c0; c < c1; ++c) fetch_rows(a.img + ((size_t)n * 3 + c) * plane, r0 - RY, 2 * RY + 1, ox0 - RX, a.TW + 2 * RX, a.H, a.W, lds, lds_floats);
    for (int y = r0; y < r1; y += a.TH) {
        const int th = min(a.TH, r1 - y);
        for (int c = c0; c < c1; ++c) fetch_rows(a.img + ((size_t)n * 3 + c) * plane, y + RY + 1, th, ox0 - RX, a.TW + 2 * RX, a.H, a.W, lds, lds_floats);
        tile_flow_and_stores(a, n, y, ox0, th, c0, c1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void __launch_bounds__(256) form_linear(const Args a) {
    const size_t plane = (size_t)a.H * a.W, nv = (size_t)a.B * plane / 4;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (size_t)gridDim.x * blockDim.x) {
        const size_t n = v / (plane / 4), pix = (v - n * (plane / 4)) * 4;
        const float4 f0 = *(const float4*)(a.flow + n * 2 * plane + pix), f1 = *(const float4*)(a.flow + n * 2 * plane + plane + pix);
        float4 s = make_float4(f0.x + f1.x, f0.y + f1.y, f0.z + f1.z, f0.w + f1.w);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 i = *(const float4*)(a.img + (n * 3 + c) * plane + pix);
            s.x += i.x; s.y += i.y; s.z += i.z; s.w += i.w;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            *(float4*)(a.out + (n * 3 + c) * plane + pix) = s;
            *(float4*)(a.mask + (n * 3 + c) * plane + pix) = s;
        }
    }
}

template <class F>
static float timed(F launch, int reps = 20) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int B = 16, H = 440, W = 1024;
    const size_t plane = (size_t)H * W;
    float *img, *flow, *out, *mask;
    CK(hipMalloc(&img, B * 3 * plane * 4)); CK(hipMalloc(&flow, B * 2 * plane * 4)); CK(hipMalloc(&out, B * 3 * plane * 4)); CK(hipMalloc(&mask, B * 3 * plane * 4));
    CK(hipMemset(img, 0, B * 3 * plane * 4)); CK(hipMemset(flow, 0, B * 2 * plane * 4));
    CK(hipFuncSetAttribute((const void*)form_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)form_slide, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const double bytes = 44.0 * B * plane;
    Args a{img, flow, out, mask, B, H, W, 0, 0, 0, 0, 0, 0, 1, 0};
    const bool round2 = argc > 1;
    {
        const float us = timed([&] { form_linear<<<2048, 256>>>(a); });
        printf("{\"form\": \"linear\", \"us\": %.1f, \"TBps_algorithmic\": %.2f}\n", us, bytes / us / 1e6);
    }
    struct TC { int th, tw, threads, lds_kb, grid; };
    const std::vector<TC> tcs = {{64, 64, 1024, 144, 256}, {64, 64, 512, 72, 512}, {32, 128, 1024, 144, 256}, {32, 128, 512, 72, 512}, {16, 256, 1024, 144, 256},
                                 {16, 256, 512, 72, 512}, {8, 512, 1024, 144, 256}, {8, 512, 512, 72, 512}, {4, 1024, 1024, 144, 256}, {8, 256, 512, 72, 512},
                                 {16, 128, 512, 72, 512}, {16, 256, 256, 36, 1024}, {8, 1024, 512, 72, 512}};
    if (!round2)
    for (const TC& t : tcs)
        for (int xo = 0; xo < 2; ++xo) {
            a.TH = t.th; a.TW = t.tw; a.tiles_x = (W + t.tw - 1) / t.tw; a.tiles_y = (H + t.th - 1) / t.th; a.xcd_order = xo;
            const int ldsf = t.lds_kb * 1024 / 4;
            const float us = timed([&] { form_tiles<<<t.grid, t.threads, t.lds_kb * 1024>>>(a, ldsf); });
            printf("{\"form\": \"tiles\", \"TH\": %d, \"TW\": %d, \"threads\": %d, \"lds_kb\": %d, \"grid\": %d, \"xcd_order\": %d, \"us\": %.1f, \"TBps_algorithmic\": %.2f}\n",
                   t.th, t.tw, t.threads, t.lds_kb, t.grid, xo, us, bytes / us / 1e6);
        }
    struct SC { int th, tw, threads, lds_kb, nseg, pc; };
    const std::vector<SC> scs2 = {{8, 128, 1024, 128, 2, 0}, {16, 128, 1024, 156, 2, 0}, {8, 128, 1024, 128, 4, 0}, {8, 128, 512, 128, 2, 0}, {4, 128, 1024, 128, 2, 0},
                                  {8, 256, 512, 72, 4, 1}, {8, 256, 512, 72, 8, 1}, {8, 256, 256, 72, 4, 1}, {16, 256, 512, 72, 4, 1}, {8, 256, 1024, 144, 4, 0},
                                  {8, 256, 1024, 144, 2, 0}, {8, 512, 1024, 144, 4, 0}, {8, 512, 512, 72, 4, 1}, {8, 1024, 1024, 72, 4, 1}, {8, 1024, 512, 72, 8, 1}};
    const std::vector<SC> scs1 = {{8, 128, 512, 72, 4, 0}, {8, 128, 256, 36, 8, 0}, {8, 256, 512, 72, 8, 0}, {4, 256, 512, 72, 8, 0}, {16, 256, 512, 72, 8, 0}, {8, 256, 1024, 144, 4, 0},
                                 {8, 512, 1024, 144, 8, 0}, {8, 512, 512, 72, 16, 0}, {4, 1024, 1024, 144, 16, 0}, {2, 1024, 512, 72, 32, 0}, {8, 256, 256, 36, 16, 0},
                                 {8, 1024, 1024, 144, 16, 0}};
    const std::vector<SC>& scs = round2 ? scs2 : scs1;
    for (const SC& s : scs)
        for (int xo = 0; xo < 2; ++xo) {
            a.TH = s.th; a.TW = s.tw; a.nseg = s.nseg; a.seg_rows = ((H + s.nseg - 1) / s.nseg + s.th - 1) / s.th * s.th; a.xcd_order = xo;
            a.nseg = (H + a.seg_rows - 1) / a.seg_rows; a.per_channel = s.pc;
            const int grid = B * (W / s.tw) * a.nseg * (s.pc ? 3 : 1), ldsf = s.lds_kb * 1024 / 4;
            const float us = timed([&] { form_slide<<<grid, s.threads, s.lds_kb * 1024>>>(a, ldsf); });
            printf("{\"form\": \"slide\", \"TH\": %d, \"TW\": %d, \"threads\": %d, \"lds_kb\": %d, \"nseg\": %d, \"seg_rows\": %d, \"grid\": %d, \"xcd_order\": %d, \"us\": %.1f, "
                   "\"TBps_algorithmic\": %.2f}\n", s.th, s.tw, s.threads, s.lds_kb, a.nseg, a.seg_rows, grid, xo, us, bytes / us / 1e6);
        }
    return 0;
}
