// Persistent form of the UNet's init conv (denoising_diffusion.py:339: 7x7, <= 8 input channels padded to 8, 64 output channels).
//
// The generic kernel (conv_igemm.hip, Cfg<8, 64>) runs one 8x32-pixel tile per workgroup and re-streams the 57 KB of weights
// through a double-buffered LDS slab per tile: 28,160 tiles x 57 KB of L2 -> LDS traffic and 8 workgroup barriers per tile for
// 112 MFMAs per wave.  Here the weights stay in LDS for the whole launch (all 7 kernel rows x 8 tap columns x 8 channels x 64
// outputs = 57,344 B), a workgroup walks many tiles, and its four waves never meet after the weights are in:
//   * a wave owns two output rows x 32 pixels of the tile and keeps ITS OWN input image (8 rows x 39 pixels x 16 B = 5 KB) in a
//     private LDS region: the global loads of the next tile's image are issued before the MFMAs of this tile and written after the
//     epilogue -- LDS operations of one wave execute in order, so no barrier and no second buffer;
//   * every operand address of the 28 k-steps is a compile-time offset from two base registers, the operands of step s + 1 are
//     read before the MFMAs of step s;
//   * the epilogue transposes an output row (32 pixels x 64 channels) through the same private region (dead once the MFMAs have
//     read it): consecutive lanes then store consecutive 16-byte units, 1 KB contiguous per instruction.  The accumulator layout
//     stores 32 bytes per pixel and instruction, a pattern this 0.92 GB output stream runs at 3.2 TB/s with (the no-MFMA
//     ablation of the first version: profiles/r04_conv7_persist_ablations.txt);
//   * two workgroups per CU (77.8 KB LDS each).
// Same tap-pair packing, operand images and MFMA order as the generic kernel; the bias is the accumulator's initial value instead of
// an add after the last MFMA (one fp32 rounding apart before the rounding to bf16).
#include <cstdlib>
#include "common.h"
#include "conv_params.h"

// diagnostic builds (tools/build_wp_variants.sh <tag> "-DOFD_C7_ABL=n" conv7.hip): bit 0 no MFMAs, bit 1 no output stores, bit 2 no input prefetch
#ifndef OFD_C7_ABL
#define OFD_C7_ABL 0
#endif

namespace ofd {
namespace c7 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TH = 8, TW = 32, NTHREADS = 256;
constexpr int IW = TW + 6 + 1;                                    // 39: the 8th tap column reads one pixel further right (zero weights)
constexpr int WR = 8, WPIX = WR * IW;                             // input rows / pixels behind a wave's two output rows
constexpr int OPITCH = 64 * 2 + 16;                               // transposition pitch of a pixel: conflict-free 16-byte writes
constexpr int WXB = 5120;                                         // a wave's region: input image (4,992 B) | one transposed output row (4,608 B)
constexpr int W_SLAB = 8 * 64 * 16;                               // one kernel row: 8 (tap column) rows x 64 outputs x 16 B
constexpr int W_ALL = 7 * W_SLAB;                                 // 57,344 B
constexpr int LDS_BYTES = W_ALL + 4 * WXB;                        // 77,824 B
constexpr int XPT = (WPIX + 63) / 64;                             // 5
static_assert(WPIX * 16 <= WXB && 32 * OPITCH <= WXB, "a wave's LDS region");

struct Tile { int b, oy0, ox0; };

// XCD-aware order (as conv_igemm.hip): the virtual index v runs on XCD v % 8 and gets a contiguous run of tiles there
__device__ __forceinline__ Tile tile_of(int v, int ntiles, int tiles_x, int tpi) {
    int tile = v;
    if (ntiles >= 8) {
        const int q = ntiles / 8, r = ntiles % 8, xcd = v % 8, idx = v / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    tile = min(tile, ntiles - 1);                                  // (past the end: a valid tile, never loaded or stored)
    const int b = tile / tpi, t_in = tile - b * tpi;
    return {b, (t_in / tiles_x) * TH, (t_in % tiles_x) * TW};
}

struct XRegs { uint4 v[XPT]; unsigned ok; };

// input rows [oy0 + 2 wave - 3, + 8) x columns [ox0 - 3, + 39) of the wave's strip: unit u = row * 39 + column, lane + 64 i
__device__ __forceinline__ void load_x(XRegs& x, const ConvParams& P, const Tile& t, int wave, int lane) {
    const ConvSrcDev& S = P.src[0];
    const bf16_t* base = S.ptr + (size_t)t.b * P.H * P.W * S.src_channels + S.ch_offset;
    x.ok = 0;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int u = min(lane + i * 64, WPIX - 1);
        const int ty = u / IW, tx = u - ty * IW;
        const int iy = t.oy0 + 2 * wave - 3 + ty, ix = t.ox0 - 3 + tx;
        const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        x.ok |= (ok ? 1u : 0u) << i;
        const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);
        x.v[i] = *(const uint4*)(base + ((size_t)cy * P.W + cx) * S.src_channels);
    }
}

__device__ __forceinline__ void write_x(const XRegs& x, unsigned char* xw, int lane) {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int u = min(lane + i * 64, WPIX - 1);                // (the surplus lanes rewrite the last pixel with identical bytes)
        const bool ok = (x.ok >> i) & 1u;
        uint4 v = x.v[i];
        v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
        *(uint4*)(xw + u * 16) = v;
    }
}

__global__ void __launch_bounds__(NTHREADS, 2) conv7x7_c8_persist_kernel(const ConvParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lds_w = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    unsigned char* const xw = smem + W_ALL + wave * WXB;
    const int tpi = P.tiles_x * P.tiles_y, ntiles = tpi * P.B;
    const int n_iter = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // weights: the prepared tensor ([ky][8 tap columns][64 outputs][8 channels]) is the LDS image
#pragma unroll
    for (int i = 0; i < W_ALL / 16 / NTHREADS; ++i)
        *(uint4*)(lds_w + (tid + i * NTHREADS) * 16) = *(const uint4*)(P.weight + (size_t)(tid + i * NTHREADS) * 8);

    // bias of the 32 channels a lane accumulates: acc[nt][.][4g + j] is channel nt*32 + 8g + 4*half + j
    f32x16 biasv[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = P.bias ? *(const float4*)(P.bias + nt * 32 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
            biasv[nt][4 * g] = bv.x; biasv[nt][4 * g + 1] = bv.y; biasv[nt][4 * g + 2] = bv.z; biasv[nt][4 * g + 3] = bv.w;
        }

    XRegs xr;
    Tile cur = tile_of((int)blockIdx.x, ntiles, P.tiles_x, tpi);
    load_x(xr, P, cur, wave, lane);
    write_x(xr, xw, lane);
    __syncthreads();                                              // the weights are in: the only workgroup barrier of the kernel

    // a lane's two operand bases: its pixel column (+ the half-wave's tap of the pair) in the wave's image; its weight row (half) and
    // output column l31
    const unsigned char* const xrow = xw + (half + l31) * 16;
    const unsigned char* const wrow = lds_w + (half * 64 + l31) * 16;

    for (int it = 0; it < n_iter; ++it) {
        const Tile nxt = tile_of((int)blockIdx.x + (it + 1) * (int)gridDim.x, ntiles, P.tiles_x, tpi);
        if (it + 1 < n_iter && !(OFD_C7_ABL & 4)) load_x(xr, P, nxt, wave, lane);

        f32x16 acc[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[nt][pt] = biasv[nt];

        // 28 k-steps (7 kernel rows x 4 tap pairs), operands of step s + 1 read before the MFMAs of step s.  The pixel operand of (row ky,
        // output row pt = 1) is the one of (ky + 1, pt = 0): an image row is read once and kept for the next kernel row.
        auto rdx = [&](int r, int ks) { return *(const bf16x8*)(xrow + (r * IW + 2 * ks) * 16); };
        auto rdw = [&](int ky, int ks, int nt) { return *(const bf16x8*)(wrow + ky * W_SLAB + (ks * 2 * 64 + nt * 32) * 16); };
        bf16x8 xkeep[2][4], wf[2][2], xn[2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xkeep[0][ks] = rdx(0, ks);
        wf[0][0] = rdw(0, 0, 0); wf[0][1] = rdw(0, 0, 1); xn[0] = rdx(1, 0);
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int st = 0; st < 28; ++st) {
            const int ky = st >> 2, ks = st & 3, cb = st & 1, nb = cb ^ 1;
            if (st + 1 < 28) {
                const int ky1 = (st + 1) >> 2, ks1 = (st + 1) & 3;
                wf[nb][0] = rdw(ky1, ks1, 0); wf[nb][1] = rdw(ky1, ks1, 1); xn[nb] = rdx(ky1 + 1, ks1);
            }
            __builtin_amdgcn_sched_barrier(0);            // (the scheduler otherwise sinks the reads to just before their use)
            const bf16x8 x0 = xkeep[ky & 1][ks], x1 = xn[cb];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (OFD_C7_ABL & 1) { asm volatile("" ::"v"(wf[cb][nt]), "v"(x0), "v"(x1)); continue; }
                acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][nt], x0, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][nt], x1, acc[nt][1], 0, 0, 0);
            }
            xkeep[(ky + 1) & 1][ks] = x1;
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);

        // ---- epilogue, one output row (pt) at a time: bf16, a lane's 16-byte units (one v_permlane32_swap per dword pairs the register
        // quads g, g + 1 of the two half-waves) into the wave's region at [pixel][channel], read back with unit (lane & 7) of pixel
        // (lane >> 3) + 8 k per lane: 8 pixels = 1 KB contiguous per store instruction
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const f32x16& a = acc[nt][pt];
                    const uint32_t q0x = f2bf2(a[4 * g], a[4 * g + 1]), q0y = f2bf2(a[4 * g + 2], a[4 * g + 3]);
                    const uint32_t q1x = f2bf2(a[4 * g + 4], a[4 * g + 5]), q1y = f2bf2(a[4 * g + 6], a[4 * g + 7]);
                    const auto rx = __builtin_amdgcn_permlane32_swap(q0x, q1x, false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(q0y, q1y, false, false);
                    *(uint4*)(xw + l31 * OPITCH + (nt * 32 + 8 * g + 8 * half) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                }
            const int oy = cur.oy0 + wave * 2 + pt;
            bf16_t* const orow = P.out + (((size_t)cur.b * P.H + min(oy, P.H - 1)) * P.W + cur.ox0) * 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pl = (lane >> 3) + 8 * k, cu = lane & 7;
                const uint4 v = *(const uint4*)(xw + pl * OPITCH + cu * 16);
                if (OFD_C7_ABL & 2) asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                else if (oy < P.H && cur.ox0 + pl < P.W) *(uint4*)(orow + pl * 64 + cu * 8) = v;
            }
        }

        // ---- the next tile's image (LDS operations of a wave execute in order: the reads above are done with the region)
        if (it + 1 < n_iter) write_x(xr, xw, lane);
        cur = nxt;
    }
}

}  // namespace c7

// 1 = shape not served (the caller falls through to the generic kernel)
int launch_conv7x7_c8_persist(const ConvParams& P, hipStream_t s) {
    using namespace c7;
    const char* e = getenv("OFD_CONV7_PERSIST");          // read per call (an A/B switch, as OFD_CONV_PC): 0 = the generic kernel
    if ((e && !atoi(e)) || P.dbg || P.Cout != 64 || P.n_src != 1 || P.Cin_total != 8 || P.src[0].mode != 0 || P.in_scale || P.residual || P.res_act ||
        P.gn_partial || P.split || P.residual_b || P.pool2)
        return 1;
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        OFD_HIP(hipFuncSetAttribute((const void*)conv7x7_c8_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        int dev = 0;
        hipDeviceProp_t prop;
        OFD_HIP(hipGetDevice(&dev));
        OFD_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    static const int grid_env = getenv("OFD_CONV7_GRID") ? atoi(getenv("OFD_CONV7_GRID")) : 0;
    const int ntiles = P.tiles_x * P.tiles_y * P.B;
    int grid = grid_env > 0 ? grid_env : 2 * (n_cu / 8 * 8);       // two workgroups per CU, a multiple of the 8 XCDs
    if (grid < 8) grid = 8;
    if (grid > ntiles) grid = ntiles;
    conv7x7_c8_persist_kernel<<<grid, NTHREADS, LDS_BYTES, s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

}  // namespace ofd
