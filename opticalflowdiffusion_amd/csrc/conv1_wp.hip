// 1x1 convolution as a streaming kernel (gfx950, MFMA 32x32x16 bf16, fp32 accumulate, NHWC bf16): the res_conv of a ResnetBlock
// with the block's output fused in (out = W x + b + SiLU(GN(h2)), DD:200,214), the Downsample conv on the pixel-unshuffled input
// (DD:95-99) and the attention output projections (DD:225,254).  These layers are HBM-bound (3.7 GB for 128 -> 64 at full
// resolution against 0.12 TFLOP); the shared-slab kernel of conv_igemm.hip ran them at 4.0 TB/s: per 8 x 32 tile it exposed two
// dependent load -> stage -> barrier round trips and an LDS transpose of the output.  Here
//   * the weights never move: a wave owns a 32-output-channel slice and keeps ALL its A fragments (Cin / 16 x 16 bytes per lane) in
//     registers for the whole launch;
//   * the input arrives by LDS-DMA (global_load_lds_dwordx4), a whole 128- or 64-pixel tile x Cin at a time, double buffered:
//     the tile of step t+1 is in flight while tile t computes -- ONE counted wait and ONE barrier per tile, no staging registers;
//   * the LDS image is [64-channel unit][pixel][128 B]; the DMA is lane-linear, so the bank swizzle (16-byte chunk ^ (pixel & 7))
//     is applied to the SOURCE address of each lane and again on the fragment read (cdna guide T2 / rule 21);
//   * the epilogue's second input (h2) is fetched one tile ahead with loads the compiler does not see (inline asm) into ONE register
//     set that is re-issued right after its last use, so that the compiler's vmcnt bookkeeping cannot drain the DMA in flight; the
//     fragment reads are inline-asm ds_read_b128 for the same reason (a visible LDS load waits for ALL outstanding LDS-DMA).  The
//     waits are counted by hand: per tile and wave the VMEM issue order is [DMA(t+1)] ... [h2(t+1)] [stores(t)], the top-of-tile
//     wait is vmcnt(#h2 + #stores), the wait before the epilogue vmcnt(#stores + #DMA).
// Persistent: workgroup w walks tiles w, w + G, ... (P.interleave; all resident workgroups inside one moving window of memory).
// Tiles are always full and stores never predicated (the hand-counted waits depend on it): when TILE does not divide H*W the LAST tile of
// a sample starts at H*W - TILE and recomputes the pixels it shares with the tile before it (same inputs, same arithmetic, the same
// bytes written twice; the host admits this only for same-size sources, an output that is none of the inputs, and never for the
// atomics of the FC form).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "conv_params.h"
#include "mfma_util.h"

namespace ofd {
namespace c1 {

constexpr int NTHREADS = 256;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for<I + 1, N>(fn);
    }
}

template <int OFF>
__device__ __forceinline__ void lds_read16(u4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm2(u4& a, u4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int F, int PEND>
__device__ __forceinline__ void lds_wait(u4* x) {      // ties the registers to the wait: no consumer may be scheduled above it
    if constexpr (F == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x[0]) : "n"(PEND) : "memory");
    else if constexpr (F == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x[0]), "+v"(x[1]) : "n"(PEND) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(PEND) : "memory");
}

__device__ __forceinline__ float silu_f(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

struct Unit {            // 64 input channels of one source; output pixel (b, y, x) reads source pixel b * bs + y * ys + x * xs + c0
    const bf16_t* ptr;   // + channel offset
    long bs;
    int ys, xs, c0;      // same-size source: (W, 1, 0); pixel-unshuffled sub-pixel (p1, p2): (2 SW, 2, p1 SW + p2)
    int stride;          // channels of the source tensor
};

struct Params {
    Unit unit[12];
    const bf16_t* weight;      // [Cin/8][Cout][8]
    const float* bias;
    const bf16_t* res_act;
    const float* res_scale;
    const float* res_shift;
    bf16_t* out;
    int B, H, W, Cout, ntiles;
    int cout0;                 // first output channel computed (the blocks of this launch start there; the tensor keeps its full pixel stride Cout)
    const float* fc_w;         // FC instantiation: the final 1x1 conv [2][Cout], its bias [2], its zeroed (B, 2, H, W) fp32 output
    const float* fc_b;
    float* fc_out;
    int interleave;            // 1: workgroup w walks tiles w, w + G, ... (the resident workgroups sweep ONE moving window of memory); 0: a contiguous range each
    // 1-D grid of gx * nb workgroups (gx tile walkers x nb output-channel blocks), channel block fastest and the nb blocks of a walker on ONE
    // XCD (workgroup i runs on XCD i % 8): the walker's input tile comes from HBM once and from that XCD's L2 for the other blocks, and every
    // workgroup is resident from the start.  (The 2-D grid ran the channel blocks one after the other: 768 -> 512 at 55 x 128 read its
    // input four times, in four rounds of workgroups.)  gx is a multiple of 8 or nb == 1.
    int gx, nb;
    // PL instantiations (data gradients, the mid attention's to_out): the epilogue input is a PLAIN residual (out = W x + b + r) and the output /
    // residual may be split between two tensors -- channels [0, split) in out / res_act with pixel stride `split`, [split, Cout) in out2 / res2
    // with stride Cout - split (split = 0: one tensor of stride Cout).  A side without a residual reads `zero64` (64 bytes of zeros, stride 0):
    // every wave issues the same number of loads per tile, which the hand-counted waits depend on.  out may BE the residual (in-place
    // accumulation): a pixel's residual is read one tile before that pixel is written, by the same wave.
    const bf16_t* res2;
    bf16_t* out2;
    int split;
    const bf16_t* zero64;
};

// NSG: 32-channel slices per workgroup (output block = 32 NSG channels), CIN, TILE pixels per step, RA: fused SiLU(affine(h2)) input,
// NCB: output blocks a workgroup computes from one staged tile (to_qkv: 3 x 128 channels -- the input is read once, not per block)
// FC (r03): the UNet's final 1x1 conv (64 -> 2, fp32) rides on the output tile -- each wave dots its 32 channels (as stored: bf16-rounded)
// with the two weight rows, one v_permlane32_swap + add joins the lane halves (lanes 0-31 end with channel 0's partial, 32-63 with
// channel 1's), and ONE float atomic per fragment adds it to the zeroed fp32 output (the other 32-channel wave adds the second addend:
// two addends onto zero, any order, same sum).  The 922 MB bf16 tensor between the two convs is neither written nor read.
template <int NSG, int CIN, int TILE, bool RA, int NCB = 1, bool FC = false, bool PL = false>
__global__ void __launch_bounds__(NTHREADS, CIN > 512 ? 1 : 2) conv1x1_wp_kernel(const Params P) {      // (768 input channels: 192 weight registers per wave, one wave per SIMD)
    constexpr int KS = CIN / 16, UNITS = CIN / 64, UNITB = TILE * 128, BUFB = UNITS * UNITB;
    constexpr int PG = 4 / NSG, F = (TILE / 32) / PG;                 // pixel groups of waves, 32-pixel fragments per wave
    constexpr int NPW = BUFB / 1024 / 4;                               // 1-KiB DMA pieces per wave and tile
    constexpr int NRA = RA ? F * 2 : 0, NST = FC ? F : F * 2 * NCB;    // per wave and tile: h2 loads, output stores (16 B per lane each; FC: one atomic per fragment instead)
    static_assert(!FC || (NCB == 1 && NSG == 2), "FC: two 32-channel waves per pixel");
    static_assert(TILE % 32 == 0 && (TILE / 32) % PG == 0 && BUFB % 4096 == 0, "tile shape");
    static_assert(!(RA && NCB > 1) && NRA + NST <= 63, "the fused epilogue input is for single-block launches; vmcnt is 6 bits");
    static_assert(!PL || (RA && !FC), "PL: the plain-residual form of the fused epilogue input");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int ns = wave % NSG, pg = wave / NSG;
    int wx = (int)blockIdx.x, wy = 0;                                 // tile walker, output-channel block
    if (P.nb > 1) {
        const int xcd = wx & 7, j = wx >> 3;
        wy = j % P.nb;
        wx = (j / P.nb) * 8 + xcd;
    }
    const int n0 = P.cout0 + wy * 32 * NSG * NCB, cb = n0 + 32 * ns;    // (block k of this workgroup: cb + 32 NSG k)

    // ---- weights: this wave's A fragments, for good
    bf16x8 wf[NCB][KS];
    float4 bias4[4];                                                  // (multi-block launches are bias-free: to_qkv, DD:222 -- host check)
#pragma unroll
    for (int k = 0; k < NCB; ++k)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[k][ks] = *(const bf16x8*)(P.weight + ((size_t)(ks * 2 + half) * P.Cout + cb + 32 * NSG * k + l31) * 8);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        bias4[g] = (NCB == 1 && P.bias) ? *(const float4*)(P.bias + cb + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    // a visible use right here: the compiler's wait for these loads lands before the loop, not (as vmcnt(0)) inside it
#pragma unroll
    for (int k = 0; k < NCB; ++k)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[k][ks]));
    if constexpr (NCB == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bias4[g].x), "+v"(bias4[g].y), "+v"(bias4[g].z), "+v"(bias4[g].w));
    }
    float fcw[FC ? 2 : 1][16], fcb = 0.0f;                             // FC: rows of the final conv for this lane's 16 channels; its bias (first wave of a pixel only)
    if constexpr (FC) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w4 = *(const float4*)(P.fc_w + k * P.Cout + cb + 8 * g + 4 * half);
                fcw[k][4 * g] = w4.x; fcw[k][4 * g + 1] = w4.y; fcw[k][4 * g + 2] = w4.z; fcw[k][4 * g + 3] = w4.w;
            }
        fcb = (ns == 0) ? P.fc_b[half] : 0.0f;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(fcw[k][i]));
        asm volatile("" : "+v"(fcb));
    }
    const int plane = P.H * P.W, tps = (plane + TILE - 1) / TILE;      // tiles per sample (the last one overlaps its predecessor when TILE does not divide the plane)
    auto tile_p0 = [&](int t, int b) { return min((t - b * tps) * TILE, plane - TILE); };
    // DMA lane constants: piece = 8 pixel rows x 128 B of one unit; lane -> row lane >> 3, 16-byte chunk (lane & 7) ^ (lane >> 3)
    const int d_row = lane >> 3, d_chunk = ((lane & 7) ^ (lane >> 3)) * 8;          // (elements)
    auto issue_dma = [&](int t, int buf) {
        const int b = t / tps, p0 = tile_p0(t, b);                     // first pixel of the tile inside its sample
        const int y = p0 / P.W, x0 = p0 - y * P.W;                     // (TILE divides W when an unshuffled source is present: host check)
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            // piece j * 4 + wave: the unit is a compile-time function of j (descriptor in SGPRs), the row block carries the wave
            constexpr int PPU = TILE / 8;
            const int u = (j * 4) / PPU, rb = (j * 4) % PPU + wave;
            const Unit& U = P.unit[u];
            const size_t pix = (size_t)b * U.bs + (size_t)y * U.ys + x0 * U.xs + U.c0;
            const bf16_t* src = U.ptr + (pix + (size_t)((rb * 8 + d_row) * U.xs)) * U.stride + d_chunk;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(smem + buf * BUFB + u * UNITB + rb * 1024), 16, 0, 0);
        }
    };
    // this wave's 32-channel slice of the output and of the epilogue input: tensor, pixel stride, first channel inside it (PL: either side of a
    // split; a side without a residual reads the zero block at stride 0)
    bf16_t* o_base = P.out;
    const bf16_t* r_base = P.res_act;
    int o_stride = P.Cout, o_c0 = cb, r_stride = P.Cout, r_c0 = cb;
    if constexpr (PL) {
        const bool second = P.split > 0 && cb >= P.split;
        o_base = second ? P.out2 : P.out;
        o_stride = P.split > 0 ? (second ? P.Cout - P.split : P.split) : P.Cout;
        o_c0 = cb - (second ? P.split : 0);
        r_base = second ? P.res2 : P.res_act;
        r_stride = o_stride; r_c0 = o_c0;
        if (!r_base) { r_base = P.zero64; r_stride = 0; r_c0 = 0; }
    }
    // h2 (PL: the residual) of the wave's pixels, one tile ahead, through loads the compiler does not count
    u4 ra[RA ? NRA : 1];
    auto issue_ra = [&](int t) {
        if constexpr (RA) {
            const int b = t / tps;
            const size_t gp0 = (size_t)b * plane + tile_p0(t, b);     // global pixel index (tiles never straddle samples)
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const bf16_t* p = r_base + (gp0 + (pg * F + f) * 32 + l31) * (size_t)r_stride + r_c0 + 16 * gi + 8 * half;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[f * 2 + gi]) : "v"(p) : "memory");
                }
        }
    };

    // tile walk: interleaved (t = w, w + G, ...) keeps the ~500 resident workgroups inside one moving window of a few MB -- DRAM pages
    // are streamed through once; contiguous per-workgroup ranges spread them over the whole tensor (measured 20 % slower on the
    // same pattern in resblock_out).  The price: the sample index changes every plane / TILE / G tiles instead of once or twice.
    const int t_step = P.interleave ? P.gx : 1;
    int t = P.interleave ? wx : (int)(((long)wx * P.ntiles) / P.gx);
    const int t_end = P.interleave ? P.ntiles : (int)(((long)(wx + 1) * P.ntiles) / P.gx);
    if (t >= t_end) return;
    int b_cur = -1;
    float4 sc4[RA ? 4 : 1], sh4[RA ? 4 : 1];
    issue_dma(t, 0);
    issue_ra(t);
    int buf = 0;
    // counted VMEM wait that also pins the h2 registers below it (their consumers cannot be scheduled above)
    auto wait_ra = [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        if constexpr (RA) {
            static_for<0, NRA / 2>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                wait_vm2<N>(ra[2 * i], ra[2 * i + 1]);
            });
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // first tile: only its own DMA and h2 loads are in flight
    // fragment read base of this lane: pixel row l31 (+ 32 f), chunk ((ks & 3) * 2 + half) ^ (l31 & 7) of unit ks >> 2
    const unsigned char* xlane = smem + (pg * F * 32 + l31) * 128;
    int cx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cx[q] = ((q * 2 + half) ^ (l31 & 7)) * 16;

    for (;;) {
        const int tn = t + t_step, tq = tn < t_end ? tn : t;                           // past the end: the same tile again (never used)
        // VMEM issue order per wave: ... [h2(t)] [stores(t-1)] | [DMA(t+1)] ... [h2(t+1)] [stores(t)] | ...   (in-order return)
        // here: all but the youngest NRA + NST have landed, i.e. this wave's DMA pieces of tile t
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRA + NST) : "memory");
        __builtin_amdgcn_s_barrier();                       // every wave's pieces of tile t are in LDS; buffer buf ^ 1 is free
        issue_dma(tq, buf ^ 1);

        const unsigned xaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(xlane + buf * BUFB);
        unsigned xa[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xa[q] = xaddr + cx[q];
        const int b = t / tps;
        const size_t gp0 = (size_t)b * plane + tile_p0(t, b);
        static_for<0, NCB>([&](auto cbc) {
        constexpr int cbk = decltype(cbc)::value;
        f32x16 acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[f][k] = 0.0f;
        // fragment reads by hand as well: an LDS load the compiler can see makes it wait for ALL LDS-DMA in flight (vmcnt(0))
        // before the first use, i.e. for tile t + 1.  Two register sets, the reads of k-step ks + 1 fly under the MFMAs of ks.
        u4 xs_[2][F];
        auto rd = [&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            static_for<0, F>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                lds_read16<(ks >> 2) * UNITB + f * 32 * 128>(xs_[ks & 1][f], xa[ks & 3]);
            });
        };
        rd(std::integral_constant<int, 0>{});
        static_for<0, KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks + 1 < KS) rd(std::integral_constant<int, ks + 1>{});
            constexpr int pend = (ks + 1 < KS) ? F : 0;
            u4* x = xs_[ks & 1];
            lds_wait<F, pend>(x);
#pragma unroll
            for (int f = 0; f < F; ++f)
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cbk][ks], __builtin_bit_cast(bf16x8, x[f]), acc[f], 0, 0, 0);
        });

        // ---- epilogue: bias, + SiLU(affine(h2)), bf16, 16-byte stores (one v_permlane32_swap per dword pairs two register quads)
        if constexpr (RA && !PL) {
            if (b != b_cur) {                                // (ordinary loads: the compiler's wait drains the queue -- once per sample)
                b_cur = b;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    sc4[g] = *(const float4*)(P.res_scale + (size_t)b * P.Cout + cb + 8 * g + 4 * half);
                    sh4[g] = *(const float4*)(P.res_shift + (size_t)b * P.Cout + cb + 8 * g + 4 * half);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {                // (the wait for them: inside this branch)
                    asm volatile("" : "+v"(sc4[g].x), "+v"(sc4[g].y), "+v"(sc4[g].z), "+v"(sc4[g].w));
                    asm volatile("" : "+v"(sh4[g].x), "+v"(sh4[g].y), "+v"(sh4[g].z), "+v"(sh4[g].w));
                }
            }
        }
        wait_ra(std::integral_constant<int, NST + NPW>{});   // h2(t): younger than it are stores(t-1) and DMA(t+1)
        uint2 qout[F][4];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            uint2 rq[4];
            if constexpr (RA) {
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const u4 t4 = ra[f * 2 + (g >> 1)];
                    const auto sx = __builtin_amdgcn_permlane32_swap(t4[0], t4[2], false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(t4[1], t4[3], false, false);
                    rq[g] = make_uint2(sx[0], sy[0]);
                    rq[g + 1] = make_uint2(sx[1], sy[1]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {acc[f][4 * g] + bias4[g].x, acc[f][4 * g + 1] + bias4[g].y, acc[f][4 * g + 2] + bias4[g].z, acc[f][4 * g + 3] + bias4[g].w};
                if constexpr (PL) {
                    v[0] += bf2f((bf16_t)(rq[g].x & 0xffffu));
                    v[1] += bf2f((bf16_t)(rq[g].x >> 16));
                    v[2] += bf2f((bf16_t)(rq[g].y & 0xffffu));
                    v[3] += bf2f((bf16_t)(rq[g].y >> 16));
                } else if constexpr (RA) {
                    const float4 sc = sc4[g], sh = sh4[g];
                    v[0] += silu_f(bf2f((bf16_t)(rq[g].x & 0xffffu)) * sc.x + sh.x);
                    v[1] += silu_f(bf2f((bf16_t)(rq[g].x >> 16)) * sc.y + sh.y);
                    v[2] += silu_f(bf2f((bf16_t)(rq[g].y & 0xffffu)) * sc.z + sh.z);
                    v[3] += silu_f(bf2f((bf16_t)(rq[g].y >> 16)) * sc.w + sh.w);
                }
                qout[f][g] = make_uint2(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]));
            }
        }
        issue_ra(tq);                                        // into the registers the lines above have just finished with
        if constexpr (FC) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float v0 = bf2f((bf16_t)(qout[f][g].x & 0xffffu)), v1 = bf2f((bf16_t)(qout[f][g].x >> 16));
                    const float v2 = bf2f((bf16_t)(qout[f][g].y & 0xffffu)), v3 = bf2f((bf16_t)(qout[f][g].y >> 16));
                    p0 += v0 * fcw[0][4 * g] + v1 * fcw[0][4 * g + 1] + v2 * fcw[0][4 * g + 2] + v3 * fcw[0][4 * g + 3];
                    p1 += v0 * fcw[1][4 * g] + v1 * fcw[1][4 * g + 1] + v2 * fcw[1][4 * g + 2] + v3 * fcw[1][4 * g + 3];
                }
                // X' = [p0 of lanes 0-31 | p1 of lanes 0-31], Y' = [p0 of lanes 32-63 | p1 of lanes 32-63]: X' + Y' = channel `half`'s sum over the wave's 32 channels
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
                const float part = __uint_as_float(sw[0]) + __uint_as_float(sw[1]) + fcb;
                const size_t pin = gp0 - (size_t)b * plane + (pg * F + f) * 32 + l31;          // pixel inside its sample (tiles never straddle samples)
                unsafeAtomicAdd(P.fc_out + ((size_t)b * 2 + half) * plane + pin, part);
            }
        } else {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const size_t pix = gp0 + (pg * F + f) * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto rx = __builtin_amdgcn_permlane32_swap(qout[f][g].x, qout[f][g + 1].x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(qout[f][g].y, qout[f][g + 1].y, false, false);
                *(uint4*)(o_base + pix * (size_t)o_stride + o_c0 + 32 * NSG * cbk + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
        }
        });
        if (tn >= t_end) break;
        t = tn;
        buf ^= 1;
    }
    // the look-ahead DMA of the last step is still writing into this workgroup's LDS: it must land before the LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NSG, int CIN, int TILE, bool RA, int NCB = 1, bool FC = false, bool PL = false>
static int launch(const Params& P, hipStream_t s) {
    constexpr int LDS = 2 * (CIN / 64) * TILE * 128;
    static bool attr = false;
    if (!attr) {
        OFD_HIP(hipFuncSetAttribute((const void*)conv1x1_wp_kernel<NSG, CIN, TILE, RA, NCB, FC, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    const int per_cu = (LDS * 2 <= 160 * 1024) ? 2 : 1;
    const int nb = (P.Cout - P.cout0) / (32 * NSG * NCB);
    int gx = 256 * per_cu;
    // several channel blocks: all of them resident at once, walkers in whole groups of 8 (one per XCD); OFD_CONV1_NB_ROUNDS=1: a full set of
    // walkers per channel block as before (nb rounds of workgroups)
    static const int nb_rounds = getenv("OFD_CONV1_NB_ROUNDS") ? atoi(getenv("OFD_CONV1_NB_ROUNDS")) : 0;
    if (nb > 1 && !nb_rounds) gx = (gx / nb) / 8 * 8;
    if (const char* e = getenv("OFD_CONV1_GRID")) gx = atoi(e) > 0 ? atoi(e) : gx;       // diagnostics / tests: long tile ranges on small inputs
    if (gx > P.ntiles) gx = P.ntiles;
    if (nb > 1 && gx > 8) gx = gx / 8 * 8;
    Params Q = P;
    Q.gx = gx;
    Q.nb = (nb > 1 && gx % 8 == 0) ? nb : 1;
    if (nb > 1 && Q.nb == 1) {              // fewer than 8 walkers (tiny inputs, OFD_CONV1_GRID): the 2-D form, one channel block per launch row
        for (int k = 0; k < nb; ++k) {
            Params R = Q;
            R.cout0 = P.cout0 + k * 32 * NSG * NCB;
            conv1x1_wp_kernel<NSG, CIN, TILE, RA, NCB, FC, PL><<<gx, NTHREADS, LDS, s>>>(R);
        }
        OFD_LAUNCH_CHECK();
        return OFD_OK;
    }
    conv1x1_wp_kernel<NSG, CIN, TILE, RA, NCB, FC, PL><<<gx * Q.nb, NTHREADS, LDS, s>>>(Q);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

}  // namespace c1

// returns 1 when the shape is not one this kernel serves (the caller falls back to conv_igemm.hip), else the launch status
int launch_conv1x1_wp(const ConvParams& C, hipStream_t s) {
    using namespace c1;
    if (C.in_scale || C.gn_partial) return 1;
    // plain residual(s) and / or a split output (the 1x1 data gradients of the training backward, the mid attention's to_out): the PL instantiations
    const char* e_pl = getenv("OFD_CONV1_NO_PL");          // read per call (A/B switch): 1 = these go to the shared-slab kernel as before
    const bool no_pl = e_pl && atoi(e_pl);
    const bool pl = (C.residual || C.residual2 || C.split > 0) && !C.res_act;
    if (!pl && (C.residual || C.residual2 || C.split || C.out2)) return 1;
    if (pl && (no_pl || C.fc_out || C.residual_b || C.pool2 || C.cout0 || (C.split > 0 && (!C.out2 || C.split % 32 != 0 || C.split >= C.Cout)))) return 1;
    if (C.res_act && !(C.res_scale && C.res_shift)) return 1;
    const int cin = C.Cin_total, plane = C.H * C.W;
    const bool ra = C.res_act != nullptr;
    if (pl) {
        const bool wide_ok = C.Cout % 128 == 0;
        if (!((cin == 64 && wide_ok) || (cin == 128 && C.Cout % 64 == 0) || (cin == 256 && wide_ok) || (cin == 512 && wide_ok))) return 1;
    } else if (cin % 64 != 0 || (cin > 384 && !(cin == 512 && !ra) && !(cin == 768 && ra)) || cin == 320 || (cin < 128 && !(cin == 64 && C.Cout == 384 && !ra && !C.bias)) || (C.Cout != 64 && C.Cout % 128 != 0)) return 1;
    const int tile = (cin <= 128) ? 128 : (cin >= 384 ? 32 : 64);
    if (plane < tile || (size_t)C.B * plane * (size_t)(C.Cout > cin ? C.Cout : cin) * 2 >= (1ull << 40)) return 1;
    const bool ragged = plane % tile != 0;                        // the last tile of a sample overlaps the one before it (see the header)
    if (ragged && (C.fc_out || (const bf16_t*)C.out == C.res_act || (pl && (C.out == C.residual || (C.out2 && C.out2 == C.residual2))))) return 1;
    Params P{};
    int nu = 0;
    for (int i = 0; i < C.n_src; ++i) {
        const ConvSrcDev& S = C.src[i];
        if (S.mode == 1) return 1;
        if (S.mode == 2 && (C.W % tile != 0)) return 1;
        if (ragged && (S.mode != 0 || S.ptr == (const bf16_t*)C.out || (C.out2 && S.ptr == (const bf16_t*)C.out2))) return 1;
        for (int k = 0; k < S.chunks; ++k) {
            if (nu >= 12) return 1;
            Unit& U = P.unit[nu++];
            U.ptr = S.ptr + S.ch_offset + k * 64;
            U.stride = S.src_channels;
            if (S.mode == 2) { U.bs = (long)S.SH * S.SW; U.ys = 2 * S.SW; U.xs = 2; U.c0 = S.p1 * S.SW + S.p2; }
            else { U.bs = plane; U.ys = C.W; U.xs = 1; U.c0 = 0; }
        }
    }
    if (nu * 64 != cin) return 1;
    P.weight = C.weight; P.bias = C.bias; P.res_act = C.res_act; P.res_scale = C.res_scale; P.res_shift = C.res_shift; P.out = C.out;
    P.fc_w = C.fc_w; P.fc_b = C.fc_b; P.fc_out = C.fc_out;
    if (C.fc_out && !(cin == 128 && C.Cout == 64 && ra)) return 1;
    P.B = C.B; P.H = C.H; P.W = C.W; P.Cout = C.Cout; P.ntiles = C.B * ((plane + tile - 1) / tile);
    static const int order = getenv("OFD_CONV1_ORDER") ? atoi(getenv("OFD_CONV1_ORDER")) : 1;
    P.interleave = order;
    const bool narrow = C.Cout == 64;
    P.cout0 = 0;
    if (pl) {
        static const bf16_t* zero64 = nullptr;             // 64 bytes of zeros: what a side without a residual reads (stride 0)
        if (!zero64) {
            void* z = nullptr;
            OFD_HIP(hipMalloc(&z, 64));
            OFD_HIP(hipMemset(z, 0, 64));
            zero64 = (const bf16_t*)z;
        }
        P.res_act = C.residual; P.res2 = C.residual2; P.out2 = C.out2; P.split = C.split; P.zero64 = zero64;
        P.res_scale = nullptr; P.res_shift = nullptr;
        if (cin == 64) return launch<4, 64, 128, true, 1, false, true>(P, s);
        if (cin == 128) return C.Cout % 128 == 0 ? launch<4, 128, 128, true, 1, false, true>(P, s) : launch<2, 128, 128, true, 1, false, true>(P, s);
        if (cin == 256) return launch<4, 256, 64, true, 1, false, true>(P, s);
        return launch<4, 512, 32, true, 1, false, true>(P, s);
    }
    if (cin == 64 && C.cout0 == 128) { P.cout0 = 128; return launch<4, 64, 128, false, 2>(P, s); }      // ... its k and v blocks only (q recomputed downstream)
    if (cin == 64) return launch<4, 64, 128, false, 3>(P, s);                 // to_qkv of the 64-channel LinearAttention (training)
    if (cin == 128 && C.Cout == 384 && !ra && !C.bias) return launch<4, 128, 128, false, 3>(P, s);
    if (cin == 128) {
        if (narrow && C.fc_out) return launch<2, 128, 128, true, 1, true>(P, s);
        if (narrow) return ra ? launch<2, 128, 128, true>(P, s) : launch<2, 128, 128, false>(P, s);
        return ra ? 1 : launch<4, 128, 128, false>(P, s);      // (no 128 -> 128+ res_conv in this UNet)
    }
    if (cin == 192) {
        if (narrow) return 1;
        return ra ? launch<4, 192, 64, true>(P, s) : launch<4, 192, 64, false>(P, s);
    }
    if (cin == 512) return narrow ? 1 : launch<4, 512, 32, false>(P, s);        // to_qkv 512 -> 384, down-sample 512 -> 256
    if (cin == 768) return (narrow || !ra) ? 1 : launch<4, 768, 32, true>(P, s); // res_conv 768 -> 512 of ups.0 (r03: was the shared-slab kernel at a third of the HBM roof)
    if (cin == 384) {
        if (narrow) return 1;
        return ra ? launch<4, 384, 32, true>(P, s) : launch<4, 384, 32, false>(P, s);
    }
    if (narrow) return ra ? launch<2, 256, 64, true>(P, s) : launch<2, 256, 64, false>(P, s);
    return ra ? launch<4, 256, 64, true>(P, s) : launch<4, 256, 64, false>(P, s);
}

}  // namespace ofd
