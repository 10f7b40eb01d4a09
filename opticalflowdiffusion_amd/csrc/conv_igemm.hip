// Implicit-GEMM convolution for gfx950 (MFMA 32x32x16 bf16, fp32 accumulate), NHWC bf16.
//
// Restates what the reference gets from cuDNN through F.conv2d for every convolution of the UNet
// (denoising_diffusion.py:92,98,114,200,222,225,253,254,297,339,354,361) plus the glue the
// reference runs as separate PyTorch ops around it, fused here:
//   prologue : GroupNorm-apply * (scale+1) + shift -> SiLU  (DD:181-187) as one per-(sample,
//              channel) affine + SiLU applied while the input tile is staged into LDS;
//              channel concat (DD:405,408,414), nearest x2 up-sampling (DD:91) and
//              pixel-unshuffle (DD:97) are address arithmetic of the loader;
//   epilogue : bias, residual add / "+ SiLU(affine(h))" (DD:214), bf16 store, and the per-tile
//              partial sums GroupNorm needs for the NEXT block (DD:181).
//
// Tiling: one workgroup (4 waves) owns 8x32 output pixels x BN output channels.  The input tile
// with its halo ((8+k-1)x(32+k-1) pixels x 64 channels) is staged ONCE per 64-channel chunk
// into LDS and reused by all k*k taps (LDS tile reuse instead of k*k global re-reads); weights
// stream per tap through a double-buffered LDS slab.  MFMA operands are read with ds_read_b128
// from XOR-swizzled images (conflict-free, see swz()).  The MFMA is issued "swapped"
// (A = weights, B = pixels) so that a lane's accumulator registers are 4 consecutive output
// channels of one pixel -> 8-byte NHWC stores.
#include <cstdlib>
#include "common.h"
#include "conv_params.h"
#include "blocks.h"

namespace ofd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TH = 8, TW = 32, NTHREADS = 256;

template <int KS, int BN>
struct Cfg {
    // KS == 8 is not a kernel size: it is the 7x7 conv over an input padded to 8 channels (16 bytes per pixel), where the two
    // channel octets of one MFMA k-step are two horizontally ADJACENT PIXELS, i.e. the taps (ky, 2j) and (ky, 2j+1): 4 k-steps per
    // kernel row instead of 7 (the 8th tap column carries zero weights).  Every other KS == 7 property applies.
    static constexpr bool K7P = (KS == 8);
    static constexpr int KSZ = K7P ? 7 : KS;                // the kernel size proper
    static constexpr int CK = K7P ? 8 : ((KS == 7) ? 16 : 64);          // channels per K-chunk
    static constexpr int NC = CK / 8;                       // 16-byte units per pixel
    static constexpr int PAD = KSZ / 2;
    static constexpr int IH = TH + KSZ - 1, IW = TW + KSZ - 1 + (K7P ? 1 : 0), NPIX = IH * IW;
    static constexpr int STAGES = (KS == 3) ? 9 : (KSZ == 7 ? 7 : (KS == 2 ? 4 : 1));   // weight slabs per chunk
    static constexpr int KSTEPS = (KS == 7) ? 7 : 4;                    // MFMA k-steps per slab
    static constexpr int SC8 = KSTEPS * 2;                              // 8-channel rows per slab
    // X tile in LDS is unit-major: [NC channel-octets][NPIX+1 slots][16 B] (see PP_US below)
    static constexpr int US = (NPIX + 1) * 16;
    static constexpr int X_BYTES = NC * US;
    static constexpr int W_BYTES = SC8 * BN * 16;
    // epilogue: every wave transposes its 64 pixels x BN channels through LDS so that the global stores are fully
    // coalesced (pixel pitch BN*2 + 16 bytes keeps the 16-byte LDS writes conflict-free)
    static constexpr int OPITCH = BN * 2 + 16;
    static constexpr int O_BYTES = 4 * 64 * OPITCH;
    static constexpr int LDS_BYTES = (KS != 1 || X_BYTES + 2 * W_BYTES > O_BYTES) ? X_BYTES + 2 * W_BYTES : O_BYTES;
    static constexpr int XPT = (NPIX * NC + NTHREADS - 1) / NTHREADS;  // 16-B units per thread
    static constexpr int WPT = (SC8 * BN + NTHREADS - 1) / NTHREADS;
    static constexpr int NTN = BN / 32;
    static constexpr int PPR = 256 / (CK * 2);              // pixels per 256-B LDS bank row
};

// XOR swizzle of the 16-byte unit index inside a pixel: 16 consecutive pixels read at the same
// logical unit by one ds_read_b128 lane group land on 16 distinct 16-B slots of the bank row.
template <int CK>
__device__ __forceinline__ int swz(int p) {
    return (CK == 64) ? ((p >> 1) & 7) : ((p >> 3) & 1);
}

__device__ __forceinline__ float silu_f(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

__device__ __forceinline__ uint32_t pack2(float a, float b) { return f2bf2(a, b); }

// reduce N per-lane values over the 64 lanes of a wave with ~N shuffles (butterfly that halves
// the value set per step): afterwards lane l holds in v[0] the total of value index
// (l >> (6 - log2 N)) (N = 32: l >> 1, N = 16: l >> 2).  Fully static indexing (no scratch).
template <int N, int NCUR, int OFF>
struct WaveReduce {
    static __device__ __forceinline__ void run(float (&v)[N]) {
        if constexpr (NCUR > 1) {
            constexpr int HALF = NCUR / 2;
            const bool upper = (threadIdx.x & OFF) != 0;
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
                const float send = upper ? v[i] : v[i + HALF];
                const float keep = upper ? v[i + HALF] : v[i];
                v[i] = keep + __shfl_xor(send, OFF, 64);
            }
            if constexpr (OFF > 1) WaveReduce<N, HALF, OFF / 2>::run(v);
        } else {
            v[0] += __shfl_xor(v[0], OFF, 64);
            if constexpr (OFF > 1) WaveReduce<N, 1, OFF / 2>::run(v);
        }
    }
};
template <int N>
__device__ __forceinline__ void wave_reduce_multi(float (&v)[N]) {
    WaveReduce<N, N, 32>::run(v);
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // first-class vector value: always in registers
template <int N>
struct U4Arr {
    u32x4 v[N];
};

template <int KS, int BN>
__device__ __forceinline__ void conv_load_w(U4Arr<Cfg<KS, BN>::WPT>& wr, const bf16_t* __restrict__ weight, int cin8, int Cout, int n0,
                                            int kc, int st, int tid) {
    using C = Cfg<KS, BN>;
    kc += st / C::STAGES;      // the slab index runs on into the following chunks
    st %= C::STAGES;
#pragma unroll
    for (int i = 0; i < C::WPT; ++i) {
        int u = tid + i * NTHREADS;
        if (C::SC8 * BN % NTHREADS != 0) u = min(u, C::SC8 * BN - 1);
        const int r = u / BN, n = u % BN;
        size_t row;
        if (KS == 7) row = (size_t)(st * 7 + r / 2) * 2 + (r & 1);
        else if (KS == 8) row = (size_t)st * 8 + r;
        else row = (size_t)st * cin8 + kc * 8 + r;
        wr.v[i] = *(const u32x4*)(weight + (row * Cout + n0 + n) * 8);   // (diagnostic bit 2 handled by the caller)
    }
}
template <int KS, int BN>
__device__ __forceinline__ void conv_store_w(const U4Arr<Cfg<KS, BN>::WPT>& wr, unsigned char* dst, int tid) {
    using C = Cfg<KS, BN>;
#pragma unroll
    for (int i = 0; i < C::WPT; ++i) {
        int u = tid + i * NTHREADS;
        if (C::SC8 * BN % NTHREADS != 0) u = min(u, C::SC8 * BN - 1);
        *(u32x4*)(dst + u * 16) = wr.v[i];
    }
}

template <int KS, int BN>
__global__ void __launch_bounds__(NTHREADS, 2) conv_igemm_kernel(const ConvParams P) {
    using C = Cfg<KS, BN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_x = smem;
    unsigned char* lds_w = smem + C::X_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware tile order: blocks that share an XCD (bid % 8) get a contiguous run of tiles, so
    // the halo rows two neighbouring tiles share are served by one L2.
    const int ntiles = P.tiles_x * P.tiles_y * P.B;
    int tile = blockIdx.x;
    // KS == 2, all four phases in this launch: phase from the block index, the per-phase fields derived here
    int ph_pad_y = P.pad_y, ph_pad_x = P.pad_x, ph_oy = P.out_oy, ph_ox = P.out_ox;
    const bf16_t* ph_weight = P.weight;
    if (KS == 2 && P.phase_all) {
        const int j = blockIdx.x, g = j >> 3, ph = g & 3;
        tile = (g >> 2) * 8 + (j & 7);
        if (tile >= ntiles) return;                       // (grid padded to whole groups of 8 tiles x 4 phases; uniform per workgroup)
        ph_pad_y = 1 - (ph >> 1); ph_pad_x = 1 - (ph & 1); ph_oy = ph >> 1; ph_ox = ph & 1;
        ph_weight = P.weight + (size_t)ph * 4 * P.Cin_total * P.Cout;
    }
    if (ntiles >= 8) {
        const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int b = tile / (P.tiles_x * P.tiles_y);
    const int t_in = tile % (P.tiles_x * P.tiles_y);
    const int oy0 = (t_in / P.tiles_x) * TH, ox0 = (t_in % P.tiles_x) * TW;
    const int n0 = blockIdx.y * BN;

    const unsigned char* xrow = lds_x + half * (C::K7P ? 16 : C::US) + (wave * 2 * C::IW + l31) * 16;

    f32x16 acc[C::NTN][2];
#pragma unroll
    for (int i = 0; i < C::NTN; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.0f;

    // 1x1 with the ResnetBlock output fused in (res_act): the activation tile the epilogue adds is independent of the K loop, so
    // its loads are issued here and land while the input tiles stream (a workgroup is otherwise a serial chain of three
    // memory round trips: X chunk 0, X chunk 1, res_act).  Only the 64-channel instantiation has the registers.
    constexpr bool PRE = (KS == 1 && BN == 64);
    uint4 pre_ra[PRE ? 2 : 1][PRE ? C::NTN : 1][2];
    if constexpr (PRE) {
        if (P.res_act) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int oy = oy0 + wave * 2 + pt, ox = ox0 + l31;
                const size_t pix = ((size_t)b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1);
#pragma unroll
                for (int nt = 0; nt < C::NTN; ++nt)
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi)
                        pre_ra[pt][nt][gi] = *(const uint4*)(P.res_act + pix * P.Cout + n0 + nt * 32 + 16 * gi + 8 * half);
            }
        }
    }

    // Weight pipeline: slab g = kc*STAGES + st lives in registers for two stages before it is written
    // to its LDS buffer (g & 1): the global load of slab g+2 is issued at the top of stage g and
    // consumed at the end of stage g+1, so ~2 stages of MFMA work cover the L2 latency.  The
    // pipeline runs on across chunk boundaries.  No predication anywhere in the staging code:
    // indices past the end are clamped (duplicates rewrite identical bytes).
    // (BN = 128 has no registers to spare for the second set: there the load of slab g+1 is issued
    // at the top of stage g and consumed at its end.)
    constexpr bool DEEP = (BN == 64);
    U4Arr<C::WPT> w1, w2;     // slabs g+1 and g+2
    const int cin8 = P.Cin_total / 8;
    const int total_slabs = P.total_chunks * C::STAGES;
    const int dbg = P.dbg;
    conv_load_w<KS, BN>(w1, ph_weight, cin8, P.Cout, n0, 0, 0, tid);                                   // slab 0
    if (DEEP) conv_load_w<KS, BN>(w2, ph_weight, cin8, P.Cout, n0, 0, (total_slabs > 1) ? 1 : 0, tid);   // slab 1

    int src_i = 0, src_first = 0;   // source that owns chunk kc, and its first chunk
    for (int kc = 0; kc < P.total_chunks; ++kc) {
        while (kc >= src_first + P.src[src_i].chunks) {
            src_first += P.src[src_i].chunks;
            ++src_i;
        }
        const bf16_t* s_ptr = P.src[src_i].ptr;
        const int s_ch = P.src[src_i].src_channels, s_off = P.src[src_i].ch_offset, s_SH = P.src[src_i].SH, s_SW = P.src[src_i].SW;
        const int s_mode = P.src[src_i].mode, s_p1 = P.src[src_i].p1, s_p2 = P.src[src_i].p2;
        const int kcl = kc - src_first;
        const int g0 = kc * C::STAGES;

        // ---- stage the input tile (+halo) of this chunk: global -> regs -> (affine+SiLU) -> LDS
        uint4 xreg[C::XPT];
        unsigned okmask = 0;
        const int c8 = tid % C::NC;
        const bf16_t* s_base = s_ptr + (size_t)b * s_SH * s_SW * s_ch + s_off + kcl * C::CK + c8 * 8;
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int p = min(tid / C::NC + i * (NTHREADS / C::NC), C::NPIX - 1);
            const int ty = p / C::IW, tx = p - ty * C::IW;
            const int iy = oy0 - (KS == 2 ? ph_pad_y : C::PAD) + ty, ix = ox0 - (KS == 2 ? ph_pad_x : C::PAD) + tx;
            const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
            okmask |= (ok ? 1u : 0u) << i;
            const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);
            int sy = cy, sx = cx;
            if (s_mode == 1) { sy = cy >> 1; sx = cx >> 1; }
            else if (s_mode == 2) { sy = 2 * cy + s_p1; sx = 2 * cx + s_p2; }
            xreg[i] = (dbg & 1) ? make_uint4(tid, i, 0, 0) : *(const uint4*)(s_base + ((size_t)sy * s_SW + sx) * s_ch);
        }
        float ps[8], pb[8];
        if (P.in_scale) {
            const int cg = kc * C::CK + c8 * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ps[j] = P.in_scale[(size_t)b * P.Cin_total + cg + j];
                pb[j] = P.in_shift[(size_t)b * P.Cin_total + cg + j];
            }
        }
        __syncthreads();   // every wave has finished reading lds_x (and the W buffer of slab g0-1)
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int p = min(tid / C::NC + i * (NTHREADS / C::NC), C::NPIX - 1);
            uint4 v = xreg[i];
            if (P.in_scale) {
                uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = silu_f(bf2f((bf16_t)(w4[j] & 0xffffu)) * ps[2 * j] + pb[2 * j]);
                    const float hi = silu_f(bf2f((bf16_t)(w4[j] >> 16)) * ps[2 * j + 1] + pb[2 * j + 1]);
                    w4[j] = pack2(lo, hi);
                }
                v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            const bool ok = (okmask >> i) & 1u;                 // zero padding stays zero (applied AFTER the prologue)
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            *(uint4*)(lds_x + c8 * C::US + p * 16) = v;
        }
        if (kc == 0) {     // slab 0 -> LDS buffer 0; rotate: w1 <- slab 1, w2 <- slab 2
            conv_store_w<KS, BN>(w1, lds_w, tid);
            if (DEEP) {
                w1 = w2;
                conv_load_w<KS, BN>(w2, ph_weight, cin8, P.Cout, n0, 0, (total_slabs > 2) ? 2 : 0, tid);
            }
        }

        // ---- stages of this chunk (BN = 64: fully unrolled, tap offsets become compile-time constants)
        constexpr int UNR = DEEP ? C::STAGES : 1;
#pragma unroll UNR
        for (int st = 0; st < C::STAGES; ++st) {
            const int g = g0 + st;
            if (!DEEP && g + 1 < total_slabs) conv_load_w<KS, BN>(w1, ph_weight, cin8, P.Cout, n0, kc, st + 1, tid);
            __syncthreads();   // lds_x (st == 0) and the W buffer of slab g are complete
            const unsigned char* wbuf = lds_w + (g & 1) * C::W_BYTES;
            const int ky = (KS == 3) ? st / 3 : (C::KSZ == 7 ? st : (KS == 2 ? st / 2 : 0));
            const int kx3 = (KS == 3) ? st % 3 : (KS == 2 ? st % 2 : 0);
#pragma unroll
            for (int ks = 0; ks < C::KSTEPS; ++ks) {
                const int kx = (KS == 7) ? ks : (C::K7P ? 2 * ks : kx3);
                const int unit = (C::KSZ == 7) ? half : (ks * 2 + half);
                bf16x8 xf[2], wf[C::NTN];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)   // base + (tap, k-step) offset: no per-read address math
                    xf[pt] = *(const bf16x8*)(xrow + (pt * C::IW + ky * C::IW + kx) * 16 + (unit - half) * C::US);
#pragma unroll
                for (int nt = 0; nt < C::NTN; ++nt)
                    wf[nt] = *(const bf16x8*)(wbuf + ((ks * 2 + half) * BN + nt * 32 + l31) * 16);
                if (!(dbg & 4)) {
                    // the MFMA issue of this wave goes ahead of the other resident workgroup's staging / epilogue instructions on the
                    // same SIMD (same-box A/B: <3,128> 12.41 -> 11.82 ms per denoise step, <3,64> 4.13 -> 3.90)
                    if (!(dbg & 128)) __builtin_amdgcn_s_setprio(2);
#pragma unroll
                    for (int nt = 0; nt < C::NTN; ++nt)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt)
                            acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[pt], acc[nt][pt], 0, 0, 0);
                    if (!(dbg & 128)) __builtin_amdgcn_s_setprio(0);
                } else {
#pragma unroll
                    for (int nt = 0; nt < C::NTN; ++nt) asm volatile("" ::"v"(wf[nt]));
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) asm volatile("" ::"v"(xf[pt]));
                }
            }
            // slab g+1 (loaded two stages ago) -> the other LDS buffer (last read in stage g-1, and
            // every wave is past this stage's barrier); then start the load of slab g+3
            if (g + 1 < total_slabs) conv_store_w<KS, BN>(w1, lds_w + ((g + 1) & 1) * C::W_BYTES, tid);
            if (DEEP) {
                w1 = w2;
                if (g + 3 < total_slabs) conv_load_w<KS, BN>(w2, ph_weight, cin8, P.Cout, n0, kc, st + 3, tid);
            }
        }
    }

    // ---- epilogue: bias, residual forms, bf16 store, GroupNorm partial sums ---------------------
    // Stores: a wave's accumulators give 32 contiguous bytes per pixel and instruction, a pattern that writes HBM at
    // 2.9 TB/s (tools/probe/store_pattern_probe.hip); transposed through a private LDS region (64 pixels x BN channels per
    // wave) consecutive lanes write consecutive 16-byte units of a pixel row: 5.3 - 5.7 TB/s.
    // Measured on one box (A/B): the 1x1 convs gain 4.5 % (64->384: 2.0 -> 1.6 ms), the MFMA-bound 3x3 / 7x7 kernels lose
    // 0.5 - 1 % to the extra LDS pass, so only the 1x1 instantiations take this path.
    constexpr bool XPOSE = (KS == 1);
    if (XPOSE) __syncthreads();                        // every wave is done with the operand buffers the regions overlap
    unsigned char* const oreg = smem + wave * 64 * C::OPITCH;
    constexpr int NV = (BN / 8) * 2;
    float stat[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) stat[i] = 0.0f;

#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int oy = oy0 + wave * 2 + pt, ox = ox0 + l31;
        const bool ok = oy < P.H && ox < P.W && !(dbg & 16);
        const size_t pix = ((size_t)b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1);   // clamped: loads stay in bounds
        // KS == 2: this launch is one phase of an up-sampled 3x3 -> the output pixel is (2y + oy, 2x + ox) of a (2H, 2W) tensor
        const size_t opix = (KS == 2) ? ((size_t)b * (2 * P.H) + 2 * min(oy, P.H - 1) + ph_oy) * (2 * P.W) + 2 * min(ox, P.W - 1) + ph_ox : pix;
#pragma unroll
        for (int nt = 0; nt < C::NTN; ++nt) {
            // destination of this 32-channel block (uniform per workgroup and nt): the second one when the output is split
            const bool second = P.split > 0 && n0 + nt * 32 >= P.split;
            bf16_t* const o_base = second ? P.out2 : P.out;
            const bf16_t* const r_base = second ? P.residual2 : P.residual;
            const int o_stride = P.split > 0 ? (second ? P.Cout - P.split : P.split) : P.Cout;
            const int o_c0 = n0 + nt * 32 - (second ? P.split : 0);
            uint2 q[4];
            // epilogue inputs as 16-byte loads: a lane reads channels 8g + 8*half .. +7 (g even) and one permlane32_swap per
            // dword hands every lane the two register quads (8g + 4*half, 8(g+1) + 4*half) it accumulates
            uint2 ra[4], rr[4];
            if (P.res_act) {
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    uint4 t4;
                    if constexpr (PRE) t4 = pre_ra[pt][nt][g >> 1];
                    else t4 = *(const uint4*)(P.res_act + pix * P.Cout + n0 + nt * 32 + 8 * g + 8 * half);
                    const auto sx = __builtin_amdgcn_permlane32_swap(t4.x, t4.z, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(t4.y, t4.w, false, false);
                    ra[g] = make_uint2(sx[0], sy[0]);
                    ra[g + 1] = make_uint2(sx[1], sy[1]);
                }
            }
            if (r_base) {
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const uint4 t4 = *(const uint4*)(r_base + pix * o_stride + o_c0 + 8 * g + 8 * half);
                    const auto sx = __builtin_amdgcn_permlane32_swap(t4.x, t4.z, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(t4.y, t4.w, false, false);
                    rr[g] = make_uint2(sx[0], sy[0]);
                    rr[g + 1] = make_uint2(sx[1], sy[1]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n0 + nt * 32 + 8 * g + 4 * half;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[nt][pt][4 * g + j];
                if (P.bias) {
                    const float4 bv = *(const float4*)(P.bias + c);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (P.res_act) {
                    const uint2 r = ra[g];
                    const float4 sc = *(const float4*)(P.res_scale + (size_t)b * P.Cout + c);
                    const float4 sh = *(const float4*)(P.res_shift + (size_t)b * P.Cout + c);
                    v[0] += silu_f(bf2f((bf16_t)(r.x & 0xffffu)) * sc.x + sh.x);
                    v[1] += silu_f(bf2f((bf16_t)(r.x >> 16)) * sc.y + sh.y);
                    v[2] += silu_f(bf2f((bf16_t)(r.y & 0xffffu)) * sc.z + sh.z);
                    v[3] += silu_f(bf2f((bf16_t)(r.y >> 16)) * sc.w + sh.w);
                }
                if (r_base) {
                    const uint2 r = rr[g];
                    v[0] += bf2f((bf16_t)(r.x & 0xffffu));
                    v[1] += bf2f((bf16_t)(r.x >> 16));
                    v[2] += bf2f((bf16_t)(r.y & 0xffffu));
                    v[3] += bf2f((bf16_t)(r.y >> 16));
                }
                q[g] = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                if (P.gn_partial && ok) {   // statistics of the values as stored (bf16), DD:181
                    const float q0 = bf2f((bf16_t)(q[g].x & 0xffffu)), q1 = bf2f((bf16_t)(q[g].x >> 16));
                    const float q2 = bf2f((bf16_t)(q[g].y & 0xffffu)), q3 = bf2f((bf16_t)(q[g].y >> 16));
                    stat[(nt * 4 + g) * 2] += (q0 + q1) + (q2 + q3);
                    stat[(nt * 4 + g) * 2 + 1] += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
                }
            }
            // 16-byte stores: swap quad g's upper-half data with quad g+1's lower-half data, so the lower
            // half-wave holds channels 8g..8g+7 and the upper half-wave 8(g+1)..8(g+1)+7 of its pixel
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
                if (!XPOSE) { if (ok) *(uint4*)(o_base + opix * o_stride + o_c0 + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]); }
                else *(uint4*)(oreg + (pt * 32 + l31) * C::OPITCH + (nt * 32 + 8 * g + 8 * half) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
    }

    if (XPOSE) {
        // copy-out of this wave's region: unit u of pixel p -> lane; UPR consecutive lanes cover one pixel row
        constexpr int UPR = BN / 8;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the wave's own LDS writes (in order per wave)
#pragma unroll
        for (int k = 0; k < UPR; ++k) {
            const int id = lane + k * 64, pl = id / UPR, cu = id % UPR;       // pl: 0..63 = (row pt, column x)
            const int oy = oy0 + wave * 2 + (pl >> 5), ox = ox0 + (pl & 31);
            const bool ok = oy < P.H && ox < P.W && !(dbg & 16);
            const size_t pix = ((size_t)b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1);
            const size_t opix = (KS == 2) ? ((size_t)b * (2 * P.H) + 2 * min(oy, P.H - 1) + ph_oy) * (2 * P.W) + 2 * min(ox, P.W - 1) + ph_ox : pix;
            const int c0 = n0 + cu * 8;
            const bool second = P.split > 0 && c0 >= P.split;
            bf16_t* const o_base = second ? P.out2 : P.out;
            const int o_stride = P.split > 0 ? (second ? P.Cout - P.split : P.split) : P.Cout;
            const uint4 v = *(const uint4*)(oreg + pl * C::OPITCH + cu * 16);
            if (ok) *(uint4*)(o_base + opix * o_stride + c0 - (second ? P.split : 0)) = v;
        }
    }
    if (P.gn_partial) {   // per-wave partial sums, slot (tile, wave) of the layout at gn_partial_index (conv_params.h); gn_finalize adds them up
        wave_reduce_multi<NV>(stat);
        constexpr int SH_ = (NV == 32) ? 1 : 2;    // lane l holds value index l >> SH_ = (octet of the channel block) * 2 + (sum | sum of squares)
        if ((lane & ((1 << SH_) - 1)) == 0) {
            const int vi = lane >> SH_;
            P.gn_partial[gn_partial_index(b, P.tiles_x * P.tiles_y * 4, t_in * 4 + wave, P.Cout / 8, n0 / 8 + (vi >> 1)) + (vi & 1)] = stat[0];
        }
    }
}

// ================================================================================================
// Persistent ping-pong variant for the full-resolution workhorse: 3x3, Cin = Cout = 64, one source.
// The generic kernel above re-streams the 73.7 KB of weights for every 8x32 tile and runs its
// phases (stage X, 9 weight slabs, epilogue) back to back, which leaves the MFMA pipe idle ~75 %
// of the time at N = 64.  Here a workgroup of 8 waves stays resident (one per CU):
//   * all 9 taps of weights live in LDS for the whole launch (loaded once);
//   * the waves form two groups of 4 that work on neighbouring tiles in opposite phases: while
//     group A issues the 144 MFMAs per wave of its tile, group B runs its epilogue (bias, bf16
//     store, GroupNorm partials) and stages its next input tile (prologue affine+SiLU, LDS
//     write) -- then they swap.  One workgroup barrier per phase;
//   * the global loads of a group's NEXT input tile are issued at the start of its MFMA phase
//     and consumed in the following staging phase, so their latency hides behind the MFMAs.
// LDS: 73,728 B weights + 2 x 43,520 B input tiles = 160,768 B.
constexpr int PP_THREADS = 512;
// X tile in LDS is unit-major: [8 channel-octets][341 pixel slots][16 B].  Consecutive pixels of one
// octet are consecutive 16-B slots (conflict-free ds_read_b128 with no swizzle), taps and k-steps are
// immediate offsets from ONE base address per output row, and 341 (odd multiple of 4 banks off 32)
// keeps the 8 octets of a pixel on distinct banks for the staging writes.
constexpr int PP_US = 341 * 16;   // bytes per octet row
constexpr int PP_XB = 8 * PP_US, PP_WB = 9 * 8 * 64 * 16, PP_LDS = PP_WB + 2 * PP_XB + 256;

struct PPTile {
    int b, oy0, ox0, t_in;
    bool valid;
};

__device__ __forceinline__ PPTile pp_tile(int t, int ntiles, int tiles_x, int tpi) {
    PPTile r;
    r.valid = t < ntiles;
    t = min(t, ntiles - 1);
    r.b = t / tpi;
    r.t_in = t - r.b * tpi;
    r.oy0 = (r.t_in / tiles_x) * TH;
    r.ox0 = (r.t_in % tiles_x) * TW;
    return r;
}

struct PPX {
    u32x4 v[11];
    unsigned okmask;
};

// staging map: wave w of a group owns the 85 tile pixels [85 w, 85 w + 85) (8 pixels x 8 channel octets per instruction, 11
// instructions; the surplus 3 re-write the last pixel).  Owning a CONTIGUOUS range is what lets the epilogue borrow exactly
// the bytes this same wave overwrites next (pp_epilogue).
constexpr int PP_PPW = 85;
__device__ __forceinline__ int pp_pixel(int gt, int i) {
    const int wv = gt >> 6, lane = gt & 63;
    return wv * PP_PPW + min((lane >> 3) + i * 8, PP_PPW - 1);
}

__device__ __forceinline__ void pp_load_x(PPX& xr, const ConvParams& P, const PPTile& T, int gt) {
    const int c8 = gt & 7;
    const bf16_t* base = P.src[0].ptr + (size_t)T.b * P.H * P.W * P.src[0].src_channels + P.src[0].ch_offset + c8 * 8;
    unsigned ok_all = 0;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int p = pp_pixel(gt, i);
        const int ty = p / 34, tx = p - ty * 34;
        const int iy = T.oy0 - 1 + ty, ix = T.ox0 - 1 + tx;
        const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        ok_all |= (ok ? 1u : 0u) << i;
        const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);
        xr.v[i] = *(const u32x4*)(base + ((size_t)cy * P.W + cx) * P.src[0].src_channels);
    }
    xr.okmask = ok_all;
}

__device__ __forceinline__ void pp_write_x(const PPX& xr, const ConvParams& P, const PPTile& T, unsigned char* xbuf, int gt) {
    const int c8 = gt & 7;
    float ps[8], pb[8];
    if (P.in_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ps[j] = P.in_scale[(size_t)T.b * 64 + c8 * 8 + j];
            pb[j] = P.in_shift[(size_t)T.b * 64 + c8 * 8 + j];
        }
    }
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int p = pp_pixel(gt, i);
        u32x4 v = xr.v[i];
        if (P.in_scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = silu_f(bf2f((bf16_t)(v[j] & 0xffffu)) * ps[2 * j] + pb[2 * j]);
                const float hi = silu_f(bf2f((bf16_t)(v[j] >> 16)) * ps[2 * j + 1] + pb[2 * j + 1]);
                v[j] = pack2(lo, hi);
            }
        }
        const bool ok = (xr.okmask >> i) & 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
        *(u32x4*)(xbuf + c8 * PP_US + p * 16) = v;
    }
}

__device__ __forceinline__ void pp_mfma(f32x16 (&acc)[2][2], const unsigned char* lds_w, const unsigned char* xbuf, int wv, int l31, int half,
                                        bool prio) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.0f;
    // 36 k-steps (9 taps x 4); the operand fragments of step s+1 are read from LDS before the
    // MFMAs of step s are issued (one wave per SIMD is in its MFMA phase: nobody else hides the
    // ds_read latency)
    const unsigned char* xrow = xbuf + half * PP_US + (wv * 2 * 34 + l31) * 16;   // every operand read = base + immediate
    const unsigned char* wrow = lds_w + (half * 64 + l31) * 16;
    bf16x8 xf[3][2], wf[3][2];
    auto read_frags = [&](int step, bf16x8 (&x2)[2], bf16x8 (&w2)[2]) {
        const int tap = step >> 2, ks = step & 3, ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) x2[pt] = *(const bf16x8*)(xrow + ((pt + ky) * 34 + kx) * 16 + ks * 2 * PP_US);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w2[nt] = *(const bf16x8*)(wrow + ((tap * 8 + ks * 2) * 64 + nt * 32) * 16);
    };
    read_frags(0, xf[0], wf[0]);
    read_frags(1, xf[1], wf[1]);
    if (prio) __builtin_amdgcn_s_setprio(2);      // ahead of the other group's VALU phase on the same SIMD
#pragma unroll
    for (int step = 0; step < 36; ++step) {
        if (step + 2 < 36) read_frags(step + 2, xf[(step + 2) % 3], wf[(step + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads two steps ahead of the MFMAs that consume them
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
                acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[step % 3][nt], xf[step % 3][pt], acc[nt][pt], 0, 0, 0);
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
}

// Epilogue with 16-byte stores: a lane holds 4 consecutive channels (8 B) per register quad; one
// v_permlane32_swap per dword between quads g and g+1 gives the lower half-wave 8 consecutive
// channels of quad g and the upper half-wave those of quad g+1 -> half as many store instructions
// (the 8-byte form is store-issue bound).
__device__ __forceinline__ void pp_epilogue(const f32x16 (&acc)[2][2], const ConvParams& P, const PPTile& T, const float* s_bias, int wv, int lane,
                                            unsigned char* xbuf) {
    const int l31 = lane & 31, half = lane >> 5;
    float stat[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) stat[i] = 0.0f;
    // The stores go through LDS so that they leave fully coalesced (a lane-per-pixel store pattern writes at 2.9 TB/s, consecutive
    // 16-byte units of a pixel row at 5.3+, tools/probe/store_pattern_probe.hip).  The group's input buffer is dead during its
    // VALU phase, but there is no group-level barrier: every wave therefore borrows only the bytes IT stages next (its 85-pixel
    // slice of each of the 8 octet rows, 1360 B each): region row r (144 B: 64 channels + pad) lives in octet row r / 9.
    unsigned char* const reg0 = xbuf + wv * PP_PPW * 16;
    auto region = [&](int r) { return reg0 + (r / 9) * PP_US + (r % 9) * 144; };
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int oy = T.oy0 + wv * 2 + pt, ox = T.ox0 + l31;
        const bool ok = oy < P.H && ox < P.W;
        unsigned char* const rrow = region(pt * 32 + l31);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            uint2 q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(s_bias + nt * 32 + 8 * g + 4 * half);
                q[g] = make_uint2(pack2(acc[nt][pt][4 * g] + bv.x, acc[nt][pt][4 * g + 1] + bv.y),
                                  pack2(acc[nt][pt][4 * g + 2] + bv.z, acc[nt][pt][4 * g + 3] + bv.w));
                if (P.gn_partial && ok) {
                    const float q0 = bf2f((bf16_t)(q[g].x & 0xffffu)), q1 = bf2f((bf16_t)(q[g].x >> 16));
                    const float q2 = bf2f((bf16_t)(q[g].y & 0xffffu)), q3 = bf2f((bf16_t)(q[g].y >> 16));
                    stat[(nt * 4 + g) * 2] += (q0 + q1) + (q2 + q3);
                    stat[(nt * 4 + g) * 2 + 1] += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
                *(uint4*)(rrow + (nt * 32 + 8 * g + 8 * half) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's own LDS writes
#pragma unroll
    for (int k = 0; k < 8; ++k) {                              // 64 pixels x 8 units: 8 lanes per pixel row, 8 pixels per instruction
        const int id = lane + k * 64, pl = id >> 3, cu = id & 7;
        const int oy = T.oy0 + wv * 2 + (pl >> 5), ox = T.ox0 + (pl & 31);
        const uint4 v = *(const uint4*)(region(pl) + cu * 16);
        if (oy < P.H && ox < P.W) *(uint4*)(P.out + (((size_t)T.b * P.H + oy) * P.W + ox) * 64 + cu * 8) = v;
    }
    if (P.gn_partial) {
        wave_reduce_multi<16>(stat);
        if ((lane & 3) == 0) {
            const int vi = lane >> 2;
            P.gn_partial[gn_partial_index(T.b, P.tiles_x * P.tiles_y * 4, T.t_in * 4 + wv, 8, vi >> 1) + (vi & 1)] = stat[0];
        }
    }
}

// Phase barrier: only LDS traffic has to be complete (lgkmcnt).  __syncthreads() would also drain
// vmcnt, i.e. wait for the epilogue's global stores and the in-flight prefetch of the next tile.
__device__ __forceinline__ void pp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ void __launch_bounds__(PP_THREADS, 1) conv3x3_c64_pingpong_kernel(const ConvParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_w = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int group = wave >> 2, wv = wave & 3, gt = tid & 255;
    unsigned char* xbuf = smem + PP_WB + group * PP_XB;
    const int tpi = P.tiles_x * P.tiles_y, ntiles = tpi * P.B;
    const int npairs = (ntiles + 1) / 2;
    const int n_iter = (npairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // pairs of this workgroup

    // weights: resident for the whole launch
#pragma unroll
    for (int i = 0; i < PP_WB / 16 / PP_THREADS; ++i)
        *(u32x4*)(lds_w + (tid + i * PP_THREADS) * 16) = *(const u32x4*)(P.weight + (size_t)(tid + i * PP_THREADS) * 8);

    float* s_bias = (float*)(smem + PP_WB + 2 * PP_XB);
    if (tid < 64) s_bias[tid] = P.bias ? P.bias[tid] : 0.0f;

    auto my_tile = [&](int i) { return pp_tile(((int)blockIdx.x + i * (int)gridDim.x) * 2 + group, ntiles, P.tiles_x, tpi); };

    PPX xr;
    PPTile cur = my_tile(0), prev = cur;
    pp_load_x(xr, P, cur, gt);
    pp_write_x(xr, P, cur, xbuf, gt);
    __syncthreads();

    f32x16 acc[2][2];
    for (int i = 0; i < n_iter; ++i) {
        const PPTile nxt = my_tile(i + 1);        // (clamped to a valid tile past the end; never stored)
        if (group == 0) {
            if (!(P.dbg & 1)) pp_load_x(xr, P, nxt, gt);
            if (!(P.dbg & 4)) pp_mfma(acc, lds_w, xbuf, wv, l31, half, !(P.dbg & 256));
        } else if (i > 0) {
            if (prev.valid && !(P.dbg & 16)) pp_epilogue(acc, P, prev, s_bias, wv, lane, xbuf);
            if (!(P.dbg & 32)) pp_write_x(xr, P, cur, xbuf, gt);
        }
        pp_barrier();
        if (group == 0) {
            if (cur.valid && !(P.dbg & 16)) pp_epilogue(acc, P, cur, s_bias, wv, lane, xbuf);
            if (!(P.dbg & 32)) pp_write_x(xr, P, nxt, xbuf, gt);
        } else {
            if (!(P.dbg & 1)) pp_load_x(xr, P, nxt, gt);
            if (!(P.dbg & 4)) pp_mfma(acc, lds_w, xbuf, wv, l31, half, !(P.dbg & 256));
        }
        pp_barrier();
        prev = cur;
        cur = nxt;
    }
    if (group == 1 && n_iter > 0 && prev.valid) pp_epilogue(acc, P, prev, s_bias, wv, lane, xbuf);
}

// ================================================================================================
// Warp-specialised persistent 3x3 kernel (all 3x3 layers except the 64->64 ping-pong case).
// Workgroup = 8 waves, one per CU: waves 0-3 are CONSUMERS (ds_read + MFMA + epilogue only), waves
// 4-7 are PRODUCERS (global loads, GroupNorm-affine+SiLU prologue, LDS writes).  A stage is one
// kernel row (3 taps) of one K-chunk: 48 MFMAs per consumer wave between two workgroup barriers,
// while the producers fill the other weight buffer (slab q+1, loaded two stages earlier into
// registers) and, per chunk, the other input-tile buffer.  Raw s_barrier + lgkmcnt only, so the
// producers' global loads stay in flight across barriers.  The workgroup walks many tiles
// (persistent, grid = 256 / n-blocks), so the producers run ahead into the next tile while the
// consumers are in their epilogue.
//   BN = 64 : K-chunk 64 channels, input tile 43.6 KB x2, weight slab 24 KB x2  (135 KB LDS)
//   BN = 128: K-chunk 32 channels, input tile 21.8 KB x2, weight slab 24 KB x2  ( 92 KB LDS)
template <int BN>
struct WsCfg {
    static constexpr int CK = (BN == 64) ? 64 : 32;
    static constexpr int NC = CK / 8;
    static constexpr int IW = 34, NPIX = 340, US = 341 * 16;
    static constexpr int XB = NC * US;
    static constexpr int WROWS = 3 * NC;
    static constexpr int WB = WROWS * BN * 16;
    static constexpr int LDS = 2 * XB + 2 * WB;
    static constexpr int XPT = (NPIX * NC + 255) / 256;
    static constexpr int WPT = WROWS * BN / 256;
    static constexpr int NTN = BN / 32;
    static constexpr int KSTEPS = CK / 16;
    static_assert(WROWS * BN % 256 == 0, "weight slab must divide over the producer threads");
};

template <int BN>
struct WsX {
    u32x4 v[WsCfg<BN>::XPT];
    unsigned okmask;
};

// producer: global -> registers of K-chunk kc (CK-channel units over the concatenated sources)
template <int BN>
__device__ __forceinline__ void ws_load_x(WsX<BN>& xr, const ConvParams& P, const PPTile& T, int kc, int gt) {
    using C = WsCfg<BN>;
    constexpr int R = 64 / C::CK;          // chunks of this kernel per 64-channel chunk of the descriptors
    int si = 0, first = 0;
    while (si + 1 < P.n_src && kc >= first + P.src[si].chunks * R) {
        first += P.src[si].chunks * R;
        ++si;
    }
    const int kcl = kc - first;
    const int s_ch = P.src[si].src_channels, s_SH = P.src[si].SH, s_SW = P.src[si].SW, s_mode = P.src[si].mode;
    const int c8 = gt % C::NC;
    const bf16_t* base = P.src[si].ptr + (size_t)T.b * s_SH * s_SW * s_ch + P.src[si].ch_offset + kcl * C::CK + c8 * 8;
    unsigned ok_all = 0;
#pragma unroll
    for (int i = 0; i < C::XPT; ++i) {
        const int p = min(gt / C::NC + i * (256 / C::NC), C::NPIX - 1);
        const int ty = p / C::IW, tx = p - ty * C::IW;
        const int iy = T.oy0 - 1 + ty, ix = T.ox0 - 1 + tx;
        const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        ok_all |= (ok ? 1u : 0u) << i;
        const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);
        int sy = cy, sx = cx;
        if (s_mode == 1) { sy = cy >> 1; sx = cx >> 1; }
        else if (s_mode == 2) { sy = 2 * cy + P.src[si].p1; sx = 2 * cx + P.src[si].p2; }
        xr.v[i] = *(const u32x4*)(base + ((size_t)sy * s_SW + sx) * s_ch);
    }
    xr.okmask = ok_all;
}

template <int BN>
__device__ __forceinline__ void ws_write_x(const WsX<BN>& xr, const ConvParams& P, const PPTile& T, int kc, unsigned char* xbuf, int gt) {
    using C = WsCfg<BN>;
    const int c8 = gt % C::NC;
    float ps[8], pb[8];
    if (P.in_scale) {
        const int cg = kc * C::CK + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ps[j] = P.in_scale[(size_t)T.b * P.Cin_total + cg + j];
            pb[j] = P.in_shift[(size_t)T.b * P.Cin_total + cg + j];
        }
    }
#pragma unroll
    for (int i = 0; i < C::XPT; ++i) {
        const int p = min(gt / C::NC + i * (256 / C::NC), C::NPIX - 1);
        u32x4 v = xr.v[i];
        if (P.in_scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = silu_f(bf2f((bf16_t)(v[j] & 0xffffu)) * ps[2 * j] + pb[2 * j]);
                const float hi = silu_f(bf2f((bf16_t)(v[j] >> 16)) * ps[2 * j + 1] + pb[2 * j + 1]);
                v[j] = pack2(lo, hi);
            }
        }
        const bool ok = (xr.okmask >> i) & 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
        *(u32x4*)(xbuf + c8 * C::US + p * 16) = v;
    }
}

// weight slab of (chunk kc, kernel row ky): rows r = kx * NC + c8
template <int BN>
__device__ __forceinline__ void ws_load_w(U4Arr<WsCfg<BN>::WPT>& w, const ConvParams& P, int n0, int kc, int ky, int gt) {
    using C = WsCfg<BN>;
    const int cin8 = P.Cin_total / 8;
#pragma unroll
    for (int i = 0; i < C::WPT; ++i) {
        const int u = gt + i * 256;
        const int r = u / BN, n = u % BN;
        const size_t row = (size_t)(ky * 3 + r / C::NC) * cin8 + kc * C::NC + r % C::NC;
        w.v[i] = *(const u32x4*)(P.weight + (row * P.Cout + n0 + n) * 8);
    }
}
template <int BN>
__device__ __forceinline__ void ws_store_w(const U4Arr<WsCfg<BN>::WPT>& w, unsigned char* wbuf, int gt) {
#pragma unroll
    for (int i = 0; i < WsCfg<BN>::WPT; ++i) *(u32x4*)(wbuf + (gt + i * 256) * 16) = w.v[i];
}

// consumer: one stage = 3 taps x KSTEPS k-steps, operand reads one k-step ahead of the MFMAs
template <int BN>
__device__ __forceinline__ void ws_compute(f32x16 (&acc)[WsCfg<BN>::NTN][2], const unsigned char* wbuf, const unsigned char* xbuf, int ky,
                                           int wv, int l31, int half) {
    using C = WsCfg<BN>;
    constexpr int NS = 3 * C::KSTEPS;
    const unsigned char* xrow = xbuf + half * C::US + ((wv * 2 + ky) * C::IW + l31) * 16;
    const unsigned char* wrow = wbuf + (half * BN + l31) * 16;
    bf16x8 xf[2][2], wf[2][C::NTN];
    auto read_frags = [&](int step, bf16x8 (&x2)[2], bf16x8 (&w2)[C::NTN]) {
        const int kx = step / C::KSTEPS, ks = step % C::KSTEPS;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) x2[pt] = *(const bf16x8*)(xrow + (pt * C::IW + kx) * 16 + ks * 2 * C::US);
#pragma unroll
        for (int nt = 0; nt < C::NTN; ++nt) w2[nt] = *(const bf16x8*)(wrow + ((kx * C::NC + ks * 2) * BN + nt * 32) * 16);
    };
    read_frags(0, xf[0], wf[0]);
#pragma unroll
    for (int step = 0; step < NS; ++step) {
        if (step + 1 < NS) read_frags(step + 1, xf[(step + 1) & 1], wf[(step + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < C::NTN; ++nt)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
                acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[step & 1][nt], xf[step & 1][pt], acc[nt][pt], 0, 0, 0);
    }
}

// shared epilogue of a 4-wave x (2 rows x 32 px) x BN tile: bias, residual forms, bf16 16-byte
// stores through v_permlane32_swap, per-wave GroupNorm partial sums
template <int BN>
__device__ __forceinline__ void conv_tile_epilogue(const f32x16 (&acc)[BN / 32][2], const ConvParams& P, const PPTile& T, int n0, int wv, int lane) {
    constexpr int NTN = BN / 32, NV = (BN / 8) * 2;
    const int l31 = lane & 31, half = lane >> 5;
    float stat[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) stat[i] = 0.0f;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int oy = T.oy0 + wv * 2 + pt, ox = T.ox0 + l31;
        const bool ok = oy < P.H && ox < P.W;
        const size_t pix = ((size_t)T.b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1);
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            uint2 q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n0 + nt * 32 + 8 * g + 4 * half;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[nt][pt][4 * g + j];
                if (P.bias) {
                    const float4 bv = *(const float4*)(P.bias + c);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (P.res_act) {
                    const uint2 r = *(const uint2*)(P.res_act + pix * P.Cout + c);
                    const float4 sc = *(const float4*)(P.res_scale + (size_t)T.b * P.Cout + c);
                    const float4 sh = *(const float4*)(P.res_shift + (size_t)T.b * P.Cout + c);
                    v[0] += silu_f(bf2f((bf16_t)(r.x & 0xffffu)) * sc.x + sh.x);
                    v[1] += silu_f(bf2f((bf16_t)(r.x >> 16)) * sc.y + sh.y);
                    v[2] += silu_f(bf2f((bf16_t)(r.y & 0xffffu)) * sc.z + sh.z);
                    v[3] += silu_f(bf2f((bf16_t)(r.y >> 16)) * sc.w + sh.w);
                }
                if (P.residual) {
                    const uint2 r = *(const uint2*)(P.residual + pix * P.Cout + c);
                    v[0] += bf2f((bf16_t)(r.x & 0xffffu));
                    v[1] += bf2f((bf16_t)(r.x >> 16));
                    v[2] += bf2f((bf16_t)(r.y & 0xffffu));
                    v[3] += bf2f((bf16_t)(r.y >> 16));
                }
                q[g] = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                if (P.gn_partial && ok) {
                    const float q0 = bf2f((bf16_t)(q[g].x & 0xffffu)), q1 = bf2f((bf16_t)(q[g].x >> 16));
                    const float q2 = bf2f((bf16_t)(q[g].y & 0xffffu)), q3 = bf2f((bf16_t)(q[g].y >> 16));
                    stat[(nt * 4 + g) * 2] += (q0 + q1) + (q2 + q3);
                    stat[(nt * 4 + g) * 2 + 1] += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
                if (ok) *(uint4*)(P.out + pix * P.Cout + n0 + nt * 32 + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
    }
    if (P.gn_partial) {
        wave_reduce_multi<NV>(stat);
        constexpr int SH_ = (NV == 32) ? 1 : 2;
        if ((lane & ((1 << SH_) - 1)) == 0) {
            const int vi = lane >> SH_;
            P.gn_partial[gn_partial_index(T.b, P.tiles_x * P.tiles_y * 4, T.t_in * 4 + wv, P.Cout / 8, n0 / 8 + (vi >> 1)) + (vi & 1)] = stat[0];
        }
    }
}

template <int BN>
__global__ void __launch_bounds__(512, 1) conv3x3_ws_kernel(const ConvParams P) {
    using C = WsCfg<BN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xb = smem;                  // [2][XB]
    unsigned char* wb = smem + 2 * C::XB;      // [2][WB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int role = wave >> 2, wv = wave & 3, gt = tid & 255;
    const int tpi = P.tiles_x * P.tiles_y, ntiles = tpi * P.B;
    const int n0 = blockIdx.y * BN;
    const int n_my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nchunks = P.total_chunks * (64 / C::CK);
    const int S = nchunks * 3, Q = n_my * S, CT = n_my * nchunks;     // steps per tile, total steps, total chunks
    auto tile_at = [&](int i) { return pp_tile((int)blockIdx.x + i * (int)gridDim.x, ntiles, P.tiles_x, tpi); };

    if (role == 1) {
        // ------------------------------------------------------------------ producers
        WsX<BN> xr;
        U4Arr<C::WPT> w1, w2;
        {
            const PPTile T0 = tile_at(0);
            ws_load_x<BN>(xr, P, T0, 0, gt);
            ws_load_w<BN>(w1, P, n0, 0, 0, gt);
            ws_write_x<BN>(xr, P, T0, 0, xb, gt);
            ws_store_w<BN>(w1, wb, gt);
            const int g1 = (Q > 1) ? 1 : 0, g2 = (Q > 2) ? 2 : 0;
            ws_load_w<BN>(w1, P, n0, (g1 % S) / 3, g1 % 3, gt);
            ws_load_w<BN>(w2, P, n0, (g2 % S) / 3, g2 % 3, gt);
        }
        pp_barrier();
        int q = 0, c = 0;
        for (int i = 0; i < n_my; ++i) {
            for (int kc = 0; kc < nchunks; ++kc, ++c) {
                const bool has_next = (c + 1 < CT);
                const int cn = c + 1, in_ = cn / nchunks, kcn = cn - in_ * nchunks;   // next chunk: tile index, local chunk
                const PPTile Tn = tile_at(has_next ? in_ : i);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky, ++q) {
                    if (q + 1 < Q) ws_store_w<BN>(w1, wb + ((q + 1) & 1) * C::WB, gt);   // slab q+1, loaded two stages ago
                    w1 = w2;
                    if (q + 3 < Q) {
                        const int g = (q + 3) % S;
                        ws_load_w<BN>(w2, P, n0, g / 3, g % 3, gt);
                    }
                    if (ky == 0 && has_next) ws_load_x<BN>(xr, P, Tn, kcn, gt);
                    if (ky == 2 && has_next) ws_write_x<BN>(xr, P, Tn, kcn, xb + (cn & 1) * C::XB, gt);
                    pp_barrier();
                }
            }
        }
    } else {
        // ------------------------------------------------------------------ consumers
        pp_barrier();
        f32x16 acc[C::NTN][2];
        int q = 0, c = 0;
        for (int i = 0; i < n_my; ++i) {
#pragma unroll
            for (int a = 0; a < C::NTN; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[a][b2][k] = 0.0f;
            for (int kc = 0; kc < nchunks; ++kc, ++c) {
                const unsigned char* xbuf = xb + (c & 1) * C::XB;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky, ++q) {
                    ws_compute<BN>(acc, wb + (q & 1) * C::WB, xbuf, ky, wv, l31, half);
                    pp_barrier();
                }
            }
            const PPTile T = tile_at(i);
            if (T.valid) conv_tile_epilogue<BN>(acc, P, T, n0, wv, lane);
        }
    }
}

template <int BN>
static int launch_conv_ws(const ConvParams& P, hipStream_t s) {
    using C = WsCfg<BN>;
    static bool attr_set = false;
    if (!attr_set) {
        OFD_HIP(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr_set = true;
    }
    const int ntiles = P.tiles_x * P.tiles_y * P.B, nb = P.Cout / BN;
    int gx = 256 / nb;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    conv3x3_ws_kernel<BN><<<dim3(gx, nb), 512, C::LDS, s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

// ---- weight preparation: OIHW fp32 -> [tap][Cin_pad/8][Cout][8] bf16 (+ weight standardisation)
__device__ __forceinline__ void conv_weight_prep_body(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int Cin_pad,
                                                      int ksize, float ws_eps, int unshuffle, int o) {
    const int tid = threadIdx.x;
    const int taps = ksize * ksize, n = Cin * taps;
    const float* wo = w + (size_t)o * n;
    __shared__ double sh[256];
    float mean = 0.0f, rstd = 1.0f;
    if (ws_eps >= 0.0f) {       // DD:109-112: biased variance over (Cin, kh, kw), fp32 result
        double s = 0.0;
        for (int i = tid; i < n; i += 256) s += (double)wo[i];
        sh[tid] = s;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) { if (tid < k) sh[tid] += sh[tid + k]; __syncthreads(); }
        const double m = sh[0] / n;
        __syncthreads();
        double v = 0.0;
        for (int i = tid; i < n; i += 256) { const double d = (double)wo[i] - m; v += d * d; }
        sh[tid] = v;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) { if (tid < k) sh[tid] += sh[tid + k]; __syncthreads(); }
        mean = (float)m;
        rstd = rsqrtf((float)(sh[0] / n) + ws_eps);
    }
    const int c8n = Cin_pad / 8;
    for (int i = tid; i < Cin_pad * taps; i += 256) {
        const int cp = i / taps, tap = i % taps;     // cp: engine channel index
        float v = 0.0f;
        if (cp < Cin) {
            int ci = cp;
            if (unshuffle) {                          // engine order (p1 p2) c  <-  reference c (p1 p2), DD:97
                const int Cq = Cin / 4, sub = cp / Cq, c = cp % Cq;
                ci = c * 4 + sub;
            }
            v = (wo[(size_t)ci * taps + tap] - mean) * rstd;
        }
        // 7x7 over 8 channels: rows [ky][8 tap columns] (Cfg::K7P), the 8th column zero
        const size_t row = (ksize == 7 && Cin_pad == 8) ? (size_t)(tap / 7) * 8 + tap % 7 : (size_t)tap * c8n + cp / 8;
        out[(row * Cout + o) * 8 + (cp % 8)] = f2bf(v);
    }
    if (ksize == 7 && Cin_pad == 8)
        for (int i = tid; i < 7 * 8; i += 256) out[(((size_t)(i / 8) * 8 + 7) * Cout + o) * 8 + (i % 8)] = 0;
}

__global__ void __launch_bounds__(256) conv_weight_prep_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout,
                                                               int Cin, int Cin_pad, int ksize, float ws_eps, int unshuffle) {
    conv_weight_prep_body(w, out, Cout, Cin, Cin_pad, ksize, ws_eps, unshuffle, blockIdx.x);
}

// every conv of a UNet in ONE launch (the training step re-prepares all weights after each optimizer step: 95 launches of ~8 us
// otherwise): block -> descriptor by binary search over the first-block table
__global__ void __launch_bounds__(256) conv_weight_prep_batched_kernel(const ofd_weight_prep_desc* __restrict__ descs, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ofd_weight_prep_desc d = descs[lo];
    conv_weight_prep_body(d.w, (bf16_t*)d.out, d.Cout, d.Cin, d.Cin_pad, d.ksize, d.ws_eps, d.unshuffle, (int)blockIdx.x - d.block0);
}

// nearest-x2 up-sample followed by a 3x3 conv (DD:89-93) = four 2x2 convs on the LOW-RES input, one per output phase
// (py, px): output row 2y+py reads up-sampled rows 2y+py-1 .. 2y+py+1 = low-res rows {y-1, y, y} (py = 0) or {y, y, y+1}
// (py = 1), so the three kernel rows collapse onto two source rows (same in x).  2.25x fewer MACs than convolving the
// up-sampled tensor.  out: 4 x [2x2 taps][Cin/8][Cout][8] bf16 (phase = py*2+px, tap = dy*2+dx); the collapsed weights are
// summed in fp32 and rounded once.
__global__ void __launch_bounds__(256) upsample_phase_weight_prep_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin) {
    const int o = blockIdx.x, c8n = Cin / 8;
    const float* wo = w + (size_t)o * Cin * 9;
    for (int i = threadIdx.x; i < Cin * 16; i += 256) {
        const int ci = i / 16, ph = (i / 4) % 4, tap = i % 4;
        const int py = ph >> 1, px = ph & 1, dy = tap >> 1, dx = tap & 1;
        // kernel rows / columns that land on low-res offset d of phase p
        const int ky0 = (py == 0) ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), ky1 = (py == 0) ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
        const int kx0 = (px == 0) ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), kx1 = (px == 0) ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
        float v = 0.0f;
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) v += wo[(size_t)ci * 9 + ky * 3 + kx];
        out[(size_t)ph * 4 * Cin * Cout + (((size_t)tap * c8n + ci / 8) * Cout + o) * 8 + (ci % 8)] = f2bf(v);
    }
}

template <int KS, int BN>
static int launch_conv(const ConvParams& P, hipStream_t s) {
    using C = Cfg<KS, BN>;
    static bool attr_set = false;
    if (!attr_set) {
        OFD_HIP(hipFuncSetAttribute((const void*)conv_igemm_kernel<KS, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    const int ntiles = P.tiles_x * P.tiles_y * P.B;
    dim3 grid((KS == 2 && P.phase_all) ? (ntiles + 7) / 8 * 32 : ntiles, P.Cout / BN);
    conv_igemm_kernel<KS, BN><<<grid, NTHREADS, C::LDS_BYTES, s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int launch_conv3x3_c64_rw(const ConvParams& P, hipStream_t s);     // conv_rw.hip
int launch_conv3x3_wp(const ConvParams& P, bool wide, hipStream_t s);   // conv_wp.hip
int launch_conv1x1_wp(const ConvParams& P, hipStream_t s);              // conv1_wp.hip: 1 = shape not served
int launch_conv_up2_phases_wp(const ConvParams& P, hipStream_t s);      // conv_wp.hip: 1 = shape not served
int launch_conv7x7_c8_persist(const ConvParams& P, hipStream_t s);      // conv7.hip: 1 = shape not served

static int conv_wp_bits() {
    static int use_wp = -1;
    if (use_wp < 0) { const char* e = getenv("OFD_CONV_WP"); use_wp = e ? atoi(e) : 7; }
    return use_wp;
}
// the pooled epilogue exists in conv_wp.hip only: 3x3, even size, one output tensor, no fused GroupNorm statistics / activation residual
bool conv_pool2_supported(const ofd_conv_args* a) {
    static const bool off = getenv("OFD_NO_DGRAD_POOL") && atoi(getenv("OFD_NO_DGRAD_POOL"));
    if (off || !a || a->ksize != 3 || a->H % 2 || a->W % 2 || a->split || a->gn_partial || a->res_act || a->Cout % 64) return false;
    int cin = 0;
    for (int i = 0; i < a->n_src; ++i) { if (a->src[i].unshuffle) return false; cin += a->src[i].channels; }
    const bool c64 = a->Cout == 64 && cin == 64;
    return !c64 && (conv_wp_bits() & 3) == 3;
}

// final 1x1 conv fused into the final res_conv (conv1_wp.hip, FC instantiation): 128 -> 64, fused h2 input, whole 128-pixel tiles, out_dim 2
bool conv_fc_fuse_supported(const ofd_conv_args* a, int out_dim) {
    static const bool off = (getenv("OFD_NO_FC_FUSE") && atoi(getenv("OFD_NO_FC_FUSE"))) || (getenv("OFD_CONV1_WP") && atoi(getenv("OFD_CONV1_WP")) == 0);
    if (off || !a || out_dim != 2 || a->ksize != 1 || a->Cout != 64 || !a->res_act || a->residual || a->gn_partial || a->split || a->in_scale) return false;
    int cin = 0;
    for (int i = 0; i < a->n_src; ++i) { if (a->src[i].upsample || a->src[i].unshuffle || a->src[i].channels % 64) return false; cin += a->src[i].channels; }
    return cin == 128 && ((long)a->H * a->W) % 128 == 0;
}

bool conv_residual_b_supported(const ofd_conv_args* a) {
    static const bool off = getenv("OFD_NO_RESIDUAL_B") && atoi(getenv("OFD_NO_RESIDUAL_B"));
    if (off || !a || a->ksize != 3 || a->split || a->Cout % 64) return false;
    for (int i = 0; i < a->n_src; ++i)
        if (a->src[i].unshuffle) return false;
    return (conv_wp_bits() & 7) == 7 || ((conv_wp_bits() & 3) == 3 && a->residual);      // (with a residual the 64 -> 64 case is conv_wp's too)
}

int conv_forward_impl(const ofd_conv_args* a, hipStream_t s, int cout0, int pool2, const bf16_t* residual_b, const FcFuse* fc) {
    OFD_CHECK_ARG(a && a->out && a->weight, "conv: null out/weight");
    OFD_CHECK_ARG(a->B > 0 && a->H > 0 && a->W > 0, "conv: bad shape");
    OFD_CHECK_ARG(a->ksize == 1 || a->ksize == 2 || a->ksize == 3 || a->ksize == 7, "conv: ksize %d unsupported", a->ksize);
    OFD_CHECK_ARG((a->ksize == 2) == (a->up2_phase >= 1 && a->up2_phase <= 5), "conv: ksize 2 is one phase (up2_phase 1..4) of an up-sampled 3x3, or all four (5)");
    OFD_CHECK_ARG(a->ksize != 2 || (!a->residual && !a->res_act && !a->gn_partial), "conv: phase convs take no residual / GroupNorm statistics");
    OFD_CHECK_ARG(a->Cout > 0 && a->Cout % 64 == 0, "conv: Cout=%d must be a multiple of 64", a->Cout);
    OFD_CHECK_ARG(a->n_src >= 1 && a->n_src <= 4, "conv: n_src=%d", a->n_src);
    OFD_CHECK_ARG(!(a->in_scale) == !(a->in_shift), "conv: in_scale/in_shift must come together");
    OFD_CHECK_ARG(!a->res_act || (a->res_scale && a->res_shift), "conv: res_act needs res_scale/res_shift");
    OFD_CHECK_ARG(a->split == 0 || (a->split > 0 && a->split < a->Cout && a->split % 64 == 0 && a->out2 && a->ksize != 2 && !a->gn_partial && !a->res_act),
                  "conv: split=%d needs out2, a multiple of 64 below Cout, no GroupNorm statistics", a->split);
    const bool k7p = a->ksize == 7 && a->n_src == 1 && a->src[0].channels == 8;      // 8-channel input: tap-pair packing (Cfg::K7P)
    const int ck = a->ksize == 7 ? (k7p ? 8 : 16) : 64;
    ConvParams P{};
    P.B = a->B; P.H = a->H; P.W = a->W; P.Cout = a->Cout; P.n_src = a->n_src;
    P.tiles_x = cdiv(a->W, TW); P.tiles_y = cdiv(a->H, TH);
    OFD_CHECK_ARG((long)P.tiles_x * P.tiles_y * P.B < (1l << 31), "conv: too many tiles");
    int cin = 0;
    for (int i = 0; i < a->n_src; ++i) {
        const ofd_conv_src& s_ = a->src[i];
        OFD_CHECK_ARG(s_.src && s_.channels > 0 && s_.channels % ck == 0, "conv: source %d channels=%d must be a multiple of %d", i, s_.channels, ck);
        OFD_CHECK_ARG(s_.ch_offset % 8 == 0 && s_.src_channels % 8 == 0 && s_.ch_offset + s_.channels <= s_.src_channels,
                      "conv: source %d channel window [%d,+%d) of %d", i, s_.ch_offset, s_.channels, s_.src_channels);
        OFD_CHECK_ARG(!(s_.upsample && s_.unshuffle), "conv: source %d both upsample and unshuffle", i);
        OFD_CHECK_ARG(!s_.upsample || (a->H % 2 == 0 && a->W % 2 == 0), "conv: upsample needs even output size");
        ConvSrcDev& d = P.src[i];
        d.ptr = (const bf16_t*)s_.src; d.chunks = s_.channels / ck; d.src_channels = s_.src_channels; d.ch_offset = s_.ch_offset;
        d.mode = s_.upsample ? 1 : (s_.unshuffle ? 2 : 0);
        d.SH = s_.upsample ? a->H / 2 : (s_.unshuffle ? a->H * 2 : a->H);
        d.SW = s_.upsample ? a->W / 2 : (s_.unshuffle ? a->W * 2 : a->W);
        d.p1 = s_.p1; d.p2 = s_.p2;
        cin += s_.channels;
        P.total_chunks += d.chunks;
    }
    P.Cin_total = cin;
    OFD_CHECK_ARG(a->ksize != 7 || (P.total_chunks == 1 && a->Cout == 64 && !a->in_scale && !a->src[0].upsample && !a->src[0].unshuffle),
                  "conv: 7x7 supports one 8- or 16-channel source, Cout=64");
    P.weight = (const bf16_t*)a->weight; P.bias = a->bias; P.in_scale = a->in_scale; P.in_shift = a->in_shift;
    P.residual = (const bf16_t*)a->residual; P.res_act = (const bf16_t*)a->res_act; P.res_scale = a->res_scale; P.res_shift = a->res_shift;
    P.out = (bf16_t*)a->out; P.gn_partial = a->gn_partial;
    P.out2 = (bf16_t*)a->out2; P.residual2 = (const bf16_t*)a->residual2; P.split = a->split;
    if (a->ksize == 2) {
        const int py = (a->up2_phase - 1) >> 1, px = (a->up2_phase - 1) & 1;
        P.pad_y = 1 - py; P.pad_x = 1 - px; P.out_oy = py; P.out_ox = px;
        P.phase_all = a->up2_phase == 5;
    }
    P.cout0 = cout0;
    P.pool2 = pool2;
    P.residual_b = residual_b;
    if (fc) { P.fc_w = fc->w; P.fc_b = fc->b; P.fc_out = fc->out; }
    OFD_CHECK_ARG(!fc || conv_fc_fuse_supported(a, 2), "conv: the fused final conv serves the 128 -> 64 streaming 1x1 with a SiLU(GN(h2)) input only");
    OFD_CHECK_ARG(!residual_b || conv_residual_b_supported(a), "conv: a second residual is served by the conv_wp 3x3 kernels only");
    OFD_CHECK_ARG(!pool2 || conv_pool2_supported(a), "conv: the 2x2-pooled epilogue does not serve this configuration");
    { static int dbg_env = -1; if (dbg_env < 0) { const char* e = getenv("OFD_CONV_DBG"); dbg_env = e ? atoi(e) : 0; } P.dbg = dbg_env; }
    // 128 output channels per workgroup unless that leaves CUs without work: small images (the reference's default 128 x 128 reaches
    // 16 x 16 at the coarsest level: 32 pixel tiles x 4 channel blocks for 256 CUs) take the 64-channel instantiation: twice the
    // workgroups
    static const int small_grid = getenv("OFD_CONV_SMALL_GRID") ? atoi(getenv("OFD_CONV_SMALL_GRID")) : 256;
    const bool wide = (a->Cout % 128 == 0) && (long)P.tiles_x * P.tiles_y * P.B * (a->Cout / 128) >= small_grid;
    // wave-private-weights kernel (conv_wp.hip): OFD_CONV_WP bit 0 = the 128-channel-block layers, bit 1 = the 64-channel-block
    // layers other than 64 -> 64, bit 2 = 64 -> 64 (instead of the ping-pong kernel)
    static int use_wp = -1;
    if (use_wp < 0) { const char* e = getenv("OFD_CONV_WP"); use_wp = e ? atoi(e) : 7; }
    if (a->ksize == 3 && use_wp) {
        bool modes_ok = true;
        for (int i = 0; i < a->n_src; ++i) modes_ok = modes_ok && P.src[i].mode != 2;
        const bool c64 = a->Cout == 64 && P.Cin_total == 64 && a->n_src == 1 && P.src[0].mode == 0 && !a->residual && !a->res_act;
        const int bit = wide ? 1 : (c64 ? 4 : 2);
        if (modes_ok && (use_wp & bit)) return launch_conv3x3_wp(P, wide, s);
    }
    static int no_pp = -1;
    if (no_pp < 0) { const char* e = getenv("OFD_NO_PINGPONG"); no_pp = (e && atoi(e)) ? 1 : 0; }
    static int use_rw = -1;
    if (use_rw < 0) { const char* e = getenv("OFD_CONV_RW"); use_rw = e ? atoi(e) : 0; }      // register-window variant: opt-in (conv_rw.hip)
    if (a->ksize == 3 && a->Cout == 64 && P.Cin_total == 64 && a->n_src == 1 && P.src[0].mode == 0 && !a->residual && !a->res_act && use_rw && !a->split)
        return launch_conv3x3_c64_rw(P, s);
    if (a->ksize == 3 && a->Cout == 64 && P.Cin_total == 64 && a->n_src == 1 && P.src[0].mode == 0 && !a->residual && !a->res_act && !no_pp) {
        static bool attr_set = false;
        if (!attr_set) {
            OFD_HIP(hipFuncSetAttribute((const void*)conv3x3_c64_pingpong_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
            attr_set = true;
        }
        const int ntiles = P.tiles_x * P.tiles_y * P.B, npairs = (ntiles + 1) / 2;
        static int pp_grid = -1;
        if (pp_grid < 0) { const char* e = getenv("OFD_PP_GRID"); pp_grid = e ? atoi(e) : 256; }
        conv3x3_c64_pingpong_kernel<<<npairs < pp_grid ? npairs : pp_grid, PP_THREADS, PP_LDS, s>>>(P);
        OFD_LAUNCH_CHECK();
        return OFD_OK;
    }
    // warp-specialised variant: measured equal-to-slightly-slower than the generic kernel this round
    // (DESIGN.md section 4), so it is opt-in for A/B runs: OFD_CONV_WS=1
    static int use_ws = -1;
    if (use_ws < 0) { const char* e = getenv("OFD_CONV_WS"); use_ws = (e && atoi(e)) ? 1 : 0; }
    if (a->ksize == 3 && use_ws && !a->split) return wide ? launch_conv_ws<128>(P, s) : launch_conv_ws<64>(P, s);
    if (a->ksize == 3) return wide ? launch_conv<3, 128>(P, s) : launch_conv<3, 64>(P, s);
    // streaming 1x1 kernel (conv1_wp.hip) where it serves the shape: OFD_CONV1_WP=0 switches it off
    static int use_c1 = -1;
    if (use_c1 < 0) { const char* e = getenv("OFD_CONV1_WP"); use_c1 = e ? atoi(e) : 1; }
    if (a->ksize == 1 && use_c1 && !P.dbg) {
        const int r = launch_conv1x1_wp(P, s);
        if (r != 1) return r;
    }
    OFD_CHECK_ARG(!P.fc_out, "conv: the fused final conv was requested for a shape the streaming 1x1 kernel does not serve");
    if (a->ksize == 1) return wide ? launch_conv<1, 128>(P, s) : launch_conv<1, 64>(P, s);
    if (a->ksize == 2 && P.phase_all && !P.dbg) {        // the four phases as four wave pairs of one workgroup (conv_wp.hip): OFD_PHASE_WP=0 switches it off
        const int r = launch_conv_up2_phases_wp(P, s);
        if (r != 1) return r;
    }
    if (a->ksize == 2) return wide ? launch_conv<2, 128>(P, s) : launch_conv<2, 64>(P, s);
    if (k7p) {                                           // weights resident in LDS, a workgroup walks many tiles (conv7.hip): OFD_CONV7_PERSIST=0 switches it off
        const int r = launch_conv7x7_c8_persist(P, s);
        if (r != 1) return r;
    }
    return k7p ? launch_conv<8, 64>(P, s) : launch_conv<7, 64>(P, s);
}

}  // namespace ofd
using namespace ofd;

extern "C" int ofd_conv_forward(const ofd_conv_args* a, void* stream) { return conv_forward_impl(a, (hipStream_t)stream, 0); }
extern "C" int ofd_conv_forward_pool2(const ofd_conv_args* a, void* stream) {
    OFD_CHECK_ARG(conv_pool2_supported(a), "conv_forward_pool2: 3x3, even H and W, Cout a multiple of 64 (not 64 -> 64), no split / GroupNorm statistics / activation residual");
    return conv_forward_impl(a, (hipStream_t)stream, 0, 1);
}

extern "C" size_t ofd_conv_gn_partial_count(int B, int H, int W, int Cout) {
    return (size_t)B * cdiv(H, TH) * cdiv(W, TW) * 4 * (Cout / 8) * 2;     // [b][8 groups][tile][wave slot][Cout/64][2] (gn_partial_index)
}

extern "C" size_t ofd_conv_weight_elems(int Cout, int Cin_pad, int ksize) {
    if (ksize == 7 && Cin_pad == 8) return (size_t)7 * 8 * 8 * Cout;      // tap-pair packing: 8 tap columns per kernel row (the 8th zero)
    return (size_t)ksize * ksize * Cin_pad * Cout;
}

extern "C" int ofd_conv_weight_prep(const float* w_oihw, void* w_out, int Cout, int Cin, int Cin_pad, int ksize, float ws_eps,
                                    int unshuffle, void* stream) {
    OFD_CHECK_ARG(w_oihw && w_out && Cout > 0 && Cin > 0 && Cin_pad >= Cin && Cin_pad % 8 == 0, "weight_prep: bad argument");
    OFD_CHECK_ARG(!unshuffle || (Cin % 4 == 0 && Cin == Cin_pad), "weight_prep: unshuffle needs Cin %% 4 == 0");
    conv_weight_prep_kernel<<<Cout, 256, 0, (hipStream_t)stream>>>(w_oihw, (bf16_t*)w_out, Cout, Cin, Cin_pad, ksize, ws_eps, unshuffle);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

namespace ofd {
int k_conv_weight_prep_batched(const ofd_weight_prep_desc* d_descs, int n, int total_blocks, hipStream_t s) {
    conv_weight_prep_batched_kernel<<<total_blocks, 256, 0, s>>>(d_descs, n);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
}  // namespace ofd

extern "C" int ofd_conv_upsample_phase_weight_prep(const float* w_oihw, void* w_out, int Cout, int Cin, void* stream) {
    OFD_CHECK_ARG(w_oihw && w_out && Cout > 0 && Cin > 0 && Cin % 8 == 0, "upsample_phase_weight_prep: bad argument");
    upsample_phase_weight_prep_kernel<<<Cout, 256, 0, (hipStream_t)stream>>>(w_oihw, (bf16_t*)w_out, Cout, Cin);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
