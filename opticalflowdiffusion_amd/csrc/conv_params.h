// Device-side descriptors shared by the forward (conv_igemm.hip) and backward (conv_bwd.hip) conv kernels.
#pragma once
#include "common.h"

namespace ofd {

struct ConvSrcDev {
    const bf16_t* ptr;
    int chunks;         // K-chunks (of CK channels) taken from this source
    int src_channels;   // pixel stride of the source tensor
    int ch_offset;
    int SH, SW;         // source spatial size
    int mode;           // 0: same size, 1: nearest x2 up-sample, 2: pixel-unshuffle sub-pixel (p1,p2)
    int p1, p2;
};

struct ConvParams {
    int B, H, W, Cout, Cin_total, n_src, total_chunks, tiles_x, tiles_y;
    ConvSrcDev src[4];
    const bf16_t* weight;
    const float* bias;
    const float* in_scale;
    const float* in_shift;
    const bf16_t* residual;
    const bf16_t* res_act;
    const float* res_scale;
    const float* res_shift;
    bf16_t* out;
    float* gn_partial;
    // split output (data gradient of a conv over two concatenated sources): channels [0, split) go to `out` / `residual`
    // with pixel stride `split`, channels [split, Cout) to out2 / residual2 with stride Cout - split; split = 0: off
    bf16_t* out2;
    const bf16_t* residual2;
    int split;
    int pad_y, pad_x;   // KS == 2 (one phase of an up-sampled 3x3): rows / columns of padding above / left of the tile
    int out_oy, out_ox; // KS == 2: the output is (2H, 2W) and this launch writes pixels (2y + out_oy, 2x + out_ox)
    int phase_all;      // KS == 2: one launch computes the four phases; block j -> XCD j % 8, phase (j / 8) % 4, tile slot j / 32 (the phases of a
                        // tile run on one XCD at the same time: its input tile comes from that L2 three times out of four)
    int cout0;          // 1x1 streaming kernel: output channels below cout0 are not computed (training to_qkv with q recomputed downstream); 0 = all
    int cy_fast;        // conv_wp: 1-D grid, the channel blocks of a pixel tile adjacent in dispatch order and on one XCD (A/B switch)
    const bf16_t* residual_b;   // conv_wp 3x3: a second plain residual (same shape as `residual`; training: gradient already in the buffer + the identity-residual gradient)
    int pool2;          // conv_wp 3x3: the epilogue sums every 2x2 block of output pixels and writes the (H/2, W/2) tensor (+ residual there): the
                        // data gradient of Upsample(x2, nearest) + conv lands in the low-resolution source's gradient without a full-size tensor
    // streaming 1x1 (128 -> 64 with the fused SiLU(GN(h2)) input) only: the UNet's final 1x1 conv (DD:361, out_dim 2) on the tile the kernel
    // holds -- fc_out (B, 2, H, W) fp32, ZEROED by the caller, receives the two 32-channel partial sums of a pixel as float atomics (two
    // addends onto zero: the order cannot change the sum); the bf16 `out` tensor is then NOT written
    const float* fc_w;  // [2][Cout]
    const float* fc_b;  // [2]
    float* fc_out;
    int dbg;            // diagnostic ablation bits (OFD_CONV_DBG), 0 in production
};

// GroupNorm partial sums written by the conv epilogues and added up by gn_finalize (blocks.hip), GROUP-major:
//     [b][group 0..7][slot][octet within the group: Cout / 64 of them][sum, sum of squares]
// slot = (8-row tile * tiles_x + tile column) * 4 + wave slot, `slots` of them per sample; noct = Cout / 8 octets (8-channel units), oct the
// octet's index in the tensor.  One (sample, group) is one contiguous run, so its gn_finalize workgroup reads its own bytes only (with the
// octet-major [b][slot][Cout/8][2] every one of a sample's 8 workgroups pulled every 64-byte line of the tensor: 17 us per launch at
// 1 x 1080 x 1920, 38 launches per UNet forward).
__host__ __device__ __forceinline__ size_t gn_partial_index(int b, int slots, int slot, int noct, int oct) {
    const int octs = noct >> 3, g = oct / octs;
    return ((((size_t)b * 8 + g) * slots + slot) * octs + (oct - g * octs)) * 2;
}

}  // namespace ofd
