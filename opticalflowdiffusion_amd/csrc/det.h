// Order-independent gradient accumulation (opt-in: ofd_unet_set_deterministic / OFD_DETERMINISTIC=1).
//
// The backward pass adds partial sums from many workgroups into one fp32 slot with global float atomics (weight gradients, bias /
// GroupNorm / LayerNorm parameter gradients, the time-embedding gradient): the adds land in whatever order the workgroups retire,
// fp32 addition is not associative, so two runs of the same step differ in the last bits and the trajectories drift apart.  In
// deterministic mode every such add goes to a 64-bit FIXED-POINT shadow of the slot instead (value * 2^38, round to nearest, integer
// atomic add): integer addition IS associative, the sum is the same whatever the order.  The shadows are flushed into the fp32
// buffers (slot += shadow * 2^-38, one rounding) before anything reads them.
//   range  +-3.3e7 per slot (sum of the partials), resolution 3.6e-12 absolute: far below Adam's eps (1e-8) scale
//   partials above 1e4 in magnitude and non-finite ones bypass the shadow and go to the fp32 slot itself (float atomic): the shadow
//   cannot wrap with fewer than 3300 partials per slot, and an overflow still surfaces as inf / nan
// The kernels keep their float* arguments: gacc_add() looks the pointer up in the (at most DET_RANGES) registered fp32 buffers of the
// executor and derives the shadow address; a pointer outside every range falls back to the float atomic and counts a miss
// (DetCtx::misses, checked by the tests).  The context is a per-translation-unit __constant__ variable: OFD_DET_DEFINE_SETTER(name)
// instantiates its host-side setter in each .hip that uses gacc_add.
#pragma once
#include <hip/hip_runtime.h>

namespace ofd {

constexpr int DET_RANGES = 3;
constexpr float DET_SCALE = 274877906944.0f;                 // 2^38
constexpr double DET_INV_SCALE = 1.0 / 274877906944.0;

struct DetRange {
    const float* f;          // fp32 buffer
    long long* fx;           // its fixed-point shadow, same element index
    size_t n;
};
struct DetCtx {
    int on;
    unsigned* misses;        // device counter of adds that matched no range (deterministic mode only)
    DetRange r[DET_RANGES];
};

#ifdef __HIPCC__
static __constant__ DetCtx g_det_ctx;      // constant address space: one scalar load per kernel, not one per add (atomics do not clobber it)

// every cross-workgroup gradient accumulation of the backward pass goes through here
__device__ __forceinline__ void gacc_add(float* p, float v) {
    if (!g_det_ctx.on) { atomicAdd(p, v); return; }
    if (!(fabsf(v) <= 1.0e4f)) { atomicAdd(p, v); return; }            // inf / nan / huge: into the fp32 slot itself (3300 partials of 1e4 fit the shadow)
#pragma unroll
    for (int i = 0; i < DET_RANGES; ++i) {
        const DetRange r = g_det_ctx.r[i];
        if (p >= r.f && p < r.f + r.n) {
            atomicAdd((unsigned long long*)(r.fx + (p - r.f)), (unsigned long long)__float2ll_rn(v * DET_SCALE));
            return;
        }
    }
    atomicAdd(p, v);
    if (g_det_ctx.misses) atomicAdd(g_det_ctx.misses, 1u);
}

#define OFD_DET_DEFINE_SETTER(name)                                                                                         \
    int name(const DetCtx* host_ctx, hipStream_t s) {                                                                       \
        return (int)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_det_ctx), host_ctx, sizeof(DetCtx), 0, hipMemcpyHostToDevice, s);   \
    }
#endif

// setters of the translation units that accumulate gradients (each returns a hipError_t as int)
int det_set_ctx_conv_bwd(const DetCtx* host_ctx, hipStream_t s);
int det_set_ctx_la_core(const DetCtx* host_ctx, hipStream_t s);
int det_set_ctx_train_ops(const DetCtx* host_ctx, hipStream_t s);
// slot[i] += shadow[i] * 2^-38 ; shadow[i] = 0        (train_ops.hip)
int k_det_flush(long long* fx, float* f, size_t n, hipStream_t s);

}  // namespace ofd
