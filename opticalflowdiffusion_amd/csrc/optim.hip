// Optimiser step of the FlowDiffuser training path (flow_diffuser.py:131-134 + exp_base.py:192,205):
// torch.optim.Adam(lr, weight_decay) = Adam with L2 added to the gradient (not AdamW), betas
// (0.9, 0.999), eps 1e-8, preceded by clip_grad_norm_(gradient_clip_val).  Multi-tensor kernels
// over a device table of {param, grad, exp_avg, exp_avg_sq, numel}: three launches per step, no
// host synchronisation (the clip coefficient stays on the device).
#include "common.h"

namespace ofd {

struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    unsigned long long n;
};

constexpr int OPT_CHUNK = 65536;   // elements per workgroup task

// sum of squares of a task's chunk -> part[task] (double): no atomics, clip_coef_kernel adds the tasks up in a fixed order (the same
// gradients give the same norm, bit for bit -- a double atomic per workgroup made the clip coefficient, and with it every parameter,
// depend on the order the workgroups retired in)
__global__ void __launch_bounds__(256) grad_sqnorm_kernel(const AdamTensor* __restrict__ tab, const unsigned* __restrict__ task_tensor,
                                                          const unsigned* __restrict__ task_chunk, double* __restrict__ part) {
    const AdamTensor t = tab[task_tensor[blockIdx.x]];
    const unsigned long long start = (unsigned long long)task_chunk[blockIdx.x] * OPT_CHUNK;
    const unsigned long long stop = min(t.n, start + OPT_CHUNK);
    double s = 0.0;
    for (unsigned long long i = start + threadIdx.x; i < stop; i += 256) {
        const float g = t.g[i];
        s += (double)g * (double)g;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// clip_grad_norm_: coef = min(1, max_norm / (sqrt(sum) + 1e-6)); max_norm <= 0 disables clipping
// acc[0] <- sum of acc[1 .. n_tasks] (thread t adds tasks t, t + 256, ...; then a fixed tree)
__global__ void __launch_bounds__(256) clip_coef_kernel(double* __restrict__ acc, int n_tasks, float max_norm, float* __restrict__ coef, float* __restrict__ total_norm) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_tasks; i += 256) s += acc[1 + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    acc[0] = sh[0];
    const float norm = (float)sqrt(sh[0]);
    if (total_norm) *total_norm = norm;
    float c = 1.0f;
    if (max_norm > 0.0f) c = fminf(1.0f, max_norm / (norm + 1e-6f));
    *coef = c;
}

__global__ void __launch_bounds__(256) adam_step_kernel(const AdamTensor* __restrict__ tab, const unsigned* __restrict__ task_tensor,
                                                        const unsigned* __restrict__ task_chunk, const float* __restrict__ coef, float lr,
                                                        float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt) {
    const AdamTensor t = tab[task_tensor[blockIdx.x]];
    const unsigned long long start = (unsigned long long)task_chunk[blockIdx.x] * OPT_CHUNK;
    const unsigned long long stop = min(t.n, start + OPT_CHUNK);
    const float c = coef ? *coef : 1.0f;
    const float step_size = lr / bc1;
    for (unsigned long long i = start + threadIdx.x; i < stop; i += 256) {
        float p = t.p[i];
        float g = t.g[i] * c;
        g = g + weight_decay * p;                                  // L2 in the gradient (torch.optim.Adam)
        const float m = beta1 * t.m[i] + (1.0f - beta1) * g;
        const float v = beta2 * t.v[i] + (1.0f - beta2) * g * g;
        t.m[i] = m;
        t.v[i] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        t.p[i] = p - step_size * (m / denom);
    }
}

}  // namespace ofd
using namespace ofd;

// table: n_tensors x {p, g, m, v, n} (device); tasks: n_tasks x (tensor index, chunk index) (device, 2 arrays)
extern "C" int ofd_adam_step(const void* table, const unsigned* task_tensor, const unsigned* task_chunk, int n_tasks, double* sqnorm_acc,
                             float* clip_coef, float* total_norm, float max_norm, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, void* stream) {
    OFD_CHECK_ARG(table && task_tensor && task_chunk && n_tasks > 0 && sqnorm_acc && clip_coef && step >= 1, "adam_step: bad argument");
    hipStream_t s = (hipStream_t)stream;
    grad_sqnorm_kernel<<<n_tasks, 256, 0, s>>>((const AdamTensor*)table, task_tensor, task_chunk, sqnorm_acc + 1);
    clip_coef_kernel<<<1, 256, 0, s>>>(sqnorm_acc, n_tasks, max_norm, clip_coef, total_norm);
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    adam_step_kernel<<<n_tasks, 256, 0, s>>>((const AdamTensor*)table, task_tensor, task_chunk, clip_coef, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_adam_chunk(void) { return OPT_CHUNK; }
