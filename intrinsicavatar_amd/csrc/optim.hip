// optim.hip -- the optimiser step of the training iteration: torch.optim.Adam as the reference configures it
// (configs/config.yaml:110-136: Adam, betas (0.9, 0.99), eps 1e-15, per-group lr / L2 weight_decay; built by
// systems/utils.py:314-325), for ALL parameter tensors of the step in one launch.
//
// The step is a pure stream over four arrays (param, grad, exp_avg, exp_avg_sq): 16 B read + 12 B written per element,
// 25.2 M elements in the two hash tables -> 706 MB -> ~0.14 ms at HBM speed.  torch's per-tensor path issues ~10 kernels
// per tensor (lerp, mul, addcmul, sqrt, div, add, addcdiv ...) over ~25 tensors; here one kernel walks a descriptor
// table, each workgroup owning one 4096-element chunk of one tensor, with float4 accesses.  The gradient all-reduce
// scaling (1/world for DDP's mean) is folded in as grad_scale.
//
// Per-element operation order follows torch/optim/adam.py:_single_tensor_adam (no amsgrad, no maximize):
//   g  = grad * grad_scale (+ weight_decay * p)
//   m  = m + (g - m) * (1 - beta1)                        exp_avg.lerp_(grad, 1 - beta1)
//   v  = v * beta2 + (1 - beta2) * g * g                  exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//   p  = p - step_size * m / (sqrt(v) / bias_correction2_sqrt + eps)
// with step_size = lr / (1 - beta1^t) and bias_correction2_sqrt = sqrt(1 - beta2^t) formed in double on the host, as
// torch does with python floats.
#include "ia_common.h"

namespace {

constexpr int THREADS = 256;
constexpr int VEC = 4;
constexpr int CHUNK = THREADS * VEC * 4;      // 4096 elements per workgroup
constexpr int MAX_T = 40;

struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    float step_size, wd;
    int block0, pad;
};

struct AdamArgs {
    int n_tensors;
    float beta1, beta2, eps, bc2_sqrt, grad_scale;
    AdamTensor t[MAX_T];
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float step_size, float wd)
{
    g = g * a.grad_scale;
    if (wd != 0.0f) g = g + wd * p;
    m = m + (g - m) * (1.0f - a.beta1);
    v = v * a.beta2 + (1.0f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + (-step_size) * (m / denom);
}

__global__ __launch_bounds__(THREADS) void adam_kernel(const AdamArgs a)
{
    // workgroup -> tensor: descriptors are kernel arguments (scalar registers), n_tensors <= 40
    int ti = 0;
    for (int k = 1; k < a.n_tensors; k++)
        if ((int)blockIdx.x >= a.t[k].block0) ti = k;
    const AdamTensor& T = a.t[ti];
    const int64_t base = (int64_t)((int)blockIdx.x - T.block0) * CHUNK;
    const float step_size = T.step_size, wd = T.wd;
    const bool vec_ok = ((((uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v) & 15) == 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int64_t i = base + ((int64_t)r * THREADS + threadIdx.x) * VEC;
        if (i >= T.n) break;
        if (vec_ok && i + VEC <= T.n) {
            float4 p = *reinterpret_cast<const float4*>(T.p + i);
            const float4 g = *reinterpret_cast<const float4*>(T.g + i);
            float4 m = *reinterpret_cast<const float4*>(T.m + i);
            float4 v = *reinterpret_cast<const float4*>(T.v + i);
            adam_one(p.x, g.x, m.x, v.x, a, step_size, wd);
            adam_one(p.y, g.y, m.y, v.y, a, step_size, wd);
            adam_one(p.z, g.z, m.z, v.z, a, step_size, wd);
            adam_one(p.w, g.w, m.w, v.w, a, step_size, wd);
            *reinterpret_cast<float4*>(T.p + i) = p;
            *reinterpret_cast<float4*>(T.m + i) = m;
            *reinterpret_cast<float4*>(T.v + i) = v;
        } else {
            for (int64_t j = i; j < i + VEC && j < T.n; j++) {
                float p = T.p[j], m = T.m[j], v = T.v[j];
                adam_one(p, T.g[j], m, v, a, step_size, wd);
                T.p[j] = p; T.m[j] = m; T.v[j] = v;
            }
        }
    }
}

}  // namespace

IA_EXPORT int ia_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, const int64_t* numel, const float* step_size,
                           const float* weight_decay, float beta1, float beta2, float eps, float bias_correction2_sqrt,
                           float grad_scale, ia_stream_t stream)
{
    if (n_tensors == 0) return IA_OK;
    IA_REQUIRE(n_tensors > 0 && params && grads && exp_avg && exp_avg_sq && numel && step_size && weight_decay,
               "null descriptor array");
    IA_REQUIRE(bias_correction2_sqrt > 0.0f, "bias_correction2_sqrt must be positive (step >= 1)");
    int t = 0;
    while (t < n_tensors) {
        AdamArgs a = {};
        a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.bc2_sqrt = bias_correction2_sqrt; a.grad_scale = grad_scale;
        int64_t blocks = 0;
        int k = 0;
        for (; t < n_tensors && k < MAX_T; t++) {
            if (numel[t] == 0) continue;
            IA_REQUIRE(numel[t] > 0 && params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t], "bad tensor descriptor");
            IA_REQUIRE(blocks < ((int64_t)1 << 30), "too many elements for one launch");
            AdamTensor& T = a.t[k++];
            T.p = params[t]; T.g = grads[t]; T.m = exp_avg[t]; T.v = exp_avg_sq[t]; T.n = numel[t];
            T.step_size = step_size[t]; T.wd = weight_decay[t]; T.block0 = (int)blocks;
            blocks += (numel[t] + CHUNK - 1) / CHUNK;
        }
        a.n_tensors = k;
        if (k == 0) continue;
        adam_kernel<<<(int)blocks, THREADS, 0, (hipStream_t)stream>>>(a);
        int r = ia::check_launch("ia_adam_step");
        if (r != IA_OK) return r;
    }
    return IA_OK;
}
