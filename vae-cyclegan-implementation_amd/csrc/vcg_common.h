// Shared host/device helpers for libvcg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vcg.h"

#define VCG_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void vcg_set_error(const char* fmt, ...);

#define VCG_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      vcg_set_error(__VA_ARGS__);     \
      return -1;                      \
    }                                 \
  } while (0)

#define VCG_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      vcg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return -2;                                                            \
    }                                                                       \
  } while (0)

// Optional per-device-kernel timing (vcg_profile_enable / vcg_profile_read, misc.hip): while enabled, the MFMA kernel
// launches bracket themselves with HIP events on their launch stream and carry the FLOPs they execute.  Off: one branch.
extern bool g_vcg_prof_on;
void vcg_prof_begin(const char* kernel, double flops, hipStream_t st);
void vcg_prof_end(hipStream_t st);
struct VcgProfScope {
  hipStream_t st;
  bool on;
  VcgProfScope(const char* kernel, double flops, hipStream_t s) : st(s), on(g_vcg_prof_on) { if (on) vcg_prof_begin(kernel, flops, s); }
  ~VcgProfScope() { if (on) vcg_prof_end(st); }
};

// division by a runtime constant: q = (umulhi(n, mul) + n) >> sh, valid for n < 2^31
struct FastDiv {
  uint32_t d, mul, sh;
};

static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) {
    f.mul = 0;
    f.sh = 0;
    return f;
  }
  uint32_t sh = 0;
  while ((1ull << sh) < d) ++sh;
  f.sh = sh;
  f.mul = (uint32_t)((((1ull << sh) - d) << 32) / d + 1);
  return f;
}

__host__ __device__ static inline uint32_t fd_div(uint32_t n, const FastDiv& f) {
#ifdef __HIP_DEVICE_COMPILE__
  return (__umulhi(n, f.mul) + n) >> f.sh;
#else
  return (uint32_t)((((uint64_t)n * f.mul) >> 32) + n) >> f.sh;
#endif
}

__host__ __device__ static inline void fd_divmod(uint32_t n, const FastDiv& f, uint32_t& q, uint32_t& r) {
  q = fd_div(n, f);
  r = n - q * f.d;
}

// reflect (no edge repeat) index map of torch 'reflect' padding; valid for -L < i < 2L-1
__host__ __device__ static inline int reflect_idx(int i, int L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// NaN goes through both (torch.relu(nan) is nan: the reference's NaN guard, Networks.py:357, relies on it reaching the loss)
__device__ static inline float act_apply(float v, int act) {
  if (act == VCG_ACT_RELU) return v < 0.f ? 0.f : v;
  if (act == VCG_ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
  if (act == VCG_ACT_TANH) return tanhf(v);
  if (act == VCG_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}
// derivative expressed on the activation OUTPUT (ReLU / LeakyReLU preserve the sign; tanh' = 1 - out^2, sigmoid' = out (1 - out))
__device__ static inline float act_grad_from_out(float out, int act) {
  if (act == VCG_ACT_RELU) return out > 0.f ? 1.f : 0.f;
  if (act == VCG_ACT_LEAKY02) return out > 0.f ? 1.f : 0.2f;
  if (act == VCG_ACT_TANH) return 1.f - out * out;
  if (act == VCG_ACT_SIGMOID) return out * (1.f - out);
  return 1.f;
}
// derivative expressed on the activation INPUT (the InstanceNorm backward holds xhat, not act(xhat))
__device__ static inline float act_grad_from_in(float x, int act) {
  if (act == VCG_ACT_TANH || act == VCG_ACT_SIGMOID) return act_grad_from_out(act_apply(x, act), act);
  return act_grad_from_out(x, act);
}

__device__ static inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// split-operand bf16 arithmetic (gemm_split.hip): 4 consecutive k of one row -> three 8-byte bf16 quads
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split4(const float4& v, uint2& h, uint2& m, uint2& l) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned short hs[4], ms[4], ls[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 hb = (__bf16)x[i];
    const float r1 = x[i] - (float)hb;
    const __bf16 mb = (__bf16)r1;
    const float r2 = r1 - (float)mb;
    const __bf16 lb = (__bf16)r2;
    hs[i] = __builtin_bit_cast(unsigned short, hb);
    ms[i] = __builtin_bit_cast(unsigned short, mb);
    ls[i] = __builtin_bit_cast(unsigned short, lb);
  }
  h = make_uint2((uint32_t)hs[0] | ((uint32_t)hs[1] << 16), (uint32_t)hs[2] | ((uint32_t)hs[3] << 16));
  m = make_uint2((uint32_t)ms[0] | ((uint32_t)ms[1] << 16), (uint32_t)ms[2] | ((uint32_t)ms[3] << 16));
  l = make_uint2((uint32_t)ls[0] | ((uint32_t)ls[1] << 16), (uint32_t)ls[2] | ((uint32_t)ls[3] << 16));
}

// InstanceNorm reductions (norm.hip): a workgroup sums one chunk of pixels for TC channel quads x TP pixel lanes; the chunk
// partials ([N][nchunk][C][2] doubles) are combined in double by k_in_final.  Shared with the conv epilogues that emit
// those partials themselves (vcg_conv_fwd_in).
struct NormPlan {
  int TC, TP, cgroups, nchunk, chunk;
};
static inline NormPlan vcg_norm_plan(int N, int HW, int C) {
  NormPlan pl;
  int c4 = C / 4;
  int tc = 1;
  while (tc * 2 <= c4 && tc * 2 <= 256) tc *= 2;
  pl.TC = tc;
  pl.TP = 256 / tc;
  pl.cgroups = (c4 + tc - 1) / tc;
  long long target = 1024 / ((long long)N * pl.cgroups);
  if (target < 1) target = 1;
  long long maxc = (HW + pl.TP * 2 - 1) / (pl.TP * 2);
  if (maxc < 1) maxc = 1;
  if (target > maxc) target = maxc;
  pl.chunk = (int)((HW + target - 1) / target);
  pl.nchunk = (HW + pl.chunk - 1) / pl.chunk;
  return pl;
}
int vcg_in_finalize(const double* part, float* mean, float* rstd, int N, int HW, int C, int nchunk, float eps, hipStream_t st);
int vcg_in_stats_pass(const float* t, float* mean, float* rstd, int N, int HW, int C, float eps, void* ws, size_t ws_bytes, hipStream_t st);

// geometry derived from the int32[16] conv descriptor
struct ConvGeom {
  int N, H, W, Cin, Cout, KH, KW, stride, pad, reflect, ups, act, cin_log, cout_log;
  int Hl, Wl, Ho, Wo, M, K, taps;
};

int vcg_conv_geom(const int32_t* cd, ConvGeom* g, const char* who);

// conv_thin.hip: direct kernels for layers with <= 4 channels on the narrow side
bool vcg_thin_fwd_ok(const ConvGeom& g);
bool vcg_thin_dgrad_ok(const ConvGeom& g);
size_t vcg_thin_dgrad_workspace(const ConvGeom& g);
int vcg_thin_fwd(const ConvGeom& g, const float* x, const float* wf, const float* bias, float* y, hipStream_t st);
int vcg_thin_dgrad(const ConvGeom& g, const float* dy, const float* wf, float* dx, void* ws, size_t ws_bytes,
                   hipStream_t st);

// conv_wino.hip: Winograd F(2x2,3x3) forward for the 3x3 / stride-1 / pad-1 layers
bool vcg_wino_weight_ok(const ConvGeom& g);
bool vcg_wino_fwd_ok(const ConvGeom& g);
size_t vcg_wino_weight_floats(const ConvGeom& g);
size_t vcg_wino_fwd_workspace(const ConvGeom& g);
int vcg_wino_weight(const ConvGeom& g, const float* w_oihw, float* u, hipStream_t st);
int vcg_wino_fwd(const ConvGeom& g, const float* x, const float* u, const float* bias, float* y, void* ws, size_t ws_bytes,
                 hipStream_t st, double* in_part = nullptr, int* in_nchunk = nullptr, float* v_keep = nullptr);
size_t vcg_wino_saved_floats(const ConvGeom& g);
size_t vcg_wino_fwd_stats_doubles(const ConvGeom& g);
// Winograd weight gradient: transforms in conv_wino.hip, batched stream-K reduction + back-transform in conv_igemm.hip
bool vcg_wino_wgrad_ok(const ConvGeom& g);
size_t vcg_wino_wgrad_workspace(const ConvGeom& g);
int vcg_wino_wgrad(const ConvGeom& g, const float* x, const float* dy, float* gw_oihw, void* ws, size_t ws_bytes, hipStream_t st,
                   const float* v_saved = nullptr);
size_t vcg_wino_wgrad_core_workspace(const ConvGeom& g, int T);
int vcg_wino_wgrad_core(const ConvGeom& g, const float* V, const float* dM, int T, float* gw_oihw, void* ws, size_t ws_bytes,
                        hipStream_t st);
bool vcg_wino_dgrad_ok(const ConvGeom& g);
size_t vcg_wino_dgrad_workspace(const ConvGeom& g);
int vcg_wino_weight_dgrad(const ConvGeom& g, const float* w_oihw, float* ud, hipStream_t st);
int vcg_wino_dgrad(const ConvGeom& g, const float* dy, const float* ud, float* dx, void* ws, size_t ws_bytes, hipStream_t st);
// conv_slab.hip: 3x3 / stride-1 layers with few channels on large maps — the input staged once per workgroup as an LDS slab
bool vcg_slab_fwd_ok(const ConvGeom& g);
bool vcg_slab_fwd_stats_ok(const ConvGeom& g);
int vcg_slab_fwd_nchunk(const ConvGeom& g);
bool vcg_slab_dgrad_ok(const ConvGeom& g);
size_t vcg_slab_dgrad_workspace(const ConvGeom& g);
int vcg_slab_fwd(const ConvGeom& g, const float* x, const void* wft_planes, size_t planes_bytes, const float* bias, float* y,
                 double* in_part, int* in_nchunk, hipStream_t st);
int vcg_slab_dgrad(const ConvGeom& g, const float* dy, const void* wfd_planes, size_t planes_bytes, float* dx, void* ws,
                   size_t ws_bytes, hipStream_t st);
// conv_ring.hip: weight gradients of U4 / head / stem with row-ring staging (the taps are address offsets of the fragment reads)
bool vcg_ring_wgrad_ok(const ConvGeom& g);
size_t vcg_ring_wgrad_workspace(const ConvGeom& g);
int vcg_ring_wgrad(const ConvGeom& g, const float* x, const float* dy, float* gw_oihw, void* ws, size_t ws_bytes, hipStream_t st);
// conv_thin.hip: thin forward with kw folded into the GEMM's N (MFMA)
bool vcg_thin_fold_ok(const ConvGeom& g);
size_t vcg_thin_fold_weight_floats(const ConvGeom& g);
size_t vcg_thin_fold_workspace(const ConvGeom& g);
int vcg_thin_fold_pack(const ConvGeom& g, const float* w_oihw, float* wk, hipStream_t st);
int vcg_thin_fold_fwd(const ConvGeom& g, const float* x, const float* wk, const float* bias, float* y, void* ws, size_t ws_bytes,
                      hipStream_t st);
bool vcg_thin_fold_dgrad_ok(const ConvGeom& g);
size_t vcg_thin_fold_dgrad_weight_floats(const ConvGeom& g);
size_t vcg_thin_fold_dgrad_workspace(const ConvGeom& g);
int vcg_thin_fold_dgrad_pack(const ConvGeom& g, const float* w_oihw, float* wk, hipStream_t st);
int vcg_thin_fold_dgrad(const ConvGeom& g, const float* dy, const float* wkd, float* dx, void* ws, size_t ws_bytes, hipStream_t st);
