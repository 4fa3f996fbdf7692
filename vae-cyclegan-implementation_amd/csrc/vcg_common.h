// Shared host/device helpers for libvcg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
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

// ---- split-operand arithmetic (gemm_split.hip) -------------------------------------------------------------------
// Round 3: fp16 x 2.  Every fp32 operand x of an MFMA kernel is staged as  x / s = h + l  with h = fp16(x / s), l = fp16(x / s - h)
// (2 x 11 = 22 mantissa bits) and s a power of two — exact to apply — that brings the operand TENSOR's largest magnitude into
// [2^14, 2^15), under fp16's 65504.  Three products of relative weight >= 2^-11 (hh, hl, lh) go into fp32 MFMA accumulators, the
// result is multiplied by sA * sB in the epilogue; the dropped ll term is below 2^-22.  Measured against float64 a GEMM rounds at
// 2.1e-7 (the bf16 x 3 / six-product form of rounds 1-2: 1.7e-7; PyTorch-CPU fp32: 2.2-3.0e-7) — and issues half the MFMAs,
// which on a chip that answers a denser MFMA stream with a lower clock is what moves the wall time (profiles/r03_gemm_fp16x2_probe.txt).
// Elements more than 2^17 below the tensor's largest magnitude lose relative (not absolute) precision: their error stays 2^-40
// of that magnitude, below the fp32 rounding of any sum they enter.
#define VCG_NP 2                          // pieces per operand
#define VCG_PBLK (32 * VCG_NP)            // 16-bit elements per (row, 32-k block) of "blocked planes": [piece][32]
#define VCG_PBYTES (2 * VCG_PBLK)         // 128 bytes
#define VCG_PFLOATS (VCG_PBYTES / 4)      // 32 floats
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));   // (tools/ probes of the rounds 1-2 arithmetic)
#define VCG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// 4 consecutive k of one row, already multiplied by 1 / s -> two 8-byte fp16 quads
__device__ __forceinline__ void split4h(const float4& v, float inv, uint2& h, uint2& l) {
  const float x[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
  _Float16 hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = (_Float16)x[i];
    ll[i] = (_Float16)(x[i] - (float)hh[i]);
  }
  const f16x2v h0 = {hh[0], hh[1]}, h1 = {hh[2], hh[3]}, l0 = {ll[0], ll[1]}, l1 = {ll[2], ll[3]};
  h = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
  l = make_uint2(__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1));
}

// The largest magnitude of an operand tensor ("amax") travels as the bit pattern of |x| (non-negative floats order like their
// bits; a NaN outranks everything and an Inf every finite value, so a poisoned tensor is seen as such) in a SLOT of 64 u64
// words {generation << 32 | bits}: producers (k_absmax, the Winograd transforms) atomicMax their block's maximum into word
// blockIdx % 64, consumers take the maximum over the words whose generation is theirs.  The host hands out (slot, generation)
// per call with a strictly increasing generation, so a slot never needs clearing: whatever an earlier call left loses
// against the first atomicMax of this one.  The slots live in a __device__ array of the code object (misc.hip).
struct VcgAmax {
  const unsigned long long* slot;     // 64 words written in THIS call sequence, or null:
  const uint32_t* stored;             //   the amax bits as an earlier call stored them (a weight pack's header, a kept V), or null:
  uint32_t gen, bits;                 //   `bits` as given by the host
  int shift;                          // the tensor the kernel reads is bounded by 2^shift * amax (Winograd transforms of a tensor)
};
struct VcgAmaxOut { unsigned long long* slot; uint32_t gen; };
VcgAmaxOut vcg_amax_new(hipStream_t st);                                // a fresh (slot, generation) on the current device
static inline VcgAmax vcg_amax_in(const VcgAmaxOut& o, int shift = 0) { VcgAmax a; a.slot = o.slot; a.stored = nullptr; a.gen = o.gen; a.bits = 0; a.shift = shift; return a; }
static inline VcgAmax vcg_amax_stored(const void* bits_ptr, int shift = 0) { VcgAmax a; a.slot = nullptr; a.stored = (const uint32_t*)bits_ptr; a.gen = 0; a.bits = 0; a.shift = shift; return a; }
static inline VcgAmax vcg_amax_const(uint32_t bits, int shift = 0) { VcgAmax a; a.slot = nullptr; a.stored = nullptr; a.gen = 0; a.bits = bits; a.shift = shift; return a; }
int vcg_absmax_launch(const float* t, size_t n, const VcgAmaxOut& out, hipStream_t st);      // t 16-byte aligned
// A HANDLE names a published amax across C-ABI calls (include/vcg.h, vcg_amax_hint / vcg_amax_last): the kernels that write an
// activation or a gradient (InstanceNorm apply / backward, activation backward) publish its largest magnitude as a by-product,
// and the convolution that reads the tensor — in the same forward, or many calls later in the backward — takes the handle
// instead of measuring the tensor again.  A handle goes stale when its slot is about to be reused (the generation counter has
// moved on by nearly the slot count): the consumer then measures, as it does for a tensor without a handle.
uint64_t vcg_amax_handle(const VcgAmaxOut& o);
// the operand's amax: from `handle` when it is still valid, else from a pass over the tensor (t, n)
int vcg_operand_amax(const float* t, size_t n, uint64_t handle, int shift, hipStream_t st, VcgAmax* out);
uint64_t vcg_take_hint_x();          // the handles the caller announced for the NEXT conv call on this thread (consumed)
uint64_t vcg_take_hint_dy();
void vcg_set_last_amax(uint64_t h);  // what vcg_amax_last() returns

#ifdef __HIPCC__
// block-wide: every thread contributes the bit pattern of a magnitude; one atomicMax per block.  `red` = 4+ words of LDS.
__device__ __forceinline__ void vcg_amax_publish(uint32_t mybits, unsigned long long* slot, uint32_t gen, uint32_t* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)mybits, o, 64);
    mybits = other > mybits ? other : mybits;
  }
  const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) red[wid] = mybits;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = red[0];
    for (int w = 1; w < nw; ++w) m = red[w] > m ? red[w] : m;
    atomicMax(slot + (blockIdx.x & 63), ((unsigned long long)gen << 32) | m);
  }
}
__device__ __forceinline__ uint32_t vcg_abs_bits(float v) { return __float_as_uint(v) & 0x7FFFFFFFu; }
__device__ __forceinline__ uint32_t vcg_abs_bits4(const float4& v) {
  const uint32_t a = vcg_abs_bits(v.x), b = vcg_abs_bits(v.y), c = vcg_abs_bits(v.z), d = vcg_abs_bits(v.w);
  const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
// every wave for itself (no LDS, no barrier): the amax bits of an operand
__device__ __forceinline__ uint32_t vcg_amax_bits(const VcgAmax& a) {
  if (!a.slot) return a.stored ? *a.stored : a.bits;
  const unsigned long long w = a.slot[threadIdx.x & 63];
  uint32_t b = (uint32_t)(w >> 32) == a.gen ? (uint32_t)w : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)b, o, 64);
    b = other > b ? other : b;
  }
  return b;
}
#endif
// s = 2^(E - 14 + shift) for amax in [2^E, 2^(E+1)): amax / s lands in [2^14, 2^15).  A zero, subnormal, infinite or NaN amax
// gives s = 2^shift (nothing to scale; a poisoned tensor stays poisoned through the fp16 conversion).
__host__ __device__ static inline void vcg_scale_of(uint32_t amax_bits, int shift, float& s, float& inv) {
  int e = (int)((amax_bits >> 23) & 0xFF);
  int f = (e == 0 || e == 255) ? 127 + shift : e - 14 + shift;
  f = f < 1 ? 1 : (f > 253 ? 253 : f);
  const uint32_t sb = (uint32_t)f << 23, ib = (uint32_t)(254 - f) << 23;
#ifdef __HIP_DEVICE_COMPILE__
  s = __uint_as_float(sb); inv = __uint_as_float(ib);
#else
  memcpy(&s, &sb, 4); memcpy(&inv, &ib, 4);
#endif
}

// InstanceNorm reductions (norm.hip): a workgroup sums one chunk of pixels for TC channel quads x TP pixel lanes; the chunk
// partials ([N][nchunk][C][2] doubles) are combined in double by k_in_final.  Shared with the conv epilogues that emit
// those partials themselves (vcg_conv_fwd_in).
struct NormPlan {
  int TC, TP, cgroups, nchunk, chunk;
};
bool vcg_in_tail_enabled();     // VCG_IN_TAIL=1: finalize in the producer (VcgInTail below); default 0 — measured slower, see there
static inline NormPlan vcg_norm_plan(int N, int HW, int C) {
  NormPlan pl;
  int c4 = C / 4;
  int tc = 1;
  // with VcgInTail on, at most 128 channels per workgroup: the workgroup that finalizes an (image, channel group) then has at
  // least two chunk lanes per channel and a short chain of partials to walk
  // at most 128 channels per workgroup (rounds 1-2: up to 1024): more, narrower workgroups — measured on the step's InstanceNorm
  // backward reductions 3.5 -> 3.3 ms per step — and what the finalizing workgroup of VcgInTail needs (>= 2 chunk lanes per channel)
  while (tc * 2 <= c4 && tc * 2 <= 32) tc *= 2;
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

// "Finalize in the producer" (round 3): the kernel that writes the chunk partials of an InstanceNorm reduction also combines
// them.  Every workgroup that has written its chunk for (image n, channel range r) bumps counter [n * ranges + r]; the one
// that arrives last — whichever it is — sums all `nchunk` partials of that range in the FIXED order k_in_final used to
// (chunk lanes, then lanes in order: bitwise reproducible) and writes mean / rstd (MODE 0) or the two backward sums (MODE 1).
// The separate 5-8 us finalize launch (134 per CycleVAEGAN step) and its dispatch gap are gone.  Counters come from a ring of
// zero-initialised words in the code object; atomicInc wraps to zero on the last arrival, so a counter cleans itself.
// MEASURED AND NOT ADOPTED (VERDICT r2 #5b; VCG_IN_TAIL=1 turns it on, every parity test passes with it): same box, 20 steps,
// twice each: 38.61 / 38.33 ms per step with the tails, 37.83 / 37.90 with the separate finalize launches.  The finalize kernel
// spreads N C / 32 blocks over the chip and is launch-latency bound (7 us); the last arriver walks nchunk x 128 channels of
// write-through partials alone, at L2-miss latency, while the rest of the chip idles — longer than the launch it replaces.
struct VcgInTail {
  float* out1;            // mean (MODE 0) or s12 (MODE 1); null: no tail, the caller finalizes with vcg_in_finalize
  float* out2;            // rstd (MODE 0)
  uint32_t* counters;     // [N * ranges]
  int HW;                 // pixels per image the partials cover
  float eps;
};
static inline VcgInTail vcg_in_tail_none() { VcgInTail t; t.out1 = nullptr; t.out2 = nullptr; t.counters = nullptr; t.HW = 0; t.eps = 0.f; return t; }
uint32_t* vcg_tail_counters(int count);                 // `count` zero words nobody else is using (misc.hip)
static inline VcgInTail vcg_in_tail_make(float* out1, float* out2, int ncounters, int HW, float eps) {
  VcgInTail t = vcg_in_tail_none();
  if (!vcg_in_tail_enabled()) return t;
  t.counters = vcg_tail_counters(ncounters);
  if (!t.counters) return t;
  t.out1 = out1; t.out2 = out2; t.HW = HW; t.eps = eps;
  return t;
}
#ifdef __HIPCC__
// Visibility between workgroups inside one launch (cdna_hip_programming.md, Guideline 16): per-CU L1s are never refreshed by other
// CUs' stores and the eight per-XCD L2s are not coherent with each other, so the partials are stored WRITE-THROUGH (sc1: relaxed
// agent-scope atomic stores), every storing wave drains its stores, the block's barrier, ONE lane's relaxed agent-scope counter
// add; the last arriver reads them back with sc1 loads.  No __threadfence(): an agent-scope release in every block writes the
// XCD's whole L2 back (measured here: the step went from 38 to 59 ms).
typedef __attribute__((address_space(1))) unsigned long long vcg_gu64;
__device__ __forceinline__ void vcg_store_sc1(double* p, double v) {
  __hip_atomic_store((vcg_gu64*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vcg_store_sc1_f2(float* p, float x, float y) {              // p 8-byte aligned
  const unsigned long long v = ((unsigned long long)__float_as_uint(y) << 32) | __float_as_uint(x);
  __hip_atomic_store((vcg_gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double vcg_load_sc1(const double* p) {
  return __builtin_bit_cast(double, __hip_atomic_load((vcg_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float2 vcg_load_sc1_f2(const float* p) {
  const unsigned long long v = __hip_atomic_load((vcg_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
}
// all threads of the block, after their sc1 stores: true in the block that arrives last at `ctr` (of `expected`).
// `flag`: one int of LDS that nothing else is using.
__device__ __forceinline__ bool vcg_last_arrival(uint32_t* ctr, uint32_t expected, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: its write-through stores have landed
  __syncthreads();
  if (threadIdx.x == 0) *flag = atomicInc(ctr, expected - 1) == expected - 1;      // agent scope; wraps to zero on the last arrival
  __syncthreads();
  return *flag != 0;
}
// call with ALL threads of a 256-thread block, after the block's partials part[((n * nchunk + chunk) * C + c) * 2 + {0, 1}]
// have been stored with vcg_store_sc1.  [c0, c0 + cn) = the channel range this block contributed to (cn a power of two), `ctr`
// its counter, `expected` = blocks per (n, range) = nchunk.  sh: 512 doubles + one int of LDS free for the taking.
template <int MODE>
__device__ __forceinline__ void vcg_in_tail_run(const VcgInTail& t, const double* part, int n, int c0, int cn, int C, int nchunk,
                                                uint32_t* ctr, uint32_t expected, double* sh) {
  if (!vcg_last_arrival(ctr, expected, reinterpret_cast<int*>(sh + 512))) return;
  const int cnl = cn < 256 ? cn : 256, L = 256 / cnl;
  const int cl = threadIdx.x % cnl, lane = threadIdx.x / cnl;
  for (int cb = 0; cb < cn; cb += cnl) {
    const int c = c0 + cb + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
      // four chunk loads in flight per thread (the loop is a chain of L2 round trips otherwise)
      double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
      const double* q = part + ((size_t)n * nchunk * C + c) * 2;
      const size_t ks = (size_t)C * 2;
      int k = lane;
      for (; k + 3 * L < nchunk; k += 4 * L) {
        const double* q0 = q + (size_t)k * ks;
        const double* q1 = q0 + (size_t)L * ks;
        const double* q2 = q1 + (size_t)L * ks;
        const double* q3 = q2 + (size_t)L * ks;
        const double x0 = vcg_load_sc1(q0), y0 = vcg_load_sc1(q0 + 1), x1 = vcg_load_sc1(q1), y1 = vcg_load_sc1(q1 + 1);
        const double x2 = vcg_load_sc1(q2), y2 = vcg_load_sc1(q2 + 1), x3 = vcg_load_sc1(q3), y3 = vcg_load_sc1(q3 + 1);
        a += x0; b += y0; a1 += x1; b1 += y1; a2 += x2; b2 += y2; a3 += x3; b3 += y3;
      }
      for (; k < nchunk; k += L) { a += vcg_load_sc1(q + (size_t)k * ks); b += vcg_load_sc1(q + (size_t)k * ks + 1); }
      a = (a + a1) + (a2 + a3);
      b = (b + b1) + (b2 + b3);
    }
    if (L > 1) {
      sh[threadIdx.x] = a;
      sh[256 + threadIdx.x] = b;
      __syncthreads();
      if (lane == 0)
        for (int k = 1; k < L; ++k) { a += sh[k * cnl + cl]; b += sh[256 + k * cnl + cl]; }
    }
    if (lane == 0 && c < C) {
      const size_t idx = (size_t)n * C + c;
      if (MODE == 0) {
        const double m = a / t.HW;
        double var = b / t.HW - m * m;
        if (var < 0.0) var = 0.0;
        t.out1[idx] = (float)m;
        t.out2[idx] = (float)(1.0 / sqrt(var + (double)t.eps));
      } else {
        t.out1[idx * 2] = (float)(a / t.HW);
        t.out1[idx * 2 + 1] = (float)(b / t.HW);
      }
    }
  }
}
#endif
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

// conv_thinin.hip: forward of the layers with a 4-channel input and 64 outputs (the 7x7 stem, the discriminators' first layer)
bool vcg_thinin_fwd_ok(const ConvGeom& g);
int vcg_thinin_nchunk(const ConvGeom& g);
int vcg_thinin_fwd(const ConvGeom& g, const float* x, const float* wf, const void* w_amax, const float* bias, float* y, double* in_part,
                   hipStream_t st, uint64_t x_handle);

// conv_wino.hip: Winograd F(2x2,3x3) forward for the 3x3 / stride-1 / pad-1 layers
bool vcg_wino_weight_ok(const ConvGeom& g);
long long vcg_wino_gate_fwd();
long long vcg_wino_gate_dgrad();
long long vcg_wino_gate_wgrad();
bool vcg_wino_fwd_ok(const ConvGeom& g);
size_t vcg_wino_weight_floats(const ConvGeom& g);
size_t vcg_wino_fwd_workspace(const ConvGeom& g);
int vcg_wino_weight(const ConvGeom& g, const float* w_oihw, float* u, const VcgAmax& amax_w, hipStream_t st);
// deferred InstanceNorm of a conv's INPUT (vcg_conv_fwd_in_pre): mean / rstd [N][Cin] and the activation that follows the norm
struct VcgPre { const float* mean; const float* rstd; int act; };
bool vcg_wino_pre_ok(const ConvGeom& g);
int vcg_wino_fwd(const ConvGeom& g, const float* x, const float* u, const void* w_amax, const float* bias, float* y, void* ws,
                 size_t ws_bytes, hipStream_t st, double* in_part = nullptr, const VcgInTail* tail = nullptr, float* v_keep = nullptr,
                 uint64_t x_handle = 0, const VcgPre* pre = nullptr);
size_t vcg_wino_saved_floats(const ConvGeom& g);
size_t vcg_wino_fwd_stats_doubles(const ConvGeom& g);
// Winograd weight gradient: transforms in conv_wino.hip, batched stream-K reduction + back-transform in conv_igemm.hip
bool vcg_wino_wgrad_ok(const ConvGeom& g);
size_t vcg_wino_wgrad_workspace(const ConvGeom& g);
int vcg_wino_wgrad(const ConvGeom& g, const float* x, const float* dy, float* gw_oihw, void* ws, size_t ws_bytes, hipStream_t st,
                   const float* v_saved = nullptr, uint64_t x_handle = 0, uint64_t dy_handle = 0);
bool vcg_wino_wgrad_tr_ok(const ConvGeom& g);
int vcg_wino_wgrad_reduce_one(const ConvGeom& g, const float* dU, float* gw_oihw, hipStream_t st);   // dU [16][Kc][Cout], one part
size_t vcg_wino_wgrad_core_workspace(const ConvGeom& g, int T);
int vcg_wino_wgrad_core(const ConvGeom& g, const float* V, const float* dM, int T, float* gw_oihw, void* ws, size_t ws_bytes,
                        hipStream_t st, const VcgAmax& amax_v, const VcgAmax& amax_dm, bool v_planes);
bool vcg_wino_dgrad_ok(const ConvGeom& g);
size_t vcg_wino_dgrad_workspace(const ConvGeom& g);
int vcg_wino_weight_dgrad(const ConvGeom& g, const float* w_oihw, float* ud, const VcgAmax& amax_w, hipStream_t st);
int vcg_wino_dgrad(const ConvGeom& g, const float* dy, const float* ud, const void* w_amax, float* dx, void* ws, size_t ws_bytes,
                   hipStream_t st, uint64_t dy_handle = 0);
// conv_slab.hip: 3x3 / stride-1 layers with few channels on large maps — the input staged once per workgroup as an LDS slab
bool vcg_slab_fwd_ok(const ConvGeom& g);
bool vcg_slab_fwd_stats_ok(const ConvGeom& g);
int vcg_slab_fwd_nchunk(const ConvGeom& g);
bool vcg_slab_dgrad_ok(const ConvGeom& g);
size_t vcg_slab_dgrad_workspace(const ConvGeom& g);
int vcg_slab_fwd(const ConvGeom& g, const float* x, const void* wft_planes, size_t planes_bytes, const void* w_amax, const float* bias,
                 float* y, double* in_part, const VcgInTail* tail, hipStream_t st, uint64_t x_handle = 0);
int vcg_slab_dgrad(const ConvGeom& g, const float* dy, const void* wfd_planes, size_t planes_bytes, const void* w_amax, float* dx,
                   void* ws, size_t ws_bytes, hipStream_t st, uint64_t dy_handle = 0);
// conv_ring.hip: weight gradients of U4 / head / stem with row-ring staging (the taps are address offsets of the fragment reads)
bool vcg_ring_wgrad_ok(const ConvGeom& g);
size_t vcg_ring_wgrad_workspace(const ConvGeom& g);
int vcg_ring_wgrad(const ConvGeom& g, const float* x, const float* dy, float* gw_oihw, void* ws, size_t ws_bytes, hipStream_t st,
                   uint64_t x_handle = 0, uint64_t dy_handle = 0);
// conv_thin.hip: thin forward with kw folded into the GEMM's N (MFMA)
bool vcg_thin_fold_ok(const ConvGeom& g);
size_t vcg_thin_fold_weight_floats(const ConvGeom& g);
size_t vcg_thin_fold_workspace(const ConvGeom& g);
int vcg_thin_fold_pack(const ConvGeom& g, const float* w_oihw, float* wk, const VcgAmax& amax_w, hipStream_t st);
int vcg_thin_fold_fwd(const ConvGeom& g, const float* x, const float* wk, const void* w_amax, const float* bias, float* y, void* ws,
                      size_t ws_bytes, hipStream_t st, uint64_t x_handle = 0);
bool vcg_thin_fold_dgrad_ok(const ConvGeom& g);
size_t vcg_thin_fold_dgrad_weight_floats(const ConvGeom& g);
size_t vcg_thin_fold_dgrad_workspace(const ConvGeom& g);
int vcg_thin_fold_dgrad_pack(const ConvGeom& g, const float* w_oihw, float* wk, const VcgAmax& amax_w, hipStream_t st);
int vcg_thin_fold_dgrad(const ConvGeom& g, const float* dy, const float* wkd, const void* w_amax, float* dx, void* ws, size_t ws_bytes,
                        hipStream_t st, uint64_t dy_handle = 0);
