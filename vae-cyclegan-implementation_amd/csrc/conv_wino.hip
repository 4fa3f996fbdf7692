// Winograd F(2x2, 3x3) for the 3x3 / stride-1 / pad-1 convolutions of the D, R and U blocks
// (Networks.py:87-136): 2.25x fewer multiplications than the direct form, whichever pipe multiplies.
//
//   V[xi][t][k]  = (B^T d B)[xi]       input transform of the 4x4 patch of output tile t (2x2 outputs), k = (i, j, c)
//   U[xi][co][k] = (G g G^T)[xi]       weight transform, once per optimizer step (vcg_pack_weight)
//   M[xi][t][co] = sum_k V U           16 independent GEMMs: one launch of the split-operand GEMM (gemm_split.hip)
//   y            = A^T M A + bias, activation
//
// Padding (reflect or zero) and the folded PixelUnshuffle of the D blocks live in the input transform's gather, so
// the GEMMs are dense.  V and M are 4x the activation they come from; they stay in the caller's workspace.
// Rounding: the transforms add a few ulp to the fp32 sums (measured in tests/test_gpu_parity.py against the
// oracle at the same 1e-4 bound as the direct kernels).
#include "vcg_common.h"
#include <stdlib.h>

int vcg_gemm_split_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, const VcgAmax& amax_a,
                           const VcgAmax& amax_b, hipStream_t st, uint32_t* amax_a_keep = nullptr);
// |G g G^T| <= (1.5)^2 max|g|: the transformed kernels are bounded by 4 x the kernel's largest magnitude
#define WINO_U_SHIFT 2
// |B^T d B| <= 4 max|d| (two +-1 pairs): the transformed input is bounded by 4 x the input's largest magnitude
#define WINO_V_SHIFT 2
int vcg_gemm_planes_batched(const void* APlanes, const void* BtPlanes, float* C, int rows, int K, int N, int batches, const VcgAmax& amax_a,
                            const VcgAmax& amax_b, hipStream_t st, uint32_t* amax_a_keep);
// VCG_WINO_PLANES=0: V as fp32, split inside the GEMM (A/B measurements).  Default: k_wino_in writes V pre-split — the same bytes —
// scaled by 4 x the input's amax, and the GEMMs (forward, data gradient, and the weight gradient that re-reads a kept V) stage it
// with plain copies.
// Gates of the three directions on Kc Cout / (Kc + Cout) (what the GEMMs save per float the transforms move); VCG_WINO_GATE_F / _D / _W
// override them for A/B measurements (tools/conv_bench.py).  Round 3 (three fp16 MFMAs per product instead of six bf16 ones: the
// direct kernels' matrix time halved, the transforms' traffic did not): re-measured per layer at batch 8 — forward D1 (85) 458 us
// Winograd vs 360 direct, U2 (85) 109 vs 84, U1 / D2 (171) 79 / 267 vs 93 / 293: the forward gate moved from 64 to 100; data
// gradient D1 476 vs 521, U2 118 vs 139: stays at 80; weight gradient D1 / U2 (85): the ring kernel as before
// (profiles/r03_wino_gates.txt)
static long long wino_gate_env(const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; }
long long vcg_wino_gate_fwd() { static const long long v = wino_gate_env("VCG_WINO_GATE_F", 100); return v; }
long long vcg_wino_gate_dgrad() { static const long long v = wino_gate_env("VCG_WINO_GATE_D", 80); return v; }
long long vcg_wino_gate_wgrad() { static const long long v = wino_gate_env("VCG_WINO_GATE_W", 128); return v; }
static bool wino_planes_on() {
  static const int on = [] { const char* e = getenv("VCG_WINO_PLANES"); return e ? atoi(e) : 1; }();
  return on != 0;
}

struct WinoP {
  const float* x;
  float* v;
  const float* m;
  const float* bias;
  float* y;
  int N, H, W, Cin, Cout, Hl, Wl, ups, reflect, act, cout_log;
  int th, tw, T, Kc;
  int off;     // patch origin = 2 * tile - off: 1 (pad 1) forward, 2 for the data gradient over the padded domain
  FastDiv fd_k4, fd_tw, fd_thtw, fd_c4, fd_co4;
  unsigned long long* amax_slot;   // k_wino_in / k_wino_dy: where the largest magnitude of what they write goes (vcg_common.h), or null
  uint32_t amax_gen;
  // k_wino_in, planes mode (vplanes != null): V is written already split — fp16 planes [xi][t][Kc / 32][2][32] of V / s, the A
  // operand of the GEMM as a pure copy — with s from the INPUT's amax: |B^T d B| <= 4 max|d|, so its scale is known before V is
  VcgAmax amax_x;
  unsigned short* vplanes;
  // k_wino_in_planes / k_wino_in_tr, deferred InstanceNorm (vcg_conv_fwd_in_pre): x is the RAW output t of the previous conv and the
  // gather normalises it on the way in — pre_act((t - mean[n][c]) * rstd[n][c]) — so the normalised tensor is never written
  const float* pre_mean;
  const float* pre_rstd;
  int pre_act;
};
__device__ __forceinline__ float4 wino_pre_apply(const float4& v, const float4& mu, const float4& rs, int act) {
  return make_float4(act_apply((v.x - mu.x) * rs.x, act), act_apply((v.y - mu.y) * rs.y, act), act_apply((v.z - mu.z) * rs.z, act),
                     act_apply((v.w - mu.w) * rs.w, act));
}

__device__ __forceinline__ float4 f4sub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4sum(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// one thread: one tile x 4 consecutive k (same unshuffle phase (i, j), channels c..c+3)
__global__ __launch_bounds__(256) void k_wino_in(WinoP p) {
  __shared__ uint32_t amax_red[4];
  uint32_t amax = 0;
  const uint32_t k4n = (uint32_t)p.Kc / 4;
  const size_t total = (size_t)p.T * k4n;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(idx / k4n);
    const uint32_t k4 = (uint32_t)(idx - (size_t)t * k4n);
    const uint32_t ph = fd_div(k4, p.fd_c4);                      // unshuffle phase i*2 + j (0 when ups == 1)
    const int c = (int)(k4 - ph * (uint32_t)(p.Cin / 4)) * 4;
    const int pi = (int)(ph >> 1), pj = (int)(ph & 1);
    const uint32_t n = fd_div(t, p.fd_thtw);
    const uint32_t rem = t - n * (uint32_t)(p.th * p.tw);
    const uint32_t ty = fd_div(rem, p.fd_tw);
    const int tx = (int)(rem - ty * (uint32_t)p.tw);
    const float* xn = p.x + (size_t)n * p.H * p.W * p.Cin;
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int ih = 2 * (int)ty - p.off + r;
      bool okh = true;
      if (p.reflect) ih = reflect_idx(ih, p.Hl);
      else okh = ih >= 0 && ih < p.Hl;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        int iw = 2 * tx - p.off + s;
        bool ok = okh;
        if (p.reflect) iw = reflect_idx(iw, p.Wl);
        else ok = ok && iw >= 0 && iw < p.Wl;
        d[r][s] = ok ? *reinterpret_cast<const float4*>(xn + ((size_t)(ih * p.ups + pi) * p.W + (iw * p.ups + pj)) * p.Cin + c)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3); then the same on the columns
    float4 e[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      e[0][s] = f4sub(d[0][s], d[2][s]);
      e[1][s] = f4sum(d[1][s], d[2][s]);
      e[2][s] = f4sub(d[2][s], d[1][s]);
      e[3][s] = f4sub(d[1][s], d[3][s]);
    }
    float* vb = p.v + (size_t)t * p.Kc + (size_t)k4 * 4;
    const size_t plane = (size_t)p.T * p.Kc;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 o0 = f4sub(e[a][0], e[a][2]), o1 = f4sum(e[a][1], e[a][2]), o2 = f4sub(e[a][2], e[a][1]), o3 = f4sub(e[a][1], e[a][3]);
      *reinterpret_cast<float4*>(vb + (size_t)(a * 4 + 0) * plane) = o0;
      *reinterpret_cast<float4*>(vb + (size_t)(a * 4 + 1) * plane) = o1;
      *reinterpret_cast<float4*>(vb + (size_t)(a * 4 + 2) * plane) = o2;
      *reinterpret_cast<float4*>(vb + (size_t)(a * 4 + 3) * plane) = o3;
      const uint32_t m01 = max(vcg_abs_bits4(o0), vcg_abs_bits4(o1)), m23 = max(vcg_abs_bits4(o2), vcg_abs_bits4(o3));
      amax = max(amax, max(m01, m23));
    }
  }
  if (p.amax_slot) vcg_amax_publish(amax, p.amax_slot, p.amax_gen, amax_red);     // uniform: the GEMM that reads V scales by it
}
// the same transform, V written as pre-split planes (WinoP::vplanes)
__global__ __launch_bounds__(256) void k_wino_in_planes(WinoP p) {
  float sc, inv;
  vcg_scale_of(vcg_amax_bits(p.amax_x), p.amax_x.shift, sc, inv);
  const uint32_t k4n = (uint32_t)p.Kc / 4;
  const size_t total = (size_t)p.T * k4n;
  const size_t KB = (size_t)p.Kc / 32;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(idx / k4n);
    const uint32_t k4 = (uint32_t)(idx - (size_t)t * k4n);
    const uint32_t ph = fd_div(k4, p.fd_c4);
    const int c = (int)(k4 - ph * (uint32_t)(p.Cin / 4)) * 4;
    const int pi = (int)(ph >> 1), pj = (int)(ph & 1);
    const uint32_t n = fd_div(t, p.fd_thtw);
    const uint32_t rem = t - n * (uint32_t)(p.th * p.tw);
    const uint32_t ty = fd_div(rem, p.fd_tw);
    const int tx = (int)(rem - ty * (uint32_t)p.tw);
    const float* xn = p.x + (size_t)n * p.H * p.W * p.Cin;
    float4 pmu = make_float4(0.f, 0.f, 0.f, 0.f), prs = pmu;
    if (p.pre_mean) {
      pmu = *reinterpret_cast<const float4*>(p.pre_mean + (size_t)n * p.Cin + c);
      prs = *reinterpret_cast<const float4*>(p.pre_rstd + (size_t)n * p.Cin + c);
    }
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int ih = 2 * (int)ty - p.off + r;
      bool okh = true;
      if (p.reflect) ih = reflect_idx(ih, p.Hl);
      else okh = ih >= 0 && ih < p.Hl;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        int iw = 2 * tx - p.off + s;
        bool ok = okh;
        if (p.reflect) iw = reflect_idx(iw, p.Wl);
        else ok = ok && iw >= 0 && iw < p.Wl;
        d[r][s] = ok ? *reinterpret_cast<const float4*>(xn + ((size_t)(ih * p.ups + pi) * p.W + (iw * p.ups + pj)) * p.Cin + c)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.pre_mean && ok) d[r][s] = wino_pre_apply(d[r][s], pmu, prs, p.pre_act);
      }
    }
    float4 e[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      e[0][s] = f4sub(d[0][s], d[2][s]);
      e[1][s] = f4sum(d[1][s], d[2][s]);
      e[2][s] = f4sub(d[2][s], d[1][s]);
      e[3][s] = f4sub(d[1][s], d[3][s]);
    }
    const uint32_t k = k4 * 4;
    unsigned short* vb = p.vplanes + ((size_t)t * KB + k / 32) * VCG_PBLK + (k & 31);
    const size_t plane = (size_t)p.T * KB * VCG_PBLK;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 o[4] = {f4sub(e[a][0], e[a][2]), f4sum(e[a][1], e[a][2]), f4sub(e[a][2], e[a][1]), f4sub(e[a][1], e[a][3])};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint2 h, l;
        split4h(o[b], inv, h, l);
        unsigned short* q = vb + (size_t)(a * 4 + b) * plane;
        *reinterpret_cast<uint2*>(q) = h;
        *reinterpret_cast<uint2*>(q + 32) = l;
      }
    }
  }
}

// one thread: one tile x 4 output channels; y = A^T m A, A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
__global__ __launch_bounds__(256) void k_wino_out(WinoP p) {
  const uint32_t c4n = (uint32_t)p.Cout / 4;
  const size_t total = (size_t)p.T * c4n;
  const size_t plane = (size_t)p.T * p.Cout;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(idx / c4n);
    const int co = (int)(idx - (size_t)t * c4n) * 4;
    const uint32_t n = fd_div(t, p.fd_thtw);
    const uint32_t rem = t - n * (uint32_t)(p.th * p.tw);
    const uint32_t ty = fd_div(rem, p.fd_tw);
    const int tx = (int)(rem - ty * (uint32_t)p.tw);
    const float* mb = p.m + (size_t)t * p.Cout + co;
    float4 s0[4], s1[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float4 m0 = *reinterpret_cast<const float4*>(mb + (size_t)(0 + b) * plane);
      const float4 m1 = *reinterpret_cast<const float4*>(mb + (size_t)(4 + b) * plane);
      const float4 m2 = *reinterpret_cast<const float4*>(mb + (size_t)(8 + b) * plane);
      const float4 m3 = *reinterpret_cast<const float4*>(mb + (size_t)(12 + b) * plane);
      s0[b] = f4sum(f4sum(m0, m1), m2);
      s1[b] = f4sub(f4sub(m1, m2), m3);
    }
    float4 y[2][2];
    y[0][0] = f4sum(f4sum(s0[0], s0[1]), s0[2]);
    y[0][1] = f4sub(f4sub(s0[1], s0[2]), s0[3]);
    y[1][0] = f4sum(f4sum(s1[0], s1[1]), s1[2]);
    y[1][1] = f4sub(f4sub(s1[1], s1[2]), s1[3]);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
      if (co + 0 < p.cout_log) bv.x = p.bias[co + 0];
      if (co + 1 < p.cout_log) bv.y = p.bias[co + 1];
      if (co + 2 < p.cout_log) bv.z = p.bias[co + 2];
      if (co + 3 < p.cout_log) bv.w = p.bias[co + 3];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int oh = 2 * (int)ty + r, ow = 2 * tx + s;
        if (oh < p.Hl && ow < p.Wl) {
          float4 o = y[r][s];
          o.x = act_apply(o.x + bv.x, p.act); o.y = act_apply(o.y + bv.y, p.act);
          o.z = act_apply(o.z + bv.z, p.act); o.w = act_apply(o.w + bv.w, p.act);
          *reinterpret_cast<float4*>(p.y + (((size_t)n * p.Hl + oh) * p.Wl + ow) * p.Cout + co) = o;
        }
      }
  }
}

// The same output transform for a layer whose output goes into an InstanceNorm: it also leaves the statistics' chunk
// partials (sum and sum of squares per (image, channel), in double: norm.hip) so that no extra pass has to re-read y.
// Grid (chunk of tiles, image, channel-quad group); thread = (channel quad, tile lane), as in k_in_partial.
__global__ __launch_bounds__(256) void k_wino_out_stats(WinoP p, double* __restrict__ part, NormPlan pl, VcgInTail tail) {
  __shared__ double r1[256 * 4];
  __shared__ double r2[256 * 4];
  const int tc = threadIdx.x % pl.TC, tp = threadIdx.x / pl.TC;
  const int c4 = blockIdx.z * pl.TC + tc;
  const int n = blockIdx.y;
  const int tiles = p.th * p.tw;
  const int pb = blockIdx.x * pl.chunk;
  int pe = pb + pl.chunk;
  if (pe > tiles) pe = tiles;
  const size_t plane = (size_t)p.T * p.Cout;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (c4 * 4 < p.Cout) {
    const int co = c4 * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
      if (co + 0 < p.cout_log) bv.x = p.bias[co + 0];
      if (co + 1 < p.cout_log) bv.y = p.bias[co + 1];
      if (co + 2 < p.cout_log) bv.z = p.bias[co + 2];
      if (co + 3 < p.cout_log) bv.w = p.bias[co + 3];
    }
    for (int tl = pb + tp; tl < pe; tl += pl.TP) {
      const int ty = tl / p.tw, tx = tl - ty * p.tw;
      const size_t t = (size_t)n * tiles + tl;
      const float* mb = p.m + t * p.Cout + co;
      float4 a0[4], a1[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float4 m0 = *reinterpret_cast<const float4*>(mb + (size_t)(0 + b) * plane);
        const float4 m1 = *reinterpret_cast<const float4*>(mb + (size_t)(4 + b) * plane);
        const float4 m2 = *reinterpret_cast<const float4*>(mb + (size_t)(8 + b) * plane);
        const float4 m3 = *reinterpret_cast<const float4*>(mb + (size_t)(12 + b) * plane);
        a0[b] = f4sum(f4sum(m0, m1), m2);
        a1[b] = f4sub(f4sub(m1, m2), m3);
      }
      float4 y[2][2];
      y[0][0] = f4sum(f4sum(a0[0], a0[1]), a0[2]);
      y[0][1] = f4sub(f4sub(a0[1], a0[2]), a0[3]);
      y[1][0] = f4sum(f4sum(a1[0], a1[1]), a1[2]);
      y[1][1] = f4sub(f4sub(a1[1], a1[2]), a1[3]);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float4 o = y[r][q];
          o.x = act_apply(o.x + bv.x, p.act); o.y = act_apply(o.y + bv.y, p.act);
          o.z = act_apply(o.z + bv.z, p.act); o.w = act_apply(o.w + bv.w, p.act);
          *reinterpret_cast<float4*>(p.y + (((size_t)n * p.Hl + 2 * ty + r) * p.Wl + 2 * tx + q) * p.Cout + co) = o;
          s1[0] += (double)o.x; s2[0] += (double)o.x * (double)o.x;
          s1[1] += (double)o.y; s2[1] += (double)o.y * (double)o.y;
          s1[2] += (double)o.z; s2[2] += (double)o.z * (double)o.z;
          s1[3] += (double)o.w; s2[3] += (double)o.w * (double)o.w;
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r1[threadIdx.x * 4 + e] = s1[e];
    r2[threadIdx.x * 4 + e] = s2[e];
  }
  __syncthreads();
  if (tp == 0 && c4 * 4 < p.Cout) {
    for (int k = 1; k < pl.TP; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[e] += r1[(k * pl.TC + tc) * 4 + e];
        s2[e] += r2[(k * pl.TC + tc) * 4 + e];
      }
    }
    double* o = part + (((size_t)n * pl.nchunk + blockIdx.x) * p.Cout + c4 * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      vcg_store_sc1(o + 2 * e, s1[e]);
      vcg_store_sc1(o + 2 * e + 1, s2[e]);
    }
  }
  if (tail.out1) {                    // the last chunk block of this (image, channel group) writes mean / rstd (vcg_common.h)
    __syncthreads();                  // r1 / r2 are free again
    vcg_in_tail_run<0>(tail, part, n, blockIdx.z * pl.TC * 4, pl.TC * 4, p.Cout, pl.nchunk, tail.counters + n * pl.cgroups + blockIdx.z,
                       (uint32_t)pl.nchunk, r1);
  }
}

// The transformed kernels (G g G^T)[xi], G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], written as the B^T operand of
// the batched GEMM, already split into fp16 "blocked planes" (gemm_split.hip): for GEMM row n and reduction index kk,
//     up[((xi * NR + n) * KD/32 + kk/32) * VCG_PBLK + piece * 32 + kk % 32]      (fp16 pieces of the value / s, s from the kernel's amax)
// DGRAD = false: the forward GEMM M = V . U^T — n = co (NR = Cout), kk = k = (phase, c) (KD = Kc);
// DGRAD = true:  the data-gradient GEMM over the padded domain — n = k (NR = Kc), kk = co (KD = Cout), kernel flipped:
//                Ud = transform of w[co][k][2 - a][2 - b].
// One thread per (n, 4 consecutive kk): 36-byte OIHW reads, 8-byte stores per piece and transform point.
template <bool DGRAD>
__global__ __launch_bounds__(256) void k_wino_weight_planes(const float* __restrict__ w, unsigned short* __restrict__ up, int Cin,
                                                            int Cout, int ups, int cin_log, int cout_log, VcgAmax amax) {
  float sc, inv;
  vcg_scale_of(vcg_amax_bits(amax), amax.shift, sc, inv);
  const int U2 = ups * ups, Kc = U2 * Cin;
  const int NR = DGRAD ? Kc : Cout, KD = DGRAD ? Cout : Kc;
  const size_t total = (size_t)NR * (KD / 4);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx / (KD / 4)), kk = (int)(idx - (size_t)n * (KD / 4)) * 4;
    float t[16][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = DGRAD ? kk + e : n, k = DGRAD ? n : kk + e;
      const int ph = k / Cin, c = k - ph * Cin;
      const bool ok = co < cout_log && c < cin_log;
      const float* wp = w + ((size_t)co * (cin_log * U2) + (size_t)c * U2 + ph) * 9;
      float g[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = ok ? (DGRAD ? wp[(2 - a) * 3 + (2 - b)] : wp[a * 3 + b]) : 0.f;
      float h[4][3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        h[0][b] = g[0][b];
        h[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        h[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        h[3][b] = g[2][b];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        t[a * 4 + 0][e] = h[a][0];
        t[a * 4 + 1][e] = 0.5f * (h[a][0] + h[a][1] + h[a][2]);
        t[a * 4 + 2][e] = 0.5f * (h[a][0] - h[a][1] + h[a][2]);
        t[a * 4 + 3][e] = h[a][2];
      }
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      uint2 hh, ll;
      split4h(make_float4(t[xi][0], t[xi][1], t[xi][2], t[xi][3]), inv, hh, ll);
      unsigned short* o = up + (((size_t)xi * NR + n) * (KD / 32) + kk / 32) * VCG_PBLK + (kk & 31);
      *reinterpret_cast<uint2*>(o) = hh;
      *reinterpret_cast<uint2*>(o + 32) = ll;
    }
  }
}

// output transform of the data-gradient GEMMs: one thread = one padded-domain tile x 4 k; writes dxp[n][qh][qw][k]
__global__ __launch_bounds__(256) void k_wino_out_pad(const float* __restrict__ m, float* __restrict__ dxp, WinoP p) {
  const uint32_t k4n = (uint32_t)p.Kc / 4;
  const size_t total = (size_t)p.T * k4n;
  const size_t plane = (size_t)p.T * p.Kc;
  const int Hp = 2 * p.th, Wp = 2 * p.tw;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(idx / k4n);
    const int k = (int)(idx - (size_t)t * k4n) * 4;
    const uint32_t n = fd_div(t, p.fd_thtw);
    const uint32_t rem = t - n * (uint32_t)(p.th * p.tw);
    const uint32_t ty = fd_div(rem, p.fd_tw);
    const int tx = (int)(rem - ty * (uint32_t)p.tw);
    const float* mb = m + (size_t)t * p.Kc + k;
    float4 s0[4], s1[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float4 m0 = *reinterpret_cast<const float4*>(mb + (size_t)(0 + b) * plane);
      const float4 m1 = *reinterpret_cast<const float4*>(mb + (size_t)(4 + b) * plane);
      const float4 m2 = *reinterpret_cast<const float4*>(mb + (size_t)(8 + b) * plane);
      const float4 m3 = *reinterpret_cast<const float4*>(mb + (size_t)(12 + b) * plane);
      s0[b] = f4sum(f4sum(m0, m1), m2);
      s1[b] = f4sub(f4sub(m1, m2), m3);
    }
    float* o = dxp + (((size_t)n * Hp + 2 * ty) * Wp + 2 * tx) * p.Kc + k;
    *reinterpret_cast<float4*>(o) = f4sum(f4sum(s0[0], s0[1]), s0[2]);
    *reinterpret_cast<float4*>(o + p.Kc) = f4sub(f4sub(s0[1], s0[2]), s0[3]);
    *reinterpret_cast<float4*>(o + (size_t)Wp * p.Kc) = f4sum(f4sum(s1[0], s1[1]), s1[2]);
    *reinterpret_cast<float4*>(o + (size_t)Wp * p.Kc + p.Kc) = f4sub(f4sub(s1[1], s1[2]), s1[3]);
  }
}

// dx[n][h*ups+i][w*ups+j][c] = sum over padded coordinates (qh, qw) that the padding maps onto (h, w) of dxp[n][qh][qw][(i,j,c)]
// pad = 1: q = h + 1 always; with reflect padding the halo row q = 0 lands on h = 1 and q = Hl + 1 on h = Hl - 2.
// k_wino_out_pad and k_wino_fold in one pass (the padded-domain image dxp never exists): a thread owns one padded-domain
// tile x 4 columns, transforms it, and writes those of its 2 x 2 outputs that are image pixels straight to dx — adding,
// for the pixels on rows / columns 1 and Hl - 2 (Wl - 2), the halo elements that reflect onto them.  A halo element belongs
// to another tile; it is recomputed here from the 9 (of 16) transform points it depends on (Y[r][c] = sum A[i][r] M[i][j] A[j][c],
// A's columns (1,1,1,0) and (0,1,-1,-1)): border threads read up to 2x (corners 3x) — a quarter of the tiles at 16 x 16 maps,
// 6 % at 64 x 64 — against a write and a re-read of the whole padded image saved.
__device__ __forceinline__ float4 wino_out_elem(const float* __restrict__ mb, size_t plane, int rr, int cc) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int xi_r = rr + i;                               // rows with a non-zero A entry: rr .. rr + 2
    const float sr = (rr == 1 && i > 0) ? -1.f : 1.f;      // column 1 of A = (0, 1, -1, -1)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int xi_c = cc + j;
      const float sc = (cc == 1 && j > 0) ? -1.f : 1.f;
      const float4 v = *reinterpret_cast<const float4*>(mb + (size_t)(xi_r * 4 + xi_c) * plane);
      const float sg = sr * sc;
      acc.x += sg * v.x; acc.y += sg * v.y; acc.z += sg * v.z; acc.w += sg * v.w;
    }
  }
  return acc;
}
__global__ __launch_bounds__(256) void k_wino_out_fold(const float* __restrict__ m, float* __restrict__ dx, WinoP p, int Hl, int Wl,
                                                       int Cin, int ups, int reflect) {
  const uint32_t k4n = (uint32_t)p.Kc / 4;
  const size_t total = (size_t)p.T * k4n;
  const size_t plane = (size_t)p.T * p.Kc;
  const int W = Wl * ups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(idx / k4n);
    const int k = (int)(idx - (size_t)t * k4n) * 4;
    const uint32_t n = fd_div(t, p.fd_thtw);
    const uint32_t rem = t - n * (uint32_t)(p.th * p.tw);
    const int ty = (int)fd_div(rem, p.fd_tw);
    const int tx = (int)(rem - (uint32_t)ty * (uint32_t)p.tw);
    const float* mb = m + (size_t)t * p.Kc + k;
    float4 s0[4], s1[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float4 m0 = *reinterpret_cast<const float4*>(mb + (size_t)(0 + b) * plane);
      const float4 m1 = *reinterpret_cast<const float4*>(mb + (size_t)(4 + b) * plane);
      const float4 m2 = *reinterpret_cast<const float4*>(mb + (size_t)(8 + b) * plane);
      const float4 m3 = *reinterpret_cast<const float4*>(mb + (size_t)(12 + b) * plane);
      s0[b] = f4sum(f4sum(m0, m1), m2);
      s1[b] = f4sub(f4sub(m1, m2), m3);
    }
    float4 o[2][2];
    o[0][0] = f4sum(f4sum(s0[0], s0[1]), s0[2]);
    o[0][1] = f4sub(f4sub(s0[1], s0[2]), s0[3]);
    o[1][0] = f4sum(f4sum(s1[0], s1[1]), s1[2]);
    o[1][1] = f4sub(f4sub(s1[1], s1[2]), s1[3]);
    const int ph = k / Cin, c = k - ph * Cin;
    const int pi = ph >> 1, pj = ph & 1;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int qy = 2 * ty + r, qx = 2 * tx + q;          // padded-domain coordinates
        if (qy < 1 || qy > Hl || qx < 1 || qx > Wl) continue;  // a halo element: folded by the pixel it reflects onto
        const int h = qy - 1, w = qx - 1;
        float4 sum = o[r][q];
        if (reflect) {
          int qh[2], qw[2], nh = 1, nw = 1;
          qh[0] = qy; qw[0] = qx;
          if (h == 1) qh[nh++] = 0;
          if (h == Hl - 2) qh[nh++] = Hl + 1;                // Hl >= 4: h == 1 and h == Hl - 2 are different rows
          if (w == 1) qw[nw++] = 0;
          if (w == Wl - 2) qw[nw++] = Wl + 1;
          for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) {
              if (a == 0 && b == 0) continue;
              const size_t t2 = ((size_t)n * p.th + (qh[a] >> 1)) * p.tw + (qw[b] >> 1);
              const float4 v = wino_out_elem(m + t2 * p.Kc + k, plane, qh[a] & 1, qw[b] & 1);
              sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        }
        *reinterpret_cast<float4*>(dx + (((size_t)n * Hl * ups + h * ups + pi) * W + (w * ups + pj)) * Cin + c) = sum;
      }
  }
}
__global__ __launch_bounds__(256) void k_wino_fold(const float* __restrict__ dxp, float* __restrict__ dx, int N, int Hl, int Wl,
                                                   int Cin, int ups, int reflect) {
  const int U2 = ups * ups, Kc = U2 * Cin, c4n = Cin / 4;
  const int Hp = Hl + 2, Wp = Wl + 2, W = Wl * ups;
  const size_t total = (size_t)N * Hl * Wl * U2 * c4n;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % c4n) * 4;
    size_t r = idx / c4n;
    const int ph = (int)(r % U2); r /= U2;
    const int w = (int)(r % Wl); r /= Wl;
    const int h = (int)(r % Hl);
    const int n = (int)(r / Hl);
    int qh[2], qw[2], nh = 1, nw = 1;
    qh[0] = h + 1; qw[0] = w + 1;
    if (reflect) {
      if (h == 1) qh[nh++] = 0;
      if (h == Hl - 2) qh[nh++] = Hl + 1;        // Hl >= 4: h == 1 and h == Hl - 2 are different rows
      if (w == 1) qw[nw++] = 0;
      if (w == Wl - 2) qw[nw++] = Wl + 1;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < nh; ++a)
      for (int b = 0; b < nw; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(dxp + (((size_t)n * Hp + qh[a]) * Wp + qw[b]) * Kc + ph * Cin + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    const int pi = ph >> 1, pj = ph & 1;
    *reinterpret_cast<float4*>(dx + (((size_t)n * Hl * ups + h * ups + pi) * W + (w * ups + pj)) * Cin + c) = s;
  }
}

// dM = A dy A^T (the adjoint of y = A^T M A), A = [[1, 0], [1, 1], [1, -1], [0, -1]]; one thread: one tile x 4 channels
__global__ __launch_bounds__(256) void k_wino_dy(const float* __restrict__ dy, float* __restrict__ dm, WinoP p) {
  __shared__ uint32_t amax_red[4];
  uint32_t amax = 0;
  const uint32_t c4n = (uint32_t)p.Cout / 4;
  const size_t total = (size_t)p.T * c4n;
  const size_t plane = (size_t)p.T * p.Cout;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(idx / c4n);
    const int co = (int)(idx - (size_t)t * c4n) * 4;
    const uint32_t n = fd_div(t, p.fd_thtw);
    const uint32_t rem = t - n * (uint32_t)(p.th * p.tw);
    const uint32_t ty = fd_div(rem, p.fd_tw);
    const int tx = (int)(rem - ty * (uint32_t)p.tw);
    const float* base = dy + (((size_t)n * p.Hl + 2 * ty) * p.Wl + 2 * tx) * p.Cout + co;
    const float4 y00 = *reinterpret_cast<const float4*>(base);
    const float4 y01 = *reinterpret_cast<const float4*>(base + p.Cout);
    const float4 y10 = *reinterpret_cast<const float4*>(base + (size_t)p.Wl * p.Cout);
    const float4 y11 = *reinterpret_cast<const float4*>(base + (size_t)p.Wl * p.Cout + p.Cout);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 z[4][2];
    z[0][0] = y00;              z[0][1] = y01;
    z[1][0] = f4sum(y00, y10);  z[1][1] = f4sum(y01, y11);
    z[2][0] = f4sub(y00, y10);  z[2][1] = f4sub(y01, y11);
    z[3][0] = f4sub(zero, y10); z[3][1] = f4sub(zero, y11);
    float* mb = dm + (size_t)t * p.Cout + co;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 o1 = f4sum(z[a][0], z[a][1]), o2 = f4sub(z[a][0], z[a][1]);
      *reinterpret_cast<float4*>(mb + (size_t)(a * 4 + 0) * plane) = z[a][0];
      *reinterpret_cast<float4*>(mb + (size_t)(a * 4 + 1) * plane) = o1;
      *reinterpret_cast<float4*>(mb + (size_t)(a * 4 + 2) * plane) = o2;
      *reinterpret_cast<float4*>(mb + (size_t)(a * 4 + 3) * plane) = f4sub(zero, z[a][1]);
      amax = max(amax, max(max(vcg_abs_bits4(z[a][0]), vcg_abs_bits4(z[a][1])), max(vcg_abs_bits4(o1), vcg_abs_bits4(o2))));
    }
  }
  if (p.amax_slot) vcg_amax_publish(amax, p.amax_slot, p.amax_gen, amax_red);
}

// ---- round 3: the weight gradient of the deep layers as a plain planes GEMM ------------------------------------------------------
// dU[xi] = V[xi]^T dM[xi] reduces over the TILES.  With few tiles and many channels (R blocks: 512 tiles, 1024 x 1024 channels; D4,
// D3) the stream-K kernel of conv_igemm.hip has 16 K-steps per 128 x 128 output tile, leaves partial tiles in slabs and runs at a
// third of the rate of the forward GEMM on the same FLOPs.  Here the two operands are written TRANSPOSED by their transforms —
// Vt[xi][k][t], dMt[xi][co][t] as blocked planes along t (the reduction index contiguous, exactly the layout of a GEMM operand) —
// and the 16 products are one launch of the forward's own GEMM (k_gemm_planes_dma: rows = Kc, N = Cout, K = T), whole tiles, no
// slabs; k_wino_wgrad_reduce then reads ONE part.  The transposition happens in LDS inside the transform kernels: a block owns
// 32 tiles x 32 channels, every thread one tile x 4 channels, and per four transform points the 4 x 2 x 32 rows (point, piece,
// channel) of 32 tiles = 64 bytes go out as whole half-lines.
constexpr int WT_PITCH = 34;                                    // fp16 elements per LDS row of 32 tiles (68 bytes: rows 4-byte aligned)
__device__ __forceinline__ void wino_tr_store(unsigned short* __restrict__ sh, const float4* __restrict__ o4, float inv, int tl, int q,
                                              unsigned short* __restrict__ dst, size_t xi_stride, size_t row0, int TB, int tb) {
  // o4[4]: the values of four transform points for this thread's (tile tl, channels 4 q .. 4 q + 3); dst rows: [xi][channel][TB][2][32]
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    uint2 h, l;
    split4h(o4[x], inv, h, l);
    const unsigned short hv[4] = {(unsigned short)h.x, (unsigned short)(h.x >> 16), (unsigned short)h.y, (unsigned short)(h.y >> 16)};
    const unsigned short lv[4] = {(unsigned short)l.x, (unsigned short)(l.x >> 16), (unsigned short)l.y, (unsigned short)(l.y >> 16)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sh[((x * 2 + 0) * 32 + 4 * q + e) * WT_PITCH + tl] = hv[e];
      sh[((x * 2 + 1) * 32 + 4 * q + e) * WT_PITCH + tl] = lv[e];
    }
  }
  __syncthreads();
  {
    // thread -> (point x, channel c, piece pc): its 32 tiles = 64 contiguous bytes; lanes 2 j, 2 j + 1 = the two pieces = one 128-byte line
    const int pc = threadIdx.x & 1, c = (threadIdx.x >> 1) & 31, x = threadIdx.x >> 6;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(sh + ((x * 2 + pc) * 32 + c) * WT_PITCH);
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = make_uint4(src[4 * j], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]);
    unsigned short* o = dst + (size_t)x * xi_stride + ((row0 + c) * TB + tb) * VCG_PBLK + pc * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(o + 8 * j) = v[j];
  }
  __syncthreads();
}
// V = B^T d B of a block of 32 tiles x 32 channels: the forward's planes (WinoP::vplanes, may be null) and / or the transposed
// planes vt[xi][k][T / 32][2][32].  Grid (T / 32, Kc / 32); T % 32 == 0, Kc % 32 == 0.
__global__ __launch_bounds__(256) void k_wino_in_tr(WinoP p, unsigned short* __restrict__ vt) {
  __shared__ __attribute__((aligned(16))) unsigned short sh[4 * 2 * 32 * WT_PITCH];
  float sc, inv;
  vcg_scale_of(vcg_amax_bits(p.amax_x), p.amax_x.shift, sc, inv);
  const int tl = threadIdx.x >> 3, q = threadIdx.x & 7;
  const uint32_t t = blockIdx.x * 32 + tl;
  const uint32_t k4 = blockIdx.y * 8 + q;
  const uint32_t ph = fd_div(k4, p.fd_c4);
  const int c = (int)(k4 - ph * (uint32_t)(p.Cin / 4)) * 4;
  const int pi = (int)(ph >> 1), pj = (int)(ph & 1);
  const uint32_t n = fd_div(t, p.fd_thtw);
  const uint32_t rem = t - n * (uint32_t)(p.th * p.tw);
  const uint32_t ty = fd_div(rem, p.fd_tw);
  const int tx = (int)(rem - ty * (uint32_t)p.tw);
  const float* xn = p.x + (size_t)n * p.H * p.W * p.Cin;
  float4 pmu = make_float4(0.f, 0.f, 0.f, 0.f), prs = pmu;
  if (p.pre_mean) {
    pmu = *reinterpret_cast<const float4*>(p.pre_mean + (size_t)n * p.Cin + c);
    prs = *reinterpret_cast<const float4*>(p.pre_rstd + (size_t)n * p.Cin + c);
  }
  float4 d[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int ih = 2 * (int)ty - p.off + r;
    bool okh = true;
    if (p.reflect) ih = reflect_idx(ih, p.Hl);
    else okh = ih >= 0 && ih < p.Hl;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int iw = 2 * tx - p.off + s;
      bool ok = okh;
      if (p.reflect) iw = reflect_idx(iw, p.Wl);
      else ok = ok && iw >= 0 && iw < p.Wl;
      d[r][s] = ok ? *reinterpret_cast<const float4*>(xn + ((size_t)(ih * p.ups + pi) * p.W + (iw * p.ups + pj)) * p.Cin + c)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.pre_mean && ok) d[r][s] = wino_pre_apply(d[r][s], pmu, prs, p.pre_act);
    }
  }
  float4 e[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    e[0][s] = f4sub(d[0][s], d[2][s]);
    e[1][s] = f4sum(d[1][s], d[2][s]);
    e[2][s] = f4sub(d[2][s], d[1][s]);
    e[3][s] = f4sub(d[1][s], d[3][s]);
  }
  const uint32_t k = k4 * 4;
  const size_t KB = (size_t)p.Kc / 32;
  const int TB = p.T / 32;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float4 o[4] = {f4sub(e[a][0], e[a][2]), f4sum(e[a][1], e[a][2]), f4sub(e[a][2], e[a][1]), f4sub(e[a][1], e[a][3])};
    if (p.vplanes) {
      unsigned short* vb = p.vplanes + ((size_t)t * KB + k / 32) * VCG_PBLK + (k & 31);
      const size_t plane = (size_t)p.T * KB * VCG_PBLK;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint2 h, l;
        split4h(o[b], inv, h, l);
        unsigned short* w = vb + (size_t)(a * 4 + b) * plane;
        *reinterpret_cast<uint2*>(w) = h;
        *reinterpret_cast<uint2*>(w + 32) = l;
      }
    }
    if (vt) wino_tr_store(sh, o, inv, tl, q, vt + (size_t)(a * 4) * p.Kc * TB * VCG_PBLK, (size_t)p.Kc * TB * VCG_PBLK, (size_t)blockIdx.y * 32, TB, (int)blockIdx.x);
  }
}
// dM = A dy A^T of a block of 32 tiles x 32 output channels, as transposed planes dmt[xi][co][T / 32][2][32] of dM / s, s from
// 4 x the largest magnitude of dy (WinoP::amax_x).  Grid (T / 32, Cout / 32).
__global__ __launch_bounds__(256) void k_wino_dy_tr(const float* __restrict__ dy, unsigned short* __restrict__ dmt, WinoP p) {
  __shared__ __attribute__((aligned(16))) unsigned short sh[4 * 2 * 32 * WT_PITCH];
  float sc, inv;
  vcg_scale_of(vcg_amax_bits(p.amax_x), p.amax_x.shift, sc, inv);
  const int tl = threadIdx.x >> 3, q = threadIdx.x & 7;
  const uint32_t t = blockIdx.x * 32 + tl;
  const int co = ((int)blockIdx.y * 8 + q) * 4;
  const uint32_t n = fd_div(t, p.fd_thtw);
  const uint32_t rem = t - n * (uint32_t)(p.th * p.tw);
  const uint32_t ty = fd_div(rem, p.fd_tw);
  const int tx = (int)(rem - ty * (uint32_t)p.tw);
  const float* base = dy + (((size_t)n * p.Hl + 2 * ty) * p.Wl + 2 * tx) * p.Cout + co;
  const float4 y00 = *reinterpret_cast<const float4*>(base);
  const float4 y01 = *reinterpret_cast<const float4*>(base + p.Cout);
  const float4 y10 = *reinterpret_cast<const float4*>(base + (size_t)p.Wl * p.Cout);
  const float4 y11 = *reinterpret_cast<const float4*>(base + (size_t)p.Wl * p.Cout + p.Cout);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 z[4][2];
  z[0][0] = y00;              z[0][1] = y01;
  z[1][0] = f4sum(y00, y10);  z[1][1] = f4sum(y01, y11);
  z[2][0] = f4sub(y00, y10);  z[2][1] = f4sub(y01, y11);
  z[3][0] = f4sub(zero, y10); z[3][1] = f4sub(zero, y11);
  const int TB = p.T / 32;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float4 o[4] = {z[a][0], f4sum(z[a][0], z[a][1]), f4sub(z[a][0], z[a][1]), f4sub(zero, z[a][1])};
    wino_tr_store(sh, o, inv, tl, q, dmt + (size_t)(a * 4) * p.Cout * TB * VCG_PBLK, (size_t)p.Cout * TB * VCG_PBLK, (size_t)blockIdx.y * 32, TB, (int)blockIdx.x);
  }
}

// ------------------------------------------------------------------ host side
static int wino_blocks(size_t work) {
  size_t b = (work + 255) / 256;
  if (b > 8192) b = 8192;
  return b < 1 ? 1 : (int)b;
}

// shape class the packed weights carry a transformed copy for (no spatial condition: packing sees no image size)
bool vcg_wino_weight_ok(const ConvGeom& g) {
  return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && g.ups * g.ups * g.Cin >= 128 && g.Cout >= 64 &&
         g.Cin % 4 == 0 && g.Cout % 64 == 0 && (g.ups * g.ups * g.Cin) % 64 == 0;     // N of both GEMMs in 64-column tiles
}
// a map Winograd can take at all (whatever the direction's gate says)
static bool wino_map_ok(const ConvGeom& g) {
  if (!vcg_wino_weight_ok(g)) return false;
#ifdef VCG_STAMP
  if (getenv("VCG_NO_WINOGRAD")) return false;     // A/B timing in the diagnostic build only
#endif
  return !(g.Ho < 4 || g.Wo < 4 || (g.Ho & 1) || (g.Wo & 1));
}
bool vcg_wino_fwd_ok(const ConvGeom& g) {
  if (!wino_map_ok(g)) return false;
  const unsigned long long T = (unsigned long long)g.N * (g.Ho / 2) * (g.Wo / 2);
  const unsigned long long Kc = (unsigned long long)g.ups * g.ups * g.Cin;
  // the transforms move 16 T (Kc + Cout) floats each way while the GEMMs save ~ T Kc Cout multiplications: the forward
  // pays from Kc Cout / (Kc + Cout) ~ 100 on (vcg_wino_gate_fwd above)
  if (Kc * g.Cout < (unsigned long long)vcg_wino_gate_fwd() * (Kc + g.Cout)) return false;
  return T * Kc * 4 < (1ull << 31) && T * g.Cout * 4 < (1ull << 31) && T * Kc * 16 < (1ull << 32);
}
// one transformed copy of the kernel as fp16 blocked planes: VCG_NP pieces x 2 bytes per value
size_t vcg_wino_weight_floats(const ConvGeom& g) { return (size_t)16 * g.ups * g.ups * g.Cin * g.Cout * VCG_NP / 2; }
size_t vcg_wino_fwd_workspace(const ConvGeom& g) {
  const size_t T = (size_t)g.N * (g.Ho / 2) * (g.Wo / 2);
  return (size_t)16 * T * ((size_t)g.ups * g.ups * g.Cin + g.Cout) * sizeof(float) + 512;
}
int vcg_wino_weight(const ConvGeom& g, const float* w_oihw, float* u, const VcgAmax& amax_w, hipStream_t st) {
  const size_t total = (size_t)g.ups * g.ups * g.Cin * g.Cout / 4;
  VcgAmax au = amax_w; au.shift += WINO_U_SHIFT;
  // Bt operand of the forward GEMM M = V . U: [xi][co][k], k contiguous, pre-split
  hipLaunchKernelGGL(k_wino_weight_planes<false>, dim3(wino_blocks(total)), dim3(256), 0, st, w_oihw, (unsigned short*)u, g.Cin, g.Cout,
                     g.ups, g.cin_log, g.cout_log, au);
  VCG_LAUNCH_CHECK("vcg_wino_weight");
  return 0;
}

static WinoP wino_params(const ConvGeom& g) {
  WinoP p = {};
  p.x = nullptr; p.v = nullptr; p.m = nullptr; p.bias = nullptr; p.y = nullptr;
  p.N = g.N; p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.Cout = g.Cout; p.Hl = g.Hl; p.Wl = g.Wl; p.ups = g.ups;
  p.reflect = g.reflect; p.act = g.act; p.cout_log = g.cout_log;
  p.th = g.Ho / 2; p.tw = g.Wo / 2; p.T = g.N * p.th * p.tw; p.Kc = g.ups * g.ups * g.Cin;
  p.fd_k4 = make_fastdiv((uint32_t)p.Kc / 4); p.fd_tw = make_fastdiv((uint32_t)p.tw);
  p.fd_thtw = make_fastdiv((uint32_t)(p.th * p.tw)); p.fd_c4 = make_fastdiv((uint32_t)g.Cin / 4);
  p.fd_co4 = make_fastdiv((uint32_t)g.Cout / 4);
  p.off = 1;
  p.amax_slot = nullptr; p.amax_gen = 0;
  p.amax_x = vcg_amax_const(0); p.vplanes = nullptr;
  p.pre_mean = nullptr; p.pre_rstd = nullptr; p.pre_act = VCG_ACT_NONE;
  return p;
}

// weight gradient: V = B^T x B — the forward's own copy when the caller kept it (vcg_conv_fwd_in's `saved`: 4x the
// activation, 3.5 GB over a CycleVAEGAN step at batch 8, against 288 GB of HBM), recomputed otherwise — dM = A dy A^T, then
// the batched reduction over tiles and the back-transform (conv_igemm.hip)
bool vcg_wino_wgrad_ok(const ConvGeom& g) {
  // the transforms move 16 T (Kc + Cout) floats each way while the GEMMs save ~ T Kc Cout multiplications: it pays
  // from Kc Cout / (Kc + Cout) ~ 128 on (measured: 171 -> 1.46x, 85 -> 0.9x)
  const long long kc = (long long)g.ups * g.ups * g.Cin;
  return vcg_wino_fwd_ok(g) && kc % 128 == 0 && kc * g.Cout >= vcg_wino_gate_wgrad() * (kc + g.Cout);
}
// VCG_WGRAD_TR=0: the stream-K reduction for every layer (A/B measurements)
static bool wgrad_tr_on() {
  static const int on = [] { const char* e = getenv("VCG_WGRAD_TR"); return e ? atoi(e) : 1; }();
  return on != 0;
}
// the weight gradient as a plain planes GEMM over transposed operands (k_wino_in_tr / k_wino_dy_tr above): enough output tiles
// of 256 x 128 to fill the chip, whole 32-tile blocks.  Measured per layer at batch 8 (profiles/r03_wgrad_tr.txt; the forward pays
// for the second, transposed copy of V): R weight gradient 154 -> 144 us but forward 81 -> 91; D3 226 -> 182 but 178 -> 223 (V is
// 134 MB there); D4 319 -> 280 against 138 -> 155 — the one layer where it nets (VCG_WGRAD_TR=2: every layer that qualifies)
bool vcg_wino_wgrad_tr_ok(const ConvGeom& g) {
  if (!wgrad_tr_on() || !vcg_wino_wgrad_ok(g)) return false;
  const long long kc = (long long)g.ups * g.ups * g.Cin, T = (long long)g.N * (g.Ho / 2) * (g.Wo / 2);
  // round 4: every qualifying layer (R, D3, D4) by default — re-measured on the round-4 build, same box, alternating runs: 36.28 / 36.07
  // ms per step with D4 alone (VCG_WGRAD_TR=1, round 3's default) against 35.59 / 35.75 with all three (+1.4 %)
  static const int all = [] { const char* e = getenv("VCG_WGRAD_TR"); return !e || atoi(e) == 2; }();
  if (!all && !(kc >= 2048 && T <= 1024)) return false;
  return T % 32 == 0 && kc % 32 == 0 && g.Cout % 128 == 0 && kc >= 256 && ((kc + 255) / 256) * (g.Cout / 128) * 16 >= 192 &&
         16 * kc * T * VCG_NP < (1ll << 32) && 16 * (long long)g.Cout * T * VCG_NP < (1ll << 32);
}
size_t vcg_wino_wgrad_workspace(const ConvGeom& g) {
  const int T = g.N * (g.Ho / 2) * (g.Wo / 2);
  return vcg_wino_fwd_workspace(g) + vcg_wino_wgrad_core_workspace(g, T);
}
// the kept V, followed by the bit pattern of its largest magnitude (the weight gradient's GEMMs scale V by it, as the forward's did)
static size_t wino_v_floats(const ConvGeom& g) { return (size_t)16 * g.N * (g.Ho / 2) * (g.Wo / 2) * g.ups * g.ups * g.Cin; }
size_t vcg_wino_saved_floats(const ConvGeom& g) { return vcg_wino_wgrad_ok(g) ? wino_v_floats(g) + 16 : 0; }
int vcg_wino_wgrad(const ConvGeom& g, const float* x, const float* dy, float* gw_oihw, void* ws, size_t ws_bytes, hipStream_t st,
                   const float* v_saved, uint64_t x_handle, uint64_t dy_handle) {
  VCG_CHECK_ARG(ws && ws_bytes >= vcg_wino_wgrad_workspace(g), "vcg_conv_wgrad: workspace too small for the Winograd path");
  WinoP p = wino_params(g);
  float* V = (float*)ws;
  float* dM = V + (((size_t)16 * p.T * p.Kc + 63) / 64) * 64;
  const size_t tbytes = vcg_wino_fwd_workspace(g);
  if (vcg_wino_wgrad_tr_ok(g)) {
    // transposed operands -> one planes GEMM dU[xi] = Vt[xi] . dMt[xi]^T (rows Kc, N Cout, K = T) -> back-transform of ONE part
    VcgAmax amax_v;
    const unsigned short* Vt;
    if (v_saved) {                     // the forward left Vt (vcg_wino_fwd) and, behind it, the word its scale came from
      Vt = reinterpret_cast<const unsigned short*>(v_saved);
      amax_v = vcg_amax_stored(v_saved + wino_v_floats(g), WINO_V_SHIFT);
    } else {
      if (vcg_operand_amax(x, (size_t)g.N * g.H * g.W * g.Cin, x_handle, WINO_V_SHIFT, st, &amax_v)) return -2;
      p.x = x; p.amax_x = amax_v; p.vplanes = nullptr;
      hipLaunchKernelGGL(k_wino_in_tr, dim3(p.T / 32, p.Kc / 32), dim3(256), 0, st, p, (unsigned short*)V);
      Vt = reinterpret_cast<const unsigned short*>(V);
    }
    VcgAmax amax_d;
    if (vcg_operand_amax(dy, (size_t)g.N * g.Ho * g.Wo * g.Cout, dy_handle, WINO_V_SHIFT, st, &amax_d)) return -2;   // |A dy A^T| <= 4 max|dy|
    WinoP pd = p;
    pd.amax_x = amax_d;
    hipLaunchKernelGGL(k_wino_dy_tr, dim3(p.T / 32, g.Cout / 32), dim3(256), 0, st, dy, (unsigned short*)dM, pd);
    VCG_LAUNCH_CHECK("vcg_conv_wgrad(winograd transforms, transposed)");
    float* dU = reinterpret_cast<float*>((char*)ws + tbytes);
    if (vcg_gemm_planes_batched(Vt, dM, dU, p.Kc, p.T, g.Cout, 16, amax_v, amax_d, st, nullptr)) return -2;
    return vcg_wino_wgrad_reduce_one(g, dU, gw_oihw, st);
  }
  p.x = x; p.v = V;
  VcgAmax amax_v;
  const bool planes = wino_planes_on();
  if (v_saved) {
    // the forward's V: fp32 with its own amax behind it, or — planes mode — pre-split by 4 x the input's amax (the word behind it)
    V = const_cast<float*>(v_saved);
    amax_v = vcg_amax_stored(v_saved + wino_v_floats(g), planes ? WINO_V_SHIFT : 0);
  } else if (planes) {
    if (vcg_operand_amax(x, (size_t)g.N * g.H * g.W * g.Cin, x_handle, WINO_V_SHIFT, st, &amax_v)) return -2;
    p.amax_x = amax_v; p.vplanes = (unsigned short*)V;
    hipLaunchKernelGGL(k_wino_in_planes, dim3(wino_blocks((size_t)p.T * p.Kc / 4)), dim3(256), 0, st, p);
  } else {
    const VcgAmaxOut av = vcg_amax_new(st);
    p.amax_slot = av.slot; p.amax_gen = av.gen;
    hipLaunchKernelGGL(k_wino_in, dim3(wino_blocks((size_t)p.T * p.Kc / 4)), dim3(256), 0, st, p);
    amax_v = vcg_amax_in(av);
  }
  const VcgAmaxOut ad = vcg_amax_new(st);
  p.amax_slot = ad.slot; p.amax_gen = ad.gen;
  hipLaunchKernelGGL(k_wino_dy, dim3(wino_blocks((size_t)p.T * g.Cout / 4)), dim3(256), 0, st, dy, dM, p);
  VCG_LAUNCH_CHECK("vcg_conv_wgrad(winograd transforms)");
  return vcg_wino_wgrad_core(g, V, dM, p.T, gw_oihw, (char*)ws + tbytes, ws_bytes - tbytes, st, amax_v, vcg_amax_in(ad), planes);
}

// data gradient over the padded domain (see k_wino_weight_dgrad)
bool vcg_wino_dgrad_ok(const ConvGeom& g) {
  if (!wino_map_ok(g)) return false;               // its own gate below: the forward's (higher since round 3) says nothing about it
  const long long kc = (long long)g.ups * g.ups * g.Cin;
  if (kc * g.Cout < vcg_wino_gate_dgrad() * (kc + g.Cout)) return false;             // measured: 85 -> 1.22..1.25x (D1, U2), 171 -> 1.3..1.5x
  const unsigned long long Tp = (unsigned long long)g.N * (g.Ho / 2 + 1) * (g.Wo / 2 + 1);
  return Tp * kc * 4 < (1ull << 31) && Tp * g.Cout * 4 < (1ull << 31) && Tp * g.Cout * 16 < (1ull << 32);
}
size_t vcg_wino_dgrad_workspace(const ConvGeom& g) {
  const size_t Tp = (size_t)g.N * (g.Ho / 2 + 1) * (g.Wo / 2 + 1);
  const size_t kc = (size_t)g.ups * g.ups * g.Cin;
  return (size_t)16 * Tp * (kc + g.Cout) * sizeof(float) + 1024;
}
int vcg_wino_weight_dgrad(const ConvGeom& g, const float* w_oihw, float* ud, const VcgAmax& amax_w, hipStream_t st) {
  const size_t total = (size_t)g.ups * g.ups * g.Cin * g.Cout / 4;
  VcgAmax au = amax_w; au.shift += WINO_U_SHIFT;
  // Bt operand of the data-gradient GEMM dXp = Vdy . Ud: [xi][k][co], co contiguous, kernel flipped, pre-split
  hipLaunchKernelGGL(k_wino_weight_planes<true>, dim3(wino_blocks(total)), dim3(256), 0, st, w_oihw, (unsigned short*)ud, g.Cin, g.Cout,
                     g.ups, g.cin_log, g.cout_log, au);
  VCG_LAUNCH_CHECK("vcg_wino_weight_dgrad");
  return 0;
}
int vcg_wino_dgrad(const ConvGeom& g, const float* dy, const float* ud, const void* w_amax, float* dx, void* ws, size_t ws_bytes,
                   hipStream_t st, uint64_t dy_handle) {
  VCG_CHECK_ARG(ws && ws_bytes >= vcg_wino_dgrad_workspace(g), "vcg_conv_dgrad: workspace too small for the Winograd path");
  const int kc = g.ups * g.ups * g.Cin;
  // input transform of dy: a plain (N, Ho, Wo, Cout) image, zero extension, patch origin 2 * tile - 2
  WinoP p = {};
  p.x = dy; p.bias = nullptr; p.y = nullptr;
  p.N = g.N; p.H = g.Ho; p.W = g.Wo; p.Cin = g.Cout; p.Cout = kc; p.Hl = g.Ho; p.Wl = g.Wo; p.ups = 1;
  p.reflect = 0; p.act = VCG_ACT_NONE; p.cout_log = kc;
  p.th = g.Ho / 2 + 1; p.tw = g.Wo / 2 + 1; p.T = g.N * p.th * p.tw; p.Kc = g.Cout; p.off = 2;
  p.amax_slot = nullptr; p.amax_gen = 0;
  p.amax_x = vcg_amax_const(0); p.vplanes = nullptr;
  p.fd_k4 = make_fastdiv((uint32_t)p.Kc / 4); p.fd_tw = make_fastdiv((uint32_t)p.tw);
  p.fd_thtw = make_fastdiv((uint32_t)(p.th * p.tw)); p.fd_c4 = make_fastdiv((uint32_t)p.Cin / 4);
  p.fd_co4 = make_fastdiv((uint32_t)kc / 4);
  float* V = (float*)ws;
  float* M = V + (((size_t)16 * p.T * g.Cout + 63) / 64) * 64;
  p.v = V; p.m = M;
  if (wino_planes_on()) {
    VcgAmax ad;
    if (vcg_operand_amax(dy, (size_t)g.N * g.Ho * g.Wo * g.Cout, dy_handle, WINO_V_SHIFT, st, &ad)) return -2;
    p.amax_x = ad; p.vplanes = (unsigned short*)V;
    hipLaunchKernelGGL(k_wino_in_planes, dim3(wino_blocks((size_t)p.T * g.Cout / 4)), dim3(256), 0, st, p);
    VCG_LAUNCH_CHECK("vcg_conv_dgrad(winograd input transform)");
    if (vcg_gemm_planes_batched(V, ud, M, p.T, g.Cout, kc, 16, ad, vcg_amax_stored(w_amax, WINO_U_SHIFT), st, nullptr)) return -2;
  } else {
    const VcgAmaxOut av = vcg_amax_new(st);
    p.amax_slot = av.slot; p.amax_gen = av.gen;
    hipLaunchKernelGGL(k_wino_in, dim3(wino_blocks((size_t)p.T * g.Cout / 4)), dim3(256), 0, st, p);
    VCG_LAUNCH_CHECK("vcg_conv_dgrad(winograd input transform)");
    if (vcg_gemm_split_batched(V, ud, M, p.T, g.Cout, kc, 16, vcg_amax_in(av), vcg_amax_stored(w_amax, WINO_U_SHIFT), st)) return -2;
  }
  WinoP q = p;
  q.Kc = kc;                                      // the output side: k columns
  // output transform and fold in one pass: the padded image is never written
  hipLaunchKernelGGL(k_wino_out_fold, dim3(wino_blocks((size_t)p.T * kc / 4)), dim3(256), 0, st, (const float*)M, dx, q, g.Hl, g.Wl,
                     g.Cin, g.ups, g.reflect);
  VCG_LAUNCH_CHECK("vcg_conv_dgrad(winograd output transform)");
  return 0;
}

// doubles the Winograd forward writes when it also leaves InstanceNorm chunk partials (vcg_conv_fwd_in)
size_t vcg_wino_fwd_stats_doubles(const ConvGeom& g) {
  const NormPlan pl = vcg_norm_plan(g.N, (g.Ho / 2) * (g.Wo / 2), g.Cout);
  return (size_t)g.N * pl.nchunk * g.Cout * 2;
}
// v_keep: where to leave V = B^T x B for the weight gradient (vcg_wino_saved_floats), instead of the workspace
// `pre` (may be null): x is the raw output of the previous conv and is normalised in the input transform's gather (deferred
// InstanceNorm, vcg_conv_fwd_in_pre).  Its amax is then not measured but BOUNDED: a normalised channel of HW values has
// |xhat| <= sqrt(HW - 1) (and ReLU / LeakyReLU / Tanh / Sigmoid of it no more), typically 2^3..2^5 above the true maximum — the
// fp16 pair keeps 22 bits for every element within 2^17 of the scale, so an element now needs to be within ~2^13 of the true
// maximum to keep them all; smaller ones keep an absolute error of 2^-36 of the maximum, far under the fp32 rounding of any sum
// they enter (the argument of vcg_common.h for the per-tensor scale).
bool vcg_wino_pre_ok(const ConvGeom& g) { return vcg_wino_fwd_ok(g) && wino_planes_on(); }
int vcg_wino_fwd(const ConvGeom& g, const float* x, const float* u, const void* w_amax, const float* bias, float* y, void* ws,
                 size_t ws_bytes, hipStream_t st, double* in_part, const VcgInTail* tail_req, float* v_keep, uint64_t x_handle,
                 const VcgPre* pre) {
  VCG_CHECK_ARG(ws && ws_bytes >= vcg_wino_fwd_workspace(g), "vcg_conv_fwd: workspace too small for the Winograd path (%zu)",
                ws_bytes);
  WinoP p = wino_params(g);
  float* V = v_keep ? v_keep : (float*)ws;
  float* M = (float*)ws + (((size_t)16 * p.T * p.Kc + 63) / 64) * 64;
  p.x = x; p.v = V; p.m = M; p.bias = bias; p.y = y;
  uint32_t pre_bits = 0;
  if (pre && pre->mean) {
    VCG_CHECK_ARG(wino_planes_on(), "vcg_conv_fwd_in_pre: needs the planes mode");
    p.pre_mean = pre->mean; p.pre_rstd = pre->rstd; p.pre_act = pre->act;
    const float bound = sqrtf((float)g.H * (float)g.W);
    memcpy(&pre_bits, &bound, 4);
  }
  uint32_t* const keep_word = v_keep ? reinterpret_cast<uint32_t*>(v_keep + wino_v_floats(g)) : nullptr;
  if (v_keep && vcg_wino_wgrad_tr_ok(g)) {
    // this layer's weight gradient wants V TRANSPOSED (vcg_wino_wgrad): one transform kernel writes the forward's planes into
    // the workspace and the transposed ones into the caller's buffer
    VCG_CHECK_ARG(wino_planes_on(), "VCG_WINO_PLANES=0 needs VCG_WGRAD_TR=0");
    VcgAmax ax;
    if (pre_bits) ax = vcg_amax_const(pre_bits, WINO_V_SHIFT);
    else if (vcg_operand_amax(x, (size_t)g.N * g.H * g.W * g.Cin, x_handle, WINO_V_SHIFT, st, &ax)) return -2;
    V = (float*)ws;
    p.v = V; p.amax_x = ax; p.vplanes = (unsigned short*)V;
    hipLaunchKernelGGL(k_wino_in_tr, dim3(p.T / 32, p.Kc / 32), dim3(256), 0, st, p, (unsigned short*)v_keep);
    VCG_LAUNCH_CHECK("vcg_conv_fwd(winograd input transform, + transposed)");
    if (vcg_gemm_planes_batched(V, u, M, p.T, p.Kc, g.Cout, 16, ax, vcg_amax_stored(w_amax, WINO_U_SHIFT), st, keep_word)) return -2;
  } else if (wino_planes_on()) {
    VcgAmax ax;
    if (pre_bits) ax = vcg_amax_const(pre_bits, WINO_V_SHIFT);
    else if (vcg_operand_amax(x, (size_t)g.N * g.H * g.W * g.Cin, x_handle, WINO_V_SHIFT, st, &ax)) return -2;
    p.amax_x = ax; p.vplanes = (unsigned short*)V;
    hipLaunchKernelGGL(k_wino_in_planes, dim3(wino_blocks((size_t)p.T * p.Kc / 4)), dim3(256), 0, st, p);
    VCG_LAUNCH_CHECK("vcg_conv_fwd(winograd input transform)");
    // a kept V keeps the word it was scaled by behind it (vcg_wino_saved_floats): the weight gradient's GEMMs need the same scale
    if (vcg_gemm_planes_batched(V, u, M, p.T, p.Kc, g.Cout, 16, ax, vcg_amax_stored(w_amax, WINO_U_SHIFT), st, keep_word)) return -2;
  } else {
    const VcgAmaxOut av = vcg_amax_new(st);
    p.amax_slot = av.slot; p.amax_gen = av.gen;
    hipLaunchKernelGGL(k_wino_in, dim3(wino_blocks((size_t)p.T * p.Kc / 4)), dim3(256), 0, st, p);
    VCG_LAUNCH_CHECK("vcg_conv_fwd(winograd input transform)");
    if (vcg_gemm_split_batched(V, u, M, p.T, p.Kc, g.Cout, 16, vcg_amax_in(av), vcg_amax_stored(w_amax, WINO_U_SHIFT), st, keep_word)) return -2;
  }
  if (in_part) {
    // statistics of y for the InstanceNorm that follows: chunk partials from this epilogue, combined by its last block
    const NormPlan pl = vcg_norm_plan(g.N, p.th * p.tw, g.Cout);
    VcgInTail tail = vcg_in_tail_make(tail_req->out1, tail_req->out2, g.N * pl.cgroups, tail_req->HW, tail_req->eps);
    hipLaunchKernelGGL(k_wino_out_stats, dim3(pl.nchunk, g.N, pl.cgroups), dim3(256), 0, st, p, in_part, pl, tail);
    VCG_LAUNCH_CHECK("vcg_conv_fwd(winograd output transform)");
    return tail.out1 ? 0 : vcg_in_finalize(in_part, tail_req->out1, tail_req->out2, g.N, tail_req->HW, g.Cout, pl.nchunk, tail_req->eps, st);
  } else {
    hipLaunchKernelGGL(k_wino_out, dim3(wino_blocks((size_t)p.T * g.Cout / 4)), dim3(256), 0, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv_fwd(winograd output transform)");
  return 0;
}
