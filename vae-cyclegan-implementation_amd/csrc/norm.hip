// InstanceNorm2d (eps 1e-5, biased variance, no affine, no running stats) forward and
// backward on NHWC activations.  Replaces nn.InstanceNorm2d at Networks.py:61,88,102,105,123
// together with the activation that sits next to it (ReLU before the norm in D/U/R.conv1
// :94,:111,:129; after it in CaSb :79-80) and, for U, the nn.PixelShuffle(2) (:121) that
// follows in the next block, folded here into the store address.
//
// All of it is HBM-bound.  Per (n, c) reductions run in two stages: a workgroup sums one
// pixel chunk for a group of channel quads (float4 per lane, 64 lanes = 1 KiB per row
// segment), then a tiny kernel combines the chunk partials in double precision in a fixed
// order, so results are bitwise reproducible and free of the E[x^2]-E[x]^2 cancellation
// a single fp32 pass would have.
#include "vcg_common.h"
#include <stdlib.h>

static NormPlan make_plan(int N, int HW, int C) { return vcg_norm_plan(N, HW, C); }

// ---- stage 1 of every per-(n,c) reduction -------------------------------------------
// MODE 0: (sum t, sum t^2)
// MODE 1: (sum g', sum g' * xhat)  with xhat=(t-mean)*rstd, g' = g * post_act'(xhat)
template <int MODE, bool TAIL>
__global__ __launch_bounds__(256) void k_in_partial(const float* __restrict__ t, const float* __restrict__ g,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    double* __restrict__ part, int H, int W, int C, NormPlan pl,
                                                    int post_act, int shuffle, VcgInTail tail) {
  // fp64 accumulators: these sums cancel (var = E[x^2] - mean^2 for channels with |mean| >> std;
  // sum g' and sum g'*xhat in the backward), and torch's CPU kernel — the reference's numerics —
  // accumulates them in double too.  The kernel is HBM-bound, the extra fp64 adds are free.
  __shared__ double r1[256 * 4];
  __shared__ double r2[256 * 4];
  const int tc = threadIdx.x % pl.TC, tp = threadIdx.x / pl.TC;
  const int c4 = blockIdx.z * pl.TC + tc;
  const int n = blockIdx.y;
  const int HW = H * W;
  const int pb = blockIdx.x * pl.chunk;
  int pe = pb + pl.chunk;
  if (pe > HW) pe = HW;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (c4 * 4 < C) {
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
      float4 m4 = *reinterpret_cast<const float4*>(mean + (size_t)n * C + c4 * 4);
      float4 r4 = *reinterpret_cast<const float4*>(rstd + (size_t)n * C + c4 * 4);
      mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
      rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
    }
    for (int pix = pb + tp; pix < pe; pix += pl.TP) {
      float4 v4 = *reinterpret_cast<const float4*>(t + ((size_t)n * HW + pix) * C + c4 * 4);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += (double)v[e];
          s2[e] += (double)v[e] * (double)v[e];
        }
      } else {
        float gv[4];
        if (shuffle) {
          // g lives in the pixel-shuffled tensor (N, 2H, 2W, C/4): element e of the quad
          // sits at pixel (2h + e/2, 2w + e%2), channel c4
          const int h = pix / W, w = pix - h * W;
          const int Cq = C / 4;
          const float* gp = g + (((size_t)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * Cq + c4;
          gv[0] = gp[0];
          gv[1] = gp[Cq];
          gv[2] = gp[(size_t)2 * W * Cq];
          gv[3] = gp[(size_t)2 * W * Cq + Cq];
        } else {
          float4 g4 = *reinterpret_cast<const float4*>(g + ((size_t)n * HW + pix) * C + c4 * 4);
          gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (v[e] - mu[e]) * rs[e];
          const float gg = gv[e] * act_grad_from_in(xh, post_act);
          s1[e] += (double)gg;
          s2[e] += (double)gg * (double)xh;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r1[threadIdx.x * 4 + e] = s1[e];
    r2[threadIdx.x * 4 + e] = s2[e];
  }
  __syncthreads();
  if (tp == 0 && c4 * 4 < C) {
    for (int k = 1; k < pl.TP; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[e] += r1[(k * pl.TC + tc) * 4 + e];
        s2[e] += r2[(k * pl.TC + tc) * 4 + e];
      }
    }
    double* o = part + (((size_t)n * pl.nchunk + blockIdx.x) * C + c4 * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (TAIL) {
        vcg_store_sc1(o + 2 * e, s1[e]);
        vcg_store_sc1(o + 2 * e + 1, s2[e]);
      } else {
        o[2 * e] = s1[e];
        o[2 * e + 1] = s2[e];
      }
    }
  }
  // the last of this (image, channel group)'s chunk blocks combines them (vcg_common.h; VCG_IN_TAIL=1)
  if (TAIL) {
    __syncthreads();                  // r1 / r2 are free again
    vcg_in_tail_run<MODE>(tail, part, n, blockIdx.z * pl.TC * 4, pl.TC * 4, C, pl.nchunk, tail.counters + n * pl.cgroups + blockIdx.z,
                          (uint32_t)pl.nchunk, r1);
  }
}

// MODE 0: mean, rstd.  MODE 1: (s1/HW, s2/HW) interleaved into out1[(n*C+c)*2 + {0,1}]
template <int MODE>
__global__ void k_in_final(const double* __restrict__ part, float* __restrict__ out1, float* __restrict__ out2,
                           int N, int HW, int C, int nchunk, float eps) {
  // 32 (n,c) pairs x 8 chunk lanes per block: the chunk loop is a dependent chain of L2 round trips,
  // so it is split 8 ways and combined through LDS in a fixed order
  __shared__ double ra[8][32], rb[8][32];
  const int il = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + il;
  const bool ok = idx < N * C;
  int n = 0, c = 0;
  double a = 0.0, b = 0.0;
  if (ok) {
    n = idx / C; c = idx - n * C;
    for (int k = kl; k < nchunk; k += 8) {
      const double* p = part + (((size_t)n * nchunk + k) * C + c) * 2;
      a += p[0];
      b += p[1];
    }
  }
  ra[kl][il] = a;
  rb[kl][il] = b;
  __syncthreads();
  if (kl != 0 || !ok) return;
  for (int k = 1; k < 8; ++k) { a += ra[k][il]; b += rb[k][il]; }
  if (MODE == 0) {
    double m = a / HW;
    double var = b / HW - m * m;
    if (var < 0.0) var = 0.0;
    out1[idx] = (float)m;
    out2[idx] = (float)(1.0 / sqrt(var + (double)eps));
  } else {
    out1[(size_t)idx * 2] = (float)(a / HW);
    out1[(size_t)idx * 2 + 1] = (float)(b / HW);
  }
}

// (every kernel below that writes an activation or a gradient also publishes its largest magnitude — vcg_common.h: the
// convolution that reads the tensor scales it by that instead of measuring it with a pass of its own)
__global__ __launch_bounds__(256) void k_in_apply(const float* __restrict__ t, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ residual,
                                                  float* __restrict__ out, int N, int H, int W, int C, int post_act,
                                                  int shuffle, unsigned long long* amax_slot, uint32_t amax_gen) {
  __shared__ uint32_t amax_red[4];
  uint32_t amax = 0;
  const int C4 = C / 4;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const size_t pixg = idx / C4;  // n*HW + pix
    const int n = (int)(pixg / ((size_t)H * W));
    float4 v = *reinterpret_cast<const float4*>(t + pixg * C + c4 * 4);
    float4 mu = *reinterpret_cast<const float4*>(mean + (size_t)n * C + c4 * 4);
    float4 rs = *reinterpret_cast<const float4*>(rstd + (size_t)n * C + c4 * 4);
    float4 o;
    o.x = act_apply((v.x - mu.x) * rs.x, post_act);
    o.y = act_apply((v.y - mu.y) * rs.y, post_act);
    o.z = act_apply((v.z - mu.z) * rs.z, post_act);
    o.w = act_apply((v.w - mu.w) * rs.w, post_act);
    if (residual) {
      float4 r = *reinterpret_cast<const float4*>(residual + pixg * C + c4 * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    amax = max(amax, vcg_abs_bits4(o));
    if (shuffle) {
      const int pix = (int)(pixg - (size_t)n * H * W);
      const int h = pix / W, w = pix - h * W;
      float* op = out + (((size_t)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * C4 + c4;
      op[0] = o.x;
      op[C4] = o.y;
      op[(size_t)2 * W * C4] = o.z;
      op[(size_t)2 * W * C4 + C4] = o.w;
    } else {
      *reinterpret_cast<float4*>(out + pixg * C + c4 * 4) = o;
    }
  }
  if (amax_slot) vcg_amax_publish(amax, amax_slot, amax_gen, amax_red);
}

// dt = epi'(t) * rstd * (g' - s1 - xhat * s2)
__global__ __launch_bounds__(256) void k_in_bwd_apply(const float* __restrict__ g, const float* __restrict__ t,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ s12, float* __restrict__ dt, int N,
                                                      int H, int W, int C, int epi_act, int post_act, int shuffle,
                                                      unsigned long long* amax_slot, uint32_t amax_gen) {
  __shared__ uint32_t amax_red[4];
  uint32_t amax = 0;
  const int C4 = C / 4;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const size_t pixg = idx / C4;
    const int n = (int)(pixg / ((size_t)H * W));
    float4 v = *reinterpret_cast<const float4*>(t + pixg * C + c4 * 4);
    float4 mu = *reinterpret_cast<const float4*>(mean + (size_t)n * C + c4 * 4);
    float4 rs = *reinterpret_cast<const float4*>(rstd + (size_t)n * C + c4 * 4);
    const float* sp = s12 + ((size_t)n * C + c4 * 4) * 2;
    float4 gv;
    if (shuffle) {
      const int pix = (int)(pixg - (size_t)n * H * W);
      const int h = pix / W, w = pix - h * W;
      const float* gp = g + (((size_t)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * C4 + c4;
      gv.x = gp[0];
      gv.y = gp[C4];
      gv.z = gp[(size_t)2 * W * C4];
      gv.w = gp[(size_t)2 * W * C4 + C4];
    } else {
      gv = *reinterpret_cast<const float4*>(g + pixg * C + c4 * 4);
    }
    float xh, gg;
    float4 o;
    xh = (v.x - mu.x) * rs.x; gg = gv.x * act_grad_from_in(xh, post_act);
    o.x = act_grad_from_out(v.x, epi_act) * rs.x * (gg - sp[0] - xh * sp[1]);
    xh = (v.y - mu.y) * rs.y; gg = gv.y * act_grad_from_in(xh, post_act);
    o.y = act_grad_from_out(v.y, epi_act) * rs.y * (gg - sp[2] - xh * sp[3]);
    xh = (v.z - mu.z) * rs.z; gg = gv.z * act_grad_from_in(xh, post_act);
    o.z = act_grad_from_out(v.z, epi_act) * rs.z * (gg - sp[4] - xh * sp[5]);
    xh = (v.w - mu.w) * rs.w; gg = gv.w * act_grad_from_in(xh, post_act);
    o.w = act_grad_from_out(v.w, epi_act) * rs.w * (gg - sp[6] - xh * sp[7]);
    *reinterpret_cast<float4*>(dt + pixg * C + c4 * 4) = o;
    amax = max(amax, vcg_abs_bits4(o));
  }
  if (amax_slot) vcg_amax_publish(amax, amax_slot, amax_gen, amax_red);
}

// The same dt, by the workgroups of k_in_partial (one pixel chunk x TC channel quads of one image), which lets the kernel leave
// the COLUMN SUMS of what it writes: the bias gradient of the conv in front of this InstanceNorm is sum over pixels of dt
// (conv -> ReLU -> IN blocks: D, U, R.conv1 — /root/reference/Networks.py:93-95), which used to be a pass of its own over dt
// (k_colsum_partial: 64 launches and ~2 GB per CycleVAEGAN step).  colpart[(n * nchunk + chunk) * C + c], summed in a fixed
// order by k_in_colsum_final.
__global__ __launch_bounds__(256) void k_in_bwd_apply_cs(const float* __restrict__ g, const float* __restrict__ t,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ s12, float* __restrict__ dt,
                                                         float* __restrict__ colpart, int H, int W, int C, NormPlan pl, int epi_act,
                                                         int post_act, int shuffle, unsigned long long* amax_slot, uint32_t amax_gen) {
  __shared__ float4 red[256];
  __shared__ uint32_t amax_red[4];
  uint32_t amax = 0;
  const int tc = threadIdx.x % pl.TC, tp = threadIdx.x / pl.TC;
  const int c4 = blockIdx.z * pl.TC + tc;
  const int n = blockIdx.y;
  const int HW = H * W;
  const int pb = blockIdx.x * pl.chunk;
  int pe = pb + pl.chunk;
  if (pe > HW) pe = HW;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < C) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + (size_t)n * C + c4 * 4);
    const float4 rs = *reinterpret_cast<const float4*>(rstd + (size_t)n * C + c4 * 4);
    const float* sp = s12 + ((size_t)n * C + c4 * 4) * 2;
    const float4 sa = *reinterpret_cast<const float4*>(sp), sb = *reinterpret_cast<const float4*>(sp + 4);   // (s1, s2) of channels 0, 1 | 2, 3
    const int C4 = C / 4;
    for (int pix = pb + tp; pix < pe; pix += pl.TP) {
      const size_t off = ((size_t)n * HW + pix) * C + c4 * 4;
      const float4 v = *reinterpret_cast<const float4*>(t + off);
      float4 gv;
      if (shuffle) {
        const int h = pix / W, w = pix - h * W;
        const float* gp = g + (((size_t)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * C4 + c4;
        gv.x = gp[0];
        gv.y = gp[C4];
        gv.z = gp[(size_t)2 * W * C4];
        gv.w = gp[(size_t)2 * W * C4 + C4];
      } else {
        gv = *reinterpret_cast<const float4*>(g + off);
      }
      float xh, gg;
      float4 o;
      xh = (v.x - mu.x) * rs.x; gg = gv.x * act_grad_from_in(xh, post_act);
      o.x = act_grad_from_out(v.x, epi_act) * rs.x * (gg - sa.x - xh * sa.y);
      xh = (v.y - mu.y) * rs.y; gg = gv.y * act_grad_from_in(xh, post_act);
      o.y = act_grad_from_out(v.y, epi_act) * rs.y * (gg - sa.z - xh * sa.w);
      xh = (v.z - mu.z) * rs.z; gg = gv.z * act_grad_from_in(xh, post_act);
      o.z = act_grad_from_out(v.z, epi_act) * rs.z * (gg - sb.x - xh * sb.y);
      xh = (v.w - mu.w) * rs.w; gg = gv.w * act_grad_from_in(xh, post_act);
      o.w = act_grad_from_out(v.w, epi_act) * rs.w * (gg - sb.z - xh * sb.w);
      *reinterpret_cast<float4*>(dt + off) = o;
      amax = max(amax, vcg_abs_bits4(o));
      cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
    }
  }
  red[threadIdx.x] = cs;
  __syncthreads();
  if (tp == 0 && c4 * 4 < C) {
    for (int k = 1; k < pl.TP; ++k) {
      const float4 r = red[k * pl.TC + tc];
      cs.x += r.x; cs.y += r.y; cs.z += r.z; cs.w += r.w;
    }
    if (colpart) *reinterpret_cast<float4*>(colpart + ((size_t)n * pl.nchunk + blockIdx.x) * C + c4 * 4) = cs;
  }
  if (amax_slot) vcg_amax_publish(amax, amax_slot, amax_gen, amax_red);
}
// gbias[c] += sum over the nrows partial rows of colpart[row][c], in a fixed order: 8 channels x 32 row lanes per block
__global__ __launch_bounds__(256) void k_in_colsum_final(const float* __restrict__ part, float* __restrict__ out, int C, int nrows, int c_log) {
  __shared__ float red[32][8];
  const int il = threadIdx.x & 7, kl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + il;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < c_log) {
    int k = kl;
    for (; k + 96 < nrows; k += 128) {
      s0 += part[(size_t)k * C + c];
      s1 += part[(size_t)(k + 32) * C + c];
      s2 += part[(size_t)(k + 64) * C + c];
      s3 += part[(size_t)(k + 96) * C + c];
    }
    for (; k < nrows; k += 32) s0 += part[(size_t)k * C + c];
  }
  red[kl][il] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kl == 0 && c < c_log) {
    float s = red[0][il];
    for (int k = 1; k < 32; ++k) s += red[k][il];
    // atomic: the two translation directions of a cycle model differentiate the SAME generator on two streams
    // (ops.DirectionFork); with two contributions into a zeroed buffer the sum does not depend on their order
    atomicAdd(out + c, s);
  }
}

__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ g, const float* __restrict__ t,
                                                 float* __restrict__ dt, size_t n4, int act, unsigned long long* amax_slot,
                                                 uint32_t amax_gen) {
  __shared__ uint32_t amax_red[4];
  uint32_t amax = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 tv = reinterpret_cast<const float4*>(t)[i];
    float4 o;
    o.x = gv.x * act_grad_from_out(tv.x, act);
    o.y = gv.y * act_grad_from_out(tv.y, act);
    o.z = gv.z * act_grad_from_out(tv.z, act);
    o.w = gv.w * act_grad_from_out(tv.w, act);
    reinterpret_cast<float4*>(dt)[i] = o;
    amax = max(amax, vcg_abs_bits4(o));
  }
  if (amax_slot) vcg_amax_publish(amax, amax_slot, amax_gen, amax_red);
}

static int ew_blocks(size_t work) {
  size_t b = (work + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" size_t vcg_in_workspace(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0 || C % 4) return 0;
  NormPlan pl = make_plan(N, HW, C);
  return (size_t)N * pl.nchunk * C * 2 * sizeof(double) + (size_t)N * C * 2 * sizeof(float) + (size_t)N * pl.nchunk * C * sizeof(float) + 512;
}

// chunk partials [N][nchunk][C][2] (sum, sum of squares, in double) -> mean, rstd
int vcg_in_finalize(const double* part, float* mean, float* rstd, int N, int HW, int C, int nchunk, float eps, hipStream_t st) {
  hipLaunchKernelGGL(k_in_final<0>, dim3((N * C + 31) / 32), dim3(256), 0, st, part, mean, rstd, N, HW, C, nchunk, eps);
  VCG_LAUNCH_CHECK("vcg_in_finalize");
  return 0;
}
int vcg_in_stats_pass(const float* t, float* mean, float* rstd, int N, int HW, int C, float eps, void* ws, size_t ws_bytes, hipStream_t st) {
  VCG_CHECK_ARG(t && mean && rstd && ws, "vcg_in_stats: null pointer");
  VCG_CHECK_ARG(N > 0 && HW > 0 && C > 0 && C % 4 == 0, "vcg_in_stats: bad dims N=%d HW=%d C=%d", N, HW, C);
  VCG_CHECK_ARG(ws_bytes >= vcg_in_workspace(N, HW, C), "vcg_in_stats: workspace too small");
  NormPlan pl = make_plan(N, HW, C);
  double* part = (double*)ws;
  VcgInTail tail = vcg_in_tail_make(mean, rstd, N * pl.cgroups, HW, eps);
  if (tail.out1)
    hipLaunchKernelGGL((k_in_partial<0, true>), dim3(pl.nchunk, N, pl.cgroups), dim3(256), 0, st, t, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, part, HW, 1, C, pl, 0, 0, tail);
  else
    hipLaunchKernelGGL((k_in_partial<0, false>), dim3(pl.nchunk, N, pl.cgroups), dim3(256), 0, st, t, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, part, HW, 1, C, pl, 0, 0, tail);
  VCG_LAUNCH_CHECK("vcg_in_stats");
  return tail.out1 ? 0 : vcg_in_finalize(part, mean, rstd, N, HW, C, pl.nchunk, eps, st);
}
extern "C" int vcg_in_stats(const float* t, float* mean, float* rstd, int N, int HW, int C, float eps,
                            void* ws, size_t ws_bytes, void* stream) {
  return vcg_in_stats_pass(t, mean, rstd, N, HW, C, eps, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int vcg_in_apply(const float* t, const float* mean, const float* rstd, const float* residual,
                            float* out, int N, int H, int W, int C, int post_act, int shuffle, void* stream) {
  VCG_CHECK_ARG(t && mean && rstd && out, "vcg_in_apply: null pointer");
  VCG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "vcg_in_apply: bad dims");
  VCG_CHECK_ARG(!(shuffle && residual), "vcg_in_apply: shuffle with residual unsupported");
  VCG_CHECK_ARG(!shuffle || C % 16 == 0, "vcg_in_apply: pixel shuffle needs C %% 16 == 0 (channel pitch stays a multiple of 4)");
  size_t total = (size_t)N * H * W * (C / 4);
  const VcgAmaxOut ao = vcg_amax_new((hipStream_t)stream);
  // (a (chunk, image, channel group) variant of this kernel, as vcg_in_bwd uses, was measured in round 4: 19.5 vs 19.1 us — no gain)
  hipLaunchKernelGGL(k_in_apply, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, t, mean, rstd, residual,
                     out, N, H, W, C, post_act, shuffle, ao.slot, ao.gen);
  VCG_LAUNCH_CHECK("vcg_in_apply");
  vcg_set_last_amax(vcg_amax_handle(ao));               // vcg_amax_last(): the largest magnitude of `out`
  return 0;
}

static int in_bwd_impl(const float* g, const float* t, const float* mean, const float* rstd, float* dt,
                       int N, int H, int W, int C, int epi_act, int post_act, int shuffle, float* gbias, int c_log,
                       void* ws, size_t ws_bytes, void* stream) {
  VCG_CHECK_ARG(g && t && mean && rstd && dt && ws, "vcg_in_bwd: null pointer");
  VCG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "vcg_in_bwd: bad dims");
  VCG_CHECK_ARG(!shuffle || C % 16 == 0, "vcg_in_bwd: pixel shuffle needs C %% 16 == 0");
  const int HW = H * W;
  VCG_CHECK_ARG(ws_bytes >= vcg_in_workspace(N, HW, C), "vcg_in_bwd: workspace too small");
  NormPlan pl = make_plan(N, HW, C);
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  float* s12 = (float*)(part + (size_t)N * pl.nchunk * C * 2);
  float* colpart = s12 + (size_t)N * C * 2 + 64;
  colpart = (float*)(((uintptr_t)colpart + 15) & ~(uintptr_t)15);
  VcgInTail tail = vcg_in_tail_make(s12, nullptr, N * pl.cgroups, HW, 0.f);
  if (tail.out1)
    hipLaunchKernelGGL((k_in_partial<1, true>), dim3(pl.nchunk, N, pl.cgroups), dim3(256), 0, st, t, g, mean, rstd, part, H, W,
                       C, pl, post_act, shuffle, tail);
  else
    hipLaunchKernelGGL((k_in_partial<1, false>), dim3(pl.nchunk, N, pl.cgroups), dim3(256), 0, st, t, g, mean, rstd, part, H, W,
                       C, pl, post_act, shuffle, tail);
  if (!tail.out1)
    hipLaunchKernelGGL(k_in_final<1>, dim3((N * C + 31) / 32), dim3(256), 0, st, (const double*)part, s12,
                       (float*)nullptr, N, HW, C, pl.nchunk, 0.f);
  size_t total = (size_t)N * HW * (C / 4);
  const VcgAmaxOut ao = vcg_amax_new(st);
  // VCG_IN_BWD_FLAT=1: the flat grid-stride kernel where no bias gradient is wanted, as before (A/B measurements).  Round 4: the
  // (chunk, image, channel group) workgroups of the column-sum variant hold mean / rstd / s12 in registers and moved 4.97 TB/s
  // where the flat kernel, which reloads them per element, moved 3.48 (profiles/r04_pmc_step_traffic.txt) — every layer takes them now
  static const int flat = [] { const char* e = getenv("VCG_IN_BWD_FLAT"); return e ? atoi(e) : 0; }();
  if (gbias || !flat) {
    // dt (and, with gbias, its column sums: the bias gradient of the conv in front of this norm) in one pass
    hipLaunchKernelGGL(k_in_bwd_apply_cs, dim3(pl.nchunk, N, pl.cgroups), dim3(256), 0, st, g, t, mean, rstd, (const float*)s12, dt,
                       gbias ? colpart : (float*)nullptr, H, W, C, pl, epi_act, post_act, shuffle, ao.slot, ao.gen);
    if (gbias)
      hipLaunchKernelGGL(k_in_colsum_final, dim3((c_log + 7) / 8), dim3(256), 0, st, (const float*)colpart, gbias, C, N * pl.nchunk, c_log);
  } else {
    hipLaunchKernelGGL(k_in_bwd_apply, dim3(ew_blocks(total)), dim3(256), 0, st, g, t, mean, rstd, (const float*)s12,
                       dt, N, H, W, C, epi_act, post_act, shuffle, ao.slot, ao.gen);
  }
  VCG_LAUNCH_CHECK("vcg_in_bwd");
  vcg_set_last_amax(vcg_amax_handle(ao));               // vcg_amax_last(): the largest magnitude of `dt`
  return 0;
}
extern "C" int vcg_in_bwd(const float* g, const float* t, const float* mean, const float* rstd, float* dt,
                          int N, int H, int W, int C, int epi_act, int post_act, int shuffle,
                          void* ws, size_t ws_bytes, void* stream) {
  return in_bwd_impl(g, t, mean, rstd, dt, N, H, W, C, epi_act, post_act, shuffle, nullptr, 0, ws, ws_bytes, stream);
}
// vcg_in_bwd that also ACCUMULATES the column sums of dt into gbias[0 .. c_log) — the bias gradient of the convolution whose
// output this InstanceNorm normalises, when an activation sits between them (otherwise it is identically zero) — instead of a
// separate pass over dt inside vcg_conv_wgrad (hand that call gbias = NULL then).  dt_amax as in vcg_in_bwd_h (may be NULL).
extern "C" int vcg_in_bwd_bias(const float* g, const float* t, const float* mean, const float* rstd, float* dt,
                               int N, int H, int W, int C, int epi_act, int post_act, int shuffle, float* gbias, int c_log,
                               void* ws, size_t ws_bytes, uint64_t* dt_amax, void* stream) {
  VCG_CHECK_ARG(gbias && c_log > 0 && c_log <= C, "vcg_in_bwd_bias: bad bias gradient arguments");
  const int rc = in_bwd_impl(g, t, mean, rstd, dt, N, H, W, C, epi_act, post_act, shuffle, gbias, c_log, ws, ws_bytes, stream);
  const uint64_t h = vcg_amax_last();
  if (dt_amax) *dt_amax = rc ? 0 : h;
  return rc;
}

extern "C" int vcg_act_bwd(const float* g, const float* t, float* dt, size_t n, int act, void* stream) {
  VCG_CHECK_ARG(g && t && dt, "vcg_act_bwd: null pointer");
  VCG_CHECK_ARG(n % 4 == 0, "vcg_act_bwd: n must be a multiple of 4");
  const VcgAmaxOut ao = vcg_amax_new((hipStream_t)stream);
  hipLaunchKernelGGL(k_act_bwd, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, g, t, dt, n / 4, act, ao.slot, ao.gen);
  VCG_LAUNCH_CHECK("vcg_act_bwd");
  vcg_set_last_amax(vcg_amax_handle(ao));
  return 0;
}

// nn.PixelShuffle(2) (Networks.py:121) as a standalone copy, and its inverse (the backward).
// (N,H,W,C) <-> (N,2H,2W,C/4): channel 4c+2i+j of pixel (h,w) <-> channel c of pixel (2h+i, 2w+j)
__global__ __launch_bounds__(256) void k_pixel_shuffle(const float* __restrict__ src, float* __restrict__ dst, int N,
                                                       int H, int W, int C, int inverse) {
  const int C4 = C / 4;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const size_t pixg = idx / C4;
    const int n = (int)(pixg / ((size_t)H * W));
    const int pix = (int)(pixg - (size_t)n * H * W);
    const int h = pix / W, w = pix - h * W;
    const size_t big = (((size_t)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * C4 + c4;
    const size_t small = pixg * C + c4 * 4;
    if (!inverse) {
      float4 v = *reinterpret_cast<const float4*>(src + small);
      dst[big] = v.x;
      dst[big + C4] = v.y;
      dst[big + (size_t)2 * W * C4] = v.z;
      dst[big + (size_t)2 * W * C4 + C4] = v.w;
    } else {
      float4 v;
      v.x = src[big];
      v.y = src[big + C4];
      v.z = src[big + (size_t)2 * W * C4];
      v.w = src[big + (size_t)2 * W * C4 + C4];
      *reinterpret_cast<float4*>(dst + small) = v;
    }
  }
}
extern "C" int vcg_pixel_shuffle(const float* src, float* dst, int N, int H, int W, int C, int inverse, void* stream) {
  VCG_CHECK_ARG(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0,
                "vcg_pixel_shuffle: bad args (C=%d must be a multiple of 16)", C);
  size_t total = (size_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(k_pixel_shuffle, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, H, W, C,
                     inverse);
  VCG_LAUNCH_CHECK("vcg_pixel_shuffle");
  return 0;
}
