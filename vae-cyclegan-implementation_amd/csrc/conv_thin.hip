// Direct convolution for the layers with <= 4 output channels per pixel: the decoder head
// CaSb(64, 3, k7) forward (Networks.py:192) and the data gradient of the encoder stem CaSb(3, 64, k7)
// (Networks.py:158), which CycleVAEGAN needs because F(G(x)) and G(F(y)) differentiate through their
// input images (Networks.py:1912, :1915).
//
// Why not the MFMA kernel: with N = 3 real columns a 32-wide MFMA tile does 10x the necessary work, and
// on gfx950 the fp32 MFMA rate equals the fp32 VALU rate (64 FLOP/clk/SIMD), so the matrix core buys
// nothing here.  Each lane owns ONE output pixel of a 16x16 tile and keeps its 3 (+1 pad) sums in
// registers; the (16+k-1)^2 input patch of a 16-channel chunk is staged once in LDS as four float4
// planes (lanes of a wave read consecutive 16-B slots: conflict-free ds_read_b128); the weights are the
// same for every lane, so they stream through the scalar unit (s_load into SGPRs) and feed v_fma as the
// scalar operand.  Per ds_read_b128: 12 useful FMAs.  HBM traffic = the activation once + 16 B/pixel out.
//
// Forward uses Wf[(tap, c)][4] as is.  The data gradient runs the same loop on dy with the taps flipped
// and weights read as Wf[(tap, ci)][co], over the PADDED domain (zero extension, no reflection), and
// k_fold_pad then folds the halo back: dx[q] = sum over {u : reflect(u) = q} of dxp[u] — the adjoint of
// reflect padding, exact and atomic-free.
#include "vcg_common.h"

struct ThinP {
  const float* x;      // (N, H, W, C) activations
  const float* w;      // Wf
  const float* bias;   // forward only (may be null)
  float* out;          // (N, Ho, Wo, 4)
  int N, H, W, C, Ho, Wo, K, pad, reflect, n_out, act;
};

#define TT 16   // output tile side
#define CC 16   // channels per LDS chunk

// MODE 0: forward      w(tap, c, j) = Wf[(tap*C + c)*4 + j]
// MODE 1: data grad    w(tap, c, j) = Wf[((K*K-1-tap)*4 + j)*C + c]   (taps flipped, c = dy channel)
template <int MODE>
__global__ __launch_bounds__(256) void k_conv_thin(ThinP p) {
  extern __shared__ __attribute__((aligned(16))) float4 patch[];   // [CC/4][PS*PS]
  const int PS = TT + p.K - 1, PP = PS * PS;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int ox0 = blockIdx.x * TT, oy0 = blockIdx.y * TT, n = blockIdx.z;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  const float* xn = p.x + (size_t)n * p.H * p.W * p.C;
  const int KK = p.K * p.K;

  for (int c0 = 0; c0 < p.C; c0 += CC) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < PP * (CC / 4); idx += 256) {
      const int j = idx & (CC / 4 - 1), pos = idx / (CC / 4);
      const int py = pos / PS, px = pos - py * PS;
      int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      bool ok;
      if (p.reflect) {
        ok = iy > -p.H && iy < 2 * p.H - 1 && ix > -p.W && ix < 2 * p.W - 1 && iy >= -p.pad && ix >= -p.pad &&
             iy <= p.H - 1 + p.pad && ix <= p.W - 1 + p.pad;
        if (ok) { iy = reflect_idx(iy, p.H); ix = reflect_idx(ix, p.W); }
      } else {
        ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      }
      if (ok) v = *reinterpret_cast<const float4*>(xn + ((size_t)iy * p.W + ix) * p.C + c0 + j * 4);
      patch[j * PP + pos] = v;
    }
    __syncthreads();
    for (int tap = 0; tap < KK; ++tap) {
      const int ky = tap / p.K, kx = tap - ky * p.K;
      const int pos = (ty + ky) * PS + tx + kx;
#pragma unroll
      for (int j = 0; j < CC / 4; ++j) {
        const float4 v = patch[j * PP + pos];
        const int c = c0 + j * 4;
        if (MODE == 0) {
          const float* wp = p.w + ((size_t)tap * p.C + c) * 4;          // wave-uniform -> scalar loads
          acc0 += v.x * wp[0] + v.y * wp[4] + v.z * wp[8] + v.w * wp[12];
          acc1 += v.x * wp[1] + v.y * wp[5] + v.z * wp[9] + v.w * wp[13];
          acc2 += v.x * wp[2] + v.y * wp[6] + v.z * wp[10] + v.w * wp[14];
        } else {
          const float* wp = p.w + ((size_t)(KK - 1 - tap) * 4) * p.C + c;
          acc0 += v.x * wp[0] + v.y * wp[1] + v.z * wp[2] + v.w * wp[3];
          const float* w1 = wp + p.C;
          acc1 += v.x * w1[0] + v.y * w1[1] + v.z * w1[2] + v.w * w1[3];
          const float* w2 = w1 + p.C;
          acc2 += v.x * w2[0] + v.y * w2[1] + v.z * w2[2] + v.w * w2[3];
        }
      }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy < p.Ho && ox < p.Wo) {
    float4 o = make_float4(acc0, acc1, acc2, acc3);
    if (p.bias) {
      if (p.n_out > 0) o.x += p.bias[0];
      if (p.n_out > 1) o.y += p.bias[1];
      if (p.n_out > 2) o.z += p.bias[2];
    }
    if (p.n_out < 3) o.z = 0.f;
    if (p.n_out < 2) o.y = 0.f;
    o.x = act_apply(o.x, p.act); o.y = act_apply(o.y, p.act); o.z = act_apply(o.z, p.act);
    *reinterpret_cast<float4*>(p.out + (((size_t)n * p.Ho + oy) * p.Wo + ox) * 4) = o;
  }
}

// dx[n,h,w,:] = sum over padded-domain coordinates that reflect onto (h,w) of dxp[n,u,v,:]
__global__ __launch_bounds__(256) void k_fold_pad(const float4* __restrict__ dxp, float4* __restrict__ dx, int N, int H,
                                                  int W, int pad, int reflect) {
  const size_t total = (size_t)N * H * W;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(idx % W);
    const size_t r = idx / W;
    const int h = (int)(r % H), n = (int)(r / H);
    int us[3], vs[3], nu = 0, nv = 0;
    us[nu++] = h + pad;
    vs[nv++] = w + pad;
    if (reflect) {
      if (h >= 1 && h <= pad) us[nu++] = pad - h;
      if (h >= H - 1 - pad && h <= H - 2) us[nu++] = 2 * (H - 1) - h + pad;
      if (w >= 1 && w <= pad) vs[nv++] = pad - w;
      if (w >= W - 1 - pad && w <= W - 2) vs[nv++] = 2 * (W - 1) - w + pad;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < nu; ++a)
      for (int b = 0; b < nv; ++b) {
        const float4 t = dxp[((size_t)n * Hp + us[a]) * Wp + vs[b]];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    dx[idx] = s;
  }
}

bool vcg_thin_fwd_ok(const ConvGeom& g) {
  return g.Cout == 4 && g.stride == 1 && g.ups == 1 && g.KH == g.KW && (g.KH & 1) && g.Cin % CC == 0 &&
         g.pad == g.KH / 2;
}
bool vcg_thin_dgrad_ok(const ConvGeom& g) {
  return g.Cin == 4 && g.stride == 1 && g.ups == 1 && g.KH == g.KW && (g.KH & 1) && g.Cout % CC == 0 &&
         g.pad == g.KH / 2;
}
size_t vcg_thin_dgrad_workspace(const ConvGeom& g) {
  return (size_t)g.N * (g.H + 2 * g.pad) * (g.W + 2 * g.pad) * 4 * sizeof(float) + 256;
}

// ---------------------------------------------------------------------------------------------------------------
// The forward of the thin layers on the matrix pipe after all: fold kw into the GEMM's N.
//   P[n][oh][pc][(kw, co)] = sum_{kh, c} xpad[n][oh + kh][pc][c] * w[co][c][kh][kw]        pc over the W + 2 pad padded columns
//   y[n][oh][ow][co]       = act(bias[co] + sum_kw P[n][oh][ow + kw][(kw, co)])
// The first line is a (KH x 1) convolution with KW * 4 (<= 32) output columns — 87 % of a 32-wide MFMA tile does
// useful work for 7 x 7 x 3 instead of 9 % — run by the forward implicit-GEMM kernel on its 128 x 32 tile; the
// second is a 17-float gather per output pixel.  4x faster than the VALU kernel above on the 64 -> 3 decoder head.
int vcg_fwd_launch(const ConvGeom& g, const float* x, const float* wf, float* y, hipStream_t st);
// conv_slab.hip: the same (KH x 1) convolution on the split-operand 16-bit pipe, the input rows staged once per workgroup
bool vcg_slab_col_ok(int KH, int C);
int vcg_slab_col(const float* x, const void* planes, size_t planes_bytes, const void* w_amax, float* P, int N, int H, int W, int C, int Ho, int Wo,
                 int KH, int pad, int reflect, hipStream_t st, uint64_t x_handle);
// Wk / Wkd are followed by their pre-split planes [32 rows (kw, co)][KH C / 32][VCG_NP pieces][32 k] when that kernel can take the layer
static size_t fold_planes_floats(int KH, int C) { return vcg_slab_col_ok(KH, C) ? (size_t)KH * C * VCG_PFLOATS : 0; }
// planes of a packed fold matrix wk[(kh, c)][32] (fp32): one thread per (row n, 4 consecutive k)
__global__ __launch_bounds__(256) void k_fold_planes(const float* __restrict__ wk, unsigned short* __restrict__ bp, int K, VcgAmax amax) {
  float sc, inv;                                                   // the planes hold wk / s, s from the kernel's amax (vcg_common.h)
  vcg_scale_of(vcg_amax_bits(amax), amax.shift, sc, inv);
  const int total = 32 * (K / 4);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int n = idx / (K / 4), k0 = (idx - n * (K / 4)) * 4;
    const float4 v = make_float4(wk[(size_t)(k0 + 0) * 32 + n], wk[(size_t)(k0 + 1) * 32 + n], wk[(size_t)(k0 + 2) * 32 + n],
                                 wk[(size_t)(k0 + 3) * 32 + n]);
    uint2 h, l;
    split4h(v, inv, h, l);
    unsigned short* o = bp + ((size_t)n * (K / 32) + k0 / 32) * VCG_PBLK + (k0 & 31);
    *reinterpret_cast<uint2*>(o) = h;
    *reinterpret_cast<uint2*>(o + 32) = l;
  }
}
static int pack_fold_planes(float* wk, int KH, int C, const VcgAmax& amax_w, hipStream_t st) {
  if (!fold_planes_floats(KH, C)) return 0;
  const int K = KH * C;
  hipLaunchKernelGGL(k_fold_planes, dim3((32 * (K / 4) + 255) / 256), dim3(256), 0, st, (const float*)wk,
                     (unsigned short*)(wk + (size_t)K * 32), K, amax_w);
  VCG_LAUNCH_CHECK("vcg_pack_weight(kw-fold planes)");
  return 0;
}

bool vcg_thin_fold_ok(const ConvGeom& g) { return vcg_thin_fwd_ok(g) && g.KW * 4 <= 32 && g.cout_log <= 3; }
size_t vcg_thin_fold_weight_floats(const ConvGeom& g) { return (size_t)g.KH * g.Cin * 32 + fold_planes_floats(g.KH, g.Cin); }
size_t vcg_thin_fold_workspace(const ConvGeom& g) {
  return (size_t)g.N * g.Ho * (g.W + 2 * g.pad) * 32 * sizeof(float) + 256;
}
// Wk[(kh, c)][kw * 4 + co] = w[co][c][kh][kw]
__global__ __launch_bounds__(256) void k_pack_kwfold(const float* __restrict__ w, float* __restrict__ wk, int C, int cin_log,
                                                     int cout_log, int KH, int KW) {
  const int total = KH * C * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int col = idx & 31, row = idx >> 5;
    const int kh = row / C, c = row - kh * C;
    const int kw = col >> 2, co = col & 3;
    float v = 0.f;
    if (kw < KW && co < cout_log && c < cin_log) v = w[(((size_t)co * cin_log + c) * KH + kh) * KW + kw];
    wk[idx] = v;
  }
}
// the data gradient of a thin-INPUT layer (the 3 -> 64 stem) is the same computation on dy with the kernel flipped and
// the channel roles swapped:  Wkd[(kh, co)][kw * 4 + ci] = w[co][ci][KH-1-kh][KW-1-kw]
__global__ __launch_bounds__(256) void k_pack_kwfold_dgrad(const float* __restrict__ w, float* __restrict__ wk, int Cout,
                                                           int cin_log, int cout_log, int KH, int KW) {
  const int total = KH * Cout * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int col = idx & 31, row = idx >> 5;
    const int kh = row / Cout, co = row - kh * Cout;
    const int kw = col >> 2, ci = col & 3;
    float v = 0.f;
    if (kw < KW && ci < cin_log && co < cout_log) v = w[(((size_t)co * cin_log + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
    wk[idx] = v;
  }
}
bool vcg_thin_fold_dgrad_ok(const ConvGeom& g) { return vcg_thin_dgrad_ok(g) && g.KW * 4 <= 32 && g.cin_log <= 3; }
size_t vcg_thin_fold_dgrad_weight_floats(const ConvGeom& g) { return (size_t)g.KH * g.Cout * 32 + fold_planes_floats(g.KH, g.Cout); }
int vcg_thin_fold_dgrad_pack(const ConvGeom& g, const float* w_oihw, float* wk, const VcgAmax& amax_w, hipStream_t st) {
  const int total = g.KH * g.Cout * 32;
  hipLaunchKernelGGL(k_pack_kwfold_dgrad, dim3((total + 255) / 256), dim3(256), 0, st, w_oihw, wk, g.Cout, g.cin_log,
                     g.cout_log, g.KH, g.KW);
  VCG_LAUNCH_CHECK("vcg_pack_weight(kw-fold dgrad)");
  return pack_fold_planes(wk, g.KH, g.Cout, amax_w, st);
}
// geometry of the full correlation over the padded domain: dy (N, Ho, Wo, Cout) -> dxp (N, H + 2 pad, W + 2 pad, 4)
static ConvGeom thin_dgrad_as_fwd(const ConvGeom& g) {
  ConvGeom f = g;
  f.H = g.Ho; f.W = g.Wo; f.Hl = g.Ho; f.Wl = g.Wo; f.Cin = g.Cout; f.cin_log = g.cout_log; f.Cout = 4; f.cout_log = g.cin_log;
  f.pad = 2 * g.pad; f.reflect = 0; f.act = VCG_ACT_NONE;
  f.Ho = g.H + 2 * g.pad; f.Wo = g.W + 2 * g.pad;
  f.M = f.N * f.Ho * f.Wo; f.K = f.KH * f.KW * f.Cin; f.taps = f.KH * f.KW;
  return f;
}
size_t vcg_thin_fold_dgrad_workspace(const ConvGeom& g) {
  const ConvGeom f = thin_dgrad_as_fwd(g);
  return vcg_thin_dgrad_workspace(g) + 256 + vcg_thin_fold_workspace(f);
}

int vcg_thin_fold_pack(const ConvGeom& g, const float* w_oihw, float* wk, const VcgAmax& amax_w, hipStream_t st) {
  const int total = g.KH * g.Cin * 32;
  hipLaunchKernelGGL(k_pack_kwfold, dim3((total + 255) / 256), dim3(256), 0, st, w_oihw, wk, g.Cin, g.cin_log, g.cout_log, g.KH,
                     g.KW);
  VCG_LAUNCH_CHECK("vcg_pack_weight(kw-fold)");
  return pack_fold_planes(wk, g.KH, g.Cin, amax_w, st);
}
__global__ __launch_bounds__(256) void k_kwfold_sum(const float* __restrict__ P, const float* __restrict__ bias,
                                                    float* __restrict__ y, int N, int H, int W, int Wp, int KW, int n_out,
                                                    int act) {
  const size_t total = (size_t)N * H * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int ow = (int)(idx % W);
    const size_t row = idx / W;                                  // n * H + oh
    const float* pr = P + (row * Wp + ow) * 32;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kw = 0; kw < KW; ++kw) {
      const float4 v = *reinterpret_cast<const float4*>(pr + (size_t)kw * 32 + kw * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
      if (n_out > 0) s.x += bias[0];
      if (n_out > 1) s.y += bias[1];
      if (n_out > 2) s.z += bias[2];
    }
    s.w = 0.f;
    if (n_out < 3) s.z = 0.f;
    if (n_out < 2) s.y = 0.f;
    s.x = act_apply(s.x, act); s.y = act_apply(s.y, act); s.z = act_apply(s.z, act);
    *reinterpret_cast<float4*>(y + idx * 4) = s;
  }
}
int vcg_thin_fold_fwd(const ConvGeom& g, const float* x, const float* wk, const void* w_amax, const float* bias, float* y, void* ws,
                      size_t ws_bytes, hipStream_t st, uint64_t x_handle) {
  VCG_CHECK_ARG(ws && ws_bytes >= vcg_thin_fold_workspace(g), "vcg_conv_fwd: workspace too small for the kw-folded path");
  ConvGeom q = g;                                 // the (KH x 1) convolution over every padded column
  q.KW = 1; q.Cout = 32; q.cout_log = g.KW * 4; q.act = VCG_ACT_NONE;
  q.Wo = g.W + 2 * g.pad; q.Ho = g.Ho; q.M = g.N * q.Ho * q.Wo; q.taps = g.KH; q.K = g.KH * g.Cin;
  if (fold_planes_floats(g.KH, g.Cin)) {
    if (vcg_slab_col(x, wk + (size_t)g.KH * g.Cin * 32, fold_planes_floats(g.KH, g.Cin) * 4, w_amax, (float*)ws, g.N, g.H, g.W, g.Cin, q.Ho,
                     q.Wo, g.KH, g.pad, g.reflect, st, x_handle))
      return -2;
  } else if (vcg_fwd_launch(q, x, wk, (float*)ws, st)) {
    return -2;
  }
  const size_t total = (size_t)g.N * g.Ho * g.Wo;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_kwfold_sum, dim3(blocks), dim3(256), 0, st, (const float*)ws, bias, y, g.N, g.Ho, g.Wo, q.Wo, g.KW,
                     g.cout_log < 3 ? g.cout_log : 3, g.act);
  VCG_LAUNCH_CHECK("vcg_conv_fwd(kw-fold sum)");
  return 0;
}

int vcg_thin_fwd(const ConvGeom& g, const float* x, const float* wf, const float* bias, float* y, hipStream_t st) {
  ThinP p;
  p.x = x; p.w = wf; p.bias = bias; p.out = y;
  p.N = g.N; p.H = g.H; p.W = g.W; p.C = g.Cin; p.Ho = g.Ho; p.Wo = g.Wo; p.K = g.KH; p.pad = g.pad;
  p.reflect = g.reflect; p.n_out = g.cout_log < 3 ? g.cout_log : 3; p.act = g.act;
  const int PS = TT + g.KH - 1;
  const size_t lds = (size_t)PS * PS * CC * sizeof(float);
  VCG_CHECK_ARG(lds <= 64 * 1024, "vcg_conv_fwd(thin): kernel %d too large for the LDS patch", g.KH);
  VCG_CHECK_ARG(g.cout_log <= 3, "vcg_conv_fwd(thin): at most 3 logical output channels");
  dim3 grid((g.Wo + TT - 1) / TT, (g.Ho + TT - 1) / TT, g.N);
  hipLaunchKernelGGL(k_conv_thin<0>, grid, dim3(256), lds, st, p);
  VCG_LAUNCH_CHECK("vcg_conv_fwd(thin)");
  return 0;
}

int vcg_thin_fold_dgrad(const ConvGeom& g, const float* dy, const float* wkd, const void* w_amax, float* dx, void* ws, size_t ws_bytes,
                        hipStream_t st, uint64_t dy_handle) {
  VCG_CHECK_ARG(ws && ws_bytes >= vcg_thin_fold_dgrad_workspace(g), "vcg_conv_dgrad(kw-fold): workspace too small");
  const ConvGeom f = thin_dgrad_as_fwd(g);
  float* dxp = (float*)ws;
  char* ws2 = (char*)ws + ((vcg_thin_dgrad_workspace(g) + 255) / 256) * 256;
  if (vcg_thin_fold_fwd(f, dy, wkd, w_amax, nullptr, dxp, ws2, ws_bytes - (size_t)(ws2 - (char*)ws), st, dy_handle)) return -2;
  size_t total = (size_t)g.N * g.H * g.W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_fold_pad, dim3(blocks), dim3(256), 0, st, (const float4*)dxp, (float4*)dx, g.N, g.H, g.W, g.pad,
                     g.reflect);
  VCG_LAUNCH_CHECK("vcg_conv_dgrad(kw-fold)");
  return 0;
}

int vcg_thin_dgrad(const ConvGeom& g, const float* dy, const float* wf, float* dx, void* ws, size_t ws_bytes,
                   hipStream_t st) {
  VCG_CHECK_ARG(ws && ws_bytes >= vcg_thin_dgrad_workspace(g), "vcg_conv_dgrad(thin): workspace too small");
  VCG_CHECK_ARG(g.cin_log <= 3, "vcg_conv_dgrad(thin): at most 3 logical input channels");
  ThinP p;
  p.x = dy; p.w = wf; p.bias = nullptr; p.out = (float*)ws;
  p.N = g.N; p.H = g.Ho; p.W = g.Wo; p.C = g.Cout; p.K = g.KH;
  p.pad = 2 * g.pad;                         // full correlation: zero-extend dy by k-1 on every side
  p.Ho = g.H + 2 * g.pad; p.Wo = g.W + 2 * g.pad;
  p.reflect = 0; p.n_out = g.cin_log; p.act = VCG_ACT_NONE;
  const int PS = TT + g.KH - 1;
  const size_t lds = (size_t)PS * PS * CC * sizeof(float);
  VCG_CHECK_ARG(lds <= 64 * 1024, "vcg_conv_dgrad(thin): kernel %d too large for the LDS patch", g.KH);
  dim3 grid((p.Wo + TT - 1) / TT, (p.Ho + TT - 1) / TT, g.N);
  hipLaunchKernelGGL(k_conv_thin<1>, grid, dim3(256), lds, st, p);
  size_t total = (size_t)g.N * g.H * g.W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_fold_pad, dim3(blocks), dim3(256), 0, st, (const float4*)ws, (float4*)dx, g.N, g.H, g.W, g.pad,
                     g.reflect);
  VCG_LAUNCH_CHECK("vcg_conv_dgrad(thin)");
  return 0;
}
