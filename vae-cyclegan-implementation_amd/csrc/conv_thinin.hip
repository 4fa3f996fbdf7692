// Forward convolutions whose INPUT has at most 4 channels and whose output has 64: the 7x7 stem of the encoders
// (/root/reference/Networks.py:158, CaSb(3, 64, 7): 9.9 GF and a 134 MB output per call at batch 8) and the first 4x4 / stride-2
// layer of the discriminators (:244, CaSb(3, 64, 4, 2, 1, LeakyReLU, no norm)).
//
// As an implicit GEMM these layers have K = KH * KW * 4 with only 16 bytes per tap and pixel: the generic kernel's gather is one
// 16-byte load per (row, tap) — 49 of them per output pixel — and it ran at 69 TF (stem) / 22 TF (discriminator) with the matrix
// pipe ~10 % busy (profiles/r03_step_shapes.txt).  Here a workgroup owns a 16 x 16 block of output pixels and stages the input
// patch under it ONCE, as two fp16 pieces of 4 channels per pixel (8 bytes per pixel and piece).  With K ordered (kh, kw, c) and kw
// padded to 8, the 32-wide K block of a kernel row kh is 8 NEIGHBOURING pixels x 4 channels = 64 contiguous bytes of the patch
// row: the im2col matrix never exists, an A fragment (8 consecutive k = 2 pixels) is two ds_read_b64 at pixel
// (oy * S + kh) * SW + ox * S + 2 * chunk.  The whole weight matrix (64 x KH x 32, zero for kw >= KW) is split into LDS once
// per workgroup from the fp32 Wf block of the pack.  256 x 64 tile, 4 waves of 64 x 64 (2 x 2 accumulators of 32 x 32, hh and
// cross-term chains as everywhere), KH x 2 MFMA slices.  Epilogue: bias + activation, NHWC store, and — for the stem — the
// InstanceNorm partial sums of the block in double (the slab kernels' layout: [N][blocks per image][Cout][2]).
#include "vcg_common.h"

struct ThinInP {
  const float* x;           // (N, H, W, 4) fp32
  const float* wf;          // Wf[(kh, kw, c)][Cout] fp32 (the pack's first block)
  const float* bias;
  float* y;                 // (N, Ho, Wo, Cout)
  double* in_part;          // [N][nbx * nby][Cout][2] or null
  int N, H, W, Ho, Wo, Cout, cout_log, pad, reflect, act, nbx, nby;
  VcgAmax amax_a, amax_b;
};

template <int KH, int KW, int S>
__global__ __launch_bounds__(256, 2) void k_conv_thinin(ThinInP p) {
  constexpr int SH = 15 * S + KH, SW = 15 * S + 8, SPX = SH * SW;       // patch rows x (columns incl. the padded taps)
  constexpr int BN = 64;
  __shared__ __attribute__((aligned(16))) unsigned char As[VCG_NP][SPX * 8];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[VCG_NP][KH * 4 * BN * 16];    // [kh][chunk of 8 k][co][16 B]
  __shared__ double red[4 * BN * 2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  float sA, invA, sB, invB;
  vcg_scale_of(vcg_amax_bits(p.amax_a), p.amax_a.shift, sA, invA);
  vcg_scale_of(vcg_amax_bits(p.amax_b), p.amax_b.shift, sB, invB);
  int blk = blockIdx.x;
  const int n = blk / (p.nbx * p.nby);
  blk -= n * p.nbx * p.nby;
  const int by = blk / p.nbx, bx = blk - by * p.nbx;
  const int oy0 = by * 16, ox0 = bx * 16;

  // ---- weights: Wf[(kh * KW + kw) * 4 + c][co] -> Bs[piece][kh][kw / 2][co][(kw & 1) * 4 + c].  KH * 2 items per thread, every
  // load issued before the first conversion (as a rolled loop this was a chain of KH * 2 L2 round trips: ~20 us per workgroup)
  {
    constexpr int NIT = KH * 8 * BN / 256;
    float4 w[NIT];
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
      const int it = tid + 256 * r;
      const int co = it % BN, q8 = it / BN, kw = q8 & 7, kh = q8 >> 3;
      w[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kw < KW && co < p.Cout) {
        const float* q = p.wf + (size_t)((kh * KW + kw) * 4) * p.Cout + co;
        w[r] = make_float4(q[0], q[p.Cout], q[2 * (size_t)p.Cout], q[3 * (size_t)p.Cout]);
      }
    }
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
      const int it = tid + 256 * r;
      const int co = it % BN, q8 = it / BN, kw = q8 & 7, kh = q8 >> 3;
      uint2 h, l;
      split4h(w[r], invB, h, l);
      const uint32_t o = (uint32_t)((((kh * 4 + (kw >> 1)) * BN + co) * 16) + (kw & 1) * 8);
      *reinterpret_cast<uint2*>(&Bs[0][o]) = h;
      *reinterpret_cast<uint2*>(&Bs[1][o]) = l;
    }
  }
  // ---- input patch: pixel (sy, sx) = input (oy0 * S - pad + sy, ox0 * S - pad + sx), reflected / zero outside
  const float* xn = p.x + (size_t)n * p.H * p.W * 4;
  {
    constexpr int NPX = (SPX + 255) / 256;
    float4 v[NPX];
#pragma unroll
    for (int r = 0; r < NPX; ++r) {
      const int spx = tid + 256 * r;
      const int sy = spx / SW, sx = spx - sy * SW;
      int iy = oy0 * S - p.pad + sy, ix = ox0 * S - p.pad + sx;
      bool ok = spx < SPX;
      if (p.reflect) {
        // columns beyond the last real tap (the padded kw) may lie more than `pad` outside: they meet zero weights, any finite
        // value does — keep them inside the reflect map's range
        ok = ok && iy > -p.H && iy < 2 * p.H - 1 && ix > -p.W && ix < 2 * p.W - 1;
        iy = reflect_idx(iy, p.H);
        ix = reflect_idx(ix, p.W);
      } else {
        ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      }
      v[r] = ok ? *reinterpret_cast<const float4*>(xn + ((size_t)iy * p.W + ix) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < NPX; ++r) {
      const int spx = tid + 256 * r;
      if (spx < SPX) {
        uint2 h, l;
        split4h(v[r], invA, h, l);
        *reinterpret_cast<uint2*>(&As[0][spx * 8]) = h;
        *reinterpret_cast<uint2*>(&As[1][spx * 8]) = l;
      }
    }
  }
  __syncthreads();

  f32x16 acc[2][2], lo[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;
  // row m of the block = pixel (m >> 4, m & 15); this wave's rows 64 wid + 32 i + l31
  int abase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wid * 64 + i * 32 + l31;
    abase[i] = ((m >> 4) * S) * SW + (m & 15) * S;
  }
#pragma unroll 1
  for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 2 + lh;                         // 8 consecutive k = pixels 2 chunk, 2 chunk + 1 of the K block
      f16x8 a[VCG_NP][2], b[VCG_NP][2];
#pragma unroll
      for (int pc = 0; pc < VCG_NP; ++pc) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t o = (uint32_t)((abase[i] + kh * SW + 2 * chunk) * 8);
          const uint2 q0 = *reinterpret_cast<const uint2*>(&As[pc][o]);
          const uint2 q1 = *reinterpret_cast<const uint2*>(&As[pc][o + 8]);
          const uint4 q = make_uint4(q0.x, q0.y, q1.x, q1.y);
          a[pc][i] = __builtin_bit_cast(f16x8, q);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[pc][j] = *reinterpret_cast<const f16x8*>(&Bs[pc][(uint32_t)(((kh * 4 + chunk) * BN + j * 32 + l31) * 16)]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = lo[i][j];
          c = VCG_MFMA(a[1][i], b[0][j], c);
          c = VCG_MFMA(a[0][i], b[1][j], c);
          lo[i][j] = c;
          acc[i][j] = VCG_MFMA(a[0][i], b[0][j], acc[i][j]);
        }
    }
  }

  // ---- epilogue
  const float oscale = sA * sB;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = j * 32 + l31;
    const float bv = (p.bias && co < p.cout_log) ? p.bias[co] : 0.f;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int m = wid * 64 + i * 32 + row;
        const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
        const float v = act_apply((acc[i][j][e] + lo[i][j][e]) * oscale + bv, p.act);
        if (co < p.Cout) p.y[(((size_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout + co] = v;
        s1 += (double)v;
        s2 += (double)v * (double)v;
      }
    }
    if (p.in_part) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lh == 0) {
        red[(wid * BN + co) * 2] = s1;
        red[(wid * BN + co) * 2 + 1] = s2;
      }
    }
  }
  if (p.in_part) {
    __syncthreads();
    if (tid < BN && tid < p.Cout) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += red[(w * BN + tid) * 2];
        t2 += red[(w * BN + tid) * 2 + 1];
      }
      double* o = p.in_part + (((size_t)n * (p.nbx * p.nby) + blk) * p.Cout + tid) * 2;
      o[0] = t1;
      o[1] = t2;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------- host side
// VCG_THININ: 0 = the generic implicit-GEMM kernel for both layers, as in round 3; 1 (default) = the discriminators' first layer
// only; 2 = the stem as well.  The stem variant is as accurate as the kernel it replaces (tests/test_gpu_fullsize.py holds both to
// 2e-6 against float64) and 144 -> 110 us per call, but it rounds differently, and on the batch-1 GAN fixture that moved ONE
// gradient tensor (G.encoder.model.5.conv1.weight, whose reference fp32-vs-fp64 error is unusually small: 6.1e-3) from under
// to over its fixture-calibrated bound — 2.84e-2 against 4 x 6.1e-3 = 2.43e-2, the ReLU-flip floor of that step being ~3e-2 on
// its neighbours.  Rather than widen the bound for 0.14 ms per step, the stem keeps the round-3 kernel by default.
static int thinin_mode() {
  static const int m = [] { const char* e = getenv("VCG_THININ"); return e ? atoi(e) : 1; }();
  return m;
}
bool vcg_thinin_fwd_ok(const ConvGeom& g) {
  const int mode = thinin_mode();
  if (mode <= 0) return false;
  const bool stem = g.KH == 7 && g.KW == 7 && g.stride == 1 && g.pad == 3 && mode >= 2;
  const bool disc = g.KH == 4 && g.KW == 4 && g.stride == 2 && g.pad == 1;
  return (stem || disc) && g.ups == 1 && g.Cin == 4 && g.Cout == 64 && g.Ho % 16 == 0 && g.Wo % 16 == 0 && g.Ho > 0 && g.Wo > 0 &&
         (!g.reflect || (g.H > g.pad + 8 && g.W > g.pad + 8));
}
int vcg_thinin_nchunk(const ConvGeom& g) { return (g.Ho / 16) * (g.Wo / 16); }
int vcg_thinin_fwd(const ConvGeom& g, const float* x, const float* wf, const void* w_amax, const float* bias, float* y, double* in_part,
                   hipStream_t st, uint64_t x_handle) {
  ThinInP p = {};
  if (vcg_operand_amax(x, (size_t)g.N * g.H * g.W * g.Cin, x_handle, 0, st, &p.amax_a)) return -2;
  p.amax_b = vcg_amax_stored(w_amax);
  p.x = x; p.wf = wf; p.bias = bias; p.y = y; p.in_part = in_part;
  p.N = g.N; p.H = g.H; p.W = g.W; p.Ho = g.Ho; p.Wo = g.Wo; p.Cout = g.Cout; p.cout_log = g.cout_log;
  p.pad = g.pad; p.reflect = g.reflect; p.act = g.act; p.nbx = g.Wo / 16; p.nby = g.Ho / 16;
  const dim3 grid((unsigned)(g.N * p.nbx * p.nby));
  const double flops = 2.0 * g.M * (double)g.K * g.Cout;
  if (g.KH == 7) {
    VcgProfScope prof("k_conv_thinin<7, 7, 1>", flops, st);
    hipLaunchKernelGGL((k_conv_thinin<7, 7, 1>), grid, dim3(256), 0, st, p);
  } else {
    VcgProfScope prof("k_conv_thinin<4, 4, 2>", flops, st);
    hipLaunchKernelGGL((k_conv_thinin<4, 4, 2>), grid, dim3(256), 0, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv_fwd(thin input)");
  return 0;
}
