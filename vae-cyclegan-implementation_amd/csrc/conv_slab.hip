// 3x3 / stride-1 convolutions of the high-resolution, few-channel layers (U3, U4 of the decoders: 128^2 and 256^2 maps,
// 32..128 channels) on the split-operand 16-bit pipe, with the input staged ONCE per workgroup as an LDS slab.
//
// Why: the implicit-GEMM kernels (conv_igemm.hip) gather, split (fp32 -> two fp16 pieces; rounds 1-2: three bf16 ones) and store the im2col tile of
// every tap — each input value nine times — and with <= 64 output columns that staging outweighs the MFMAs: the
// 64-column tiles ran their matrix pipe 17-23 % busy (profiles/r01_pmc_mfma_util.txt).  Here a workgroup owns a PR x 16
// block of output pixels and, per 32-channel chunk, loads the (PR + 2) x 18 input pixels under it once: 180 (or 324)
// 16-byte gathers, splits and LDS stores instead of 9 x 128 (256).  The image has the same [row][32 fp16] swizzled
// layout as the GEMM kernels' A tile with "row" = slab pixel, so a tap is nothing but a row offset (kh * 18 + kw) in the
// fragment address.  The weight tile of a (chunk, tap) — BN rows x 32 k, pre-split planes from the pack — is copied
// global -> registers -> LDS under the previous tap's MFMAs, one barrier per tap.
//
// Forward (reflect or zero padding 1):  y = act(conv(x) + bias), optionally the InstanceNorm chunk partials of y.
// Data gradient: the same kernel on dy, zero-extended by 2, with the taps flipped and the weight planes of the data
//   gradient (WFD: rows = input channel, k = output channel): dxp over the padded domain (H + 2) x (W + 2), then
//   k_fold_pad_c adds the halo back onto the image (adjoint of reflect padding; a crop for zero padding).
#include "vcg_common.h"

typedef unsigned int sl_u32x4 __attribute__((ext_vector_type(4)));
#define SL_OOB 0x80000000u

struct SlabP {
  const float* in;          // (N, H, W, C) fp32
  const void* planes;       // weight planes, VCG_PBYTES per (row, 32-k block): VCG_NP pieces x 32 fp16 of w / sB (vcg_common.h)
  const float* bias;
  float* out;               // (N, Ho, Wo, Cout)
  double* in_part;          // InstanceNorm chunk partials [N][nchunk][Cout][2], or null
  VcgInTail in_tail;        // the last pixel block of an (image, column tile) combines them (vcg_common.h)
  int N, H, W, C, Ho, Wo, Cout, cout_log;
  int pad, reflect, act, tap_flip, nchunks, in_nchunk;
  int nbx, nby;             // pixel blocks per image
  uint32_t in_bytes, b_bytes, row_stride, tap_stride;
  VcgAmax amax_a, amax_b;   // largest magnitudes of the input tensor and of the kernel the planes were split from
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sl_srd(const void* ptr, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, (int)bytes, 0x00020000);
}

// WM x WN waves of 64 x 32 outputs: BM = 64 WM pixels (PR = 4 WM rows of 16), BN = 32 WN output channels; KH x KW taps
// (3 x 3; 7 x 1 for the kw-folded thin layers of conv_thin.hip, whose column taps live in the GEMM's N)
template <int WM, int WN, int KH, int KW>
__global__ __launch_bounds__(256, 2) void k_conv_slab(SlabP p) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int BM = 64 * WM, BN = 32 * WN, PR = BM / 16, SH = PR + KH - 1, SW = 16 + KW - 1, SPX = SH * SW, NTAP = KH * KW;
  constexpr int AQ = (SPX * 8 + 255) / 256;                  // slab quads per thread
  constexpr int CPR = 4 * VCG_NP;                            // 16-byte chunks per weight row and 32-k block
  constexpr int BQ = (BN * CPR + 255) / 256;                 // 16-byte weight chunks per thread and tap
  __shared__ __attribute__((aligned(16))) unsigned char As[VCG_NP][SPX * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][VCG_NP][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, lh = lane >> 5;
  float sA, invA, sB, invB;
  vcg_scale_of(vcg_amax_bits(p.amax_a), p.amax_a.shift, sA, invA);
  vcg_scale_of(vcg_amax_bits(p.amax_b), p.amax_b.shift, sB, invB);
  const float oscale = sA * sB;

  int blk = blockIdx.x;
  const int n = blk / (p.nbx * p.nby);
  blk -= n * p.nbx * p.nby;
  const int by = blk / p.nbx, bx = blk - by * p.nbx;
  const int oy0 = by * PR, ox0 = bx * 16;
  const int n0 = blockIdx.y * BN;

  const __amdgpu_buffer_rsrc_t ra = sl_srd(p.in, p.in_bytes), rb = sl_srd(p.planes, p.b_bytes);

  // this thread's slab quads: (slab pixel, 4-channel quad) -> global byte offset of channel chunk 0 and LDS byte offset
  uint32_t goff[AQ], soff[AQ];
#pragma unroll
  for (int a = 0; a < AQ; ++a) {
    const int idx = tid + 256 * a;
    const int spx = idx >> 3, q = idx & 7;
    const int sy = spx / SW, sx = spx - sy * SW;
    int iy = oy0 + sy - p.pad, ix = ox0 + sx - p.pad;
    bool ok = spx < SPX;
    if (p.reflect) {
      ok = ok && iy >= -p.pad && iy <= p.H - 1 + p.pad && ix >= -p.pad && ix <= p.W - 1 + p.pad;
      iy = reflect_idx(iy, p.H);
      ix = reflect_idx(ix, p.W);
    } else {
      ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    }
    goff[a] = ok ? (uint32_t)((((n * p.H + iy) * p.W + ix) * p.C + q * 4) * 4) : SL_OOB;
    soff[a] = spx < SPX ? (uint32_t)(spx * 64 + (((q >> 1) ^ ((spx >> 2) & 3)) << 4) + ((q & 1) << 3)) : SL_OOB;
  }
  // weight chunks: idx -> (row, piece, 16-byte quarter)
  uint32_t boff[BQ], bsoff[BQ];
  int bpc[BQ];
#pragma unroll
  for (int j = 0; j < BQ; ++j) {
    const int idx = tid + 256 * j;
    const int row = idx / CPR, r12 = idx - row * CPR, pc = r12 >> 2, q = r12 & 3;
    const bool ok = row < BN;
    boff[j] = ok ? (uint32_t)(n0 + row) * p.row_stride + (uint32_t)(r12 * 16) : SL_OOB;
    bsoff[j] = ok ? (uint32_t)(row * 64 + ((q ^ ((row >> 2) & 3)) << 4)) : SL_OOB;
    bpc[j] = pc;
  }

  f32x16 acc[2], lo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = lo[i][e] = 0.f;

  // fragment rows: M index m = wm * 64 + i * 32 + l31 -> block pixel (m >> 4, m & 15) -> slab pixel of tap (0, 0)
  int rbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wm * 64 + i * 32 + l31;
    rbase[i] = (m >> 4) * SW + (m & 15);
  }
  const int brow = wn * 32 + l31;
  const uint32_t fb = (uint32_t)(brow * 64);
  const int sb = (brow >> 2) & 3;

  // Weight tiles are fetched PD (chunk, tap) steps ahead into a ring of register sets (round 4).  With one step of look-ahead
  // (round 2) a step's MFMAs — 12 per wave, ~400 cycles — hid a fraction of the ~1 us an L2 / Infinity-Cache read takes, and the
  // layers with one or two channel chunks (U4: 9 steps in all) spent most of their time waiting for 8 KB weight tiles:
  // SQ_WAIT_ANY 54 % of the wave cycles (gpurun_out/thinin_pmc.txt).  PD divides the tap count, so the ring slot tap % PD is static.
  constexpr int PD = NTAP % 3 == 0 ? 3 : NTAP;
  sl_u32x4 vb[PD][BQ];
  const int total = p.nchunks * NTAP;
  auto load_b = [&](sl_u32x4 (&dst)[BQ], int step) {                  // step = chunk * NTAP + tap
    const int chunk = step / NTAP, tap = step - chunk * NTAP;
    const uint32_t t = (uint32_t)(p.tap_flip ? NTAP - 1 - tap : tap);
    const uint32_t o = t * p.tap_stride + (uint32_t)chunk * (uint32_t)VCG_PBYTES;
#pragma unroll
    for (int j = 0; j < BQ; ++j) dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(boff[j] != SL_OOB ? boff[j] + o : SL_OOB), 0, 0);
  };
  auto store_b = [&](const sl_u32x4 (&src)[BQ], int buf) {
#pragma unroll
    for (int j = 0; j < BQ; ++j)
      if (bsoff[j] != SL_OOB) *reinterpret_cast<sl_u32x4*>(&Bs[buf][bpc[j]][bsoff[j]]) = src[j];
  };
#pragma unroll
  for (int k = 0; k < PD; ++k)
    if (k < total) load_b(vb[k], k);

  int s = 0;                                              // linear (chunk, tap) step: B buffer = s & 1
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    // ---- the slab of this channel chunk: gather, split, store (the previous chunk's last MFMAs are behind a barrier)
    {
      float4 va[AQ];
#pragma unroll
      for (int a = 0; a < AQ; ++a) {
        const sl_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(goff[a] != SL_OOB ? goff[a] + (uint32_t)chunk * 128u : SL_OOB), 0, 0);
        va[a] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
      if (chunk > 0) __syncthreads();                     // every wave is done reading the previous slab
#pragma unroll
      for (int a = 0; a < AQ; ++a) {
        if (soff[a] == SL_OOB) continue;
        uint2 h, l;
        split4h(va[a], invA, h, l);
        *reinterpret_cast<uint2*>(&As[0][soff[a]]) = h;
        *reinterpret_cast<uint2*>(&As[1][soff[a]]) = l;
      }
      if (chunk == 0) {
        store_b(vb[0], 0);
        if (PD < total) load_b(vb[0], PD);
      }
      __syncthreads();
    }
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap, ++s) {
      const int buf = s & 1;
      const int kh = tap / KW, kw = tap - kh * KW;
      uint32_t fa[2];
      int sa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = rbase[i] + kh * SW + kw;
        fa[i] = (uint32_t)(r * 64);
        sa[i] = (r >> 2) & 3;
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        f16x8 a[VCG_NP][2], b[VCG_NP];
#pragma unroll
        for (int pc = 0; pc < VCG_NP; ++pc) {
#pragma unroll
          for (int i = 0; i < 2; ++i) a[pc][i] = *reinterpret_cast<const f16x8*>(&As[pc][fa[i] + (((2 * k2 + lh) ^ sa[i]) << 4)]);
          b[pc] = *reinterpret_cast<const f16x8*>(&Bs[buf][pc][fb + (((2 * k2 + lh) ^ sb) << 4)]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x16 c = lo[i];
          c = VCG_MFMA(a[1][i], b[0], c);                  // the cross terms in their own chain
          c = VCG_MFMA(a[0][i], b[1], c);
          lo[i] = c;
          acc[i] = VCG_MFMA(a[0][i], b[0], acc[i]);
        }
      }
      // the next step's tile (fetched PD steps ago) goes into the buffer step s - 1 read — every wave passed the barrier after it —
      // and its register set is handed to the fetch of step s + 1 + PD
      if (s + 1 < total) {
        constexpr int slot = 0;                            // (placeholder: the static slot is chosen by the switch below)
        (void)slot;
        const int nslot = (tap + 1) % PD;
#pragma unroll
        for (int k = 0; k < PD; ++k)
          if (k == nslot) {
            store_b(vb[k], buf ^ 1);
            if (s + 1 + PD < total) load_b(vb[k], s + 1 + PD);
          }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: bias + activation, NHWC store; optionally the InstanceNorm partials of this pixel block
  double* const red = reinterpret_cast<double*>(&As[0][0]);   // [wm][BN][2]; the last barrier freed As.  Sums in double: conv_igemm.hip
  const int cl = wn * 32 + l31, co = n0 + cl;
  const bool cv = co < p.Cout;
  const float bv = (cv && p.bias && co < p.cout_log) ? p.bias[co] : 0.f;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int m = wm * 64 + i * 32 + row;
      const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
      const float v = act_apply((acc[i][e] + lo[i][e]) * oscale + bv, p.act);
      if (cv && oy < p.Ho && ox < p.Wo) p.out[(((size_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout + co] = v;
      if (p.in_part) {
        s1 += (double)v;
        s2 += (double)v * (double)v;
      }
    }
  }
  if (p.in_part) {                      // uniform; the host sets it only when every block lies inside the image
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lh == 0) {
      red[(wm * BN + cl) * 2] = s1;
      red[(wm * BN + cl) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        t1 += red[(w * BN + tid) * 2];
        t2 += red[(w * BN + tid) * 2 + 1];
      }
      double* o = p.in_part + (((size_t)n * p.in_nchunk + blk) * p.Cout + n0 + tid) * 2;
      vcg_store_sc1(o, t1);
      vcg_store_sc1(o + 1, t2);
    }
    if (p.in_tail.out1) {
      __syncthreads();                // `red` (As) has been read
      vcg_in_tail_run<0>(p.in_tail, p.in_part, n, n0, BN, p.Cout, p.in_nchunk, p.in_tail.counters + n * gridDim.y + blockIdx.y,
                         (uint32_t)p.in_nchunk, reinterpret_cast<double*>(&As[0][0]));
    }
  }
}

// dx[n,h,w,:] = sum over the padded-domain coordinates that reflect onto (h,w) of dxp[n,u,v,:] (pad 1); zero padding: the crop
__global__ __launch_bounds__(256) void k_fold_pad_c(const float4* __restrict__ dxp, float4* __restrict__ dx, int N, int H, int W, int C4,
                                                    int reflect) {
  const size_t total = (size_t)N * H * W * C4;
  const int Hp = H + 2, Wp = W + 2;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C4);
    size_t r = idx / C4;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H), n = (int)(r / H);
    int us[2], vs[2], nu = 0, nv = 0;
    us[nu++] = h + 1;
    vs[nv++] = w + 1;
    if (reflect) {
      if (h == 1) us[nu++] = 0;
      if (h == H - 2) us[nu++] = H + 1;
      if (w == 1) vs[nv++] = 0;
      if (w == W - 2) vs[nv++] = W + 1;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < nu; ++a)
      for (int b = 0; b < nv; ++b) {
        const float4 t = dxp[(((size_t)n * Hp + us[a]) * Wp + vs[b]) * C4 + c];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    dx[idx] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------- host side
static bool slab_geom_ok(const ConvGeom& g) {
  return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && g.ups == 1 && g.Hl >= 3 && g.Wl >= 3;
}
// forward: Cin in 32-channel chunks (<= 4 of them: beyond that the GEMM-shaped kernels amortise their staging), 64-column tiles
bool vcg_slab_fwd_ok(const ConvGeom& g) {
  return slab_geom_ok(g) && g.Cin % 32 == 0 && g.Cin <= 128 && g.Cout % 64 == 0 && g.Cout <= 128 && (long long)g.Ho * g.Wo >= 64 * 64;
}
// every pixel block inside the image: the tile epilogue can leave the InstanceNorm partials
bool vcg_slab_fwd_stats_ok(const ConvGeom& g) { return vcg_slab_fwd_ok(g) && g.Ho % 8 == 0 && g.Wo % 16 == 0; }
int vcg_slab_fwd_nchunk(const ConvGeom& g) { return (g.Ho / 8) * (g.Wo / 16); }
// data gradient: k = Cout in 32-chunks, N = Cin as one 32- or 64-column tile per workgroup column
bool vcg_slab_dgrad_ok(const ConvGeom& g) {
  return slab_geom_ok(g) && g.Cout % 32 == 0 && g.Cout <= 128 && (g.Cin == 32 || g.Cin % 64 == 0) && g.Cin <= 128 &&
         (long long)g.H * g.W >= 64 * 64;
}
size_t vcg_slab_dgrad_workspace(const ConvGeom& g) { return (size_t)g.N * (g.H + 2) * (g.W + 2) * g.Cin * sizeof(float) + 256; }

int vcg_slab_fwd(const ConvGeom& g, const float* x, const void* wft_planes, size_t planes_bytes, const void* w_amax, const float* bias,
                 float* y, double* in_part, const VcgInTail* tail_req, hipStream_t st, uint64_t x_handle) {
  SlabP p = {};
  if (vcg_operand_amax(x, (size_t)g.N * g.H * g.W * g.Cin, x_handle, 0, st, &p.amax_a)) return -2;
  p.amax_b = vcg_amax_stored(w_amax);
  p.in = x; p.planes = wft_planes; p.bias = bias; p.out = y; p.in_part = in_part;
  p.N = g.N; p.H = g.H; p.W = g.W; p.C = g.Cin; p.Ho = g.Ho; p.Wo = g.Wo; p.Cout = g.Cout; p.cout_log = g.cout_log;
  p.pad = 1; p.reflect = g.reflect; p.act = g.act; p.tap_flip = 0; p.nchunks = g.Cin / 32;
  p.nbx = (g.Wo + 15) / 16; p.nby = (g.Ho + 7) / 8;
  p.in_nchunk = p.nbx * p.nby;
  p.in_tail = in_part ? vcg_in_tail_make(tail_req->out1, tail_req->out2, g.N * (g.Cout / 64), tail_req->HW, tail_req->eps) : vcg_in_tail_none();
  const unsigned long long ab = (unsigned long long)g.N * g.H * g.W * g.Cin * 4;
  VCG_CHECK_ARG(ab < (1ull << 31) && planes_bytes < (1ull << 31), "vcg_conv_fwd: tensor extents must stay below 2 GiB");
  p.in_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)planes_bytes;
  p.row_stride = (uint32_t)(9 * (g.Cin / 32) * VCG_PBYTES); p.tap_stride = (uint32_t)((g.Cin / 32) * VCG_PBYTES);
  {
    VcgProfScope prof("k_conv_slab<2, 2, 3, 3>", 2.0 * g.M * (double)g.K * g.Cout, st);
    hipLaunchKernelGGL((k_conv_slab<2, 2, 3, 3>), dim3(g.N * p.nbx * p.nby, g.Cout / 64), dim3(256), 0, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv_fwd(slab)");
  if (in_part && !p.in_tail.out1)
    return vcg_in_finalize(in_part, tail_req->out1, tail_req->out2, g.N, tail_req->HW, g.Cout, p.in_nchunk, tail_req->eps, st);
  return 0;
}

int vcg_slab_dgrad(const ConvGeom& g, const float* dy, const void* wfd_planes, size_t planes_bytes, const void* w_amax, float* dx,
                   void* ws, size_t ws_bytes, hipStream_t st, uint64_t dy_handle) {
  VCG_CHECK_ARG(ws && ws_bytes >= vcg_slab_dgrad_workspace(g), "vcg_conv_dgrad(slab): workspace too small");
  SlabP p = {};
  if (vcg_operand_amax(dy, (size_t)g.N * g.Ho * g.Wo * g.Cout, dy_handle, 0, st, &p.amax_a)) return -2;
  p.amax_b = vcg_amax_stored(w_amax);
  p.in = dy; p.planes = wfd_planes; p.bias = nullptr; p.out = (float*)ws; p.in_part = nullptr;
  p.N = g.N; p.H = g.Ho; p.W = g.Wo; p.C = g.Cout; p.Ho = g.H + 2; p.Wo = g.W + 2; p.Cout = g.Cin; p.cout_log = g.Cin;
  p.pad = 2; p.reflect = 0; p.act = VCG_ACT_NONE; p.tap_flip = 1; p.nchunks = g.Cout / 32;
  const bool wide = g.Cin == 32;                        // 32 columns: four waves along the pixels, 16 x 16 blocks
  p.nbx = (p.Wo + 15) / 16; p.nby = (p.Ho + (wide ? 15 : 7)) / (wide ? 16 : 8);
  const unsigned long long ab = (unsigned long long)g.N * g.Ho * g.Wo * g.Cout * 4;
  VCG_CHECK_ARG(ab < (1ull << 31) && planes_bytes < (1ull << 31), "vcg_conv_dgrad: tensor extents must stay below 2 GiB");
  p.in_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)planes_bytes;
  p.row_stride = (uint32_t)((g.Cout / 32) * VCG_PBYTES); p.tap_stride = (uint32_t)(g.Cin * (g.Cout / 32) * VCG_PBYTES);
  {
    const double flops = 2.0 * g.N * (double)p.Ho * p.Wo * 9.0 * g.Cout * g.Cin;
    VcgProfScope prof(wide ? "k_conv_slab<4, 1, 3, 3>" : "k_conv_slab<2, 2, 3, 3>", flops, st);
    if (wide) hipLaunchKernelGGL((k_conv_slab<4, 1, 3, 3>), dim3(g.N * p.nbx * p.nby, 1), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_conv_slab<2, 2, 3, 3>), dim3(g.N * p.nbx * p.nby, g.Cin / 64), dim3(256), 0, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv_dgrad(slab)");
  const size_t total = (size_t)g.N * g.H * g.W * (g.Cin / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_fold_pad_c, dim3(blocks), dim3(256), 0, st, (const float4*)ws, (float4*)dx, g.N, g.H, g.W, g.Cin / 4, g.reflect);
  VCG_LAUNCH_CHECK("vcg_conv_dgrad(slab fold)");
  return 0;
}

// ---- the (KH x 1) convolution of the kw-folded thin layers (conv_thin.hip): P[n][oh][pc][32] over the padded columns --------
// planes: [32 rows = (kw, co)][KH * C / 32 blocks][VCG_NP][32], k = (kh, c).  Reflect or zero padding `pad` on both axes (the
// column taps being part of N, a padded column pc simply reads input column reflect(pc - pad)).
bool vcg_slab_col_ok(int KH, int C) { return KH == 7 && C % 32 == 0 && C <= 128; }
int vcg_slab_col(const float* x, const void* planes, size_t planes_bytes, const void* w_amax, float* P, int N, int H, int W, int C,
                 int Ho, int Wo, int KH, int pad, int reflect, hipStream_t st, uint64_t x_handle) {
  VCG_CHECK_ARG(vcg_slab_col_ok(KH, C), "vcg_conv(kw-fold slab): unsupported KH=%d C=%d", KH, C);
  SlabP p = {};
  if (vcg_operand_amax(x, (size_t)N * H * W * C, x_handle, 0, st, &p.amax_a)) return -2;
  p.amax_b = vcg_amax_stored(w_amax);
  p.in = x; p.planes = planes; p.bias = nullptr; p.out = P; p.in_part = nullptr;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.Cout = 32; p.cout_log = 32;
  p.pad = pad; p.reflect = reflect; p.act = VCG_ACT_NONE; p.tap_flip = 0; p.nchunks = C / 32;
  p.nbx = (Wo + 15) / 16; p.nby = (Ho + 15) / 16;
  const unsigned long long ab = (unsigned long long)N * H * W * C * 4;
  VCG_CHECK_ARG(ab < (1ull << 31) && planes_bytes < (1ull << 31), "vcg_conv(kw-fold slab): tensor extents must stay below 2 GiB");
  p.in_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)planes_bytes;
  p.row_stride = (uint32_t)(KH * (C / 32) * VCG_PBYTES); p.tap_stride = (uint32_t)((C / 32) * VCG_PBYTES);
  {
    VcgProfScope prof("k_conv_slab<4, 1, 7, 1>", 2.0 * N * (double)Ho * Wo * KH * C * 32, st);
    hipLaunchKernelGGL((k_conv_slab<4, 1, 7, 1>), dim3(N * p.nbx * p.nby, 1), dim3(256), 0, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv(kw-fold slab)");
  return 0;
}
