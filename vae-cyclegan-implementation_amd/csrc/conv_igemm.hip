// Implicit-GEMM convolution on the gfx950 fp32 matrix core (v_mfma_f32_32x32x2_f32).
//
// Replaces nn.Conv2d(padding_mode='reflect') forward/backward at the call sites
// Networks.py:60 (CaSb), :87 (D, with PixelUnshuffle :86 folded into the gather),
// :101/:104 (R), :122 (U), :136 (S), :145 (L).
//
// Data layout in HBM
//   activations  NHWC fp32, channel pitch % 4 == 0
//   weights      Wf[K][Cout], K = (kh, kw, i, j, c)   (i,j = PixelUnshuffle phase, ups in {1,2})
//
// One workgroup = 256 threads = 4 waves in a 2x2 grid; a wave owns (BM/2)x(BN/2) of the
// BMxBN output tile as 32x32 MFMA accumulators.  K advances 32 per step; the next
// step's global loads are issued before the MFMAs of the current one (register staged),
// so with two workgroups per CU the matrix pipe always has a wave to run.
//
//   forward   C[m=(n,oh,ow)][co]  = sum_k A[m][k] * Wf[k][co]         A gathered from x
//   dgrad     C[m=(n,h,w)][(q,c)] = sum_(tap,co) A[m][(tap,co)] * Wf[(tap,q,c)][co]
//             A gathered from dy through the ADJOINT of the padding: a padded-domain
//             coordinate q folds onto h when reflect(q) == h, so each (h, kh) has up to
//             three source rows (h itself, -h near the top edge, 2(H-1)-h near the bottom).
//             stride 2 is decomposed into 4 parity classes (blockIdx.z) so no MFMA runs on
//             structurally-zero taps.
//   wgrad     C[(tap,q,c)][co] = sum_m A[m][(tap,q,c)] * dy[m][co], split over m across
//             blockIdx.z into fp32 slabs that a second kernel sums in a fixed order
//             (bitwise reproducible, no float atomics) and scatters to OIHW.
#include "vcg_common.h"
#include <stdlib.h>

struct ConvP {
  const float* a;
  const float* b;
  const float* bias;
  float* out;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, reflect, ups, act;
  int Hl, Wl, Ho, Wo, M, K;
  int cin4, cout4, cout_log;
  FastDiv fd_howo, fd_wo, fd_cin4, fd_cout4, fd_kw, fd_cin;
  // dgrad
  int Hc, Wc, Mc, NB;
  FastDiv fd_hcwc, fd_wc;
  // wgrad: "stream-K" work split.  The (tile, K' step) pairs are numbered tile-major; workgroup b owns the sk_len
  // consecutive units [b * sk_len, (b + 1) * sk_len), i.e. the tail of one tile, whole tiles, the head of another.
  // Every workgroup does the same number of MFMA steps whatever the tile count; a tile's partial sums go to
  // slab[j], j = b - (first workgroup touching the tile), and the reduce kernels sum sk_parts(tile) of them in order.
  int ktiles_total, sk_len, sk_units, sk_ntn, sk_bm_shift, sk_bn_shift;
  int sk_ntr_pb;               // row tiles per batch (batched wgrad: the row tiles of batch z follow those of z-1)
  FastDiv fd_sklen;
  // split-K of fwd / dgrad (stride 1): blockIdx.z = K slice, raw partial tiles go to slab[z][M][N]
  int ksplit, kt_per;
  float* slab;
  // wgrad with swapped roles (thin Cout): rows are gathered from dy through the adjoint of the padding
  int adjoint;
  int src_pitch;   // channel pitch of the tensor adjoint_gather reads (dy)
  uint32_t a_bytes, b_bytes;   // extents of a / b for the bounds-checked buffer loads (< 2 GiB each)
  int dbl_mirror;              // dgrad: some pixel has BOTH a top and a bottom (or left and right) mirror
  // batched launch of the forward kernel (the 16 GEMMs of a Winograd conv): blockIdx.z selects the batch
  int nbatch;
  uint32_t a_bstride, b_bstride;   // floats between consecutive batches of a / b
  size_t out_bstride;
  // forward into an InstanceNorm (vcg_conv_fwd_in): the tile also leaves sum / sum of squares of its 128 output rows per
  // channel, as chunk `(m0 % HoWo) / 128` of image `m0 / HoWo` in the [N][nchunk][Cout][2] double partials of norm.hip
  double* in_part;
  int in_nchunk;
  VcgInTail in_tail;             // ... and the last tile of an (image, column tile) combines them into mean / rstd (vcg_common.h)
  // fp16 x 2 split-operand kernels (vcg_common.h): the largest magnitude of the tensor behind `a` and of the one behind `b`
  VcgAmax amax_a, amax_b;
  // batched Winograd weight gradient: `a` (the kept V) is pre-split planes [batch][T][K / 32][2][32] (k_wino_in_planes), not fp32
  int a_planes;
};

#define BK 32
#define AS_STRIDE 33

// Diagnostic build only (tools/stamp_probe.py compiles a private copy with -DVCG_STAMP; libvcg.so never has it):
// per-workgroup wall-clock (100 MHz) and shader-clock stamps into a buffer of their own, so that launch skew,
// workgroup duration spread and the clock the chip holds can be read off (MI355X_MICROARCH.md, DVFS item 6).
#ifdef VCG_STAMP
__device__ unsigned long long* g_vcg_stamp = nullptr;
extern "C" int vcg_debug_set_stamp(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_vcg_stamp), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define VCG_STAMP_AT(slot)                                                                                   \
  do {                                                                                                       \
    if (g_vcg_stamp && threadIdx.x == 0) {                                                                   \
      unsigned long long* s__ = g_vcg_stamp + 8ull * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); \
      s__[slot] = __builtin_amdgcn_s_memrealtime();                                                          \
      if (slot == 0) { s__[4] = __builtin_amdgcn_s_memtime(); s__[6] = __builtin_amdgcn_s_getreg(63492); s__[7] = __builtin_amdgcn_s_getreg(63508); } \
      if (slot == 3) s__[5] = __builtin_amdgcn_s_memtime();                                                  \
    }                                                                                                        \
  } while (0)
#else
#define VCG_STAMP_AT(slot) do { } while (0)
#endif

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4add(float4& a, const float4& b) {
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

// Two-level accumulation.  An MFMA chain is a sequential fp32 sum: over K = 18 432 its rounding error
// reaches 2.5e-6 of the result (measured, tools/conv_accuracy.py) against 2.5e-7 for the CPU kernels
// the reference runs on.  That matters beyond the 1e-3 budget: a pre-activation within rounding of 0
// flips relu'(.) and moves that element's gradient by its full size, and the flip rate is proportional
// to this error.  Every FLUSH_TILES K-steps the chain is cut: acc is added into `tot` and restarted,
// which brings the error to ~u(sqrt(c)+sqrt(n/c)) (5x lower at K = 18 432) for 16*MI*NI more VGPRs.
#define FLUSH_TILES 8

template <int MI, int NI>
__device__ __forceinline__ void flush_acc(f32x16 (&acc)[MI][NI], f32x16 (&tot)[MI][NI]) {
  asm volatile("" ::: "memory");   // keeps this a real (rare) branch: if-converted it is 96 VALU ops in EVERY K-step
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      tot[i][j] += acc[i][j];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
}

// One K-step tile (BK = 32 -> 16 MFMA steps) with the operand fragments software-pipelined: the LDS reads of
// step s+1 are issued before the MFMAs of step s.  Left to itself hipcc places each step's ds_reads right in
// front of its MFMAs behind an s_waitcnt lgkmcnt(0), exposing the LDS latency every 4 MFMAs (~60 % MFMA
// utilisation measured with SQ_VALU_MFMA_BUSY_CYCLES); the double-buffered fragments cost MI+NI registers.
//
// Spreading the staging over the MFMA shadows (LDS writes in the first steps, buffer loads in the last ones, pinned
// with sched_barrier fences) was measured on k_conv_fwd: the two co-resident workgroups then finish together, but
// the launch takes the same time ("what did not pay"), so the kernels keep the simpler phases.
template <int MI, int NI, class FA, class FB>
__device__ __forceinline__ void mma_ktile(f32x16 (&acc)[MI][NI], FA ldA, FB ldB, int lh) {
  float a[2][MI], b[2][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) a[0][i] = ldA(lh, i);
#pragma unroll
  for (int j = 0; j < NI; ++j) b[0][j] = ldB(lh, j);
#pragma unroll
  for (int ks = 0; ks < BK / 2; ++ks) {
    const int cur = ks & 1, nxt = cur ^ 1;
    if (ks + 1 < BK / 2) {
#pragma unroll
      for (int i = 0; i < MI; ++i) a[nxt][i] = ldA((ks + 1) * 2 + lh, i);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[nxt][j] = ldB((ks + 1) * 2 + lh, j);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);   // the next step's LDS reads ...
    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);   // ... then this step's MFMAs
  }
}

// Bounds-checked 16-byte load through a buffer descriptor (SRD): a 32-bit byte offset per lane, and an
// offset >= num_records returns zeros — so rows past M, K-tail columns and zero-padding taps need no
// predication, no zero-initialised staging registers and no 64-bit address arithmetic.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define VCG_OOB 0x80000000u   // > any tensor we accept (host checks extents < 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const float* ptr, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// chunk g (4 consecutive k) of the forward K axis -> (kh, kw, i, j, c)
__device__ __forceinline__ void decode_tap(const ConvP& p, uint32_t g, int& kh, int& kw, int& ii, int& jj, int& c) {
  uint32_t tap = fd_div(g, p.fd_cin4);
  c = (int)(g - tap * (uint32_t)p.cin4) * 4;
  ii = 0; jj = 0;
  if (p.ups == 2) { jj = tap & 1; ii = (tap >> 1) & 1; tap >>= 2; }
  uint32_t q = fd_div(tap, p.fd_kw);
  kh = (int)q; kw = (int)(tap - q * (uint32_t)p.KW);
}

// sum of src[n, oh, ow, co..co+3] over every output pixel (oh, ow) whose padded-domain tap (kh, kw) lands on
// input pixel (h, w): the adjoint of (reflect) padding.  A padded coordinate q folds onto h when
// reflect(q) == h, so there are up to three source rows: h itself, -h near the top edge, 2(H-1)-h near the
// bottom one (same for columns).  nHo = n * Ho.
// `main` gets the always-present source (the tap read at (h, w) itself) as a plain load with NO use, so
// the caller can keep it in flight under the MFMAs; `extra` sums the fold sources, which exist only for
// pixels within `pad` of an edge (that add forces a wait, but only in those few waves).
__device__ __forceinline__ void adjoint_gather(const ConvP& p, const float* __restrict__ src, int nHo, int h, int w,
                                               int kh, int kw, int co, int sshift, float4& main, float4& extra) {
  main = f4zero();
  extra = f4zero();
  const bool edge = p.reflect && (h <= p.pad || h >= p.Hl - 1 - p.pad || w <= p.pad || w >= p.Wl - 1 - p.pad);
  {
    int numh = h - kh + p.pad, numw = w - kw + p.pad;
    int oh = numh >> sshift, ow = numw >> sshift;
    if (numh >= 0 && oh < p.Ho && numw >= 0 && ow < p.Wo)
      main = ldg4(src + ((size_t)(nHo + oh) * p.Wo + ow) * p.src_pitch + co);
  }
  if (!edge) return;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int qh; bool okh;
    if (a == 0) { qh = h; okh = true; }
    else if (a == 1) { qh = -h; okh = h >= 1 && h <= p.pad; }
    else { qh = 2 * (p.Hl - 1) - h; okh = h >= p.Hl - 1 - p.pad && h <= p.Hl - 2; }
    int numh = qh - kh + p.pad;
    int oh = numh >> sshift;
    okh = okh && numh >= 0 && oh < p.Ho;
    if (!okh) continue;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      if (a == 0 && b == 0) continue;            // the main source
      int qw; bool okw;
      if (b == 0) { qw = w; okw = true; }
      else if (b == 1) { qw = -w; okw = w >= 1 && w <= p.pad; }
      else { qw = 2 * (p.Wl - 1) - w; okw = w >= p.Wl - 1 - p.pad && w <= p.Wl - 2; }
      int numw = qw - kw + p.pad;
      int ow = numw >> sshift;
      okw = okw && numw >= 0 && ow < p.Wo;
      if (!okw) continue;
      f4add(extra, ldg4(src + ((size_t)(nHo + oh) * p.Wo + ow) * p.src_pitch + co));
    }
  }
}

// fold sources only (everything adjoint_gather puts into `extra`); used when the main source is fetched by a
// bounds-checked buffer load
__device__ __forceinline__ float4 adjoint_extras(const ConvP& p, const float* __restrict__ src, int nHo, int h, int w,
                                                 int kh, int kw, int co, int sshift) {
  float4 m, e;
  adjoint_gather(p, src, nHo, h, w, kh, kw, co, sshift, m, e);
  return e;
}

// ------------------------------------------------------------------ forward
// WN = wave columns: 2 (waves 2 x 2) or, for 32-column tiles, 1 (waves 4 x 1, each 32 rows x 32 columns)
// TWO = two-level accumulation (see flush_acc).  The Winograd GEMMs have K <= 2048: a single chain of that length
// rounds like the CPU's blocked kernels do, and without the second accumulator set the kernel fits three
// workgroups per CU.
template <int BM, int BN, int WN = 2, bool TWO = true>
__global__ __launch_bounds__(256, TWO ? 2 : 3) void k_conv_fwd(ConvP p) {   // TWO: acc + tot must fit 256 regs at 2 waves/SIMD
  VCG_STAMP_AT(0);
  constexpr int WM = 4 / WN;
  constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN), AR = BM / 32, BE = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[BM * AS_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, lh = lane >> 5;
  // Tile assignment.  Batched GEMMs (XCD-aware, cdna_hip_programming.md T1): workgroup ids round-robin over the 8
  // XCDs, each with its own L2, so the launch is cut into 8 contiguous runs of the N-fastest tile order — the N
  // tiles that share an A tile then run back to back on ONE XCD instead of fetching it into eight L2s.
  int mt = blockIdx.x, nt = blockIdx.y, zb = 0;
  if (p.nbatch > 1) {
    const uint32_t per = gridDim.x * gridDim.y, nwg = per * gridDim.z;
    const uint32_t gid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const uint32_t q = nwg >> 3, r = nwg & 7, xcd = gid & 7;
    const uint32_t swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (gid >> 3);   // bijective
    zb = (int)(swz / per);
    const uint32_t l = swz - (uint32_t)zb * per;
    mt = (int)(l / gridDim.y);
    nt = (int)(l - (uint32_t)mt * gridDim.y);
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int a_row = tid >> 3, a_u = tid & 7;

  const __amdgpu_buffer_rsrc_t ra = make_srd(p.a + (size_t)zb * p.a_bstride, p.a_bytes),
                               rb = make_srd(p.b + (size_t)zb * p.b_bstride, p.b_bytes);
  int pnH[AR], boh[AR], bow[AR];
  bool pv[AR];
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    uint32_t m = (uint32_t)(m0 + a_row + 32 * r);
    pv[r] = m < (uint32_t)p.M;
    uint32_t n = fd_div(m, p.fd_howo);
    uint32_t rem = m - n * (uint32_t)(p.Ho * p.Wo);
    uint32_t oh = fd_div(rem, p.fd_wo);
    uint32_t ow = rem - oh * (uint32_t)p.Wo;
    pnH[r] = (int)n * p.H;
    boh[r] = (int)oh * p.stride - p.pad;
    bow[r] = (int)ow * p.stride - p.pad;
  }

  f32x16 acc[MI][NI], tot[TWO ? MI : 1][TWO ? NI : 1];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        if (TWO) tot[i][j][e] = 0.f;
      }

  float4 va[AR], vb[BE];
  int nkt = (p.K + BK - 1) / BK;
  int kt0 = 0;
  if (p.ksplit > 1) {
    kt0 = (int)blockIdx.z * p.kt_per;
    int kt1 = kt0 + p.kt_per;
    nkt = kt1 < nkt ? kt1 : nkt;
  }

  // byte offsets of this thread's rows for the CURRENT tap; recomputed only when the tap changes (every
  // Cin/32 K-steps), so a steady-state K-step costs one add per row instead of the reflect arithmetic
  uint32_t rowoff[AR];
  int tap_cur = -1;
  // weight tile: this thread's float4 slots walk down K by 32 rows per step
  uint32_t boff[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    int idx = tid + 256 * e;
    int kk = idx / (BN / 4), j4 = idx % (BN / 4);
    int co = n0 + j4 * 4;
    boff[e] = co < p.Cout ? (uint32_t)(((kt0 * BK + kk) * p.Cout + co) * 4) : VCG_OOB;
  }
  const uint32_t bstep = (uint32_t)(BK * p.Cout * 4);

  auto load_tiles = [&](int kt) {
    const uint32_t g = (uint32_t)(kt * 8 + a_u);
    const bool kv = (int)(g * 4) < p.K;
    uint32_t tap = fd_div(g, p.fd_cin4);
    const int c = (int)(g - tap * (uint32_t)p.cin4) * 4;
    if ((int)tap != tap_cur) {
      tap_cur = (int)tap;
      int ii = 0, jj = 0;
      if (p.ups == 2) { jj = tap & 1; ii = (tap >> 1) & 1; tap >>= 2; }
      const uint32_t q = fd_div(tap, p.fd_kw);
      const int kh = (int)q, kw = (int)(tap - q * (uint32_t)p.KW);
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        int ih = boh[r] + kh, iw = bow[r] + kw;
        bool ok = pv[r];
        if (p.reflect) {
          ih = reflect_idx(ih, p.Hl);
          iw = reflect_idx(iw, p.Wl);
        } else {
          ok = ok && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
        }
        rowoff[r] = ok ? (uint32_t)(((pnH[r] + ih * p.ups + ii) * p.W + (iw * p.ups + jj)) * p.Cin) * 4u : VCG_OOB;
      }
    }
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      const uint32_t off = (kv && rowoff[r] != VCG_OOB) ? rowoff[r] + (uint32_t)c * 4u : VCG_OOB;
      va[r] = bload4(ra, off);
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      vb[e] = bload4(rb, boff[e]);
      if (boff[e] != VCG_OOB) boff[e] += bstep;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      float* d = &As[(a_row + 32 * r) * AS_STRIDE + a_u * 4];
      d[0] = va[r].x; d[1] = va[r].y; d[2] = va[r].z; d[3] = va[r].w;
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      int idx = tid + 256 * e;
      int kk = idx / (BN / 4), j4 = idx % (BN / 4);
      *reinterpret_cast<float4*>(&Bs[kk * BN + j4 * 4]) = vb[e];
    }
  };

  if (kt0 < nkt) {
    load_tiles(kt0);
    store_tiles();
  }
  __syncthreads();
  VCG_STAMP_AT(1);
  for (int kt = kt0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_tiles(kt + 1);
    mma_ktile<MI, NI>(
        acc, [&](int kk, int i) { return As[(wm * (BM / WM) + i * 32 + l31) * AS_STRIDE + kk]; },
        [&](int kk, int j) { return Bs[kk * BN + wn * (BN / WN) + j * 32 + l31]; }, lh);
    if constexpr (TWO)
      if (((kt - kt0 + 1) & (FLUSH_TILES - 1)) == 0 && kt + 1 < nkt) flush_acc<MI, NI>(acc, tot);
    __syncthreads();
    if (kt + 1 < nkt) {
      store_tiles();
      __syncthreads();
    }
  }
  VCG_STAMP_AT(2);
  if constexpr (TWO) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] += tot[i][j];
  }

  // epilogue: bias + activation, NHWC store (32 consecutive channels per half-wave);
  // a K slice stores its raw partial tile instead (bias/activation happen in k_splitk_finish)
  float* const dst = p.ksplit > 1 ? p.slab + (size_t)blockIdx.z * p.M * p.Cout : p.out + (size_t)zb * p.out_bstride;
  const int act = p.ksplit > 1 ? VCG_ACT_NONE : p.act;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + wn * (BN / WN) + j * 32 + l31;
    if (co >= p.Cout) continue;
    const float bv = (p.ksplit <= 1 && p.bias && co < p.cout_log) ? p.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int m = m0 + wm * (BM / WM) + i * 32 + row;
        if (m < p.M) dst[(size_t)m * p.Cout + co] = act_apply(acc[i][j][e] + bv, act);
      }
    }
  }
  VCG_STAMP_AT(3);
}

// one K-step (32) of the split-operand product: two 16-wide slices x three fp16 MFMAs per 32x32 accumulator (vcg_common.h:
// x / s = h + l).  The dominant h*h products go to `acc`, the two cross terms (<= 2^-11 of them) to `lo`: every add into an
// fp32 accumulator rounds relative to the accumulator's magnitude, so feeding all three into one chain would cost three
// roundings of the big running sum per slice instead of one.
template <int MI, int NI, int RA, int RB>
__device__ __forceinline__ void split_mma_ktile(f32x16 (&acc)[MI][NI], f32x16 (&lo)[MI][NI], const unsigned char (&As)[VCG_NP][RA],
                                                const unsigned char (&Bs)[VCG_NP][RB], const uint32_t (&fa)[MI], const uint32_t (&fb)[NI],
                                                const int (&sa)[MI], const int (&sb)[NI], int lh) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    f16x8 a[VCG_NP][MI], b[VCG_NP][NI];
#pragma unroll
    for (int pc = 0; pc < VCG_NP; ++pc) {
#pragma unroll
      for (int i = 0; i < MI; ++i) a[pc][i] = *reinterpret_cast<const f16x8*>(&As[pc][fa[i] + (((2 * s + lh) ^ sa[i]) << 4)]);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[pc][j] = *reinterpret_cast<const f16x8*>(&Bs[pc][fb[j] + (((2 * s + lh) ^ sb[j]) << 4)]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        f32x16 c = lo[i][j];
        c = VCG_MFMA(a[1][i], b[0][j], c);
        c = VCG_MFMA(a[0][i], b[1][j], c);
        lo[i][j] = c;
        acc[i][j] = VCG_MFMA(a[0][i], b[0][j], acc[i][j]);
      }
  }
}
// the operand scales of a split-operand kernel: 1 / sA (applied while A is split), 1 / sB (B, where the kernel splits it itself)
// and sA * sB (the epilogue)
struct SplitScales { float inv_a, inv_b, out; };
__device__ __forceinline__ SplitScales split_scales(const ConvP& p) {
  float sa, ia, sb, ib;
  vcg_scale_of(vcg_amax_bits(p.amax_a), p.amax_a.shift, sa, ia);
  vcg_scale_of(vcg_amax_bits(p.amax_b), p.amax_b.shift, sb, ib);
  SplitScales r; r.inv_a = ia; r.inv_b = ib; r.out = sa * sb;
  return r;
}

// The forward implicit GEMM on the 16-bit matrix pipe with split operands (vcg_common.h / gemm_split.hip explain the arithmetic:
// two fp16 pieces of x / s per fp32 value, three MFMA products, fp32 accumulation — fp32-level rounding).  Same gather as
// k_conv_fwd; B comes from the transposed pack WfT[Cout][K] so that both operands are staged as k-contiguous quads: A scaled,
// split and written to the swizzled [row][32 fp16] images that ds_read_b128 feeds to the MFMA, B copied from its planes.
template <int BN>
__global__ __launch_bounds__(256, 2) void k_conv_fwd_split(ConvP p) {
  VCG_STAMP_AT(0);
  constexpr int BM = 128, WN = 2, WM = 2;
  constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN), AR = BM / 32;
  __shared__ __attribute__((aligned(16))) unsigned char As[VCG_NP][BM * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[VCG_NP][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, lh = lane >> 5;
  const SplitScales sc = split_scales(p);
  // Tile assignment.  Batched GEMMs (XCD-aware, cdna_hip_programming.md T1): workgroup ids round-robin over the 8
  // XCDs, each with its own L2, so the launch is cut into 8 contiguous runs of the N-fastest tile order — the N
  // tiles that share an A tile then run back to back on ONE XCD instead of fetching it into eight L2s.
  int mt = blockIdx.x, nt = blockIdx.y, zb = 0;
  if (p.nbatch > 1) {
    const uint32_t per = gridDim.x * gridDim.y, nwg = per * gridDim.z;
    const uint32_t gid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const uint32_t q = nwg >> 3, r = nwg & 7, xcd = gid & 7;
    const uint32_t swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (gid >> 3);   // bijective
    zb = (int)(swz / per);
    const uint32_t l = swz - (uint32_t)zb * per;
    mt = (int)(l / gridDim.y);
    nt = (int)(l - (uint32_t)mt * gridDim.y);
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int a_row = tid >> 3, a_u = tid & 7;

  const __amdgpu_buffer_rsrc_t ra = make_srd(p.a + (size_t)zb * p.a_bstride, p.a_bytes),
                               rb = make_srd(p.b + (size_t)zb * p.b_bstride, p.b_bytes);
  int pnH[AR], boh[AR], bow[AR];
  bool pv[AR];
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    uint32_t m = (uint32_t)(m0 + a_row + 32 * r);
    pv[r] = m < (uint32_t)p.M;
    uint32_t n = fd_div(m, p.fd_howo);
    uint32_t rem = m - n * (uint32_t)(p.Ho * p.Wo);
    uint32_t oh = fd_div(rem, p.fd_wo);
    uint32_t ow = rem - oh * (uint32_t)p.Wo;
    pnH[r] = (int)n * p.H;
    boh[r] = (int)oh * p.stride - p.pad;
    bow[r] = (int)ow * p.stride - p.pad;
  }

  f32x16 acc[MI][NI], lo[MI][NI];               // h*h chain and cross-term chain (split_mma_ktile)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  float4 va[AR];
  int nkt = (p.K + BK - 1) / BK;
  int kt0 = 0;
  if (p.ksplit > 1) {
    kt0 = (int)blockIdx.z * p.kt_per;
    int kt1 = kt0 + p.kt_per;
    nkt = kt1 < nkt ? kt1 : nkt;
  }

  // byte offsets of this thread's rows for the CURRENT tap; recomputed only when the tap changes (every
  // Cin/32 K-steps), so a steady-state K-step costs one add per row instead of the reflect arithmetic
  uint32_t rowoff[AR];
  int tap_cur = -1;
  // weight tile from the WfT PLANES (the transposed pack, split into fp16 pieces when it was packed: [Cout][K/32][2][32],
  // K zero-padded to 32): thread (row b_r = tid >> 2, 16-byte chunk b_q = tid & 3 of a 64-byte piece row); pass j =
  // (64-row half, piece).  Rows past Cout fall off the end of the buffer and read as zeros.
  constexpr int BP = VCG_NP * BN / 64;
  const int b_q = tid & 3, b_r = tid >> 2;
  const int KB = (p.K + 31) / 32;
  const uint32_t boff0 = (uint32_t)(((size_t)(n0 + b_r) * KB) * VCG_PBYTES + b_q * 16);
  const uint32_t bhalf = (uint32_t)KB * (64u * VCG_PBYTES);
  const uint32_t bsoff0 = (uint32_t)(b_r * 64 + ((b_q ^ ((b_r >> 2) & 3)) << 4));
  u32x4 vbp[BP];
  // LDS byte offset of this thread's quad in a piece image (row r, chunk a_u >> 1 swizzled by (r >> 2) & 3, half a_u & 1)
  uint32_t soff[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = a_row + 32 * i;
    soff[i] = (uint32_t)(r * 64 + (((a_u >> 1) ^ ((r >> 2) & 3)) << 4) + ((a_u & 1) << 3));
  }

  auto load_tiles = [&](int kt) {
    const uint32_t g = (uint32_t)(kt * 8 + a_u);
    const bool kv = (int)(g * 4) < p.K;
    uint32_t tap = fd_div(g, p.fd_cin4);
    const int c = (int)(g - tap * (uint32_t)p.cin4) * 4;
    if ((int)tap != tap_cur) {
      tap_cur = (int)tap;
      int ii = 0, jj = 0;
      if (p.ups == 2) { jj = tap & 1; ii = (tap >> 1) & 1; tap >>= 2; }
      const uint32_t q = fd_div(tap, p.fd_kw);
      const int kh = (int)q, kw = (int)(tap - q * (uint32_t)p.KW);
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        int ih = boh[r] + kh, iw = bow[r] + kw;
        bool ok = pv[r];
        if (p.reflect) {
          ih = reflect_idx(ih, p.Hl);
          iw = reflect_idx(iw, p.Wl);
        } else {
          ok = ok && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
        }
        rowoff[r] = ok ? (uint32_t)(((pnH[r] + ih * p.ups + ii) * p.W + (iw * p.ups + jj)) * p.Cin) * 4u : VCG_OOB;
      }
    }
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      const uint32_t off = (kv && rowoff[r] != VCG_OOB) ? rowoff[r] + (uint32_t)c * 4u : VCG_OOB;
      va[r] = bload4(ra, off);
    }
#pragma unroll
    for (int j = 0; j < BP; ++j)
      vbp[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(boff0 + (uint32_t)(j / VCG_NP) * bhalf + (uint32_t)(j % VCG_NP) * 64u + (uint32_t)kt * VCG_PBYTES), 0, 0);
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      uint2 h, l;
      split4h(va[r], sc.inv_a, h, l);
      *reinterpret_cast<uint2*>(&As[0][soff[r]]) = h;
      *reinterpret_cast<uint2*>(&As[1][soff[r]]) = l;
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) *reinterpret_cast<u32x4*>(&Bs[j % VCG_NP][bsoff0 + 4096 * (j / VCG_NP)]) = vbp[j];
  };
  uint32_t fa[MI], fb[NI];
  int sa[MI], sb[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { const int r = wm * (BM / WM) + i * 32 + l31; fa[i] = (uint32_t)(r * 64); sa[i] = (r >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < NI; ++j) { const int r = wn * (BN / WN) + j * 32 + l31; fb[j] = (uint32_t)(r * 64); sb[j] = (r >> 2) & 3; }

  if (kt0 < nkt) {
    load_tiles(kt0);
    store_tiles();
  }
  __syncthreads();
  VCG_STAMP_AT(1);
  for (int kt = kt0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_tiles(kt + 1);
    split_mma_ktile<MI, NI>(acc, lo, As, Bs, fa, fb, sa, sb, lh);
    __syncthreads();
    if (kt + 1 < nkt) {
      store_tiles();
      __syncthreads();
    }
  }
  VCG_STAMP_AT(2);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] += lo[i][j];

  // epilogue: bias + activation, NHWC store (32 consecutive channels per half-wave);
  // a K slice stores its raw partial tile instead (bias/activation happen in k_splitk_finish)
  float* const dst = p.ksplit > 1 ? p.slab + (size_t)blockIdx.z * p.M * p.Cout : p.out + (size_t)zb * p.out_bstride;
  const int act = p.ksplit > 1 ? VCG_ACT_NONE : p.act;
  // the statistics' sums are kept in double from the first element on, as norm.hip's own pass and the Winograd output
  // transform do: var = E[x^2] - mean^2 cancels for channels whose mean dwarfs their spread, and 64 squares summed in fp32
  // left mean / rstd ~1e-5 off there — which the InstanceNorm backward (rstd up to 316 on near-dead channels) turned into
  // gradient errors of 4e-3 on the 64 x 64 fixtures once D1 and U2 took this path (round 3; 2e-4 with the sums in double)
  double* const red = reinterpret_cast<double*>(&As[0][0]);   // [wm][BN][2] doubles; the K loop's last barrier freed As
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int cl = wn * (BN / WN) + j * 32 + l31;
    const int co = n0 + cl;
    const bool cv = co < p.Cout;
    const float bv = (cv && p.ksplit <= 1 && p.bias && co < p.cout_log) ? p.bias[co] : 0.f;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int m = m0 + wm * (BM / WM) + i * 32 + row;
        const float v = act_apply(acc[i][j][e] * sc.out + bv, act);
        if (cv && m < p.M) dst[(size_t)m * p.Cout + co] = v;
        if (p.in_part) {
          s1 += (double)v;
          s2 += (double)v * (double)v;
        }
      }
    }
    if (p.in_part) {                    // uniform; the host only sets it when every tile row is a valid pixel of ONE image
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lh == 0) {
        red[(wm * BN + cl) * 2] = s1;
        red[(wm * BN + cl) * 2 + 1] = s2;
      }
    }
  }
  if (p.in_part) {
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      const uint32_t n = fd_div((uint32_t)m0, p.fd_howo);
      const uint32_t chunk = ((uint32_t)m0 - n * (uint32_t)(p.Ho * p.Wo)) / BM;
      double* o = p.in_part + (((size_t)n * p.in_nchunk + chunk) * p.Cout + n0 + tid) * 2;
      vcg_store_sc1(o, red[tid * 2] + red[(BN + tid) * 2]);
      vcg_store_sc1(o + 1, red[tid * 2 + 1] + red[(BN + tid) * 2 + 1]);
    }
    if (p.in_tail.out1) {
      const uint32_t n = fd_div((uint32_t)m0, p.fd_howo);
      __syncthreads();                // `red` (As) has been read
      vcg_in_tail_run<0>(p.in_tail, p.in_part, (int)n, n0, BN, p.Cout, p.in_nchunk, p.in_tail.counters + n * gridDim.y + nt,
                         (uint32_t)p.in_nchunk, reinterpret_cast<double*>(&As[0][0]));
    }
  }
  VCG_STAMP_AT(3);
}

// out[m][c] = act(sum_z slab[z][m][c] + bias[c]) — fixed summation order, float4 per lane
__global__ __launch_bounds__(256) void k_splitk_finish(const float* __restrict__ slab, const float* __restrict__ bias,
                                                       float* __restrict__ out, size_t rows, int C, int nsplit,
                                                       int c_log, int act) {
  const int C4 = C / 4;
  const size_t total4 = rows * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    float4 s = f4zero();
    for (int z = 0; z < nsplit; ++z) f4add(s, reinterpret_cast<const float4*>(slab)[(size_t)z * total4 + i]);
    const int c = (int)(i % C4) * 4;
    if (bias) {
      if (c + 0 < c_log) s.x += bias[c + 0];
      if (c + 1 < c_log) s.y += bias[c + 1];
      if (c + 2 < c_log) s.z += bias[c + 2];
      if (c + 3 < c_log) s.w += bias[c + 3];
    }
    s.x = act_apply(s.x, act); s.y = act_apply(s.y, act); s.z = act_apply(s.z, act); s.w = act_apply(s.w, act);
    reinterpret_cast<float4*>(out)[i] = s;
  }
}

// ------------------------------------------------------------------ dgrad
template <int BM, int BN, int WN = 2>
__global__ __launch_bounds__(256, 2) void k_conv_dgrad(ConvP p) {
  VCG_STAMP_AT(0);
  constexpr int WM = 4 / WN;                                   // wave rows x WN wave columns (see k_conv_fwd)
  constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN), AR = BM / 32, BR = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[2][BM * AS_STRIDE];   // double-buffered, see k_conv_fwd
  __shared__ __attribute__((aligned(16))) float Bt[2][BN * AS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int a_row = tid >> 3, a_u = tid & 7;
  const int s = p.stride, sshift = s - 1;
  // blockIdx.z = parity class (stride 2: s * s of them) + s * s * K slice
  const int cls = (int)blockIdx.z % (s * s), kslice = (int)blockIdx.z / (s * s);
  const int ca = cls / s, cb = cls % s;
  const int kh0 = (ca + p.pad) % s, kw0 = (cb + p.pad) % s;
  const int nKH = (p.KH - kh0 + s - 1) / s, nKW = (p.KW - kw0 + s - 1) / s;
  const int Kc = nKH * nKW * p.Cout;

  int pnHo[AR], ph[AR], pw[AR];
  bool pv[AR];
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    uint32_t m = (uint32_t)(m0 + a_row + 32 * r);
    pv[r] = m < (uint32_t)p.Mc;
    uint32_t n = fd_div(m, p.fd_hcwc);
    uint32_t rem = m - n * (uint32_t)(p.Hc * p.Wc);
    uint32_t hq = fd_div(rem, p.fd_wc);
    uint32_t wq = rem - hq * (uint32_t)p.Wc;
    pnHo[r] = (int)n * p.Ho;
    ph[r] = (int)hq * s + ca;
    pw[r] = (int)wq * s + cb;
  }

  f32x16 acc[MI][NI], tot[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = tot[i][j][e] = 0.f;

  float4 va[AR], ve[AR], vb[BR];
  int nkt = (Kc + BK - 1) / BK;
  int kt0 = 0;
  if (p.ksplit > 1) {
    kt0 = kslice * p.kt_per;
    int kt1 = kt0 + p.kt_per;
    nkt = kt1 < nkt ? kt1 : nkt;
  }

  // Bounds-checked buffer loads + per-tap row offsets (see k_conv_fwd).  A pixel within `pad` of a border
  // (but not on it) also receives the contributions that reflect padding folded onto it: one mirrored row
  // coordinate eh and/or one mirrored column coordinate ew, i.e. up to three extra sources per tap, whose
  // offsets are likewise computed once per tap.
  const __amdgpu_buffer_rsrc_t ra = make_srd(p.a, p.a_bytes), rb = make_srd(p.b, p.b_bytes);
  const int NONE = -(1 << 20);
  int eh[AR], ew[AR];
  bool edge[AR], any_edge = false;
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const int h = ph[r], w = pw[r];
    eh[r] = (h >= 1 && h <= p.pad) ? -h : (h >= p.Hl - 1 - p.pad && h <= p.Hl - 2) ? 2 * (p.Hl - 1) - h : NONE;
    ew[r] = (w >= 1 && w <= p.pad) ? -w : (w >= p.Wl - 1 - p.pad && w <= p.Wl - 2) ? 2 * (p.Wl - 1) - w : NONE;
    edge[r] = pv[r] && p.reflect && (eh[r] != NONE || ew[r] != NONE);
    any_edge = any_edge || edge[r];
    ve[r] = f4zero();
  }
  uint32_t rowoff[AR], xoff[AR][3], wbase[BR];
  int tap_cur = -1;

  auto load_tiles = [&](int kt) {
    const uint32_t g = (uint32_t)(kt * 8 + a_u);
    const bool kv = (int)(g * 4) < Kc;
    const uint32_t tapc = fd_div(g, p.fd_cout4);
    const int co = (int)(g - tapc * (uint32_t)p.cout4) * 4;
    if ((int)tapc != tap_cur) {
      tap_cur = (int)tapc;
      const int u = (int)tapc / nKW, v_ = (int)tapc % nKW;
      const int kh = kh0 + u * s, kw = kw0 + v_ * s;
      auto src = [&](int r, int qh, int qw) -> uint32_t {      // dy offset of the output pixel whose tap hits (qh, qw)
        const int numh = qh - kh + p.pad, numw = qw - kw + p.pad;
        const int oh = numh >> sshift, ow = numw >> sshift;
        const bool ok = pv[r] && qh != NONE && qw != NONE && numh >= 0 && oh < p.Ho && numw >= 0 && ow < p.Wo;
        return ok ? (uint32_t)(((pnHo[r] + oh) * p.Wo + ow) * p.Cout) * 4u : VCG_OOB;
      };
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        rowoff[r] = src(r, ph[r], pw[r]);
        if (any_edge) {
          xoff[r][0] = edge[r] ? src(r, eh[r], pw[r]) : VCG_OOB;
          xoff[r][1] = edge[r] ? src(r, ph[r], ew[r]) : VCG_OOB;
          xoff[r][2] = edge[r] ? src(r, eh[r], ew[r]) : VCG_OOB;
        }
      }
      const int tapfull = kh * p.KW + kw;
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        const int J = n0 + a_row + 32 * r;
        wbase[r] = J < p.NB ? (uint32_t)((tapfull * p.NB + J) * p.Cout) * 4u : VCG_OOB;
      }
    }
    const uint32_t cb4 = (uint32_t)co * 4u;
#pragma unroll
    for (int r = 0; r < AR; ++r) va[r] = bload4(ra, (kv && rowoff[r] != VCG_OOB) ? rowoff[r] + cb4 : VCG_OOB);
    if (p.dbl_mirror) {                      // tiny maps (e.g. 3x3 with pad 1): general 3x3 candidate search
      const int u = tap_cur / nKW, v_ = tap_cur % nKW;
#pragma unroll
      for (int r = 0; r < AR; ++r)
        ve[r] = (pv[r] && kv) ? adjoint_extras(p, p.a, pnHo[r], ph[r], pw[r], kh0 + u * s, kw0 + v_ * s, co, sshift) : f4zero();
    } else if (any_edge) {
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        float4 e = f4zero();
        if (edge[r] && kv) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (xoff[r][k] != VCG_OOB) f4add(e, bload4(ra, xoff[r][k] + cb4));
        }
        ve[r] = e;
      }
    }
#pragma unroll
    for (int r = 0; r < BR; ++r) vb[r] = bload4(rb, (kv && wbase[r] != VCG_OOB) ? wbase[r] + cb4 : VCG_OOB);
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      float* d = &As[buf][(a_row + 32 * r) * AS_STRIDE + a_u * 4];
      d[0] = va[r].x + ve[r].x; d[1] = va[r].y + ve[r].y; d[2] = va[r].z + ve[r].z; d[3] = va[r].w + ve[r].w;   // ve == 0 off the edges
    }
#pragma unroll
    for (int r = 0; r < BR; ++r) {
      float* d = &Bt[buf][(a_row + 32 * r) * AS_STRIDE + a_u * 4];
      d[0] = vb[r].x; d[1] = vb[r].y; d[2] = vb[r].z; d[3] = vb[r].w;
    }
  };

  if (kt0 < nkt) {
    load_tiles(kt0);
    store_tiles(0);
  }
  __syncthreads();
  VCG_STAMP_AT(1);
  for (int kt = kt0; kt < nkt; ++kt) {
    const int cur = (kt - kt0) & 1;
    const float* const Ac = As[cur];
    const float* const Bc = Bt[cur];
    if (kt + 1 < nkt) load_tiles(kt + 1);
    mma_ktile<MI, NI>(
        acc, [&](int kk, int i) { return Ac[(wm * (BM / WM) + i * 32 + l31) * AS_STRIDE + kk]; },
        [&](int kk, int j) { return Bc[(wn * (BN / WN) + j * 32 + l31) * AS_STRIDE + kk]; }, lh);
    if (((kt - kt0 + 1) & (FLUSH_TILES - 1)) == 0 && kt + 1 < nkt) flush_acc<MI, NI>(acc, tot);
    if (kt + 1 < nkt) store_tiles(cur ^ 1);
    __syncthreads();
  }
  VCG_STAMP_AT(2);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] += tot[i][j];

  // epilogue: column J = (q, c) -> physical pixel (h*ups + i, w*ups + j), channel c.
  // A K slice writes its raw partial into slab[z] in the same physical layout; k_splitk_finish sums them.
  float* const dst = p.ksplit > 1 ? p.slab + (size_t)kslice * p.N * p.H * p.W * p.Cin : p.out;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int J = n0 + wn * (BN / WN) + j * 32 + l31;
    if (J >= p.NB) continue;
    int q = 0, c = J;
    if (p.ups == 2) { q = (int)fd_div((uint32_t)J, p.fd_cin); c = J - q * p.Cin; }
    const int qi = q >> 1, qj = q & 1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const uint32_t m = (uint32_t)(m0 + wm * (BM / WM) + i * 32 + row);
        if (m < (uint32_t)p.Mc) {
          uint32_t n = fd_div(m, p.fd_hcwc);
          uint32_t rem = m - n * (uint32_t)(p.Hc * p.Wc);
          uint32_t hq = fd_div(rem, p.fd_wc);
          uint32_t wq = rem - hq * (uint32_t)p.Wc;
          int h = (int)hq * s + ca, w = (int)wq * s + cb;
          size_t off = ((size_t)((int)n * p.H + h * p.ups + qi) * p.W + (w * p.ups + qj)) * p.Cin + c;
          dst[off] = acc[i][j][e];
        }
      }
    }
  }
  VCG_STAMP_AT(3);
}

// The data gradient on the split-operand path (see k_conv_fwd_split / gemm_split.hip).  Same gather as
// k_conv_dgrad — including the fold of the reflect halo, summed in fp32 BEFORE the split — and the same weight rows
// (Wf viewed as [tap][J][co] is already k-contiguous per output column J).
template <int BN, int WN, bool BPL = (BN >= 64)>
__global__ __launch_bounds__(256, 2) void k_conv_dgrad_split(ConvP p) {
  VCG_STAMP_AT(0);
  constexpr int BM = 128, WM = 4 / WN;
  constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN), AR = BM / 32, BR = BN / 32;
  __shared__ __attribute__((aligned(16))) unsigned char As[VCG_NP][BM * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[VCG_NP][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, lh = lane >> 5;
  const SplitScales sc = split_scales(p);
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int a_row = tid >> 3, a_u = tid & 7;
  const int s = p.stride, sshift = s - 1;
  // blockIdx.z = parity class (stride 2: s * s of them) + s * s * K slice
  const int cls = (int)blockIdx.z % (s * s), kslice = (int)blockIdx.z / (s * s);
  const int ca = cls / s, cb = cls % s;
  const int kh0 = (ca + p.pad) % s, kw0 = (cb + p.pad) % s;
  const int nKH = (p.KH - kh0 + s - 1) / s, nKW = (p.KW - kw0 + s - 1) / s;
  const int Kc = nKH * nKW * p.Cout;

  int pnHo[AR], ph[AR], pw[AR];
  bool pv[AR];
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    uint32_t m = (uint32_t)(m0 + a_row + 32 * r);
    pv[r] = m < (uint32_t)p.Mc;
    uint32_t n = fd_div(m, p.fd_hcwc);
    uint32_t rem = m - n * (uint32_t)(p.Hc * p.Wc);
    uint32_t hq = fd_div(rem, p.fd_wc);
    uint32_t wq = rem - hq * (uint32_t)p.Wc;
    pnHo[r] = (int)n * p.Ho;
    ph[r] = (int)hq * s + ca;
    pw[r] = (int)wq * s + cb;
  }

  f32x16 acc[MI][NI], lo[MI][NI];               // h*h chain and cross-term chain (split_mma_ktile)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  float4 va[AR], ve[AR], vb[BPL ? 1 : BR];
  // BPL: the weight rows come pre-split from the pack's dgrad planes [(tap, J)][Cout/32][3][32] (Cout % 32 == 0): thread
  // (row b_r = tid >> 2, 16-byte chunk b_q = tid & 3 of a 64-byte piece row), pass j = (64-row half, piece)
  constexpr int BP = BPL ? VCG_NP * BN / 64 : 1;
  const int b_q = tid & 3, b_r = tid >> 2;
  const int CB = p.Cout / 32;
  const uint32_t bsoff0 = (uint32_t)(b_r * 64 + ((b_q ^ ((b_r >> 2) & 3)) << 4));
  u32x4 vbp[BP];
  int nkt = (Kc + BK - 1) / BK;
  int kt0 = 0;
  if (p.ksplit > 1) {
    kt0 = kslice * p.kt_per;
    int kt1 = kt0 + p.kt_per;
    nkt = kt1 < nkt ? kt1 : nkt;
  }

  // Bounds-checked buffer loads + per-tap row offsets (see k_conv_fwd).  A pixel within `pad` of a border
  // (but not on it) also receives the contributions that reflect padding folded onto it: one mirrored row
  // coordinate eh and/or one mirrored column coordinate ew, i.e. up to three extra sources per tap, whose
  // offsets are likewise computed once per tap.
  const __amdgpu_buffer_rsrc_t ra = make_srd(p.a, p.a_bytes), rb = make_srd(p.b, p.b_bytes);
  const int NONE = -(1 << 20);
  int eh[AR], ew[AR];
  bool edge[AR], any_edge = false;
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const int h = ph[r], w = pw[r];
    eh[r] = (h >= 1 && h <= p.pad) ? -h : (h >= p.Hl - 1 - p.pad && h <= p.Hl - 2) ? 2 * (p.Hl - 1) - h : NONE;
    ew[r] = (w >= 1 && w <= p.pad) ? -w : (w >= p.Wl - 1 - p.pad && w <= p.Wl - 2) ? 2 * (p.Wl - 1) - w : NONE;
    edge[r] = pv[r] && p.reflect && (eh[r] != NONE || ew[r] != NONE);
    any_edge = any_edge || edge[r];
    ve[r] = f4zero();
  }
  uint32_t rowoff[AR], xoff[AR][3], wbase[BR];
  int tap_cur = -1;

  auto load_tiles = [&](int kt) {
    const uint32_t g = (uint32_t)(kt * 8 + a_u);
    const bool kv = (int)(g * 4) < Kc;
    const uint32_t tapc = fd_div(g, p.fd_cout4);
    const int co = (int)(g - tapc * (uint32_t)p.cout4) * 4;
    if ((int)tapc != tap_cur) {
      tap_cur = (int)tapc;
      const int u = (int)tapc / nKW, v_ = (int)tapc % nKW;
      const int kh = kh0 + u * s, kw = kw0 + v_ * s;
      auto src = [&](int r, int qh, int qw) -> uint32_t {      // dy offset of the output pixel whose tap hits (qh, qw)
        const int numh = qh - kh + p.pad, numw = qw - kw + p.pad;
        const int oh = numh >> sshift, ow = numw >> sshift;
        const bool ok = pv[r] && qh != NONE && qw != NONE && numh >= 0 && oh < p.Ho && numw >= 0 && ow < p.Wo;
        return ok ? (uint32_t)(((pnHo[r] + oh) * p.Wo + ow) * p.Cout) * 4u : VCG_OOB;
      };
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        rowoff[r] = src(r, ph[r], pw[r]);
        if (any_edge) {
          xoff[r][0] = edge[r] ? src(r, eh[r], pw[r]) : VCG_OOB;
          xoff[r][1] = edge[r] ? src(r, ph[r], ew[r]) : VCG_OOB;
          xoff[r][2] = edge[r] ? src(r, eh[r], ew[r]) : VCG_OOB;
        }
      }
      const int tapfull = kh * p.KW + kw;
      if constexpr (!BPL) {
#pragma unroll
        for (int r = 0; r < BR; ++r) {
          const int J = n0 + a_row + 32 * r;
          wbase[r] = J < p.NB ? (uint32_t)((tapfull * p.NB + J) * p.Cout) * 4u : VCG_OOB;
        }
      }
    }
    const uint32_t cb4 = (uint32_t)co * 4u;
#pragma unroll
    for (int r = 0; r < AR; ++r) va[r] = bload4(ra, (kv && rowoff[r] != VCG_OOB) ? rowoff[r] + cb4 : VCG_OOB);
    if (p.dbl_mirror) {                      // tiny maps (e.g. 3x3 with pad 1): general 3x3 candidate search
      const int u = tap_cur / nKW, v_ = tap_cur % nKW;
#pragma unroll
      for (int r = 0; r < AR; ++r)
        ve[r] = (pv[r] && kv) ? adjoint_extras(p, p.a, pnHo[r], ph[r], pw[r], kh0 + u * s, kw0 + v_ * s, co, sshift) : f4zero();
    } else if (any_edge) {
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        float4 e = f4zero();
        if (edge[r] && kv) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (xoff[r][k] != VCG_OOB) f4add(e, bload4(ra, xoff[r][k] + cb4));
        }
        ve[r] = e;
      }
    }
    if constexpr (BPL) {
      // the K-step's 32 reduction indices are 32 consecutive co of ONE tap (Cout % 32 == 0): wave-uniform
      const int tk = (kt * 32) / p.Cout, cob = (kt * 32 - tk * p.Cout) >> 5;
      const int tf = (kh0 + (tk / nKW) * s) * p.KW + (kw0 + (tk % nKW) * s);
      const bool kvt = kt * 32 < Kc;
#pragma unroll
      for (int j = 0; j < BP; ++j) {
        const int J = n0 + b_r + 64 * (j / VCG_NP);
        const uint32_t off = (kvt && J < p.NB) ? (uint32_t)((((size_t)tf * p.NB + J) * CB + cob) * VCG_PBYTES + (j % VCG_NP) * 64 + b_q * 16) : VCG_OOB;
        vbp[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)off, 0, 0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < BR; ++r) vb[r] = bload4(rb, (kv && wbase[r] != VCG_OOB) ? wbase[r] + cb4 : VCG_OOB);
    }
  };
  uint32_t soff[AR > BR ? AR : BR];
#pragma unroll
  for (int i = 0; i < (AR > BR ? AR : BR); ++i) {
    const int r = a_row + 32 * i;
    soff[i] = (uint32_t)(r * 64 + (((a_u >> 1) ^ ((r >> 2) & 3)) << 4) + ((a_u & 1) << 3));
  }
  auto store_tiles = [&]() {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      const float4 v = make_float4(va[r].x + ve[r].x, va[r].y + ve[r].y, va[r].z + ve[r].z, va[r].w + ve[r].w);   // ve == 0 off the edges
      uint2 h, l;
      split4h(v, sc.inv_a, h, l);
      *reinterpret_cast<uint2*>(&As[0][soff[r]]) = h;
      *reinterpret_cast<uint2*>(&As[1][soff[r]]) = l;
    }
    if constexpr (BPL) {
#pragma unroll
      for (int j = 0; j < BP; ++j) *reinterpret_cast<u32x4*>(&Bs[j % VCG_NP][bsoff0 + 4096 * (j / VCG_NP)]) = vbp[j];
    } else {
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        uint2 h, l;
        split4h(vb[r], sc.inv_b, h, l);
        *reinterpret_cast<uint2*>(&Bs[0][soff[r]]) = h;
        *reinterpret_cast<uint2*>(&Bs[1][soff[r]]) = l;
      }
    }
  };
  uint32_t fa[MI], fb[NI];
  int sa[MI], sb[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { const int r = wm * (BM / WM) + i * 32 + l31; fa[i] = (uint32_t)(r * 64); sa[i] = (r >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < NI; ++j) { const int r = wn * (BN / WN) + j * 32 + l31; fb[j] = (uint32_t)(r * 64); sb[j] = (r >> 2) & 3; }

  if (kt0 < nkt) {
    load_tiles(kt0);
    store_tiles();
  }
  __syncthreads();
  VCG_STAMP_AT(1);
  for (int kt = kt0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_tiles(kt + 1);
    split_mma_ktile<MI, NI>(acc, lo, As, Bs, fa, fb, sa, sb, lh);
    __syncthreads();
    if (kt + 1 < nkt) {
      store_tiles();
      __syncthreads();
    }
  }
  VCG_STAMP_AT(2);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] += lo[i][j];

  // epilogue: column J = (q, c) -> physical pixel (h*ups + i, w*ups + j), channel c.
  // A K slice writes its raw partial into slab[z] in the same physical layout; k_splitk_finish sums them.
  float* const dst = p.ksplit > 1 ? p.slab + (size_t)kslice * p.N * p.H * p.W * p.Cin : p.out;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int J = n0 + wn * (BN / WN) + j * 32 + l31;
    if (J >= p.NB) continue;
    int q = 0, c = J;
    if (p.ups == 2) { q = (int)fd_div((uint32_t)J, p.fd_cin); c = J - q * p.Cin; }
    const int qi = q >> 1, qj = q & 1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const uint32_t m = (uint32_t)(m0 + wm * (BM / WM) + i * 32 + row);
        if (m < (uint32_t)p.Mc) {
          uint32_t n = fd_div(m, p.fd_hcwc);
          uint32_t rem = m - n * (uint32_t)(p.Hc * p.Wc);
          uint32_t hq = fd_div(rem, p.fd_wc);
          uint32_t wq = rem - hq * (uint32_t)p.Wc;
          int h = (int)hq * s + ca, w = (int)wq * s + cb;
          size_t off = ((size_t)((int)n * p.H + h * p.ups + qi) * p.W + (w * p.ups + qj)) * p.Cin + c;
          dst[off] = acc[i][j][e] * sc.out;
        }
      }
    }
  }
  VCG_STAMP_AT(3);
}

// Stream-K workgroup numbering, XCD-aware: hardware workgroup ids go round-robin over the 8 XCDs, so XCD x gets the
// ids x, x+8, ...; handing it a CONTIGUOUS range of logical workgroups (= consecutive tiles: the column tiles of one
// row tile, then the next row tile of the same batch) lets its L2 serve the operand re-reads.  With ids dealt
// round-robin the weight-gradient GEMMs fetched 7x their operands from the fabric (PMC: 464 MB per launch).
__device__ __forceinline__ int sk_logical_wg() {
  const uint32_t nwg = gridDim.x, gid = blockIdx.x;
  const uint32_t q = nwg >> 3, r = nwg & 7, xcd = gid & 7;
  return (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (gid >> 3));
}

// ------------------------------------------------------------------ wgrad
// NT = 256: four waves, two such workgroups per CU.  NT = 512: eight waves in lockstep on one 256-row tile, one
// workgroup per CU — the two waves of a SIMD then advance together (the per-K-step barrier), where two independent
// workgroups do not: the hardware favours wave slot 0, which finishes ~20 % early and leaves its partner alone on
// the CU for the rest of the launch (tools/stamp_probe.py).
template <int BM, int BN, int NT = 256>
__global__ __launch_bounds__(NT) void k_conv_wgrad(ConvP p) {
  VCG_STAMP_AT(0);
  constexpr int WR = NT / 128;                                 // wave rows (x 2 wave columns)
  constexpr int MI = BM / (32 * WR), NI = BN / 64;
  constexpr int RQ = BM / 4, PS = NT / RQ, AP = BK / PS, BE = (BK * BN / 4) / NT;
  __shared__ __attribute__((aligned(16))) float Xs[2][BK * BM];   // double-buffered, see k_conv_fwd
  __shared__ __attribute__((aligned(16))) float Ds[2][BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
  const int rq = tid % RQ, ps = tid / RQ;
  const uint32_t bstep = (uint32_t)(BK * p.Cout * 4);
  // When Wo % 32 == 0 the 32 pixels of a K' step lie in one image row: (n, oh) and the row part of the
  // reflect/address arithmetic are shared by the thread's AP slots.
  const bool row_aligned = !p.adjoint && (p.Wo % BK) == 0;

  const int wg = sk_logical_wg();
  int unit = wg * p.sk_len;
  int unit_end = unit + p.sk_len;
  if (unit_end > p.sk_units) unit_end = p.sk_units;
  while (unit < unit_end) {                       // one segment = one tile's K' range [kt_begin, kt_end)
  const int tile = unit / p.ktiles_total;
  const int kt_begin = unit - tile * p.ktiles_total;
  int kt_end = kt_begin + (unit_end - unit);
  if (kt_end > p.ktiles_total) kt_end = p.ktiles_total;
  const int tr = tile / p.sk_ntn;
  const int zb = tr / p.sk_ntr_pb;                 // batch (0 unless this is a batched launch)
  const int r0 = (tr - zb * p.sk_ntr_pb) * BM, n0 = (tile - tr * p.sk_ntn) * BN;
  const __amdgpu_buffer_rsrc_t ra = make_srd(p.a + (size_t)zb * p.a_bstride, p.a_bytes),
                               rb = make_srd(p.b + (size_t)zb * p.b_bstride, p.b_bytes);

  // this thread's K-row quad (fixed for the segment)
  const int R = r0 + rq * 4;
  const bool rv = R < p.K;
  int kh, kw, ii, jj, c;
  decode_tap(p, (uint32_t)(R >> 2), kh, kw, ii, jj, c);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 va[AP], vb[BE];

  // Pixel coordinates of this thread's K' slots, advanced incrementally (32 pixels per step): the two
  // divisions per pixel per step of a from-scratch decode were most of this kernel's 8 VALU ops per MFMA.
  uint32_t sm[AP];
  int sn[AP], soh[AP], sow[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) {
    uint32_t m = (uint32_t)(kt_begin * BK + ps + PS * a);
    uint32_t n = fd_div(m, p.fd_howo);
    uint32_t rem = m - n * (uint32_t)(p.Ho * p.Wo);
    uint32_t oh = fd_div(rem, p.fd_wo);
    sm[a] = m; sn[a] = (int)n; soh[a] = (int)oh; sow[a] = (int)(rem - oh * (uint32_t)p.Wo);
  }
  uint32_t boff[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    int idx = tid + NT * e;
    int pp = idx / (BN / 4), j4 = idx % (BN / 4);
    const int co = n0 + j4 * 4;
    // rows past M fall off the end of the buffer (b_bytes = M * Cout * 4) and read as zeros
    boff[e] = co < p.Cout ? (uint32_t)(((kt_begin * BK + pp) * p.Cout + co) * 4) : VCG_OOB;
  }

  auto load_tiles = [&](int /*kt: tiles are visited strictly in order*/) {
    if (row_aligned) {
      const int n = sn[0], oh = soh[0];
      int ih = oh * p.stride - p.pad + kh;
      bool okr = rv && sm[0] < (uint32_t)p.M;            // M % 32 == 0 here, so all slots agree
      if (p.reflect) ih = reflect_idx(ih, p.Hl);
      else okr = okr && ih >= 0 && ih < p.Hl;
      const uint32_t rowbase = (uint32_t)(((n * p.H + ih * p.ups + ii) * p.W + jj) * p.Cin + c) * 4u;
      const uint32_t colstep = (uint32_t)(p.ups * p.Cin) * 4u;
#pragma unroll
      for (int a = 0; a < AP; ++a) {
        int iw = sow[a] * p.stride - p.pad + kw;
        bool ok = okr;
        if (p.reflect) iw = reflect_idx(iw, p.Wl);
        else ok = ok && iw >= 0 && iw < p.Wl;
        va[a] = bload4(ra, ok ? rowbase + (uint32_t)iw * colstep : VCG_OOB);
        sm[a] += BK;
        sow[a] += BK;
        if (sow[a] >= p.Wo) { sow[a] -= p.Wo; ++soh[a]; if (soh[a] >= p.Ho) { soh[a] = 0; ++sn[a]; } }
      }
    } else {
#pragma unroll
      for (int a = 0; a < AP; ++a) {
        const int n = sn[a], oh = soh[a], ow = sow[a];
        if (p.adjoint) {
          // swapped roles: this K' pixel is an INPUT pixel; its row entries come from the 4-channel dy
          float4 v = f4zero();
          if (rv && sm[a] < (uint32_t)p.M) {
            float4 ex;
            adjoint_gather(p, p.a, n * p.Ho, oh, ow, kh, kw, 0, 0, v, ex);
            f4add(v, ex);
          }
          va[a] = v;
        } else {
          int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
          bool ok = rv && sm[a] < (uint32_t)p.M;
          if (p.reflect) {
            ih = reflect_idx(ih, p.Hl);
            iw = reflect_idx(iw, p.Wl);
          } else {
            ok = ok && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
          }
          const uint32_t off = ok ? (uint32_t)(((n * p.H + ih * p.ups + ii) * p.W + (iw * p.ups + jj)) * p.Cin + c) * 4u : VCG_OOB;
          va[a] = bload4(ra, off);
        }
        sm[a] += BK;
        sow[a] += BK;
        while (sow[a] >= p.Wo) { sow[a] -= p.Wo; ++soh[a]; }
        while (soh[a] >= p.Ho) { soh[a] -= p.Ho; ++sn[a]; }
      }
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      vb[e] = bload4(rb, boff[e]);
      if (boff[e] != VCG_OOB) boff[e] += bstep;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int a = 0; a < AP; ++a)
      *reinterpret_cast<float4*>(&Xs[buf][(ps + PS * a) * BM + rq * 4]) = va[a];
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      int idx = tid + NT * e;
      int pp = idx / (BN / 4), j4 = idx % (BN / 4);
      *reinterpret_cast<float4*>(&Ds[buf][pp * BN + j4 * 4]) = vb[e];
    }
  };

  if (kt_begin < kt_end) {
    load_tiles(kt_begin);
    store_tiles(0);
  }
  __syncthreads();
  VCG_STAMP_AT(1);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const float* const Xc = Xs[cur];
    const float* const Dc = Ds[cur];
    if (kt + 1 < kt_end) load_tiles(kt + 1);
    mma_ktile<MI, NI>(
        acc, [&](int kk, int i) { return Xc[kk * BM + wm * (BM / WR) + i * 32 + l31]; },
        [&](int kk, int j) { return Dc[kk * BN + wn * (BN / 2) + j * 32 + l31]; }, lh);
    if (kt + 1 < kt_end) store_tiles(cur ^ 1);
    __syncthreads();
  }

  VCG_STAMP_AT(2);
  const int part = wg - (int)fd_div((uint32_t)(tile * p.ktiles_total), p.fd_sklen);
  float* slab = p.out + ((size_t)part * p.nbatch + zb) * p.K * p.Cout;      // slab[part][batch][K][Cout]
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + wn * (BN / 2) + j * 32 + l31;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int Rr = r0 + wm * (BM / WR) + i * 32 + row;
        if (Rr < p.K) slab[(size_t)Rr * p.Cout + co] = acc[i][j][e];
      }
    }
  }
  unit += kt_end - kt_begin;
  if (unit < unit_end) __syncthreads();           // the next segment's prologue overwrites LDS half 0
  }
  VCG_STAMP_AT(3);
}

// The weight gradient on the split-operand path.  Here the reduction index (the pixel) is the SLOW index of both
// operands — x and dy are [pixel][channel] — while the MFMA wants 8 consecutive reduction indices per lane.  The
// tiles are therefore staged as they come, [32 pixels][128 channels] fp16 per piece (256-byte rows, 16-byte chunks
// XOR-swizzled by ((row & 3) << 2) | ((row >> 2) & 3)), and read with gfx950's transposing LDS read
// ds_read_b64_tr_b16: per 16-lane group it takes a 4-row x 16-column block and hands each lane one column's four
// rows — two such reads give a lane its 8 reduction indices.  Everything else (stream-K segments, gather, slabs) is
// k_conv_wgrad's.
template <int COLS>
__device__ __forceinline__ uint32_t tr_off(int row, int col) {       // byte offset of element (row, col) in a [rows][COLS] fp16 image
  const int f = (((row & 3) << 2) | ((row >> 2) & 3)) & (COLS / 8 - 1);
  return (uint32_t)(2 * COLS * row + 16 * ((col >> 3) ^ f) + (col & 7) * 2);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int COLS>
__device__ __forceinline__ f16x8 tr_frag(const unsigned char* img, int row0, int col) {   // rows row0 .. row0+7 of column `col`
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + tr_off<COLS>(row0, col)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + tr_off<COLS>(row0 + 4, col)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(f16x8, v);
}

template <int BN>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_split(ConvP p) {
  VCG_STAMP_AT(0);
  constexpr int BM = 128, NT = 256;
  constexpr int WR = NT / 128;                                 // wave rows (x 2 wave columns)
  constexpr int MI = BM / (32 * WR), NI = BN / 64;
  constexpr int RQ = BM / 4, PS = NT / RQ, AP = BK / PS, BE = (BK * BN / 4) / NT;
  // TWO images of each tile (round 4: 64 KB per workgroup at BN = 128, two workgroups per CU): tile kt + 1 is split and stored into
  // the other image right behind the issue of tile kt's MFMAs — VALU and LDS stores run under the matrix pipe — and ONE barrier
  // per K' step publishes it; with one image the stores waited behind a barrier for every wave's MFMAs to drain and a second
  // barrier followed them (SQ_WAIT_ANY 55 % of the wave cycles, matrix pipe 22 % busy: gpurun_out/r04a_pmc_step.txt)
  __shared__ __attribute__((aligned(16))) unsigned char Xs[2][VCG_NP][BK * BM * 2];   // [image][piece][32 pixels][128 rows of dW] fp16
  __shared__ __attribute__((aligned(16))) unsigned char Ds[2][VCG_NP][BK * BN * 2];   // [image][piece][32 pixels][128 columns] fp16
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
  const SplitScales sc = split_scales(p);
  const int rq = tid % RQ, ps = tid / RQ;
  const uint32_t bstep = (uint32_t)(BK * p.Cout * 4);
  // When Wo % 32 == 0 the 32 pixels of a K' step lie in one image row: (n, oh) and the row part of the
  // reflect/address arithmetic are shared by the thread's AP slots.
  const bool row_aligned = !p.adjoint && (p.Wo % BK) == 0;

  const int wg = sk_logical_wg();
  int unit = wg * p.sk_len;
  int unit_end = unit + p.sk_len;
  if (unit_end > p.sk_units) unit_end = p.sk_units;
  while (unit < unit_end) {                       // one segment = one tile's K' range [kt_begin, kt_end)
  const int tile = unit / p.ktiles_total;
  const int kt_begin = unit - tile * p.ktiles_total;
  int kt_end = kt_begin + (unit_end - unit);
  if (kt_end > p.ktiles_total) kt_end = p.ktiles_total;
  const int tr = tile / p.sk_ntn;
  const int zb = tr / p.sk_ntr_pb;                 // batch (0 unless this is a batched launch)
  const int r0 = (tr - zb * p.sk_ntr_pb) * BM, n0 = (tile - tr * p.sk_ntn) * BN;
  const __amdgpu_buffer_rsrc_t ra = make_srd(p.a + (size_t)zb * p.a_bstride, p.a_bytes),
                               rb = make_srd(p.b + (size_t)zb * p.b_bstride, p.b_bytes);

  // this thread's K-row quad (fixed for the segment)
  const int R = r0 + rq * 4;
  const bool rv = R < p.K;
  int kh, kw, ii, jj, c;
  decode_tap(p, (uint32_t)(R >> 2), kh, kw, ii, jj, c);

  f32x16 acc[MI][NI], lo[MI][NI];               // h*h chain and cross-term chain
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  float4 va[AP], vb[BE];

  // Pixel coordinates of this thread's K' slots, advanced incrementally (32 pixels per step): the two
  // divisions per pixel per step of a from-scratch decode were most of this kernel's 8 VALU ops per MFMA.
  uint32_t sm[AP];
  int sn[AP], soh[AP], sow[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) {
    uint32_t m = (uint32_t)(kt_begin * BK + ps + PS * a);
    uint32_t n = fd_div(m, p.fd_howo);
    uint32_t rem = m - n * (uint32_t)(p.Ho * p.Wo);
    uint32_t oh = fd_div(rem, p.fd_wo);
    sm[a] = m; sn[a] = (int)n; soh[a] = (int)oh; sow[a] = (int)(rem - oh * (uint32_t)p.Wo);
  }
  uint32_t boff[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    int idx = tid + NT * e;
    int pp = idx / (BN / 4), j4 = idx % (BN / 4);
    const int co = n0 + j4 * 4;
    // rows past M fall off the end of the buffer (b_bytes = M * Cout * 4) and read as zeros
    boff[e] = co < p.Cout ? (uint32_t)(((kt_begin * BK + pp) * p.Cout + co) * 4) : VCG_OOB;
  }

  auto load_tiles = [&](int /*kt: tiles are visited strictly in order*/) {
    if (p.a_planes) {
      // the batched 1x1 geometry of the Winograd path (N = 1, H = 1, W = T, Cin = K): pixel = tile index sow, row quad c = R.
      // The quad's two pieces sit 64 bytes apart in its 128-byte (tile, 32-k block) group; they travel as the two halves of va
      const uint32_t cblk = (uint32_t)(R >> 5) * VCG_PBYTES + (uint32_t)(R & 31) * 2u;
      const uint32_t rowb = (uint32_t)(p.K >> 5) * VCG_PBYTES;
#pragma unroll
      for (int a = 0; a < AP; ++a) {
        const bool ok = rv && sm[a] < (uint32_t)p.M;
        const uint32_t off = ok ? sm[a] * rowb + cblk : VCG_OOB;
        const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(ra, (int)off, 0, 0);
        const u32x2 l = __builtin_amdgcn_raw_buffer_load_b64(ra, (int)(ok ? off + 64u : VCG_OOB), 0, 0);
        va[a] = make_float4(__uint_as_float(h.x), __uint_as_float(h.y), __uint_as_float(l.x), __uint_as_float(l.y));
        sm[a] += BK;
      }
    } else if (row_aligned) {
      const int n = sn[0], oh = soh[0];
      int ih = oh * p.stride - p.pad + kh;
      bool okr = rv && sm[0] < (uint32_t)p.M;            // M % 32 == 0 here, so all slots agree
      if (p.reflect) ih = reflect_idx(ih, p.Hl);
      else okr = okr && ih >= 0 && ih < p.Hl;
      const uint32_t rowbase = (uint32_t)(((n * p.H + ih * p.ups + ii) * p.W + jj) * p.Cin + c) * 4u;
      const uint32_t colstep = (uint32_t)(p.ups * p.Cin) * 4u;
#pragma unroll
      for (int a = 0; a < AP; ++a) {
        int iw = sow[a] * p.stride - p.pad + kw;
        bool ok = okr;
        if (p.reflect) iw = reflect_idx(iw, p.Wl);
        else ok = ok && iw >= 0 && iw < p.Wl;
        va[a] = bload4(ra, ok ? rowbase + (uint32_t)iw * colstep : VCG_OOB);
        sm[a] += BK;
        sow[a] += BK;
        if (sow[a] >= p.Wo) { sow[a] -= p.Wo; ++soh[a]; if (soh[a] >= p.Ho) { soh[a] = 0; ++sn[a]; } }
      }
    } else {
#pragma unroll
      for (int a = 0; a < AP; ++a) {
        const int n = sn[a], oh = soh[a], ow = sow[a];
        if (p.adjoint) {
          // swapped roles: this K' pixel is an INPUT pixel; its row entries come from the 4-channel dy
          float4 v = f4zero();
          if (rv && sm[a] < (uint32_t)p.M) {
            float4 ex;
            adjoint_gather(p, p.a, n * p.Ho, oh, ow, kh, kw, 0, 0, v, ex);
            f4add(v, ex);
          }
          va[a] = v;
        } else {
          int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
          bool ok = rv && sm[a] < (uint32_t)p.M;
          if (p.reflect) {
            ih = reflect_idx(ih, p.Hl);
            iw = reflect_idx(iw, p.Wl);
          } else {
            ok = ok && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
          }
          const uint32_t off = ok ? (uint32_t)(((n * p.H + ih * p.ups + ii) * p.W + (iw * p.ups + jj)) * p.Cin + c) * 4u : VCG_OOB;
          va[a] = bload4(ra, off);
        }
        sm[a] += BK;
        sow[a] += BK;
        while (sow[a] >= p.Wo) { sow[a] -= p.Wo; ++soh[a]; }
        while (soh[a] >= p.Ho) { soh[a] -= p.Ho; ++sn[a]; }
      }
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      vb[e] = bload4(rb, boff[e]);
      if (boff[e] != VCG_OOB) boff[e] += bstep;
    }
  };
  auto store_tiles = [&](int img) {
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      uint2 h, l;
      if (p.a_planes) {
        h = make_uint2(__float_as_uint(va[a].x), __float_as_uint(va[a].y));
        l = make_uint2(__float_as_uint(va[a].z), __float_as_uint(va[a].w));
      } else {
        split4h(va[a], sc.inv_a, h, l);
      }
      const uint32_t o = tr_off<BM>(ps + PS * a, rq * 4);
      *reinterpret_cast<uint2*>(&Xs[img][0][o]) = h;
      *reinterpret_cast<uint2*>(&Xs[img][1][o]) = l;
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      const int idx = tid + NT * e;
      const int pp = idx / (BN / 4), j4 = idx % (BN / 4);
      uint2 h, l;
      split4h(vb[e], sc.inv_b, h, l);
      const uint32_t o = tr_off<BN>(pp, j4 * 4);
      *reinterpret_cast<uint2*>(&Ds[img][0][o]) = h;
      *reinterpret_cast<uint2*>(&Ds[img][1][o]) = l;
    }
  };
  // transposing fragment reads: 16-lane group g = lane >> 4 handles columns 16 (g & 1) .. +15 of a 32-wide MFMA tile and
  // the reduction rows of half lh = g >> 1; inside it lane 4q + p addresses row q, columns 4p .. 4p+3
  const int tq = (lane & 15) >> 2, tp = lane & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * tp;

  if (kt_begin < kt_end) {
    load_tiles(kt_begin);
    store_tiles(0);
  }
  __syncthreads();
  VCG_STAMP_AT(1);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int img = (kt - kt_begin) & 1;
    if (kt + 1 < kt_end) load_tiles(kt + 1);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int row0 = 16 * s2 + 8 * lh + tq;
      f16x8 af[VCG_NP][MI], bfr[VCG_NP][NI];
#pragma unroll
      for (int pc = 0; pc < VCG_NP; ++pc) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[pc][i] = tr_frag<BM>(Xs[img][pc], row0, wm * (BM / WR) + i * 32 + tcol);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[pc][j] = tr_frag<BN>(Ds[img][pc], row0, wn * (BN / 2) + j * 32 + tcol);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          f32x16 c = lo[i][j];
          c = VCG_MFMA(af[1][i], bfr[0][j], c);
          c = VCG_MFMA(af[0][i], bfr[1][j], c);
          lo[i][j] = c;
          acc[i][j] = VCG_MFMA(af[0][i], bfr[0][j], acc[i][j]);
        }
    }
    // the other image was last read in step kt - 1, and every wave has passed the barrier that ended it
    if (kt + 1 < kt_end) store_tiles(img ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] += lo[i][j];
  VCG_STAMP_AT(2);
  const int part = wg - (int)fd_div((uint32_t)(tile * p.ktiles_total), p.fd_sklen);
  float* slab = p.out + ((size_t)part * p.nbatch + zb) * p.K * p.Cout;      // slab[part][batch][K][Cout]
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + wn * (BN / 2) + j * 32 + l31;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int Rr = r0 + wm * (BM / WR) + i * 32 + row;
        if (Rr < p.K) slab[(size_t)Rr * p.Cout + co] = acc[i][j][e] * sc.out;
      }
    }
  }
  unit += kt_end - kt_begin;
  if (unit < unit_end) __syncthreads();           // the next segment's prologue overwrites LDS image 0
  }
  VCG_STAMP_AT(3);
}

// number of partial sums the stream-K split left for the tile that holds element (R, co)
__device__ __forceinline__ int sk_parts(const ConvP& p, int R, int co) {
  const int tile = (R >> p.sk_bm_shift) * p.sk_ntn + (co >> p.sk_bn_shift);
  const uint32_t u0 = (uint32_t)(tile * p.ktiles_total);
  return (int)(fd_div(u0 + (uint32_t)p.ktiles_total - 1u, p.fd_sklen) - fd_div(u0, p.fd_sklen)) + 1;
}

// stage 0 when there are many slabs: out[g][idx] = sum over the g-th group of slabs (fixed order),
// fully parallel over elements and groups
__global__ __launch_bounds__(256) void k_slab_sum(const float* __restrict__ slabs, float* __restrict__ out,
                                                  size_t total4, ConvP p, int per_group) {
  const int g = blockIdx.y;
  const int z0 = g * per_group;
  const int c4 = p.Cout / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int R = (int)(i / c4), co = (int)(i - (size_t)R * c4) * 4;
    int z1 = z0 + per_group;
    const int np = sk_parts(p, R, co);            // slabs past a tile's part count were never written
    if (z1 > np) z1 = np;
    float4 s = f4zero();
    for (int z = z0; z < z1; ++z) f4add(s, reinterpret_cast<const float4*>(slabs)[(size_t)z * total4 + i]);
    reinterpret_cast<float4*>(out)[(size_t)g * total4 + i] = s;
  }
}

// small weights: one thread per (R, co), scattered OIHW read-modify-write (irrelevant at this size)
// nsplit > 0: that many fully written slabs (pre-summed groups); nsplit == 0: the tile's own part count
__global__ __launch_bounds__(256) void k_wgrad_scatter(const float* __restrict__ slabs, float* __restrict__ gw,
                                                       ConvP p, int nsplit, int cin_log, int cout_log) {
  const size_t total = (size_t)p.K * p.Cout;
  const int U2 = p.ups * p.ups, KK = p.KH * p.KW;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t R = (uint32_t)(idx / p.Cout);
    const int co = (int)(idx - (size_t)R * p.Cout);
    const uint32_t t = R / (uint32_t)p.Cin;
    const int c = (int)(R - t * (uint32_t)p.Cin);
    if (co >= cout_log || c >= cin_log) continue;
    const int tap9 = (int)t / U2, ph = (int)t - tap9 * U2;
    const int nz = nsplit ? nsplit : sk_parts(p, (int)R, co);
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += slabs[(size_t)z * total + idx];
    gw[((size_t)co * (cin_log * U2) + (size_t)c * U2 + ph) * KK + tap9] += s;
  }
}

// swapped-role wgrad: slabs[z][(tap, j)][c] -> gw_oihw[j][c][tap] += sum_z
__global__ __launch_bounds__(256) void k_wgrad_scatter_swapped(const float* __restrict__ slabs, float* __restrict__ gw,
                                                               ConvP p, int T, int C, int nsplit, int cin_real,
                                                               int cout_real) {
  const size_t total = (size_t)T * 4 * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int R = (int)(idx / C);
    const int t = R >> 2, j = R & 3;
    if (j >= cout_real || c >= cin_real) continue;
    const int nz = nsplit ? nsplit : sk_parts(p, R, c);
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += slabs[(size_t)z * total + idx];
    gw[((size_t)j * cin_real + c) * T + t] += s;
  }
}

// slabs[z][K][Cout] -> gw_oihw += sum_z (fixed order), transposed through LDS.
// A block owns 32 output channels x 8 input channels x ALL taps: slab reads are 128-B row segments
// (co fastest), and for every co the block's OIHW target [8 c][U*U][KH*KW] is one contiguous run,
// so the read-modify-write of the gradient is coalesced too.  Dynamic LDS: T*8*33 floats.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ gw,
                                                      ConvP p, int nsplit, int cin_log, int cout_log) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int U2 = p.ups * p.ups, KK = p.KH * p.KW, T = KK * U2;
  const int co0 = blockIdx.x * 32, c0 = blockIdx.y * 8;
  const int cl_ = threadIdx.x >> 5, col = threadIdx.x & 31;      // read mapping: 8 c x 32 co
  const size_t total = (size_t)p.K * p.Cout;
  {
    const int c = c0 + cl_, co = co0 + col;
    const bool ok = c < p.Cin && co < p.Cout;
    for (int t = 0; t < T; ++t) {
      float s = 0.f;
      if (ok) {
        const size_t idx = ((size_t)t * p.Cin + c) * p.Cout + co;
        const int nz = nsplit ? nsplit : sk_parts(p, t * p.Cin + c, co);
        for (int z = 0; z < nz; ++z) s += slabs[(size_t)z * total + idx];
      }
      tile[(t * 8 + cl_) * 33 + col] = s;
    }
  }
  __syncthreads();
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cinL = cin_log * U2;
  const int run = 8 * U2 * KK;                                   // OIHW elements per co in this block
  // a wave's 8 output channels x `run` elements as one flat index space (full trips of 64 lanes; see k_wino_wgrad_reduce)
  for (int idx = lane; idx < 8 * run; idx += 64) {
    const int j = idx / run, q = idx - j * run;
    const int colw = wid * 8 + j, co = co0 + colw;
    const int clq = q / KK, tap9 = q - clq * KK;                   // clq = c_local*U2 + phase
    const int c_local = clq / U2, ph = clq - c_local * U2;
    const int c = c0 + c_local;
    if (co >= cout_log || c >= cin_log) continue;
    const int t = tap9 * U2 + ph;
    const size_t o = ((size_t)co * cinL + (size_t)c * U2 + ph) * KK + tap9;
    gw[o] += tile[(t * 8 + c_local) * 33 + colw];
  }
}

// Winograd weight gradient, last stage: slabs[part][xi][k][co] hold dU = sum_t V^T dM per transform point xi.
// dg (3x3) = G^T dU G is the adjoint of U = G g G^T; the sum over parts runs in fixed order first.  A block owns
// 64 co x 8 c of ONE unshuffle phase (blockIdx.z); a thread owns two neighbouring co (8-byte loads: a wave row is a
// 256-byte segment of a slab row) and keeps the 16 transform points of a part as 16 independent loads in flight (one
// dependent load at a time made the D2 reduce latency-bound: 227 us for 42 MB).  The OIHW read-modify-write goes
// through LDS as in k_wgrad_reduce (9-float runs per (co, c) when ups == 2).  Cout % 64 == 0 (vcg_wino_weight_ok).
__global__ __launch_bounds__(256) void k_wino_wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ gw, ConvP p,
                                                           int Cin, int ups, int cin_log, int cout_log) {
  __shared__ float tile[9 * 8 * 65];
  const int U2 = ups * ups, KK = 9;
  const int co0 = blockIdx.x * 64, c0 = blockIdx.y * 8, ph = blockIdx.z;
  const int cl_ = threadIdx.x >> 5, col = (threadIdx.x & 31) * 2;
  const size_t plane = (size_t)p.nbatch * p.K * p.Cout;          // one part: [16][Kc][Cout]
  {
    const int c = c0 + cl_, co = co0 + col;
    const bool ok = c < Cin;
    float2 s[16];
    int nz[16], nzmax = 0;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      nz[xi] = ok ? sk_parts(p, xi * p.K + ph * Cin + c, co) : 0;
      nzmax = nz[xi] > nzmax ? nz[xi] : nzmax;
      s[xi] = make_float2(0.f, 0.f);
    }
    const float* src = slabs + (size_t)(ph * Cin + (ok ? c : 0)) * p.Cout + co;
    const size_t xstride = (size_t)p.K * p.Cout;
    for (int z = 0; z < nzmax; ++z) {                              // parts in fixed order; adding 0 past a tile's count
      float2 v[16];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi)
        v[xi] = z < nz[xi] ? *reinterpret_cast<const float2*>(src + (size_t)z * plane + (size_t)xi * xstride) : make_float2(0.f, 0.f);
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) { s[xi].x += v[xi].x; s[xi].y += v[xi].y; }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float h[3][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float s0 = e ? s[0 + b].y : s[0 + b].x, s1 = e ? s[4 + b].y : s[4 + b].x;
        const float s2 = e ? s[8 + b].y : s[8 + b].x, s3 = e ? s[12 + b].y : s[12 + b].x;
        h[0][b] = s0 + 0.5f * (s1 + s2);
        h[1][b] = 0.5f * (s1 - s2);
        h[2][b] = 0.5f * (s1 + s2) + s3;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        tile[((a * 3 + 0) * 8 + cl_) * 65 + col + e] = h[a][0] + 0.5f * (h[a][1] + h[a][2]);
        tile[((a * 3 + 1) * 8 + cl_) * 65 + col + e] = 0.5f * (h[a][1] - h[a][2]);
        tile[((a * 3 + 2) * 8 + cl_) * 65 + col + e] = 0.5f * (h[a][1] + h[a][2]) + h[a][3];
      }
    }
  }
  __syncthreads();
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cinL = cin_log * U2;
  // a wave's 16 output channels x 72 (c, tap) elements as ONE flat index space: 18 full trips of 64 lanes (round 4; per channel
  // the 72-element run took two trips with 8 lanes of the second one active)
  for (int idx = lane; idx < 16 * 8 * KK; idx += 64) {
    const int j = idx / (8 * KK), q = idx - j * (8 * KK);
    const int colw = wid * 16 + j, co = co0 + colw;
    const int c_local = q / KK, tap9 = q - c_local * KK;
    const int c = c0 + c_local;
    if (co >= cout_log || c >= cin_log) continue;
    gw[((size_t)co * cinL + (size_t)c * U2 + ph) * KK + tap9] += tile[(tap9 * 8 + c_local) * 65 + colw];
  }
}

// OIHW -> Wf[K][Cout] (pad rows / pad columns are zero)
// (both Wf pack kernels also leave the pack's header: the bit pattern of the kernel's largest magnitude — p.amax_b, measured by the
// k_absmax launch at the head of vcg_pack_weight — where the kernels that multiply by the pack's planes read it.  Round 3: this was
// a launch of its own, 44 per step)
__device__ __forceinline__ void pack_header(const ConvP& p, uint32_t* hdr) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {       // wave 0, all lanes: vcg_amax_bits is a wave-wide maximum
    const uint32_t b = vcg_amax_bits(p.amax_b);
    if (threadIdx.x == 0) *hdr = b;
  }
}
__global__ void k_pack_weight(const float* __restrict__ w, float* __restrict__ wf, ConvP p, int cin_log,
                              int cout_log, uint32_t* hdr) {
  pack_header(p, hdr);
  const size_t total = (size_t)p.K * p.Cout;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    uint32_t R = (uint32_t)(idx / p.Cout);
    int co = (int)(idx - (size_t)R * p.Cout);
    uint32_t tap = R / (uint32_t)p.Cin;
    int c = (int)(R - tap * (uint32_t)p.Cin);
    float v = 0.f;
    if (co < cout_log && c < cin_log) {
      int ii = 0, jj = 0;
      if (p.ups == 2) { jj = tap & 1; ii = (tap >> 1) & 1; tap >>= 2; }
      int kh = (int)tap / p.KW, kw = (int)tap % p.KW;
      int cl = (p.ups == 2) ? (c * 4 + ii * 2 + jj) : c;
      int cinL = (p.ups == 2) ? cin_log * 4 : cin_log;
      v = w[(((size_t)co * cinL + cl) * p.KH + kh) * p.KW + kw];
    }
    wf[idx] = v;
  }
}

// The same repack for large weights, transposed through LDS (the inverse of k_wgrad_reduce): a block owns 32 output
// channels x 8 input channels x all taps; for every co its OIHW source [8 c][U*U][KH*KW] is one contiguous run, and
// the Wf rows it writes are 128-B segments (co fastest).  The one-thread-per-element kernel reads OIHW at a stride of
// Cin*KH*KW floats: 0.7 TB/s.  Dynamic LDS: T*8*33 floats.
__global__ __launch_bounds__(256) void k_pack_weight_t(const float* __restrict__ w, float* __restrict__ wf, ConvP p,
                                                       int cin_log, int cout_log, uint32_t* hdr) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  pack_header(p, hdr);
  const int U2 = p.ups * p.ups, KK = p.KH * p.KW, T = KK * U2;
  const int co0 = blockIdx.x * 32, c0 = blockIdx.y * 8;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cinL = cin_log * U2;
  const int run = 8 * U2 * KK;
  for (int j = 0; j < 8; ++j) {
    const int colw = wid * 8 + j, co = co0 + colw;
    for (int q = lane; q < run; q += 64) {
      const int clq = q / KK, tap9 = q - clq * KK;                 // clq = c_local*U2 + phase
      const int c_local = clq / U2, ph = clq - c_local * U2;
      const int c = c0 + c_local;
      const int t = tap9 * U2 + ph;
      float v = 0.f;
      if (co < cout_log && c < cin_log) v = w[((size_t)co * cinL + (size_t)c * U2 + ph) * KK + tap9];
      tile[(t * 8 + c_local) * 33 + colw] = v;
    }
  }
  __syncthreads();
  const int cl_ = threadIdx.x >> 5, col = threadIdx.x & 31;      // write mapping: 8 c x 32 co
  const int c = c0 + cl_, co = co0 + col;
  if (c < p.Cin && co < p.Cout)
    for (int t = 0; t < T; ++t) wf[((size_t)t * p.Cin + c) * p.Cout + co] = tile[(t * 8 + cl_) * 33 + col];
}

// OIHW -> the pre-split ("blocked planes", gemm_split.hip) weight operands of the direct split-operand kernels:
//   WFT planes  [Cout][KB][3][32], KB = ceil(K / 32), zero padded: the transpose of Wf, B^T of k_conv_fwd_split;
//               one thread per (co, 4 consecutive k = one tap, 4 channels)
//   WFD planes  [(kh, kw)][J][Cout / 32][3][32], J = (phase, c): the rows of Wf as k_conv_dgrad_split reads them (reduction
//               index co); one thread per ((tap, J), 4 consecutive co)
template <bool DGRAD>
__global__ __launch_bounds__(256) void k_pack_planes(const float* __restrict__ w, unsigned short* __restrict__ bp, ConvP p, int cin_log,
                                                     int cout_log) {
  float psc, pinv;                                                  // the planes hold w / s, s from the kernel's amax (p.amax_b)
  vcg_scale_of(vcg_amax_bits(p.amax_b), p.amax_b.shift, psc, pinv);
  const int U2 = p.ups * p.ups, KK = p.KH * p.KW;
  const int cinL = cin_log * U2;
  const int KB = (p.K + 31) / 32, CB = p.Cout / 32;
  const size_t total = DGRAD ? (size_t)KK * U2 * p.Cin * (p.Cout / 4) : (size_t)p.Cout * KB * 8;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned short* o;
    if (DGRAD) {
      const size_t row = idx / (p.Cout / 4);                     // (tapfull, J)
      const int co0 = (int)(idx - row * (p.Cout / 4)) * 4;
      const int NB = U2 * p.Cin;
      const int tapfull = (int)(row / NB), J = (int)(row - (size_t)tapfull * NB);
      const int q = J / p.Cin, c = J - q * p.Cin;
      const int kh = tapfull / p.KW, kw = tapfull - kh * p.KW;
      if (c < cin_log) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co0 + e < cout_log) v[e] = w[(((size_t)(co0 + e) * cinL + (size_t)c * U2 + q) * p.KH + kh) * p.KW + kw];
      }
      o = bp + (row * CB + co0 / 32) * VCG_PBLK + (co0 & 31);
    } else {
      const int co = (int)(idx / ((size_t)KB * 8));
      const int k0 = (int)(idx - (size_t)co * KB * 8) * 4;       // 4 consecutive k: one tap, channels c .. c + 3
      if (co < cout_log && k0 < p.K) {
        uint32_t tap = (uint32_t)k0 / (uint32_t)p.Cin;
        const int c = k0 - (int)tap * p.Cin;
        int ii = 0, jj = 0;
        if (p.ups == 2) { jj = tap & 1; ii = (tap >> 1) & 1; tap >>= 2; }
        const int kh = (int)tap / p.KW, kw = (int)tap % p.KW;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < cin_log) {
            const int cl = (p.ups == 2) ? ((c + e) * 4 + ii * 2 + jj) : c + e;
            v[e] = w[(((size_t)co * cinL + cl) * p.KH + kh) * p.KW + kw];
          }
      }
      o = bp + ((size_t)co * KB + k0 / 32) * VCG_PBLK + (k0 & 31);
    }
    uint2 h, l;
    split4h(make_float4(v[0], v[1], v[2], v[3]), pinv, h, l);
    *reinterpret_cast<uint2*>(o) = h;
    *reinterpret_cast<uint2*>(o + 32) = l;
  }
}

// gbias[co] += sum_m dy[m][co]: per-chunk partials then a fixed-order final sum
// float4 per lane (TC channel quads x TP row lanes per block), 4 independent rows in flight per lane
// The last chunk block of a channel group (arrival counter `ctr`, vcg_common.h / VcgInTail) sums the group's chunk partials
// in a fixed order and adds them to gbias: no separate finalize launch.  ctr == null: the caller runs k_colsum_final.
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ dy, float* __restrict__ part,
                                                        int M, int C, int rows_per_chunk, int TC, uint32_t* ctr,
                                                        float* __restrict__ gbias, int c_log) {
  __shared__ float4 red[256];
  __shared__ int last;
  const int TP = 256 / TC;
  const int tc = threadIdx.x % TC, tp = threadIdx.x / TC;
  const int c4 = blockIdx.x * TC + tc;
  const int mb = blockIdx.y * rows_per_chunk;
  int me = mb + rows_per_chunk;
  if (me > M) me = M;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s, s3 = s, s4 = s;
  if (c4 * 4 < C) {
    const float* base = dy + (size_t)c4 * 4;
    int m = mb + tp;
    for (; m + 3 * TP < me; m += 4 * TP) {
      float4 a = ldg4(base + (size_t)m * C), b = ldg4(base + (size_t)(m + TP) * C);
      float4 c = ldg4(base + (size_t)(m + 2 * TP) * C), d = ldg4(base + (size_t)(m + 3 * TP) * C);
      f4add(s, a); f4add(s2, b); f4add(s3, c); f4add(s4, d);
    }
    for (; m < me; m += TP) f4add(s, ldg4(base + (size_t)m * C));
    f4add(s, s2); f4add(s3, s4); f4add(s, s3);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (tp == 0 && c4 * 4 < C) {
    for (int k = 1; k < TP; ++k) f4add(s, red[k * TC + tc]);
    float* o = part + (size_t)blockIdx.y * C + c4 * 4;
    vcg_store_sc1_f2(o, s.x, s.y);
    vcg_store_sc1_f2(o + 2, s.z, s.w);
  }
  if (!ctr) return;
  if (!vcg_last_arrival(ctr + blockIdx.x, gridDim.y, &last)) return;
  // thread (tc, tp): channel quad tc, chunks tp, tp + TP, ... — four loads in flight — then the TP lanes in order
  const int nchunk = (int)gridDim.y;
  float4 t0 = f4zero(), t1 = t0, t2 = t0, t3 = t0;
  if (c4 * 4 < C) {
    const float* base = part + (size_t)c4 * 4;
    int k = tp;
    auto ld = [](const float* q) { const float2 lo = vcg_load_sc1_f2(q), hi = vcg_load_sc1_f2(q + 2); return make_float4(lo.x, lo.y, hi.x, hi.y); };
    for (; k + 3 * TP < nchunk; k += 4 * TP) {
      const float4 v0 = ld(base + (size_t)k * C), v1 = ld(base + (size_t)(k + TP) * C);
      const float4 v2 = ld(base + (size_t)(k + 2 * TP) * C), v3 = ld(base + (size_t)(k + 3 * TP) * C);
      f4add(t0, v0); f4add(t1, v1); f4add(t2, v2); f4add(t3, v3);
    }
    for (; k < nchunk; k += TP) f4add(t0, ld(base + (size_t)k * C));
    f4add(t0, t1); f4add(t2, t3); f4add(t0, t2);
  }
  red[threadIdx.x] = t0;
  __syncthreads();
  if (tp == 0 && c4 * 4 < C) {
    for (int k = 1; k < TP; ++k) f4add(t0, red[k * TC + tc]);
    const int c = c4 * 4;
    if (c + 0 < c_log) gbias[c + 0] += t0.x;
    if (c + 1 < c_log) gbias[c + 1] += t0.y;
    if (c + 2 < c_log) gbias[c + 2] += t0.z;
    if (c + 3 < c_log) gbias[c + 3] += t0.w;
  }
}
// 8 channels x 32 chunk lanes per block: the loop over chunk partials is a dependent chain of L2 round trips
// (150 us per launch as one thread per channel, 16 us split 8 ways), so it is split 32 ways with 4 loads in
// flight per lane and combined in LDS in a fixed order
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ part, float* __restrict__ out, int C,
                                                      int nchunk, int c_log) {
  __shared__ float red[32][8];
  const int il = threadIdx.x & 7, kl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + il;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < c_log) {
    int k = kl;
    for (; k + 96 < nchunk; k += 128) {
      s0 += part[(size_t)k * C + c];
      s1 += part[(size_t)(k + 32) * C + c];
      s2 += part[(size_t)(k + 64) * C + c];
      s3 += part[(size_t)(k + 96) * C + c];
    }
    for (; k < nchunk; k += 32) s0 += part[(size_t)k * C + c];
  }
  red[kl][il] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kl == 0 && c < c_log) {
    float s = red[0][il];
    for (int k = 1; k < 32; ++k) s += red[k][il];
    out[c] += s;
  }
}

// ------------------------------------------------------------------ host side
int vcg_conv_geom(const int32_t* cd, ConvGeom* g, const char* who) {
  g->N = cd[VCG_CD_N]; g->H = cd[VCG_CD_H]; g->W = cd[VCG_CD_W];
  g->Cin = cd[VCG_CD_CIN]; g->Cout = cd[VCG_CD_COUT];
  g->KH = cd[VCG_CD_KH]; g->KW = cd[VCG_CD_KW];
  g->stride = cd[VCG_CD_STRIDE]; g->pad = cd[VCG_CD_PAD];
  g->reflect = cd[VCG_CD_REFLECT]; g->ups = cd[VCG_CD_UPS]; g->act = cd[VCG_CD_ACT];
  g->cin_log = cd[VCG_CD_CIN_LOGICAL]; g->cout_log = cd[VCG_CD_COUT_LOGICAL];
  VCG_CHECK_ARG(g->N > 0 && g->H > 0 && g->W > 0, "%s: bad N/H/W %d %d %d", who, g->N, g->H, g->W);
  VCG_CHECK_ARG(g->Cin > 0 && g->Cin % 4 == 0, "%s: Cin pitch %d must be a positive multiple of 4", who, g->Cin);
  VCG_CHECK_ARG(g->Cout > 0 && g->Cout % 4 == 0, "%s: Cout pitch %d must be a positive multiple of 4", who, g->Cout);
  VCG_CHECK_ARG(g->ups == 1 || g->ups == 2, "%s: ups must be 1 or 2", who);
  VCG_CHECK_ARG(g->stride == 1 || g->stride == 2, "%s: stride must be 1 or 2", who);
  VCG_CHECK_ARG(g->H % g->ups == 0 && g->W % g->ups == 0, "%s: H,W must be divisible by ups", who);
  VCG_CHECK_ARG(g->cin_log > 0 && g->cin_log <= g->Cin && g->cout_log > 0 && g->cout_log <= g->Cout,
                "%s: logical channels out of range", who);
  VCG_CHECK_ARG(g->act >= 0 && g->act <= VCG_ACT_SIGMOID, "%s: bad act", who);
  g->Hl = g->H / g->ups; g->Wl = g->W / g->ups;
  VCG_CHECK_ARG(g->KH > 0 && g->KW > 0 && g->pad >= 0, "%s: bad kernel/pad", who);
  VCG_CHECK_ARG(g->Hl + 2 * g->pad >= g->KH && g->Wl + 2 * g->pad >= g->KW, "%s: kernel larger than padded input", who);
  if (g->reflect)
    VCG_CHECK_ARG(g->pad < g->Hl && g->pad < g->Wl, "%s: reflect pad %d needs input > pad (got %dx%d)", who, g->pad, g->Hl, g->Wl);
  g->Ho = (g->Hl + 2 * g->pad - g->KH) / g->stride + 1;
  g->Wo = (g->Wl + 2 * g->pad - g->KW) / g->stride + 1;
  long long M = (long long)g->N * g->Ho * g->Wo;
  long long in_elems = (long long)g->N * g->H * g->W * g->Cin;
  long long out_elems = M * g->Cout;
  VCG_CHECK_ARG(M < (1ll << 30) && in_elems < (1ll << 40) && out_elems < (1ll << 40), "%s: tensor too large", who);
  g->M = (int)M;
  g->taps = g->KH * g->KW * g->ups * g->ups;
  long long K = (long long)g->taps * g->Cin;
  VCG_CHECK_ARG(K < (1ll << 30), "%s: K too large", who);
  g->K = (int)K;
  return 0;
}

static void fill_params(const ConvGeom& g, ConvP& p) {
  p.N = g.N; p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.Cout = g.Cout; p.KH = g.KH; p.KW = g.KW;
  p.stride = g.stride; p.pad = g.pad; p.reflect = g.reflect; p.ups = g.ups; p.act = g.act;
  p.Hl = g.Hl; p.Wl = g.Wl; p.Ho = g.Ho; p.Wo = g.Wo; p.M = g.M; p.K = g.K;
  p.cin4 = g.Cin / 4; p.cout4 = g.Cout / 4; p.cout_log = g.cout_log;
  p.fd_howo = make_fastdiv((uint32_t)(g.Ho * g.Wo));
  p.fd_wo = make_fastdiv((uint32_t)g.Wo);
  p.fd_cin4 = make_fastdiv((uint32_t)p.cin4);
  p.fd_cout4 = make_fastdiv((uint32_t)p.cout4);
  p.fd_kw = make_fastdiv((uint32_t)g.KW);
  p.fd_cin = make_fastdiv((uint32_t)g.Cin);
  p.Hc = p.Wc = p.Mc = p.NB = 0;
  p.fd_hcwc = make_fastdiv(1); p.fd_wc = make_fastdiv(1);
  p.nbatch = 1; p.a_bstride = p.b_bstride = 0; p.out_bstride = 0;
  p.ktiles_total = 0; p.sk_len = 1; p.sk_units = 0; p.sk_ntn = 1; p.sk_bm_shift = p.sk_bn_shift = 7; p.sk_ntr_pb = 1 << 30;
  p.fd_sklen = make_fastdiv(1);
  p.ksplit = 1; p.kt_per = 0; p.slab = nullptr; p.adjoint = 0; p.src_pitch = g.Cout;
  p.a_bytes = p.b_bytes = 0; p.dbl_mirror = 0;
  p.bias = nullptr;
  p.in_part = nullptr; p.in_nchunk = 0; p.in_tail = vcg_in_tail_none();
  p.amax_a = p.amax_b = vcg_amax_const(0);      // scale 1 (the fp32-MFMA kernels never look)
  p.a_planes = 0;
}

// Tile and K-slice choice.  The 256 CUs want >= 512 workgroups.  If the largest tile that reaches that
// exists, use it.  Otherwise (deep layers: M = N*16*16 pixels, K up to 18 432) keep the big, efficient
// tile and slice K across blockIdx.z instead of shrinking the tile: partial tiles go to fp32 slabs that
// k_splitk_finish sums in a fixed order (+ bias + activation).
// VCG_PLAN_TSCALE: multiplies the planners' time per K-step (their constants were measured on the fp32-MFMA kernels of round 1;
// the fp16 x 2 kernels step ~2x faster while a K slice's slab traffic and finish launch cost what they did) — A/B measurements
static double plan_tscale() {
  static const double v = [] { const char* e = getenv("VCG_PLAN_TSCALE"); return e ? atof(e) : 1.0; }();
  return v;
}
static void gemm_plan(long long rows, long long cols, int nkt, bool allow_split, int& bm, int& bn, int& nsplit,
                      int& kt_per, int batches = 1, bool allow_bn32 = false, bool single_level = false) {
  // cost model (us): rounds of resident workgroups x K-steps per workgroup x time per K-step of that tile,
  // plus the slab write+read of a K-sliced launch.  Same constants as wgrad_plan.
  struct Cand { int bm, bn, resident; double t_step; };
  // single_level: the 128x128 kernel without the second accumulator set keeps three workgroups per CU
  const Cand cands[5] = {{128, 128, single_level ? 3 : 2, single_level ? 5.4 : 5.1}, {128, 64, 3, 4.7}, {64, 128, 3, 4.7},
                         {64, 64, 4, 4.8}, {128, 32, 4, 3.6}};
  double best = 1e30;
  bm = 64; bn = 64; nsplit = 1; kt_per = nkt;
  for (int ci = 0; ci < 5; ++ci) {
    const Cand& c = cands[ci];
    if (c.bn == 32 && !(allow_bn32 && cols <= 32)) continue;
    if (c.bn == 128 && cols <= 64) continue;
    if (c.bm == 128 && rows <= 64) continue;
    const long long tiles = ((rows + c.bm - 1) / c.bm) * ((cols + c.bn - 1) / c.bn) * batches;
    const long long slots = 256LL * c.resident;
    const int max_ns = allow_split ? 32 : 1;
    for (int ns = 1; ns <= max_ns; ++ns) {
      int kt = (nkt + ns - 1) / ns;
      if (ns > 1 && kt < 8) break;
      int real_ns = (nkt + kt - 1) / kt;
      long long rounds = (tiles * real_ns + slots - 1) / slots;
      double t = rounds * kt * c.t_step * plan_tscale();
      if (real_ns > 1) t += (double)real_ns * rows * cols * 8.0 / 3.0e6 + 3.0;   // + one more launch
      if (t < best * 0.97) { best = t; bm = c.bm; bn = c.bn; nsplit = real_ns; kt_per = kt; }
    }
  }
}

#define DISPATCH_DGRAD(bm, bn, grid, stream, p)                                               \
  do {                                                                                        \
    if (bn == 32) hipLaunchKernelGGL((k_conv_dgrad<128, 32, 1>), grid, dim3(256), 0, stream, p); \
    else DISPATCH_TILE(k_conv_dgrad, bm, bn, grid, stream, p);                                 \
  } while (0)
#define DISPATCH_FWD(bm, bn, grid, stream, p)                                                 \
  do {                                                                                        \
    if (bn == 32) hipLaunchKernelGGL((k_conv_fwd<128, 32, 1>), grid, dim3(256), 0, stream, p); \
    else DISPATCH_TILE(k_conv_fwd, bm, bn, grid, stream, p);                                   \
  } while (0)
#define DISPATCH_TILE(KERNEL, bm, bn, grid, stream, p)                                        \
  do {                                                                                        \
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((KERNEL<128, 128>), grid, dim3(256), 0, stream, p); \
    else if (bm == 128 && bn == 64) hipLaunchKernelGGL((KERNEL<128, 64>), grid, dim3(256), 0, stream, p); \
    else if (bm == 64 && bn == 128) hipLaunchKernelGGL((KERNEL<64, 128>), grid, dim3(256), 0, stream, p); \
    else hipLaunchKernelGGL((KERNEL<64, 64>), grid, dim3(256), 0, stream, p);                  \
  } while (0)

// C[z][m][n] = sum_k A[z][m][k] * B[z][k][n]   (row-major, k and n multiples of 4) — the forward kernel run as a
// 1x1 convolution over `rows` pixels, one batch per blockIdx.z.  Used by the Winograd path (conv_wino.hip).
int vcg_gemm_batched(const float* A, const float* B, float* C, int rows, int K, int Ncols, int batches, hipStream_t st) {
  ConvGeom g = {};
  g.N = 1; g.H = 1; g.W = rows; g.Cin = K; g.Cout = Ncols; g.KH = g.KW = 1; g.stride = 1; g.pad = 0; g.reflect = 0;
  g.ups = 1; g.act = VCG_ACT_NONE; g.cin_log = K; g.cout_log = Ncols;
  g.Hl = 1; g.Wl = rows; g.Ho = 1; g.Wo = rows; g.M = rows; g.K = K; g.taps = 1;
  ConvP p; fill_params(g, p);
  p.a = A; p.b = B; p.bias = nullptr; p.out = C;
  VCG_CHECK_ARG((unsigned long long)rows * K * 4 < (1ull << 31) && (unsigned long long)K * Ncols * 4 < (1ull << 31),
                "vcg_gemm_batched: operand extents must stay below 2 GiB per batch");
  p.a_bytes = (uint32_t)((size_t)rows * K * 4); p.b_bytes = (uint32_t)((size_t)K * Ncols * 4);
  p.nbatch = batches; p.a_bstride = (uint32_t)((size_t)rows * K); p.b_bstride = (uint32_t)((size_t)K * Ncols);
  p.out_bstride = (size_t)rows * Ncols;
  VCG_CHECK_ARG((unsigned long long)rows * K * (unsigned long long)batches < (1ull << 32), "vcg_gemm_batched: batch stride overflow");
  int bm, bn, nsplit, kt_per;
  gemm_plan(rows, Ncols, (K + BK - 1) / BK, false, bm, bn, nsplit, kt_per, batches, false, K <= 2048);
  dim3 grid((rows + bm - 1) / bm, (Ncols + bn - 1) / bn, batches);
  {
    VcgProfScope prof("k_conv_fwd<fp32 MFMA>", 2.0 * rows * (double)K * Ncols * batches, st);
    if (bm == 128 && bn == 128 && K <= 2048) hipLaunchKernelGGL((k_conv_fwd<128, 128, 2, false>), grid, dim3(256), 0, st, p);
    else DISPATCH_FWD(bm, bn, grid, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_gemm_batched");
  return 0;
}

// floats of the packed-weight buffer: Wf[K][Cout], then (3x3 stride-1 layers) the Winograd-transformed U[16][Kc][Cout]
static size_t wf_floats(const ConvGeom& g) { return (((size_t)g.K * g.Cout + 63) / 64) * 64; }
// offset of Wkd (kw-folded thin data gradient) in the packed buffer: after Wf and, for a 4 -> 4 layer, after Wk
// The direct split-operand kernels take their weight operand pre-split (k_pack_planes): the forward the WFT planes, the data
// gradient the WFD planes.  Layers that own Winograd copies (every D, R and U block from 128 reduction channels on) run
// through those in all three directions and get neither; should one of them meet a map Winograd cannot take (odd sizes), it
// runs on the fp32-MFMA kernels from Wf.  Thin (Cout == 4) layers and the kw-folded data gradient never take them either.
// (a layer with Winograd copies whose channel product is under a direction's gate — conv_wino.hip: forward from
// Kc Cout / (Kc + Cout) = 64, data gradient from 80; the 1024 -> 64 latent convs sit at 60 — runs that direction direct, and
// gets the planes for it)
static bool wino_takes_fwd(const ConvGeom& g) {
  const long long kc = (long long)g.ups * g.ups * g.Cin;
  return vcg_wino_weight_ok(g) && kc * g.Cout >= vcg_wino_gate_fwd() * (kc + g.Cout);
}
static bool wino_takes_dgrad(const ConvGeom& g) {
  const long long kc = (long long)g.ups * g.ups * g.Cin;
  return vcg_wino_weight_ok(g) && kc * g.Cout >= vcg_wino_gate_dgrad() * (kc + g.Cout);
}
// floats of the Winograd copies a layer keeps: U if some map can take the forward (and with it the weight gradient), Ud if
// some map can take the data gradient
static size_t wino_u_floats(const ConvGeom& g) { return wino_takes_fwd(g) ? vcg_wino_weight_floats(g) : 0; }
static size_t wino_ud_floats(const ConvGeom& g) { return wino_takes_dgrad(g) ? vcg_wino_weight_floats(g) : 0; }
static bool wft_wanted(const ConvGeom& g) { return g.Cout >= 64 && g.Cin % 4 == 0 && !wino_takes_fwd(g); }
static bool wfd_wanted(const ConvGeom& g) {
  return g.Cout >= 64 && g.Cout % 32 == 0 && !wino_takes_dgrad(g) && !vcg_thin_fold_dgrad_ok(g) && !vcg_thin_dgrad_ok(g);
}
static size_t wft_floats(const ConvGeom& g) { return (size_t)g.Cout * ((g.K + 31) / 32) * VCG_PFLOATS; }      // VCG_PBYTES per (co, K block)
static size_t wfd_floats(const ConvGeom& g) { return (size_t)g.KH * g.KW * g.ups * g.ups * g.Cin * (g.Cout / 32) * VCG_PFLOATS; }
static size_t wkd_offset(const ConvGeom& g) { return wf_floats(g) + (vcg_thin_fold_ok(g) ? vcg_thin_fold_weight_floats(g) : 0); }
static size_t wft_offset(const ConvGeom& g) {
  return wf_floats(g) + wino_u_floats(g) + wino_ud_floats(g) +
         (vcg_thin_fold_ok(g) ? vcg_thin_fold_weight_floats(g) : 0) +
         (vcg_thin_fold_dgrad_ok(g) ? vcg_thin_fold_dgrad_weight_floats(g) : 0);
}
static size_t wfd_offset(const ConvGeom& g) { return wft_offset(g) + (wft_wanted(g) ? wft_floats(g) : 0); }
// the pack ends with a 16-float header: word 0 = bit pattern of the kernel's largest magnitude (what every plane set of
// the pack was scaled by: vcg_common.h), read by the kernels that multiply by those planes
static size_t wamax_offset(const ConvGeom& g) { return wfd_offset(g) + (wfd_wanted(g) ? wfd_floats(g) : 0); }
const void* vcg_pack_amax(const ConvGeom& g, const float* wf) { return wf + wamax_offset(g); }
// VCG_SLAB=0 keeps the slab kernels (conv_slab.hip) out of the dispatch: A/B measurements only
static bool slab_enabled() {
  static const int on = [] { const char* e = getenv("VCG_SLAB"); return e ? atoi(e) : 1; }();
  return on != 0;
}
// VCG_RING=0: the row-ring weight gradients (conv_ring.hip) out of the dispatch
static bool wgrad_ring_ok(const ConvGeom& g) {
  static const int on = [] { const char* e = getenv("VCG_RING"); return e ? atoi(e) : 1; }();
  return on != 0 && !vcg_wino_wgrad_ok(g) && vcg_ring_wgrad_ok(g);
}
static bool fwd_slab_ok(const ConvGeom& g) {
  return slab_enabled() && !vcg_thin_fold_ok(g) && !vcg_thin_fwd_ok(g) && !vcg_wino_fwd_ok(g) && wft_wanted(g) && vcg_slab_fwd_ok(g);
}
static bool dgrad_slab_ok(const ConvGeom& g) {
  return slab_enabled() && !vcg_thin_fold_dgrad_ok(g) && !vcg_thin_dgrad_ok(g) && !vcg_wino_dgrad_ok(g) && wfd_wanted(g) && vcg_slab_dgrad_ok(g);
}

// the forward kernel on a caller-built geometry (no bias, no activation, no K slicing): conv_thin.hip's kw-folded path
int vcg_fwd_launch(const ConvGeom& g, const float* x, const float* wf, float* y, hipStream_t st) {
  ConvP p; fill_params(g, p);
  p.a = x; p.b = wf; p.bias = nullptr; p.out = y; p.act = VCG_ACT_NONE;
  const unsigned long long ab = (unsigned long long)g.N * g.H * g.W * g.Cin * 4, bb = (unsigned long long)g.K * g.Cout * 4;
  VCG_CHECK_ARG(ab < (1ull << 31) && bb < (1ull << 31), "vcg_conv_fwd: tensor extents must stay below 2 GiB");
  p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  int bm, bn, nsplit, kt_per;
  gemm_plan(g.M, g.Cout, (g.K + BK - 1) / BK, false, bm, bn, nsplit, kt_per, 1, true);
  dim3 grid((g.M + bm - 1) / bm, (g.Cout + bn - 1) / bn, 1);
  {
    VcgProfScope prof("k_conv_fwd<fp32 MFMA>", 2.0 * g.M * (double)g.K * g.Cout, st);
    DISPATCH_FWD(bm, bn, grid, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv_fwd(raw)");
  return 0;
}

extern "C" size_t vcg_pack_weight_floats(const int32_t* cd) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_pack_weight_floats")) return 0;
  return wf_floats(g) + wino_u_floats(g) + wino_ud_floats(g)                           // + U (forward) + Ud (data gradient)
         + (vcg_thin_fold_ok(g) ? vcg_thin_fold_weight_floats(g) : 0)                   // + Wk (kw-folded thin forward)
         + (vcg_thin_fold_dgrad_ok(g) ? vcg_thin_fold_dgrad_weight_floats(g) : 0)       // + Wkd (kw-folded thin data gradient)
         + (wft_wanted(g) ? wft_floats(g) : 0)                                          // + WFT planes (split-operand direct forward)
         + (wfd_wanted(g) ? wfd_floats(g) : 0)                                          // + WFD planes (split-operand direct data gradient)
         + 16;                                                                          // + the header (wamax_offset)
}

extern "C" int vcg_pack_weight(const float* w_oihw, float* wf, const int32_t* cd, void* stream) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_pack_weight")) return -1;
  VCG_CHECK_ARG(w_oihw && wf, "vcg_pack_weight: null pointer");
  // the kernel's largest magnitude: every pre-split plane set below holds w / s (vcg_common.h), and the header keeps the
  // bits for the kernels that consume the planes
  const VcgAmaxOut aw = vcg_amax_new((hipStream_t)stream);
  if (vcg_absmax_launch(w_oihw, (size_t)g.cout_log * g.cin_log * g.ups * g.ups * g.KH * g.KW, aw, (hipStream_t)stream)) return -2;
  const VcgAmax amax_w = vcg_amax_in(aw);                  // the header word is written by the Wf pack kernel at the end of this call
  if (vcg_thin_fold_ok(g) && vcg_thin_fold_pack(g, w_oihw, wf + wf_floats(g), amax_w, (hipStream_t)stream)) return -2;
  if (vcg_thin_fold_dgrad_ok(g) && vcg_thin_fold_dgrad_pack(g, w_oihw, wf + wkd_offset(g), amax_w, (hipStream_t)stream)) return -2;
  if (wft_wanted(g)) {
    ConvP q; fill_params(g, q);
    q.amax_b = amax_w;
    const size_t tot = (size_t)g.Cout * ((g.K + 31) / 32) * 8;
    int blocks = (int)((tot + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_pack_planes<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)(wf + wft_offset(g)), q,
                       g.cin_log, g.cout_log);
    VCG_LAUNCH_CHECK("vcg_pack_weight(WFT planes)");
  }
  if (wfd_wanted(g)) {
    ConvP q; fill_params(g, q);
    q.amax_b = amax_w;
    const size_t tot = (size_t)g.KH * g.KW * g.ups * g.ups * g.Cin * (g.Cout / 4);
    int blocks = (int)((tot + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_pack_planes<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)(wf + wfd_offset(g)), q,
                       g.cin_log, g.cout_log);
    VCG_LAUNCH_CHECK("vcg_pack_weight(WFD planes)");
  }
  if (wino_takes_fwd(g) && vcg_wino_weight(g, w_oihw, wf + wf_floats(g), amax_w, (hipStream_t)stream)) return -2;
  if (wino_takes_dgrad(g) && vcg_wino_weight_dgrad(g, w_oihw, wf + wf_floats(g) + wino_u_floats(g), amax_w, (hipStream_t)stream)) return -2;
  ConvP p; fill_params(g, p);
  p.amax_b = amax_w;
  uint32_t* const hdr = (uint32_t*)(wf + wamax_offset(g));
  if (cd[VCG_CD_PACK_FLAGS] & 1) {                          // the caller knows nothing reads Wf (vcg_conv_reads_wf): the header only
    ConvP q = p; q.K = 0; q.Cout = 0;
    hipLaunchKernelGGL(k_pack_weight, dim3(1), dim3(64), 0, (hipStream_t)stream, w_oihw, wf, q, g.cin_log, g.cout_log, hdr);
    VCG_LAUNCH_CHECK("vcg_pack_weight(header)");
    return 0;
  }
  size_t total = (size_t)g.K * g.Cout;
  const int T = g.KH * g.KW * g.ups * g.ups;
  const size_t lds = (size_t)T * 8 * 33 * sizeof(float);
  if (total < (1u << 20) || lds > 64 * 1024) {
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pack_weight, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, wf, p, g.cin_log, g.cout_log, hdr);
  } else {
    hipLaunchKernelGGL(k_pack_weight_t, dim3((g.Cout + 31) / 32, (g.Cin + 7) / 8), dim3(256), lds, (hipStream_t)stream,
                       w_oihw, wf, p, g.cin_log, g.cout_log, hdr);
  }
  VCG_LAUNCH_CHECK("vcg_pack_weight");
  return 0;
}

static void fwd_plan(const ConvGeom& g, int& bm, int& bn, int& nsplit, int& kt_per) {
  gemm_plan(g.M, g.Cout, (g.K + BK - 1) / BK, true, bm, bn, nsplit, kt_per, 1, true);
}

extern "C" size_t vcg_conv_fwd_workspace(const int32_t* cd) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_fwd_workspace")) return 0;
  if (vcg_thin_fold_ok(g)) return vcg_thin_fold_workspace(g);
  if (vcg_thin_fwd_ok(g) || vcg_thinin_fwd_ok(g)) return 0;
  if (vcg_wino_fwd_ok(g)) return vcg_wino_fwd_workspace(g);
  if (fwd_slab_ok(g)) return 0;
  int bm, bn, nsplit, kt_per;
  fwd_plan(g, bm, bn, nsplit, kt_per);
  return nsplit > 1 ? (size_t)nsplit * g.M * g.Cout * sizeof(float) + 256 : 0;
}

static int ew_grid(size_t work) {
  size_t b = (work + 255) / 256;
  if (b > 2048) b = 2048;
  return b < 1 ? 1 : (int)b;
}

// does the direct split-operand forward leave the InstanceNorm partials itself?  (every 128-row tile inside one image)
static bool fwd_tile_stats_ok(const ConvGeom& g) {
  if (vcg_thin_fold_ok(g) || vcg_thin_fwd_ok(g) || vcg_thinin_fwd_ok(g) || vcg_wino_fwd_ok(g) || fwd_slab_ok(g)) return false;
  int bm, bn, nsplit, kt_per;
  fwd_plan(g, bm, bn, nsplit, kt_per);
  return bm == 128 && bn >= 64 && nsplit == 1 && wft_wanted(g) && (g.Ho * g.Wo) % 128 == 0;
}

// in_part != nullptr: also leave the InstanceNorm chunk partials of y there (the caller checked that this launch plan
// can: Winograd, or fwd_tile_stats_ok) and report the chunk count per image.
static int conv_fwd_impl(const float* x, const float* wf, const float* bias, float* y, const int32_t* cd, void* ws,
                         size_t ws_bytes, void* stream, double* in_part, const VcgInTail* tail_req, float* saved = nullptr,
                         const VcgPre* pre = nullptr) {
  const uint64_t x_handle = vcg_take_hint_x();              // vcg_amax_hint: who wrote x left its largest magnitude (or 0)
  (void)vcg_take_hint_dy();
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_fwd")) return -1;
  VCG_CHECK_ARG(x && wf && y, "vcg_conv_fwd: null pointer");
  if (pre && pre->mean)
    VCG_CHECK_ARG(!vcg_thin_fold_ok(g) && !vcg_thin_fwd_ok(g) && vcg_wino_pre_ok(g),
                  "vcg_conv_fwd_in_pre: this geometry has no normalising gather (ask vcg_conv_pre_ok first)");
  if (vcg_thin_fold_ok(g)) return vcg_thin_fold_fwd(g, x, wf + wf_floats(g), vcg_pack_amax(g, wf), bias, y, ws, ws_bytes, (hipStream_t)stream, x_handle);
  if (vcg_thin_fwd_ok(g)) return vcg_thin_fwd(g, x, wf, bias, y, (hipStream_t)stream);
  if (vcg_thinin_fwd_ok(g)) {
    if (vcg_thinin_fwd(g, x, wf, vcg_pack_amax(g, wf), bias, y, in_part, (hipStream_t)stream, x_handle)) return -2;
    if (in_part)
      return vcg_in_finalize(in_part, tail_req->out1, tail_req->out2, g.N, tail_req->HW, g.Cout, vcg_thinin_nchunk(g), tail_req->eps,
                             (hipStream_t)stream);
    return 0;
  }
  if (vcg_wino_fwd_ok(g))
    return vcg_wino_fwd(g, x, wf + wf_floats(g), vcg_pack_amax(g, wf), bias, y, ws, ws_bytes, (hipStream_t)stream, in_part, tail_req, saved, x_handle,
                        pre);
  if (fwd_slab_ok(g))
    return vcg_slab_fwd(g, x, wf + wft_offset(g), wft_floats(g) * 4, vcg_pack_amax(g, wf), bias, y, in_part, tail_req, (hipStream_t)stream, x_handle);
  ConvP p; fill_params(g, p);
  if (in_part) {
    p.in_part = in_part;
    p.in_nchunk = g.Ho * g.Wo / 128;
  }
  p.a = x; p.b = wf; p.bias = bias; p.out = y;
  {
    const unsigned long long ab = (unsigned long long)g.N * g.H * g.W * g.Cin * 4, bb = (unsigned long long)g.K * g.Cout * 4;
    VCG_CHECK_ARG(ab < (1ull << 31) && bb < (1ull << 31), "vcg_conv_fwd: tensor extents must stay below 2 GiB");
    p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  }
  int bm, bn, nsplit, kt_per;
  fwd_plan(g, bm, bn, nsplit, kt_per);
  if (nsplit > 1) {
    VCG_CHECK_ARG(ws && ws_bytes >= vcg_conv_fwd_workspace(cd), "vcg_conv_fwd: workspace too small (%zu)", ws_bytes);
    p.ksplit = nsplit; p.kt_per = kt_per; p.slab = (float*)ws;
  }
  dim3 grid((g.M + bm - 1) / bm, (g.Cout + bn - 1) / bn, nsplit);
  hipStream_t st = (hipStream_t)stream;
  const double gemm_flops = 2.0 * g.M * (double)g.K * g.Cout;
  if (bm == 128 && bn >= 64 && wft_wanted(g)) {          // split-operand fp16 kernel, B^T from the pre-split WFT planes of the pack
    p.b = wf + wft_offset(g);
    p.b_bytes = (uint32_t)(wft_floats(g) * 4);
    // the input's largest magnitude (its scale, vcg_common.h): from its writer's handle, else measured
    if (vcg_operand_amax(x, (size_t)g.N * g.H * g.W * g.Cin, x_handle, 0, st, &p.amax_a)) return -2;
    p.amax_b = vcg_amax_stored(vcg_pack_amax(g, wf));
    if (in_part) p.in_tail = vcg_in_tail_make(tail_req->out1, tail_req->out2, g.N * (int)grid.y, tail_req->HW, tail_req->eps);
    VcgProfScope prof(bn == 128 ? "k_conv_fwd_split<128>" : "k_conv_fwd_split<64>", gemm_flops, st);
    if (bn == 128) hipLaunchKernelGGL((k_conv_fwd_split<128>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_conv_fwd_split<64>), grid, dim3(256), 0, st, p);
  } else {
    VcgProfScope prof("k_conv_fwd<fp32 MFMA>", gemm_flops, st);
    if (bm == 128 && bn == 128 && g.K <= 2048) hipLaunchKernelGGL((k_conv_fwd<128, 128, 2, false>), grid, dim3(256), 0, st, p);
    else DISPATCH_FWD(bm, bn, grid, st, p);
  }
  if (nsplit > 1)
    hipLaunchKernelGGL(k_splitk_finish, dim3(ew_grid((size_t)g.M * g.Cout / 4)), dim3(256), 0, st, (const float*)ws, bias,
                       y, (size_t)g.M, g.Cout, nsplit, g.cout_log, g.act);
  VCG_LAUNCH_CHECK("vcg_conv_fwd");
  if (in_part && !p.in_tail.out1)
    return vcg_in_finalize(in_part, tail_req->out1, tail_req->out2, g.N, tail_req->HW, g.Cout, p.in_nchunk, tail_req->eps, st);
  return 0;
}

extern "C" int vcg_conv_fwd(const float* x, const float* wf, const float* bias, float* y,
                            const int32_t* cd, void* ws, size_t ws_bytes, void* stream) {
  return conv_fwd_impl(x, wf, bias, y, cd, ws, ws_bytes, stream, nullptr, nullptr);
}

// ---- convolution + the statistics of the InstanceNorm that follows it (CaSb with norm=True, Networks.py:93-95) ----------
// Workspace: [conv workspace, rounded to 256 B][statistics partials].  Where the conv's launch plan can, the partial sums
// come out of the conv's own epilogue (Winograd output transform; the direct split-operand tiles) and only the tiny
// finalize kernel follows; otherwise y is reduced by the same pass vcg_in_stats runs.
static size_t fwd_in_conv_ws(const int32_t* cd) { return (vcg_conv_fwd_workspace(cd) + 255) / 256 * 256; }

extern "C" size_t vcg_conv_fwd_in_workspace(const int32_t* cd) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_fwd_in_workspace")) return 0;
  size_t part;
  const bool thinin = !vcg_thin_fold_ok(g) && !vcg_thin_fwd_ok(g) && vcg_thinin_fwd_ok(g);
  if (thinin) part = (size_t)g.N * vcg_thinin_nchunk(g) * g.Cout * 2 * sizeof(double);
  else if (vcg_wino_fwd_ok(g)) part = vcg_wino_fwd_stats_doubles(g) * sizeof(double);
  else if (fwd_slab_ok(g) && vcg_slab_fwd_stats_ok(g)) part = (size_t)g.N * vcg_slab_fwd_nchunk(g) * g.Cout * 2 * sizeof(double);
  else if (fwd_tile_stats_ok(g)) part = (size_t)g.N * (g.Ho * g.Wo / 128) * g.Cout * 2 * sizeof(double);
  else part = vcg_in_workspace(g.N, g.Ho * g.Wo, g.Cout);
  return fwd_in_conv_ws(cd) + part;
}

// floats of forward state worth keeping for the weight gradient of this layer (0: nothing) — today the Winograd-transformed
// input V of the layers whose weight gradient runs through Winograd as well
extern "C" size_t vcg_conv_saved_floats(const int32_t* cd) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_saved_floats")) return 0;
  return vcg_wino_fwd_ok(g) ? vcg_wino_saved_floats(g) : 0;
}

static int conv_fwd_in_impl(const float* x, const float* wf, const float* bias, float* y, float* mean, float* rstd,
                            float eps, float* saved, const int32_t* cd, void* ws, size_t ws_bytes, void* stream, const VcgPre* pre);
extern "C" int vcg_conv_fwd_in(const float* x, const float* wf, const float* bias, float* y, float* mean, float* rstd,
                               float eps, float* saved, const int32_t* cd, void* ws, size_t ws_bytes, void* stream) {
  return conv_fwd_in_impl(x, wf, bias, y, mean, rstd, eps, saved, cd, ws, ws_bytes, stream, nullptr);
}
// 1 if vcg_conv_fwd_in_pre exists for this geometry: the forward's input gather can normalise on the fly (today: the Winograd
// input transform — the D2..D4, R and U1 layers at the training sizes)
extern "C" int vcg_conv_pre_ok(const int32_t* cd) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_pre_ok")) return 0;
  return (!vcg_thin_fold_ok(g) && !vcg_thin_fwd_ok(g) && vcg_wino_pre_ok(g)) ? 1 : 0;
}
extern "C" int vcg_conv_fwd_in_pre(const float* t_prev, const float* pre_mean, const float* pre_rstd, int pre_act, const float* wf,
                                   const float* bias, float* y, float* mean, float* rstd, float eps, float* saved, const int32_t* cd,
                                   void* ws, size_t ws_bytes, void* stream) {
  VCG_CHECK_ARG(pre_mean && pre_rstd, "vcg_conv_fwd_in_pre: null statistics");
  VCG_CHECK_ARG(pre_act >= VCG_ACT_NONE && pre_act <= VCG_ACT_SIGMOID, "vcg_conv_fwd_in_pre: bad activation %d", pre_act);
  const VcgPre pre = {pre_mean, pre_rstd, pre_act};
  (void)vcg_take_hint_x();              // the input's magnitude is bounded, not measured (conv_wino.hip, vcg_wino_fwd)
  return conv_fwd_in_impl(t_prev, wf, bias, y, mean, rstd, eps, saved, cd, ws, ws_bytes, stream, &pre);
}
static int conv_fwd_in_impl(const float* x, const float* wf, const float* bias, float* y, float* mean, float* rstd,
                            float eps, float* saved, const int32_t* cd, void* ws, size_t ws_bytes, void* stream, const VcgPre* pre) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_fwd_in")) return -1;
  VCG_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "vcg_conv_fwd_in: mean and rstd go together");
  if (saved && !vcg_conv_saved_floats(cd)) saved = nullptr;
  if (!mean) {                                           // no statistics wanted: the plain forward (+ saved state)
    VCG_CHECK_ARG(ws_bytes >= vcg_conv_fwd_workspace(cd), "vcg_conv_fwd_in: workspace too small (%zu)", ws_bytes);
    return conv_fwd_impl(x, wf, bias, y, cd, ws, ws_bytes, stream, nullptr, nullptr, saved, pre);
  }
  VCG_CHECK_ARG(ws, "vcg_conv_fwd_in: null pointer");
  VCG_CHECK_ARG(ws_bytes >= vcg_conv_fwd_in_workspace(cd), "vcg_conv_fwd_in: workspace too small (%zu)", ws_bytes);
  const size_t cws = fwd_in_conv_ws(cd);
  double* part = reinterpret_cast<double*>(static_cast<char*>(ws) + cws);
  const bool thinin = !vcg_thin_fold_ok(g) && !vcg_thin_fwd_ok(g) && vcg_thinin_fwd_ok(g);
  const bool fused = thinin || vcg_wino_fwd_ok(g) || (fwd_slab_ok(g) && vcg_slab_fwd_stats_ok(g)) || fwd_tile_stats_ok(g);
  // fused: the conv's epilogue leaves the partials and its last block per (image, channel range) finalizes them
  VcgInTail req = vcg_in_tail_none();
  req.out1 = mean; req.out2 = rstd; req.HW = g.Ho * g.Wo; req.eps = eps;
  if (conv_fwd_impl(x, wf, bias, y, cd, ws, cws, stream, fused ? part : nullptr, &req, saved, pre)) return -1;
  if (fused) return 0;
  return vcg_in_stats_pass(y, mean, rstd, g.N, g.Ho * g.Wo, g.Cout, eps, part, ws_bytes - cws, (hipStream_t)stream);
}

static int dgrad_setup(const ConvGeom& g, ConvP& p, int& bm, int& bn, int& nsplit, int& kt_per) {
  p.Hc = g.Hl / g.stride; p.Wc = g.Wl / g.stride; p.Mc = g.N * p.Hc * p.Wc;
  p.NB = g.ups * g.ups * g.Cin;
  p.fd_hcwc = make_fastdiv((uint32_t)(p.Hc * p.Wc));
  p.fd_wc = make_fastdiv((uint32_t)p.Wc);
  // stride s: s * s parity classes, each a GEMM over its own (KH / s) x (KW / s) taps — the planner sees them as batches,
  // and blockIdx.z = class + s * s * K slice
  const int s = g.stride;
  gemm_plan(p.Mc, p.NB, (g.KH * g.KW * g.Cout + BK - 1) / BK, s == 1, bm, bn, nsplit, kt_per, 1, true);
  if (s > 1 && g.KH % s == 0 && g.KW % s == 0) {
    // the unsliced plan fills the chip for the large maps (measured: slicing made 64 -> 128 at 128^2 and 128 -> 256 at 64^2
    // slower); the deep layer (256 -> 512 at 32^2: 256 workgroups) is the one that needs its K cut
    const long long wgs = ((p.Mc + bm - 1) / bm) * ((p.NB + bn - 1) / bn) * s * s;
    if (wgs < 384)
      gemm_plan(p.Mc, p.NB, ((g.KH / s) * (g.KW / s) * g.Cout + BK - 1) / BK, true, bm, bn, nsplit, kt_per, s * s, true);
  }
  return 0;
}

// does the forward / the data gradient at this geometry read the fp32 Wf block?  (mirrors conv_fwd_impl and vcg_conv_dgrad)
static bool fwd_reads_wf(const ConvGeom& g) {
  if (vcg_thin_fold_ok(g)) return false;
  if (vcg_thin_fwd_ok(g) || vcg_thinin_fwd_ok(g)) return true;
  if (vcg_wino_fwd_ok(g) || fwd_slab_ok(g)) return false;
  int bm, bn, nsplit, kt_per;
  fwd_plan(g, bm, bn, nsplit, kt_per);
  return !(bm == 128 && bn >= 64 && wft_wanted(g));
}
static bool dgrad_reads_wf(const ConvGeom& g) {
  if (g.Hl % g.stride || g.Wl % g.stride) return true;
  if (vcg_thin_fold_dgrad_ok(g)) return false;
  if (vcg_thin_dgrad_ok(g)) return true;
  if (vcg_wino_dgrad_ok(g) || dgrad_slab_ok(g)) return false;
  ConvP p; fill_params(g, p);
  int bm, bn, nsplit, kt_per;
  dgrad_setup(g, p, bm, bn, nsplit, kt_per);
  return !(bm == 128 && bn >= 64 && wfd_wanted(g));          // planes; everything else (no planes, 32-column, fp32) reads Wf
}
extern "C" int vcg_conv_reads_wf(const int32_t* cd) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_reads_wf")) return 1;
  return (fwd_reads_wf(g) || dgrad_reads_wf(g)) ? 1 : 0;
}

extern "C" size_t vcg_conv_dgrad_workspace(const int32_t* cd) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_dgrad_workspace")) return 0;
  if (g.Hl % g.stride || g.Wl % g.stride) return 0;
  if (vcg_thin_fold_dgrad_ok(g)) return vcg_thin_fold_dgrad_workspace(g);
  if (vcg_thin_dgrad_ok(g)) return vcg_thin_dgrad_workspace(g);
  if (vcg_wino_dgrad_ok(g)) return vcg_wino_dgrad_workspace(g);
  if (dgrad_slab_ok(g)) return vcg_slab_dgrad_workspace(g);
  ConvP p; fill_params(g, p);
  int bm, bn, nsplit, kt_per;
  dgrad_setup(g, p, bm, bn, nsplit, kt_per);
  return nsplit > 1 ? (size_t)nsplit * g.N * g.H * g.W * g.Cin * sizeof(float) + 256 : 0;
}

extern "C" int vcg_conv_dgrad(const float* dy, const float* wf, float* dx, const int32_t* cd, void* ws,
                              size_t ws_bytes, void* stream) {
  (void)vcg_take_hint_x();
  const uint64_t dy_handle = vcg_take_hint_dy();            // vcg_amax_hint
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_dgrad")) return -1;
  VCG_CHECK_ARG(dy && wf && dx, "vcg_conv_dgrad: null pointer");
  VCG_CHECK_ARG(g.Hl % g.stride == 0 && g.Wl % g.stride == 0, "vcg_conv_dgrad: input %dx%d not divisible by stride", g.Hl, g.Wl);
  VCG_CHECK_ARG(g.stride == 1 || g.ups == 1, "vcg_conv_dgrad: stride 2 with ups 2 unsupported");
  if (vcg_thin_fold_dgrad_ok(g)) return vcg_thin_fold_dgrad(g, dy, wf + wkd_offset(g), vcg_pack_amax(g, wf), dx, ws, ws_bytes, (hipStream_t)stream, dy_handle);
  if (vcg_thin_dgrad_ok(g)) return vcg_thin_dgrad(g, dy, wf, dx, ws, ws_bytes, (hipStream_t)stream);
  if (vcg_wino_dgrad_ok(g))
    return vcg_wino_dgrad(g, dy, wf + wf_floats(g) + wino_u_floats(g), vcg_pack_amax(g, wf), dx, ws, ws_bytes, (hipStream_t)stream, dy_handle);
  if (dgrad_slab_ok(g))
    return vcg_slab_dgrad(g, dy, wf + wfd_offset(g), wfd_floats(g) * 4, vcg_pack_amax(g, wf), dx, ws, ws_bytes, (hipStream_t)stream, dy_handle);
  ConvP p; fill_params(g, p);
  p.a = dy; p.b = wf; p.out = dx;
  {
    const unsigned long long ab = (unsigned long long)g.M * g.Cout * 4, bb = (unsigned long long)g.K * g.Cout * 4;
    VCG_CHECK_ARG(ab < (1ull << 31) && bb < (1ull << 31), "vcg_conv_dgrad: tensor extents must stay below 2 GiB");
    p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  }
  {
    auto dbl = [&](int L) { int lo = 1 > L - 1 - g.pad ? 1 : L - 1 - g.pad, hi = g.pad < L - 2 ? g.pad : L - 2; return lo <= hi; };
    p.dbl_mirror = g.reflect && (dbl(g.Hl) || dbl(g.Wl));
  }
  int bm, bn, nsplit, kt_per;
  dgrad_setup(g, p, bm, bn, nsplit, kt_per);
  if (nsplit > 1) {
    VCG_CHECK_ARG(ws && ws_bytes >= vcg_conv_dgrad_workspace(cd), "vcg_conv_dgrad: workspace too small (%zu)", ws_bytes);
    p.ksplit = nsplit; p.kt_per = kt_per; p.slab = (float*)ws;
  }
  dim3 grid((p.Mc + bm - 1) / bm, (p.NB + bn - 1) / bn, nsplit * g.stride * g.stride);
  hipStream_t st = (hipStream_t)stream;
  {
    const double gemm_flops = 2.0 * g.M * (double)g.K * g.Cout;   // stride 2: the parity classes together visit every tap once
    const bool planes = bm == 128 && bn >= 64 && wfd_wanted(g);       // weight rows from the pre-split WFD planes of the pack
    if (planes) { p.b = wf + wfd_offset(g); p.b_bytes = (uint32_t)(wfd_floats(g) * 4); }
    // layers without planes (the 4-channel dy of the 7x7 head: K = (tap, co) is not a multiple of 32 per tap) still run on the
    // 16-bit pipe when their tile is 128 x 64: the weight rows are split in the kernel from Wf
    const bool nopl64 = !planes && bm == 128 && bn == 64 && g.Cout % 4 == 0;
    const bool split = planes || bn == 32 || nopl64;
    if (split) {                                            // fp16 x 2 kernels: the operands' largest magnitudes (vcg_common.h)
      // the kernel ADDS the sources that reflect padding folds onto a pixel before it splits the sum: up to 4 of them (9 on maps
      // so small that a pixel has mirrors on both sides) — the operand is bounded by 2^4 x dy's largest magnitude, not by it
      if (vcg_operand_amax(dy, (size_t)g.M * g.Cout, dy_handle, g.reflect ? 4 : 0, st, &p.amax_a)) return -2;
      p.amax_b = vcg_amax_stored(vcg_pack_amax(g, wf));
    }
    VcgProfScope prof(!split ? "k_conv_dgrad<fp32 MFMA>" : bn == 128 ? "k_conv_dgrad_split<128, 2>" : bn == 64 ? "k_conv_dgrad_split<64, 2>"
                                                                                                             : "k_conv_dgrad_split<32, 1>",
                      gemm_flops, st);
    if (planes) {                                           // split-operand kernel
      if (bn == 128) hipLaunchKernelGGL((k_conv_dgrad_split<128, 2>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((k_conv_dgrad_split<64, 2>), grid, dim3(256), 0, st, p);
    } else if (bn == 32) hipLaunchKernelGGL((k_conv_dgrad_split<32, 1>), grid, dim3(256), 0, st, p);
    else if (nopl64) hipLaunchKernelGGL((k_conv_dgrad_split<64, 2, false>), grid, dim3(256), 0, st, p);
    else DISPATCH_DGRAD(bm, bn, grid, st, p);
  }
  if (nsplit > 1)
    hipLaunchKernelGGL(k_splitk_finish, dim3(ew_grid((size_t)g.N * g.H * g.W * g.Cin / 4)), dim3(256), 0, st,
                       (const float*)ws, (const float*)nullptr, dx, (size_t)g.N * g.H * g.W, g.Cin, nsplit, g.Cin,
                       (int)VCG_ACT_NONE);
  VCG_LAUNCH_CHECK("vcg_conv_dgrad");
  return 0;
}

// Weight-gradient launch plan ("stream-K").  The (tile, K' step) units are dealt out in equal consecutive runs of
// `len` to `grid` workgroups, so every workgroup does the same number of MFMA steps; with one workgroup per
// (tile, split) the count is quantised in rounds of (256 CUs x resident workgroups) — 2304 workgroups on 512
// slots cost five rounds for 4.5 rounds of work (the R blocks, -16 %).  Cost model in us: one 32-pixel K' step of
// a resident workgroup per tile shape (measured on the D/R/U layers), plus the partial-tile write + read at
// ~3 TB/s; every workgroup boundary that falls inside a tile adds one partial.
struct WgradPlan { int bm, bn, grid, len, parts, total, ntr, ntn; };
static WgradPlan wgrad_plan(const ConvGeom& g, int batches = 1) {
  WgradPlan best_p = {};
  const int total = (g.M + BK - 1) / BK;
  struct Cand { int bm, bn, resident; double t_step; };
  // t_step from measured rates: 128x128 ~105 TF, 128x64 ~85 TF, 64x64 ~70 TF at full residency;
  // bm = 256 is the eight-wave lockstep workgroup (one per CU)
  const Cand cands[4] = {{256, 128, 1, 4.3}, {128, 128, 2, 5.1}, {128, 64, 3, 4.7}, {64, 64, 5, 4.8}};
  double best = 1e30;
  int force = 0;
#ifdef VCG_STAMP
  { const char* e = getenv("VCG_WGRAD_BM"); force = e ? atoi(e) : 0; }
#endif
  for (int ci = 0; ci < 4; ++ci) {
    const Cand& c = cands[ci];
    if (force ? c.bm != force : c.bm == 256) continue;    // the lockstep tile measured 0..10 % slower: diagnostics only
    if (c.bn == 128 && g.Cout <= 64) continue;
    if (c.bm >= 128 && g.K <= 64) continue;
    if (c.bm == 256 && g.K < 512) continue;
    if (batches > 1 && g.K % c.bm) continue;               // batched: the row tiles of batch z follow those of z-1
    const int ntr = ((g.K + c.bm - 1) / c.bm) * batches, ntn = (g.Cout + c.bn - 1) / c.bn;
    const long long tiles = (long long)ntr * ntn, units = tiles * total;
    if (units >= (1LL << 30)) continue;
    const long long slots = 256LL * c.resident;
    // one workgroup per resident slot: several shorter workgroups per slot measured 0..30 % slower (more partial
    // tiles, more prologues), so the run length is simply units / slots
    long long len = (units + slots - 1) / slots;
    if (len < 8) len = units < 8 ? units : 8;
    const long long grid = (units + len - 1) / len;
    const int parts = (int)((total + len - 1) / len) + 1;
    double t = (double)len * c.t_step + (double)(grid + tiles) * c.bm * c.bn * 8.0 / 3.0e6;
    if (t < best * 0.97) {
      best = t;
      best_p = {c.bm, c.bn, (int)grid, (int)len, parts, total, ntr, ntn};
    }
  }
  return best_p;
}

static const int kMaxDirectSlabs = 24;

// bias-gradient column sums: TC channel quads per block, ~1024 blocks in flight
static void colsum_plan(const ConvGeom& g, int& tc, int& cgroups, int& rows, int& nchunk) {
  int c4 = g.Cout / 4;
  tc = 1;
  const int cap = vcg_in_tail_enabled() ? 16 : 256;    // tails on: <= 64 channels per workgroup, so the finalizing workgroup has >= 16 chunk lanes per quad
  while (tc * 2 <= c4 && tc * 2 <= cap) tc *= 2;
  cgroups = (c4 + tc - 1) / tc;
  int tp = 256 / tc;
  long long want = 512 / cgroups;
  if (want < 1) want = 1;
  long long r = (g.M + want - 1) / want;
  if (r < 4LL * tp) r = 4LL * tp;
  rows = (int)r;
  nchunk = (g.M + rows - 1) / rows;
}

static bool wgrad_swapped_ok(const ConvGeom& g);
static ConvGeom swapped_geom(const ConvGeom& g);

// Winograd weight gradient core (conv_wino.hip provides V = B^T x B and dM = A dy A^T): the 16 reductions
// dU[xi] = V[xi]^T dM[xi] over the T tiles run as ONE stream-K launch of the wgrad kernel (a 1x1 "convolution" whose
// row tiles enumerate (xi, k-tile)), then k_wino_wgrad_reduce transforms back and accumulates into the OIHW gradient.
static ConvGeom wino_gemm_geom(const ConvGeom& g, int T) {
  ConvGeom q = {};
  q.N = 1; q.H = 1; q.W = T; q.Cin = g.ups * g.ups * g.Cin; q.Cout = g.Cout; q.KH = q.KW = 1; q.stride = 1; q.pad = 0;
  q.reflect = 0; q.ups = 1; q.act = VCG_ACT_NONE; q.cin_log = q.Cin; q.cout_log = g.Cout;
  q.Hl = 1; q.Wl = T; q.Ho = 1; q.Wo = T; q.M = T; q.K = q.Cin; q.taps = 1;
  return q;
}
size_t vcg_wino_wgrad_core_workspace(const ConvGeom& g, int T) {
  const ConvGeom q = wino_gemm_geom(g, T);
  const WgradPlan wp = wgrad_plan(q, 16);
  return (size_t)wp.parts * 16 * q.K * q.Cout * sizeof(float) + 256;
}
int vcg_wino_wgrad_core(const ConvGeom& g, const float* V, const float* dM, int T, float* gw_oihw, void* ws, size_t ws_bytes,
                        hipStream_t st, const VcgAmax& amax_v, const VcgAmax& amax_dm, bool v_planes) {
  const ConvGeom q = wino_gemm_geom(g, T);
  const WgradPlan wp = wgrad_plan(q, 16);
  VCG_CHECK_ARG(wp.grid > 0, "vcg_conv_wgrad: no launch plan for the Winograd path");
  VCG_CHECK_ARG(ws_bytes >= vcg_wino_wgrad_core_workspace(g, T), "vcg_conv_wgrad: Winograd slab workspace too small");
  ConvP p; fill_params(q, p);
  p.a = V; p.b = dM; p.out = (float*)ws;
  p.amax_a = amax_v; p.amax_b = amax_dm;
  p.a_planes = v_planes ? 1 : 0;
  VCG_CHECK_ARG(!v_planes || (wp.bm == 128 && q.K % 32 == 0), "vcg_conv_wgrad: pre-split V needs the split-operand tile");
  p.a_bytes = (uint32_t)((size_t)T * q.K * 4); p.b_bytes = (uint32_t)((size_t)T * q.Cout * 4);
  p.nbatch = 16; p.a_bstride = (uint32_t)((size_t)T * q.K); p.b_bstride = (uint32_t)((size_t)T * q.Cout);
  p.ktiles_total = wp.total; p.sk_len = wp.len; p.sk_units = wp.ntr * wp.ntn * wp.total; p.sk_ntn = wp.ntn;
  p.sk_bm_shift = wp.bm == 256 ? 8 : wp.bm == 128 ? 7 : 6; p.sk_bn_shift = wp.bn == 128 ? 7 : 6;
  p.sk_ntr_pb = wp.ntr / 16;
  p.fd_sklen = make_fastdiv((uint32_t)wp.len);
  dim3 grid(wp.grid);
  {
    VcgProfScope prof(wp.bm == 128 ? (wp.bn == 128 ? "k_conv_wgrad_split<128>" : "k_conv_wgrad_split<64>") : "k_conv_wgrad<fp32 MFMA>",
                      2.0 * 16 * (double)T * q.K * q.Cout, st);
    if (wp.bm == 128 && wp.bn == 128) hipLaunchKernelGGL(k_conv_wgrad_split<128>, grid, dim3(256), 0, st, p);        // split-operand
    else if (wp.bm == 128 && wp.bn == 64) hipLaunchKernelGGL(k_conv_wgrad_split<64>, grid, dim3(256), 0, st, p);
    else if (wp.bm == 64 && wp.bn == 128) hipLaunchKernelGGL((k_conv_wgrad<64, 128>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_conv_wgrad<64, 64>), grid, dim3(256), 0, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv_wgrad(winograd gemm)");
  hipLaunchKernelGGL(k_wino_wgrad_reduce, dim3(g.Cout / 64, (g.Cin + 7) / 8, g.ups * g.ups), dim3(256), 0, st,
                     (const float*)ws, gw_oihw, p, g.Cin, g.ups, g.cin_log, g.cout_log);
  VCG_LAUNCH_CHECK("vcg_conv_wgrad(winograd reduce)");
  return 0;
}

// the back-transform of a dU that is already whole (vcg_wino_wgrad's transposed-operand GEMM): k_wino_wgrad_reduce over one part
int vcg_wino_wgrad_reduce_one(const ConvGeom& g, const float* dU, float* gw_oihw, hipStream_t st) {
  const ConvGeom q = wino_gemm_geom(g, 32);
  ConvP p; fill_params(q, p);
  p.nbatch = 16;
  p.ktiles_total = 1; p.sk_len = 1; p.sk_ntn = 1; p.sk_bm_shift = 30; p.sk_bn_shift = 30;      // sk_parts() == 1 everywhere
  p.fd_sklen = make_fastdiv(1u);
  hipLaunchKernelGGL(k_wino_wgrad_reduce, dim3(g.Cout / 64, (g.Cin + 7) / 8, g.ups * g.ups), dim3(256), 0, st, dU, gw_oihw, p, g.Cin,
                     g.ups, g.cin_log, g.cout_log);
  VCG_LAUNCH_CHECK("vcg_conv_wgrad(winograd reduce)");
  return 0;
}

extern "C" size_t vcg_conv_wgrad_workspace(const int32_t* cd) {
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_wgrad_workspace")) return 0;
  const ConvGeom gorig = g;
  int tc, cgroups, rows, nchunk;
  colsum_plan(gorig, tc, cgroups, rows, nchunk);
  size_t cols = (size_t)nchunk * gorig.Cout * sizeof(float);
  if (vcg_wino_wgrad_ok(g)) return ((vcg_wino_wgrad_workspace(g) + 255) / 256) * 256 + cols + 1024;
  if (wgrad_ring_ok(g)) return ((vcg_ring_wgrad_workspace(g) + 255) / 256) * 256 + cols + 1024;
  if (wgrad_swapped_ok(g)) g = swapped_geom(g);
  const WgradPlan wp = wgrad_plan(g);
  const int nsplit = wp.parts;
  size_t slabs = (size_t)nsplit * g.K * g.Cout * sizeof(float);
  size_t groups = (size_t)16 * g.K * g.Cout * sizeof(float);     // k_slab_sum output (used when nsplit > 8)
  return slabs + groups + cols + 1024;
}

// Thin Cout (the decoder head, 64 -> 3): dW[(tap,c)][co<4] as an MFMA GEMM wastes 15/16 of the tile.  With
// the roles swapped — rows (tap, co) gathered from the 4-channel dy through the adjoint of the padding,
// columns c from x, summed over INPUT pixels — it becomes a well-shaped 196 x 64 weight gradient.
static bool wgrad_swapped_ok(const ConvGeom& g) {
  return g.Cout == 4 && g.stride == 1 && g.ups == 1 && g.Ho == g.H && g.Wo == g.W && g.Cin >= 32;
}
static ConvGeom swapped_geom(const ConvGeom& g) {
  ConvGeom s = g;
  s.Cin = 4; s.Cout = g.Cin; s.cin_log = g.cout_log; s.cout_log = g.cin_log;
  s.taps = g.KH * g.KW; s.K = s.taps * 4; s.M = g.N * g.H * g.W; s.act = 0;
  return s;
}

static int launch_colsum(const ConvGeom& gorig, const float* dy, float* gbias, float* part, hipStream_t st) {
  int tc, cgroups, rows, nchunk;
  colsum_plan(gorig, tc, cgroups, rows, nchunk);
  uint32_t* ctr = vcg_in_tail_enabled() ? vcg_tail_counters(cgroups) : nullptr;
  hipLaunchKernelGGL(k_colsum_partial, dim3(cgroups, nchunk), dim3(256), 0, st, dy, part, gorig.M, gorig.Cout, rows, tc, ctr, gbias,
                     gorig.cout_log);
  if (!ctr)
    hipLaunchKernelGGL(k_colsum_final, dim3((gorig.cout_log + 7) / 8), dim3(256), 0, st, (const float*)part, gbias,
                       gorig.Cout, nchunk, gorig.cout_log);
  VCG_LAUNCH_CHECK("vcg_conv_wgrad(bias)");
  return 0;
}

extern "C" int vcg_conv_wgrad_saved(const float* x, const float* dy, float* gw_oihw, float* gbias, const float* saved,
                                    const int32_t* cd, void* ws, size_t ws_bytes, void* stream);
extern "C" int vcg_conv_wgrad(const float* x, const float* dy, float* gw_oihw, float* gbias,
                              const int32_t* cd, void* ws, size_t ws_bytes, void* stream) {
  return vcg_conv_wgrad_saved(x, dy, gw_oihw, gbias, nullptr, cd, ws, ws_bytes, stream);
}
// `saved`: what vcg_conv_fwd_in left for this very (x, cd) in its `saved` buffer (vcg_conv_saved_floats), or null
extern "C" int vcg_conv_wgrad_saved(const float* x, const float* dy, float* gw_oihw, float* gbias, const float* saved,
                                    const int32_t* cd, void* ws, size_t ws_bytes, void* stream) {
  const uint64_t x_handle = vcg_take_hint_x(), dy_handle = vcg_take_hint_dy();      // vcg_amax_hint
  ConvGeom g;
  if (vcg_conv_geom(cd, &g, "vcg_conv_wgrad")) return -1;
  VCG_CHECK_ARG(x && dy && gw_oihw && ws, "vcg_conv_wgrad: null pointer");
  size_t need = vcg_conv_wgrad_workspace(cd);
  VCG_CHECK_ARG(ws_bytes >= need, "vcg_conv_wgrad: workspace %zu < %zu", ws_bytes, need);
  if (vcg_wino_wgrad_ok(g)) {
    const size_t wbytes = vcg_wino_wgrad_workspace(g);
    if (vcg_wino_wgrad(g, x, dy, gw_oihw, ws, wbytes, (hipStream_t)stream, saved, x_handle, dy_handle)) return -2;
    if (gbias) return launch_colsum(g, dy, gbias, (float*)((char*)ws + ((wbytes + 255) / 256) * 256), (hipStream_t)stream);
    return 0;
  }
  if (wgrad_ring_ok(g)) {
    const size_t rbytes = vcg_ring_wgrad_workspace(g);
    if (vcg_ring_wgrad(g, x, dy, gw_oihw, ws, rbytes, (hipStream_t)stream, x_handle, dy_handle)) return -2;
    if (gbias) return launch_colsum(g, dy, gbias, (float*)((char*)ws + ((rbytes + 255) / 256) * 256), (hipStream_t)stream);
    return 0;
  }
  const bool swapped = wgrad_swapped_ok(g);
  const ConvGeom gorig = g;
  if (swapped) g = swapped_geom(g);
  ConvP p; fill_params(g, p);
  const WgradPlan wp = wgrad_plan(g);
  const int bm = wp.bm, bn = wp.bn, nsplit = wp.parts;
  p.a = x; p.b = dy; p.out = (float*)ws;
  if (swapped) { p.a = dy; p.b = x; p.adjoint = 1; p.src_pitch = 4; }
  {
    // a: the gathered side (x; dy in swapped mode), b: the plain [K' pixel][Cout] side
    const unsigned long long ab = swapped ? (unsigned long long)gorig.M * 4 * 4 : (unsigned long long)g.N * g.H * g.W * g.Cin * 4;
    const unsigned long long bb = (unsigned long long)g.M * g.Cout * 4;
    VCG_CHECK_ARG(ab < (1ull << 31) && bb < (1ull << 31), "vcg_conv_wgrad: tensor extents must stay below 2 GiB");
    p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  }
  p.ktiles_total = wp.total; p.sk_len = wp.len; p.sk_units = wp.ntr * wp.ntn * wp.total; p.sk_ntn = wp.ntn;
  p.sk_bm_shift = bm == 256 ? 8 : bm == 128 ? 7 : 6; p.sk_bn_shift = bn == 128 ? 7 : 6;
  p.fd_sklen = make_fastdiv((uint32_t)wp.len);
  dim3 grid(wp.grid);
  hipStream_t st = (hipStream_t)stream;
  if (bm == 128) {                                          // fp16 x 2 kernel: both operands are activations, scaled by their own amax
    // swapped roles (a = dy, b = x): the rows are gathered from dy through the adjoint of the padding, i.e. as sums of up to 4 (9) sources
    if (vcg_operand_amax(p.a, (size_t)p.a_bytes / 4, swapped ? dy_handle : x_handle, (swapped && gorig.reflect) ? 4 : 0, st, &p.amax_a) ||
        vcg_operand_amax(p.b, (size_t)p.b_bytes / 4, swapped ? x_handle : dy_handle, 0, st, &p.amax_b))
      return -2;
  }
  {
    VcgProfScope prof(bm == 128 ? (bn == 128 ? "k_conv_wgrad_split<128>" : "k_conv_wgrad_split<64>") : "k_conv_wgrad<fp32 MFMA>",
                      2.0 * g.M * (double)g.K * g.Cout, st);
    if (bm == 256) hipLaunchKernelGGL((k_conv_wgrad<256, 128, 512>), grid, dim3(512), 0, st, p);
    else if (bm == 128 && bn == 128) hipLaunchKernelGGL(k_conv_wgrad_split<128>, grid, dim3(256), 0, st, p);     // split-operand
    else if (bm == 128 && bn == 64) hipLaunchKernelGGL(k_conv_wgrad_split<64>, grid, dim3(256), 0, st, p);
    else if (bm == 64 && bn == 128) hipLaunchKernelGGL((k_conv_wgrad<64, 128>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_conv_wgrad<64, 64>), grid, dim3(256), 0, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv_wgrad");
  const size_t totalw = (size_t)g.K * g.Cout;
  const size_t slab_bytes = (((size_t)nsplit * totalw * sizeof(float) + 255) / 256) * 256;
  const float* src = (const float*)ws;
  int ns = 0;                                  // 0: the final kernels sum each tile's own part count
  if (nsplit > kMaxDirectSlabs) {              // many thin slabs: parallel pre-sum into <= 16 group slabs
    float* grp = (float*)((char*)ws + slab_bytes);
    int per_group = (nsplit + 15) / 16;
    int G = (nsplit + per_group - 1) / per_group;
    size_t total4 = totalw / 4;
    int bx = (int)((total4 + 255) / 256); if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_slab_sum, dim3(bx, G), dim3(256), 0, st, src, grp, total4, p, per_group);
    src = grp;
    ns = G;
  }
  const int T = g.KH * g.KW * g.ups * g.ups;
  const size_t lds = (size_t)T * 8 * 33 * sizeof(float);
  if (swapped) {
    int blocks = (int)((totalw + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_wgrad_scatter_swapped, dim3(blocks), dim3(256), 0, st, src, gw_oihw, p, g.KH * g.KW, g.Cout, ns,
                       gorig.cin_log, gorig.cout_log);
  } else if (totalw < (1u << 20) || lds > 64 * 1024) {
    int blocks = (int)((totalw + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_wgrad_scatter, dim3(blocks), dim3(256), 0, st, src, gw_oihw, p, ns, g.cin_log, g.cout_log);
  } else {
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((g.Cout + 31) / 32, (g.Cin + 7) / 8), dim3(256), lds, st, src, gw_oihw, p,
                       ns, g.cin_log, g.cout_log);
  }
  VCG_LAUNCH_CHECK("vcg_conv_wgrad(reduce)");
  if (gbias) {
    float* part = (float*)((char*)ws + slab_bytes + (((size_t)16 * totalw * sizeof(float) + 255) / 256) * 256);
    return launch_colsum(gorig, dy, gbias, part, st);
  }
  return 0;
}
