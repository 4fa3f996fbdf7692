// Weight gradients of the layers whose 64-column implicit-GEMM tiles starve (U4: 32 -> 64 at 256^2; the 7x7 head 64 -> 3
// and stem 3 -> 64): k_conv_wgrad_split<64> ran them at 13-14 % of the (then bf16 x 3) pipe, the im2col operand being gathered, split
// and stored once per tap (and, for the 4-channel tensors, through the adjoint-of-padding gather per element).
//
// Here the reduction walks the image row by row.  A workgroup owns a 32-pixel column segment of a band of rows; per row it
// stages ONE row segment of each operand:
//   * the "wide" operand — 64 channels, [32 px][64] — as the plain transposable tile of k_conv_wgrad_split;
//   * the "ring" operand — the one the taps shift — into a ring of KH + 1 row slots, where it stays for KH steps:
//       U mode  (3x3, 32 channels per group):  [34 px][32 ch]; tap (kh, kw) = slot kh, pixel offset kw;
//       T mode  (7x7, a 4-channel tensor):     [39 px][4 ch] — the 32 columns of a tap row kh are (j = 0..7, c = 0..3),
//               i.e. 32 CONTIGUOUS 16-bit values starting at pixel px + j: the Toeplitz matrix the kw-folded GEMM needs never
//               exists, a lane of ds_read_b64_tr_b16 simply addresses pixel (row + q + j).
// The MFMA's reduction index is the pixel, which is the slow index of both images: fragments come out through the
// transposing LDS read, as in k_conv_wgrad_split.  Four waves: wave w owns wide column block w & 1 and the upper or lower
// half of the taps (5 + 4 of 9, 4 + 3 of 7): <= 5 accumulator tiles; the six split products of a tile and slice are summed
// in a zero-initialised chain and added to the accumulator once (one rounding of the running sum per slice).
// Every workgroup leaves its partial [rows][64] in its own slab; k_ring_sum / k_ring_scatter add them in a fixed order
// (bit-reproducible) and accumulate into the OIHW gradient.
#include "vcg_common.h"

typedef unsigned int rg_u32x4 __attribute__((ext_vector_type(4)));
typedef short rg_s16x4 __attribute__((ext_vector_type(4)));
typedef short rg_s16x8 __attribute__((ext_vector_type(8)));
#define RG_OOB 0x80000000u

struct RingP {
  const float* ring;      // the shifted operand: (N, Hr, Wr, Cr) fp32
  const float* wide;      // the plain operand:   (N, Hw, Ww, 64 * nwt) fp32 (channel pitch Cw)
  float* slabs;           // [workgroup][NR][64]
  int N, Hr, Wr, Cr, Hw, Ww, Cw;
  int Yd, Xd;             // the reduction domain (rows, columns of K positions)
  int r_reflect, r_off;   // ring coordinate of (K row y, tap row kh) = y + kh + r_shift - r_off, then reflect / zero outside
  int r_shift;
  int w_reflect, w_off;   // wide coordinate of K position (y, x) = (y - w_off, x - w_off), then reflect / zero outside
  int nseg, nrs, rows_per_band;
  int r_ups, r_gpp;       // U mode with a folded PixelUnshuffle: the ring tensor is (N, Hr * ups, Wr * ups, Cr), channel group z holds
                          // phase z / r_gpp (i = phase >> 1, j = phase & 1), channels 32 * (z % r_gpp) .. + 31   (ups 1: r_gpp = Cr / 32)
  uint32_t ring_bytes, wide_bytes;
  VcgAmax amax_r, amax_w;   // largest magnitudes of the ring and of the wide tensor (fp16 x 2 operand scales, vcg_common.h)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rg_srd(const void* ptr, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 rg_load4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  const rg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// byte offset of element (row, col) in a [rows][64] fp16 image (128-byte rows, 16-byte chunks XOR-swizzled: conv_igemm.hip tr_off<64>)
__device__ __forceinline__ uint32_t rg_woff(int row, int col) {
  const int f = (((row & 3) << 2) | ((row >> 2) & 3)) & 7;
  return (uint32_t)(128 * row + 16 * ((col >> 3) ^ f) + (col & 7) * 2);
}
typedef __attribute__((address_space(3))) unsigned char rg_lds_t;       // 32-bit LDS addresses: no flat-pointer arithmetic per fragment
__device__ __forceinline__ f16x8 rg_tr2(rg_lds_t* a0, rg_lds_t* a1) {
  const rg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rg_s16x4*)a0);
  const rg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rg_s16x4*)a1);
  const rg_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(f16x8, v);
}

template <bool THIN>
__global__ __launch_bounds__(256, 2) void k_wgrad_ring(RingP p) {
  constexpr int KH = THIN ? 7 : 3, NT = THIN ? 7 : 9, SLOTS = KH + 1;
  constexpr int RPX = THIN ? 40 : 34;                        // ring pixels per row segment (32 + taps - 1, T mode padded to 40)
  constexpr int RROW = THIN ? 8 : 64;                        // bytes per ring pixel and piece
  constexpr int RQ = THIN ? RPX : RPX * 8;                   // float4 quads per ring row segment
  constexpr int RA = (RQ + 255) / 256;
  constexpr int MAXT = THIN ? 4 : 5;                         // accumulator tiles per wave
  __shared__ __attribute__((aligned(16))) unsigned char Rs[SLOTS][VCG_NP][RPX * RROW];
  __shared__ __attribute__((aligned(16))) unsigned char Ws[2][VCG_NP][32 * 128];
  float sR, invR, sW, invW;
  vcg_scale_of(vcg_amax_bits(p.amax_r), p.amax_r.shift, sR, invR);
  vcg_scale_of(vcg_amax_bits(p.amax_w), p.amax_w.shift, sW, invW);
  const float oscale = sR * sW;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and known to be
  const int l31 = lane & 31, lh = lane >> 5;
  const int nb = wid & 1, th = wid >> 1;
  const int tap0 = th ? (NT + 1) / 2 : 0, ntap = th ? NT / 2 : (NT + 1) / 2;

  int blk = blockIdx.x;
  const int n = blk / (p.nseg * p.nrs);
  blk -= n * p.nseg * p.nrs;
  const int seg = blk / p.nrs, rs = blk - seg * p.nrs;
  const int x0 = seg * 32;
  const int y0 = rs * p.rows_per_band;
  int y1 = y0 + p.rows_per_band;
  if (y1 > p.Yd) y1 = p.Yd;
  const int rph = THIN ? 0 : (int)blockIdx.z / p.r_gpp;      // ring channel group (U mode): unshuffle phase and first channel
  const int cg = THIN ? 0 : ((int)blockIdx.z - rph * p.r_gpp) * 32;
  const int rpi = rph >> 1, rpj = rph & 1;
  const int wt = (int)blockIdx.y * 64;                       // wide channel tile

  const __amdgpu_buffer_rsrc_t rr = rg_srd(p.ring, p.ring_bytes), rw = rg_srd(p.wide, p.wide_bytes);

  // ---- staging maps -------------------------------------------------------------------------------------------------
  // ring quads: (pixel, 4-channel quad) of a row segment -> column offset in the tensor (bytes), LDS offset
  uint32_t rcol[RA], rlds[RA];
#pragma unroll
  for (int a = 0; a < RA; ++a) {
    const int idx = tid + 256 * a;
    const int px = THIN ? idx : idx >> 3, q = THIN ? 0 : idx & 7;
    bool ok = idx < RQ;
    int xc = x0 + px + p.r_shift - p.r_off;                  // ring column of segment pixel px
    if (p.r_reflect) {
      ok = ok && xc > -p.Wr && xc < 2 * p.Wr - 1;
      xc = reflect_idx(xc, p.Wr);
    } else {
      ok = ok && xc >= 0 && xc < p.Wr;
    }
    rcol[a] = ok ? (uint32_t)(((xc * p.r_ups + rpj) * p.Cr + cg + q * 4) * 4) : RG_OOB;
    rlds[a] = idx < RQ ? (uint32_t)(px * RROW + q * 8) : RG_OOB;
  }
  // wide quads: 32 px x 16 quads = 512 -> 2 per thread
  uint32_t wcol[2], wlds[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + 256 * e;
    const int px = idx >> 4, q = idx & 15;
    int xc = x0 + px - p.w_off;
    bool ok = x0 + px < p.Xd;                                // K positions past the domain contribute nothing
    if (p.w_reflect) {
      ok = ok && xc > -p.Ww && xc < 2 * p.Ww - 1;
      xc = reflect_idx(xc, p.Ww);
    } else {
      ok = ok && xc >= 0 && xc < p.Ww;
    }
    wcol[e] = ok ? (uint32_t)((xc * p.Cw + wt + q * 4) * 4) : RG_OOB;
    wlds[e] = rg_woff(px, q * 4);
  }
  // byte offset of ring row rho (ring coordinates) / of wide row y in their tensors, or RG_OOB (-> zeros)
  auto ring_row = [&](int rho) -> uint32_t {
    int yr = rho + p.r_shift - p.r_off;
    if (p.r_reflect) {
      if (yr <= -p.Hr || yr >= 2 * p.Hr - 1) return RG_OOB;
      yr = reflect_idx(yr, p.Hr);
    } else if (yr < 0 || yr >= p.Hr) {
      return RG_OOB;
    }
    return (uint32_t)((((n * p.Hr + yr) * p.r_ups + rpi) * p.Wr * p.r_ups) * p.Cr * 4);
  };
  auto wide_row = [&](int y) -> uint32_t {
    int yw = y - p.w_off;
    if (p.w_reflect) {
      if (yw <= -p.Hw || yw >= 2 * p.Hw - 1) return RG_OOB;
      yw = reflect_idx(yw, p.Hw);
    } else if (yw < 0 || yw >= p.Hw) {
      return RG_OOB;
    }
    return (uint32_t)(((n * p.Hw + yw) * p.Ww) * p.Cw * 4);
  };

  float4 vr[RA], vw[2];
  auto load_ring = [&](int rho) {
    const uint32_t base = ring_row(rho);
#pragma unroll
    for (int a = 0; a < RA; ++a) vr[a] = rg_load4(rr, (base != RG_OOB && rcol[a] != RG_OOB) ? base + rcol[a] : RG_OOB);
  };
  auto store_ring = [&](int rho) {
    const int slot = rho % SLOTS;
#pragma unroll
    for (int a = 0; a < RA; ++a) {
      if (rlds[a] == RG_OOB) continue;
      uint2 h, l;
      split4h(vr[a], invR, h, l);
      *reinterpret_cast<uint2*>(&Rs[slot][0][rlds[a]]) = h;
      *reinterpret_cast<uint2*>(&Rs[slot][1][rlds[a]]) = l;
    }
  };
  auto load_wide = [&](int y) {
    const uint32_t base = wide_row(y);
#pragma unroll
    for (int e = 0; e < 2; ++e) vw[e] = rg_load4(rw, (base != RG_OOB && wcol[e] != RG_OOB) ? base + wcol[e] : RG_OOB);
  };
  auto store_wide = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      uint2 h, l;
      split4h(vw[e], invW, h, l);
      *reinterpret_cast<uint2*>(&Ws[buf][0][wlds[e]]) = h;
      *reinterpret_cast<uint2*>(&Ws[buf][1][wlds[e]]) = l;
    }
  };

  f32x16 acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // transposing fragment reads: 16-lane group g16 = (lane >> 4) & 1 covers columns 16 g16 .. + 15 of a 32-wide tile, the
  // reduction rows of half lh; inside it lane 4 q + pp addresses row q, columns 4 pp .. 4 pp + 3
  const int tq = (lane & 15) >> 2, tp = lane & 3, g16 = (lane >> 4) & 1;
  const int tcol = 16 * g16 + 4 * tp;
  rg_lds_t* const ldsR = (rg_lds_t*)&Rs[0][0][0];
  rg_lds_t* const ldsW = (rg_lds_t*)&Ws[0][0][0];
  constexpr int RPIECE = RPX * RROW, RSLOT = VCG_NP * RPIECE, WPIECE = 32 * 128, WBUF = VCG_NP * WPIECE;

  if (y0 < y1) {
    for (int k = 0; k < KH; ++k) {                          // the first step's tap rows
      load_ring(y0 + k);
      store_ring(y0 + k);
    }
    load_wide(y0);
    store_wide(0);
  }
  __syncthreads();
  for (int y = y0; y < y1; ++y) {
    const int buf = (y - y0) & 1;
    const bool more = y + 1 < y1;
    if (more) {
      load_ring(y + KH);
      load_wide(y + 1);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int row0 = 16 * s2 + 8 * lh + tq;                // reduction row (segment pixel) of this lane's first read
      f16x8 bfr[VCG_NP];
#pragma unroll
      for (int pc = 0; pc < VCG_NP; ++pc)
        bfr[pc] = rg_tr2(ldsW + buf * WBUF + pc * WPIECE + rg_woff(row0, nb * 32 + tcol),
                         ldsW + buf * WBUF + pc * WPIECE + rg_woff(row0 + 4, nb * 32 + tcol));
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t >= ntap) break;
        const int tap = tap0 + t;
        int slot, off;
        if (THIN) {                                          // tap row kh = tap; column (j, c): j = 4 g16 + tp
          slot = (y + tap) % SLOTS;
          off = (row0 + 4 * g16 + tp) * 8;
        } else {                                             // tap (kh, kw); column = channel tcol
          const int kh = tap / 3, kw = tap - kh * 3;
          slot = (y + kh) % SLOTS;
          off = (row0 + kw) * 64 + tcol * 2;
        }
        f16x8 af[VCG_NP];
#pragma unroll
        for (int pc = 0; pc < VCG_NP; ++pc)
          af[pc] = rg_tr2(ldsR + slot * RSLOT + pc * RPIECE + off, ldsR + slot * RSLOT + pc * RPIECE + off + 4 * RROW);
        f32x16 c;
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = 0.f;
        c = VCG_MFMA(af[1], bfr[0], c);                     // smallest contributions first
        c = VCG_MFMA(af[0], bfr[1], c);
        c = VCG_MFMA(af[0], bfr[0], c);
        acc[t] += c;
        asm volatile("" : "+v"(acc[t]));                     // pin the add here: sunk to the end of the step, all chains stay live
        __builtin_amdgcn_sched_barrier(0);                   // keep one tap's fragments and chain live at a time (register budget)
      }
    }
    if (more) {
      store_ring(y + KH);                                    // slot (y + KH) % SLOTS: not one of this step's KH slots
      store_wide(buf ^ 1);
    }
    __syncthreads();
  }

  // ---- the workgroup's partial: slab[(tap, m)][wide channel], m = the ring-side column of the tile -----------------------------
  const int wgid = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  float* slab = p.slabs + (size_t)wgid * (NT * 32) * 64;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t >= ntap) break;
    const int tap = tap0 + t;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = (e & 3) + 8 * (e >> 2) + 4 * lh;
      slab[(size_t)(tap * 32 + m) * 64 + nb * 32 + l31] = acc[t][e] * oscale;
    }
  }
}

// out[g][i] = sum over the g-th group of workgroup slabs (fixed order); the slabs of one (wide tile, channel group) are contiguous
__global__ __launch_bounds__(256) void k_ring_sum(const float4* __restrict__ slabs, float4* __restrict__ out, int total4, int nslab,
                                                  int per_group) {
  const int g = blockIdx.y, z0 = g * per_group;
  int z1 = z0 + per_group;
  if (z1 > nslab) z1 = nslab;
  const size_t sub = (size_t)blockIdx.z * nslab;             // (wide tile, channel group) index
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = z0; z < z1; ++z) {
      const float4 v = slabs[(sub + z) * total4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[((size_t)blockIdx.z * gridDim.y + g) * total4 + i] = s;
  }
}
// mode 0 (U): row (tap, c in group), column co      -> gw[co][(cg + c) * U2 + phase][tap]   (U2 = ups^2 phases of a folded PixelUnshuffle)
// mode 1 (stem): row (kh, j, c), column co          -> gw[co][c][kh * 7 + j]            (j < 7, c < cin_log)
// mode 2 (head): row (kh', j, co), column c          -> gw[co][c][(6 - kh') * 7 + 6 - j]  (j < 7, co < cout_log)
__global__ __launch_bounds__(256) void k_ring_scatter(const float* __restrict__ grp, float* __restrict__ gw, int G, int NR, int mode,
                                                      int cin_log, int cout_log, int ngroups, int ntiles, int U2, int gpp) {
  const int total = NR * 64;
  const int sub = blockIdx.y;                                // = wide tile + ntiles * channel group (the launch's (y, z) order)
  const int wtile = sub % ntiles, cgrp = sub / ntiles;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i >> 6, col = i & 63;
    int co, c, tap, KK, ph = 0;
    if (mode == 0) {
      ph = cgrp / gpp;
      tap = row >> 5; c = (cgrp - ph * gpp) * 32 + (row & 31); co = wtile * 64 + col; KK = 9;
    } else {
      const int kh = row >> 5, j = (row >> 2) & 7, q = row & 3;
      if (j >= 7) continue;
      KK = 49;
      if (mode == 1) { tap = kh * 7 + j; c = q; co = col; }
      else { tap = (6 - kh) * 7 + (6 - j); co = q; c = col; }
    }
    if (co >= cout_log || c >= cin_log) continue;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += grp[((size_t)sub * G + g) * total + i];
    gw[((size_t)co * (cin_log * U2) + (size_t)c * U2 + ph) * KK + tap] += s;
  }
}

// ------------------------------------------------------------------------------------------------------------- host side
static int ring_mode(const ConvGeom& g) {
  if (g.stride != 1 || g.KH != g.KW || (long long)g.Ho * g.Wo < 64 * 64 || g.Ho != g.Hl || g.Wo != g.Wl) return -1;
  if (g.KH == 3 && g.pad == 1 && g.Cin % 32 == 0 && g.ups * g.ups * g.Cin <= 256 && g.Cout % 64 == 0 && g.Cout <= 256) return 0;
  if (g.ups != 1) return -1;
  if (g.KH == 7 && g.pad == 3 && g.reflect && g.Cin == 4 && g.Cout == 64) return 1;
  if (g.KH == 7 && g.pad == 3 && g.reflect && g.Cin == 64 && g.Cout == 4) return 2;
  return -1;
}
bool vcg_ring_wgrad_ok(const ConvGeom& g) { return ring_mode(g) >= 0; }

struct RingPlan { int nseg, nrs, rows, nwg, ntiles, ngroups, NR, G, per_group; };
static RingPlan ring_plan(const ConvGeom& g) {
  RingPlan pl;
  const int mode = ring_mode(g);
  const int Yd = mode == 2 ? g.H + 6 : g.Ho, Xd = mode == 2 ? g.W + 6 : g.Wo;
  pl.nseg = (Xd + 31) / 32;
  pl.ntiles = mode == 0 ? g.Cout / 64 : 1;
  pl.ngroups = mode == 0 ? g.ups * g.ups * g.Cin / 32 : 1;
  const int sub = pl.ntiles * pl.ngroups;
  int nrs = 768 / (g.N * pl.nseg * sub);                     // ~768 resident workgroups (3 per CU: 164 / 134 VGPRs, 51 / 32 KB of LDS)
  if (nrs < 1) nrs = 1;
  int rows = (Yd + nrs - 1) / nrs;
  if (rows < 8) rows = 8;                                    // a band pays KH - 1 ring rows of prologue
  pl.rows = rows;
  pl.nrs = (Yd + rows - 1) / rows;
  pl.nwg = g.N * pl.nseg * pl.nrs;
  pl.NR = (mode == 0 ? 9 : 7) * 32;
  pl.per_group = (pl.nwg + 15) / 16;
  pl.G = (pl.nwg + pl.per_group - 1) / pl.per_group;
  return pl;
}
size_t vcg_ring_wgrad_workspace(const ConvGeom& g) {
  const RingPlan pl = ring_plan(g);
  const size_t sub = (size_t)pl.ntiles * pl.ngroups;
  return (sub * pl.nwg + sub * pl.G) * pl.NR * 64 * sizeof(float) + 256;
}

int vcg_ring_wgrad(const ConvGeom& g, const float* x, const float* dy, float* gw_oihw, void* ws, size_t ws_bytes, hipStream_t st,
                   uint64_t x_handle, uint64_t dy_handle) {
  const int mode = ring_mode(g);
  VCG_CHECK_ARG(mode >= 0, "vcg_conv_wgrad(ring): unsupported layer");
  VCG_CHECK_ARG(ws && ws_bytes >= vcg_ring_wgrad_workspace(g), "vcg_conv_wgrad(ring): workspace too small");
  const RingPlan pl = ring_plan(g);
  RingP p = {};
  p.N = g.N;
  const unsigned long long xb = (unsigned long long)g.N * g.H * g.W * g.Cin * 4, db = (unsigned long long)g.N * g.Ho * g.Wo * g.Cout * 4;
  VCG_CHECK_ARG(xb < (1ull << 31) && db < (1ull << 31), "vcg_conv_wgrad: tensor extents must stay below 2 GiB");
  if (mode == 2) {                       // head: the 4-channel dy shifts (zero outside), x is the plain operand, read through the padding
    p.ring = dy; p.Hr = g.Ho; p.Wr = g.Wo; p.Cr = 4; p.ring_bytes = (uint32_t)db;
    p.wide = x; p.Hw = g.H; p.Ww = g.W; p.Cw = g.Cin; p.wide_bytes = (uint32_t)xb;
    p.Yd = g.H + 6; p.Xd = g.W + 6;
    p.r_reflect = 0; p.r_off = 0; p.r_shift = -6;
    p.w_reflect = 1; p.w_off = 3;
  } else {                               // U / stem: x shifts (through the padding), dy is the plain operand
    const int pad = g.pad;
    p.ring = x; p.Hr = g.Hl; p.Wr = g.Wl; p.Cr = g.Cin; p.ring_bytes = (uint32_t)xb;
    p.wide = dy; p.Hw = g.Ho; p.Ww = g.Wo; p.Cw = g.Cout; p.wide_bytes = (uint32_t)db;
    p.Yd = g.Ho; p.Xd = g.Wo;
    p.r_reflect = g.reflect; p.r_off = pad; p.r_shift = 0;
    p.w_reflect = 0; p.w_off = 0;
  }
  p.r_ups = mode == 0 ? g.ups : 1; p.r_gpp = mode == 0 ? g.Cin / 32 : 1;
  // the operands' largest magnitudes: from the handles of whoever wrote x / dy, else measured (mode 2: dy is the ring)
  if (vcg_operand_amax(p.ring, (size_t)p.ring_bytes / 4, mode == 2 ? dy_handle : x_handle, 0, st, &p.amax_r) ||
      vcg_operand_amax(p.wide, (size_t)p.wide_bytes / 4, mode == 2 ? x_handle : dy_handle, 0, st, &p.amax_w))
    return -2;
  p.nseg = pl.nseg; p.nrs = pl.nrs; p.rows_per_band = pl.rows;
  p.slabs = (float*)ws;
  const dim3 grid(pl.nwg, pl.ntiles, pl.ngroups);
  {
    const double flops = 2.0 * g.N * (double)p.Yd * p.Xd * pl.NR * 64.0 * pl.ntiles * pl.ngroups;
    VcgProfScope prof(mode == 0 ? "k_wgrad_ring<false>" : "k_wgrad_ring<true>", flops, st);
    if (mode == 0) hipLaunchKernelGGL(k_wgrad_ring<false>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_wgrad_ring<true>, grid, dim3(256), 0, st, p);
  }
  VCG_LAUNCH_CHECK("vcg_conv_wgrad(ring)");
  const int sub = pl.ntiles * pl.ngroups;
  const int total4 = pl.NR * 64 / 4;
  float* grp = (float*)ws + (size_t)sub * pl.nwg * pl.NR * 64;
  hipLaunchKernelGGL(k_ring_sum, dim3((total4 + 255) / 256, pl.G, sub), dim3(256), 0, st, (const float4*)ws, (float4*)grp, total4, pl.nwg,
                     pl.per_group);
  hipLaunchKernelGGL(k_ring_scatter, dim3((pl.NR * 64 + 255) / 256, sub), dim3(256), 0, st, (const float*)grp, gw_oihw, pl.G, pl.NR, mode,
                     g.cin_log, g.cout_log, pl.ngroups, pl.ntiles, g.ups * g.ups, mode == 0 ? g.Cin / 32 : 1);
  VCG_LAUNCH_CHECK("vcg_conv_wgrad(ring reduce)");
  return 0;
}
