// EXPERIMENT (round 3, not part of libvcg.so; built by tools/gemm_ws_probe.hip): a wave-specialised schedule for the split-operand
// GEMM of the Winograd layers, in the bf16 x 3 arithmetic of rounds 1-2 (fp32 operands as three bf16 pieces, six MFMAs per
// product, two fp32 accumulation chains).  Result (profiles/r03_gemm_ws_probe*.txt): within +-5 % of the 4-wave kernel in every
// variant — the chip clocks down as the MFMA stream gets denser — so the schedule was not adopted; what was: half the MFMAs.
//
//   C[z][m][n] = sum_k A[z][m][k] * Bt[z][n][k]        A fp32 (or, MODE_WINO, produced on the fly from the image: below),
//                                                      Bt pre-split "blocked planes" (gemm_split.hip), C fp32
//
// k_gemm_split runs two 4-wave workgroups per CU; every wave alternates between staging (global -> split -> LDS) and its
// MFMAs, with two barriers per 32-deep K-step, and sits at 36-38 % of the bf16 pipe peak whatever feeds it (DESIGN.md §3:
// the "128^2 tile, two barriers per K-step" ceiling of cdna_hip_programming.md §5).  Here ONE 8-wave workgroup owns the CU:
//
//   waves 0-3  PRODUCERS   global loads (one K-step ahead, in registers) -> split -> ds_write into the stage the consumers
//                          are NOT reading; no MFMA, no accumulators: their VALU and LDS-store work runs on the vector
//                          pipe beside the partner wave's matrix instructions (the two pipes are separate per SIMD);
//   waves 4-7  CONSUMERS   one per SIMD, 64 x 64 of the 128 x 128 tile each: ds_read_b128 fragments (double-buffered in
//                          registers, the next slice's reads issued before the current slice's 24 MFMAs) and MFMAs, nothing
//                          else in the loop.
//
// Two LDS stages (2 x 48 KB) and ONE barrier per K-step: at barrier k the producers have finished writing stage k+1 and
// the consumers have all their fragments of stage k in registers; the consumers' last 24 MFMAs of stage k are issued
// AFTER that barrier, behind the first fragment reads of stage k+1, so the matrix pipe has work on both sides of it.
//
// MODE_WINO: the A operand of the forward Winograd GEMM, V[xi] = (B^T d B)[xi], is computed by the producers from the
// activation itself (reflect / zero padding and the folded PixelUnshuffle in the gather, as k_wino_in does): per value of
// transform point xi = (a, b) a 2 x 2 sub-patch is read (B^T has two non-zeros per row) — 4 loads and 3 adds instead of
// one load — and V never exists in HBM.  Used for the layers whose V traffic bounds them and that have one or two
// N tiles (D1, D2, U2: no transform is repeated more than twice); the compute-bound many-N-tile layers keep a
// materialised V (33-134 MB, L2 / MALL resident) so that the transform is done once.
#include "../vae-cyclegan-implementation_amd/csrc/vcg_common.h"
#include <stdlib.h>

// (this experiment runs the bf16 x 3 arithmetic of rounds 1-2: x = h + m + l, six products)
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

typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));

struct GemmWsP {
  const float* a;          // MODE 0: A [z][rows][K] fp32;  MODE_WINO: the image x (N, H, W, Cin) fp32 NHWC
  const void* bt;          // blocked planes [z][N][K/32][3][32] bf16
  float* c;
  int rows, K, N;
  uint32_t a_bytes, b_bytes;
  uint32_t a_bstride, b_bstride;
  size_t c_bstride;
  // MODE_WINO geometry (conv_wino.hip's WinoP)
  int H, W, Cin, Hl, Wl, ups, reflect, off, th, tw;
  FastDiv fd_tw, fd_thtw;
};

#define GW_OOB 0x80000000u

__device__ __forceinline__ float4 gw_bload4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// raw workgroup barrier: LDS traffic of this wave complete, no wait on outstanding global loads (a producer's prefetch of the
// next K-step stays in flight across it; __syncthreads() would drain it with vmcnt(0))
// (the wait is the builtin, not inline asm: hipcc's own waitcnt pass sees it and does not wait again for the same reads later)
__device__ __forceinline__ void gw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0), vmcnt / expcnt untouched (gfx9 encoding)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

constexpr int GW_STAGE = 2 * 3 * 128 * 64;       // bytes per stage: A pieces then B pieces, 8 KB each

// Diagnostic build only (tools/gemm_ws_probe.hip, -DVCG_WS_STAMP; libvcg.so never has it): per-wave shader-clock totals of the
// loop's phases, into a buffer of their own ([workgroup][wave][8] u64)
#ifdef VCG_WS_STAMP
__device__ unsigned long long* g_ws_stamp = nullptr;
int vcg_ws_set_stamp(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stamp), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#define WS_T() __builtin_amdgcn_s_memtime()
#define WS_ACC(slot, t0) do { const unsigned long long t1__ = WS_T(); st_acc[slot] += t1__ - (t0); (t0) = t1__; } while (0)
#else
#define WS_T() 0ull
#define WS_ACC(slot, t0) do { } while (0)
#endif

// 16-byte chunk swizzle of a 64-byte LDS row.  SHAPE 32 (v_mfma_f32_32x32x16_bf16: a ds_read_b128 has lane = row, 16 consecutive
// rows per LDS cycle, one chunk index): chunk ^ ((row >> 2) & 3).  SHAPE 16 (v_mfma_f32_16x16x32_bf16: lane l reads row l & 15,
// chunk l >> 4; the hardware's 16-lane groups {0-3, 12-15, 20-27}, ... then hold rows {0-3, 12-15} of one chunk and rows 4-11 of
// the next): chunk ^ f(row >> 2) with f = (0, 2, 3, 1) puts every group on 16 different 16-byte bank slots.
template <int SHAPE>
__device__ __forceinline__ int gw_swz(int row) {
  const int g = (row >> 2) & 3;
  return SHAPE == 32 ? g : ((0x78 >> (2 * g)) & 3);
}
typedef float f32x4w __attribute__((ext_vector_type(4)));

template <int MODE, int SHAPE = 32>
__global__ __launch_bounds__(512, 2) void k_gemm_ws(GemmWsP p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * GW_STAGE];
  const int tid = threadIdx.x;
  // XCD-aware tile order (as k_gemm_split): the N tiles that share an A tile run back to back on one XCD
  int mt, nt, zb;
  {
    const uint32_t per = gridDim.x * gridDim.y, nwg = per * gridDim.z;
    const uint32_t gid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const uint32_t q = nwg >> 3, r = nwg & 7, xcd = gid & 7;
    const uint32_t swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (gid >> 3);
    zb = (int)(swz / per);
    const uint32_t l = swz - (uint32_t)zb * per;
    mt = (int)(l / gridDim.y);
    nt = (int)(l - (uint32_t)mt * gridDim.y);
  }
  const int m0 = mt * 128, n0 = nt * 128;
  const int nkt = p.K / 32;
#ifdef VCG_WS_STAMP
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_t = WS_T();
  const unsigned long long st_begin = st_t;
  auto st_flush = [&]() {
    if (g_ws_stamp && (tid & 63) == 0) {
      unsigned long long* o = g_ws_stamp + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (tid >> 6)) * 8;
      for (int i = 0; i < 7; ++i) o[i] = st_acc[i];
      o[7] = WS_T() - st_begin;
    }
  };
#endif

  if (tid < 256) {
    // ------------------------------------------------------------------------------------------------ producers
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(MODE == 0 ? (const void*)(p.a + (size_t)zb * p.a_bstride)
                          : MODE == 2 ? (const void*)((const unsigned short*)p.a + (size_t)zb * p.a_bstride) : (const void*)p.a),
        0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const unsigned short*)p.bt + (size_t)zb * p.b_bstride), 0, (int)p.b_bytes, 0x00020000);
    const int s_row = tid >> 3, s_u = tid & 7;                    // A staging: rows s_row + 32 i, k quad s_u
    uint32_t soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = s_row + 32 * i;
      soff[i] = (uint32_t)(r * 64 + (((s_u >> 1) ^ gw_swz<SHAPE>(r)) << 4) + ((s_u & 1) << 3));
    }
    // B planes: thread (row b_r + 64 (j / 3), piece j % 3, 16-byte chunk b_q of the 64-byte piece row)
    const int b_q = tid & 3, b_r = tid >> 2;
    const int KB = nkt;
    const uint32_t boff0 = (uint32_t)(((size_t)(n0 + b_r) * KB) * 192 + b_q * 16);      // N % 128 == 0: every row is in range
    const uint32_t bhalf = (uint32_t)KB * (64u * 192u);
    const uint32_t bsoff0 = (uint32_t)(b_r * 64 + ((b_q ^ gw_swz<SHAPE>(b_r)) << 4));

    // A addressing
    uint32_t aoff[MODE == 1 ? 16 : (MODE == 0 ? 4 : 2)];
    float sgn[3] = {1.f, 1.f, 1.f};                               // MODE_WINO: signs of patch elements 1, 2, 3 (element 0: +)
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = m0 + s_row + 32 * i;
        aoff[i] = r < p.rows ? (uint32_t)(((size_t)r * p.K + s_u * 4) * 4) : GW_OOB;
      }
    } else if constexpr (MODE == 2) {                             // A pre-split by its producer: the same 16-byte copies as B
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = m0 + b_r + 64 * h;
        aoff[h] = r < p.rows ? (uint32_t)(((size_t)r * KB) * 192 + b_q * 16) : GW_OOB;
      }
    } else {
      // transform point xi = zb = (a, b): V[a][b] = sum over rows R_a and columns S_b of sign * d[r][s], with
      //   B^T rows: a = 0: d0 - d2;  a = 1: d1 + d2;  a = 2: d2 - d1;  a = 3: d1 - d3    (and the same on the columns)
      const int ta = zb >> 2, tb = zb & 3;
      const int r0 = ta == 0 ? 0 : (ta == 2 ? 2 : 1), r1 = ta == 0 ? 2 : (ta == 1 ? 2 : (ta == 2 ? 1 : 3));
      const int c0 = tb == 0 ? 0 : (tb == 2 ? 2 : 1), c1 = tb == 0 ? 2 : (tb == 1 ? 2 : (tb == 2 ? 1 : 3));
      const float sr = ta == 1 ? 1.f : -1.f, sc = tb == 1 ? 1.f : -1.f;          // sign of the second row / column
      sgn[0] = sc; sgn[1] = sr; sgn[2] = sr * sc;                                 // elements (r0,c1), (r1,c0), (r1,c1)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = m0 + s_row + 32 * i;
        const bool rok = t < p.rows;
        const uint32_t n = fd_div((uint32_t)(rok ? t : 0), p.fd_thtw);
        const uint32_t rem = (uint32_t)(rok ? t : 0) - n * (uint32_t)(p.th * p.tw);
        const uint32_t ty = fd_div(rem, p.fd_tw);
        const int tx = (int)(rem - ty * (uint32_t)p.tw);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ih = 2 * (int)ty - p.off + ((e >> 1) ? r1 : r0);
          int iw = 2 * tx - p.off + ((e & 1) ? c1 : c0);
          bool ok = rok;
          if (p.reflect) { ih = reflect_idx(ih, p.Hl); iw = reflect_idx(iw, p.Wl); }
          else ok = ok && ih >= 0 && ih < p.Hl && iw >= 0 && iw < p.Wl;
          // pixel (ih * ups, iw * ups) of image n; the unshuffle phase (pi, pj) and the channel are added per K-step
          aoff[i * 4 + e] = ok ? (uint32_t)((((size_t)n * p.H + (size_t)ih * p.ups) * p.W + (size_t)iw * p.ups) * p.Cin * 4 + s_u * 16) : GW_OOB;
        }
      }
    }
    float4 va[MODE == 1 ? 16 : 4];
    u32x4w vb[6], vap[6];
    (void)vap;
    auto load_tiles = [&](int kt) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) va[i] = gw_bload4(ra, aoff[i] != GW_OOB ? aoff[i] + (uint32_t)kt * 128u : GW_OOB);
      } else if constexpr (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
          vap[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(aoff[j / 3] != GW_OOB ? aoff[j / 3] + (uint32_t)(j % 3) * 64u + (uint32_t)kt * 192u : GW_OOB), 0, 0);
      } else {
        // k = 32 kt + 4 s_u + {0..3} = (phase, channel): phase = k / Cin (Cin % 32 == 0: one phase per K-step)
        const int k0 = kt * 32;
        const int ph = k0 / p.Cin, cb = k0 - ph * p.Cin;
        const uint32_t koff = (uint32_t)((((size_t)(ph >> 1) * p.W + (ph & 1)) * p.Cin + cb) * 4);
#pragma unroll
        for (int i = 0; i < 16; ++i) va[i] = gw_bload4(ra, aoff[i] != GW_OOB ? aoff[i] + koff : GW_OOB);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
        vb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(boff0 + (uint32_t)(j / 3) * bhalf + (uint32_t)(j % 3) * 64u + (uint32_t)kt * 192u), 0, 0);
    };
    auto store_tiles = [&](int stage) {
      unsigned char* As = smem + stage * GW_STAGE;
      unsigned char* Bs = As + 3 * 8192;
      if constexpr (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4w*>(As + (j % 3) * 8192 + bsoff0 + 4096 * (j / 3)) = vap[j];
      } else
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v;
        if constexpr (MODE == 0) v = va[i];
        else {
          const float4 d0 = va[i * 4], d1 = va[i * 4 + 1], d2 = va[i * 4 + 2], d3 = va[i * 4 + 3];
          // ((d00 + s0 d01) + s1 d10) + s2 d11: the order k_wino_in's two passes produce is (d00 + sr d10) + sc (d01 + sr d11);
          // keep THAT association so the fused path rounds exactly like the materialised V
          v.x = (d0.x + sgn[1] * d2.x) + sgn[0] * (d1.x + sgn[1] * d3.x);
          v.y = (d0.y + sgn[1] * d2.y) + sgn[0] * (d1.y + sgn[1] * d3.y);
          v.z = (d0.z + sgn[1] * d2.z) + sgn[0] * (d1.z + sgn[1] * d3.z);
          v.w = (d0.w + sgn[1] * d2.w) + sgn[0] * (d1.w + sgn[1] * d3.w);
        }
        uint2 h, m, l;
        split4(v, h, m, l);
        *reinterpret_cast<uint2*>(As + soff[i]) = h;
        *reinterpret_cast<uint2*>(As + 8192 + soff[i]) = m;
        *reinterpret_cast<uint2*>(As + 16384 + soff[i]) = l;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4w*>(Bs + (j % 3) * 8192 + bsoff0 + 4096 * (j / 3)) = vb[j];
    };
    load_tiles(0);
    store_tiles(0);
    if (nkt > 1) load_tiles(1);
    gw_barrier();                                                 // stage 0 is ready
#ifdef VCG_WS_STAMP
    st_t = WS_T();
#endif
    for (int kt = 0; kt < nkt; ++kt) {
#ifdef VCG_WS_STAMP
      __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0): the loads' exposed latency, by itself
      WS_ACC(0, st_t);
#endif
      if (kt + 1 < nkt) store_tiles((kt + 1) & 1);                // the registers hold K-step kt + 1
#ifdef VCG_WS_STAMP
      __builtin_amdgcn_s_waitcnt(0xC07F);
      WS_ACC(1, st_t);
#endif
      if (kt + 2 < nkt) load_tiles(kt + 2);                       // in flight across the barrier
      WS_ACC(2, st_t);
      gw_barrier();                                               // stage kt + 1 is ready; the consumers are done reading stage kt
      WS_ACC(3, st_t);
    }
#ifdef VCG_WS_STAMP
    st_flush();
#endif
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int cw = (tid >> 6) - 4, lane = tid & 63;
  const int wm = cw >> 1, wn = cw & 1, l31 = lane & 31, lh = lane >> 5;
  if constexpr (SHAPE == 16) {
    // v_mfma_f32_16x16x32_bf16: the wave's 64 x 64 as 4 x 4 tiles of 16 x 16; one MFMA spans the whole 32-deep K-step.
    // Fragments: row blocks {0, 1} (A01) and {2, 3} (A23) of A, all four column blocks of B; rows 0-1 are multiplied while
    // A23 lands, rows 2-3 while the next stage's A01 lands; B of the next stage is read after the last MFMA that uses B.
    const int l15 = lane & 15, lq = lane >> 4;
    uint32_t fa16[4], fb16[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ra_ = wm * 64 + i * 16 + l15, rb_ = wn * 64 + i * 16 + l15;
      fa16[i] = (uint32_t)(ra_ * 64 + ((lq ^ gw_swz<16>(ra_)) << 4));
      fb16[i] = (uint32_t)(3 * 8192 + rb_ * 64 + ((lq ^ gw_swz<16>(rb_)) << 4));
    }
    f32x4w acc16[4][4], lo16[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[i][j][e] = lo16[i][j][e] = 0.f;
    bf16x8 A01[3][2], A23[3][2], Bf[3][4];
    auto read_a = [&](int stage, int half, bf16x8 (&a)[3][2]) {
      const unsigned char* S = smem + stage * GW_STAGE;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int i = 0; i < 2; ++i) a[pc][i] = *reinterpret_cast<const bf16x8*>(S + pc * 8192 + fa16[2 * half + i]);
    };
    auto read_b = [&](int stage) {
      const unsigned char* S = smem + stage * GW_STAGE;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int j = 0; j < 4; ++j) Bf[pc][j] = *reinterpret_cast<const bf16x8*>(S + pc * 8192 + fb16[j]);
    };
    auto mma_half = [&](int half, const bf16x8 (&a)[3][2]) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4w c = lo16[2 * half + i][j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], Bf[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][i], Bf[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], Bf[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], Bf[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], Bf[1][j], c, 0, 0, 0);
          lo16[2 * half + i][j] = c;
          acc16[2 * half + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], Bf[0][j], acc16[2 * half + i][j], 0, 0, 0);
        }
    };
    gw_barrier();                                                 // stage 0 is ready
    read_b(0);
    read_a(0, 0, A01);
#ifdef VCG_WS_STAMP
    st_t = WS_T();
#endif
    for (int kt = 0; kt < nkt; ++kt) {
      read_a(kt & 1, 1, A23);
      mma_half(0, A01);
      WS_ACC(0, st_t);
      __builtin_amdgcn_sched_barrier(0);
      gw_barrier();                                               // every fragment of stage kt is in registers; stage kt + 1 is ready
      __builtin_amdgcn_sched_barrier(0);
      WS_ACC(1, st_t);
      if (kt + 1 < nkt) read_a((kt + 1) & 1, 0, A01);
      mma_half(1, A23);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nkt) read_b((kt + 1) & 1);
      WS_ACC(2, st_t);
    }
    float* const dst16 = p.c + (size_t)zb * p.c_bstride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * 64 + i * 16 + lq * 4 + e;
          const int n = n0 + wn * 64 + j * 16 + l15;
          if (m < p.rows) dst16[(size_t)m * p.N + n] = acc16[i][j][e] + lo16[i][j][e];
        }
#ifdef VCG_WS_STAMP
    WS_ACC(3, st_t);
    st_flush();
#endif
    return;
  }
  uint32_t fa[2], fb[2];
  int sa[2], sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int r = wm * 64 + i * 32 + l31; fa[i] = (uint32_t)(r * 64); sa[i] = (r >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = wn * 64 + j * 32 + l31; fb[j] = (uint32_t)(3 * 8192 + r * 64); sb[j] = (r >> 2) & 3; }
  f32x16 acc[2][2], lo[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  bf16x8 a0[3][2], b0[3][2], a1[3][2], b1[3][2];                  // fragments of slice 0 / slice 1 of a K-step
  auto read_frags = [&](int stage, int s, bf16x8 (&a)[3][2], bf16x8 (&b)[3][2]) {
    const unsigned char* S = smem + stage * GW_STAGE;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[pc][i] = *reinterpret_cast<const bf16x8*>(S + pc * 8192 + fa[i] + (((2 * s + lh) ^ sa[i]) << 4));
#pragma unroll
      for (int j = 0; j < 2; ++j) b[pc][j] = *reinterpret_cast<const bf16x8*>(S + pc * 8192 + fb[j] + (((2 * s + lh) ^ sb[j]) << 4));
    }
  };
  auto mma = [&](const bf16x8 (&a)[3][2], const bf16x8 (&b)[3][2]) {
    // smallest contributions first: mm, lh, hl, mh, hm into the cross-term chain, hh into its own
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = lo[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
        lo[i][j] = c;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
      }
  };
  gw_barrier();                                                   // stage 0 is ready
  read_frags(0, 0, a0, b0);
#ifdef VCG_WS_STAMP
  st_t = WS_T();
#endif
  for (int kt = 0; kt < nkt; ++kt) {
    read_frags(kt & 1, 1, a1, b1);                                // slice 1's fragments land under slice 0's MFMAs
    mma(a0, b0);
    WS_ACC(0, st_t);
    // the scheduler may not move slice 0's MFMAs below the barrier (it did: the matrix pipe then idles while the wave waits
    // for its 12 reads and for the other waves) nor the next stage's reads above it
    __builtin_amdgcn_sched_barrier(0);
    gw_barrier();                                                 // all fragments of stage kt are in registers; stage kt + 1 is ready
    __builtin_amdgcn_sched_barrier(0);
    WS_ACC(1, st_t);
    if (kt + 1 < nkt) read_frags((kt + 1) & 1, 0, a0, b0);        // the next stage's first fragments land under slice 1's MFMAs
    mma(a1, b1);
    WS_ACC(2, st_t);
  }
  float* const dst = p.c + (size_t)zb * p.c_bstride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int m = m0 + wm * 64 + i * 32 + row;
        if (m < p.rows) dst[(size_t)m * p.N + n] = acc[i][j][e] + lo[i][j][e];
      }
  }
#ifdef VCG_WS_STAMP
  WS_ACC(3, st_t);
  st_flush();
#endif
}

// VCG_GEMM_WS=1 routes the Winograd GEMMs through this kernel.  OFF by default: measured on the step's shapes
// (tools/gemm_ws_probe.hip, profiles/r03_gemm_ws_probe.txt) it lands within +-5 % of the 4-wave kernel in every variant —
// fp32 A, pre-split A, 16x16x32 MFMAs — although its consumers keep the matrix pipe 66 % busy in shader clocks (in-kernel
// stamps) where the 4-wave kernel has 44 %: the chip answers a denser MFMA stream with a lower clock (1.6 GHz against
// 2.2 GHz; cdna_hip_programming.md §5.4 rule 28).  What moved the kernel was fewer MFMAs per product (fp16x2).
bool vcg_gemm_ws_enabled() {
  static const int on = [] { const char* e = getenv("VCG_GEMM_WS"); return e ? atoi(e) : 0; }();
  return on != 0;
}
bool vcg_gemm_ws_ok(int rows, int K, int N) { return vcg_gemm_ws_enabled() && K % 32 == 0 && K >= 64 && N % 128 == 0 && rows > 0; }

// rows x K (fp32) times (N x K)^T (blocked planes) per batch on the wave-specialised kernel; N % 128 == 0
int vcg_gemm_ws_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st) {
  VCG_CHECK_ARG(K % 32 == 0 && N % 128 == 0 && rows > 0, "vcg_gemm_ws_batched: bad shape rows=%d K=%d N=%d", rows, K, N);
  VCG_CHECK_ARG((unsigned long long)rows * K * 4 < (1ull << 31) && (unsigned long long)N * K * 6 < (1ull << 31),
                "vcg_gemm_ws_batched: operand extents must stay below 2 GiB per batch");
  VCG_CHECK_ARG((unsigned long long)rows * K * (unsigned long long)batches < (1ull << 32) &&
                    (unsigned long long)N * K * 3 * (unsigned long long)batches < (1ull << 32),
                "vcg_gemm_ws_batched: batch stride overflow");
  GemmWsP p = {};
  p.a = A; p.bt = BtPlanes; p.c = C; p.rows = rows; p.K = K; p.N = N;
  p.a_bytes = (uint32_t)((size_t)rows * K * 4); p.b_bytes = (uint32_t)((size_t)N * K * 6);
  p.a_bstride = (uint32_t)((size_t)rows * K); p.b_bstride = (uint32_t)((size_t)N * K * 3);
  p.c_bstride = (size_t)rows * N;
  dim3 grid((rows + 127) / 128, N / 128, batches);
  static const int shape = [] { const char* e = getenv("VCG_WS_SHAPE"); return e ? atoi(e) : 32; }();
  VcgProfScope prof(shape == 16 ? "k_gemm_ws<0, 16>" : "k_gemm_ws<0>", 2.0 * rows * (double)K * N * batches, st);
  if (shape == 16) hipLaunchKernelGGL((k_gemm_ws<0, 16>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_ws<0, 32>), grid, dim3(512), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_ws_batched");
  return 0;
}
// the same on v_mfma_f32_16x16x32_bf16 (probe: which MFMA shape the chip clocks higher on, cdna_hip_programming.md §5.4 rule 28)
int vcg_gemm_ws16_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st) {
  GemmWsP p = {};
  p.a = A; p.bt = BtPlanes; p.c = C; p.rows = rows; p.K = K; p.N = N;
  p.a_bytes = (uint32_t)((size_t)rows * K * 4); p.b_bytes = (uint32_t)((size_t)N * K * 6);
  p.a_bstride = (uint32_t)((size_t)rows * K); p.b_bstride = (uint32_t)((size_t)N * K * 3);
  p.c_bstride = (size_t)rows * N;
  dim3 grid((rows + 127) / 128, N / 128, batches);
  hipLaunchKernelGGL((k_gemm_ws<0, 16>), grid, dim3(512), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_ws16_batched");
  return 0;
}

// both operands pre-split (A planes [z][rows][K/32][3][32] written by A's producer): the producer waves only copy
int vcg_gemm_ws_planes_batched(const void* APlanes, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st) {
  VCG_CHECK_ARG(K % 32 == 0 && N % 128 == 0 && rows > 0, "vcg_gemm_ws_planes_batched: bad shape rows=%d K=%d N=%d", rows, K, N);
  VCG_CHECK_ARG((unsigned long long)rows * K * 6 < (1ull << 31) && (unsigned long long)N * K * 6 < (1ull << 31),
                "vcg_gemm_ws_planes_batched: operand extents must stay below 2 GiB per batch");
  VCG_CHECK_ARG((unsigned long long)rows * K * 3 * (unsigned long long)batches < (1ull << 32) &&
                    (unsigned long long)N * K * 3 * (unsigned long long)batches < (1ull << 32),
                "vcg_gemm_ws_planes_batched: batch stride overflow");
  GemmWsP p = {};
  p.a = (const float*)APlanes; p.bt = BtPlanes; p.c = C; p.rows = rows; p.K = K; p.N = N;
  p.a_bytes = (uint32_t)((size_t)rows * K * 6); p.b_bytes = (uint32_t)((size_t)N * K * 6);
  p.a_bstride = (uint32_t)((size_t)rows * K * 3); p.b_bstride = (uint32_t)((size_t)N * K * 3);
  p.c_bstride = (size_t)rows * N;
  dim3 grid((rows + 127) / 128, N / 128, batches);
  VcgProfScope prof("k_gemm_ws<planes>", 2.0 * rows * (double)K * N * batches, st);
  hipLaunchKernelGGL((k_gemm_ws<2, 32>), grid, dim3(512), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_ws_planes_batched");
  return 0;
}

// The forward Winograd GEMMs with the input transform inside: M[xi] = (B^T x B)[xi] . U[xi]^T for all 16 xi, from the image.
// geometry: as conv_wino.hip's WinoP (T tiles of th x tw per image, patch origin 2 * tile - off)
int vcg_gemm_ws_wino(const float* x, const void* UPlanes, float* M, int Nimg, int H, int W, int Cin, int Hl, int Wl, int ups,
                     int reflect, int off, int th, int tw, int Ncols, hipStream_t st) {
  const int T = Nimg * th * tw, Kc = ups * ups * Cin;
  VCG_CHECK_ARG(Cin % 32 == 0 && Ncols % 128 == 0 && T > 0, "vcg_gemm_ws_wino: bad shape T=%d Cin=%d N=%d", T, Cin, Ncols);
  VCG_CHECK_ARG((unsigned long long)Nimg * H * W * Cin * 4 < (1ull << 31) && (unsigned long long)Ncols * Kc * 6 < (1ull << 31),
                "vcg_gemm_ws_wino: operand extents must stay below 2 GiB");
  VCG_CHECK_ARG((unsigned long long)Ncols * Kc * 3 * 16ull < (1ull << 32), "vcg_gemm_ws_wino: batch stride overflow");
  GemmWsP p = {};
  p.a = x; p.bt = UPlanes; p.c = M; p.rows = T; p.K = Kc; p.N = Ncols;
  p.a_bytes = (uint32_t)((size_t)Nimg * H * W * Cin * 4); p.b_bytes = (uint32_t)((size_t)Ncols * Kc * 6);
  p.a_bstride = 0; p.b_bstride = (uint32_t)((size_t)Ncols * Kc * 3);
  p.c_bstride = (size_t)T * Ncols;
  p.H = H; p.W = W; p.Cin = Cin; p.Hl = Hl; p.Wl = Wl; p.ups = ups; p.reflect = reflect; p.off = off; p.th = th; p.tw = tw;
  p.fd_tw = make_fastdiv((uint32_t)tw); p.fd_thtw = make_fastdiv((uint32_t)(th * tw));
  dim3 grid((T + 127) / 128, Ncols / 128, 16);
  VcgProfScope prof("k_gemm_ws<wino>", 2.0 * T * (double)Kc * Ncols * 16, st);
  hipLaunchKernelGGL((k_gemm_ws<1, 32>), grid, dim3(512), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_ws_wino");
  return 0;
}
