// Batched dense GEMM for the Winograd layers on the 16-bit matrix pipe with fp32-level accuracy ("split-operand" arithmetic):
//     C[z][m][n] = sum_k A[z][m][k] * Bt[z][n][k]      (A, C fp32 in memory; Bt pre-split planes; k contiguous in both)
//
// gfx950 runs v_mfma_f32_32x32x16_{f16,bf16} at 16x the rate of v_mfma_f32_32x32x2_f32.  Rounds 1-2 split every fp32 operand
// into three bf16 pieces and accumulated six products; measured from inside (tools/gemm_ws_probe.hip, round 3) those kernels
// were not held back by their schedule but by the chip's answer to a dense MFMA stream — a lower clock — so the lever is
// MFMAs per product.  Round 3: x / s = h + l as two fp16 pieces (22 mantissa bits), s a power of two from the tensor's largest
// magnitude (vcg_common.h), and the THREE products hh, hl, lh — h*h in one fp32 accumulation chain, the two cross terms
// (<= 2^-11 of it) in a second one, summed and multiplied by sA * sB in the epilogue.  Dropped: l*l, below 2^-22.
// Against float64 the GEMM rounds at 2.1e-7 (bf16 x 3: 1.7e-7; PyTorch-CPU fp32 on the same data 2.2-3.0e-7) and runs
// 1.3-1.6x faster on the step's shapes (profiles/r03_gemm_fp16x2_probe.txt).
//
// The B operand of every call site is a WEIGHT (the Winograd-transformed kernels U / Ud): it is split once per optimizer
// step, when it is packed ("blocked planes": for Bt[n][k], K % 32 == 0,  bp[(n * K/32 + kb) * 64 + piece * 32 + j]  as fp16,
// piece 0 / 1 = h / l of Bt[n][32 kb + j] / sB — 128 contiguous bytes per row and K block), so this kernel stages B with plain
// 16-byte copies and only the activations (A) are scaled and split in the K loop.
//
// Tile 128 x BN (BN = 128 or 64), BK = 32, 256 threads = 2 x 2 waves, each wave (64 x BN/2) as 32x32 accumulators.
// LDS images: per piece [rows][32] fp16, 64-byte rows, the four 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3
// so that the ds_read_b128 fragment reads (lane = row, 16 consecutive rows per LDS cycle) are conflict-free.
#include "vcg_common.h"
#include <stdlib.h>

typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));

struct GemmSplitP {
  const float* a;
  const float* bt;
  float* c;
  int rows, K, N;
  uint32_t a_bytes, b_bytes;          // per batch (buffer-load bounds)
  uint32_t a_bstride, b_bstride;      // floats (a) / fp16 elements (bt planes: VCG_NP per value) between batches
  size_t c_bstride;
  VcgAmax amax_a, amax_b;             // largest magnitudes of A (all batches) and of the tensor Bt was split from
  uint32_t* amax_a_keep;              // where to leave A's amax bits for a later call that reads the same A (the kept V), or null
};

__device__ __forceinline__ float4 gs_bload4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  u32x4g v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#define GS_OOB 0x80000000u

// APL: A comes pre-split too (k_wino_in_planes: fp16 planes [rows][K / 32][2][32] of A / sA) — both operands are staged with
// plain 16-byte copies and the K loop has no conversion arithmetic at all.
template <int BN, bool APL = false>
__global__ __launch_bounds__(256, 2) void k_gemm_split(GemmSplitP p) {
  constexpr int BM = 128, NI = BN / 64, MI = 2, AR = BM / 32;
  // [piece][row][32 fp16] as raw bytes: 64 B per row
  __shared__ __attribute__((aligned(16))) unsigned char As[VCG_NP][BM * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[VCG_NP][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
  float sa_, inv_a, sb_, inv_b;
  {
    const uint32_t abits = vcg_amax_bits(p.amax_a);
    vcg_scale_of(abits, p.amax_a.shift, sa_, inv_a);
    vcg_scale_of(vcg_amax_bits(p.amax_b), p.amax_b.shift, sb_, inv_b);
    if (p.amax_a_keep && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) *p.amax_a_keep = abits;
  }
  // XCD-aware tile order (see k_conv_fwd): the N tiles that share an A tile run back to back on one XCD
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
  const int m0 = mt * BM, n0 = nt * BN;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                                   APL ? (void*)((const unsigned short*)p.a + (size_t)zb * p.a_bstride) : (void*)(p.a + (size_t)zb * p.a_bstride), 0,
                                   (int)p.a_bytes, 0x00020000),
                               rb = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)p.bt + (size_t)zb * p.b_bstride), 0, (int)p.b_bytes, 0x00020000);
  const int s_row = tid >> 3, s_u = tid & 7;                    // staging: row (+32 i), k quad
  uint32_t aoff[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = m0 + s_row + 32 * i;
    aoff[i] = r < p.rows ? (uint32_t)(((size_t)r * p.K + s_u * 4) * 4) : GS_OOB;
  }
  // B planes: thread (row b_r = tid >> 2, 16-byte chunk b_q = tid & 3 of a 64-byte piece row); pass j = (row half, piece)
  constexpr int BP = VCG_NP * BN / 64;                            // 16-byte copies per thread and K-step
  const int b_q = tid & 3, b_r = tid >> 2;
  const int KB = p.K / 32;
  const uint32_t boff0 = (uint32_t)(((size_t)(n0 + b_r) * KB) * VCG_PBYTES + b_q * 16);      // N % BN == 0: every row is in range
  const uint32_t bhalf = (uint32_t)KB * (64u * VCG_PBYTES);
  const uint32_t bsoff0 = (uint32_t)(b_r * 64 + ((b_q ^ ((b_r >> 2) & 3)) << 4));     // row + 64 keeps the swizzle term
  // LDS byte offset of this thread's quad inside a piece image: row r, chunk (u >> 1) swizzled, half (u & 1)
  uint32_t soff[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = s_row + 32 * i;
    soff[i] = (uint32_t)(r * 64 + (((s_u >> 1) ^ ((r >> 2) & 3)) << 4) + ((s_u & 1) << 3));
  }

  // acc: the h*h chain; lo: the two cross terms (<= 2^-11 of it).  One chain for all three would round the big running
  // sum three times per slice instead of once.
  f32x16 acc[MI][NI], lo[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  float4 va[AR];
  u32x4g vb[BP], vap[2 * VCG_NP];
  // A planes: the B pattern on the 128 rows of the M tile (rows past the matrix read as zeros)
  uint32_t apoff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = m0 + b_r + 64 * h;
    apoff[h] = r < p.rows ? (uint32_t)(((size_t)r * KB) * VCG_PBYTES + b_q * 16) : GS_OOB;
  }
  (void)vap; (void)apoff;
  const int nkt = KB;                                           // K % 32 == 0 (the planes' block size)
  auto load_tiles = [&](int kt) {
    if constexpr (APL) {
#pragma unroll
      for (int j = 0; j < 2 * VCG_NP; ++j)
        vap[j] = __builtin_amdgcn_raw_buffer_load_b128(
            ra, (int)(apoff[j / VCG_NP] != GS_OOB ? apoff[j / VCG_NP] + (uint32_t)(j % VCG_NP) * 64u + (uint32_t)kt * VCG_PBYTES : GS_OOB), 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < AR; ++i) va[i] = gs_bload4(ra, aoff[i] != GS_OOB ? aoff[i] + (uint32_t)kt * 128u : GS_OOB);
    }
#pragma unroll
    for (int j = 0; j < BP; ++j)
      vb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(boff0 + (uint32_t)(j / VCG_NP) * bhalf + (uint32_t)(j % VCG_NP) * 64u + (uint32_t)kt * VCG_PBYTES), 0, 0);
  };
  auto store_tiles = [&]() {
    if constexpr (APL) {
#pragma unroll
      for (int j = 0; j < 2 * VCG_NP; ++j) *reinterpret_cast<u32x4g*>(&As[j % VCG_NP][bsoff0 + 4096 * (j / VCG_NP)]) = vap[j];
    } else {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        uint2 h, l;
        split4h(va[i], inv_a, h, l);
        *reinterpret_cast<uint2*>(&As[0][soff[i]]) = h;
        *reinterpret_cast<uint2*>(&As[1][soff[i]]) = l;
      }
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) *reinterpret_cast<u32x4g*>(&Bs[j % VCG_NP][bsoff0 + 4096 * (j / VCG_NP)]) = vb[j];
  };
  // fragment byte offsets (per k slice s: chunk 2s + lh)
  uint32_t fa[MI], fb[NI];
  int sa[MI], sb[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { const int r = wm * 64 + i * 32 + l31; fa[i] = (uint32_t)(r * 64); sa[i] = (r >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < NI; ++j) { const int r = wn * (BN / 2) + j * 32 + l31; fb[j] = (uint32_t)(r * 64); sb[j] = (r >> 2) & 3; }

  load_tiles(0);
  store_tiles();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_tiles(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 a[VCG_NP][MI], b[VCG_NP][NI];
#pragma unroll
      for (int pc = 0; pc < VCG_NP; ++pc) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          a[pc][i] = *reinterpret_cast<const f16x8*>(&As[pc][fa[i] + (((2 * s + lh) ^ sa[i]) << 4)]);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          b[pc][j] = *reinterpret_cast<const f16x8*>(&Bs[pc][fb[j] + (((2 * s + lh) ^ sb[j]) << 4)]);
      }
      // the cross terms lh, hl in one chain, hh in the other
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
    __syncthreads();
    if (kt + 1 < nkt) {
      store_tiles();
      __syncthreads();
    }
  }
  float* const dst = p.c + (size_t)zb * p.c_bstride;
  const float os = sa_ * sb_;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn * (BN / 2) + j * 32 + l31;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int m = m0 + wm * 64 + i * 32 + row;
        if (m < p.rows) dst[(size_t)m * p.N + n] = (acc[i][j][e] + lo[i][j][e]) * os;
      }
  }
}

// ---- round 3: the same GEMM with LDS-DMA staging (both operands pre-split planes) ------------------------------------------------
// k_gemm_split<.., planes> moves 32 KB per 128 x 128 x 32 tile step through registers (8 buffer loads + 8 ds_write_b128 per
// thread and K-step, two barriers) and sits at ~7 TB/s of L2 -> LDS traffic with its matrix pipe a third busy.  Here ONE 8-wave
// workgroup per CU owns a 256 x 128 tile (a quarter fewer staged bytes per product), the planes go global -> LDS directly
// (buffer_load_dwordx4 ... lds: 1 KB per wave-instruction, no VGPRs, no ds_write; the XOR swizzle of the LDS image is applied on
// the SOURCE address, the destination of an LDS-DMA being lane-linear), into a ring of three stages with the loads two K-steps
// ahead, ONE raw barrier per K-step: at step kt a wave waits for its own DMAs of stage kt (counted vmcnt, the next stage's stay
// in flight), the barrier publishes everybody's, and the stage read at step kt - 1 is handed to the DMAs of step kt + 2.
// The DMA is inline asm (through the builtin hipcc drains vmcnt(0) in front of every fragment read: tools/mfma_probe.hip).
struct GdSrd { uint32_t x, y, z, w; };
typedef uint32_t gd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gd_u32x4 gd_srd(const void* ptr, uint32_t bytes) {
  const uint64_t a = (uint64_t)ptr;
  gd_u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
// 16 bytes per lane, global (buffer offset voff; out of range: zeros) -> LDS at lds_dst + 16 * lane (lds_dst wave-uniform)
__device__ __forceinline__ void gd_dma16(gd_u32x4 srd, uint32_t lds_dst, uint32_t voff) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(srd) : "memory");
}
// raw workgroup barrier: this wave's LDS reads have returned; outstanding DMAs stay in flight (a __syncthreads() would drain them)
__device__ __forceinline__ void gd_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
constexpr int GD_BM = 256, GD_BN = 128;
constexpr int GD_A_BYTES = GD_BM * VCG_PBYTES, GD_B_BYTES = GD_BN * VCG_PBYTES;     // one stage: [row][128 B] images, 32 KB + 16 KB
constexpr int GD_STAGE = GD_A_BYTES + GD_B_BYTES;
constexpr int GD_STAGES = 3;
// LDS image of an operand: 128 bytes per row — the row's K block exactly as it lies in memory, piece h (4 x 16 B) then piece l —
// with the eight 16-byte slots XOR-swizzled by (row >> 1) & 7: a ds_read_b128 fragment read (lane = row, one logical slot) then
// touches 16 different bank groups per 16-lane group, and one DMA wave-instruction moves 8 rows x 128 B = eight FULL cache lines
// (the [piece][row][64 B] images of k_gemm_split fetch every line twice, half a line per instruction).
// Diagnostic build only (tools/gemm_dma_probe.hip, -DVCG_GD_STAMP; libvcg.so never has it): per-wave shader-clock totals of the
// K loop's phases, into a buffer of their own ([workgroup][wave][8] u64)
#ifdef VCG_GD_STAMP
__device__ unsigned long long* g_gd_stamp = nullptr;
int vcg_gd_set_stamp(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_gd_stamp), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#define GD_T() (__builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_memtime())
#define GD_ACC(slot, t0) do { const unsigned long long t1__ = GD_T(); gd_acc[slot] += t1__ - (t0); (t0) = t1__; } while (0)
#else
#define GD_T() 0ull
#define GD_ACC(slot, t0) do { } while (0)
#endif
// SHAPE 32: v_mfma_f32_32x32x16_f16, a wave's 64 x 64 as 2 x 2 tiles x two 16-deep slices; SHAPE 16: v_mfma_f32_16x16x32_f16, 4 x 4
// tiles x one 32-deep slice — the same fragment bytes, LDS reads, accumulator registers and matrix-pipe cycles, but the chip,
// which answers an MFMA-dense loop with a lower clock (1.15-1.76 GHz here: profiles/r03_gemm_dma_stamps.txt), holds a higher one on
// the 16 x 16 shape (MI355X_MICROARCH.md, DVFS item 7)
// STAG (SHAPE 16 only; round 4): the two waves that share a SIMD (wave w and w + 4 of the workgroup) run half a K-step apart.
// Measured from the instruction stream (profiles/r04_power_probe.txt has the clocks; the ISA: all 16 ds_read_b128 of a step are
// issued right behind the barrier, by all 8 waves at once): a K-step was an LDS phase — 8 waves x 16 KB of fragments + 48 KB of
// DMA writes at 128 B / clock: ~1 400 clocks with the matrix pipe idle — followed by a matrix phase (2 x 768 clocks per SIMD)
// with the LDS idle.  With STAG the waves 4..7 keep the fragments of step kt in registers across the barrier and multiply them
// in the FIRST half of interval kt + 1, while the waves 0..3 read theirs; in the second half the roles swap.  Same barriers, same
// ring (stage kt is read in interval kt by both halves; the DMAs of interval kt overwrite the stage read in interval kt - 1), same
// registers, same order of accumulation in every wave: bit-identical results.
template <bool GD_SPREAD, int SHAPE, bool STAG = false>
__global__ __launch_bounds__(512, 1) void k_gemm_planes_dma(GemmSplitP p) {
  constexpr int MI = 2, NI = 2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[GD_STAGES * GD_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
  float sa_, inv_a, sb_, inv_b;
  {
    const uint32_t abits = vcg_amax_bits(p.amax_a);
    vcg_scale_of(abits, p.amax_a.shift, sa_, inv_a);
    vcg_scale_of(vcg_amax_bits(p.amax_b), p.amax_b.shift, sb_, inv_b);
    if (p.amax_a_keep && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) *p.amax_a_keep = abits;
  }
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
  const int m0 = mt * GD_BM, n0 = nt * GD_BN;
  const gd_u32x4 ra = gd_srd((const unsigned short*)p.a + (size_t)zb * p.a_bstride, p.a_bytes),
                 rb = gd_srd((const unsigned short*)p.bt + (size_t)zb * p.b_bstride, p.b_bytes);
  const int KB = p.K / 32;
  // one DMA wave-instruction = 8 rows x 128 B: lane -> (row l >> 3, physical slot l & 7), whose content is the logical slot
  // (l & 7) ^ ((row >> 1) & 7) = byte 16 * that of the row's K block.  Wave w stages row groups w, w + 8, w + 16, w + 24 of A and
  // w, w + 8 of B: six instructions per K-step.
  const int drow = lane >> 3, dslot = lane & 7;
  uint32_t aoff[4], boff[2];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int rl = (wid + 8 * h) * 8 + drow;                 // row of the tile
    const int r = m0 + rl;
    aoff[h] = r < p.rows ? (uint32_t)((size_t)r * KB * VCG_PBYTES + ((dslot ^ ((rl >> 1) & 7)) << 4)) : GS_OOB;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int rl = (wid + 8 * h) * 8 + drow;
    boff[h] = (uint32_t)((size_t)(n0 + rl) * KB * VCG_PBYTES + ((dslot ^ ((rl >> 1) & 7)) << 4));      // N % 128 == 0: in range
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // DMA number q (0..5) of a stage: four A pieces, two B pieces
  auto issue_one = [&](int stage, int kt, int q) {
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)stage * GD_STAGE);
    const uint32_t ko = (uint32_t)kt * VCG_PBYTES;
    if (q < 4) gd_dma16(ra, __builtin_amdgcn_readfirstlane(sbase + (uint32_t)((wid + 8 * q) * 1024)), aoff[q] != GS_OOB ? aoff[q] + ko : GS_OOB);
    else gd_dma16(rb, __builtin_amdgcn_readfirstlane(sbase + (uint32_t)(GD_A_BYTES + (wid + 8 * (q - 4)) * 1024)), boff[q - 4] + ko);
  };
  auto issue = [&](int stage, int kt) {
#pragma unroll
    for (int q = 0; q < 6; ++q) issue_one(stage, kt, q);
  };

  f32x16 acc[MI][NI], lo[MI][NI];
  f32x4 acc16[4][4], lo16[4][4];
  if constexpr (SHAPE == 32) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[i][j][e] = lo16[i][j][e] = 0.f;
  }
  uint32_t fa[MI], fb[NI];
  int sa[MI], sb[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { const int r = wm * 64 + i * 32 + l31; fa[i] = (uint32_t)(r * VCG_PBYTES); sa[i] = (r >> 1) & 7; }
#pragma unroll
  for (int j = 0; j < NI; ++j) { const int r = wn * 64 + j * 32 + l31; fb[j] = (uint32_t)(GD_A_BYTES + r * VCG_PBYTES); sb[j] = (r >> 1) & 7; }
  // SHAPE 16: lane -> (row lane & 15 of a 16-row tile, 8-deep k slot lane >> 4)
  const int r16 = lane & 15, kq = lane >> 4;
  uint32_t fa16[4], fb16[4];
  int sa16[4], sb16[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + r16; fa16[i] = (uint32_t)(r * VCG_PBYTES); sa16[i] = (r >> 1) & 7; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int r = wn * 64 + j * 16 + r16; fb16[j] = (uint32_t)(GD_A_BYTES + r * VCG_PBYTES); sb16[j] = (r >> 1) & 7; }

  const int nkt = KB;
#ifdef VCG_GD_STAMP
  unsigned long long gd_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long gd_start = GD_T(), gd_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  issue(0, 0);
  if (nkt > 1) issue(1, 1);
  unsigned long long gd_t = GD_T();
  (void)gd_t;
  // SHAPE 16 fragments live across the loop (STAG: the late half multiplies step kt - 1 while it is in iteration kt)
  f16x8 a16[VCG_NP][4], b16[VCG_NP][4];
  auto read16 = [&](int kt) {
    const unsigned char* st = smem + (kt % GD_STAGES) * GD_STAGE;
#pragma unroll
    for (int pc = 0; pc < VCG_NP; ++pc) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a16[pc][i] = *reinterpret_cast<const f16x8*>(st + fa16[i] + (((pc * 4 + kq) ^ sa16[i]) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        b16[pc][j] = *reinterpret_cast<const f16x8*>(st + fb16[j] + (((pc * 4 + kq) ^ sb16[j]) << 4));
    }
  };
  // the 48 MFMAs of one K-step on the fragments in registers; the six DMAs of stage `dkt` (if dkt < nkt) go out one at a time
  // between the MFMA groups: an LDS-DMA costs the issuing wave 60-185 cycles (MI355X_MICROARCH.md), spread out it sits beside
  // the partner wave's MFMAs
  auto mma16 = [&](int dkt) {
    const bool more = dkt < nkt;
    const int nstage = dkt % GD_STAGES;
    int q = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 c = lo16[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[1][i], b16[0][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[0][i], b16[1][j], c, 0, 0, 0);
        lo16[i][j] = c;
        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[0][i], b16[0][j], acc16[i][j], 0, 0, 0);
        if (GD_SPREAD && (j & 1) && q < 6) {                                   // after every second group: 8 slots for 6 DMAs
          __builtin_amdgcn_sched_barrier(0);
          if (more) issue_one(nstage, dkt, q);
          __builtin_amdgcn_sched_barrier(0);
          ++q;
        }
      }
    if (!GD_SPREAD && more) issue(nstage, dkt);
  };
  const bool late = STAG && wid >= 4;                                       // wave-uniform
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // my six DMAs of stage kt have landed; stage kt + 1's stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GD_ACC(0, gd_t);
    gd_barrier();                                                           // everybody's have; everybody is done with stage kt - 1
    GD_ACC(1, gd_t);
    if constexpr (SHAPE == 32) {
      // The six DMAs of stage kt + 2 are issued ONE AT A TIME between the MFMA groups of this step, not in a burst behind the
      // barrier (see mma16)
      const bool more = kt + 2 < nkt;
      const int nstage = (kt + 2) % GD_STAGES;
      const unsigned char* st = smem + (kt % GD_STAGES) * GD_STAGE;
      int q = 0;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f16x8 a[VCG_NP][MI], b[VCG_NP][NI];
#pragma unroll
        for (int pc = 0; pc < VCG_NP; ++pc) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            a[pc][i] = *reinterpret_cast<const f16x8*>(st + fa[i] + (((pc * 4 + 2 * s + lh) ^ sa[i]) << 4));
#pragma unroll
          for (int j = 0; j < NI; ++j)
            b[pc][j] = *reinterpret_cast<const f16x8*>(st + fb[j] + (((pc * 4 + 2 * s + lh) ^ sb[j]) << 4));
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
            if (GD_SPREAD && q < 6 && !(s == 1 && i == 1 && j == 1)) {          // after 7 of the 8 groups: 6 DMAs + one spare slot
              __builtin_amdgcn_sched_barrier(0);
              if (more) issue_one(nstage, kt + 2, q);
              __builtin_amdgcn_sched_barrier(0);
              ++q;
            }
          }
      }
      if (!GD_SPREAD && more) issue(nstage, kt + 2);
    } else if (!late) {
      read16(kt);
      mma16(kt + 2);
    } else {
      // the late half: multiply step kt - 1 (fragments read in the previous interval) under the early half's reads, then read
      // step kt under the early half's MFMAs.  Its DMAs of stage kt + 2 go out in the same interval as the early half's.
      if (kt > 0) mma16(kt + 2);
      else if (kt + 2 < nkt) issue((kt + 2) % GD_STAGES, kt + 2);
      __builtin_amdgcn_sched_barrier(0);
      read16(kt);
    }
    GD_ACC(2, gd_t);
  }
  if constexpr (SHAPE == 16) {
    if (late) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mma16(nkt);                                                            // the last step's products (no DMA left to issue)
    }
  }
  float* const dst = p.c + (size_t)zb * p.c_bstride;
  const float os = sa_ * sb_;
  if constexpr (SHAPE == 32) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
          const int m = m0 + wm * 64 + i * 32 + row;
          if (m < p.rows) dst[(size_t)m * p.N + n] = (acc[i][j][e] + lo[i][j][e]) * os;
        }
    }
  } else {
    // 16 x 16 accumulator: lane -> column lane & 15, rows 4 (lane >> 4) + e
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + r16;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * 64 + i * 16 + 4 * kq + e;
          if (m < p.rows) dst[(size_t)m * p.N + n] = (acc16[i][j][e] + lo16[i][j][e]) * os;
        }
    }
  }
#ifdef VCG_GD_STAMP
  GD_ACC(3, gd_t);
  if (g_gd_stamp && lane == 0) {
    unsigned long long* o = g_gd_stamp + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wid) * 8;
    o[0] = gd_acc[0]; o[1] = gd_acc[1]; o[2] = gd_acc[2]; o[3] = gd_acc[3];
    o[6] = __builtin_amdgcn_s_memrealtime() - gd_rt0; o[7] = gd_t - gd_start;
  }
#endif
}
// VCG_GEMM_DMA=0: the register-staged 128 x 128 kernel for every shape (A/B measurements)
static bool gemm_dma_on() {
  static const int on = [] { const char* e = getenv("VCG_GEMM_DMA"); return e ? atoi(e) : 1; }();
  return on != 0;
}

// X[rows][K] fp32 -> blocked planes of X / s (see the top of this file), s from `amax`; one thread per 4 consecutive k.  K % 32 == 0.
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ x, unsigned short* __restrict__ bp, size_t quads, int K, VcgAmax amax) {
  float s, inv;
  vcg_scale_of(vcg_amax_bits(amax), amax.shift, s, inv);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4, row = e / K;
    const int k = (int)(e - row * K);
    uint2 h, l;
    split4h(*reinterpret_cast<const float4*>(x + e), inv, h, l);
    unsigned short* o = bp + (row * (K / 32) + k / 32) * VCG_PBLK + (k & 31);
    *reinterpret_cast<uint2*>(o) = h;
    *reinterpret_cast<uint2*>(o + 32) = l;
  }
}
int vcg_split_planes(const float* x, void* bp, size_t rows, int K, const VcgAmax& amax, hipStream_t st) {
  VCG_CHECK_ARG(K % 32 == 0, "vcg_split_planes: K must be a multiple of 32");
  const size_t quads = rows * K / 4;
  size_t blocks = (quads + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_split_planes, dim3((unsigned)blocks), dim3(256), 0, st, x, (unsigned short*)bp, quads, K, amax);
  VCG_LAUNCH_CHECK("vcg_split_planes");
  return 0;
}

// both operands as blocked planes (A from k_wino_in_planes: [batch][rows][K / 32][2][32]); K % 32 == 0, N % 64 == 0
int vcg_gemm_planes_batched(const void* APlanes, const void* BtPlanes, float* C, int rows, int K, int N, int batches, const VcgAmax& amax_a,
                            const VcgAmax& amax_b, hipStream_t st, uint32_t* amax_a_keep) {
  VCG_CHECK_ARG(K % 32 == 0 && N % 64 == 0 && rows > 0, "vcg_gemm_planes_batched: bad shape rows=%d K=%d N=%d", rows, K, N);
  VCG_CHECK_ARG((unsigned long long)rows * K * 2 * VCG_NP < (1ull << 31) && (unsigned long long)N * K * 2 * VCG_NP < (1ull << 31),
                "vcg_gemm_planes_batched: operand extents must stay below 2 GiB per batch");
  VCG_CHECK_ARG((unsigned long long)rows * K * VCG_NP * (unsigned long long)batches < (1ull << 32) &&
                    (unsigned long long)N * K * VCG_NP * (unsigned long long)batches < (1ull << 32),
                "vcg_gemm_planes_batched: batch stride overflow");
  GemmSplitP p;
  p.a = (const float*)APlanes; p.bt = (const float*)BtPlanes; p.c = C; p.rows = rows; p.K = K; p.N = N;
  p.a_bytes = (uint32_t)((size_t)rows * K * 2 * VCG_NP); p.b_bytes = (uint32_t)((size_t)N * K * 2 * VCG_NP);
  p.a_bstride = (uint32_t)((size_t)rows * K * VCG_NP); p.b_bstride = (uint32_t)((size_t)N * K * VCG_NP);
  p.c_bstride = (size_t)rows * N;
  p.amax_a = amax_a; p.amax_b = amax_b; p.amax_a_keep = amax_a_keep;
  const int bn = (N % 128 == 0) ? 128 : 64;
  if (bn == 128 && rows >= 256 && gemm_dma_on()) {
    dim3 grid((rows + GD_BM - 1) / GD_BM, N / GD_BN, batches);
    VcgProfScope prof("k_gemm_planes_dma", 2.0 * rows * (double)K * N * batches, st);
    // VCG_GEMM_SPREAD=0: the DMAs of a stage in one burst behind the barrier (A/B measurements)
    static const int spread = [] { const char* e = getenv("VCG_GEMM_SPREAD"); return e ? atoi(e) : 1; }();
    // VCG_GEMM_SHAPE=32: v_mfma_f32_32x32x16_f16 tiles (A/B measurements)
    static const int shape = [] { const char* e = getenv("VCG_GEMM_SHAPE"); return e ? atoi(e) : 16; }();
    // VCG_GEMM_STAGGER=0: both waves of a SIMD in lockstep, as in round 3 (A/B measurements)
    static const int stagger = [] { const char* e = getenv("VCG_GEMM_STAGGER"); return e ? atoi(e) : 1; }();
    if (shape == 16 && stagger && spread) {
      hipLaunchKernelGGL((k_gemm_planes_dma<true, 16, true>), grid, dim3(512), 0, st, p);
    } else if (shape == 16) {
      if (spread) hipLaunchKernelGGL((k_gemm_planes_dma<true, 16>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((k_gemm_planes_dma<false, 16>), grid, dim3(512), 0, st, p);
    } else {
      if (spread) hipLaunchKernelGGL((k_gemm_planes_dma<true, 32>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((k_gemm_planes_dma<false, 32>), grid, dim3(512), 0, st, p);
    }
    VCG_LAUNCH_CHECK("vcg_gemm_planes_batched(dma)");
    return 0;
  }
  dim3 grid((rows + 127) / 128, N / bn, batches);
  VcgProfScope prof(bn == 128 ? "k_gemm_split<128, planes>" : "k_gemm_split<64, planes>", 2.0 * rows * (double)K * N * batches, st);
  if (bn == 128) hipLaunchKernelGGL((k_gemm_split<128, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_split<64, true>), grid, dim3(256), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_planes_batched");
  return 0;
}

// rows x K (fp32) times (N x K)^T (blocked planes) per batch; K % 32 == 0, N % 64 == 0
int vcg_gemm_split_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, const VcgAmax& amax_a,
                           const VcgAmax& amax_b, hipStream_t st, uint32_t* amax_a_keep) {
  const float* Bt = (const float*)BtPlanes;
  VCG_CHECK_ARG(K % 32 == 0 && N % 64 == 0 && rows > 0, "vcg_gemm_split_batched: bad shape rows=%d K=%d N=%d", rows, K, N);
  VCG_CHECK_ARG((unsigned long long)rows * K * 4 < (1ull << 31) && (unsigned long long)N * K * 2 * VCG_NP < (1ull << 31),
                "vcg_gemm_split_batched: operand extents must stay below 2 GiB per batch");
  VCG_CHECK_ARG((unsigned long long)rows * K * (unsigned long long)batches < (1ull << 32) &&
                    (unsigned long long)N * K * VCG_NP * (unsigned long long)batches < (1ull << 32),
                "vcg_gemm_split_batched: batch stride overflow");
  GemmSplitP p;
  p.a = A; p.bt = Bt; p.c = C; p.rows = rows; p.K = K; p.N = N;
  p.a_bytes = (uint32_t)((size_t)rows * K * 4); p.b_bytes = (uint32_t)((size_t)N * K * 2 * VCG_NP);
  p.a_bstride = (uint32_t)((size_t)rows * K); p.b_bstride = (uint32_t)((size_t)N * K * VCG_NP);
  p.c_bstride = (size_t)rows * N;
  p.amax_a = amax_a; p.amax_b = amax_b; p.amax_a_keep = amax_a_keep;
  const int bn = (N % 128 == 0) ? 128 : 64;
  dim3 grid((rows + 127) / 128, N / bn, batches);
  VcgProfScope prof(bn == 128 ? "k_gemm_split<128>" : "k_gemm_split<64>", 2.0 * rows * (double)K * N * batches, st);
  if (bn == 128) hipLaunchKernelGGL((k_gemm_split<128, false>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_split<64, false>), grid, dim3(256), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_split_batched");
  return 0;
}
