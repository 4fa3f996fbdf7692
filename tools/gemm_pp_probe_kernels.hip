// EXPERIMENT (round 2), not part of libvcg.so: built only by tools/gemm_split_probe.hip.  Results and why it was not
// kept: DESIGN.md §3 "ping-pong GEMM"; profiles/r02_gemm_pp_probe.txt.  k_gemm_pp (inline-asm loads, two-step prefetch)
// still returns wrong tiles on one shape (rows 8712, K 256, N 512: 0.02 % of the outputs) in the build WITHOUT stamps.
//
// Batched split-operand GEMM, "ping-pong" form:   C[z][m][n] = sum_k A[z][m][k] * Bt[z][n][k]   (fp32 in memory, k contiguous)
//
// Same arithmetic as gemm_split.hip (three bf16 pieces per fp32 operand, six bf16 MFMAs per product, fp32 accumulation in two
// chains), different schedule.  gemm_split.hip runs two independent 4-wave workgroups per CU, each alternating an MFMA
// phase with a staging phase (global -> split -> LDS) between two barriers; nothing keeps the two workgroups out of step,
// and measured they overlap poorly (MFMA pipe 44 % busy, the LDS store path alone costs a fifth of the launch).  Here ONE
// 8-wave workgroup owns a 256 x 128 tile (25 % fewer staged bytes per MFMA) and its two halves run the two phases in
// opposite order inside every K-step, with a single barrier per step:
//
//     waves 0-3:   MFMA(tile kt)          ; split + store(tile kt+1) ; issue loads(kt+2)
//     waves 4-7:   split + store(kt+1) ; issue loads(kt+2) ; MFMA(tile kt)
//
// Waves w and w+4 share a SIMD, so on every SIMD one wave is in its matrix phase while its partner stages — by
// construction, not by luck.  LDS holds two images of the tile pair (2 x 72 KiB): tile kt+1 is written into the other
// image while tile kt is read, and the one barrier per K-step publishes it.
#include "../vae-cyclegan-implementation_amd/csrc/vcg_common.h"

typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));

struct GemmPPParams {
  const float* a;
  const float* bt;
  float* c;
  int rows, K, N;
  uint32_t a_bytes, b_bytes;          // per batch (buffer-load bounds)
  uint32_t a_bstride, b_bstride;      // floats between batches
  size_t c_bstride;
};

// The staging loads are issued from inline asm so that hipcc does not count them: with two tiles in flight it waits
// vmcnt(0) before the first use of the OLDER tile (it cannot tell the two register sets' loads apart across the loop
// back-edge), which drains the newer tile too and turns a two-step prefetch back into a one-step one.  The waits are
// therefore ours: pp_wait<N> leaves the N youngest loads in flight and names every destination register as an operand,
// so that no use of them can be scheduled above it (cdna_hip_programming.md §5.7, item 1, form (ii)).
__device__ __forceinline__ u32x4p pp_srd(const float* ptr, uint32_t bytes) {
  const uint64_t a = (uint64_t)ptr;
  u32x4p r;
  r.x = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ f32x4 pp_aload4(u32x4p srd, uint32_t off) {
  f32x4 dst;
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(off), "s"(srd) : "memory");
  return dst;
}
template <int N>
__device__ __forceinline__ void pp_wait(f32x4 (&a)[4], f32x4 (&b)[2]) {
  f32x4 x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3], y0 = b[0], y1 = b[1];
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1) : "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
  a[0] = x0; a[1] = x1; a[2] = x2; a[3] = x3; b[0] = y0; b[1] = y1;
}
#define PP_OOB 0x80000000u

// Diagnostic build only (tools/gemm_split_probe.hip is compiled with -DVCG_PP_STAMP; libvcg.so never is): shader-clock
// time each wave spends in its matrix phase, its split+store phase, issuing loads, and waiting at the barrier.
#ifdef VCG_PP_STAMP
__device__ unsigned long long* g_pp_stamp = nullptr;
int vcg_pp_set_stamp(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamp), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#define PP_T() __builtin_amdgcn_s_memtime()
#else
#define PP_T() 0ull
#endif

__global__ __launch_bounds__(512, 2) void k_gemm_pp(GemmPPParams p) {
  constexpr int BM = 256, BN = 128, MI = 2, NI = 2, AR = BM / 64, BR = BN / 64;
  // [image][piece][row][32 bf16] as raw bytes: 64 B per row, 16-byte chunks XOR-swizzled by (row >> 2) & 3 (gemm_split.hip)
  __shared__ __attribute__((aligned(16))) unsigned char As[2][3][BM * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][3][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool late = wid >= 4;                                    // waves 4-7 stage first, multiply second
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
  // XCD-aware tile order (see gemm_split.hip): the N tiles that share an A tile run back to back on one XCD
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
  const u32x4p ra = pp_srd(p.a + (size_t)zb * p.a_bstride, p.a_bytes), rb = pp_srd(p.bt + (size_t)zb * p.b_bstride, p.b_bytes);
  const int s_row = tid >> 3, s_u = tid & 7;                    // staging: row (+64 i), k quad
  // Rows past the operand fall off the end of the buffer (a_bytes = rows * K * 4) and read as zeros, so one base offset
  // per operand and a scalar row stride are all the address state there is (registers are what this kernel is short of).
  const uint32_t aoff0 = (uint32_t)(((size_t)(m0 + s_row) * p.K + s_u * 4) * 4), boff0 = (uint32_t)(((size_t)(n0 + s_row) * p.K + s_u * 4) * 4);
  const uint32_t rstride = (uint32_t)p.K * 256u;                // 64 rows
  // LDS byte offset of this thread's quad in a piece image; row + 64 i keeps the swizzle term, so image offsets are + 4096 i
  const uint32_t soff0 = (uint32_t)(s_row * 64 + (((s_u >> 1) ^ ((s_row >> 2) & 3)) << 4) + ((s_u & 1) << 3));

  f32x16 acc[MI][NI], lo[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  // Two staging register sets: tile t lives in set t & 1 from its loads' issue (during step t - 3) to its split (step
  // t - 1), so every load has two full K-steps to return.  With one set the loads of the waves that stage first had only
  // their own matrix phase (~1 us) to come back — under load a miss beyond the L2 takes 2-3 us, and in-kernel stamps
  // showed those waves stalled for two thirds of every K-step (tools/gemm_split_probe.hip, profiles/r02_gemm_pp_stamps.txt).
#ifdef VCG_PP_STAMP
  unsigned long long st_mma = 0, st_stage = 0, st_bar = 0, st_wait = 0, st_issue = 0;
  const unsigned long long t_begin = PP_T();
#endif
  f32x4 va0[AR], vb0[BR], va1[AR], vb1[BR];
  const int nkt = (p.K + 31) / 32;
  auto load_tiles = [&](f32x4 (&va)[AR], f32x4 (&vb)[BR], int kt) {
    const bool kv = kt * 32 + s_u * 4 < p.K;                    // K is a multiple of 4; also false for every kt >= nkt
    const uint32_t ao = kv ? aoff0 + (uint32_t)kt * 128u : PP_OOB, bo = kv ? boff0 + (uint32_t)kt * 128u : PP_OOB;
#pragma unroll
    for (int i = 0; i < AR; ++i) va[i] = pp_aload4(ra, kv ? ao + (uint32_t)i * rstride : PP_OOB);
#pragma unroll
    for (int i = 0; i < BR; ++i) vb[i] = pp_aload4(rb, kv ? bo + (uint32_t)i * rstride : PP_OOB);
  };
  auto store_tiles = [&](const f32x4 (&va)[AR], const f32x4 (&vb)[BR], int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      uint2 h, m, l;
      split4(make_float4(va[i][0], va[i][1], va[i][2], va[i][3]), h, m, l);
      *reinterpret_cast<uint2*>(&As[buf][0][soff0 + 4096 * i]) = h;
      *reinterpret_cast<uint2*>(&As[buf][1][soff0 + 4096 * i]) = m;
      *reinterpret_cast<uint2*>(&As[buf][2][soff0 + 4096 * i]) = l;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      uint2 h, m, l;
      split4(make_float4(vb[i][0], vb[i][1], vb[i][2], vb[i][3]), h, m, l);
      *reinterpret_cast<uint2*>(&Bs[buf][0][soff0 + 4096 * i]) = h;
      *reinterpret_cast<uint2*>(&Bs[buf][1][soff0 + 4096 * i]) = m;
      *reinterpret_cast<uint2*>(&Bs[buf][2][soff0 + 4096 * i]) = l;
    }
  };
  // fragment rows: wave row block + 32 i + l31; + 32 i keeps the swizzle term, so fragment offsets are + 2048 i
  const uint32_t fa0 = (uint32_t)((wm * 64 + l31) * 64), fb0 = (uint32_t)((wn * 64 + l31) * 64);
  const int sa0 = (l31 >> 2) & 3;                               // (row >> 2) & 3 for both operands (wave blocks are multiples of 64)
  auto mma = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][MI], b[3][NI];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[pc][i] = *reinterpret_cast<const bf16x8*>(&As[buf][pc][fa0 + 2048 * i + (((2 * s + lh) ^ sa0) << 4)]);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[pc][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][pc][fb0 + 2048 * j + (((2 * s + lh) ^ sa0) << 4)]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          f32x16 c = lo[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);     // smallest contributions first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
          lo[i][j] = c;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };

  load_tiles(va0, vb0, 0);
  pp_wait<0>(va0, vb0);
  store_tiles(va0, vb0, 0);
  load_tiles(va1, vb1, 1);                    // past the last tile every offset is out of bounds: zeros, never multiplied
  load_tiles(va0, vb0, 2);
  pp_wait<0>(va1, vb1);                       // the loop is entered with landed registers (any copy hipcc places at the
  pp_wait<0>(va0, vb0);                       //  loop's entry must not read a register whose load is still in flight)
  __syncthreads();
  // One K-step.  `va`/`vb` is the register set of tile kt + 1 (parity known at compile time: the loop is unrolled by
  // two).  Its reload — tile kt + 3 — is issued at ONE program point for both wave halves, the end of the step, and
  // unconditionally, so that the asm loads' destinations never meet another definition in a phi (hipcc would resolve
  // that with v_mov copies of registers whose data has not landed).
  auto step = [&](f32x4 (&va)[AR], f32x4 (&vb)[BR], int kt, int cur) {
    const unsigned long long t0 = PP_T();
    // tile kt + 1's loads are the older ones; tile kt + 2's (the other set, 6 loads) stay in flight across this wait
    unsigned long long tw = t0;
    if (late) {
      pp_wait<6>(va, vb);
      tw = PP_T();
      store_tiles(va, vb, cur ^ 1);
    }
    const unsigned long long t1 = PP_T();
    mma(cur);                                   // one code copy per parity: the accumulators live in one place
    const unsigned long long t2 = PP_T();
    unsigned long long tw2 = t2;
    if (!late) {
      pp_wait<6>(va, vb);
      tw2 = PP_T();
      store_tiles(va, vb, cur ^ 1);
    }
    const unsigned long long ts = PP_T();
    load_tiles(va, vb, kt + 3);
    const unsigned long long t3 = PP_T();
    __syncthreads();
#ifdef VCG_PP_STAMP
    const unsigned long long t4 = PP_T();
    st_mma += t2 - t1; st_stage += (t1 - tw) + (ts - tw2); st_bar += t4 - t3; st_wait += (tw - t0) + (tw2 - t2); st_issue += t3 - ts;
#else
    (void)t0; (void)t1; (void)t2; (void)t3; (void)tw; (void)tw2; (void)ts;
#endif
  };
  for (int kt = 0; kt < nkt; kt += 2) {        // a step past the last tile stores zeros into the idle image: harmless
    step(va1, vb1, kt, 0);
    step(va0, vb0, kt + 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef VCG_PP_STAMP
  if (g_pp_stamp && lane == 0) {
    unsigned long long* o = g_pp_stamp + 8ull * ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wid);
    o[0] = st_mma; o[1] = st_stage; o[2] = st_bar; o[3] = PP_T() - t_begin; o[4] = st_wait; o[5] = st_issue;
  }
#endif
  float* const dst = p.c + (size_t)zb * p.c_bstride;
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
        if (m < p.rows) dst[(size_t)m * p.N + n] = acc[i][j][e] + lo[i][j][e];
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same GEMM on operands that were split by their PRODUCERS ("blocked planes"): for X[rows][K], K % 32 == 0,
//     bp[(row * K/32 + kb) * 96 + piece * 32 + j]   (bf16; piece 0 / 1 / 2 = h / m / l of X[row][32 kb + j])
// i.e. 192 contiguous bytes per (row, 32-wide K block).  The weights are split once per optimizer step when they are
// packed, the activations by the Winograd input transform that writes them (an HBM-bound kernel with VALU to spare), so
// this kernel's staging is a plain 16-byte copy global -> LDS: no conversion arithmetic in the K loop at all.  In-kernel
// stamps had shown that arithmetic (about 130 VALU instructions per thread and K-step) to take as long as the 48 MFMAs of
// the step, beside which it does not overlap well (profiles/r02_gemm_pp_stamps.txt).
__global__ __launch_bounds__(512, 2) void k_gemm_pp_planes(GemmPPParams p) {
  constexpr int BM = 256, BN = 128, MI = 2, NI = 2;
  __shared__ __attribute__((aligned(16))) unsigned char As[2][3][BM * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][3][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool late = wid >= 4;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
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
  // a_bstride / b_bstride count bf16 elements here (rows * K * 3 per batch), a_bytes / b_bytes the bytes of one batch
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)p.a + (size_t)zb * p.a_bstride), 0, (int)p.a_bytes, 0x00020000),
                               rb = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)p.bt + (size_t)zb * p.b_bstride), 0, (int)p.b_bytes, 0x00020000);
  const int KB = p.K / 32;
  const int sq = tid & 3, sr = tid >> 2;                        // staging: 16-byte chunk of a 64-byte piece row, row (0..127)
  const uint32_t aoff0 = (uint32_t)((size_t)(m0 + sr) * KB * 192 + sq * 16), boff0 = (uint32_t)((size_t)(n0 + sr) * KB * 192 + sq * 16);
  const uint32_t ahalf = (uint32_t)KB * (128u * 192u);          // + 128 rows
  const uint32_t soff0 = (uint32_t)(sr * 64 + ((sq ^ ((sr >> 2) & 3)) << 4));

  f32x16 acc[MI][NI], lo[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  u32x4p va[6], vb[3];
  auto load_tiles = [&](int kt) {                               // past the last K block every offset is out of bounds only for
    const uint32_t ko = (uint32_t)kt * 192u;                    // the last rows; callers never load kt >= KB
#pragma unroll
    for (int j = 0; j < 6; ++j)
      va[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(aoff0 + (uint32_t)(j / 3) * ahalf + (uint32_t)(j % 3) * 64u + ko), 0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) vb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(boff0 + (uint32_t)j * 64u + ko), 0, 0);
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4p*>(&As[buf][j % 3][soff0 + 8192 * (j / 3)]) = va[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4p*>(&Bs[buf][j][soff0]) = vb[j];
  };
  const uint32_t fa0 = (uint32_t)((wm * 64 + l31) * 64), fb0 = (uint32_t)((wn * 64 + l31) * 64);
  const int sa0 = (l31 >> 2) & 3;
  auto mma = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][MI], b[3][NI];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[pc][i] = *reinterpret_cast<const bf16x8*>(&As[buf][pc][fa0 + 2048 * i + (((2 * s + lh) ^ sa0) << 4)]);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[pc][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][pc][fb0 + 2048 * j + (((2 * s + lh) ^ sa0) << 4)]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          f32x16 c = lo[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);     // smallest contributions first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
          lo[i][j] = c;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };

  load_tiles(0);
  store_tiles(0);
  if (KB > 1) load_tiles(1);
  __syncthreads();
  for (int kt = 0; kt < KB; ++kt) {
    const int cur = kt & 1;
    if (late) {
      if (kt + 1 < KB) store_tiles(cur ^ 1);
      if (kt + 2 < KB) load_tiles(kt + 2);
    }
    mma(cur);
    if (!late) {
      if (kt + 1 < KB) store_tiles(cur ^ 1);
      if (kt + 2 < KB) load_tiles(kt + 2);
    }
    __syncthreads();
  }
  float* const dst = p.c + (size_t)zb * p.c_bstride;
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
        if (m < p.rows) dst[(size_t)m * p.N + n] = acc[i][j][e] + lo[i][j][e];
      }
  }
}

// A, Bt: blocked planes of rows x K and N x K per batch; K % 32 == 0, N % 128 == 0
int vcg_gemm_pp_planes_batched(const void* Ap, const void* Btp, float* C, int rows, int K, int N, int batches, hipStream_t st) {
  VCG_CHECK_ARG(K % 32 == 0 && N % 128 == 0 && rows > 0, "vcg_gemm_pp_planes_batched: bad shape rows=%d K=%d N=%d", rows, K, N);
  VCG_CHECK_ARG((unsigned long long)rows * K * 6 < (1ull << 31) && (unsigned long long)N * K * 6 < (1ull << 31),
                "vcg_gemm_pp_planes_batched: operand extents must stay below 2 GiB per batch");
  VCG_CHECK_ARG((unsigned long long)rows * K * 3 * (unsigned long long)batches < (1ull << 32) &&
                    (unsigned long long)N * K * 3 * (unsigned long long)batches < (1ull << 32),
                "vcg_gemm_pp_planes_batched: batch stride overflow");
  GemmPPParams p;
  p.a = (const float*)Ap; p.bt = (const float*)Btp; p.c = C; p.rows = rows; p.K = K; p.N = N;
  p.a_bytes = (uint32_t)((size_t)rows * K * 6); p.b_bytes = (uint32_t)((size_t)N * K * 6);
  p.a_bstride = (uint32_t)((size_t)rows * K * 3); p.b_bstride = (uint32_t)((size_t)N * K * 3);
  p.c_bstride = (size_t)rows * N;
  dim3 grid((rows + 255) / 256, N / 128, batches);
  VcgProfScope prof("k_gemm_pp_planes", 2.0 * rows * (double)K * N * batches, st);
  hipLaunchKernelGGL(k_gemm_pp_planes, grid, dim3(512), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_pp_planes_batched");
  return 0;
}

// rows x K times (N x K)^T per batch; K % 4 == 0, N % 128 == 0
int vcg_gemm_pp_batched(const float* A, const float* Bt, float* C, int rows, int K, int N, int batches, hipStream_t st) {
  VCG_CHECK_ARG(K % 4 == 0 && N % 128 == 0 && rows > 0, "vcg_gemm_pp_batched: bad shape rows=%d K=%d N=%d", rows, K, N);
  VCG_CHECK_ARG((unsigned long long)rows * K * 4 < (1ull << 31) && (unsigned long long)N * K * 4 < (1ull << 31),
                "vcg_gemm_pp_batched: operand extents must stay below 2 GiB per batch");
  VCG_CHECK_ARG((unsigned long long)rows * K * (unsigned long long)batches < (1ull << 32) &&
                    (unsigned long long)N * K * (unsigned long long)batches < (1ull << 32),
                "vcg_gemm_pp_batched: batch stride overflow");
  GemmPPParams p;
  p.a = A; p.bt = Bt; p.c = C; p.rows = rows; p.K = K; p.N = N;
  p.a_bytes = (uint32_t)((size_t)rows * K * 4); p.b_bytes = (uint32_t)((size_t)N * K * 4);
  p.a_bstride = (uint32_t)((size_t)rows * K); p.b_bstride = (uint32_t)((size_t)N * K);
  p.c_bstride = (size_t)rows * N;
  dim3 grid((rows + 255) / 256, N / 128, batches);
  VcgProfScope prof("k_gemm_pp", 2.0 * rows * (double)K * N * batches, st);
  hipLaunchKernelGGL(k_gemm_pp, grid, dim3(512), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_pp_batched");
  return 0;
}
