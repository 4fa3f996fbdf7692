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
#include "vcg_common.h"

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

__device__ __forceinline__ float4 pp_bload4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  u32x4p v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
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
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + (size_t)zb * p.a_bstride), 0, (int)p.a_bytes, 0x00020000),
                               rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bt + (size_t)zb * p.b_bstride), 0, (int)p.b_bytes, 0x00020000);
  const int s_row = tid >> 3, s_u = tid & 7;                    // staging: row (+64 i), k quad
  uint32_t aoff[AR], boff[BR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = m0 + s_row + 64 * i;
    aoff[i] = r < p.rows ? (uint32_t)(((size_t)r * p.K + s_u * 4) * 4) : PP_OOB;
  }
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int r = n0 + s_row + 64 * i;
    boff[i] = r < p.N ? (uint32_t)(((size_t)r * p.K + s_u * 4) * 4) : PP_OOB;
  }
  uint32_t soff[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = s_row + 64 * i;
    soff[i] = (uint32_t)(r * 64 + (((s_u >> 1) ^ ((r >> 2) & 3)) << 4) + ((s_u & 1) << 3));
  }

  f32x16 acc[MI][NI], lo[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  float4 va[AR], vb[BR];
  const int nkt = (p.K + 31) / 32;
  auto load_tiles = [&](int kt) {
    const bool kv = kt * 32 + s_u * 4 < p.K;                    // K is a multiple of 4
#pragma unroll
    for (int i = 0; i < AR; ++i) va[i] = pp_bload4(ra, (kv && aoff[i] != PP_OOB) ? aoff[i] + (uint32_t)kt * 128u : PP_OOB);
#pragma unroll
    for (int i = 0; i < BR; ++i) vb[i] = pp_bload4(rb, (kv && boff[i] != PP_OOB) ? boff[i] + (uint32_t)kt * 128u : PP_OOB);
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      uint2 h, m, l;
      split4(va[i], h, m, l);
      *reinterpret_cast<uint2*>(&As[buf][0][soff[i]]) = h;
      *reinterpret_cast<uint2*>(&As[buf][1][soff[i]]) = m;
      *reinterpret_cast<uint2*>(&As[buf][2][soff[i]]) = l;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      uint2 h, m, l;
      split4(vb[i], h, m, l);
      *reinterpret_cast<uint2*>(&Bs[buf][0][soff[i]]) = h;
      *reinterpret_cast<uint2*>(&Bs[buf][1][soff[i]]) = m;
      *reinterpret_cast<uint2*>(&Bs[buf][2][soff[i]]) = l;
    }
  };
  uint32_t fa[MI], fb[NI];
  int sa[MI], sb[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { const int r = wm * 64 + i * 32 + l31; fa[i] = (uint32_t)(r * 64); sa[i] = (r >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < NI; ++j) { const int r = wn * 64 + j * 32 + l31; fb[j] = (uint32_t)(r * 64); sb[j] = (r >> 2) & 3; }
  auto mma = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][MI], b[3][NI];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[pc][i] = *reinterpret_cast<const bf16x8*>(&As[buf][pc][fa[i] + (((2 * s + lh) ^ sa[i]) << 4)]);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[pc][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][pc][fb[j] + (((2 * s + lh) ^ sb[j]) << 4)]);
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

#ifdef VCG_PP_STAMP
  unsigned long long st_mma = 0, st_stage = 0, st_bar = 0;
  const unsigned long long t_begin = PP_T();
#endif
  load_tiles(0);
  store_tiles(0);
  if (nkt > 1) load_tiles(1);                 // both halves enter the loop holding the next tile in registers
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const unsigned long long t0 = PP_T();
    if (late) {
      if (kt + 1 < nkt) store_tiles(cur ^ 1);
      if (kt + 2 < nkt) load_tiles(kt + 2);
    }
    const unsigned long long t1 = PP_T();
    mma(cur);                                   // one code copy: the accumulators live in one place
    const unsigned long long t2 = PP_T();
    if (!late) {
      if (kt + 1 < nkt) store_tiles(cur ^ 1);
      if (kt + 2 < nkt) load_tiles(kt + 2);
    }
    const unsigned long long t3 = PP_T();
    __syncthreads();
#ifdef VCG_PP_STAMP
    const unsigned long long t4 = PP_T();
    st_mma += t2 - t1; st_stage += (t1 - t0) + (t3 - t2); st_bar += t4 - t3;
#endif
  }
#ifdef VCG_PP_STAMP
  if (g_pp_stamp && lane == 0) {
    unsigned long long* o = g_pp_stamp + 4ull * ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wid);
    o[0] = st_mma; o[1] = st_stage; o[2] = st_bar; o[3] = PP_T() - t_begin;
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
  hipLaunchKernelGGL(k_gemm_pp, grid, dim3(512), 0, st, p);
  VCG_LAUNCH_CHECK("vcg_gemm_pp_batched");
  return 0;
}
