// Diagnostic (not part of libvcg.so): what does the conv main loop's MFMA stream sustain on its own?
// The real mma_ktile of conv_igemm.hip with the global-memory side removed, in variants:
//   0  operands in registers (no LDS)                  3  as 1 + one barrier per K-step
//   1  ds_read_b32 fragments, the kernels' LDS layout  4  ds_read_b128 A fragments from an XOR-swizzled [m][32] image
//   2  as 1 without the fragment pipelining fence      5  ds_read_b128 A and B fragments
// each at 1 and 2 workgroups per CU.
// Build: C=vae-cyclegan-implementation_amd/csrc; hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/mfma_probe \
//          tools/mfma_probe.hip $C/conv_thin.hip $C/conv_wino.hip $C/norm.hip $C/misc.hip
#include "../vae-cyclegan-implementation_amd/csrc/conv_igemm.hip"

#include <math.h>
#include <vector>

template <int MODE, int NT = 256>
__global__ __launch_bounds__(NT, 512 / NT) void k_probe(float* __restrict__ out, const float* __restrict__ seed, int nkt,
                                                  int lds_pad_dummy) {
  constexpr int BM = 128, BN = 128, MI = 2, NI = 2;
  __shared__ __attribute__((aligned(16))) float As[2][BM * AS_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wid = (tid >> 6) & 3;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * BM * AS_STRIDE; i += NT) (&As[0][0])[i] = seed[i & 4095];
  for (int i = tid; i < 2 * BK * BN; i += NT) (&Bs[0][0])[i] = seed[(i * 7) & 4095];
  __syncthreads();
  f32x16 acc[MI][NI], tot[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = tot[i][j][e] = 0.f;
  float ra[2] = {seed[tid & 255], seed[(tid & 255) + 256]}, rb[2] = {seed[(tid & 255) + 512], seed[(tid & 255) + 768]};

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const float* const Ac = As[cur];
    const float* const Bc = Bs[cur];
    if (MODE == 0) {
      mma_ktile<MI, NI>(acc, [&](int kk, int i) { return ra[i]; }, [&](int kk, int j) { return rb[j]; }, lh);
    } else if (MODE == 1 || MODE == 3) {
      mma_ktile<MI, NI>(
          acc, [&](int kk, int i) { return Ac[(wm * (BM / 2) + i * 32 + l31) * AS_STRIDE + kk]; },
          [&](int kk, int j) { return Bc[kk * BN + wn * (BN / 2) + j * 32 + l31]; }, lh);
      if (MODE == 3) __syncthreads();
    } else if (MODE == 2) {
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        float a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = Ac[(wm * (BM / 2) + i * 32 + l31) * AS_STRIDE + ks * 2 + lh];
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = Bc[(ks * 2 + lh) * BN + wn * (BN / 2) + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // b128 fragments: A image [m][32] with 16-byte chunks XOR-swizzled by (m & 7); lane (l31, lh) reads chunk 2g+lh
      // -> k = 8g + 4lh + t for the 4 MFMA steps t of group g.  B either b32 from the k-major image at the same k
      // (MODE 4) or b128 from a swizzled [n][32] image (MODE 5; aliases As/Bs memory, contents are irrelevant here).
      const float* const Af = &As[0][0] + cur * (BM * 32);
      const float* const Bf = &Bs[0][0] + cur * (BN * 32) / 2;   // only addresses matter for the probe
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 a4[MI], b4[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = wm * (BM / 2) + i * 32 + l31;
          a4[i] = *reinterpret_cast<const float4*>(&Af[row * 32 + (((2 * g + lh) ^ (row & 7)) << 2)]);
        }
        if (MODE == 5) {
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int col = wn * (BN / 2) + j * 32 + l31;
            b4[j] = *reinterpret_cast<const float4*>(&Bf[(col * 32 + (((2 * g + lh) ^ (col & 7)) << 2)) & 4095]);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float b[NI];
#pragma unroll
          for (int j = 0; j < NI; ++j)
            b[j] = MODE == 5 ? (t == 0 ? b4[j].x : t == 1 ? b4[j].y : t == 2 ? b4[j].z : b4[j].w)
                             : Bc[(8 * g + 4 * lh + t) * BN + wn * (BN / 2) + j * 32 + l31];
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const float av = t == 0 ? a4[i].x : t == 1 ? a4[i].y : t == 2 ? a4[i].z : a4[i].w;
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    if (((kt + 1) & (FLUSH_TILES - 1)) == 0 && kt + 1 < nkt) flush_acc<MI, NI>(acc, tot);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e] + tot[i][j][e];
  out[(size_t)blockIdx.x * NT + tid] = s + (float)lds_pad_dummy;
}


// ---------------------------------------------------------------------------------------------------------
// Mode 6: what a whole staged GEMM loop sustains with LDS-DMA staging: C[m][n] = sum_k A[m][k] B[n][k],
// 512 threads = 8 waves (4 x 2), tile 256 x 128, BK = 32, both operand images [row][32] with the 16-byte chunks
// XOR-swizzled by (row >> 1) & 7 (swizzle applied on the DMA's SOURCE address), ds_read_b128 fragments,
// 2 LDS buffers, one __syncthreads per K-step.
// The DMA is issued from inline asm: through the builtin, hipcc treats every later ds_read as a possible reader of
// the DMA's LDS destination and drains vmcnt(0) in front of the CURRENT tile's fragment reads (seen in the ISA),
// which serialises the staging with the MFMAs.  Invisible to hipcc's counters, the DMA needs our own vmcnt(0)
// before the barrier that publishes the tile.
struct Srd { uint32_t w[4]; };
__device__ __forceinline__ u32x4 make_srd_words(const float* ptr, uint32_t bytes) {
  const uint64_t a = (uint64_t)ptr;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void dma16(u32x4 srd, const float* lds_dst, uint32_t voff) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)lds_dst);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(dst), "s"(srd) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int BM, int BN, int STAGES = 2, bool SAME = false>
__global__ __launch_bounds__(512, 1) void k_gemm_dma(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ C, int M, int N, int K) {
  constexpr int MI = 2, NI = 2, WM = BM / 64;                 // waves: WM x (8 / WM)
  constexpr int WN = 8 / WM;
  static_assert(WN * 64 == BN, "tile/wave mismatch");
  __shared__ __attribute__((aligned(1024))) float As[STAGES][BM * 32];
  __shared__ __attribute__((aligned(1024))) float Bs[STAGES][BN * 32];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const u32x4 ra = make_srd_words(A, (uint32_t)((size_t)M * K * 4)), rb = make_srd_words(B, (uint32_t)((size_t)N * K * 4));
  // DMA pieces: 8 rows x 128 B each.  A: BM/8 pieces, B: BN/8 pieces, dealt round-robin to the 8 waves.
  constexpr int PA = BM / 64, PB = BN / 64;                   // pieces per wave
  const int prow = lane >> 3, pslot = lane & 7;
  uint32_t aoff[PA], boff[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wid * PA + i) * 8 + prow;
    const int c = pslot ^ ((row >> 1) & 7);
    aoff[i] = (uint32_t)(((size_t)((SAME ? 0 : m0) + row) * K + c * 4) * 4);
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = (wid * PB + i) * 8 + prow;
    const int c = pslot ^ ((row >> 1) & 7);
    boff[i] = (uint32_t)(((size_t)((SAME ? 0 : n0) + row) * K + c * 4) * 4);
  }
  auto issue = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i) { dma16(ra, &As[buf][(wid * PA + i) * 256], aoff[i]); aoff[i] += 128; }
#pragma unroll
    for (int i = 0; i < PB; ++i) { dma16(rb, &Bs[buf][(wid * PB + i) * 256], boff[i]); boff[i] += 128; }
  };
  f32x16 acc[MI][NI], tot[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = tot[i][j][e] = 0.f;
  int arow[MI], brow[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) arow[i] = wm * 64 + i * 32 + l31;
#pragma unroll
  for (int j = 0; j < NI; ++j) brow[j] = wn * 64 + j * 32 + l31;

  const int nkt = K / 32;
  // STAGES LDS buffers: tiles kt+1 .. kt+STAGES-1 are in flight while tile kt is multiplied; each wave waits only
  // for its own DMAs of tile kt+1 (counted vmcnt leaves the newer tiles' in flight), then the barrier publishes it
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st) issue(st);
  if (STAGES == 2) dma_wait_all();
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (PA + PB)) : "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const float* const Ac = As[cur];
    const float* const Bc = Bs[cur];
    {
      int nb = cur + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      issue(nb);                                  // tile kt+STAGES-1 (past the end: in-bounds garbage or zeros, never read)
    }
    float4 a4[2][MI], b4[2][NI];
    auto frag = [&](int g, int s) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a4[s][i] = *reinterpret_cast<const float4*>(&Ac[arow[i] * 32 + (((2 * g + lh) ^ ((arow[i] >> 1) & 7)) << 2)]);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        b4[s][j] = *reinterpret_cast<const float4*>(&Bc[brow[j] * 32 + (((2 * g + lh) ^ ((brow[j] >> 1) & 7)) << 2)]);
    };
    frag(0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int s = g & 1;
      if (g + 1 < 4) frag(g + 1, s ^ 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const float av = t == 0 ? a4[s][i].x : t == 1 ? a4[s][i].y : t == 2 ? a4[s][i].z : a4[s][i].w;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const float bv = t == 0 ? b4[s][j].x : t == 1 ? b4[s][j].y : t == 2 ? b4[s][j].z : b4[s][j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
        }
      }
      if (g == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);                  // frags one group ahead
      else if (g < 3) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * MI * NI, 0);
    }
    if (((kt + 1) & (FLUSH_TILES - 1)) == 0 && kt + 1 < nkt) flush_acc<MI, NI>(acc, tot);
    if (STAGES == 2) dma_wait_all();
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (PA + PB)) : "memory");
    __syncthreads();
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  }
  dma_wait_all();
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      acc[i][j] += tot[i][j];
      const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        C[(size_t)(m0 + wm * 64 + i * 32 + row) * N + n] = acc[i][j][e];
      }
    }
}

template <int STAGES, bool SAME>
static void run_gemm(int M, int N, int K) {
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 4);
  hipMalloc(&B, (size_t)N * K * 4);
  hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  uint32_t x = 777;
  for (auto& v : ha) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xFFFF) / 65536.f - 0.5f; }
  for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xFFFF) / 65536.f - 0.5f; }
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  dim3 grid(M / 256, N / 128);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_gemm_dma<256, 128, STAGES, SAME>), grid, dim3(512), 0, 0, A, B, C, M, N, K);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 5;
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_gemm_dma<256, 128, STAGES, SAME>), grid, dim3(512), 0, 0, A, B, C, M, N, K);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> hc(4 * (size_t)N);
  hipMemcpy(hc.data(), C + (size_t)300 * N, 4 * (size_t)N * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int r = 0; r < 4; ++r)
    for (int n = 0; n < N; n += 37) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)ha[(size_t)(300 + r) * K + k] * hb[(size_t)n * K + k];
      maxerr = fmax(maxerr, fabs(ref - hc[(size_t)r * N + n]));
      maxref = fmax(maxref, fabs(ref));
    }
  const double us = ms * 1e3 / reps;
  printf("mode 6 LDS-DMA GEMM 256x128, 8 waves, stages %d%s, M=%d N=%d K=%d (%d WGs): %8.1f us %6.1f TF   max err %.2e (ref %.2f)\n", STAGES, SAME ? " SAME-TILE" : "", M, N, K,
         grid.x * grid.y, us, 2.0 * M * N * K / us * 1e-6, maxerr, maxref);
  hipFree(A); hipFree(B); hipFree(C);
}

template <int MODE, int NT = 256>
static void run(const char* what, float* out, const float* seed, int wgs, int nkt) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_probe<MODE, NT>), dim3(wgs), dim3(NT), 0, 0, out, seed, nkt, 0);
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_probe<MODE, NT>), dim3(wgs), dim3(NT), 0, 0, out, seed, nkt, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double flop = (double)wgs * nkt * (NT / 64) /*waves*/ * 64 /*mfma*/ * 4096.0;
  printf("mode %d %-44s wgs %4d: %8.1f us  %6.1f TF  (%.0f cycles@2.4GHz per wave-K-step)\n", MODE, what, wgs, us,
         flop / us * 1e-6, us * 2400.0 / nkt);
}

int main() {
  float *out, *seed;
  hipMalloc(&out, 1024 * 512 * 4);
  hipMalloc(&seed, 4096 * 4);
  std::vector<float> h(4096);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xFFFF) / 65536.f - 0.5f; }
  hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const int nkt = 144;
  for (int wgs : {256, 512}) {
    run<0>("registers only", out, seed, wgs, nkt);
    run<1>("ds_read_b32, pipelined (the kernels' loop)", out, seed, wgs, nkt);
    run<2>("ds_read_b32, compiler-scheduled", out, seed, wgs, nkt);
    run<3>("as 1 + barrier per K-step", out, seed, wgs, nkt);
    run<4>("A ds_read_b128 swizzled, B ds_read_b32", out, seed, wgs, nkt);
    run<5>("A and B ds_read_b128 swizzled", out, seed, wgs, nkt);
  }
  run<1, 512>("8-wave workgroup, 1/CU, no barrier", out, seed, 256, nkt);
  run<3, 512>("8-wave workgroup, 1/CU, barrier per K-step", out, seed, 256, nkt);
  run<4, 512>("8-wave, b128 A frags, no barrier", out, seed, 256, nkt);
  run_gemm<2, false>(32768, 256, 4608);      // the d2 forward shape: 256 workgroups
  run_gemm<2, true>(32768, 256, 4608);
  run_gemm<3, false>(32768, 256, 4608);
  run_gemm<3, true>(32768, 256, 4608);
  run_gemm<3, false>(65536, 256, 2304);
  run_gemm<3, false>(8192, 2048, 2304);
  return 0;
}
