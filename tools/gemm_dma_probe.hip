// Diagnostic: where k_gemm_planes_dma (csrc/gemm_split.hip) spends its shader clocks per K-step — the wait for its own DMAs, the
// barrier, the fragment reads + MFMAs + DMA issue — on the Winograd GEMM shapes of the step, and the clock the chip holds.
// Build (GPU box):  C=vae-cyclegan-implementation_amd/csrc; mkdir -p tools/_build; hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -fno-slp-vectorize -DVCG_GD_STAMP -o tools/_build/gemm_dma_probe tools/gemm_dma_probe.hip $C/gemm_split.hip $C/misc.hip
#include "../vae-cyclegan-implementation_amd/csrc/vcg_common.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

int vcg_gemm_planes_batched(const void* APlanes, const void* BtPlanes, float* C, int rows, int K, int N, int batches, const VcgAmax& amax_a,
                            const VcgAmax& amax_b, hipStream_t st, uint32_t* amax_a_keep);
int vcg_split_planes(const float* x, void* bp, size_t rows, int K, const VcgAmax& amax, hipStream_t st);
int vcg_gd_set_stamp(void* buf);
extern "C" const char* vcg_last_error();

static void run(const char* what, int rows, int K, int N, int batches) {
  const size_t na = (size_t)batches * rows * K, nb = (size_t)batches * N * K, nc = (size_t)batches * rows * N;
  std::vector<float> ha(na), hb(nb);
  uint32_t x = 99;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xFFFFFF) / 16777216.f - 0.5f; };
  for (auto& v : ha) v = rnd() * 3.f;
  for (auto& v : hb) v = rnd();
  float *A, *B, *C;
  void *Ap, *Bp;
  hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, nc * 4); hipMalloc(&Ap, na * 4); hipMalloc(&Bp, nb * 4);
  hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice);
  const float ma = 1.5f, mb = 0.5f;
  uint32_t ba, bb; memcpy(&ba, &ma, 4); memcpy(&bb, &mb, 4);
  const VcgAmax aa = vcg_amax_const(ba), ab = vcg_amax_const(bb);
  vcg_split_planes(A, Ap, (size_t)batches * rows, K, aa, 0);
  vcg_split_planes(B, Bp, (size_t)batches * N, K, ab, 0);
  const int wgs = ((rows + 255) / 256) * (N / 128) * batches;
  unsigned long long* d;
  hipMalloc(&d, (size_t)wgs * 64 * 8);
  for (int w = 0; w < 20; ++w) vcg_gemm_planes_batched(Ap, Bp, C, rows, K, N, batches, aa, ab, 0, nullptr);     // warm: the clock settles
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) vcg_gemm_planes_batched(Ap, Bp, C, rows, K, N, batches, aa, ab, 0, nullptr);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemset(d, 0, (size_t)wgs * 64 * 8);
  vcg_gd_set_stamp(d);
  if (vcg_gemm_planes_batched(Ap, Bp, C, rows, K, N, batches, aa, ab, 0, nullptr)) { printf("failed: %s\n", vcg_last_error()); return; }
  hipDeviceSynchronize();
  vcg_gd_set_stamp(nullptr);
  std::vector<unsigned long long> h((size_t)wgs * 64);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s[8] = {};
  for (size_t i = 0; i < h.size(); ++i) s[i & 7] += (double)h[i];
  const double cnt = (double)wgs * 8, nk = K / 32;
  printf("%-18s rows %6d K %5d N %5d x%2d: %7.1f us per launch (%6.1f TF fp32-equivalent), %d workgroups\n", what, rows, K, N, batches, ms * 100.0,
         2.0 * batches * rows * (double)N * K / (ms * 1e-4) * 1e-12, wgs);
  printf("   shader clocks per K-step and wave: wait for own DMAs %6.0f | barrier %6.0f | reads + 24 MFMAs + 6 DMA issues %6.0f  (24 MFMAs alone: 768; two waves share a SIMD)\n",
         s[0] / cnt / nk, s[1] / cnt / nk, s[2] / cnt / nk);
  printf("   per wave: K loop %8.0f clocks, epilogue %6.0f, whole kernel %8.0f clocks = %6.1f us -> clock %.2f GHz\n", (s[0] + s[1] + s[2]) / cnt, s[3] / cnt,
         s[7] / cnt, s[6] / cnt / 100.0, (s[7] / cnt) / (s[6] / cnt / 100.0) * 1e-3);
  hipFree(A); hipFree(B); hipFree(C); hipFree(Ap); hipFree(Bp); hipFree(d);
}

int main() {
  run("R forward", 512, 1024, 1024, 16);
  run("D4 forward", 512, 2048, 1024, 16);
  run("D3 forward", 2048, 1024, 512, 16);
  run("D2 forward", 8192, 512, 256, 16);
  run("R data gradient", 648, 1024, 1024, 16);
  return 0;
}
