// Diagnostic: accuracy (against float64) and speed of the split-operand bf16 GEMM next to the fp32-MFMA GEMM the
// Winograd layers use.  Build: C=vae-cyclegan-implementation_amd/csrc; hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVCG_PP_STAMP \
//   -o tools/_build/gemm_split_probe tools/gemm_split_probe.hip $C/conv_igemm.hip $C/conv_thin.hip $C/conv_wino.hip $C/gemm_split.hip tools/gemm_pp_probe_kernels.hip $C/norm.hip $C/misc.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

int vcg_gemm_split_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st);
int vcg_gemm_batched(const float* A, const float* B, float* C, int rows, int K, int Ncols, int batches, hipStream_t st);
int vcg_gemm_pp_batched(const float* A, const float* Bt, float* C, int rows, int K, int N, int batches, hipStream_t st);
extern "C" const char* vcg_last_error();

#ifdef VCG_PP_STAMP
int vcg_pp_set_stamp(void* buf);
#endif
int vcg_split_planes(const float* x, void* bp, size_t rows, int K, hipStream_t st);
int vcg_gemm_pp_planes_batched(const void* Ap, const void* Btp, float* C, int rows, int K, int N, int batches, hipStream_t st);

static void stamps(const float* A, const float* Bt, float* C3, int rows, int K, int N, int batches) {
#ifndef VCG_PP_STAMP
  return;
#else
  const int wgs = ((rows + 255) / 256) * (N / 128) * batches;
  unsigned long long* d; hipMalloc(&d, (size_t)wgs * 8 * 8 * 8); hipMemset(d, 0, (size_t)wgs * 8 * 8 * 8);
  vcg_pp_set_stamp(d);
  for (int r = 0; r < 3; ++r) vcg_gemm_pp_batched(A, Bt, C3, rows, K, N, batches, 0);
  hipDeviceSynchronize();
  vcg_pp_set_stamp(nullptr);
  std::vector<unsigned long long> h((size_t)wgs * 8 * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s[2][8] = {};
  for (int w = 0; w < wgs; ++w) for (int v = 0; v < 8; ++v) for (int q = 0; q < 8; ++q) s[v >= 4][q] += (double)h[((size_t)w * 8 + v) * 8 + q];
  const double nk = ((K + 31) / 32 + 1) / 2 * 2, cnt = (double)wgs * 4;
  for (int g = 0; g < 2; ++g)
    printf("   stamps %s waves: per K-step  matrix %6.0f  vmcnt wait %6.0f  split+store %6.0f  load issue %6.0f  barrier %6.0f | kernel %8.0f clocks (%d K-steps)\n",
           g ? "late (4-7) " : "early (0-3)", s[g][0] / cnt / nk, s[g][4] / cnt / nk, s[g][1] / cnt / nk, s[g][5] / cnt / nk, s[g][2] / cnt / nk, s[g][3] / cnt, (int)nk);
  hipFree(d);
#endif
}

static void run(int rows, int K, int N, int batches) {
  const size_t na = (size_t)batches * rows * K, nb = (size_t)batches * N * K, nc = (size_t)batches * rows * N;
  std::vector<float> ha(na), hbt(nb), hb(nb);
  uint32_t x = 99;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xFFFFFF) / 16777216.f - 0.5f; };
  for (auto& v : ha) v = rnd() * 3.f;
  for (size_t z = 0; z < (size_t)batches; ++z)
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) { float v = rnd(); hbt[(z * N + n) * K + k] = v; hb[(z * K + k) * N + n] = v; }
  float *A, *Bt, *B, *C, *C2, *C3;
  hipMalloc(&A, na * 4); hipMalloc(&Bt, nb * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, nc * 4); hipMalloc(&C2, nc * 4); hipMalloc(&C3, nc * 4);
  hipMemset(C3, 0, nc * 4);
  hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(Bt, hbt.data(), nb * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms_split = 0, ms_f32 = 0;
  if (K % 32) { printf("rows %d K %d N %d: K %% 32 != 0, skipped (the weight operand comes as blocked planes)\n", rows, K, N); return; }
  void* BtP; hipMalloc(&BtP, nb * 6);
  vcg_split_planes(Bt, BtP, (size_t)batches * N, K, 0);
  for (int w = 0; w < 2; ++w) {
    if (vcg_gemm_split_batched(A, BtP, C, rows, K, N, batches, 0)) { printf("split failed: %s\n", vcg_last_error()); return; }
    if (vcg_gemm_batched(A, B, C2, rows, K, N, batches, 0)) { printf("f32 failed: %s\n", vcg_last_error()); return; }
  }
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) vcg_gemm_split_batched(A, BtP, C, rows, K, N, batches, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_split, e0, e1);
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) vcg_gemm_batched(A, B, C2, rows, K, N, batches, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_f32, e0, e1);
  float ms_pp = 0;
  const bool pp = N % 128 == 0;
  if (pp) {
    for (int w = 0; w < 2; ++w) if (vcg_gemm_pp_batched(A, Bt, C3, rows, K, N, batches, 0)) { printf("pp failed: %s\n", vcg_last_error()); return; }
    hipDeviceSynchronize();
    for (int round = 0; round < 3; ++round) {          // interleaved rounds: split, pp, split, pp ...
      float a = 0, b = 0;
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) vcg_gemm_split_batched(A, BtP, C, rows, K, N, batches, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&a, e0, e1);
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) vcg_gemm_pp_batched(A, Bt, C3, rows, K, N, batches, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&b, e0, e1);
      printf("   round %d: split %8.1f us   ping-pong %8.1f us\n", round, a * 1e3 / reps, b * 1e3 / reps);
      ms_pp = b;
    }
  }
  if (pp) stamps(A, Bt, C3, rows, K, N, batches);
  if (pp && K % 32 == 0) {                     // operands pre-split into blocked planes by their producers
    void *Ap, *Bp; float* C4;
    hipMalloc(&Ap, na * 6); hipMalloc(&Bp, nb * 6); hipMalloc(&C4, nc * 4); hipMemset(C4, 0, nc * 4);
    vcg_split_planes(A, Ap, (size_t)batches * rows, K, 0);
    vcg_split_planes(Bt, Bp, (size_t)batches * N, K, 0);
    for (int w = 0; w < 2; ++w) if (vcg_gemm_pp_planes_batched(Ap, Bp, C4, rows, K, N, batches, 0)) { printf("planes failed: %s\n", vcg_last_error()); return; }
    hipDeviceSynchronize();
    float tg = 0, ts = 0;
    for (int round = 0; round < 3; ++round) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) vcg_gemm_pp_planes_batched(Ap, Bp, C4, rows, K, N, batches, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&tg, e0, e1);
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) vcg_split_planes(A, Ap, (size_t)batches * rows, K, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ts, e0, e1);
      printf("   round %d: planes GEMM %8.1f us (%6.1f TF)   + splitting A as its own pass %8.1f us\n", round, tg * 1e3 / reps,
             2.0 * batches * rows * N * (double)K / (tg * 1e-3 / reps) * 1e-12, ts * 1e3 / reps);
    }
    std::vector<float> h4(nc), h3(nc);
    hipMemcpy(h4.data(), C4, nc * 4, hipMemcpyDeviceToHost); hipMemcpy(h3.data(), C3, nc * 4, hipMemcpyDeviceToHost);
    double md = 0; size_t bad = 0;
    for (size_t i = 0; i < nc; ++i) { const double d = fabs((double)h4[i] - h3[i]); if (d > md) md = d; if (!(d <= 1e-3)) ++bad; }
    printf("   planes vs ping-pong over ALL outputs: max |diff| %.3e (%zu beyond 1e-3)\n", md, bad);
    hipFree(Ap); hipFree(Bp); hipFree(C4);
  }
  std::vector<float> hc(nc), hc2(nc), hc3(nc);
  hipMemcpy(hc3.data(), C3, nc * 4, hipMemcpyDeviceToHost);
  double e_pp = 0, maxdiff = 0;
  if (pp) for (size_t i = 0; i < nc; ++i) { }
  hipMemcpy(hc.data(), C, nc * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hc2.data(), C2, nc * 4, hipMemcpyDeviceToHost);
  double e_split = 0, e_f32 = 0, nrm = 0;
  for (int s = 0; s < 4000; ++s) {
    x = x * 1664525u + 1013904223u; const size_t z = (x >> 8) % batches;
    x = x * 1664525u + 1013904223u; const size_t m = (x >> 8) % rows;
    x = x * 1664525u + 1013904223u; const size_t n = (x >> 8) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)ha[(z * rows + m) * K + k] * hbt[(z * N + n) * K + k];
    const size_t ci = (z * rows + m) * N + n;
    e_split += (hc[ci] - ref) * (hc[ci] - ref); e_f32 += (hc2[ci] - ref) * (hc2[ci] - ref); nrm += ref * ref;
    e_pp += (hc3[ci] - ref) * (hc3[ci] - ref);
  }
  size_t nbad = 0;
  if (pp) for (size_t i = 0; i < nc; ++i) { const double d = fabs((double)hc3[i] - hc[i]); if (d > maxdiff) maxdiff = d; if (!(d <= 1e-3)) ++nbad; }
  if (pp) printf("   ping-pong %8.1f us %6.1f TF  rel err %.2e | max |pp - split| over ALL outputs %.3e (%zu beyond 1e-3)\n", ms_pp * 1e3 / reps,
                 2.0 * batches * rows * N * (double)K / (ms_pp * 1e-3 / reps) * 1e-12, sqrt(e_pp / nrm), maxdiff, nbad);
  const double fl = 2.0 * batches * rows * N * (double)K;
  printf("rows %6d K %5d N %5d x%2d | split %8.1f us %6.1f TF  rel err %.2e | fp32 MFMA %8.1f us %6.1f TF  rel err %.2e\n", rows, K, N,
         batches, ms_split * 1e3 / reps, fl / (ms_split * 1e-3 / reps) * 1e-12, sqrt(e_split / nrm), ms_f32 * 1e3 / reps,
         fl / (ms_f32 * 1e-3 / reps) * 1e-12, sqrt(e_f32 / nrm));
  hipFree(A); hipFree(Bt); hipFree(B); hipFree(C); hipFree(C2); hipFree(C3);
}

int main() {
  run(32768, 256, 128, 16);    // D1 forward
  run(512, 1024, 1024, 16);    // R forward
  run(8192, 512, 256, 16);     // D2 forward
  run(2048, 1024, 512, 16);    // D3 forward
  run(8712, 256, 512, 16);     // D2 data gradient
  run(300, 100, 128, 3);       // ragged: row and K tails
  return 0;
}
