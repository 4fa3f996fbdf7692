// Diagnostic: the wave-specialised split-operand GEMM (csrc/gemm_ws.hip) next to the 4-wave kernel (csrc/gemm_split.hip) on the
// Winograd GEMM shapes of the step: agreement over ALL outputs, time, and — in a -DVCG_WS_STAMP build — where the producer and
// consumer waves spend their shader clocks per K-step.
// Build (tools/run_gemm_ws_probe.sh):
//   C=vae-cyclegan-implementation_amd/csrc; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DVCG_WS_STAMP \
//     -o tools/_build/gemm_ws_probe tools/gemm_ws_probe.hip $C/gemm_ws.hip $C/gemm_split.hip $C/misc.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

int vcg_gemm_split_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st);
int vcg_gemm_ws_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st);
int vcg_gemm_ws16_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st);
int vcg_gemm_ws_planes_batched(const void* APlanes, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st);
int vcg_split_planes(const float* x, void* bp, size_t rows, int K, hipStream_t st);
extern "C" const char* vcg_last_error();
#ifdef VCG_WS_STAMP
int vcg_ws_set_stamp(void* buf);
#endif

static void stamps(const char* what, int mode, const float* A, const void* Ap, const void* BtP, float* C, int rows, int K, int N, int batches) {
#ifdef VCG_WS_STAMP
  const int wgs = ((rows + 127) / 128) * (N / 128) * batches;
  unsigned long long* d;
  hipMalloc(&d, (size_t)wgs * 64 * 8);
  hipMemset(d, 0, (size_t)wgs * 64 * 8);
  vcg_ws_set_stamp(d);
  if (mode == 0) vcg_gemm_ws_batched(A, BtP, C, rows, K, N, batches, 0);
  else if (mode == 2) vcg_gemm_ws16_batched(A, BtP, C, rows, K, N, batches, 0);
  else vcg_gemm_ws_planes_batched(Ap, BtP, C, rows, K, N, batches, 0);
  hipDeviceSynchronize();
  vcg_ws_set_stamp(nullptr);
  std::vector<unsigned long long> h((size_t)wgs * 64);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s[2][8] = {};
  for (int w = 0; w < wgs; ++w)
    for (int v = 0; v < 8; ++v)
      for (int q = 0; q < 8; ++q) s[v >= 4][q] += (double)h[((size_t)w * 8 + v) * 8 + q];
  const double nk = K / 32, cnt = (double)wgs * 4;
  printf("   %s stamps, shader clocks per K-step and wave (%d K-steps, %d workgroups):\n", what, (int)nk, wgs);
  printf("     producers: wait for loads %6.0f | split + ds_write %6.0f | issue next loads %6.0f | barrier %6.0f | whole kernel %8.0f\n",
         s[0][0] / cnt / nk, s[0][1] / cnt / nk, s[0][2] / cnt / nk, s[0][3] / cnt / nk, s[0][7] / cnt);
  printf("     consumers: reads + 24 MFMA (slice 0) %6.0f | barrier %6.0f | reads + 24 MFMA (slice 1) %6.0f | epilogue %6.0f | whole kernel %8.0f\n",
         s[1][0] / cnt / nk, s[1][1] / cnt / nk, s[1][2] / cnt / nk, s[1][3] / cnt, s[1][7] / cnt);
  hipFree(d);
#else
  (void)what; (void)mode; (void)A; (void)Ap; (void)BtP; (void)C; (void)rows; (void)K; (void)N; (void)batches;
#endif
}

static void run(int rows, int K, int N, int batches) {
  const size_t na = (size_t)batches * rows * K, nb = (size_t)batches * N * K, nc = (size_t)batches * rows * N;
  std::vector<float> ha(na), hbt(nb);
  uint32_t x = 99;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xFFFFFF) / 16777216.f - 0.5f; };
  for (auto& v : ha) v = rnd() * 3.f;
  for (auto& v : hbt) v = rnd();
  float *A, *Bt, *C0, *C1, *C2, *C3;
  void *BtP, *Ap;
  hipMalloc(&A, na * 4); hipMalloc(&Bt, nb * 4); hipMalloc(&C0, nc * 4); hipMalloc(&C1, nc * 4); hipMalloc(&C2, nc * 4); hipMalloc(&C3, nc * 4); hipMemset(C3, 0xFF, nc * 4);
  hipMalloc(&BtP, nb * 6); hipMalloc(&Ap, na * 6);
  hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(Bt, hbt.data(), nb * 4, hipMemcpyHostToDevice);
  hipMemset(C0, 0, nc * 4); hipMemset(C1, 0xFF, nc * 4); hipMemset(C2, 0xFF, nc * 4);
  vcg_split_planes(Bt, BtP, (size_t)batches * N, K, 0);
  vcg_split_planes(A, Ap, (size_t)batches * rows, K, 0);
  setenv("VCG_GEMM_WS", "0", 1);                       // vcg_gemm_split_batched: the 4-wave kernel (read once, at the first call)
  for (int w = 0; w < 2; ++w) {
    if (vcg_gemm_split_batched(A, BtP, C0, rows, K, N, batches, 0)) { printf("split failed: %s\n", vcg_last_error()); return; }
    if (vcg_gemm_ws_batched(A, BtP, C1, rows, K, N, batches, 0)) { printf("ws failed: %s\n", vcg_last_error()); return; }
    if (vcg_gemm_ws_planes_batched(Ap, BtP, C2, rows, K, N, batches, 0)) { printf("ws planes failed: %s\n", vcg_last_error()); return; }
    if (vcg_gemm_ws16_batched(A, BtP, C3, rows, K, N, batches, 0)) { printf("ws16 failed: %s\n", vcg_last_error()); return; }
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 5;
  const double fl = 2.0 * batches * rows * N * (double)K;
  printf("rows %6d K %5d N %5d x%2d\n", rows, K, N, batches);
  for (int round = 0; round < 3; ++round) {            // interleaved rounds
    float t[4];
    for (int v = 0; v < 4; ++v) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) {
        if (v == 0) vcg_gemm_split_batched(A, BtP, C0, rows, K, N, batches, 0);
        else if (v == 1) vcg_gemm_ws_batched(A, BtP, C1, rows, K, N, batches, 0);
        else if (v == 2) vcg_gemm_ws_planes_batched(Ap, BtP, C2, rows, K, N, batches, 0);
        else vcg_gemm_ws16_batched(A, BtP, C3, rows, K, N, batches, 0);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&t[v], e0, e1);
    }
    printf("   round %d: 4-wave %8.1f us %6.1f TF | wave-specialised %8.1f us %6.1f TF | ... with A pre-split %8.1f us %6.1f TF | ... on 16x16x32 MFMAs %8.1f us %6.1f TF\n", round,
           t[0] * 1e3 / reps, fl / (t[0] * 1e-3 / reps) * 1e-12, t[1] * 1e3 / reps, fl / (t[1] * 1e-3 / reps) * 1e-12, t[2] * 1e3 / reps,
           fl / (t[2] * 1e-3 / reps) * 1e-12, t[3] * 1e3 / reps, fl / (t[3] * 1e-3 / reps) * 1e-12);
  }
  std::vector<float> h0(nc), h1(nc), h2(nc), h3(nc);
  hipMemcpy(h3.data(), C3, nc * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h0.data(), C0, nc * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h1.data(), C1, nc * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h2.data(), C2, nc * 4, hipMemcpyDeviceToHost);
  size_t d1 = 0, d2 = 0, d3 = 0;
  double md3 = 0;
  for (size_t i = 0; i < nc; ++i) {
    d1 += memcmp(&h0[i], &h1[i], 4) != 0; d2 += memcmp(&h0[i], &h2[i], 4) != 0; d3 += memcmp(&h0[i], &h3[i], 4) != 0;
    const double d = fabs((double)h3[i] - h0[i]); if (!(d <= md3)) md3 = d;
  }
  printf("   16x16x32 variant: %zu outputs differ bitwise from the 4-wave kernel, max |diff| %.3e\n", d3, md3);
  double err = 0, nrm = 0;
  for (int s = 0; s < 2000; ++s) {
    x = x * 1664525u + 1013904223u; const size_t z = (x >> 8) % batches;
    x = x * 1664525u + 1013904223u; const size_t m = (x >> 8) % rows;
    x = x * 1664525u + 1013904223u; const size_t n = (x >> 8) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)ha[(z * rows + m) * K + k] * hbt[(z * N + n) * K + k];
    const size_t ci = (z * rows + m) * N + n;
    err += (h1[ci] - ref) * (h1[ci] - ref); nrm += ref * ref;
  }
  printf("   outputs that differ bitwise from the 4-wave kernel: wave-specialised %zu, pre-split A %zu of %zu; rel err vs float64 %.2e\n", d1, d2, nc,
         sqrt(err / nrm));
  stamps("fp32 A", 0, A, Ap, BtP, C1, rows, K, N, batches);
  stamps("pre-split A", 1, A, Ap, BtP, C2, rows, K, N, batches);
  stamps("16x16x32", 2, A, Ap, BtP, C3, rows, K, N, batches);
  hipFree(A); hipFree(Bt); hipFree(C0); hipFree(C1); hipFree(C2); hipFree(C3); hipFree(BtP); hipFree(Ap);
}

int main() {
  run(512, 1024, 1024, 16);    // R forward
  run(2048, 1024, 512, 16);    // D3 forward
  run(8192, 512, 256, 16);     // D2 forward
  run(32768, 256, 128, 16);    // D1 forward
  run(648, 1024, 1024, 16);    // R data gradient (ragged rows)
  return 0;
}
