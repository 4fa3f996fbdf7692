// Diagnostic (round 3): is the split-operand GEMM limited by the energy of its MFMAs?  The same 4-wave 128 x 128 kernel structure
// as csrc/gemm_split.hip with each fp32 operand split into TWO fp16 pieces (x / s = h + l, 22 mantissa bits, s a power of two
// that brings the tensor's largest magnitude under fp16's range) and THREE products per multiply (hh, hl, lh; ll is below 2^-22)
// instead of three bf16 pieces and six products.  Prints time next to the bf16x3 kernel and the error of both against float64.
// Build: C=vae-cyclegan-implementation_amd/csrc; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize \
//   -o tools/_build/gemm_h2_probe tools/gemm_h2_probe.hip $C/gemm_split.hip $C/gemm_ws.hip $C/misc.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

int vcg_gemm_split_batched(const float* A, const void* BtPlanes, float* C, int rows, int K, int N, int batches, hipStream_t st);
int vcg_split_planes(const float* x, void* bp, size_t rows, int K, hipStream_t st);
extern "C" const char* vcg_last_error();

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// x * inv -> (h, l) fp16 pairs packed as 2 x uint2 (4 consecutive k)
__device__ __forceinline__ void split4h(const float4& v, float inv, uint2& h, uint2& l) {
  const float x[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
  _Float16 hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { hh[i] = (_Float16)x[i]; ll[i] = (_Float16)(x[i] - (float)hh[i]); }
  const f16x2 h0 = {hh[0], hh[1]}, h1 = {hh[2], hh[3]}, l0 = {ll[0], ll[1]}, l1 = {ll[2], ll[3]};
  h = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
  l = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}
// planes: [row][K/32][2 pieces][32] fp16 = 128 bytes per (row, K block)
__global__ __launch_bounds__(256) void k_split_planes_h(const float* __restrict__ x, unsigned short* __restrict__ bp, size_t quads, int K, float inv) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4, row = e / K;
    const int k = (int)(e - row * K);
    uint2 h, l;
    split4h(*reinterpret_cast<const float4*>(x + e), inv, h, l);
    unsigned short* o = bp + (row * (K / 32) + k / 32) * 64 + (k & 31);
    *reinterpret_cast<uint2*>(o) = h;
    *reinterpret_cast<uint2*>(o + 32) = l;
  }
}

struct P { const float* a; const void* bt; float* c; int rows, K, N; uint32_t a_bytes, b_bytes, a_bstride, b_bstride; size_t c_bstride; float inv_a, out_scale; };
#define OOB 0x80000000u

template <int WGS>
__global__ __launch_bounds__(256, WGS) void k_gemm_h2(P p) {
  constexpr int BM = 128, BN = 128, NI = 2, MI = 2, AR = 4;
  __shared__ __attribute__((aligned(16))) unsigned char As[2][BM * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + (size_t)zb * p.a_bstride), 0, (int)p.a_bytes, 0x00020000),
                               rb = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)p.bt + (size_t)zb * p.b_bstride), 0, (int)p.b_bytes, 0x00020000);
  const int s_row = tid >> 3, s_u = tid & 7;
  uint32_t aoff[AR], soff[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = m0 + s_row + 32 * i;
    aoff[i] = r < p.rows ? (uint32_t)(((size_t)r * p.K + s_u * 4) * 4) : OOB;
    const int rl = s_row + 32 * i;
    soff[i] = (uint32_t)(rl * 64 + (((s_u >> 1) ^ ((rl >> 2) & 3)) << 4) + ((s_u & 1) << 3));
  }
  constexpr int BP = 4;                                       // 2 pieces x 2 row halves
  const int b_q = tid & 3, b_r = tid >> 2;
  const int KB = p.K / 32;
  const uint32_t boff0 = (uint32_t)(((size_t)(n0 + b_r) * KB) * 128 + b_q * 16);
  const uint32_t bhalf = (uint32_t)KB * (64u * 128u);
  const uint32_t bsoff0 = (uint32_t)(b_r * 64 + ((b_q ^ ((b_r >> 2) & 3)) << 4));
  f32x16 acc[MI][NI], lo[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;
  float4 va[AR];
  u32x4 vb[BP];
  const float inv = p.inv_a;
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(aoff[i] != OOB ? aoff[i] + (uint32_t)kt * 128u : OOB), 0, 0);
      va[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
#pragma unroll
    for (int j = 0; j < BP; ++j)
      vb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(boff0 + (uint32_t)(j / 2) * bhalf + (uint32_t)(j % 2) * 64u + (uint32_t)kt * 128u), 0, 0);
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      uint2 h, l;
      split4h(va[i], inv, h, l);
      *reinterpret_cast<uint2*>(&As[0][soff[i]]) = h;
      *reinterpret_cast<uint2*>(&As[1][soff[i]]) = l;
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) *reinterpret_cast<u32x4*>(&Bs[j % 2][bsoff0 + 4096 * (j / 2)]) = vb[j];
  };
  uint32_t fa[MI], fb[NI];
  int sa[MI], sb[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { const int r = wm * 64 + i * 32 + l31; fa[i] = (uint32_t)(r * 64); sa[i] = (r >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < NI; ++j) { const int r = wn * 64 + j * 32 + l31; fb[j] = (uint32_t)(r * 64); sb[j] = (r >> 2) & 3; }
  const int nkt = KB;
  load_tiles(0);
  store_tiles();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_tiles(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 a[2][MI], b[2][NI];
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
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
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], c, 0, 0, 0);
          lo[i][j] = c;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      store_tiles();
      __syncthreads();
    }
  }
  float* const dst = p.c + (size_t)zb * p.c_bstride;
  const float os = p.out_scale;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
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

static void run(int rows, int K, int N, int batches, float amp_a, float amp_b) {
  const size_t na = (size_t)batches * rows * K, nb = (size_t)batches * N * K, nc = (size_t)batches * rows * N;
  std::vector<float> ha(na), hbt(nb);
  uint32_t x = 99;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xFFFFFF) / 16777216.f - 0.5f; };
  for (auto& v : ha) v = rnd() * amp_a;
  for (auto& v : hbt) v = rnd() * amp_b;
  float *A, *Bt, *C0, *C1;
  void *BtP, *BtH;
  hipMalloc(&A, na * 4); hipMalloc(&Bt, nb * 4); hipMalloc(&C0, nc * 4); hipMalloc(&C1, nc * 4);
  hipMalloc(&BtP, nb * 6); hipMalloc(&BtH, nb * 4);
  hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(Bt, hbt.data(), nb * 4, hipMemcpyHostToDevice);
  // per-tensor power-of-two scales: largest magnitude (a bound: amp / 2) into [2^14, 2^15)
  auto scale_of = [](float amax) { int e; frexpf(amax, &e); return ldexpf(1.f, e - 15); };      // amax / s in [2^14, 2^15)
  const float sA = scale_of(amp_a * 0.5f), sB = scale_of(amp_b * 0.5f);
  vcg_split_planes(Bt, BtP, (size_t)batches * N, K, 0);
  hipLaunchKernelGGL(k_split_planes_h, dim3(4096), dim3(256), 0, 0, Bt, (unsigned short*)BtH, nb / 4, K, 1.f / sB);
  P p;
  p.a = A; p.bt = BtH; p.c = C1; p.rows = rows; p.K = K; p.N = N;
  p.a_bytes = (uint32_t)((size_t)rows * K * 4); p.b_bytes = (uint32_t)((size_t)N * K * 4);
  p.a_bstride = (uint32_t)((size_t)rows * K); p.b_bstride = (uint32_t)((size_t)N * K * 2);
  p.c_bstride = (size_t)rows * N; p.inv_a = 1.f / sA; p.out_scale = sA * sB;
  dim3 grid((rows + 127) / 128, N / 128, batches);
  setenv("VCG_GEMM_WS", "0", 1);
  for (int w = 0; w < 2; ++w) {
    vcg_gemm_split_batched(A, BtP, C0, rows, K, N, batches, 0);
    hipLaunchKernelGGL(k_gemm_h2<2>, grid, dim3(256), 0, 0, p);
    hipLaunchKernelGGL(k_gemm_h2<3>, grid, dim3(256), 0, 0, p);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error %s\n", hipGetErrorString(hipGetLastError())); return; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 5;
  const double fl = 2.0 * batches * rows * N * (double)K;
  printf("rows %6d K %5d N %5d x%2d   (|a| <= %.3g, |b| <= %.3g: scales 2^%d, 2^%d)\n", rows, K, N, batches, amp_a / 2, amp_b / 2, (int)log2f(sA), (int)log2f(sB));
  for (int round = 0; round < 4; ++round) {
    float t[3];
    for (int v = 0; v < 3; ++v) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) {
        if (v == 0) vcg_gemm_split_batched(A, BtP, C0, rows, K, N, batches, 0);
        else if (v == 1) hipLaunchKernelGGL(k_gemm_h2<2>, grid, dim3(256), 0, 0, p);
        else hipLaunchKernelGGL(k_gemm_h2<3>, grid, dim3(256), 0, 0, p);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&t[v], e0, e1);
    }
    printf("   round %d: bf16x3 (6 MFMA) %8.1f us %6.1f TF | fp16x2 (3 MFMA) 2 WG/CU %8.1f us %6.1f TF | 3 WG/CU %8.1f us %6.1f TF\n", round, t[0] * 1e3 / reps,
           fl / (t[0] * 1e-3 / reps) * 1e-12, t[1] * 1e3 / reps, fl / (t[1] * 1e-3 / reps) * 1e-12, t[2] * 1e3 / reps, fl / (t[2] * 1e-3 / reps) * 1e-12);
  }
  std::vector<float> h0(nc), h1(nc);
  hipMemcpy(h0.data(), C0, nc * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h1.data(), C1, nc * 4, hipMemcpyDeviceToHost);
  double e0s = 0, e1s = 0, nrm = 0;
  for (int s = 0; s < 4000; ++s) {
    x = x * 1664525u + 1013904223u; const size_t z = (x >> 8) % batches;
    x = x * 1664525u + 1013904223u; const size_t m = (x >> 8) % rows;
    x = x * 1664525u + 1013904223u; const size_t n = (x >> 8) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)ha[(z * rows + m) * K + k] * hbt[(z * N + n) * K + k];
    const size_t ci = (z * rows + m) * N + n;
    e0s += (h0[ci] - ref) * (h0[ci] - ref); e1s += (h1[ci] - ref) * (h1[ci] - ref); nrm += ref * ref;
  }
  printf("   rel L2 error vs float64: bf16x3 %.2e   fp16x2 %.2e\n", sqrt(e0s / nrm), sqrt(e1s / nrm));
  hipFree(A); hipFree(Bt); hipFree(C0); hipFree(C1); hipFree(BtP); hipFree(BtH);
}

int main() {
  run(512, 1024, 1024, 16, 3.f, 1.f);        // R forward
  run(2048, 1024, 512, 16, 3.f, 1.f);        // D3 forward
  run(8192, 512, 256, 16, 3.f, 1.f);         // D2 forward
  run(32768, 256, 128, 16, 3.f, 1.f);        // D1 forward
  run(512, 1024, 1024, 16, 3e-6f, 4e3f);     // tiny gradients times large weights: the scales do the work
  return 0;
}
