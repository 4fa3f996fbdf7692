// Device-side input transforms: decoded uint8 HWC images -> the fp32 NHWC (pitch 4) batches the networks take.
//
// Replaces, on the GPU, the torchvision pipelines the reference builds on the host for every sample
// (/root/reference/train.py:184-190 hypersim, :248-262 maps, :309-319 summer2winter):
//     RandomHorizontalFlip / RandomVerticalFlip -> RandomResizedCrop(S, scale (0.33, 1), ratio (1, 1), BICUBIC)
//     -> [ColorJitter] -> ToTensor            and  Resize((S, S)) -> ToTensor  for test data.
// The random DRAWS stay on the host (a dozen numbers per image, input_pipeline.py); everything that touches pixels runs
// here, on a side stream, so that the step loop is never host-bound (a PIL + torchvision loader does ~100-200 images/s
// per core; one MI355X consumes 330 image pairs/s at the headline config and 1 500 images/s as an autoencoder).
//
// Semantics (restated in oracle/input_oracle.py, which cites the published algorithms):
//   k_input_resample   Pillow's convolution resize of the CROP (antialiased: the filter is stretched by max(scale, 1)),
//                      BICUBIC (Keys a = -0.5, support 2) or BILINEAR, flips folded into the tap addresses, ToTensor's
//                      1/255 at the end; floating point through both passes (Pillow rounds to uint8 after each of its two),
//                      then — params[12] — rounded and clipped to the uint8 grid the reference's PIL image lives on;
//   k_color_jitter     ColorJitter as the reference runs it, on a uint8 PIL image (train.py:316; torchvision's _functional_pil
//                      path): brightness / contrast / saturation = PIL.ImageEnhance, i.e. Image.blend against black / the
//                      rounded mean of the L image / the L image, in C float, truncated to uint8 after every op; hue = a
//                      wrapping integer shift of H in Pillow's uint8 HSV conversion (Convert.c, restated with float where the C
//                      has float and double where a double literal promotes); the four in the drawn order; contrast needs the
//                      image's mean -> one workgroup per image, integer reduction in LDS.  Bit-exact against the oracle's
//                      restatement, which is pinned on Pillow itself (tests/test_input_pipeline.py).
//   hypersim's colour modality is jittered BEFORE the crop, on the whole image (Data_Manager.py:164-171: color_transform, then
//   the shared spatial transform): k_u8_to_f4 unpacks the decoded image into a float4 buffer, k_color_jitter runs on it at
//   full resolution (image sizes from `var`), and k_input_resample reads that buffer instead of the uint8 arena (params[11]).
// All of it is HBM/latency-trivial (a batch is 16 x 256 x 256 pixels, or 16 full frames); written for clarity.
#include "vcg_common.h"

struct ResampleP {
  const unsigned char* arena;
  const float4* fsrc;        // float4 image buffer for the samples with params[11] == 1 (offset = PIXEL index into it), or null
  const int32_t* params;     // [N][16]: arena offset lo, hi, H, W, box y0, x0, h, w (flipped-image coordinates), flip_h, flip_v, filter,
                             //          source (0: uint8 RGB in the arena, offset in bytes; 1: float4 in fsrc, already in [0, 1]),
                             //          quantise (1: round + clip the result to the uint8 grid, as the PIL image the reference resizes)
  float* out;                // (N, S, S, 4), channel 3 = 0
  int N, S;
};

// a value in [0, 1] -> the uint8 level Pillow would store: floor(255 v + 0.5) clipped to 0..255 (as a float holding an integer)
__device__ __forceinline__ float u8_of(float v) {
  const float t = floorf(v * 255.f + 0.5f);
  return t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
}

__device__ __forceinline__ float filt_bicubic(float x) {
  const float a = -0.5f;
  x = fabsf(x);
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}
__device__ __forceinline__ float filt_triangle(float x) {
  x = fabsf(x);
  return x < 1.f ? 1.f - x : 0.f;
}
__device__ __forceinline__ float filt(float x, int kind) { return kind == 0 ? filt_bicubic(x) : filt_triangle(x); }

// taps [t0, t1) in crop coordinates and the centre of output index o
__device__ __forceinline__ void tap_range(int o, int box_len, int S, int kind, float& center, float& fscale, int& t0, int& t1) {
  const float scale = (float)box_len / (float)S;
  fscale = scale > 1.f ? scale : 1.f;
  const float sup = (kind == 0 ? 2.f : 1.f) * fscale;
  center = ((float)o + 0.5f) * scale;
  t0 = (int)(center - sup + 0.5f);
  if (t0 < 0) t0 = 0;
  t1 = (int)(center + sup + 0.5f);
  if (t1 > box_len) t1 = box_len;
}

__global__ __launch_bounds__(256) void k_input_resample(ResampleP p) {
  const int S = p.S;
  const size_t total = (size_t)p.N * S * S;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx / ((size_t)S * S));
    const int rem = (int)(idx - (size_t)n * S * S);
    const int oy = rem / S, ox = rem - oy * S;
    const int32_t* q = p.params + (size_t)n * 16;
    const size_t soff = ((size_t)(uint32_t)q[1]) << 32 | (size_t)(uint32_t)q[0];
    const unsigned char* src = p.arena + soff;
    const bool fl = q[11] == 1;
    const float4* fs = p.fsrc + (fl ? soff : 0);
    const int H = q[2], W = q[3], y0 = q[4], x0 = q[5], bh = q[6], bw = q[7], fh = q[8], fv = q[9], kind = q[10];
    float cy, cx, fsy, fsx;
    int ty0, ty1, tx0, tx1;
    tap_range(oy, bh, S, kind, cy, fsy, ty0, ty1);
    tap_range(ox, bw, S, kind, cx, fsx, tx0, tx1);
    float sx = 0.f, sy = 0.f;
    for (int t = tx0; t < tx1; ++t) sx += filt(((float)t - cx + 0.5f) / fsx, kind);
    for (int t = ty0; t < ty1; ++t) sy += filt(((float)t - cy + 0.5f) / fsy, kind);
    float r = 0.f, g = 0.f, b = 0.f;
    for (int ty = ty0; ty < ty1; ++ty) {
      const float wy = filt(((float)ty - cy + 0.5f) / fsy, kind);
      int yy = y0 + ty;                           // flipped-image row -> source row
      if (fv) yy = H - 1 - yy;
      const unsigned char* row = src + (size_t)yy * W * 3;
      float rr = 0.f, rg = 0.f, rb = 0.f;
      for (int tx = tx0; tx < tx1; ++tx) {
        const float wx = filt(((float)tx - cx + 0.5f) / fsx, kind);
        int xx = x0 + tx;
        if (fh) xx = W - 1 - xx;
        if (fl) {
          const float4 v = fs[(size_t)yy * W + xx];
          rr += wx * v.x; rg += wx * v.y; rb += wx * v.z;
        } else {
          const unsigned char* px = row + (size_t)xx * 3;
          rr += wx * (float)px[0];
          rg += wx * (float)px[1];
          rb += wx * (float)px[2];
        }
      }
      r += wy * rr; g += wy * rg; b += wy * rb;
    }
    const float norm = 1.f / ((fl ? 1.f : 255.f) * (sx != 0.f ? sx : 1.f) * (sy != 0.f ? sy : 1.f));
    float4 o = make_float4(r * norm, g * norm, b * norm, 0.f);
    if (q[12] == 1) {                             // Pillow: clip8(value + 0.5); ToTensor: uint8 / 255
      o.x = __fdiv_rn(u8_of(o.x), 255.f); o.y = __fdiv_rn(u8_of(o.y), 255.f); o.z = __fdiv_rn(u8_of(o.z), 255.f);
    }
    *reinterpret_cast<float4*>(p.out + idx * 4) = o;
  }
}

// ---- PIL-path ColorJitter on integer levels 0..255 held in floats --------------------------------------------------------
// Pillow RGB -> L
__device__ __forceinline__ int pil_L(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
// Image.blend(degenerate, image, alpha) for one channel: C float  t = in1 + alpha * (in2 - in1),  truncated; clipped when alpha
// is outside [0, 1] (Blend.c).  __fmul_rn / __fadd_rn: no fused multiply-add, as the C compiled for plain x86-64.
__device__ __forceinline__ int pil_blend1(int deg, int v, float alpha, bool interp) {
  const float t = __fadd_rn((float)deg, __fmul_rn(alpha, (float)(v - deg)));
  if (interp) return (int)t & 255;                               // (UINT8)(float): truncation (0 <= t <= 255 here)
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
// Pillow Convert.c rgb2hsv_row + torchvision's wrapping H shift + hsv2rgb
__device__ __forceinline__ void pil_hue(int& r, int& g, int& b, int shift) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = (int)((double)h * 255.0); uh = uh < 0 ? 0 : (uh > 255 ? 255 : uh);
    us = (int)((double)s * 255.0); us = us < 0 ? 0 : (us > 255 ? 255 : us);
  }
  uh = (uh + shift) & 255;
  if (us == 0) { r = g = b = uv; return; }
  const double h6 = (double)uh * 6.0 / 255.0;
  const int i = (int)floor(h6);
  const float f = (float)(h6 - (double)(float)i);
  const float fs = (float)((double)us / 255.0);
  const double v = (double)uv;
  int p = (int)floor(v * (1.0 - (double)fs) + 0.5);
  int q = (int)floor(v * (1.0 - (double)__fmul_rn(fs, f)) + 0.5);
  int t = (int)floor(v * (1.0 - (double)fs * (1.0 - (double)f)) + 0.5);
  p = p < 0 ? 0 : (p > 255 ? 255 : p); q = q < 0 ? 0 : (q > 255 ? 255 : q); t = t < 0 ? 0 : (t > 255 ? 255 : t);
  switch (i % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

// one workgroup per image; jitter[n][8] = enabled, brightness, contrast, saturation, hue, order code (o0 + 4 o1 + 16 o2 + 64 o3).
// var == null: N images of S x S pixels back to back; else var[n][4] = pixel offset lo, hi, pixel count (whole decoded frames).
// Pixels come in and go out as floats in [0, 1]; inside they are the uint8 levels of the PIL image the reference jitters.
__global__ __launch_bounds__(1024) void k_color_jitter(float* __restrict__ img, const float* __restrict__ jitter, int S,
                                                       const int32_t* __restrict__ var) {
  __shared__ unsigned long long red[1024];
  __shared__ int mean_s;
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* j = jitter + (size_t)n * 8;
  if (j[0] == 0.f) return;                                       // uniform per workgroup
  const float fb = j[1], fc = j[2], fs = j[3], fhue = j[4];
  const int code = (int)j[5];
  const int shift = (int)(fhue * 255.0) & 255;                   // np.uint8(hue_factor * 255): truncation toward zero, wrapping
  float4* px = reinterpret_cast<float4*>(img) + (var ? (((size_t)(uint32_t)var[n * 4 + 1]) << 32 | (size_t)(uint32_t)var[n * 4]) : (size_t)n * S * S);
  const int npx = var ? var[n * 4 + 2] : S * S;
  for (int i = tid; i < npx; i += 1024) {                        // onto the uint8 grid (a no-op for levels that are already on it)
    float4 v = px[i];
    v.x = u8_of(v.x); v.y = u8_of(v.y); v.z = u8_of(v.z);
    px[i] = v;
  }
  for (int k = 0; k < 4; ++k) {
    const int op = (code >> (2 * k)) & 3;
    if (op == 1) {                                               // contrast: the degenerate image is int(mean(L) + 0.5) of the CURRENT image
      __syncthreads();
      unsigned long long s = 0;
      for (int i = tid; i < npx; i += 1024) {
        const float4 v = px[i];
        s += (unsigned long long)pil_L((int)v.x, (int)v.y, (int)v.z);
      }
      red[tid] = s;
      __syncthreads();
      for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
      }
      if (tid == 0) mean_s = (int)((double)red[0] / (double)npx + 0.5);
      __syncthreads();
    }
    const int m = mean_s;
    const float alpha = op == 0 ? fb : (op == 1 ? fc : fs);
    const bool interp = alpha >= 0.f && alpha <= 1.f;
    for (int i = tid; i < npx; i += 1024) {
      const float4 v = px[i];
      int r = (int)v.x, g = (int)v.y, b = (int)v.z;
      if (op == 0) {
        r = pil_blend1(0, r, alpha, interp); g = pil_blend1(0, g, alpha, interp); b = pil_blend1(0, b, alpha, interp);
      } else if (op == 1) {
        r = pil_blend1(m, r, alpha, interp); g = pil_blend1(m, g, alpha, interp); b = pil_blend1(m, b, alpha, interp);
      } else if (op == 2) {
        const int L = pil_L(r, g, b);
        r = pil_blend1(L, r, alpha, interp); g = pil_blend1(L, g, alpha, interp); b = pil_blend1(L, b, alpha, interp);
      } else {
        pil_hue(r, g, b, shift);
      }
      px[i] = make_float4((float)r, (float)g, (float)b, 0.f);
    }
  }
  for (int i = tid; i < npx; i += 1024) {                        // ToTensor: uint8 / 255
    float4 v = px[i];
    v.x = __fdiv_rn(v.x, 255.f); v.y = __fdiv_rn(v.y, 255.f); v.z = __fdiv_rn(v.z, 255.f);
    px[i] = v;
  }
}

extern "C" int vcg_input_resample(const unsigned char* arena, const float* fsrc, const int32_t* params, float* out, int N, int S,
                                  void* stream) {
  VCG_CHECK_ARG(arena && params && out, "vcg_input_resample: null pointer");
  VCG_CHECK_ARG(N > 0 && S > 0 && S <= 4096, "vcg_input_resample: bad N=%d S=%d", N, S);
  ResampleP p;
  p.arena = arena; p.fsrc = (const float4*)fsrc; p.params = params; p.out = out; p.N = N; p.S = S;
  size_t blocks = ((size_t)N * S * S + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_input_resample, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  VCG_LAUNCH_CHECK("vcg_input_resample");
  return 0;
}

extern "C" int vcg_input_color_jitter(float* img, const float* jitter, int N, int S, void* stream) {
  VCG_CHECK_ARG(img && jitter, "vcg_input_color_jitter: null pointer");
  VCG_CHECK_ARG(N > 0 && S > 0, "vcg_input_color_jitter: bad N=%d S=%d", N, S);
  hipLaunchKernelGGL(k_color_jitter, dim3(N), dim3(1024), 0, (hipStream_t)stream, img, jitter, S, (const int32_t*)nullptr);
  VCG_LAUNCH_CHECK("vcg_input_color_jitter");
  return 0;
}

// decoded uint8 RGB frames -> float4 pixels in [0, 1] (pad 0): frames[n][8] = arena byte offset lo, hi, pixel count, float-buffer
// pixel offset lo, hi
__global__ __launch_bounds__(256) void k_u8_to_f4(const unsigned char* __restrict__ arena, const int32_t* __restrict__ frames,
                                                  float4* __restrict__ fbuf) {
  const int32_t* q = frames + (size_t)blockIdx.y * 8;
  const unsigned char* src = arena + (((size_t)(uint32_t)q[1]) << 32 | (size_t)(uint32_t)q[0]);
  float4* dst = fbuf + (((size_t)(uint32_t)q[4]) << 32 | (size_t)(uint32_t)q[3]);
  const int npx = q[2];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x)
    dst[i] = make_float4(src[3 * (size_t)i] * (1.f / 255.f), src[3 * (size_t)i + 1] * (1.f / 255.f), src[3 * (size_t)i + 2] * (1.f / 255.f), 0.f);
}

// ColorJitter on whole decoded frames, before any crop (hypersim's colour modality): unpack + jitter; the resample then reads
// `fbuf` for these samples (params[11] == 1).  jitter[n][8] as above; var[n][4] = float-buffer pixel offset lo, hi, pixel count.
extern "C" int vcg_input_prejitter(const unsigned char* arena, const int32_t* frames, const float* jitter, const int32_t* var,
                                   float* fbuf, int N, void* stream) {
  VCG_CHECK_ARG(arena && frames && jitter && var && fbuf, "vcg_input_prejitter: null pointer");
  VCG_CHECK_ARG(N > 0, "vcg_input_prejitter: bad N=%d", N);
  hipLaunchKernelGGL(k_u8_to_f4, dim3(256, N), dim3(256), 0, (hipStream_t)stream, arena, frames, (float4*)fbuf);
  hipLaunchKernelGGL(k_color_jitter, dim3(N), dim3(1024), 0, (hipStream_t)stream, fbuf, jitter, 0, var);
  VCG_LAUNCH_CHECK("vcg_input_prejitter");
  return 0;
}
