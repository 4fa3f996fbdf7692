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
//                      1/255 at the end; floating point throughout (Pillow rounds to uint8 after each of its two passes);
//   k_color_jitter     torchvision's tensor-path formulas: brightness / contrast / saturation as clamped blends, hue as an
//                      HSV rotation, the four in the drawn order; contrast needs the image's mean grey -> one workgroup
//                      per image, reduction in LDS.
//   hypersim's colour modality is jittered BEFORE the crop, on the whole image (Data_Manager.py:164-171: color_transform, then
//   the shared spatial transform): k_u8_to_f4 unpacks the decoded image into a float4 buffer, k_color_jitter runs on it at
//   full resolution (image sizes from `var`), and k_input_resample reads that buffer instead of the uint8 arena (params[11]).
// All of it is HBM/latency-trivial (a batch is 16 x 256 x 256 pixels, or 16 full frames); written for clarity.
#include "vcg_common.h"

struct ResampleP {
  const unsigned char* arena;
  const float4* fsrc;        // float4 image buffer for the samples with params[11] == 1 (offset = PIXEL index into it), or null
  const int32_t* params;     // [N][16]: arena offset lo, hi, H, W, box y0, x0, h, w (flipped-image coordinates), flip_h, flip_v, filter,
                             //          source (0: uint8 RGB in the arena, offset in bytes; 1: float4 in fsrc, already in [0, 1])
  float* out;                // (N, S, S, 4), channel 3 = 0
  int N, S;
};

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
    *reinterpret_cast<float4*>(p.out + idx * 4) = make_float4(r * norm, g * norm, b * norm, 0.f);
  }
}

__device__ __forceinline__ float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
__device__ __forceinline__ float gray_of(const float4& v) { return 0.299f * v.x + 0.587f * v.y + 0.114f * v.z; }

__device__ __forceinline__ float4 hue_shift(float4 v, float hue) {
  // torchvision _rgb2hsv / _hsv2rgb
  const float r = v.x, g = v.y, b = v.z;
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.f : maxc);
  const float crd = eqc ? 1.f : cr;
  const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
  const float hr = (maxc == r) ? (bc - gc) : 0.f;
  const float hg = (maxc == g && maxc != r) ? (2.f + rc - bc) : 0.f;
  const float hb = (maxc != g && maxc != r) ? (4.f + gc - rc) : 0.f;
  float h = (hr + hg + hb) / 6.f + 1.f;
  h = h - floorf(h);
  h = h + hue;
  h = h - floorf(h);
  const float h6 = h * 6.f;
  float fi = floorf(h6);
  const float f = h6 - fi;
  int i = ((int)fi) % 6;
  if (i < 0) i += 6;
  const float val = maxc;
  const float pp = clamp01(val * (1.f - s)), qq = clamp01(val * (1.f - f * s)), tt = clamp01(val * (1.f - (1.f - f) * s));
  float4 o = v;
  switch (i) {
    case 0: o.x = val; o.y = tt; o.z = pp; break;
    case 1: o.x = qq; o.y = val; o.z = pp; break;
    case 2: o.x = pp; o.y = val; o.z = tt; break;
    case 3: o.x = pp; o.y = qq; o.z = val; break;
    case 4: o.x = tt; o.y = pp; o.z = val; break;
    default: o.x = val; o.y = pp; o.z = qq; break;
  }
  return o;
}

// one workgroup per image; jitter[n][8] = enabled, brightness, contrast, saturation, hue, order code (o0 + 4 o1 + 16 o2 + 64 o3).
// var == null: N images of S x S pixels back to back; else var[n][4] = pixel offset lo, hi, pixel count (whole decoded frames)
__global__ __launch_bounds__(1024) void k_color_jitter(float* __restrict__ img, const float* __restrict__ jitter, int S,
                                                       const int32_t* __restrict__ var) {
  __shared__ double red[1024];
  __shared__ float mean_s;
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* j = jitter + (size_t)n * 8;
  if (j[0] == 0.f) return;                                       // uniform per workgroup
  const float fb = j[1], fc = j[2], fs = j[3], fhue = j[4];
  const int code = (int)j[5];
  float4* px = reinterpret_cast<float4*>(img) + (var ? (((size_t)(uint32_t)var[n * 4 + 1]) << 32 | (size_t)(uint32_t)var[n * 4]) : (size_t)n * S * S);
  const int npx = var ? var[n * 4 + 2] : S * S;
  for (int i = tid; i < npx; i += 1024) {                        // a uint8 PIL image: the resize's overshoot was clipped
    float4 v = px[i];
    v.x = clamp01(v.x); v.y = clamp01(v.y); v.z = clamp01(v.z);
    px[i] = v;
  }
  for (int k = 0; k < 4; ++k) {
    const int op = (code >> (2 * k)) & 3;
    if (op == 1) {                                               // contrast: blend with the mean grey of the CURRENT image
      __syncthreads();
      double s = 0.0;
      for (int i = tid; i < npx; i += 1024) s += (double)gray_of(px[i]);
      red[tid] = s;
      __syncthreads();
      for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
      }
      if (tid == 0) mean_s = (float)(red[0] / npx);
      __syncthreads();
    }
    const float m = mean_s;
    for (int i = tid; i < npx; i += 1024) {
      float4 v = px[i];
      if (op == 0) {
        v.x = clamp01(fb * v.x); v.y = clamp01(fb * v.y); v.z = clamp01(fb * v.z);
      } else if (op == 1) {
        v.x = clamp01(fc * v.x + (1.f - fc) * m); v.y = clamp01(fc * v.y + (1.f - fc) * m); v.z = clamp01(fc * v.z + (1.f - fc) * m);
      } else if (op == 2) {
        const float gr = gray_of(v);
        v.x = clamp01(fs * v.x + (1.f - fs) * gr); v.y = clamp01(fs * v.y + (1.f - fs) * gr); v.z = clamp01(fs * v.z + (1.f - fs) * gr);
      } else {
        v = hue_shift(v, fhue);
      }
      px[i] = v;
    }
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
