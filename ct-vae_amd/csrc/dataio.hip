// Input side of the hot path: the reference's per-sample transform pipeline
//     transforms.ToTensor() -> transforms.CenterCrop(crop) -> transforms.Resize(size)        (dataset.py:72-80)
// for a whole batch of rows of a uint8 dataset [N][H][W][3] that lives in HBM, written as the NHWC fp32 batch the
// encoder reads.  One thread per output pixel; HBM-bound (a few MB per batch).
//   ToTensor          : u8 -> f32 / 255
//   CenterCrop(crop)  : window origin round((H-crop)/2), round((W-crop)/2) (Python round: half to even); an image
//                       smaller than the crop is zero-padded first with floor((crop-H)/2) rows above (torchvision's
//                       center_crop), i.e. every window pixel outside the image reads 0
//   Resize(size)      : on a tensor = F.interpolate(mode="bilinear", align_corners=False), no antialias:
//                       src = scale*(dst+0.5)-0.5 clamped at 0, scale = crop/size in fp32, neighbours clamped at the
//                       window edge, weights (1-l, l), rows combined after columns -- the order of ATen's
//                       upsample_bilinear2d kernel, so results agree to rounding.
#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

namespace {

__device__ __forceinline__ int round_half_even_div2(int v) {   // Python round(v / 2.0)
  const int q = v >> 1;                                         // floor(v/2) (arithmetic shift)
  return (v & 1) ? ((q & 1) ? q + 1 : q) : q;                   // x.5 -> the even neighbour
}

__global__ __launch_bounds__(256) void crop_resize_u8_kernel(const unsigned char* __restrict__ img,
                                                            const long long* __restrict__ rows, float* __restrict__ out,
                                                            int B, int N, int H, int W, int crop, int S) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)B * S * S) return;
  const int ox = (int)(e % S), oy = (int)((e / S) % S), b = (int)(e / ((long)S * S));
  const long long row = rows[b];
  // window origin in image coordinates (negative when the image is smaller than the crop: zero padding)
  const int top = H >= crop ? round_half_even_div2(H - crop) : -((crop - H) / 2);
  const int left = W >= crop ? round_half_even_div2(W - crop) : -((crop - W) / 2);
  const float scale = (float)crop / (float)S;
  float sy = scale * ((float)oy + 0.5f) - 0.5f, sx = scale * ((float)ox + 0.5f) - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  sx = sx < 0.f ? 0.f : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < crop - 1 ? 1 : 0), x1 = x0 + (x0 < crop - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const bool ok = row >= 0 && row < N;
  const unsigned char* base = img + (ok ? row : 0) * (long)H * W * 3;
  auto px = [&](int wy, int wx, int c) -> float {
    const int iy = top + wy, ix = left + wx;
    if (!ok || (unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return 0.f;
    return (float)base[((long)iy * W + ix) * 3 + c] / 255.f;
  };
  float* dst = out + e * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    dst[c] = hy * (hx * px(y0, x0, c) + lx * px(y0, x1, c)) + ly * (hx * px(y1, x0, c) + lx * px(y1, x1, c));
}

}  // namespace

int launch_crop_resize_u8(const unsigned char* img, const long long* rows, float* out, int B, int N, int H, int W, int crop,
                          int S, hipStream_t st) {
  if (B <= 0 || N <= 0 || H <= 0 || W <= 0 || crop <= 0 || S <= 0) return kErrBadArg;
  const long n = (long)B * S * S;
  ProfScope ps("crop_resize_u8_kernel", st, 0.0, 12.0 * (double)n + 12.0 * (double)n);
  hipLaunchKernelGGL(crop_resize_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, img, rows, out, B, N, H, W, crop,
                     S);
  CTVAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ctvae
