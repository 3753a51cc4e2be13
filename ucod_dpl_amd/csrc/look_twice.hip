// Look-Twice support (engine/runner/loop_UCOD_DPL.py:326-384):
//   * ucod_ccl8_host        -- 8-connected component labelling (cv2.connectedComponents, :366) on the host, labels numbered as OpenCV numbers them;
//   * ucod_pil_resize_u8_host -- Pillow's 8-bit antialiased resample (Image.resize on the 'L' mask, :350) on the host;
//   * ucod_crop_resize_norm -- batched GPU crop + Pillow-BILINEAR resize to the network input + /255 + ImageNet normalise
//                              (PIL crop + torchvision Resize/ToTensor/Normalize, :282-286,341-342), one launch pair for all boxes.
// The resampler restates Pillow's Resample.c: coefficients in double, normalised, rounded to 22-bit fixed point; horizontal
// pass then vertical pass with the intermediate rounded to uint8 -- so the GPU result is bit-identical to PIL's.
#include "common.h"
#include "../../include/ucod_dpl.h"
#include <cmath>
#include <vector>
#include <algorithm>

namespace ucod {

constexpr int PRECISION_BITS = 32 - 8 - 2;

static inline double bilinear_filter(double x) { x = x < 0 ? -x : x; return x < 1.0 ? 1.0 - x : 0.0; }
static inline double bicubic_filter(double x) {
  const double a = -0.5;
  x = x < 0 ? -x : x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc.  bounds[2*xx] = xmin, bounds[2*xx+1] = count; kk[xx*ksize + x].
static int pil_coeffs(int in_size, double in0, double in1, int out_size, int filter, std::vector<int>& bounds, std::vector<int>& kk) {
  const double support0 = filter == 1 ? 2.0 : 1.0;
  const double scale = (in1 - in0) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = support0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> w(ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      const double arg = (x + xmin - center + 0.5) * ss;
      w[x] = filter == 1 ? bicubic_filter(arg) : bilinear_filter(arg);
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
      const double v = ww != 0.0 ? w[x] / ww : w[x];
      kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return ksize;
}

__host__ __device__ static inline unsigned char clip8(long long v) {
  v >>= PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// ------------------------------------------------------------------------------------------- GPU crop + resize
struct BoxMeta { const unsigned char* img; int H, W; int x, y, w, h, kh, kv, off_bh, off_kh, off_bv, off_kv; };   // (each box names its source image: boxes of many images in one launch pair)

// pass 1: horizontal.  tmp[box][row < h][ox][3] u8.  grid (cdiv(ow*3,256), max_h, nbox)
__global__ __launch_bounds__(256) void crop_hpass_kernel(const BoxMeta* __restrict__ meta,
                                                         const int* __restrict__ tab, unsigned char* __restrict__ tmp, int ow, int max_h) {
  const BoxMeta m = meta[blockIdx.z];
  const unsigned char* __restrict__ img = m.img;
  const int H = m.H, W = m.W;
  const int row = blockIdx.y;
  if (row >= m.h) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ow * 3) return;
  const int ox = i / 3, c = i - ox * 3;
  const int xmin = tab[m.off_bh + 2 * ox], n = tab[m.off_bh + 2 * ox + 1];
  const int* k = tab + m.off_kh + ox * m.kh;
  const int sy = m.y + row;
  long long acc = 1LL << (PRECISION_BITS - 1);
  if (sy >= 0 && sy < H) {
    for (int t = 0; t < n; ++t) {
      const int sx = m.x + xmin + t;                         // PIL crop zero-fills outside the image
      const int v = (sx >= 0 && sx < W) ? img[((size_t)sy * W + sx) * 3 + c] : 0;
      acc += (long long)v * k[t];
    }
  }
  tmp[(((size_t)blockIdx.z * max_h + row) * ow + ox) * 3 + c] = clip8(acc);
}

// pass 2: vertical + ToTensor + Normalize.  out[box][c][oy][ox] f32.  grid (cdiv(ow,256), oh, nbox)
__global__ __launch_bounds__(256) void crop_vpass_kernel(const unsigned char* __restrict__ tmp, const BoxMeta* __restrict__ meta,
                                                         const int* __restrict__ tab, float* __restrict__ out, int oh, int ow, int max_h) {
  const BoxMeta m = meta[blockIdx.z];
  const int oy = blockIdx.y, ox = blockIdx.x * 256 + threadIdx.x;
  if (ox >= ow) return;
  const int ymin = tab[m.off_bv + 2 * oy], n = tab[m.off_bv + 2 * oy + 1];
  const int* k = tab + m.off_kv + oy * m.kv;
  long long acc[3] = {1LL << (PRECISION_BITS - 1), 1LL << (PRECISION_BITS - 1), 1LL << (PRECISION_BITS - 1)};
  for (int t = 0; t < n; ++t) {
    const unsigned char* p = tmp + (((size_t)blockIdx.z * max_h + ymin + t) * ow + ox) * 3;
    const long long kv = k[t];
    acc[0] += p[0] * kv;
    acc[1] += p[1] * kv;
    acc[2] += p[2] * kv;
  }
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)clip8(acc[c]) / 255.0f;          // ToTensor
    out[(((size_t)blockIdx.z * 3 + c) * oh + oy) * ow + ox] = (v - mean[c]) / stdv[c];
  }
}

}  // namespace ucod

using namespace ucod;

extern "C" int ucod_ccl8_host(const uint8_t* mask, int H, int W, int32_t* labels) {
  if (!mask || !labels || H <= 0 || W <= 0) return UCOD_EINVAL;
  std::vector<int> parent(1, 0);
  auto find = [&](int a) {
    while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; }
    return a;
  };
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      int32_t& L = labels[(size_t)y * W + x];
      L = 0;
      if (!mask[(size_t)y * W + x]) continue;
      int best = 0;
      auto join = [&](int n) {
        if (!n) return;
        const int r = find(n);
        if (!best) best = r;
        else if (r != best) { const int lo = r < best ? r : best, hi = r < best ? best : r; parent[hi] = lo; best = lo; }
      };
      if (x > 0) join(labels[(size_t)y * W + x - 1]);
      if (y > 0) {
        if (x > 0) join(labels[(size_t)(y - 1) * W + x - 1]);
        join(labels[(size_t)(y - 1) * W + x]);
        if (x + 1 < W) join(labels[(size_t)(y - 1) * W + x + 1]);
      }
      if (!best) { parent.push_back((int)parent.size()); best = (int)parent.size() - 1; }
      L = best;
    }
  // Final numbering = OpenCV's.  cv2.connectedComponents(connectivity=8) labels 2 x 2 BLOCKS in raster order (Grana's BBDT before 4.5.2,
  // Bolelli's Spaghetti since: modules/imgproc/src/connectedcomponents.cpp), a block that touches nothing labelled so far takes the next
  // provisional label, unions keep the smaller label as root and flattenL() renumbers the roots in increasing order -- so component k is
  // the k-th component in raster order of its first 2 x 2 block (all foreground pixels of a block are mutually 8-adjacent: one component
  // per block).  That differs from the raster order of first PIXELS exactly when a component starts on the odd row of a block row to the
  // left of one that starts on the even row; it decides the order of equal-area boxes in process_preds' stable sort (:382) and with it the
  // paste order of look_twice (:346-352).
  std::vector<int> remap(parent.size(), 0);
  int next = 0;
  for (int by = 0; by < H; by += 2)
    for (int bx = 0; bx < W; bx += 2)
      for (int dy = 0; dy < 2 && by + dy < H; ++dy)
        for (int dx = 0; dx < 2 && bx + dx < W; ++dx) {
          const int l = labels[(size_t)(by + dy) * W + bx + dx];
          if (!l) continue;
          const int r = find(l);
          if (!remap[r]) remap[r] = ++next;
        }
  for (size_t i = 0; i < (size_t)H * W; ++i)
    if (labels[i]) labels[i] = remap[find(labels[i])];
  return next + 1;                                           // cv2 convention: background counts as label 0
}

extern "C" int ucod_pil_resize_u8_host(const uint8_t* src, int h, int w, uint8_t* dst, int oh, int ow, int filter) {
  if (!src || !dst || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || (filter != 0 && filter != 1)) return UCOD_EINVAL;
  std::vector<int> bh, kh, bv, kv;
  std::vector<uint8_t> tmp;
  const uint8_t* cur = src;
  int cw = w;
  if (ow != w) {
    const int ks = pil_coeffs(w, 0.0, (double)w, ow, filter, bh, kh);
    tmp.resize((size_t)h * ow);
    for (int y = 0; y < h; ++y)
      for (int xx = 0; xx < ow; ++xx) {
        long long acc = 1LL << (PRECISION_BITS - 1);
        const int xmin = bh[2 * xx], n = bh[2 * xx + 1];
        for (int t = 0; t < n; ++t) acc += (long long)src[(size_t)y * w + xmin + t] * kh[(size_t)xx * ks + t];
        tmp[(size_t)y * ow + xx] = clip8(acc);
      }
    cur = tmp.data();
    cw = ow;
  }
  if (oh != h) {
    const int ks = pil_coeffs(h, 0.0, (double)h, oh, filter, bv, kv);
    for (int yy = 0; yy < oh; ++yy) {
      const int ymin = bv[2 * yy], n = bv[2 * yy + 1];
      for (int x = 0; x < cw; ++x) {
        long long acc = 1LL << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) acc += (long long)cur[(size_t)(ymin + t) * cw + x] * kv[(size_t)yy * ks + t];
        dst[(size_t)yy * ow + x] = clip8(acc);
      }
    }
  } else {
    std::copy(cur, cur + (size_t)oh * ow, dst);
  }
  return UCOD_OK;
}

// workspace: [meta nbox*sizeof(BoxMeta) rounded to 256] [tables int32] [tmp u8 nbox*max_h*ow*3]
static size_t tables_ints(int nbox, int max_dim, int oh, int ow);
extern "C" size_t ucod_crop_workspace_bytes(int nbox, int max_crop_h, int max_crop_w, int oh, int ow);
static size_t tables_ints(int nbox, int max_dim, int oh, int ow) {
  const int kmax = (int)std::ceil(std::max(1.0, (double)max_dim / std::min(oh, ow))) * 2 + 1;
  return (size_t)nbox * ((size_t)ow * (2 + kmax) + (size_t)oh * (2 + kmax));
}
extern "C" size_t ucod_crop_workspace_bytes(int nbox, int max_crop_h, int max_crop_w, int oh, int ow) {
  const size_t meta = ((size_t)nbox * sizeof(BoxMeta) + 255) / 256 * 256;
  const size_t tabs = (tables_ints(nbox, std::max(max_crop_h, max_crop_w), oh, ow) * 4 + 255) / 256 * 256;
  return meta + tabs + (size_t)nbox * max_crop_h * ow * 3;
}

// boxes of SEVERAL source images in one launch pair (batched Look-Twice validation, BASELINE configs[3]): box i is cut from image box_image_host[i]
extern "C" int ucod_crop_resize_norm_multi(const uint8_t* const* imgs_host, const int32_t* hw_host, int nimg, const int32_t* box_image_host, const int32_t* boxes_host,
                                           int nbox, float* out, int oh, int ow, void* workspace, size_t workspace_bytes, void* stream) {
  if (!imgs_host || !hw_host || !box_image_host || !boxes_host || !out || !workspace || nimg <= 0 || nbox <= 0 || oh <= 0 || ow <= 0) return UCOD_EINVAL;
  for (int i = 0; i < nimg; ++i)
    if (!imgs_host[i] || hw_host[2 * i] <= 0 || hw_host[2 * i + 1] <= 0) return UCOD_EINVAL;
  int max_h = 0, max_w = 0;
  for (int i = 0; i < nbox; ++i) {
    if (boxes_host[4 * i + 2] <= 0 || boxes_host[4 * i + 3] <= 0 || box_image_host[i] < 0 || box_image_host[i] >= nimg) return UCOD_EINVAL;
    max_w = std::max(max_w, boxes_host[4 * i + 2]);
    max_h = std::max(max_h, boxes_host[4 * i + 3]);
  }
  if (workspace_bytes < ucod_crop_workspace_bytes(nbox, max_h, max_w, oh, ow)) return UCOD_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  std::vector<BoxMeta> meta(nbox);
  std::vector<int> tab;
  std::vector<int> b, k;
  for (int i = 0; i < nbox; ++i) {
    BoxMeta& m = meta[i];
    m.img = imgs_host[box_image_host[i]];
    m.H = hw_host[2 * box_image_host[i]];
    m.W = hw_host[2 * box_image_host[i] + 1];
    m.x = boxes_host[4 * i];
    m.y = boxes_host[4 * i + 1];
    m.w = boxes_host[4 * i + 2];
    m.h = boxes_host[4 * i + 3];
    m.kh = pil_coeffs(m.w, 0.0, (double)m.w, ow, 0, b, k);
    m.off_bh = (int)tab.size();
    tab.insert(tab.end(), b.begin(), b.end());
    m.off_kh = (int)tab.size();
    tab.insert(tab.end(), k.begin(), k.end());
    m.kv = pil_coeffs(m.h, 0.0, (double)m.h, oh, 0, b, k);
    m.off_bv = (int)tab.size();
    tab.insert(tab.end(), b.begin(), b.end());
    m.off_kv = (int)tab.size();
    tab.insert(tab.end(), k.begin(), k.end());
  }
  const size_t meta_bytes = ((size_t)nbox * sizeof(BoxMeta) + 255) / 256 * 256;
  const size_t tab_cap = (tables_ints(nbox, std::max(max_h, max_w), oh, ow) * 4 + 255) / 256 * 256;
  if (tab.size() * 4 > tab_cap) return UCOD_ENOMEM;
  char* ws = (char*)workspace;
  BoxMeta* dmeta = (BoxMeta*)ws;
  int* dtab = (int*)(ws + meta_bytes);
  unsigned char* tmp = (unsigned char*)(ws + meta_bytes + tab_cap);
  hipError_t e = hipMemcpyAsync(dmeta, meta.data(), nbox * sizeof(BoxMeta), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);          // host tables are stack-owned: make the copies complete before returning
  if (e != hipSuccess) return (int)e;
  UCOD_PROF(PROF_CROP, s);
  hipLaunchKernelGGL(crop_hpass_kernel, dim3(cdiv((long)ow * 3, 256), max_h, nbox), dim3(256), 0, s, dmeta, dtab, tmp, ow, max_h);
  hipLaunchKernelGGL(crop_vpass_kernel, dim3(cdiv(ow, 256), oh, nbox), dim3(256), 0, s, tmp, dmeta, dtab, out, oh, ow, max_h);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_crop_resize_norm(const uint8_t* img, int H, int W, const int32_t* boxes_host, int nbox, float* out, int oh, int ow,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  if (!img || !boxes_host || nbox <= 0 || H <= 0 || W <= 0) return UCOD_EINVAL;
  const int32_t hw[2] = {H, W};
  const std::vector<int32_t> which((size_t)nbox, 0);
  return ucod_crop_resize_norm_multi(&img, hw, 1, which.data(), boxes_host, nbox, out, oh, ow, workspace, workspace_bytes, stream);
}
