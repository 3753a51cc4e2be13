// GPU Look-Twice tail (engine/runner/loop_UCOD_DPL.py:362-384 and :346-352):
//   * ucod_ccl8_components  -- 8-connected component labelling ON THE DEVICE (union-find with the smaller linear index as the
//                              root) and a compact table of components (root, area, xmin, xmax, ymin, ymax, order): `order` = the raster
//                              index of the component's first 2 x 2 block, i.e. the key by which cv2.connectedComponents numbers labels
//                              (look_twice.hip: ucod_ccl8_host); only that table (a few dozen ints) goes to the host,
//                              where the reference's float box arithmetic runs unchanged.  Replaces a 268 KB mask D2H + host CCL.
//   * ucod_paste_resized_u8 -- for every box: Pillow-BICUBIC (8-bit, 22-bit fixed point, horizontal then vertical pass with
//                              the intermediate rounded to uint8) resize of a small refined mask to the box size, pasted into
//                              the canvas; boxes in order, later boxes overwrite earlier ones (:346-352).  Bit-identical
//                              to PIL's resize + paste.
#include "common.h"
#include "../../include/ucod_dpl.h"
#include <cmath>
#include <vector>

namespace ucod {
namespace {

__device__ __forceinline__ int uf_find(const int* __restrict__ parent, int i) {
  int p = parent[i];
  while (p != i) {
    i = p;
    p = parent[i];
  }
  return i;
}

// union by smaller root index, lock-free (atomicMin on the larger root's parent, retry with what was found there)
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&parent[b], a);
    if (old == b) return;
    b = old;
  }
}

// Run-based labelling.  A pixel-per-thread union-find (every foreground pixel unions with its four scanned neighbours, every pixel adds
// itself to its component's statistics) costs 0.6 ms on a 518 x 518 blob mask: long find chains across solid regions and ~10^5 atomics
// on one statistics row.  Horizontal runs remove both: a row scan points every foreground pixel at the first pixel of its run, only
// run starts are ever union-find roots, one union is issued per pair of 8-adjacent runs, and one set of atomics per run.
// (a) one wave per row, 64 columns per step: latest run start at or before x = running max of (start ? x : -1).
__global__ __launch_bounds__(256) void ccl_runs_kernel(const unsigned char* __restrict__ mask, int* __restrict__ parent, int H, int W) {
  const int lane = threadIdx.x & 63, y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y >= H) return;
  const unsigned char* row = mask + (size_t)y * W;
  int carry = -1;                                         // latest run start seen in earlier chunks of this row
  bool prev_fg = false;                                   // column x0 - 1
  for (int x0 = 0; x0 < W; x0 += 64) {
    const int x = x0 + lane;
    const bool fg = x < W && row[x] != 0;
    const int left = __shfl_up((int)fg, 1, 64);
    const bool before = lane == 0 ? prev_fg : (left != 0);
    int v = (fg && !before) ? x : -1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o, 64);
      if (lane >= o) v = v > u ? v : u;
    }
    v = v > carry ? v : carry;
    if (x < W) parent[(size_t)y * W + x] = fg ? y * W + v : -1;
    carry = __shfl(v, 63, 64);
    prev_fg = __shfl((int)fg, 63, 64) != 0;
  }
}

// (b) one union per pair of 8-adjacent runs of rows y-1 and y: at the first column where they overlap (where either run starts), or
//     through the diagonal when they only touch corner to corner
__global__ __launch_bounds__(256) void ccl_merge_kernel(const unsigned char* __restrict__ mask, int* __restrict__ parent, int H, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W || !mask[i]) return;
  const int y = i / W, x = i - y * W;
  if (y == 0) return;
  const bool left = x > 0 && mask[i - 1], right = x + 1 < W && mask[i + 1];
  const bool up = mask[i - W] != 0, upl = x > 0 && mask[i - W - 1], upr = x + 1 < W && mask[i - W + 1];
  if (up) {
    if (!left || !upl) uf_union(parent, i, i - W);
  } else {
    if (upl && !left) uf_union(parent, i, i - W - 1);
    if (upr && !right) uf_union(parent, i, i - W + 1);
  }
}

// stats[root] = {area, xmin, xmax, ymin, ymax, first 2 x 2 block}: the LAST pixel of every run adds the whole run
__global__ __launch_bounds__(256) void ccl_stats_kernel(const unsigned char* __restrict__ mask, int* __restrict__ parent, int* __restrict__ stats, int H,
                                                        int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W || !mask[i]) return;
  const int y = i / W, x = i - y * W;
  if (x + 1 < W && mask[i + 1]) return;                   // not the end of its run
  const bool is_start = x == 0 || !mask[i - 1];
  const int start = is_start ? i : parent[i];             // non-start pixels keep pointing at their run start for ever
  const int r = uf_find(parent, start);
  int* s = stats + (size_t)r * 6;
  atomicAdd(&s[0], i - start + 1);
  atomicMin(&s[1], start - y * W);
  atomicMax(&s[2], x);
  atomicMin(&s[3], y);
  atomicMax(&s[4], y);
  atomicMin(&s[5], (y >> 1) * ((W + 1) >> 1) + ((start - y * W) >> 1));       // the run's first block, in block-raster order
}

__global__ __launch_bounds__(256) void ccl_stats_init_kernel(const int* __restrict__ parent, int* __restrict__ stats, int n, int W, int H) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || parent[i] != i) return;                         // only roots own a stats row (parent is final after merge)
  int* s = stats + (size_t)i * 6;
  s[0] = 0;
  s[1] = W;
  s[2] = -1;
  s[3] = H;
  s[4] = -1;
  s[5] = 0x7FFFFFFF;
}

__global__ __launch_bounds__(256) void ccl_compact_kernel(const int* __restrict__ parent, const int* __restrict__ stats, int n, int* __restrict__ count,
                                                          int* __restrict__ table, int cap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || parent[i] != i) return;
  const int slot = atomicAdd(count, 1);
  if (slot >= cap) return;
  int* t = table + (size_t)slot * 7;
  t[0] = i;
#pragma unroll
  for (int k = 0; k < 6; ++k) t[1 + k] = stats[(size_t)i * 6 + k];
}

// ----------------------------------------------------------------------------------------------- Pillow resample (as look_twice.hip)
constexpr int PRECISION_BITS = 32 - 8 - 2;

double bicubic_filter(double x) {
  const double a = -0.5;
  x = x < 0 ? -x : x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

int bicubic_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> w(ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      w[x] = bicubic_filter((x + xmin - center + 0.5) * ss);
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

__device__ __forceinline__ unsigned char clip8(long long v) {
  v >>= PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct PasteMeta { int x, y, w, h, kh, kv, off_bh, off_kh, off_bv, off_kv; };

// one launch per box (boxes are pasted in order); thread = one destination pixel of the box that lies inside the canvas.
// vertical tap t needs the horizontally resampled source row ymin+t at column ox: recomputed per tap (sources are 37x37).
__global__ __launch_bounds__(256) void paste_box_kernel(const unsigned char* __restrict__ src, int sh, int sw, PasteMeta m, const int* __restrict__ tab,
                                                        unsigned char* __restrict__ canvas, int CH, int CW) {
  const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
  if (ox >= m.w) return;
  const int cx = m.x + ox, cy = m.y + oy;
  if (cx < 0 || cx >= CW || cy < 0 || cy >= CH) return;
  const int xmin = tab[m.off_bh + 2 * ox], nx = tab[m.off_bh + 2 * ox + 1];
  const int* kx = tab + m.off_kh + ox * m.kh;
  const int ymin = tab[m.off_bv + 2 * oy], ny = tab[m.off_bv + 2 * oy + 1];
  const int* ky = tab + m.off_kv + oy * m.kv;
  long long acc = 1LL << (PRECISION_BITS - 1);
  for (int t = 0; t < ny; ++t) {
    const unsigned char* row = src + (size_t)(ymin + t) * sw;
    long long h = 1LL << (PRECISION_BITS - 1);
    for (int u = 0; u < nx; ++u) h += (long long)row[xmin + u] * kx[u];
    acc += (long long)clip8(h) * ky[t];
  }
  canvas[(size_t)cy * CW + cx] = clip8(acc);
}

}  // namespace
}  // namespace ucod

using namespace ucod;

extern "C" size_t ucod_ccl8_workspace_bytes(int H, int W) { return H > 0 && W > 0 ? (size_t)H * W * 7 * sizeof(int) + 256 : 0; }

extern "C" int ucod_ccl8_components(const uint8_t* mask, int H, int W, int32_t* table, int capacity, int32_t* count, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  if (!mask || !table || !count || !workspace || H <= 0 || W <= 0 || capacity <= 0 || (long)H * W > (1L << 30)) return UCOD_EINVAL;
  if (workspace_bytes < ucod_ccl8_workspace_bytes(H, W)) return UCOD_ENOMEM;
  const int n = H * W;
  int* parent = (int*)workspace;
  int* stats = parent + n;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(cdiv(n, 256)), block(256);
  { const hipError_t e = hipMemsetAsync(count, 0, sizeof(int), s); if (e != hipSuccess) return (int)e; }
  hipLaunchKernelGGL(ccl_runs_kernel, dim3(cdiv(H, 4)), block, 0, s, mask, parent, H, W);
  hipLaunchKernelGGL(ccl_merge_kernel, grid, block, 0, s, mask, parent, H, W);
  hipLaunchKernelGGL(ccl_stats_init_kernel, grid, block, 0, s, parent, stats, n, W, H);
  hipLaunchKernelGGL(ccl_stats_kernel, grid, block, 0, s, mask, parent, stats, H, W);
  hipLaunchKernelGGL(ccl_compact_kernel, grid, block, 0, s, parent, stats, n, count, table, capacity);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" size_t ucod_paste_workspace_bytes(int nbox, int max_w, int max_h, int sh, int sw) {
  if (nbox <= 0 || max_w <= 0 || max_h <= 0 || sh <= 0 || sw <= 0) return 0;
  auto ks = [](int in, int out) { const double sc = (double)in / out; return (int)std::ceil(2.0 * (sc < 1.0 ? 1.0 : sc)) * 2 + 1; };
  // worst case taps: the smallest destination (1 pixel) has the widest support
  const size_t per_box = (size_t)max_w * (2 + ks(sw, 1)) + (size_t)max_h * (2 + ks(sh, 1));
  return (size_t)nbox * per_box * sizeof(int) + 256;
}

// masks of SEVERAL canvases in one call (batched Look-Twice validation): mask i is pasted onto canvas box_canvas_host[i] of `canvases` [ncanvas][CH][CW], in
// the order given (the boxes of one canvas keep their order: later ones overwrite earlier ones, loop_UCOD_DPL.py:346-352).  box_canvas_host NULL = one canvas.
extern "C" int ucod_paste_resized_u8_multi(const uint8_t* masks, int nbox, int sh, int sw, const int32_t* boxes_host, const int32_t* box_canvas_host, uint8_t* canvases,
                                           int ncanvas, int CH, int CW, void* workspace, size_t workspace_bytes, void* stream) {
  if (!masks || !boxes_host || !canvases || !workspace || nbox <= 0 || sh <= 0 || sw <= 0 || CH <= 0 || CW <= 0 || ncanvas <= 0) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  std::vector<int> tab;
  std::vector<PasteMeta> metas(nbox);
  std::vector<int> b, k;
  for (int i = 0; i < nbox; ++i) {
    PasteMeta& m = metas[i];
    m.x = boxes_host[4 * i];
    m.y = boxes_host[4 * i + 1];
    m.w = boxes_host[4 * i + 2];
    m.h = boxes_host[4 * i + 3];
    if (m.w <= 0 || m.h <= 0) return UCOD_EINVAL;                  // PIL: "height and width must be > 0"
    if (box_canvas_host && (box_canvas_host[i] < 0 || box_canvas_host[i] >= ncanvas)) return UCOD_EINVAL;
    m.kh = bicubic_coeffs(sw, m.w, b, k);
    m.off_bh = (int)tab.size();
    tab.insert(tab.end(), b.begin(), b.end());
    m.off_kh = (int)tab.size();
    tab.insert(tab.end(), k.begin(), k.end());
    m.kv = bicubic_coeffs(sh, m.h, b, k);
    m.off_bv = (int)tab.size();
    tab.insert(tab.end(), b.begin(), b.end());
    m.off_kv = (int)tab.size();
    tab.insert(tab.end(), k.begin(), k.end());
  }
  if (workspace_bytes < tab.size() * sizeof(int)) return UCOD_ENOMEM;
  // pageable host -> device copy of the tables: synchronous with respect to the host buffer, ordered on the stream
  hipError_t e = hipMemcpyAsync(workspace, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);              // `tab` dies at return
  if (e != hipSuccess) return (int)e;
  for (int i = 0; i < nbox; ++i) {
    const PasteMeta& m = metas[i];
    uint8_t* canvas = canvases + (size_t)(box_canvas_host ? box_canvas_host[i] : 0) * CH * CW;
    hipLaunchKernelGGL(paste_box_kernel, dim3(cdiv(m.w, 256), m.h), dim3(256), 0, s, masks + (size_t)i * sh * sw, sh, sw, m, (const int*)workspace, canvas,
                       CH, CW);
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_paste_resized_u8(const uint8_t* masks, int nbox, int sh, int sw, const int32_t* boxes_host, uint8_t* canvas, int CH, int CW,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  return ucod_paste_resized_u8_multi(masks, nbox, sh, sw, boxes_host, nullptr, canvas, 1, CH, CW, workspace, workspace_bytes, stream);
}
