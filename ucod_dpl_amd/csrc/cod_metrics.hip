// COD measures of engine/utils/metrics/metric.py::statistics on the device ("COD metrics vectorised").
//
// One call prices a batch of same-sized (prediction, ground-truth) pairs and writes one record of doubles per image: MAE (:187-207),
// ACC (:139-159), IoU (:161-185), S-measure (:209-313), adaptive + 256-threshold E-measure (:315-435), adaptive + 256-threshold
// F-measure with its precision / recall curves (:437-500), weighted F-measure (:503-560).  The reference computes each measure with
// numpy float64 on the host, seven passes of temporaries per image; here every per-pixel quantity is formed once per pass in
// registers (float64 throughout, like the reference) and reduced by a fixed tree, so results are deterministic and agree with the
// reference to summation-order rounding (tests: 1e-9).
//
// Passes (1024-thread workgroups over up to 64 pixel chunks per image, chunk partials combined in chunk order):
//   1  min / max of prediction and ground truth (the min-max normalisation of _prepare_data :125-133);
//   2  sums over pixels of everything that needs only p and g: MAE, ACC, IoU, the object-similarity moments, the centroid, and the
//      two 256-bin histograms of uint8(p * 255) inside / outside the ground truth (integer LDS atomics);
//   3  (needs the mean of p and the centroid) adaptive-threshold counts and the raw moments of the four S-measure quadrants;
//   4a per row: nearest foreground column to the left / right of every pixel;  4b per pixel: exact Euclidean distance to the nearest
//      foreground pixel by walking rows outward (ties: smallest column, then smallest row -- what
//      scipy.ndimage.distance_transform_edt(return_indices=True) returns) and the error |p - g| at that pixel;
//   4c 7x7 Gaussian (sigma 5, zero padded, taps in raster order) of that error, pixel weights, the two weighted sums;
//   5  one 256-thread workgroup per image: the threshold curves from the cumulative histograms and every scalar formula.
#include <cmath>
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {
namespace {

constexpr double EPS = 2.220446049250313e-16;            // np.spacing(1)
constexpr int NS2 = 12, NS3 = 18;                        // doubles reduced by pass 2 / pass 3

struct Norm {                                            // _prepare_data: how to turn the raw inputs into p (float64) and g (bool)
  double pmin, pden;                                     // p = (v - pmin) / pden, a true division as in the reference (not a reciprocal multiply)
  double gmin, gden;
  bool p_const, g_const;
};

__device__ __forceinline__ Norm load_norm(const float* mm) {
  Norm n;
  n.pmin = (double)mm[0];
  n.pden = (double)mm[1] - (double)mm[0];
  n.gmin = (double)mm[2];
  n.gden = (double)mm[3] - (double)mm[2];
  n.p_const = mm[1] == mm[0];
  n.g_const = mm[3] == mm[2];
  return n;
}
__device__ __forceinline__ double norm_p(const Norm& n, float v) { return n.p_const ? trunc((double)v) : ((double)v - n.pmin) / n.pden; }
__device__ __forceinline__ bool norm_g(const Norm& n, float v) { return (n.g_const ? (double)v : ((double)v - n.gmin) / n.gden) > 0.5; }

// fixed-tree block reduction of K doubles per thread (blockDim.x = 1024): lane tree, then wave 0 over the 16 wave partials
template <int K>
__device__ __forceinline__ void block_reduce(double (&v)[K], double* smem /* 16 * K */, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double x = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    if (lane == 0) smem[wave * K + k] = x;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double x = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) x += smem[w * K + threadIdx.x];
    out[threadIdx.x] = x;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void cod_minmax_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int n, float* __restrict__ mm) {
  const float* p = pred + (size_t)blockIdx.x * n;
  const float* g = gt + (size_t)blockIdx.x * n;
  float v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    v[0] = fminf(v[0], p[i]); v[1] = fmaxf(v[1], p[i]);
    v[2] = fminf(v[2], g[i]); v[3] = fmaxf(v[3], g[i]);
  }
  __shared__ float sm[16 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float x = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float y = __shfl_down(x, o, 64); x = (k & 1) ? fmaxf(x, y) : fminf(x, y); }
    if (lane == 0) sm[wave * 4 + k] = x;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float x = sm[threadIdx.x];
    for (int w = 1; w < 16; ++w) x = (threadIdx.x & 1) ? fmaxf(x, sm[w * 4 + threadIdx.x]) : fminf(x, sm[w * 4 + threadIdx.x]);
    mm[blockIdx.x * 4 + threadIdx.x] = x;
  }
}

// The per-pixel passes run on gridDim.x chunks per image (chunk c owns pixels [c*per, (c+1)*per)): each workgroup reduces its chunk
// with the fixed tree and writes one partial; cod_combine_kernel then adds the chunk partials in chunk order.  Deterministic, and a
// 1024 x 1024 map uses 64 CUs instead of one.
__device__ __forceinline__ void chunk_range(int n, int& lo, int& hi) {
  const int per = (n + gridDim.x - 1) / gridDim.x;
  lo = blockIdx.x * per;
  hi = lo + per < n ? lo + per : n;
}

__global__ void cod_combine_kernel(const double* __restrict__ part, int chunks, int K, double* __restrict__ out) {
  const int b = blockIdx.x, k = threadIdx.x;
  if (k >= K) return;
  double x = 0.0;
  for (int c = 0; c < chunks; ++c) x += part[((size_t)b * chunks + c) * K + k];
  out[(size_t)b * K + k] = x;
}

__global__ __launch_bounds__(1024) void cod_pass2_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W,
                                                         const float* __restrict__ mm, double* __restrict__ S2, unsigned* __restrict__ hist) {
  __shared__ double red[16 * NS2];
  __shared__ unsigned h[512];
  const int n = H * W, b = blockIdx.y;
  int lo, hi;
  chunk_range(n, lo, hi);
  const Norm nm = load_norm(mm + b * 4);
  for (int i = threadIdx.x; i < 512; i += blockDim.x) h[i] = 0;
  __syncthreads();
  double s[NS2];
#pragma unroll
  for (int k = 0; k < NS2; ++k) s[k] = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const double p = norm_p(nm, pred[(size_t)b * n + i]);
    const bool g = norm_g(nm, gt[(size_t)b * n + i]);
    const double gf = g ? 1.0 : 0.0;
    s[0] += gf;
    s[1] += p;
    s[2] += fabs(p - gf);
    s[3] += (p == gf) ? 1.0 : 0.0;
    s[4] += (p != 0.0 && g) ? 1.0 : 0.0;
    s[5] += (p != 0.0 || g) ? 1.0 : 0.0;
    if (g) { s[6] += p; s[7] += p * p; s[10] += (double)(i / W); s[11] += (double)(i % W); }
    else { const double q = 1.0 - p; s[8] += q; s[9] += q * q; }
    atomicAdd(&h[(g ? 0 : 256) + (int)(unsigned char)(p * 255.0)], 1u);
  }
  block_reduce<NS2>(s, red, S2 + ((size_t)b * gridDim.x + blockIdx.x) * NS2);
  for (int i = threadIdx.x; i < 512; i += blockDim.x)
    if (h[i]) atomicAdd(&hist[(size_t)b * 512 + i], h[i]);             // integer adds: order does not matter
}

// centroid of the ground truth (:259-268): np.round is round-half-even; an empty mask takes the image centre
__device__ __forceinline__ void centroid(const double* S2, int H, int W, int& cx, int& cy) {
  if (S2[0] == 0.0) { cx = (int)rint(W / 2.0) + 1; cy = (int)rint(H / 2.0) + 1; }
  else { cy = (int)rint(S2[10] / S2[0]) + 1; cx = (int)rint(S2[11] / S2[0]) + 1; }
}

__global__ __launch_bounds__(1024) void cod_pass3_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W,
                                                         const float* __restrict__ mm, const double* __restrict__ S2, double* __restrict__ S3) {
  __shared__ double red[16 * NS3];
  const int n = H * W, b = blockIdx.y;
  int lo, hi;
  chunk_range(n, lo, hi);
  const Norm nm = load_norm(mm + b * 4);
  const double* s2 = S2 + (size_t)b * NS2;
  const double thr = fmin(2.0 * (s2[1] / (double)n), 1.0);            // _get_adaptive_threshold (:135-136)
  int cx, cy;
  centroid(s2, H, W, cx, cy);
  double s[NS3];
#pragma unroll
  for (int k = 0; k < NS3; ++k) s[k] = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const double p = norm_p(nm, pred[(size_t)b * n + i]);
    const bool g = norm_g(nm, gt[(size_t)b * n + i]);
    if (p >= thr) s[g ? 0 : 1] += 1.0;
    const int y = i / W, x = i - y * W;
    const int q = (y < cy ? 0 : 2) + (x < cx ? 0 : 1);                 // LT, RT, LB, RB
    const double gf = g ? 1.0 : 0.0;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
      if (q == qq) { s[2 + 4 * qq] += p; s[3 + 4 * qq] += p * p; s[4 + 4 * qq] += gf; s[5 + 4 * qq] += p * gf; }
  }
  block_reduce<NS3>(s, red, S3 + ((size_t)b * gridDim.x + blockIdx.x) * NS3);
}

// ---- weighted F-measure -------------------------------------------------------------------------------------------------
__global__ void cod_rowscan_kernel(const float* __restrict__ gt, int H, int W, const float* __restrict__ mm, int* __restrict__ Lc, int* __restrict__ Rc) {
  const int b = blockIdx.y, y = blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= H) return;
  const Norm nm = load_norm(mm + b * 4);
  const float* g = gt + ((size_t)b * H + y) * W;
  int* l = Lc + ((size_t)b * H + y) * W;
  int* r = Rc + ((size_t)b * H + y) * W;
  int last = -1;
  for (int x = 0; x < W; ++x) { if (norm_g(nm, g[x])) last = x; l[x] = last; }
  last = -1;
  for (int x = W - 1; x >= 0; --x) { if (norm_g(nm, g[x])) last = x; r[x] = last; }
}

__global__ __launch_bounds__(256) void cod_edt_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W, const float* __restrict__ mm,
                                                      const int* __restrict__ Lc, const int* __restrict__ Rc, double* __restrict__ dist, double* __restrict__ Et) {
  const int b = blockIdx.y, n = H * W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Norm nm = load_norm(mm + b * 4);
  const float* pb = pred + (size_t)b * n;
  const float* gb = gt + (size_t)b * n;
  const int y = i / W, x = i - y * W;
  auto err = [&](int j) { return fabs(norm_p(nm, pb[j]) - (norm_g(nm, gb[j]) ? 1.0 : 0.0)); };
  if (norm_g(nm, gb[i])) { dist[(size_t)b * n + i] = 0.0; Et[(size_t)b * n + i] = err(i); return; }
  long best = 0x7fffffffffffffffL;
  int br = -1, bc = -1;
  const int* lb = Lc + (size_t)b * n;
  const int* rb = Rc + (size_t)b * n;
  auto consider = [&](int r, int c, long dy2) {
    if (c < 0) return;
    const long dx = c - x, d2 = dy2 + dx * dx;
    if (d2 < best || (d2 == best && (c < bc || (c == bc && r < br)))) { best = d2; br = r; bc = c; }
  };
  for (int dy = 0; dy < H; ++dy) {
    const long dy2 = (long)dy * dy;
    if (dy2 > best) break;                                             // equal distances still have to be seen for the tie rule
    const int up = y - dy, dn = y + dy;
    if (up < 0 && dn >= H) break;
    if (up >= 0) { consider(up, lb[up * W + x], dy2); consider(up, rb[up * W + x], dy2); }
    if (dy != 0 && dn < H) { consider(dn, lb[dn * W + x], dy2); consider(dn, rb[dn * W + x], dy2); }
  }
  if (br < 0) { dist[(size_t)b * n + i] = 0.0; Et[(size_t)b * n + i] = err(i); return; }      // no foreground at all: the measure is defined as 0
  dist[(size_t)b * n + i] = sqrt((double)best);
  Et[(size_t)b * n + i] = err(br * W + bc);
}

struct Gauss7 { double k[49]; };

__global__ __launch_bounds__(1024) void cod_wfm_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W, const float* __restrict__ mm,
                                                       const double* __restrict__ dist, const double* __restrict__ Et, Gauss7 gk, double* __restrict__ T) {
  __shared__ double red[16 * 2];
  const int n = H * W, b = blockIdx.y;
  int lo, hi;
  chunk_range(n, lo, hi);
  const Norm nm = load_norm(mm + b * 4);
  const double* et = Et + (size_t)b * n;
  const double* ds = dist + (size_t)b * n;
  const double c = log(0.5) / 5.0;
  double s[2] = {0.0, 0.0};
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const int y = i / W, x = i - y * W;
    double ea = 0.0;
    for (int dy = 0; dy < 7; ++dy) {
      const int yy = y + dy - 3;
      for (int dx = 0; dx < 7; ++dx) {
        const int xx = x + dx - 3;
        const double v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? et[yy * W + xx] : 0.0;
        ea += gk.k[dy * 7 + dx] * v;                                   // raster order over the 49 taps, zero padded (scipy convolve, mode constant)
      }
    }
    if (nm.p_const) ea = trunc(ea);                                    // constant prediction: the reference's arrays are integer there
    const bool g = norm_g(nm, gt[(size_t)b * n + i]);
    const double e = fabs(norm_p(nm, pred[(size_t)b * n + i]) - (g ? 1.0 : 0.0));
    const double m = (g && ea < e) ? ea : e;
    const double w = g ? 1.0 : 2.0 - exp(c * ds[i]);
    s[g ? 0 : 1] += m * w;
  }
  block_reduce<2>(s, red, T + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

// ---- pass 5: curves and scalar formulas ------------------------------------------------------------------------------------
__device__ __forceinline__ double enhanced_alignment(double fg_fg, double fg_bg, double n_fg, double n) {
  const double pred_fg = fg_fg + fg_bg, pred_bg = n - pred_fg;
  double total;
  if (n_fg == 0.0) total = pred_bg;
  else if (n_fg == n) total = pred_fg;
  else {
    const double bg_fg = n_fg - fg_fg, bg_bg = pred_bg - bg_fg;
    const double mp = pred_fg / n, mg = n_fg / n;
    const double dpf = 1.0 - mp, dpb = 0.0 - mp, dgf = 1.0 - mg, dgb = 0.0 - mg;
    auto part = [&](double cnt, double dp, double dg) {
      const double align = 2.0 * (dp * dg) / (dp * dp + dg * dg + EPS);
      return (align + 1.0) * (align + 1.0) / 4.0 * cnt;
    };
    total = part(fg_fg, dpf, dgf);
    total += part(fg_bg, dpf, dgb);
    total += part(bg_fg, dpb, dgf);
    total += part(bg_bg, dpb, dgb);
  }
  return total / (n - 1.0 + EPS);
}

__device__ __forceinline__ double ssim_q(double N, double sp, double spp, double sg, double spg) {
  const double x = sp / N, y = sg / N;
  const double vx = (spp - N * x * x) / (N - 1.0), vy = (sg - N * y * y) / (N - 1.0), cxy = (spg - N * x * y) / (N - 1.0);
  const double alpha = 4.0 * x * y * cxy, beta = (x * x + y * y) * (vx + vy);
  if (alpha != 0.0) return alpha / (beta + EPS);
  return (beta == 0.0) ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void cod_finalize_kernel(int H, int W, const double* __restrict__ S2, const double* __restrict__ S3, const unsigned* __restrict__ hist,
                                                           const double* __restrict__ T, double* __restrict__ out) {
  __shared__ double cfg[256], cbg[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const double n = (double)H * W;
  const double* s2 = S2 + (size_t)b * NS2;
  const double* s3 = S3 + (size_t)b * NS3;
  const unsigned* hf = hist + (size_t)b * 512;
  double* o = out + (size_t)b * UCOD_COD_RECORD;
  if (t == 0) {                                                        // cumulative counts from the top bin down (np.cumsum(np.flip(hist)))
    double a = 0.0, c = 0.0;
    for (int i = 0; i < 256; ++i) { a += hf[255 - i]; c += hf[256 + 255 - i]; cfg[i] = a; cbg[i] = c; }
  }
  __syncthreads();
  const double n_fg = s2[0];
  {
    const double tp = cfg[t];
    double ps = tp + cbg[t];
    if (ps == 0.0) ps = 1.0;
    const double T_ = n_fg > 1.0 ? n_fg : 1.0;
    const double prec = tp / ps, rec = tp / T_;
    const double num = (1.0 + 0.3) * prec * rec;
    const double den = (num == 0.0) ? 1.0 : 0.3 * prec + rec;
    o[8 + t] = enhanced_alignment(cfg[t], cbg[t], n_fg, n);            // E-measure curve
    o[8 + 256 + t] = num / den;                                        // F-measure curve
    o[8 + 512 + t] = prec;
    o[8 + 768 + t] = rec;
  }
  if (t != 0) return;
  o[0] = s2[2] / n;                                                    // MAE
  o[1] = s2[3] / n;                                                    // ACC
  o[2] = (s2[5] == 0.0) ? 1.0 : s2[4] / s2[5];                         // IoU (:173-179)
  // S-measure (:222-231)
  const double y = n_fg / n, mean_p = s2[1] / n;
  double sm;
  if (y == 0.0) sm = 1.0 - mean_p;
  else if (y == 1.0) sm = mean_p;
  else {
    auto s_object = [](double cnt, double sum, double sumsq) {
      const double x = sum / cnt;
      const double sd = sqrt((sumsq - cnt * x * x) / (cnt - 1.0));
      return 2.0 * x / (x * x + 1.0 + sd + EPS);
    };
    const double n_bg = n - n_fg;
    const double obj = y * s_object(n_fg, s2[6], s2[7]) + (1.0 - y) * s_object(n_bg, s2[8], s2[9]);
    int cx, cy;
    centroid(s2, H, W, cx, cy);
    const double Nq[4] = {(double)cx * cy, (double)(W - cx) * cy, (double)cx * (H - cy), (double)(W - cx) * (H - cy)};
    const double w1 = (double)cx * cy / n, w2 = (double)cy * (W - cx) / n, w3 = (double)(H - cy) * cx / n, w4 = 1.0 - w1 - w2 - w3;
    const double wq[4] = {w1, w2, w3, w4};
    double reg = 0.0;
    for (int q = 0; q < 4; ++q) reg += wq[q] * ssim_q(Nq[q], s3[2 + 4 * q], s3[3 + 4 * q], s3[4 + 4 * q], s3[5 + 4 * q]);
    sm = 0.5 * obj + 0.5 * reg;
    sm = sm > 0.0 ? sm : 0.0;                                          // max(0, sm); a NaN (degenerate quadrant) ends as 0, as in the reference
  }
  o[3] = sm;
  // weighted F (:514-545)
  double wfm = 0.0;
  if (n_fg != 0.0) {
    const double tpw = n_fg - T[b * 2], fpw = T[b * 2 + 1];
    const double R = 1.0 - T[b * 2] / n_fg, P = tpw / (tpw + fpw + EPS);
    wfm = (1.0 + 1.0) * R * P / (R + 1.0 * P + EPS);
  }
  o[4] = wfm;
  o[5] = enhanced_alignment(s3[0], s3[1], n_fg, n);                    // adaptive E-measure
  double adp_fm = 0.0;                                                 // adaptive F-measure (:462-472)
  if (s3[0] != 0.0) {
    const double pre = s3[0] / (s3[0] + s3[1]), rec = s3[0] / n_fg;
    adp_fm = (1.0 + 0.3) * pre * rec / (0.3 * pre + rec);
  }
  o[6] = adp_fm;
  o[7] = n_fg;
}

constexpr int MAX_CHUNKS = 64;
static int chunks_for(int n) { const int c = cdiv(n, 8192); return c < 1 ? 1 : (c > MAX_CHUNKS ? MAX_CHUNKS : c); }

struct Layout { size_t mm, s2, s3, hist, t, part, lc, rc, dist, et, total; };
static Layout layout(int B, int H, int W) {
  Layout l;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t n = (size_t)H * W;
  l.mm = take((size_t)B * 4 * sizeof(float));
  l.s2 = take((size_t)B * NS2 * sizeof(double));
  l.s3 = take((size_t)B * NS3 * sizeof(double));
  l.hist = take((size_t)B * 512 * sizeof(unsigned));
  l.t = take((size_t)B * 2 * sizeof(double));
  l.part = take((size_t)B * MAX_CHUNKS * NS3 * sizeof(double));      // chunk partials of the pass in flight (NS3 is the widest)
  l.lc = take((size_t)B * n * sizeof(int));
  l.rc = take((size_t)B * n * sizeof(int));
  l.dist = take((size_t)B * n * sizeof(double));
  l.et = take((size_t)B * n * sizeof(double));
  l.total = off;
  return l;
}

// fspecial('gaussian', 7, 5) as numpy builds it (:547-559), including the order of the normalising sum (numpy's pairwise sum of 49
// contiguous doubles: eight running partials over the first 48, combined as a balanced tree, then the last element)
static Gauss7 gauss7() {
  Gauss7 g;
  double mx = 0.0;
  for (int y = 0; y < 7; ++y)
    for (int x = 0; x < 7; ++x) {
      const double dy = y - 3.0, dx = x - 3.0;
      g.k[y * 7 + x] = std::exp(-(dx * dx + dy * dy) / (2.0 * 5.0 * 5.0));
      mx = g.k[y * 7 + x] > mx ? g.k[y * 7 + x] : mx;
    }
  for (double& v : g.k) if (v < EPS * mx) v = 0.0;
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = g.k[j];
  for (int i = 8; i < 48; i += 8)
    for (int j = 0; j < 8; ++j) r[j] += g.k[i + j];
  double sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  sum += g.k[48];
  for (double& v : g.k) v /= sum;
  return g;
}

}  // namespace
}  // namespace ucod

extern "C" size_t ucod_cod_metrics_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return ucod::layout(B, H, W).total;
}

extern "C" int ucod_cod_metrics(const float* pred, const float* gt, int B, int H, int W, double* out, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace ucod;
  if (!pred || !gt || !out || !workspace || B <= 0 || H <= 0 || W <= 0 || (long)H * W > (1L << 30)) return UCOD_EINVAL;
  const Layout l = layout(B, H, W);
  if (workspace_bytes < l.total) return UCOD_ENOMEM;
  char* ws = (char*)workspace;
  float* mm = (float*)(ws + l.mm);
  double* S2 = (double*)(ws + l.s2);
  double* S3 = (double*)(ws + l.s3);
  unsigned* hist = (unsigned*)(ws + l.hist);
  double* T = (double*)(ws + l.t);
  int* Lc = (int*)(ws + l.lc);
  int* Rc = (int*)(ws + l.rc);
  double* dist = (double*)(ws + l.dist);
  double* Et = (double*)(ws + l.et);
  hipStream_t s = (hipStream_t)stream;
  const int n = H * W;
  static const Gauss7 gk = gauss7();
  double* part = (double*)(ws + l.part);
  const int ch = chunks_for(n);
  hipLaunchKernelGGL(cod_minmax_kernel, dim3(B), dim3(1024), 0, s, pred, gt, n, mm);
  if (const hipError_t e = hipMemsetAsync(hist, 0, (size_t)B * 512 * sizeof(unsigned), s); e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(cod_pass2_kernel, dim3(ch, B), dim3(1024), 0, s, pred, gt, H, W, mm, part, hist);
  hipLaunchKernelGGL(cod_combine_kernel, dim3(B), dim3(32), 0, s, part, ch, NS2, S2);
  hipLaunchKernelGGL(cod_pass3_kernel, dim3(ch, B), dim3(1024), 0, s, pred, gt, H, W, mm, S2, part);
  hipLaunchKernelGGL(cod_combine_kernel, dim3(B), dim3(32), 0, s, part, ch, NS3, S3);
  hipLaunchKernelGGL(cod_rowscan_kernel, dim3(cdiv(H, 64), B), dim3(64), 0, s, gt, H, W, mm, Lc, Rc);
  hipLaunchKernelGGL(cod_edt_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, s, pred, gt, H, W, mm, Lc, Rc, dist, Et);
  hipLaunchKernelGGL(cod_wfm_kernel, dim3(ch, B), dim3(1024), 0, s, pred, gt, H, W, mm, dist, Et, gk, part);
  hipLaunchKernelGGL(cod_combine_kernel, dim3(B), dim3(32), 0, s, part, ch, 2, T);
  hipLaunchKernelGGL(cod_finalize_kernel, dim3(B), dim3(256), 0, s, H, W, S2, S3, hist, T, out);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
