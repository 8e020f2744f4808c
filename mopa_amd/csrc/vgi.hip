// Valid Ground-based Insertion (VGI) on the device -- SURVEY.md 8f-4, the data-side step of a MoPA iteration that the
// reference times as `g_insert_time` (mopa/train/train_xmuda_mopa.py:483-555).  Integer / byte work on dense grids over
// the bounded search region; HBM-bound, no MFMA.
//
// Replaces, per target scan:
//   mopa/data/mixmatch_ss.py:215-331  check_overlap   occupancy grid (*) all-ones box via float64 F.conv3d on the GPU,
//                                                      torch.nonzero, host round trips       -> mopa_vgi_first_points,
//                                                                                               mopa_vgi_box_free
//   mopa/data/mixmatch_ss.py:139-160  centre filters   (in front, inside the image, range)   -> mopa_vgi_candidates
//   mopa/data/mixmatch_ss.py:372-409  obj_on_road      ground-cell lookup by an (n_ground x n_centres) broadcast
//                                                      compare on the GPU                    -> mopa_vgi_ground_cells,
//                                                                                               mopa_vgi_candidates, _compact_cells
//   mopa/data/mixmatch_ss.py:431-446  road height      mean z of the lowest ground voxel     -> mopa_vgi_road_height
//   mopa/data/utils/augmentation_3d.py:161-290 (+ :81-111) range_projection / occulusion_detector: lexsort + an
//                                                      (N x n_obj) broadcast compare          -> mopa_vgi_range_keep
//   mopa/data/utils/augmentation_3d.py:48-59 on the float64 concatenated cloud (post_process, mixmatch_ss.py:522-545)
//                                                                                            -> mopa_voxelize_f64
// Oracle: oracle/vgi.py (pinned by fixture G8 = the reference's own outputs).
//
// Layout: `first[(cx * Y + cy) * ZR + (vz - zlo)]` = smallest index of the points whose voxel floor(p / voxel_size) is
// (ox + cx, oy + cy, vz), INT_MAX if none: torchsparse's sparse_quantize keeps the FIRST point of a voxel as its
// representative, and the reference's ground flag of a voxel is that point's flag (mixmatch_ss.py:397).  The map covers the
// search region in x / y (everything downstream only looks at cells inside it) and ZR = 128 voxel slots in z.
#include "common.h"
#include <limits.h>
#include <math.h>
#include <string.h>

#define VGI_ZR 128

MOPA_API int mopa_vgi_zslots(void) { return VGI_ZR; }

// ---------------------------------------------------------------- first point of every voxel of the search map
__global__ void k_vgi_fill_i32(int* __restrict__ p, int64_t n, int v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_vgi_first(const float* __restrict__ pts, int stride, int n, float vs, int ox, int oy, int X, int Y, int zlo,
                            const unsigned char* __restrict__ g_mask, int* __restrict__ first, int* __restrict__ status) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = pts + (int64_t)i * stride;
    // float32 / float32 like numpy's `pc[:, :3] / voxel_size` with a python-float voxel size (correctly rounded division)
    const int vx = (int)floorf(__fdiv_rn(p[0], vs)), vy = (int)floorf(__fdiv_rn(p[1], vs)), vz = (int)floorf(__fdiv_rn(p[2], vs));
    const int cx = vx - ox, cy = vy - oy, cz = vz - zlo;
    if (cx < 0 || cx >= X || cy < 0 || cy >= Y) continue;
    if (cz < 0 || cz >= VGI_ZR) {
      if (g_mask && g_mask[i]) atomicOr(status, 1);   // a ground point outside the z slots: the map would miss its voxel
      continue;
    }
    atomicMin(&first[((int64_t)cx * Y + cy) * VGI_ZR + cz], i);
  }
}
MOPA_API int mopa_vgi_first_points(const float* points, int32_t stride, int32_t n, float voxel_size, int32_t ox, int32_t oy,
                                   int32_t X, int32_t Y, int32_t zlo, const uint8_t* g_mask, int32_t* first, int32_t* status,
                                   void* stream) {
  if (n <= 0 || stride < 3 || X <= 0 || Y <= 0 || !(voxel_size > 0.f)) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t cells = (int64_t)X * Y * VGI_ZR;
  k_vgi_fill_i32<<<stream_grid(cells, 256), 256, 0, st>>>(first, cells, INT_MAX);
  if (hipMemsetAsync(status, 0, sizeof(int), st) != hipSuccess) return MOPA_ERR_LAUNCH;
  k_vgi_first<<<stream_grid(n, 256), 256, 0, st>>>(points, stride, n, voxel_size, ox, oy, X, Y, zlo, g_mask, first, status);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- overlap test: window OR of the occupancy, axis by axis
// in: [X][Y][Zi] bytes (or, for the first pass, the `first` volume restricted to z slots [z0, z0 + Zi)); out[..][..][Zo],
// out = OR over a window of `w` cells along the given axis.  Equivalent to the reference's dense conv3d with an all-ones
// box followed by `== 0` (the sum of non-negative integers is zero iff every term is).
__global__ void k_vgi_or_z(const int* __restrict__ first, int XY, int z0, int Zi, int w, unsigned char* __restrict__ out) {
  const int Zo = Zi - w + 1;
  const int64_t total = (int64_t)XY * Zo;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i / Zo), z = (int)(i - (int64_t)col * Zo);
    const int* f = first + (int64_t)col * VGI_ZR + z0 + z;
    unsigned char o = 0;
    for (int k = 0; k < w; ++k) o |= f[k] != INT_MAX;
    out[i] = o;
  }
}
__global__ void k_vgi_or_axis(const unsigned char* __restrict__ in, int X, int Y, int Z, int axis, int w, unsigned char* __restrict__ out,
                              int invert) {
  const int Xo = axis == 0 ? X - w + 1 : X, Yo = axis == 1 ? Y - w + 1 : Y;
  const int64_t total = (int64_t)Xo * Yo * Z;
  const int64_t step = axis == 0 ? (int64_t)Y * Z : Z;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int z = (int)(i % Z);
    const int64_t t = i / Z;
    const int y = (int)(t % Yo), x = (int)(t / Yo);
    const unsigned char* p = in + ((int64_t)x * Y + y) * Z + z;
    unsigned char o = 0;
    for (int k = 0; k < w; ++k) o |= p[k * step];
    out[i] = invert ? !o : o;
  }
}
MOPA_API size_t mopa_vgi_box_free_workspace_bytes(int32_t X, int32_t Y, int32_t Z) { return align_up((size_t)2 * X * Y * Z, 256); }

// free[(X-bx+1)][(Y-by+1)][(Z-bz+1)] = 1 where the bx x by x bz box placed at that cell meets no occupied voxel of the grid
// [X][Y][Z] whose z origin is voxel slot (gz0 - zlo) of the `first` volume.
MOPA_API int mopa_vgi_box_free(const int32_t* first, int32_t X, int32_t Y, int32_t zlo, int32_t gz0, int32_t Z, int32_t bx, int32_t by,
                               int32_t bz, uint8_t* free_cells, void* ws, size_t ws_bytes, void* stream) {
  const int z0 = gz0 - zlo;
  if (X <= 0 || Y <= 0 || Z <= 0 || z0 < 0 || z0 + Z > VGI_ZR || bx <= 0 || by <= 0 || bz <= 0) return MOPA_ERR_ARG;
  if (bx > X || by > Y || bz > Z) return MOPA_ERR_ARG;   // caller: no candidate at all
  if (ws_bytes < mopa_vgi_box_free_workspace_bytes(X, Y, Z)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* t1 = (unsigned char*)ws;
  unsigned char* t2 = t1 + (size_t)X * Y * Z;
  const int Zo = Z - bz + 1, Yo = Y - by + 1, Xo = X - bx + 1;
  k_vgi_or_z<<<stream_grid((int64_t)X * Y * Zo, 256), 256, 0, st>>>(first, X * Y, z0, Z, bz, t1);
  k_vgi_or_axis<<<stream_grid((int64_t)X * Yo * Zo, 256), 256, 0, st>>>(t1, X, Y, Zo, 1, by, t2, 0);
  k_vgi_or_axis<<<stream_grid((int64_t)Xo * Yo * Zo, 256), 256, 0, st>>>(t2, X, Yo, Zo, 0, bx, free_cells, 1);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- ground columns
__global__ void k_vgi_ground2d(const int* __restrict__ first, const unsigned char* __restrict__ g_mask, int XY, unsigned char* __restrict__ g2d) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < XY; c += gridDim.x * blockDim.x) {
    const int* f = first + (int64_t)c * VGI_ZR;
    unsigned char o = 0;
    for (int z = 0; z < VGI_ZR; ++z) {
      const int i = f[z];
      if (i != INT_MAX && g_mask[i]) { o = 1; break; }
    }
    g2d[c] = o;
  }
}
MOPA_API int mopa_vgi_ground_cells(const int32_t* first, const uint8_t* g_mask, int32_t X, int32_t Y, uint8_t* ground2d, void* stream) {
  if (X <= 0 || Y <= 0) return MOPA_ERR_ARG;
  k_vgi_ground2d<<<stream_grid((int64_t)X * Y, 256), 256, 0, (hipStream_t)stream>>>(first, g_mask, X * Y, ground2d);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- candidate centres -> ground cells
struct VgiCand {
  double ext[3], off[3], vs, ori_range;
  float P[12], img_w, img_h;
  int Xo, Yo, Zo, ox, oy, X, Y;
};
__global__ void k_vgi_candidates(const unsigned char* __restrict__ free_cells, const VgiCand a, const unsigned char* __restrict__ g2d,
                                 unsigned char* __restrict__ cand2d, int* __restrict__ counts) {
  const int64_t total = (int64_t)a.Xo * a.Yo * a.Zo;
  int n_free = 0, n_kept = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (!free_cells[i]) continue;
    ++n_free;
    const int z = (int)(i % a.Zo);
    const int64_t t = i / a.Zo;
    const int y = (int)(t % a.Yo), x = (int)(t / a.Yo);
    // centre in metres, float64 like the reference: (start + (extent - 1) / 2 + v2g_offset) * voxel_size
    const double cx = ((double)x + (a.ext[0] - 1.0) / 2.0 + a.off[0]) * a.vs;
    const double cy = ((double)y + (a.ext[1] - 1.0) / 2.0 + a.off[1]) * a.vs;
    const double cz = ((double)z + (a.ext[2] - 1.0) / 2.0 + a.off[2]) * a.vs;
    if (!(cx > 0.0)) continue;                                           // condition 1a: in front (mixmatch_ss.py:141)
    const float hx = (float)cx, hy = (float)cy, hz = (float)cz;          // float32 projection (:143-147)
    float p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      p[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a.P[4 * r], hx), __fmul_rn(a.P[4 * r + 1], hy)), __fmul_rn(a.P[4 * r + 2], hz)), a.P[4 * r + 3]);
    const float u = __fdiv_rn(p[0], p[2]), v = __fdiv_rn(p[1], p[2]);
    if (!(u > 0.f && v > 0.f && u < a.img_w && v < a.img_h)) continue;  // condition 1b: inside the image (:148)
    if (!(sqrt(__dadd_rn(__dmul_rn(cx, cx), __dmul_rn(cy, cy))) >= a.ori_range)) continue;             // condition 2: not nearer than the object was (:152-156)
    ++n_kept;
    const int fx = (int)floor(cx / a.vs) - a.ox, fy = (int)floor(cy / a.vs) - a.oy;   // np.floor(valid_centers / voxel_size)
    if (fx < 0 || fx >= a.X || fy < 0 || fy >= a.Y) continue;
    if (g2d[fx * a.Y + fy]) cand2d[fx * a.Y + fy] = 1;
  }
  n_free = wave_sum_i(n_free);
  n_kept = wave_sum_i(n_kept);
  if ((threadIdx.x & 63) == 0) {
    if (n_free) atomicAdd(&counts[0], n_free);
    if (n_kept) atomicAdd(&counts[1], n_kept);
  }
}
// params_host: ext[3], off[3], voxel_size, ori_range (8 doubles), then P[12], img_w, img_h as doubles (14) = 22 doubles.
// counts (device int32[2]): number of free cells, number of centres that pass the three filters.
MOPA_API int mopa_vgi_candidates(const uint8_t* free_cells, int32_t Xo, int32_t Yo, int32_t Zo, const double* params_host,
                                 const uint8_t* ground2d, int32_t ox, int32_t oy, int32_t X, int32_t Y, uint8_t* cand2d,
                                 int32_t* counts, void* stream) {
  if (Xo <= 0 || Yo <= 0 || Zo <= 0 || X <= 0 || Y <= 0 || !params_host) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  VgiCand a;
  for (int i = 0; i < 3; ++i) { a.ext[i] = params_host[i]; a.off[i] = params_host[3 + i]; }
  a.vs = params_host[6]; a.ori_range = params_host[7];
  for (int i = 0; i < 12; ++i) a.P[i] = (float)params_host[8 + i];
  a.img_w = (float)params_host[20]; a.img_h = (float)params_host[21];
  a.Xo = Xo; a.Yo = Yo; a.Zo = Zo; a.ox = ox; a.oy = oy; a.X = X; a.Y = Y;
  if (hipMemsetAsync(cand2d, 0, (size_t)X * Y, st) != hipSuccess || hipMemsetAsync(counts, 0, 2 * sizeof(int), st) != hipSuccess)
    return MOPA_ERR_LAUNCH;
  k_vgi_candidates<<<stream_grid((int64_t)Xo * Yo * Zo, 256), 256, 0, st>>>(free_cells, a, ground2d, cand2d, counts);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ordered compaction of the marked (x, y) cells: lexicographic = row-major over the map, the order np.unique(..., axis=0)
// gives the reference (mixmatch_ss.py:408).  The map has at most a few 10^4 cells: one block, ballot prefix per 1024 cells.
__global__ __launch_bounds__(1024) void k_vgi_compact(const unsigned char* __restrict__ cand2d, int XY, int Y, int ox, int oy,
                                                       int* __restrict__ cells, int* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < XY; b0 += 1024) {
    const int i = b0 + tid;
    const bool m = i < XY && cand2d[i];
    const unsigned long long bal = __ballot(m);
    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (m) { cells[2 * (off + pos)] = ox + i / Y; cells[2 * (off + pos) + 1] = oy + i % Y; }
    __syncthreads();
    if (tid == 0) { int s = 0; for (int k = 0; k < 16; ++k) s += wsum[k]; base += s; }
    __syncthreads();
  }
  if (tid == 0) *count = base;
}
MOPA_API int mopa_vgi_compact_cells(const uint8_t* cand2d, int32_t X, int32_t Y, int32_t ox, int32_t oy, int32_t* cells /*[X*Y][2]*/,
                                    int32_t* count, void* stream) {
  if (X <= 0 || Y <= 0) return MOPA_ERR_ARG;
  k_vgi_compact<<<1, 1024, 0, (hipStream_t)stream>>>(cand2d, X * Y, Y, ox, oy, cells, count);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- road height of a chosen cell
// out[0] = sum of z over the points of the LOWEST ground voxel of column (cell_x, cell_y), out[1] = their count
// (mixmatch_ss.py:431-444: argmin over the ground voxels of the cell, then the mean z of that voxel's points).
// ONE block: every thread sums its points in index order, the waves' fixed xor-butterflies and an ordered sum over the
// waves follow -- the same bits run to run (no floating-point atomics; a cloud is at most a few hundred thousand points).
__global__ __launch_bounds__(1024) void k_vgi_road(const float* __restrict__ pts, int stride, int n, float vs, const int* __restrict__ first,
                                                    const unsigned char* __restrict__ g_mask, int col, int zlo, int vx, int vy,
                                                    double* __restrict__ out) {
  __shared__ int s_vz;
  __shared__ double ws[16], wc[16];
  if (threadIdx.x == 0) {
    int vz = INT_MAX;
    const int* f = first + (int64_t)col * VGI_ZR;
    for (int z = 0; z < VGI_ZR; ++z)
      if (f[z] != INT_MAX && g_mask[f[z]]) { vz = zlo + z; break; }
    s_vz = vz;
  }
  __syncthreads();
  const int vz = s_vz;
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float* p = pts + (int64_t)i * stride;
    if ((int)floorf(__fdiv_rn(p[0], vs)) == vx && (int)floorf(__fdiv_rn(p[1], vs)) == vy && (int)floorf(__fdiv_rn(p[2], vs)) == vz) {
      s += (double)p[2];
      c += 1.0;
    }
  }
  s = wave_sum_d(s);
  c = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = s; wc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tc = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += ws[w]; tc += wc[w]; }
    out[0] = ts;
    out[1] = tc;
  }
}
MOPA_API int mopa_vgi_road_height(const float* points, int32_t stride, int32_t n, float voxel_size, const int32_t* first,
                                  const uint8_t* g_mask, int32_t ox, int32_t oy, int32_t X, int32_t Y, int32_t zlo, int32_t cell_x,
                                  int32_t cell_y, double* out /*[2]*/, void* stream) {
  const int cx = cell_x - ox, cy = cell_y - oy;
  if (n <= 0 || cx < 0 || cx >= X || cy < 0 || cy >= Y) return MOPA_ERR_ARG;
  k_vgi_road<<<1, 1024, 0, (hipStream_t)stream>>>(points, stride, n, voxel_size, first, g_mask, cx * Y + cy, zlo, cell_x, cell_y, out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- range-image occlusion culling
// In every range-image pixel that holds at least one inserted-object point only the NEAREST point of the whole cloud
// survives (ties: the smaller point index, = the reference's stable lexsort); every other pixel keeps all its points.
__device__ __forceinline__ int vgi_pixel(const double* __restrict__ p, double fov_down_abs, double fov, int W, int H, double* depth) {
  const double x = p[0], y = p[1], z = p[2];
  // np.linalg.norm(points, 2, axis=1): separately rounded squares, left-to-right sum (no FMA contraction)
  const double d = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)), __dmul_rn(z, z)));
  *depth = d;
  const double yaw = -atan2(y, x), pitch = asin(z / d);
  double px = floor(0.5 * (yaw / M_PI + 1.0) * (double)W);
  px = fmax(0.0, fmin((double)(W - 1), px));
  double py = floor((1.0 - (pitch + fov_down_abs) / fov) * (double)H);
  py = fmax(0.0, fmin((double)(H - 1), py));
  return (int)py * W + (int)px;
}
__global__ void k_vgi_range_a(const double* __restrict__ pts, int n, int n0, double fda, double fov, int W, int H,
                              unsigned char* __restrict__ objpix, unsigned long long* __restrict__ best_d, int* __restrict__ pix) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double d;
    const int q = vgi_pixel(pts + 3 * (int64_t)i, fda, fov, W, H, &d);
    pix[i] = q;
    if (i >= n0) objpix[q] = 1;
    atomicMin(&best_d[q], (unsigned long long)__double_as_longlong(d));   // depth >= 0: the bit pattern orders like the value
  }
}
__global__ void k_vgi_range_b(const double* __restrict__ pts, int n, const int* __restrict__ pix, const unsigned long long* __restrict__ best_d,
                              int* __restrict__ best_i) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double* p = pts + 3 * (int64_t)i;
    const double d = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(p[0], p[0]), __dmul_rn(p[1], p[1])), __dmul_rn(p[2], p[2])));
    if ((unsigned long long)__double_as_longlong(d) == best_d[pix[i]]) atomicMin(&best_i[pix[i]], i);
  }
}
__global__ void k_vgi_range_c(int n, const int* __restrict__ pix, const unsigned char* __restrict__ objpix, const int* __restrict__ best_i,
                              unsigned char* __restrict__ keep) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    keep[i] = (!objpix[pix[i]] || best_i[pix[i]] == i) ? 1 : 0;
}
MOPA_API size_t mopa_vgi_range_keep_workspace_bytes(int32_t n, int32_t W, int32_t H) {
  return align_up((size_t)W * H * (8 + 4 + 1) + (size_t)n * 4 + 64, 256);
}
// points [n][3] float64 (scan points first, the inserted object's points from index n_scan on) -> keep[n].
MOPA_API int mopa_vgi_range_keep(const double* points, int32_t n, int32_t n_scan, double fov_up, double fov_down, int32_t W, int32_t H,
                                 uint8_t* keep, void* ws, size_t ws_bytes, void* stream) {
  if (n <= 0 || n_scan < 0 || n_scan > n || W <= 0 || H <= 0) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_vgi_range_keep_workspace_bytes(n, W, H)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int npx = W * H;
  unsigned long long* best_d = (unsigned long long*)ws;
  int* best_i = (int*)(best_d + npx);
  int* pix = best_i + npx;
  unsigned char* objpix = (unsigned char*)(pix + n);
  if (hipMemsetAsync(best_d, 0xFF, (size_t)npx * 8, st) != hipSuccess || hipMemsetAsync(objpix, 0, npx, st) != hipSuccess) return MOPA_ERR_LAUNCH;
  k_vgi_fill_i32<<<stream_grid(npx, 256), 256, 0, st>>>(best_i, npx, INT_MAX);
  const double fda = fabs(fov_down), fov = fabs(fov_down) + fabs(fov_up);
  const int g = stream_grid(n, 256);
  k_vgi_range_a<<<g, 256, 0, st>>>(points, n, n_scan, fda, fov, W, H, objpix, best_d, pix);
  k_vgi_range_b<<<g, 256, 0, st>>>(points, n, pix, best_d, best_i);
  k_vgi_range_c<<<g, 256, 0, st>>>(n, pix, objpix, best_i, keep);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- float64 voxeliser of post_process
// coords = trunc(round((p @ R) * scale) - min + clip(full_scale - max - 0.001, 0) * u) for the points with keep_in != 0
// (augmentation_3d.py:48-59 on float64 points; int cast + field filter: mixmatch_ss.py:535-540).  R, u: host arrays drawn
// by the caller from numpy's global RNG like the reference does (R may be NULL = identity).
__global__ void k_vox64_minmax(const double* __restrict__ pts, int n, const unsigned char* __restrict__ keep_in, const double* __restrict__ Rm,
                               double scale, unsigned long long* __restrict__ mm) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (keep_in && !keep_in[i]) continue;
    const double* p = pts + 3 * (int64_t)i;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = Rm ? p[0] * Rm[a] + p[1] * Rm[3 + a] + p[2] * Rm[6 + a] : p[a];
      const double r = rint(v * scale);
      // order-preserving map double -> uint64
      long long b = __double_as_longlong(r);
      const unsigned long long o = b >= 0 ? (unsigned long long)b | 0x8000000000000000ull : ~(unsigned long long)b;
      atomicMin(&mm[a], o);
      atomicMax(&mm[3 + a], o);
    }
  }
}
__device__ __forceinline__ double vgi_ord2d(unsigned long long o) {
  const long long b = (o & 0x8000000000000000ull) ? (long long)(o & 0x7FFFFFFFFFFFFFFFull) : (long long)~o;
  return __longlong_as_double(b);
}
__global__ void k_vox64_coords(const double* __restrict__ pts, int n, const unsigned char* __restrict__ keep_in, const double* __restrict__ Rm,
                               double scale, int full_scale, const unsigned long long* __restrict__ mm, double u0, double u1, double u2,
                               int transl, int64_t batch, int64_t* __restrict__ coords, unsigned char* __restrict__ keep) {
  const double u[3] = {u0, u1, u2};
  double mn[3], off[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = vgi_ord2d(mm[a]);
    const double mx = vgi_ord2d(mm[3 + a]) - mn[a];
    off[a] = transl ? fmax((double)full_scale - mx - 0.001, 0.0) * u[a] : 0.0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bool ok = !keep_in || keep_in[i];
    const double* p = pts + 3 * (int64_t)i;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = Rm ? p[0] * Rm[a] + p[1] * Rm[3 + a] + p[2] * Rm[6 + a] : p[a];
      const double c = rint(v * scale) - mn[a] + off[a];
      const int64_t ci = (int64_t)c;
      coords[4 * (int64_t)i + a] = ci;
      ok = ok && c >= 0.0 && c < (double)full_scale;   // the reference filters on the float coordinates (:535)
    }
    coords[4 * (int64_t)i + 3] = batch;
    keep[i] = ok ? 1 : 0;
  }
}
MOPA_API size_t mopa_voxelize_f64_workspace_bytes(void) { return 256; }
MOPA_API int mopa_voxelize_f64(const double* points, int32_t n, const uint8_t* keep_in, const double* rot_host /*[9] or NULL*/, double scale,
                               int32_t full_scale, const double* u_host, int32_t transl, int64_t batch_index, int64_t* coords /*[n][4]*/,
                               uint8_t* keep /*[n]*/, void* ws, size_t ws_bytes, void* stream) {
  if (n <= 0 || full_scale <= 0 || (transl && !u_host)) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_voxelize_f64_workspace_bytes()) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* mm = (unsigned long long*)ws;
  double* Rd = (double*)ws + 8;
  if (hipMemsetAsync(mm, 0xFF, 3 * 8, st) != hipSuccess || hipMemsetAsync(mm + 3, 0, 3 * 8, st) != hipSuccess) return MOPA_ERR_LAUNCH;
  if (rot_host && hipMemcpyAsync(Rd, rot_host, 9 * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return MOPA_ERR_LAUNCH;
  const double* Rm = rot_host ? Rd : nullptr;
  k_vox64_minmax<<<stream_grid(n, 256), 256, 0, st>>>(points, n, keep_in, Rm, scale, mm);
  k_vox64_coords<<<stream_grid(n, 256), 256, 0, st>>>(points, n, keep_in, Rm, scale, full_scale, mm, transl ? u_host[0] : 0.0,
                                                      transl ? u_host[1] : 0.0, transl ? u_host[2] : 0.0, transl, batch_index, coords, keep);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
