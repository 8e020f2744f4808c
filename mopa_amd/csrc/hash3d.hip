// Voxel hash, active-set construction and rule tables on the GPU.
//
// Replaces the host-side hash / rulebook construction of sparseconvnet that the
// reference reaches through mopa/models/scn_unet.py:26-28 (InputLayer mode 4,
// SubmanifoldConvolution, scn.UNet's Convolution/Deconvolution) -- SURVEY.md
// Appendix A.2/A.4/A.5.  Integer work only: results must be bit-exact with
// oracle/scn3d.py (Geometry).
//
// Canonical row order: level-0 rows in first-seen point order (A.2); level l+1
// rows in first-seen order of the parents of the level-l rows.  It is obtained
// without sorting: open-addressing insert with atomicMin(first item index) per
// slot, a flag "I am my voxel's first item", and an exclusive scan of the flags.
#include "common.h"

#define KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
#define SCAN_ITEMS 1024  // items per scan block (256 threads x 4)

__device__ __forceinline__ uint64_t mix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

// Home slot of a voxel key: the voxel's 4x4x4 block picks a 64-slot bucket (512 B of keys, 256 B of values) and the
// position inside the block the slot within it.  The 27 probes of a submanifold rule row then touch at most 8 buckets, and
// rows that are neighbours in space (consecutive rows of a lidar ring) probe the same ones -- with a plain mix64(key) home
// every probe was its own 64-byte sector: k_rulebook_subm moved 11x its algorithmic bytes (profiles/r1_joint_hbm_traffic.json).
// Voxels of one block never collide with each other; two blocks sharing a bucket do, and linear probing resolves it inside the
// bucket's lines.  The table is an index only: row numbers and rule tables do not depend on it.
#ifndef MOPA_HASH_LOCAL
#define MOPA_HASH_LOCAL 1   // 0: plain mix64(key) home (round 1) -- A/B builds only (profiles/ab)
#endif
__device__ __forceinline__ uint32_t home_slot(uint64_t key, uint32_t mask) {
  if (!MOPA_HASH_LOCAL) return (uint32_t)mix64(key) & mask;
  const uint64_t blk = key & ~0x003003003ull;   // low 2 bits of x, y, z cleared
  const uint32_t local = (uint32_t)((((key >> 24) & 3) << 4) | (((key >> 12) & 3) << 2) | (key & 3));
  return ((((uint32_t)mix64(blk)) << 6) | local) & mask;
}

__device__ __forceinline__ int resolve_n(int n_host, const int* n_dev) { return n_dev ? *n_dev : n_host; }

// ---------------------------------------------------------------- key packing
// key = b<<36 | x<<24 | y<<12 | z  (x,y,z < 4096; collate.py:183-185 gives [x,y,z,b]).
__global__ void k_pack_keys(const int64_t* __restrict__ coords, int n, uint64_t* __restrict__ keys,
                            int* __restrict__ status) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int64_t x = coords[4 * (int64_t)i + 0], y = coords[4 * (int64_t)i + 1];
    int64_t z = coords[4 * (int64_t)i + 2], b = coords[4 * (int64_t)i + 3];
    if ((uint64_t)x >= 4096 || (uint64_t)y >= 4096 || (uint64_t)z >= 4096 || (uint64_t)b >= (1u << 27)) {
      atomicOr(status, 1);
      x &= 4095; y &= 4095; z &= 4095; b &= (1 << 27) - 1;
    }
    keys[i] = ((uint64_t)b << 36) | ((uint64_t)x << 24) | ((uint64_t)y << 12) | (uint64_t)z;
  }
}

__global__ void k_coarse_keys(const uint64_t* __restrict__ fine, int n_host, const int* n_dev,
                              uint64_t* __restrict__ coarse) {
  int n = resolve_n(n_host, n_dev);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint64_t k = fine[i];
    // halve x,y,z: clear the low bit of each 12-bit field, then shift the fields right by one.
    uint64_t xyz = (k & 0xFFFFFFFFFull & ~0x001001001ull) >> 1;
    coarse[i] = (k & ~0xFFFFFFFFFull) | xyz;
  }
}

// ---------------------------------------------------------------- hash insert
__global__ void k_table_clear(uint64_t* __restrict__ tk, int* __restrict__ tv, int64_t cap) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    tk[i] = KEY_EMPTY;
    tv[i] = 0x7FFFFFFF;
  }
}

__global__ void k_insert(const uint64_t* __restrict__ keys, int n_host, const int* n_dev,
                         uint64_t* __restrict__ tk, int* __restrict__ tv, uint32_t mask,
                         int* __restrict__ slot_of) {
  int n = resolve_n(n_host, n_dev);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint64_t key = keys[i];
    uint32_t s = home_slot(key, mask);
    while (true) {
      unsigned long long prev = atomicCAS((unsigned long long*)&tk[s], (unsigned long long)KEY_EMPTY,
                                          (unsigned long long)key);
      if (prev == KEY_EMPTY || prev == key) break;
      s = (s + 1) & mask;
    }
    atomicMin(&tv[s], i);
    slot_of[i] = (int)s;
  }
}

__global__ void k_first_flags(const int* __restrict__ slot_of, const int* __restrict__ tv, int n_host,
                              const int* n_dev, int* __restrict__ flags) {
  int n = resolve_n(n_host, n_dev);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    flags[i] = (tv[slot_of[i]] == i) ? 1 : 0;
}

// ---------------------------------------------------------------- exclusive scan (int32)
// Three launches: per-block sums, one-block scan of the sums, per-block local scan + offset.
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* lds /*>=8 ints*/, int* block_total) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int s = lds[k];
    if (k < w) base += s;
    tot += s;
  }
  __syncthreads();
  *block_total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(256) void k_scan_block_sums(const int* __restrict__ in, int n_host, const int* n_dev,
                                                          int* __restrict__ sums) {
  int n = resolve_n(n_host, n_dev);
  int base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n) s += in[base + k];
  s = wave_sum_i(s);
  __shared__ int l[4];
  if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = l[0] + l[1] + l[2] + l[3];
}

__global__ __launch_bounds__(256) void k_scan_sums(int* __restrict__ sums, int nblocks_cap, int n_host,
                                                    const int* n_dev, int* __restrict__ total) {
  int n = resolve_n(n_host, n_dev);
  int nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
  if (nb > nblocks_cap) nb = nblocks_cap;
  __shared__ int l[8];
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    int i = b0 + threadIdx.x;
    int v = i < nb ? sums[i] : 0;
    int tot;
    int ex = block_exclusive_scan_256(v, l, &tot);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_scan_final(const int* __restrict__ in, int n_host, const int* n_dev,
                                                     const int* __restrict__ sums, int* __restrict__ out) {
  int n = resolve_n(n_host, n_dev);
  int base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
  if (blockIdx.x * SCAN_ITEMS >= n) return;
  int v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  __shared__ int l[8];
  int tot;
  int ex = block_exclusive_scan_256(s, l, &tot) + sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
}

// Short inputs (group counts per tile, the deep levels): one block walks the array 1024 items at a time -- one launch, not three.
__global__ __launch_bounds__(256) void k_scan_small(const int* __restrict__ in, int n_host, const int* n_dev, int* __restrict__ out,
                                                     int* __restrict__ total) {
  const int n = resolve_n(n_host, n_dev);
  __shared__ int l[8];
  int carry = 0;
  for (int b0 = 0; b0 < n; b0 += 1024) {
    const int base = b0 + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = (base + k < n) ? in[base + k] : 0;
      s += v[k];
    }
    int tot;
    int ex = carry + block_exclusive_scan_256(s, l, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (base + k < n) out[base + k] = ex;
      ex += v[k];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

static int scan_exclusive(const int* in, int* out, int n_cap, const int* n_dev, int* sums, int* total,
                          hipStream_t st) {
  if (n_cap <= 8192) {
    k_scan_small<<<1, 256, 0, st>>>(in, n_cap, n_dev, out, total);
    return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
  }
  int nb = (int)cdiv64(n_cap > 0 ? n_cap : 1, SCAN_ITEMS);
  k_scan_block_sums<<<nb, 256, 0, st>>>(in, n_cap, n_dev, sums);
  k_scan_sums<<<1, 256, 0, st>>>(sums, nb, n_cap, n_dev, total);
  k_scan_final<<<nb, 256, 0, st>>>(in, n_cap, n_dev, sums, out);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

MOPA_API size_t mopa_scan_workspace_bytes(int64_t n) { return align_up((size_t)(cdiv64(n, SCAN_ITEMS) + 16) * sizeof(int), 256); }

// out[i] = sum_{j<i} in[j] (int32), *total = sum of all n inputs (device scalar).  in/out may not alias.
MOPA_API int mopa_scan_exclusive_i32(const int32_t* in, int32_t* out, int32_t n, int32_t* total, void* ws, size_t ws_bytes,
                                     void* stream) {
  if (n <= 0) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_scan_workspace_bytes(n)) return MOPA_ERR_WORKSPACE;
  return scan_exclusive(in, out, n, nullptr, (int*)ws, total, (hipStream_t)stream);
}

// ---------------------------------------------------------------- row assignment
__global__ void k_assign_rows(const uint64_t* __restrict__ keys, const int* __restrict__ flags,
                              const int* __restrict__ scan, const int* __restrict__ slot_of, int n_host,
                              const int* n_dev, int* __restrict__ tv, uint64_t* __restrict__ row_keys) {
  int n = resolve_n(n_host, n_dev);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (flags[i]) {
      int r = scan[i];
      tv[slot_of[i]] = r;
      row_keys[r] = keys[i];
    }
}

__global__ void k_item_rows(const int* __restrict__ slot_of, const int* __restrict__ tv, int n_host,
                            const int* n_dev, int* __restrict__ item_row) {
  int n = resolve_n(n_host, n_dev);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    item_row[i] = tv[slot_of[i]];
}

// Shared driver: unique keys in first-seen order.
//   keys[n] -> table (key -> row), item_row[n], row_keys[num_rows], *num_rows
// ws layout (ints): slot_of[n_cap] | flags[n_cap] | scan[n_cap] | sums[ceil(n_cap/1024)]
static size_t unique_ws_bytes(int64_t n_cap) {
  return align_up((size_t)(3 * n_cap + cdiv64(n_cap, SCAN_ITEMS) + 16) * sizeof(int), 256);
}

static int unique_first_seen(const uint64_t* keys, int n_cap, const int* n_dev, uint64_t* tk, int* tv,
                             int64_t cap, int* item_row, uint64_t* row_keys, int* num_rows, void* ws,
                             hipStream_t st) {
  if (cap <= 0 || (cap & (cap - 1)) != 0 || cap < 2 * (int64_t)n_cap || cap > (1ll << 31)) return MOPA_ERR_ARG;
  int* slot_of = (int*)ws;
  int* flags = slot_of + n_cap;
  int* scan = flags + n_cap;
  int* sums = scan + n_cap;
  int g = stream_grid(n_cap, 256);
  k_table_clear<<<stream_grid(cap, 256), 256, 0, st>>>(tk, tv, cap);
  k_insert<<<g, 256, 0, st>>>(keys, n_cap, n_dev, tk, tv, (uint32_t)(cap - 1), slot_of);
  k_first_flags<<<g, 256, 0, st>>>(slot_of, tv, n_cap, n_dev, flags);
  int rc = scan_exclusive(flags, scan, n_cap, n_dev, sums, num_rows, st);
  if (rc) return rc;
  k_assign_rows<<<g, 256, 0, st>>>(keys, flags, scan, slot_of, n_cap, n_dev, tv, row_keys);
  k_item_rows<<<g, 256, 0, st>>>(slot_of, tv, n_cap, n_dev, item_row);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- C-ABI: level 0
MOPA_API size_t mopa_voxel_hash_workspace_bytes(int64_t n_points) {
  return align_up((size_t)n_points * sizeof(uint64_t), 256) + unique_ws_bytes(n_points);
}

// coords [N,4] int64 (x,y,z,batch) on the device.  status: bit0 set if a coordinate was out of range.
MOPA_API int mopa_voxel_hash_build(const int64_t* coords, int64_t n_points, uint64_t* table_keys,
                                   int32_t* table_vals, int64_t table_cap, int32_t* point_row,
                                   uint64_t* row_keys, int32_t* num_rows, int32_t* status, void* ws,
                                   size_t ws_bytes, void* stream) {
  if (n_points <= 0 || n_points > (1 << 30)) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_voxel_hash_workspace_bytes(n_points)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  uint64_t* keys = (uint64_t*)ws;
  void* uws = (char*)ws + align_up((size_t)n_points * sizeof(uint64_t), 256);
  k_pack_keys<<<stream_grid(n_points, 256), 256, 0, st>>>(coords, (int)n_points, keys, status);
  return unique_first_seen(keys, (int)n_points, nullptr, table_keys, table_vals, table_cap, point_row, row_keys,
                           num_rows, uws, st);
}

// ---------------------------------------------------------------- C-ABI: level l -> l+1
MOPA_API size_t mopa_coarsen_workspace_bytes(int64_t n_fine_cap) { return mopa_voxel_hash_workspace_bytes(n_fine_cap); }

// fine_keys[*n_fine] -> coarse table, parent[*n_fine], coarse_keys[*num_coarse].  n_fine is read on the device
// so that all levels can be chained without a host sync; n_fine_cap bounds it (grid + buffer sizes).
MOPA_API int mopa_coarsen_build(const uint64_t* fine_keys, int32_t n_fine_cap, const int32_t* n_fine_dev,
                                uint64_t* table_keys, int32_t* table_vals, int64_t table_cap, int32_t* parent,
                                uint64_t* coarse_keys, int32_t* num_coarse, void* ws, size_t ws_bytes,
                                void* stream) {
  if (n_fine_cap <= 0) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_coarsen_workspace_bytes(n_fine_cap)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  uint64_t* keys = (uint64_t*)ws;
  void* uws = (char*)ws + align_up((size_t)n_fine_cap * sizeof(uint64_t), 256);
  k_coarse_keys<<<stream_grid(n_fine_cap, 256), 256, 0, st>>>(fine_keys, n_fine_cap, n_fine_dev, keys);
  return unique_first_seen(keys, n_fine_cap, n_fine_dev, table_keys, table_vals, table_cap, parent, coarse_keys,
                           num_coarse, uws, st);
}

// ---------------------------------------------------------------- row ranges of scan groups
// Rows are numbered in first-seen order (points in the order given; a coarse row at the first fine row that maps to it), and a
// voxel key carries its scan index: when the points of scans 0 .. B0-1 come first in the batch, their rows come first at EVERY
// level.  out[0] = max(item_row[i]) + 1 over i < n (n = *n_dev if n_dev else n_host): the number of rows of the first group at the
// level `item_row` maps into, given the number of its items (points, or rows of the finer level).  out[0] must be 0 before.
__global__ void k_group_split(const int* __restrict__ item_row, const int* __restrict__ n_dev, int n_host, int* __restrict__ out) {
  const int n = n_dev ? *n_dev : n_host;
  int m = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = max(m, item_row[i] + 1);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  // one atomic per BLOCK on the one output word (one per wave of a 2048-block grid was 8192 serialised atomics: 54 us per launch)
  __shared__ int wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    if (m > 0) atomicMax(out, m);
  }
}
// Integer max: the result does not depend on the order of the atomics.
MOPA_API int mopa_group_split(const int32_t* item_row, const int32_t* n_dev, int32_t n_host, int32_t n_cap, int32_t* out, void* stream) {
  if (!item_row || !out || n_cap <= 0 || (!n_dev && (n_host < 0 || n_host > n_cap))) return MOPA_ERR_ARG;
  int grid = stream_grid(n_cap, 256 * 8);   // grid-stride, 8+ items per thread
  if (grid > 512) grid = 512;
  k_group_split<<<grid, 256, 0, (hipStream_t)stream>>>(item_row, n_dev, n_host, out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- rule tables
__device__ __forceinline__ int table_lookup(const uint64_t* __restrict__ tk, const int* __restrict__ tv,
                                            uint32_t mask, uint64_t key) {
  uint32_t s = home_slot(key, mask);
  while (true) {
    uint64_t k = tk[s];
    if (k == key) return tv[s];
    if (k == KEY_EMPTY) return -1;
    s = (s + 1) & mask;
  }
}

// nbr[o][i] = row of voxel at pos(i) + off(o), o = (dx+1)*9 + (dy+1)*3 + (dz+1)  (A.4), or -1.
// The table is point-symmetric -- nbr[o][i] = j  <=>  nbr[26 - o][j] = i -- so only offsets 0 .. 12 are probed (13 random hash
// probes per row instead of 26: the probes were 7x the kernel's algorithmic bytes, a 64-byte sector per 8-byte key); a hit also
// writes the mirrored entry, rows 14 .. 26 are pre-filled with -1 (every entry is written by at most one thread: the mirror of
// (o, i) is unique).  Same table bit for bit.
__global__ __launch_bounds__(256) void k_rulebook_subm(const uint64_t* __restrict__ row_keys, int A,
                                                        const uint64_t* __restrict__ tk, const int* __restrict__ tv,
                                                        uint32_t mask, int size, int* __restrict__ nbr) {
  int64_t total = (int64_t)A * 14;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int o = (int)(t / A), i = (int)(t - (int64_t)o * A);
    int r;
    if (o == 13) {
      r = i;
    } else {
      uint64_t k = row_keys[i];
      int x = (int)((k >> 24) & 4095) + o / 9 - 1;
      int y = (int)((k >> 12) & 4095) + (o / 3) % 3 - 1;
      int z = (int)(k & 4095) + o % 3 - 1;
      if ((unsigned)x >= (unsigned)size || (unsigned)y >= (unsigned)size || (unsigned)z >= (unsigned)size) {
        r = -1;
      } else {
        uint64_t q = (k & ~0xFFFFFFFFFull) | ((uint64_t)x << 24) | ((uint64_t)y << 12) | (uint64_t)z;
        r = table_lookup(tk, tv, mask, q);
      }
      if (r >= 0) nbr[(int64_t)(26 - o) * A + r] = i;
    }
    nbr[t] = r;
  }
}

__global__ void k_fill_i32(int* __restrict__ p, int64_t n, int v);

MOPA_API int mopa_rulebook_subm(const uint64_t* row_keys, int32_t num_rows, const uint64_t* table_keys,
                                const int32_t* table_vals, int64_t table_cap, int32_t spatial_size,
                                int32_t* nbr /*[27][num_rows]*/, void* stream) {
  if (num_rows <= 0 || (table_cap & (table_cap - 1)) != 0) return MOPA_ERR_ARG;
  k_fill_i32<<<stream_grid((int64_t)num_rows * 13, 256), 256, 0, (hipStream_t)stream>>>(nbr + (int64_t)14 * num_rows, (int64_t)num_rows * 13, -1);
  MOPA_CHECK_LAUNCH();
  k_rulebook_subm<<<stream_grid((int64_t)num_rows * 14, 256), 256, 0, (hipStream_t)stream>>>(
      row_keys, num_rows, table_keys, table_vals, (uint32_t)(table_cap - 1), spatial_size, nbr);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ch[o][q] = fine row that is child o of coarse row q (or -1); up[o][p] = parent(p) iff octant(p)==o else -1.
// o = (x&1)*4 + (y&1)*2 + (z&1)  (A.5).
__global__ void k_fill_i32(int* __restrict__ p, int64_t n, int v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_rulebook_updown(const uint64_t* __restrict__ fine_keys, const int* __restrict__ parent, int A_f,
                                  int A_c, int* __restrict__ ch, int* __restrict__ up) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < A_f; p += gridDim.x * blockDim.x) {
    uint64_t k = fine_keys[p];
    int o = (int)(((k >> 24) & 1) * 4 + ((k >> 12) & 1) * 2 + (k & 1));
    int q = parent[p];
    ch[(int64_t)o * A_c + q] = p;
#pragma unroll
    for (int j = 0; j < 8; ++j) up[(int64_t)j * A_f + p] = (j == o) ? q : -1;
  }
}

MOPA_API int mopa_rulebook_updown(const uint64_t* fine_keys, const int32_t* parent, int32_t num_fine,
                                  int32_t num_coarse, int32_t* ch /*[8][num_coarse]*/,
                                  int32_t* up /*[8][num_fine]*/, void* stream) {
  if (num_fine <= 0 || num_coarse <= 0) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  k_fill_i32<<<stream_grid((int64_t)num_coarse * 8, 256), 256, 0, st>>>(ch, (int64_t)num_coarse * 8, -1);
  k_rulebook_updown<<<stream_grid(num_fine, 256), 256, 0, st>>>(fine_keys, parent, num_fine, num_coarse, ch, up);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- CSR of points per voxel row
// row_start[A+1], row_points[N]: points of each row in increasing point index (deterministic sums).
__global__ void k_count_rows(const int* __restrict__ item_row, int n, int* __restrict__ cnt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(&cnt[item_row[i]], 1);
}
__global__ void k_fill_rows(const int* __restrict__ item_row, int n, const int* __restrict__ row_start,
                            int* __restrict__ cursor, int* __restrict__ row_items) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int r = item_row[i];
    row_items[row_start[r] + atomicAdd(&cursor[r], 1)] = i;
  }
}
__global__ void k_sort_rows(const int* __restrict__ row_start, int A, int n, int* __restrict__ row_items) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < A; r += gridDim.x * blockDim.x) {
    int s = row_start[r], e = (r + 1 < A) ? row_start[r + 1] : n;
    for (int a = s + 1; a < e; ++a) {  // insertion sort: rows hold a handful of points
      int v = row_items[a], b = a - 1;
      while (b >= s && row_items[b] > v) { row_items[b + 1] = row_items[b]; --b; }
      row_items[b + 1] = v;
    }
  }
}
__global__ void k_set_last(int* row_start, int A, int n) { row_start[A] = n; }

MOPA_API size_t mopa_points_csr_workspace_bytes(int64_t num_rows) {
  return align_up((size_t)(2 * num_rows + cdiv64(num_rows, SCAN_ITEMS) + 32) * sizeof(int), 256);
}

MOPA_API int mopa_points_csr(const int32_t* point_row, int32_t n_points, int32_t num_rows,
                             int32_t* row_start /*[num_rows+1]*/, int32_t* row_points /*[n_points]*/, void* ws,
                             size_t ws_bytes, void* stream) {
  if (n_points <= 0 || num_rows <= 0) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_points_csr_workspace_bytes(num_rows)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* cnt = (int*)ws;
  int* cursor = cnt + num_rows;
  int* sums = cursor + num_rows;
  int* total = sums + cdiv64(num_rows, SCAN_ITEMS) + 8;
  if (hipMemsetAsync(cnt, 0, (size_t)2 * num_rows * sizeof(int), st) != hipSuccess) return MOPA_ERR_LAUNCH;
  int g = stream_grid(n_points, 256);
  k_count_rows<<<g, 256, 0, st>>>(point_row, n_points, cnt);
  int rc = scan_exclusive(cnt, row_start, num_rows, nullptr, sums, total, st);
  if (rc) return rc;
  k_set_last<<<1, 1, 0, st>>>(row_start, num_rows, n_points);
  k_fill_rows<<<g, 256, 0, st>>>(point_row, n_points, row_start, cursor, row_points);
  k_sort_rows<<<stream_grid(num_rows, 256), 256, 0, st>>>(row_start, num_rows, n_points, row_points);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ---------------------------------------------------------------- device voxeliser
// coords = trunc( float( double( rint(p * scale) - min_p rint(p * scale) ) + offset ) ),  offset_a = clip(full_scale -
// max_a - 0.001f, 0) * u_a  -- the arithmetic of augment_and_scale_3d after its rotation
// (mopa/data/utils/augmentation_3d.py:48-59) followed by the dataset's int64 cast and in-range filter
// (mopa/data/nuscenes/nuscenes_dataloader.py:419-424).  Output rows are [x, y, z, batch] int64 = the collate layout
// (mopa/data/collate.py:183-185); keep[i] = 0 marks points the dataset would drop.  Bit-exact with the reference
// (fixture G4) for a given set of (already rotated) points and translation draws u.
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void k_vox_init(int* __restrict__ mm) {
  if (threadIdx.x < 3) { mm[threadIdx.x] = 0x7FFFFFFF; mm[3 + threadIdx.x] = (int)0x80000000; }
}
__global__ __launch_bounds__(256) void k_vox_minmax(const float* __restrict__ pts, int n, float scale, int* __restrict__ mm) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float r = rintf(pts[3 * (int64_t)i + a] * scale);
      lo[a] = fminf(lo[a], r);
      hi[a] = fmaxf(hi[a], r);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int o = 32; o > 0; o >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
    }
    if ((threadIdx.x & 63) == 0) {  // min / max are order-independent: integer atomics on the ordered bit pattern
      atomicMin(&mm[a], f2ord(lo[a]));
      atomicMax(&mm[3 + a], f2ord(hi[a]));
    }
  }
}
__global__ void k_vox_coords(const float* __restrict__ pts, int n, float scale, int full_scale, const int* __restrict__ mm,
                             double u0, double u1, double u2, int transl, int64_t batch, int64_t* __restrict__ coords,
                             unsigned char* __restrict__ keep) {
  const double u[3] = {u0, u1, u2};
  float mn[3];
  double off[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = ord2f(mm[a]);
    const float mx = ord2f(mm[3 + a]) - mn[a];                    // max of the min-shifted coordinates (exact)
    float t = (float)full_scale - mx;                             // float32 like numpy's weak-scalar arithmetic
    t = t - 0.001f;
    t = fmaxf(t, 0.f);
    off[a] = transl ? (double)t * u[a] : 0.0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float c = rintf(pts[3 * (int64_t)i + a] * scale) - mn[a];
      if (transl) c = (float)((double)c + off[a]);
      const int64_t ci = (int64_t)c;
      coords[4 * (int64_t)i + a] = ci;
      ok = ok && ci >= 0 && ci < full_scale;
    }
    coords[4 * (int64_t)i + 3] = batch;
    keep[i] = ok ? 1 : 0;
  }
}

// The rotation / flip stage of augment_and_scale_3d (mopa/data/utils/augmentation_3d.py:26-50): out = points @ R for a
// 3x3 float32 matrix drawn by the caller (numpy's global RNG, like the reference).  float32 like numpy's float32 dot
// (sgemm: acc = fma(a_k, b_k, acc) over k); a last-ulp difference against a particular BLAS is possible, the voxel
// coordinates downstream are pinned by fixture G4.
__global__ void k_rotate_f32(const float* __restrict__ pts, int n, float r00, float r01, float r02, float r10, float r11, float r12,
                             float r20, float r21, float r22, float* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = pts[3 * (int64_t)i], y = pts[3 * (int64_t)i + 1], z = pts[3 * (int64_t)i + 2];
    out[3 * (int64_t)i] = fmaf(z, r20, fmaf(y, r10, __fmul_rn(x, r00)));
    out[3 * (int64_t)i + 1] = fmaf(z, r21, fmaf(y, r11, __fmul_rn(x, r01)));
    out[3 * (int64_t)i + 2] = fmaf(z, r22, fmaf(y, r12, __fmul_rn(x, r02)));
  }
}
MOPA_API int mopa_rotate_points_f32(const float* points, int32_t n, const float* rot_host /*[9] row-major*/, float* out, void* stream) {
  if (n <= 0 || !rot_host) return MOPA_ERR_ARG;
  const float* r = rot_host;
  k_rotate_f32<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(points, n, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

MOPA_API size_t mopa_voxelize_workspace_bytes(void) { return 256; }

// points [n][3] fp32 (already rotated / scaled by the augmentation), u_host: the 3 uniform draws of the translation.
MOPA_API int mopa_voxelize(const float* points, int32_t n, float scale, int32_t full_scale, const double* u_host,
                           int32_t transl, int64_t batch_index, int64_t* coords /*[n][4]*/, uint8_t* keep /*[n]*/, void* ws,
                           size_t ws_bytes, void* stream) {
  if (n <= 0 || full_scale <= 0 || (transl && !u_host)) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_voxelize_workspace_bytes()) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* mm = (int*)ws;
  k_vox_init<<<1, 64, 0, st>>>(mm);
  k_vox_minmax<<<stream_grid(n, 256), 256, 0, st>>>(points, n, scale, mm);
  k_vox_coords<<<stream_grid(n, 256), 256, 0, st>>>(points, n, scale, full_scale, mm, transl ? u_host[0] : 0.0,
                                                    transl ? u_host[1] : 0.0, transl ? u_host[2] : 0.0, transl, batch_index,
                                                    coords, keep);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
