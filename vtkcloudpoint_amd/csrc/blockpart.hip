// blockpart.hip -- the reference's block partition on MI355X without a sort of the cloud.
//
// MainForm.getClusterFromMotor (FrmMain.cs:1214-1291; twin getClusterFromList :1136-1213) sorts the whole list by
// d = max(x - x_Min, y - y_Min) (:1229-1251), takes the first ptsInCell points as block 0 and as the measure of the block
// size (:1253-1258), and files every other point under the (lo, hi] rectangle it falls into (Tools.getListByScale2,
// BaseClass/Tools.cs:510-513); inside a block the points keep the order of the sorted list.  What the sort is needed for
// is (a) the ptsInCell-th element and (b) the order INSIDE each block.  Here:
//
//   select    radix select of the take-th smallest (d, index) pair over the 96-bit string [bits of d | index], twelve bits
//             per pass: histogram of the digit over the keys that match the prefix found so far (k_sel_hist), the bin
//             that holds the rank (k_sel_pick); as soon as that bin holds <= 65536 keys they are collected together with
//             the largest x and y of everything below them (k_sel_collect) and one workgroup finishes the selection and
//             the first block's extent (k_sel_final).  On ordinary clouds the first digit -- sign and exponent of d --
//             already isolates a few hundred keys: two passes over the coordinates in all.
//   partition per chunk of the input an LDS histogram over buckets (k_blk_hist, which also stores the block of every
//             point; the counts chunk-major, transposed for the one scan that wants them bucket-major and back), then a
//             scatter of 32-byte records (motor coordinates, bits of d, index, block) through LDS cursors (k_blk_scatter).
//             Up to 32768 blocks every block is its own bucket: the records land in block order and the block starts are
//             the scanned counts (k_blk_starts).  Beyond, a bucket is a super-bucket of 2^fsh consecutive block ids and one
//             workgroup per super-bucket counts per block, publishes the block starts and moves the records into block
//             order (k_blk_split).  Points that fall in no block ride along as block `nblocks` (they end up behind the m
//             points that did).
//   order     inside a block: ranks by distribution.  d spans a known range inside a block, so a monotone map of d onto
//             ~m/2 sub-buckets (k_blk_sort) leaves groups of a few records; a record's final place is its group's start
//             plus the number of group members with a smaller (d, index).  No comparison network, no padding, any
//             block size; a block whose points all share one d is mapped by index instead.
//
// Declared deviation from the C# (as before): List.Sort's unstable tie order is replaced by (d, index).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "blocks_state.hpp"

namespace {
constexpr int BT = 256;
constexpr int PT = 1024;             // threads of the chunked passes (one workgroup per CU at 256 chunks)
constexpr int PCH_MIN = 8192;        // smallest chunk
constexpr uint32_t MAXS = 32768;     // super-buckets (LDS histogram of the chunked passes: up to 128 KB, one workgroup per CU)
constexpr uint32_t MAXF = 8192;      // blocks per super-bucket (LDS counters of the split pass): MAXS * MAXF >= 2^26 blocks
constexpr uint32_t CAND_CAP = 65536; // keys the single-workgroup end of the selection takes

struct alignas(16) Rec32 {   // one record of the partition
  double x, y;               // motor coordinates (what every DBImproved clusters on)
  unsigned long long key;    // bits of d = max(x - x_Min, y - y_Min) of the PARTITION coordinates
  uint32_t idx, blk;         // index in the caller's list; block id (nblocks = in no block)
};
struct alignas(16) KeyPart {
  unsigned long long key;
  uint32_t idx, src;         // src = position of the record inside its block's slice of rec2
};
struct alignas(16) Cand {
  unsigned long long key;
  uint32_t idx, pad;
  double x, y;               // partition coordinates
};
struct SelState {
  unsigned long long phi, plo;   // the digits found so far, left-aligned in the 128-bit string [bits of d | index << 32]
  unsigned long long rank;       // 1-based rank still to find among the keys that match the prefix
  uint32_t count;                // keys that match the prefix
  uint32_t pass;                 // digits fixed so far
  uint32_t collect;              // count <= CAND_CAP: the collect / final kernels run
  uint32_t done;
  uint32_t ncand;                // append cursor of the candidate list
  uint32_t nbig;                 // append cursors (k_blk_split): large blocks ...
  uint32_t nbinfo, nslice, nvirt, nfall;  // ... those split into sub-ranges, their slices and sub-ranges; blocks for the slow kernel
  unsigned long long key_T;      // the take-th smallest (d, index) ...
  uint32_t idx_T, pad;
  double fx_max, fy_max;         // ... and the largest x, y over the first block
};

__device__ __forceinline__ unsigned long long dkey(double x, double y, double x_Min, double y_Min) {
  // FrmMain.cs:1231-1232: d = Math.Max(x - x_Min, y - y_Min); non-negative, so the IEEE bit pattern orders it
  const double a = x - x_Min, b = y - y_Min;
  const double d = a > b ? a : b;
  return (unsigned long long)__double_as_longlong(d + 0.0);
}

// digit `pass` (twelve bits) of the 96-bit string [k | idx], counted from the top
__device__ __forceinline__ uint32_t sel_digit(unsigned long long k, uint32_t idx, uint32_t pass) {
  const uint32_t start = 12u * pass;
  const unsigned long long lo = (unsigned long long)idx << 32;
  unsigned long long top;
  if (start == 0) top = k;
  else if (start < 64) top = (k << start) | (lo >> (64 - start));
  else top = lo << (start - 64);
  return (uint32_t)(top >> 52);
}
// the top nbits of [k | idx] against the prefix: -1 below, 0 equal, +1 above
__device__ __forceinline__ int sel_cmp(unsigned long long k, uint32_t idx, unsigned long long phi, unsigned long long plo,
                                       uint32_t nbits) {
  if (nbits == 0) return 0;
  const unsigned long long lo = (unsigned long long)idx << 32;
  const unsigned long long mh = nbits >= 64 ? ~0ull : ~(~0ull >> nbits);
  const unsigned long long ml = nbits <= 64 ? 0ull : ~(~0ull >> (nbits - 64));
  const unsigned long long kh = k & mh, kl = lo & ml;
  if (kh != phi) return kh < phi ? -1 : 1;
  if (kl != plo) return kl < plo ? -1 : 1;
  return 0;
}
__device__ __forceinline__ void sel_push(unsigned long long& phi, unsigned long long& plo, uint32_t digit, uint32_t pass) {
  const int s = 116 - 12 * (int)pass;  // left shift of the digit inside the 128-bit string
  if (s >= 64) {
    phi |= (unsigned long long)digit << (s - 64);
  } else {
    plo |= (unsigned long long)digit << s;
    if (s > 52) phi |= (unsigned long long)digit >> (64 - s);
  }
}

__device__ __forceinline__ double wmin(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmin(v, __shfl_down(v, d, 64));
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_down(v, d, 64));
  return v;
}

// per-workgroup partials [nb][5]: x min, x max, y min, y max, number of non-finite coordinates
__global__ __launch_bounds__(BT) void k_minmax2_part(const double* __restrict__ c, int64_t n, double* __restrict__ part) {
  double xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY, bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x; i < n; i += (int64_t)gridDim.x * BT) {
    const double2 v = *reinterpret_cast<const double2*>(c + 2 * i);
    if (!isfinite(v.x) || !isfinite(v.y)) bad += 1.0;
    xmn = fmin(xmn, v.x);
    xmx = fmax(xmx, v.x);
    ymn = fmin(ymn, v.y);
    ymx = fmax(ymx, v.y);
  }
  __shared__ double sm[BT / 64][5];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double a = wmin(xmn), b = wmax(xmx), cc = wmin(ymn), d = wmax(ymx), e = bad;
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) e += __shfl_down(e, k, 64);
  if (lane == 0) {
    sm[w][0] = a;
    sm[w][1] = b;
    sm[w][2] = cc;
    sm[w][3] = d;
    sm[w][4] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BT / 64; k++) {
      sm[0][0] = fmin(sm[0][0], sm[k][0]);
      sm[0][1] = fmax(sm[0][1], sm[k][1]);
      sm[0][2] = fmin(sm[0][2], sm[k][2]);
      sm[0][3] = fmax(sm[0][3], sm[k][3]);
      sm[0][4] += sm[k][4];
    }
    for (int k = 0; k < 5; k++) part[(size_t)blockIdx.x * 5 + k] = sm[0][k];
  }
}
__global__ __launch_bounds__(BT) void k_minmax2_final(const double* __restrict__ part, int nb, double* __restrict__ out) {
  double a = INFINITY, b = -INFINITY, c = INFINITY, d = -INFINITY, e = 0;
  for (int k = threadIdx.x; k < nb; k += BT) {
    a = fmin(a, part[k * 5]);
    b = fmax(b, part[k * 5 + 1]);
    c = fmin(c, part[k * 5 + 2]);
    d = fmax(d, part[k * 5 + 3]);
    e += part[k * 5 + 4];
  }
  __shared__ double sm[BT / 64][5];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  a = wmin(a);
  b = wmax(b);
  c = wmin(c);
  d = wmax(d);
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) e += __shfl_down(e, k, 64);
  if (lane == 0) {
    sm[w][0] = a;
    sm[w][1] = b;
    sm[w][2] = c;
    sm[w][3] = d;
    sm[w][4] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BT / 64; k++) {
      sm[0][0] = fmin(sm[0][0], sm[k][0]);
      sm[0][1] = fmax(sm[0][1], sm[k][1]);
      sm[0][2] = fmin(sm[0][2], sm[k][2]);
      sm[0][3] = fmax(sm[0][3], sm[k][3]);
      sm[0][4] += sm[k][4];
    }
    for (int k = 0; k < 5; k++) out[k] = sm[0][k];
  }
}

// ---- select ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PT) void k_sel_init(SelState* __restrict__ st, uint32_t* __restrict__ ghist, uint64_t take,
                                                 uint32_t n) {
  for (uint32_t k = threadIdx.x; k < 4096u; k += PT) ghist[k] = 0u;
  if (threadIdx.x == 0) {
    st->phi = st->plo = 0ull;
    st->rank = take;
    st->count = n;
    st->pass = 0;
    st->collect = 0;
    st->done = 0;
    st->ncand = 0;
    st->nbig = 0;
    st->nbinfo = st->nslice = st->nvirt = st->nfall = 0;
  }
}

// mm: the bounds pass's result in device memory ([0] x min, [2] y min): the selection starts without a host round trip
__global__ __launch_bounds__(PT) void k_sel_hist(const double* __restrict__ key, int64_t n, const double* __restrict__ mm,
                                                 uint32_t chunk, uint32_t pass, const SelState* __restrict__ st,
                                                 uint32_t* __restrict__ ghist) {
  __shared__ uint32_t h[4096];
  if (st->done) return;
  const double x_Min = mm[0], y_Min = mm[2];
  for (uint32_t k = threadIdx.x; k < 4096u; k += PT) h[k] = 0u;
  __syncthreads();
  const unsigned long long phi = st->phi, plo = st->plo;
  const uint32_t nbits = 12u * pass;
  const int64_t first = (int64_t)blockIdx.x * chunk, last = min(first + (int64_t)chunk, n);
#pragma unroll 4
  for (int64_t i = first + threadIdx.x; i < last; i += PT) {
    const double2 v = *reinterpret_cast<const double2*>(key + 2 * i);
    const unsigned long long k = dkey(v.x, v.y, x_Min, y_Min);
    if (sel_cmp(k, (uint32_t)i, phi, plo, nbits) == 0) atomicAdd(&h[sel_digit(k, (uint32_t)i, pass)], 1u);
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < 4096u; k += PT)
    if (h[k]) atomicAdd(&ghist[k], h[k]);
}

// the bin of h [4096] whose cumulative range holds `rank` (1-based); all PT threads call it
__device__ __forceinline__ void find_bin(const uint32_t* h, unsigned long long rank, uint32_t* s_out /*[3]*/,
                                         uint32_t* wsum /*[PT / 64]*/) {
  const uint32_t t = threadIdx.x;
  uint32_t v[4];
#pragma unroll
  for (int q = 0; q < 4; q++) v[q] = h[4u * t + q];
  const uint32_t loc = v[0] + v[1] + v[2] + v[3];
  const int lane = t & 63, w = t >> 6;
  uint32_t inc = loc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t x = __shfl_up(inc, d, 64);
    if (lane >= d) inc += x;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  unsigned long long cum = inc - loc;
  for (int k = 0; k < w; k++) cum += wsum[k];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (cum < rank && rank <= cum + v[q]) {
      s_out[0] = 4u * t + q;
      s_out[1] = (uint32_t)cum;
      s_out[2] = v[q];
    }
    cum += v[q];
  }
  __syncthreads();
}

__global__ __launch_bounds__(PT) void k_sel_pick(SelState* __restrict__ st, uint32_t* __restrict__ ghist) {
  __shared__ uint32_t s_out[3], wsum[PT / 64];
  if (st->done) return;
  const unsigned long long rank = st->rank;
  const uint32_t pass = st->pass;
  unsigned long long phi = st->phi, plo = st->plo;
  __syncthreads();
  find_bin(ghist, rank, s_out, wsum);
#pragma unroll
  for (int q = 0; q < 4; q++) ghist[4u * threadIdx.x + q] = 0u;  // ready for the next pass
  if (threadIdx.x == 0) {
    sel_push(phi, plo, s_out[0], pass);
    st->phi = phi;
    st->plo = plo;
    st->rank = rank - s_out[1];
    st->count = s_out[2];
    st->pass = pass + 1;
    st->collect = s_out[2] <= CAND_CAP ? 1u : 0u;
    st->ncand = 0;
  }
}

// keys below the prefix are in the first block for sure (their x, y feed its extent); keys that match it are candidates
__global__ __launch_bounds__(PT) void k_sel_collect(const double* __restrict__ key, int64_t n, const double* __restrict__ mm,
                                                    uint32_t chunk, SelState* __restrict__ st, Cand* __restrict__ cand,
                                                    double* __restrict__ part) {
  if (st->done || !st->collect) return;
  const double x_Min = mm[0], y_Min = mm[2];
  const unsigned long long phi = st->phi, plo = st->plo;
  const uint32_t nbits = 12u * st->pass;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double mx = -INFINITY, my = -INFINITY;
  const int64_t first = (int64_t)blockIdx.x * chunk, last = min(first + (int64_t)chunk, n);
  for (int64_t i = first + threadIdx.x; i < last; i += PT) {
    const double2 v = *reinterpret_cast<const double2*>(key + 2 * i);
    const unsigned long long k = dkey(v.x, v.y, x_Min, y_Min);
    const int c = sel_cmp(k, (uint32_t)i, phi, plo, nbits);
    if (c < 0) {
      mx = fmax(mx, v.x);
      my = fmax(my, v.y);
    }
    const unsigned long long mask = __ballot(c == 0);
    if (mask) {  // one append per wave
      const int leader = __ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&st->ncand, (uint32_t)__popcll(mask));
      base = (uint32_t)__shfl((int)base, leader, 64);
      if (c == 0) {
        Cand cd;
        cd.key = k;
        cd.idx = (uint32_t)i;
        cd.pad = 0;
        cd.x = v.x;
        cd.y = v.y;
        cand[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = cd;
      }
    }
  }
  __shared__ double sm[PT / 64][2];
  mx = wmax(mx);
  my = wmax(my);
  if (lane == 0) {
    sm[w][0] = mx;
    sm[w][1] = my;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < PT / 64; k++) {
      mx = fmax(mx, sm[k][0]);
      my = fmax(my, sm[k][1]);
    }
    part[2 * blockIdx.x] = mx;
    part[2 * blockIdx.x + 1] = my;
  }
}

__global__ __launch_bounds__(PT) void k_sel_final(SelState* __restrict__ st, const Cand* __restrict__ cand,
                                                  const double* __restrict__ part, uint32_t npart) {
  __shared__ uint32_t h[4096];
  __shared__ uint32_t s_out[3], wsum[PT / 64], s_T;
  __shared__ double sm[PT / 64][2];
  if (st->done || !st->collect) return;
  const uint32_t c = st->ncand;
  unsigned long long phi = st->phi, plo = st->plo, rank = st->rank;
  uint32_t count = st->count, pass = st->pass;
  while (pass < 8u && count > 1u) {
    for (uint32_t k = threadIdx.x; k < 4096u; k += PT) h[k] = 0u;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < c; j += PT) {
      const unsigned long long k = cand[j].key;
      const uint32_t idx = cand[j].idx;
      if (sel_cmp(k, idx, phi, plo, 12u * pass) == 0) atomicAdd(&h[sel_digit(k, idx, pass)], 1u);
    }
    __syncthreads();
    find_bin(h, rank, s_out, wsum);
    sel_push(phi, plo, s_out[0], pass);
    rank -= s_out[1];
    count = s_out[2];
    pass++;
    __syncthreads();
  }
  // one candidate matches the prefix: the take-th smallest (d, index)
  for (uint32_t j = threadIdx.x; j < c; j += PT)
    if (sel_cmp(cand[j].key, cand[j].idx, phi, plo, 12u * pass) == 0) s_T = j;
  __syncthreads();
  const unsigned long long key_T = cand[s_T].key;
  const uint32_t idx_T = cand[s_T].idx;
  double mx = -INFINITY, my = -INFINITY;
  for (uint32_t j = threadIdx.x; j < c; j += PT) {
    const Cand cd = cand[j];
    if (cd.key < key_T || (cd.key == key_T && cd.idx <= idx_T)) {
      mx = fmax(mx, cd.x);
      my = fmax(my, cd.y);
    }
  }
  for (uint32_t k = threadIdx.x; k < npart; k += PT) {
    mx = fmax(mx, part[2 * k]);
    my = fmax(my, part[2 * k + 1]);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  mx = wmax(mx);
  my = wmax(my);
  if (lane == 0) {
    sm[w][0] = mx;
    sm[w][1] = my;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < PT / 64; k++) {
      mx = fmax(mx, sm[k][0]);
      my = fmax(my, sm[k][1]);
    }
    st->key_T = key_T;
    st->idx_T = idx_T;
    st->fx_max = mx;
    st->fy_max = my;
    st->done = 1;
  }
}

// ---- partition by block id ---------------------------------------------------------------------------------------
struct PartP {
  double x_Min, x_Max, y_Min, y_Max, cell_x, cell_y, inv_x, inv_y;
  int rows, cols;
};

// unique q with lo(q) < v <= hi(q) (Tools.getListByScale2: strict > on the low edge, <= on the high edge;
// last row / column stretched to the max), or -1.  lo/hi are evaluated exactly as FrmMain.cs:1262-1285 does.
__device__ __forceinline__ int find_axis(double v, double vmin, double vmax, double cellw, double inv_cellw, int cnt) {
  // a guess (the window below decides with the C#'s own expressions); the division only where 1 / cellw overflows
  const double g = isfinite(inv_cellw) ? (v - vmin) * inv_cellw : (v - vmin) / cellw;
  const long long q0 = isfinite(g) ? (long long)floor(g) : 0;
  // the intervals (lo, hi] are disjoint (hi(q) and lo(q + 1) are the same expression), so the order of the tests is free:
  // the guess first -- it is right for all but the points within a rounding of an edge
  const long long offs[5] = {0, -1, 1, -2, 2};
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const long long q = q0 + offs[k];
    if (q < 0 || q >= cnt) continue;
    const double lo = vmin + (double)(int)q * cellw;
    const double hi = (q == cnt - 1) ? vmax : vmin + (double)((int)q + 1) * cellw;
    if (v > lo && v <= hi) return (int)q;
  }
  {
    const int q = cnt - 1;
    const double lo = vmin + (double)q * cellw;
    if (v > lo && v <= vmax) return q;
  }
  return -1;
}

// The first block is rawData.Take(ptsInCell) of the list sorted by (d, index): exactly the points whose pair is <= that
// of the take-th element (key_T, idx_T).  Everybody else: the rectangle, rectangle 0 skipped (FrmMain.cs:1266).
__device__ __forceinline__ int32_t block_of_point(double x, double y, int64_t i, const PartP& P, unsigned long long key_T,
                                                  uint32_t idx_T, unsigned long long& k) {
  k = dkey(x, y, P.x_Min, P.y_Min);
  if (k < key_T || (k == key_T && (uint32_t)i <= idx_T)) return 0;  // cells[0] = rawData.Take(ptsInCell), :1254,1260
  const int q = find_axis(x, P.x_Min, P.x_Max, P.cell_x, P.inv_x, P.cols);
  const int p = find_axis(y, P.y_Min, P.y_Max, P.cell_y, P.inv_y, P.rows);
  if (p >= 0 && q >= 0) {
    const long long index = (long long)p * P.cols + q;
    if (index != 0) return (int32_t)index;
  }
  return -1;
}

__global__ __launch_bounds__(PT) void k_blk_hist(const double* __restrict__ key, int64_t n, PartP P, unsigned long long key_T,
                                                 uint32_t idx_T, uint32_t nblocks, uint32_t fsh, uint32_t NS, uint32_t chunk,
                                                 uint32_t nchunk, int32_t* __restrict__ blockof, uint32_t* __restrict__ counts) {
  (void)nchunk;
  extern __shared__ uint32_t h[];
  for (uint32_t k = threadIdx.x; k < NS; k += PT) h[k] = 0u;
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * chunk, last = min(first + (int64_t)chunk, n);
#pragma unroll 2
  for (int64_t i = first + threadIdx.x; i < last; i += PT) {
    const double2 v = *reinterpret_cast<const double2*>(key + 2 * i);
    unsigned long long k;
    const int32_t b = block_of_point(v.x, v.y, i, P, key_T, idx_T, k);
    blockof[i] = b;
    atomicAdd(&h[(b < 0 ? nblocks : (uint32_t)b) >> fsh], 1u);
  }
  __syncthreads();
  // (chunk-major: consecutive lanes, consecutive words -- k_transpose_u32 turns it bucket-major for the scan)
  for (uint32_t k = threadIdx.x; k < NS; k += PT) counts[(size_t)blockIdx.x * NS + k] = h[k];
}

__global__ __launch_bounds__(PT) void k_blk_scatter(const double* __restrict__ key, const double* __restrict__ motor, int64_t n,
                                                    double x_Min, double y_Min, uint32_t nblocks, uint32_t fsh, uint32_t NS,
                                                    uint32_t chunk, uint32_t nchunk, const int32_t* __restrict__ blockof,
                                                    const uint32_t* __restrict__ base, Rec32* __restrict__ rec, uint32_t S_lo,
                                                    uint32_t S_hi, uint32_t off0) {
  // only the super-buckets [S_lo, S_hi) are built (a rank's share; everything for a single device); record positions
  // are relative to the first of them (off0)
  extern __shared__ uint32_t h[];
  (void)nchunk;
  for (uint32_t k = S_lo + threadIdx.x; k < S_hi; k += PT) h[k] = base[(size_t)blockIdx.x * NS + k] - off0;  // (chunk-major copy)
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * chunk, last = min(first + (int64_t)chunk, n);
#pragma unroll 2
  for (int64_t i = first + threadIdx.x; i < last; i += PT) {
    const int32_t b = blockof[i];
    const uint32_t sb = (b < 0 ? nblocks : (uint32_t)b) >> fsh;
    if (sb < S_lo || sb >= S_hi) continue;
    const double2 kv = *reinterpret_cast<const double2*>(key + 2 * i);
    const double2 mv = *reinterpret_cast<const double2*>(motor + 2 * i);
    Rec32 r;
    r.x = mv.x;
    r.y = mv.y;
    r.key = dkey(kv.x, kv.y, x_Min, y_Min);
    r.idx = (uint32_t)i;
    r.blk = b < 0 ? nblocks : (uint32_t)b;
    rec[atomicAdd(&h[r.blk >> fsh], 1u)] = r;
  }
}

// A large block (more than VCP_BIG_BLOCK records: the heart of a blob holds 10^4) is cut once more, into V sub-ranges of d
// ("virtual blocks" of a few hundred records, in ascending d), so that the LDS kernel orders it too and many workgroups
// share the work: one workgroup walking 25 k records pass after pass set the time of the whole partition (0.54 ms).
struct BigInfo {
  uint32_t b, s, m, V, voff, pad;
  double dmin, scale;  // sub-range of d: min(V - 1, (d - dmin) * scale), monotone in d
};
struct Desc {  // a run of records to order: [s, e) of rec2 (src 0) or rec (src 1), all of block b
  uint32_t s, e, b, src;
};
constexpr uint32_t VF = 512;        // blocks per super-bucket up to which the split pass tracks the range of d per block
constexpr uint32_t VMAX = 4096;     // sub-ranges per large block at most
constexpr uint32_t SLICE = 4096;    // records per slice of a large block (one workgroup each in the count / move passes)
constexpr uint32_t VTARGET = 256;   // records per sub-range aimed at

__device__ __forceinline__ uint32_t vmap(const BigInfo& B, unsigned long long key) {
  const double v = (__longlong_as_double((long long)key) - B.dmin) * B.scale;  // (monotone in d; clamped at both ends)
  return v >= (double)B.V ? B.V - 1u : v > 0.0 ? (uint32_t)v : 0u;
}

// One workgroup per super-bucket: count its records per block in LDS, publish the block starts, move the records into
// block order (rec -> rec2; the super-bucket's records stay in the writing XCD's L2), list the large blocks and register
// their sub-ranges and slices.
__global__ __launch_bounds__(PT) void k_blk_split(const Rec32* __restrict__ rec, Rec32* __restrict__ rec2,
                                                  const uint32_t* __restrict__ base, const uint32_t* __restrict__ total,
                                                  uint32_t nchunk, uint32_t NS, uint32_t fsh, uint32_t nblocks,
                                                  uint32_t* __restrict__ blockstart, uint32_t* __restrict__ biglist,
                                                  SelState* __restrict__ st, BigInfo* __restrict__ binfo,
                                                  uint2* __restrict__ slicelist, Desc* __restrict__ fall, uint32_t S_lo,
                                                  uint32_t S_hi, uint32_t off0) {
  __shared__ uint32_t h[MAXF];
  __shared__ uint32_t wsum[PT / 64];
  __shared__ unsigned long long kmn[VF], kmx[VF];
  __shared__ uint32_t bcnt[VF], bst[VF];
  const uint32_t S = S_lo + blockIdx.x, F = 1u << fsh, fm = F - 1u;
  const bool track = F <= VF;
  const uint32_t s = base[(size_t)S * nchunk] - off0;
  const uint32_t e = ((S + 1 < NS) ? base[(size_t)(S + 1) * nchunk] : *total) - off0;
  for (uint32_t k = threadIdx.x; k < F; k += PT) h[k] = 0u;
  __syncthreads();
  for (uint32_t j = s + threadIdx.x; j < e; j += PT) atomicAdd(&h[rec[j].blk & fm], 1u);
  __syncthreads();
  // exclusive scan of h[0..F): thread t owns PERF consecutive entries
  {
    constexpr uint32_t PERF = MAXF / PT;
    const uint32_t t = threadIdx.x;
    uint32_t v[PERF], loc = 0;
#pragma unroll
    for (uint32_t q = 0; q < PERF; q++) {
      v[q] = PERF * t + q < F ? h[PERF * t + q] : 0u;
      loc += v[q];
    }
    const int lane = t & 63, w = t >> 6;
    uint32_t inc = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t x = __shfl_up(inc, d, 64);
      if (lane >= d) inc += x;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t cum = s + inc - loc;
    for (int k = 0; k < w; k++) cum += wsum[k];
#pragma unroll
    for (uint32_t q = 0; q < PERF; q++) {
      const uint32_t f = PERF * t + q;
      if (f < F) {
        const uint32_t vb = (S << fsh) + f;  // nblocks = the points in no block; nblocks + 1 = the end of everything
        if (vb <= nblocks + 1u) blockstart[vb] = cum;
        h[f] = cum;
        if (track) {
          bcnt[f] = vb < nblocks ? v[q] : 0u;
          bst[f] = cum;
          kmn[f] = ~0ull;
          kmx[f] = 0ull;
        } else if (vb < nblocks && v[q] > VCP_BIG_BLOCK) {
          biglist[atomicAdd(&st->nbig, 1u)] = vb;
          fall[atomicAdd(&st->nfall, 1u)] = Desc{cum, cum + v[q], vb, 0u};
        }
      }
      cum += v[q];
    }
    if (S + 1 == NS && t == 0) blockstart[nblocks + 1u] = e;  // (also when nblocks + 1 is a multiple of F)
    if (S + 1 == S_hi && S_hi < NS && t == 0) blockstart[S_hi << fsh] = e;  // the end of a rank's share
  }
  __syncthreads();
  for (uint32_t j = s + threadIdx.x; j < e; j += PT) {
    const Rec32 r = rec[j];
    const uint32_t f = r.blk & fm;
    rec2[atomicAdd(&h[f], 1u)] = r;
    if (track && bcnt[f] > VCP_BIG_BLOCK) {
      atomicMin(&kmn[f], r.key);
      atomicMax(&kmx[f], r.key);
    }
  }
  if (!track) return;
  __syncthreads();
  for (uint32_t f = threadIdx.x; f < F; f += PT) {
    const uint32_t m = bcnt[f];
    if (m <= VCP_BIG_BLOCK) continue;
    const uint32_t vb = (S << fsh) + f;
    biglist[atomicAdd(&st->nbig, 1u)] = vb;
    uint32_t V = 2u;
    while (V < VMAX && V * VTARGET < m) V <<= 1;
    const double dmin = __longlong_as_double((long long)kmn[f]);
    const double scale = (double)V / (__longlong_as_double((long long)kmx[f]) - dmin);
    if (!(kmx[f] > kmn[f]) || !(scale > 0.0) || !isfinite(scale) || (unsigned long long)V * VTARGET * 8ull < m) {
      // one value of d, or more records than the sub-ranges are meant for: the general kernel
      fall[atomicAdd(&st->nfall, 1u)] = Desc{bst[f], bst[f] + m, vb, 0u};
      continue;
    }
    BigInfo B;
    B.b = vb;
    B.s = bst[f];
    B.m = m;
    B.V = V;
    B.voff = atomicAdd(&st->nvirt, V);
    B.pad = 0;
    B.dmin = dmin;
    B.scale = scale;
    const uint32_t kb = atomicAdd(&st->nbinfo, 1u);
    binfo[kb] = B;
    const uint32_t nsl = (m + SLICE - 1u) / SLICE;
    const uint32_t so = atomicAdd(&st->nslice, nsl);
    for (uint32_t i = 0; i < nsl; i++) slicelist[so + i] = make_uint2(kb, i);
  }
}

// Up to MAXS blocks (+ the points in no block): every block is its own super-bucket, the scatter has left the records in
// block order and nothing has to be moved -- the block starts are the scanned counts, and a large block's sub-ranges of d
// are laid over the range of d its RECTANGLE allows (max of the offsets of its lower / upper corner from the minimum
// corner; the first block: [0, d of its last point]) instead of the range its points take: no pass over the records (the
// map only has to be monotone -- vmap clamps at both ends -- and a sub-range that comes out too full goes to the general
// kernel as before).  Replaces k_blk_split there (0.23 of the partition's 1.1 ms on the 10 M-point cloud).
__global__ __launch_bounds__(BT) void k_blk_starts(const uint32_t* __restrict__ base, const uint32_t* __restrict__ total,
                                                  uint32_t nchunk, uint32_t NS, uint32_t nblocks, PartP P,
                                                  unsigned long long key_T, uint32_t* __restrict__ blockstart,
                                                  uint32_t* __restrict__ biglist, SelState* __restrict__ st,
                                                  BigInfo* __restrict__ binfo, uint2* __restrict__ slicelist,
                                                  Desc* __restrict__ fall, uint32_t S_lo, uint32_t S_hi, uint32_t off0) {
  const uint32_t S = S_lo + blockIdx.x * BT + threadIdx.x;  // = the block id (NS = nblocks + 1: the last one = no block)
  if (S >= S_hi) return;
  const uint32_t s = base[(size_t)S * nchunk] - off0;
  const uint32_t e = ((S + 1 < NS) ? base[(size_t)(S + 1) * nchunk] : *total) - off0;
  blockstart[S] = s;
  if (S + 1 == NS) blockstart[nblocks + 1u] = e;
  if (S + 1 == S_hi && S_hi < NS) blockstart[S_hi] = e;  // the end of a rank's share
  const uint32_t m = e - s;
  if (S >= nblocks || m <= VCP_BIG_BLOCK) return;
  biglist[atomicAdd(&st->nbig, 1u)] = S;
  double dmin = 0.0, dmax = __longlong_as_double((long long)key_T);
  if (S != 0u) {  // the rectangle as FrmMain.cs:1262-1285 evaluates it (last row / column stretched to the max)
    const int p = (int)(S / (uint32_t)P.cols), q = (int)(S - (uint32_t)p * (uint32_t)P.cols);
    const double lox = (double)q * P.cell_x, hix = q == P.cols - 1 ? P.x_Max - P.x_Min : (double)(q + 1) * P.cell_x;
    const double loy = (double)p * P.cell_y, hiy = p == P.rows - 1 ? P.y_Max - P.y_Min : (double)(p + 1) * P.cell_y;
    dmin = fmax(lox, loy);
    dmax = fmax(hix, hiy);
  }
  uint32_t V = 2u;
  while (V < VMAX && V * VTARGET < m) V <<= 1;
  const double scale = (double)V / (dmax - dmin);
  if (!(dmax > dmin) || !(scale > 0.0) || !isfinite(scale) || (unsigned long long)V * VTARGET * 8ull < m) {
    // no usable range, or more records than the sub-ranges are meant for: the general kernel
    fall[atomicAdd(&st->nfall, 1u)] = Desc{s, e, S, 0u};
    return;
  }
  BigInfo B;
  B.b = S;
  B.s = s;
  B.m = m;
  B.V = V;
  B.voff = atomicAdd(&st->nvirt, V);
  B.pad = 0;
  B.dmin = dmin;
  B.scale = scale;
  const uint32_t kb = atomicAdd(&st->nbinfo, 1u);
  binfo[kb] = B;
  const uint32_t nsl = (m + SLICE - 1u) / SLICE;
  const uint32_t so = atomicAdd(&st->nslice, nsl);
  for (uint32_t i = 0; i < nsl; i++) slicelist[so + i] = make_uint2(kb, i);
}

// records per sub-range, slice by slice: LDS histogram, then one global add per sub-range the slice touches
__global__ __launch_bounds__(PT) void k_big_count(const Rec32* __restrict__ rec2, const BigInfo* __restrict__ binfo,
                                                  const uint2* __restrict__ slicelist, const SelState* __restrict__ st,
                                                  uint32_t* __restrict__ gcnt) {
  __shared__ uint32_t h[VMAX];
  const uint32_t ns = st->nslice;
  for (uint32_t q = blockIdx.x; q < ns; q += gridDim.x) {
    const uint2 sl = slicelist[q];
    const BigInfo B = binfo[sl.x];
    const uint32_t lo = B.s + sl.y * SLICE, hi = min(lo + SLICE, B.s + B.m);
    for (uint32_t k = threadIdx.x; k < B.V; k += PT) h[k] = 0u;
    __syncthreads();
    for (uint32_t j = lo + threadIdx.x; j < hi; j += PT) atomicAdd(&h[vmap(B, rec2[j].key)], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < B.V; k += PT)
      if (h[k]) atomicAdd(&gcnt[B.voff + k], h[k]);
    __syncthreads();
  }
}

// per large block: starts of its sub-ranges (the counters become the cursors of the move pass) and their descriptors
__global__ __launch_bounds__(BT) void k_big_scan(const BigInfo* __restrict__ binfo, const SelState* __restrict__ st,
                                                 uint32_t* __restrict__ gcnt, Desc* __restrict__ vlist) {
  __shared__ uint32_t wsum[BT / 64];
  constexpr uint32_t PERV = VMAX / BT;
  const uint32_t nbi = st->nbinfo;
  for (uint32_t q = blockIdx.x; q < nbi; q += gridDim.x) {
    const BigInfo B = binfo[q];
    const uint32_t t = threadIdx.x;
    uint32_t v[PERV], loc = 0;
#pragma unroll
    for (uint32_t k = 0; k < PERV; k++) {
      v[k] = t * PERV + k < B.V ? gcnt[B.voff + t * PERV + k] : 0u;
      loc += v[k];
    }
    const int lane = t & 63, w = t >> 6;
    uint32_t inc = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t x = __shfl_up(inc, d, 64);
      if (lane >= d) inc += x;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t cum = B.s + inc - loc;
    for (int k = 0; k < w; k++) cum += wsum[k];
#pragma unroll
    for (uint32_t k = 0; k < PERV; k++) {
      const uint32_t vi = t * PERV + k;
      if (vi < B.V) {
        gcnt[B.voff + vi] = cum;
        vlist[B.voff + vi] = Desc{cum, cum + v[k], B.b, 1u};
      }
      cum += v[k];
    }
    __syncthreads();
  }
}

// rec2 -> rec, sub-range by sub-range: each slice reserves its run in every sub-range it touches with one global add
__global__ __launch_bounds__(PT) void k_big_move(const Rec32* __restrict__ rec2, Rec32* __restrict__ rec,
                                                 const BigInfo* __restrict__ binfo, const uint2* __restrict__ slicelist,
                                                 const SelState* __restrict__ st, uint32_t* __restrict__ gcnt) {
  __shared__ uint32_t h[VMAX];
  constexpr int RPT = SLICE / PT;
  const uint32_t ns = st->nslice;
  for (uint32_t q = blockIdx.x; q < ns; q += gridDim.x) {
    const uint2 sl = slicelist[q];
    const BigInfo B = binfo[sl.x];
    const uint32_t lo = B.s + sl.y * SLICE, hi = min(lo + SLICE, B.s + B.m);
    for (uint32_t k = threadIdx.x; k < B.V; k += PT) h[k] = 0u;
    __syncthreads();
    Rec32 r[RPT];
    uint32_t v[RPT], rk[RPT];
#pragma unroll
    for (int u = 0; u < RPT; u++) {
      const uint32_t j = lo + threadIdx.x + (uint32_t)u * PT;
      if (j < hi) {
        r[u] = rec2[j];
        v[u] = vmap(B, r[u].key);
        rk[u] = atomicAdd(&h[v[u]], 1u);
      }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < B.V; k += PT) {
      const uint32_t c = h[k];
      if (c) h[k] = atomicAdd(&gcnt[B.voff + k], c);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RPT; u++) {
      const uint32_t j = lo + threadIdx.x + (uint32_t)u * PT;
      if (j < hi) rec[h[v[u]] + rk[u]] = r[u];
    }
    __syncthreads();
  }
}

// ---- order inside a block: ranks by distribution -----------------------------------------------------------------
struct SbMap {
  int by_idx;
  uint32_t nsb;
  double dmin, scale;
  uint32_t imin;
  unsigned long long ispan;
};
__device__ __forceinline__ uint32_t sb_of(const SbMap& M, unsigned long long key, uint32_t idx) {
  if (!M.by_idx) {
    // monotone in d: subtraction of a constant, product with a positive constant and truncation all are
    const double v = (__longlong_as_double((long long)key) - M.dmin) * M.scale;
    return v >= (double)M.nsb ? M.nsb - 1u : (uint32_t)v;  // (a NaN -- inf * 0 -- lands in 0 together with everybody)
  }
  return (uint32_t)(((unsigned long long)(idx - M.imin) * M.nsb) / M.ispan);
}
__device__ __forceinline__ SbMap sb_map(uint32_t m, uint32_t sbmax, unsigned long long kmin, unsigned long long kmax,
                                        uint32_t imin, uint32_t imax) {
  SbMap M;
  uint32_t want = m / 2u, nsb = 1u;
  while (nsb < want && nsb < sbmax) nsb <<= 1;
  M.nsb = nsb;
  M.dmin = __longlong_as_double((long long)kmin);
  const double span = __longlong_as_double((long long)kmax) - M.dmin;
  M.scale = (double)nsb / span;
  M.by_idx = !(kmax > kmin) || !(M.scale > 0.0) || !isfinite(M.scale);
  M.imin = imin;
  M.ispan = (unsigned long long)(imax - imin) + 1ull;
  return M;
}

// BIG = false: a workgroup of 256 per run of up to VCP_BIG_BLOCK records, everything in LDS and registers.  LIST = false:
// run = block blockIdx.x of rec2; LIST = true: the descriptors of the sub-ranges (a run that came out larger goes to the
// general kernel's list; the sub-range's counter is left at zero for the next call).
// BIG = true: workgroups of 1024 walk the general kernel's list: any size, ranks and the grouped keys through global
// memory (L2: a run is contiguous); also copies the indices of the points in no block.
template <bool BIG, bool LIST>
__global__ __launch_bounds__(BIG ? 1024 : 128) void k_blk_sort(const Rec32* __restrict__ rec2, const Rec32* __restrict__ rec,
                                                               const uint32_t* __restrict__ blockstart, uint32_t nblocks,
                                                               const Desc* __restrict__ list, SelState* __restrict__ st,
                                                               Desc* __restrict__ fall, uint32_t* __restrict__ gcnt,
                                                               KeyPart* __restrict__ stage_g, uint32_t* __restrict__ rk_g,
                                                               double* __restrict__ motor_bm, uint32_t* __restrict__ bl,
                                                               uint32_t* __restrict__ blk_t, uint32_t b_lo, int has_dropped,
                                                               int32_t* __restrict__ grp_big, uint32_t brute_thr,
                                                               uint32_t nshare) {
  // nshare (LIST = BIG = false): blocks of the share; the grid is capped and walks them (a launch of one workgroup per
  // block wraps at 2^32 threads: 33 M blocks)
  constexpr int NT = BIG ? 1024 : 128;
  constexpr uint32_t SBMAX = BIG ? 8192u : 512u;
  constexpr uint32_t PER = SBMAX / NT;
  constexpr int RPT = BIG ? 1 : (int)(VCP_BIG_BLOCK / NT);  // records per thread kept in registers (small runs)
  __shared__ uint32_t cnt[SBMAX + 1];
  __shared__ KeyPart stage_l[BIG ? 1 : VCP_BIG_BLOCK];
  __shared__ unsigned long long s_k[2];
  __shared__ uint32_t s_i[2];
  __shared__ uint32_t wsum[NT / 64];
  if (BIG && has_dropped) {  // the points in no block (FrmMain.cs:1266; Tools.cs:512): only their indices are wanted, behind the m others
    const uint32_t s = blockstart[nblocks], e = blockstart[nblocks + 1];
    for (uint32_t j = s + blockIdx.x * NT + threadIdx.x; j < e; j += gridDim.x * NT) bl[j] = rec2[j].idx;
  }
  const uint32_t nlist = BIG ? st->nfall : LIST ? st->nvirt : nshare;
  for (uint32_t q = blockIdx.x; q < nlist; q += gridDim.x) {
    uint32_t b, s, e;
    const Rec32* src = rec2;
    if (BIG || LIST) {
      const Desc d = list[q];
      b = d.b;
      s = d.s;
      e = d.e;
      if (d.src) src = rec;
      if (!BIG) {
        if (threadIdx.x == 0) {
          gcnt[q] = 0u;  // left at zero for the next call
          if (e - s > VCP_BIG_BLOCK) fall[atomicAdd(&st->nfall, 1u)] = d;
        }
        if (e - s > VCP_BIG_BLOCK || e == s) continue;  // (uniform over the workgroup)
      }
    } else {
      b = b_lo + q;
      s = blockstart[b];
      e = blockstart[b + 1];
      if (e == s || e - s > VCP_BIG_BLOCK) continue;  // (uniform over the workgroup)
    }
    const uint32_t m = e - s;
    // what the grid engine reads as the point's group: the points of blocks small enough for the all-pairs kernel are not its
    const int32_t gb = ((BIG || LIST) ? blockstart[b + 1] - blockstart[b] : m) > brute_thr ? (int32_t)b : -1;
    if (threadIdx.x == 0) {
      s_k[0] = ~0ull;
      s_k[1] = 0ull;
      s_i[0] = 0xFFFFFFFFu;
      s_i[1] = 0u;
    }
    for (uint32_t k = threadIdx.x; k <= SBMAX; k += NT) cnt[k] = 0u;
    __syncthreads();
    Rec32 r[RPT];
    {
      unsigned long long kmn = ~0ull, kmx = 0ull;
      uint32_t imn = 0xFFFFFFFFu, imx = 0u;
      if constexpr (BIG) {
        for (uint32_t j = threadIdx.x; j < m; j += NT) {
          const Rec32 x = src[s + j];
          kmn = min(kmn, x.key);
          kmx = max(kmx, x.key);
          imn = min(imn, x.idx);
          imx = max(imx, x.idx);
        }
      } else {
#pragma unroll
        for (int u = 0; u < RPT; u++) {
          const uint32_t j = threadIdx.x + (uint32_t)u * NT;
          if (j < m) {
            r[u] = src[s + j];
            kmn = min(kmn, r[u].key);
            kmx = max(kmx, r[u].key);
            imn = min(imn, r[u].idx);
            imx = max(imx, r[u].idx);
          }
        }
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        kmn = min(kmn, (unsigned long long)__shfl_xor((long long)kmn, d, 64));
        kmx = max(kmx, (unsigned long long)__shfl_xor((long long)kmx, d, 64));
        imn = min(imn, (uint32_t)__shfl_xor((int)imn, d, 64));
        imx = max(imx, (uint32_t)__shfl_xor((int)imx, d, 64));
      }
      if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_k[0], kmn);
        atomicMax(&s_k[1], kmx);
        atomicMin(&s_i[0], imn);
        atomicMax(&s_i[1], imx);
      }
    }
    __syncthreads();
    const SbMap M = sb_map(m, SBMAX, s_k[0], s_k[1], s_i[0], s_i[1]);
    // count per sub-bucket; the returning atomic is the record's rank inside its group
    uint32_t sbv[RPT], rkv[RPT];
    if constexpr (BIG) {
      for (uint32_t j = threadIdx.x; j < m; j += NT) {
        const Rec32 x = src[s + j];
        rk_g[s + j] = atomicAdd(&cnt[sb_of(M, x.key, x.idx)], 1u);
      }
    } else {
#pragma unroll
      for (int u = 0; u < RPT; u++) {
        const uint32_t j = threadIdx.x + (uint32_t)u * NT;
        if (j < m) {
          sbv[u] = sb_of(M, r[u].key, r[u].idx);
          rkv[u] = atomicAdd(&cnt[sbv[u]], 1u);
        }
      }
    }
    __syncthreads();
    {  // exclusive scan of cnt[0..SBMAX): thread t owns PER consecutive entries; cnt[SBMAX] <- m
      const uint32_t t = threadIdx.x;
      uint32_t v[PER], loc = 0;
#pragma unroll
      for (uint32_t k = 0; k < PER; k++) {
        v[k] = cnt[t * PER + k];
        loc += v[k];
      }
      const int lane = t & 63, w = t >> 6;
      uint32_t inc = loc;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(inc, d, 64);
        if (lane >= d) inc += x;
      }
      if (lane == 63) wsum[w] = inc;
      __syncthreads();
      uint32_t cum = inc - loc;
      for (int k = 0; k < w; k++) cum += wsum[k];
#pragma unroll
      for (uint32_t k = 0; k < PER; k++) {
        cnt[t * PER + k] = cum;
        cum += v[k];
      }
      if (t == NT - 1) cnt[SBMAX] = cum;
    }
    __syncthreads();
    // the keys, group by group (in no particular order inside a group)
    if constexpr (BIG) {
      for (uint32_t j = threadIdx.x; j < m; j += NT) {
        const Rec32 x = src[s + j];
        KeyPart kp;
        kp.key = x.key;
        kp.idx = x.idx;
        kp.src = j;
        stage_g[s + cnt[sb_of(M, x.key, x.idx)] + rk_g[s + j]] = kp;
      }
      __threadfence_block();
    } else {
#pragma unroll
      for (int u = 0; u < RPT; u++) {
        const uint32_t j = threadIdx.x + (uint32_t)u * NT;
        if (j < m) {
          KeyPart kp;
          kp.key = r[u].key;
          kp.idx = r[u].idx;
          kp.src = j;
          stage_l[cnt[sbv[u]] + rkv[u]] = kp;
        }
      }
    }
    __syncthreads();
    // final place = start of the group + members with a smaller (d, index)
    const KeyPart* stg = BIG ? stage_g + s : stage_l;
    if constexpr (BIG) {
      for (uint32_t j = threadIdx.x; j < m; j += NT) {
        const Rec32 x = src[s + j];
        const uint32_t g = sb_of(M, x.key, x.idx);
        const uint32_t gs = cnt[g], ge = cnt[g + 1];
        uint32_t c = 0;
        for (uint32_t k = gs; k < ge; k++) {
          const KeyPart o = stg[k];
          c += (o.key < x.key || (o.key == x.key && o.idx < x.idx)) ? 1u : 0u;
        }
        const uint32_t p = s + gs + c;
        *reinterpret_cast<double2*>(motor_bm + 2 * (size_t)p) = make_double2(x.x, x.y);
        bl[p] = x.idx;
        blk_t[p] = b;
        if (grp_big) grp_big[p] = gb;
      }
    } else {
#pragma unroll
      for (int u = 0; u < RPT; u++) {
        const uint32_t j = threadIdx.x + (uint32_t)u * NT;
        if (j < m) {
          const uint32_t gs = cnt[sbv[u]], ge = cnt[sbv[u] + 1];
          uint32_t c = 0;
          for (uint32_t k = gs; k < ge; k++) {
            const KeyPart o = stg[k];
            c += (o.key < r[u].key || (o.key == r[u].key && o.idx < r[u].idx)) ? 1u : 0u;
          }
          const uint32_t p = s + gs + c;
          *reinterpret_cast<double2*>(motor_bm + 2 * (size_t)p) = make_double2(r[u].x, r[u].y);
          bl[p] = r[u].idx;
          blk_t[p] = b;
          if (grp_big) grp_big[p] = gb;
        }
      }
    }
    __syncthreads();  // (lists: the LDS counters are reused by the next run)
  }
}

}  // namespace

int vcp_blocks_ens(vcp_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return VCP_OK;
  if (b.p) {
    VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VCP_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return vcp_fail(ctx, VCP_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  b.cap = want;
  return VCP_OK;
}

namespace {
// out [cols][rows] = in [rows][cols]^T, 32 x 32 tiles through LDS.  The per-chunk counts are written and read chunk-major by
// the chunked passes (a workgroup's row: coalesced) and scanned bucket-major: with 27 k super-buckets the strided form of
// those accesses had doubled the histogram pass (64 -> 124 us).
__global__ __launch_bounds__(BT) void k_transpose_u32(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t rows,
                                                     uint32_t cols) {
  __shared__ uint32_t t[32][33];
  const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;  // 32 x 8
  const uint32_t c0 = blockIdx.x * 32u, r0 = blockIdx.y * 32u;
  for (uint32_t j = ty; j < 32u; j += 8u)
    if (r0 + j < rows && c0 + tx < cols) t[j][tx] = in[(size_t)(r0 + j) * cols + c0 + tx];
  __syncthreads();
  for (uint32_t j = ty; j < 32u; j += 8u)
    if (c0 + j < cols && r0 + tx < rows) out[(size_t)(c0 + j) * rows + r0 + tx] = t[tx][j];
}

// starts of the super-buckets in the list of all n points, dense (the scanned counts hold them at a stride)
__global__ __launch_bounds__(BT) void k_sb_starts(const uint32_t* __restrict__ base, const uint32_t* __restrict__ total,
                                                  uint32_t nchunk, uint32_t NS, uint32_t* __restrict__ sbstart) {
  const uint32_t S = blockIdx.x * BT + threadIdx.x;
  if (S < NS) sbstart[S] = base[(size_t)S * nchunk];
  else if (S == NS) sbstart[S] = *total;
}

}  // namespace

// Stage 1, the same on every rank: bounds, the first block, the block of every point and the population of every
// super-bucket (streaming passes over the whole list).  want_cuts: also bring the super-bucket starts to the host
// (vcp_blocks_plan_cuts needs them; a single device building everything does not).
int vcp_blocks_plan(vcp_ctx* ctx, BlocksState* s, const double* d_key, const double* d_motor, int64_t n, int pts_in_cell,
                    bool want_cuts) {
  hipStream_t st = ctx->stream;
  const bool keyed = d_key != d_motor;
  s->planned = false;  // (set again at the end: a plan that fails half way leaves nothing a build could use)
  s->built = false;
  s->d_key = d_key;
  s->d_motor = d_motor;
  // bounds (FrmMain.cs:1224-1227) and the finiteness check
  const int rb = (int)vcp_blocks(n, BT, 1024);
  VCP_TRY(vcp_blocks_ens(ctx, s->misc, (size_t)(rb * 5 + 64) * 8));
  double* part = s->misc.as<double>();
  double* out = part + (size_t)rb * 5;
  double* h = reinterpret_cast<double*>(ctx->pinned);
  // The bounds stay on the device for the selection's first pass (k_sel_hist / k_sel_collect read them there) and come to
  // the host together with its result: one read-back for bounds + selection instead of two (three with a separate key).
  double* out_motor = out + 8;
  if (keyed) {
    // a non-finite motor coordinate would reach DBImproved only; the partition's own check below covers the keys
    hipLaunchKernelGGL(k_minmax2_part, dim3(rb), dim3(BT), 0, st, d_motor, n, part);
    hipLaunchKernelGGL(k_minmax2_final, dim3(1), dim3(BT), 0, st, part, rb, out_motor);
  }
  hipLaunchKernelGGL(k_minmax2_part, dim3(rb), dim3(BT), 0, st, d_key, n, part);
  hipLaunchKernelGGL(k_minmax2_final, dim3(1), dim3(BT), 0, st, part, rb, out);

  // chunks of the passes over the list
  int64_t chunk = (n + 255) / 256;
  if (chunk < PCH_MIN) chunk = PCH_MIN;
  chunk = (chunk + PT - 1) / PT * PT;
  const uint32_t nchunk = (uint32_t)((n + chunk - 1) / chunk);
  s->chunk = (uint32_t)chunk;
  s->nchunk = nchunk;

  // the take-th smallest (d, index) and the extent of the first block (FrmMain.cs:1229-1258)
  s->take = (int)std::min<int64_t>(pts_in_cell, n);
  const size_t sel_bytes = 256 + 4096 * 4 + (size_t)nchunk * 16;
  VCP_TRY(vcp_blocks_ens(ctx, s->sel, sel_bytes));
  VCP_TRY(vcp_blocks_ens(ctx, s->cand, (size_t)CAND_CAP * sizeof(Cand)));
  SelState* d_sel = s->sel.as<SelState>();
  uint32_t* ghist = reinterpret_cast<uint32_t*>(s->sel.as<char>() + 256);
  double* selpart = reinterpret_cast<double*>(s->sel.as<char>() + 256 + 4096 * 4);
  static_assert(sizeof(SelState) <= 256, "SelState");
  hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(PT), 0, st, d_sel, ghist, (uint64_t)s->take, (uint32_t)n);
  SelState* hs = reinterpret_cast<SelState*>(reinterpret_cast<char*>(ctx->pinned) + 1024);
  bool done = false;
  for (uint32_t pass = 0; pass < 8 && !done; pass++) {
    hipLaunchKernelGGL(k_sel_hist, dim3(nchunk), dim3(PT), 0, st, d_key, n, out, (uint32_t)chunk, pass, d_sel, ghist);
    hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(PT), 0, st, d_sel, ghist);
    hipLaunchKernelGGL(k_sel_collect, dim3(nchunk), dim3(PT), 0, st, d_key, n, out, (uint32_t)chunk, d_sel,
                       s->cand.as<Cand>(), selpart);
    hipLaunchKernelGGL(k_sel_final, dim3(1), dim3(PT), 0, st, d_sel, s->cand.as<Cand>(), selpart, nchunk);
    VCP_HIP(ctx, hipMemcpyAsync(hs, d_sel, sizeof(SelState), hipMemcpyDeviceToHost, st));
    if (pass == 0) VCP_HIP(ctx, hipMemcpyAsync(h, out, 16 * 8, hipMemcpyDeviceToHost, st));  // both sets of bounds
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (pass == 0) {
      if (keyed && h[8 + 4] != 0.0) return vcp_fail(ctx, VCP_ERR_ARG, "non-finite motor coordinates");
      if (h[4] != 0.0) return vcp_fail(ctx, VCP_ERR_ARG, "non-finite partition coordinates");
      s->x_Min = h[0];
      s->x_Max = h[1];
      s->y_Min = h[2];
      s->y_Max = h[3];
      for (int k = 0; k < 4; k++) s->mbox[k] = keyed ? h[8 + k] : h[k];
    }
    done = hs->done != 0;
  }
  if (!done) return vcp_fail(ctx, VCP_ERR_HIP, "the selection of the first block did not finish");
  s->key_T = hs->key_T;
  s->idx_T = hs->idx_T;
  s->cell_x = hs->fx_max - s->x_Min;
  s->cell_y = hs->fy_max - s->y_Min;
  const double fr = (s->y_Max - s->y_Min) / s->cell_y, fc = (s->x_Max - s->x_Min) / s->cell_x;
  if (!std::isfinite(fr) || !std::isfinite(fc))
    return vcp_fail(ctx, VCP_ERR_DEGENERATE, "first block has zero extent: rows/cols undefined (FrmMain.cs:1256-1259)");
  if (fr >= 2147483646.0 || fc >= 2147483646.0) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "rows/cols overflow int");
  s->rows = (int)fr + 1;
  s->cols = (int)fc + 1;
  s->nblocks = (int64_t)s->rows * s->cols;
  // (2^26 - 4: the kernels with a wave per block launch nblocks * 64 threads, and a launch holds fewer than 2^32)
  if (s->nblocks > ((int64_t)1 << 26) - 4) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "%lld blocks", (long long)s->nblocks);

  // block of every point (FrmMain.cs:1259-1285, Tools.cs:510-513) and the population of every super-bucket
  const PartP P{s->x_Min, s->x_Max, s->y_Min, s->y_Max, s->cell_x, s->cell_y, 1.0 / s->cell_x, 1.0 / s->cell_y, s->rows, s->cols};
  const uint32_t nblocks = (uint32_t)s->nblocks;
  const uint64_t nb1 = (uint64_t)nblocks + 1;  // + the points in no block
  uint32_t fsh = 0;
  while (((nb1 + (1ull << fsh) - 1) >> fsh) > MAXS) fsh++;  // nblocks <= 2^26: 2^fsh <= MAXF
  if ((1u << fsh) > MAXF) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "%lld blocks", (long long)s->nblocks);
  const uint32_t NS = (uint32_t)((nb1 + (1ull << fsh) - 1) >> fsh);
  s->fsh = fsh;
  s->NS = NS;
  const size_t nc = (size_t)NS * nchunk;
  VCP_TRY(vcp_blocks_ens(ctx, s->blockof, (size_t)n * 4));
  VCP_TRY(vcp_blocks_ens(ctx, s->counts, (nc + 8) * 4));
  VCP_TRY(vcp_blocks_ens(ctx, s->counts_t, (nc + 8) * 4));
  VCP_TRY(vcp_blocks_ens(ctx, s->blockstart, (size_t)(nb1 + 2) * 4));
  uint32_t* counts = s->counts.as<uint32_t>();
  uint32_t* total = counts + nc;
  if ((size_t)NS * 4 > 65536) {  // (more than 64 KB of dynamic LDS has to be allowed per kernel)
    VCP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_blk_hist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(MAXS * 4)));
    VCP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_blk_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(MAXS * 4)));
  }
  hipLaunchKernelGGL(k_blk_hist, dim3(nchunk), dim3(PT), (size_t)NS * 4, st, d_key, n, P, s->key_T, s->idx_T, nblocks, fsh, NS,
                     (uint32_t)chunk, nchunk, s->blockof.as<int32_t>(), s->counts_t.as<uint32_t>());
  // chunk-major counts -> bucket-major, scanned (the positions of every (bucket, chunk) run), and back for the scatter
  hipLaunchKernelGGL(k_transpose_u32, dim3((NS + 31) / 32, (nchunk + 31) / 32), dim3(BT), 0, st, s->counts_t.as<uint32_t>(),
                     counts, nchunk, NS);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, counts, counts, (int64_t)nc, total));
  hipLaunchKernelGGL(k_transpose_u32, dim3((nchunk + 31) / 32, (NS + 31) / 32), dim3(BT), 0, st, counts,
                     s->counts_t.as<uint32_t>(), NS, nchunk);
  VCP_HIP(ctx, hipGetLastError());
  s->h_sbstart.clear();
  if (want_cuts) {
    VCP_TRY(vcp_blocks_ens(ctx, s->sbstart, ((size_t)NS + 2) * 4));
    hipLaunchKernelGGL(k_sb_starts, dim3(vcp_blocks((int64_t)NS + 1, BT)), dim3(BT), 0, st, counts, total, nchunk, NS,
                       s->sbstart.as<uint32_t>());
    s->h_sbstart.resize((size_t)NS + 1);
    VCP_HIP(ctx, hipMemcpyAsync(s->h_sbstart.data(), s->sbstart.p, ((size_t)NS + 1) * 4, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
  }
  s->planned = true;
  return VCP_OK;
}

// Stage 2: the block-major list of the super-buckets [S_lo, S_hi) -- a rank's share, cut at super-bucket boundaries
// (vcp_blocks_plan_cuts), or everything.  n_loc = the points in that share (known to the caller: n for everything).
int vcp_blocks_build(vcp_ctx* ctx, BlocksState* s, uint32_t S_lo, uint32_t S_hi, uint32_t off0, int64_t n_loc) {
  hipStream_t st = ctx->stream;
  if (!s->planned) return vcp_fail(ctx, VCP_ERR_ARG, "the partition has not been planned");
  if (S_lo > S_hi || S_hi > s->NS) return vcp_fail(ctx, VCP_ERR_ARG, "share of super-buckets");
  const int64_t n = s->n;
  const uint32_t nblocks = (uint32_t)s->nblocks, fsh = s->fsh, NS = s->NS, nchunk = s->nchunk;
  const size_t nc = (size_t)NS * nchunk;
  const size_t nl = (size_t)(n_loc > 0 ? n_loc : 1);
  s->S_lo = S_lo;
  s->S_hi = S_hi;
  s->b_lo = (int64_t)S_lo << fsh;
  s->b_hi = std::min<int64_t>((int64_t)S_hi << fsh, s->nblocks);
  s->n_loc = n_loc;
  const bool has_dropped = S_hi == NS;
  SelState* d_sel = s->sel.as<SelState>();
  SelState* hs = reinterpret_cast<SelState*>(reinterpret_cast<char*>(ctx->pinned) + 1024);
  VCP_TRY(vcp_blocks_ens(ctx, s->rec, nl * sizeof(Rec32)));
  VCP_TRY(vcp_blocks_ens(ctx, s->rec2, nl * sizeof(Rec32)));
  VCP_TRY(vcp_blocks_ens(ctx, s->stage, nl * sizeof(KeyPart)));
  VCP_TRY(vcp_blocks_ens(ctx, s->rank, nl * 4));
  VCP_TRY(vcp_blocks_ens(ctx, s->bl, nl * 4));
  VCP_TRY(vcp_blocks_ens(ctx, s->motor_bm, nl * 16));
  VCP_TRY(vcp_blocks_ens(ctx, s->blk_t, (nl + 1) * 4));
  {
    // blocks of up to this many points go to the all-pairs kernel (VCP_BRUTE_MAX: test switch, 0 = every block to the engine)
    static const int brute_env = [] {
      const char* e = getenv("VCP_BRUTE_MAX");
      const int v = e ? atoi(e) : 1024;
      return v < 0 ? 0 : v > (int)VCP_BRUTE_MAX ? (int)VCP_BRUTE_MAX : v;
    }();
    s->brute_thr = (uint32_t)brute_env;
    if (s->brute_thr) VCP_TRY(vcp_blocks_ens(ctx, s->grp_big, (nl + 1) * 4));
  }
  // large blocks: descriptors, slices, sub-ranges (V <= m / 128 each, at least 2) and the general kernel's list
  const size_t nbig_cap = nl / VCP_BIG_BLOCK + 8, nvirt_cap = nl / 64 + 2 * nbig_cap + 64;
  VCP_TRY(vcp_blocks_ens(ctx, s->biglist, nbig_cap * 4));
  VCP_TRY(vcp_blocks_ens(ctx, s->binfo, nbig_cap * sizeof(BigInfo)));
  VCP_TRY(vcp_blocks_ens(ctx, s->slicelist, (nbig_cap + nl / SLICE + 8) * sizeof(uint2)));
  VCP_TRY(vcp_blocks_ens(ctx, s->vlist, nvirt_cap * sizeof(Desc)));
  VCP_TRY(vcp_blocks_ens(ctx, s->fall, (nvirt_cap + nbig_cap) * sizeof(Desc)));
  {
    // the sub-range counters are zero between calls (their last reader leaves them so): cleared only when the array is
    // new or a call did not get to its end
    const void* before = s->gcnt.p;
    VCP_TRY(vcp_blocks_ens(ctx, s->gcnt, nvirt_cap * 4));
    if (s->gcnt.p != before || !s->virt_clean) VCP_HIP(ctx, hipMemsetAsync(s->gcnt.p, 0, s->gcnt.cap, st));
    s->virt_clean = false;
  }
  uint32_t* counts = s->counts.as<uint32_t>();
  uint32_t* total = counts + nc;
  const size_t lds_h = (size_t)NS * 4;
  Rec32* rec = s->rec.as<Rec32>();
  Rec32* rec2 = s->rec2.as<Rec32>();
  uint32_t* blockstart = s->blockstart.as<uint32_t>();
  uint32_t* gcnt = s->gcnt.as<uint32_t>();
  Desc* vlist = s->vlist.as<Desc>();
  Desc* fall = s->fall.as<Desc>();
  const uint32_t b_lo = (uint32_t)s->b_lo;
  const unsigned nbl = (unsigned)std::max<int64_t>(s->b_hi - s->b_lo, 1);
  if (S_hi > S_lo) {
    hipLaunchKernelGGL(k_blk_scatter, dim3(nchunk), dim3(PT), lds_h, st, s->d_key, s->d_motor, n, s->x_Min, s->y_Min, nblocks,
                       fsh, NS, s->chunk, nchunk, s->blockof.as<int32_t>(), s->counts_t.as<uint32_t>(), rec, S_lo, S_hi, off0);
    if (fsh == 0) {
      // every block its own super-bucket: the scattered records ARE the block-major list; the two arrays swap roles
      const PartP P{s->x_Min, s->x_Max, s->y_Min, s->y_Max, s->cell_x, s->cell_y, 1.0 / s->cell_x, 1.0 / s->cell_y, s->rows,
                    s->cols};
      hipLaunchKernelGGL(k_blk_starts, dim3(vcp_blocks((int64_t)(S_hi - S_lo), BT)), dim3(BT), 0, st, counts, total, nchunk, NS,
                         nblocks, P, s->key_T, blockstart, s->biglist.as<uint32_t>(), d_sel, s->binfo.as<BigInfo>(),
                         s->slicelist.as<uint2>(), fall, S_lo, S_hi, off0);
      std::swap(rec, rec2);
    } else {
      hipLaunchKernelGGL(k_blk_split, dim3(S_hi - S_lo), dim3(PT), 0, st, rec, rec2, counts, total, nchunk, NS, fsh, nblocks,
                         blockstart, s->biglist.as<uint32_t>(), d_sel, s->binfo.as<BigInfo>(), s->slicelist.as<uint2>(), fall,
                         S_lo, S_hi, off0);
    }
#define VCP_SORT_ARGS(list) rec2, rec, blockstart, nblocks, list, d_sel, fall, gcnt, s->stage.as<KeyPart>(),              \
                            s->rank.as<uint32_t>(), s->motor_bm.as<double>(), s->bl.as<uint32_t>(), s->blk_t.as<uint32_t>(), \
                            b_lo, has_dropped ? 1 : 0, s->brute_thr ? s->grp_big.as<int32_t>() : nullptr, s->brute_thr, \
                            (uint32_t)(s->b_hi - s->b_lo)
    if (s->b_hi > s->b_lo)
      hipLaunchKernelGGL((k_blk_sort<false, false>), dim3(std::min(nbl, 1u << 22)), dim3(128), 0, st, VCP_SORT_ARGS(nullptr));
    const unsigned gsl = (unsigned)std::min<size_t>(2048, nbig_cap + nl / SLICE);
    hipLaunchKernelGGL(k_big_count, dim3(gsl), dim3(PT), 0, st, rec2, s->binfo.as<BigInfo>(), s->slicelist.as<uint2>(), d_sel,
                       gcnt);
    hipLaunchKernelGGL(k_big_scan, dim3((unsigned)std::min<size_t>(1024, nbig_cap)), dim3(BT), 0, st, s->binfo.as<BigInfo>(),
                       d_sel, gcnt, vlist);
    hipLaunchKernelGGL(k_big_move, dim3(gsl), dim3(PT), 0, st, rec2, rec, s->binfo.as<BigInfo>(), s->slicelist.as<uint2>(),
                       d_sel, gcnt);
    hipLaunchKernelGGL((k_blk_sort<false, true>), dim3((unsigned)std::min<size_t>(8192, nvirt_cap)), dim3(128), 0, st,
                       VCP_SORT_ARGS(vlist));
    hipLaunchKernelGGL((k_blk_sort<true, true>), dim3(256), dim3(1024), 0, st, VCP_SORT_ARGS(fall));
#undef VCP_SORT_ARGS
  }
  VCP_HIP(ctx, hipGetLastError());
  // host copy of the share's block starts [b_lo, b_hi] (positions relative to the share)
  s->h_blockstart.assign((size_t)nblocks + 2, 0u);
  if (S_hi > S_lo) {
    const size_t cnt = (size_t)(s->b_hi - s->b_lo) + 1 + (has_dropped ? 1 : 0);
    VCP_HIP(ctx, hipMemcpyAsync(s->h_blockstart.data() + s->b_lo, blockstart + s->b_lo, cnt * 4, hipMemcpyDeviceToHost, st));
  }
  VCP_HIP(ctx, hipMemcpyAsync(hs, d_sel, sizeof(SelState), hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  s->m = S_hi > S_lo ? s->h_blockstart[(size_t)s->b_hi] : 0;
  s->nbig = hs->nbig;
  s->virt_clean = true;
  s->built = true;
  return VCP_OK;
}

int vcp_blocks_partition(vcp_ctx* ctx, BlocksState* s, const double* d_key, const double* d_motor, int64_t n,
                         int pts_in_cell) {
  VCP_TRY(vcp_blocks_plan(ctx, s, d_key, d_motor, n, pts_in_cell, false));
  return vcp_blocks_build(ctx, s, 0u, s->NS, 0u, n);
}
