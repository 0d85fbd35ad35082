// gridbuild.hip -- caller order <-> cell order by two-level partition (MI355X, gfx950).
//
// The round-1 build sorted (cell id, index) pairs with a library radix sort and then gathered the coordinates by
// index: every gathered point paid a 128-byte line for 16 bytes (3.7x the algorithmic traffic), and the output pass
// paid the same on the way back (9x).  Here a 16-byte record TRAVELS with the index: the point's binary32 coordinates
// relative to the grid origin -- the values the grid bins on and the search kernels screen with (grid_common.hpp:
// rel32) -- and its index.  The binary64 coordinates stay where the caller put them; the exact re-test of a screened
// pair reads them by index (ExactSrc).
//
//   k_part_hist     per chunk of >= 8192 points: bucket histogram in LDS (bucket = 2^csh consecutive cell ids, i.e. a band
//                   of grid rows) -> counts[bucket][chunk]
//   (scan)          exclusive scan of counts in bucket-major order = where each chunk's share of each bucket starts
//   k_part_scatter  per chunk: the scanned counts become per-bucket cursors in LDS; one returning LDS atomic per point
//                   gives the slot of its record in the bucket-major record array -- each chunk owns a contiguous run
//                   per bucket, so partial lines are completed in the writing XCD's L2
//   (k_part_split)  grids of more than 8192 buckets (very sparse: C5's 1.4 G cells): the coarse passes run on
//                   super-buckets of up to 32 buckets and this pass splits each into its buckets, cursors in LDS
//   k_part_fine     one workgroup per bucket: count the bucket's records per cell in LDS, scan, store the bucket's
//                   slice of the CELL TABLE (grid_common.hpp: CellTab -- two bit planes + a start per word of 32 cells,
//                   full starts only for words with a populous cell; written exactly once, never zero-filled), then
//                   assemble the bucket in LDS in its final order and store it as full lines: binary32 coordinates,
//                   list position, and the optional per-point extras.
//
// and back (vcp_grid_output_partition): (list position, label word) pairs partitioned by windows of 2^OWSH list
// positions, every window then written by workgroups that share an XCD (b and b+8 share an L2).
//
// The order of points inside a cell follows the LDS atomics and may differ from run to run; every result of the engine
// is order-free by construction (numbering by smallest LIST position, border rule by max id), which
// tests/test_determinism_gpu.py re-checks on the GPU.
#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "grid_common.hpp"

using namespace vcpg;

namespace {

constexpr int PT = 1024;           // threads per workgroup, histogram / scatter passes (one workgroup per CU at 256 chunks:
                                   // 512 threads measured 0.061 / 0.179 ms for the two passes, 1024 threads 0.043 / 0.153-0.175)
constexpr int PCH_MIN = 8192;      // smallest chunk
constexpr int FT = 1024;           // threads per workgroup, fine pass
constexpr uint32_t MAXB = 8192;    // buckets (LDS histogram of the coarse passes: 32 KB)
// fine pass: a bucket of at most wcap records is staged in LDS in its final order (counters + staging <= 160 KB)
constexpr uint32_t wcap(int) { return 4096u; }
constexpr uint32_t QUEUE_FROM = 32;  // buckets of more windows than this hand them to k_part_fine_windows (at 6 windows per
                                     // bucket -- 140 M uniform points -- the queue form was measured slower: 10.9 against 7.2 ms)

// One record of the bucket-major intermediate: 16 bytes = the binary32 coordinates relative to the grid origin (what
// the grid bins on and what the search kernels screen with, grid_common.hpp: rel32) + the point's index; in 2-D the
// fourth word carries the cell id, in 3-D the fine pass recomputes it from the three coordinates.  The binary64
// coordinates do NOT travel: the exact re-test of a screened pair (rare) reads them from the caller's array by index.
// One 16-byte store per lane (a divergent store costs per instruction and per line touched -- tools/micro/part_bench.hip:
// 10 M records into 4096 buckets take 0.23 ms at 16 B, 0.28 ms at 32 B, 0.33 ms as separate 16 + 4 B arrays).
typedef float4 Rec;

template <int GD>
__device__ __forceinline__ Rec rec_make(const float* qf, uint32_t idx, uint32_t cell) {
  return GD == 2 ? make_float4(qf[0], qf[1], __uint_as_float(idx), __uint_as_float(cell))
                 : make_float4(qf[0], qf[1], qf[2], __uint_as_float(idx));
}
// -> cell id; idx = the point's index
template <int GD>
__device__ __forceinline__ uint32_t rec_cell(const Rec& r, const GridP& g, uint32_t& idx) {
  if (GD == 2) {
    idx = __float_as_uint(r.z);
    return __float_as_uint(r.w);
  }
  idx = __float_as_uint(r.w);
  const float qf[3] = {r.x, r.y, r.z};
  int cc[3];
  return cell_of32<3>(qf, g, cc);
}

// caller-order point i -> its binary32 relative coordinates and cell
template <int GD>
__device__ __forceinline__ uint32_t point_cell(const double* __restrict__ c, int64_t i, int stride, const GridP& g, float* qf) {
  double q[3];
  int cc[3];
  load_in<GD>(c, i, stride, q);
#pragma unroll
  for (int a = 0; a < GD; a++) qf[a] = rel32(q[a], g.mn[a], g.scale);
  return cell_of32<GD>(qf, g, cc);
}

struct PartGeom {
  uint32_t csh;     // log2(cells per bucket)
  uint32_t B;       // buckets
  uint32_t a;       // log2(buckets per super-bucket): 0 unless B > MAXB
  uint32_t NS;      // super-buckets = what the coarse passes split into (= B when a == 0)
  uint32_t chunk;   // points per chunk (a multiple of PT)
  uint32_t nchunk;
};

inline int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  int v = e ? atoi(e) : dflt;
  return v < lo ? lo : v > hi ? hi : v;
}

inline PartGeom part_geom(int64_t n, uint32_t ncells) {
  static const int csh_max = env_int("VCP_CPB_LOG2", 14, 10, 14);
  static const int target = env_int("VCP_PART_CHUNKS", 256, 64, 8192);  // long runs per (bucket, chunk) matter more
  PartGeom p;                                                           // than workgroups per CU (part_bench)
  // cells per bucket: 2^14 on the sparse grids of this path (tens of cells per point); on a coarse grid (eps well above
  // the point spacing: the 10 M-point cloud at eps 0.7 has 1 M cells) that would leave a few dozen buckets of 10^5 records
  // for 256 CUs -- the fine pass took 5.0 ms there, 0.3 ms with 2^10 -- so: about 4096 records per bucket, 2^10 .. 2^14 cells
  int csh = csh_max;
  while (csh > 10 && ((uint64_t)ncells >> csh) * 4096ull < (uint64_t)n) csh--;
  p.csh = (uint32_t)csh;
  p.B = (uint32_t)((((uint64_t)ncells + 1) + ((1ull << p.csh) - 1)) >> p.csh);  // the table has ncells + 1 entries
  p.a = 0;
  while (((p.B + (1u << p.a) - 1u) >> p.a) > MAXB) p.a++;  // ncells < 2^31, csh >= 10: a <= 8; with csh 14 a <= 4
  p.NS = (p.B + (1u << p.a) - 1u) >> p.a;
  int64_t chunk = (n + target - 1) / target;
  if (chunk < PCH_MIN) chunk = PCH_MIN;
  chunk = (chunk + PT - 1) / PT * PT;
  p.chunk = (uint32_t)chunk;
  p.nchunk = (uint32_t)((n + chunk - 1) / chunk);
  return p;
}

template <int GD, bool GROUPED>
__global__ __launch_bounds__(PT) void k_part_hist(const double* __restrict__ c, int64_t n, int stride, GridP g,
                                                 const int32_t* __restrict__ group, int glo, int ghi, uint32_t csh,
                                                 uint32_t B, uint32_t chunk, uint32_t nchunk,
                                                 uint32_t* __restrict__ counts, uint32_t* __restrict__ qcount,
                                                 uint32_t* __restrict__ ctcount) {
  extern __shared__ uint32_t h[];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *qcount = 0u;      // the fine pass's queue of windows (k_part_fine_windows)
    ctcount[0] = 0u;   // populous words of the cell table handed out (k_part_fine)
  }
  for (uint32_t k = threadIdx.x; k < B; k += PT) h[k] = 0;
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * chunk;
  const int64_t last = min(first + (int64_t)chunk, n);
#pragma unroll 4
  for (int64_t i = first + threadIdx.x; i < last; i += PT) {
    if (GROUPED) {
      const int gg = group[i];
      if (gg < glo || gg >= ghi) continue;
    }
    float qf[3];
    atomicAdd(&h[point_cell<GD>(c, i, stride, g, qf) >> csh], 1u);
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < B; k += PT) counts[(size_t)k * nchunk + blockIdx.x] = h[k];
}

// The scanned counts say where this chunk's share of every bucket starts: loaded into LDS they are per-bucket
// cursors, and one returning LDS atomic per point yields the record's final slot.
template <int GD, bool GROUPED>
__global__ __launch_bounds__(PT) void k_part_scatter(const double* __restrict__ c, int64_t n, int stride, GridP g,
                                                    const int32_t* __restrict__ group, int glo, int ghi, uint32_t csh,
                                                    uint32_t B, uint32_t chunk, uint32_t nchunk,
                                                    const uint32_t* __restrict__ base, Rec* __restrict__ rec,
                                                    uint32_t* __restrict__ pos) {
  extern __shared__ uint32_t h[];
  for (uint32_t k = threadIdx.x; k < B; k += PT) h[k] = base[(size_t)k * nchunk + blockIdx.x];
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * chunk;
  const int64_t last = min(first + (int64_t)chunk, n);
#pragma unroll 4
  for (int64_t i = first + threadIdx.x; i < last; i += PT) {
    if (GROUPED) {
      const int gg = group[i];
      if (gg < glo || gg >= ghi) {
        if (pos) pos[i] = NONE;  // left out of this call
        continue;
      }
    }
    float qf[3];
    const uint32_t cell = point_cell<GD>(c, i, stride, g, qf);
    rec[atomicAdd(&h[cell >> csh], 1u)] = rec_make<GD>(qf, (uint32_t)i, cell);
  }
}

// one pad word per 32 counters: a thread's run of consecutive counters in the scan then walks all banks
__device__ __forceinline__ uint32_t padded(uint32_t i) { return i + (i >> 5); }

// One workgroup per bucket.  LDS: the bucket's cell counters, then (behind them) a staging area of WCAP records.
// Every bucket is assembled in LDS in its final order and stored with consecutive lanes on consecutive addresses (a
// scattered 16-byte store per record was measured 7x slower than the same bytes as full lines).  A bucket of up to
// WCAP records takes one window: the scanned counters serve as per-cell cursors.  A larger one (dense clouds, the heart
// of a cluster) keeps, from the counting pass, each record's RANK inside its cell (rk, 4 B per record), so that its
// final position start[cell] + rank is known without a cursor, and walks its records once per window of WCAP
// positions (the records of a bucket stay L2-resident).
template <int GD, bool GROUPED, int NT = FT>
__device__ __forceinline__ void fine_flush(uint32_t s0, uint32_t m, const float* s32, const uint32_t* sidx,
                                           const uint32_t* __restrict__ ord, const uint8_t* __restrict__ in_classed,
                                           const int32_t* __restrict__ group, float* __restrict__ sorted32,
                                           uint32_t* __restrict__ sord, uint32_t* __restrict__ sidx_out,
                                           int32_t* __restrict__ sgroup, uint8_t* __restrict__ flags,
                                           uint32_t* __restrict__ pos) {
  if (GD == 2) {
    const float2* src = reinterpret_cast<const float2*>(s32);
    float2* dst = reinterpret_cast<float2*>(sorted32) + s0;
    for (uint32_t k = threadIdx.x; k < m; k += NT) dst[k] = src[k];
  } else {
    const float4* src = reinterpret_cast<const float4*>(s32);
    float4* dst = reinterpret_cast<float4*>(sorted32) + s0;
    for (uint32_t k = threadIdx.x; k < m; k += NT) dst[k] = src[k];
  }
  for (uint32_t k = threadIdx.x; k < m; k += NT) {
    const uint32_t i = sidx[k], p = s0 + k;
    sord[p] = ord ? ord[i] : i;
    if (sidx_out) sidx_out[p] = i;
    if (GROUPED) sgroup[p] = group[i];
    if (in_classed) flags[p] = in_classed[i] ? F_CLASSED : 0;
    if (pos) pos[i] = p;
  }
}

// staging store of one record at window slot p
template <int GD>
__device__ __forceinline__ void stage_put(float* s32, uint32_t* sidx, uint32_t p, const Rec& r, uint32_t idx) {
  if (GD == 2) reinterpret_cast<float2*>(s32)[p] = make_float2(r.x, r.y);
  else reinterpret_cast<float4*>(s32)[p] = make_float4(r.x, r.y, r.z, 0.0f);
  sidx[p] = idx;
}

// first / one-past-last record of bucket b: from the scanned (bucket, chunk) counts of the coarse passes (stride =
// chunks) or, after k_part_split, from the array of bucket starts (stride 1); the last bucket ends at *total
__device__ __forceinline__ void bucket_range(const uint32_t* __restrict__ bstart, uint32_t bstride, const uint32_t* __restrict__ total,
                                             uint32_t b, uint32_t B, uint32_t& s, uint32_t& e) {
  s = bstart[(size_t)b * bstride];
  e = (b + 1 < B) ? bstart[(size_t)(b + 1) * bstride] : *total;
}

// Grids of more than MAXB buckets: the coarse passes ran on super-buckets of 2^a consecutive buckets.  One workgroup per
// super-bucket counts its records per bucket in LDS (at most 32 counters), publishes the bucket starts and moves the
// records into bucket order (rec -> rec2; the super-bucket's ~10^4 records stay in the writing XCD's L2).
template <int GD>
__global__ __launch_bounds__(FT) void k_part_split(const Rec* __restrict__ rec, Rec* __restrict__ rec2,
                                                  const uint32_t* __restrict__ base, const uint32_t* __restrict__ total,
                                                  uint32_t nchunk, uint32_t NS, uint32_t a, uint32_t csh, uint32_t B, GridP g,
                                                  uint32_t* __restrict__ bstart) {
  __shared__ uint32_t h[32];
  const uint32_t S = blockIdx.x;
  uint32_t s, e;
  bucket_range(base, nchunk, total, S, NS, s, e);
  const uint32_t subm = (1u << a) - 1u;
  if (threadIdx.x < 32) h[threadIdx.x] = 0u;
  __syncthreads();
  for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
    uint32_t i;
    atomicAdd(&h[(rec_cell<GD>(rec[j], g, i) >> csh) & subm], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const uint32_t v = h[threadIdx.x];
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if ((int)threadIdx.x >= d) inc += t;
    }
    const uint32_t first = s + inc - v;
    h[threadIdx.x] = first;
    const uint32_t bk = (S << a) + threadIdx.x;
    if (threadIdx.x <= subm && bk <= B) bstart[bk] = first;  // bstart[B] = *total: behind the last bucket nothing follows
  }
  __syncthreads();
  for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
    const Rec r = rec[j];
    uint32_t i;
    rec2[atomicAdd(&h[(rec_cell<GD>(r, g, i) >> csh) & subm], 1u)] = r;
  }
}

// ---- fine pass, small buckets -------------------------------------------------------------------------------
// A bucket of up to SCAP records is worked at the granularity of the cell table's WORDS (512 per bucket) instead of its
// 2^14 cells: the dense per-cell counters of k_part_fine cost the same ~11 us per bucket however few records it holds
// (zero, scan and read back 16 k counters at one workgroup per CU), which made the fine pass the largest phase on sparse
// grids (noise pass of the block pipeline: 6 k buckets of ~800 records; C5: 85 k buckets of ~600).  Here: records are
// counted and grouped per word; a word whose cells all hold at most two records (the sparse background) gets its bit
// planes from a short sequential walk by one thread and its records their final place from the planes; a POPULOUS word
// (some cell with three or more: the inside of a blob) gets 32 cell counters in LDS, worked by a half-wave.  Work is
// O(records + words), 50 KB of LDS: three workgroups per CU.  Buckets with more than SCAP records or more than SPOP
// populous words are left to k_part_fine (bstate[b] = 1).
constexpr int FS = 512;            // threads
constexpr uint32_t scap(int gd) { return gd == 2 ? 2048u : 1536u; }  // records: 42 / 50 KB of LDS, three workgroups per CU
// (4096 records and 256 populous words -- 75 KB, two workgroups per CU -- was measured: C4 fine pass 0.20 ms either way,
// C5 1.65 against 1.22 ms, noise pass of the block pipeline 0.160 against 0.131 ms)
constexpr uint32_t SPOP = 128;     // populous words

template <int GD, bool GROUPED>
__global__ __launch_bounds__(FS) void k_part_fine_small(const Rec* __restrict__ rec, const uint32_t* __restrict__ bstart,
                                                       uint32_t bstride, const uint32_t* __restrict__ total, uint32_t B,
                                                       uint32_t csh, GridP g, const uint32_t* __restrict__ ord,
                                                       const uint8_t* __restrict__ in_classed, const int32_t* __restrict__ group,
                                                       uint4* __restrict__ ctwords, uint32_t* __restrict__ ctdense,
                                                       uint32_t* __restrict__ ctcount, uint8_t* __restrict__ bstate,
                                                       float* __restrict__ sorted32, uint32_t* __restrict__ sord,
                                                       uint32_t* __restrict__ sidx_out, int32_t* __restrict__ sgroup,
                                                       uint8_t* __restrict__ flags, uint32_t* __restrict__ pos) {
  constexpr uint32_t FPR = GD == 2 ? 2 : 4;
  constexpr uint32_t SCAP = scap(GD);
  constexpr int SRPT = SCAP / FS;             // records per thread, kept in registers between the phases
  __shared__ uint32_t ws[512 + 2];            // records per word, then first position of each word (+ the bucket's end)
  __shared__ uint32_t wlo[512], whi[512];     // the words' bit planes
  __shared__ uint16_t wpi[512];               // populous words: index among the bucket's populous words
  __shared__ uint8_t sci[SCAP];               // cell inside the word of each record, word-grouped order
  __shared__ __attribute__((aligned(16))) uint16_t st32[SPOP][32];         // populous words: cell counters -> starts -> cursors, relative to the bucket
  __shared__ __attribute__((aligned(16))) float s32[SCAP * FPR];  // final order: coordinates ...
  __shared__ uint32_t sidx[SCAP];                                 // ... and the points' indices
  __shared__ uint32_t wsum[FS / 64];
  __shared__ uint32_t s_np, s_dbase;
  const uint32_t b = blockIdx.x;
  uint32_t s, e;
  bucket_range(bstart, bstride, total, b, B, s, e);
  const uint32_t m = e - s;
  if (b + 1 == B && threadIdx.x == 0) ctcount[1] = *total;  // points in the grid (CellTab::nin)
  if (m > SCAP) {                                            // k_part_fine's
    if (threadIdx.x == 0) bstate[b] = 1;
    return;
  }
  const uint32_t CPB = 1u << csh, NW = CPB >> 5, c0 = b << csh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t k = threadIdx.x; k < NW + 1; k += FS) ws[k] = 0u;
  __syncthreads();
  // A. count per word; the record stays in registers
  Rec r[SRPT];
  uint32_t idx[SRPT], lw[SRPT], ci[SRPT], rnk[SRPT];
#pragma unroll
  for (int k = 0; k < SRPT; k++) {
    const uint32_t j = s + (uint32_t)k * FS + threadIdx.x;
    lw[k] = NONE;
    if (j < e) {
      r[k] = rec[j];
      const uint32_t c = rec_cell<GD>(r[k], g, idx[k]) - c0;
      lw[k] = c >> 5;
      ci[k] = c & 31u;
      rnk[k] = atomicAdd(&ws[lw[k]], 1u);
    }
  }
  __syncthreads();
  // B. word starts: exclusive scan of the NW (<= 512 = FS) counts
  {
    const uint32_t v = threadIdx.x < NW ? ws[threadIdx.x] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t pre = inc - v;
    for (int k = 0; k < w; k++) pre += wsum[k];
    if (threadIdx.x < NW) ws[threadIdx.x] = s + pre;
    if (threadIdx.x == 0) ws[NW] = e;
  }
  __syncthreads();
  // C. the cells of a word's records, side by side
#pragma unroll
  for (int k = 0; k < SRPT; k++)
    if (lw[k] != NONE) sci[ws[lw[k]] - s + rnk[k]] = (uint8_t)ci[k];
  __syncthreads();
  // D. bit planes of the words that can have them from a short walk (more than 64 records: some cell holds three)
  bool popl = false;
  uint32_t lo = 0u, hi = 0u;
  if (threadIdx.x < NW) {
    const uint32_t a = ws[threadIdx.x] - s, k = ws[threadIdx.x + 1] - s - a;
    if (k > 64u) {
      popl = true;
    } else {
      for (uint32_t i = 0; i < k; i++) {
        const uint32_t bit = 1u << sci[a + i];
        if (lo & hi & bit) continue;  // saturated at three
        if (lo & bit) hi ^= bit;      // 1 -> 2, (2 -> 3 below)
        lo ^= bit;
      }
      popl = (lo & hi) != 0u;
    }
  }
  {
    const unsigned long long dm = __ballot(popl);
    if (lane == 0) wsum[w] = (uint32_t)__popcll(dm);
    __syncthreads();
    uint32_t before = (uint32_t)__popcll(dm & ((1ull << lane) - 1ull)), tot = 0;
    for (int k = 0; k < FS / 64; k++) {
      if (k < w) before += wsum[k];
      tot += wsum[k];
    }
    if (threadIdx.x == 0) s_np = tot;
    if (threadIdx.x < NW) wpi[threadIdx.x] = popl ? (uint16_t)before : (uint16_t)0xFFFFu;
  }
  __syncthreads();
  const uint32_t np = s_np;
  if (np > SPOP) {  // a bucket in the thick of a blob: the dense pass does it (uniform over the workgroup)
    if (threadIdx.x == 0) bstate[b] = 1;
    return;
  }
  if (threadIdx.x == 0) {
    bstate[b] = 0;
    s_dbase = np ? atomicAdd(&ctcount[0], np) : 0u;
  }
  for (uint32_t k = threadIdx.x; k < np * 16u; k += FS) reinterpret_cast<uint32_t*>(&st32[0][0])[k] = 0u;
  if (threadIdx.x < NW && !popl) {
    wlo[threadIdx.x] = lo;
    whi[threadIdx.x] = hi;
  }
  __syncthreads();
  // E. populous words: count per cell ...
#pragma unroll
  for (int k = 0; k < SRPT; k++)
    if (lw[k] != NONE && wpi[lw[k]] != 0xFFFFu) {  // 16-bit counters: an atomic add on the containing word
      uint16_t* c = &st32[wpi[lw[k]]][ci[k]];
      atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(c) & ~(uintptr_t)3), ((uintptr_t)c & 2) ? 0x10000u : 1u);
    }
  __syncthreads();
  // ... starts, bit planes and the full slice of the table, one half-wave per populous word
  {
    const uint32_t hw = threadIdx.x >> 5, li = threadIdx.x & 31u;
    for (uint32_t wd = hw; wd < NW; wd += FS / 32) {  // (uniform per half-wave)
      const uint32_t pi = wpi[wd];
      if (pi == 0xFFFFu) continue;
      const uint32_t c = st32[pi][li];
      uint32_t inc = c;  // (a cell of a bucket of <= 4096 records holds < 65536)
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 32);
        if ((int)li >= d) inc += t;
      }
      const uint32_t first = ws[wd] + inc - c;
      st32[pi][li] = (uint16_t)(first - s);  // cursor
      ctdense[(size_t)(s_dbase + pi) * 32u + li] = first;
      const uint32_t cs = min(c, 3u);
      // planes by a 32-lane ballot: the half-wave's bits of the wave-wide mask
      const unsigned long long blo = __ballot(cs & 1u), bhi = __ballot(cs >> 1);
      const int sh = (threadIdx.x & 32) ? 32 : 0;
      if (li == 0) {
        wlo[wd] = (uint32_t)(blo >> sh);
        whi[wd] = (uint32_t)(bhi >> sh);
      }
    }
  }
  __syncthreads();
  // F. every record's final place
#pragma unroll
  for (int k = 0; k < SRPT; k++) {
    if (lw[k] == NONE) continue;
    const uint32_t pi = wpi[lw[k]];
    uint32_t p;
    if (pi != 0xFFFFu) {
      uint16_t* c = &st32[pi][ci[k]];
      const bool up = ((uintptr_t)c & 2) != 0;
      const uint32_t old = atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(c) & ~(uintptr_t)3), up ? 0x10000u : 1u);
      p = up ? old >> 16 : old & 0xFFFFu;
    } else {
      const uint32_t l = wlo[lw[k]], h = whi[lw[k]], below = (1u << ci[k]) - 1u;
      const uint32_t a = ws[lw[k]] - s;
      p = a + (uint32_t)__popc(l & below) + 2u * (uint32_t)__popc(h & below);
      if ((h >> ci[k]) & 1u)  // two records in this cell: the one met first in the word goes first
        for (uint32_t i = 0; i < rnk[k]; i++) p += sci[a + i] == (uint8_t)ci[k] ? 1u : 0u;
    }
    stage_put<GD>(s32, sidx, p, r[k], idx[k]);
  }
  // the bucket's words of the table
  if (threadIdx.x < NW) {
    const size_t W = (size_t)(c0 >> 5) + threadIdx.x;
    const uint32_t pi = wpi[threadIdx.x];
    if (W < vcp_ct_words(g.ncells))
      ctwords[W] = make_uint4(wlo[threadIdx.x], whi[threadIdx.x], ws[threadIdx.x], pi != 0xFFFFu ? s_dbase + pi : NONE);
  }
  __syncthreads();
  fine_flush<GD, GROUPED, FS>(s, m, s32, sidx, ord, in_classed, group, sorted32, sord, sidx_out, sgroup, flags, pos);
}

template <int GD, bool GROUPED>
__global__ __launch_bounds__(FT) void k_part_fine(const Rec* __restrict__ rec, uint32_t* __restrict__ rk,
                                                 const uint32_t* __restrict__ bstart, uint32_t bstride,
                                                 const uint32_t* __restrict__ total, uint32_t B, uint32_t csh, GridP g,
                                                 const uint32_t* __restrict__ ord, const uint8_t* __restrict__ in_classed,
                                                 const int32_t* __restrict__ group, uint4* __restrict__ ctwords,
                                                 uint32_t* __restrict__ ctdense, uint32_t* __restrict__ ctcount,
                                                 float* __restrict__ sorted32, uint32_t* __restrict__ sord,
                                                 uint32_t* __restrict__ sidx_out, int32_t* __restrict__ sgroup,
                                                 uint8_t* __restrict__ flags, uint32_t* __restrict__ pos,
                                                 uint2* __restrict__ queue, uint32_t* __restrict__ qcount,
                                                 const uint8_t* __restrict__ bstate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  __shared__ uint32_t wsum[FT / 64];
  __shared__ uint32_t w_lo[512], w_hi[512], w_slot[512];  // the bucket's words of the cell table (2^14 cells at most)
  __shared__ uint32_t s_dbase;
  const uint32_t b = blockIdx.x;
  if (bstate[b] == 0) return;  // done by k_part_fine_small
  constexpr uint32_t WCAP = wcap(GD);
  constexpr uint32_t FPR = GD == 2 ? 2 : 4;  // staged floats per record
  const uint32_t CPB = 1u << csh;
  uint32_t* cnt = reinterpret_cast<uint32_t*>(lds);                                  // [padded(CPB)]
  float* s32 = reinterpret_cast<float*>(lds + (((size_t)padded(CPB) + 4) & ~3ull) * 4);  // [WCAP * FPR]
  uint32_t* sidx = reinterpret_cast<uint32_t*>(s32 + (size_t)WCAP * FPR);            // [WCAP]
  const uint32_t c0 = b << csh;
  uint32_t s, e;
  bucket_range(bstart, bstride, total, b, B, s, e);
  const uint32_t m = e - s;
  const bool big = m > WCAP;
  for (uint32_t k = threadIdx.x; k < padded(CPB); k += FT) cnt[k] = 0;
  __syncthreads();
  for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
    uint32_t i;
    uint32_t* slot = &cnt[padded(rec_cell<GD>(rec[j], g, i) - c0)];
    if (big) rk[j] = atomicAdd(slot, 1u);  // rank inside the cell
    else atomicAdd(slot, 1u);
  }
  __syncthreads();
  // exclusive scan of the CPB counters, offset by the bucket's first position: thread t owns PER consecutive ones
  const uint32_t PER = CPB / FT;
  const uint32_t first = threadIdx.x * PER;
  uint32_t loc = 0;
  for (uint32_t k = 0; k < PER; k++) loc += cnt[padded(first + k)];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = loc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t pre = s + inc - loc;
  for (int k = 0; k < w; k++) pre += wsum[k];
  // second pass of the scan; the thread's run of PER cells is part of ONE word of the cell table (PER <= 16): its bits of
  // the two population planes are built on the way and OR-ed together over the 32 / PER threads that share the word
  const uint32_t NW = CPB >> 5;
  uint32_t plo = 0u, phi = 0u;
  const uint32_t wstart = pre;  // start of the thread's first cell
  {
    const uint32_t bit0 = first & 31u;
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t v = cnt[padded(first + k)];
      cnt[padded(first + k)] = pre;
      pre += v;
      const uint32_t c = min(v, 3u);
      plo |= (c & 1u) << (bit0 + k);
      phi |= (c >> 1) << (bit0 + k);
    }
    for (uint32_t d = 1; d < 32u / PER; d <<= 1) {
      plo |= (uint32_t)__shfl_xor((int)plo, (int)d, 64);
      phi |= (uint32_t)__shfl_xor((int)phi, (int)d, 64);
    }
    if ((first & 31u) == 0u) {  // the word's first thread
      w_lo[first >> 5] = plo;
      w_hi[first >> 5] = phi;
      w_slot[first >> 5] = wstart;  // (its slot is settled below; until then the start of the word's first cell)
    }
  }
  __syncthreads();
  {
    // populous words (some cell with 3 or more points) keep their 32 starts in full: they are numbered inside the
    // workgroup and take consecutive slots behind ONE atomic per bucket (a single hot word serialises at ~11 ns)
    const bool popl = threadIdx.x < NW && (w_lo[threadIdx.x] & w_hi[threadIdx.x]) != 0u;
    const unsigned long long dm = __ballot(popl);
    if (lane == 0) wsum[w] = (uint32_t)__popcll(dm);
    __syncthreads();
    uint32_t before = (uint32_t)__popcll(dm & ((1ull << lane) - 1ull)), tot = 0;
    for (int k = 0; k < FT / 64; k++) {
      if (k < w) before += wsum[k];
      tot += wsum[k];
    }
    if (threadIdx.x == 0) s_dbase = tot ? atomicAdd(&ctcount[0], tot) : 0u;
    __syncthreads();
    if (threadIdx.x < NW) {
      const uint32_t slot = popl ? s_dbase + before : NONE;
      const uint32_t wpos = w_slot[threadIdx.x];
      w_slot[threadIdx.x] = slot;
      const size_t W = (size_t)(c0 >> 5) + threadIdx.x;
      if (W < vcp_ct_words(g.ncells))
        ctwords[W] = make_uint4(w_lo[threadIdx.x], w_hi[threadIdx.x], wpos, slot);
    }
    __syncthreads();
    const uint32_t hw = threadIdx.x >> 5, li = threadIdx.x & 31u;  // half-waves: one populous word (128 bytes) each
    for (uint32_t j = hw; j < NW; j += FT / 32) {
      const uint32_t slot = w_slot[j];
      if (slot != NONE) ctdense[(size_t)slot * 32u + li] = cnt[padded(32u * j + li)];
    }
  }
  __syncthreads();
  if (!big) {
    // one window: the scanned counters serve as per-cell cursors
    for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
      const Rec r = rec[j];
      uint32_t i;
      const uint32_t p = atomicAdd(&cnt[padded(rec_cell<GD>(r, g, i) - c0)], 1u) - s;
      stage_put<GD>(s32, sidx, p, r, i);
    }
    __syncthreads();
    fine_flush<GD, GROUPED>(s, m, s32, sidx, ord, in_classed, group, sorted32, sord, sidx_out, sgroup, flags, pos);
    return;
  }
  if (m > QUEUE_FROM * WCAP) {
    // many windows: other workgroups assemble them (k_part_fine_windows), from the ranks left in rk and the bucket's
    // slice of the cell table just stored -- this one would walk its m records once per window all by itself (a cloud
    // that sits in ONE bucket: 445 k records, 109 windows, 37 ms)
    __shared__ uint32_t q0;
    const uint32_t nwin = (m + WCAP - 1) / WCAP;
    if (threadIdx.x == 0) q0 = atomicAdd(qcount, nwin);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nwin; k += FT) queue[q0 + k] = make_uint2(b, k * WCAP);
    return;
  }
  for (uint32_t w0 = 0; w0 < m; w0 += WCAP) {
    for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
      const Rec r = rec[j];
      uint32_t i;
      const uint32_t p = cnt[padded(rec_cell<GD>(r, g, i) - c0)] + rk[j] - s - w0;  // wraps below the window
      if (p < WCAP) stage_put<GD>(s32, sidx, p, r, i);
    }
    __syncthreads();
    fine_flush<GD, GROUPED>(s + w0, min(WCAP, m - w0), s32, sidx, ord, in_classed, group, sorted32, sord, sidx_out, sgroup,
                            flags, pos);
    __syncthreads();
  }
}

// The windows of the large buckets, one per workgroup (queue filled by k_part_fine): final position of a record = its
// cell's start (the cell table is in place) + its rank inside the cell (rk); the records whose position falls into the
// window are staged in LDS and stored as full lines like everywhere else.
template <int GD, bool GROUPED>
__global__ __launch_bounds__(FT) void k_part_fine_windows(const Rec* __restrict__ rec, const uint32_t* __restrict__ rk,
                                                         const uint32_t* __restrict__ bstart, uint32_t bstride,
                                                         const uint32_t* __restrict__ total, uint32_t B, GridP g,
                                                         const uint32_t* __restrict__ ord, const uint8_t* __restrict__ in_classed,
                                                         const int32_t* __restrict__ group, CellTab ct,
                                                         float* __restrict__ sorted32,
                                                         uint32_t* __restrict__ sord, uint32_t* __restrict__ sidx_out,
                                                         int32_t* __restrict__ sgroup, uint8_t* __restrict__ flags,
                                                         uint32_t* __restrict__ pos, const uint2* __restrict__ queue,
                                                         const uint32_t* __restrict__ qcount) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr uint32_t WCAP = wcap(GD);
  constexpr uint32_t FPR = GD == 2 ? 2 : 4;
  float* s32 = reinterpret_cast<float*>(lds);
  uint32_t* sidx = reinterpret_cast<uint32_t*>(s32 + (size_t)WCAP * FPR);
  const uint32_t nq = *qcount;
  for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    const uint2 q = queue[qi];
    const uint32_t b = q.x, w0 = q.y;
    uint32_t s, e;
    bucket_range(bstart, bstride, total, b, B, s, e);
    const uint32_t m = e - s;
    for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
      const Rec r = rec[j];
      uint32_t i;
      const uint32_t p = ct_start(ct, rec_cell<GD>(r, g, i)) + rk[j] - s - w0;  // wraps below the window
      if (p < WCAP) stage_put<GD>(s32, sidx, p, r, i);
    }
    __syncthreads();
    fine_flush<GD, GROUPED>(s + w0, min(WCAP, m - w0), s32, sidx, ord, in_classed, group, sorted32, sord, sidx_out, sgroup,
                            flags, pos);
    __syncthreads();
  }
}

// ---- cell order -> caller order --------------------------------------------------------------------------
// Windows of 2^OWSH list positions: the label words are assembled in LDS and stored as full lines (0.022 ms for
// 10 M points, against 0.15 ms for the same stores made straight from the lanes -- tools/micro/part_bench.hip).
// OWSH (log2 of the window) is chosen per call: 15 from 8 M points (longer runs per window in the scatter pass: 0.086 ms
// against 0.134 ms at 13 for 10 M points), 14 / 13 below, so that smaller clouds still fill the CUs with windows.
constexpr int OT = 1024, OPT = 16;  // scatter pass: OT * OPT positions per workgroup (longer runs per window)
constexpr int OWT = 512;           // write pass: one workgroup per window

template <int OWSH>
__global__ __launch_bounds__(OT) void k_out_scatter(const uint32_t* __restrict__ sord, const uint32_t* __restrict__ labk,
                                                   int64_t n, uint32_t OB, uint32_t* __restrict__ gcur,
                                                   uint2* __restrict__ rec) {
  extern __shared__ uint32_t h[];
  for (uint32_t k = threadIdx.x; k < OB; k += OT) h[k] = 0;
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * (OT * OPT);
  uint32_t o[OPT], w[OPT], r[OPT];
#pragma unroll
  for (int k = 0; k < OPT; k++) {
    const int64_t p = first + (int64_t)k * OT + threadIdx.x;
    o[k] = NONE;
    if (p < n) {
      o[k] = sord[p];
      w[k] = labk[p];
      r[k] = atomicAdd(&h[o[k] >> OWSH], 1u);
    }
  }
  __syncthreads();
  // claim this workgroup's run in every window it touches (one returning atomic per touched window)
  for (uint32_t k = threadIdx.x; k < OB; k += OT) {
    const uint32_t c = h[k];
    if (c) h[k] = atomicAdd(&gcur[k], c);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < OPT; k++) {
    if (o[k] == NONE) continue;
    const uint32_t bk = o[k] >> OWSH;
    rec[((size_t)bk << OWSH) + h[bk] + r[k]] = make_uint2(o[k], w[k]);
  }
}

template <int OWSH>
__global__ __launch_bounds__(OWT) void k_out_write(const uint2* __restrict__ rec, int64_t n, bool have_in_classed,
                                                  int32_t cf_in, int32_t* __restrict__ labels,
                                                  uint8_t* __restrict__ is_core, uint8_t* __restrict__ is_classed,
                                                  unsigned long long* __restrict__ counters, uint32_t* __restrict__ gcur) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sw[];  // [2^OWSH] label words by list position
  if (threadIdx.x == 0) gcur[blockIdx.x] = 0u;  // the window cursors are left at zero for the next call
  const int64_t lo = (int64_t)blockIdx.x << OWSH;
  const uint32_t cnt = (uint32_t)min((int64_t)1 << OWSH, n - lo);  // every list position appears exactly once
  unsigned unclassed = 0;
  for (uint32_t j = threadIdx.x; j < cnt; j += OWT) {
    const uint2 v = rec[lo + j];
    sw[v.x - (uint32_t)lo] = v.y;  // (1 + seed rank) << 2 | core | classed << 1
    if (!(v.y & 2u)) unclassed++;
  }
  __syncthreads();
  // one word per position in LDS (64 KB for 2^14 positions: two workgroups per CU); labels and the two byte arrays are
  // derived from it on the way out, 4 positions per lane where the window is whole (lo is a multiple of 2^OWSH)
  auto lab_of = [&](uint32_t w) { return (w >> 2) ? cf_in + (int32_t)(w >> 2) : 0; };
  auto core_of = [](uint32_t w) { return (uint32_t)((w & 3u) == 1u); };                // core and not classed on entry
  auto cls_of = [](uint32_t w) { return (uint32_t)((w & 2u) != 0u || (w >> 2) != 0u); };  // classed, or labelled now
  const bool vec = (((uintptr_t)labels & 15u) | ((uintptr_t)is_core & 3u) | ((uintptr_t)is_classed & 3u)) == 0;  // caller's pointers
  const uint32_t c4 = vec ? cnt >> 2 : 0u;
  for (uint32_t j = threadIdx.x; j < c4; j += OWT) {
    const uint4 w = reinterpret_cast<const uint4*>(sw)[j];
    const int4 l = make_int4(lab_of(w.x), lab_of(w.y), lab_of(w.z), lab_of(w.w));
    if (have_in_classed) {  // labels are in/out: a point this call does not label keeps its id
      if (l.x) labels[lo + 4 * j] = l.x;
      if (l.y) labels[lo + 4 * j + 1] = l.y;
      if (l.z) labels[lo + 4 * j + 2] = l.z;
      if (l.w) labels[lo + 4 * j + 3] = l.w;
    } else {
      reinterpret_cast<int4*>(labels + lo)[j] = l;
    }
    if (is_core)
      reinterpret_cast<uint32_t*>(is_core + lo)[j] = core_of(w.x) | core_of(w.y) << 8 | core_of(w.z) << 16 | core_of(w.w) << 24;
    if (is_classed)
      reinterpret_cast<uint32_t*>(is_classed + lo)[j] = cls_of(w.x) | cls_of(w.y) << 8 | cls_of(w.z) << 16 | cls_of(w.w) << 24;
  }
  for (uint32_t j = (c4 << 2) + threadIdx.x; j < cnt; j += OWT) {
    const uint32_t w = sw[j];
    const int32_t l = lab_of(w);
    if (!have_in_classed || l) labels[lo + j] = l;
    if (is_core) is_core[lo + j] = (uint8_t)core_of(w);
    if (is_classed) is_classed[lo + j] = (uint8_t)cls_of(w);
  }
  if (have_in_classed) {  // one atomic per workgroup, spread over 32 slots (a single hot word serialises)
    __shared__ unsigned wc[OWT / 64];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) unclassed += __shfl_down(unclassed, d, 64);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = unclassed;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int k = 0; k < OWT / 64; k++) t += wc[k];
      if (t) atomicAdd(&counters[4 + (blockIdx.x & 31)], (unsigned long long)t);
    }
  }
}

// dynamic LDS beyond 64 KB has to be allowed per kernel: the record is keyed on the kernel's ADDRESS (instantiations
// with the same signature are different kernels), one entry per kernel and process
int allow_lds(vcp_ctx* ctx, const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, size_t> allowed;  // (kernel, device): code objects are loaded per device
  std::lock_guard<std::mutex> lk(mu);
  const uint64_t key = (uint64_t)reinterpret_cast<uintptr_t>(kernel) * 64u + (uint64_t)(ctx->device & 63);
  size_t& have = allowed.emplace(key, (size_t)64 * 1024).first->second;
  if (bytes > have) {
    VCP_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
  }
  return VCP_OK;
}
template <class K>
int allow_lds(vcp_ctx* ctx, K kernel, size_t bytes) {
  return allow_lds(ctx, reinterpret_cast<const void*>(kernel), bytes);
}

template <int GD, bool GROUPED>
int build(vcp_ctx* ctx, const GridBuildArgs& a) {
  hipStream_t st = ctx->stream;
  const PartGeom pg = part_geom(a.n, a.g.ncells);
  const size_t nc = (size_t)pg.NS * pg.nchunk;
  VCP_TRY(vcp_ensure(ctx, ctx->b_hist, (nc + 8) * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_rec, (size_t)a.n * sizeof(Rec)));
  uint32_t* counts = ctx->b_hist.as<uint32_t>();
  uint32_t* total = counts + nc;
  Rec* rec = ctx->b_rec.as<Rec>();
  const size_t lds_h = (size_t)pg.NS * 4;
  const uint32_t CPB = 1u << pg.csh;
  const size_t lds_f = ((((size_t)CPB + CPB / 32) + 4) & ~3ull) * 4 + (size_t)wcap(GD) * ((GD == 2 ? 8 : 16) + 4);
  VCP_TRY(allow_lds(ctx, k_part_fine<GD, GROUPED>, lds_f));
  // queue of (bucket, window) pairs for the buckets of many windows: at most n / WCAP + B entries, counter in front
  VCP_TRY(vcp_ensure(ctx, ctx->b_fineq, ((size_t)a.n / wcap(GD) + pg.B + 8) * sizeof(uint2) + 64));
  uint32_t* qcount = ctx->b_fineq.as<uint32_t>();
  uint2* queue = reinterpret_cast<uint2*>(ctx->b_fineq.as<char>() + 64);
  vcp_phase(ctx, "part_hist");
  const uint32_t csh_a = pg.csh + pg.a;  // the coarse passes split by super-bucket
  hipLaunchKernelGGL((k_part_hist<GD, GROUPED>), dim3(pg.nchunk), dim3(PT), lds_h, st, a.d_coords, a.n, a.stride, a.g,
                     a.d_group, a.glo, a.ghi, csh_a, pg.NS, pg.chunk, pg.nchunk, counts, qcount, a.ctcount);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, counts, counts, (int64_t)nc, total));
  vcp_phase(ctx, "part_scatter");
  hipLaunchKernelGGL((k_part_scatter<GD, GROUPED>), dim3(pg.nchunk), dim3(PT), lds_h, st, a.d_coords, a.n, a.stride, a.g,
                     a.d_group, a.glo, a.ghi, csh_a, pg.NS, pg.chunk, pg.nchunk, counts, rec, a.pos);
  const uint32_t* bstart = counts;
  uint32_t bstride = pg.nchunk;
  if (pg.a > 0) {
    vcp_phase(ctx, "part_split");
    VCP_TRY(vcp_ensure(ctx, ctx->b_rec2, (size_t)a.n * sizeof(Rec)));
    VCP_TRY(vcp_ensure(ctx, ctx->b_bstart, ((size_t)pg.B + 2) * 4));
    hipLaunchKernelGGL(k_part_split<GD>, dim3(pg.NS), dim3(FT), 0, st, rec, ctx->b_rec2.as<Rec>(), counts, total, pg.nchunk,
                       pg.NS, pg.a, pg.csh, pg.B, a.g, ctx->b_bstart.as<uint32_t>());
    rec = ctx->b_rec2.as<Rec>();
    bstart = ctx->b_bstart.as<uint32_t>();
    bstride = 1;
  }
  vcp_phase(ctx, "part_fine");
  VCP_TRY(vcp_ensure(ctx, ctx->b_rank, (size_t)a.n * 4));  // ranks inside the cell, written for large buckets only
  VCP_TRY(vcp_ensure(ctx, ctx->b_bstate, (size_t)pg.B + 16));
  uint8_t* bstate = ctx->b_bstate.as<uint8_t>();
  hipLaunchKernelGGL((k_part_fine_small<GD, GROUPED>), dim3(pg.B), dim3(FS), 0, st, rec, bstart, bstride, total, pg.B, pg.csh,
                     a.g, a.d_ord, a.d_in_classed, a.d_group, a.ctwords, a.ctdense, a.ctcount, bstate, a.sorted32, a.sord,
                     a.sidx, a.sgroup, a.flags, a.pos);
  hipLaunchKernelGGL((k_part_fine<GD, GROUPED>), dim3(pg.B), dim3(FT), lds_f, st, rec, ctx->b_rank.as<uint32_t>(), bstart,
                     bstride, total, pg.B, pg.csh, a.g, a.d_ord, a.d_in_classed, a.d_group, a.ctwords, a.ctdense, a.ctcount,
                     a.sorted32, a.sord, a.sidx, a.sgroup, a.flags, a.pos, queue, qcount, bstate);
  {
    const size_t lds_w = (size_t)wcap(GD) * ((GD == 2 ? 8 : 16) + 4);
    VCP_TRY(allow_lds(ctx, k_part_fine_windows<GD, GROUPED>, lds_w));
    const unsigned gw = (unsigned)std::min<size_t>(768, (size_t)a.n / wcap(GD) + 1);  // 3 workgroups per CU (48 KB LDS each in 2-D)
    const CellTab ct{a.ctwords, a.ctdense, a.ctcount + 1};
    hipLaunchKernelGGL((k_part_fine_windows<GD, GROUPED>), dim3(gw), dim3(FT), lds_w, st, rec, ctx->b_rank.as<uint32_t>(),
                       bstart, bstride, total, pg.B, a.g, a.d_ord, a.d_in_classed, a.d_group, ct, a.sorted32,
                       a.sord, a.sidx, a.sgroup, a.flags, a.pos, queue, qcount);
  }
  VCP_HIP(ctx, hipGetLastError());
  return VCP_OK;
}

}  // namespace

int vcp_grid_build_partition(vcp_ctx* ctx, const GridBuildArgs& a) {
  const bool grouped = a.d_group != nullptr;
  if (a.gd == 3) {
    if (grouped) return vcp_fail(ctx, VCP_ERR_ARG, "grouped calls are 2-D");
    return build<3, false>(ctx, a);
  }
  return grouped ? build<2, true>(ctx, a) : build<2, false>(ctx, a);
}

template <int OWSH>
static int output_partition(vcp_ctx* ctx, const GridOutputArgs& a) {
  hipStream_t st = ctx->stream;
  const uint32_t OB = (uint32_t)((a.n + (1 << OWSH) - 1) >> OWSH);
  // window cursors: zero between calls (k_out_write leaves them so), cleared here only when the array is new
  const void* before = ctx->b_outcur.p;
  const size_t before_cap = ctx->b_outcur.cap;
  VCP_TRY(vcp_ensure(ctx, ctx->b_outcur, ((size_t)OB + 8) * 4));
  if (ctx->b_outcur.p != before || ctx->b_outcur.cap != before_cap)
    VCP_HIP(ctx, hipMemsetAsync(ctx->b_outcur.p, 0, ctx->b_outcur.cap, st));
  VCP_TRY(vcp_ensure(ctx, ctx->b_rec, (size_t)OB << (OWSH + 3)));
  uint32_t* gcur = ctx->b_outcur.as<uint32_t>();
  uint2* rec = ctx->b_rec.as<uint2>();
  // every fallible host step comes BEFORE the scatter: between the two kernels the window cursors are not zero
  VCP_TRY(allow_lds(ctx, k_out_write<OWSH>, (size_t)4 << OWSH));
  vcp_phase(ctx, "out_scatter");
  hipLaunchKernelGGL(k_out_scatter<OWSH>, dim3(vcp_blocks(a.n, OT * OPT)), dim3(OT), (size_t)OB * 4, st, a.sord, a.labk, a.n,
                     OB, gcur, rec);
  vcp_phase(ctx, "out_write");
  hipLaunchKernelGGL(k_out_write<OWSH>, dim3(OB), dim3(OWT), (size_t)4 << OWSH, st, rec, a.n, a.have_in_classed, a.cf_in,
                     a.labels, a.is_core, a.is_classed, a.counters, gcur);
  VCP_HIP(ctx, hipGetLastError());
  return VCP_OK;
}

int vcp_grid_output_partition(vcp_ctx* ctx, const GridOutputArgs& a) {
  if (a.n > ((int64_t)1 << 27)) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "more than 2^27 points: use the gather output");
  if (a.n >= ((int64_t)1 << 23)) return output_partition<15>(ctx, a);
  if (a.n >= ((int64_t)1 << 22)) return output_partition<14>(ctx, a);
  return output_partition<13>(ctx, a);
}
