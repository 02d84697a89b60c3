// sort.hip -- the two sorts the path needs, written for 64-wide wavefronts:
//   radix_sort_u64        stable LSD radix sort of (64-bit key, 32-bit value) records, 8 bits per pass -- the Morton keys
//                         of the LBVH (broadphase.hip)
//   sort_segments_u32     every segment of a CSR-like array ascending -- neighbour-list rows (pairs come out sorted by
//                         (i, j) without a global sort) and incidence lists.  Short segments (<= 32, the monodisperse
//                         case: degree ~15) are sorted by one thread each; long ones (a large body among small ones has
//                         thousands of neighbours) by one workgroup each with the same radix passes, so the cost is
//                         O(L) per pass instead of the O(L^2) of an insertion sort.
// One digit pass = histogram over the digit, exclusive scan, stable scatter.  The scatter keeps the input order of equal
// digits with wavefront ballots: the lanes of a wave that hold the same digit find each other with 8 ballots (one per
// digit bit), popc(lanes below) is a lane's rank among them, and the waves of a workgroup are chained through LDS
// counters.  Integer work only, HBM/LDS bound.
#include "mhip_internal.hpp"

namespace mhip {

constexpr int kRadix = 256;          // 8-bit digits
constexpr int kSortRounds = 4;       // a workgroup takes kSortRounds chunks of kBlock keys, in order
constexpr int kSortTile = kBlock * kSortRounds;

// Stable placement of one chunk of up to kBlock keys (one per thread, `valid` lanes only): returns the output position
// of this thread's key given base[d] = position of the first not-yet-placed key with digit d, and advances base[].
// wave_cnt: [kBlock/64][kRadix] ints of LDS.  All threads of the workgroup must call it (barriers inside).
__device__ inline int place_chunk(bool valid, unsigned digit, int* base, int (*wave_cnt)[kRadix]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int k = threadIdx.x; k < nw * kRadix; k += blockDim.x) (&wave_cnt[0][0])[k] = 0;
  __syncthreads();
  unsigned long long mask = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (digit >> b) & 1u;
    const unsigned long long bal = __ballot(valid && bit);
    mask &= bit ? bal : ~bal;
  }
  const int rank = __popcll(mask & ((1ull << lane) - 1ull));
  if (valid && rank == 0) wave_cnt[w][digit] = __popcll(mask);
  __syncthreads();
  int pos = 0;
  if (valid) {
    pos = base[digit] + rank;
    for (int v = 0; v < w; ++v) pos += wave_cnt[v][digit];
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kRadix; d += blockDim.x) {
    int add = 0;
    for (int v = 0; v < nw; ++v) add += wave_cnt[v][d];
    base[d] += add;
  }
  __syncthreads();
  return pos;
}

// ---- global sort of (u64 key, u32 value) ------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_radix_hist(size_t n, const unsigned long long* __restrict__ keys, int shift, unsigned nblocks,
                 int32_t* __restrict__ hist) {
  __shared__ int h[kRadix];
  for (int d = threadIdx.x; d < kRadix; d += blockDim.x) h[d] = 0;
  __syncthreads();
  const size_t t0 = static_cast<size_t>(blockIdx.x) * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const size_t i = t0 + static_cast<size_t>(r) * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255ull], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kRadix; d += blockDim.x) hist[static_cast<size_t>(d) * nblocks + blockIdx.x] = h[d];
}

__global__ void __launch_bounds__(kBlock)
    k_radix_scatter(size_t n, const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals, int shift,
                    unsigned nblocks, const int32_t* __restrict__ offs, unsigned long long* __restrict__ keys_out,
                    unsigned* __restrict__ vals_out) {
  __shared__ int base[kRadix];
  __shared__ int wave_cnt[kBlock / 64][kRadix];
  for (int d = threadIdx.x; d < kRadix; d += blockDim.x) base[d] = offs[static_cast<size_t>(d) * nblocks + blockIdx.x];
  __syncthreads();
  const size_t t0 = static_cast<size_t>(blockIdx.x) * kSortTile;
  for (int r = 0; r < kSortRounds; ++r) {
    const size_t i = t0 + static_cast<size_t>(r) * kBlock + threadIdx.x;
    const bool valid = i < n;
    const unsigned long long key = valid ? keys[i] : 0ull;
    const int pos = place_chunk(valid, static_cast<unsigned>((key >> shift) & 255ull), base, wave_cnt);
    if (valid) {
      keys_out[pos] = key;
      vals_out[pos] = vals[i];
    }
  }
}

size_t radix_sort_workspace_bytes(size_t n) {
  const size_t nblocks = (n + kSortTile - 1) / kSortTile + 1;
  const size_t m = nblocks * kRadix + 2;
  return 2 * m * sizeof(int32_t) + scan_workspace_bytes(m) + 256;
}

// Sorts n records by the key bits [0, 8 * passes); passes must be even so that the result is back in (keys, vals).
int radix_sort_u64(size_t n, unsigned long long* keys, unsigned* vals, unsigned long long* keys_tmp, unsigned* vals_tmp,
                   int passes, void* workspace, hipStream_t stream) {
  MHIP_REQUIRE(passes >= 2 && passes <= 8 && (passes % 2) == 0, MHIP_ERR_LOGIC, "radix sort: passes must be even");
  if (n < 2) return MHIP_SUCCESS;
  const size_t nb = (n + kSortTile - 1) / kSortTile;
  MHIP_REQUIRE(nb * kRadix < (size_t(1) << 31), MHIP_ERR_RUNTIME, "radix sort: too many keys");
  const unsigned nblocks = static_cast<unsigned>(nb);
  const size_t m = nb * kRadix;
  int32_t* hist = static_cast<int32_t*>(workspace);
  int32_t* offs = hist + (m + 2);
  void* scanws = offs + (m + 2);
  for (int p = 0; p < passes; ++p) {
    const unsigned long long* kin = (p & 1) ? keys_tmp : keys;
    const unsigned* vin = (p & 1) ? vals_tmp : vals;
    unsigned long long* kout = (p & 1) ? keys : keys_tmp;
    unsigned* vout = (p & 1) ? vals : vals_tmp;
    k_radix_hist<<<nblocks, kBlock, 0, stream>>>(n, kin, 8 * p, nblocks, hist);
    MHIP_LAUNCH_CHECK();
    if (int e = exclusive_scan_i32(hist, offs, m, scanws, stream)) return e;
    k_radix_scatter<<<nblocks, kBlock, 0, stream>>>(n, kin, vin, 8 * p, nblocks, offs, kout, vout);
    MHIP_LAUNCH_CHECK();
  }
  return MHIP_SUCCESS;
}

// ---- segments ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_seg_sort_short(size_t nseg, const int32_t* __restrict__ seg_ptr, unsigned* __restrict__ data,
                     int32_t* __restrict__ long_count, int32_t* __restrict__ long_list) {
  const size_t s = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (s >= nseg) return;
  const int32_t beg = seg_ptr[s], end = seg_ptr[s + 1];
  if (end - beg > kShortSegment) {
    long_list[atomicAdd(long_count, 1)] = static_cast<int32_t>(s);
    return;
  }
  for (int32_t a = beg + 1; a < end; ++a) {
    const unsigned v = data[a];
    int32_t b = a - 1;
    while (b >= beg && data[b] > v) {
      data[b + 1] = data[b];
      --b;
    }
    data[b + 1] = v;
  }
}

// one workgroup per long segment: LSD radix passes over the key bits [0, 8 * passes), ping-pong with tmp (same offsets)
__global__ void __launch_bounds__(kBlock)
    k_seg_sort_long(const int32_t* __restrict__ seg_ptr, unsigned* __restrict__ data, unsigned* __restrict__ tmp,
                    const int32_t* __restrict__ long_count, const int32_t* __restrict__ long_list, int passes) {
  __shared__ int base[kRadix];
  __shared__ int wave_cnt[kBlock / 64][kRadix];
  __shared__ int scan_scratch[kBlock / 64];
  const int nlong = *long_count;
  for (int q = blockIdx.x; q < nlong; q += gridDim.x) {
    const int32_t s = long_list[q];
    const int32_t beg = seg_ptr[s], len = seg_ptr[s + 1] - beg;
    unsigned* a = data + beg;
    unsigned* b = tmp + beg;
    for (int p = 0; p < passes; ++p) {
      const int shift = 8 * p;
      for (int d = threadIdx.x; d < kRadix; d += blockDim.x) base[d] = 0;
      __syncthreads();
      for (int32_t i = threadIdx.x; i < len; i += blockDim.x) atomicAdd(&base[(a[i] >> shift) & 255u], 1);
      __syncthreads();
      {  // exclusive scan of the 256 counts (kBlock == kRadix: one count per thread)
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int v = base[threadIdx.x];
        int inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_up(inc, off, 64);
          if (lane >= off) inc += t;
        }
        if (lane == 63) scan_scratch[w] = inc;
        __syncthreads();
        int pre = 0;
        for (int k = 0; k < w; ++k) pre += scan_scratch[k];
        base[threadIdx.x] = pre + inc - v;
        __syncthreads();
      }
      for (int32_t c0 = 0; c0 < len; c0 += blockDim.x) {  // chunks in order: stable
        const int32_t i = c0 + threadIdx.x;
        const bool valid = i < len;
        const unsigned key = valid ? a[i] : 0u;
        const int pos = place_chunk(valid, (key >> shift) & 255u, base, wave_cnt);
        if (valid) b[pos] = key;
      }
      __syncthreads();
      unsigned* t = a;
      a = b;
      b = t;
    }
    if (passes & 1) {  // the result sits in tmp: copy it home
      for (int32_t i = threadIdx.x; i < len; i += blockDim.x) b[i] = a[i];
    }
    __syncthreads();
  }
}

// Sorts the segments named in long_list[0 .. *long_count) (device resident: a fixed grid strides over the list, which is
// usually empty -- the launch then costs ~2 us).  key_bits: significant low bits of the entries.  tmp: as long as data.
int sort_listed_segments_u32(const int32_t* seg_ptr, unsigned* data, unsigned* tmp, int key_bits,
                             const int32_t* long_count, const int32_t* long_list, hipStream_t stream) {
  static_assert(kBlock == kRadix, "the in-workgroup digit scan assumes one count per thread");
  MHIP_REQUIRE(key_bits >= 1 && key_bits <= 32, MHIP_ERR_LOGIC, "segment sort: key_bits out of range");
  k_seg_sort_long<<<1024, kBlock, 0, stream>>>(seg_ptr, data, tmp, long_count, long_list, (key_bits + 7) / 8);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

// Sorts every segment of data[seg_ptr[s] .. seg_ptr[s+1]) ascending (unsigned compare): segments of <= kShortSegment
// entries by one thread each, the others through sort_listed_segments_u32.  scratch: (nseg + 16) int32.
int sort_segments_u32(size_t nseg, const int32_t* seg_ptr, unsigned* data, unsigned* tmp, int key_bits,
                      int32_t* scratch, hipStream_t stream) {
  if (nseg == 0) return MHIP_SUCCESS;
  int32_t* long_count = scratch;
  int32_t* long_list = scratch + 16;
  MHIP_HIP(hipMemsetAsync(long_count, 0, sizeof(int32_t), stream));
  k_seg_sort_short<<<grid_exact(nseg), kBlock, 0, stream>>>(nseg, seg_ptr, data, long_count, long_list);
  MHIP_LAUNCH_CHECK();
  return sort_listed_segments_u32(seg_ptr, data, tmp, key_bits, long_count, long_list, stream);
}

}  // namespace mhip
