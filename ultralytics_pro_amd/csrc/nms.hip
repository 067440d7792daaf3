// Batched non_max_suppression, zero host synchronisation, fixed-shape outputs.
// Replaces ultralytics/utils/nms.py:13-166 (candidate filter, best-class / multi-label expansion, class filter,
// top-max_nms by score, class offset, greedy NMS, max_det) and TorchNMS.nms (:239-296).
//
// Three kernels per call:
//  1. nms_candidates   (B x anchor-chunks workgroups)  per anchor: max/argmax over classes (first max wins) or every
//                      class above conf (multi_label); survivors are appended as unique 64-bit keys
//                        key = (~score_bits << 32) | (anchor*nc + cls)
//                      with one slot reservation per workgroup.  Ascending key order == score descending, ties by
//                      ascending candidate order == the stable sort of the reference's candidate list.
//  2. nms_sort         (one 1024-thread workgroup per image)  if n > max_nms an MSB-first radix select finds the exact
//                      max_nms-th key (keys are unique) and compacts; then a bitonic sort (LDS when it fits).
//  3. nms_greedy       (one 1024-thread workgroup per image)  walks the sorted candidates 64 at a time: every wave
//                      tests the chunk against a slice of the kept list held in LDS, the waves build the chunk's suppression
//                      matrix, wave 0 resolves the in-chunk dependencies by iteration, appends survivors, stops at max_det.
// Long multi-label lists (validation: conf 0.001, up to A * nc = 672 k candidates per image, max_nms 30000) run in stages instead:
// the greedy pass almost always has its max_det boxes after the first thousand or so candidates, so (sort, greedy) pairs run over
// growing PREFIXES of the score order and a pair after the first only touches the images the pass before it flagged (it ran out of
// candidates before max_det boxes were kept).  A prefix is "the candidates in the coarse score bins before the one where the running
// count crosses ~4096": nms_hist_kernel builds a 2048-bin score histogram per image (64 bins per octave), nms_emit_kernel finds that
// bin and writes keys for the prefix only (one more sweep over the scores), nms_sort sorts them in LDS.  The full key list (170 MB per
// batch of 32) is written only for flagged images (nms_candidates with `only_redo`), which then take the 16384-prefix and at last the
// exact top-max_nms path.  The result is the reference's in every case: greedy NMS over the score-ordered candidates cut at max_nms,
// cut at max_det.
// IoU arithmetic follows the reference op for op in f32 (class offset added to the boxes first, areas from the offset
// boxes, no eps, survivor iff iou <= thr); FP contraction is disabled so no FMA changes a keep/suppress decision.
#include "common.h"
#pragma clang fp contract(off)

typedef unsigned long long u64;

namespace {

struct NmsWs {
  int* count;   // [B] candidates appended
  int* nsorted; // [B] candidates after select
  u64* keys;    // [B][cap]
  u64* sel;     // [B][selcap]
};

// Coarse score digit of a key for the staged multi-label path: 64 bins per octave of the score counted down from 1.0 (sign, exponent and
// the top 6 mantissa bits of the inverted score word), clamped to [0, COARSE_BINS) - a monotone function of the key, so "bin < b" is a
// prefix of the score order.  conf = 0.001 .. 1 spans 639 bins.  (UPA_COARSE_SHIFT 16 / 18 = 4096 / 1024 bins: same-box A/B 52.6-52.8 k /
// 52.85-52.99 k images/s against 52.87-52.93 k on the validate path - the histogram is zeroed per call.)
#ifndef UPA_COARSE_SHIFT
#define UPA_COARSE_SHIFT 17
#endif
constexpr int COARSE_BINS = 4096 >> (UPA_COARSE_SHIFT - 16);
__device__ __forceinline__ int coarse_bin(unsigned inv_score_bits) {
  const int d = (int)(inv_score_bits >> UPA_COARSE_SHIFT) - (int)(0xC07FFFFFu >> UPA_COARSE_SHIFT);  // 0xC07FFFFF = ~bits(1.0f)
  return d < 0 ? 0 : (d > COARSE_BINS - 1 ? COARSE_BINS - 1 : d);
}

__global__ __launch_bounds__(256) void nms_candidates_kernel(const float* pred, int B, int nc, int A, float conf,
                                                             int multi_label, const uint8_t* cmask, int* count, u64* keys,
                                                             long cap, int* coarse, const int* only_redo) {
  const int b = blockIdx.y;
  if (only_redo && !only_redo[b]) return;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const bool valid = a < A;
  const float* pb = pred + (size_t)b * (4 + nc) * A + (size_t)4 * A + (valid ? a : 0);
  u64* kb = keys + (size_t)b * cap;
  const int lane = threadIdx.x & 63;
  if (!multi_label) {
    float best = -INFINITY;
    int bc = 0;
    if (valid) {
      int c = 0;
      for (; c + 8 <= nc; c += 8) {  // 8 independent loads in flight, then the ordered first-max scan
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = pb[(size_t)(c + q) * A];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (v[q] > best) { best = v[q]; bc = c + q; }
      }
      for (; c < nc; ++c) {
        const float v = pb[(size_t)c * A];
        if (v > best) { best = v; bc = c; }
      }
    }
    // slots: waves reserve inside the workgroup (LDS), the workgroup reserves once in count[b] - one global atomic per 256
    // anchors instead of one per wave with a candidate (same-address atomics serialise in L2: they, not the 86 MB of scores,
    // were a third of this kernel's time)
    __shared__ int blk_n, blk_base;
    if (threadIdx.x == 0) blk_n = 0;
    __syncthreads();
    const bool cand = valid && best > conf && (!cmask || cmask[bc]);
    const u64 m = __ballot(cand);
    int local = 0;
    if (m) {
      if (lane == 0) local = atomicAdd(&blk_n, __popcll(m));
      local = __shfl(local, 0) + __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x == 0 && blk_n) blk_base = atomicAdd(&count[b], blk_n);
    __syncthreads();
    if (cand) {
      const int slot = blk_base + local;
      if (slot >= 0 && slot < cap) kb[slot] = ((u64)(~__float_as_uint(best)) << 32) | (unsigned)(a * nc + bc);
    }
  } else {
    // `coarse` (two-stage sort): the workgroup also counts its candidates per coarse score bin in LDS and adds the non-empty bins to
    // the image's histogram at the end - the sort kernel then knows which bins hold the top candidates before it reads a single key
    __shared__ int lh[COARSE_BINS];
    if (coarse) {
      for (int i = threadIdx.x; i < COARSE_BINS; i += 256) lh[i] = 0;
      __syncthreads();
    }
    // Two sweeps over the workgroup's 256 x nc scores (the second one hits L2).  Sweep 1 counts: every wave adds up its candidates
    // (ballots, scalar), the four waves reserve inside the workgroup (LDS) and the workgroup reserves ONCE in count[b] - one
    // same-address global atomic per 256 anchors.  (One per wave and group of eight classes were 1320 serialised atomics per image at
    // A = 8400: they, not the 86 MB of scores and 170 MB of keys, were the 0.29 ms this kernel took per validation batch.)  Sweep 2
    // writes the keys: the lanes of a wave that hold a candidate of one class take adjacent slots.
    __shared__ int blk_n, blk_base;
    if (threadIdx.x == 0) blk_n = 0;
    __syncthreads();
    int wave_tot = 0;
    for (int c0 = 0; c0 < nc; c0 += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (valid && c0 + q < nc) ? pb[(size_t)(c0 + q) * A] : -INFINITY;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool cand = valid && c0 + q < nc && v[q] > conf && (!cmask || cmask[c0 + q]);
        wave_tot += __popcll(__ballot(cand));
        if (coarse && cand) atomicAdd(&lh[coarse_bin(~__float_as_uint(v[q]))], 1);
      }
    }
    int base = 0;
    if (lane == 0 && wave_tot) base = atomicAdd(&blk_n, wave_tot);
    __syncthreads();
    if (threadIdx.x == 0 && blk_n) blk_base = atomicAdd(&count[b], blk_n);
    __syncthreads();
    base = __shfl(base, 0) + blk_base;
    if (wave_tot)  // uniform
      for (int c0 = 0; c0 < nc; c0 += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (valid && c0 + q < nc) ? pb[(size_t)(c0 + q) * A] : -INFINITY;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool cand = valid && c0 + q < nc && v[q] > conf && (!cmask || cmask[c0 + q]);
          const u64 m = __ballot(cand);
          if (cand) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            if (slot >= 0 && slot < cap) kb[slot] = ((u64)(~__float_as_uint(v[q])) << 32) | (unsigned)(a * nc + c0 + q);
          }
          base += __popcll(m);
        }
      }
    if (coarse) {
      __syncthreads();
      int* gh = coarse + (size_t)b * COARSE_BINS;
      for (int i = threadIdx.x; i < COARSE_BINS; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
    }
  }
}

// Where the running count of an image's coarse histogram crosses `target`: bstar = the crossing bin, m = candidates in the bins before it
// (bstar = -1: fewer than target candidates), total = all candidates.  Whole workgroup of NT threads.
template <int NT>
__device__ void coarse_split(const int* gh, int target, int& bstar, int& m, int& total) {
  __shared__ int wsum[NT / 64];
  __shared__ int s_b, s_m;
  constexpr int PER = COARSE_BINS / NT;
  int c[PER], t = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) { c[q] = gh[threadIdx.x * PER + q]; t += c[q]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  if (threadIdx.x == 0) { s_b = -1; s_m = 0; }
  __syncthreads();
  int before = incl - t;
  total = 0;
  for (int w2 = 0; w2 < NT / 64; ++w2) {
    if (w2 < wave) before += wsum[w2];
    total += wsum[w2];
  }
  if (before < target && before + t >= target) {  // exactly one thread: the crossing bin is one of its PER
    int cum = before, q = 0;
    for (; q < PER; ++q) {
      if (cum + c[q] >= target) break;
      cum += c[q];
    }
    s_b = threadIdx.x * PER + q;
    s_m = cum;
  }
  __syncthreads();
  bstar = s_b;
  m = s_m;
  __syncthreads();  // (the shared words may be rewritten by a second call)
}

__device__ __forceinline__ bool coarse_prefix_usable(int total, int target, int bstar, int m) {
  return total > target && bstar > 0 && m >= target / 4 && m <= target;
}

// Multi-label candidates of long lists, first half: ONLY the coarse score histogram of every image (no keys yet).
// Also: the best (lowest) coarse bin among each wave's candidates per group of eight classes, group_best[b][wave][group] - the emit
// kernel reads again only the class rows of the (wave, group) pairs that hold a candidate of the prefix (the top few thousand candidates
// of an image sit in a handful of classes).
__global__ __launch_bounds__(256) void nms_hist_kernel(const float* pred, int nc, int A, float conf, const uint8_t* cmask,
                                                       int* coarse, int* group_best) {
  const int b = blockIdx.y;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const bool valid = a < A;
  const float* pb = pred + (size_t)b * (4 + nc) * A + (size_t)4 * A + (valid ? a : 0);
  __shared__ int lh[COARSE_BINS];
  for (int i = threadIdx.x; i < COARSE_BINS; i += 256) lh[i] = 0;
  __syncthreads();
  const int ng = (nc + 7) >> 3;
  int* gb = group_best + (((size_t)b * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * ng;
  for (int c0 = 0; c0 < nc; c0 += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (valid && c0 + q < nc) ? pb[(size_t)(c0 + q) * A] : -INFINITY;
    int best = COARSE_BINS;  // (no candidate)
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (valid && c0 + q < nc && v[q] > conf && (!cmask || cmask[c0 + q])) {
        const int bin = coarse_bin(~__float_as_uint(v[q]));
        atomicAdd(&lh[bin], 1);
        best = bin < best ? bin : best;
      }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const int o = __shfl_xor(best, d);
      best = o < best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) gb[c0 >> 3] = best;
  }
  __syncthreads();
  int* gh = coarse + (size_t)b * COARSE_BINS;
  for (int i = threadIdx.x; i < COARSE_BINS; i += 256)
    if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// Second half: every workgroup finds the image's prefix bin from the finished histogram (the same few hundred additions in each of the
// image's workgroups - cheaper than another launch), then sweeps its 256 x nc scores again and writes keys ONLY for the candidates of
// the prefix (coarse bin < bstar: fewer than `target` per image) - straight into the image's sorted-list buffer `sel`, where the sort
// kernel picks them up.  The other candidates (up to A * nc = 672 k per image, 170 MB of keys per validation batch) are never
// written unless the greedy pass flags the image (nms_candidates_kernel with `only_redo` then writes them all).  Images without a
// usable prefix (few candidates, or one bin holds them all) get all their keys into `keys` here: mode[b] = 0.
__global__ __launch_bounds__(256) void nms_emit_kernel(const float* pred, int nc, int A, float conf, const uint8_t* cmask,
                                                       const int* coarse, const int* group_best, int target, int* count, int* mode,
                                                       int* pcount, u64* keys, long cap, u64* sel, int selcap) {
  const int b = blockIdx.y;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const bool valid = a < A;
  const float* pb = pred + (size_t)b * (4 + nc) * A + (size_t)4 * A + (valid ? a : 0);
  const int lane = threadIdx.x & 63;
  int bstar, m, total;
  coarse_split<256>(coarse + (size_t)b * COARSE_BINS, target, bstar, m, total);
  const bool usable = coarse_prefix_usable(total, target, bstar, m) && m <= selcap;
  if (blockIdx.x == 0 && threadIdx.x == 0) { mode[b] = usable ? 1 : 0; count[b] = total; }
  const int lim = usable ? bstar : COARSE_BINS;
  u64* dst = usable ? sel + (size_t)b * selcap : keys + (size_t)b * cap;
  const long dcap = usable ? (long)selcap : cap;
  // One sweep: the workgroup's prefix candidates (about target / workgroups-per-image of them) are collected in LDS, then the workgroup
  // reserves their slots with ONE atomic in pcount[b] and copies them out.  Candidates beyond the LDS buffer (only when one workgroup holds
  // a large part of the list: a few candidates in all, or the one-bin fallback) take slots straight from pcount[b], one atomic per ballot.
  constexpr int LCAP = 1024;
  __shared__ u64 staged[LCAP];
  __shared__ int s_cnt, blk_base;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int ng = (nc + 7) >> 3;
  const int* gb = group_best + (((size_t)b * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * ng;
  const int my_gb = (ng <= 64 && lane < ng) ? gb[lane] : 0;  // lane g holds the wave's best bin of class group g
    for (int c0 = 0; c0 < nc; c0 += 8) {
      if (ng <= 64 && __shfl(my_gb, c0 >> 3) >= lim) continue;  // uniform: no candidate of the prefix in these eight class rows
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (valid && c0 + q < nc) ? pb[(size_t)(c0 + q) * A] : -INFINITY;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool cand = valid && c0 + q < nc && v[q] > conf && (!cmask || cmask[c0 + q]) && coarse_bin(~__float_as_uint(v[q])) < lim;
        const u64 mk = __ballot(cand);
        if (!mk) continue;  // uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_cnt, __popcll(mk));
        base = __shfl(base, 0);
        const int slot = base + __popcll(mk & ((1ull << lane) - 1ull));
        const u64 key = ((u64)(~__float_as_uint(v[q])) << 32) | (unsigned)(a * nc + c0 + q);
        if (cand && slot < LCAP) staged[slot] = key;
        const u64 over = __ballot(cand && slot >= LCAP);
        if (over) {  // uniform
          int gbase = 0;
          if (lane == 0) gbase = atomicAdd(&pcount[b], __popcll(over));
          gbase = __shfl(gbase, 0);
          const int gs = gbase + __popcll(over & ((1ull << lane) - 1ull));
          if (cand && slot >= LCAP && gs >= 0 && gs < dcap) dst[gs] = key;
        }
      }
    }
  __syncthreads();
  const int nl = s_cnt < LCAP ? s_cnt : LCAP;
  if (threadIdx.x == 0 && nl) blk_base = atomicAdd(&pcount[b], nl);
  __syncthreads();
  if (nl)
    for (int i = threadIdx.x; i < nl; i += 256) {
      const int gs = blk_base + i;
      if (gs >= 0 && gs < dcap) dst[gs] = staged[i];
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's global STORES (s_waitcnt vmcnt(0)): in the greedy
// kernel's chunk loop that made wave 0 sit out the acknowledgement of the kept boxes' output rows - ~1.5-2 us per chunk, what the phase
// profile (tools/experiments/r05_greedy_phases.py) showed as "phase 2" - although nobody in the workgroup reads them back.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

#ifndef UPA_GREEDY_ROWS_MIN
#define UPA_GREEDY_ROWS_MIN 8
#endif
// bitonic sort of `buf[0..npad)` (npad power of two) ascending, all threads of the workgroup
template <int NT>
__device__ void bitonic_sort(u64* buf, int npad) {
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (npad >> 1); t += NT) {
        const int i = 2 * t - (t & (j - 1));  // index with bit j clear
        const int l = i + j;
        const bool up = (i & k) == 0;
        const u64 x = buf[i], y = buf[l];
        if ((x > y) == up) { buf[i] = y; buf[l] = x; }
      }
      __syncthreads();
    }
  }
}

constexpr int SORT_NT = 1024;
constexpr int LDS_SORT_CAP = 16384;  // u64 -> 128 KiB of dynamic LDS (covers every single-label case, A <= 16384)

// best_keys != nullptr (NMS prefilter): the candidates come from the dense (B, A) best-class keys the Detect class tails wrote
// (single-label rule: best class of an anchor, nms.py:109) - this workgroup, which owns the image, first filters them by
// conf_thres / class mask and compacts them into `keys` (slot counter in LDS, wave-aggregated: no global atomics - with a few
// hundred candidates per image the atomics on count[b] were what the scan kernels spent their time on), then sorts as usual.
// Two-stage form for long candidate lists (multi-label validation: up to A * nc candidates per image, max_nms = 30000): with `prefix`
// > 0 the kernel selects and sorts only the top `prefix` (<= LDS_SORT_CAP, so the sort runs in LDS) and sets partial[b]; the greedy
// pass stops at max_det kept boxes long before that list ends in all but pathological images, and flags redo[b] when it does not -
// a second (sort, greedy) pair with prefix = 0 then runs the full top-max_nms path for the flagged images only (`only_redo`: every
// other workgroup returns at once).  The result is the reference's in every case: greedy NMS over the score-ordered candidates cut
// at max_nms, cut at max_det.
__global__ __launch_bounds__(SORT_NT) void nms_sort_kernel(const int* count, int* nsorted, u64* keys, u64* sel, long cap,
                                                           int selcap, int max_nms, const u64* best_keys, int nc, int A,
                                                           float conf, const uint8_t* cmask, int prefix, int* partial,
                                                           const int* only_redo, const int* coarse, const int* mode,
                                                           const int* pcount) {
  if (only_redo && !only_redo[blockIdx.x]) return;
  extern __shared__ __attribute__((aligned(16))) u64 lbuf[];  // LDS_SORT_CAP keys
  __shared__ int hist[256];
  __shared__ u64 s_prefix;
  __shared__ int s_remaining, s_n;
  const int b = blockIdx.x;
  u64* kb = keys + (size_t)b * cap;
  u64* sb = sel + (size_t)b * selcap;
  int n;
  if (best_keys) {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    // every candidate list this call can produce fits the LDS sort buffer and stays under max_nms (the usual single-label call: A = 8400):
    // the keys are compacted straight into LDS and sorted there - no trip through `keys` in global memory, no wait for those stores
    const bool in_lds = prefix == 0 && A <= LDS_SORT_CAP && A <= max_nms && (long)A <= cap;
    constexpr int U = 8;  // keys in flight per thread (one load per loop trip left the scan at a memory round trip per 1024 anchors)
    for (int a0 = 0; a0 < A; a0 += U * SORT_NT) {
      u64 kk[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int a = a0 + u * SORT_NT + (int)threadIdx.x;
        kk[u] = a < A ? best_keys[(size_t)b * A + a] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int a = a0 + u * SORT_NT + (int)threadIdx.x;
        const u64 k = kk[u];
        bool cand = false;
        if (a < A) {
          const float best = __uint_as_float(~(unsigned)(k >> 32));
          const int bc = (int)((unsigned)k - (unsigned)a * (unsigned)nc);
          cand = best > conf && (!cmask || cmask[bc]);
        }
        const u64 m = __ballot(cand);
        if (m) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_n, __popcll(m));
          base = __shfl(base, 0);
          if (cand) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            if (in_lds) lbuf[slot] = k;  // (slot < A <= LDS_SORT_CAP)
            else if (slot >= 0 && slot < cap) kb[slot] = k;
          }
        }
      }
    }
    if (in_lds) {
      lds_barrier();
      n = s_n;
      int npad = 2;
      while (npad < n) npad <<= 1;
      for (int i = n + threadIdx.x; i < npad; i += SORT_NT) lbuf[i] = ~0ull;
      lds_barrier();
      bitonic_sort<SORT_NT>(lbuf, npad);
      for (int i = threadIdx.x; i < n; i += SORT_NT) sb[i] = lbuf[i];
      if (threadIdx.x == 0) nsorted[b] = n;
      if (partial && threadIdx.x == 0) partial[b] = 0;
      return;
    }
    __syncthreads();  // the compacted keys (global) and the counter are visible to the whole workgroup
    n = s_n;
    __syncthreads();  // before s_n is reused below
  } else {
    n = count[b];
  }
  if (n > cap) n = (int)cap;
  const int target = (prefix > 0 && prefix < max_nms) ? prefix : max_nms;  // how many of the best candidates this call keeps
  if (partial && threadIdx.x == 0) partial[b] = (prefix > 0 && n > target && target < max_nms) ? 1 : 0;
  // The prefix keys of this image were written by nms_emit_kernel (mode[b] = 1): pcount[b] of them (fewer than target), in `sb`
  if (mode && mode[b] && prefix > 0) {
    n = pcount[b];
    if (n > LDS_SORT_CAP) n = LDS_SORT_CAP;  // (cannot happen: m <= target <= LDS_SORT_CAP)
    int npad = 2;
    while (npad < n) npad <<= 1;
    for (int i = threadIdx.x; i < npad; i += SORT_NT) lbuf[i] = i < n ? sb[i] : ~0ull;
    __syncthreads();
    bitonic_sort<SORT_NT>(lbuf, npad);
    for (int i = threadIdx.x; i < n; i += SORT_NT) sb[i] = lbuf[i];
    if (threadIdx.x == 0) nsorted[b] = n;
    return;
  }
  // A prefix from the image's coarse score histogram: the sorted prefix need not be exactly `target` long - any prefix of the score
  // order does, the greedy pass flags the image when it runs out.  So the prefix is "every key in the coarse bins before the one where
  // the running count crosses target": found from the histogram alone, then ONE pass over the n keys compacts those (fewer than target
  // <= LDS_SORT_CAP) straight into LDS, where they are sorted.  No radix select (three to five passes over up to A * nc keys through
  // this one workgroup, then the compaction pass: 0.59 ms per validation batch).
  if (coarse && prefix > 0 && n > target && target < max_nms && target <= LDS_SORT_CAP) {
    int bstar, m, total;
    coarse_split<SORT_NT>(coarse + (size_t)b * COARSE_BINS, target, bstar, m, total);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (coarse_prefix_usable(n, target, bstar, m)) {  // (uniform) else: the exact path below
      constexpr int U = 8;
      for (int i0 = threadIdx.x; i0 < n; i0 += U * SORT_NT) {
        u64 k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * SORT_NT;
          k[u] = i < n ? kb[i] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * SORT_NT;
          const bool take = i < n && coarse_bin((unsigned)(k[u] >> 32)) < bstar;
          const u64 mk = __ballot(take);
          if (mk) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_n, __popcll(mk));
            base = __shfl(base, 0);
            if (take) {
              const int slot = base + __popcll(mk & ((1ull << lane) - 1ull));
              if (slot < LDS_SORT_CAP) lbuf[slot] = k[u];
            }
          }
        }
      }
      __syncthreads();
      n = s_n < LDS_SORT_CAP ? s_n : LDS_SORT_CAP;  // (== m)
      int npad = 2;
      while (npad < n) npad <<= 1;
      for (int i = n + threadIdx.x; i < npad; i += SORT_NT) lbuf[i] = ~0ull;
      __syncthreads();
      bitonic_sort<SORT_NT>(lbuf, npad);
      for (int i = threadIdx.x; i < n; i += SORT_NT) sb[i] = lbuf[i];
      if (threadIdx.x == 0) nsorted[b] = n;
      return;
    }
    __syncthreads();
  }
  if (n > target) {
    // exact threshold key K*: exactly `target` keys are <= K* (keys are unique)
    if (threadIdx.x == 0) { s_prefix = 0ull; s_remaining = target; s_n = 0; }
    // Every pass reads all n keys (up to A * nc = 672 k per image in multi-label validation) through ONE workgroup: the loads are issued
    // eight deep per thread (one load per iteration behind an LDS atomic left the loop at a memory round trip per key: 7 of the 8.4 ms
    // this kernel took per validation batch), and the digit loop stops as soon as the chosen bin is taken whole (the low digits only
    // order keys of equal score: usually three or four passes instead of eight).
    constexpr int U = 8;
    u64 done_mask = 0ull;  // low bits of K* once the loop stops early (uniform)
    for (int shift = 56; shift >= 0; shift -= 8) {
      for (int i = threadIdx.x; i < 256; i += SORT_NT) hist[i] = 0;
      __syncthreads();
      const u64 prefix = s_prefix;
      const u64 himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
      for (int i0 = threadIdx.x; i0 < n; i0 += U * SORT_NT) {
        u64 k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * SORT_NT;
          k[u] = i < n ? kb[i] : ~0ull;  // (the all-ones key can match no prefix of a real key's range: real keys have a clear low part < a * nc)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * SORT_NT;
          if (i < n && (k[u] & himask) == (prefix & himask)) atomicAdd(&hist[(int)((k[u] >> shift) & 255ull)], 1);
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int rem = s_remaining, cum = 0, d = 0;
        for (; d < 256; ++d) {
          if (cum + hist[d] >= rem) break;
          cum += hist[d];
        }
        s_remaining = rem - cum;
        s_prefix = prefix | ((u64)d << shift);
        s_n = (cum + hist[d] == rem) ? 1 : 0;  // the whole bin is needed: every key with this prefix is below K*
      }
      __syncthreads();
      if (s_n) {  // uniform
        done_mask = shift == 0 ? 0ull : ((1ull << shift) - 1ull);
        break;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const u64 kstar = s_prefix | done_mask;
    const int lane_ = threadIdx.x & 63;
    for (int i0 = threadIdx.x; i0 < n; i0 += U * SORT_NT) {
      u64 k[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * SORT_NT;
        k[u] = i < n ? kb[i] : ~0ull;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * SORT_NT;
        const bool take = i < n && k[u] <= kstar;
        const u64 m = __ballot(take);  // one LDS atomic per wave and iteration, not one per key
        if (m) {
          int base = 0;
          if (lane_ == 0) base = atomicAdd(&s_n, __popcll(m));
          base = __shfl(base, 0);
          if (take) {
            const int slot = base + __popcll(m & ((1ull << lane_) - 1ull));
            if (slot < selcap) sb[slot] = k[u];
          }
        }
      }
    }
    __syncthreads();
    n = s_n < target ? s_n : target;
    kb = sb;  // source is now the compacted list
    __syncthreads();
  }
  int npad = 1;
  while (npad < n) npad <<= 1;
  if (npad < 2) npad = 2;
  if (npad <= LDS_SORT_CAP) {
    for (int i = threadIdx.x; i < npad; i += SORT_NT) lbuf[i] = i < n ? kb[i] : ~0ull;
    __syncthreads();
    bitonic_sort<SORT_NT>(lbuf, npad);
    for (int i = threadIdx.x; i < n; i += SORT_NT) sb[i] = lbuf[i];
  } else {
    if (kb != sb)
      for (int i = threadIdx.x; i < n; i += SORT_NT) sb[i] = kb[i];
    for (int i = n + threadIdx.x; i < npad; i += SORT_NT) sb[i] = ~0ull;
    __syncthreads();
    bitonic_sort<SORT_NT>(sb, npad);  // global memory; visibility inside one workgroup via __syncthreads
  }
  if (threadIdx.x == 0) nsorted[b] = n;
}

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1,
                                       float bx2, float by2, float barea, float thr) {
  const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
  const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float inter = w * h;
  const float iou = inter / (aarea + barea - inter);
  return !(iou <= thr);  // reference keeps `iou <= thr`; NaN is not kept either
}
// (A wave-uniform shortcut - skip the IEEE division when no lane's box overlaps the broadcast one at all, inter == 0 - was measured
// SLOWER: the vote and the branch cost more than the division they save in one wave out of a few: phase 1 1744 -> 2501 cycles per chunk
// on the headline batches, 4842 -> 9017 in validation, serial step +1.4 %.  Replacing the division by inter * rcp(den) with an exact
// repeat for quotients within a few ulp of thr was slower too: 1788 -> 1897 (phase 1), 1689 -> 2066 (columns) cycles per chunk - the
// margin test costs what the division's extra instructions do.  profiles/r05_greedy_phases.txt)

// -DUPA_GREEDY_PROF (tools/experiments/r05_greedy_phases.sh builds it into a separate library): wave 0 adds the shader cycles it spends per
// phase of the greedy kernel to g_greedy_prof (read and cleared by upa_debug_greedy_prof); the product build carries none of it.
#ifdef UPA_GREEDY_PROF
__device__ unsigned long long g_greedy_prof[12];  // init, stage load, phase 1, barrier 1, phase 2 rest, barrier 2, chunks, workgroups, alive mask, rows, resolve loop, output rows
#define GP_DECL unsigned long long gp_t = __builtin_amdgcn_s_memtime(), gp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define GP_AT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); gp_acc[i] += t_ - gp_t; gp_t = t_; } while (0)
#define GP_COUNT(i) (gp_acc[i] += 1)
#define GP_FLUSH do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 12; ++i_) if (i_ != 7) atomicAdd(&g_greedy_prof[i_], gp_acc[i_]); atomicAdd(&g_greedy_prof[7], 1ull); } } while (0)
#else
#define GP_DECL
#define GP_AT(i)
#define GP_COUNT(i)
#define GP_FLUSH
#endif
constexpr int GREEDY_IL = 2;  // suppression columns: candidates evaluated per loop trip by a wave (1 / 4: 1828 / 1679 against 1594 cycles per chunk)
constexpr int GREEDY_ROWS_MIN = UPA_GREEDY_ROWS_MIN;  // alive candidates in a chunk from which the suppression columns are computed by all waves
#ifndef UPA_GREEDY_NT
#define UPA_GREEDY_NT 1024  // 16 waves per image: with the parallel phase 2, 3 us faster per serial step than 8 (0.8013-0.8029 vs 0.8043-0.8062 ms, same box)
#endif
constexpr int GREEDY_NT = UPA_GREEDY_NT;


constexpr int GREEDY_NW = GREEDY_NT / 64;
constexpr int MAX_DET_CAP = 1024;

__global__ __launch_bounds__(GREEDY_NT) void nms_greedy_kernel(const float* pred, int nc, int A, const int* nsorted,
                                                               const u64* sel, int selcap, float iou_thr, int agnostic,
                                                               float max_wh, int max_det, float* out, int* counts,
                                                               int* keep_idx, const int* partial, int* redo, const int* only_redo) {
  if (only_redo && !only_redo[blockIdx.x]) return;
  __shared__ float kx1[MAX_DET_CAP], ky1[MAX_DET_CAP], kx2[MAX_DET_CAP], ky2[MAX_DET_CAP], kar[MAX_DET_CAP];
  __shared__ u64 alive_w[GREEDY_NW];
  __shared__ u64 colm[64];  // suppression-matrix columns of the chunk's candidates (phase 2), zero between chunks
  __shared__ int s_kept;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GP_DECL;
  const int n = nsorted[b];
  const u64* sb = sel + (size_t)b * selcap;
  const float* pb = pred + (size_t)b * (4 + nc) * A;
  float* ob = out + (size_t)b * max_det * 6;
  if (threadIdx.x == 0) s_kept = 0;
  if (threadIdx.x < 64) colm[threadIdx.x] = 0ull;
  __syncthreads();
  GP_AT(0);
  // Candidates are staged 512 at a time: every thread decodes ONE key and gathers its box (two dependent global round trips, paid
  // once per 512 candidates instead of once per 64-candidate chunk - the gathers were most of this kernel's time: a typical image
  // has 100-500 candidates and the chunks are resolved one after the other), then the eight chunks of the stage run out of LDS.
  __shared__ float c_rx1[GREEDY_NT], c_ry1[GREEDY_NT], c_rx2[GREEDY_NT], c_ry2[GREEDY_NT], c_score[GREEDY_NT], c_cls[GREEDY_NT];
  __shared__ int c_anchor[GREEDY_NT];
  for (int stage = 0; stage < n; stage += GREEDY_NT) {
    if (s_kept >= max_det) break;
    {
      const int ci = stage + (int)threadIdx.x;
      if (ci < n) {
        const u64 key = sb[ci];
        const unsigned ok = (unsigned)(key & 0xFFFFFFFFull);
        const int anchor = (int)(ok / (unsigned)nc);
        const int c = (int)(ok - (unsigned)anchor * (unsigned)nc);
        const float cx = pb[anchor], cy = pb[(size_t)A + anchor], w = pb[(size_t)2 * A + anchor],
                    h = pb[(size_t)3 * A + anchor];
        const float hw = w / 2.f, hh = h / 2.f;  // xywh2xyxy, utils/ops.py:268-284
        c_rx1[threadIdx.x] = cx - hw; c_ry1[threadIdx.x] = cy - hh; c_rx2[threadIdx.x] = cx + hw; c_ry2[threadIdx.x] = cy + hh;
        c_score[threadIdx.x] = __uint_as_float(~(unsigned)(key >> 32));
        c_cls[threadIdx.x] = (float)c;
        c_anchor[threadIdx.x] = anchor;
      }
    }
    __syncthreads();
    GP_AT(1);
    const int stage_n = n - stage < GREEDY_NT ? n - stage : GREEDY_NT;
    for (int base = 0; base < stage_n; base += 64) {
      const int kept = s_kept;
      if (kept >= max_det) break;
      const int li = base + lane;
      const bool valid = li < stage_n;
      float rx1 = 0, ry1 = 0, rx2 = 0, ry2 = 0, score = 0, clsf = 0;  // raw (un-offset) box
      float x1 = 0, y1 = 0, x2 = 0, y2 = 0, area = 0;
      int anchor = 0;
      if (valid) {
        rx1 = c_rx1[li]; ry1 = c_ry1[li]; rx2 = c_rx2[li]; ry2 = c_ry2[li]; score = c_score[li]; clsf = c_cls[li];
        anchor = c_anchor[li];
        const float off = clsf * (agnostic ? 0.f : max_wh);  // nms.py:143
        x1 = rx1 + off; y1 = ry1 + off; x2 = rx2 + off; y2 = ry2 + off;
        area = (x2 - x1) * (y2 - y1);
      }
      // phase 1: this wave tests the chunk against its slice of the kept list
      bool sup = false;
#pragma unroll 2
      for (int k = wave; k < kept; k += GREEDY_NW)
        sup |= iou_gt(kx1[k], ky1[k], kx2[k], ky2[k], kar[k], x1, y1, x2, y2, area, iou_thr);
      const u64 am = __ballot(valid && !sup);
      if (lane == 0) alive_w[wave] = am;
      GP_AT(2);
      lds_barrier();
      GP_AT(3);
      GP_COUNT(6);
      // phase 2: resolve the chunk greedily: in candidate order, an alive candidate is kept unless an earlier KEPT candidate of the chunk
      // suppresses it.  Walking the set bits one by one in a single wave - broadcast the kept box, evaluate its IoU against the 64
      // lanes, drop the suppressed bits - was a serial chain of ~200 cycles per kept box: 60-65 % of this kernel's cycles on the
      // headline batches, 48 % in validation (tools/experiments/r05_greedy_phases.py); even with precomputed suppression rows the walk
      // cost ~100 cycles per kept box (scalar <-> vector register round trips).  Now:
      //  a. all eight waves (each holds the chunk's 64 boxes in its lanes) evaluate the pairs: the alive candidates j are dealt
      //     round-robin to the waves, two per loop trip (independent chains); for its j a wave sets bit j in the lanes i > j that j
      //     would suppress - iou_gt(j, i), the same call the chain made - so lane i collects its COLUMN of the suppression matrix;
      //     the waves' partial columns meet in LDS (ds_or_b64).
      //  b. wave 0 solves  kept = alive & ~(some kept j in my column)  by iteration from kept = alive: one ballot per round, and the
      //     bits settle from the low indices upward, so the rounds needed = the longest suppress chain in the chunk (2-4), not
      //     the number of kept boxes.  The fixed point is unique (bit i depends on bits < i only) and equals the serial walk.
      // Chunks with few alive candidates keep the serial walk (no second barrier).
      u64 alive = alive_w[0];
#pragma unroll
      for (int q = 1; q < GREEDY_NW; ++q) alive &= alive_w[q];
      const u64 alive_s = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(alive >> 32)) << 32) |
                          (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)alive);  // (the builtin returns int: no sign extension)
      const bool by_cols = __popcll(alive_s) >= GREEDY_ROWS_MIN;  // uniform over the workgroup
      const bool me_alive = (alive_s >> lane) & 1ull;
      GP_AT(8);
      if (by_cols) {
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        auto hits = [&](int j) __attribute__((always_inline)) {  // would candidate j, if kept, suppress this lane's candidate?
          const float jx1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x1), j));
          const float jy1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y1), j));
          const float jx2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x2), j));
          const float jy2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y2), j));
          const float jar = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(area), j));
          return me_alive && lane > j && iou_gt(jx1, jy1, jx2, jy2, jar, x1, y1, x2, y2, area, iou_thr);
        };
        u64 col = 0ull;
        // this wave's candidates: the alive lanes whose rank among the alive ones is wave_s modulo the wave count (one ballot; the walk
        // below then visits only them - walking all 64 alive bits in every wave cost more than the IoU tests: 3603 -> 1594 cycles per
        // chunk), GREEDY_IL at a time (independent chains)
        u64 it = __ballot(me_alive && (__popcll(alive_s & ((1ull << lane) - 1ull)) & (GREEDY_NW - 1)) == wave_s);
        while (it) {
          int j[GREEDY_IL];
          bool h[GREEDY_IL];
#pragma unroll
          for (int q = 0; q < GREEDY_IL; ++q) {
            j[q] = it ? __builtin_ctzll((unsigned long long)it) : -1;
            it &= it - 1ull;  // (0 stays 0)
          }
#pragma unroll
          for (int q = 0; q < GREEDY_IL; ++q) h[q] = j[q] >= 0 ? hits(j[q]) : false;
#pragma unroll
          for (int q = 0; q < GREEDY_IL; ++q) col |= h[q] ? 1ull << (j[q] & 63) : 0ull;
        }
        if (col) atomicOr((unsigned long long*)&colm[lane], (unsigned long long)col);
        GP_AT(9);
        lds_barrier();
        GP_AT(5);
      }
      if (wave == 0) {
        u64 keepmask = 0ull;
        const int kept_s = __builtin_amdgcn_readfirstlane(kept);
        const int room = max_det - kept_s;  // >= 1 (the chunk loop stops at max_det)
        if (by_cols) {
          const u64 col = colm[lane];
          colm[lane] = 0ull;  // for the next chunk (its columns are written after that chunk's first barrier)
          u64 k = alive_s;
          for (;;) {
            const u64 kn = __ballot(me_alive && (col & k) == 0ull);
            if (kn == k) break;
            k = kn;
          }
          // the first `room` kept candidates (the serial walk stopped there)
          keepmask = __ballot(((k >> lane) & 1ull) && __popcll(k & ((1ull << lane) - 1ull)) < room);
        } else {
          // (the serial walk: the mask and the kept count are wave-uniform, held in scalar registers, so the lane index of the kept box
          // is scalar too and its five values are broadcast by v_readlane_b32 instead of five ds_bpermute round trips)
          u64 rem = alive_s;
          int nkeep = 0;
          while (rem) {
            const int i = __builtin_ctzll((unsigned long long)rem);
            keepmask |= 1ull << i;
            rem &= rem - 1ull;
            ++nkeep;
            if (nkeep >= room) break;
            const float ix1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x1), i));
            const float iy1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y1), i));
            const float ix2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x2), i));
            const float iy2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y2), i));
            const float iar = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(area), i));
            const bool sup2 = ((rem >> lane) & 1ull) && iou_gt(ix1, iy1, ix2, iy2, iar, x1, y1, x2, y2, area, iou_thr);
            rem &= ~__ballot(sup2);
          }
        }
        GP_AT(10);
        if ((keepmask >> lane) & 1ull) {
          const int idx = kept + __popcll(keepmask & ((1ull << lane) - 1ull));
          if (idx < max_det) {
            kx1[idx] = x1; ky1[idx] = y1; kx2[idx] = x2; ky2[idx] = y2; kar[idx] = area;
            float* o = ob + (size_t)idx * 6;
            o[0] = rx1; o[1] = ry1; o[2] = rx2; o[3] = ry2; o[4] = score; o[5] = clsf;
            if (keep_idx) keep_idx[(size_t)b * max_det + idx] = anchor;
          }
        }
        if (lane == 0) {
          const int nk = kept + __popcll(keepmask);
          s_kept = nk < max_det ? nk : max_det;
        }
        GP_AT(11);
      }
      GP_AT(4);
      lds_barrier();
      GP_AT(5);
    }
    lds_barrier();  // the stage buffers are rewritten by the next stage
  }
  GP_FLUSH;
  {
    // the rest of the fixed-shape output: rows past the kept ones are zero (keep_idx: -1).  Written here, not up front: zeroing first
    // meant a full barrier (with its wait for the stores) before the first kept row could be written over the zeros
    const int k = s_kept;  // (final: every exit of the loops above is behind a barrier that follows its last update)
    for (int i = k * 6 + (int)threadIdx.x; i < max_det * 6; i += GREEDY_NT) ob[i] = 0.f;
    if (keep_idx)
      for (int i = k + (int)threadIdx.x; i < max_det; i += GREEDY_NT) keep_idx[(size_t)b * max_det + i] = -1;
  }
  if (threadIdx.x == 0) {
    counts[b] = s_kept;
    // the sorted list was only the top part of the candidates and ran out before max_det boxes were kept: this image goes through
    // the full path (second pair of launches)
    if (redo) redo[b] = (partial && partial[b] && s_kept < max_det) ? 1 : 0;
  }
}

constexpr int NMS_COUNTERS = 8;  // int arrays of length B in front of the workspace: count, nsorted, partial, redo, redo (second stage), spare
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int pow2_ge(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

extern "C" size_t upa_nms_workspace_bytes(int b, int nc, int a, int multi_label, int max_nms) {
  const size_t cap = (size_t)a * (multi_label ? nc : 1);
  const size_t selcap = (size_t)pow2_ge(max_nms < 2 ? 2 : max_nms);
  // count, nsorted, partial, redo flags (+ coarse histograms and per-wave best bins)
  const size_t counters = (size_t)b * NMS_COUNTERS + (multi_label ? (size_t)b * (COARSE_BINS + 4 * (size_t)cdiv(a, 256) * (size_t)cdiv(nc, 8)) : 0);
  return 256 + align_up(counters * sizeof(int), 256) + (size_t)b * cap * 8 + (size_t)b * selcap * 8;
}

static int nms_batched_impl(const float* pred, int b, int nc, int a, float conf_thres, float iou_thres, int multi_label,
                            int agnostic, const uint8_t* classes_mask, int max_det, int max_nms, float max_wh,
                            float* out, int32_t* counts, int32_t* keep_idx, void* workspace, size_t workspace_bytes,
                            const u64* best_keys, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(pred && out && counts && workspace, "nms: null pointer");
  const int stages_mode = UPA_OPT(opts, nms_stages);
  const int first_prefix_opt = UPA_OPT(opts, nms_first_prefix);
  UPA_CHECK_ARG(stages_mode >= 0 && stages_mode <= 2, "nms: opts.nms_stages must be 0, 1 or 2");
  UPA_CHECK_ARG(first_prefix_opt == 0 || first_prefix_opt == -1 || (first_prefix_opt >= 256 && first_prefix_opt < LDS_SORT_CAP),
                "nms: opts.nms_first_prefix must be 0, -1 or in [256, %d)", LDS_SORT_CAP);
  const int first_prefix = first_prefix_opt == 0 ? 4096 : first_prefix_opt;
  UPA_CHECK_ARG(b > 0 && nc > 0 && a > 0, "nms: bad shape");
  UPA_CHECK_ARG(conf_thres >= 0.f && conf_thres <= 1.f, "Invalid Confidence threshold %f, valid values are between 0.0 and 1.0",
                conf_thres);
  UPA_CHECK_ARG(iou_thres >= 0.f && iou_thres <= 1.f, "Invalid IoU %f, valid values are between 0.0 and 1.0", iou_thres);
  UPA_CHECK_ARG(max_det >= 1 && max_det <= MAX_DET_CAP, "nms: max_det must be in [1, %d]", MAX_DET_CAP);
  UPA_CHECK_ARG(max_nms >= 1 && max_nms <= (1 << 20), "nms: max_nms out of range");
  UPA_CHECK_ARG((long)a * nc < (1L << 31), "nms: a*nc overflows the 32-bit candidate index");
  if (workspace_bytes < upa_nms_workspace_bytes(b, nc, a, multi_label, max_nms)) {
    upa_set_error("nms: workspace too small");
    return UPA_EWORKSPACE;
  }
  const bool ws_has_coarse = multi_label != 0;
  multi_label = multi_label && nc > 1;  // nms.py:82
  const long cap = (long)a * (multi_label ? nc : 1);
  const int selcap = pow2_ge(max_nms < 2 ? 2 : max_nms);
  char* ws = (char*)workspace;
  ws = (char*)align_up((size_t)ws, 256);
  int* count = (int*)ws;
  int* nsorted = count + b;
  int* partial = count + 2 * b;
  int* redo = count + 3 * b;
  ws += align_up(((size_t)b * NMS_COUNTERS + (ws_has_coarse ? (size_t)b * (COARSE_BINS + 4 * (size_t)cdiv(a, 256) * (size_t)cdiv(nc, 8)) : 0)) * sizeof(int), 256);
  u64* keys = (u64*)ws;
  u64* sel = keys + (size_t)b * cap;
  hipStream_t s = (hipStream_t)stream;
  // counters zeroed by a kernel, not hipMemsetAsync (see upa_zero_words, common.h); with best-class keys nothing counts in
  // global memory: the sort kernel compacts its own candidates and always writes nsorted[b]
  // long candidate lists (multi-label validation): first only a sorted prefix of at most LDS_SORT_CAP candidates, sorted in LDS - the
  // full top-max_nms select + global-memory sort (8.4 ms per batch-32 call, two thirds of the validate step's GPU time) only for images
  // whose greedy pass ran out of candidates before max_det boxes were kept
  const bool two_stage = cap > LDS_SORT_CAP && max_nms > LDS_SORT_CAP;
  int* coarse = (two_stage && multi_label && !best_keys && stages_mode != 2) ? count + NMS_COUNTERS * b : nullptr;
  int prefixes[3], np = 0;
  if (two_stage && coarse && first_prefix > 0) prefixes[np++] = first_prefix;
  if (two_stage) prefixes[np++] = LDS_SORT_CAP;
  prefixes[np++] = 0;
  // with the histogram, the first prefix's keys are the only ones written up front (nms_hist_kernel + nms_emit_kernel)
  const bool emit = coarse && stages_mode == 0;
  int* wave_best = count + NMS_COUNTERS * b + b * COARSE_BINS;  // [B][4 * workgroups per image][class groups], every word written by nms_hist_kernel
  int *redo2 = count + 4 * b, *mode = count + 5 * b, *pcount = count + 6 * b, *count2 = count + 7 * b;
  if (!best_keys) upa_zero_words(count, coarse ? NMS_COUNTERS * b + b * COARSE_BINS : 2 * b, s);
  const dim3 cgrid((unsigned)cdiv(a, 256), (unsigned)b);
  if (emit) {
    hipLaunchKernelGGL(nms_hist_kernel, cgrid, dim3(256), 0, s, pred, nc, a, conf_thres, classes_mask, coarse, wave_best);
    hipLaunchKernelGGL(nms_emit_kernel, cgrid, dim3(256), 0, s, pred, nc, a, conf_thres, classes_mask, (const int*)coarse,
                       (const int*)wave_best, prefixes[0], count, mode, pcount, keys, cap, sel, selcap);
  } else if (!best_keys) {  // (with best-class keys the sort kernel compacts its own candidates)
    hipLaunchKernelGGL(nms_candidates_kernel, cgrid, dim3(256), 0, s, pred, b, nc, a, conf_thres, multi_label, classes_mask, count,
                       keys, cap, coarse, (const int*)nullptr);
  }
  UPA_LAUNCH_CHECK();
  {
    hipError_t e = upa_full_lds<nms_sort_kernel>();
    if (e != hipSuccess) { upa_set_error("nms: cannot raise LDS limit: %s", hipGetErrorString(e)); return UPA_ELAUNCH; }
  }
  // Stages: (sort a prefix of the score order, greedy pass) pairs over growing prefixes; a pair after the first only works on the
  // images the greedy pass before it flagged (it ran out of candidates before max_det boxes were kept - every other workgroup returns
  // at once), the last pair is the full top-max_nms path.  With the coarse histogram a prefix costs little, so the first one is short
  // (about 4096 candidates: a bitonic sort a quarter the length of the LDS_SORT_CAP one).
  int* flags[3] = {redo, redo2, nullptr};  // written by pair i, read by pair i + 1 (redo2 only exists zeroed: `coarse`)
  for (int i = 0; i < np; ++i) {
    const int* only = i ? flags[i - 1] : nullptr;
    const bool last = i == np - 1;
    if (emit && i == 1)  // flagged images: now all their keys (own slot counter: count[] already holds the images' totals)
      hipLaunchKernelGGL(nms_candidates_kernel, cgrid, dim3(256), 0, s, pred, b, nc, a, conf_thres, multi_label, classes_mask, count2,
                         keys, cap, (int*)nullptr, only);
    hipLaunchKernelGGL(nms_sort_kernel, dim3((unsigned)b), dim3(SORT_NT), LDS_SORT_CAP * 8, s, (const int*)((emit && i) ? count2 : count),
                       nsorted, keys, sel, cap, selcap, max_nms, best_keys, nc, a, conf_thres, classes_mask, prefixes[i], partial, only,
                       (const int*)((last || (emit && i == 0)) ? nullptr : coarse), (const int*)((emit && i == 0) ? mode : nullptr),
                       (const int*)pcount);
    UPA_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_greedy_kernel, dim3((unsigned)b), dim3(GREEDY_NT), 0, s, pred, nc, a, nsorted, sel, selcap,
                       iou_thres, agnostic, max_wh, max_det, out, counts, keep_idx, (const int*)(last ? nullptr : partial),
                       last ? (int*)nullptr : flags[i], only);
    UPA_LAUNCH_CHECK();
  }
  return UPA_OK;
}

#ifdef UPA_GREEDY_PROF
extern "C" int upa_debug_greedy_prof(unsigned long long* out12) {
  unsigned long long* out8 = out12;
  unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_greedy_prof), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_greedy_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int upa_nms_batched(const float* pred, int b, int nc, int a, float conf_thres, float iou_thres, int multi_label,
                               int agnostic, const uint8_t* classes_mask, int max_det, int max_nms, float max_wh,
                               float* out, int32_t* counts, int32_t* keep_idx, void* workspace, size_t workspace_bytes,
                               void* stream) {
  return nms_batched_impl(pred, b, nc, a, conf_thres, iou_thres, multi_label, agnostic, classes_mask, max_det, max_nms, max_wh,
                          out, counts, keep_idx, workspace, workspace_bytes, nullptr, nullptr, stream);
}

extern "C" int upa_nms_batched_opts(const float* pred, int b, int nc, int a, float conf_thres, float iou_thres, int multi_label,
                                    int agnostic, const uint8_t* classes_mask, int max_det, int max_nms, float max_wh,
                                    float* out, int32_t* counts, int32_t* keep_idx, void* workspace, size_t workspace_bytes,
                                    const upa_opts* opts, void* stream) {
  return nms_batched_impl(pred, b, nc, a, conf_thres, iou_thres, multi_label, agnostic, classes_mask, max_det, max_nms, max_wh,
                          out, counts, keep_idx, workspace, workspace_bytes, nullptr, opts, stream);
}

extern "C" int upa_nms_batched_hot(const float* pred, int b, int nc, int a, float conf_thres, float iou_thres, int multi_label,
                                   int agnostic, const uint8_t* classes_mask, int max_det, int max_nms, float max_wh,
                                   float* out, int32_t* counts, int32_t* keep_idx, void* workspace, size_t workspace_bytes,
                                   const unsigned long long* best_keys, void* stream) {
  UPA_CHECK_ARG(best_keys, "nms_hot: best-class keys missing");
  UPA_CHECK_ARG(!(multi_label && nc > 1), "nms_hot: the keys follow the single-label rule (best class per anchor)");
  return nms_batched_impl(pred, b, nc, a, conf_thres, iou_thres, multi_label, agnostic, classes_mask, max_det, max_nms, max_wh,
                          out, counts, keep_idx, workspace, workspace_bytes, (const u64*)best_keys, nullptr, stream);
}
