// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the dilated-CNN patch path.
// Written for wave64 / MFMA / 160 KiB LDS only; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRS_OK 0
#define DRS_ERR_ARG 1
#define DRS_ERR_HIP 2

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A view of a spatially padded NHWC activation slab [B][S+2P][S+2P][ld]; the slice starts at channel `coff`.
// `base` addresses the padded element (b=0, y=-P, x=-P, ch=0).
struct ActView {
  float* base;     // may be null when only the bf16 terms are wanted
  int S;     // patch side (interior)
  int P;     // halo width (zeros)
  int ld;    // floats per pixel
  int coff;  // first channel of the slice
  uint16_t* terms; // or null: the split-bf16 image of the same slab (conv_split.hip), written alongside
  int nt;          // terms per element (2 or 3)
};

// 4 consecutive channels (element offset e, a multiple of 4, in the slab's fp32 indexing) -> the fp32 slab and/or its
// bf16 term image: term s of element e lives at (e & ~31) * nt + 32 s + (e & 31); term s = rne_bf16(x - earlier terms).
__device__ __forceinline__ void view_store4(const ActView& v, size_t e, f32x4 x) {
  if (v.base) *reinterpret_cast<f32x4*>(v.base + e) = x;
  if (v.terms) {
    uint16_t* t = v.terms + (e & ~(size_t)31) * v.nt + (e & 31);
    f32x4 r = x;
    for (int s = 0; s < v.nt; ++s) {
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = (__bf16)r[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] -= (float)h[j];
      *reinterpret_cast<bf16x4_t*>(t + 32 * s) = h;
    }
  }
}

// q = n / d, r = n % d for 0 <= n < 2^24 (exact in f32), d >= 1; rcp = 1.0f/d.
__device__ __forceinline__ void divmod24(int n, int d, float rcp, int& q, int& r) {
  q = (int)((float)n * rcp);
  r = n - q * d;
  if (r < 0) { q -= 1; r += d; }
  else if (r >= d) { q += 1; r -= d; }
}

// flat interior pixel p = (b*S + y)*S + x  ->  element offset of padded pixel (b, y+dy, x+dx), channel 0 of the buffer
__device__ __forceinline__ uint32_t padded_pixel_off(int p, int S, int P, int ld, float rcpS, float rcpSS, int dy, int dx) {
  int b, rem, y, x;
  divmod24(p, S * S, rcpSS, b, rem);
  divmod24(rem, S, rcpS, y, x);
  const int Sp = S + 2 * P;
  return (uint32_t)(((b * Sp + y + P + dy) * Sp + (x + P + dx))) * (uint32_t)ld;
}

// drs_step_prep (conv_mfma.hip; the step engine's one launch between forward and backward)
constexpr int STEP_PREP_MAX = 24;
struct StepPrepArgs {
  const float* w[STEP_PREP_MAX];      // filters [k][k][cin][cout] ...
  float* wt[STEP_PREP_MAX];           // ... flipped and transposed into [k][k][cout][cin]
  int k[STEP_PREP_MAX], cin[STEP_PREP_MAX], cout[STEP_PREP_MAX];
  int n;
  unsigned int* z0; int nz0;          // zero fills
  float* z1; int nz1;
};

// Filter-tap rows whose whole input row range lies in the zero halo contribute exact zeros to every pixel of an M tile:
// for the tile of pixels [m0, m0 + BM) (clipped to M) return the range [lo, hi) of tap rows u, dy = u * rate - pad, for
// which some pixel row y of the tile has 0 <= y + dy < S.  A tile that crosses an image boundary keeps every tap row.
// (Rows are skipped, columns are not: a tile spans whole image rows.)  Adding the skipped products would add +-0.
__device__ __forceinline__ void live_tap_rows(int m0, int BM, int M, int S, int k, int rate, int pad, float rcpS, float rcpSS,
                                              int& lo, int& hi) {
  const int p1 = (m0 + BM - 1 < M ? m0 + BM : M) - 1;
  int b0, r0, b1, r1, y0, y1, x;
  divmod24(m0, S * S, rcpSS, b0, r0);
  divmod24(p1, S * S, rcpSS, b1, r1);
  divmod24(r0, S, rcpS, y0, x);
  divmod24(r1, S, rcpS, y1, x);
  if (b0 != b1) { y0 = 0; y1 = S - 1; }
  // u * rate - pad >= -y1   and   u * rate - pad <= S - 1 - y0
  const int a = pad - y1;
  lo = a > 0 ? (a + rate - 1) / rate : 0;
  const int bnum = S - 1 - y0 + pad;
  hi = bnum / rate + 1;
  hi = hi < k ? hi : k;
  if (lo >= hi) { lo = 0; hi = k; }        // cannot happen for SAME padding (the centre taps always land inside); stay safe
}

// Filter gradient: a tile of filter rows belongs to tap rows u_first .. u_last; pixel rows y whose shifted rows y + u*rate - pad
// all fall outside the image meet only halo zeros in X.  [lo, hi) = the pixel range (inside one image of S*S pixels) of the rows
// that do meet image data; everything outside contributes exact zeros and is never fetched.
__host__ __device__ __forceinline__ void live_pixel_range(int row_first, int row_last, int Cin, int k, int rate, int pad, int S, int enable,
                                                          int& lo, int& hi) {
  const int dy_min = (row_first / Cin / k) * rate - pad, dy_max = (row_last / Cin / k) * rate - pad;
  int ylo = -dy_max > 0 ? -dy_max : 0;
  int yhi = S - dy_min < S ? S - dy_min : S;
  if (!enable || ylo >= yhi) { ylo = 0; yhi = S; }
  lo = ylo * S;
  hi = yhi * S;
}

// first 32-pixel chunk after chunk c that contains a pixel of some image's live range [b*S2 + lo, b*S2 + hi)
__device__ __forceinline__ int next_live_chunk(int c, int S2, float rcpSS, int lo, int hi) {
  const int p = 32 * (c + 1);
  int b, r;
  divmod24(p, S2, rcpSS, b, r);
  if (r + 31 >= lo && r < hi) return c + 1;           // meets the live rows of image b
  if (r + 31 >= S2 + lo) return c + 1;                // ... or reaches into those of image b + 1
  return ((r < lo ? b : b + 1) * S2 + lo) >> 5;       // the chunk holding the first live pixel ahead
}

// Train-mode batch-norm statistics of one output tile of a convolution kernel (the first half of isprs:655-663), taken in the
// epilogue from the accumulator registers: per tile column (sum v, M2 = sum (v - tile mean)^2), TWO-PASS inside the tile as
// TensorFlow's batch_norm is two-pass over the batch.  The tiles are combined in fp64 by Chan's formula
// (drs_conv_stats_reduce: sum z^2 = sum_i (M2_i + s_i^2 / n_i)), so nothing is lost to cancellation in fp32 whatever
// |mean| / std is; a plain per-tile (sum v, sum v^2) loses ~(mean/std)^2 * 2^-24 of the variance.
//   NSLOT column slots per lane; slot ni is tile column col(ni); `lanesum` adds the lanes of a wave that hold the same column
//   (a fixed butterfly); the WM waves that share a column are added through LDS in wave order; `each(ni, f)` calls f(v) for
//   every valid value this lane holds of slot ni; `writer` = this lane writes its wave's column totals;
//   red: 2 * WM * BN floats of LDS no wave reads any more; dst: this tile's [BN][2] run of the statistics slab row.
template <int NSLOT, int WM, int BN, class Col, class LaneSum, class Each>
__device__ __forceinline__ void tile_column_stats(float* red, int t, int wm, bool writer, float n_tile, Col col, LaneSum lanesum,
                                                  Each each, float* dst) {
  float* red2 = red + WM * BN;
#pragma unroll
  for (int ni = 0; ni < NSLOT; ++ni) {
    float s = 0.f;
    each(ni, [&](float v) { s += v; });
    s = lanesum(s);
    if (writer) red[wm * BN + col(ni)] = s;
  }
  __syncthreads();
  const float rn = 1.0f / n_tile;
#pragma unroll
  for (int ni = 0; ni < NSLOT; ++ni) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) tot += red[w * BN + col(ni)];
    const float mean = tot * rn;
    float q = 0.f;
    each(ni, [&](float v) { const float d = v - mean; q += d * d; });
    q = lanesum(q);
    if (writer) red2[wm * BN + col(ni)] = q;
  }
  __syncthreads();
  if (t >= 0 && t < BN && dst) {
    float u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) { u1 += red[w * BN + t]; u2 += red2[w * BN + t]; }
    dst[2 * t] = u1;
    dst[2 * t + 1] = u2;
  }
}

// Host-side hint, set by the step engine around its two-stream backward pass (engine.hip): the launches this thread enqueues while it
// is set belong to the chain the step waits for (batch-norm backward -> input gradient) and share the chip with a filter gradient on a
// stream of its own; their waves then take the top priority (s_setprio 3).  1 = the convolution launches (input gradient, stream-K
// fix-up); 2 = the batch-norm backward launches as well -- what the engine asks for when the step carries collectives: with an
// all-reduce launch between the statistics and `bn_bwd_apply` of every block the elementwise launches at the top priority win 0.4-2.5 %
// at every patch side, without collectives they lose 1-4 % from S = 45 (profiles/r05/chain_priority_ab.txt, last block).  A hint
// about scheduling only: never changes a result.  drs_g_chain_mode (development switch drs_debug_chain_mode): -1 = as the engine
// asks (default), 0 / 1 / 2 = that level whatever it asks.
inline int drs_chain_level(int asked, int mode) { return asked == 0 ? 0 : (mode < 0 ? asked : mode); }
extern thread_local int drs_tl_chain;
extern int drs_g_chain_mode;

// development switch shared by the convolution kernels (drs_debug_skip_taps): 0 = multiply the all-halo taps / chunks too,
// 1 = skip them where it pays, 2 = skip them always
extern int drs_g_skip_halo_taps;
// Skipping makes the workgroups of a launch unequal, and that only pays on a grid of many rounds: in-process A/B at 64x64,
// forward conv8 -6 % at 8192 workgroups (B = 128), -4.6 % at 4096, +3 % at 2048, +12 % at 1024 (B = 16); filter gradient
// -1..-4 % at B = 128, +6..18 % on several layers at B = 32.  So: forward / input gradient from 4096 workgroups (128-row
// tiles assumed), filter gradient from 2^19 pixels.
static inline int drs_skip_halo_taps_fwd(long long M, int cout) {
  if (drs_g_skip_halo_taps != 1) return drs_g_skip_halo_taps == 2;
  const long long wgs = ((M + 127) / 128) * ((cout + 127) / 128);
  return wgs >= 4096;
}
static inline int drs_skip_halo_taps_wgrad(long long M) {
  if (drs_g_skip_halo_taps != 1) return drs_g_skip_halo_taps == 2;
  return M >= (1LL << 19);
}

// Bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch), so give every
// XCD one contiguous chunk of the logical tile order (neighbouring tiles then share that XCD's 4 MiB L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + (bid >> 3);
}

static inline int drs_check(hipError_t e) { return e == hipSuccess ? DRS_OK : DRS_ERR_HIP; }
#define DRS_LAUNCH_CHECK() drs_check(hipGetLastError())
// launch with the thread's sticky "last error" cleared first, so that DRS_LAUNCH_CHECK reports this launch only
#define DRS_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
