// HBM-bound kernels of the dilated-CNN patch path on gfx950: batch-norm statistics and normalisation fused with the
// (leaky) ReLU and the 3x3/stride-1 max-pool (forward and backward), the 1x1 classifier fused with the per-pixel
// softmax cross-entropy, arg-max, confusion matrix and their gradients, and the momentum update.
//
// Reference call sites (/root/reference/isprs_dilated_random.py): _batch_norm :655-663, activation :717-721 and
// leaky_relu :620-621, _max_pool :745-746, classifier :1024-1031, loss_def :1089-1099 (masked form:
// contest_dilated_random.py:881-901), MomentumOptimizer :1685-1687, tf.argmax :1690, calc_accuracy_by_crop :510-531.
//
// All cross-workgroup reductions go through per-workgroup slabs that are summed in a fixed order (no float atomics),
// so a step is bitwise reproducible.  Everything is channels-last; a thread owns 4 consecutive channels (16-B
// accesses), consecutive lanes own consecutive channel quads, so every wave access is a run of full cache lines.
#include "drs_common.hpp"

namespace {

constexpr float BN_EPS = 0.001f;

__device__ __forceinline__ float act(float x, float alpha) { return fmaxf(alpha * x, x); }   // alpha = 0 -> ReLU

// ------------------------------------------------------------------------------------------------ BN statistics
// Column sums of a row-major fp32 slab [nrows][ncols] in a FIXED order (bitwise reproducible), accumulated in fp64:
// level 1: COLSUM_BLOCKS row bands x 4 row lanes per band -> scratch[COLSUM_BLOCKS][ncols] (fp64);
// level 2: one thread per column adds the bands in order.  Used for the batch-norm statistic slabs
// (ncols = 2*C: per channel (sum, sum of squares) or (sum g, sum g*xhat)) and for the classifier gradient slabs.
constexpr int COLSUM_BLOCKS = 32;

__global__ void colsum_l1_kernel(const float* __restrict__ in, int nrows, int ncols, double* __restrict__ scratch) {
  __shared__ double sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int rpb = (nrows + COLSUM_BLOCKS - 1) / COLSUM_BLOCKS;
  const int r0 = blockIdx.y * rpb;
  int r1 = r0 + rpb;
  r1 = r1 < nrows ? r1 : nrows;
  double s = 0.0;
  if (c < ncols)
    for (int r = r0 + ty; r < r1; r += 4) s += (double)in[(size_t)r * ncols + c];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < ncols) scratch[(size_t)blockIdx.y * ncols + c] = ((sh[0][tx] + sh[1][tx]) + sh[2][tx]) + sh[3][tx];
}

template <typename OUT>
__global__ void colsum_l2_kernel(const double* __restrict__ scratch, int ncols, OUT* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  double s = 0.0;
#pragma unroll 4
  for (int b = 0; b < COLSUM_BLOCKS; ++b) s += scratch[(size_t)b * ncols + c];
  out[c] = (OUT)s;
}

// Single-launch form for the batch-norm statistic slabs [nrows][C][2]: one workgroup per 4 channels, 64 row lanes, fp64
// accumulation, a fixed halving tree in LDS (bitwise reproducible).  CHAN: the slab comes from a convolution epilogue
// (tile_column_stats): row r holds per channel (s_r, M2_r) over n_r = min(mtile, M - r*mtile) pixels and the rows combine by
// Chan's formula, sum z^2 = sum_r (M2_r + s_r^2 / n_r); otherwise both columns are plain sums (the backward pass's
// (sum g, sum g*xhat)).  With mean_rstd != null the workgroup also finishes the batch norm of its channels (bn_finish_kernel's
// arithmetic): statistics reduction and finish are then ONE launch (single-rank case; under data parallelism the all-reduce of
// `sums` sits between the two).
// hp: the launch belongs to the chain a step waits for while another launch shares the chip (engine.hip, two-stream backward pass):
// its waves take the top priority
#define DRS_CHAIN_PRIO() do { if (hp) __builtin_amdgcn_s_setprio(3); } while (0)
constexpr int STAT_CH = 2, STAT_LANES = 128;       // (4 x 64 measured 32 us at 4096 rows x 256 channels: latency-bound, so more row lanes and workgroups)
template <bool CHAN>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ partial, int nrows, int C, int M, int mtile,
                                                       double* __restrict__ sums, double count, float* __restrict__ mean_rstd,
                                                       float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                       float one_minus_decay, int bessel, float* __restrict__ means, int hp) {
  DRS_CHAIN_PRIO();
  __shared__ double sh[STAT_LANES][STAT_CH][2];
  const int ch = threadIdx.x & (STAT_CH - 1), rl = threadIdx.x / STAT_CH;
  const int c = blockIdx.x * STAT_CH + ch;
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
    const float2* src = reinterpret_cast<const float2*>(partial) + c;
#pragma unroll 8
    for (int r = rl; r < nrows; r += STAT_LANES) {
      const float2 v = src[(size_t)r * C];
      const double s = (double)v.x;
      a0 += s;
      if (CHAN) {
        const int rem = M - r * mtile;
        a1 += (double)v.y + s * s / (double)(rem < mtile ? rem : mtile);
      } else {
        a1 += (double)v.y;
      }
    }
  }
  sh[rl][ch][0] = a0;
  sh[rl][ch][1] = a1;
  __syncthreads();
  for (int d = STAT_LANES / 2; d >= 1; d >>= 1) {
    if (rl < d) { sh[rl][ch][0] += sh[rl + d][ch][0]; sh[rl][ch][1] += sh[rl + d][ch][1]; }
    __syncthreads();
  }
  if (rl != 0 || c >= C) return;
  const double s0 = sh[0][ch][0], s1 = sh[0][ch][1];
  if (sums) { sums[2 * c] = s0; sums[2 * c + 1] = s1; }
  if (means) { means[2 * c] = (float)(s0 / count); means[2 * c + 1] = (float)(s1 / count); }     // bn_bwd_apply_kernel's coefficients, its expressions
  if (mean_rstd) {
    const double m = s0 / count;
    double var = s1 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, vf = (float)var;
    mean_rstd[2 * c] = mf;
    mean_rstd[2 * c + 1] = 1.0f / sqrtf(vf + BN_EPS);
    if (moving_mean) {
      const float vu = bessel && count > 1.0 ? (float)(var * (count / (count - 1.0))) : vf;
      moving_mean[c] = moving_mean[c] - (moving_mean[c] - mf) * one_minus_decay;
      moving_var[c] = moving_var[c] - (moving_var[c] - vu) * one_minus_decay;
    }
  }
}

// sums (global over the batch, all ranks) -> mean, rstd; moving averages updated as
// moving -= (moving - value) * (1 - decay)  (tf assign_moving_average); the moving variance takes the
// Bessel-corrected batch variance (TF fused batch norm), the normalisation the biased one.
__global__ void bn_finish_kernel(const double* __restrict__ sums, double count, int C, float* __restrict__ mean_rstd,
                                 float* __restrict__ moving_mean, float* __restrict__ moving_var, float one_minus_decay,
                                 int bessel) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = sums[2 * c] / count;
  double var = sums[2 * c + 1] / count - m * m;
  if (var < 0.0) var = 0.0;
  const float mf = (float)m, vf = (float)var;
  mean_rstd[2 * c] = mf;
  mean_rstd[2 * c + 1] = 1.0f / sqrtf(vf + BN_EPS);
  if (moving_mean) {
    const float vu = bessel && count > 1.0 ? (float)(var * (count / (count - 1.0))) : vf;
    moving_mean[c] = moving_mean[c] - (moving_mean[c] - mf) * one_minus_decay;
    moving_var[c] = moving_var[c] - (moving_var[c] - vu) * one_minus_decay;
  }
}

// eval mode: (mean, rstd) from the moving statistics
__global__ void bn_eval_coeffs_kernel(const float* __restrict__ moving_mean, const float* __restrict__ moving_var, int C,
                                      float* __restrict__ mean_rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean_rstd[2 * c] = moving_mean[c];
  mean_rstd[2 * c + 1] = 1.0f / sqrtf(moving_var[c] + BN_EPS);
}

// ------------------------------------------------------------------------------------------------ BN + act + pool, forward
// z [B*S*S][C] (raw conv output) -> out view (zero halo written here), optional arg-max code per element.
// grid.y = padded rows (b, yy), grid.x * block covers (xx, channel quad) of one padded row.
template <bool POOL>
__global__ void bn_act_pool_fwd_kernel(const float* __restrict__ z, int B, int S, int C, const float* __restrict__ mean_rstd,
                                       float alpha, ActView out, unsigned char* __restrict__ idx) {
  const int CQ = C >> 2;
  const int Sp = S + 2 * out.P;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Sp * CQ) return;
  const int xx = e / CQ, cq = e - xx * CQ;
  const int b = blockIdx.y / Sp, yy = blockIdx.y - b * Sp;
  const size_t dst = ((size_t)(b * Sp + yy) * Sp + xx) * out.ld + out.coff + cq * 4;
  const int y = yy - out.P, x = xx - out.P;
  if (y < 0 || y >= S || x < 0 || x >= S) {
    view_store4(out, dst, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  const f32x4 mr0 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8);        // (m0, r0, m1, r1)
  const f32x4 mr1 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8 + 4);    // (m2, r2, m3, r3)
  const float mu[4] = {mr0[0], mr0[2], mr1[0], mr1[2]};
  const float rs[4] = {mr0[1], mr0[3], mr1[1], mr1[3]};
  const size_t pix = ((size_t)b * S + y) * S + x;
  f32x4 best;
  unsigned code[4] = {4u, 4u, 4u, 4u};
  if (!POOL) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(z + pix * C + cq * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) best[j] = act((v[j] - mu[j]) * rs[j], alpha);
  } else {
    bool first = true;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int ny = y + dy, nx = x + dx;
        if (ny < 0 || ny >= S || nx < 0 || nx >= S) continue;       // SAME padding never wins
        const f32x4 v = *reinterpret_cast<const f32x4*>(z + (((size_t)b * S + ny) * S + nx) * C + cq * 4);
        const unsigned cd = (unsigned)((dy + 1) * 3 + (dx + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = act((v[j] - mu[j]) * rs[j], alpha);
          if (first || a > best[j]) { best[j] = a; code[j] = cd; }   // first maximum in scan order wins
        }
        first = false;
      }
    }
  }
  view_store4(out, dst, best);
  if (POOL && idx) {
    const unsigned packed = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
    *reinterpret_cast<unsigned*>(idx + pix * C + cq * 4) = packed;
  }
}

// ------------------------------------------------------------------------------------------------ BN + act + pool, backward
// pass A: route the incoming gradient back through the pool (gather form over the saved arg-max codes) and the
// activation; write g_xhat and per-workgroup partial sums of (g_xhat, g_xhat * xhat) per channel.
// block = (C/4) x PT threads; a workgroup walks `rows_per_block` pixels.
template <bool POOL>
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ ga, int ld_ga, int coff_ga, const float* __restrict__ z,
                                     const unsigned char* __restrict__ idx, int B, int S, int C,
                                     const float* __restrict__ mean_rstd, float alpha, float* __restrict__ gxh,
                                     float* __restrict__ partial, int rows_per_block, int hp) {
  DRS_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float red[];   // [PT][C][2]
  const int CQ = C >> 2;
  const int cq = threadIdx.x % CQ, tp = threadIdx.x / CQ, PT = blockDim.x / CQ;
  const int M = B * S * S;
  const f32x4 mr0 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8);
  const f32x4 mr1 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8 + 4);
  const float mu[4] = {mr0[0], mr0[2], mr1[0], mr1[2]};
  const float rs[4] = {mr0[1], mr0[3], mr1[1], mr1[3]};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int p0 = blockIdx.x * rows_per_block;
  int pend = p0 + rows_per_block;
  pend = pend < M ? pend : M;
  const float rcpS = 1.0f / (float)S, rcpSS = 1.0f / (float)(S * S);
  for (int p = p0 + tp; p < pend; p += PT) {
    f32x4 g;
    if (!POOL) {
      g = *reinterpret_cast<const f32x4*>(ga + (size_t)p * ld_ga + coff_ga + cq * 4);
    } else {
      int b, rem, y, x;
      divmod24(p, S * S, rcpSS, b, rem);
      divmod24(rem, S, rcpS, y, x);
      g = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int qy = y + dy, qx = x + dx;
          if (qy < 0 || qy >= S || qx < 0 || qx >= S) continue;
          const size_t q = ((size_t)b * S + qy) * S + qx;
          const unsigned packed = *reinterpret_cast<const unsigned*>(idx + q * C + cq * 4);
          const unsigned want = (unsigned)((1 - dy) * 3 + (1 - dx));   // where p sits in q's window
          const f32x4 gq = *reinterpret_cast<const f32x4*>(ga + q * ld_ga + coff_ga + cq * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (((packed >> (8 * j)) & 0xffu) == want) g[j] += gq[j];
        }
      }
    }
    const f32x4 zv = *reinterpret_cast<const f32x4*>(z + (size_t)p * C + cq * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (zv[j] - mu[j]) * rs[j];
      const float gx = xh > 0.f ? g[j] : g[j] * alpha;
      o[j] = gx;
      s1[j] += gx;
      s2[j] += gx * xh;
    }
    *reinterpret_cast<f32x4*>(gxh + (size_t)p * C + cq * 4) = o;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[((size_t)tp * C + cq * 4 + j) * 2] = s1[j];
    red[((size_t)tp * C + cq * 4 + j) * 2 + 1] = s2[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float u1 = 0.f, u2 = 0.f;
    for (int r = 0; r < PT; ++r) { u1 += red[((size_t)r * C + c) * 2]; u2 += red[((size_t)r * C + c) * 2 + 1]; }
    partial[((size_t)blockIdx.x * C + c) * 2] = u1;
    partial[((size_t)blockIdx.x * C + c) * 2 + 1] = u2;
  }
}

// ------------------------------------------------------------------------------------------------ pooled nets: sliding kernels
// With the 3x3 pool every output needs a 3x3 neighbourhood.  Instead of 9 loads per output, a thread owns one image
// column x and four channels and slides down a strip of rows: ONE load per output of its own column, the horizontal
// neighbours through LDS, the two previous rows' partial results in registers (forward: row maxima + positions; backward: the
// running sums of the rows above / at / below the window row that arrives).
// block = TX columns x (C/4) channel quads; grid.x = column blocks, grid.y = (image, row strip).
struct SlideCfg { int TX, ncol, nstrips, rps; };
// (development switches: drs_debug_slide_*.  Workgroup target 2048 -> 5120 in r04: in-step sweep at B = 128 over 1280 .. 7680 --
//  the forward kernel does not care (1.19-1.21 ms for the 8 layers), the backward one 1.85 -> 1.75 ms: profiles/r04/slide_blocks_sweep.txt)
static int g_slide_blocks = 5120, g_slide_minrows = 8;
static int g_slide_rowpad = 0;      // development switch (drs_debug_slide_rowpad; timing experiment): phantom pixels after every stored image row of z / idx / ga / gxh
static bool slide_ok(int C) { return C / 4 <= 128; }        // two columns or more per workgroup (the neighbours go through LDS)
static SlideCfg slide_cfg(int B, int S, int C) {
  SlideCfg c;
  const int CQ = C / 4;
  c.TX = 256 / CQ;
  c.ncol = (S + c.TX - 1) / c.TX;
  c.nstrips = 1;
  while ((long long)B * c.ncol * c.nstrips < g_slide_blocks && S / (c.nstrips * 2) >= g_slide_minrows) c.nstrips *= 2;
  // the per-rank batches of data parallelism (16 patches of 25 .. 85 pixels): few workgroups, each a chain of dependent row loads
  // -- the launch is latency-bound, so shorter strips (down to 2-3 rows: 2 extra rows loaded per strip, out of L2) and more of them
  while ((long long)B * c.ncol * c.nstrips < 768 && S / (c.nstrips * 2) >= 2) c.nstrips *= 2;
  c.rps = (S + c.nstrips - 1) / c.nstrips;
  return c;
}

// zero the halo of a view (the sliding forward kernel writes interiors only)
__global__ void zero_halo_kernel(ActView out, int B, int C) {
  const int CQ = C >> 2;
  const int Sp = out.S + 2 * out.P;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Sp * CQ) return;
  const int xx = e / CQ, cq = e - xx * CQ;
  const int b = blockIdx.y / Sp, yy = blockIdx.y - b * Sp;
  const int y = yy - out.P, x = xx - out.P;
  if (y >= 0 && y < out.S && x >= 0 && x < out.S) return;
  view_store4(out, ((size_t)(b * Sp + yy) * Sp + xx) * out.ld + out.coff + cq * 4, f32x4{0.f, 0.f, 0.f, 0.f});
}

// Both sliding kernels, r03: a thread is a chain of dependent rows; such a launch is bound by the instruction stream of the few waves
// a SIMD holds and by what they keep in flight, not by HBM (one 16-byte load per wave at first, each neighbour under its own branch
// and wait: 0.56 of HBM; profiles/r03/elementwise_sliding_kernels.txt).  So each thread loads ONLY its
// own column, three rows ahead of the row it works on, and the horizontal neighbours come through LDS: every thread publishes what
// it derived from its own load (the activation forward; the (position word, gradient) pair backward), one barrier per row, two
// cells in turn.  The first / last column thread of a workgroup also loads and publishes the column beyond its edge.  Columns
// outside the image are read from the nearest column inside (clamped): forward such a duplicate cannot change a maximum and is
// never chosen as its position; backward its position word is replaced by one that matches no window position.  Addresses are a
// per-row base every lane shares plus a per-thread byte offset fixed for the strip; the row registers rotate by renaming (the loops
// are unrolled over their rings of three).
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load and store in flight, which is
// exactly the prefetch these kernels live on
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// the second half of a batch norm whose sums arrive from an all-reduce (data parallelism): (mean, rstd) of a thread's four channels
// from the global sums, bn_finish_kernel's arithmetic exactly; the chosen thread also leaves them in mean_rstd (the backward pass
// reads them) and updates the moving averages -- the "finish" launch of the multi-rank forward pass folded into its consumer
struct BnFinish { const double* sums; double count; float* moving_mean; float* moving_var; float one_minus_decay; int bessel; };
__device__ __forceinline__ void bn_finish4(const BnFinish& f, int c0, bool writer, float* __restrict__ mean_rstd, float (&mu)[4], float (&rs)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    const double m = f.sums[2 * c] / f.count;
    double var = f.sums[2 * c + 1] / f.count - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, vf = (float)var;
    mu[j] = mf;
    rs[j] = 1.0f / sqrtf(vf + BN_EPS);
    if (writer) {
      mean_rstd[2 * c] = mf;
      mean_rstd[2 * c + 1] = rs[j];
      if (f.moving_mean) {
        const float vu = f.bessel && f.count > 1.0 ? (float)(var * (f.count / (f.count - 1.0))) : vf;
        f.moving_mean[c] = f.moving_mean[c] - (f.moving_mean[c] - mf) * f.one_minus_decay;
        f.moving_var[c] = f.moving_var[c] - (f.moving_var[c] - vu) * f.one_minus_decay;
      }
    }
  }
}

__global__ __launch_bounds__(256) void bn_act_pool_fwd_slide_kernel(const float* __restrict__ z, int B, int S, int C,
                                                                    float* __restrict__ mean_rstd, float alpha, ActView out,
                                                                    unsigned char* __restrict__ idx, int nstrips, int rps, const BnFinish fin, int SW) {   // SW: pixels per stored image row of z / idx (S; development arm: more)
  extern __shared__ __attribute__((aligned(16))) float xch[];                // [2][TX + 2][C]: the activations of a row
  const int CQ = C >> 2, TX = blockDim.x / CQ;                               // TX >= 2
  const int tx = threadIdx.x / CQ, cq = threadIdx.x - tx * CQ;
  const int xr = blockIdx.x * TX + tx;
  const bool live = xr < S;                                                  // (the last column block may overhang: such threads take part, store nothing)
  const int x = live ? xr : S - 1;
  const int b = blockIdx.y / nstrips, strip = blockIdx.y - b * nstrips;
  const int y0 = strip * rps;
  int y1 = y0 + rps;
  y1 = y1 < S ? y1 : S;
  float mu[4], rs[4];
  if (fin.sums) {       // (kernel-uniform)
    bn_finish4(fin, cq * 4, blockIdx.x == 0 && blockIdx.y == 0 && tx == 0, mean_rstd, mu, rs);
  } else {
    const f32x4 mr0 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8);
    const f32x4 mr1 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8 + 4);
    mu[0] = mr0[0]; mu[1] = mr0[2]; mu[2] = mr1[0]; mu[3] = mr1[2];
    rs[0] = mr0[1]; rs[1] = mr0[3]; rs[2] = mr1[1]; rs[3] = mr1[3];
  }
  const float NEG = -__builtin_inff();
  const bool vl = x > 0;
  const bool edge_l = tx == 0, edge = edge_l || tx == TX - 1;
  const int xh = edge_l ? (x > 0 ? x - 1 : 0) : (x + 1 < S ? x + 1 : S - 1);
  const unsigned oc = (unsigned)((x * C + cq * 4) * 4), oh = edge ? (unsigned)((xh * C + cq * 4) * 4) : oc;   // (inner threads: their own column again, out of L1)
  const size_t zrow = (size_t)SW * C * 4;                                    // bytes per image row of z
  const char* zimg = reinterpret_cast<const char*>(z) + (size_t)b * S * zrow;
  f32x4* cells = reinterpret_cast<f32x4*>(xch);
  const int lc = (tx + 1) * CQ + cq, lh = (edge_l ? 0 : TX + 1) * CQ + cq, lslot = (TX + 2) * CQ;
  auto load_row = [&](int r, f32x4& vc, f32x4& vh) {
    const int rc = r < 0 ? 0 : (r >= S ? S - 1 : r);
    const char* zr = zimg + (size_t)rc * zrow;
    vc = *reinterpret_cast<const f32x4*>(zr + oc);
    vh = *reinterpret_cast<const f32x4*>(zr + oh);                           // (every thread: a load under a divergent branch makes the
  };                                                                         //  compiler drain the loads in flight at the join)
  const int Sp = S + 2 * out.P;
  const size_t orow = (size_t)Sp * out.ld * 4;                               // bytes per padded row of the output view
  char* oimg = reinterpret_cast<char*>(out.base) + ((size_t)b * Sp + out.P) * orow;
  const unsigned oo = (unsigned)(((x + out.P) * out.ld + out.coff + cq * 4) * 4);
  unsigned char* iimg = idx + (size_t)b * S * SW * C;
  // row r: (max, position 0..2 of the first maximum in scan order) over its three horizontal neighbours into (mC, cC), the registers
  // of row r refilled with row r + 3; then output row r - 1 from rows r - 2, r - 1 (mA, mB) and r.  ONE path for every row -- a row
  // outside the image is computed from the clamped row and then overridden -- because the compiler answers a second path through
  // the loads with a full drain of the loads in flight.
  const int lhx = edge ? lh : lc;
  auto iter = [&](int r, f32x4& vc, f32x4& vh, const f32x4& mA, const unsigned (&cA)[4], const f32x4& mB, const unsigned (&cB)[4], f32x4& mC,
                  unsigned (&cC)[4]) {
    f32x4 a1, ah;
#pragma unroll
    for (int j = 0; j < 4; ++j) a1[j] = act((vc[j] - mu[j]) * rs[j], alpha);
#pragma unroll
    for (int j = 0; j < 4; ++j) ah[j] = act((vh[j] - mu[j]) * rs[j], alpha);
    load_row(r + 3, vc, vh);
    f32x4* cell = cells + (r & 1) * lslot;
    cell[lc] = a1;
    cell[lhx] = ah;                                              // (inner threads: their own cell again, the same value)
    lds_barrier();
    const f32x4 a0 = cell[lc - CQ], a2 = cell[lc + CQ];
    const bool rv = r >= 0 && r < S;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mm = fmaxf(fmaxf(a0[j], a1[j]), a2[j]);
      mC[j] = rv ? mm : NEG;
      cC[j] = (vl && a0[j] == mm) ? 0u : (a1[j] == mm ? 1u : 2u);
    }
    const int y = r - 1;
    f32x4 best;
    unsigned code[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float bb = fmaxf(fmaxf(mA[j], mB[j]), mC[j]);
      best[j] = bb;
      const unsigned kB = 3u + cB[j], kC = 6u + cC[j];          // (both formed before the selects: keeps them selects, not branches)
      unsigned k = mB[j] == bb ? kB : kC;
      k = mA[j] == bb ? cA[j] : k;
      code[j] = k;
    }
    if (y >= y0 && live) {
      *reinterpret_cast<f32x4*>(oimg + (size_t)y * orow + oo) = best;
      if (idx) *reinterpret_cast<unsigned*>(iimg + (size_t)y * SW * C + (oc >> 2)) = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
    }
  };
  f32x4 m0 = {NEG, NEG, NEG, NEG}, m1 = m0, m2 = m0;
  unsigned c0[4] = {0u, 0u, 0u, 0u}, c1[4] = {0u, 0u, 0u, 0u}, c2[4] = {0u, 0u, 0u, 0u};
  f32x4 v0c, v0h = {0.f, 0.f, 0.f, 0.f}, v1c, v1h = v0h, v2c, v2h = v0h;
  load_row(y0 - 1, v0c, v0h);
  load_row(y0, v1c, v1h);
  load_row(y0 + 1, v2c, v2h);
  for (int r = y0 - 1; r <= y1; r += 3) {                       // rows y0-1 .. y1; row r's maxima live in m[(r - y0 + 1) % 3]
    iter(r, v0c, v0h, m1, c1, m2, c2, m0, c0);
    if (r + 1 > y1) break;
    iter(r + 1, v1c, v1h, m2, c2, m0, c0, m1, c1);
    if (r + 2 > y1) break;
    iter(r + 2, v2c, v2h, m0, c0, m1, c1, m2, c2);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_slide_kernel(const float* __restrict__ ga, int ld_ga, int coff_ga,
                                                                  const float* __restrict__ z, const unsigned char* __restrict__ idx, int B,
                                                                  int S, int C, const float* __restrict__ mean_rstd, float alpha,
                                                                  float* __restrict__ gxh, float* __restrict__ partial, int nstrips, int rps, int SW, int hp) {
  DRS_CHAIN_PRIO();
  // the exchange cells [2][TX + 2][CQ] x (gradient f32x4, position word), then (after the rows) the reduction image [TX][C][2]
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int CQ = C >> 2;
  const int TX = blockDim.x / CQ;                                            // TX >= 2
  const int tx = threadIdx.x / CQ, cq = threadIdx.x - tx * CQ;
  const int xr = blockIdx.x * TX + tx;
  const bool live = xr < S;
  const int x = live ? xr : S - 1;
  const int b = blockIdx.y / nstrips, strip = blockIdx.y - b * nstrips;
  const int y0 = strip * rps;
  int y1 = y0 + rps;
  y1 = y1 < S ? y1 : S;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const f32x4 mr0 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8);
    const f32x4 mr1 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8 + 4);
    const float mu[4] = {mr0[0], mr0[2], mr1[0], mr1[2]};
    const float rs[4] = {mr0[1], mr0[3], mr1[1], mr1[3]};
    const bool edge_l = tx == 0, edge = edge_l || tx == TX - 1;
    const int xh = edge_l ? x - 1 : x + 1;
    const bool vh_ok = xh >= 0 && xh < S;                                    // the column beyond the workgroup's edge exists
    const int xhc = xh < 0 ? 0 : (xh >= S ? S - 1 : xh);
    const unsigned oi = (unsigned)(x * C + cq * 4), oih = edge ? (unsigned)(xhc * C + cq * 4) : oi;      // bytes into a row of positions
    const unsigned og = (unsigned)((x * ld_ga + coff_ga + cq * 4) * 4), ogh = edge ? (unsigned)((xhc * ld_ga + coff_ga + cq * 4) * 4) : og;
    const size_t irow = (size_t)SW * C, grow = (size_t)SW * ld_ga * 4, zrow = (size_t)SW * C * 4;
    const unsigned char* iimg = idx + (size_t)b * S * irow;
    const char* gimg = reinterpret_cast<const char*>(ga) + (size_t)b * S * grow;
    const char* zimg = reinterpret_cast<const char*>(z) + (size_t)b * S * zrow;
    char* ximg = reinterpret_cast<char*>(gxh) + (size_t)b * S * zrow;
    f32x4* gcells = reinterpret_cast<f32x4*>(red);
    const int lslot = (TX + 2) * CQ;
    unsigned* icells = reinterpret_cast<unsigned*>(red) + 2 * lslot * 4;
    const int lc = (tx + 1) * CQ + cq, lh = (edge_l ? 0 : TX + 1) * CQ + cq;
    // the loads of window row r (own column; the column beyond the edge) and z of output row r - 1
    auto load_row = [&](int r, unsigned& iw, f32x4& gw, unsigned& ih, f32x4& gh, f32x4& zv) {
      const int rc = r < 0 ? 0 : (r >= S ? S - 1 : r);
      iw = *reinterpret_cast<const unsigned*>(iimg + (size_t)rc * irow + oi);
      gw = *reinterpret_cast<const f32x4*>(gimg + (size_t)rc * grow + og);
      ih = *reinterpret_cast<const unsigned*>(iimg + (size_t)rc * irow + oih);     // (every thread, as in the forward kernel)
      gh = *reinterpret_cast<const f32x4*>(gimg + (size_t)rc * grow + ogh);
      const int ry = r - 1 < 0 ? 0 : (r - 1 >= S ? S - 1 : r - 1);
      zv = *reinterpret_cast<const f32x4*>(zimg + (size_t)ry * zrow + (oi << 2));
    };
    // window row r arrives: pooled output (r, x + k - 1) with position c = 3 cy + cx routed its gradient to input
    // (r + cy - 1, x + k - 1 + cx - 1) -- to this thread's column iff cx == 2 - k, and then to row r - 1 / r / r + 1 for cy = 0 / 1 / 2,
    // whose sums are aP / aC / aN.  After row r the sum of row r - 1 is complete: gradient wrt the normalised activation, its sums.
    const int lhx = edge ? lh : lc;
    auto iter = [&](int r, unsigned& iw, f32x4& gw, unsigned& ih, f32x4& gh, f32x4& zv, f32x4& aP, f32x4& aC, f32x4& aN) {
      const f32x4 zy = zv;
      const bool rv = r >= 0 && r < S;                                       // (one path for every row, as in the forward kernel)
      const unsigned mi = (rv && live) ? iw : 0xffffffffu, mh = (rv && vh_ok) ? ih : 0xffffffffu;
      const f32x4 mg = gw, mgh = gh;
      load_row(r + 3, iw, gw, ih, gh, zv);
      f32x4* gcell = gcells + (r & 1) * lslot;
      unsigned* icell = icells + (r & 1) * lslot;
      gcell[lc] = mg;
      icell[lc] = mi;
      gcell[lhx] = edge ? mgh : mg;                                          // (inner threads: their own cell again, the same values)
      icell[lhx] = edge ? mh : mi;
      lds_barrier();
      const unsigned wi[3] = {icell[lc - CQ], mi, icell[lc + CQ]};
      const f32x4 wg[3] = {gcell[lc - CQ], mg, gcell[lc + CQ]};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const unsigned wP = (unsigned)(2 - k), wC = (unsigned)(3 + 2 - k), wN = (unsigned)(6 + 2 - k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned c = (wi[k] >> (8 * j)) & 0xffu;
          aP[j] += c == wP ? wg[k][j] : 0.f;
          aC[j] += c == wC ? wg[k][j] : 0.f;
          aN[j] += c == wN ? wg[k][j] : 0.f;
        }
      }
      const int y = r - 1;
      f32x4 o;
      float t1[4], t2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xhat = (zy[j] - mu[j]) * rs[j];
        const float gx = xhat > 0.f ? aP[j] : aP[j] * alpha;
        o[j] = gx;
        t1[j] = gx;
        t2[j] = gx * xhat;
      }
      if (y >= y0 && live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[j] += t1[j]; s2[j] += t2[j]; }
        *reinterpret_cast<f32x4*>(ximg + (size_t)y * zrow + (oi << 2)) = o;
      }
      aP = f32x4{0.f, 0.f, 0.f, 0.f};                                        // becomes the sum of row r + 2
    };
    unsigned i0, i1, i2, h0 = 0u, h1 = 0u, h2 = 0u;
    f32x4 g0, g1, g2, q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0, z0, z1, z2;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
    load_row(y0 - 1, i0, g0, h0, q0, z0);
    load_row(y0, i1, g1, h1, q1, z1);
    load_row(y0 + 1, i2, g2, h2, q2, z2);
    for (int r = y0 - 1; r <= y1; r += 3) {
      iter(r, i0, g0, h0, q0, z0, a0, a1, a2);
      if (r + 1 > y1) break;
      iter(r + 1, i1, g1, h1, q1, z1, a1, a2, a0);
      if (r + 2 > y1) break;
      iter(r + 2, i2, g2, h2, q2, z2, a2, a0, a1);
    }
  }
  __syncthreads();                                                           // the exchange cells are done with
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[((size_t)tx * C + cq * 4 + j) * 2] = s1[j];
    red[((size_t)tx * C + cq * 4 + j) * 2 + 1] = s2[j];
  }
  __syncthreads();
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float u1 = 0.f, u2 = 0.f;
    for (int r = 0; r < TX; ++r) { u1 += red[((size_t)r * C + c) * 2]; u2 += red[((size_t)r * C + c) * 2 + 1]; }
    partial[(blk * C + c) * 2] = u1;
    partial[(blk * C + c) * 2 + 1] = u2;
  }
}

// ------------------------------------------------------------------------------------------------ average pool (k x k, stride 1, SAME)
// tf.nn.avg_pool(..., 'SAME') of the `_avgpool` net variant (isprs:818-854): the divisor counts only the pixels inside
// the image.  Forward: in [B*S*S][C] -> interior of a haloed view; backward: every output hands g/count to each pixel of
// its window, gathered here per input pixel (odd k: p is in q's window iff q is in p's).
__device__ __forceinline__ int win_count(int c, int S, int r) {       // valid positions of a (2r+1)-window centred at c
  const int lo = c - r < 0 ? 0 : c - r, hi = c + r > S - 1 ? S - 1 : c + r;
  return hi - lo + 1;
}

template <bool BWD>
__global__ void avg_pool_kernel(const float* __restrict__ in, int ld_in, int coff_in, int B, int S, int C, int k, ActView out) {
  const int CQ = C >> 2;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * CQ) return;
  const int x = e / CQ, cq = e - x * CQ;
  const int b = blockIdx.y / S, y = blockIdx.y - b * S;
  const int r = k >> 1;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int dy = -r; dy <= r; ++dy) {
    const int qy = y + dy;
    if (qy < 0 || qy >= S) continue;
    for (int dx = -r; dx <= r; ++dx) {
      const int qx = x + dx;
      if (qx < 0 || qx >= S) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((size_t)b * S + qy) * S + qx) * ld_in + coff_in + cq * 4);
      if (BWD) {
        const float w = 1.0f / (float)(win_count(qy, S, r) * win_count(qx, S, r));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j] * w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      }
    }
  }
  if (!BWD) {
    const float w = 1.0f / (float)(win_count(y, S, r) * win_count(x, S, r));
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] *= w;
  }
  const int Sp = S + 2 * out.P;
  *reinterpret_cast<f32x4*>(out.base + ((size_t)(b * Sp + y + out.P) * Sp + x + out.P) * out.ld + out.coff + cq * 4) = acc;
}

// ------------------------------------------------------------------------------------------------ squeeze-and-excitation
// _squeeze_excitation_layer (isprs:682-697) of the `_SE` variant: s = mean_hw(x); e1 = relu(s W1 + b1); e2 = sigmoid(e1 W2 + b2);
// y = x * e2 (per image and channel).  The spatial reductions run in a fixed order (4 pixel lanes per channel, then lanes in
// order); the two tiny fully connected layers run one workgroup per image.
// out2 == nullptr: s[b][c] = scale * sum_p a[p][c];   else: s[b][c] = scale * sum_p a[p][c] * out2... (see callers)
template <bool DOT>
__global__ void se_spatial_reduce_kernel(const float* __restrict__ a, int ld_a, int coff_a, const float* __restrict__ b2, int S, int C,
                                         float scale, float* __restrict__ out) {
  __shared__ float sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int b = blockIdx.y;
  const int npix = S * S;
  float acc = 0.f;
  if (c < C)
    for (int p = ty; p < npix; p += 4) {
      const size_t q = (size_t)b * npix + p;
      const float v = a[q * ld_a + coff_a + c];
      acc += DOT ? v * b2[q * C + c] : v;
    }
  sh[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < C) out[(size_t)b * C + c] = (((sh[0][tx] + sh[1][tx]) + sh[2][tx]) + sh[3][tx]) * scale;
}

__global__ void se_excite_kernel(const float* __restrict__ s, const float* __restrict__ w1, const float* __restrict__ b1,
                                 const float* __restrict__ w2, const float* __restrict__ b2, int C, int R, float* __restrict__ e1,
                                 float* __restrict__ e2) {
  extern __shared__ float sm[];           // s[C] then e1[R]
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sm[c] = s[(size_t)b * C + c];
  __syncthreads();
  for (int j = threadIdx.x; j < R; j += blockDim.x) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += sm[c] * w1[(size_t)c * R + j];
    a = fmaxf(a + b1[j], 0.f);
    sm[C + j] = a;
    e1[(size_t)b * R + j] = a;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int j = 0; j < R; ++j) a += sm[C + j] * w2[(size_t)j * C + c];
    e2[(size_t)b * C + c] = 1.0f / (1.0f + expf(-(a + b2[c])));
  }
}

// y = x * e2 into the interior of a view (FWD), or gx = gy * e2 + gs (BWD, plain [M][C] output)
template <bool BWD>
__global__ void se_scale_kernel(const float* __restrict__ x, int ld_x, int coff_x, const float* __restrict__ e2,
                                const float* __restrict__ gs, int B, int S, int C, ActView out) {
  const int CQ = C >> 2;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * CQ) return;
  const int xx = e / CQ, cq = e - xx * CQ;
  const int b = blockIdx.y / S, y = blockIdx.y - b * S;
  const size_t p = ((size_t)b * S + y) * S + xx;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ld_x + coff_x + cq * 4);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(e2 + (size_t)b * C + cq * 4);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = v[j] * sc[j];
  if (BWD) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gs + (size_t)b * C + cq * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += g[j];
  }
  const int Sp = S + 2 * out.P;
  *reinterpret_cast<f32x4*>(out.base + ((size_t)(b * Sp + y + out.P) * Sp + xx + out.P) * out.ld + out.coff + cq * 4) = o;
}

// per image: back through sigmoid, FC2, ReLU, FC1; gs already carries the 1/(S*S) of the spatial mean
__global__ void se_excite_bwd_kernel(const float* __restrict__ ge2, const float* __restrict__ e1, const float* __restrict__ e2,
                                     const float* __restrict__ w1, const float* __restrict__ w2, int C, int R, float inv_hw,
                                     float* __restrict__ gpre2, float* __restrict__ gpre1, float* __restrict__ gs) {
  extern __shared__ float sm[];           // gpre2[C] then gpre1[R]
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float ev = e2[(size_t)b * C + c];
    const float g = ge2[(size_t)b * C + c] * ev * (1.0f - ev);
    sm[c] = g;
    gpre2[(size_t)b * C + c] = g;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < R; j += blockDim.x) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += sm[c] * w2[(size_t)j * C + c];
    a = e1[(size_t)b * R + j] > 0.f ? a : 0.f;
    sm[C + j] = a;
    gpre1[(size_t)b * R + j] = a;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int j = 0; j < R; ++j) a += sm[C + j] * w1[(size_t)c * R + j];
    gs[(size_t)b * C + c] = a * inv_hw;
  }
}

// dW[i][j] = sum_b u[b][i] * v[b][j] (fixed order over b); i == rows -> the bias row: sum_b v[b][j]
__global__ void se_fc_wgrad_kernel(const float* __restrict__ u, const float* __restrict__ v, int B, int rows, int cols,
                                   float* __restrict__ dw, float* __restrict__ db) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (rows + 1) * cols) return;
  const int i = e / cols, j = e - i * cols;
  float a = 0.f;
  if (i < rows) {
    for (int b = 0; b < B; ++b) a += u[(size_t)b * rows + i] * v[(size_t)b * cols + j];
    dw[e] = a;
  } else {
    for (int b = 0; b < B; ++b) a += v[(size_t)b * cols + j];
    db[j] = a;
  }
}

// pass B: g_z = rstd * (g_xhat - mean(g_xhat) - xhat * mean(g_xhat * xhat)), written into a zero-haloed view
__global__ void bn_bwd_apply_kernel(const float* __restrict__ gxh, const float* __restrict__ z, int B, int S, int C,
                                    const float* __restrict__ mean_rstd, const double* __restrict__ sums, double count,
                                    ActView out, int hp) {
  DRS_CHAIN_PRIO();
  // per-channel coefficients once per workgroup (they used to be 8 fp64 divisions and 96 bytes of loads per THREAD, for 48 bytes of
  // payload: the kernel was bound by that arithmetic, not by memory): [mu | rstd | sum g / N | sum g xhat / N][C], so that a
  // thread's four channels are one 16-byte LDS read per coefficient.  Same expressions, same rounding as before.
  extern __shared__ __attribute__((aligned(16))) float coef[];               // [4][C]
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    coef[c] = mean_rstd[2 * c];
    coef[C + c] = mean_rstd[2 * c + 1];
    coef[2 * C + c] = (float)(sums[2 * c] / count);
    coef[3 * C + c] = (float)(sums[2 * c + 1] / count);
  }
  __syncthreads();
  const int CQ = C >> 2;
  const int Sp = S + 2 * out.P;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Sp * CQ) return;
  const int xx = e / CQ, cq = e - xx * CQ;
  const int b = blockIdx.y / Sp, yy = blockIdx.y - b * Sp;
  const size_t dst = ((size_t)(b * Sp + yy) * Sp + xx) * out.ld + out.coff + cq * 4;
  const int y = yy - out.P, x = xx - out.P;
  if (y < 0 || y >= S || x < 0 || x >= S) {
    view_store4(out, dst, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  const size_t pix = ((size_t)b * S + y) * S + x;
  const f32x4 gv = *reinterpret_cast<const f32x4*>(gxh + pix * C + cq * 4);
  const f32x4 zv = *reinterpret_cast<const f32x4*>(z + pix * C + cq * 4);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(&coef[cq * 4]);
  const f32x4 rs = *reinterpret_cast<const f32x4*>(&coef[C + cq * 4]);
  const f32x4 m1 = *reinterpret_cast<const f32x4*>(&coef[2 * C + cq * 4]);
  const f32x4 m2 = *reinterpret_cast<const f32x4*>(&coef[3 * C + cq * 4]);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xh = (zv[j] - mu[j]) * rs[j];
    o[j] = rs[j] * (gv[j] - m1[j] - xh * m2[j]);
  }
  view_store4(out, dst, o);
}

// The same with the two means handed in as fp32 (drs_stats_reduce_means works them out where it finishes the sums: single rank, the
// sums need no all-reduce): no fp64 division, no LDS and no barrier in the 204 800 one-row workgroups of a 128-patch launch -- a
// thread's coefficient loads (64 bytes out of L1) go out together with its payload loads.  Same expressions on the same fp32
// values: the same bits.
__global__ void bn_bwd_apply_means_kernel(const float* __restrict__ gxh, const float* __restrict__ z, int B, int S, int C,
                                          const float* __restrict__ mean_rstd, const float* __restrict__ means, ActView out, int hp) {
  DRS_CHAIN_PRIO();
  const int CQ = C >> 2;
  const int Sp = S + 2 * out.P;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Sp * CQ) return;
  const int xx = e / CQ, cq = e - xx * CQ;
  const int b = blockIdx.y / Sp, yy = blockIdx.y - b * Sp;
  const size_t dst = ((size_t)(b * Sp + yy) * Sp + xx) * out.ld + out.coff + cq * 4;
  const int y = yy - out.P, x = xx - out.P;
  if (y < 0 || y >= S || x < 0 || x >= S) {
    view_store4(out, dst, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  const size_t pix = ((size_t)b * S + y) * S + x;
  const f32x4 gv = *reinterpret_cast<const f32x4*>(gxh + pix * C + cq * 4);
  const f32x4 zv = *reinterpret_cast<const f32x4*>(z + pix * C + cq * 4);
  const f32x4 mr0 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8), mr1 = *reinterpret_cast<const f32x4*>(mean_rstd + cq * 8 + 4);
  const f32x4 mm0 = *reinterpret_cast<const f32x4*>(means + cq * 8), mm1 = *reinterpret_cast<const f32x4*>(means + cq * 8 + 4);
  const float mu[4] = {mr0[0], mr0[2], mr1[0], mr1[2]}, rs[4] = {mr0[1], mr0[3], mr1[1], mr1[3]};
  const float m1[4] = {mm0[0], mm0[2], mm1[0], mm1[2]}, m2[4] = {mm0[1], mm0[3], mm1[1], mm1[3]};
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xh = (zv[j] - mu[j]) * rs[j];
    o[j] = rs[j] * (gv[j] - m1[j] - xh * m2[j]);
  }
  view_store4(out, dst, o);
}

// ------------------------------------------------------------------------------------------------ classifier + loss
// One wave walks pixels; lane l owns channels l, l+64, ... of the feature vector.  K <= 8 classes.
struct ClsArgs {
  ActView feat; int C, K, M;
  const float* w;            // [C][K]
  const float* bias;         // [K]
  const unsigned char* labels;   // [M] or null (inference)
  const unsigned char* loss_mask;  // [M] or null: pixels that enter the loss (contest void mask)
  const unsigned char* acc_mask;   // [M] or null: pixels that enter the confusion matrix (augmentation validity)
  float inv_n;               // 1 / (number of pixels the loss averages over, all ranks)
  float* logits;             // [M][K] or null
  unsigned char* pred;       // [M] or null
  float* gfeat; int ld_g, coff_g;   // gradient wrt features [M][ld_g] or null
  float* dw_partial;         // [nblk][C][K] or null
  float* db_partial;         // [nblk][K]
  double* loss_partial;      // [nblk]
  unsigned int* conf;        // [K][K] counts (integer atomics) or null
  int rows_per_block;
  int dma_span, nrows;       // classifier_dma_kernel: slab rows' worth of pixels per workgroup; slab rows in all
  float rcpS, rcpSS;
};

// CI = C / 64; KM = class slots carried per lane: the exact class count for the reference's 2 / 6 / 7 classes (no masked
// slots, no wasted multiplies), 8 otherwise
template <int CI, int KM>
__global__ __launch_bounds__(256) void classifier_loss_kernel(const ClsArgs a) {
  __shared__ float red[4][CI * 64 * KM];
  __shared__ float redb[4][KM];
  __shared__ double redl[4];
  __shared__ unsigned int confs[KM * KM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = KM < 8 ? KM : a.K;      // exact instantiations know their class count at compile time
  // a lane owns its CI channels in groups of V consecutive ones, so that a pixel's features move as 16- / 8-byte accesses
  // (1 KiB / 512 B per wave-instruction instead of 256 B): channel of slot i = (i / V) * 64 V + lane * V + i % V
  constexpr int V = CI % 4 == 0 ? 4 : (CI % 2 == 0 ? 2 : 1);
  typedef float fvec __attribute__((ext_vector_type(V)));
  auto chan = [&](int i) { return (i / V) * 64 * V + lane * V + (i % V); };
  if (threadIdx.x < KM * KM) confs[threadIdx.x] = 0u;
  float wr[CI][KM], dw[CI][KM];
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      wr[i][k] = k < K ? a.w[(size_t)chan(i) * K + k] : 0.f;
      dw[i][k] = 0.f;
    }
  float bk[KM], db[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) { bk[k] = k < K ? a.bias[k] : 0.f; db[k] = 0.f; }
  double lsum = 0.0;
  __syncthreads();
  const int p0 = blockIdx.x * a.rows_per_block;
  int pend = p0 + a.rows_per_block;
  pend = pend < a.M ? pend : a.M;
  // a wave takes PIX consecutive pixels per iteration and issues all their feature loads before the arithmetic of the first:
  // the kernel is bound by bytes in flight (one pixel = 4 x 256 B per wave), not by its arithmetic
  constexpr int PIX = 4;
  auto one_pixel = [&](const int p, const float (&f)[CI]) {
    // the KM = 8 per-lane partial dot products are summed over the 64 lanes by a halving butterfly: at distance 32 / 16 / 8 a lane
    // hands its partner the half of its values the partner keeps (4 + 2 + 1 exchanges), then three plain steps finish the one
    // value left (class 4*b5 + 2*b4 + b3 of the lane id); 10 cross-lane moves per pixel instead of 48, summation order fixed
    float v8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = 0.f;
      if (k < KM) {
#pragma unroll
        for (int i = 0; i < CI; ++i) s += f[i] * wr[i][k];
      }
      v8[k] = s;
    }
    float v4[4], v2[2], v1;
    {
      const bool hi = lane & 32;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float give = hi ? v8[j] : v8[4 + j], keep = hi ? v8[4 + j] : v8[j];
        v4[j] = keep + __shfl_xor(give, 32);
      }
    }
    {
      const bool hi = lane & 16;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float give = hi ? v4[j] : v4[2 + j], keep = hi ? v4[2 + j] : v4[j];
        v2[j] = keep + __shfl_xor(give, 16);
      }
    }
    {
      const bool hi = lane & 8;
      const float give = hi ? v2[0] : v2[1], keep = hi ? v2[1] : v2[0];
      v1 = keep + __shfl_xor(give, 8);
    }
    v1 += __shfl_xor(v1, 4);
    v1 += __shfl_xor(v1, 2);
    v1 += __shfl_xor(v1, 1);
    float lg[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) lg[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v1), 8 * k)) + bk[k];     // class k's total lives in lanes 8k .. 8k+7
    // arg-max (first maximum) and softmax over the K live classes; every lane holds the same values
    int am = 0;
    float mx = lg[0];
#pragma unroll
    for (int k = 1; k < KM; ++k)
      if (k < K && lg[k] > mx) { mx = lg[k]; am = k; }
    if (lane == 0) {
      if (a.logits)
        for (int k = 0; k < K; ++k) a.logits[(size_t)p * K + k] = lg[k];
      if (a.pred) a.pred[p] = (unsigned char)am;
    }
    if (!a.labels) return;
    const int y = a.labels[p];
    if (lane == 0 && a.conf && (!a.acc_mask || a.acc_mask[p]) && y < K) atomicAdd(&confs[y * KM + am], 1u);
    const bool in_loss = (!a.loss_mask || a.loss_mask[p]) && y < K;      // a label outside [0, K) (TF would raise) never trains the net
    float ex[KM], se = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) { ex[k] = k < K ? __expf(lg[k] - mx) : 0.f; se += ex[k]; }
    const float inv = 1.0f / se;
    float ly = 0.f;
    float dl[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const float pk = ex[k] * inv;
      dl[k] = in_loss && k < K ? (pk - (k == y ? 1.f : 0.f)) * a.inv_n : 0.f;
      if (k == y) ly = lg[k];
    }
    if (in_loss) lsum += (double)(__logf(se) + mx - ly);
    if (a.gfeat) {
#pragma unroll
      for (int gi = 0; gi < CI / V; ++gi) {
        fvec gv;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const int i = gi * V + e;
          float g = 0.f;
#pragma unroll
          for (int k = 0; k < KM; ++k) g += dl[k] * wr[i][k];
          gv[e] = g;
#pragma unroll
          for (int k = 0; k < KM; ++k) dw[i][k] += f[i] * dl[k];
        }
        *reinterpret_cast<fvec*>(a.gfeat + (size_t)p * a.ld_g + a.coff_g + gi * 64 * V + lane * V) = gv;
      }
#pragma unroll
      for (int k = 0; k < KM; ++k) db[k] += dl[k];
    }
    };
  for (int pb = p0 + wave * PIX; pb < pend; pb += 4 * PIX) {
    float f[PIX][CI];
#pragma unroll
    for (int j = 0; j < PIX; ++j) {
      const int p = pb + j < pend ? pb + j : pend - 1;
      const uint32_t off = padded_pixel_off(p, a.feat.S, a.feat.P, a.feat.ld, a.rcpS, a.rcpSS, 0, 0) + a.feat.coff;
#pragma unroll
      for (int g = 0; g < CI / V; ++g) {
        const fvec v = *reinterpret_cast<const fvec*>(a.feat.base + off + g * 64 * V + lane * V);
#pragma unroll
        for (int e = 0; e < V; ++e) f[j][g * V + e] = v[e];
      }
    }
#pragma unroll
    for (int j = 0; j < PIX; ++j)
      if (pb + j < pend) one_pixel(pb + j, f[j]);
  }
  if (!a.labels) return;
  // workgroup reduction in wave order, then one slab row per workgroup
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int k = 0; k < KM; ++k) red[wave][chan(i) * KM + k] = dw[i][k];
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < KM; ++k) redb[wave][k] = db[k];
    redl[wave] = lsum;
  }
  __syncthreads();
  if (a.dw_partial) {
    for (int e = threadIdx.x; e < a.C * K; e += 256) {
      const int c = e / K, k = e - c * K;
      const float s = ((red[0][c * KM + k] + red[1][c * KM + k]) + red[2][c * KM + k]) + red[3][c * KM + k];
      a.dw_partial[(size_t)blockIdx.x * a.C * K + e] = s;
    }
    if (threadIdx.x < K)
      a.db_partial[(size_t)blockIdx.x * K + threadIdx.x] =
          ((redb[0][threadIdx.x] + redb[1][threadIdx.x]) + redb[2][threadIdx.x]) + redb[3][threadIdx.x];
  }
  if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = ((redl[0] + redl[1]) + redl[2]) + redl[3];
  if (a.conf && threadIdx.x < K * K) {
    const int r = threadIdx.x / K, c = threadIdx.x - r * K;
    const unsigned v = confs[r * KM + c];
    if (v) atomicAdd(&a.conf[threadIdx.x], v);
  }
}

// The same classifier block on the matrix cores (isprs:1024-1031 is a true dense contraction: [M x C] . [C x K]): three
// products per tile of 16 pixels, all on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains), the class dimension padded to 16 / 8:
//   logits^T [class][px] = W^T [class][c] . feat^T [c][px]            (k = channel;  C / 4 MFMAs)
//   gfeat^T  [c][px]     = W [c][class]   . dlogits^T [class][px]     (k = class;    C / 8 MFMAs)
//   dW^T     [class][c] += dlogits^T [class][px] . feat [px][c]       (k = pixel;    C / 4 MFMAs)
// The orientations are chosen so that what one product leaves in a lane is what the next one wants there.  The rows of the first
// product are the classes in the order 0, 4, 8, 12, 1, 5, ...: lane l then holds, for pixel l & 15, class (l >> 4) + 4 r in
// accumulator register r -- the real classes (< 8) of a pixel sit in registers 0 and 1 of its four lanes, softmax / arg-max /
// loss need the exchanges with lanes l ^ 16 and l ^ 32, and the logit gradients are already the B operand of the second
// product (k slot l >> 4 <-> class (l >> 4) + 4 s at step s = 0, 1).  Only the third product needs them transposed: through a
// 1-KiB LDS tile per wave.  The features are read twice, in the lane arrangement each product wants -- 16 pixels x 64 B per
// instruction for the first (k = channel on l >> 4), 4 pixels x 256 B for the third (k = pixel on l >> 4) -- the second time
// out of L2.  One workgroup = 4 waves, each walking its own 16-pixel tiles of the workgroup's pixel range; the filter sits in
// LDS in the two operand arrangements.  Sums over pixels (dW, db, the loss) stay per workgroup and are added in wave order.
template <int CQ, bool TRAIN>
__global__ __launch_bounds__(256) void classifier_mfma_kernel(const ClsArgs a) {
  constexpr int C = CQ * 64, NJ = C / 16, KP = 16;
  __shared__ __attribute__((aligned(16))) float W1[C * KP];      // [c / 16][(c % 16) / 4][row i <-> class (i >> 2) + 4 (i & 3)][c % 4]: A operand of the first product
  __shared__ __attribute__((aligned(16))) float W2[C * 8];       // [c][2 G + s <-> class G + 4 s]: A operand of the second
  __shared__ __attribute__((aligned(16))) float DL[4][16 * KP];  // per wave: logit gradients [px][class] of the current tile (classes 8 .. 15 stay zero)
  __shared__ float redb[4][8];
  __shared__ double redl[4];
  __shared__ unsigned int confs[64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int p = lane & 15, G = lane >> 4;
  const int K = a.K;
  for (int i = t; i < C * KP; i += 256) {
    const int c = i >> 4, row = i & 15;
    const int cls = (row >> 2) + 4 * (row & 3);
    W1[(((c >> 4) * 4 + ((c & 15) >> 2)) * KP + row) * 4 + (c & 3)] = cls < K ? a.w[(size_t)c * K + cls] : 0.f;
  }
  for (int i = t; i < C * 8; i += 256) {
    const int c = i >> 3, slot = i & 7;
    const int cls = (slot >> 1) + 4 * (slot & 1);
    W2[i] = cls < K ? a.w[(size_t)c * K + cls] : 0.f;
  }
  for (int i = t; i < 4 * 16 * KP; i += 256) (&DL[0][0])[i] = 0.f;
  if (t < 64) confs[t] = 0u;
  const int cls0 = G, cls1 = G + 4;                 // this lane's two real class slots
  const float bk0 = cls0 < K ? a.bias[cls0] : 0.f, bk1 = cls1 < K ? a.bias[cls1] : 0.f;
  __syncthreads();

  f32x4 dw[CQ][4];
#pragma unroll
  for (int q = 0; q < CQ; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) dw[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db0 = 0.f, db1 = 0.f;
  double lsum = 0.0;

  const int p0 = blockIdx.x * a.rows_per_block;
  int pend = p0 + a.rows_per_block;
  pend = pend < a.M ? pend : a.M;
  const float* fb = a.feat.base + a.feat.coff;
  for (int tb = p0 + 16 * wave; tb < pend; tb += 64) {
    const int pix = tb + p;
    const bool valid = pix < pend;
    const int pixc = valid ? pix : pend - 1;
    const uint32_t off1 = padded_pixel_off(pixc, a.feat.S, a.feat.P, a.feat.ld, a.rcpS, a.rcpSS, 0, 0);
    // ---- logits: k = channel 16 jj + 4 G + e
    // (two accumulation chains, even / odd 16-channel groups: a single chain of this MFMA is paced by its 40-cycle dependent
    // latency, not its 32-cycle issue rate)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    constexpr int JC = NJ < 8 ? NJ : 8;          // feature loads in flight per lane (NJ is a multiple of 4)
#pragma unroll
    for (int j0 = 0; j0 < NJ; j0 += JC) {
      f32x4 fr[JC];
#pragma unroll
      for (int j = 0; j < JC; ++j)
        if (j0 + j < NJ) fr[j] = *reinterpret_cast<const f32x4*>(fb + off1 + 16 * (j0 + j) + 4 * G);
#pragma unroll
      for (int j = 0; j < JC; ++j) {
        if (j0 + j >= NJ) continue;
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&W1[(((j0 + j) * 4 + G) * KP + p) * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if ((j0 + j) & 1) acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], fr[j][e], acc_b, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], fr[j][e], acc, 0, 0, 0);
        }
      }
    }
    acc[0] += acc_b[0];
    acc[1] += acc_b[1];
    // this lane: pixel p, classes G (register 0) and G + 4 (register 1); registers 2, 3 are padding
    const float lg0 = acc[0] + bk0, lg1 = acc[1] + bk1;
    float mv = -INFINITY;
    int mc = 0;
    if (cls0 < K) { mv = lg0; mc = cls0; }
    if (cls1 < K && lg1 > mv) { mv = lg1; mc = cls1; }
    // maximum and its FIRST class over the four lanes of the pixel: the larger value, the lower class on a tie
#pragma unroll
    for (int d = 16; d <= 32; d <<= 1) {
      const float ov = __shfl_xor(mv, d);
      const int oc = __shfl_xor(mc, d);
      if (ov > mv || (ov == mv && oc < mc)) { mv = ov; mc = oc; }
    }
    const float mx = mv;
    const int am = mc;
    if (valid) {
      if (a.logits) {
        if (cls0 < K) a.logits[(size_t)pix * K + cls0] = lg0;
        if (cls1 < K) a.logits[(size_t)pix * K + cls1] = lg1;
      }
      if (a.pred && G == 0) a.pred[pix] = (unsigned char)am;
    }
    if (!TRAIN) continue;
    const int y = a.labels[pixc];
    if (G == 0 && valid && a.conf && (!a.acc_mask || a.acc_mask[pix]) && y < K) atomicAdd(&confs[y * 8 + am], 1u);
    const bool in_loss = valid && (!a.loss_mask || a.loss_mask[pixc]) && y < K;     // a label outside [0, K) never trains the net
    const float ex0 = cls0 < K ? __expf(lg0 - mx) : 0.f, ex1 = cls1 < K ? __expf(lg1 - mx) : 0.f;
    float se = ex0 + ex1;
    se += __shfl_xor(se, 16);                       // (a + b == b + a bit for bit: every lane of the pixel gets the same sum)
    se += __shfl_xor(se, 32);
    const float inv = 1.0f / se;
    const float dl0 = (in_loss && cls0 < K) ? (ex0 * inv - (cls0 == y ? 1.f : 0.f)) * a.inv_n : 0.f;
    const float dl1 = (in_loss && cls1 < K) ? (ex1 * inv - (cls1 == y ? 1.f : 0.f)) * a.inv_n : 0.f;
    if (in_loss && cls0 == y) lsum += (double)(__logf(se) + mx - lg0);
    if (in_loss && cls1 == y) lsum += (double)(__logf(se) + mx - lg1);
    db0 += dl0;
    db1 += dl1;
    if (!a.gfeat) continue;
    // ---- gradient wrt the features: k slot G <-> class G + 4 s (the lane's own registers), rows = 16 channels per product
    float* gdst = a.gfeat + (size_t)pixc * a.ld_g + a.coff_g + 4 * G;
#pragma unroll
    for (int tt = 0; tt < NJ; ++tt) {
      const float2 wv = *reinterpret_cast<const float2*>(&W2[(16 * tt + p) * 8 + 2 * G]);
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      g = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, dl0, g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, dl1, g, 0, 0, 0);
      if (valid) *reinterpret_cast<f32x4*>(gdst + 16 * tt) = g;
    }
    // ---- filter gradient: k = pixel; the logit gradients transposed through this wave's LDS tile
    DL[wave][p * KP + cls0] = dl0;
    DL[wave][p * KP + cls1] = dl1;
    __builtin_amdgcn_wave_barrier();
    float aop[4];
    uint32_t off2[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      aop[t4] = DL[wave][(4 * t4 + G) * KP + p];          // class p of pixel 4 t4 + G
      off2[t4] = __shfl(off1, 4 * t4 + G);                // that pixel's (clamped) feature row
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      f32x4 fv[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) fv[q] = *reinterpret_cast<const f32x4*>(fb + off2[t4] + 64 * q + 4 * p);
#pragma unroll
      for (int q = 0; q < CQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[q][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[t4], fv[q][e], dw[q][e], 0, 0, 0);
    }
  }
  if (!TRAIN) return;
  // ---- workgroup sums in wave order, one slab row per workgroup.  dw[q][e][r] = dW[c = 64 q + 4 p + e][class 4 G + r]
  __syncthreads();                       // every wave is done with W1: it becomes the [C][8] accumulator
  float* red = W1;
  for (int w = 0; w < 4; ++w) {
    if (wave == w && G < 2) {
#pragma unroll
      for (int q = 0; q < CQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = (64 * q + 4 * p + e) * 8 + 4 * G + r;
            red[idx] = (w ? red[idx] : 0.f) + dw[q][e][r];
          }
    }
    __syncthreads();
  }
  // db: over the 16 pixels of a lane row (fixed butterfly), then over the waves
  {
    float v0 = db0, v1 = db1;
    for (int d = 1; d < 16; d <<= 1) { v0 += __shfl_xor(v0, d); v1 += __shfl_xor(v1, d); }
    if (p == 0) { redb[wave][cls0] = v0; redb[wave][cls1] = v1; }
  }
  {
    double v = lsum;
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    if (lane == 0) redl[wave] = v;
  }
  __syncthreads();
  if (a.dw_partial) {
    for (int e = t; e < C * K; e += 256) {
      const int c = e / K, k = e - c * K;
      a.dw_partial[(size_t)blockIdx.x * C * K + e] = red[c * 8 + k];
    }
    if (t < K) a.db_partial[(size_t)blockIdx.x * K + t] = ((redb[0][t] + redb[1][t]) + redb[2][t]) + redb[3][t];
  }
  if (t == 0) a.loss_partial[blockIdx.x] = ((redl[0] + redl[1]) + redl[2]) + redl[3];
  if (a.conf && t < K * K) {
    const int r = t / K, c = t - r * K;
    const unsigned v = confs[r * 8 + c];
    if (v) atomicAdd(&a.conf[t], v);
  }
}

// The MFMA classifier with its features brought in ONCE, by LDS-DMA, one tile ahead: a wave owns two 16-pixel feature tiles in LDS
// (16 x C floats each), the DMA of tile i+1 (global_load_lds_dwordx4, no registers) lands while tile i is multiplied, and both
// feature operands -- k = channel for the logits, k = pixel for the filter gradient -- are ds_read_b128 fragments of that one
// image (16-byte piece c of pixel p sits in slot c ^ (p & 15), realised on the SOURCE address: both read patterns are
// conflict-free).  One workgroup of 4 waves per CU (C = 256: 128 KiB of tiles + 28 KiB of filter images); the latency that
// classifier_mfma_kernel pays per tile at two waves per SIMD (HBM round trip, then four L2 round trips for the second read) is
// hidden behind a whole tile of arithmetic.  Every vector-memory operation inside the loop is counted by hand: the label / mask
// bytes are inline-asm loads waited for with vmcnt(#DMA instructions), the DMA itself with vmcnt(#feature-gradient stores).
// Products, orientations, class permutation and every sum are those of classifier_mfma_kernel: results are bitwise the same.
template <int CQ, bool TRAIN>
__global__ __launch_bounds__(256, 1) void classifier_dma_kernel(const ClsArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = CQ * 64, NJ = C / 16, KP = 16, NI = 4 * CQ;      // NI: 1-KiB DMA instructions per tile
  __shared__ __attribute__((aligned(1024))) float FT[4][2][16 * C];
  __shared__ __attribute__((aligned(16))) float W1[C * KP];
  __shared__ __attribute__((aligned(16))) float W2[C * 8];
  __shared__ __attribute__((aligned(16))) float DL[4][16 * KP];
  __shared__ float redb[4][8];
  __shared__ double redl[4];
  __shared__ unsigned int confs[64];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int p = lane & 15, G = lane >> 4;
  const int K = a.K;
  for (int i = t; i < C * KP; i += 256) {
    const int c = i >> 4, row = i & 15;
    const int cls = (row >> 2) + 4 * (row & 3);
    W1[(((c >> 4) * 4 + ((c & 15) >> 2)) * KP + row) * 4 + (c & 3)] = cls < K ? a.w[(size_t)c * K + cls] : 0.f;
  }
  for (int i = t; i < C * 8; i += 256) {
    const int c = i >> 3, slot = i & 7;
    const int cls = (slot >> 1) + 4 * (slot & 1);
    W2[i] = cls < K ? a.w[(size_t)c * K + cls] : 0.f;
  }
  for (int i = t; i < 4 * 16 * KP; i += 256) (&DL[0][0])[i] = 0.f;
  if (t < 64) confs[t] = 0u;
  const int cls0 = G, cls1 = G + 4;
  const float bk0 = cls0 < K ? a.bias[cls0] : 0.f, bk1 = cls1 < K ? a.bias[cls1] : 0.f;
  __syncthreads();

  f32x4 dw[CQ][4];
#pragma unroll
  for (int q = 0; q < CQ; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) dw[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db0 = 0.f, db1 = 0.f;
  double lsum = 0.0;

  // this form wants long pixel ranges (one workgroup per CU, a DMA pipeline to fill): the launch has at most 256 workgroups, each
  // covering the pixels of a.dma_span slab rows; it writes its sums to row blockIdx.x and zeroes the rows nobody writes
  const int p0 = blockIdx.x * a.rows_per_block * a.dma_span;
  int pend = p0 + a.rows_per_block * a.dma_span;
  pend = pend < a.M ? pend : a.M;
  const float* fb = a.feat.base + a.feat.coff;
  const unsigned char* lm = a.loss_mask ? a.loss_mask : a.labels;      // (absent masks: any readable bytes keep the load count fixed)
  const unsigned char* am_ = a.acc_mask ? a.acc_mask : a.labels;
  // DMA of the tile that starts at pixel `tb` into buffer `buf`: instruction i moves bytes [1024 i, 1024 (i + 1)) of the image
  auto dma = [&](int tb, int buf) {
    int px = tb + p;
    px = px < pend ? px : pend - 1;
    const uint32_t off = padded_pixel_off(px, a.feat.S, a.feat.P, a.feat.ld, a.rcpS, a.rcpSS, 0, 0);
    float* dst = &FT[wave][buf][0];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int X = 1024 * i + 16 * lane;                 // byte of the image this lane fills
      const int pixel = X / (4 * C), slot = (X % (4 * C)) / 16;
      const uint32_t po = (C == 256) ? (uint32_t)__builtin_amdgcn_readlane((int)off, i)      // one pixel per instruction: scalar
                                     : (uint32_t)__shfl(off, pixel);                       // that pixel's feature row (lane `pixel` computed it)
      __builtin_amdgcn_global_load_lds(fb + po + 4 * (slot ^ pixel), (lds_ptr)(dst + 256 * i), 16, 0, 0);
    }
  };
  int tb = p0 + 16 * wave;
  int cur = 0;
  if (tb < pend) dma(tb, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (; tb < pend; tb += 64, cur ^= 1) {
    const int pix = tb + p;
    const bool valid = pix < pend;
    const int pixc = valid ? pix : pend - 1;
    __builtin_amdgcn_wave_barrier();
    // label / mask bytes of this tile: hidden from the compiler's wait bookkeeping (it would drain the DMA below at their first use)
    unsigned ylab = 0, vlm = 1, vam = 1;
    if (TRAIN) {
      asm volatile("global_load_ubyte %0, %1, off" : "=v"(ylab) : "v"(a.labels + pixc) : "memory");
      asm volatile("global_load_ubyte %0, %1, off" : "=v"(vlm) : "v"(lm + pixc) : "memory");
      asm volatile("global_load_ubyte %0, %1, off" : "=v"(vam) : "v"(am_ + pixc) : "memory");
    }
    if (tb + 64 < pend) dma(tb + 64, cur ^ 1);           // the next tile lands while this one is multiplied
    const float* F = &FT[wave][cur][0];
    // ---- logits: k = channel 16 jj + 4 G + e
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};          // two chains, as in classifier_mfma_kernel
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const f32x4 fr = *reinterpret_cast<const f32x4*>(&F[p * C + 4 * ((4 * jj + G) ^ p)]);
      const f32x4 wv = *reinterpret_cast<const f32x4*>(&W1[((jj * 4 + G) * KP + p) * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (jj & 1) acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], fr[e], acc_b, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], fr[e], acc, 0, 0, 0);
      }
    }
    acc[0] += acc_b[0];
    acc[1] += acc_b[1];
    const float lg0 = acc[0] + bk0, lg1 = acc[1] + bk1;
    float mv = -INFINITY;
    int mc = 0;
    if (cls0 < K) { mv = lg0; mc = cls0; }
    if (cls1 < K && lg1 > mv) { mv = lg1; mc = cls1; }
#pragma unroll
    for (int d = 16; d <= 32; d <<= 1) {
      const float ov = __shfl_xor(mv, d);
      const int oc = __shfl_xor(mc, d);
      if (ov > mv || (ov == mv && oc < mc)) { mv = ov; mc = oc; }
    }
    const float mx = mv;
    const int am = mc;
    if (valid) {
      if (a.logits) {
        if (cls0 < K) a.logits[(size_t)pix * K + cls0] = lg0;
        if (cls1 < K) a.logits[(size_t)pix * K + cls1] = lg1;
      }
      if (a.pred && G == 0) a.pred[pix] = (unsigned char)am;
    }
    if (TRAIN) {
      // the three byte loads are older than the NI DMA instructions (or nothing, on the last tile) issued after them
      if (tb + 64 < pend) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(ylab), "+v"(vlm), "+v"(vam) : "n"(NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ylab), "+v"(vlm), "+v"(vam) :: "memory");
      const int y = (int)ylab;
      if (G == 0 && valid && a.conf && (!a.acc_mask || vam) && y < K) atomicAdd(&confs[y * 8 + am], 1u);
      const bool in_loss = valid && (!a.loss_mask || vlm) && y < K;
      const float ex0 = cls0 < K ? __expf(lg0 - mx) : 0.f, ex1 = cls1 < K ? __expf(lg1 - mx) : 0.f;
      float se = ex0 + ex1;
      se += __shfl_xor(se, 16);
      se += __shfl_xor(se, 32);
      const float inv = 1.0f / se;
      const float dl0 = (in_loss && cls0 < K) ? (ex0 * inv - (cls0 == y ? 1.f : 0.f)) * a.inv_n : 0.f;
      const float dl1 = (in_loss && cls1 < K) ? (ex1 * inv - (cls1 == y ? 1.f : 0.f)) * a.inv_n : 0.f;
      if (in_loss && cls0 == y) lsum += (double)(__logf(se) + mx - lg0);
      if (in_loss && cls1 == y) lsum += (double)(__logf(se) + mx - lg1);
      db0 += dl0;
      db1 += dl1;
      if (a.gfeat) {
        // ---- filter gradient first (k = pixel; the logit gradients transposed through this wave's LDS tile) ...
        DL[wave][p * KP + cls0] = dl0;
        DL[wave][p * KP + cls1] = dl1;
        __builtin_amdgcn_wave_barrier();
        float aop[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) aop[t4] = DL[wave][(4 * t4 + G) * KP + p];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int px = 4 * t4 + G;
#pragma unroll
          for (int q = 0; q < CQ; ++q) {
            const f32x4 fv = *reinterpret_cast<const f32x4*>(&F[px * C + 4 * ((16 * q + p) ^ px)]);
#pragma unroll
            for (int e = 0; e < 4; ++e) dw[q][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[t4], fv[e], dw[q][e], 0, 0, 0);
          }
        }
        // ---- ... then the gradient wrt the features: its NJ stores are the youngest vector-memory operations of the tile
        float* gdst = a.gfeat + (size_t)pixc * a.ld_g + a.coff_g + 4 * G;
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
          const float2 wv = *reinterpret_cast<const float2*>(&W2[(16 * tt + p) * 8 + 2 * G]);
          f32x4 g = {0.f, 0.f, 0.f, 0.f};
          g = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, dl0, g, 0, 0, 0);
          g = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, dl1, g, 0, 0, 0);
          if (valid) *reinterpret_cast<f32x4*>(gdst + 16 * tt) = g;
        }
        // the next tile's DMA is older than those NJ stores (every tile has a valid lane, so all NJ are issued)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NJ) : "memory");
        continue;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (!TRAIN) return;
  __syncthreads();
  float* red = W1;
  for (int w = 0; w < 4; ++w) {
    if (wave == w && G < 2) {
#pragma unroll
      for (int q = 0; q < CQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = (64 * q + 4 * p + e) * 8 + 4 * G + r;
            red[idx] = (w ? red[idx] : 0.f) + dw[q][e][r];
          }
    }
    __syncthreads();
  }
  {
    float v0 = db0, v1 = db1;
    for (int d = 1; d < 16; d <<= 1) { v0 += __shfl_xor(v0, d); v1 += __shfl_xor(v1, d); }
    if (p == 0) { redb[wave][cls0] = v0; redb[wave][cls1] = v1; }
  }
  {
    double v = lsum;
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    if (lane == 0) redl[wave] = v;
  }
  __syncthreads();
  if (a.dw_partial) {
    for (int e = t; e < C * K; e += 256) {
      const int c = e / K, k = e - c * K;
      a.dw_partial[(size_t)blockIdx.x * C * K + e] = red[c * 8 + k];
    }
    if (t < K) a.db_partial[(size_t)blockIdx.x * K + t] = ((redb[0][t] + redb[1][t]) + redb[2][t]) + redb[3][t];
  }
  if (t == 0) a.loss_partial[blockIdx.x] = ((redl[0] + redl[1]) + redl[2]) + redl[3];
  for (int r = gridDim.x + blockIdx.x; r < a.nrows; r += gridDim.x) {        // slab rows beyond the launch: zero (the caller sums a.nrows rows)
    if (a.dw_partial) {
      for (int e = t; e < C * K; e += 256) a.dw_partial[(size_t)r * C * K + e] = 0.f;
      if (t < K) a.db_partial[(size_t)r * K + t] = 0.f;
    }
    if (t == 0) a.loss_partial[r] = 0.0;
  }
  if (a.conf && t < K * K) {
    const int r = t / K, c = t - r * K;
    const unsigned v = confs[r * 8 + c];
    if (v) atomicAdd(&a.conf[t], v);
  }
#endif
}

__global__ void sum_f64_kernel(const double* __restrict__ in, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += in[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

__global__ void scale_f64_kernel(double* __restrict__ x, int n, double s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

// 0.5 * sum(w^2) over [0, n): per-workgroup partials in fp64 (tf.nn.l2_loss, isprs:648)
__global__ void l2_partial_kernel(const float* __restrict__ w, size_t n, double* __restrict__ partial) {
  __shared__ double sh[256];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (double)w[i] * (double)w[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = 0.5 * sh[0];
}

// ApplyMomentum (use_nesterov = False): g = grad*gscale (+ wd*w for the first n_decay entries: the kernels);
// accum = momentum*accum + g; w -= lr*accum.
__global__ void momentum_kernel(float* __restrict__ w, const float* __restrict__ grad, float* __restrict__ accum, size_t n,
                                size_t n_decay, float lr, float wd, float momentum, float gscale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float g = grad[i] * gscale;
    const float wv = w[i];
    if (i < n_decay) g += wd * wv;
    const float a = momentum * accum[i] + g;
    accum[i] = a;
    w[i] = wv - lr * a;
  }
}

// K x K confusion matrix of (label, prediction) under an optional mask; integer atomics (order-independent)
__global__ void confusion_kernel(const unsigned char* __restrict__ labels, const unsigned char* __restrict__ pred,
                                 const unsigned char* __restrict__ mask, size_t n, int K, int ignore_label,
                                 unsigned int* __restrict__ conf) {
  __shared__ unsigned int sh[64];
  if (threadIdx.x < 64) sh[threadIdx.x] = 0u;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (mask && !mask[i]) continue;
    const int y = labels[i], p = pred[i];
    if (y == ignore_label || y >= K || p >= K) continue;
    atomicAdd(&sh[y * 8 + p], 1u);
  }
  __syncthreads();
  if (threadIdx.x < K * K) {
    const int r = threadIdx.x / K, c = threadIdx.x - r * K;
    const unsigned v = sh[r * 8 + c];
    if (v) atomicAdd(&conf[threadIdx.x], v);
  }
}

inline int chain_hp() { return drs_chain_level(drs_tl_chain, drs_g_chain_mode) >= 2 ? 1 : 0; }

inline ActView mkview(float* base, int S, int P, int ld, int coff) {
  ActView v; v.base = base; v.S = S; v.P = P; v.ld = ld; v.coff = coff; v.terms = nullptr; v.nt = 0; return v;
}

}  // namespace

extern "C" {

int drs_colsum_scratch_doubles(int ncols) { return COLSUM_BLOCKS * ncols; }

int drs_stats_reduce(const float* partial, int nrows, int C, double* sums, double* scratch, void* stream) {
  (void)scratch;      // kept in the signature: earlier revisions reduced in two launches through it
  if (!partial || !sums || nrows < 1 || C < 1) return DRS_ERR_ARG;
  DRS_LAUNCH(bn_stats_kernel<false>, dim3((C + STAT_CH - 1) / STAT_CH), dim3(256), 0, (hipStream_t)stream, partial, nrows, C, 0, 1, sums,
             1.0, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0, (float*)nullptr, chain_hp());
  return DRS_LAUNCH_CHECK();
}

int drs_stats_reduce_means(const float* partial, int nrows, int C, double count, double* sums, float* means, void* stream) {
  if (!partial || !means || nrows < 1 || C < 1 || count < 1.0) return DRS_ERR_ARG;
  DRS_LAUNCH(bn_stats_kernel<false>, dim3((C + STAT_CH - 1) / STAT_CH), dim3(256), 0, (hipStream_t)stream, partial, nrows, C, 0, 1, sums,
             count, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0, means, chain_hp());
  return DRS_LAUNCH_CHECK();
}

int drs_conv_stats_reduce(const float* partial, int M, int mtile, int C, double* sums, double* scratch, void* stream) {
  (void)scratch;
  if (!partial || !sums || M < 1 || mtile < 1 || C < 1) return DRS_ERR_ARG;
  DRS_LAUNCH(bn_stats_kernel<true>, dim3((C + STAT_CH - 1) / STAT_CH), dim3(256), 0, (hipStream_t)stream, partial, (M + mtile - 1) / mtile, C,
             M, mtile, sums, 1.0, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0, (float*)nullptr, chain_hp());
  return DRS_LAUNCH_CHECK();
}

int drs_conv_stats_finish(const float* partial, int M, int mtile, int C, double count, float* mean_rstd, float* moving_mean,
                          float* moving_var, double decay, int bessel, double* sums, void* stream) {
  if (!partial || !mean_rstd || M < 1 || mtile < 1 || C < 1 || count < 1.0) return DRS_ERR_ARG;
  DRS_LAUNCH(bn_stats_kernel<true>, dim3((C + STAT_CH - 1) / STAT_CH), dim3(256), 0, (hipStream_t)stream, partial, (M + mtile - 1) / mtile, C,
             M, mtile, sums, count, mean_rstd, moving_mean, moving_var, (float)(1.0 - decay), bessel, (float*)nullptr, chain_hp());
  return DRS_LAUNCH_CHECK();
}

int drs_bn_finish(const double* sums, double count, int C, float* mean_rstd, float* moving_mean, float* moving_var,
                  double decay, int bessel, void* stream) {
  if (!sums || !mean_rstd || count < 1.0) return DRS_ERR_ARG;
  DRS_LAUNCH(bn_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, count, C, mean_rstd,
                     moving_mean, moving_var, (float)(1.0 - decay), bessel);
  return DRS_LAUNCH_CHECK();
}

int drs_bn_eval_coeffs(const float* moving_mean, const float* moving_var, int C, float* mean_rstd, void* stream) {
  if (!moving_mean || !moving_var || !mean_rstd) return DRS_ERR_ARG;
  DRS_LAUNCH(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, moving_mean, moving_var, C,
                     mean_rstd);
  return DRS_LAUNCH_CHECK();
}

static int bn_act_pool_forward_impl(const float* z, int B, int S, int C, const float* mean_rstd, float alpha, int pool,
                                    float* out, int P_out, int ld_out, int coff_out, unsigned char* argmax,
                                    unsigned short* terms, int nterms, void* stream, const BnFinish* fin = nullptr) {
  if (!z || !mean_rstd || (!out && !terms) || C % 4) return DRS_ERR_ARG;
  if (fin && !((pool & 1) && slide_ok(C) && !terms)) return DRS_ERR_ARG;     // the folded finish exists in the pooled sliding kernel only
  if (terms && ((nterms != 2 && nterms != 3) || (ld_out & 31) || (coff_out & 31))) return DRS_ERR_ARG;
  const int Sp = S + 2 * P_out;
  if ((long long)B * Sp > 65535) return DRS_ERR_ARG;
  const int per_row = Sp * (C / 4);
  dim3 grid((per_row + 255) / 256, B * Sp);
  ActView v = mkview(out, S, P_out, ld_out, coff_out);
  v.terms = terms; v.nt = terms ? nterms : 0;
  const bool halo_is_zero = (pool & 2) != 0;     // the caller vouches for the halo (same B, S, P as the call that last zeroed it)
  pool &= 1;
  if (pool && slide_ok(C) && !terms) {             // (the split-bf16 images of the opt-in arithmetic: the gathering form below)
    const SlideCfg c = slide_cfg(B, S, C);
    if (P_out > 0 && !halo_is_zero) DRS_LAUNCH(zero_halo_kernel, grid, dim3(256), 0, (hipStream_t)stream, v, B, C);
    const size_t shm = (size_t)2 * (c.TX + 2) * C * sizeof(float);
    const BnFinish none = {nullptr, 1.0, nullptr, nullptr, 0.f, 0};
    DRS_LAUNCH(bn_act_pool_fwd_slide_kernel, dim3(c.ncol, B * c.nstrips), dim3(c.TX * (C / 4)), shm, (hipStream_t)stream, z, B, S, C,
               const_cast<float*>(mean_rstd), alpha, v, argmax, c.nstrips, c.rps, fin ? *fin : none, S + g_slide_rowpad);
  } else if (pool)
    DRS_LAUNCH(bn_act_pool_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, z, B, S, C, mean_rstd, alpha, v, argmax);
  else
    DRS_LAUNCH(bn_act_pool_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, z, B, S, C, mean_rstd, alpha, v, argmax);
  return DRS_LAUNCH_CHECK();
}

int drs_bn_act_pool_forward(const float* z, int B, int S, int C, const float* mean_rstd, float alpha, int pool,
                            float* out, int P_out, int ld_out, int coff_out, unsigned char* argmax, void* stream) {
  if (!out) return DRS_ERR_ARG;
  return bn_act_pool_forward_impl(z, B, S, C, mean_rstd, alpha, pool, out, P_out, ld_out, coff_out, argmax, nullptr, 0, stream);
}

// drs_bn_finish + drs_bn_act_pool_forward in ONE launch (pooled blocks, C <= 512): the multi-rank form of the forward batch norm is
// tile statistics -> fp64 sums -> all-reduce -> finish -> normalise; the finish (mean, rstd, moving averages) is worked out by the
// normalising kernel itself from the all-reduced sums (same arithmetic: same bits), which also leaves (mean, rstd) in mean_rstd
int drs_bn_finish_act_pool_forward(const double* sums, double count, float* mean_rstd, float* moving_mean, float* moving_var, double decay,
                                   int bessel, const float* z, int B, int S, int C, float alpha, int pool, float* out, int P_out, int ld_out,
                                   int coff_out, unsigned char* argmax, void* stream) {
  if (!sums || !mean_rstd || !out || count < 1.0 || !(pool & 1) || !slide_ok(C)) return DRS_ERR_ARG;
  const BnFinish fin = {sums, count, moving_mean, moving_var, (float)(1.0 - decay), bessel};
  return bn_act_pool_forward_impl(z, B, S, C, mean_rstd, alpha, pool, out, P_out, ld_out, coff_out, argmax, nullptr, 0, stream, &fin);
}

int drs_bn_act_pool_forward_terms(const float* z, int B, int S, int C, const float* mean_rstd, float alpha, int pool,
                                  float* out, int P_out, int ld_out, int coff_out, unsigned char* argmax,
                                  unsigned short* terms, int nterms, void* stream) {
  if (!terms) return DRS_ERR_ARG;
  return bn_act_pool_forward_impl(z, B, S, C, mean_rstd, alpha, pool, out, P_out, ld_out, coff_out, argmax, terms, nterms, stream);
}

// rows of the slab drs_bn_backward_reduce writes (partial must hold rows * C * 2 floats)
// pixels per workgroup of drs_bn_backward_reduce: 256 for big batches, down to 32 so that a small per-rank batch still
// fills the chip (>= ~2048 workgroups)
static int bn_bwd_rows_per_block(long long M) {
  int r = 256;
  while (r > 32 && M / r < 2048) r >>= 1;
  return r;
}
int drs_bn_backward_rows(int B, int S, int C, int pool) {
  if (pool && slide_ok(C)) {
    const SlideCfg c = slide_cfg(B, S, C);
    return c.ncol * B * c.nstrips;
  }
  const long long M = (long long)B * S * S;
  const int r = bn_bwd_rows_per_block(M);
  return (int)((M + r - 1) / r);
}

int drs_bn_backward_reduce(const float* ga, int ld_ga, int coff_ga, const float* z, const unsigned char* argmax, int B, int S,
                           int C, const float* mean_rstd, float alpha, int pool, float* gxhat, float* partial, void* stream) {
  if (!ga || !z || !mean_rstd || !gxhat || !partial || C % 4 || (pool && !argmax)) return DRS_ERR_ARG;
  const int CQ = C / 4;
  if (CQ > 256) return DRS_ERR_ARG;
  if (pool && slide_ok(C)) {
    const SlideCfg c = slide_cfg(B, S, C);
    const size_t xchg = (size_t)2 * (c.TX + 2) * CQ * 20, redu = (size_t)c.TX * C * 2 * sizeof(float);
    const size_t shm = xchg > redu ? xchg : redu;
    DRS_LAUNCH(bn_bwd_reduce_slide_kernel, dim3(c.ncol, B * c.nstrips), dim3(c.TX * CQ), shm, (hipStream_t)stream, ga, ld_ga, coff_ga, z,
               argmax, B, S, C, mean_rstd, alpha, gxhat, partial, c.nstrips, c.rps, S + g_slide_rowpad, chain_hp());
    return DRS_LAUNCH_CHECK();
  }
  const int PT = 256 / CQ;
  const int nblk = drs_bn_backward_rows(B, S, C, 0);
  const int rpb = bn_bwd_rows_per_block((long long)B * S * S);
  const size_t shm = (size_t)PT * C * 2 * sizeof(float);
  if (pool)       // wider than 512 channels: one column per workgroup, no neighbour to exchange with -- the gathering form
    DRS_LAUNCH(bn_bwd_reduce_kernel<true>, dim3(nblk), dim3(CQ * PT), shm, (hipStream_t)stream, ga, ld_ga, coff_ga, z,
               argmax, B, S, C, mean_rstd, alpha, gxhat, partial, rpb, chain_hp());
  else
    DRS_LAUNCH(bn_bwd_reduce_kernel<false>, dim3(nblk), dim3(CQ * PT), shm, (hipStream_t)stream, ga, ld_ga, coff_ga, z,
               argmax, B, S, C, mean_rstd, alpha, gxhat, partial, rpb, chain_hp());
  return DRS_LAUNCH_CHECK();
}

static int bn_backward_apply_impl(const float* gxhat, const float* z, int B, int S, int C, const float* mean_rstd,
                                  const double* sums, double count, float* gz, int P_out, int ld_out, int coff_out,
                                  unsigned short* terms, int nterms, void* stream) {
  if (!gxhat || !z || !mean_rstd || !sums || (!gz && !terms) || C % 4 || C > 1024) return DRS_ERR_ARG;
  if (terms && ((nterms != 2 && nterms != 3) || (ld_out & 31) || (coff_out & 31))) return DRS_ERR_ARG;
  const int Sp = S + 2 * P_out;
  if ((long long)B * Sp > 65535) return DRS_ERR_ARG;
  const int per_row = Sp * (C / 4);
  dim3 grid((per_row + 255) / 256, B * Sp);
  ActView v = mkview(gz, S, P_out, ld_out, coff_out);
  v.terms = terms; v.nt = terms ? nterms : 0;
  DRS_LAUNCH(bn_bwd_apply_kernel, grid, dim3(256), (size_t)4 * C * sizeof(float), (hipStream_t)stream, gxhat, z, B, S, C, mean_rstd, sums,
             count, v, chain_hp());
  return DRS_LAUNCH_CHECK();
}

int drs_bn_backward_apply(const float* gxhat, const float* z, int B, int S, int C, const float* mean_rstd, const double* sums,
                          double count, float* gz, int P_out, int ld_out, int coff_out, void* stream) {
  if (!gz) return DRS_ERR_ARG;
  return bn_backward_apply_impl(gxhat, z, B, S, C, mean_rstd, sums, count, gz, P_out, ld_out, coff_out, nullptr, 0, stream);
}

int drs_bn_backward_apply_means(const float* gxhat, const float* z, int B, int S, int C, const float* mean_rstd, const float* means,
                                float* gz, int P_out, int ld_out, int coff_out, void* stream) {
  if (!gxhat || !z || !mean_rstd || !means || !gz || C % 4 || C > 1024) return DRS_ERR_ARG;
  const int Sp = S + 2 * P_out;
  if ((long long)B * Sp > 65535) return DRS_ERR_ARG;
  const int per_row = Sp * (C / 4);
  DRS_LAUNCH(bn_bwd_apply_means_kernel, dim3((per_row + 255) / 256, B * Sp), dim3(256), 0, (hipStream_t)stream, gxhat, z, B, S, C, mean_rstd,
             means, mkview(gz, S, P_out, ld_out, coff_out), chain_hp());
  return DRS_LAUNCH_CHECK();
}

int drs_bn_backward_apply_terms(const float* gxhat, const float* z, int B, int S, int C, const float* mean_rstd,
                                const double* sums, double count, float* gz, int P_out, int ld_out, int coff_out,
                                unsigned short* terms, int nterms, void* stream) {
  if (!terms) return DRS_ERR_ARG;
  return bn_backward_apply_impl(gxhat, z, B, S, C, mean_rstd, sums, count, gz, P_out, ld_out, coff_out, terms, nterms, stream);
}

// slab rows = workgroups of a classifier launch: 64 .. pixels per workgroup, at most 1024 workgroups (monotone in B * S * S: a
// slab sized for (b_max, s_max) serves every smaller call)
int drs_classifier_rows(int B, int S) {
  const long long M = (long long)B * S * S;
  const long long n = (M + 63) / 64;
  return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}

int g_cls_variant = 1;       // development switch (drs_debug_cls_variant): 1 = by the class count and width (below), 2 = register MFMA form always, 3 = LDS-DMA MFMA form where it fits, 0 = vector-ALU form always

#ifdef DRS_DEV
int drs_debug_slide_blocks(int v) { const int old = g_slide_blocks; if (v >= 0) g_slide_blocks = v; return old; }
int drs_debug_slide_rowpad(int v) { const int old = g_slide_rowpad; if (v >= 0) g_slide_rowpad = v; return old; }
int drs_debug_slide_minrows(int v) { const int old = g_slide_minrows; if (v >= 1) g_slide_minrows = v; return old; }
int drs_debug_cls_variant(int v) { const int old = g_cls_variant; if (v >= 0) g_cls_variant = v; return old; }
#endif

int drs_classifier_loss(const float* feat, int B, int S, int P, int ld, int coff, int C, int K, const float* w,
                        const float* bias, const unsigned char* labels, const unsigned char* loss_mask,
                        const unsigned char* acc_mask, float inv_n, float* logits, unsigned char* pred, float* gfeat,
                        int ld_g, int coff_g, float* dw_partial, float* db_partial, double* loss_partial,
                        unsigned int* conf, void* stream) {
  if (!feat || !w || !bias || K < 1 || K > 8 || C % 64 || C / 64 > 7) return DRS_ERR_ARG;
  const long long M = (long long)B * S * S;
  if (M <= 0 || M >= (1 << 24)) return DRS_ERR_ARG;
  if (labels && !loss_partial) return DRS_ERR_ARG;
  if (gfeat && (!dw_partial || !db_partial || !labels)) return DRS_ERR_ARG;
  if (ld % 4 || coff % 4 || (gfeat && (ld_g % 4 || coff_g % 4))) return DRS_ERR_ARG;       // 16-byte feature / gradient accesses
  ClsArgs a;
  a.feat = mkview(const_cast<float*>(feat), S, P, ld, coff); a.C = C; a.K = K; a.M = (int)M; a.w = w; a.bias = bias;
  a.labels = labels; a.loss_mask = loss_mask; a.acc_mask = acc_mask; a.inv_n = inv_n; a.logits = logits; a.pred = pred;
  a.gfeat = gfeat; a.ld_g = ld_g; a.coff_g = coff_g; a.dw_partial = dw_partial; a.db_partial = db_partial;
  a.loss_partial = loss_partial; a.conf = conf;
  a.rcpS = 1.0f / (float)S; a.rcpSS = 1.0f / (float)(S * S);
  const int nblk = drs_classifier_rows(B, S);
  a.rows_per_block = (int)(((M + nblk - 1) / nblk + 63) / 64 * 64);      // whole 16-pixel tiles per wave; trailing workgroups may be empty
  hipStream_t st = (hipStream_t)stream;
  // the MFMA form pads the class dimension to 16 / 8, the vector-ALU form multiplies exactly K classes: in-process A/B
  // (tools/bench_classifier.py, profiles/r03/bench_classifier.log) K = 6, C = 256: MFMA 0.31 against 0.40 ms training and 0.11
  // against 0.19 ms inference at B = 128; K = 2, C = 448: vector-ALU 0.24 against 0.47 ms.  So: MFMA from four classes up.
  const bool mfma = g_cls_variant == 1 ? K >= 4 : g_cls_variant >= 2;
  a.dma_span = 1; a.nrows = nblk;
  // the LDS-DMA form (two feature tiles per wave fit the 160 KiB up to C = 256) is one workgroup per CU with a pipeline to fill: it
  // pays from 2^18 pixels (tools/bench_classifier.py: B = 128, S = 64 training 0.31 -> 0.26 ms; B = 16, S = 64: 0.059 -> 0.077 ms)
  if (mfma && C / 64 <= 4 && g_cls_variant != 2 && (g_cls_variant == 3 || M >= (1 << 18))) {
    a.dma_span = (nblk + 255) / 256;
    const int ndma = (nblk + a.dma_span - 1) / a.dma_span;
#define DRS_CLS_D(cq) do { if (labels) DRS_LAUNCH((classifier_dma_kernel<cq, true>), dim3(ndma), dim3(256), 0, st, a); \
                           else DRS_LAUNCH((classifier_dma_kernel<cq, false>), dim3(ndma), dim3(256), 0, st, a); } while (0)
    switch (C / 64) { case 1: DRS_CLS_D(1); break; case 2: DRS_CLS_D(2); break; case 3: DRS_CLS_D(3); break; default: DRS_CLS_D(4); break; }
#undef DRS_CLS_D
    return DRS_LAUNCH_CHECK();
  }
  if (mfma) {
#define DRS_CLS_M(cq) do { if (labels) DRS_LAUNCH((classifier_mfma_kernel<cq, true>), dim3(nblk), dim3(256), 0, st, a); \
                           else DRS_LAUNCH((classifier_mfma_kernel<cq, false>), dim3(nblk), dim3(256), 0, st, a); } while (0)
    switch (C / 64) { case 1: DRS_CLS_M(1); break; case 2: DRS_CLS_M(2); break; case 3: DRS_CLS_M(3); break; case 4: DRS_CLS_M(4); break;
                      case 5: DRS_CLS_M(5); break; case 6: DRS_CLS_M(6); break; default: DRS_CLS_M(7); break; }
#undef DRS_CLS_M
    return DRS_LAUNCH_CHECK();
  }
#define DRS_CLS_CASE(ci, km) DRS_LAUNCH((classifier_loss_kernel<ci, km>), dim3(nblk), dim3(256), 0, st, a)
#define DRS_CLS_KM(km) switch (C / 64) { case 1: DRS_CLS_CASE(1, km); break; case 2: DRS_CLS_CASE(2, km); break; case 3: DRS_CLS_CASE(3, km); break; \
    case 4: DRS_CLS_CASE(4, km); break; case 5: DRS_CLS_CASE(5, km); break; case 6: DRS_CLS_CASE(6, km); break; default: DRS_CLS_CASE(7, km); break; }
  if (K == 2) DRS_CLS_KM(2) else if (K == 6) DRS_CLS_KM(6) else if (K == 7) DRS_CLS_KM(7) else DRS_CLS_KM(8)
#undef DRS_CLS_KM
#undef DRS_CLS_CASE
  return DRS_LAUNCH_CHECK();
}

int drs_rows_reduce_f32(const float* in, int nrows, int ncols, float* out, double* scratch, void* stream) {
  if (!in || !out || !scratch || nrows < 1) return DRS_ERR_ARG;
  DRS_LAUNCH(colsum_l1_kernel, dim3((ncols + 63) / 64, COLSUM_BLOCKS), dim3(256), 0, (hipStream_t)stream, in, nrows, ncols, scratch);
  DRS_LAUNCH(colsum_l2_kernel<float>, dim3((ncols + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, ncols, out);
  return DRS_LAUNCH_CHECK();
}

int drs_sum_f64(const double* in, int n, double* out, void* stream) {
  if (!in || !out) return DRS_ERR_ARG;
  DRS_LAUNCH(sum_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, n, out);
  return DRS_LAUNCH_CHECK();
}

int drs_scale_f64(double* x, int n, double s, void* stream) {
  if (!x || n < 1) return DRS_ERR_ARG;
  DRS_LAUNCH(scale_f64_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, n, s);
  return DRS_LAUNCH_CHECK();
}

// l2 = 0.5 * sum(w[0..n)^2) -> out[0] (fp64); scratch holds 256 doubles
int drs_l2_loss(const float* w, size_t n, double* scratch, double* out, void* stream) {
  if (!w || !scratch || !out) return DRS_ERR_ARG;
  DRS_LAUNCH(l2_partial_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w, n, scratch);
  DRS_LAUNCH(sum_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, 256, out);
  return DRS_LAUNCH_CHECK();
}

int drs_momentum_update(float* w, const float* grad, float* accum, size_t n, size_t n_decay, float lr, float weight_decay,
                        float momentum, float grad_scale, void* stream) {
  if (!w || !grad || !accum) return DRS_ERR_ARG;
  const size_t nb = (n + 255) / 256;
  DRS_LAUNCH(momentum_kernel, dim3(nb < 2048 ? (unsigned)nb : 2048u), dim3(256), 0, (hipStream_t)stream, w, grad, accum, n,
                     n_decay, lr, weight_decay, momentum, grad_scale);
  return DRS_LAUNCH_CHECK();
}

int drs_confusion(const unsigned char* labels, const unsigned char* pred, const unsigned char* mask, size_t n, int K,
                  int ignore_label, unsigned int* conf, void* stream) {
  if (!labels || !pred || !conf || K < 1 || K > 8) return DRS_ERR_ARG;
  const size_t nb = (n + 255) / 256;
  DRS_LAUNCH(confusion_kernel, dim3(nb < 1024 ? (unsigned)nb : 1024u), dim3(256), 0, (hipStream_t)stream, labels, pred, mask,
                     n, K, ignore_label, conf);
  return DRS_LAUNCH_CHECK();
}

int drs_avg_pool_forward(const float* in, int B, int S, int C, int k, float* out, int P_out, int ld_out, int coff_out, void* stream) {
  if (!in || !out || C % 4 || k < 1 || !(k & 1) || (long long)B * (S + 2 * P_out) > 65535) return DRS_ERR_ARG;
  ActView v = mkview(out, S, P_out, ld_out, coff_out);
  if (P_out > 0) {
    const int Sp = S + 2 * P_out;
    DRS_LAUNCH(zero_halo_kernel, dim3((Sp * (C / 4) + 255) / 256, B * Sp), dim3(256), 0, (hipStream_t)stream, v, B, C);
  }
  DRS_LAUNCH(avg_pool_kernel<false>, dim3((S * (C / 4) + 255) / 256, B * S), dim3(256), 0, (hipStream_t)stream, in, C, 0, B, S, C, k, v);
  return DRS_LAUNCH_CHECK();
}

int drs_avg_pool_backward(const float* gout, int ld_g, int coff_g, int B, int S, int C, int k, float* gin, void* stream) {
  if (!gout || !gin || C % 4 || k < 1 || !(k & 1) || (long long)B * S > 65535) return DRS_ERR_ARG;
  DRS_LAUNCH(avg_pool_kernel<true>, dim3((S * (C / 4) + 255) / 256, B * S), dim3(256), 0, (hipStream_t)stream, gout, ld_g, coff_g, B, S, C,
             k, mkview(gin, S, 0, C, 0));
  return DRS_LAUNCH_CHECK();
}

// state: s [B][C], e1 [B][R], e2 [B][C] (kept for the backward pass)
int drs_se_forward(const float* act, int B, int S, int C, int R, const float* w1, const float* b1, const float* w2, const float* b2,
                   float* s, float* e1, float* e2, float* out, int P_out, int ld_out, int coff_out, void* stream) {
  if (!act || !w1 || !b1 || !w2 || !b2 || !s || !e1 || !e2 || !out || C % 4 || R < 1 || (long long)B * (S + 2 * P_out) > 65535)
    return DRS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  DRS_LAUNCH(se_spatial_reduce_kernel<false>, dim3((C + 63) / 64, B), dim3(256), 0, st, act, C, 0, nullptr, S, C, 1.0f / (float)(S * S), s);
  DRS_LAUNCH(se_excite_kernel, dim3(B), dim3(256), (C + R) * sizeof(float), st, s, w1, b1, w2, b2, C, R, e1, e2);
  ActView v = mkview(out, S, P_out, ld_out, coff_out);
  if (P_out > 0) {
    const int Sp = S + 2 * P_out;
    DRS_LAUNCH(zero_halo_kernel, dim3((Sp * (C / 4) + 255) / 256, B * Sp), dim3(256), 0, st, v, B, C);
  }
  DRS_LAUNCH(se_scale_kernel<false>, dim3((S * (C / 4) + 255) / 256, B * S), dim3(256), 0, st, act, C, 0, e2, nullptr, B, S, C, v);
  return DRS_LAUNCH_CHECK();
}

// gy [B*S*S][ld_g]+coff_g -> gact [B*S*S][C]; dw1 [C][R], db1 [R], dw2 [R][C], db2 [C]; scratch: B*(3*C + R) floats
int drs_se_backward(const float* gy, int ld_g, int coff_g, const float* act, const float* s, const float* e1, const float* e2,
                    const float* w1, const float* w2, int B, int S, int C, int R, float* gact, float* dw1, float* db1, float* dw2,
                    float* db2, float* scratch, void* stream) {
  if (!gy || !act || !s || !e1 || !e2 || !w1 || !w2 || !gact || !dw1 || !db1 || !dw2 || !db2 || !scratch || C % 4 || (long long)B * S > 65535)
    return DRS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* ge2 = scratch;                       // [B][C]
  float* gpre2 = ge2 + (size_t)B * C;         // [B][C]
  float* gs = gpre2 + (size_t)B * C;          // [B][C]
  float* gpre1 = gs + (size_t)B * C;          // [B][R]
  DRS_LAUNCH(se_spatial_reduce_kernel<true>, dim3((C + 63) / 64, B), dim3(256), 0, st, gy, ld_g, coff_g, act, S, C, 1.0f, ge2);
  DRS_LAUNCH(se_excite_bwd_kernel, dim3(B), dim3(256), (C + R) * sizeof(float), st, ge2, e1, e2, w1, w2, C, R, 1.0f / (float)(S * S), gpre2,
             gpre1, gs);
  DRS_LAUNCH(se_fc_wgrad_kernel, dim3(((R + 1) * C + 255) / 256), dim3(256), 0, st, e1, gpre2, B, R, C, dw2, db2);
  DRS_LAUNCH(se_fc_wgrad_kernel, dim3(((C + 1) * R + 255) / 256), dim3(256), 0, st, s, gpre1, B, C, R, dw1, db1);
  DRS_LAUNCH(se_scale_kernel<true>, dim3((S * (C / 4) + 255) / 256, B * S), dim3(256), 0, st, gy, ld_g, coff_g, e2, gs, B, S, C,
             mkview(gact, S, 0, C, 0));
  return DRS_LAUNCH_CHECK();
}

}  // extern "C"
