// Dilated stride-1 SAME convolution on gfx950 as an fp32 implicit GEMM on the MFMA pipe.
//
// Replaces the TensorFlow ops behind /root/reference/isprs_dilated_random.py:710-713
// (tf.nn.atrous_conv2d / tf.nn.conv2d + bias_add) and their gradients.
//
//   forward : z[p, o]  = sum_{u,v,c} X[p + (u,v)*rate - pad, c] * W[u,v,c,o] + bias[o]
//   dgrad   : the same kernel run on the (zero-haloed) output gradient with the filter flipped in
//             (u,v) and transposed in (c,o), and pad_before := pad_after
//   wgrad   : dW[u,v,c,o] = sum_p X[p + (u,v)*rate - pad, c] * G[p, o]   (split over pixels, slabs, ordered reduce)
//
// Data layout in HBM: activations are channels-last with an explicit zero halo ("padded NHWC"), so a filter tap is a
// wave-uniform address offset and the inner loop has no bounds checks; filters are HWIO = a row-major [k*k*Cin][Cout]
// GEMM B operand as they stand.  GEMM view: M = B*S*S pixels, N = Cout, K = k*k*Cin.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the fp32 roof of the chip).  One
// workgroup = 4 waves (one per SIMD); every wave owns 64x64 (or 64x32) of the output tile = 2x2 MFMA tiles = 64
// accumulator VGPRs; A tile [BM][32] and B tile [32][BN] are staged through LDS (register prefetch of the next
// K-step while the current one is multiplied).
#include "drs_common.hpp"
#include <cstdlib>

namespace {

constexpr int BK = 32;        // channels per K-step (every Cin on this path is a multiple of 32; conv1 is zero-padded)
constexpr int LDA = BK + 4;   // LDS row stride of the A tile: 36 floats -> ds_read_b128 of 16 rows hit 16 distinct slots

struct ConvArgs {
  const float* in; int S, P, ld_in, coff_in;
  int M;
  const float* w;       // [k*k*Cin][Cout]
  const float* bias;    // [Cout] or null
  float* out; int ld_out, coff_out;
  float* stats;         // [mtiles][Cout][2] partial (sum, sumsq) of the output rows of each M tile, or null
  int k, rate, pad, Cin, Cout;
  int accumulate;
  int skip_halo;
  float rcpS, rcpSS;
  // stream-K (sk_W > 0; conv_dma_kernel only): the launch has sk_W workgroups and the K-steps of ALL tiles, in tile-major order
  // (sk_U = tiles * sk_nks of them), are cut into sk_W equal consecutive ranges; a tile that one range covers whole is finished
  // by that workgroup, the others leave their partial sums in sk_slab and conv_sk_fixup_kernel adds them in a fixed order
  // Hybrid (r05): only the tiles [0, sk_T) are cut that way, into sk_W ranges (sk_U = sk_T * sk_nks); the other sk_tiles - sk_T tiles stay
  // whole and are dealt out with stride sk_G = the launch's workgroup count (tile sk_T + w + j sk_G to workgroup w): the full rounds of a
  // launch run whole tiles, only the remainder that would make a nearly empty last round is cut across all workgroups.  sk_T == sk_tiles,
  // sk_G == sk_W is the pure stream-K launch of r03.  sk_order: which part a workgroup does first (0 its range, 1 its whole tiles,
  // 2 alternating by workgroup slot)
  float* sk_slab;
  int sk_W, sk_nks, sk_U;
  int sk_T, sk_tiles, sk_G, sk_order;
  int prio;             // waves lower their priority as their workgroup advances (set_prio_by_progress)
  // launch order "long tiles first" (plain launches with halo-tap skipping, whole patches per XCD chunk): lpt_T = M tiles per patch
  // (0 = natural order), tiles [lpt_ta, lpt_tb) of a patch multiply every tap row, the others skip some; lpt_P = patches per XCD chunk
  int lpt_T, lpt_ta, lpt_tb, lpt_P;
#ifdef DRS_DEV
  unsigned long long* trace;      // development build: [workgroup][2] = (start, end) of the workgroup on the 100 MHz real-time clock, or null
#endif
};

// Wave priority by REMAINING work (r04).  The SIMD arbiter serves its oldest ready wave first, so the workgroups that share a CU do
// not advance together: stamps of every workgroup's start and end (tools/wgrad_spread.py) show the equal-length workgroups of a
// filter-gradient launch ending over +-13 % of their duration, all of it INSIDE a CU (CU means agree to 1 %), and the last
// workgroup of a CU running alone at the end of a launch.  A wave that is further along lowers its own priority (3 in its first
// quarter ... 0 in its last), the ones behind get the matrix pipe first and catch up, and the co-resident workgroups of the last
// round end together (spread 13 -> 5 %).  In-process A/B over the filter gradients of conv3..conv8 at B = 128: 15.21 -> 15.03 ms
// (conv1 / conv2 on the 64-wide register-staged form: -3.6 / -2.4 %); the reverse order (the one ahead keeps the pipe) 15.31,
// classes crowded towards the end 15.17, two levels 15.17.  Only from 2^18 pixels (wgrad_setup).  NOT in the forward / input-gradient kernel: its many short workgroups
// gain 0.3-0.7 % on the 128-wide tiles and LOSE 4-7 % on the 192-wide ones (three workgroups per CU).
// mode 1: levels 3, 2, 1, 0 by quarter; mode 2: 2, 1, 0, 0 (a launch that shares the chip with a chain the step waits for leaves the top level to it)
__device__ __forceinline__ void set_prio_by_progress(int done, int total, int& quarter, int mode = 1) {
  const int q = __builtin_amdgcn_readfirstlane((done * 4) / total);
  if (q == quarter) return;
  quarter = q;
  if (mode == 2) {
    if (q <= 0) __builtin_amdgcn_s_setprio(2);
    else if (q == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    return;
  }
  if (q <= 0) __builtin_amdgcn_s_setprio(3);
  else if (q == 1) __builtin_amdgcn_s_setprio(2);
  else if (q == 2) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// Launch order "full tiles first" of a plain launch: logical workgroup index (after the XCD remap: every XCD owns one contiguous
// chunk of `P` whole patches) -> tile.  Inside a chunk first the tiles [ta, tb) of every patch (they multiply every tap row), patch
// by patch in natural order, then the tiles that skip halo tap rows (the top / bottom ones of every patch).  A bijection on
// [0, chunks * P * T * ntn) that keeps the ntn column tiles of an M tile adjacent.
__host__ __device__ __forceinline__ int lpt_tile(int wg, int T, int ta, int tb, int P, int ntn) {
  const int per = P * T * ntn, nI = tb - ta, nE = T - nI;
  const int chunk = wg / per;
  int wl = wg - chunk * per, p, t;
  if (wl < P * nI * ntn) {
    p = wl / (nI * ntn);
    wl -= p * nI * ntn;
    t = ta + wl / ntn;
  } else {
    wl -= P * nI * ntn;
    p = wl / (nE * ntn);
    wl -= p * nE * ntn;
    const int e = wl / ntn;
    t = e < ta ? e : tb + (e - ta);
  }
  return ((chunk * P + p) * T + t) * ntn + wl % ntn;
}

// first K-step (of the tile-major sequence) of workgroup w, and the workgroup that owns K-step x: the static cut both the
// convolution kernel and the fix-up kernel work out for themselves
__host__ __device__ __forceinline__ int sk_first_unit(int w, int U, int W) { return (int)((long long)w * U / W); }
__host__ __device__ __forceinline__ int sk_owner(int x, int U, int W) { return (int)((((long long)x + 1) * W - 1) / U); }

// Epilogue shared by the forward / input-gradient kernels: bias, optional accumulate, store, and the tile's batch-norm statistics.
// C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  `scratch`: LDS no wave reads any more.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* scratch, int m0, int n0) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  float bv[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + wn * WTN + ni * 32 + li;
    bv[ni] = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < a.M) {
          float v = acc[mi][ni][r] + bv[ni];
          float* dst = a.out + (size_t)row * a.ld_out + a.coff_out + col;
          if (a.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
  if (a.stats) {
    // batch-norm statistics of the tile (never together with accumulate): fixed-order reduction over the lane halves, then
    // over the WM waves that share a column; one slab row per M tile
    const int rem = a.M - m0;
    tile_column_stats<TN, WM, BN>(
        scratch, t, wm, h == 0, (float)(rem < BM ? rem : BM), [&](int ni) { return wn * WTN + ni * 32 + li; },
        [](float s) { return s + __shfl_xor(s, 32); },
        [&](int ni, auto f) {
#pragma unroll
          for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (m0 + wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < a.M) f(acc[mi][ni][r] + bv[ni]);
        },
        a.stats + ((size_t)(m0 / BM) * a.Cout + n0) * 2);
  }
}

// PACK: the input has fewer than 32 channels (conv1: 3..5 bands in an 8-channel slab).  A K-step is then 32 / Cin consecutive
// filter taps x Cin channels instead of one tap x 32 channels, so the contraction is k*k*Cin long rather than k*k*32 (conv1: 224
// instead of 800); every thread adds the offset of ITS tap.  The filter is [round_up(k*k*Cin, 32)][Cout] with zero tail rows.
template <int BM, int BN, int WM, int WN, bool PACK = false>
__global__ __launch_bounds__(256, (BM == 256 || BN == 128) ? 3 : 1) void conv_igemm_kernel(const ConvArgs a) {   // 256-row and 128x128 tiles: hold the allocation to 3 waves per SIMD
  static_assert(WM * WN == 4, "4 waves");
  constexpr int LDB = BN + 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;   // MFMA tiles per wave
  constexpr int NA = BM / 32;                   // float4 A loads per thread per K-step
  constexpr int NB = BN / 32;                   // float4 B loads per thread per K-step
  constexpr int BROWS = 256 / (BN / 4);         // B rows covered by one pass of the 256 threads

  __shared__ __attribute__((aligned(16))) float lds[BM * LDA + BK * LDB];
  float* As = lds;
  float* Bs = lds + BM * LDA;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int ntn = a.Cout / BN;
  const int nblk = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, nblk);
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  const int Sp = a.S + 2 * a.P;
  // Addressing without per-K-step vector arithmetic: every load is (wave-uniform base in SGPRs) + (loop-invariant 32-bit byte
  // offset of this thread), i.e. `global_load_dwordx4 v, voff, s[base]`.  The K loop then issues nothing on the vector ALU
  // besides the MFMAs: VALU instructions of any wave on a SIMD take issue slots from that SIMD's matrix pipe (measured on the
  // filter-gradient kernel: its ~200 address / select instructions per chunk cost 10-15 % of the MFMA rate).
  // per-thread A source byte offsets: pixel part + slice + this thread's 4-channel column (slab bytes < 2^32, checked by the host)
  uint32_t offA[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int p = m0 + (t >> 3) + 32 * i;
    p = p < a.M ? p : a.M - 1;
    offA[i] = (padded_pixel_off(p, a.S, a.P, a.ld_in, a.rcpS, a.rcpSS, -a.pad, -a.pad) + (uint32_t)(a.coff_in + (PACK ? 0 : (t & 7) * 4))) * 4u;
  }
  const int brow = t / (BN / 4), bcol = (t % (BN / 4)) * 4;
  uint32_t offB[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) offB[i] = (uint32_t)((brow + BROWS * i) * a.Cout + n0 + bcol) * 4u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int cpt = a.Cin / BK;                 // K-steps per filter tap
  // tap rows that fall entirely into the zero halo for this tile's pixel rows are not multiplied at all (exact zeros)
  int u_lo, u_hi;
  live_tap_rows(m0, BM, a.M, a.S, a.k, a.rate, a.pad, a.rcpS, a.rcpSS, u_lo, u_hi);
  if (PACK || !a.skip_halo) { u_lo = 0; u_hi = a.k; }
  u_lo = __builtin_amdgcn_readfirstlane(u_lo);
  u_hi = __builtin_amdgcn_readfirstlane(u_hi);
  const int nks = PACK ? (a.k * a.k * a.Cin + BK - 1) / BK : (u_hi - u_lo) * a.k * cpt;   // K-steps (of the live tap rows), from 0
  const char* wlive = reinterpret_cast<const char*>(a.w + (size_t)u_lo * a.k * a.Cin * a.Cout);      // filter rows of the first live tap row
  const char* inb = reinterpret_cast<const char*>(a.in);
  f32x4 ra[NA], rb[NB];
  int lu = u_lo, lv = 0, lc = 0;              // (tap row, tap col, channel chunk) of the next K-step to fetch: wave-uniform, in SGPRs

  auto gload = [&](int ks) {
    if (PACK) {     // this thread's 4 channels belong to tap (32 ks + 4 (t & 7)) / Cin; taps past the last meet zero filter rows
      const int kidx = ks * BK + (t & 7) * 4;
      int tap = kidx / a.Cin;
      const int c = kidx - tap * a.Cin;
      tap = tap < a.k * a.k ? tap : a.k * a.k - 1;
      const int u = tap / a.k, v = tap - u * a.k;
      const uint32_t soff = (uint32_t)((u * a.rate * Sp + v * a.rate) * a.ld_in + c) * 4u;
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(inb + (offA[i] + soff));
    } else {
      const char* ab = inb + (size_t)(uint32_t)((lu * a.rate * Sp + lv * a.rate) * a.ld_in + lc * BK) * 4u;
#pragma unroll
      for (int i = 0; i < NA; ++i) { uint32_t o = offA[i]; asm volatile("" : "+v"(o)); ra[i] = *reinterpret_cast<const f32x4*>(ab + o); }   // (opaque: keeps the 32-bit offset form, global_load v, voff, s[base])
    }
    const char* wb = wlive + (size_t)(uint32_t)((PACK ? ks : ((lu - u_lo) * a.k + lv) * cpt + lc) * BK) * (uint32_t)a.Cout * 4u;
#pragma unroll
    for (int i = 0; i < NB; ++i) { uint32_t o = offB[i]; asm volatile("" : "+v"(o)); rb[i] = *reinterpret_cast<const f32x4*>(wb + o); }
    if (!PACK) { if (++lv == a.k) { lv = 0; if (++lu == u_hi) { lu = u_lo; ++lc; } } }     // K-step order as in conv_dma_kernel: the two forms add in the same order
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[((t >> 3) + 32 * i) * LDA + (t & 7) * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bs[(brow + BROWS * i) * LDB + bcol]) = rb[i];
  };

  gload(0);
  lstore();
  __syncthreads();
  const int arow = wm * WTM + li, bcolw = wn * WTN + li;
  for (int ks = 0; ks < nks; ++ks) {
    if (ks + 1 < nks) gload(ks + 1);
    // K is consumed in the order the wide LDS reads deliver it: lane-half h of read q supplies k = 8q+4h+e at step e.
    // The fragments of step st+1 are read from LDS BEFORE the MFMAs of step st are issued (two register sets used in turn, so no
    // copies; order pinned with sched_group_barrier), so an LDS round trip never sits between two MFMA groups.
    {
      constexpr int NST = BK / 2;             // MFMA k-steps per K-step
      f32x4 af[2][TM];
      float bf[2][TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[0][mi] = *reinterpret_cast<const f32x4*>(&As[(arow + mi * 32) * LDA + h * 4]);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) bf[0][ni] = Bs[(h * 4) * LDB + bcolw + ni * 32];
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int e = st & 3;
        if (st + 1 < NST) {
          const int q1 = (st + 1) >> 2, e1 = (st + 1) & 3;
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) bf[(st + 1) & 1][ni] = Bs[(q1 * 8 + h * 4 + e1) * LDB + bcolw + ni * 32];
          if (e1 == 0) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[q1 & 1][mi] = *reinterpret_cast<const f32x4*>(&As[(arow + mi * 32) * LDA + q1 * 8 + h * 4]);
            __builtin_amdgcn_sched_group_barrier(0x100, TN + TM, 0);                  // DS reads of the next step first
          } else {
            __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);
          }
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(st >> 2) & 1][mi][e], bf[st & 1][ni], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, TM * TN, 0);                        // then this step's MFMAs
      }
    }
    __syncthreads();
    if (ks + 1 < nks) { lstore(); __syncthreads(); }
  }

  conv_epilogue<BM, BN, WM, WN>(a, acc, lds, m0, n0);
}

// The forward / input-gradient tile with its operands brought in by LDS-DMA (as wgrad_dma_kernel): a K-step (32 channels of one
// tap) is multiplied as two 16-channel halves out of a double-buffered LDS image -- A half [BM pixels][16 ch] (64-B rows), B half
// [16 k][BN] -- the DMA of half h+1 landing while half h is multiplied; one barrier per half; no staging registers, no ds_write
// pass.  A DMA wave-instruction writes 64 lanes x 16 B to consecutive LDS bytes, so the A image cannot be padded; its
// ds_read_b128 fragment reads stay conflict-free through a swizzle realised on the SOURCE address: the 16-byte piece c of pixel
// row r sits in slot c ^ ((r >> 2) & 3) (the 16 lanes of every ds_read_b128 group then cover the 16 slots of the 256-byte bank
// row once).  K order, accumulation order and epilogue are those of conv_igemm_kernel: results are bitwise the same.
// SK: the stream-K form (a.sk_W workgroups, each a range of the tile-major K-step sequence); !SK: one workgroup per tile.
template <int BM, int BN, int WM, int WN, bool SK>
__global__ __launch_bounds__(256, 3) void conv_dma_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass has no LDS-DMA builtin; it only needs the launch stub
  static_assert(WM * WN == 4 && BM == 128, "4 waves, 128-pixel tiles");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int HK = 16;                         // channels per pipeline stage (half a K-step)
  constexpr int ASTAGE = BM * HK, STAGE = ASTAGE + HK * BN;   // floats
  constexpr int IA = (BM / 16) / 4;              // A DMA instructions per wave and half (16 pixel rows each)
  constexpr int BQ = BN / 4;                     // 16-byte pieces per B k-row
  constexpr int IB = HK * BQ / 64 / 4;           // B DMA instructions per wave and half: the half image [HK][BN] as one linear run of 1-KiB pieces
  static_assert(IA >= 1 && IB >= 1 && IB * 4 * 64 == HK * BQ, "tile / wave layout");

  __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);      // (scalar: so is the LDS address of each DMA piece)
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = a.Cout / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
#ifdef DRS_DEV
  if (a.trace && t == 0) a.trace[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
  const int Sp = a.S + 2 * a.P;
  const int cpt = a.Cin / BK;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // raw buffer descriptors over the input slab and the filter (stride 0, no swizzle; every offset below is a byte offset INSIDE its
  // tensor, < 2^32: the host checks the slab size)
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0xffffffff, 0x00020000);
  // fragment addresses (floats): A row r = arow + 32 mi, piece c = 2 q + h in slot c ^ ((r >> 2) & 3); B row k, column
  const int arow = wm * WTM + li, bcolw = wn * WTN + li;
  const int sw = (li >> 2) & 3;                 // (arow + 32 mi) >> 2 & 3 == (li >> 2) & 3: WTM and 32 are multiples of 16

  // the K-steps this workgroup multiplies: one whole tile (tile = wg), or -- stream-K -- the range [u, u_end) of the tile-major
  // sequence, i.e. the tail of one tile, whole tiles, the head of another
  // (hybrid: + the whole tiles sk_T + wg, sk_T + wg + sk_G, ...; the sk_W <= sk_G ranges are spread evenly over the workgroups: range j
  //  goes to the first workgroup w with ceil(w sk_W / sk_G) == j)
  int u = 0, u_end = 1, rng = 0, dp_next = 0;
  bool dp_first = false;
  if (SK) {
    const int jb = (int)(((long long)wg * a.sk_W + a.sk_G - 1) / a.sk_G), je = (int)(((long long)(wg + 1) * a.sk_W + a.sk_G - 1) / a.sk_G);
    rng = __builtin_amdgcn_readfirstlane(jb);
    u = u_end = 0;
    if (je > jb) {
      u = __builtin_amdgcn_readfirstlane(sk_first_unit(jb, a.sk_U, a.sk_W));
      u_end = __builtin_amdgcn_readfirstlane(sk_first_unit(jb + 1, a.sk_U, a.sk_W));
    }
    dp_next = a.sk_T + wg;
    if (u >= u_end && dp_next >= a.sk_tiles) return;
    dp_first = a.sk_order == 1 || (a.sk_order == 2 && ((blockIdx.x >> 8) & 1));
  }
  // the workgroups of a launch start in index order and the launch ends with its last round draining: let that round be the SHORT
  // tiles (r04, tools/conv_tail.py: the drain of conv8's forward launch 419 -> 216 us, idle workgroup slots 6.4 -> 3.5 % of the launch;
  // in-process A/B over the 14 forward / input-gradient launches at B = 128: -1.2 .. -5.5 % each, 29.7 -> 28.8 ms)
  const int tile0 = (!SK && a.lpt_T) ? __builtin_amdgcn_readfirstlane(lpt_tile(wg, a.lpt_T, a.lpt_ta, a.lpt_tb, a.lpt_P, ntn)) : wg;
  bool first_seg = true;
  int quart = -1, done = 0, total = 1;
  if (a.prio == 3) __builtin_amdgcn_s_setprio(3);      // the whole launch above a filter gradient that shares the chip (drs_tl_chain)
  if (SK && a.prio) total = (u_end - u) + (dp_next < a.sk_tiles ? ((a.sk_tiles - 1 - dp_next) / a.sk_G + 1) * a.sk_nks : 0);
  for (;;) {
    int tile = tile0, kb = 0, ke = 0;
    bool in_range = false;
    if (SK) {
      in_range = u < u_end && !(dp_first && dp_next < a.sk_tiles);
      if (in_range) {
        tile = u / a.sk_nks;
        kb = u - tile * a.sk_nks;
        ke = kb + (u_end - u);
        ke = ke < a.sk_nks ? ke : a.sk_nks;
      } else {
        tile = dp_next;
        dp_next += a.sk_G;
        kb = 0;
        ke = a.sk_nks;
      }
    }
    const int m0 = (tile / ntn) * BM;
    const int n0 = (tile % ntn) * BN;

    // DMA lane roles.  A: instruction j covers pixel rows 16 j + (lane >> 2); this lane fills slot (lane & 3) of its row with
    // the source piece (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3)
    uint32_t offA[IA], offB[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      int p = m0 + (wave + 4 * i) * 16 + (lane >> 2);
      p = p < a.M ? p : a.M - 1;
      offA[i] = (padded_pixel_off(p, a.S, a.P, a.ld_in, a.rcpS, a.rcpSS, -a.pad, -a.pad) + (uint32_t)a.coff_in) * 4u +
                (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
    }
    // B: instruction j moves pieces 64 j + lane of the linear half image: piece f is the 16-byte column f % BQ of k-row f / BQ
    // (BN = 192: a k-row is 768 B, so an instruction spans rows; every lane has its own (row, column) per instruction)
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int f = 64 * (wave + 4 * i) + lane;
      offB[i] = (uint32_t)(((f / BQ) * a.Cout + n0 + (f % BQ) * 4)) * 4u;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    int u_lo, u_hi;
    live_tap_rows(m0, BM, a.M, a.S, a.k, a.rate, a.pad, a.rcpS, a.rcpSS, u_lo, u_hi);
    if (!a.skip_halo || SK) { u_lo = 0; u_hi = a.k; }          // (stream-K: every tile has the same number of K-steps)
    u_lo = __builtin_amdgcn_readfirstlane(u_lo);
    u_hi = __builtin_amdgcn_readfirstlane(u_hi);
    const int nks = (u_hi - u_lo) * a.k * cpt;
    if (!SK) ke = nks;
    const uint32_t wlive_off = (uint32_t)(u_lo * a.k * a.Cin * a.Cout) * 4u;      // byte offset of the first live tap row's filter rows
    // (tap row, tap col, channel chunk) of the K-step being fetched; K-step j of a tile is (chunk, tap row, tap col) = (j / (rows k), ...)
    int lu = u_lo, lv = 0, lc = 0;
    if (SK && kb) {
      const int per_chunk = (u_hi - u_lo) * a.k;
      lc = kb / per_chunk;
      const int rem = kb - lc * per_chunk;
      lu = u_lo + rem / a.k;
      lv = rem - (rem / a.k) * a.k;
      lc = __builtin_amdgcn_readfirstlane(lc); lu = __builtin_amdgcn_readfirstlane(lu); lv = __builtin_amdgcn_readfirstlane(lv);
    }

    // The DMA as BUFFER loads: address = descriptor base + the lane's loop-invariant byte offset (a VGPR that is never touched
    // again) + the K-step's byte offset (an SGPR) -- no vector instruction per piece at all (the global form took a 64-bit address
    // built per piece: one v_mov / v_lshl_add_u64 each, and every VALU instruction issued in this loop costs matrix-pipe time)
    auto issue = [&](int half, int stage) {
      float* sa = lds + stage * STAGE;
      float* sb = sa + ASTAGE;
      const uint32_t ao = (uint32_t)((lu * a.rate * Sp + lv * a.rate) * a.ld_in + lc * BK + half * HK) * 4u;
      const uint32_t wo = wlive_off + (uint32_t)((((lu - u_lo) * a.k + lv) * cpt + lc) * BK + half * HK) * (uint32_t)a.Cout * 4u;
#pragma unroll
      for (int i = 0; i < IA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr)(sa + (wave + 4 * i) * 256), 16, (int)offA[i], (int)ao, 0, 0);
#pragma unroll
      for (int i = 0; i < IB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr)(sb + (wave + 4 * i) * 256), 16, (int)offB[i], (int)wo, 0, 0);
    };
    // K order (channel chunk, tap row, tap column): the k*k shifted reads of one 32-channel chunk follow each other, so the lines a
    // tap shares with the one before it (a column shift keeps 7/8 of a row) and with the neighbouring tiles' taps are still in L2
    // when they are read again.  Against (tap row, tap column, chunk), in-process A/B with a run-time switch (profiles/r02/
    // conv_korder_ab.txt): fabric-side fetch per launch 6.1 -> 0.9 GB (conv6), 7.8 -> 2.2 GB (conv8); forward -1 %, dgrad -2.4 %.
    // (The switch itself is gone: it sent the loop counters to scratch memory and their arithmetic to the vector ALU, -7 %.)
    auto next_kstep = [&]() { if (++lv == a.k) { lv = 0; if (++lu == u_hi) { lu = u_lo; ++lc; } } };

    auto compute = [&](int stage) {
      const float* As = lds + stage * STAGE;
      const float* Bs = As + ASTAGE;
      constexpr int NST = HK / 2;                 // MFMA k-steps per half
      f32x4 af[2][TM];
      float bf[2][TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) af[0][mi] = *reinterpret_cast<const f32x4*>(&As[(arow + mi * 32) * HK + ((h ^ sw) * 4)]);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) bf[0][ni] = Bs[(h * 4) * BN + bcolw + ni * 32];
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int e = st & 3;
        if (st + 1 < NST) {
          const int q1 = (st + 1) >> 2, e1 = (st + 1) & 3;
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) bf[(st + 1) & 1][ni] = Bs[(q1 * 8 + h * 4 + e1) * BN + bcolw + ni * 32];
          if (e1 == 0) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
              af[q1 & 1][mi] = *reinterpret_cast<const f32x4*>(&As[(arow + mi * 32) * HK + (((2 * q1 + h) ^ sw) * 4)]);
            __builtin_amdgcn_sched_group_barrier(0x100, TN + TM, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);
          }
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(st >> 2) & 1][mi][e], bf[st & 1][ni], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, TM * TN, 0);
      }
    };

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!SK && a.prio) total = nks;
    for (int ks = kb; ks < ke; ++ks) {
      if (a.prio == 1) set_prio_by_progress(SK ? done + (ks - kb) : ks, total, quart);
      issue(1, 1);                                  // second half of this K-step lands while the first is multiplied
      compute(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      next_kstep();
      if (ks + 1 < ke) issue(0, 0);                 // first half of the next K-step (every wave is done with stage 0)
      compute(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (!SK || (kb == 0 && ke == nks)) {
      conv_epilogue<BM, BN, WM, WN>(a, acc, lds, m0, n0);
    } else {
      // partial sums of a tile this workgroup shares with others: a piece of the slab in accumulator order, four registers per
      // access ([fragment quad][lane][4]: 1-KB wave stores); a workgroup has at most two such segments, its first (piece 2 w)
      // and its last (piece 2 w + 1)
      float* piece = a.sk_slab + (size_t)(2 * rng + (first_seg ? 0 : 1)) * (BM * BN) + (size_t)wave * (TM * TN * 16 * 64) + lane * 4;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(piece + ((mi * TN + ni) * 4 + q) * 256) =
                f32x4{acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
    }
    if (!SK) break;
    done += ke - kb;
    if (in_range) { u += ke - kb; first_seg = false; }
    if (u >= u_end && dp_next >= a.sk_tiles) break;
    __syncthreads();                                // the epilogue's LDS scratch is free before the next segment's DMA lands
  }
#ifdef DRS_DEV
  if (a.trace && t == 0) a.trace[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
#endif
}

// Stream-K fix-up: one workgroup per output tile.  A tile whose K-steps lie in ONE workgroup's range was finished there; the
// others are the sum of the pieces their workgroups left in the slab, added in workgroup (= K) order, then the same epilogue
// (bias, accumulate, store, batch-norm statistics of the tile) as the convolution kernel's.  Thread -> element mapping is the
// convolution kernel's accumulator layout, so pieces are read as they were written.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_sk_fixup_kernel(const ConvArgs a) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  __shared__ float lds[2 * WM * BN];
  if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int tile = blockIdx.x;
  const int x0 = tile * a.sk_nks;
  const int w_first = sk_owner(x0, a.sk_U, a.sk_W), w_last = sk_owner(x0 + a.sk_nks - 1, a.sk_U, a.sk_W);
  if (w_first == w_last) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntn = a.Cout / BN;
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  // Pieces are added in workgroup (= K) order.  A tile has ~5 of them at the per-rank batches, each an HBM / fabric round trip for
  // this workgroup: two pieces are fetched at a time (all their loads in flight before the first addition), which halves the
  // chain of round trips (r04: the fix-up launches were 21 us each at 16 patches of 25 x 25, 270 us of a 1.8 ms step).
  auto piece_of = [&](int w) {
    const int slot = 2 * w + (sk_first_unit(w, a.sk_U, a.sk_W) >= x0 ? 0 : 1);      // the workgroup's first segment, or its last
    return a.sk_slab + (size_t)slot * (BM * BN) + (size_t)wave * (TM * TN * 16 * 64) + lane * 4;
  };
  constexpr int NF = TM * TN * 4;                  // 16-byte fragments of a piece per thread
  int w = w_first;
  if (NF <= 16) {
    for (; w + 1 <= w_last; w += 2) {
      const float* pa = piece_of(w);
      const float* pb = piece_of(w + 1);
      f32x4 va[NF], vb[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) va[f] = *reinterpret_cast<const f32x4*>(pa + f * 256);
#pragma unroll
      for (int f = 0; f < NF; ++f) vb[f] = *reinterpret_cast<const f32x4*>(pb + f * 256);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[(f >> 2) / TN][(f >> 2) % TN][4 * (f & 3) + j] += va[f][j];
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[(f >> 2) / TN][(f >> 2) % TN][4 * (f & 3) + j] += vb[f][j];
    }
  }
  for (; w <= w_last; ++w) {
    const float* piece = piece_of(w);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(piece + ((mi * TN + ni) * 4 + q) * 256);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[mi][ni][4 * q + j] += v[j];
        }
  }
  conv_epilogue<BM, BN, WM, WN>(a, acc, lds, m0, n0);
}

// ------------------------------------------------------------------------------------------------ wgrad
// How the pixel dimension of the filter gradient is cut into workgroups.  A tile of filter rows meets image data only in the
// live rows of every image (live_pixel_range); with the dead chunks skipped, equal chunk ranges make unequal workgroups, and a
// launch that fits the chip in one or two rounds then ends with its longest ones (measured: skipping LOST time below 2^19
// pixels).  So the cut is by LIVE pixels: row tiles fall into classes by their live pixels per image (<= WG_MAXC distinct values:
// tap rows entirely inside / reaching over the top / the bottom edge), class c is cut into n[c] splits of equal live-pixel
// count, n[c] proportional to the class's live pixels, and every workgroup of the launch then has the same length.
// Workgroup order stays split-major (the tiles of one split read the same pixels at the same time, which is what keeps X and G
// in L2): classes are sorted by n ascending, levels [n[j-1], n[j]) hold the row tiles of classes >= j.
constexpr int WG_MAXT = 64, WG_MAXC = 8;
struct WgradPlan {
  int nclass;                           // 0: every row tile has `nsplit` equal chunk ranges (chunks_per_split)
  int n[WG_MAXC];                       // splits of class c, ascending
  int seg_start[WG_MAXC + 1];           // first workgroup of segment j = levels n[j-1] .. n[j] - 1
  int seg_width[WG_MAXC];               // workgroups per level in segment j
  unsigned char cls[WG_MAXT];           // class of row tile r
};

struct WgradArgs {
  const float* x; int S, Px, ld_x, coff_x;
  const float* g; int Pg, ld_g, coff_g;
  int M;
  int k, rate, pad, Cin, Cout;
  float* slab;               // [split][k*k*Cin][Cout]
  int chunks_per_split;      // 32-pixel chunks per split (uniform cut)
  int ntr, nto;
  int skip_halo;             // the walk jumps over the dead chunks (0: it multiplies their zeros -- same sums, bitwise)
  int live_cut;              // the live ranges shape the cut (plan) whether or not the walk skips
  int prio;                  // waves lower their priority as they advance (set_prio_by_progress)
  int ablate;                // timing experiments only (libdrs_hip_dev.so): 1 = every tap reads the un-shifted pixels (wrong sums)
  float rcpS, rcpSS;
  WgradPlan plan;
#ifdef DRS_DEV
  unsigned long long* trace;      // development build: [workgroup][2] = (start, end) on the 100 MHz real-time clock, or null
#endif
};

// workgroup id -> (tile, split), the split's chunk range [cbeg, cend) and the tile's live pixel range
__host__ __device__ __forceinline__ int wave_uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}

__host__ __device__ __forceinline__ void wgrad_assign(const WgradArgs& a, int TR, int id, int& tile, int& split, int& cbeg, int& cend,
                                                      int& live_lo, int& live_hi) {
  const int nchunks_total = (a.M + 31) / 32;
  const int rows_all = a.k * a.k * a.Cin;
  int nr = 0;
  if (a.plan.nclass == 0) {
    const int ntile = a.ntr * a.nto;
    split = id / ntile;
    tile = id % ntile;
  } else {
    int j = 0;
    while (j + 1 < a.plan.nclass && id >= a.plan.seg_start[j + 1]) ++j;
    const int rel = id - a.plan.seg_start[j];
    split = (j ? a.plan.n[j - 1] : 0) + rel / a.plan.seg_width[j];
    const int pos = rel % a.plan.seg_width[j];
    const int want = pos / a.nto;
    int r = 0, seen = 0;
    for (; r < a.ntr; ++r)
      if (a.plan.cls[r] >= j) { if (seen == want) break; ++seen; }
    r = r < a.ntr ? r : a.ntr - 1;              // (cannot happen for a plan of wgrad_live_plan; never index past the tables)
    tile = r * a.nto + pos % a.nto;
    nr = a.plan.n[a.plan.cls[r]];
  }
  tile = wave_uniform(tile);
  split = wave_uniform(split);
  const int R0 = (tile / a.nto) * TR;
  const int rlast = (R0 + TR < rows_all ? R0 + TR : rows_all) - 1;
  int lo, hi;
  live_pixel_range(R0, rlast, a.Cin, a.k, a.rate, a.pad, a.S, a.live_cut, lo, hi);
  if (a.plan.nclass == 0) {
    cbeg = split * a.chunks_per_split;
    cend = cbeg + a.chunks_per_split;
    cend = cend < nchunks_total ? cend : nchunks_total;
  } else {
    // split s of nr owns the live pixels [s q, (s+1) q) of the tile, counted through the images in order; the chunk that holds a
    // boundary pixel starts the later split (every chunk then belongs to exactly one split, dead ones included)
    const int S2 = a.S * a.S, lp = hi - lo;
    const int live_total = (a.M / S2) * lp;
    const int q = (live_total + nr - 1) / nr;
    auto bound = [&](int s) -> int {
      if (s <= 0) return 0;
      const long long jp = (long long)s * q;
      if (s >= nr || jp >= live_total) return nchunks_total;
      const int img = (int)jp / lp, rem = (int)jp - img * lp;
      return (img * S2 + lo + rem) >> 5;
    };
    cbeg = bound(split);
    cend = bound(split + 1);
  }
  cbeg = wave_uniform(cbeg);
  cend = wave_uniform(cend);
  if (!a.skip_halo) { lo = 0; hi = a.S * a.S; }
  live_lo = lo; live_hi = hi;
}

// Walk over the 32-pixel chunks of a pixel range that meet the live rows [lo, hi) of some image (drs_common.hpp), kept entirely in
// wave-uniform integers: no division after init(), so the walk costs scalar-ALU instructions only.  c = chunk index; (b, r) =
// image and in-image position of the chunk's first pixel; (y, x0) = its row / column, maintained when S % 32 == 0 (a chunk then
// lies inside one image row and the chunk holding pixel `lo` starts exactly at it).
struct ChunkWalk {
  int c, b, r, y, x0;
  int S, S2, lo, hi, ylo;
  __device__ __forceinline__ void init(int cfirst, int S_, float rcpS, float rcpSS, int lo_, int hi_) {
    S = S_; S2 = S_ * S_; lo = lo_; hi = hi_;
    int q;
    divmod24(lo, S, rcpS, ylo, q);
    c = next_live_chunk(cfirst - 1, S2, rcpSS, lo, hi);
    divmod24(c * 32, S2, rcpSS, b, r);
    divmod24(r, S, rcpS, y, x0);
    c = __builtin_amdgcn_readfirstlane(c); b = __builtin_amdgcn_readfirstlane(b); r = __builtin_amdgcn_readfirstlane(r);
    y = __builtin_amdgcn_readfirstlane(y); x0 = __builtin_amdgcn_readfirstlane(x0); ylo = __builtin_amdgcn_readfirstlane(ylo);
  }
  // to the next chunk that meets a live row; returns the number of chunks advanced
  __device__ __forceinline__ int advance() {
    int r1 = r + 32, b1 = b;
    if (r1 >= S2) { r1 -= S2; ++b1; }
    if ((r1 + 31 >= lo && r1 < hi) || r1 + 31 >= S2 + lo) {
      ++c; r = r1; b = b1;
      x0 += 32;
      if (x0 >= S) { x0 -= S; ++y; }               // (S >= 32: at most one row wrap per chunk; smaller sides do not use y / x0)
      if (y >= S) y = 0;
      return 1;
    }
    const int tb = r1 < lo ? b1 : b1 + 1;          // image whose live rows come next
    const int T = tb * S2 + lo, c_old = c;
    const int f = T & 31;                          // the chunk holding pixel T starts f pixels before it
    c = T >> 5;
    if (f <= lo) { b = tb; r = lo - f; } else { b = tb - 1; r = S2 + lo - f; }
    // pixel T is the first of image row ylo; the chunk starts f < 32 <= S pixels before it: at the row start (S % 32 == 0: always), or
    // S - f into the row above -- the last row of the image before when ylo == 0
    if (f == 0) { y = ylo; x0 = 0; } else { x0 = S - f; y = ylo - 1; if (y < 0) y = S - 1; }
    return c - c_old;
  }
};

// The filter-gradient tile with a K loop that issues no vector-ALU instruction besides its MFMAs: VALU instructions of ANY
// wave on a SIMD take issue slots from that SIMD's matrix pipe (measured on wgrad_kernel: removing its ~200 address / select
// instructions per chunk raises the MFMA rate by 10-15 %).  Every global load is (wave-uniform base in SGPRs) + (32-bit byte
// offset of this thread): loop-invariant when S % 32 == 0, one table read + add otherwise; the chunk walk is scalar
// (ChunkWalk); rows beyond k*k*Cin are loaded as they are (they only feed accumulator rows that are never stored); pixels
// beyond the tensor exist only in the last chunk, which alone pays for the select.  Chunk order, tile geometry, slab layout
// and the order of every sum are those of the first form of this kernel: results are bitwise the same
// (profiles/r02/wgrad_ablation.txt has the A/B and the ablations).
template <int TR, int TO, bool AFF>      // AFF: S % 32 == 0, as in wgrad_dma_kernel
__global__ __launch_bounds__(64 * (TR >= 64 ? 2 : 1) * (TO >= 64 ? 2 : 1), 3) void wgrad_kernel(const WgradArgs a) {
  constexpr int WR = TR >= 64 ? 2 : 1, WC = TO >= 64 ? 2 : 1;
  constexpr int NT = 64 * WR * WC;
  constexpr int WTR = TR / WR, WTO = TO / WC;
  constexpr int TMr = WTR / 32, TNo = WTO / 32;
  constexpr int BP = 32;                      // pixels per K-step
  constexpr int LDX = TR + 4, LDG = TO + 4;
  constexpr int XQ = TR / 4, GQ = TO / 4;     // float4 per pixel row
  constexpr int NX = BP * XQ / NT, NG = BP * GQ / NT;
  constexpr int XPS = NT / XQ, GPS = NT / GQ; // pixel stride between a thread's successive loads

  __shared__ __attribute__((aligned(16))) float lds[BP * LDX + BP * LDG];
  __shared__ uint32_t tabx[2][BP], tabg[2][BP];   // BYTE offsets of the chunk's pixels (walk with S % 32 != 0), double-buffered
  float* Xs = lds;
  float* Gs = lds + BP * LDX;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;

  int tile, split, cbeg, cend, live_lo, live_hi;
#ifdef DRS_DEV
  if (a.trace && t == 0) {
    a.trace[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    a.trace[32768 + blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);   // HW_ID, XCC_ID
  }
#endif
  wgrad_assign(a, TR, xcd_remap(blockIdx.x, gridDim.x), tile, split, cbeg, cend, live_lo, live_hi);
  const int R0 = (tile / a.nto) * TR;         // first row of the [k*k*Cin] dimension; a tile may span several taps
  const int o0 = (tile % a.nto) * TO;         // and the last one may be ragged (rows beyond k*k*Cin are not stored)
  const int rows_all = a.k * a.k * a.Cin;
  // this thread always stages the same 4 rows (tap, c..c+3) of the tile: its tap shift is a per-thread constant
  const int myR = R0 + (t % XQ) * 4;
  const int myRc = myR < rows_all ? myR : 0;                 // rows past the end: any valid address will do
  const int tap = myRc / a.Cin, c0 = myRc % a.Cin;
  const int u = tap / a.k, v = tap % a.k;
  const int Sxp = a.S + 2 * a.Px, Sgp = a.S + 2 * a.Pg;
  const uint32_t xconst = (uint32_t)((u * a.rate * Sxp + v * a.rate) * a.ld_x + a.coff_x + c0) * 4u;
  const uint32_t gconst = (uint32_t)(a.coff_g + o0 + (t % GQ) * 4) * 4u;
  const int xpix = t / XQ, gpix = t / GQ;
  constexpr bool affine = AFF;
  uint32_t xoff[NX], goff[NG];                // S % 32 == 0: the whole per-thread part of the address
#pragma unroll
  for (int i = 0; i < NX; ++i) xoff[i] = xconst + (affine ? (uint32_t)((xpix + XPS * i) * a.ld_x) * 4u : 0u);
#pragma unroll
  for (int i = 0; i < NG; ++i) goff[i] = gconst + (affine ? (uint32_t)((gpix + GPS * i) * a.ld_g) * 4u : 0u);

  f32x16 acc[TMr][TNo];
#pragma unroll
  for (int mi = 0; mi < TMr; ++mi)
#pragma unroll
    for (int ni = 0; ni < TNo; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nchunks_total = (a.M + BP - 1) / BP;
  const int clast = (a.M & 31) ? nchunks_total - 1 : -1;     // the one chunk that holds pixels past the end, if any
  ChunkWalk w;
  w.init(cbeg, a.S, a.rcpS, a.rcpSS, live_lo, live_hi);

  // S % 32 != 0: a chunk crosses image rows, so its 32 pixels get a table of byte offsets, written by the first 32 threads.
  // A thread follows ITS pixel (pb, py, px) through the walk: +32 pixels is a couple of compares; only a jump over dead rows
  // (once per image and tile) or a pixel past the end pays for the divisions.
  // The thread carries the BYTE OFFSETS of its pixel in the two slabs along with (py, px): +32 pixels is three additions, a row
  // wrap adds the slab's row jump, an image wrap its image jump -- compares, selects and additions only; the integer multiplies of
  // the offset formula (quarter rate) are paid once per jump.  (r04: the fills cost 3.5-5 % of a launch at S % 32 != 0 -- one wave
  // of four does them and the workgroup moves at its pace.)
  int py = 0, px = 0;
  uint32_t ox = 0, og = 0;
  const uint32_t stepx = (uint32_t)(32 * a.ld_x) * 4u, stepg = (uint32_t)(32 * a.ld_g) * 4u;
  const uint32_t rowjx = (uint32_t)((Sxp - a.S) * a.ld_x) * 4u, rowjg = (uint32_t)((Sgp - a.S) * a.ld_g) * 4u;
  const uint32_t imgjx = (uint32_t)(Sxp * (Sxp - a.S) * a.ld_x) * 4u, imgjg = (uint32_t)(Sgp * (Sgp - a.S) * a.ld_g) * 4u;
  auto pixel_from_index = [&](int p) {
    int pb, rem;
    divmod24(p < a.M ? p : a.M - 1, w.S2, a.rcpSS, pb, rem);
    divmod24(rem, a.S, a.rcpS, py, px);
    ox = (uint32_t)(((pb * Sxp + py + a.Px - a.pad) * Sxp + px + a.Px - a.pad) * a.ld_x) * 4u;
    og = (uint32_t)(((pb * Sgp + py + a.Pg) * Sgp + px + a.Pg) * a.ld_g) * 4u;
  };
  if (!affine && t < BP) pixel_from_index(w.c * BP + t);
  auto fill_tables = [&](int slot, int stepped) {      // for chunk w.c; `stepped` = chunks the walk advanced since this thread's pixel was set
    if (affine || t >= BP || w.c >= cend) return;
    const int p = w.c * BP + t;
    if (stepped == 1 && p < a.M && a.S >= 11) {
      px += 32; ox += stepx; og += stepg;
      if (a.S >= 32) {                           // (uniform) at most one row wrap per 32 pixels
        if (px >= a.S) { px -= a.S; ++py; ox += rowjx; og += rowjg; }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) if (px >= a.S) { px -= a.S; ++py; ox += rowjx; og += rowjg; }
      }
      if (py >= a.S) { py -= a.S; ox += imgjx; og += imgjg; }
    } else if (stepped != 0) {
      pixel_from_index(p);
    }
    tabx[slot][t] = ox;
    tabg[slot][t] = og;
  };

  const char* xbase = reinterpret_cast<const char*>(a.x);
  const char* gbase = reinterpret_cast<const char*>(a.g);
  f32x4 rx[NX], rg[NG];
  // global -> registers for the chunk the walk stands on (tables in `slot` when S % 32 != 0)
  // (S % 32 != 0: the lane's byte offsets come out of the table a phase BEFORE the loads that use them -- fetch_offsets after the
  // barrier that publishes the table, gload at the top of the next iteration -- so that no LDS round trip sits in front of the loads)
  uint32_t nox[NX], nog[NG];
#pragma unroll
  for (int i = 0; i < NX; ++i) nox[i] = 0u;
#pragma unroll
  for (int i = 0; i < NG; ++i) nog[i] = 0u;
  auto fetch_offsets = [&](int slot) {
    if (affine) return;
#pragma unroll
    for (int i = 0; i < NX; ++i) nox[i] = tabx[slot][xpix + XPS * i] + xoff[i];
#pragma unroll
    for (int i = 0; i < NG; ++i) nog[i] = tabg[slot][gpix + GPS * i] + goff[i];
  };
  auto gload = [&]() {
    if (affine) {
      const char* xb = xbase + (size_t)(uint32_t)(((w.b * Sxp + w.y + a.Px - a.pad) * Sxp + w.x0 + a.Px - a.pad) * a.ld_x) * 4u;
      const char* gb = gbase + (size_t)(uint32_t)(((w.b * Sgp + w.y + a.Pg) * Sgp + w.x0 + a.Pg) * a.ld_g) * 4u;
#pragma unroll
      for (int i = 0; i < NX; ++i) { uint32_t o = xoff[i]; asm volatile("" : "+v"(o)); rx[i] = *reinterpret_cast<const f32x4*>(xb + o); }
#pragma unroll
      for (int i = 0; i < NG; ++i) { uint32_t o = goff[i]; asm volatile("" : "+v"(o)); rg[i] = *reinterpret_cast<const f32x4*>(gb + o); }
    } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) rx[i] = *reinterpret_cast<const f32x4*>(xbase + nox[i]);
#pragma unroll
      for (int i = 0; i < NG; ++i) rg[i] = *reinterpret_cast<const f32x4*>(gbase + nog[i]);
    }
  };
  auto lstore = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NX; ++i) *reinterpret_cast<f32x4*>(&Xs[(xpix + XPS * i) * LDX + (t % XQ) * 4]) = rx[i];
    if (!AFF && chunk == clast) {               // pixels past the end were loaded from a clamped address: their G rows are zero (S % 32 == 0: no such chunk)
#pragma unroll
      for (int i = 0; i < NG; ++i)
        *reinterpret_cast<f32x4*>(&Gs[(gpix + GPS * i) * LDG + (t % GQ) * 4]) =
            chunk * BP + gpix + GPS * i < a.M ? rg[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < NG; ++i) *reinterpret_cast<f32x4*>(&Gs[(gpix + GPS * i) * LDG + (t % GQ) * 4]) = rg[i];
    }
  };

  int cA = w.c;                                  // chunk in LDS / being multiplied
  if (cA < cend) {
    fill_tables(0, 0);
    __syncthreads();
    fetch_offsets(0);
    gload();
    int stepped = w.advance();                   // the walk now stands on B, the chunk to prefetch next
    int cB = w.c;
    fill_tables(1, stepped);
    lstore(cA);
    __syncthreads();
    fetch_offsets(1);
    const int xr = wr * WTR + li, gc = wc * WTO + li;
    int quart = -1;
    for (int it = 0; cA < cend; ++it) {
      if (a.prio) set_prio_by_progress(cA - cbeg, cend - cbeg, quart, a.prio);
#ifdef DRS_DEV
      if (a.ablate == 2) fetch_offsets((it + 1) & 1);      // A/B arm: the table reads right in front of the loads, as before r04
#endif
      if (cB < cend) gload();                     // B: its offsets were fetched after the last barrier
      stepped = w.advance();                      // ... and on C, whose table goes into the slot read two barriers ago
      fill_tables(it & 1, stepped);
      {
        // fragments of pixel pair s+1 are read before the MFMAs of pair s are issued (two register sets used in turn)
        float af[2][TMr], bf[2][TNo];
#pragma unroll
        for (int mi = 0; mi < TMr; ++mi) af[0][mi] = Xs[h * LDX + xr + mi * 32];
#pragma unroll
        for (int ni = 0; ni < TNo; ++ni) bf[0][ni] = Gs[h * LDG + gc + ni * 32];
#pragma unroll
        for (int s = 0; s < BP / 2; ++s) {
          if (s + 1 < BP / 2) {
#pragma unroll
            for (int mi = 0; mi < TMr; ++mi) af[(s + 1) & 1][mi] = Xs[(2 * s + 2 + h) * LDX + xr + mi * 32];
#pragma unroll
            for (int ni = 0; ni < TNo; ++ni) bf[(s + 1) & 1][ni] = Gs[(2 * s + 2 + h) * LDG + gc + ni * 32];
            __builtin_amdgcn_sched_group_barrier(0x100, TMr + TNo, 0);
          }
#pragma unroll
          for (int mi = 0; mi < TMr; ++mi)
#pragma unroll
            for (int ni = 0; ni < TNo; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][mi], bf[s & 1][ni], acc[mi][ni], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, TMr * TNo, 0);
        }
      }
      __syncthreads();
      fetch_offsets(it & 1);                      // C's table is published; the reads land under the stores below
      if (cB < cend) { lstore(cB); __syncthreads(); }
      cA = cB;
      cB = w.c;
    }
  }
  const size_t rows_total = (size_t)a.k * a.k * a.Cin;
  float* dst = a.slab + ((size_t)split * rows_total + R0) * a.Cout + o0;
#pragma unroll
  for (int mi = 0; mi < TMr; ++mi)
#pragma unroll
    for (int ni = 0; ni < TNo; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * WTR + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = wc * WTO + ni * 32 + li;
        if (R0 + row < rows_all) dst[(size_t)row * a.Cout + col] = acc[mi][ni][r];
      }
#ifdef DRS_DEV
  if (a.trace && t == 0) a.trace[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// The same tile with its operands brought in by LDS-DMA (global_load_lds_dwordx4: global memory -> LDS without passing through
// vector registers and without a ds_write pass) into a DOUBLE-BUFFERED image of two 16-pixel half-chunks (the bytes of one
// 32-pixel stage): while half h is multiplied the DMA of half h+1 lands in the other stage; one barrier per half.  Measured on
// wgrad_kernel (profiles/r02/wgrad_ablation.txt): with the address arithmetic gone, the register-staged loads still cost 6-13 %
// and the LDS stores 3-4 % of the MFMA rate -- both disappear here.  A DMA wave-instruction writes 64 lanes x 16 B to consecutive
// LDS bytes, so image rows are unpadded (TR / TO floats); the ds_read_b32 fragment reads of this kernel (32 consecutive floats
// per lane group) are conflict-free on such rows.  Chunk walk, tile geometry, slab layout and the order of every sum are those
// of wgrad_kernel: results are bitwise the same.
// AFF: S % 32 == 0 (a chunk lies inside one image row: wave-uniform bases + loop-invariant lane offsets); !AFF: the table form.  Two
// instantiations rather than a run-time flag: the table form keeps a lane's next offsets in registers ACROSS the multiply phases, and
// the 128 x 192 tile has none to spare (161 VGPRs of 168; with the flag the S % 32 == 0 loop of the headline carried 3 spilled registers).
template <int TR, int TO, int MODE>
__global__ __launch_bounds__(64 * (TR >= 64 ? 2 : 1) * (TO >= 64 ? 2 : 1), 3) void wgrad_dma_kernel(const WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass has no LDS-DMA builtin; it only needs the launch stub
  constexpr int WR = TR >= 64 ? 2 : 1, WC = TO >= 64 ? 2 : 1;
  constexpr int NW = WR * WC, NT = 64 * NW;
  constexpr int WTR = TR / WR, WTO = TO / WC;
  constexpr int TMr = WTR / 32, TNo = WTO / 32;
  constexpr int BP = 32, HP = 16;             // pixels per chunk (walk / table unit) and per pipeline stage
  constexpr int XQ = TR / 4, GQ = TO / 4;     // 16-byte pieces per pixel row
  // a DMA wave-instruction moves 64 pieces = 1 KiB of the half-stage image [HP][TR] (resp. [HP][TO]) taken as one linear run
  constexpr int IX = HP * XQ / 64 / NW, IG = HP * GQ / 64 / NW;   // DMA instructions per wave and half
  static_assert(IX >= 1 && IG >= 1 && IX * NW * 64 == HP * XQ && IG * NW * 64 == HP * GQ, "tile / wave layout");
  constexpr int XSTAGE = HP * TR, STAGE = HP * (TR + TO);     // floats

  __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
  __shared__ uint32_t tabx[2][BP], tabg[2][BP];

  constexpr bool AFF = MODE == 1;              // S % 32 == 0: a chunk lies inside one image row
  constexpr bool SEG = MODE == 2;              // S >= 32, not a multiple: a chunk is at most two row segments (the table-free form, r05)
  const int t = threadIdx.x;
  // (S % 32 == 0: the wave index as a scalar -- the LDS address of every DMA piece then is one too, instead of a v_or + v_readfirstlane
  //  in front of each of the 8 DMA instructions of a chunk: the 8 launches at B = 128 15.10 -> 14.88 ms; the table form, whose loop is short
  //  of scalar registers, loses 1 % with it)
  const int lane = t & 63, wave = (AFF || SEG) ? __builtin_amdgcn_readfirstlane(t >> 6) : t >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;

  int tile, split, cbeg, cend, live_lo, live_hi;
#ifdef DRS_DEV
  if (a.trace && t == 0) {
    a.trace[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    a.trace[32768 + blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);   // HW_ID, XCC_ID
  }
#endif
  wgrad_assign(a, TR, xcd_remap(blockIdx.x, gridDim.x), tile, split, cbeg, cend, live_lo, live_hi);
  const int R0 = (tile / a.nto) * TR;
  const int o0 = (tile % a.nto) * TO;
  const int rows_all = a.k * a.k * a.Cin;
  const int Sxp = a.S + 2 * a.Px, Sgp = a.S + 2 * a.Pg;
  constexpr bool affine = AFF || SEG;         // wave-uniform chunk bases + loop-invariant lane offsets
  // DMA lane roles: instruction i of this wave moves pieces 64 (wave + NW i) + lane of the linear image: piece f belongs to pixel
  // f / XQ, 16-byte column f % XQ.  (TR is a power of two, so the X column -- and with it the filter tap of the rows this lane
  // stages -- is the same for all of a lane's instructions; TO = 192 gives every instruction its own (pixel, column) pair.)
  const int xq = lane % XQ;
  const int myR = R0 + xq * 4;
  const int myRc = myR < rows_all ? myR : 0;
  const int tap = myRc / a.Cin, c0 = myRc % a.Cin;
  const int u = tap / a.k, v = tap % a.k;
  // SEG: every lane offset carries a bias of 32 pixels' worth of bytes that the chunk bases take off again, so that the NEGATIVE jump of
  // the ragged last chunk (below) never takes a 32-bit lane offset below zero
  const uint32_t biasx = SEG ? (uint32_t)(32 * a.ld_x) * 4u : 0u, biasg = SEG ? (uint32_t)(32 * a.ld_g) * 4u : 0u;
  const uint32_t xconst = (uint32_t)((a.ablate == 1 ? 0 : (u * a.rate * Sxp + v * a.rate)) * a.ld_x + a.coff_x + c0) * 4u + biasx;
  uint32_t xoff[IX], goff[IG];
  int xpix[IX], gpix[IG];
#pragma unroll
  for (int i = 0; i < IX; ++i) {
    xpix[i] = (64 * (wave + NW * i) + lane) / XQ;
    xoff[i] = xconst + (affine ? (uint32_t)(xpix[i] * a.ld_x) * 4u : 0u);
  }
#pragma unroll
  for (int i = 0; i < IG; ++i) {
    const int f = 64 * (wave + NW * i) + lane;
    gpix[i] = f / GQ;
    goff[i] = (uint32_t)(a.coff_g + o0 + (f % GQ) * 4) * 4u + (affine ? (uint32_t)(gpix[i] * a.ld_g) * 4u : 0u) + biasg;
  }

  f32x16 acc[TMr][TNo];
#pragma unroll
  for (int mi = 0; mi < TMr; ++mi)
#pragma unroll
    for (int ni = 0; ni < TNo; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nchunks_total = (a.M + BP - 1) / BP;
  const int clast = (a.M & 31) ? nchunks_total - 1 : -1;
  ChunkWalk w;
  w.init(cbeg, a.S, a.rcpS, a.rcpSS, live_lo, live_hi);

  // The thread carries the BYTE OFFSETS of its pixel in the two slabs along with (py, px): +32 pixels is three additions, a row
  // wrap adds the slab's row jump, an image wrap its image jump -- compares, selects and additions only; the integer multiplies of
  // the offset formula (quarter rate) are paid once per jump.  (r04: the fills cost 3.5-5 % of a launch at S % 32 != 0 -- one wave
  // of four does them and the workgroup moves at its pace.)
  int py = 0, px = 0;
  uint32_t ox = 0, og = 0;
  const uint32_t stepx = (uint32_t)(32 * a.ld_x) * 4u, stepg = (uint32_t)(32 * a.ld_g) * 4u;
  const uint32_t rowjx = (uint32_t)((Sxp - a.S) * a.ld_x) * 4u, rowjg = (uint32_t)((Sgp - a.S) * a.ld_g) * 4u;
  const uint32_t imgjx = (uint32_t)(Sxp * (Sxp - a.S) * a.ld_x) * 4u, imgjg = (uint32_t)(Sgp * (Sgp - a.S) * a.ld_g) * 4u;
  auto pixel_from_index = [&](int p) {
    int pb, rem;
    divmod24(p < a.M ? p : a.M - 1, w.S2, a.rcpSS, pb, rem);
    divmod24(rem, a.S, a.rcpS, py, px);
    ox = (uint32_t)(((pb * Sxp + py + a.Px - a.pad) * Sxp + px + a.Px - a.pad) * a.ld_x) * 4u;
    og = (uint32_t)(((pb * Sgp + py + a.Pg) * Sgp + px + a.Pg) * a.ld_g) * 4u;
  };
  if (!affine && t < BP) pixel_from_index(w.c * BP + t);
  auto fill_tables = [&](int slot, int stepped) {
    if (affine || t >= BP || w.c >= cend) return;
    const int p = w.c * BP + t;
    if (stepped == 1 && p < a.M && a.S >= 11) {
      px += 32; ox += stepx; og += stepg;
      if (a.S >= 32) {                           // (uniform) at most one row wrap per 32 pixels
        if (px >= a.S) { px -= a.S; ++py; ox += rowjx; og += rowjg; }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) if (px >= a.S) { px -= a.S; ++py; ox += rowjx; og += rowjg; }
      }
      if (py >= a.S) { py -= a.S; ox += imgjx; og += imgjg; }
    } else if (stepped != 0) {
      pixel_from_index(p);
    }
    tabx[slot][t] = ox;
    tabg[slot][t] = og;
  };

  const char* xbase = reinterpret_cast<const char*>(a.x);
  const char* gbase = reinterpret_cast<const char*>(a.g);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // scalar bases of the chunk the walk stands on (S % 32 == 0), taken when the walk is there
  // SEG: the chunk the walk stands on is the pixels x0 .. of image row y and, when the row ends inside it (after brk = S - x0 < 32
  // pixels), the start of the next row -- in the slab `jump` bytes further on than linear addressing says (the two halos between the
  // rows; after an image's last row also the halo rows between the images).  The ragged LAST chunk of the tensor ends with the last
  // image's last row: its pixels past the end take a jump BACK by 32 pixels instead (any address inside the slab will do: their G rows
  // are zeroed, zero_tail) -- the same mechanism, one more value of the jump.
  struct Seg { int brk; int32_t jx, jg; };
  auto chunk_bases = [&](const char*& xb, const char*& gb, Seg& sg) {
    xb = xbase + (size_t)(uint32_t)(((w.b * Sxp + w.y + a.Px - a.pad) * Sxp + w.x0 + a.Px - a.pad) * a.ld_x) * 4u;
    gb = gbase + (size_t)(uint32_t)(((w.b * Sgp + w.y + a.Pg) * Sgp + w.x0 + a.Pg) * a.ld_g) * 4u;
    if (SEG) {
      xb -= biasx; gb -= biasg;
      sg.brk = a.S - w.x0;
      const bool last = w.c == clast, img_end = w.y == a.S - 1;
      sg.jx = last ? -(int32_t)biasx : (int32_t)(rowjx + (img_end ? imgjx : 0u));
      sg.jg = last ? -(int32_t)biasg : (int32_t)(rowjg + (img_end ? imgjg : 0u));
    }
  };
  // S % 32 != 0: a lane's byte offsets for the two halves of a chunk, table entry + lane constant, fetched into registers a phase
  // BEFORE the DMA that uses them is issued (the table read used to sit in front of every issue: an LDS round trip on the critical
  // path of each half, twice per chunk)
  struct LaneOffsets { uint32_t x[2][IX], g[2][IG]; };
  auto fetch_offsets = [&](int slot, LaneOffsets& o) {
    if (affine) return;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < IX; ++i) o.x[half][i] = tabx[slot][half * HP + xpix[i]] + xoff[i];
#pragma unroll
      for (int i = 0; i < IG; ++i) o.g[half][i] = tabg[slot][half * HP + gpix[i]] + goff[i];
    }
  };
  // DMA of half `half` of a chunk (bases xb / gb, or the lane offsets `o`) into LDS stage `stage`
  auto issue = [&](const char* xb, const char* gb, const Seg& seg, const LaneOffsets& o, int half, int stage) {
    float* sx = lds + stage * STAGE;
    float* sg = sx + XSTAGE;
    if (affine) {
      const char* xh = xb + (size_t)(uint32_t)(half * HP * a.ld_x) * 4u;
      const char* gh = gb + (size_t)(uint32_t)(half * HP * a.ld_g) * 4u;
      int inner = HP;                            // SEG: first pixel of this half that lies past the row break (HP: none does)
      if (SEG) {
        const int hb = seg.brk - half * HP;
        if (hb <= 0) { xh += (ptrdiff_t)seg.jx; gh += (ptrdiff_t)seg.jg; }      // the whole half lies past it: a scalar matter
        else if (hb < HP) inner = hb;
      }
      if (SEG && inner < HP) {                   // (wave-uniform) the break falls inside this half: a compare per lane and piece
#pragma unroll
        for (int i = 0; i < IX; ++i) {
          uint32_t v = xoff[i] + (xpix[i] >= inner ? (uint32_t)seg.jx : 0u);
          __builtin_amdgcn_global_load_lds(xh + v, (lds_ptr)(sx + (wave + NW * i) * 256), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IG; ++i) {
          uint32_t v = goff[i] + (gpix[i] >= inner ? (uint32_t)seg.jg : 0u);
          __builtin_amdgcn_global_load_lds(gh + v, (lds_ptr)(sg + (wave + NW * i) * 256), 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < IX; ++i) {
          uint32_t v = xoff[i]; asm volatile("" : "+v"(v));
          __builtin_amdgcn_global_load_lds(xh + v, (lds_ptr)(sx + (wave + NW * i) * 256), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IG; ++i) {
          uint32_t v = goff[i]; asm volatile("" : "+v"(v));
          __builtin_amdgcn_global_load_lds(gh + v, (lds_ptr)(sg + (wave + NW * i) * 256), 16, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < IX; ++i) __builtin_amdgcn_global_load_lds(xbase + o.x[half][i], (lds_ptr)(sx + (wave + NW * i) * 256), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < IG; ++i) __builtin_amdgcn_global_load_lds(gbase + o.g[half][i], (lds_ptr)(sg + (wave + NW * i) * 256), 16, 0, 0);
    }
  };
  // pixels past the end (only in the last chunk) were fetched from a clamped address: their G rows must read as zero.  Called AFTER
  // the barrier that follows every wave's `s_waitcnt vmcnt(0)`: a thread zeroes rows that ANOTHER wave's DMA wrote, so all of the
  // stage's DMA must have landed first (zeroing after only this wave's own wait let a late DMA put the clamped row back on top of
  // the zeros -- harmless-looking alone, a run-to-run difference as soon as another kernel shared the chip: tools/soak.py), and a
  // second barrier publishes the zeros before anybody multiplies.  `chunk` is uniform over the workgroup, so are the barriers.
  auto zero_tail = [&](int chunk, int half, int stage) {
    if (AFF || chunk != clast) return;          // (S % 32 == 0: the pixel count is a multiple of 32, there is no such chunk)
    float* sg = lds + stage * STAGE + XSTAGE;
    for (int e = t; e < HP * TO; e += NT)
      if (chunk * BP + half * HP + e / TO >= a.M) sg[e] = 0.f;
    __syncthreads();
  };
  // Wave (wr, wc) multiplies the 32-row groups wr, wr + WR, ... and the 32-column groups wc, wc + WC, ...: the fragments of one MFMA
  // step then lie 64 floats (256 B) apart in an LDS row, and a pair of them, every step of a stage and both stages are within the
  // reach of ds_read2st64_b32's immediate offsets from ONE base register per operand (with groups 2 wr, 2 wr + 1 the pair was 128 B
  // apart: 16 hoisted base registers for stage 0 and 16 v_add per half chunk for stage 1).
  const int xr = wr * 32 + li, gc = wc * 32 + li;
  constexpr int XG = 32 * WR, GG = 32 * WC;      // floats between a wave's consecutive row / column groups
  auto compute = [&](int stage) {
    const float* Xs = lds + stage * STAGE;
    const float* Gs = Xs + XSTAGE;
    float af[2][TMr], bf[2][TNo];
#pragma unroll
    for (int mi = 0; mi < TMr; ++mi) af[0][mi] = Xs[h * TR + xr + mi * XG];
#pragma unroll
    for (int ni = 0; ni < TNo; ++ni) bf[0][ni] = Gs[h * TO + gc + ni * GG];
#pragma unroll
    for (int s = 0; s < HP / 2; ++s) {
      if (s + 1 < HP / 2) {
#pragma unroll
        for (int mi = 0; mi < TMr; ++mi) af[(s + 1) & 1][mi] = Xs[(2 * s + 2 + h) * TR + xr + mi * XG];
#pragma unroll
        for (int ni = 0; ni < TNo; ++ni) bf[(s + 1) & 1][ni] = Gs[(2 * s + 2 + h) * TO + gc + ni * GG];
        __builtin_amdgcn_sched_group_barrier(0x100, TMr + TNo, 0);
      }
#pragma unroll
      for (int mi = 0; mi < TMr; ++mi)
#pragma unroll
        for (int ni = 0; ni < TNo; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][mi], bf[s & 1][ni], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, TMr * TNo, 0);
    }
  };

  int cA = w.c;
  if (cA < cend) {
    const char *xbA, *gbA, *xbB = xbase, *gbB = gbase;
    LaneOffsets oA, oB;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < IX; ++i) oA.x[half][i] = oB.x[half][i] = 0u;
#pragma unroll
      for (int i = 0; i < IG; ++i) oA.g[half][i] = oB.g[half][i] = 0u;
    }
    Seg sgA = {32, 0, 0}, sgB = {32, 0, 0};
    chunk_bases(xbA, gbA, sgA);
    fill_tables(0, 0);
    if (!affine) __syncthreads();
    fetch_offsets(0, oA);
    issue(xbA, gbA, sgA, oA, 0, 0);              // chunk A, first half -> stage 0
    int stepped = w.advance();                   // the walk now stands on B
    int cB = w.c;
    chunk_bases(xbB, gbB, sgB);
    fill_tables(1, stepped);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA has landed; the barrier then publishes everybody's
    __syncthreads();
    zero_tail(cA, 0, 0);
    int quart = -1;
    for (int it = 0; cA < cend; ++it) {
      if (a.prio) set_prio_by_progress(cA - cbeg, cend - cbeg, quart, a.prio);
#ifdef DRS_DEV
      const bool late = a.ablate == 2;           // A/B arm: the table reads where they used to be, right in front of each issue
      if (late) fetch_offsets(it & 1, oA); else
#endif
      fetch_offsets((it + 1) & 1, oB);           // B's table was published by the last barrier; its entries are used after compute(0)
      issue(xbA, gbA, sgA, oA, 1, 1);            // second half of A -> stage 1, lands while stage 0 is multiplied
      compute(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      zero_tail(cA, 1, 1);
#ifdef DRS_DEV
      if (late) fetch_offsets((it + 1) & 1, oB);
#endif
      if (cB < cend) issue(xbB, gbB, sgB, oB, 0, 0);  // first half of B -> stage 0 (every wave is done with it)
      stepped = w.advance();                     // ... and on C, whose table goes into the slot A's table was in
      compute(1);
      fill_tables(it & 1, stepped);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (cB < cend) zero_tail(cB, 0, 0);
      cA = cB; xbA = xbB; gbA = gbB; sgA = sgB;
      oA = oB;
      cB = w.c;
      chunk_bases(xbB, gbB, sgB);
    }
  }
  const size_t rows_total = (size_t)a.k * a.k * a.Cin;
  float* dst = a.slab + ((size_t)split * rows_total + R0) * a.Cout + o0;
#pragma unroll
  for (int mi = 0; mi < TMr; ++mi)
#pragma unroll
    for (int ni = 0; ni < TNo; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mi * XG + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = ni * GG + wc * 32 + li;
        if (R0 + row < rows_all) dst[(size_t)row * a.Cout + col] = acc[mi][ni][r];
      }
#ifdef DRS_DEV
  if (a.trace && t == 0) a.trace[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
#endif
}

// grad[tap][c][o] = sum over splits (fixed order) of slab[split][tap][c (of cin_pad)][o], c < cin_real
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int nsplit_all, int taps,
                                    int cin_pad, int cin_real, int cout, int tr, const WgradPlan plan) {
  const int n = taps * cin_real * cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int o = i % cout;
    const int rc = i / cout;
    const int c = rc % cin_real, tap = rc / cin_real;
    const size_t src = ((size_t)tap * cin_pad + c) * cout + o;
    const int nsplit = plan.nclass ? plan.n[plan.cls[(tap * cin_pad + c) / tr]] : nsplit_all;     // the splits this row's tile wrote
    const size_t stride = (size_t)taps * cin_pad * cout;
    // the sum runs in split order (fixed, so a step is reproducible); the loads of 8 splits are in flight together
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = slab[(size_t)(k + j) * stride + src];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < nsplit; ++k) s += slab[(size_t)k * stride + src];
    grad[i] = s;
  }
}

// Wt[k-1-u][k-1-v][o][c] = W[u][v][c][o]      (filter for the input-gradient pass)
__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int k, int cin, int cout) {
  const int n = k * k * cin * cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i % cin;
    int rest = i / cin;
    const int o = rest % cout;
    rest /= cout;
    const int v = rest % k, u = rest / k;
    wt[i] = w[(((k - 1 - u) * k + (k - 1 - v)) * cin + c) * cout + o];
  }
}

// What a training step needs done between its forward and its backward pass, in ONE launch (the step engine; at a per-rank batch
// of 16 small patches a step is ~100 launches of ~15 us and every launch counts): the flipped / transposed filter of every layer
// with an input gradient (blockIdx.y = entry) and two zero fills (the confusion matrix; the conv-bias gradients).
__global__ void step_prep_kernel(const StepPrepArgs a) {
  const int e = blockIdx.y;
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (e < a.n) {
    const int k = a.k[e], cin = a.cin[e], cout = a.cout[e];
    const float* __restrict__ w = a.w[e];
    float* __restrict__ wt = a.wt[e];
    const int n = k * k * cin * cout;
    for (int i = i0; i < n; i += stride) {
      const int c = i % cin;
      int rest = i / cin;
      const int o = rest % cout;
      rest /= cout;
      const int v = rest % k, u = rest / k;
      wt[i] = w[(((k - 1 - u) * k + (k - 1 - v)) * cin + c) * cout + o];
    }
  } else if (e == a.n) {
    for (int i = i0; i < a.nz0; i += stride) a.z0[i] = 0u;
  } else {
    for (int i = i0; i < a.nz1; i += stride) a.z1[i] = 0.f;
  }
}

// Wp[tap][c < cin_pad][o] = c < cin ? W[tap][c][o] : 0
__global__ void pad_cin_kernel(const float* __restrict__ w, float* __restrict__ wp, int taps, int cin, int cin_pad, int cout) {
  const int n = taps * cin_pad * cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int o = i % cout;
    const int rc = i / cout;
    const int c = rc % cin_pad, tap = rc / cin_pad;
    wp[i] = c < cin ? w[((size_t)tap * cin + c) * cout + o] : 0.f;
  }
}

int g_conv_variant = -1;     // development switch (drs_debug_conv_variant): 0 = register-staged tiles, 1 = LDS-DMA double-buffered halves, -1 = per tile

#ifdef DRS_DEV
unsigned long long* g_conv_trace = nullptr;      // development build (drs_debug_conv_trace): per-workgroup (start, end) stamps of the next launches
#endif
int g_conv_lpt = 1;          // development switch (drs_debug_conv_lpt): 1 = plain launches start their full tiles first, the halo-skipping ones last
int g_conv_splitk = -1;      // development switch (drs_debug_conv_splitk): -1 = stream-K by the rule below, 0 = never, n >= 1: n workgroups
int g_conv_sk_order = 1;     // development switch (drs_debug_conv_sk_order): a hybrid workgroup does 0 its range first, 1 its whole tiles first, 2 alternating by slot
int g_conv_prio = -1;        // development switch (drs_debug_conv_prio): wave priority by remaining work in the forward / input-gradient kernel: -1 = by the rule in launch_conv_dma, 0 = never, 1 = always

constexpr int SK_MAX_TILES = 4096;     // from here on the tiles fill the chip many times over (and the all-halo tap rows are skipped instead)

// compute units of the CURRENT device (MI355X: 256), asked once per device; the stream-K geometry and the filter-gradient workgroup
// counts are "a whole number of workgroups per CU", so they follow the part instead of a literal.  (A host without a GPU -- the CPU
// tests that read the net tables back -- gets 256.  The query starts the HIP runtime on whatever device is current: hosts set their
// device before they create a net, as engine.py does.)
int cu_count() {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    hipDeviceProp_t p;
    cached[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 1) ? p.multiProcessorCount : 256;
  }
  return cached[dev];
}
inline int sk_max_w() { return 3 * cu_count(); }      // most workgroups of a stream-K launch: 3 per CU

// Stream-K launch geometry of the forward / input-gradient pass: G == 0 = one workgroup per tile.
// A launch of few tiles leaves CUs idle (B = 16, S = 25: 79 M tiles for 256 CUs) and ends with its longest K loop; a launch of a
// few hundred tiles ends with a round that is nearly empty (B = 16, S = 65: 1058 tiles for 1024 places: 0.65 of the rate the same
// kernel reaches at S = 75).  Cutting the K-steps of all tiles into W equal ranges, W a whole number of workgroups per CU, gives
// every CU the same work whatever the tile count; the price is the partial-sum slab of the tiles that are cut (at most two
// pieces per workgroup) and the fix-up launch.  Per-size table: profiles/r03/small_m.
// Hybrid (r05): a launch of MORE tiles than the G = 3 x CUs workgroups the chip holds of this form runs its full rounds as whole
// tiles (tile T + w + j G on workgroup w: no piece, the fast epilogue) and cuts only the remaining T = tiles mod G tiles into W <= G
// ranges -- every workgroup still does the same number of K-steps, the slab traffic and the fix-up launch shrink from `tiles` to T tiles.
struct SkGeom { int G, W, T; };      // workgroups of the launch, ranges the first T tiles are cut into
int g_conv_hybrid = 1;       // development switch (drs_debug_conv_hybrid): 1 = whole tiles for the full rounds, 0 = every tile cut (r03)
SkGeom sk_geometry(int tiles, int nks, int bn, size_t ws_floats) {
  const SkGeom plain = {0, 0, 0};
  if (g_conv_splitk == 0 || tiles >= SK_MAX_TILES || nks < 2) return plain;
  const long long NCU = cu_count();
  long long cap = (long long)(ws_floats / (2ull * 128 * (size_t)bn));
  if (cap > sk_max_w()) cap = sk_max_w();
  if (cap < 1) return plain;
  const int MINU = 12;                                         // K-steps a range should at least have (prologue + epilogue cost ~2)
  const int occ = 3;                                           // workgroups a CU holds of the stream-K form (136 / 168 VGPRs)
  long long G, W, T = tiles;
  if (g_conv_splitk > 0) {
    G = W = g_conv_splitk;
  } else {
    // one workgroup per tile loads the busiest CU with ceil(tiles / 256) tiles: where that is within 7 % of the mean the plain
    // launch is the faster one (no slab, no fix-up, 4 instead of 3 workgroups per CU); measured at B = 16, S = 25 .. 85
    // (profiles/r03/ab_streamk_b16.log): plain 0.85-0.87 of the fp32 roof at a perfect fit and proportionally less otherwise,
    // stream-K 0.78-0.85 at every tile count
    const long long per = (tiles + NCU - 1) / NCU;
    if ((double)tiles >= 0.93 * (double)NCU * (double)per) return plain;
    const long long U = (long long)tiles * nks;
    long long per_cu = U / (NCU * MINU);
    per_cu = per_cu < 1 ? 1 : (per_cu > occ ? occ : per_cu);
    G = W = NCU * per_cu;
    if (U < NCU * MINU) G = W = U / MINU > 0 ? U / MINU : 1;
  }
  if (g_conv_hybrid && tiles > G) {
    T = tiles % G;
    if (T == 0) return plain;                                  // (whole rounds of the 3-per-CU form: the 4-per-CU plain form does them better)
    // (every workgroup takes a range however short: with fewer, longer ranges the workgroups that have one run that much longer than the
    //  others -- B = 16, S = 85, conv3: 362 ranges of 12 K-steps on top of 32-step tiles, +8 % against 768 ranges of 5-6)
    W = G;
  }
  if (W > T * nks) W = T * nks;
  if (W > G) W = G;
  // the cut must follow from the SHAPE alone: a workspace too small for it gets the plain launch, not a smaller cut (the sums of a
  // convolution would otherwise associate differently from one caller's workspace to another's)
  if (W > cap) {
    if (g_conv_splitk <= 0) return plain;
    W = cap; if (G < W || T == tiles) G = W;
  }
  return SkGeom{(int)G, (int)W, (int)T};
}

// parameters of the "full tiles first" launch order (lpt_tile): only for plain launches whose M tiles are whole image rows of whole
// patches (S * S and the tile height multiples of each other's parts: S = 32, 64, 128 ...) and whose XCD chunks are whole patches;
// everything else keeps the natural order.  Where the order applies the all-halo tap rows are skipped at ANY tile count: with the
// natural order skipping only paid from 4096 workgroups (+3 % time at 2048, +12 % at 1024: drs_common.hpp); with the full tiles
// spread over the CUs first and the short ones after them it pays everywhere it was tried (in-process A/B against multiplying every
// tap, forward + input gradient of the 7 layers: B = 64 -5.3 %, B = 32 -3.4 %, B = 16 -2 % (conv8 -8 %), B = 8 +-0).
void conv_lpt_setup(ConvArgs& a, int BM, int mt, int nt, int W) {
  a.lpt_T = 0; a.lpt_ta = a.lpt_tb = a.lpt_P = 0;
  if (W || !g_conv_lpt || drs_g_skip_halo_taps == 0) return;
  const int S2 = a.S * a.S, T = S2 % BM == 0 ? S2 / BM : 0, nblk = mt * nt;
  if (T < 2 || BM % a.S != 0 || nblk % 8 != 0 || (nblk / 8) % (T * nt) != 0) return;
  const int rows = BM / a.S;                                   // image rows per tile
  auto full = [&](int t) {                                     // live_tap_rows' rule for the tile of rows [t rows, (t + 1) rows)
    const int y0 = t * rows, y1 = y0 + rows - 1;
    const int lo = a.pad - y1 > 0 ? (a.pad - y1 + a.rate - 1) / a.rate : 0;
    int hi = (a.S - 1 - y0 + a.pad) / a.rate + 1;
    hi = hi < a.k ? hi : a.k;
    return lo == 0 && hi == a.k;
  };
  int ta = 0, tb = T;
  while (ta < T && !full(ta)) ++ta;
  while (tb > ta && !full(tb - 1)) --tb;
  bool ok = ta < tb;
  for (int t = ta; ok && t < tb; ++t) ok = full(t);
  if (ok && (ta > 0 || tb < T)) { a.lpt_T = T; a.lpt_ta = ta; a.lpt_tb = tb; a.lpt_P = nblk / 8 / (T * nt); a.skip_halo = 1; }
}

template <int BM, int BN, int WM, int WN>
int launch_conv_dma(ConvArgs& a, float* ws, size_t ws_floats, hipStream_t st) {
  const int mt = (a.M + BM - 1) / BM, nt = a.Cout / BN;
  const int nks = a.k * a.k * (a.Cin / BK);
  const SkGeom g = ws && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 ? sk_geometry(mt * nt, nks, BN, ws_floats) : SkGeom{0, 0, 0};   // (16-byte piece accesses)
  const int W = g.W;
  a.sk_W = W; a.sk_nks = nks; a.sk_U = g.T * nks; a.sk_slab = ws;
  a.sk_T = g.T; a.sk_tiles = mt * nt; a.sk_G = g.G; a.sk_order = g_conv_sk_order;
  // Wave priority by remaining work (set_prio_by_progress) in the stream-K / hybrid launches: their equal-length workgroups all start
  // together and the SIMD arbiter lets the oldest run ahead, so the last of a CU finishes alone.  In-process A/B over the 14 forward /
  // input-gradient launches at B = 16 (tools/ab_conv_sched.py, profiles/r05/conv_schedule_ab.txt): S = 85 7.18 -> 7.04 ms, 65: 4.38 ->
  // 4.34, 55 -1 %, 35 -1 %, 25 +0.7 % (ranges of ~15 K-steps: nothing to catch up on) -- so from 20 K-steps per workgroup.  NOT in
  // plain launches: -0.6 % on the one-round ones of B = 16, S = 64, but +2.4 % at B = 32 and B = 128 and +18 % on a 192-wide tile.
  a.prio = g_conv_prio >= 0 ? g_conv_prio : (g.G && (long long)mt * nt * nks >= 20LL * g.G ? 1 : 0);
  // a launch of the chain the step waits for, beside a filter gradient on a stream of its own (engine.hip sets drs_tl_chain around its
  // two-stream backward pass): the whole launch at the top priority, no lowering by progress
  if (g_conv_prio < 0 && drs_chain_level(drs_tl_chain, drs_g_chain_mode) >= 1) a.prio = 3;
  conv_lpt_setup(a, BM, mt, nt, g.G);
  if (g.G) DRS_LAUNCH((conv_dma_kernel<BM, BN, WM, WN, true>), dim3(g.G), dim3(256), 0, st, a);
  else DRS_LAUNCH((conv_dma_kernel<BM, BN, WM, WN, false>), dim3(mt * nt), dim3(256), 0, st, a);
  int rc = DRS_LAUNCH_CHECK();
  if (rc || !g.G) return rc;
  if (a.sk_U % W == 0 && (a.sk_U / W) % nks == 0) return rc;        // every range is a whole number of tiles: nothing to add up
  DRS_LAUNCH((conv_sk_fixup_kernel<BM, BN, WM, WN>), dim3(g.T), dim3(256), 0, st, a);
  return DRS_LAUNCH_CHECK();
}

template <int BM, int BN, int WM, int WN>
int launch_conv(ConvArgs& a, float* ws, size_t ws_floats, hipStream_t st) {
  const int mt = (a.M + BM - 1) / BM, nt = a.Cout / BN;
  if (a.Cin < BK) DRS_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, true>), dim3(mt * nt), dim3(256), 0, st, a);
  else if constexpr (BM == 128) {
    const bool dma = g_conv_variant != 0;       // in-process A/B (tools/ab_conv.py): the DMA form is 1-6 % faster on every Dilated8Pooling shape at B = 128 and 4 % at B = 16
    if (dma) return launch_conv_dma<BM, BN, WM, WN>(a, ws, ws_floats, st);
    DRS_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN>), dim3(mt * nt), dim3(256), 0, st, a);
  } else DRS_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN>), dim3(mt * nt), dim3(256), 0, st, a);
  return DRS_LAUNCH_CHECK();
}

int g_wgrad_variant = -1;    // development switch (drs_debug_wgrad_variant): 0 = register-staged tiles, 1 = LDS-DMA double-buffered halves, -1 = per tile

// the S % 32 == 0 form of wgrad_dma_kernel (development build, ablate 5: at any S -- a timing experiment with WRONG sums)
inline bool wgrad_affine(const WgradArgs& a) { return (a.S & 31) == 0 || a.ablate == 5; }
// the table-free form of wgrad_dma_kernel for the other sides (r05): a 32-pixel chunk of a side >= 32 is at most two row segments.  The
// lane offsets of that form are 32-bit byte offsets with a bias of 32 pixels: both slabs must leave that much room below 4 GiB
int g_wgrad_seg = 1;         // development switch (drs_debug_wgrad_seg): 0 = the table form for every side that is not a multiple of 32
inline bool wgrad_segments(const WgradArgs& a) {
  const long long Sxp = a.S + 2 * a.Px, Sgp = a.S + 2 * a.Pg, B = a.M / ((long long)a.S * a.S);
  const long long xb = (B * Sxp * Sxp + 64) * a.ld_x * 4, gb = (B * Sgp * Sgp + 64) * a.ld_g * 4;
  return g_wgrad_seg && a.S >= 32 && !wgrad_affine(a) && xb < (1LL << 32) && gb < (1LL << 32);
}

template <int TR, int TO>
int launch_wgrad(const WgradArgs& a, int nwg, hipStream_t st) {
  constexpr int NT = 64 * (TR >= 64 ? 2 : 1) * (TO >= 64 ? 2 : 1);
  // in-process A/B at B = 128 (profiles/r02/wgrad_ablation.txt): the LDS-DMA form wins 3-6 % on the 128-wide tiles and loses 7-10 % on the
  // 64-wide ones (Cout = 64, 192), so each layer takes the form that is faster for its tile
  // (r04, after the loop clean-ups of the LDS-DMA form: at S % 32 == 0 it now also wins on the 64-wide tiles -- conv2 at B = 128 0.830 -> 0.806 ms,
  //  at B = 16 0.140 -> 0.132; the table form still loses 2 % there)
  // (r05, row-segment addressing: on the 64-wide tiles it wins at the per-rank batches -- conv1 / conv2 at B = 16, S = 45 / 65 / 85: -4 .. -5 % --
  //  and is a wash at B = 128 (S = 63 +1 %, 65 -4 %, 85 +2.5 %): taken below 2^18 pixels; profiles/r05/wgrad_segments_ab.txt)
  const bool dma = g_wgrad_variant < 0 ? (TO == 128 || (TO == 64 && (wgrad_affine(a) || (wgrad_segments(a) && a.M < (1 << 18))))) : g_wgrad_variant == 1;
  if (!dma && wgrad_affine(a)) DRS_LAUNCH((wgrad_kernel<TR, TO, true>), dim3(nwg), dim3(NT), 0, st, a);
  else if (!dma) DRS_LAUNCH((wgrad_kernel<TR, TO, false>), dim3(nwg), dim3(NT), 0, st, a);
  else if (wgrad_affine(a)) DRS_LAUNCH((wgrad_dma_kernel<TR, TO, 1>), dim3(nwg), dim3(NT), 0, st, a);
  else if (wgrad_segments(a)) DRS_LAUNCH((wgrad_dma_kernel<TR, TO, 2>), dim3(nwg), dim3(NT), 0, st, a);
  else DRS_LAUNCH((wgrad_dma_kernel<TR, TO, 0>), dim3(nwg), dim3(NT), 0, st, a);
  return DRS_LAUNCH_CHECK();
}

template <int TR, int TO>
int launch_wgrad_dma_only(const WgradArgs& a, int nwg, hipStream_t st) {
  if (wgrad_affine(a)) DRS_LAUNCH((wgrad_dma_kernel<TR, TO, 1>), dim3(nwg), dim3(256), 0, st, a);
  else if (wgrad_segments(a)) DRS_LAUNCH((wgrad_dma_kernel<TR, TO, 2>), dim3(nwg), dim3(256), 0, st, a);
  else DRS_LAUNCH((wgrad_dma_kernel<TR, TO, 0>), dim3(nwg), dim3(256), 0, st, a);
  return DRS_LAUNCH_CHECK();
}

}  // namespace

// (used by engine.hip)
__attribute__((visibility("hidden"))) int drs_step_prep(const StepPrepArgs& a, hipStream_t stream) {
  if (a.n < 0 || a.n > STEP_PREP_MAX) return DRS_ERR_ARG;
  DRS_LAUNCH(step_prep_kernel, dim3(512, a.n + 2), dim3(256), 0, stream, a);
  return DRS_LAUNCH_CHECK();
}


int drs_g_skip_halo_taps = 1;
thread_local int drs_tl_chain = 0;
int drs_g_chain_mode = -1;

namespace {

int g_wgrad_target = 2048;   // workgroups the pixel split of the filter gradient aims at: two rounds of the 4 x 256 resident 128x128 DMA workgroups (sweep in profiles/r02/wgrad_ablation.txt; development switch drs_debug_wgrad_target)

int pick_tile(int c) { return c % 128 == 0 ? 128 : (c % 64 == 0 ? 64 : 32); }
int g_conv_wide192 = 1;      // development switch (drs_debug_conv_wide192): Cout = 192 as one 128 x 192 tile (0: three 128 x 64 tiles)
// forward / input-gradient N tile: Cout = 192 (conv5 / conv6, and the input gradients of conv6 / conv7) takes ONE 192-wide tile
// (wave tile 64 x 96: 6 MFMAs per 5 fragment reads, the A tile fetched once instead of three times) in the LDS-DMA form
int pick_conv_tile(int c, int cin) { return (g_conv_wide192 && c % 192 == 0 && c % 128 != 0 && cin % 32 == 0 && g_conv_variant != 0) ? 192 : pick_tile(c); }

// wgrad row tile: rows = k*k*Cin may be cut anywhere (a tile spans taps, the last one may be ragged), so take the
// tall 128-row tile whenever the ragged remainder wastes little; measured MFMA-busy: 128-row tiles 75 %, 64: 65-73 %, 32: 46 %
int pick_wgrad_rows(int rows) {
  const int n128 = (rows + 127) / 128;
  if ((double)rows / (n128 * 128.0) >= 0.85) return 128;
  if (rows % 64 == 0) return 64;
  return rows % 32 == 0 && rows < 128 ? 32 : 128;
}

// wgrad column tile: Cout = 192 (conv5 / conv6 of Dilated8Pooling) takes ONE 192-wide tile with the 128-row LDS-DMA form (wave
// tile 64 x 96: 6 MFMAs per 5 fragment reads, 320 operand floats per pixel for 24576 products) instead of three 64-wide ones
int pick_wgrad_cols(int tr, int cout) { return (tr == 128 && cout % 192 == 0 && cout % 128 != 0 && g_wgrad_variant != 0) ? 192 : pick_tile(cout); }

int g_wgrad_balance = 1;     // development switch (drs_debug_wgrad_balance): 0 = equal chunk ranges (and the old rule for skipping), 1 = cut by live pixels
int g_wgrad_len = 0;        // development switch (drs_debug_wgrad_len): chunks per workgroup the launches below the `big` class aim at (0 = by the pixel count, below)
// r03 tuned 96 at the per-rank batches; with the waves' priority by remaining work (from 2^18 pixels) the workgroups of a launch end
// together and longer ones pay: in-process A/B over conv2..conv8, B = 128: 96 -> 15.89 ms, 128 -> 15.78, 160 -> 15.79, 192 -> 15.80,
// 256 -> 16.08; B = 64: 8.09 / 8.08 / 8.19 (96 / 128 / 192); B = 16: 2.21 / 2.31 / 2.40
// (r04, with the kernels' loops cleaned up: B = 128 128 -> 14.77 ms, 192 -> 14.43 (conv6's 1534 workgroups of two rounds become 767 of one:
//  2.67 -> 2.35 ms), 256 -> 14.68; B = 128 at S = 75 / 85 and B = 64 at S = 100: 192 -0.4 .. -0.7 %; B = 64: 96 = 128, 192 +1.5 %)
inline int wgrad_len(int nchunks) { return g_wgrad_len ? g_wgrad_len : (nchunks >= (1 << 14) ? 192 : 96); }
int g_wgrad_minchunks = 8;  // development switch (drs_debug_wgrad_minchunks): fewest 32-pixel chunks a split may have
int g_wgrad_target_big = 0;  // development switch: workgroups aimed at on launches with many tiles and pixels under the live cut (0 = default)

int g_wgrad_prio = -1;       // development switch (drs_debug_wgrad_prio): -1 = by the rule in wgrad_setup, 0 = never, 1 = levels 3..0, 2 = levels 2..0
int g_wgrad_ablate = 0;      // development switch (drs_debug_wgrad_ablate): timing experiments with wrong sums
int g_wgrad_model = 1;       // development switch (drs_debug_wgrad_model): 1 = per-CU cost model for launches below the `big` class, 0 = the r02 table

// workgroups a filter-gradient launch of `work` chunk-tiles (32-pixel chunks x tiles) aims at; occ = workgroups of this tile
// shape a CU holds (4; the 128 x 192 tile: 3)
int wgrad_target(long long work, int ntile, int nchunks, bool balanced, int occ, int len = 0) {
  if (len <= 0) len = wgrad_len(nchunks);
  // fill the 256 CUs evenly.  Small launches (the per-rank batches of data parallelism) want long workgroups more than many:
  // >= 96 chunks each, down to one round of 512 (sweeps at B = 16 / 32 in profiles/r02/wgrad_ablation.txt).  Launches with many
  // tiles and pixels: with equal chunk ranges and the dead chunks skipped the workgroups differ in length by up to a quarter and
  // two rounds quantise the gain away, so twice as many (measured at B = 128: conv6 3.48 -> 3.01 ms, conv8 5.19 -> 4.85).
  long long fit = work / len;
  fit = fit < 512 ? 512 : (fit > g_wgrad_target ? g_wgrad_target : fit);
  const bool big = ntile >= 24 && nchunks >= 8192;
  if (!balanced) return big ? 2 * g_wgrad_target : (int)fit;
  // equal-length workgroups: the chip holds occ x 256 of them, and one workgroup over a whole number of rounds costs a round
  // (measured: 513 workgroups instead of 507 +27 %, 2049 instead of 2030 +7 %), so aim AT a whole number of workgroups per CU
  // (one round) or of rounds, and never above it (wgrad_live_plan rounds the splits down)
  if (big) return g_wgrad_target_big ? g_wgrad_target_big : 3 * 1024;
  if (g_wgrad_model) {
    // n workgroups per CU, all resident: a CU's time ~ (its chunks + n * o) / eff(n), o = what a workgroup costs besides its
    // chunks (assignment, pipeline fill, the 64 KB slab tile: ~6 chunks' worth), eff(n) = share of the MFMA rate n co-resident
    // workgroups reach (1: 0.75, 2: 0.88, 3: 0.93, 4: 0.95: the barrier bubbles of one are filled by the others).  Sweeps at
    // B = 16, S = 25 .. 85: profiles/r03/ab_wgrad_small.log.  Longer than g_wgrad_len chunks per workgroup: whole further rounds.
    static const double eff[5] = {1.0, 0.75, 0.88, 0.93, 0.95};
    const double W0 = (double)work / (double)cu_count(), o = 6.0;
    int n = 1;
    double bt = 1e30;
    for (int i = 1; i <= occ && i <= 4; ++i) {
      const double t = (W0 + i * o) / eff[i];
      if (t < bt) { bt = t; n = i; }
    }
    int r = (int)(W0 / n / (double)len + 0.5);
    r = r < 1 ? 1 : r;
    const long long t = (long long)cu_count() * n * r;
    return (int)(t > 4096 ? 4096 : t);
  }
  static const int steps[] = {512, 768, 1024, 2048, 3072, 4096};
  int best = steps[0];
  for (int s : steps)
    if (s <= g_wgrad_target && (double)(s > fit ? s : fit) / (s > fit ? fit : s) < (double)(best > fit ? best : fit) / (best > fit ? fit : best)) best = s;
  return best;
}

// equal chunk ranges: the number of splits.  Never less than g_wgrad_minchunks chunks per split.
int wgrad_uniform_splits(int B, int S, int k, int cin, int cout, int len = 0) {
  const long long M = (long long)B * S * S;
  const int tr = pick_wgrad_rows(k * k * cin), to = pick_wgrad_cols(tr, cout);
  const int ntile = ((k * k * cin + tr - 1) / tr) * (cout / to);
  const int nchunks = (int)((M + 31) / 32);
  int want = wgrad_target((long long)nchunks * ntile, ntile, nchunks, false, 4, len) / ntile;
  int maxs = (nchunks + g_wgrad_minchunks - 1) / g_wgrad_minchunks;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  const int cps = (nchunks + want - 1) / want;
  return (nchunks + cps - 1) / cps;
}

// the cut by live pixels (WgradPlan); false = not applicable (too many row tiles / classes, or nothing to skip): use the equal cut
bool wgrad_live_plan(int B, int S, int k, int rate, int pad, int cin, int cout, int tr, int to, int cap, WgradPlan& p, int& nwg) {
  const int rows = k * k * cin, ntr = (rows + tr - 1) / tr, nto = cout / to;
  p.nclass = 0;
  if (ntr > WG_MAXT) return false;
  int lp[WG_MAXT], vals[WG_MAXC], nv = 0;
  for (int r = 0; r < ntr; ++r) {
    int lo, hi;
    const int rlast = ((r + 1) * tr < rows ? (r + 1) * tr : rows) - 1;
    live_pixel_range(r * tr, rlast, cin, k, rate, pad, S, 1, lo, hi);
    lp[r] = hi - lo;
    int c = 0;
    while (c < nv && vals[c] != lp[r]) ++c;
    if (c == nv) { if (nv == WG_MAXC) return false; vals[nv++] = lp[r]; }
  }
  if (nv == 1 && vals[0] == S * S) return false;            // every tile meets every pixel: the equal cut is the live cut
  for (int i = 1; i < nv; ++i)                              // classes ascending by live pixels (so by splits)
    for (int j = i; j > 0 && vals[j] < vals[j - 1]; --j) { const int t = vals[j]; vals[j] = vals[j - 1]; vals[j - 1] = t; }
  int cnt[WG_MAXC] = {0};
  for (int r = 0; r < ntr; ++r) {
    int c = 0;
    while (vals[c] != lp[r]) ++c;
    p.cls[r] = (unsigned char)c;
    ++cnt[c];
  }
  const long long M = (long long)B * S * S;
  const int nchunks = (int)((M + 31) / 32);
  double work = 0;                                          // live chunk-tiles of the launch
  for (int c = 0; c < nv; ++c) work += (double)cnt[c] * nto * B * vals[c] / 32.0;
  const int target = wgrad_target((long long)work, ntr * nto, nchunks, true, (tr == 128 && to == 192) ? 3 : 4);
  const double len = work / target;                          // chunks per workgroup
  int maxs = (nchunks + g_wgrad_minchunks - 1) / g_wgrad_minchunks;
  if (maxs > cap) maxs = cap;
  if (maxs < 1) maxs = 1;
  double frac[WG_MAXC];
  int total = 0;
  for (int c = 0; c < nv; ++c) {                             // rounded down: never more workgroups than aimed at ...
    const double x = (double)B * vals[c] / 32.0 / len;
    int n = (int)x;
    frac[c] = x - n;
    if (n > maxs) { n = maxs; frac[c] = 0; }
    if (n < 1) { n = 1; frac[c] = 0; }
    p.n[c] = n;
    total += n * cnt[c] * nto;
  }
  for (;;) {                                                 // ... then the classes nearest their next split take one more while it fits
    int pick = -1;
    for (int c = 0; c < nv; ++c)
      if (frac[c] > 0 && p.n[c] < maxs && total + cnt[c] * nto <= target && (c + 1 == nv || p.n[c] < p.n[c + 1]) && (pick < 0 || frac[c] > frac[pick])) pick = c;
    if (pick < 0) break;
    ++p.n[pick];
    total += cnt[pick] * nto;
    frac[pick] = 0;
  }
  for (int c = 1; c < nv; ++c)
    if (p.n[c] < p.n[c - 1]) p.n[c] = p.n[c - 1];             // (ascending by construction; stay safe)
  int above = ntr, start = 0;                               // row tiles of classes >= j
  for (int j = 0; j < nv; ++j) {
    p.seg_start[j] = start;
    p.seg_width[j] = above * nto;
    start += (p.n[j] - (j ? p.n[j - 1] : 0)) * above * nto;
    above -= cnt[j];
  }
  p.seg_start[nv] = start;
  p.nclass = nv;
  nwg = start;
  return true;
}

int wgrad_bound_splits(int B, int S, int k, int cin, int cout) {
  // (worked out with the SHORTEST workgroup length the rule ever takes, whatever this shape's own: the host sizes the slab at
  //  (b_max, s_max) and every smaller step, whose launches may take the shorter length and so MORE splits, must fit into it:
  //  tests/test_wgrad_cut.py::test_slab_sized_at_the_largest_step_serves_every_smaller_one)
  const int shortest = g_wgrad_len && g_wgrad_len < 96 ? g_wgrad_len : 96;
  const int ua = wgrad_uniform_splits(B, S, k, cin, cout), ub = wgrad_uniform_splits(B, S, k, cin, cout, shortest);
  const int u = ua > ub ? ua : ub;
  int maxs = (int)(((long long)B * S * S + 32LL * g_wgrad_minchunks - 1) / (32LL * g_wgrad_minchunks));
  if (maxs < 1) maxs = 1;
  const int bound = u + u / 2 + 1;
  return bound < maxs ? bound : (maxs > u ? maxs : u);
}

// everything of a filter-gradient launch that follows from the shape: tile sizes, the cut, the workgroup count
int wgrad_setup(int B, int S, int k, int rate, int pad_before, int cin, int cout, WgradArgs& a, int& nwg, int& tr, int& to, int& nsplit) {
  if ((cin % 32 && cin != 8 && cin != 16) || cout % 32 || B < 1 || S < 1 || k < 1 || rate < 1) return DRS_ERR_ARG;
  const long long M = (long long)B * S * S;
  if (M <= 0 || M >= (1 << 24)) return DRS_ERR_ARG;
  a.x = nullptr; a.g = nullptr; a.slab = nullptr; a.Px = a.Pg = a.ld_x = a.ld_g = a.coff_x = a.coff_g = 0;
  a.S = S; a.M = (int)M;
  a.k = k; a.rate = rate; a.pad = pad_before; a.Cin = cin; a.Cout = cout;
  tr = pick_wgrad_rows(k * k * cin); to = pick_wgrad_cols(tr, cout);
  a.ntr = (k * k * cin + tr - 1) / tr; a.nto = cout / to;
  nsplit = wgrad_uniform_splits(B, S, k, cin, cout);
  const int nchunks = (int)((M + 31) / 32);
  a.chunks_per_split = (nchunks + nsplit - 1) / nsplit;
  a.rcpS = 1.0f / (float)S; a.rcpSS = 1.0f / (float)(S * S);
  a.ablate = g_wgrad_ablate;
  // priority by remaining work: only where the filter gradient has the chip to itself.  Below 2^18 pixels the step engine runs it on a
  // stream of its own BESIDE the batch-norm-backward -> input-gradient chain (engine.hip), and raised priorities there take the matrix
  // pipe from the chain that the step waits for (B = 16: 6.87 -> 6.94 ms with them on)
  a.prio = g_wgrad_prio >= 0 ? g_wgrad_prio : ((g_wgrad_ablate != 3 && M >= (1LL << 18)) ? 1 : 0);
#ifdef DRS_DEV
  a.trace = g_conv_trace;
#endif
  // the cut: by live pixels (every workgroup of the launch the same length, the dead chunks skipped at any size), or equal chunk
  // ranges where that does not apply; development switches: drs_debug_wgrad_balance(0) = the equal cut with the dead chunks
  // skipped from 2^19 pixels only, drs_debug_skip_taps(0) = same cut, the dead chunks multiplied (same sums: bitwise)
  nwg = nsplit * a.ntr * a.nto;
  a.plan.nclass = 0;
  a.live_cut = 0;
  if (g_wgrad_balance && wgrad_live_plan(B, S, k, rate, pad_before, cin, cout, tr, to, wgrad_bound_splits(B, S, k, cin, cout), a.plan, nwg)) {
    a.live_cut = 1;
    a.skip_halo = drs_g_skip_halo_taps != 0;
  } else {
    a.skip_halo = drs_skip_halo_taps_wgrad(M);
    a.live_cut = a.skip_halo;
  }
  return DRS_OK;
}

}  // namespace

extern "C" {

#ifdef DRS_DEV   /* development switches (include/drs_dev.h): only libdrs_hip_dev.so exports them; process-global */
/* 0 = also multiply the filter taps / pixel chunks that meet only halo zeros */
int drs_debug_skip_taps(int v) { const int old = drs_g_skip_halo_taps; if (v >= 0) drs_g_skip_halo_taps = v; return old; }

int drs_debug_wgrad_target(int v) { const int old = g_wgrad_target; if (v > 0) g_wgrad_target = v; return old; }

int drs_debug_wgrad_balance(int v) { const int old = g_wgrad_balance; if (v >= 0) g_wgrad_balance = v; return old; }

int drs_debug_wgrad_len(int v) { const int old = g_wgrad_len; if (v >= 0) g_wgrad_len = v; return old; }

int drs_debug_wgrad_minchunks(int v) { const int old = g_wgrad_minchunks; if (v > 0) g_wgrad_minchunks = v; return old; }

int drs_debug_wgrad_model(int v) { const int old = g_wgrad_model; if (v >= 0) g_wgrad_model = v; return old; }

int drs_debug_wgrad_prio(int v) { const int old = g_wgrad_prio; if (v >= -1) g_wgrad_prio = v; return old; }

int drs_debug_wgrad_ablate(int v) { const int old = g_wgrad_ablate; if (v >= 0) g_wgrad_ablate = v; return old; }

int drs_debug_wgrad_target_big(int v) { const int old = g_wgrad_target_big; if (v >= 0) g_wgrad_target_big = v; return old; }

int drs_debug_conv_wide192(int v) { const int old = g_conv_wide192; if (v >= 0) g_conv_wide192 = v; return old; }

int drs_debug_conv_variant(int v) { const int old = g_conv_variant; if (v >= -1) g_conv_variant = v; return old; }

size_t drs_conv_workspace_floats(int cout);
int drs_debug_conv_lpt(int v) { const int old = g_conv_lpt; if (v >= 0) g_conv_lpt = v; return old; }
/* the tile every logical workgroup index (after the XCD remap) of a plain forward launch of this shape would take: out[w] = tile;
   returns the number of workgroups, 0 when the launch keeps the natural order, negative on a rejected shape */
int drs_debug_conv_order(int B, int S, int k, int rate, int pad_before, int cin, int cout, int* out, int cap) {
  const long long M = (long long)B * S * S;
  if (M <= 0 || M >= (1 << 24) || cout % 32 || k < 1 || rate < 1) return -1;
  ConvArgs a;
  a.S = S; a.M = (int)M; a.k = k; a.rate = rate; a.pad = pad_before; a.Cin = cin; a.Cout = cout;
  a.skip_halo = drs_skip_halo_taps_fwd(M, cout);
  const int bn = pick_conv_tile(cout, cin);
  if (bn < 64) return 0;
  const int mt = (int)((M + 127) / 128), nt = cout / bn;
  const int W = cin >= 32 ? sk_geometry(mt * nt, k * k * (cin / 32), bn, drs_conv_workspace_floats(cout)).G : 0;      // (a caller with the full workspace)
  conv_lpt_setup(a, 128, mt, nt, W);
  if (!a.lpt_T) return 0;
  for (int w = 0; w < mt * nt && w < cap; ++w) out[w] = lpt_tile(w, a.lpt_T, a.lpt_ta, a.lpt_tb, a.lpt_P, nt);
  return mt * nt;
}
int drs_debug_conv_trace(void* dev_buffer) { g_conv_trace = (unsigned long long*)dev_buffer; return 0; }

int drs_debug_conv_splitk(int v) { const int old = g_conv_splitk; if (v >= -1) g_conv_splitk = v; return old; }
int drs_debug_conv_hybrid(int v) { const int old = g_conv_hybrid; if (v >= 0) g_conv_hybrid = v; return old; }
int drs_debug_conv_sk_order(int v) { const int old = g_conv_sk_order; if (v >= 0) g_conv_sk_order = v; return old; }
int drs_debug_chain_mode(int v) { const int old = drs_g_chain_mode; if (v >= -1) drs_g_chain_mode = v; return old; }
int drs_debug_conv_prio(int v) { const int old = g_conv_prio; if (v >= -1) g_conv_prio = v; return old; }
/* the stream-K geometry drs_conv_forward_ws takes for a launch of `tiles` tiles of nks K-steps, N tile bn, with the full workspace:
   out3 = (workgroups, ranges, tiles that are cut); returns the workgroup count (0 = one workgroup per tile) */
int drs_debug_conv_sk_geometry(int tiles, int nks, int bn, int* out3) {
  const SkGeom g = sk_geometry(tiles, nks, bn, 2ull * (size_t)sk_max_w() * 128 * (size_t)bn);
  if (out3) { out3[0] = g.G; out3[1] = g.W; out3[2] = g.T; }
  return g.G;
}

int drs_debug_wgrad_variant(int v) { const int old = g_wgrad_variant; if (v >= -1) g_wgrad_variant = v; return old; }
int drs_debug_wgrad_seg(int v) { const int old = g_wgrad_seg; if (v >= 0) g_wgrad_seg = v; return old; }

#endif

// M-tile height the forward/dgrad kernel uses for this Cout (= rows per BN-statistics slab row)
int drs_conv_mtile(int cout) { return pick_tile(cout) >= 64 ? 128 : 256; }

size_t drs_conv_workspace_floats(int cout) {
  const int bn = pick_conv_tile(cout, 32);
  return bn >= 64 ? 2ull * (size_t)sk_max_w() * 128 * (size_t)bn : 0;
}

// does the forward launch of this shape (a caller with the full workspace) leave out the filter-tap rows that meet only the zero halo?
// (bench.py prices the executed share of the algorithmic flops with it; the rule lives here and nowhere else)
int drs_conv_halo_skip(int B, int S, int k, int rate, int pad_before, int cin, int cout) {
  const long long M = (long long)B * S * S;
  if (M <= 0 || M >= (1 << 24) || cout % 32 || cin < 32 || cin % 32 || k < 1 || rate < 1) return 0;
  ConvArgs a;
  a.S = S; a.M = (int)M; a.k = k; a.rate = rate; a.pad = pad_before; a.Cin = cin; a.Cout = cout;
  a.skip_halo = drs_skip_halo_taps_fwd(M, cout);
  const int bn = pick_conv_tile(cout, cin);
  if (bn < 64) return a.skip_halo;
  const int mt = (int)((M + 127) / 128), nt = cout / bn;
  const int W = sk_geometry(mt * nt, k * k * (cin / 32), bn, drs_conv_workspace_floats(cout)).G;
  if (W) return 0;                                        // stream-K: every tile has the same number of K-steps
  conv_lpt_setup(a, 128, mt, nt, 0);
  return a.skip_halo;
}


int drs_conv_forward_ws(const float* in, int B, int S, int P, int ld_in, int coff_in, const float* w, const float* bias,
                        int k, int rate, int pad_before, int cin, int cout, float* out, int ld_out, int coff_out,
                        int accumulate, float* stats_partial, float* workspace, size_t workspace_floats, void* stream) {
  // cin: a multiple of 32, or 8 / 16 (few-band input: several taps share a K-step; w then has round_up(k*k*cin, 32) rows)
  if (!in || !w || !out || (cin % 32 && cin != 8 && cin != 16) || cout % 32 || k < 1 || rate < 1 || P < pad_before) return DRS_ERR_ARG;
  if (cin < 32 && (ld_in % 4 || coff_in % 4)) return DRS_ERR_ARG;
  if (accumulate && stats_partial) return DRS_ERR_ARG;              // the tile statistics are those of this call's own sums
  if (P < (k - 1) * rate - pad_before) return DRS_ERR_ARG;          // halo must cover pad_after too
  const long long M = (long long)B * S * S;
  if (M <= 0 || M >= (1 << 24)) return DRS_ERR_ARG;
  if ((long long)B * (S + 2 * P) * (S + 2 * P) * ld_in >= (1LL << 30)) return DRS_ERR_ARG;      // 32-bit BYTE offsets (slab < 4 GiB)
  ConvArgs a;
  a.in = in; a.S = S; a.P = P; a.ld_in = ld_in; a.coff_in = coff_in; a.M = (int)M;
  a.w = w; a.bias = bias; a.out = out; a.ld_out = ld_out; a.coff_out = coff_out; a.stats = stats_partial;
  a.k = k; a.rate = rate; a.pad = pad_before; a.Cin = cin; a.Cout = cout; a.accumulate = accumulate;
  a.rcpS = 1.0f / (float)S; a.rcpSS = 1.0f / (float)(S * S);
  a.skip_halo = drs_skip_halo_taps_fwd(M, cout);
  a.sk_slab = nullptr; a.sk_W = 0; a.sk_nks = 0; a.sk_U = 0; a.sk_T = a.sk_tiles = a.sk_G = a.sk_order = 0; a.prio = 0;
  a.lpt_T = a.lpt_ta = a.lpt_tb = a.lpt_P = 0;
#ifdef DRS_DEV
  a.trace = g_conv_trace;
#endif
  hipStream_t st = (hipStream_t)stream;
  if (!workspace) workspace_floats = 0;
  switch (pick_conv_tile(cout, cin)) {
    case 192: return launch_conv_dma<128, 192, 2, 2>(a, workspace, workspace_floats, st);
    case 128: return launch_conv<128, 128, 2, 2>(a, workspace, workspace_floats, st);
    case 64:  return launch_conv<128, 64, 2, 2>(a, workspace, workspace_floats, st);     // in-process A/B against 256 x 64: -4..-9 % at B = 128, -2..-12 % at B = 16
    default:  return launch_conv<256, 32, 4, 1>(a, workspace, workspace_floats, st);
  }
}

int drs_conv_forward(const float* in, int B, int S, int P, int ld_in, int coff_in, const float* w, const float* bias,
                     int k, int rate, int pad_before, int cin, int cout, float* out, int ld_out, int coff_out,
                     int accumulate, float* stats_partial, void* stream) {
  return drs_conv_forward_ws(in, B, S, P, ld_in, coff_in, w, bias, k, rate, pad_before, cin, cout, out, ld_out, coff_out, accumulate,
                             stats_partial, nullptr, 0, stream);
}

// upper bound of the splits any row tile of this layer is cut into, whatever its rate / padding (the cut by live pixels may give
// the full tiles more splits than the equal cut has: up to half as many again): workspace = that many * k*k*cin * cout floats
int drs_conv_wgrad_splits(int B, int S, int k, int cin, int cout) { return wgrad_bound_splits(B, S, k, cin, cout); }

#ifdef DRS_DEV
/* development aid (not part of the documented ABI): the workgroups drs_conv_wgrad would launch for this shape, worked out on the
   host by the kernels' own assignment code: out[5 i ..] = (row tile, column tile, split, first chunk, end chunk) of workgroup i,
   out_tile_splits[r] = splits of row tile r.  Returns the number of workgroups (also when it exceeds cap; nothing is then written
   past cap), negative on a rejected shape. */
int drs_debug_wgrad_cut(int B, int S, int k, int rate, int pad_before, int cin, int cout, int* out, int cap, int* out_tile_splits,
                        int* out_tile_rows) {
  WgradArgs a;
  int nwg, tr, to, nsplit;
  if (wgrad_setup(B, S, k, rate, pad_before, cin, cout, a, nwg, tr, to, nsplit)) return -1;
  if (out_tile_rows) *out_tile_rows = tr;
  for (int r = 0; r < a.ntr && out_tile_splits; ++r) out_tile_splits[r] = a.plan.nclass ? a.plan.n[a.plan.cls[r]] : nsplit;
  for (int i = 0; i < nwg && i < cap && out; ++i) {
    int tile, split, cbeg, cend, lo, hi;
    wgrad_assign(a, tr, i, tile, split, cbeg, cend, lo, hi);
    out[5 * i] = tile / a.nto; out[5 * i + 1] = tile % a.nto; out[5 * i + 2] = split; out[5 * i + 3] = cbeg; out[5 * i + 4] = cend;
  }
  return nwg;
}
#endif

int drs_conv_wgrad(const float* x, int B, int S, int Px, int ld_x, int coff_x, const float* g, int Pg, int ld_g,
                   int coff_g, int k, int rate, int pad_before, int cin, int cin_real, int cout, float* slab,
                   float* grad, void* stream) {
  if (!x || !g || !slab || !grad || cin_real > cin) return DRS_ERR_ARG;
  if ((long long)B * (S + 2 * Px) * (S + 2 * Px) * ld_x >= (1LL << 30)) return DRS_ERR_ARG;     // 32-bit BYTE offsets (slabs < 4 GiB)
  if ((long long)B * (S + 2 * Pg) * (S + 2 * Pg) * ld_g >= (1LL << 30)) return DRS_ERR_ARG;
  WgradArgs a;
  int nwg, tr, to, nsplit;
  if (wgrad_setup(B, S, k, rate, pad_before, cin, cout, a, nwg, tr, to, nsplit)) return DRS_ERR_ARG;
  a.x = x; a.Px = Px; a.ld_x = ld_x; a.coff_x = coff_x;
  a.g = g; a.Pg = Pg; a.ld_g = ld_g; a.coff_g = coff_g; a.slab = slab;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (tr == 128 && to == 192) rc = launch_wgrad_dma_only<128, 192>(a, nwg, st);
  else if (tr == 128 && to == 128) rc = launch_wgrad<128, 128>(a, nwg, st);
  else if (tr == 128 && to == 64) rc = launch_wgrad<128, 64>(a, nwg, st);
  else if (tr == 128 && to == 32) rc = launch_wgrad<128, 32>(a, nwg, st);
  else if (tr == 64 && to == 128) rc = launch_wgrad<64, 128>(a, nwg, st);
  else if (tr == 64 && to == 64) rc = launch_wgrad<64, 64>(a, nwg, st);
  else if (tr == 64 && to == 32) rc = launch_wgrad<64, 32>(a, nwg, st);
  else if (tr == 32 && to == 128) rc = launch_wgrad<32, 128>(a, nwg, st);
  else if (tr == 32 && to == 64) rc = launch_wgrad<32, 64>(a, nwg, st);
  else rc = launch_wgrad<32, 32>(a, nwg, st);
  if (rc) return rc;
  const int n = k * k * cin_real * cout;
  DRS_LAUNCH(wgrad_reduce_kernel, dim3((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), dim3(256), 0, st, slab,
                     grad, nsplit, k * k, cin, cin_real, cout, tr, a.plan);
  return DRS_LAUNCH_CHECK();
}

int drs_filter_flip_transpose(const float* w, float* wt, int k, int cin, int cout, void* stream) {
  if (!w || !wt) return DRS_ERR_ARG;
  const int n = k * k * cin * cout;
  DRS_LAUNCH(flip_transpose_kernel, dim3((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), dim3(256), 0,
                     (hipStream_t)stream, w, wt, k, cin, cout);
  return DRS_LAUNCH_CHECK();
}

int drs_filter_pad_cin(const float* w, float* wp, int k, int cin, int cin_pad, int cout, void* stream) {
  if (!w || !wp || cin_pad < cin) return DRS_ERR_ARG;
  const int n = k * k * cin_pad * cout;
  DRS_LAUNCH(pad_cin_kernel, dim3((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), dim3(256), 0,
                     (hipStream_t)stream, w, wp, k * k, cin, cin_pad, cout);
  return DRS_LAUNCH_CHECK();
}

}  // extern "C"
