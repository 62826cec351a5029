// Dilated stride-1 SAME convolution on gfx950 with fp32 operands carried as sums of bf16 terms ("split-bf16").
//
// Same operator as conv_mfma.hip (tf.nn.atrous_conv2d / tf.nn.conv2d + bias_add and their gradients,
// /root/reference/isprs_dilated_random.py:710-713); what differs is the arithmetic.  Every fp32 operand x is stored
// as NS bf16 terms x = x_0 + x_1 (+ x_2) (+ residual), each term the round-to-nearest bf16 of what the previous
// ones left, and a product a*b is evaluated on the bf16 MFMA pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA
// rate; this file uses the 16x16x32 form in the LDS-DMA kernel) as the partial products a_i*b_j with i + j < NS, accumulated in fp32:
//   NS = 2 ("bf16x3"):  a0 b0 + a0 b1 + a1 b0                         relative error of a product <= ~2^-16
//   NS = 3 ("bf16x6"):  a0 b0 + a0 b1 + a1 b0 + a0 b2 + a2 b0 + a1 b1  relative error <= ~2^-25 (below one fp32 ulp)
// Each bf16 x bf16 product is exact in fp32, so with NS = 3 the only roundings left are those of the fp32
// accumulation, as in the exact-fp32 kernel.  The fp32 path (conv_mfma.hip) stays the default arithmetic.
//
// Layouts: the terms of a slab are interleaved at 32-channel granularity: fp32 element e (the padded NHWC indexing of
// the slab the terms were split from: view (S, P, ld, coff), ld and coff multiples of 32) has its term s at
// (e & ~31) * NS + 32 * s + (e & 31).  The NS 64-byte pieces one K-step needs of a pixel are then one contiguous
// 64*NS-byte run = whole 128-byte cache lines, instead of NS half-used lines in NS separate planes.  Filters are split
// into the K-contiguous form the MFMA B operand wants, interleaved the same way: forward [Cout][k*k*Cin/32][NS][32],
// input gradient [Cin][k*k*Cout/32][NS][32] (taps reversed), so both operands of the implicit GEMM are rows of 16-byte
// K-chunks.
#include "drs_common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float x) {
  const __bf16 h = (__bf16)x;                       // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return (uint32_t)__builtin_bit_cast(unsigned short, h);
}

// x -> NS bf16 terms, term s = rne_bf16(x - sum of the earlier terms)   (the subtraction is exact in fp32)
template <int NS>
__device__ __forceinline__ void split_terms(float x, uint32_t (&t)[NS]) {
  float r = x;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    t[s] = f32_to_bf16_bits(r);
    r -= bf16_bits_to_f32(t[s]);
  }
}

// ------------------------------------------------------------------------------------------------ splitting
// term s of src[e] -> dst[(e & ~31) * NS + 32 s + (e & 31)], e < n (n a multiple of 32; slabs are)
template <int NS>
__global__ void split_planes_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i];
    const f32x4 b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
    uint32_t t[8][NS];
#pragma unroll
    for (int e = 0; e < 4; ++e) { split_terms<NS>(a[e], t[e]); split_terms<NS>(b[e], t[4 + e]); }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = t[2 * e][s] | (t[2 * e + 1][s] << 16);
      *reinterpret_cast<u32x4*>(dst + ((i >> 2) * NS + s) * 32 + (i & 3) * 8) = o;
    }
  }
}

// forward filter:  row o, K index kf = tap*cin_pad + c          holds (c < cin ? w[tap][c][o] : 0)
// gradient filter: row c, K index kd = (taps-1-tap)*cout + o    holds w[tap][c][o]        (c < cin)
// NS = 2: term s of K index kk of a row of length Ktot sits at row*Ktot*NS + (kk & ~31)*NS + 32 s + (kk & 31)
// NS = 3: blocked by 16-deep HALF K-steps -- [K/16][NS][rows][16]: term s of K index kk of row r (of nrows) sits at
//         (((kk >> 4) * NS + s) * nrows + r) * 16 + (kk & 15).  The B tile of one half K-step and term is then ONE contiguous run of
//         32-byte rows: the forward kernels fetch it in whole cache lines (with the row-major form a 16-channel piece is 32 bytes out
//         of every K*NS*2, a quarter of each line fetched; an ablation put the filter fetch at a third of the forward kernel's time)
template <int NS>
__device__ __forceinline__ size_t filter_term_off(int row, int kk, int s, int nrows, int Ktot) {
  if constexpr (NS == 3) return ((size_t)((kk >> 4) * NS + s) * nrows + row) * 16 + (kk & 15);
  else return (size_t)row * Ktot * NS + (size_t)(kk & ~31) * NS + 32 * s + (kk & 31);
}

template <int NS>
__global__ void filter_split_kernel(const float* __restrict__ w, int taps, int cin, int cin_pad, int cout,
                                    uint16_t* __restrict__ wf, uint16_t* __restrict__ wd) {
  const int nf = cout * taps * cin_pad;
  const int nd = wd ? cin * taps * cout : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += gridDim.x * blockDim.x) {
    float v;
    uint16_t* dst;
    int row, kk, nrows, ktot;
    if (i < nf) {
      const int c = i % cin_pad;
      const int rest = i / cin_pad;
      const int tap = rest % taps, o = rest / taps;
      v = c < cin ? w[((size_t)tap * cin + c) * cout + o] : 0.f;
      dst = wf;
      row = o; kk = tap * cin_pad + c; nrows = cout; ktot = taps * cin_pad;
    } else {
      const int j = i - nf;
      const int o = j % cout;
      const int rest = j / cout;
      const int tapr = rest % taps, c = rest / taps;
      v = w[((size_t)(taps - 1 - tapr) * cin + c) * cout + o];
      dst = wd;
      row = c; kk = tapr * cout + o; nrows = cin; ktot = taps * cout;
    }
    uint32_t t[NS];
    split_terms<NS>(v, t);
#pragma unroll
    for (int s = 0; s < NS; ++s) dst[filter_term_off<NS>(row, kk, s, nrows, ktot)] = (uint16_t)t[s];
  }
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
constexpr int BK = 32;          // channels per K-step (one filter tap x 32 channels = two MFMA k-steps of 16)
constexpr int LDR = BK + 8;     // LDS row stride in bf16 elements: 80 B -> 16 consecutive rows hit 16 distinct 16-B slots

struct SplitConvArgs {
  const uint16_t* in; int S, P, ld_in, coff_in;
  int M;
  const uint16_t* w;                    // [Cout][k*k*Cin/32][NS][32]
  const float* bias;
  float* out; int ld_out, coff_out;
  float* stats;
  int k, rate, pad, Cin, Cout;
  int accumulate;
  int skip_halo;
  float rcpS, rcpSS;
};

template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256) void conv_split_kernel(const SplitConvArgs a) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int NA = BM / 64, NB = BN / 64;     // 16-byte loads per thread per plane per K-step
  static_assert(BN >= 64, "B tile rows are loaded 64 at a time");

  __shared__ __attribute__((aligned(16))) uint16_t lds[NS * (BM + BN) * LDR];
  uint16_t* As = lds;                           // [NS][BM][LDR]
  uint16_t* Bs = lds + NS * BM * LDR;           // [NS][BN][LDR]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int ntn = a.Cout / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  const int Sp = a.S + 2 * a.P;
  const int Ktot = a.k * a.k * a.Cin;
  const int lrow = t >> 2, lchk = (t & 3) * 8;  // this thread stages 16-byte chunk lchk of rows lrow + 64 i
  uint32_t offA[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int p = m0 + lrow + 64 * i;
    p = p < a.M ? p : a.M - 1;
    offA[i] = NS * (padded_pixel_off(p, a.S, a.P, a.ld_in, a.rcpS, a.rcpSS, -a.pad, -a.pad) + (uint32_t)a.coff_in) + (uint32_t)lchk;
  }
  uint32_t offB[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) offB[i] = (uint32_t)(NS * (n0 + lrow + 64 * i) * Ktot + lchk);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int cpt = a.Cin / BK;
  int u_lo, u_hi;                               // tap rows that are not entirely in the zero halo for this tile (drs_common.hpp)
  live_tap_rows(m0, BM, a.M, a.S, a.k, a.rate, a.pad, a.rcpS, a.rcpSS, u_lo, u_hi);
  if (!a.skip_halo) { u_lo = 0; u_hi = a.k; }
  const int ks0 = u_lo * a.k * cpt;
  const int nks = u_hi * a.k * cpt;
  u32x4 ra[NS][NA], rb[NS][NB];
  int lu = u_lo, lv = 0, lc = 0;

  auto gload = [&](int ks) {
    const uint32_t soff = (uint32_t)(NS * ((lu * a.rate * Sp + lv * a.rate) * a.ld_in + lc * BK));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[s][i] = *reinterpret_cast<const u32x4*>(a.in + offA[i] + soff + 32 * s);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if constexpr (NS == 3)      // blocked filter image: this thread's 16-byte chunk (t & 3) is half (t & 3) >> 1, piece t & 1 of its row
          rb[s][i] = *reinterpret_cast<const u32x4*>(a.w + ((size_t)((ks * 2 + ((t & 3) >> 1)) * NS + s) * a.Cout + (n0 + lrow + 64 * i)) * 16 + (t & 1) * 8);
        else
          rb[s][i] = *reinterpret_cast<const u32x4*>(a.w + offB[i] + NS * ks * BK + 32 * s);
      }
    }
    if (++lc == cpt) { lc = 0; if (++lv == a.k) { lv = 0; ++lu; } }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(&As[(s * BM + lrow + 64 * i) * LDR + lchk]) = ra[s][i];
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(&Bs[(s * BN + lrow + 64 * i) * LDR + lchk]) = rb[s][i];
    }
  };

  gload(ks0);
  lstore();
  __syncthreads();
  const int arow = wm * WTM + li, brow = wn * WTN + li;
  for (int ks = ks0; ks < nks; ++ks) {
    if (ks + 1 < nks) gload(ks + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fa[NS][TM], fb[NS][TN];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
          fa[s][mi] = *reinterpret_cast<const bf16x8*>(&As[(s * BM + arow + mi * 32) * LDR + kk * 16 + h * 8]);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          fb[s][ni] = *reinterpret_cast<const bf16x8*>(&Bs[(s * BN + brow + ni * 32) * LDR + kk * 16 + h * 8]);
      }
      // partial products a_i * b_j, i + j < NS, smallest terms first
#pragma unroll
      for (int d = NS - 1; d >= 0; --d)
#pragma unroll
        for (int i = 0; i <= d; ++i)
#pragma unroll
          for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[d - i][ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
    if (ks + 1 < nks) { lstore(); __syncthreads(); }
  }

  // ---- epilogue (as conv_igemm_kernel): C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
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
  if (a.stats) {      // two-pass tile statistics, see tile_column_stats (drs_common.hpp); lds aliases the (finished) A tile
    const int rem = a.M - m0;
    tile_column_stats<TN, WM, BN>(
        reinterpret_cast<float*>(lds), t, wm, h == 0, (float)(rem < BM ? rem : BM), [&](int ni) { return wn * WTN + ni * 32 + li; },
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

// ---- the same GEMM with the tiles brought in by LDS-DMA (global_load_lds_dwordx4) into a double-buffered LDS image:
// no staging registers, no ds_write pass, one barrier per K-step, the load of K-step ks+1 in flight during the MFMAs of
// ks.  An LDS-DMA wave-instruction writes 64 lanes x 16 B to consecutive LDS bytes, so an image row is the bare 64 B of
// one pixel's 32 channels (16 rows per instruction) and the bank spread comes from a swizzle instead of padding: the
// 16-byte chunk c of row r sits in slot c ^ ((r >> 2) & 3); the DMA realises it by fetching, for LDS slot s, source
// chunk s ^ ((r >> 2) & 3), and the fragment reads apply the same XOR.  16 consecutive rows at one k-chunk then cover
// the 16 slots of the 256-byte bank row once.  MFMA shape: v_mfma_f32_16x16x32_bf16 (one 32-deep k-step per K-step; the
// 32x32x16 form of this kernel costs the same cycles per flop but the chip holds a lower clock on it: -4 % measured).
template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256) void conv_split_dma_kernel(const SplitConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass has no LDS-DMA builtin; it only needs the launch stub
  static_assert(WM * WN == 4, "4 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;          // 16x16 MFMA tiles per wave (v_mfma_f32_16x16x32_bf16)
  constexpr int IA = BM / 64, IB = BN / 64;            // DMA instructions per wave per plane (16 rows each)
  constexpr int ROWB = BK * 2;                         // bytes per image row (64)
  constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
  constexpr int STAGE = NS * (PLANE_A + PLANE_B);

  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, h = lane >> 4;               // 16x16x32 operand: row / column li, k-chunk h (8 of the 32 k)
  const int wm = wave / WN, wn = wave % WN;

  const int ntn = a.Cout / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  const int Sp = a.S + 2 * a.P;
  const int Ktot = a.k * a.k * a.Cin;
  // DMA lane roles: instruction j of a plane covers image rows 16 j .. 16 j + 15; lane l fills row 16 j + (l >> 2), slot l & 3
  const int drow = lane >> 2;
  const int dchk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;           // source chunk (elements) for this lane's slot
  uint32_t offA[IA], offB[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    int p = m0 + 16 * (wave + 4 * i) + drow;
    p = p < a.M ? p : a.M - 1;
    offA[i] = NS * (padded_pixel_off(p, a.S, a.P, a.ld_in, a.rcpS, a.rcpSS, -a.pad, -a.pad) + (uint32_t)a.coff_in) + (uint32_t)dchk;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) offB[i] = (uint32_t)(NS * (n0 + 16 * (wave + 4 * i) + drow) * Ktot + dchk);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;

  const int cpt = a.Cin / BK;
  int u_lo, u_hi;                               // tap rows that are not entirely in the zero halo for this tile (drs_common.hpp)
  live_tap_rows(m0, BM, a.M, a.S, a.k, a.rate, a.pad, a.rcpS, a.rcpSS, u_lo, u_hi);
  if (!a.skip_halo) { u_lo = 0; u_hi = a.k; }
  const int ks0 = u_lo * a.k * cpt;
  const int nks = u_hi * a.k * cpt;
  int lu = u_lo, lv = 0, lc = 0;

  auto issue = [&](int ks, int stage) {
    const uint32_t soff = (uint32_t)(NS * ((lu * a.rate * Sp + lv * a.rate) * a.ld_in + lc * BK));
    unsigned char* sb = lds + stage * STAGE;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < IA; ++i)
        __builtin_amdgcn_global_load_lds(a.in + offA[i] + soff + 32 * s,
                                         (__attribute__((address_space(3))) void*)(sb + s * PLANE_A + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < IB; ++i)
        __builtin_amdgcn_global_load_lds(a.w + offB[i] + NS * ks * BK + 32 * s,
                                         (__attribute__((address_space(3))) void*)(sb + NS * PLANE_A + s * PLANE_B + (wave + 4 * i) * 1024), 16, 0, 0);
    }
    if (++lc == cpt) { lc = 0; if (++lv == a.k) { lv = 0; ++lu; } }
  };

  // fragment read offsets (bytes inside a plane): row * 64 + (chunk ^ ((row >> 2) & 3)) * 16, chunk = h (one 32-deep k-step)
  const int sw = (li >> 2) & 3;
  const uint32_t ra = (uint32_t)((wm * WTM + li) * ROWB + ((h ^ sw) * 16));
  const uint32_t rb = (uint32_t)((wn * WTN + li) * ROWB + ((h ^ sw) * 16));

  issue(ks0, ks0 & 1);
  __syncthreads();
  for (int ks = ks0; ks < nks; ++ks) {
    if (ks + 1 < nks) issue(ks + 1, (ks + 1) & 1);
    const unsigned char* sa = lds + (ks & 1) * STAGE;
    const unsigned char* sbb = sa + NS * PLANE_A;
    {
      bf16x8 fa[NS][TM], fb[NS][TN];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
          fa[s][mi] = *reinterpret_cast<const bf16x8*>(sa + s * PLANE_A + mi * 16 * ROWB + ra);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          fb[s][ni] = *reinterpret_cast<const bf16x8*>(sbb + s * PLANE_B + ni * 16 * ROWB + rb);
      }
#pragma unroll
      for (int d = NS - 1; d >= 0; --d)
#pragma unroll
        for (int i = 0; i <= d; ++i)
#pragma unroll
          for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][mi], fb[d - i][ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  float bv[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + wn * WTN + ni * 16 + li;
    bv[ni] = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * WTM + mi * 16 + 4 * h + r;
        if (row < a.M) {
          float v = acc[mi][ni][r] + bv[ni];
          float* dst = a.out + (size_t)row * a.ld_out + a.coff_out + col;
          if (a.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
  if (a.stats) {      // two-pass tile statistics, see tile_column_stats (drs_common.hpp); the 16x16 C/D map puts a column in 4 lanes
    const int rem = a.M - m0;
    tile_column_stats<TN, WM, BN>(
        reinterpret_cast<float*>(lds), t, wm, h == 0, (float)(rem < BM ? rem : BM), [&](int ni) { return wn * WTN + ni * 16 + li; },
        [](float s) { s += __shfl_xor(s, 16); return s + __shfl_xor(s, 32); },
        [&](int ni, auto f) {
#pragma unroll
          for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (m0 + wm * WTM + mi * 16 + 4 * h + r < a.M) f(acc[mi][ni][r] + bv[ni]);
        },
        a.stats + ((size_t)(m0 / BM) * a.Cout + n0) * 2);
  }
#endif
}

// ---- three terms (bf16x6), the half-stage form: operands by LDS-DMA into a double-buffered image of 16-channel HALF K-steps, so that
// all three terms of both operands fit (48 KB) and two to three workgroups share a CU; every global address is (wave-uniform base in
// SGPRs) + (loop-invariant 32-bit byte offset of this lane), the (channel chunk, tap row, tap column) counters live in SGPRs -- the K loop
// issues no vector-ALU instruction besides its MFMAs -- and the K order is channel-major, as in conv_dma_kernel (conv_mfma.hip).  An image
// row is the bare 32 B of one pixel's 16 channels of one term; a DMA wave-instruction fills 32 rows; the filter image is blocked by half
// K-steps (filter_term_off), so a B tile is one contiguous run.  (A 32x32x16 form of this kernel, a 128x256 tile, rings of 3 / 4 stages
// and a whole-K-step A image were measured and dropped: profiles/r02/split_forms.txt.)
// On the 16x16x32 MFMA shape:  The chip holds a higher clock under v_mfma_f32_16x16x32_bf16 than under the 32x32x16
// form (bare loops on random operands, 4 workgroups per CU: 1.75 against 1.33 PFLOP/s, tools/ubench/mfma_shape.hip), but its K of 32
// is a whole K-step, two LDS stages of the half-stage image.  So one MFMA multiplies TWO of the six partial products of a 16-channel
// half instead: its 32-deep K is [16 channels of one term | 16 channels of another], A-side and B-side chosen so that the pairs
//   (a1 b0 + a2 b0), (a0 b1 + a0 b2), (a0 b0 + a1 b1)
// come out -- three MFMAs per 16x16 tile and half, the same flops as six 32x32x16 ones per 32x32 tile.  A lane's k-chunk q = lane / 16
// (8 of the 32 k) therefore reads piece q & 1 of the 32-byte image row of term T[q >> 1]; with 32-byte rows the 16 lanes of every
// ds_read_b128 group already cover the 256-byte bank row once, so this image is NOT swizzled.
template <int BM, int BN>
__global__ __launch_bounds__((BM / 64) * (BN == 256 ? 4 : 2) * 64, 2) void conv_split_half16_kernel(const SplitConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass has no LDS-DMA builtin; it only needs the launch stub
  // BM = 256 (8 waves, 4 x 2): the filter tile is fetched once per 256 pixels -- half the L2 requests for B per product
  // BN = 256 (8 waves, 2 x 4): the activation tile is fetched once for all 256 output channels -- half the L2 requests for A
  constexpr int WM = BM / 64, WN = BN == 256 ? 4 : 2, NS = 3, NW = WM * WN;
  static_assert((BM == 128 || BM == 256) && NW <= 8, "tile / wave layout");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int HK = 16;                               // channels per half
  constexpr int ROWB = HK * 2;                         // bytes per image row (32)
  constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
  constexpr int STAGE = NS * (PLANE_A + PLANE_B);
  static_assert(WTN % 16 == 0, "wave tile");

  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = a.Cout / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;
  const int Sp = a.S + 2 * a.P;
  const int Ktot = a.k * a.k * a.Cin;

  // DMA lane roles: a wave-instruction fills rows 32 j .. 32 j + 31 of one term plane; lane l fills row 32 j + (l >> 1), piece l & 1.
  // Wave w takes A row block w and B row blocks w, w + NW, ... of every term.
  const int dpiece = (lane & 1) * 16;                 // bytes
  constexpr int IBW = (BN / 32 + NW - 1) / NW;          // B row blocks per wave and term (some waves have fewer)
  constexpr int IAW = (BM / 32 + NW - 1) / NW;          // A row blocks per wave and term
  uint32_t offA[IAW], offB[IBW];
#pragma unroll
  for (int i = 0; i < IAW; ++i) {
    int p = m0 + 32 * (wave + NW * i) + (lane >> 1);
    p = p < a.M ? p : a.M - 1;
    offA[i] = (NS * (padded_pixel_off(p, a.S, a.P, a.ld_in, a.rcpS, a.rcpSS, -a.pad, -a.pad) + (uint32_t)a.coff_in)) * 2u + (uint32_t)dpiece;
  }
#pragma unroll
  for (int i = 0; i < IBW; ++i) {
    int o = n0 + 32 * (wave + NW * i) + (lane >> 1);
    o = o < a.Cout ? o : a.Cout - 1;                    // (BN = 64 / 192: row blocks past the tile are not issued)
    offB[i] = NS == 3 ? (uint32_t)o * 32u + (uint32_t)dpiece : (uint32_t)(NS * o * Ktot) * 2u + (uint32_t)dpiece;
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;

  const int cpt = a.Cin / BK;
  int u_lo, u_hi;
  live_tap_rows(m0, BM, a.M, a.S, a.k, a.rate, a.pad, a.rcpS, a.rcpSS, u_lo, u_hi);
  if (!a.skip_halo) { u_lo = 0; u_hi = a.k; }
  u_lo = __builtin_amdgcn_readfirstlane(u_lo);
  u_hi = __builtin_amdgcn_readfirstlane(u_hi);
  const int nks = (u_hi - u_lo) * a.k * cpt;
  const char* inb = reinterpret_cast<const char*>(a.in);
  const char* wbase = reinterpret_cast<const char*>(a.w);
  int lu = u_lo, lv = 0, lc = 0;                       // (tap row, tap column, channel chunk) of the K-step being fetched
  typedef __attribute__((address_space(3))) void* lds_ptr;

  auto issue = [&](int half, int stage) {
    unsigned char* sa = lds + stage * STAGE;
    unsigned char* sb = sa + NS * PLANE_A;
    // fp32 element offset of (tap, chunk) in the slab -> term image: * NS, term s at + 32 s elements, half at + 16 elements
    const uint32_t aoff = (uint32_t)(NS * ((lu * a.rate * Sp + lv * a.rate) * a.ld_in + lc * BK) + half * HK) * 2u;
    const uint32_t boff = NS == 3 ? (uint32_t)(((((lu * a.k + lv) * cpt + lc) * 2 + half) * NS) * a.Cout) * 32u
                                  : (uint32_t)(NS * (((lu * a.k + lv) * cpt + lc) * BK) + half * HK) * 2u;
    const uint32_t bterm = NS == 3 ? (uint32_t)a.Cout * 32u : 64u;       // bytes from one term's rows to the next
    const char* ab = inb + (size_t)aoff;
    const char* wb = wbase + (size_t)boff;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      // one scalar base per term, kept opaque: otherwise LLVM folds the term offset into a 64-bit VECTOR add per load
      const char* as = ab + s * 64;
      const char* ws = wb + s * bterm;
      asm volatile("" : "+s"(as));
      asm volatile("" : "+s"(ws));
#pragma unroll
      for (int i = 0; i < IAW; ++i) {
        if (32 * (wave + NW * i) < BM) {               // wave-uniform
          uint32_t o = offA[i]; asm volatile("" : "+v"(o));
          __builtin_amdgcn_global_load_lds(as + o, (lds_ptr)(sa + s * PLANE_A + (wave + NW * i) * 1024), 16, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < IBW; ++i) {
        if (32 * (wave + NW * i) < BN) {               // wave-uniform
          uint32_t o = offB[i]; asm volatile("" : "+v"(o));
          __builtin_amdgcn_global_load_lds(ws + o, (lds_ptr)(sb + s * PLANE_B + (wave + NW * i) * 1024), 16, 0, 0);
        }
      }
    }
  };
  auto next_kstep = [&]() { if (++lv == a.k) { lv = 0; if (++lu == u_hi) { lu = u_lo; ++lc; } } };

  // fragment offsets (bytes): row * 32 + (q & 1) * 16 inside the plane of the term this lane's k-chunk belongs to.
  // kinds: 0 = [t0 | t1], 1 = [t0 | t0] (A) / [t1 | t2] (B), 2 = [t1 | t2] (A) / [t0 | t0] (B)
  const int hi = q >> 1;
  const uint32_t rowa = (uint32_t)((wm * WTM + li) * ROWB + (q & 1) * 16);
  const uint32_t rowb = (uint32_t)((wn * WTN + li) * ROWB + (q & 1) * 16);
  const uint32_t fa01 = rowa + (uint32_t)(hi ? 1 : 0) * PLANE_A, fa00 = rowa, fa12 = rowa + (uint32_t)(hi ? 2 : 1) * PLANE_A;
  const uint32_t fb01 = rowb + (uint32_t)(hi ? 1 : 0) * PLANE_B, fb12 = rowb + (uint32_t)(hi ? 2 : 1) * PLANE_B, fb00 = rowb;
  auto compute = [&](int stage) {
    const unsigned char* sa = lds + stage * STAGE;
    const unsigned char* sb = sa + NS * PLANE_A;
    bf16x8 a01[TM], a00[TM], a12[TM], b01[TN], b12[TN], b00[TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      a12[mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 16 * ROWB + fa12);
      a00[mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 16 * ROWB + fa00);
      a01[mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 16 * ROWB + fa01);
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      b00[ni] = *reinterpret_cast<const bf16x8*>(sb + ni * 16 * ROWB + fb00);
      b12[ni] = *reinterpret_cast<const bf16x8*>(sb + ni * 16 * ROWB + fb12);
      b01[ni] = *reinterpret_cast<const bf16x8*>(sb + ni * 16 * ROWB + fb01);
    }
    // smallest partial products first: (a1 b0 + a2 b0), (a0 b1 + a0 b2), (a0 b0 + a1 b1)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a12[mi], b00[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a00[mi], b12[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01[mi], b01[ni], acc[mi][ni], 0, 0, 0);
  };

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ks = 0; ks < nks; ++ks) {
    issue(1, 1);                                  // second half of this K-step lands while the first is multiplied
    compute(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    next_kstep();
    if (ks + 1 < nks) issue(0, 0);                // first half of the next K-step (every wave is done with stage 0)
    compute(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: 16x16 C/D map col = lane & 15, row = 4 * (lane >> 4) + reg
  float bv[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + wn * WTN + ni * 16 + li;
    bv[ni] = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * WTM + mi * 16 + 4 * q + r;
        if (row < a.M) {
          float v = acc[mi][ni][r] + bv[ni];
          float* dst = a.out + (size_t)row * a.ld_out + a.coff_out + col;
          if (a.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
  if (a.stats) {      // per 128-pixel statistics row (a 256-pixel tile holds two: waves wm 0-1 and 2-3); the 16x16 C/D map puts a column in 4 lanes
    const int hrow = wm >> 1;                                  // which 128-pixel half of the tile this wave belongs to
    const int mh = m0 + 128 * hrow;
    const int rem = a.M - mh;
    const float nrow = rem <= 0 ? 1.f : (float)(rem < 128 ? rem : 128);
    float* dst = a.stats + ((size_t)(mh / 128) * a.Cout + n0) * 2;
    tile_column_stats<TN, 2, BN>(
        reinterpret_cast<float*>(lds) + hrow * 4 * BN, t - 256 * hrow, wm & 1, q == 0, nrow, [&](int ni) { return wn * WTN + ni * 16 + li; },
        [](float s) { s += __shfl_xor(s, 16); return s + __shfl_xor(s, 32); },
        [&](int ni, auto f) {
#pragma unroll
          for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (m0 + wm * WTM + mi * 16 + 4 * q + r < a.M) f(acc[mi][ni][r] + bv[ni]);
        },
        rem > 0 ? dst : nullptr);
  }
#endif
}

int g_variant = 1;      // 0: register-staged tiles everywhere, 1: LDS-DMA double-buffered tiles for the two-term arithmetic
                        // (development switch, see drs_debug_variant)

template <int BM, int BN, int WM, int WN, int NS>
int launch_split(const SplitConvArgs& a, hipStream_t st) {
  const int mt = (a.M + BM - 1) / BM, nt = a.Cout / BN;
  // three terms: the double-buffered LDS-DMA image (96 KiB) leaves one workgroup per CU; the register-staged kernel keeps two
  if (g_variant == 0 || NS == 3) DRS_LAUNCH((conv_split_kernel<BM, BN, WM, WN, NS>), dim3(mt * nt), dim3(256), 0, st, a);
  else DRS_LAUNCH((conv_split_dma_kernel<BM, BN, WM, WN, NS>), dim3(mt * nt), dim3(256), 0, st, a);
  return DRS_LAUNCH_CHECK();
}

template <int NS>
int dispatch_split(const SplitConvArgs& a, hipStream_t st) {
  // three terms (bf16x6): the 16x16x32 half-stage form on every tile width (with the blocked filter image it is the fastest form on
  // every Dilated8Pooling shape, profiles/r02/split_forms.txt), 8-wave tiles where Cout allows: 128 x 256 (the activation tile fetched once
  // for all 256 output channels: conv7 +10 %, conv8 +3 % over 256 x 128) or 256 x 128 (the filter tile fetched once per 256 pixels); development arms: 0 = register-staged, 8 = 128-pixel tiles only
  if constexpr (NS == 3) {
    const int mt = (a.M + 127) / 128;
    if (g_variant != 0) {
      if (a.Cout % 256 == 0 && g_variant != 8) DRS_LAUNCH((conv_split_half16_kernel<128, 256>), dim3(mt * (a.Cout / 256)), dim3(512), 0, st, a);
      else if (a.Cout % 128 == 0 && g_variant != 8) DRS_LAUNCH((conv_split_half16_kernel<256, 128>), dim3(((a.M + 255) / 256) * (a.Cout / 128)), dim3(512), 0, st, a);
      else if (a.Cout % 128 == 0) DRS_LAUNCH((conv_split_half16_kernel<128, 128>), dim3(mt * (a.Cout / 128)), dim3(256), 0, st, a);
      else if (a.Cout % 192 == 0) DRS_LAUNCH((conv_split_half16_kernel<128, 192>), dim3(mt * (a.Cout / 192)), dim3(256), 0, st, a);
      else DRS_LAUNCH((conv_split_half16_kernel<128, 64>), dim3(mt * (a.Cout / 64)), dim3(256), 0, st, a);
      return DRS_LAUNCH_CHECK();
    }
  }
  if (a.Cout % 128 == 0) return launch_split<128, 128, 2, 2, NS>(a, st);
  if (a.Cout % 192 == 0) return launch_split<128, 192, 2, 2, NS>(a, st);
  return launch_split<128, 64, 2, 2, NS>(a, st);
}

// ------------------------------------------------------------------------------------------------ wgrad
// dW[(tap, c)][o] = sum_p X[p + tap][c] * G[p][o]: the contraction runs over pixels, so both MFMA operands are wanted
// pixel-major per lane while the planes are stored [pixel][channel].  The LDS images keep the stored orientation
// (rows = pixels) and the fragments are fetched with ds_read_b64_tr_b16, the gfx950 transposing LDS read: per 16-lane
// group it reads a 4-row x 16-column block of 16-bit elements (lane 4q+p supplies the address of row q, columns
// 4p..4p+3) and hands lane i column i of the 4 rows.  Two such reads = the 8 consecutive k (pixels) of one MFMA
// operand.  Row stride = tile width + 32 elements (64 B), so the 4 rows of a read fall on the four 64-byte quarters
// of the 256-byte bank row: conflict-free.
struct SplitWgradArgs {
  const uint16_t* x; int S, Px, ld_x, coff_x;
  const uint16_t* g; int Pg, ld_g, coff_g;
  int M;
  int k, rate, pad, Cin, Cout;
  float* slab;
  int chunks_per_split;
  int ntr, nto;
  int o_base;                // first output column of this launch (Cout = 192 runs as a 128-wide and a 64-wide launch)
  int skip_halo;
  float rcpS, rcpSS;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* p0, const uint16_t* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int TR, int TO, int NS>
__global__ __launch_bounds__((TR / 64) * (TO / 64) * 64) void wgrad_split_kernel(const SplitWgradArgs a) {
  constexpr int WR = TR / 64, WC = TO / 64;
  constexpr int NT = 64 * WR * WC;
  constexpr int BP = 32;                        // pixels per K-step = two MFMA k-steps
  constexpr int LDX = TR + 32, LDG = TO + 32;   // LDS row strides (elements)
  constexpr int XQ = TR / 8, GQ = TO / 8;       // 16-byte chunks per pixel row
  constexpr int NX = BP * XQ / NT, NG = BP * GQ / NT;
  constexpr int XPS = NT / XQ, GPS = NT / GQ;

  __shared__ __attribute__((aligned(16))) uint16_t lds[NS * BP * (LDX + LDG)];
  __shared__ uint32_t tabx[2][BP], tabg[2][BP];
  uint16_t* Xs = lds;                           // [NS][BP][LDX]
  uint16_t* Gs = lds + NS * BP * LDX;           // [NS][BP][LDG]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;

  const int ntile = a.ntr * a.nto;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int split = id / ntile;
  const int tile = id % ntile;
  const int R0 = (tile / a.nto) * TR;
  const int o0 = a.o_base + (tile % a.nto) * TO;
  const int rows_all = a.k * a.k * a.Cin;
  const int myR = R0 + (t % XQ) * 8;            // this thread always stages the same 8 rows (tap, c..c+7)
  const bool row_ok = myR < rows_all;
  const int tap = (row_ok ? myR : 0) / a.Cin, c0 = (row_ok ? myR : 0) % a.Cin;
  const int u = tap / a.k, v = tap % a.k;
  const int Sxp = a.S + 2 * a.Px;
  // interleaved-term addressing: NS * (32-aligned fp32 element index) + 32 * term + (index & 31)
  const uint32_t xconst = (uint32_t)(NS * ((u * a.rate * Sxp + v * a.rate) * a.ld_x + a.coff_x + (c0 & ~31)) + (c0 & 31));
  const int og0 = o0 + (t % GQ) * 8;
  const uint32_t gconst = (uint32_t)(NS * (a.coff_g + (og0 & ~31)) + (og0 & 31));
  const int xpix = t / XQ, gpix = t / GQ;

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nchunks_total = (a.M + BP - 1) / BP;
  const int cbeg = split * a.chunks_per_split;
  int cend = cbeg + a.chunks_per_split;
  cend = cend < nchunks_total ? cend : nchunks_total;

  // chunks whose pixel rows meet only halo zeros for this tile's tap rows are jumped over (wgrad_kernel in conv_mfma.hip)
  int live_lo, live_hi;
  {
    const int rlast = (R0 + TR < rows_all ? R0 + TR : rows_all) - 1;
    live_pixel_range(R0, rlast, a.Cin, a.k, a.rate, a.pad, a.S, a.skip_halo, live_lo, live_hi);
  }
  const int S2 = a.S * a.S;
  auto next_chunk = [&](int c) { return next_live_chunk(c, S2, a.rcpSS, live_lo, live_hi); };
  auto fill_tables = [&](int chunk, int slot) {
    if (t < BP && chunk < cend) {
      const int p = chunk * BP + t;
      const int pc = p < a.M ? p : a.M - 1;
      tabx[slot][t] = padded_pixel_off(pc, a.S, a.Px, a.ld_x, a.rcpS, a.rcpSS, -a.pad, -a.pad);
      tabg[slot][t] = padded_pixel_off(pc, a.S, a.Pg, a.ld_g, a.rcpS, a.rcpSS, 0, 0) | (p < a.M ? 0u : 0x80000000u);
    }
  };

  u32x4 rx[NS][NX], rg[NS][NG];
  uint32_t gflag[NG];
  auto gload = [&](int slot) {
    const uint32_t* tx = tabx[slot];
    const uint32_t* tg = tabg[slot];
    uint32_t ox[NX], og[NG];
#pragma unroll
    for (int i = 0; i < NX; ++i) ox[i] = tx[xpix + XPS * i];
#pragma unroll
    for (int i = 0; i < NG; ++i) og[i] = tg[gpix + GPS * i];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < NX; ++i) rx[s][i] = *reinterpret_cast<const u32x4*>(a.x + NS * ox[i] + xconst + 32 * s);
#pragma unroll
      for (int i = 0; i < NG; ++i) rg[s][i] = *reinterpret_cast<const u32x4*>(a.g + NS * (og[i] & 0x7fffffffu) + gconst + 32 * s);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) gflag[i] = og[i] & 0x80000000u;
  };
  // rows that must read as zero (G rows of pixels >= M, X rows of a ragged tile) are zeroed here, at the LDS write, so that
  // nothing waits on the loads before the MFMAs of the current chunk
  auto lstore = [&]() {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < NX; ++i)
        *reinterpret_cast<u32x4*>(&Xs[(s * BP + xpix + XPS * i) * LDX + (t % XQ) * 8]) = row_ok ? rx[s][i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < NG; ++i)
        *reinterpret_cast<u32x4*>(&Gs[(s * BP + gpix + GPS * i) * LDG + (t % GQ) * 8]) = gflag[i] ? u32x4{0u, 0u, 0u, 0u} : rg[s][i];
    }
  };

  int ck0 = next_chunk(cbeg - 1);
  if (ck0 < cend) {
    int ck1 = next_chunk(ck0);
    fill_tables(ck0, 0);
    fill_tables(ck1, 1);
    __syncthreads();
    gload(0);
    lstore();
    __syncthreads();
    // transposing-read lane roles: group = lane>>4 -> channel block (group&1)*16, pixel block (group>>1)*8;
    // inside the group lane 4q+p addresses pixel row q, columns 4p..4p+3
    const int l16 = lane & 15;
    const int trow = h * 8 + (l16 >> 2);
    const int tcol = ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
    const uint16_t* xbase = Xs + trow * LDX + wr * 64 + tcol;
    const uint16_t* gbase = Gs + trow * LDG + wc * 64 + tcol;
    for (int it = 0; ck0 < cend; ++it) {
      const int ck2 = next_chunk(ck1);
      if (ck1 < cend) gload((it + 1) & 1);
      fill_tables(ck2, it & 1);
#pragma unroll
      for (int kk = 0; kk < BP / 16; ++kk) {
        bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const uint16_t* p = xbase + (s * BP + kk * 16) * LDX + mi * 32;
            fa[s][mi] = tr_frag(p, p + 4 * LDX);
          }
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const uint16_t* p = gbase + (s * BP + kk * 16) * LDG + ni * 32;
            fb[s][ni] = tr_frag(p, p + 4 * LDG);
          }
        }
#pragma unroll
        for (int d = NS - 1; d >= 0; --d)
#pragma unroll
          for (int i = 0; i <= d; ++i)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[d - i][ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();
      if (ck1 < cend) { lstore(); __syncthreads(); }
      ck0 = ck1;
      ck1 = ck2;
    }
  }
  const size_t rows_total = (size_t)rows_all;
  float* dst = a.slab + ((size_t)split * rows_total + R0) * a.Cout + o0;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = wc * 64 + ni * 32 + li;
        if (R0 + row < rows_all) dst[(size_t)row * a.Cout + col] = acc[mi][ni][r];
      }
}

// ---- the same filter-gradient GEMM with LDS-DMA tiles (double-buffered, one barrier per 32-pixel chunk).  The DMA
// writes lane-linear LDS, so the images are unpadded and swizzled instead: X rows are 256 B (128 rows of the
// [k*k*Cin] dimension), 16-byte chunk c of pixel row r at slot c ^ (((r & 3) << 2) | ((r >> 2) & 3)); G rows are 256 B
// (TO = 128, same rule) or 128 B (TO = 64, slot c ^ (((r >> 1) & 1) << 2)).  With these the four pixel rows of every
// transposing read fall on the four 64-byte quarters of the bank row.  Pixels past the end read G from the slab's first
// halo pixel (zeros; needs Pg > 0), so no select is needed; X rows past k*k*Cin read a valid address and only feed
// accumulator rows that are never stored.
template <int TO, int NS>
__global__ __launch_bounds__((TO / 64) * 128) void wgrad_split_dma_kernel(const SplitWgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass has no LDS-DMA builtin; it only needs the launch stub
  constexpr int TR = 128;
  constexpr int WC = TO / 64;
  constexpr int NW = 2 * WC;                    // waves
  constexpr int BP = 32;
  constexpr int XROW = TR * 2, GROW = TO * 2;   // bytes per image row
  constexpr int XT = BP * XROW, GT = BP * GROW; // bytes per term tile
  constexpr int STAGE = NS * (XT + GT);
  constexpr int IX = 8 / NW;                    // X DMA instructions per wave per term (4 pixel rows each)
  constexpr int IG = (GT / 1024) / NW;          // G DMA instructions per wave per term (4 or 8 pixel rows each)
  constexpr int GRPI = 1024 / GROW;             // pixel rows per G instruction

  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  __shared__ uint32_t tabx[2][BP], tabg[2][BP];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;

  const int ntile = a.ntr * a.nto;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int split = id / ntile;
  const int tile = id % ntile;
  const int R0 = (tile / a.nto) * TR;
  const int o0 = a.o_base + (tile % a.nto) * TO;
  const int rows_all = a.k * a.k * a.Cin;
  const int Sxp = a.S + 2 * a.Px;

  // DMA lane roles.  X instruction j: pixel row 4 j + (lane >> 4), slot lane & 15 <- source chunk slot ^ fX(row)
  uint32_t xconst[IX];
  int xrow[IX];
#pragma unroll
  for (int i = 0; i < IX; ++i) {
    const int j = wave + NW * i;
    xrow[i] = 4 * j + (lane >> 4);
    const int chunk = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (j & 3));
    int R = R0 + chunk * 8;
    R = R < rows_all ? R : 0;
    const int tap = R / a.Cin, c0 = R % a.Cin;
    const int u = tap / a.k, v = tap % a.k;
    xconst[i] = (uint32_t)(NS * ((u * a.rate * Sxp + v * a.rate) * a.ld_x + a.coff_x + (c0 & ~31)) + (c0 & 31));
  }
  uint32_t gconst;
  int grow[IG];
  {
    const int slot = TO == 128 ? (lane & 15) : (lane & 7);
    const int chunk = TO == 128 ? (slot ^ ((((lane >> 4) & 3) << 2) | (wave & 3))) : (slot ^ (((lane >> 4) & 1) << 2));
    const int og0 = o0 + chunk * 8;
    gconst = (uint32_t)(NS * (a.coff_g + (og0 & ~31)) + (og0 & 31));
#pragma unroll
    for (int i = 0; i < IG; ++i) grow[i] = GRPI * (wave + NW * i) + (TO == 128 ? (lane >> 4) : (lane >> 3));
  }
  static_assert(TO == 64 || NW == 4, "the TO = 128 G swizzle uses j & 3 == wave");

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nchunks_total = (a.M + BP - 1) / BP;
  const int cbeg = split * a.chunks_per_split;
  int cend = cbeg + a.chunks_per_split;
  cend = cend < nchunks_total ? cend : nchunks_total;

  // chunks whose pixel rows meet only halo zeros for this tile's tap rows are jumped over (wgrad_kernel in conv_mfma.hip)
  int live_lo, live_hi;
  {
    const int rlast = (R0 + TR < rows_all ? R0 + TR : rows_all) - 1;
    live_pixel_range(R0, rlast, a.Cin, a.k, a.rate, a.pad, a.S, a.skip_halo, live_lo, live_hi);
  }
  const int S2 = a.S * a.S;
  auto next_chunk = [&](int c) { return next_live_chunk(c, S2, a.rcpSS, live_lo, live_hi); };
  auto fill_tables = [&](int chunk, int slot) {
    if (t < BP && chunk < cend) {
      const int p = chunk * BP + t;
      const int pc = p < a.M ? p : a.M - 1;
      tabx[slot][t] = padded_pixel_off(pc, a.S, a.Px, a.ld_x, a.rcpS, a.rcpSS, -a.pad, -a.pad);
      tabg[slot][t] = p < a.M ? padded_pixel_off(pc, a.S, a.Pg, a.ld_g, a.rcpS, a.rcpSS, 0, 0) : 0u;   // 0 = a halo pixel: zeros
    }
  };
  auto issue = [&](int slot, int stage) {       // table slot == LDS stage == parity of the iteration that consumes the chunk
    const uint32_t* tx = tabx[slot];
    const uint32_t* tg = tabg[slot];
    unsigned char* sb = lds + stage * STAGE;
    uint32_t ox[IX], og[IG];
#pragma unroll
    for (int i = 0; i < IX; ++i) ox[i] = NS * tx[xrow[i]] + xconst[i];
#pragma unroll
    for (int i = 0; i < IG; ++i) og[i] = NS * tg[grow[i]] + gconst;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < IX; ++i)
        __builtin_amdgcn_global_load_lds(a.x + ox[i] + 32 * s,
                                         (__attribute__((address_space(3))) void*)(sb + s * XT + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < IG; ++i)
        __builtin_amdgcn_global_load_lds(a.g + og[i] + 32 * s,
                                         (__attribute__((address_space(3))) void*)(sb + NS * XT + s * GT + (wave + NW * i) * 1024), 16, 0, 0);
    }
  };

  int ck0 = next_chunk(cbeg - 1);
  if (ck0 < cend) {
    int ck1 = next_chunk(ck0);
    // transposing-read offsets (bytes inside a term tile, kk = 0): lane 4q+p of a 16-lane group addresses pixel row
    // r0 + q, 16-byte chunk c0 + (p >> 1), half p & 1, with r0 = 16 kk + 8 h + 4 j2 and c0 = the group's 16 columns
    const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, g1 = (lane >> 4) & 1;
    uint32_t xo[2][2], go[2][2];      // [mi | ni][j2]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int row = 8 * h + 4 * j2 + q;
        const int fx = (q << 2) | ((2 * h + j2) & 3);
        const int chx = (wr * 8 + m * 4 + g1 * 2 + (pp >> 1)) ^ fx;
        xo[m][j2] = (uint32_t)(XROW * row + 16 * chx + 8 * (pp & 1));
        if (TO == 128) {
          const int chg = (wc * 8 + m * 4 + g1 * 2 + (pp >> 1)) ^ fx;
          go[m][j2] = (uint32_t)(GROW * row + 16 * chg + 8 * (pp & 1));
        } else {
          const int chg = (m * 4 + g1 * 2 + (pp >> 1)) ^ (((q >> 1) & 1) << 2);
          go[m][j2] = (uint32_t)(GROW * row + 16 * chg + 8 * (pp & 1));
        }
      }
    fill_tables(ck0, 0);
    fill_tables(ck1, 1);
    __syncthreads();
    issue(0, 0);
    __syncthreads();
    for (int it = 0; ck0 < cend; ++it) {
      const int stage = it & 1;
      const int ck2 = next_chunk(ck1);
      if (ck1 < cend) issue(stage ^ 1, stage ^ 1);
      fill_tables(ck2, stage);
      const uint16_t* sx = reinterpret_cast<const uint16_t*>(lds + stage * STAGE);
      const uint16_t* sg = reinterpret_cast<const uint16_t*>(lds + stage * STAGE + NS * XT);
#pragma unroll
      for (int kk = 0; kk < BP / 16; ++kk) {
        bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            fa[s][mi] = tr_frag(sx + (s * XT + kk * 16 * XROW + xo[mi][0]) / 2, sx + (s * XT + kk * 16 * XROW + xo[mi][1]) / 2);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            fb[s][ni] = tr_frag(sg + (s * GT + kk * 16 * GROW + go[ni][0]) / 2, sg + (s * GT + kk * 16 * GROW + go[ni][1]) / 2);
        }
#pragma unroll
        for (int d = NS - 1; d >= 0; --d)
#pragma unroll
          for (int i = 0; i <= d; ++i)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[d - i][ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();
      ck0 = ck1;
      ck1 = ck2;
    }
  }
  const size_t rows_total = (size_t)rows_all;
  float* dst = a.slab + ((size_t)split * rows_total + R0) * a.Cout + o0;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = wc * 64 + ni * 32 + li;
        if (R0 + row < rows_all) dst[(size_t)row * a.Cout + col] = acc[mi][ni][r];
      }
#endif
}

// ---- the LDS-DMA filter gradient in HALF stages (as wgrad_dma_kernel does it for exact fp32): the double-buffered image holds two
// 16-pixel halves of a 32-pixel chunk, so three terms of both operands take 48 KB (TO = 128) and three workgroups share a CU; one
// v_mfma_f32_32x32x16_bf16 k-step per half; the DMA of half h + 1 lands while half h is multiplied.  Swizzles, lane roles and the
// order of every sum are those of wgrad_split_dma_kernel (its swizzle is periodic in 16 pixel rows).
template <int TO, int NS>
__global__ __launch_bounds__((TO / 64) * 128) void wgrad_split_half_kernel(const SplitWgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass has no LDS-DMA builtin; it only needs the launch stub
  constexpr int TR = 128;
  constexpr int WC = TO / 64;
  constexpr int NW = 2 * WC;                    // waves
  constexpr int BP = 32;
  constexpr int XROW = TR * 2, GROW = TO * 2;   // bytes per image row
  constexpr int HP = 16;                        // pixels per stage
  constexpr int XT = HP * XROW, GT = HP * GROW; // bytes per term half tile
  constexpr int STAGE = NS * (XT + GT);
  constexpr int IX = 4 / NW;                    // X DMA instructions per wave, term and half (4 pixel rows each)
  constexpr int IG = (GT / 1024) / NW;          // G DMA instructions per wave, term and half (4 or 8 pixel rows each)
  static_assert(IX >= 1 && IG >= 1, "wave layout");
  constexpr int GRPI = 1024 / GROW;             // pixel rows per G instruction

  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  __shared__ uint32_t tabx[2][BP], tabg[2][BP];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;

  const int ntile = a.ntr * a.nto;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int split = id / ntile;
  const int tile = id % ntile;
  const int R0 = (tile / a.nto) * TR;
  const int o0 = a.o_base + (tile % a.nto) * TO;
  const int rows_all = a.k * a.k * a.Cin;
  const int Sxp = a.S + 2 * a.Px;

  // DMA lane roles.  X instruction j: pixel row 4 j + (lane >> 4), slot lane & 15 <- source chunk slot ^ fX(row)
  uint32_t xconst[IX];
  int xrow[IX];
#pragma unroll
  for (int i = 0; i < IX; ++i) {
    const int j = wave + NW * i;
    xrow[i] = 4 * j + (lane >> 4);
    const int chunk = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (j & 3));
    int R = R0 + chunk * 8;
    R = R < rows_all ? R : 0;
    const int tap = R / a.Cin, c0 = R % a.Cin;
    const int u = tap / a.k, v = tap % a.k;
    xconst[i] = (uint32_t)(NS * ((u * a.rate * Sxp + v * a.rate) * a.ld_x + a.coff_x + (c0 & ~31)) + (c0 & 31));
  }
  uint32_t gconst;
  int grow[IG];
  {
    const int slot = TO == 128 ? (lane & 15) : (lane & 7);
    const int chunk = TO == 128 ? (slot ^ ((((lane >> 4) & 3) << 2) | (wave & 3))) : (slot ^ (((lane >> 4) & 1) << 2));
    const int og0 = o0 + chunk * 8;
    gconst = (uint32_t)(NS * (a.coff_g + (og0 & ~31)) + (og0 & 31));
#pragma unroll
    for (int i = 0; i < IG; ++i) grow[i] = GRPI * (wave + NW * i) + (TO == 128 ? (lane >> 4) : (lane >> 3));
  }
  static_assert(TO == 64 || NW == 4, "the TO = 128 G swizzle uses j & 3 == wave");

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nchunks_total = (a.M + BP - 1) / BP;
  const int cbeg = split * a.chunks_per_split;
  int cend = cbeg + a.chunks_per_split;
  cend = cend < nchunks_total ? cend : nchunks_total;

  // chunks whose pixel rows meet only halo zeros for this tile's tap rows are jumped over (wgrad_kernel in conv_mfma.hip)
  int live_lo, live_hi;
  {
    const int rlast = (R0 + TR < rows_all ? R0 + TR : rows_all) - 1;
    live_pixel_range(R0, rlast, a.Cin, a.k, a.rate, a.pad, a.S, a.skip_halo, live_lo, live_hi);
  }
  const int S2 = a.S * a.S;
  auto next_chunk = [&](int c) { return next_live_chunk(c, S2, a.rcpSS, live_lo, live_hi); };
  auto fill_tables = [&](int chunk, int slot) {
    if (t < BP && chunk < cend) {
      const int p = chunk * BP + t;
      const int pc = p < a.M ? p : a.M - 1;
      tabx[slot][t] = padded_pixel_off(pc, a.S, a.Px, a.ld_x, a.rcpS, a.rcpSS, -a.pad, -a.pad);
      tabg[slot][t] = p < a.M ? padded_pixel_off(pc, a.S, a.Pg, a.ld_g, a.rcpS, a.rcpSS, 0, 0) : 0u;   // 0 = a halo pixel: zeros
    }
  };
  auto issue = [&](int slot, int half, int stage) {       // half `half` of the chunk whose pixel offsets are in table slot `slot`
    const uint32_t* tx = tabx[slot] + half * HP;
    const uint32_t* tg = tabg[slot] + half * HP;
    unsigned char* sb = lds + stage * STAGE;
    uint32_t ox[IX], og[IG];
#pragma unroll
    for (int i = 0; i < IX; ++i) ox[i] = NS * tx[xrow[i]] + xconst[i];
#pragma unroll
    for (int i = 0; i < IG; ++i) og[i] = NS * tg[grow[i]] + gconst;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < IX; ++i)
        __builtin_amdgcn_global_load_lds(a.x + ox[i] + 32 * s,
                                         (__attribute__((address_space(3))) void*)(sb + s * XT + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < IG; ++i)
        __builtin_amdgcn_global_load_lds(a.g + og[i] + 32 * s,
                                         (__attribute__((address_space(3))) void*)(sb + NS * XT + s * GT + (wave + NW * i) * 1024), 16, 0, 0);
    }
  };

  int ck0 = next_chunk(cbeg - 1);
  if (ck0 < cend) {
    int ck1 = next_chunk(ck0);
    // transposing-read offsets (bytes inside a term tile, kk = 0): lane 4q+p of a 16-lane group addresses pixel row
    // r0 + q, 16-byte chunk c0 + (p >> 1), half p & 1, with r0 = 16 kk + 8 h + 4 j2 and c0 = the group's 16 columns
    const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, g1 = (lane >> 4) & 1;
    uint32_t xo[2][2], go[2][2];      // [mi | ni][j2]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int row = 8 * h + 4 * j2 + q;
        const int fx = (q << 2) | ((2 * h + j2) & 3);
        const int chx = (wr * 8 + m * 4 + g1 * 2 + (pp >> 1)) ^ fx;
        xo[m][j2] = (uint32_t)(XROW * row + 16 * chx + 8 * (pp & 1));
        if (TO == 128) {
          const int chg = (wc * 8 + m * 4 + g1 * 2 + (pp >> 1)) ^ fx;
          go[m][j2] = (uint32_t)(GROW * row + 16 * chg + 8 * (pp & 1));
        } else {
          const int chg = (m * 4 + g1 * 2 + (pp >> 1)) ^ (((q >> 1) & 1) << 2);
          go[m][j2] = (uint32_t)(GROW * row + 16 * chg + 8 * (pp & 1));
        }
      }
    fill_tables(ck0, 0);
    fill_tables(ck1, 1);
    __syncthreads();
    issue(0, 0, 0);
    __syncthreads();
    auto compute = [&](int stage) {
      const uint16_t* sx = reinterpret_cast<const uint16_t*>(lds + stage * STAGE);
      const uint16_t* sg = reinterpret_cast<const uint16_t*>(lds + stage * STAGE + NS * XT);
      bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[s][mi] = tr_frag(sx + (s * XT + xo[mi][0]) / 2, sx + (s * XT + xo[mi][1]) / 2);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[s][ni] = tr_frag(sg + (s * GT + go[ni][0]) / 2, sg + (s * GT + go[ni][1]) / 2);
      }
#pragma unroll
      for (int d = NS - 1; d >= 0; --d)
#pragma unroll
        for (int i = 0; i <= d; ++i)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[d - i][ni], acc[mi][ni], 0, 0, 0);
    };
    for (int it = 0; ck0 < cend; ++it) {
      const int slot = it & 1;                    // table slot of the chunk being multiplied
      const int ck2 = next_chunk(ck1);
      issue(slot, 1, 1);                          // its second half lands while the first is multiplied
      compute(0);
      __syncthreads();                            // (drains the DMA; every wave has issued from this chunk's table slot)
      if (ck1 < cend) issue(slot ^ 1, 0, 0);      // first half of the next chunk
      fill_tables(ck2, slot);
      compute(1);
      __syncthreads();
      ck0 = ck1;
      ck1 = ck2;
    }
  }
  const size_t rows_total = (size_t)rows_all;
  float* dst = a.slab + ((size_t)split * rows_total + R0) * a.Cout + o0;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = wc * 64 + ni * 32 + li;
        if (R0 + row < rows_all) dst[(size_t)row * a.Cout + col] = acc[mi][ni][r];
      }
#endif
}

// grad[tap][c][o] = sum over splits (fixed order) of slab[split][tap][c (of cin_pad)][o], c < cin_real
__global__ void wgrad_split_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int nsplit, int taps,
                                          int cin_pad, int cin_real, int cout) {
  const int n = taps * cin_real * cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int o = i % cout;
    const int rc = i / cout;
    const int c = rc % cin_real, tap = rc / cin_real;
    const size_t src = ((size_t)tap * cin_pad + c) * cout + o;
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

template <int TR, int TO, int NS>
int launch_wgrad_split(const SplitWgradArgs& a, int nsplit, hipStream_t st) {
  DRS_LAUNCH((wgrad_split_kernel<TR, TO, NS>), dim3(nsplit * a.ntr * a.nto), dim3((TR / 64) * (TO / 64) * 64), 0, st, a);
  return DRS_LAUNCH_CHECK();
}

// the LDS-DMA filter gradient: two terms (wgrad_split_dma_kernel) and three (wgrad_split_half_kernel, 128-wide column tiles only;
// development arm 6 = register-staged everywhere)
inline bool wgrad_dma(int nterms, int Pg, int cout = 128) { return g_variant != 0 && g_variant != 6 && Pg > 0 && (nterms == 2 || cout >= 128); }
int split_wgrad_rows(int rows, int Pg, int nterms, int cout) {
  if (wgrad_dma(nterms, Pg, cout)) return 128;       // the LDS-DMA kernel: 128-row tiles only, zeros fetched from the halo
  const int n128 = (rows + 127) / 128;
  return (double)rows / (n128 * 128.0) >= 0.85 ? 128 : 64;
}

inline int grid_for(size_t n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace

extern "C" {

#ifdef DRS_DEV   /* development switch between kernel variants (include/drs_dev.h; libdrs_hip_dev.so only) */
int drs_debug_variant(int v) { const int old = g_variant; if (v >= 0) g_variant = v; return old; }
#endif


int drs_split_conv_mtile(int cout) { (void)cout; return 128; }

int drs_split_terms(const float* src, size_t n, int nterms, unsigned short* terms, void* stream) {
  if (!src || !terms || (n & 31) || (nterms != 2 && nterms != 3)) return DRS_ERR_ARG;
  if (n == 0) return DRS_OK;
  hipStream_t st = (hipStream_t)stream;
  if (nterms == 2) DRS_LAUNCH(split_planes_kernel<2>, dim3(grid_for(n / 8)), dim3(256), 0, st, src, terms, n / 8);
  else DRS_LAUNCH(split_planes_kernel<3>, dim3(grid_for(n / 8)), dim3(256), 0, st, src, terms, n / 8);
  return DRS_LAUNCH_CHECK();
}

int drs_filter_split(const float* w, int k, int cin, int cin_pad, int cout, int nsplit, unsigned short* wf,
                     unsigned short* wd, void* stream) {
  if (!w || !wf || cin_pad < cin || cin_pad % 32 || cout % 32 || (nsplit != 2 && nsplit != 3)) return DRS_ERR_ARG;
  if (wd && cin % 32) return DRS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)k * k * cout * (cin_pad + (wd ? cin : 0));
  if (nsplit == 2) DRS_LAUNCH(filter_split_kernel<2>, dim3(grid_for(n)), dim3(256), 0, st, w, k * k, cin, cin_pad, cout, wf, wd);
  else DRS_LAUNCH(filter_split_kernel<3>, dim3(grid_for(n)), dim3(256), 0, st, w, k * k, cin, cin_pad, cout, wf, wd);
  return DRS_LAUNCH_CHECK();
}

int drs_conv_forward_split(const unsigned short* in, int B, int S, int P, int ld_in, int coff_in, const unsigned short* w,
                           const float* bias, int k, int rate, int pad_before, int cin, int cout, float* out, int ld_out,
                           int coff_out, int accumulate, float* stats_partial, int nsplit, void* stream) {
  if (!in || !w || !out || cin % 32 || cout % 64 || k < 1 || rate < 1 || P < pad_before) return DRS_ERR_ARG;
  if (P < (k - 1) * rate - pad_before || (ld_in & 31) || (coff_in & 31) || (nsplit != 2 && nsplit != 3)) return DRS_ERR_ARG;
  if (accumulate && stats_partial) return DRS_ERR_ARG;
  const long long M = (long long)B * S * S;
  if (M <= 0 || M >= (1 << 24)) return DRS_ERR_ARG;
  // the kernels address with 32-bit element offsets: the whole term image of the input slab and the filter must stay below 2^32
  if ((long long)B * (S + 2 * P) * (S + 2 * P) * ld_in * nsplit >= (1LL << 32)) return DRS_ERR_ARG;
  if ((long long)k * k * cin * cout * nsplit >= (1LL << 32)) return DRS_ERR_ARG;
  // (the half-stage LDS-DMA form addresses in 32-bit BYTE offsets: term images below 4 GiB)
  if ((long long)B * (S + 2 * P) * (S + 2 * P) * ld_in * nsplit >= (1LL << 31) || (long long)k * k * cin * cout * nsplit >= (1LL << 31)) return DRS_ERR_ARG;
  SplitConvArgs a;
  a.in = in; a.S = S; a.P = P; a.ld_in = ld_in; a.coff_in = coff_in; a.M = (int)M;
  a.w = w; a.bias = bias; a.out = out; a.ld_out = ld_out; a.coff_out = coff_out;
  a.stats = stats_partial; a.k = k; a.rate = rate; a.pad = pad_before; a.Cin = cin; a.Cout = cout; a.accumulate = accumulate;
  a.rcpS = 1.0f / (float)S; a.rcpSS = 1.0f / (float)(S * S);
  a.skip_halo = drs_skip_halo_taps_fwd(M, cout);
  hipStream_t st = (hipStream_t)stream;
  return nsplit == 2 ? dispatch_split<2>(a, st) : dispatch_split<3>(a, st);
}

// splits of the pixel dimension: *bound = the most any (b <= B, s <= S) is cut into (monotone in B and S: what a workspace sized
// once for (b_max, s_max) needs -- the exact count is not monotone in S); returns the exact count for (B, S)
static int wgrad_split_count(int B, int S, int k, int cin, int cout, int Pg, int nterms, int* bound) {
  const long long M = (long long)B * S * S;
  const int tr = split_wgrad_rows(k * k * cin, Pg, nterms, cout), to = cout % 128 == 0 ? 128 : 64;
  int ntile = ((k * k * cin + tr - 1) / tr) * (cout / to);
  if ((wgrad_dma(nterms, Pg, cout) || nterms == 3) && to == 64 && cout > 64) ntile = ((k * k * cin + tr - 1) / tr) * ((cout / 128) + 1);   // 128-wide tiles + one 64-wide
  const int nchunks = (int)((M + 31) / 32);
  int want = ((ntile >= 24 && nchunks >= 8192) ? 3072 : 1536) / ntile;      // see drs_conv_wgrad_splits (conv_mfma.hip)
  int maxs = (nchunks + 31) / 32;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (bound) *bound = want;
  const int cps = (nchunks + want - 1) / want;
  return (nchunks + cps - 1) / cps;
}

int drs_conv_wgrad_split_splits(int B, int S, int k, int cin, int cout, int Pg, int nterms) {
  int bound = 1;
  (void)wgrad_split_count(B, S, k, cin, cout, Pg, nterms, &bound);
  return bound;
}

int drs_conv_wgrad_split(const unsigned short* x, int B, int S, int Px, int ld_x, int coff_x, const unsigned short* g,
                         int Pg, int ld_g, int coff_g, int k, int rate, int pad_before, int cin, int cin_real, int cout,
                         float* slab, float* grad, int nsplit_terms, void* stream) {
  if (!x || !g || !slab || !grad || cin % 32 || cout % 64 || cin_real > cin) return DRS_ERR_ARG;
  if ((ld_x & 31) || (coff_x & 31) || (ld_g & 31) || (coff_g & 31) || (nsplit_terms != 2 && nsplit_terms != 3)) return DRS_ERR_ARG;
  const long long M = (long long)B * S * S;
  if (M <= 0 || M >= (1 << 24)) return DRS_ERR_ARG;
  if ((long long)B * (S + 2 * Px) * (S + 2 * Px) * ld_x * nsplit_terms >= (1LL << 32)) return DRS_ERR_ARG;   // 32-bit element offsets
  if ((long long)B * (S + 2 * Pg) * (S + 2 * Pg) * ld_g * nsplit_terms >= (1LL << 31)) return DRS_ERR_ARG;   // bit 31 flags a row
  SplitWgradArgs a;
  a.x = x; a.S = S; a.Px = Px; a.ld_x = ld_x; a.coff_x = coff_x;
  a.g = g; a.Pg = Pg; a.ld_g = ld_g; a.coff_g = coff_g; a.M = (int)M;
  a.k = k; a.rate = rate; a.pad = pad_before; a.Cin = cin; a.Cout = cout; a.slab = slab;
  const int tr = split_wgrad_rows(k * k * cin, Pg, nsplit_terms, cout), to = cout % 128 == 0 ? 128 : 64;
  a.ntr = (k * k * cin + tr - 1) / tr; a.nto = cout / to;
  const int nsplit = wgrad_split_count(B, S, k, cin, cout, Pg, nsplit_terms, nullptr);
  const int nchunks = (int)((M + 31) / 32);
  a.chunks_per_split = (nchunks + nsplit - 1) / nsplit;
  a.rcpS = 1.0f / (float)S; a.rcpSS = 1.0f / (float)(S * S);
  a.o_base = 0;
  a.skip_halo = drs_skip_halo_taps_wgrad(M);
  hipStream_t st = (hipStream_t)stream;
  int rc = DRS_OK;
  if (wgrad_dma(nsplit_terms, Pg, cout)) {
    // 128-wide column tiles wherever they fit, one 64-wide tile for what is left (Cout = 64, 192); three terms: the 64-wide rest
    // goes to the register-staged kernel (the half-stage form loses 8 % there)
    const int n128 = cout / 128, rest = cout % 128;
    if (n128) {
      a.nto = n128; a.o_base = 0;
      const dim3 grid(nsplit * a.ntr * a.nto);
      if (nsplit_terms == 3) DRS_LAUNCH((wgrad_split_half_kernel<128, 3>), grid, dim3(256), 0, st, a);
      else DRS_LAUNCH((wgrad_split_dma_kernel<128, 2>), grid, dim3(256), 0, st, a);
      rc = DRS_LAUNCH_CHECK();
    }
    if (rest && rc == DRS_OK) {
      a.nto = 1; a.o_base = n128 * 128;
      const dim3 grid(nsplit * a.ntr);
      if (nsplit_terms == 3) rc = launch_wgrad_split<128, 64, 3>(a, nsplit, st);
      else { DRS_LAUNCH((wgrad_split_dma_kernel<64, 2>), grid, dim3(128), 0, st, a); rc = DRS_LAUNCH_CHECK(); }
    }
  } else if (nsplit_terms == 2) {
    if (tr == 128 && to == 128) rc = launch_wgrad_split<128, 128, 2>(a, nsplit, st);
    else if (tr == 128) rc = launch_wgrad_split<128, 64, 2>(a, nsplit, st);
    else if (to == 128) rc = launch_wgrad_split<64, 128, 2>(a, nsplit, st);
    else rc = launch_wgrad_split<64, 64, 2>(a, nsplit, st);
  } else if (to == 64 && cout > 64) {
    // three terms, Cout = 192: 128-wide column tiles where they fit (they run at the rate of the other layers, the 64-wide
    // ones a quarter below it) and one 64-wide tile for the rest
    const int n128 = cout / 128;
    a.nto = n128; a.o_base = 0;
    rc = tr == 128 ? launch_wgrad_split<128, 128, 3>(a, nsplit, st) : launch_wgrad_split<64, 128, 3>(a, nsplit, st);
    if (rc == DRS_OK) {
      a.nto = 1; a.o_base = n128 * 128;
      rc = tr == 128 ? launch_wgrad_split<128, 64, 3>(a, nsplit, st) : launch_wgrad_split<64, 64, 3>(a, nsplit, st);
    }
  } else {
    if (tr == 128 && to == 128) rc = launch_wgrad_split<128, 128, 3>(a, nsplit, st);
    else if (tr == 128) rc = launch_wgrad_split<128, 64, 3>(a, nsplit, st);
    else if (to == 128) rc = launch_wgrad_split<64, 128, 3>(a, nsplit, st);
    else rc = launch_wgrad_split<64, 64, 3>(a, nsplit, st);
  }
  if (rc) return rc;
  const int n = k * k * cin_real * cout;
  DRS_LAUNCH(wgrad_split_reduce_kernel, dim3((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), dim3(256), 0, st, slab, grad,
             nsplit, k * k, cin, cin_real, cout);
  return DRS_LAUNCH_CHECK();
}

}  // extern "C"
