// Patch materialisation and whole-tile stitching on gfx950 (HBM-bound gather / scatter-free accumulate).
//
// Reference call sites (/root/reference/isprs_dilated_random.py):
//   dynamically_create_patches :245-334  (crop, border shift-back done by the host, rotate / noise / flip augmentation)
//   normalize_images           :74-81    ((x - mean[c]) / std[c] for c = 0,1,2 ONLY)
//   create_patches_per_map     :337-400  (same gather, window positions from the host)
//   overlap-add of logits      :1261-1284 / :1925-1949, arg-max of the average
//
// The crop writes straight into the zero-haloed, channel-padded input slab of conv1, so no separate pad/normalise
// pass exists.  Arithmetic on pixel values is fp64 (the reference normalises float64 patches, then feeds float32),
// rounded once to fp32 on the store, so the result is bit-identical to the reference's feed.
#include "drs_common.hpp"

// Everything in this file restates host arithmetic that the reference does in numpy / scipy and is held to it BIT FOR BIT: no
// multiply-add may be fused.  hipcc contracts a * b + c into an fma by default, and HIP's __dmul_rn / __dadd_rn are no
// barrier against that (they are folded into fmas all the same): the rotation's source coordinate ((i m00) + j m01) + off came out as fma(j, m01, i m00) + off, which
// picks the other neighbour than scipy.ndimage at exact ties (multiples of 15 / 45 degrees at some sides: tests/fuzz/check_rotation.py).
#pragma clang fp contract(off)

namespace {

struct CropArgs {
  const void* tiles;            // pool of HWC tiles (double or float)
  const unsigned char* labels;  // pool of HW label maps
  const long long* tile_off;    // [nmaps] element offset of each tile in the pool
  const long long* lab_off;     // [nmaps]
  const int* tile_h; const int* tile_w;
  int C;                        // real channels
  const int* inst;              // [B][4]: map, x (row), y (col), flip (0 none, 1 flipud, 2 fliplr)
  const double* rot;            // [B][6]: m00 m01 m10 m11 off0 off1 (scipy affine, output->input) or null
  const unsigned char* rot_on;  // [B] or null
  const double* noise;          // [B][S][S][C] additive noise (pre-flip coordinates) or null
  const unsigned char* noise_on;  // [B] or null
  unsigned long long seed;      // device noise (Philox) when noise == null and noise_on[b]
  int quantize_f16;             // coffee:293 + :67-74: value, (value - mean) and (... / std) each rounded to float16
  int b0;                       // index of this call's first patch in the global batch: the noise of a patch does not depend on how the batch is sharded
  int void_label;               // pixels carrying this label are masked out too (contest:235-239); -1 = none
  double mean[3], stdv[3];
  float* out; int S, P, ld;     // conv1 input slab [B][S+2P][S+2P][ld]
  unsigned char* out_lab;       // [B][S][S]
  unsigned char* out_mask;      // [B][S][S] validity (0 where the rotation pulled in fill)
};

// Philox-4x32-10 -> two N(0,1) (Box-Muller) per call, keyed by (seed, element index)
__device__ __forceinline__ void philox(unsigned long long seed, unsigned long long ctr, unsigned (&o)[4]) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x1234567u, c3 = 0x89abcdefu;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ double normal_from(unsigned a, unsigned b) {
  const double u1 = ((double)a + 1.0) * (1.0 / 4294967296.0), u2 = (double)b * (1.0 / 4294967296.0);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

template <typename T>
__global__ void crop_kernel(const CropArgs a) {
  const int Sp = a.S + 2 * a.P;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= Sp) return;
  const int b = blockIdx.y / Sp, yy = blockIdx.y - b * Sp;
  float* dst = a.out + ((size_t)(b * Sp + yy) * Sp + xx) * a.ld;
  const int i = yy - a.P, j = xx - a.P;
  const bool inside = i >= 0 && i < a.S && j >= 0 && j < a.S;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (inside) {
    const int map = a.inst[4 * b], px = a.inst[4 * b + 1], py = a.inst[4 * b + 2], flip = a.inst[4 * b + 3];
    // undo the flip (applied last by the reference), then the rotation (applied first)
    const int fi = flip == 1 ? a.S - 1 - i : i, fj = flip == 2 ? a.S - 1 - j : j;
    int si = fi, sj = fj;
    bool valid = true;
    if (a.rot_on && a.rot_on[b]) {
      const double* m = a.rot + 6 * b;
      // scipy.ndimage geometric transform, order 0: in = M . out + offset (no FMA contraction), nearest = floor(c + 0.5)
      // (plain operators under this file's `fp contract(off)`: every product and every sum rounded on its own, as in ndimage's C)
      const double p00 = (double)fi * m[0], p01 = (double)fj * m[1], p10 = (double)fi * m[2], p11 = (double)fj * m[3];
      double c0 = 0.0 + p00;
      c0 = c0 + p01;
      c0 = c0 + m[4];
      double c1 = 0.0 + p10;
      c1 = c1 + p11;
      c1 = c1 + m[5];
      valid = !(c0 < 0.0 || c0 > (double)(a.S - 1) || c1 < 0.0 || c1 > (double)(a.S - 1));
      si = (int)floor(c0 + 0.5);
      sj = (int)floor(c1 + 0.5);
    }
    const size_t opix = ((size_t)b * a.S + i) * a.S + j;
    unsigned char lab = 0;
    const T* src = nullptr;
    if (valid) {
      const int W = a.tile_w[map];
      src = reinterpret_cast<const T*>(a.tiles) + a.tile_off[map] + ((size_t)(px + si) * W + (py + sj)) * a.C;
      lab = a.labels[a.lab_off[map] + (size_t)(px + si) * W + (py + sj)];
    }
    // rotated-in fill is value 0 (+ noise), label 0, mask 0: what ndimage.rotate(cval=0) leaves behind
    const bool noisy = a.noise_on && a.noise_on[b];
    for (int c = 0; c < a.C; ++c) {
      double e = valid ? (double)src[c] : 0.0;      // fp64 until the single rounding on the store
      if (noisy) {
        const size_t ne = (((size_t)b * a.S + fi) * a.S + fj) * a.C + c;   // noise is indexed before the flip
        const size_t ng = ne + (size_t)a.b0 * a.S * a.S * a.C;             // ... and by the patch's place in the GLOBAL batch on the device path
        if (a.noise) e = e + a.noise[ne];
        else {
          unsigned r[4];
          philox(a.seed, (unsigned long long)ng, r);
          e = e + 0.01 * normal_from(r[0], r[1]);
        }
      }
      if (a.quantize_f16) {
        // coffee:293 casts the patches to float16, coffee:67-74 normalises in that array.  NumPy >= 2 (NEP 50) evaluates
        // float16-array (op) numpy-scalar in the SCALAR's type when that is wider, and the assignment rounds to float16:
        // 1 = float32 scalars (what coffee's own compute_image_mean gives: np.mean / np.std of float32 patches), 2 = float64 scalars
        _Float16 q = (_Float16)(float)e;
        if (c < 3) {
          if (a.quantize_f16 == 2) {
            q = (_Float16)((double)q - a.mean[c]);
            q = (_Float16)((double)q / a.stdv[c]);
          } else {
            q = (_Float16)((float)q - (float)a.mean[c]);
            q = (_Float16)((float)q / (float)a.stdv[c]);
          }
        }
        v[c] = (float)q;
        continue;
      }
      if (c < 3) e = (e - a.mean[c]) / a.stdv[c];
      v[c] = (float)e;
    }
    if (a.out_lab) a.out_lab[opix] = lab;
    if (a.out_mask) a.out_mask[opix] = (valid && (int)lab != a.void_label) ? 1 : 0;
  }
  // one pixel = ld floats: the real channels, then zero padding up to the conv1 K-step
  for (int c4 = 0; c4 < a.ld; c4 += 4) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (c4 + k) < 8 ? v[(c4 + k) & 7] : 0.f;
    *reinterpret_cast<f32x4*>(dst + c4) = o;
  }
}

// ------------------------------------------------------------------------------------------------ stitch
struct StitchArgs {
  float* prob;            // [h][w][K]
  unsigned int* occur;    // [h][w]   (the reference replicates the count over K; one copy is kept)
  const float* logits;    // [nb][S][S][K]
  int h, w, K, S, stride, n_h, n_w;
  int f0, nb;             // windows f0 .. f0+nb-1 (row-major flat index) are in `logits`
  int row0, nrows;        // image rows touched by this batch
};

// windows along one axis that cover coordinate v, in ascending window index: regular ones at r*stride, plus the
// last one when it was shifted back to end at the border (isprs:366-375)
__device__ __forceinline__ int covering(int v, int len, int S, int stride, int n, int (&idx)[6], int (&pos)[6]) {
  const int n_reg = (len - S) / stride + 1;
  int lo = v - S + 1; lo = lo <= 0 ? 0 : (lo + stride - 1) / stride;
  int hi = v / stride; if (hi > n_reg - 1) hi = n_reg - 1;
  int cnt = 0;
  for (int r = lo; r <= hi && cnt < 5; ++r) { idx[cnt] = r; pos[cnt] = r * stride; ++cnt; }
  if (n > n_reg && v >= len - S) { idx[cnt] = n - 1; pos[cnt] = len - S; ++cnt; }
  return cnt;
}

// one thread per image pixel of the touched band; windows are added in their flat (reference) order
__global__ void stitch_accumulate_kernel(const StitchArgs a) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = a.row0 + blockIdx.y;
  if (x >= a.w || y >= a.h) return;
  int ri[6], rp[6], ci[6], cp[6];
  const int nr = covering(y, a.h, a.S, a.stride, a.n_h, ri, rp);
  const int nc = covering(x, a.w, a.S, a.stride, a.n_w, ci, cp);
  float acc[8];
  float* pp = a.prob + ((size_t)y * a.w + x) * a.K;
  for (int k = 0; k < a.K; ++k) acc[k] = pp[k];
  unsigned cnt = 0;
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nc; ++j) {
      const int f = ri[i] * a.n_w + ci[j] - a.f0;
      if (f < 0 || f >= a.nb) continue;
      const float* lg = a.logits + (((size_t)f * a.S + (y - rp[i])) * a.S + (x - cp[j])) * a.K;
      for (int k = 0; k < a.K; ++k) acc[k] += lg[k];
      ++cnt;
    }
  if (cnt) {
    for (int k = 0; k < a.K; ++k) pp[k] = acc[k];
    a.occur[(size_t)y * a.w + x] += cnt;
  }
}

// arg-max over classes of prob / max(occur, 1) (first maximum); the division is by a per-pixel positive constant
__global__ void stitch_finalize_kernel(const float* __restrict__ prob, const unsigned int* __restrict__ occur, size_t npix, int K,
                                       unsigned char* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned oc = occur[i] ? occur[i] : 1u;
    int am = 0;
    double best = (double)prob[i * K] / (double)oc;
    for (int k = 1; k < K; ++k) {
      const double v = (double)prob[i * K + k] / (double)oc;
      if (v > best) { best = v; am = k; }
    }
    out[i] = (unsigned char)am;
  }
}

// multi-scale evaluation (isprs:1347-1474): per scale, softmax over classes of the averaged logits, summed over scales
__global__ void softmax_accumulate_kernel(const float* __restrict__ prob, const unsigned int* __restrict__ occur, size_t npix, int K,
                                          float* __restrict__ acc) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const double oc = (double)(occur[i] ? occur[i] : 1u);
    float e[8], sum = 0.f;
    for (int k = 0; k < K; ++k) {
      e[k] = expf((float)((double)prob[i * K + k] / oc));     // the reference's softmax() has no max subtraction (isprs:38-43)
      sum += e[k];
    }
    for (int k = 0; k < K; ++k) acc[i * K + k] += e[k] / sum;
  }
}

}  // namespace

extern "C" {

int drs_crop_normalize(const void* tiles, int tiles_are_f64, const unsigned char* labels, const long long* tile_off,
                       const long long* lab_off, const int* tile_h, const int* tile_w, int C, const int* inst,
                       const double* rot, const unsigned char* rot_on, const double* noise, const unsigned char* noise_on,
                       unsigned long long seed, int noise_index0, const double* mean3, const double* std3, int B, int S, int P, int ld,
                       float* out, unsigned char* out_lab, unsigned char* out_mask, int void_label, int quantize_f16, void* stream) {
  if (!tiles || !labels || !tile_off || !lab_off || !tile_h || !tile_w || !inst || !out || !mean3 || !std3) return DRS_ERR_ARG;
  if (C < 1 || C > 8 || ld < C || ld % 4) return DRS_ERR_ARG;
  const int Sp = S + 2 * P;
  if ((long long)B * Sp > 65535) return DRS_ERR_ARG;
  CropArgs a;
  a.tiles = tiles; a.labels = labels; a.tile_off = tile_off; a.lab_off = lab_off; a.tile_h = tile_h; a.tile_w = tile_w; a.C = C;
  a.inst = inst; a.rot = rot; a.rot_on = rot ? rot_on : nullptr; a.noise = noise; a.noise_on = noise_on; a.seed = seed; a.b0 = noise_index0; a.void_label = void_label; a.quantize_f16 = quantize_f16;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
  a.out = out; a.S = S; a.P = P; a.ld = ld; a.out_lab = out_lab; a.out_mask = out_mask;
  dim3 grid((Sp + 63) / 64, B * Sp);
  if (tiles_are_f64) DRS_LAUNCH(crop_kernel<double>, grid, dim3(64), 0, (hipStream_t)stream, a);
  else DRS_LAUNCH(crop_kernel<float>, grid, dim3(64), 0, (hipStream_t)stream, a);
  return DRS_LAUNCH_CHECK();
}

int drs_stitch_accumulate(float* prob, unsigned int* occur, const float* logits, int h, int w, int K, int S, int stride,
                          int first_window, int n_windows, void* stream) {
  if (!prob || !occur || !logits || K < 1 || K > 8 || S > h || S > w || stride < 1) return DRS_ERR_ARG;
  StitchArgs a;
  a.prob = prob; a.occur = occur; a.logits = logits; a.h = h; a.w = w; a.K = K; a.S = S; a.stride = stride;
  a.n_h = (h - S) % stride == 0 ? (h - S) / stride + 1 : (h - S) / stride + 2;
  a.n_w = (w - S) % stride == 0 ? (w - S) / stride + 1 : (w - S) / stride + 2;
  if (first_window < 0 || n_windows < 1 || first_window + n_windows > a.n_h * a.n_w) return DRS_ERR_ARG;
  a.f0 = first_window; a.nb = n_windows;
  const int r_first = first_window / a.n_w, r_last = (first_window + n_windows - 1) / a.n_w;
  const int y0 = r_first * stride < h - S ? r_first * stride : h - S;
  const int y1 = (r_last * stride < h - S ? r_last * stride : h - S) + S;
  a.row0 = y0; a.nrows = y1 - y0;
  dim3 grid((w + 255) / 256, a.nrows);
  DRS_LAUNCH(stitch_accumulate_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return DRS_LAUNCH_CHECK();
}

int drs_stitch_finalize(const float* prob, const unsigned int* occur, int h, int w, int K, unsigned char* out, void* stream) {
  if (!prob || !occur || !out || K < 1 || K > 8) return DRS_ERR_ARG;
  const size_t n = (size_t)h * w;
  const size_t nb = (n + 255) / 256;
  DRS_LAUNCH(stitch_finalize_kernel, dim3(nb < 4096 ? (unsigned)nb : 4096u), dim3(256), 0, (hipStream_t)stream, prob, occur,
                     n, K, out);
  return DRS_LAUNCH_CHECK();
}

int drs_softmax_accumulate(const float* prob, const unsigned int* occur, int h, int w, int K, float* acc, void* stream) {
  if (!prob || !occur || !acc || K < 1 || K > 8) return DRS_ERR_ARG;
  const size_t n = (size_t)h * w;
  const size_t nb = (n + 255) / 256;
  DRS_LAUNCH(softmax_accumulate_kernel, dim3(nb < 4096 ? (unsigned)nb : 4096u), dim3(256), 0, (hipStream_t)stream, prob, occur, n, K, acc);
  return DRS_LAUNCH_CHECK();
}

}  // extern "C"
