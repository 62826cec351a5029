// Microbenchmark: FLOP/s of bare MFMA loops on random register operands, per MFMA shape.  Cycles per FLOP are the same for the two
// shapes of a type; what differs is the clock the chip holds under them (MI355X_MICROARCH.md, DVFS give-back (7)).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shape.hip -o tools/ubench/mfma_shape && tools/ubench/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 64 x 64 wave tile, fp32: 2 x 2 tiles of 32x32x2 (4 accumulators), K advanced by 2 per MFMA group
__device__ int g_prio_shift = -1;
__device__ __forceinline__ void static_prio() {
  const int sh = g_prio_shift;
  if (sh >= 0) {
    const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> sh) & 3);
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 3) __builtin_amdgcn_s_setprio(3);
  }
}
template <int OCC>
__global__ __launch_bounds__(256, OCC) void f32_32(const float* in, float* out, int iters) {
  static_prio();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a[2][8], b[2][8];
  for (int i = 0; i < 2; ++i) for (int k = 0; k < 8; ++k) { a[i][k] = in[(threadIdx.x * 16 + i * 8 + k) & 65535]; b[i][k] = in[(threadIdx.x * 16 + 4096 + i * 8 + k) & 65535]; }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][k], b[ni][k], acc[mi * 2 + ni], 0, 0, 0);
  float r = 0.f;
  for (int t = 0; t < 4; ++t) r += acc[t][t];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
// the same tile and K range as 4 x 4 tiles of 16x16x4 (16 accumulators): 16 MFMAs per 4 k, i.e. per 2 of the groups above
__global__ __launch_bounds__(256) void f32_16(const float* in, float* out, int iters) {
  f32x4 acc[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
  float a[4][4], b[4][4];
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) { a[i][k] = in[(threadIdx.x * 16 + i * 4 + k) & 65535]; b[i][k] = in[(threadIdx.x * 16 + 4096 + i * 4 + k) & 65535]; }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][k], b[ni][k], acc[mi * 4 + ni], 0, 0, 0);
  float r = 0.f;
  for (int t = 0; t < 16; ++t) r += acc[t][t & 3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void bf16_32(const float* in, float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 a[2][2], b[2][2];
  for (int i = 0; i < 2; ++i) for (int k = 0; k < 2; ++k) for (int e = 0; e < 8; ++e) {
    a[i][k][e] = (__bf16)in[(threadIdx.x * 64 + i * 16 + k * 8 + e) & 65535]; b[i][k][e] = (__bf16)in[(threadIdx.x * 64 + 32 + i * 16 + k * 8 + e) & 65535]; }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][k], b[ni][k], acc[mi * 2 + ni], 0, 0, 0);
  float r = 0.f;
  for (int t = 0; t < 4; ++t) r += acc[t][t];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void bf16_16(const float* in, float* out, int iters) {
  f32x4 acc[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)in[(threadIdx.x * 64 + i * 8 + e) & 65535]; b[i][e] = (__bf16)in[(threadIdx.x * 64 + 32 + i * 8 + e) & 65535]; }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi * 4 + ni], 0, 0, 0);
  float r = 0.f;
  for (int t = 0; t < 16; ++t) r += acc[t][t & 3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <class K>
double run(K kern, const float* in, float* out, int blocks, int iters, double flop_per_wave_iter, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  double best = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = flop_per_wave_iter * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;
    if (tf > best) best = tf;
  }
  printf("%-28s %8.1f TFLOP/s\n", name, best);
  return best;
}

int main(int argc, char** argv) {
  const int zero = argc > 1 && atoi(argv[1]) == 0;
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = zero ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *out;
  hipMalloc(&in, 65536 * 4); hipMalloc(&out, 4096 * 256 * 4);
  hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  const int per_cu = argc > 2 ? atoi(argv[2]) : 4;
  const int blocks = 256 * per_cu;   // default: 4 workgroups of 4 waves per CU, as the convolution kernels run
  const int shift = argc > 3 ? atoi(argv[3]) : -1;
  hipMemcpyToSymbol(HIP_SYMBOL(g_prio_shift), &shift, sizeof(int));
  printf("%d workgroups of 4 waves per CU, static priority shift %d\n", per_cu, shift);
  printf("%s operands\n", zero ? "all-zero" : "random");
  for (int round = 0; round < 2; ++round) {
    run(f32_32<1>, in, out, blocks, 2000, 8 * 4 * 2.0 * 32 * 32 * 2, "f32  32x32x2 (AGPR acc)");
    run(f32_32<3>, in, out, blocks, 2000, 8 * 4 * 2.0 * 32 * 32 * 2, "f32  32x32x2 (VGPR acc)");
    run(f32_16, in, out, blocks, 2000, 4 * 16 * 2.0 * 16 * 16 * 4, "f32  16x16x4");
    run(bf16_32, in, out, blocks, 4000, 2 * 4 * 2.0 * 32 * 32 * 16, "bf16 32x32x16");
    run(bf16_16, in, out, blocks, 4000, 16 * 2.0 * 16 * 16 * 32, "bf16 16x16x32");
  }
  return 0;
}
