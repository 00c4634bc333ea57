// Microbenchmark behind DESIGN 22.3: what does HBM give a 2-read : 1-write stream of the forward products' size,
// as a function of how the work is handed to workgroups?  No arithmetic to speak of, no LDS.
//   mode 0: one float4 per thread, grid = elements / 256 (the "copy" shape)
//   mode 1: persistent workgroups (G), round-robin 64-row tiles
//   mode 2: persistent workgroups (G), one contiguous range of tiles each (what the product kernels do)
//   mode 3: as 2, but the tile's loads are all issued, then a barrier, then all stores (the bursty shape)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/stream_mix.hip -o tools/micro/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kCols4 = 16;  // 64 floats per row

__global__ void __launch_bounds__(256) k_flat(const float4* a, const float4* b, float4* o, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const float4 x = a[i], y = b[i];
    o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

template <int MODE, int NT>
__global__ void __launch_bounds__(NT) k_tiles(const float4* a, const float4* b, float4* o, int n_tiles) {
  constexpr int PASS = 64 * kCols4 / NT;
  const int G = gridDim.x, w = blockIdx.x;
  const int t0 = MODE == 1 ? w : (int)((long long)w * n_tiles / G);
  const int t1 = MODE == 1 ? n_tiles : (int)((long long)(w + 1) * n_tiles / G);
  const int step = MODE == 1 ? G : 1;
  float4 x[PASS], y[PASS];
  for (int t = t0; t < t1; t += step) {
    const size_t base = (size_t)t * 64 * kCols4;
#pragma unroll
    for (int p = 0; p < PASS; ++p) {
      x[p] = a[base + threadIdx.x + p * NT];
      y[p] = b[base + threadIdx.x + p * NT];
    }
    if (MODE == 3) __syncthreads();
#pragma unroll
    for (int p = 0; p < PASS; ++p)
      o[base + threadIdx.x + p * NT] = make_float4(x[p].x + y[p].x, x[p].y + y[p].y, x[p].z + y[p].z, x[p].w + y[p].w);
    if (MODE == 3) __syncthreads();
  }
}

int main(int argc, char** argv) {
  const size_t rows = argc > 1 ? atoll(argv[1]) : 1198503;
  const int n_tiles = (int)(rows / 64);
  const size_t n4 = (size_t)n_tiles * 64 * kCols4;
  float4 *a, *b, *o;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&o, n4 * 16));
  CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, us, 3.0 * n4 * 16 / us / 1e6);
  };
  run("flat, one float4 per thread", [&] { hipLaunchKernelGGL(k_flat, dim3((n4 + 255) / 256), dim3(256), 0, 0, a, b, o, n4); });
  for (int G : {256, 512, 1024, 2048}) {
    char nm[96];
    snprintf(nm, sizeof nm, "round-robin tiles, %d x 256 threads", G);
    run(nm, [&] { hipLaunchKernelGGL((k_tiles<1, 256>), dim3(G), dim3(256), 0, 0, a, b, o, n_tiles); });
    snprintf(nm, sizeof nm, "contiguous ranges, %d x 256 threads", G);
    run(nm, [&] { hipLaunchKernelGGL((k_tiles<2, 256>), dim3(G), dim3(256), 0, 0, a, b, o, n_tiles); });
    snprintf(nm, sizeof nm, "contiguous, load | barrier | store, %d x 256", G);
    run(nm, [&] { hipLaunchKernelGGL((k_tiles<3, 256>), dim3(G), dim3(256), 0, 0, a, b, o, n_tiles); });
  }
  for (int G : {256, 512}) {
    char nm[96];
    snprintf(nm, sizeof nm, "contiguous, load | barrier | store, %d x 512", G);
    run(nm, [&] { hipLaunchKernelGGL((k_tiles<3, 512>), dim3(G), dim3(512), 0, 0, a, b, o, n_tiles); });
    snprintf(nm, sizeof nm, "round-robin, %d x 512 threads", G);
    run(nm, [&] { hipLaunchKernelGGL((k_tiles<1, 512>), dim3(G), dim3(512), 0, 0, a, b, o, n_tiles); });
  }
  return 0;
}
