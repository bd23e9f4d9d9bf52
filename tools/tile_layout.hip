// Does a tile-blocked layout of the carried plane (each 64 x 64 tile 32 KB contiguous) beat the row-major plane
// (64 row segments of 512 B, pitch N * 8 B) for the ADI access pattern?   tools/bin/tile_layout
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BLOCKED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) rmw(double* __restrict__ a, int n) {
  const int tiles_x = n / 64;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  double* p = BLOCKED ? a + (long)blockIdx.x * 4096 + threadIdx.x : a + (long)ty * 64 * n + tx * 64 + threadIdx.x;
  const long pitch = BLOCKED ? 64 : n;
  double v[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) v[r] = p[r * pitch];
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 64; ++r) { s = fma(s, 0.25, v[r]); v[r] = s; }
#pragma unroll
  for (int r = 63; r >= 0; --r) { s = fma(s, 0.25, v[r]); v[r] = s; }
#pragma unroll
  for (int r = 0; r < 64; ++r) p[r * pitch] = v[r];
}

template <int BLOCKED>
static void run(double* a, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles = (n / 64) * (n / 64);
  for (int r = 0; r < 3; ++r) rmw<BLOCKED><<<tiles, 64>>>(a, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) rmw<BLOCKED><<<tiles, 64>>>(a, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("N=%5d %s  %8.2f us  %6.2f TB/s\n", n, BLOCKED ? "tile-blocked" : "row-major   ", 1e3 * ms / 20, 16.0 * n * n / (1e3 * ms / 20) / 1e6);
}

int main() {
  double* a;
  hipMalloc(&a, 16384L * 16384 * 8);
  hipMemset(a, 0, 16384L * 16384 * 8);
  for (int n : {1024, 2048, 4096, 8192, 16384}) { run<0>(a, n); run<1>(a, n); }
  return 0;
}
