// Does a third wave per SIMD help once the waves spend part of their time computing (as the ADI kernels do)?
//   tools/bin/tile_occupancy      (N = 4096, in place, 64 x 64 tile per wave, dependent chains of CH * 128 FMAs between
//   the loads and the stores; W waves per SIMD by register budget)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int W, int CH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W))) rmw(double* __restrict__ a) {
  const int ty = blockIdx.x >> 6, tx = blockIdx.x & 63;
  double* p = a + (long)ty * 64 * 4096 + tx * 64 + threadIdx.x;
  double v[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) v[r] = p[(long)r * 4096];
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int r = 0; r < 64; ++r) { s = fma(s, 0.25, v[r]); v[r] = s; }
#pragma unroll
    for (int r = 63; r >= 0; --r) { s = fma(s, 0.25, v[r]); v[r] = s; }
  }
#pragma unroll
  for (int r = 0; r < 64; ++r) p[(long)r * 4096] = v[r];
}

template <int W, int CH>
static void run(double* a) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) rmw<W, CH><<<4096, 64>>>(a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 30; ++r) rmw<W, CH><<<4096, 64>>>(a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD %d  chains %d  %7.2f us\n", W, CH, 1e3 * ms / 30);
}

int main() {
  double* a;
  hipMalloc(&a, 4096L * 4096 * 8);
  hipMemset(a, 0, 4096L * 4096 * 8);
  run<2, 1>(a); run<3, 1>(a); run<2, 4>(a); run<3, 4>(a); run<2, 8>(a); run<3, 8>(a); run<1, 4>(a); run<1, 8>(a);
  return 0;
}
