// Environment effects on the 4096^2 in-place tile copy (tools/tile_occupancy.hip, 2 waves per SIMD, chain 1):
//   buffer: first 128 MiB of a 2 GiB allocation | its own 128 MiB hipMalloc;   stream: NULL | created stream;
//   kernels: one kernel repeated | two identical kernels alternating | with a 1-block tiny kernel in between
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ID>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) rmw(double* __restrict__ a, int n, double seed = 0.0) {
  const int tiles_x = n >> 6;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  double* p = a + (long)ty * 64 * n + tx * 64 + threadIdx.x;
  double v[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) v[r] = p[(long)r * n];
  double s = ID + seed;
#pragma unroll
  for (int r = 0; r < 64; ++r) { s = fma(s, 0.25, v[r]); v[r] = s; }
#pragma unroll
  for (int r = 63; r >= 0; --r) { s = fma(s, 0.25, v[r]); v[r] = s; }
#pragma unroll
  for (int r = 0; r < 64; ++r) p[(long)r * n] = v[r];
}
__global__ void tiny(double* a) { if (threadIdx.x == 999) a[0] = 1.0; }
__global__ void fill(double* a, long n, double scale) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
    a[t] = scale * (1.0 + 1e-3 * (double)(t % 1000));
}

static void run(const char* tag, double* a, hipStream_t st, int mode) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto one = [&](int i) {
    if (mode == 1 && (i & 1)) rmw<1><<<4096, 64, 0, st>>>(a, 4096);
    else if (mode == 3) rmw<0><<<4096, 64, 0, st>>>(a, 4096, (double)(i & 1));          // same kernel, alternating argument
    else if (mode == 4 && (i & 2)) rmw<1><<<4096, 64, 0, st>>>(a, 4096);               // A A B B A A B B
    else if (mode == 5) { if (i & 1) rmw<1><<<4095, 64, 0, st>>>(a, 4096); else rmw<0><<<4096, 64, 0, st>>>(a, 4096); }
    else rmw<0><<<4096, 64, 0, st>>>(a, 4096);
    if (mode == 2) tiny<<<1, 64, 0, st>>>(a);
  };
  for (int r = 0; r < 4; ++r) one(r);
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int r = 0; r < 40; ++r) one(r);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.2f us per sweep\n", tag, 1e3 * ms / 40);
}

int main() {
  double *big, *own;
  hipMalloc(&big, 2048L << 20);
  hipMalloc(&own, 128L << 20);
  hipMemset(big, 0, 2048L << 20);
  hipMemset(own, 0, 128L << 20);
  hipStream_t st;
  hipStreamCreate(&st);
  run("2 GiB allocation, NULL stream, one kernel", big, 0, 0);
  run("own 128 MiB allocation, NULL stream, one kernel", own, 0, 0);
  run("own 128 MiB allocation, created stream, one kernel", own, st, 0);
  run("own 128 MiB allocation, created stream, two kernels alternating", own, st, 1);
  run("own 128 MiB allocation, created stream, tiny kernel between", own, st, 2);
  run("same kernel, argument alternating", own, st, 3);
  run("two kernels, A A B B", own, st, 4);
  run("two kernels alternating, grids 4096 / 4095", own, st, 5);
  run("one kernel again", own, st, 0);
  hipMemset(own, 0, 128L << 20);
  run("one kernel, buffer zeroed again", own, st, 0);
  fill<<<4096, 256>>>(own, 4096L * 4096, 1e-4);
  hipDeviceSynchronize();
  run("one kernel, buffer filled with 1e-4 (1 + ...)", own, st, 0);
  hipMemset(own, 0, 128L << 20);
  run("one kernel, zeroed once more", own, st, 0);
  return 0;
}
