// What a kernel with the ADI sweeps' access pattern can reach on one MI355X, without their arithmetic:
//   hipcc --offload-arch=gfx950 -O3 tools/tile_ceiling.hip -o /tmp/tile_ceiling && /tmp/tile_ceiling
// One wave per ROWS x 64 tile of an N x N fp64 plane: ROWS row-segment loads of 512 B (lane <-> column), all loads before
// all stores (the solve needs the whole chunk), in place, after a warm-up that brings the clocks to their sustained state.  Variants: tile height 64 / 32 / 16, waves per SIMD forced
// through the register budget, cached vs non-temporal accesses.  Prints the time of one pass and 16 N^2 B / time.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ROWS, int NT, int WAVES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
tile_rmw(double* __restrict__ a, int n, int tiles_x) {
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  double* p = a + (long)ty * ROWS * n + tx * 64 + threadIdx.x;
  double v[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) v[r] = (NT & 1) ? __builtin_nontemporal_load(p + (long)r * n) : p[(long)r * n];
  // a dependent chain through all rows, like the forward / backward substitution (keeps loads before stores)
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) { s = fma(s, 0.25, 0.75 * v[r]); v[r] = s; }      // a smoothing recurrence: values stay O(input)
#pragma unroll
  for (int r = ROWS - 1; r >= 0; --r) { s = fma(s, 0.25, 0.75 * v[r]); v[r] = s; }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (NT & 2) __builtin_nontemporal_store(v[r], p + (long)r * n);
    else p[(long)r * n] = v[r];
  }
}

template <int ROWS, int NT, int WAVES>
static void run(double* a, int n, const char* tag) {
  const int tiles_x = n / 64, tiles = tiles_x * (n / ROWS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) tile_rmw<ROWS, NT, WAVES><<<tiles, 64>>>(a, n, tiles_x);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) tile_rmw<ROWS, NT, WAVES><<<tiles, 64>>>(a, n, tiles_x);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = 1e3 * ms / reps;
  printf("N=%5d rows=%2d nt=%d waves/SIMD=%d %-8s tiles=%6d  %8.2f us  %6.2f TB/s\n", n, ROWS, NT, WAVES, tag, tiles, us,
         16.0 * n * n / us / 1e6);
}

__global__ void fill(double* a, long n, double scale) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
    a[t] = scale * (1.0 + 1e-3 * (double)((t * 2654435761u) % 1000));
}

// The rate depends on the DATA: an all-zero plane runs 14 % faster than a plane of ordinary numbers (36.7 vs 41.8 us at
// 4096^2, reproducibly, whatever ran before) - zero data is what hipMemset leaves and what a quick micro-benchmark
// measures.  Argument "zero" reproduces that; the default fills the plane with 1e-4 (1 + hash), like the benchmark fields.
int main(int argc, char** argv) {
  double* a;
  const long nmax = 16384;
  hipMalloc(&a, nmax * nmax * 8);
  hipMemset(a, 0, nmax * nmax * 8);
  const bool zero = argc > 1 && argv[1][0] == 'z';
  if (!zero) fill<<<8192, 256>>>(a, nmax * nmax, 1e-4);
  printf("data: %s\n", zero ? "all zero" : "1e-4 (1 + hash)");
  // half a second of the same work first: the first milliseconds after an idle period run at boost clocks and read ~13 %
  // faster (36.4 vs 41.7 us at 4096^2) than the sustained rate every real time loop sees
  for (int r = 0; r < 12000; ++r) tile_rmw<64, 0, 2><<<4096, 64>>>(a, 4096, 64);
  hipDeviceSynchronize();
  const int sizes[] = {1024, 2048, 2880, 4096, 5760, 8192, 16384};
  for (int n : sizes) {
    run<64, 0, 2>(a, n, "cached");
    run<64, 2, 2>(a, n, "nt-st");
    run<64, 3, 2>(a, n, "nt-both");
    run<32, 0, 2>(a, n, "cached");
    run<32, 0, 4>(a, n, "cached");
    run<32, 2, 4>(a, n, "nt-st");
    run<32, 3, 4>(a, n, "nt-both");
    run<16, 0, 8>(a, n, "cached");
    run<16, 3, 8>(a, n, "nt-both");
    printf("\n");
  }
  return 0;
}
