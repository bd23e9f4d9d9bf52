// Does a tile written by one launch come back faster when the NEXT launch reads it from the same XCD?
//   hipcc --offload-arch=gfx950 -O3 tools/tile_xcd.hip -o tools/bin/tile_xcd && tools/bin/tile_xcd
// One wave per 32 x 64 tile of an N x N fp64 plane, in place, as tools/tile_ceiling.hip; consecutive launches alternate
// between the mapping tile = block and tile = (block + shift) mod tiles.  Blocks b and b + 8 share an XCD (round-robin
// dispatch), so shift = 8 k keeps every tile on its XCD and any other shift moves all of them.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ROWS = 32;

__global__ void __launch_bounds__(64) tile_rmw(double* __restrict__ a, int n, int tiles_x, int tiles, int shift) {
  int id = blockIdx.x + shift;
  if (id >= tiles) id -= tiles;
  const int ty = id / tiles_x, tx = id % tiles_x;
  double* p = a + (long)ty * ROWS * n + tx * 64 + threadIdx.x;
  double v[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) v[r] = p[(long)r * n];
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) { s = fma(s, 0.25, 0.75 * v[r]); v[r] = s; }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) p[(long)r * n] = v[r];
}

__global__ void fill(double* a, long n, double scale) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
    a[t] = scale * (1.0 + 1e-3 * (double)((t * 2654435761u) % 1000));
}

int main() {
  double* a;
  const long nmax = 4096;
  hipMalloc(&a, nmax * nmax * 8);
  fill<<<8192, 256>>>(a, nmax * nmax, 1e-4);
  for (int r = 0; r < 12000; ++r) tile_rmw<<<4096 * 2, 64>>>(a, 4096, 64, 8192, 0);     // sustained clocks
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int sizes[] = {1024, 2048, 2880, 4096};
  const int shifts[] = {0, 8, 64, 1, 3, 4, 100};
  for (int n : sizes) {
    const int tiles_x = n / 64, tiles = tiles_x * (n / ROWS);
    for (int shift : shifts) {
      for (int r = 0; r < 10; ++r) tile_rmw<<<tiles, 64>>>(a, n, tiles_x, tiles, (r & 1) ? shift : 0);
      hipDeviceSynchronize();
      const int reps = 40;
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) tile_rmw<<<tiles, 64>>>(a, n, tiles_x, tiles, (r & 1) ? shift : 0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("N=%5d tiles=%6d shift=%4d  %8.2f us  %6.2f TB/s\n", n, tiles, shift, 1e3 * ms / reps, 16.0 * n * n / (1e3 * ms / reps) / 1e6);
    }
    printf("\n");
  }
  return 0;
}
