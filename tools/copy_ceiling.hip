// Copy-bandwidth ceiling of one MI355X for the access patterns of this library:
//   hipcc --offload-arch=gfx950 -O3 tools/copy_ceiling.hip -o copy_ceiling && ./copy_ceiling
// flat: contiguous fp64 copy; "planes": every thread touches one cell of each of 47 planes of 4096^2 (the collision kernel at
// NE = 12, Nw = 35), either load-store per plane ("loop") or all loads before all stores ("batched").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// copy NP planes: thread p handles VEC consecutive cells of every plane (plane-strided streams per wave)
template <int VEC>
__global__ void __launch_bounds__(128) copy_planes(const double* __restrict__ in, double* __restrict__ out, long ncell, int np) {
  const long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (p >= ncell) return;
  for (int i = 0; i < np; ++i) {
    if (VEC == 1) { out[(long)i * ncell + p] = in[(long)i * ncell + p] * 1.0000001; }
    else if (VEC == 2) { double2 v = *(const double2*)(in + (long)i * ncell + p); v.x *= 1.0000001; v.y *= 1.0000001; *(double2*)(out + (long)i * ncell + p) = v; }
    else { double4 v = *(const double4*)(in + (long)i * ncell + p); v.x *= 1.0000001; *(double4*)(out + (long)i * ncell + p) = v; }
  }
}
// all loads first (like the collision kernel: many streams in flight), then stores
template <int VEC, int NP>
__global__ void __launch_bounds__(128) copy_planes_batched(const double* __restrict__ in, double* __restrict__ out, long ncell) {
  const long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (p >= ncell) return;
  double v[NP][VEC];
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[i][k] = __builtin_nontemporal_load(&in[(long)i * ncell + p + k]);
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int k = 0; k < VEC; ++k) __builtin_nontemporal_store(v[i][k] * 1.0000001, &out[(long)i * ncell + p + k]);
}
__global__ void fill(double* a, long n) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
    a[t] = 1e-4 * (1.0 + 1e-3 * (double)((t * 2654435761u) % 1000));
}
// The rate depends on the data (see tools/tile_ceiling.hip): argument "zero" copies an all-zero buffer (what round 1
// measured: 5.99 TB/s flat), the default a buffer of ordinary numbers.
int main(int argc, char** argv) {
  const long ncell = 4096L * 4096; const int np = 47;
  double *in, *out; hipMalloc(&in, ncell * np * 8); hipMalloc(&out, ncell * np * 8); hipMemset(in, 0, ncell * np * 8);
  const bool zero = argc > 1 && argv[1][0] == 'z';
  if (!zero) { fill<<<8192, 256>>>(in, ncell * np); hipDeviceSynchronize(); }
  printf("data: %s\n", zero ? "all zero" : "1e-4 (1 + hash)");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5; printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, 2.0 * ncell * np * 8 / ms / 1e9);
  };
  timeit("loop vec1", [&] { copy_planes<1><<<ncell / 128, 128>>>(in, out, ncell, np); });
  timeit("loop vec2", [&] { copy_planes<2><<<ncell / 256, 128>>>(in, out, ncell, np); });
  timeit("loop vec4", [&] { copy_planes<4><<<ncell / 512, 128>>>(in, out, ncell, np); });
  timeit("batched vec1 (47 in flight)", [&] { copy_planes_batched<1, 47><<<ncell / 128, 128>>>(in, out, ncell); });
  timeit("batched vec2", [&] { copy_planes_batched<2, 47><<<ncell / 256, 128>>>(in, out, ncell); });
  // contiguous copy of the same bytes
  timeit("flat copy vec1", [&] { copy_planes<1><<<ncell * np / 128, 128>>>(in, out, ncell * np, 1); });
  timeit("flat copy vec2", [&] { copy_planes<2><<<ncell * np / 256, 128>>>(in, out, ncell * np, 1); });
  return 0;
}
