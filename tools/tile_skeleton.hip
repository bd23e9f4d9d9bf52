// Which part of the ADI kernels' skeleton (beyond loads, a dependent chain and stores) costs time?
//   hipcc --offload-arch=gfx950 -O3 tools/tile_skeleton.hip -o tools/bin/tile_skeleton && tools/bin/tile_skeleton
// Variants of tools/tile_ceiling.hip's 64-row kernel at N = 4096 (in place):
//   bit 0: 16.9 KB of static LDS per block (the transposes' buffer, touched once)
//   bit 1: a 256-byte by-value kernel argument + tile coordinates by integer division of blockIdx
//   bit 2: 200 VGPRs forced live (two waves per SIMD by register count instead of by attribute)
//   bit 3: two LDS round trips of the tile (the transposes' traffic: 64 ds_write_b64 + 64 ds_read_b64 each)
#include <hip/hip_runtime.h>
#include <cstdio>

struct BigArg {
  int ny, nx, nfield, py, px, gny, gnx, j0, i0, gpy, gpx, stream;
  const double* p[16];
  double s[8];
  int tail[8];
};

template <int V>
__global__ void __launch_bounds__(64) skel(BigArg a, double* __restrict__ buf) {
  __shared__ double lds[2112];
  int ty, tx;
  if (V & 2) {
    int id = blockIdx.x;
    tx = id % a.px; id /= a.px; ty = id % a.py;
  } else {
    ty = blockIdx.x >> 6; tx = blockIdx.x & 63;
  }
  const int n = (V & 2) ? a.nx : 4096;
  double* p = buf + (long)ty * 64 * n + tx * 64 + threadIdx.x;
  double v[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) v[r] = p[(long)r * n];
  if (V & 1) { lds[threadIdx.x] = v[0]; __syncthreads(); v[1] += lds[63 - threadIdx.x]; }
  if (V & 8) {
    const int l = threadIdx.x & 31;
    double* blk = lds + (threadIdx.x >> 5) * 1056;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int k = 0; k < 32; ++k) blk[k * 33 + l] = v[half * 32 + k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) v[half * 32 + k] = blk[l * 33 + k];
        __syncthreads();
      }
  }
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 64; ++r) { s = fma(s, 0.25, v[r]); v[r] = s; }
#pragma unroll
  for (int r = 63; r >= 0; --r) { s = fma(s, 0.25, v[r]); v[r] = s; }
  if (V & 4) {
    double extra[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) extra[k] = v[k] * 1.5;
#pragma unroll
    for (int k = 0; k < 36; ++k) asm volatile("" : "+v"(extra[k]));
#pragma unroll
    for (int k = 0; k < 36; ++k) v[k] += extra[k] * 1e-300;
  }
#pragma unroll
  for (int r = 0; r < 64; ++r) p[(long)r * n] = v[r];
}

template <int V>
static void run(double* a) {
  BigArg arg{};
  arg.ny = arg.nx = 4096; arg.py = arg.px = 64; arg.nfield = 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) skel<V><<<4096, 64>>>(arg, a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 40; ++r) skel<V><<<4096, 64>>>(arg, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("variant %2d  %7.2f us\n", V, 1e3 * ms / 40);
}

int main() {
  double* a;
  hipMalloc(&a, 4096L * 4096 * 8);
  hipMemset(a, 0, 4096L * 4096 * 8);
  run<0>(a); run<1>(a); run<2>(a); run<4>(a); run<8>(a); run<3>(a); run<5>(a); run<7>(a); run<9>(a); run<13>(a); run<15>(a); run<0>(a);
  return 0;
}
