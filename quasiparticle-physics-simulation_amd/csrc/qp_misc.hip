// Small elementwise / reduction kernels of the time loop: external generation, Pauli guard statistics,
// energy integrals, max-norm.
#include "qp_common.h"

namespace qp {

__global__ void __launch_bounds__(256) add_constant_kernel(const uint8_t* __restrict__ flags, long ncell,
                                                           int nfield, double* __restrict__ s, double amount) {
  const long total = ncell * nfield;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long p = t % ncell;
    if (flags[p] & QP_FLAG_ACTIVE) s[t] += amount;
  }
}

// out[f][p] = inside the mask ? scale * in[f][p] : NaN  (frames as the reference returns them, reconstruct_field)
__global__ void __launch_bounds__(256) nan_pad_kernel(const uint8_t* __restrict__ flags, long ncell, int nfield,
                                                      const double* __restrict__ in, double scale,
                                                      double* __restrict__ out) {
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < ncell; p += (long)gridDim.x * blockDim.x) {
    const bool on = flags[p] & QP_FLAG_ACTIVE;
    for (int f = 0; f < nfield; ++f) out[(long)f * ncell + p] = on ? scale * in[(long)f * ncell + p] : qnan;
  }
}

__global__ void __launch_bounds__(256) add_scaled_kernel(long n, double* __restrict__ s,
                                                         const double* __restrict__ g, double scale) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
    s[t] += scale * g[t];
}

__global__ void __launch_bounds__(256) energy_integrate_kernel(const double* __restrict__ s, int ne, long ncell,
                                                               double dE, double* __restrict__ out) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < ncell; p += (long)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int i = 0; i < ne; ++i) acc += s[(long)i * ncell + p];
    out[p] = acc * dE;
  }
}

__global__ void __launch_bounds__(256) weighted_sum_kernel(const double* __restrict__ s,
                                                           const double* __restrict__ w, int n, long ncell,
                                                           double* __restrict__ out) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < ncell; p += (long)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += s[(long)i * ncell + p] * w[i];
    out[p] = acc;
  }
}

// ---- reductions: per-block partials in workspace, finished by a single-block kernel -----------------------
constexpr int kRedBlocks = 1024;

// np.argmax order (solver.py:993): a NaN occupation is the maximum and the first NaN in C order wins; otherwise the
// largest value, the smallest linear index on ties
__device__ __forceinline__ bool pauli_better(double of, long ofi, double f, long fi) {
  const bool on = of != of, fn = f != f;
  if (on || fn) return on && (!fn || ofi < fi);
  return of > f || (of == f && ofi < fi);
}

__device__ __forceinline__ void pauli_merge(double& f, long& fi, long& forb, double of, long ofi, long oforb) {
  if (pauli_better(of, ofi, f, fi)) { f = of; fi = ofi; }
  if (oforb >= 0 && (forb < 0 || oforb < forb)) forb = oforb;
}

__device__ void pauli_block_reduce(double f, long fi, long forb, PauliPartial* dst) {
  __shared__ double sf[256];
  __shared__ long si[256];
  __shared__ long sb[256];
  const int t = threadIdx.x;
  sf[t] = f; si[t] = fi; sb[t] = forb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) pauli_merge(sf[t], si[t], sb[t], sf[t + s], si[t + s], sb[t + s]);
    __syncthreads();
  }
  if (t == 0) { dst->maxf = sf[0]; dst->maxidx = si[0]; dst->forb = sb[0]; }
}

__global__ void __launch_bounds__(256) pauli_partial_kernel(const double* __restrict__ s,
                                                            const double* __restrict__ rho,
                                                            const int32_t* __restrict__ cls,
                                                            const uint8_t* __restrict__ flags, int ne, long ncell,
                                                            double floor_, PauliPartial* part) {
  // f defaults to 0 where rho <= 1e-30 (np.divide(..., where=rho_mask) into zeros, solver.py:987-992);
  // np.argmax returns the first maximum in C order, so ties resolve to the smallest linear index.
  double f = -__builtin_huge_val();
  long fi = 0x7fffffffffffffffL, forb = -1;
  // blockIdx.y strides over energy bins, blockIdx.x over cells: no 64-bit div / mod per element.  Four independent
  // cells per trip so that four loads are in flight per thread (the loop is latency-bound otherwise: ~1.9 TB/s).
  const long stride = (long)gridDim.x * blockDim.x;
  for (int i = blockIdx.y; i < ne; i += gridDim.y) {
    const double* si = s + (long)i * ncell;
    const double r0 = rho[i];
    for (long p0 = (long)blockIdx.x * blockDim.x + threadIdx.x; p0 < ncell; p0 += 4 * stride) {
      double nv[4];
      unsigned fl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long p = p0 + u * stride;
        const bool in = p < ncell;
        fl[u] = in ? flags[p] : 0u;
        nv[u] = in ? si[p] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!(fl[u] & QP_FLAG_ACTIVE)) continue;
        const long p = p0 + u * stride;
        const double r = cls ? rho[(long)cls[p] * ne + i] : r0;
        const double n = nv[u];
        const long t = (long)i * ncell + p;
        double occ = 0.0;
        if (r > 1e-30) occ = n / fmax(r, 1e-30);
        else if (n > floor_ && (forb < 0 || t < forb)) forb = t;
        if (pauli_better(occ, t, f, fi)) { f = occ; fi = t; }
      }
    }
  }
  pauli_block_reduce(f, fi, forb, part + (long)blockIdx.y * gridDim.x + blockIdx.x);
}

__global__ void __launch_bounds__(256) pauli_final_kernel(const PauliPartial* part, int nparts, double* out_vals,
                                                          long* out_idx) {
  double f = -__builtin_huge_val();
  long fi = 0x7fffffffffffffffL, forb = -1;
  for (int k = threadIdx.x; k < nparts; k += blockDim.x) pauli_merge(f, fi, forb, part[k].maxf, part[k].maxidx, part[k].forb);
  __shared__ PauliPartial res;
  pauli_block_reduce(f, fi, forb, &res);
  __syncthreads();
  if (threadIdx.x == 0) { out_vals[0] = res.maxf; out_idx[0] = res.maxidx; out_idx[1] = res.forb; }
}

// middle stage for the per-wave partials of the fused guard (collision kernels): block b reduces its contiguous share
__global__ void __launch_bounds__(256) pauli_merge_kernel(const PauliPartial* part, long nparts, PauliPartial* out) {
  double f = -__builtin_huge_val();
  long fi = 0x7fffffffffffffffL, forb = -1;
  const long per = (nparts + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = lo + per < nparts ? lo + per : nparts;
  for (long k = lo + threadIdx.x; k < hi; k += blockDim.x) pauli_merge(f, fi, forb, part[k].maxf, part[k].maxidx, part[k].forb);
  pauli_block_reduce(f, fi, forb, out + blockIdx.x);
}

void pauli_finish(const PauliPartial* parts, long nparts, PauliPartial* scratch, double* out_vals, long* out_idx,
                  hipStream_t stream) {
  hipLaunchKernelGGL(pauli_merge_kernel, dim3(kGuardMergeBlocks), dim3(256), 0, stream, parts, nparts, scratch);
  hipLaunchKernelGGL(pauli_final_kernel, dim3(1), dim3(256), 0, stream, (const PauliPartial*)scratch, kGuardMergeBlocks,
                     out_vals, out_idx);
}

__global__ void __launch_bounds__(256) absmax_partial_kernel(const double* __restrict__ a, long n, double* part) {
  double m = 0.0;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const double v = fabs(a[t]);
    // NaN must not be lost: propagate it as +inf so a diverged iteration is visible
    m = (v != v) ? __builtin_huge_val() : fmax(m, v);
  }
  __shared__ double sm[256];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

__global__ void __launch_bounds__(256) absmax_final_kernel(const double* part, int nparts, double* out) {
  double m = 0.0;
  for (int k = threadIdx.x; k < nparts; k += blockDim.x) m = fmax(m, part[k]);
  __shared__ double sm[256];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sm[0];
}

void absmax_finish(const double* parts, int nparts, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(256), 0, stream, parts, nparts, out);
}

// direction and iterate update of the Chebyshev iteration in one pass: d = c1 z + c2 d (d not read when c2 == 0), v += d
__global__ void __launch_bounds__(256) cheb_update_kernel(long n, double c1, const double* __restrict__ z, double c2,
                                                          double* __restrict__ d, double* __restrict__ v) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const double dn = c2 == 0.0 ? c1 * z[t] : fma(c1, z[t], c2 * d[t]);
    d[t] = dn;
    v[t] += dn;
  }
}

__global__ void __launch_bounds__(256) axpy_kernel(long n, double alpha, const double* __restrict__ x,
                                                   double* __restrict__ y) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
    y[t] += alpha * x[t];
}

// Windows of an extended block <-> one packed buffer (halo refresh of the overlapped-halo decomposition).  blockIdx.y is
// the window; a thread moves two neighbouring doubles of one window row when the window width and both addresses allow
// 16-byte accesses (the 64-wide strips of the decomposition always do), single doubles otherwise.
struct HaloRects {
  int n;
  int row0[8], col0[8], rows[8], cols[8];
  long offset[8];      // start of the window in the packed buffer, in doubles
};

template <bool PACK>
__global__ void __launch_bounds__(256) halo_pack_kernel(double* __restrict__ u, int nfield, int ey, int ex, HaloRects r,
                                                        double* __restrict__ buf) {
  const int w = blockIdx.y;
  const int rows = r.rows[w], cols = r.cols[w];
  const long plane = (long)ey * ex;
  double* b = buf + r.offset[w];
  double* base = u + (long)r.row0[w] * ex + r.col0[w];
  const bool pair = (cols & 1) == 0 && (ex & 1) == 0 && (r.col0[w] & 1) == 0 && (r.offset[w] & 1) == 0 &&
                    (((unsigned long long)u | (unsigned long long)buf) & 15) == 0;
  if (pair) {
    const int hc = cols >> 1;
    const long total = (long)nfield * rows * hc;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
      const int c = (int)(t % hc);
      const long fr = t / hc;
      const int j = (int)(fr % rows), f = (int)(fr / rows);
      double2* pu = reinterpret_cast<double2*>(base + (long)f * plane + (long)j * ex) + c;
      double2* pb = reinterpret_cast<double2*>(b + ((long)f * rows + j) * cols) + c;
      if (PACK) *pb = *pu; else *pu = *pb;
    }
  } else {
    const long total = (long)nfield * rows * cols;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
      const int c = (int)(t % cols);
      const long fr = t / cols;
      const int j = (int)(fr % rows), f = (int)(fr / rows);
      double* pu = base + (long)f * plane + (long)j * ex + c;
      double* pb = b + ((long)f * rows + j) * cols + c;
      if (PACK) *pb = *pu; else *pu = *pb;
    }
  }
}

static inline unsigned grid_for(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace qp

extern "C" {

int qp_add_constant(const uint8_t* flags, int64_t ncell, int32_t nfield, double* state, double amount, void* stream) {
  QP_REQUIRE(flags && state && ncell > 0 && nfield > 0, "bad arguments");
  hipLaunchKernelGGL(qp::add_constant_kernel, dim3(qp::grid_for(ncell * nfield)), dim3(256), 0, (hipStream_t)stream,
                     flags, (long)ncell, (int)nfield, state, amount);
  return qp::check_launch("qp_add_constant");
}

int qp_nan_pad(const uint8_t* flags, int64_t ncell, int32_t nfield, const double* in, double scale, double* out,
               void* stream) {
  QP_REQUIRE(flags && in && out && ncell > 0 && nfield > 0, "bad arguments");
  hipLaunchKernelGGL(qp::nan_pad_kernel, dim3(qp::grid_for(ncell)), dim3(256), 0, (hipStream_t)stream, flags, (long)ncell,
                     (int)nfield, in, scale, out);
  return qp::check_launch("qp_nan_pad");
}

int qp_add_scaled(int64_t n, double* state, const double* g, double scale, void* stream) {
  QP_REQUIRE(state && g && n > 0, "bad arguments");
  hipLaunchKernelGGL(qp::add_scaled_kernel, dim3(qp::grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long)n, state,
                     g, scale);
  return qp::check_launch("qp_add_scaled");
}

int64_t qp_pauli_workspace_bytes(void) { return (int64_t)qp::kRedBlocks * (int64_t)sizeof(qp::PauliPartial); }

int qp_pauli_stats(const double* state, const double* rho, const int32_t* cls, const uint8_t* flags, int32_t ne,
                   int32_t nclass, int64_t ncell, double density_floor, void* workspace, double* out_vals,
                   int64_t* out_idx, void* stream) {
  QP_REQUIRE(state && rho && flags && workspace && out_vals && out_idx, "NULL argument");
  QP_REQUIRE(ne > 0 && nclass > 0 && ncell > 0, "ne, nclass, ncell must be positive");
  QP_REQUIRE(nclass == 1 || cls, "cls is required when nclass > 1");
  const long by = ne < qp::kRedBlocks ? ne : qp::kRedBlocks;
  long bx = (ncell + 255) / 256;
  if (bx > qp::kRedBlocks / by) bx = qp::kRedBlocks / by;
  if (bx < 1) bx = 1;
  const long blocks = bx * by;
  auto* part = (qp::PauliPartial*)workspace;
  hipLaunchKernelGGL(qp::pauli_partial_kernel, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream, state,
                     rho, cls, flags, (int)ne, (long)ncell, density_floor, part);
  hipLaunchKernelGGL(qp::pauli_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, (int)blocks, out_vals,
                     (long*)out_idx);
  return qp::check_launch("qp_pauli_stats");
}

int qp_energy_integrate(const double* state, int32_t ne, int64_t ncell, double dE, double* out, void* stream) {
  QP_REQUIRE(state && out && ne > 0 && ncell > 0, "bad arguments");
  hipLaunchKernelGGL(qp::energy_integrate_kernel, dim3(qp::grid_for(ncell)), dim3(256), 0, (hipStream_t)stream, state,
                     (int)ne, (long)ncell, dE, out);
  return qp::check_launch("qp_energy_integrate");
}

int qp_weighted_sum(const double* state, const double* weights, int32_t n, int64_t ncell, double* out, void* stream) {
  QP_REQUIRE(state && weights && out && n > 0 && ncell > 0, "bad arguments");
  hipLaunchKernelGGL(qp::weighted_sum_kernel, dim3(qp::grid_for(ncell)), dim3(256), 0, (hipStream_t)stream, state,
                     weights, (int)n, (long)ncell, out);
  return qp::check_launch("qp_weighted_sum");
}

int qp_absmax(const double* a, int64_t n, void* workspace, double* out_val, void* stream) {
  QP_REQUIRE(a && workspace && out_val && n > 0, "bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > qp::kRedBlocks) blocks = qp::kRedBlocks;
  hipLaunchKernelGGL(qp::absmax_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, (long)n,
                     (double*)workspace);
  hipLaunchKernelGGL(qp::absmax_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace,
                     (int)blocks, out_val);
  return qp::check_launch("qp_absmax");
}

int qp_halo_pack(double* u, int32_t nfield, int32_t ey, int32_t ex, const int32_t* rects, int32_t nrect, int32_t op,
                 double* buf, void* stream) {
  QP_REQUIRE(u && buf && rects, "u, buf, rects must be non-NULL");
  QP_REQUIRE(nfield > 0 && ey > 0 && ex > 0, "nfield, ey, ex must be positive");
  QP_REQUIRE(nrect >= 1 && nrect <= 8, "1 to 8 windows per call");
  QP_REQUIRE(op == 0 || op == 1, "op must be 0 (pack) or 1 (unpack)");
  qp::HaloRects r;
  r.n = nrect;
  long off = 0, largest = 0;
  for (int w = 0; w < nrect; ++w) {
    r.row0[w] = rects[4 * w]; r.col0[w] = rects[4 * w + 1]; r.rows[w] = rects[4 * w + 2]; r.cols[w] = rects[4 * w + 3];
    QP_REQUIRE(r.rows[w] > 0 && r.cols[w] > 0 && r.row0[w] >= 0 && r.col0[w] >= 0 && r.row0[w] + r.rows[w] <= ey &&
                   r.col0[w] + r.cols[w] <= ex, "window outside the block");
    r.offset[w] = off;
    const long cells = (long)nfield * r.rows[w] * r.cols[w];
    off += cells;
    if (cells > largest) largest = cells;
  }
  long bx = (largest / 2 + 255) / 256;
  if (bx > 2048) bx = 2048;
  if (bx < 1) bx = 1;
  if (op == 0)
    hipLaunchKernelGGL(qp::halo_pack_kernel<true>, dim3((unsigned)bx, (unsigned)nrect), dim3(256), 0, (hipStream_t)stream, u,
                       (int)nfield, (int)ey, (int)ex, r, buf);
  else
    hipLaunchKernelGGL(qp::halo_pack_kernel<false>, dim3((unsigned)bx, (unsigned)nrect), dim3(256), 0, (hipStream_t)stream, u,
                       (int)nfield, (int)ey, (int)ex, r, buf);
  return qp::check_launch("qp_halo_pack");
}

int qp_axpy(int64_t n, double alpha, const double* x, double* y, void* stream) {
  QP_REQUIRE(x && y && n > 0, "bad arguments");
  hipLaunchKernelGGL(qp::axpy_kernel, dim3(qp::grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long)n, alpha, x, y);
  return qp::check_launch("qp_axpy");
}

int qp_cheb_update(int64_t n, double c1, const double* z, double c2, double* d, double* v, void* stream) {
  QP_REQUIRE(z && d && v && n > 0, "bad arguments");
  hipLaunchKernelGGL(qp::cheb_update_kernel, dim3(qp::grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long)n, c1, z, c2, d, v);
  return qp::check_launch("qp_cheb_update");
}

}  // extern "C"
