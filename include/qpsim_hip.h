/*
 * qpsim_hip.h -- C ABI of libqpsim_hip.so: the MI355X (gfx950) kernels behind the time loop of
 * qpsim.solver.run_2d_crank_nicolson (reference: /root/reference/qpsim/solver.py, cited per entry).
 *
 * The reference is pure Python and has no FFI of its own; the boundary it exposes for this path is one
 * Python function plus the step helpers.  This header is the binding a maintainer of the reference would
 * call from Python (ctypes) in place of the NumPy/SciPy bodies cited below.
 *
 * Conventions
 *   - Plain C types only: device pointers (HIP global memory), sizes, scalars, and a hipStream_t passed as
 *     void*.  No torch types.  All real data is IEEE fp64.
 *   - Every call returns 0 on success or a negative qp_status code; qp_last_error() gives the message of the
 *     last failure on the calling thread.  Nothing throws across the boundary.
 *   - Calls only enqueue work on `stream`; they never allocate, free or synchronise (graph-capture safe),
 *     except the *_create / *_destroy plan calls.
 *   - Fields live on the FULL ny x nx grid, row-major, one contiguous "plane" of ncell = ny*nx doubles per
 *     field; cells outside the geometry mask are kept at 0.  A batch is `nfield` consecutive planes
 *     (energy bins, ensemble members x bins).  The reference packs only interior cells
 *     (solver.py:53-58,1281-1285); the host layer converts at the boundary.
 *
 * Geometry / operator encoding (shared by every field of a problem)
 *   flags[ncell] (uint8): bit0 link to x-1, bit1 link to x+1, bit2 link to y-1, bit3 link to y+1,
 *                         bit4 cell is interior.  A link exists when both cells are interior.
 *   ex, ey[ncell]: boundary-face diagonal terms of the x- / y-faces of the cell in units of 1/dx^2:
 *                  absorbing 2, dirichlet 2, robin beta*dx, reflective/neumann 0     (solver.py:112-149)
 *   sx, sy[ncell]: boundary-face source terms in units of 1/dx^2:
 *                  dirichlet 2g, neumann q*dx, robin gamma*dx                          (solver.py:112-149)
 *   Diffusion coefficient of field b: scalar dcoef[b] (uniform gap) or plane dfield[b*ncell + p]
 *   (non-uniform gap: harmonic-mean face values 2 Dp Dq / max(Dp+Dq, 1e-30), solver.py:235-321).
 *   With r = dt/(2 dx^2) the directional operators are
 *      (r Lx u)_p = r [ w-(u_{p-1}-u_p) + w+(u_{p+1}-u_p) - ex_p D_p u_p ],   source r D_p sx_p   (same in y).
 */
#ifndef QPSIM_HIP_H
#define QPSIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum qp_status {
  QP_OK = 0,
  QP_ERR_INVALID_ARGUMENT = -1,
  QP_ERR_LAUNCH = -2,
  QP_ERR_UNSUPPORTED = -3,
  QP_ERR_ALLOC = -4
} qp_status;

#define QP_FLAG_LINK_XM 1u
#define QP_FLAG_LINK_XP 2u
#define QP_FLAG_LINK_YM 4u
#define QP_FLAG_LINK_YP 8u
#define QP_FLAG_ACTIVE 16u

/* Library version (major*10000 + minor*100 + patch) and last error text of this thread. */
int qp_version(void);
const char* qp_last_error(void);
/* Identity of the build as one JSON object: {"version": .., "arch": "gfx950", "qp_abl": N, "source_hash": "16 hex digits"}.
 * qp_abl != 0 marks a timing-only ablation build (tools/ablate.sh) whose results are wrong by construction: bench.py
 * prints this object with every number and refuses to measure such a build. */
const char* qp_build_info(void);

/* Geometry + per-field diffusivity description passed to the general diffusion kernels. */
typedef struct qp_grid_desc {
  uint32_t struct_size;    /* sizeof(qp_grid_desc) as the CALLER compiled it; a mismatch is QP_ERR_INVALID_ARGUMENT */
  int32_t ny, nx;          /* grid extent; ncell = ny*nx */
  int32_t nfield;          /* planes in the batch */
  const uint8_t* flags;    /* [ncell] */
  const double* ex;        /* [ncell] */
  const double* ey;        /* [ncell] */
  const double* sx;        /* [ncell] */
  const double* sy;        /* [ncell] */
  const double* dcoef;     /* [nfield] uniform diffusivity per field, or NULL */
  const double* dfield;    /* [nfield*ncell] diffusivity planes, or NULL (exactly one of dcoef/dfield) */
} qp_grid_desc;

/*
 * out = c0*u + cx*(r Lx u) + cy*(r Ly u) + cs*(r D (sx+sy)) + cr*rin          (rin may be NULL when cr == 0)
 *
 * One pass of the 5-point operator of solver.py:152-212 / :235-321 split by direction.  With
 * (c0,cx,cy,cs,cr) = (1,0,1,1,0) it is the explicit half of the first ADI sweep, (1,1,1,2,0) the
 * Crank-Nicolson right-hand side B u + dt D source of solver.py:1440,1451,1554, and (-1,1,1,0,1) the residual
 * rin - (I - r L) u used by the exact-CN iteration.  r = dt/(2 dx^2).
 */
int qp_stencil_combine(const qp_grid_desc* g, double r, const double* u, const double* rin, double* out,
                       double c0, double cx, double cy, double cs, double cr, void* stream);
/* The same with norm_out[0] = max |out| formed in the same pass (NaN counts as +inf; workspace:
 * qp_pauli_workspace_bytes() bytes): the residual and its norm of the exact-CN iteration in one kernel.
 * ex, ey, sx, sy are read only at cells with a missing link - boundary terms belong to boundary faces. */
int qp_stencil_combine_norm(const qp_grid_desc* g, double r, const double* u, const double* rin, double* out,
                            double c0, double cx, double cy, double cs, double cr, void* workspace, double* norm_out,
                            void* stream);

/*
 * Implicit sweep: solve (I - r L_dir) x = rhs along every grid line of every field (dir 0 = x, 1 = y).
 * Replaces `lu.solve(rhs)` of solver.py:1441,1452,1555 one direction at a time.  `scratch` holds
 * 2*nfield*ncell doubles.  rhs and x may alias.  General (masked / variable-D) path: Thomas per line.
 */
int qp_implicit_sweep(const qp_grid_desc* g, double r, int dir, const double* rhs, double* x, double* scratch,
                      void* stream);

/*
 * Collision tables for qp_collision_step (all device pointers).  Built on the host exactly as
 * solver.py:463-490 (kernels), :324-342 (rho), :668-683 (phonon bin maps); `cls` selects the per-pixel
 * gap class (solver.py:1203-1232 copies per-pixel tables; here only one table set per distinct gap is kept).
 */
typedef struct qp_collision_tables {
  uint32_t struct_size;    /* sizeof(qp_collision_tables) as the CALLER compiled it: a binding written against another
                            * revision of this header (fewer / more members) is rejected with QP_ERR_INVALID_ARGUMENT
                            * instead of being read past its end */
  int32_t ne;              /* quasiparticle energy bins */
  int32_t nw;              /* phonon bins */
  int32_t nclass;          /* distinct gap classes (1 = uniform) */
  const double* kr0;       /* [nclass][ne][ne] or NULL when recombination is off */
  const double* ks0;       /* [nclass][ne][ne] or NULL when scattering is off */
  const double* rho;       /* [nclass][ne] */
  const int32_t* idx_diff; /* [ne][ne] phonon bin of |Ei-Ej| */
  const int32_t* idx_sum;  /* [ne][ne] phonon bin of Ei+Ej */
  const int8_t* sign;      /* [ne][ne] sign(Ei-Ej) */
  const int32_t* cls;      /* [ncell] gap class per cell, or NULL when nclass == 1 */
  /* Optional structure hint (both or neither).  On the reference's uniform energy grid idx_diff[i][j] = diag_bin[|i-j|]
   * and idx_sum[i][j] = anti_bin[i+j], sign[i][j] = sign(i-j); pass the two arrays when the host has verified that.
   * If a phonon bin is shared between a diagonal k and an anti-diagonal m (merged bins, e.g. whenever 2 E_min / dE is an
   * integer) also set QP_COLL_SHARED_BINS and tag BOTH entries, diag_bin[k] and anti_bin[m], with (slot + 1) << 16 on top of
   * the bin index (slot = 0, 1, ... numbering the merged bins): the register-resident kernels then park the diagonal's sums
   * in ph_scratch[2 slot], ph_scratch[2 slot + 1] (planes of ncell doubles) until the anti-diagonal finalises the bin, so
   * ph_scratch must hold 2 * (number of merged bins) * ncell doubles when phonons are updated with both processes on. */
  const int32_t* diag_bin; /* [ne] or NULL */
  const int32_t* anti_bin; /* [2*ne-1] or NULL */
  /* Kernel selection.  Default (0): register-resident kernel when (diag_bin, nclass == 1,
   * qp_collision_register_kernel_available(ne)); otherwise, for
   * ne <= 64, one wave per pixel with lanes <-> energy bins (deterministic when diag_bin vouches for the bin-map structure,
   * LDS atomics otherwise); otherwise the generic one-thread-per-cell kernel.  The FORCE bits exist for tests. */
  uint32_t flags;
  /* Optional, for nclass > 1 (non-uniform gap): the reference's kernels are separable in the gap (solver.py:463-490),
   *   K^r_0 = kr_amp[i][j] (1 + gap^2 pair_inv[i][j]),   K^s_0 = ks_amp[i][j] max(1 - gap^2 pair_inv[i][j], 0),
   * with kr_amp = (1/tau_r) ((Ei+Ej)/kTc)^2 / kTc, ks_amp = (1/tau_s) (Ei-Ej)^2 / kTc^3 (zero diagonal),
   * pair_inv = 1 / max(Ei Ej, 1e-30).  When gap_sq[nclass] and pair_inv are given (kr_amp / ks_amp per enabled process),
   * the register-resident kernel serves gap classes too (qp_collision_register_kernel_classes(ne)): it forms K per pixel from these three shared tables
   * instead of reading per-class tables; at ne = 50 with at most 16 classes that is the one-pass kernel as well (the tables
   * staged in LDS, K formed per lane).  All NULL: gap classes run the one-wave-per-pixel kernel. */
  const double* gap_sq;
  const double* kr_amp;
  const double* ks_amp;
  const double* pair_inv;
  /* Optional, for the ONE-PASS register kernel (qp_collision_onepass_available(ne); one gap class): one launch that reads
   * the old quasiparticle planes once and sweeps the phonon planes once per target block, where ne >= 32 otherwise runs
   * three launches.  Its phonon phase
   * walks the pairs (anti)diagonal by (anti)diagonal and wants the kernel values of a diagonal contiguous:
   *   ks0_diag  [ne][ne]      ks0_diag [k*ne + i] =     ks0[i][i-k]   for k <= i < ne,      0 elsewhere (k = 0..ne-1)
   *   kr0_anti2 [2ne-1][ne]   kr0_anti2[m*ne + i] = 2 * kr0[i][m-i]   for 0 <= m-i < ne,    0 elsewhere (m = 0..2ne-2)
   * Each is needed only for an enabled process; both NULL is always valid (three-launch path). */
  const double* ks0_diag;
  const double* kr0_anti2;
} qp_collision_tables;
#define QP_COLL_FORCE_GENERIC 1u
#define QP_COLL_FORCE_WAVE 2u
#define QP_COLL_SHARED_BINS 4u

/*
 * One local coupled quasiparticle-phonon collision update of every interior cell
 * (solver.py:703-791 per pixel, :794-875 the pixel loops).  state_in [ne][ncell] and phonon [nw][ncell]
 * are read as the OLD values; the new quasiparticle density goes to state_out (must not alias state_in),
 * phonons are updated in place when update_phonons != 0.
 * ph_scratch: the generic one-thread-per-cell kernel (ne > 64 or QP_COLL_FORCE_GENERIC) needs 2*nw*ncell doubles of
 * accumulators when phonons are updated; the register-resident kernels need 2*(merged bins)*ncell doubles only when
 * QP_COLL_SHARED_BINS is set and both processes update phonons (see diag_bin above); every other case takes NULL.
 * Cells whose flags lack QP_FLAG_ACTIVE are copied through unchanged.
 */
int qp_collision_step(const qp_collision_tables* t, const uint8_t* flags, int64_t ncell, const double* state_in,
                      double* state_out, double* phonon, double* ph_scratch, double dE, double dt,
                      int enable_recombination, int enable_scattering, int update_phonons, void* stream);

/*
 * qp_collision_step followed by qp_pauli_stats on state_out (the guard of solver.py:1477 after the last collision of a step),
 * as ONE call: the single-pass register kernels (ne < 32) reduce the occupation statistics of the new densities while they
 * are still in registers - one partial per wave into guard_workspace, finished by two small launches - which saves the
 * separate pass over the ne planes (7 % of a 4096^2, ne = 12 coupled step).  Other kernels run the separate pass.
 * out_vals / out_idx as qp_pauli_stats (bit-identical results); guard_workspace: qp_collision_guard_workspace_bytes(ncell).
 */
int64_t qp_collision_guard_workspace_bytes(int64_t ncell);
int qp_collision_step_guarded(const qp_collision_tables* t, const uint8_t* flags, int64_t ncell, const double* state_in,
                              double* state_out, double* phonon, double* ph_scratch, double dE, double dt,
                              int enable_recombination, int enable_scattering, int update_phonons, double density_floor,
                              void* guard_workspace, double* out_vals, int64_t* out_idx, void* stream);

/*
 * TWO consecutive collision half-steps in one pass over the state: under Strang splitting the closing half-step of step k
 * and the opening half-step of step k + 1 (solver.py:1469-1475) act on the same state with nothing between them but the
 * Pauli guard of step k (solver.py:1477) and the explicit generation term of step k + 1 (solver.py:1459-1464) - unless step
 * k is a store point.  Equivalent, bit for bit, to
 *     qp_collision_step_guarded(state_in -> tmp, dt_first);   tmp += gen_amount on interior cells;
 *     qp_collision_step(tmp -> state_out, dt_second)
 * but n and the phonon planes make one round trip through HBM instead of two (the intermediate state never leaves the
 * registers).  out_vals / out_idx: guard statistics of the INTERMEDIATE state (before the generation term), as
 * qp_pauli_stats.  gen_amount: dt_{k+1} * rate for constant / active pulse generation, 0 otherwise.
 * Returns QP_ERR_UNSUPPORTED (nothing launched, nothing written) when no fused kernel fits - qp_collision_pair_available(ne)
 * is 0, more than one gap class, merged phonon bins (QP_COLL_SHARED_BINS), no diag_bin / anti_bin, a FORCE flag, no enabled
 * process: the caller then issues the two calls above.  guard_workspace: qp_collision_guard_workspace_bytes(ncell).
 */
int qp_collision_pair_available(int32_t ne);
int qp_collision_double_step_guarded(const qp_collision_tables* t, const uint8_t* flags, int64_t ncell, const double* state_in,
                                     double* state_out, double* phonon, double dE, double dt_first, double dt_second,
                                     double gen_amount, int enable_recombination, int enable_scattering, int update_phonons,
                                     double density_floor, void* guard_workspace, double* out_vals, int64_t* out_idx,
                                     void* stream);

/* 1 when qp_collision_step has a register-resident kernel for `ne` energy bins (structured, unshared bin maps and one gap
 * class are the other conditions); other sizes <= 64 run the one-wave-per-pixel kernel, larger ones the generic kernel. */
int qp_collision_register_kernel_available(int32_t ne);
/* 1 when the one-pass kernel (ks0_diag / kr0_anti2 above) is instantiated for `ne` energy bins: ne = 30, 32, 40 and 50 (the
 * reference's default num_energy_bins, solver.py:1012). */
int qp_collision_onepass_available(int32_t ne);
/* 1 when the register-resident kernel also has its gap-class variant for `ne` (single-pass sizes: ne <= 16, 18, 20, 24, 30). */
int qp_collision_register_kernel_classes(int32_t ne);

/*
 * Explicit fixed-bath collision helpers of the reference's step API (not on its time loop; API parity):
 * rhs = [g_therm - 2 n dE (K_r n)] (if kr) + [dE rho (1-f) (K_s^T n) - n dE ((K_s rho) (1-f))] (if ks), per cell;
 * out = rhs (rhs_only != 0: solver.py:608-637 _collision_rhs) or max(n + dt rhs, 0) (solver.py:551-605
 * apply_scattering_step / apply_recombination_step).  state_in, out: [ne][ncell], must not alias.
 */
int qp_euler_collision(int32_t ne, int64_t ncell, const double* state_in, double* out, const double* kr,
                       const double* g_therm, const double* ks, const double* rho, double dE, double dt, int rhs_only,
                       void* stream);

/*
 * state[f][p] += amount for every interior cell of `nfield` planes (constant / pulse external generation,
 * solver.py:910-916,1464), or state += scale * g[f][p] (custom generation evaluated on the host).
 */
int qp_add_constant(const uint8_t* flags, int64_t ncell, int32_t nfield, double* state, double amount, void* stream);
/* out[f][p] = scale * in[f][p] inside the mask, NaN outside: the NaN-padded [ny, nx] frames of reconstruct_field
 * (solver.py:215-218, store block :1479-1494) formed on the device, so a stored state crosses PCIe once, ready to hand out. */
int qp_nan_pad(const uint8_t* flags, int64_t ncell, int32_t nfield, const double* in, double scale, double* out,
               void* stream);
int qp_add_scaled(int64_t n, double* state, const double* g, double scale, void* stream);

/*
 * Pauli-guard statistics over state[ne][ncell] (solver.py:967-996): occupation f = n / rho where rho > 1e-30.
 * out_vals[0] = max f, out_idx[0] = its linear index ie*ncell + p (first in C order on ties; a NaN occupation counts as
 * the maximum and the first NaN wins, as np.argmax does),
 * out_idx[1] = first linear index with rho <= 1e-30 and n > density_floor, or -1.
 * workspace: qp_pauli_workspace_bytes() bytes.  Results land in device memory (copy them back yourself).
 */
int64_t qp_pauli_workspace_bytes(void);
int qp_pauli_stats(const double* state, const double* rho, const int32_t* cls, const uint8_t* flags, int32_t ne,
                   int32_t nclass, int64_t ncell, double density_floor, void* workspace, double* out_vals,
                   int64_t* out_idx, void* stream);

/* out[p] = dE * sum_i state[i][p]  (energy integral of solver.py:1367,1480; sequential in i). */
int qp_energy_integrate(const double* state, int32_t ne, int64_t ncell, double dE, double* out, void* stream);

/* out[p] = sum_i state[i][p] * weights[i]  (phonon integrated occupation, solver.py:1358). */
int qp_weighted_sum(const double* state, const double* weights, int32_t n, int64_t ncell, double* out, void* stream);

/* out_val[0] = max_p |a[p]| over n doubles (convergence check of the exact-CN iteration). workspace as above. */
int qp_absmax(const double* a, int64_t n, void* workspace, double* out_val, void* stream);

/* y[i] += alpha * x[i] */
int qp_axpy(int64_t n, double alpha, const double* x, double* y, void* stream);
/* d[i] = c1 * z[i] + c2 * d[i] (d is not read when c2 == 0: it may be uninitialised); v[i] += d[i]: direction and iterate
 * update of the Chebyshev-accelerated exact-CN iteration in one pass */
int qp_cheb_update(int64_t n, double c1, const double* z, double c2, double* d, double* v, void* stream);

/*
 * Fast CN-ADI path: full ny x nx rectangle, one diffusivity per field, one boundary condition per side.
 * Replaces, for that geometry class, the whole `rhs = B u + dt D source; u = lu.solve(rhs)` loop of
 * solver.py:1443-1452 / :1545-1555 by the Peaceman-Rachford factorisation of the same Crank-Nicolson step
 * (identical when ny == 1 or nx == 1; O(dt^3) splitting term per step otherwise).
 *
 *   r          dt / (2 dx^2)
 *   dcoef_host [nfield] diffusivities (HOST pointer; tables are built on the host at plan creation)
 *   bc_diag    [4] boundary diagonal terms of the left, right, up, down sides in 1/dx^2 units (see ex/ey above)
 *   bc_src     [4] boundary source terms of the same sides (see sx/sy above)
 *   force_banded  0: when the far couplings between 64-cell chunks underflow fp64 significance (< 1e-22, true for
 *              r*D <~ 1.5) the interface unknowns are solved as independent 2x2 systems inside the sweep kernels;
 *              otherwise, or when force_banded != 0, a banded reduced solve runs between the sweeps.
 *              qp_adi_rect_plan_decoupled(plan, dir) reports which one a direction uses (1 / 0).
 * The plan owns its device tables and work planes (hipMalloc at creation, the only allocating call).
 * qp_adi_rect_steps advances u [nfield][ny*nx] in place by `nsteps` consecutive diffusion steps; intermediate
 * fields are not materialised (steady-state traffic: one read + one write of the field per sweep).
 */
typedef struct qp_adi_rect_plan qp_adi_rect_plan;
int qp_adi_rect_plan_create(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                            const double* bc_diag, const double* bc_src, int32_t force_banded,
                            qp_adi_rect_plan** out);
int qp_adi_rect_plan_destroy(qp_adi_rect_plan* plan);
int qp_adi_rect_plan_decoupled(const qp_adi_rect_plan* plan, int32_t dir);
/* 1 when the plan runs its passes on fine tiles (lines cut into 32-cell chunks, one wave per 32 x 64 / 64 x 32 tile:
 * undecomposed grids whose extents are multiples of 64, r*D <~ 0.32 for every field, size rule or QPSIM_FINE_TILES=1),
 * 0 for the 64 x 64 tiles.  Same results to rounding (the dropped far couplings are < 1e-22 either way). */
int qp_adi_rect_plan_fine(const qp_adi_rect_plan* plan);
/* Peaceman-Rachford iteration for the UNSPLIT Crank-Nicolson system A u = b, A = I - r D (Lx + Ly) (the matrix the
 * reference factorises with SuperLU, solver.py:231,1155-1161), on full rectangles whose Lx and Ly commute:
 *   (H + p) u* = b - (V - p) u,   (V + p) u' = b - (H - p) u*,     H = I/2 - r D Lx,  V = I/2 - r D Ly.
 * qp_adi_rect_plan_create_pr builds the plan of ONE parameter p > 0 (tables of the sweeps with r D / (1/2 + p), explicit
 * operators with the shifted diagonal; boundary sources belong to b); it runs the fine tiles where the grid qualifies
 * (extents multiples of 64, r D / (1/2 + p) <~ 0.32) and the 64 x 64 tiles - any extents, banded reduced systems for
 * stiff steps - elsewhere.  `share` (may be NULL): another plan of the same shape whose work
 * plane is borrowed - a cycle of J parameters then holds one work plane, not J; destroy the lender last.
 * qp_adi_rect_pr_iteration overwrites u with the next iterate (three passes: 8 B read + 8 B read of b + 8 B written,
 * twice, and 8 + 8 B for the last solve, per cell).  With parameters spread over the spectrum of H and V, [1/2, 1/2 +
 * r D (4 + boundary term)], J = 5-7 iterations reduce the error by 1e-13 (the host chooses them: qpsim_amd/engine.py). */
int qp_adi_rect_plan_create_pr(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                               const double* bc_diag, double p, qp_adi_rect_plan* share, qp_adi_rect_plan** out);
int qp_adi_rect_pr_iteration(qp_adi_rect_plan* plan, double* u, const double* b, void* stream);
/* One whole cycle, plans[0 .. nplans) in order, u overwritten with the last iterate.  When every plan runs the fine tiles
 * the iterations are carried (the y-pass of iteration j forms the right-hand side of iteration j + 1: two passes and
 * 6 plane transfers per iteration instead of three passes and 8); otherwise one qp_adi_rect_pr_iteration per plan. */
int qp_adi_rect_pr_cycle(qp_adi_rect_plan* const* plans, int32_t nplans, double* u, const double* b, void* stream);
int qp_adi_rect_steps(qp_adi_rect_plan* plan, double* u, int32_t nsteps, void* stream);
/* qp_stencil_combine for the plan's operator without per-cell geometry arrays (positions decide which side term applies):
 * out = c0 u + cx (a Lx u) + cy (a Ly u) + cs a S + cr rin on [nfield][ny*nx]; with norm_out non-NULL also
 * norm_out[0] = max |out| (workspace: qp_pauli_workspace_bytes() bytes), which saves the exact-CN iteration its
 * separate norm pass.  (1,1,1,2,0) = CN right-hand side, (-1,1,1,0,1) = residual rin - (I - a L) u.  `out` may be NULL when
 * norm_out is given: only the norm is formed (the residual CHECK after a Peaceman-Rachford cycle: one plane less to move). */
int qp_adi_rect_combine(qp_adi_rect_plan* plan, const double* u, const double* rin, double* out, double c0, double cx,
                        double cy, double cs, double cr, void* workspace, double* norm_out, void* stream);
/* x[nfield][ny*nx] <- (I - a Ly)^-1 (I - a Lx)^-1 x in place: the ADI factorisation applied as the preconditioner of
 * the exact Crank-Nicolson iteration (replaces two qp_implicit_sweep calls on full rectangles). */
int qp_adi_rect_solve(qp_adi_rect_plan* plan, double* x, void* stream);

/*
 * Domain decomposition (one rank per GPU owns a ny x nx block at offset (j0, i0) of a gny x gnx grid; offsets and
 * interior block extents are multiples of 64).  Only the decoupled-interface regime is supported across ranks
 * (QP_ERR_UNSUPPORTED otherwise): the coupling between blocks is then the same 2x2 interface system as between
 * 64-cell chunks inside a block, and each sweep needs from each neighbour one row of reduced right-hand sides
 * (nfield x nlines doubles), the entry pass one row of the field.  The host sequences the phases and moves the rows
 * (RCCL send/recv); a time step that starts from a materialised field is
 *     [set_field_halo up/down] ENTRY [iface x] REDUCED_X SWEEP_X [iface y] REDUCED_Y SWEEP_Y_EXIT
 * and consecutive steps replace SWEEP_Y_EXIT by SWEEP_Y_CARRY [iface x] REDUCED_X SWEEP_X ...
 * qp_adi_rect_phase works on undecomposed plans too (it is what qp_adi_rect_steps calls).
 */
enum {
  QP_ADI_ENTRY = 0,         /* u -> rhs of the x-solve (needs field halo rows), reduced rhs for x */
  QP_ADI_REDUCED_X = 1,     /* banded reduced solve along x (no-op in the decoupled regime) */
  QP_ADI_SWEEP_X = 2,       /* x-solve, rhs of the y-solve, reduced rhs for y */
  QP_ADI_REDUCED_Y = 3,
  QP_ADI_SWEEP_Y_CARRY = 4, /* y-solve, rhs of the next step's x-solve, reduced rhs for x */
  QP_ADI_SWEEP_Y_EXIT = 5   /* y-solve, u' stored to u */
};
int qp_adi_rect_plan_create_block(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                                  const double* bc_diag, const double* bc_src, int32_t force_banded, int32_t gny,
                                  int32_t gnx, int32_t j0, int32_t i0, qp_adi_rect_plan** out);
int qp_adi_rect_phase(qp_adi_rect_plan* plan, int32_t phase, double* u, void* stream);
/* dir 0: neighbours left (side 0) / right (side 1); dir 1: up / down.  op 0 packs the row the neighbour needs into
 * buf[nfield][nlines], op 1 stores the neighbour's row into the halo slot on that side. */
int qp_adi_rect_iface_halo(qp_adi_rect_plan* plan, int32_t dir, int32_t side, int32_t op, double* buf, void* stream);
/* rows[nfield][nx]: the field row just above (side 0) / below (side 1) the block, consumed by QP_ADI_ENTRY. */
int qp_adi_rect_set_field_halo(qp_adi_rect_plan* plan, int32_t side, const double* rows, void* stream);

/*
 * Halo refresh of the overlapped-halo decomposition (one rank per GPU holds its block + halos as u[nfield][ey][ex]):
 * copies `nrect` rectangular windows between u and ONE packed buffer in a single launch, so that a refresh is
 * pack -> one batch of RCCL send/recv on persistent buffers -> unpack, with no host synchronisation in between.
 *   rects  HOST array [nrect][4] = (row0, col0, rows, cols) of each window inside the ey x ex block, nrect <= 8
 *   buf    windows back to back in `rects` order, each stored [nfield][rows][cols]
 *   op     0: pack (u -> buf), 1: unpack (buf -> u)
 * The reference has no distributed path; this serves north_star's "RCCL halo exchange over xGMI".
 */
int qp_halo_pack(double* u, int32_t nfield, int32_t ey, int32_t ex, const int32_t* rects, int32_t nrect, int32_t op,
                 double* buf, void* stream);

/*
 * Tiled CN-ADI path for MASKED grids: any mask, any per-face boundary condition, one diffusivity per field
 * (qp_adi_tile_plan_create) or a diffusivity field per energy bin (qp_adi_tile_plan_create_var).
 * Replaces the same reference loop as the rectangle path (solver.py:1443-1452 / :1545-1555) on the geometries the
 * reference actually ships (strips with holes, donuts, GDS shapes: build_laplacian_with_boundaries solver.py:152-212).
 *   flags, ex, ey, sx, sy   HOST arrays [ny*nx] with the meaning of qp_grid_desc (the plan compresses them into 16-bit
 *              per-cell codes + a table of distinct boundary terms, classifies the 64 x 64 tiles as empty / clean /
 *              general and computes the chunk-interface coefficients on the device)
 * Returns QP_ERR_UNSUPPORTED when r*D is too large for 64-cell chunks to decouple (same 1e-22 criterion as the
 * rectangle path; r*D <~ 1.3) or when there are more than 1024 distinct boundary-term combinations: the caller then
 * uses qp_stencil_combine / qp_implicit_sweep.
 * qp_adi_tile_steps: `nsteps` Peaceman-Rachford steps in place on u[nfield][ny*nx]; cells outside the mask must be 0
 * and stay 0.  qp_adi_tile_solve: x <- (I - a Ly)^-1 (I - a Lx)^-1 x (preconditioner of the exact-CN iteration).
 * qp_adi_tile_plan_info: counts[3] = empty, clean, general tiles; far = largest coupling across a chunk.
 */
typedef struct qp_adi_tile_plan qp_adi_tile_plan;
int qp_adi_tile_plan_create(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                            const uint8_t* flags, const double* ex, const double* ey, const double* sx, const double* sy,
                            qp_adi_tile_plan** out);
/* Same for spatially varying diffusivity (build_variable_diffusion_laplacian, solver.py:235-321: harmonic-mean face values,
 * boundary terms scaled by the cell's D): dfield is a DEVICE array [nfield][ny*nx] read once at plan creation (face weights
 * and r*D are stored per field in the layouts the sweeps read: 4 planes per field).  Every non-empty tile is "general". */
int qp_adi_tile_plan_create_var(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dfield,
                                const uint8_t* flags, const double* ex, const double* ey, const double* sx,
                                const double* sy, qp_adi_tile_plan** out);
int qp_adi_tile_plan_destroy(qp_adi_tile_plan* plan);
int qp_adi_tile_plan_info(const qp_adi_tile_plan* plan, int32_t* counts, double* far);
int qp_adi_tile_steps(qp_adi_tile_plan* plan, double* u, int32_t nsteps, void* stream);
int qp_adi_tile_solve(qp_adi_tile_plan* plan, double* x, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QPSIM_HIP_H */
