/*
 * slampv.h -- C ABI of libslampv.so: the MI355X (gfx950) implementation of
 * SLAM pvfitting's mock position-velocity likelihood, batched over emcee
 * walkers (BASELINE.json config 5).
 *
 * The reference (jinshisai/SLAM) is pure Python and has no FFI of its own; the
 * de-facto boundary this library sits behind is
 *
 *   - the per-walker closure `lnprob(p)`          pvfitting/_pvfitting.py:181-195
 *   - `makemodel` / `MockPVD.generate_mockpvd`    pvfitting/_pvfitting.py:142-151,
 *                                                 pvfitting/mockpvd.py:97-156
 *
 * Plain C: opaque handle, caller-owned row-major arrays, `int` status (0 = ok,
 * <0 = error, text via slampv_last_error).  One context drives ONE GPU (the
 * current HIP device at create time); walkers are independent, so multi-GPU use
 * is one process per GPU, each with its own context and shard of the walkers.
 *
 * Parameter vectors are 6 float64 in the reference's order
 * (_pvfitting.py:154), natural units (NOT log10):
 *   Mstar, Rc, alphainfall, taumax, frho, sig_mdl
 */
#ifndef SLAMPV_H
#define SLAMPV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMPV_NPARAM 6
#define SLAMPV_NTAB 5 /* rho_env, rho_disk, vr_env, vt_env, vp_env+vp_disk */

enum {
  SLAMPV_OK = 0,
  SLAMPV_ERR_ARG = -1,
  SLAMPV_ERR_HIP = -2,
  SLAMPV_ERR_NOMEM = -3,
  SLAMPV_ERR_STATE = -4
};

/* One level of the nested 3-D grid (pvfitting/grid.py:129-183) on its FULL
 * (nx, ny, nz) box, stored z-slowest / x-fastest ([nz][ny][nx]) so that the
 * line-of-sight columns of neighbouring pixels are adjacent in memory.  The
 * per-cell arrays are the module-level caches the reference fills on its first
 * evaluation (pvfitting/precalculation.py:169-179), one set per PV axis
 * ([0] = major, [1] = minor).  A cell that can never contribute -- covered by
 * the next level (mockpvd.py:298), r <= rin or r > rout (mockpvd.py:228-229) --
 * carries idx_t < 0. */
typedef struct slampv_level {
  int32_t nx, ny, nz;
  int32_t bx0, bx1;  /* pixels [bx0,bx1] x [by0,by1] of this level are refined  */
  int32_t by0, by1;  /* by the next level (grid.py:136-137); -1 on the last one */
  int32_t nsub;      /* refinement factor of the next level (grid.py:32)        */
  int32_t pad_;
  double dz;         /* zaxes[l][1]-zaxes[l][0]            (mockpvd.py:283,291) */
  const int32_t* idx_t[2]; /* theta row of the tables   (precalculation.py:176-177) */
  const double* j_org[2];  /* ln r column, fractional   (precalculation.py:179)     */
  const double* elos_r[2]; /* line-of-sight projections (precalculation.py:42-48)   */
  const double* elos_t[2];
  const double* elos_p[2];
} slampv_level;

/* Everything `MockPVD.__init__` + the first evaluation + `fit_mockpvd`'s
 * preamble hold constant (mockpvd.py:64-95, 168-192; _pvfitting.py:110-141).
 * All pointers are HOST pointers, copied to the GPU by slampv_create. */
typedef struct slampv_desc {
  int32_t nlevels;   /* grid.nlevels                              grid.py:33 */
  int32_t nv;        /* len(v)                                               */
  int32_t nx_out;    /* len(x) of the observed PV (before sub-gridding)      */
  int32_t nsubgrid;  /* level-0 nx = nx_out * nsubgrid       mockpvd.py:70-76 */
  int32_t ntheta, nr; /* table shape                precalculation.py:146-151 */
  int32_t signmajor, signminor; /* x flips of the two cuts  mockpvd.py:88-90,145 */
  double dlnr;       /* exp(lnr[1]-lnr[0])-1            precalculation.py:149 */
  double g_si, m_sun, au_m; /* constants behind vunit         mockpvd.py:11-13 */
  double sigma;      /* noise rms                          _pvfitting.py:123 */
  double rarea;      /* beam area / dx^2               _pvfitting.py:114-115 */
  const double* table;    /* [ntheta][nr][SLAMPV_NTAB]  precalculation.py:156-159 */
  const double* vedge;    /* [nv+1] channel edges            mockpvd.py:260-261 */
  const double* gauss_v;  /* [nv] or NULL (linewidth=None)   mockpvd.py:307-312 */
  const double* gauss_xy; /* [ny0][nx0] or NULL (beam=None)  mockpvd.py:319-325 */
  const double* obs_major; /* [nv][nx_out], NaN = masked; NULL = model only  */
  const double* obs_minor; /*                              _pvfitting.py:119-122 */
  const slampv_level* levels; /* [nlevels]                                  */
  int64_t max_scratch_bytes; /* cap on per-batch device scratch; 0 = default */
} slampv_desc;

typedef struct slampv_ctx slampv_ctx;

/* Replaces the state MockPVD.__init__ and the first generate_mockpvd call
 * leave behind (mockpvd.py:64-95, 203-219). */
int slampv_create(const slampv_desc* desc, slampv_ctx** out);

/* Replaces `lnprob`'s body (_pvfitting.py:185-195) for a whole batch: for each
 * of the `nw` parameter rows builds both mock PV diagrams
 * (mockpvd.py:97-147), scales them by the common flux factor
 * (_pvfitting.py:149-151) and returns
 *   -0.5 * (chi2maj + chi2min) / sqrt(Rarea),
 * chi2 including the log(sigma'^2) term.  The caller applies the box prior
 * (utils.py:32-36).  Host pointers; synchronous. */
int slampv_lnprob_batch(slampv_ctx* ctx, const double* params, int nw,
                        double* lnp_out);

/* Same with `params_dev` [nw][6] and `lnp_dev` [nw] resident on the context's
 * GPU; enqueued on `stream` (hipStream_t as void*, NULL = default) and NOT
 * waited for.  Every call on a context works in that context's one scratch
 * set: a call enqueued on another stream than the previous call first waits
 * (event) for the previous call's last kernel, so calls on one context are
 * ordered and never run concurrently; calls from several host threads on one
 * context must be serialised by the caller.  For concurrency use one context
 * per stream. */
int slampv_lnprob_batch_dev(slampv_ctx* ctx, const double* params_dev, int nw,
                            double* lnp_dev, void* stream);

/* Hot path with its scalars exposed: per walker
 *   {fflux (_pvfitting.py:149-150), chi2 without the log term / sqrt(Rarea)
 *    (_pvfitting.py:230-232), lnprob (:195), max tau major, max tau minor,
 *    number of unmasked pixels}.
 * `stats_out` [nw][6], host pointers; synchronous. */
int slampv_stats(slampv_ctx* ctx, const double* params, int nw,
                 double* stats_out);

/* Replaces `MockPVD.generate_mockpvd(..., axis='both')` (mockpvd.py:122-147)
 * for one parameter vector (sig_mdl ignored): UNSCALED major and minor PV
 * diagrams, each [nv][nx_out], x already flipped by the axis sign. */
int slampv_mockpvd(slampv_ctx* ctx, const double* params, double* major_out,
                   double* minor_out);

/* Parity taps.  _own: level `level`'s own column sums, summed over z and not
 * yet multiplied by dz or divided by the global density maximum (rho2tau,
 * precalculation.py:136-144), [nv][ny_l][nx_l]; NaN marks the columns that are
 * never computed because the nested add (mockpvd.py:272-277, diagonal
 * sub-cells only) cannot carry them to level 0.  _tau: the nested optical-depth
 * cube on level 0 after the optional spectral smoothing (mockpvd.py:279-314),
 * [nv][ny0][nx0]. */
int slampv_own_tau(slampv_ctx* ctx, const double* params, int axis, int level,
                   double* out);
int slampv_tau(slampv_ctx* ctx, const double* params, int axis, double* out);

/* Introspection: cells that can contribute and are computed (both axes), walkers per launch
 * group, column tasks per walker. */
int slampv_info(slampv_ctx* ctx, int64_t* live_cells, int32_t* walkers_per_group,
                int32_t* tasks_per_walker);

/* Measurement aid (no counterpart in the reference): per-kernel HIP-event
 * timing on the launch stream between _begin and _end. */
int slampv_kernel_count(void);
const char* slampv_kernel_name(int i);
int slampv_profile_begin(slampv_ctx* ctx);
int slampv_profile_end(slampv_ctx* ctx, double* total_ms, int64_t* launches);

const char* slampv_last_error(slampv_ctx* ctx);
void slampv_destroy(slampv_ctx* ctx);
int slampv_device_count(void);
const char* slampv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLAMPV_H */
