// slamhip.hip -- MI355X (gfx950) kernels + C ABI for SLAM channelfit's batched
// likelihood: Keplerian-disk channel-map model + chi-square against the
// observed cube, one launch group for many emcee walkers.
//
// Reference behaviour restated (jinshisai/SLAM, channelfit/_channelfit.py):
//   update_pa :373-374   update_incl :376-380   update_xdisk :382-410
//   update_prof :412-427 getvlos :429-449       get_Iunif :454-473
//   convolve :530        rgi2d :475-483         get_scale :485-490
//   chi2 :601-602
// The design (sample list, kernels, HBM layout, rooflines) is in DESIGN.md.
//
// Pipeline per walker group (all fp64), scaling='uniform':
//   k_prep      : per-walker constants, 45-entry line-profile table, regrid cells
//   k_surface   : deprojection + line-of-sight velocity per nested-grid sample
//   k_profile   : line profile into velocity channels + nested-layer merge ->
//                 sparse Iunif (only 32x8 tiles with emission) + occupancy bytes
//   k_compact   : list of conv windows that contain emission
//   k_conv_rows / k_conv_chi2 : per window beam convolution in LDS (factorised
//                 beam: rows owned by waves for one-term beams, else phase by
//                 phase; or direct taps), regrid to observed pixels (folded into
//                 the taps where it is a constant shift), sum(I*d), sum(I*I)
//   k_finalize  : fixed-order partial reduction -> scale, chi2
// 'mom0*' scalings: k_prep/k_surface/k_profile, k_tile_krange, k_mom_model,
// k_regrid_mom0, then k_compact + k_conv_rows<OBS> on the observed grid (one-term
// beam of 25..32 padded taps) or k_obs_list + k_conv_obs_live, k_sum_chi2.  Dense single-vector products (cubemodel, parity taps):
// k_conv_sep / k_conv_2d, k_regrid, k_scale, k_peak_unity.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "slamhip.h"

namespace {

constexpr int NP = SLAMHIP_NPARAM;
constexpr int PROF_LEN = SLAMHIP_PROF_LEN;   // 45
constexpr int PROF_N = PROF_LEN - 1;         // 44 (:427)
constexpr double IDX_GUARD = 1.0e-9;         // |iv - integer| below this -> exact path
constexpr int CH = 16;                       // channels accumulated in registers
constexpr int CONV_TX = 64;                  // conv tile width (= one wave)
constexpr int CONV_R = 8;                    // outputs per thread along the tap axis
constexpr int RED_THREADS = 256;
// Fixed-point table index of k_profile: t (table steps) is formed as
// (t + IDX_OFF) * 2^32 + IDX_FGUARD + 2^52, a double in [2^52, 2^53) whose low
// mantissa word is t's fraction in units of 2^-32 and whose HIGH word is
// IDX_HI0 + floor(t) + IDX_OFF.
constexpr double IDX_SCALE = 4294967296.0;            // 2^32
constexpr int IDX_OFF = 2048;                         // makes the scaled value positive (|t| < 2^11)
constexpr unsigned IDX_FGUARD = 4096;                 // guard, in units of 2^-32 table steps
constexpr double IDX_BIAS = (22.5 + IDX_OFF) * IDX_SCALE + IDX_FGUARD + 4503599627370496.0;   // ... + 2^52
constexpr int IDX_HI0 = 0x43300000 + IDX_OFF;         // high word of the sum for floor(t) = 0
constexpr int MAX_RANK = 24;                 // beam outer-product terms the fused kernel accepts
constexpr unsigned LEAF_PAD = 0xffffffffu;    // slot without a sample
constexpr int SEP_PITCH = (CONV_TX + 32) | 1;          // fused factorised beam, kpad == 32: the pitch fz_tile_bytes gives, fixed at compile time
constexpr int DIRECT_PITCH = (CONV_TX + 32) | 1;       // fused direct taps, kpad <= 32: LDS tile pitch (room for the staging's column shift)
constexpr int DIRECT_MIN_RANK = 6;           // beams needing this many terms or more get direct taps in the fused kernel
constexpr int PREP_TILES = 1024;             // conv windows per axis pair k_prep can index (ntx + nty)
// counter words of a launch group (zeroed by k_prep): [0] live windows, [1] window ticket, [2] k_plan's
// items, [16..31] slamhip_group_counts, [32..47] phase stamps, [64 + 32 x] window ticket of XCD x
constexpr int LIVECOUNT_WORDS = 320;
constexpr int FLAG_MAX = 512;                // occupancy bytes one conv window may span

// ---------------------------------------------------------------------------
// Device-resident constants of one fit (what makegrid leaves on `self`).
// ---------------------------------------------------------------------------
struct FitK {
  int nv, ny, nx, nlayer, npix, nq1, nq3, ineed0, ineed1, nneed, kbeam, ngauss;
  int disk, envelope;
  int nvpad;            // nv rounded up to a multiple of CH
  long long nleaf;      // nested-grid samples that survive the layer merge
  long long nslot;      // sample slots: per 32x8 tile, [sample rank][pixel], padded
  double dv, sigma, pixperbeam, incl0, signmajor, signminor, vunit;
  double ssq_data;      // sum(data_valid^2), fixed per fit
  int data_has_nan;
  int fast_idx;         // k_profile may use the fixed-point table index
  int kpad;             // beam taps padded to a multiple of CONV_R
  int fold;             // the fused kernel may fold the regrid into the beam taps (one-term beam)
  double* tk;           // per-group scratch [walker][nvpad]: v_k * rinv * 2^32 (k_prep -> k_profile)
  const double* Xnest0;
  const double* Ynest0;
  const double* xneed;
  const double* yneed;
  const double* xobs;
  const double* yobs;
  const double* vpad;       // [nvpad] v_valid padded with +1e300
  const double* ch_vmin;    // [nvpad/CH]
  const double* ch_vmax;
  const double* data;       // [nv][ny][nx]
  const double* beam;       // [K][K]
  const double* mom0decon;  // [ny][nx] deconvolved moment 0, or null (scaling='uniform')
  const double* gx_pad;     // separable factors, flipped + zero padded
  const double* gy_pad;
  const double* beam_ft;    // [K][kpad] direct taps for the fused kernel: beam_ft[b][a] = beam[K-1-a][K-1-b], zero padded
  const unsigned* leaf;     // [nslot] packed (layer:4 | row:14 | col:14), LEAF_PAD = none
  const int* tile_goff;     // [tiles] first slot of each 32x8 pixel tile
  const int* tile_nl;       // [tiles] sample ranks in the tile (max samples per pixel)
};

// Per-walker constants written by k_prep.
struct WalkerK {
  double cp, sp;                 // cos/sin(radians(paoff))            :374
  double sini, cosi, tani;       // :377-380
  double sqrtM, Rc, Rin, pI, Ienv, voff, xoff, yoff;
  double inv_ienv1;              // 1/(Ienv+1)
  double prof_d, rinv;           // table step (:418), 1/(dv*prof_d)
  double halfwidth;              // |v - vlos - voff| beyond this -> prof == 0
  int nsurf;                     // live surfaces (<= 4)
  int slot_surf[4];              // reference slot -> live surface (-1 = None)
  int kind[4];                   // 1 thin, 2 one root, 3 root +, 4 root -
  double h2[4];                  // h^2
  double hfac[4];                // (1+h^2)^(-3/4)                    :435
  double qa[4];                  // tan^-2 i - h^2                    :393
  double iqa[4];                 // 1 / qa (the roots multiply by it: <= 1 ulp from the division of :405,:408)
  double qk[4];                  // tan^2 i - h^2                     :395
  double wmul[4];                // 2 for the doubled thin surface    :389-390
  double prof[PROF_LEN];
  // regrid folded into the beam (k_prep): per axis (x, y) the observed pixel of model
  // cell c is fi0 + fstep * (c - fc0), for pixels flo .. fhi
  int fold;
  int fc0[2], fi0[2], fstep[2], flo[2], fhi[2];
};

// ---------------------------------------------------------------------------
// k_prep: one 256-thread block per walker.
// ---------------------------------------------------------------------------
__device__ inline double sgn(double x) { return (x > 0) - (x < 0); }

__device__ inline int find_cell(const double* __restrict__ gr, int n, double q,
                                double inv_step) {
  // i with gr[i] <= q < gr[i+1], clipped to [0, n-2] (searchsorted semantics)
  int i = (int)floor((q - gr[0]) * inv_step);
  i = min(max(i, 0), n - 2);
  while (i > 0 && q < gr[i]) --i;
  while (i < n - 2 && q >= gr[i + 1]) ++i;
  return i;
}

// Cell index and fractional position of one observed coordinate on the model
// axis `gr` (scipy RegularGridInterpolator, linear, bounds_error=False,
// fill_value=0; :479-481): cell -1 = outside (value 0), -2 = NaN query.
__device__ inline void regrid_cell(const double* __restrict__ gr, int n, double q,
                                   int* cell, double* frac) {
  int c;
  double t = 0.0;
  if (q != q) {
    c = -2;
  } else if (q < gr[0] || q > gr[n - 1]) {
    c = -1;
  } else {
    c = find_cell(gr, n, q, 1.0 / (gr[1] - gr[0]));
    t = (q - gr[c]) / (gr[c + 1] - gr[c]);
  }
  *cell = c;
  *frac = t;
}

__global__ void k_prep(FitK g, const double* __restrict__ params, int nw,
                       WalkerK* __restrict__ wk, int* __restrict__ cellx,
                       double* __restrict__ fracx, int* __restrict__ celly,
                       double* __restrict__ fracy, int* __restrict__ tile_ix,
                       int* __restrict__ tile_jy, int ntx, int nty, int sx, int sy,
                       unsigned char* __restrict__ winflag, int winbytes, int* __restrict__ livecount,
                       double* __restrict__ gxw, double* __restrict__ gyw,
                       float* __restrict__ gxw32, float* __restrict__ gyw32) {
  const int w = blockIdx.x;
  if (w >= nw) return;
  if (winflag) {
    // the marks of this walker's conv windows and the live-window / ticket
    // counters start every group at zero
    unsigned char* wf = winflag + (long long)w * winbytes;
    for (int i = threadIdx.x; i < winbytes; i += blockDim.x) wf[i] = 0;
    if (w == 0) for (int i = threadIdx.x; i < LIVECOUNT_WORDS; i += blockDim.x) livecount[i] = 0;
  }
  const double* p = params + (long long)w * NP;
  WalkerK& o = wk[w];
  const int t = threadIdx.x;
  if (t == 0) o.fold = 0;
  // rgi2d's query points X - xoff, Y - yoff (:481), per axis
  for (int i = t; i < g.nx; i += blockDim.x)
    regrid_cell(g.xneed, g.nneed, g.xobs[i] - p[8], &cellx[(long long)w * g.nx + i],
                &fracx[(long long)w * g.nx + i]);
  for (int j = t; j < g.ny; j += blockDim.x)
    regrid_cell(g.yneed, g.nneed, g.yobs[j] - p[9], &celly[(long long)w * g.ny + j],
                &fracy[(long long)w * g.ny + j]);
  // observed index range [lo, hi) whose cells fall in each conv tile's span
  // (tile tt owns cells [tt*s, (tt+1)*s)): min / max over the observed indices
  __shared__ int s_lo[PREP_TILES], s_hi[PREP_TILES];
  const bool tiles_ok = ntx + nty <= PREP_TILES;           // else the fused kernel is not used (create)
  for (int tl = t; tiles_ok && tl < ntx + nty; tl += blockDim.x) { s_lo[tl] = 0x7fffffff; s_hi[tl] = 0; }
  __syncthreads();
  for (int i = t; tiles_ok && i < g.nx + g.ny; i += blockDim.x) {
    const bool isx = i < g.nx;
    const int ii = isx ? i : i - g.nx;
    const int cell = isx ? cellx[(long long)w * g.nx + ii] : celly[(long long)w * g.ny + ii];
    if (cell < 0) continue;                                // outside the model grid, or NaN
    const int tt = cell / (isx ? sx : sy);
    if (tt >= (isx ? ntx : nty)) continue;
    atomicMin(&s_lo[isx ? tt : ntx + tt], ii);
    atomicMax(&s_hi[isx ? tt : ntx + tt], ii + 1);
  }
  __syncthreads();
  for (int tl = t; tiles_ok && tl < ntx + nty; tl += blockDim.x) {
    const bool isx = tl < ntx;
    const int tt = isx ? tl : tl - ntx;
    int* o2 = isx ? &tile_ix[((long long)w * ntx + tt) * 2] : &tile_jy[((long long)w * nty + tt) * 2];
    o2[0] = s_hi[tl] == 0 ? 0 : s_lo[tl];
    o2[1] = s_hi[tl];
  }
  // Folding the regrid into the beam (k_conv_chi2): when, on an axis, the observed
  // pixels inside the model grid step through consecutive cells with one and the
  // same fractional position t, the bilinear regrid (:475-483) is itself a two-tap
  // convolution [1-t, t], and beam-then-regrid equals ONE pass with the K+1 taps
  // (1-t) g[u] + t g[u-1] whose outputs land on the observed pixels directly.
  // Checked per walker and axis (the offsets move t); otherwise the general path.
  if (gxw) {
    __shared__ int s_first[2], s_last[2], s_cnt[2], s_ok;
    if (t == 0) {
      s_first[0] = s_first[1] = 0x7fffffff;
      s_last[0] = s_last[1] = -1;
      s_cnt[0] = s_cnt[1] = 0;
      s_ok = g.fold;
    }
    __syncthreads();
    for (int i = t; i < g.nx + g.ny; i += blockDim.x) {
      const bool isx = i < g.nx;
      const int ii = isx ? i : i - g.nx;
      const int cell = isx ? cellx[(long long)w * g.nx + ii] : celly[(long long)w * g.ny + ii];
      if (cell >= 0) {
        atomicMin(&s_first[isx ? 0 : 1], ii);
        atomicMax(&s_last[isx ? 0 : 1], ii);
        atomicAdd(&s_cnt[isx ? 0 : 1], 1);
      }
      if (cell == -2) s_ok = 0;                            // NaN query point
    }
    __syncthreads();
    bool ok = true;
    double tf[2] = {0.0, 0.0};
    int fc0[2] = {0, 0}, fi0[2] = {0, 0}, fst[2] = {1, 1};
    for (int ax = 0; ax < 2; ++ax) {
      const int nobs = ax == 0 ? g.nx : g.ny;
      const int* cl = ax == 0 ? cellx + (long long)w * g.nx : celly + (long long)w * g.ny;
      const double* fr = ax == 0 ? fracx + (long long)w * g.nx : fracy + (long long)w * g.ny;
      const int i0 = s_first[ax];
      if (i0 >= nobs) continue;                            // no pixel inside the grid: nothing to regrid
      if (s_cnt[ax] != s_last[ax] - i0 + 1) ok = false;    // the pixels inside the grid are one run
      const int c0 = cl[i0];
      const double t0 = fr[i0];
      tf[ax] = t0;
      const int step = (i0 + 1 < nobs && cl[i0 + 1] >= 0) ? cl[i0 + 1] - c0 : 1;
      if (step != 1 && step != -1) ok = false;
      fc0[ax] = c0; fi0[ax] = i0; fst[ax] = step;
      for (int i = t; i < nobs; i += blockDim.x) {
        const int c = cl[i];
        if (c < 0) continue;
        if (c != c0 + step * (i - i0) || !(fabs(fr[i] - t0) <= 1e-12)) ok = false;
      }
    }
    if (!ok) s_ok = 0;
    __syncthreads();
    const bool fold = s_ok != 0;
    if (t == 0) {
      o.fold = fold ? 1 : 0;
      for (int ax = 0; ax < 2; ++ax) {
        o.fc0[ax] = fc0[ax]; o.fi0[ax] = fi0[ax]; o.fstep[ax] = fst[ax];
        o.flo[ax] = s_first[ax]; o.fhi[ax] = s_last[ax];   // empty run: flo > fhi
      }
    }
    if (fold) {
      for (int u = t; u < 2 * g.kpad; u += blockDim.x) {
        const bool isx = u < g.kpad;
        const int v = isx ? u : u - g.kpad;
        const double* gp = isx ? g.gx_pad : g.gy_pad;
        const double tt = tf[isx ? 0 : 1];
        const double tap = (1.0 - tt) * gp[v] + (v > 0 ? tt * gp[v - 1] : 0.0);
        (isx ? gxw : gyw)[(long long)w * g.kpad + v] = tap;
        if (gxw32) (isx ? gxw32 : gyw32)[(long long)w * g.kpad + v] = (float)tap;
      }
    }
  }
  const double cs = p[2];
  // update_prof (:412-427)
  const double c = cs / g.dv;
  const double wid = fmax(c * 2.35482, 1.0);
  const double vmax = 2.0 * wid;
  const double d = wid / 11.0;
  if (t < PROF_LEN) {
    const double step = (vmax - (-vmax)) / (double)PROF_N;
    double u = (double)t * step + (-vmax);
    if (t == PROF_N) u = vmax;
    double val;
    if (c < 0.01) {
      val = (1.0 + sgn(u + 0.5)) * (1.0 - sgn(u - 0.5)) / 4.0;
    } else {
      const double s2 = sqrt(2.0);
      val = erf((u + 0.5) / s2 / c) - erf((u - 0.5) / s2 / c);
      if (t == 0 || t == PROF_N) val = 0.0;
    }
    o.prof[t] = val;
  }
  {
    // channel velocities in units of 2^-32 table steps, for k_profile's
    // fixed-point index; padding channels land on prof[44] = 0
    const double R = (1.0 / (g.dv * d)) * IDX_SCALE;
    for (int k = t; k < g.nvpad; k += blockDim.x)
      g.tk[(long long)w * g.nvpad + k] = k < g.nv ? g.vpad[k] * R : 1000.0 * IDX_SCALE;
  }
  if (t == 0) {
    const double Mstar = p[0], Rc = p[1], h1 = p[3], h2v = p[4], pI = p[5];
    const double Rin = p[6], Ienv = p[7];
    const double deg = M_PI / 180.0;
    const double pa = p[12] * deg;
    o.cp = cos(pa);
    o.sp = sin(pa);
    const double inc = (g.incl0 + p[11]) * deg;
    o.sini = sin(inc);
    o.cosi = cos(inc);
    o.tani = tan(inc);
    o.sqrtM = sqrt(Mstar);
    o.Rc = Rc; o.Rin = Rin; o.pI = pI; o.Ienv = Ienv;
    o.xoff = p[8]; o.yoff = p[9]; o.voff = p[10];
    o.inv_ienv1 = 1.0 / (Ienv + 1.0);
    o.prof_d = d;
    o.rinv = 1.0 / (g.dv * d);
    // prof[idx] != 0 only for idx in [1, 43]: |(v-vlos-voff) * rinv| < 22.5
    o.halfwidth = 22.6 * fabs(g.dv * d);
    int ns = 0;
    const double hs[2] = {h1, h2v};
    for (int k = 0; k < 2; ++k) {
      const double h = hs[k];
      int* slot = &o.slot_surf[2 * k];
      slot[0] = slot[1] = -1;
      if (h < 0) continue;                       // :385-387
      const double hh = h * h;
      const double hf = pow(1.0 + hh, -3.0 / 4.0);
      const double a = pow(o.tani, -2.0) - hh;   // :393
      const double kq = o.tani * o.tani - hh;    // :395
      if (h < 0.01) {                            // :388-390, counted twice
        o.kind[ns] = 1; o.h2[ns] = hh; o.hfac[ns] = hf; o.qa[ns] = a;
        o.qk[ns] = kq; o.wmul[ns] = 2.0;
        slot[0] = slot[1] = ns++;
      } else if (-1e-3 < a && a < 1e-3) {        // :397-399
        o.kind[ns] = 2; o.h2[ns] = hh; o.hfac[ns] = hf; o.qa[ns] = a;
        o.qk[ns] = kq; o.wmul[ns] = 1.0;
        slot[0] = ns++;
      } else {                                   // :400-408
        for (int r = 0; r < 2; ++r) {
          o.kind[ns] = 3 + r; o.h2[ns] = hh; o.hfac[ns] = hf; o.qa[ns] = a;
          o.iqa[ns] = 1.0 / a;
          o.qk[ns] = kq; o.wmul[ns] = 1.0;
          slot[r] = ns++;
        }
      }
    }
    o.nsurf = ns;
  }
}

// ---------------------------------------------------------------------------
// Geometry of one sample on one surface: update_xdisk + getvlos.
// ---------------------------------------------------------------------------
// The kernels of the hot path exist in two arithmetic types T: double (the
// reference's, parity to 1e-9) and float (`precision` = 1 in the descriptor:
// geometry, line-profile accumulation and beam passes in single precision,
// 1e-3 parity; the truncating table index and the sums of :486-487, :601 stay
// in double in both).
template <typename T> struct Real;
template <> struct Real<double> {
  using vec2 = double2;
  static constexpr double masked = 1.0e300;              // stands in for a NaN vlos
  static constexpr double huge = 1.79769313486231570815e308;   // nan_to_num(inf), :468
  static constexpr double beyond = 1.0 + 1e-12;          // |Y| > Rc * this: masked whatever x is
  __device__ static double2 make2(double a, double b) { return make_double2(a, b); }
  __device__ static double nan() { return __builtin_nan(""); }
  // r**(-1/2) from the hardware estimate and two Newton steps (r is a finite, normal
  // radius here: masked and NaN samples never arrive): <= 3 ulp, a third of the
  // instructions of a square root and a division
  __device__ static double rsqrt(double r) {
    double y = __builtin_amdgcn_rsq(r);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const double e = fma(-(r * y), y, 1.0);
      y = fma(0.5 * y, e, y);
    }
    return y;
  }
  // r**(-p) as 2**(-p log2 r), for the weight r**(-pI) (:466-467).  Same accuracy as
  // exp(-p * log(r)) (relative error ~ (1 + |p log2 r|) ulp: measured max 4.4e-15, mean 3.6e-16
  // over r in [1e-2, 3e4], |p| <= 3 against numpy's pow, identical to the exp/log form) in about
  // half its instructions: no special cases inside (they take the library path below), the
  // logarithm from 2 atanh((m-1)/(m+1)) with m in [sqrt(1/2), sqrt(2)), both series in plain
  // Horner form.
  __device__ static double powneg(double r, double p) {
    if (!(r > 1.0e-300 && r < 1.0e300)) return exp(-p * log(r));      // 0, negative, inf, NaN, denormal
    double m = __builtin_amdgcn_frexp_mant(r);                        // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(r);
    if (m < 0.70710678118654752) { m = m + m; e -= 1; }
    const double f = m - 1.0, d = 2.0 + f;
    double q = __builtin_amdgcn_rcp(d);
    q = q * fma(-d, q, 2.0);
    q = q * fma(-d, q, 2.0);
    double s = f * q;
    s = fma(fma(-s, d, f), q, s);                                     // s = f / d to the last bit or so
    const double z = s * s;
    double P = 2.0 / 23.0;
    P = fma(P, z, 2.0 / 21.0); P = fma(P, z, 2.0 / 19.0); P = fma(P, z, 2.0 / 17.0);
    P = fma(P, z, 2.0 / 15.0); P = fma(P, z, 2.0 / 13.0); P = fma(P, z, 2.0 / 11.0);
    P = fma(P, z, 2.0 / 9.0);  P = fma(P, z, 2.0 / 7.0);  P = fma(P, z, 2.0 / 5.0);
    P = fma(P, z, 2.0 / 3.0);
    const double lnm = fma(s * z, P, s + s);                          // log(m)
    const double y = -p * ((double)e + lnm * 1.4426950408889634);      // -p log2(r)
    const double n = __builtin_rint(y);
    const double u = (y - n) * 0.6931471805599453;                    // |u| <= 0.347
    double E = 1.0 / 6227020800.0;                                    // sum u^k / k!, k <= 13
    E = fma(E, u, 1.0 / 479001600.0); E = fma(E, u, 1.0 / 39916800.0); E = fma(E, u, 1.0 / 3628800.0);
    E = fma(E, u, 1.0 / 362880.0);    E = fma(E, u, 1.0 / 40320.0);    E = fma(E, u, 1.0 / 5040.0);
    E = fma(E, u, 1.0 / 720.0);       E = fma(E, u, 1.0 / 120.0);      E = fma(E, u, 1.0 / 24.0);
    E = fma(E, u, 1.0 / 6.0);         E = fma(E, u, 0.5);              E = fma(E, u, 1.0);
    E = fma(E, u, 1.0);
    const double nc = fmin(fmax(n, -2200.0), 2200.0);                 // ldexp saturates to 0 / inf beyond
    return __builtin_amdgcn_ldexp(E, (int)nc);
  }
};
template <> struct Real<float> {
  using vec2 = float2;
  static constexpr float masked = 1.0e30f;
  static constexpr float huge = 3.402823466e38f;
  static constexpr float beyond = 1.0f + 1e-5f;
  __device__ static float2 make2(float a, float b) { return make_float2(a, b); }
  __device__ static float nan() { return __builtin_nanf(""); }
  __device__ static float rsqrt(float r) {
    float y = __builtin_amdgcn_rsqf(r);
    const float e = fmaf(-(r * y), y, 1.0f);             // one Newton step: ~1 ulp
    return fmaf(0.5f * y, e, y);
  }
  __device__ static float powneg(float r, float p) { return __expf(-p * __logf(r)); }
};

template <typename T> struct Surf { T x, r, vlos; };

// Deprojected minor-axis coordinate on surface s (update_xdisk, :382-410).
template <typename T>
__device__ inline T surface_x(const WalkerK& w, int s, T X, T Y) {
  const int kind = w.kind[s];
  const T hh = (T)w.h2[s];
  if (kind == 1) return X / (T)w.cosi;                     // :389
  const T Xc = X * (T)w.cosi;                              // :392
  const T b = ((T)1 + hh) * Xc;                            // :394
  const T c = (T)w.qk[s] * (Xc * Xc) - hh * (Y * Y);       // :395-396
  if (kind == 2) return Xc + c / b / (T)2;                 // :398
  const T a = (T)w.qa[s];
  const T D = b * b - a * c;                               // :403
  if (!(D >= 0)) return Real<T>::nan();
  const T rt = sqrt(D);
  return Xc + (kind == 3 ? (b + rt) : (b - rt)) * (T)w.iqa[s];   // :405-408
}

// True where getvlos (:443-447) leaves NaN whatever the velocity is.
template <typename T>
__device__ inline bool surface_masked(const FitK& g, const WalkerK& w, T r) {
  const bool outer = r > (T)w.Rc;                          // :434
  return !(r == r) || r < (T)w.Rin || (!g.envelope && outer) || (!g.disk && !outer);
}

// Line-of-sight velocity for Mstar = 1 at (x, Y), radius r (getvlos, :433-447).
template <typename T>
__device__ inline T surface_vlos(const FitK& g, const WalkerK& w, int s, T x, T Y, T r) {
  const bool outer = r > (T)w.Rc;                          // :434
  const T rmh = Real<T>::rsqrt(r);                         // r**(-1/2); 1/r as its square
  const T ir = rmh * rmh;
  T vp = rmh * (T)w.hfac[s];                               // :435
  T vr = r * (T)0;                                         // :436
  if (g.envelope && outer) {
    vp = sqrt((T)w.Rc) * ir;                               // :438
    vr = -rmh * sqrt((T)2 - (T)w.Rc * ir);                 // :439
  }
  const T erot = Y * (T)g.signmajor * ir;                  // :440
  if (!(g.envelope && outer))                              // vr = 0: vr*erad adds (+-)0 for the finite r > 0 that get here
    return (vp * erot + vr) * (T)w.sini * (T)g.vunit;
  const T erad = x * (T)g.signminor * ir;                  // :441
  return (vp * erot + vr * erad) * (T)w.sini * (T)g.vunit; // :442
}

template <typename T>
__device__ inline Surf<T> surface_eval(const FitK& g, const WalkerK& w, int s, T X, T Y) {
  Surf<T> o;
  o.x = surface_x<T>(w, s, X, Y);
  // hypot without the overflow rescaling (|x|,|Y| are grid coordinates): <= 1 ulp
  o.r = sqrt(fma(o.x, o.x, Y * Y));                        // :433
  o.vlos = surface_masked<T>(g, w, o.r) ? Real<T>::nan() : surface_vlos<T>(g, w, s, o.x, Y, o.r);
  return o;
}

constexpr int SURF_WPT = 8;   // walkers per k_surface thread

// k_surface: thread per (SURF_WPT walkers, surviving sample).  Writes, per live surface,
//   vl = vlos*sqrt(Mstar)          (Real<T>::masked where the reference has NaN)
//   wt = intensity weight * layer-merge weight 4^-layer
// as one {vl, wt} pair laid out [walker][surface][slot]; slots are ordered per
// 32x8 pixel tile as [sample rank][pixel] so k_profile's lanes (= pixels) read
// consecutive entries for every rank.
template <typename T>
__global__ void __launch_bounds__(256)
k_surface(FitK g, const WalkerK* __restrict__ wk, typename Real<T>::vec2* __restrict__ geom,
          unsigned* __restrict__ gflag, float2* __restrict__ grange, int nw) {
  const long long leaf = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // slot
  if (leaf >= g.nslot) return;                             // whole waves: nslot is a multiple of 256
  const unsigned code = g.leaf[leaf];
  const bool pad = code == LEAF_PAD;                       // slot without a sample: an entry that adds nothing
  const int l = code >> 28, j = (code >> 14) & 0x3fff, i = code & 0x3fff;
  const long long idx = pad ? 0 : ((long long)l * g.npix + j) * g.npix + i;
  const T X0 = (T)g.Xnest0[idx], Y0 = (T)g.Ynest0[idx];
  constexpr T MASKED = Real<T>::masked;
  // the sample's coordinates serve SURF_WPT walkers
  for (int wi = blockIdx.z * SURF_WPT, we = min(wi + SURF_WPT, nw); wi < we; ++wi) {
  const WalkerK& w = wk[wi];
  const T X = X0 * (T)w.cp - Y0 * (T)w.sp;                 // utils.py:355
  const T Y = X0 * (T)w.sp + Y0 * (T)w.cp;                 // utils.py:356
  const T lw = (T)1 / (T)(1 << (2 * (pad ? 0 : l)));       // avefour nesting :469-471
  T pXc = 0, pb = 0, prt = 0;                              // root '+' leaves these for root '-' (same h)
  // Most of the sky is masked (r > Rc, r < Rin): a wave whose 64 entries of a
  // surface all add nothing stores none of them and clears that surface's bit in
  // the wave's flag word, which k_profile reads before it asks for the entries.
  unsigned fm = 0;
  // r = hypot(x, Y) >= |Y|: without an envelope a sample with |Y| > Rc is masked on
  // every surface whatever x is (:443-447); a wave of such samples is done here
  const bool beyond = !g.envelope && fabs(Y) > (T)w.Rc * Real<T>::beyond;
  if (__ballot(!pad && !beyond) == 0ull) {
    if ((threadIdx.x & 63) == 0) {
      gflag[(long long)wi * (g.nslot >> 6) + (leaf >> 6)] = 0u;
      grange[(long long)wi * (g.nslot >> 6) + (leaf >> 6)] = make_float2(3.0e38f, -3.0e38f);
    }
    continue;
  }
  float vlo = 3.0e38f, vhi = -3.0e38f;                     // range of the stored vl of this lane
  for (int s = 0; s < w.nsurf; ++s) {
    T xs;
    if (w.kind[s] == 3) {                                  // :392-405, as surface_x
      pXc = X * (T)w.cosi;
      pb = ((T)1 + (T)w.h2[s]) * pXc;
      const T c = (T)w.qk[s] * (pXc * pXc) - (T)w.h2[s] * (Y * Y);
      const T D = pb * pb - (T)w.qa[s] * c;
      prt = D >= 0 ? sqrt(D) : Real<T>::nan();
      xs = pXc + (pb + prt) * (T)w.iqa[s];
    } else if (w.kind[s] == 4) {                           // :408, always right after its root '+'
      xs = pXc + (pb - prt) * (T)w.iqa[s];
    } else {
      xs = surface_x<T>(w, s, X, Y);
    }
    const T r = sqrt(fma(xs, xs, Y * Y));                  // :433
    T vl = MASKED, wt = 0;
    if (!pad && !surface_masked<T>(g, w, r)) {             // else NaN vlos: most of the sky
      vl = surface_vlos<T>(g, w, s, xs, Y, r) * (T)w.sqrtM;   // :460
      wt = (r < (T)w.Rc ? (T)1 : (T)w.Ienv) * (T)w.inv_ienv1;  // :465
      if (w.pI != 0.0) wt = wt * Real<T>::powneg(r, (T)w.pI);  // r**(-pI), :466-467
      wt = wt * (lw * (T)w.wmul[s]);
      if (!(vl == vl) || !(wt == wt)) {                    // NaN -> contributes 0 (:463,:468)
        vl = MASKED;
        wt = 0;
      } else {
        if (fabs(vl) > MASKED) vl = MASKED;
        if (fabs(wt) > Real<T>::huge)                      // nan_to_num(inf) :468
          wt = copysign(Real<T>::huge, wt);
      }
    }
    if (__ballot(wt != (T)0 && vl != MASKED) != 0ull) {    // wave-uniform
      geom[((long long)wi * 4 + s) * g.nslot + leaf] = Real<T>::make2(vl, wt);
      fm |= 1u << s;
      if (wt != (T)0 && vl != MASKED) {
        // single precision, widened by more than its rounding: only k_plan's cull reads it
        const float f = (float)vl, e = fabsf(f) * 2.4e-7f + 1.0e-30f;
        vlo = fminf(vlo, f - e);
        vhi = fmaxf(vhi, f + e);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    vlo = fminf(vlo, __shfl_xor(vlo, off, 64));
    vhi = fmaxf(vhi, __shfl_xor(vhi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    gflag[(long long)wi * (g.nslot >> 6) + (leaf >> 6)] = fm;
    grange[(long long)wi * (g.nslot >> 6) + (leaf >> 6)] = make_float2(vlo, vhi);
  }
  }
}

// Tap for parity tests: xdisk / vlos(Mstar=1) on the full nested grid.
__global__ void k_surface_tap(FitK g, const WalkerK* __restrict__ wk, int s,
                              double* __restrict__ x_out,
                              double* __restrict__ v_out) {
  const long long n = (long long)g.nlayer * g.npix * g.npix;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const WalkerK& w = wk[0];
  const double X0 = g.Xnest0[idx], Y0 = g.Ynest0[idx];
  const double X = X0 * w.cp - Y0 * w.sp;
  const double Y = X0 * w.sp + Y0 * w.cp;
  const Surf<double> f = surface_eval<double>(g, w, s, X, Y);
  x_out[idx] = f.x;
  v_out[idx] = f.vlos;
}

// ---------------------------------------------------------------------------
// k_profile: thread per output pixel of the cropped layer-0 grid.  For each
// block of CH channels the thread walks its samples x surfaces, looks the
// line profile up with the reference's truncating index (:462-463) and
// accumulates in registers; then stores Iunif[k][j][i] (coalesced along i).
// ---------------------------------------------------------------------------
__device__ inline int prof_index(double vk, double vl, double o, double rinv,
                                 const FitK& g, const WalkerK& w) {
  // fast form (o = vl + voff): differs from the reference's iv by < 1e-12
  const double t = fma(vk - o, rinv, 22.5);
  int idx = (int)t;                       // saturating; NaN -> 0
  if (fabs(t - rint(t)) < IDX_GUARD && t > 0.5 && t < 45.5) {
    // too close to a table boundary: redo in the reference's exact order
    //   iv = ((v_valid - vlos) - voff) / dv / prof_d + prof_n//2 + 0.5   (:461-462)
    double v = vk - vl;
    v = v - w.voff;
    const double iv = v / g.dv / w.prof_d + 22.0 + 0.5;
    idx = (int)iv;
  }
  return min(max(idx, 0), PROF_N);
}

constexpr int PT_X = 32, PT_Y = 8;   // k_profile tile: 256 pixels, 2 rows per wave

__device__ inline unsigned wave_or(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
  return v;
}

// k_plan: which (walker, 32x8 pixel tile, 16-channel block) triples can hold
// emission at all.  A thread takes one (walker, tile): the velocity range of
// the tile's stored samples (k_surface's per-wave ranges) against every channel
// block's window, exactly the test k_profile applies per sample but on the
// tile's range, so nothing that could light a block is left out.  The triples
// that pass are appended to `items` (count in *nitems), a tile's blocks next to
// each other; the occupancy bytes of all others stay at the zero the launch
// group's memset left.  Most of the sky is empty in most channels: at the
// headline workload about one triple in eight is listed.
__global__ void __launch_bounds__(256)
k_plan(FitK g, const WalkerK* __restrict__ wk, const float2* __restrict__ grange, int nw, int ntile,
       uint4* __restrict__ items, int* __restrict__ nitems) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nchunk = g.nvpad / CH;
  int wi = 0, tile = 0, goff = 0, nl = 0;
  float lo = 3.0e38f, hi = -3.0e38f;
  const bool live = t < nw * ntile;
  if (live) {
    wi = t / ntile;
    tile = t - wi * ntile;
    goff = g.tile_goff[tile];
    nl = g.tile_nl[tile];
    const int g0 = goff >> 6, ng = nl * 4;
    const float2* r = grange + (long long)wi * (g.nslot >> 6) + g0;
    for (int q = 0; q < ng; ++q) {
      const float2 v = r[q];
      lo = fminf(lo, v.x);
      hi = fmaxf(hi, v.y);
    }
  }
  // items of this thread, counted first, then one slot reservation per wave
  int mine = 0;
  double hw = 0.0, voff = 0.0;
  if (live && lo <= hi) {
    hw = wk[wi].halfwidth;
    voff = wk[wi].voff;
    for (int cix = 0; cix < nchunk; ++cix) {
      const double cmid = 0.5 * (g.ch_vmin[cix] + g.ch_vmax[cix]);
      const double hwc = hw + 0.5 * (g.ch_vmax[cix] - g.ch_vmin[cix]);
      mine += ((double)lo + voff <= cmid + hwc) && ((double)hi + voff >= cmid - hwc);
    }
  }
  const int lane = threadIdx.x & 63;
  int incl = mine;                                 // inclusive prefix sum over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  // one slot reservation per workgroup (a single counter serves its returning atomics one
  // after the other at memory): the waves' totals meet in LDS
  __shared__ int s_tot[4], s_base;
  if (lane == 63) s_tot[threadIdx.x >> 6] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int all = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    s_base = all ? atomicAdd(nitems, all) : 0;
  }
  __syncthreads();
  int base = s_base + incl - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_tot[w];
  if (mine) {
    for (int cix = 0; cix < nchunk; ++cix) {
      const double cmid = 0.5 * (g.ch_vmin[cix] + g.ch_vmax[cix]);
      const double hwc = hw + 0.5 * (g.ch_vmax[cix] - g.ch_vmin[cix]);
      if (((double)lo + voff <= cmid + hwc) && ((double)hi + voff >= cmid - hwc))
        items[base++] = make_uint4((unsigned)(wi * ntile + tile), (unsigned)cix, (unsigned)goff, (unsigned)nl);
    }
  }
}

// Iunif is sparse (a channel map is mostly empty sky), so each (walker,
// channel, 32x8 tile) gets an occupancy byte and only occupied tiles are
// written.  Consumers either honour the flags (k_conv_chi2) or are handed a
// pre-zeroed buffer (dense taps).
#ifdef SLAMHIP_PHASE_STAMPS
__device__ unsigned long long g_pstamp[8];      // k_profile's phases, cycles of wave 0 summed over items
#define PSTAMP(p_) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if (threadIdx.x == 0) atomicAdd(&g_pstamp[p_], t_ - pprev); pprev = t_; } while (0)
#else
#define PSTAMP(p_) do { } while (0)
#endif
// One listed (walker, tile, channel block) triple, by the whole 256-thread workgroup.
template <bool FAST, typename T, typename TG>   // T: table and accumulators, TG: geometry entries
__device__ __forceinline__ void
profile_item(const FitK& g, const WalkerK* __restrict__ wk, const typename Real<TG>::vec2* __restrict__ geom,
             const unsigned* __restrict__ gflag, T* __restrict__ iunif, unsigned char* __restrict__ flags, int ftx, int fty,
             unsigned char* __restrict__ winflag, int wnx, int wny, int wsx, int wsy,
             const uint4* __restrict__ items, int it, T* s_prof, unsigned* s_mask) {
  using T2 = typename Real<TG>::vec2;
#ifdef SLAMHIP_PHASE_STAMPS
  unsigned long long pprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pprev) :: "memory");
#endif
  // double geometry: the reference's table entry, bit for bit (lanes near a table
  // boundary redo the index in the reference's operation order); float geometry
  // cannot tell, its fast index is the index
  constexpr bool EXACT = sizeof(TG) == 8;
  // the item carries its tile's first slot and rank count, so that the flag words and
  // the first rank's entries can be asked for at once (no table lookup in between)
  const uint4 item = items[it];                           // block-uniform: kept in SGPRs, so that
  const unsigned ixw = __builtin_amdgcn_readfirstlane(item.x);   // everything indexed by it is scalar too
  const int goff = (int)__builtin_amdgcn_readfirstlane(item.z);
  const int nl = (int)__builtin_amdgcn_readfirstlane(item.w);
  const int ntile = ftx * fty;
  const int wi = (int)(ixw / (unsigned)ntile);
  const int tile = (int)(ixw - (unsigned)wi * (unsigned)ntile);
  const int cix = (int)__builtin_amdgcn_readfirstlane(item.y);
  const int tby = tile / ftx, tbx = tile - tby * ftx;
  const WalkerK& w = wk[wi];
  const int tid = threadIdx.x;
  if (tid < PROF_LEN) s_prof[tid] = (T)w.prof[tid];
  const int n = g.nneed;
  const int npx = n * n;
  const int x = tbx * PT_X + (tid & (PT_X - 1));
  const int y = tby * PT_Y + tid / PT_X;
  const bool inb = x < n && y < n;
  const int pix = inb ? y * n + x : 0;
  const int nsurf = w.nsurf;
  const double rinv = w.rinv, hw = w.halfwidth, voff = w.voff;
  // this thread's first slot; rank i of surface s is at gw[s*nslot + i*256]
  const T2* gw = geom + (long long)wi * 4 * g.nslot + goff + tid;
  const long long ftile = ntile;
  unsigned char* fl = flags + (long long)wi * g.nv * ftile + tile;
  // k_surface's flag word of this wave's 64 samples of a rank: bit s = surface s
  // was stored (else every entry adds nothing and nothing is read).  The first
  // two are requested here, under the table's round trip.
  const unsigned* gf = gflag + (long long)wi * (g.nslot >> 6) + (goff >> 6) + (tid >> 6);
  const unsigned f0v = nl > 0 ? gf[0] : 0u, f1v = nl > 1 ? gf[4] : 0u;
  // the first rank's entries, asked for before its flag word is known (a surface that was
  // not stored holds stale bytes there: dropped below)
  T2 spec[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    spec[s] = nl > 0 ? gw[(long long)s * g.nslot] : Real<TG>::make2(Real<TG>::masked, (TG)0);

  PSTAMP(0);
  __syncthreads();                                        // s_prof ready
  PSTAMP(1);
  {
    const int kf = cix * CH + tid;                        // flag writer: thread c -> channel c
    // a sample lights the block only if vlos + voff lies within hw of [cmin, cmax]
    const double cmid = 0.5 * (g.ch_vmin[cix] + g.ch_vmax[cix]);
    const double hwc = hw + 0.5 * (g.ch_vmax[cix] - g.ch_vmin[cix]);
    const double* vk = g.vpad + cix * CH;
    T acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0;
    bool hit = false;
    double vks[CH];                                       // this block's channel velocities (scalar)
#pragma unroll
    for (int c = 0; c < CH; ++c) vks[c] = FAST ? 0.0 : vk[c];
    // fast table index (FAST: create checked that every |t| of a lit block is
    // below 2^11): this block's channels as k_prep scaled them, block-uniform
    const double R = rinv * IDX_SCALE;
    const double* __restrict__ tkp = g.tk + (long long)wi * g.nvpad + cix * CH;
    double tks[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) tks[c] = FAST ? tkp[c] : 0.0;
    // one stored {vlos, weight} entry into the block's 16 channels
    auto lookups = [&](const TG vl, const TG wtg) {
      const T wt = (T)wtg;
      const double o = (double)vl + voff;                 // the index arithmetic is double in both types
      if (fabs(o - cmid) > hwc) return;                   // no channel of this block is lit
      hit = true;
      // table index of every channel, fast form (differs from the reference's
      // iv by < 1e-12); lanes within IDX_GUARD of an integer redo it exactly
      int idx[CH];
      bool near = false;
      if (FAST) {
        // one fp64 add per channel (see IDX_BIAS): the low word of the sum is
        // t's fraction, the high word its integer part + IDX_HI0, kept in that
        // form until the clamp.  The sum is off by at most one unit of 2^-32 and
        // the fast t by < 1e-11 from the reference's; the bias carries
        // +IDX_FGUARD, so "near an integer" is fraction < 2*IDX_FGUARD, and the
        // high word is the reference's index whenever the lane is not near.
        const double base = fma(-o, R, IDX_BIAS);
        unsigned lowest = 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const double sum = tks[c] + base;
          lowest = min(lowest, (unsigned)__double2loint(sum));
          idx[c] = __double2hiint(sum);
        }
        near = EXACT && lowest < 2u * IDX_FGUARD;          // float: the fast index is the index
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const double t = fma(vks[c] - o, rinv, 22.5);
          near |= EXACT && fabs(t - rint(t)) < IDX_GUARD;
          idx[c] = (int)t;                                // saturating; NaN -> 0
        }
      }
      if (near) {
        // iv = ((v_valid - vlos) - voff) / dv / prof_d + prof_n//2 + 0.5   (:461-462)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          double v = (FAST ? vk[c] : vks[c]) - (double)vl;
          v = v - voff;
          const int ix = (int)(v / g.dv / w.prof_d + 22.0 + 0.5);
          acc[c] = fma(s_prof[min(max(ix, 0), PROF_N)], wt, acc[c]);
        }
      } else {
        constexpr int I0 = FAST ? IDX_HI0 : 0;            // FAST: idx holds IDX_HI0 + index
#pragma unroll
        for (int c = 0; c < CH; ++c)
          acc[c] = fma(s_prof[min(max(idx[c], I0), I0 + PROF_N) - I0], wt, acc[c]);
      }
    };
    // Flags run two ranks ahead of the work, entries one (the word for rank i+2
    // stays in a VGPR until it is needed: moving it to an SGPR at once would
    // wait for the load).  Tried and not kept: walking only the stored (rank, surface)
    // pairs with four loads in flight (29.8 instead of 28.2 us/eval); entries two ranks
    // ahead in three register sets that trade places, every rank issuing the same four
    // loads (139 VGPRs, 3 waves per SIMD: 31.0 us/eval); one register set refilled surface
    // by surface with exactly counted loads, the guard and the lookups as two passes over
    // the 16 sums (96 VGPRs, 5 waves per SIMD: 30.5; the same at 4 waves: 32.1); this form
    // capped at 5 waves per SIMD (spills 52 B: 32.6).  Halving the resident workgroups
    // (2 per CU) costs 41.0: the kernel wants occupancy, but every cheaper-register form
    // so far pays more in extra loads and instructions than the fifth wave returns.
    // Two listed items per workgroup as straight-line copies (to halve the grid's empty
    // tail, 2.4 us/eval): the kernel's scalars then stay live across the first item, the 16
    // scaled channel values (32 SGPRs) move to vector registers and 112 B spill inside the
    // rank loop; the scalar budget, not the loop form, is what a second item runs into.
    PSTAMP(2);
    unsigned fcur = 0, fnxt = __builtin_amdgcn_readfirstlane(f0v);
    unsigned fnn = f1v;
    T2 q[4], qn[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      qn[s] = ((fnxt >> s) & 1u) ? spec[s] : Real<TG>::make2(Real<TG>::masked, (TG)0);
    for (int i = 0; i < nl; ++i) {
#pragma unroll
      for (int s = 0; s < 4; ++s) q[s] = qn[s];
      fcur = fnxt;
      fnxt = __builtin_amdgcn_readfirstlane(fnn);
      if (i + 1 < nl) {                                   // next rank's loads fly during this one
        fnn = i + 2 < nl ? gf[(i + 2) * 4] : 0u;
#pragma unroll
        for (int s = 0; s < 4; ++s)
          qn[s] = ((fnxt >> s) & 1u) ? gw[(long long)s * g.nslot + (i + 1) * 256] : Real<TG>::make2(Real<TG>::masked, (TG)0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (!((fcur >> s) & 1u)) continue;                // surface absent, or nothing stored for this wave
        lookups(q[s].x, q[s].y);
      }
    }
    PSTAMP(3);
    const bool anyhit = __syncthreads_or(hit);
    PSTAMP(4);
    if (!anyhit) return;                                  // empty sky after all: the flags stay 0
    unsigned m = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) m |= (unsigned)(acc[c] != (T)0) << c;
    m = wave_or(m);
    if ((tid & 63) == 0) s_mask[tid >> 6] = m;
    __syncthreads();
    m = s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3];
    PSTAMP(5);
    if (tid < CH && kf < g.nv) {
      if ((m >> tid) & 1u) {                              // lit: occupancy byte, and the conv windows this tile feeds
        fl[(long long)kf * ftile] = 1;
        const int px0 = tbx * PT_X, py0 = tby * PT_Y;
        const int wbx0 = max(0, (px0 - (CONV_TX - 1) - g.ngauss + wsx - 1) / wsx);
        const int wbx1 = min(wnx - 1, (px0 + PT_X - 1 + g.ngauss) / wsx);
        const int wby0 = max(0, (py0 - wsy - g.ngauss + wsy - 1) / wsy);
        const int wby1 = min(wny - 1, (py0 + PT_Y - 1 + g.ngauss) / wsy);
        unsigned char* wf = winflag + ((long long)wi * g.nv + kf) * wnx * wny;
        for (int by = wby0; by <= wby1; ++by)
          for (int bx = wbx0; bx <= wbx1; ++bx) wf[by * wnx + bx] = 1;
      }
    }
    T* out = iunif + (long long)wi * g.nv * npx + pix;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = cix * CH + c;
      if (((m >> c) & 1u) && inb && k < g.nv) out[(long long)k * npx] = acc[c];
    }
    PSTAMP(6);
  }
}

// k_profile: one workgroup per listed (walker, tile, channel block).  XCD-aware order:
// blocks b and b+8 share an XCD (and its L2), and a tile's channel blocks, which all read
// the tile's geometry, are neighbours in the list: XCD x takes the x-th eighth of the list.
// The host cannot know the list's length without waiting for it, and a grid sized for every
// possible triple is mostly workgroups that leave at once (2.4 us/eval at C3): this launch
// covers the first gridDim.x / 8 items of every eighth (a quarter of all possible triples),
// k_profile_rest whatever a longer list leaves over.
template <bool FAST, typename T, typename TG>
__global__ void __launch_bounds__(256)
k_profile(FitK g, const WalkerK* __restrict__ wk, const typename Real<TG>::vec2* __restrict__ geom,
          const unsigned* __restrict__ gflag, T* __restrict__ iunif, unsigned char* __restrict__ flags, int ftx, int fty,
          unsigned char* __restrict__ winflag, int wnx, int wny, int wsx, int wsy,
          const uint4* __restrict__ items, const int* __restrict__ nitems) {
  __shared__ T s_prof[PROF_LEN];
  __shared__ unsigned s_mask[4];
  const int nit = *nitems;
  const int per_xcd = (nit + 7) >> 3;
  const int it = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || it >= nit) return;
  profile_item<FAST, T, TG>(g, wk, geom, gflag, iunif, flags, ftx, fty, winflag, wnx, wny, wsx, wsy,
                            items, it, s_prof, s_mask);
}

// The items of each eighth beyond the first `done8`: a fixed number of workgroups walk them.
// The loop costs registers (3 waves per SIMD instead of 4), so this is the overflow path:
// it has work only when more than a quarter of all possible triples can hold emission.
template <bool FAST, typename T, typename TG>
__global__ void __launch_bounds__(256)
k_profile_rest(FitK g, const WalkerK* __restrict__ wk, const typename Real<TG>::vec2* __restrict__ geom,
               const unsigned* __restrict__ gflag, T* __restrict__ iunif, unsigned char* __restrict__ flags, int ftx, int fty,
               unsigned char* __restrict__ winflag, int wnx, int wny, int wsx, int wsy,
               const uint4* __restrict__ items, const int* __restrict__ nitems, int done8) {
  __shared__ T s_prof[PROF_LEN];
  __shared__ unsigned s_mask[4];
  const int nit = *nitems;
  const int per_xcd = (nit + 7) >> 3;
  const int rest8 = per_xcd - done8;
  if (rest8 <= 0) return;
  for (int r = blockIdx.x; r < 8 * rest8; r += gridDim.x) {
    const int it = (r & 7) * per_xcd + done8 + (r >> 3);
    if (it < nit)                                          // block-uniform
      profile_item<FAST, T, TG>(g, wk, geom, gflag, iunif, flags, ftx, fty, winflag, wnx, wny, wsx, wsy,
                                items, it, s_prof, s_mask);
    __syncthreads();                                       // table and masks free for the next item
  }
}

// ---------------------------------------------------------------------------
// Beam convolution (:530): zero-padded 'same' convolution per channel.
//
// k_conv_sep: separable beam, row pass then column pass on one LDS tile.
// The row-pass results are held in registers across the barrier so the same
// LDS buffer serves as input tile and as the intermediate.  Each thread slides
// a CONV_R-wide register window along the tap axis: 1 LDS read per CONV_R FMAs.
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over the wave in the VALU's data-parallel-primitive network, no LDS round trips:
// butterflies within each row of 16 lanes (every lane then holds its row's sum), row 0
// into row 1 and row 2 into row 3, rows 0+1 into row 3.  Returns the total in every lane.
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
  return v + __hiloint2double(hi, lo);                   // lanes outside ROWS add +0
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v = dpp_add<0xB1, 0xf>(v);                             // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);                             // quad_perm [2,3,0,1]
  v = dpp_add<0x124, 0xf>(v);                            // row_ror:4
  v = dpp_add<0x128, 0xf>(v);                            // row_ror:8
  v = dpp_add<0x142, 0xa>(v);                            // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);                            // row_bcast:31 into rows 2 and 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

template <int TY>
struct ConvAcc {
  static constexpr int NYB = (TY / CONV_R) / (256 / CONV_TX);   // y-blocks per thread
  double v[NYB][CONV_R];
};

// Convolves the TX x TY output nodes whose corner is (x0, y0) with the
// separable beam.  On return thread (lane, wave) holds, for q < NYB, rows
// (wave + 4q)*R .. +R-1 of column `lane`.  Returns false (nothing computed)
// when the staged input tile is all zero.  Leaves the block past a barrier:
// LDS may be reused after one more __syncthreads().
// `fl` (may be null = dense input): occupancy bytes of this plane's 32x8 tiles,
// `ftx`/`fty` tiles per row/column; tiles flagged 0 are not read.
template <int TY, int MAXR>
__device__ inline bool conv_tile(const FitK& g, const double* __restrict__ src,
                                 const unsigned char* __restrict__ fl, int ftx, int fty,
                                 int x0, int y0, int kpad, double* lds,
                                 ConvAcc<TY>& acc) {
  constexpr int TX = CONV_TX, R = CONV_R;
  const int K = g.kbeam, ng = g.ngauss, n = g.nneed;
  const int rows = TY + kpad - 1;              // tile rows incl. halo + zero pad
  const int cols = TX + kpad - 1;
  const int P = cols | 1;                      // odd pitch: lanes<->rows conflict-free
  const int P2 = TX + 1;
  const int tid = threadIdx.x;

  // 1. stage the input tile (zero outside the image and in the pad).  Wave w
  //    takes rows w, w+4, ...; 8 independent loads in flight per lane before
  //    the first LDS write.
  const int lane = tid & 63, wv = tid >> 6;
  const int vrows = TY + K - 1, vcols = TX + K - 1;
  // occupancy of the 32x8 input tiles under this window
  __shared__ unsigned char s_fl[FLAG_MAX];
  const int ftx0 = max(x0 - ng, 0) / PT_X, fty0 = max(y0 - ng, 0) / PT_Y;
  const int fnx = min(x0 - ng + vcols - 1, n - 1) / PT_X - ftx0 + 1;
  const int fny = min(y0 - ng + vrows - 1, n - 1) / PT_Y - fty0 + 1;
  if (fl) {
    bool some = false;
    for (int e = tid; e < fnx * fny; e += 256) {
      const int fy = e / fnx, fx = e - fy * fnx;
      const unsigned char f = fl[(long long)(fty0 + fy) * ftx + ftx0 + fx];
      s_fl[e] = f;
      some |= f != 0;
    }
    if (!__syncthreads_or(some)) return false;             // nothing but empty sky
  }
  bool any = false;
  for (int c0 = 0; c0 < P; c0 += 64) {
    const int cc = c0 + lane;
    const int gx = x0 - ng + cc;
    const bool colok = cc < vcols && gx >= 0 && gx < n;
    const int fxl = colok ? gx / PT_X - ftx0 : 0;
    for (int r0 = wv; r0 < rows; r0 += 4 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = r0 + 4 * u;
        const int gy = y0 - ng + rr;
        bool ok = colok && rr < vrows && gy >= 0 && gy < n;
        if (fl && ok) ok = s_fl[(gy / PT_Y - fty0) * fnx + fxl] != 0;
        v[u] = ok ? src[(long long)gy * n + gx] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = r0 + 4 * u;
        any |= v[u] != 0.0;
        if (rr < rows && cc < P) lds[rr * P + cc] = v[u];
      }
    }
  }
  // an all-zero tile (most of a channel map is empty sky) convolves to zero
  if (!__syncthreads_or(any)) return false;

  // 2. row pass: tmp[row][x] = sum_u gxf[u] * tile[row][x+u]
  const int nitem = rows * (TX / R);
  double res[MAXR][R];
  const double* wx = g.gx_pad;                 // flipped taps, zero padded to kpad
#pragma unroll
  for (int q = 0; q < MAXR; ++q) {
    const int item = q * 256 + tid;
#pragma unroll
    for (int r = 0; r < R; ++r) res[q][r] = 0.0;
    if (item < nitem) {
      const int xb = item / rows, rr = item - xb * rows;
      const double* row = lds + rr * P + xb * R;
      double win[R];
#pragma unroll
      for (int r = 0; r < R - 1; ++r) win[r] = row[r];
      for (int u0 = 0; u0 < kpad; u0 += R) {
#pragma unroll
        for (int qq = 0; qq < R; ++qq) {
          win[(qq + R - 1) % R] = row[u0 + qq + R - 1];
          const double wgt = wx[u0 + qq];
#pragma unroll
          for (int r = 0; r < R; ++r)
            res[q][r] = fma(wgt, win[(r + qq) % R], res[q][r]);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < MAXR; ++q) {
    const int item = q * 256 + tid;
    if (item < nitem) {
      const int xb = item / rows, rr = item - xb * rows;
#pragma unroll
      for (int r = 0; r < R; ++r) lds[rr * P2 + xb * R + r] = res[q][r];
    }
  }
  __syncthreads();

  // 3. column pass: out[y][x] = sum_u gyf[u] * tmp[y+u][x]; lanes <-> x
  const double* wy = g.gy_pad;
#pragma unroll
  for (int q = 0; q < ConvAcc<TY>::NYB; ++q) {
    const int yb = wv + 4 * q;
    const double* col = lds + (yb * R) * P2 + lane;
    double win[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc.v[q][r] = 0.0;
#pragma unroll
    for (int r = 0; r < R - 1; ++r) win[r] = col[r * P2];
    for (int u0 = 0; u0 < kpad; u0 += R) {
#pragma unroll
      for (int qq = 0; qq < R; ++qq) {
        win[(qq + R - 1) % R] = col[(u0 + qq + R - 1) * P2];
        const double wgt = wy[u0 + qq];
#pragma unroll
        for (int r = 0; r < R; ++r)
          acc.v[q][r] = fma(wgt, win[(r + qq) % R], acc.v[q][r]);
      }
    }
  }
  return true;
}

// k_conv_sep: convolved cube to HBM (cubemodel / parity tap path).
template <int TY, int MAXR>
__global__ void __launch_bounds__(256)
k_conv_sep(FitK g, const double* __restrict__ in, double* __restrict__ out,
           int kpad) {
  extern __shared__ double lds[];
  constexpr int TX = CONV_TX, R = CONV_R;
  const int n = g.nneed;
  const long long plane = (long long)n * n;
  const double* src = in + (long long)blockIdx.z * plane;
  double* dst = out + (long long)blockIdx.z * plane;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  ConvAcc<TY> acc;
  if (!conv_tile<TY, MAXR>(g, src, nullptr, 0, 0, x0, y0, kpad, lds, acc)) {
    for (int e = tid; e < TX * TY; e += 256) {
      const int gy = y0 + e / TX, gxo = x0 + (e & (TX - 1));
      if (gxo < n && gy < n) dst[(long long)gy * n + gxo] = 0.0;
    }
    return;
  }
  const int gx = x0 + lane;
#pragma unroll
  for (int q = 0; q < ConvAcc<TY>::NYB; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gy = y0 + (wv + 4 * q) * R + r;
      if (gx < n && gy < n) dst[(long long)gy * n + gx] = acc.v[q][r];
    }
}

#ifndef CV_WB
#define CV_WB 8                       // walkers per block of k_compact's list order (0: walker-major)
#endif
constexpr int WIN_IDS = 56;           // covers every tile under a 64-row window of a K <= 32 beam
// One live conv window, everything k_conv_chi2 needs to start loading at once.
struct __attribute__((aligned(16))) WinDesc {
  int e;                         // window id: (walker*nv + channel)*wnx*wny + by*wnx + bx
  int ixa, ixb, jya, jyb;        // observed pixel ranges whose cells it owns
  int nocc;                      // occupied 32x8 input tiles under the window
  int z, tb;                     // plane walker*nv + channel; window column | row << 8 | walker (of the group) << 16 (spares the kernel's divisions)
  unsigned long long rowmask[4]; // window rows that hold emission
  unsigned short ids[WIN_IDS];   // the first WIN_IDS occupied tiles (fy << 8 | fx)
  unsigned long long occ[FLAG_MAX / 64];   // all of them, as a bitmap
};

// k_compact: list of conv windows that contain emission (order irrelevant:
// every window writes its partial sums to its own slot), each with the
// occupancy bitmap of the input tiles under it.
__global__ void __launch_bounds__(256)
k_compact(FitK g, const unsigned char* __restrict__ winflag, int total,
          const unsigned char* __restrict__ flags, int ftx, int fty,
          const int* __restrict__ tile_ix, const int* __restrict__ tile_jy,
          int wnx, int wny, int sx, int sy, int ty,
          WinDesc* __restrict__ list, int* __restrict__ count, const WalkerK* __restrict__ wk,
          int split, int nw, int nrow_arg) {
  // LIST ORDER.  Thread t is window (channel, window row, block of CV_WB walkers, window column,
  // walker of the block), the walker running fastest: neighbours in the list are the SAME window
  // of the observed cube for up to CV_WB walkers (their `data` pixels are one read from memory
  // and L2 hits after it; a workgroup's batch of four tickets keeps them in registers), then
  // the next window column of those walkers (shared halo tiles still in the L2).  Walker-major
  // order re-read `data` from memory for every walker: 6.9 of the launch's 13.2 GB.
  int e = blockIdx.x * 256 + threadIdx.x;
  bool inrange = e < total;
  if (CV_WB > 0) {
    const int t = e, nblk = (nw + CV_WB - 1) / CV_WB, nv = total / (nw * wnx * wny);
    const int wi_in = t % CV_WB, t1 = t / CV_WB;
    const int cx = t1 % wnx, t2 = t1 / wnx;
    const int wblk = t2 % nblk, t3 = t2 / nblk;
    const int cy = t3 % wny, kk = t3 / wny;
    const int wi_ = wblk * CV_WB + wi_in;
    inrange = wi_ < nw && kk < nv;
    e = ((wi_ * nv + kk) * wny + cy) * wnx + cx;
  }
  // one counter update per wave (a single address takes ~1 atomic per clock):
  // lane 0 reserves the wave's slots, each live lane takes its rank among them
  const bool live_any = inrange && winflag[e];
  // split: windows of walkers whose regrid does not fold into the beam go to a second list that
  // grows down from the end of `list` (count[3] entries: list[total - count[3] ..]), so that the
  // windows of folded walkers can be run by a kernel instance without the regrid phase
  const bool second = split && live_any && !wk[(e / (wnx * wny)) / g.nv].fold;
  const bool live = live_any && !second;
  const unsigned long long lanes = __ballot(live);
  const int lane = threadIdx.x & 63;
  const unsigned long long lanes2 = __ballot(second);
  // ONE counter update per workgroup: returning atomics on a single address are served
  // one after the other at memory (~12 ns each on this part: one per wave made them the
  // kernel's whole duration), so the four waves' counts meet in LDS first
  __shared__ int s_cnt[4], s_base, s_cnt2[4], s_base2;
  if (lane == 0) { s_cnt[threadIdx.x >> 6] = __popcll(lanes); s_cnt2[threadIdx.x >> 6] = __popcll(lanes2); }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    s_base = tot ? atomicAdd(count, tot) : 0;
    const int tot2 = s_cnt2[0] + s_cnt2[1] + s_cnt2[2] + s_cnt2[3];
    s_base2 = tot2 ? atomicAdd(count + 3, tot2) : 0;
  }
  __syncthreads();
  int slot = second ? s_base2 : s_base;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) slot += second ? s_cnt2[w] : s_cnt[w];
  slot += __popcll((second ? lanes2 : lanes) & ((1ull << lane) - 1ull));
  if (second) slot = total - 1 - slot;
  if (!live_any) return;
  const int n = g.nneed;
  const int z = e / (wnx * wny), tl = e - z * (wnx * wny);
  const int tby = tl / wnx, tbx = tl - tby * wnx;
  const int wi = z / g.nv;
  const int wx0 = tbx * sx - g.ngauss, wy0 = tby * sy - g.ngauss;
  const int vcols = CONV_TX + g.kbeam - 1, vrows = ty + g.kbeam - 1;
  const int ftx0 = max(wx0, 0) / PT_X, fty0 = max(wy0, 0) / PT_Y;
  const int fnx = min(wx0 + vcols - 1, n - 1) / PT_X - ftx0 + 1;
  const int nrow = nrow_arg ? nrow_arg : n;               // (the observed grid of the 'mom0*' path may be rectangular)
  const int fny = min(wy0 + vrows - 1, nrow - 1) / PT_Y - fty0 + 1;
  // The descriptor is assembled in its slot of the list; the bitmaps are kept in
  // named registers (static word indices) so that nothing lives in scratch.
  WinDesc* D = &list[slot];
  unsigned long long occ[FLAG_MAX / 64], rowm[4];
#pragma unroll
  for (int q = 0; q < FLAG_MAX / 64; ++q) occ[q] = 0ull;
#pragma unroll
  for (int q = 0; q < 4; ++q) rowm[q] = 0ull;
  const unsigned char* fl = flags + (long long)z * ftx * fty;
  int nocc = 0;
  auto lit_tile = [&](int fy, int q) {
#pragma unroll
    for (int wd = 0; wd < FLAG_MAX / 64; ++wd)
      if ((q >> 6) == wd) occ[wd] |= 1ull << (q & 63);
    if (nocc < WIN_IDS) D->ids[nocc] = (unsigned short)((fy << 8) | (q - fy * fnx));   // entries past nocc are never read
    ++nocc;
  };
  auto lit_row = [&](int fy) {
    const int r0 = max((fty0 + fy) * PT_Y - wy0, 0);
    const int r1 = min((fty0 + fy + 1) * PT_Y - wy0, vrows);   // exclusive
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
      const int lo = max(r0 - 64 * wd, 0), hi = min(r1 - 64 * wd, 64);
      if (lo < hi)
        rowm[wd] |= (hi - lo == 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
    }
  };
  constexpr int CR = 16, CC = 8;      // rows x bytes per row requested together
  if (fny <= CR && fnx <= CC) {
    // every row's occupancy bytes in one 8-byte request (the buffer is padded
    // for the over-read), all rows in flight together
    unsigned long long rb[CR];
#pragma unroll
    for (int fy = 0; fy < CR; ++fy) {
      rb[fy] = 0ull;
      if (fy < fny) __builtin_memcpy(&rb[fy], fl + (long long)(fty0 + fy) * ftx + ftx0, 8);
    }
#pragma unroll
    for (int fy = 0; fy < CR; ++fy) {
      if (fy >= fny) continue;
      bool any = false;
#pragma unroll
      for (int fx = 0; fx < CC; ++fx)
        if (fx < fnx && ((rb[fy] >> (8 * fx)) & 0xffull)) { lit_tile(fy, fy * fnx + fx); any = true; }
      if (any) lit_row(fy);
    }
  } else {
    for (int fy = 0; fy < fny; ++fy) {
      bool any = false;
      for (int fx = 0; fx < fnx; ++fx)
        if (fl[(long long)(fty0 + fy) * ftx + ftx0 + fx]) { lit_tile(fy, fy * fnx + fx); any = true; }
      if (any) lit_row(fy);
    }
  }
  D->e = e;
  if (wk[wi].fold) {
    // folded regrid: the observed pixel of the window's cell column j is ixa + (+-1) j for
    // ja <= j < jb; ixb packs {bit 0: step +1, bits 8-15: ja, bits 16-23: jb, bit 31: folded};
    // jya / jyb the same for rows
    const WalkerK& w = wk[wi];
    int out[4];
#pragma unroll
    for (int ax = 0; ax < 2; ++ax) {
      const int c0 = ax == 0 ? tbx * sx : tby * sy;            // the window's first cell
      const int ncell = min(ax == 0 ? sx : sy, n - 1 - c0);    // cells it owns
      const int p0 = w.fi0[ax] + w.fstep[ax] * (c0 - w.fc0[ax]);
      int ja, jb;
      if (w.fstep[ax] > 0) { ja = max(0, w.flo[ax] - p0); jb = min(ncell, w.fhi[ax] - p0 + 1); }
      else                 { ja = max(0, p0 - w.fhi[ax]); jb = min(ncell, p0 - w.flo[ax] + 1); }
      if (jb < ja) jb = ja;
      out[2 * ax] = p0;
      out[2 * ax + 1] = (int)((w.fstep[ax] > 0 ? 1u : 0u) | ((unsigned)ja << 8) | ((unsigned)jb << 16) | 0x80000000u);
    }
    D->ixa = out[0]; D->ixb = out[1]; D->jya = out[2]; D->jyb = out[3];
  } else if (!tile_ix) {                                  // windows on the observed grid itself (k_conv_rows<OBS>)
    D->ixa = D->ixb = D->jya = D->jyb = 0;
  } else {
    D->ixa = tile_ix[((long long)wi * wnx + tbx) * 2];
    D->ixb = tile_ix[((long long)wi * wnx + tbx) * 2 + 1];
    D->jya = tile_jy[((long long)wi * wny + tby) * 2];
    D->jyb = tile_jy[((long long)wi * wny + tby) * 2 + 1];
  }
  D->nocc = nocc;
  D->z = z;
  D->tb = tbx | (tby << 8) | (wi << 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) D->rowmask[q] = rowm[q];
#pragma unroll
  for (int q = 0; q < FLAG_MAX / 64; ++q) D->occ[q] = occ[q];
}

// k_conv_chi2: the hot-path form.  Persistent 512-thread workgroups walk the
// list of windows that contain emission.  Per window: stage only the occupied
// 32x8 input tiles into a zeroed LDS tile, separable row pass over the rows
// that hold emission (results in registers across the barrier), column pass
// (tap blocks over empty rows skipped), keep the convolved 64xTY nodes in LDS,
// regrid them bilinearly onto the observed pixels whose interpolation cell the
// window owns (:475-483) and reduce sum(I*d), sum(I*I) for get_scale
// (:485-490) with wave shuffles, then a block reduction.  The convolved cube
// never reaches HBM; empty sky costs neither FLOPs nor a read of the data cube.
// Windows overlap by one node: window (bx, by) owns cells [x0, x0+TX-1) x
// [y0, y0+TY-1) with x0 = bx*(TX-1), y0 = by*(TY-1).
constexpr int CV_NT = 512;                   // threads per workgroup
constexpr int CV_NW = CV_NT / 64;            // waves
#ifndef CV_NSB
#define CV_NSB 7                             // staged tiles per wave held in registers
#endif
#ifndef CV_MINW_48
#define CV_MINW_48 4
#endif
#ifndef CV_MINW_F32
// waves per SIMD the float instances are compiled for: their tile is half the size, three
// workgroups fit a CU's LDS, and 80 VGPRs (44 B of scratch in the hot instance) buy that
// third workgroup: k_conv_chi2 22.9 instead of 26.2 us/eval at C3
#define CV_MINW_F32 6
#endif

// Workgroup barrier for LDS traffic only: waits for this wave's LDS operations, not
// for its global loads (__syncthreads() also drains vmcnt, which would expose the
// latency of every load that is meant to fly across the barrier).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Row pass of one lane: C consecutive outputs res[c] += sum_u wx[u] row[c + u] over kpad
// taps, the C-wide window of `row` rotating through registers (one LDS read per tap);
// the taps come through the scalar cache (block-uniform address).  kpad is a multiple of RU.
// LDS read that the compiler may not pair with its neighbour: a ds_read2_b64 costs 8 LDS cycles
// for its 1 KiB, two ds_read_b64 cost 2 each (MI355X_MICROARCH.md, LDS table), and these kernels
// wait on LDS round trips (k_conv_rows: 30.7 instead of 32.2 us/eval at C3)
template <typename T> __device__ __forceinline__ T lds_rd(const T* p) {
  if constexpr (sizeof(T) == 8) {
    typedef const volatile __attribute__((address_space(3))) T* lds_ptr;   // stays an LDS (ds_read_b64) access
    return *(lds_ptr)(p);
  } else {
    return *p;                         // floats: the paired / wide read is the cheaper one
  }
}

template <int C, int RU, typename T>
__device__ __forceinline__ void row_taps(const T* row, const T* __restrict__ wx, int kpad, bool tail4, T* res) {
  static_assert(RU % C == 0, "the window rotation must close over one unrolled step");
  T win[C];
#pragma unroll
  for (int r = 0; r < C - 1; ++r) win[r] = lds_rd(row + r);
  // tail4 (block-uniform): the last RU / 2 of the kpad padded taps are zero padding (K + 1 <=
  // kpad - 4, e.g. the folded 28 taps of a K = 27 beam in 32): the last step runs half its taps
  const int kfull = tail4 ? kpad - RU : kpad;
  for (int u0 = 0; u0 < kfull; u0 += RU) {
#pragma unroll
    for (int qq = 0; qq < RU; ++qq) {
      win[(qq + C - 1) % C] = lds_rd(row + u0 + qq + C - 1);
      const T wgt = wx[u0 + qq];
#pragma unroll
      for (int r = 0; r < C; ++r) res[r] = fma(wgt, win[(r + qq) % C], res[r]);
    }
  }
  if (tail4) {
#pragma unroll
    for (int qq = 0; qq < RU / 2; ++qq) {
      win[(qq + C - 1) % C] = lds_rd(row + kfull + qq + C - 1);
      const T wgt = wx[kfull + qq];
#pragma unroll
      for (int r = 0; r < C; ++r) res[r] = fma(wgt, win[(r + qq) % C], res[r]);
    }
  }
}

// T: arithmetic type of tile and passes; PITCH: LDS tile pitch fixed at compile time (0: from
// kpad); MULTI: the factorised beam may have several terms (false: exactly one, the column
// pass's sums are then born after the row pass and cost it no registers)
template <int TY, int MAXR, bool DIRECT, int PITCH = 0, typename T = double, bool MULTI = true>
__global__ void __launch_bounds__(CV_NT, (sizeof(T) == 4 && TY == 64 && MAXR == 2 ? CV_MINW_F32 : TY == 48 ? CV_MINW_48 : 4))   // only the small float tile fits a CU three times
k_conv_chi2(FitK g, const T* __restrict__ iunif,
            const int* __restrict__ cellx, const double* __restrict__ fracx,
            const int* __restrict__ celly, const double* __restrict__ fracy,
            const WinDesc* __restrict__ list, int* __restrict__ count, int wnx, int wny,
            double* __restrict__ partial, int kpad_arg, int nrank,
            const T* __restrict__ gxp, const T* __restrict__ gyp,
            const T* __restrict__ gbt,
            const T* __restrict__ gxw, const T* __restrict__ gyw, int tail4_arg) {
  // the last four of the kpad padded taps are padding in every tap set of this launch
  // (K + 1 <= kpad - 4 with folded walkers, K <= kpad - 4 without): the last tap block of
  // both passes / of the direct taps' rows runs half its length (K = 27: 28 of 32 taps;
  // direct taps 166 instead of 175 us/eval at C3; the factorised form is not bound by its
  // multiply-adds and stays at 31.6)
  const bool tail4 = tail4_arg != 0;
  // the fixed-pitch factorised instance is only launched with 32 padded taps
  const int kpad = (PITCH != 0 && !DIRECT && sizeof(T) == 8) ? 32 : kpad_arg;
  // DIRECT (nrank == 0): the beam is not worth factorising (a rotated Gaussian
  // needs ~14 terms): direct K x K taps on the staged tile instead of row +
  // column passes
  extern __shared__ unsigned char lds_raw[];
  T* const lds = reinterpret_cast<T*>(lds_raw);
  using T2 = typename Real<T>::vec2;
  constexpr int TX = CONV_TX, R = CONV_R, PC = TX + 1, P2 = TX + 1;
  constexpr bool direct = DIRECT;
  constexpr int NRQ = (TY - 1 + CV_NW - 1) / CV_NW;         // regrid rows per wave
  constexpr int NSB = CV_NSB;                                  // staged tiles per wave held in registers
  __shared__ unsigned char s_rowlist[256];
  __shared__ double sg[CV_NW], sf[CV_NW];
  const int K = g.kbeam, ng = g.ngauss, n = g.nneed;
  const int rows = TY + kpad - 1, cols = TX + kpad - 1;
  // a compile-time pitch turns the direct taps' row offsets into immediates
  const int P = PITCH ? PITCH : ((cols + 1) | 1);           // odd, with room for the column shift of the staging
  const int vrows = TY + K - 1, vcols = TX + K - 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index, kept scalar
  // beam = sum over nrank outer products gy[r] (x) gx[r] (1 for an axis-aligned
  // Gaussian, the leading singular triplets for a rotated one).  With one term
  // the row-pass output overlays the input tile; with several the tile must
  // survive, so the intermediate gets LDS of its own.
  // taps sit after the tile; DIRECT also parks two TY x PC half sums in the tile's place
  T* wxl = lds + (DIRECT ? max(rows * P + 1, 2 * TY * PC) : rows * P + 1);   // [nrank][kpad] flipped taps, zero padded
  T* wyl = wxl + nrank * kpad;
  T* tmp = nrank == 1 ? lds : wyl + nrank * kpad;      // [rows][P2]
  for (int e = tid; e < nrank * kpad; e += CV_NT) { wxl[e] = (T)g.gx_pad[e]; wyl[e] = (T)g.gy_pad[e]; }
  __syncthreads();
  const int nlive = *count;
  constexpr int DW = sizeof(WinDesc) / 4;                   // descriptor dwords
  __shared__ __attribute__((aligned(16))) unsigned s_desc[2][DW];
  __shared__ int s_ticket;
  // Windows differ in cost (rows and tiles with emission), so a workgroup takes
  // its first window by position and every later one from a shared ticket
  // counter (`count[1]`, zeroed with the list).  Tickets are drawn two windows
  // ahead: one window hides the counter's round trip, the next the descriptor's.
  int* queue = count + 1;
  if ((int)blockIdx.x >= nlive) return;                     // whole workgroup
  if (tid < DW) s_desc[0][tid] = reinterpret_cast<const unsigned*>(&list[blockIdx.x])[tid];
  // Tickets come TICKET_B at a time: the counter is one address, its returning atomics are
  // served one after the other at memory, and a ticket per window kept that queue two thirds busy.
  constexpr int TICKET_B = 4;
  if (tid == 0) s_ticket = (int)gridDim.x + atomicAdd(queue, TICKET_B);
  __syncthreads();
  int nxt = __builtin_amdgcn_readfirstlane(s_ticket);     // block-uniform, kept scalar
  int held = TICKET_B - 1;                                // further windows of the batch after nxt
  // every wave reaches the loop exit: tickets only grow, and a workgroup stops
  // drawing once it holds one past the end of the list
  // Diagnostic build only (-DSLAMHIP_PHASE_STAMPS, `python -m slam_amd.build --stamps`):
  // cycles of wave 0 per phase of the window loop, summed into the spare counter
  // words behind the ticket; the product build has no stamp.
#ifdef SLAMHIP_PHASE_STAMPS
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
#define STAMP_NOW(t_) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP(p_) do { unsigned long long t_; STAMP_NOW(t_); tacc[p_] += t_ - tprev; tprev = t_; } while (0)
  STAMP_NOW(tprev);
#else
#define STAMP(p_) do { } while (0)
#endif
  int par = 0;
  for (int it = blockIdx.x; it < nlive; par ^= 1) {
    // next window's descriptor: in flight during this window, parked in LDS at its end
    unsigned nextw = 0;
    if (tid < DW && nxt < nlive)
      nextw = reinterpret_cast<const unsigned*>(&list[nxt])[tid];
    int fut = nlive;
    if (held == 0 && tid == 0 && nxt < nlive) fut = (int)gridDim.x + atomicAdd(queue, TICKET_B);
    const WinDesc* cur = reinterpret_cast<const WinDesc*>(&s_desc[par][0]);   // in LDS
    // The descriptor's fields are block-uniform and live in SGPRs.  ONE LDS round
    // trip fetches them all: lane l reads dword l of the head (16 dwords: ids and
    // ranges, then the row masks) and, in the same wait, lane l's tile id; the
    // values are then picked lane by lane (v_readlane needs no wait).
    const unsigned hd = s_desc[par][lane & 15];
    const unsigned idv = reinterpret_cast<const unsigned short*>(cur->ids)[lane < WIN_IDS ? lane : 0];
    auto head = [&](int i) { return (int)__builtin_amdgcn_readlane(hd, i); };
    const int e = head(0);
    const int z = head(6);
    const int tb = head(7);
    const int tby = (tb >> 8) & 0xff, tbx = tb & 0xff;
    const int wi = (int)((unsigned)tb >> 16), k = z - wi * g.nv;
    const T* src = iunif + (long long)z * n * n;
    const int x0 = tbx * (TX - 1), y0 = tby * (TY - 1);
    const int wx0 = x0 - ng, wy0 = y0 - ng;                 // window origin (image coords)
    const int ftx0 = max(wx0, 0) / PT_X, fty0 = max(wy0, 0) / PT_Y;
    const int fnx = min(wx0 + vcols - 1, n - 1) / PT_X - ftx0 + 1;
    const int nocc = head(5);
    // Walkers whose regrid folds into the beam (k_prep): the passes below run with the
    // walker's K+1 taps, their outputs ARE the observed pixels of the window's cells,
    // and the regrid phase drops out.  This lane's observed column, asked for now:
    // (k_compact left the pixel of the window's first cell, the direction and the range of
    // cells that have a pixel in the descriptor's range words.)
    const int fxw = head(2), fyw = head(4);
    const bool fold_w = !DIRECT && fxw < 0;                 // block-uniform
    const int fja = (fxw >> 8) & 255, fjb = (fxw >> 16) & 255;
    const int pixx = (fold_w && lane >= fja && lane < fjb) ? head(1) + ((fxw & 1) ? lane : -lane) : -1;

    // -- 0. issue the loads of the occupied input tiles now (NSB per wave in registers).
    // A lane takes one row of a 32x8 tile and two 16-byte column pairs of it (columns
    // pcol, pcol+1 and pcol+16, pcol+17).  Image columns are paired from an even index
    // and rows have an even pitch, so a pair is one aligned 16-byte load; the window's
    // column 0 sits at LDS column `woff` = wx0 & 1, which keeps every pair whole inside
    // the tile (a pair may spill one column into the pad on either side: column -1 is
    // never read, column vcols only against zero taps).
    const int woff = wx0 & 1;
    T* tile = lds + woff;
    const int prow = lane >> 3, pcol = (lane & 7) * 2;
    // position of tile (fyy, fxx) for this lane: LDS offset of its first pair, or -1;
    // bit 0 / 1 of `ok`: first / second pair inside the window
    // Everything that depends on the tile and the window only is scalar; the lane adds its
    // row / column within the tile and tests them against [lo, lo + span) ranges (one
    // subtraction and one unsigned compare each).
    const int lane_gofs = prow * n + pcol, lane_lofs = prow * P + pcol;
    auto tile_pos = [&](int fyy, int fxx, int& gofs, int& ok) {
      const int sgy = (fty0 + fyy) * PT_Y, sgx = (ftx0 + fxx) * PT_X;      // tile origin (image)
      const int srr = sgy - wy0, scc = sgx - wx0;                          // ... in the window
      const int rlo = max(-srr, 0), rspan = max(min(n - sgy, vrows - srr) - rlo, 0);
      const int chi = min(n - sgx, vcols - scc);
      const int c1lo = max(-woff - scc, 0), c1span = max(chi - c1lo, 0);
      const int c2lo = max(-woff - scc - 16, 0), c2span = max(chi - 16 - c2lo, 0);
      const bool rowok = (unsigned)(prow - rlo) < (unsigned)rspan;
      ok = (rowok && (unsigned)(pcol - c1lo) < (unsigned)c1span ? 1 : 0) |
           (rowok && (unsigned)(pcol - c2lo) < (unsigned)c2span ? 2 : 0);
      gofs = sgy * n + sgx + lane_gofs;
      return srr * P + scc + lane_lofs;
    };
    T2 sa[NSB], sb[NSB];
    int sofs[NSB], sok[NSB];
#pragma unroll
    for (int u = 0; u < NSB; ++u) {
      const int bq = wv + CV_NW * u;
      sok[u] = 0;
      sofs[u] = 0;
      sa[u] = sb[u] = Real<T>::make2((T)0, (T)0);
      if (bq < nocc && bq < WIN_IDS) {
        const int q = (int)__builtin_amdgcn_readlane(idv, bq);             // wave-uniform
        int gofs;
        sofs[u] = tile_pos(q >> 8, q & 255, gofs, sok[u]);
        const T2* rp = reinterpret_cast<const T2*>(src + gofs);
        if (sok[u] & 1) sa[u] = rp[0];
        if (sok[u] & 2) sb[u] = rp[8];
      }
    }
    const double* dat = g.data + (long long)k * g.ny * g.nx;
    const int ixa = head(1), ixb = head(2);
    const int jya = head(3), jyb = head(4);
    const int ix = ixa + lane;
    int i1 = -1;
    double t1 = 0.0;
    double dv[NRQ], t0q[NRQ];
    int i0q[NRQ];

    STAMP(0);
    // -- 1. rows that hold emission; zero them in the LDS tile
    auto head64 = [&](int i) {
      return ((unsigned long long)(unsigned)head(i + 1) << 32) | (unsigned)head(i);
    };
    const unsigned long long m0 = head64(8), m1 = head64(10), m2 = head64(12), m3 = head64(14);
    const int nlr = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
    if (tid < rows) {
      const unsigned long long mw = tid < 64 ? m0 : tid < 128 ? m1 : tid < 192 ? m2 : m3;
      if ((mw >> (tid & 63)) & 1ull) {
        int pos = __popcll(mw & ((1ull << (tid & 63)) - 1ull));
        if (tid >= 64) pos += __popcll(m0);
        if (tid >= 128) pos += __popcll(m1);
        if (tid >= 192) pos += __popcll(m2);
        s_rowlist[pos] = (unsigned char)tid;
      }
    }
    if constexpr (PITCH != 0 && MAXR == 2) {
      // Fixed pitch: wave <-> rows wv, wv + 8, ..., a lane <-> one pair of columns.  The row
      // masks shifted by wv once; bit 8j is then row wv + 8j, every row offset an immediate.
      static_assert(TY + 31 <= 128, "two mask words cover the window's rows");
      const unsigned long long z0 = m0 >> wv, z1 = m1 >> wv;
      T* zp = lds + wv * P + lane * 2;
      const bool pair = lane * 2 + 1 < P, last = lane * 2 + 1 == P;
      constexpr int NJ = (TY + 31 + CV_NW - 1) / CV_NW;     // kpad <= 32
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool lit = DIRECT ? wv + CV_NW * j < rows    // direct taps read every row of the window
                       : j < 8 ? ((z0 >> (8 * j)) & 1ull) : ((z1 >> (8 * (j - 8))) & 1ull);   // wave-uniform; rows past the window are never lit
        if (!lit) continue;
        if (pair) { zp[j * CV_NW * P] = 0; zp[j * CV_NW * P + 1] = 0; }
        else if (last) zp[j * CV_NW * P] = 0;
      }
    } else if constexpr (!DIRECT && MAXR == 2) {
      // Any pitch, rows < 128: this wave's rows wv, wv + 8, ... as one bit each (bit j of a
      // byte-strided gather of the two mask words), then only the lit ones are visited.
      const unsigned long long every8 = 0x0101010101010101ull;
      unsigned long long g0 = (m0 >> wv) & every8, g1 = (m1 >> wv) & every8;   // bit 8j: row wv + 8j (+64)
      T* zp = lds + wv * P;
      for (int half = 0; half < 2; ++half) {
        unsigned long long gb = half ? g1 : g0;
        while (gb) {                                       // wave-uniform
          const int j8 = __builtin_ctzll(gb);              // 8j
          gb &= gb - 1;
          T* rowp = zp + (half * 64 + j8) * P;
          for (int c2 = lane * 2; c2 < P; c2 += 128) {
            if (c2 + 1 < P) { rowp[c2] = 0; rowp[c2 + 1] = 0; } else rowp[c2] = 0;
          }
        }
      }
    } else {
      // wave <-> rows, lanes <-> columns; only rows the row pass will read
      for (int rr = wv; rr < rows; rr += CV_NW) {
        const unsigned long long mw = rr < 64 ? m0 : rr < 128 ? m1 : rr < 192 ? m2 : m3;
        if (!direct && !((mw >> (rr & 63)) & 1ull)) continue;   // wave-uniform; direct taps read every row
        T* rowp = lds + rr * P;
        for (int c2 = lane * 2; c2 < P; c2 += 128) {
          if (c2 + 1 < P) { rowp[c2] = 0; rowp[c2 + 1] = 0; } else rowp[c2] = 0;
        }
      }
    }
    lds_barrier();
    STAMP(1);

    // -- 2. staged tiles into LDS
#pragma unroll
    for (int u = 0; u < NSB; ++u) {
      if (wv + CV_NW * u >= nocc) continue;                 // wave-uniform: this slot held no tile
      if (sok[u] & 1) { tile[sofs[u]] = sa[u].x; tile[sofs[u] + 1] = sa[u].y; }
      if (sok[u] & 2) { tile[sofs[u] + 16] = sb[u].x; tile[sofs[u] + 17] = sb[u].y; }
    }
    // folded walkers: the data under this wave's 8 cell rows, asked for only now (the wait above
    // was for the tiles alone) and in flight until the passes are done
    double fdv[R];
    unsigned fvalid = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) fdv[r] = 0.0;
    if (fold_w && wv < TY / R) {
      // this wave's cell rows wv*R .. wv*R + 7 against the window's valid range [ya, yb);
      // the observed row of the first one, then one row pitch up or down per cell row
      const int ya = (fyw >> 8) & 255, yb = (fyw >> 16) & 255;
      const int rlo = max(ya - wv * R, 0), rhi = min(yb - wv * R, R);
      if (rhi > rlo) fvalid = ((1u << rhi) - 1u) & ~((1u << rlo) - 1u);
      const long long step = (fyw & 1) ? (long long)g.nx : -(long long)g.nx;
      const double* rowp = dat + (long long)(head(3) + ((fyw & 1) ? wv * R : -(wv * R))) * g.nx;
      if (pixx >= 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((fvalid >> r) & 1u) fdv[r] = rowp[pixx];      // wave-uniform test
          rowp += step;
        }
      }
    }
    if (nocc > NSB * CV_NW) {
      // crowded window: the remaining occupied tiles, loaded and stored directly
      const unsigned long long* occ = cur->occ;
      int seen = 0;
      for (int qw = 0; qw < FLAG_MAX / 64; ++qw) {
        unsigned long long m = occ[qw];
        while (m) {
          const int q = qw * 64 + __builtin_ctzll(m);
          m &= m - 1;
          const int b = seen++;
          if (b < NSB * CV_NW || (b & (CV_NW - 1)) != wv) continue;
          const int fyy = q / fnx;
          int gofs, ok;
          const int o = tile_pos(fyy, q - fyy * fnx, gofs, ok);
          const T2* rp = reinterpret_cast<const T2*>(src + gofs);
          if (ok & 1) { const T2 v = rp[0]; tile[o] = v.x; tile[o + 1] = v.y; }
          if (ok & 2) { const T2 v = rp[8]; tile[o + 16] = v.x; tile[o + 17] = v.y; }
        }
      }
    }
    lds_barrier();
    STAMP(2);

    // -- 3+4. per beam term: row pass over the rows with emission
    //         tmp[row][x] = sum_u gxf[u] tile[row][x+u]   (wave <-> R output columns,
    //         lanes <-> rows; odd pitch: conflict-free), then column pass
    //         out[y][x] += sum_u gyf[u] tmp[y+u][x]        (lanes <-> x, waves <-> y blocks)
    T acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0;
    // direct taps: wave <-> (block of RB output rows, half of the tap columns);
    // the two halves are added through LDS afterwards
    constexpr int RB = TY == 64 ? 2 * R : R;
    static_assert(!DIRECT || TY / RB == CV_NW / 2, "direct taps: four row blocks, two tap-column halves");
    const int rb = wv % (TY / RB), tapset = wv / (TY / RB);
    T dacc[DIRECT ? RB : 1];
    if (DIRECT) {
#pragma unroll
      for (int r = 0; r < RB; ++r) dacc[r] = 0;
      // out[y][x] = sum_{a,b} beam[K-1-a][K-1-b] tile[y+a][x+b]: per tap column b a
      // sliding pass along y (lanes <-> x; RB+R-1 tile reads and R tap reads per
      // RB*R multiply-adds); tap blocks over rows without emission are skipped
      for (int b = tapset; b < K; b += 2) {
        const T* __restrict__ wb = gbt + b * kpad;     // block-uniform address: scalar loads
        const T* col = tile + (rb * RB) * P + lane + b;
        for (int u0 = 0; u0 < kpad; u0 += R) {
          const int ra = rb * RB + u0;
          unsigned long long bits = 0ull;
          {
            const int w0 = ra >> 6, sh = ra & 63;
            const unsigned long long lo = w0 == 0 ? m0 : w0 == 1 ? m1 : w0 == 2 ? m2 : m3;
            const unsigned long long hi = w0 == 0 ? m1 : w0 == 1 ? m2 : w0 == 2 ? m3 : 0ull;
            bits = (lo >> sh) | (sh ? (hi << (64 - sh)) : 0ull);
          }
          const bool half = tail4 && u0 == kpad - R;     // block-uniform: last tap rows are padding
          if (!(bits & ((1ull << (half ? RB + R / 2 - 1 : RB + R - 1)) - 1ull))) continue;
          // three row pointers the compiler cannot relate: rows 3 apart lie more
          // than a ds_read2_b64's reach apart, so the reads stay single
          // ds_read_b64 (2 LDS cycles each; the paired form costs 8)
          int o1 = P, o2 = 2 * P;
          asm volatile("" : "+v"(o1));
          asm volatile("" : "+v"(o2));
          const T* c0 = col + u0 * P;
          const T* c1 = c0 + o1;
          const T* c2 = c0 + o2;
          T win[RB + R - 1];
#pragma unroll
          for (int r = 0; r < RB + R / 2 - 1; ++r)
            win[r] = (r % 3 == 0 ? c0 : r % 3 == 1 ? c1 : c2)[(r / 3) * 3 * P];
#pragma unroll
          for (int qq = 0; qq < R / 2; ++qq) {
            const T wgt = wb[u0 + qq];
#pragma unroll
            for (int r = 0; r < RB; ++r) dacc[r] = fma(wgt, win[r + qq], dacc[r]);
          }
          if (half) continue;
#pragma unroll
          for (int r = RB + R / 2 - 1; r < RB + R - 1; ++r)
            win[r] = (r % 3 == 0 ? c0 : r % 3 == 1 ? c1 : c2)[(r / 3) * 3 * P];
#pragma unroll
          for (int qq = R / 2; qq < R; ++qq) {
            const T wgt = wb[u0 + qq];
#pragma unroll
            for (int r = 0; r < RB; ++r) dacc[r] = fma(wgt, win[r + qq], dacc[r]);
          }
        }
      }
    }
    for (int rk = 0; !DIRECT && rk < (MULTI ? nrank : 1); ++rk) {
      // taps through the scalar cache (block-uniform addresses): the LDS is busy enough
      const T* __restrict__ wxr = fold_w ? gxw + (long long)wi * kpad : gxp + rk * kpad;
      const T* __restrict__ wyr = fold_w ? gyw + (long long)wi * kpad : gyp + rk * kpad;
      // A group of up to 64 listed rows per step.  A lane takes one row and, of the wave's
      // R output columns, all R when the group holds more than 32 rows, half of them when
      // it holds 17..32 (lanes l and l + 32 share a row), a quarter when it holds fewer:
      // a thin group costs a half or a quarter of a full one instead of idling its lanes.
      T res[MAXR][R];
      // rows of group q, and the lanes-per-row shift (0 / 1 / 2); both block-uniform
      auto group = [&](int q, int& m, int& sh) {
        m = min(nlr - q * 64, 64);
        sh = m > 32 ? 0 : m > 16 ? 1 : 2;
      };
#pragma unroll
      for (int q = 0; q < MAXR; ++q) {
#pragma unroll
        for (int r = 0; r < R; ++r) res[q][r] = 0;
        int m, sh;
        group(q, m, sh);
        if (m <= 0) continue;
        const int l = lane & (63 >> sh);
        if (l < m) {
          const T* row = tile + s_rowlist[q * 64 + l] * P + wv * R + (lane >> (6 - sh)) * (R >> sh);
          if (sh == 0) row_taps<R, R>(row, wxr, kpad, tail4, res[q]);
          else if (sh == 1) row_taps<R / 2, R>(row, wxr, kpad, tail4, res[q]);
          else row_taps<R / 4, R>(row, wxr, kpad, tail4, res[q]);
        }
      }
      STAMP(3);
      lds_barrier();                  // tile reads (overlay) / previous column pass done
      // rows without emission convolve to zero; the others come from registers
#pragma unroll
      for (int q = 0; q < MAXR; ++q) {
        const int rr = q * 64 + lane;
        if (rr < rows && rk == 0) {
          const unsigned long long mw = q == 0 ? m0 : q == 1 ? m1 : q == 2 ? m2 : m3;
          if (!((mw >> lane) & 1ull)) {
#pragma unroll
            for (int r = 0; r < R; ++r) tmp[rr * P2 + wv * R + r] = 0;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < MAXR; ++q) {
        int m, sh;
        group(q, m, sh);
        if (m <= 0) continue;
        const int l = lane & (63 >> sh);
        if (l < m) {
          T* o = tmp + s_rowlist[q * 64 + l] * P2 + wv * R + (lane >> (6 - sh)) * (R >> sh);
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (r < (R >> sh)) o[r] = res[q][r];
        }
      }
      lds_barrier();
      STAMP(4);

      if (wv < TY / R) {
        const T* col = tmp + (wv * R) * P2 + lane;
        // fixed instance: the row masks from this wave's first row on, shifted once per window
        // (rows wv*R .. wv*R + 63 cover every tap block of a 32-tap beam)
        unsigned long long mcol = 0ull, mcol_hi = 0ull;
        if constexpr (MAXR == 2) {
          mcol = wv ? (m0 >> (wv * R)) | (m1 << (64 - wv * R)) : m0;
          mcol_hi = m1 >> (wv * R);
        }
        for (int u0 = 0; u0 < kpad; u0 += R) {
          // rows wv*R+u0 .. +2R-2 feed this tap block: skip it if none holds emission
          const int ra = wv * R + u0;
          unsigned long long bits = 0ull;
          if constexpr (PITCH != 0 && MAXR == 2) {
            bits = mcol >> u0;
          } else if constexpr (MAXR == 2) {
            // any tap count, rows < 128: the 128-bit mask slides down one tap block per trip
            bits = mcol;
            mcol = (mcol >> R) | (mcol_hi << (64 - R));
            mcol_hi >>= R;
          } else {
            const int w0 = ra >> 6, sh = ra & 63;
            const unsigned long long lo = w0 == 0 ? m0 : w0 == 1 ? m1 : w0 == 2 ? m2 : m3;
            const unsigned long long hi = w0 == 0 ? m1 : w0 == 1 ? m2 : w0 == 2 ? m3 : 0ull;
            bits = (lo >> sh) | (sh ? (hi << (64 - sh)) : 0ull);
          }
          if (tail4 && u0 == kpad - R) {
            // half block: R / 2 taps over the R + R / 2 - 1 rows they reach
            if (!(bits & ((1ull << (R + R / 2 - 1)) - 1ull))) continue;
            T win[R + R / 2 - 1];
#pragma unroll
            for (int r = 0; r < R + R / 2 - 1; ++r) win[r] = col[(u0 + r) * P2];
#pragma unroll
            for (int qq = 0; qq < R / 2; ++qq) {
              const T wgt = wyr[u0 + qq];
#pragma unroll
              for (int r = 0; r < R; ++r) acc[r] = fma(wgt, win[r + qq], acc[r]);
            }
            continue;
          }
          if (!(bits & ((1ull << (2 * R - 1)) - 1ull))) continue;
          T win[2 * R - 1];
#pragma unroll
          for (int r = 0; r < 2 * R - 1; ++r) win[r] = col[(u0 + r) * P2];
#pragma unroll
          for (int qq = 0; qq < R; ++qq) {
            const T wgt = wyr[u0 + qq];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fma(wgt, win[r + qq], acc[r]);
          }
        }
      }
    }
    STAMP(5);
    double gf = 0.0, ff = 0.0;
    if (fold_w) {
      // the column pass left cell (row wv*8 + r, column lane) of the regridded model in
      // acc[r]: straight into sum(I*d), sum(I*I) (cells without an observed pixel add nothing)
      if (wv < TY / R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double a = (pixx >= 0 && ((fvalid >> r) & 1u)) ? (double)acc[r] : 0.0;
          gf = fma(a, fdv[r], gf);
          ff = fma(a, a, ff);
        }
      }
      STAMP(6);
    } else {
      // what the regrid needs from HBM, requested only now: the passes above want the registers
      if (ix < ixb) {
        i1 = cellx[(long long)wi * g.nx + ix] - x0;
        t1 = fracx[(long long)wi * g.nx + ix];
      }
  #pragma unroll
      for (int q = 0; q < NRQ; ++q) {
        const int jy = jya + wv + CV_NW * q;
        const bool rowin = jy < jyb;
        i0q[q] = rowin ? celly[(long long)wi * g.ny + jy] - y0 : -1;   // wave-uniform
        t0q[q] = rowin ? fracy[(long long)wi * g.ny + jy] : 0.0;
        dv[q] = (ix < ixb && rowin) ? dat[(long long)jy * g.nx + ix] : 0.0;
      }
      const bool xown = i1 >= 0 && i1 < TX - 1;               // else owned by another window
      lds_barrier();                                        // column pass done reading LDS
      if (DIRECT) {
        T* half = lds + TY * PC;                         // the odd tap columns' sums
        if (tapset == 1) {
  #pragma unroll
          for (int r = 0; r < RB; ++r) half[(rb * RB + r) * PC + lane] = dacc[r];
        }
        lds_barrier();
        if (tapset == 0) {
  #pragma unroll
          for (int r = 0; r < RB; ++r)
            lds[(rb * RB + r) * PC + lane] = dacc[r] + half[(rb * RB + r) * PC + lane];
        }
      } else if (wv < TY / R) {
  #pragma unroll
        for (int r = 0; r < R; ++r) lds[(wv * R + r) * PC + lane] = acc[r];
      }
      lds_barrier();

      STAMP(6);
      // -- 5. regrid + chi-square sums from the LDS tile
  #pragma unroll
      for (int q = 0; q < NRQ; ++q) {
        const int jy = jya + wv + CV_NW * q;
        if (jy < jyb) {
          const int i0 = i0q[q];                              // wave-uniform
          if (xown && i0 >= 0 && i0 < TY - 1) {
            const double t0 = t0q[q];
            const T* c = lds + i0 * PC + i1;
            const double v00 = c[0], v01 = c[1], v10 = c[PC], v11 = c[PC + 1];
            const double val = v00 * (1 - t0) * (1 - t1) + v01 * (1 - t0) * t1 +
                               v10 * t0 * (1 - t1) + v11 * t0 * t1;
            gf = fma(val, dv[q], gf);
            ff = fma(val, val, ff);
          }
        }
      }
      // (only if the observed grid is finer than the model grid) further columns / rows
      for (int ix2 = ix; ix2 < ixb; ix2 += 64) {
        const int i1b = cellx[(long long)wi * g.nx + ix2] - x0;
        if (i1b < 0 || i1b >= TX - 1) continue;
        const double t1b = fracx[(long long)wi * g.nx + ix2];
        for (int jy = jya + wv + (ix2 == ix ? CV_NW * NRQ : 0); jy < jyb; jy += CV_NW) {
          const int i0 = celly[(long long)wi * g.ny + jy] - y0;
          if (i0 < 0 || i0 >= TY - 1) continue;
          const double t0 = fracy[(long long)wi * g.ny + jy];
          const T* c = lds + i0 * PC + i1b;
          const double val = (double)c[0] * (1 - t0) * (1 - t1b) + (double)c[1] * (1 - t0) * t1b +
                             (double)c[PC] * t0 * (1 - t1b) + (double)c[PC + 1] * t0 * t1b;
          gf = fma(val, dat[(long long)jy * g.nx + ix2], gf);
          ff = fma(val, val, ff);
        }
      }
    }
    gf = wave_sum_dpp(gf);
    ff = wave_sum_dpp(ff);
    if (lane == 0) { sg[wv] = gf; sf[wv] = ff; }
    if (tid < DW) s_desc[par ^ 1][tid] = nextw;
    if (tid == 0) s_ticket = fut;
    lds_barrier();
    if (tid == 0) {
      double a = 0.0, b2 = 0.0;
#pragma unroll
      for (int q = 0; q < CV_NW; ++q) { a += sg[q]; b2 += sf[q]; }
      partial[3 * (long long)e] = a;
      partial[3 * (long long)e + 1] = b2;
      partial[3 * (long long)e + 2] = 1.0;                  // window was convolved
    }
    STAMP(7);
    it = nxt;
    if (held > 0) { nxt = nxt + 1; --held; }              // still inside the batch
    else { nxt = __builtin_amdgcn_readfirstlane(s_ticket); held = TICKET_B - 1; }
    // the next window's first barrier orders its LDS writes after these reads
  }
#ifdef SLAMHIP_PHASE_STAMPS
  if (tid == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(count + 32);
    for (int p_ = 0; p_ < 8; ++p_) atomicAdd(&dbg[p_], tacc[p_]);
  }
#endif
#undef STAMP
}

// One round of k_conv_rows' row pass: `cnt` (<= 64 / (64 / C) ...) listed rows, 64 / C lanes per row,
// C outputs per lane.  Groups of four adjacent lanes take four adjacent column blocks of a row, the
// next four lanes the next row.
template <int C, typename T>
__device__ __forceinline__ void row_round(T* tile, const unsigned char* rowlist, int cnt, int lane,
                                          const T* __restrict__ wx, int kp, bool tail4, int P) {
  constexpr int SLOTS = C;                                  // rows per round: 64 lanes / (64 / C lanes per row)
  constexpr int LG = C == 8 ? 3 : C == 4 ? 2 : 1;
  static_assert(C == 8 || C == 4 || C == 2, "lane layout");
  const int slot = (lane >> 2) & (SLOTS - 1);
  const int cb = ((lane >> (2 + LG)) << 2) | (lane & 3);
  if (slot < cnt) {
    T res[C];
#pragma unroll
    for (int r = 0; r < C; ++r) res[r] = 0;
    T* o = tile + rowlist[slot] * P + cb * C;
    row_taps<C, CONV_R>(o, wx, kp, tail4, res);
#pragma unroll
    for (int r = 0; r < C; ++r) o[r] = res[r];
  }
}

// k_conv_rows: k_conv_chi2's one-term cases -- K + 1 <= 32 taps on 64-row windows <TY 64, KP 32>,
// 80 padded taps of a K <= 77 beam on 48-row windows <48, 80>, double or float, and <OBS> for the
// 'mom0*' scalings' beam on the observed grid -- with the window's ROWS OWNED BY WAVES.  k_conv_chi2 walks a window in phases that every wave must finish before
// any wave may go on (zero the tile | stage the tiles | row pass | write the intermediate | column
// pass | sums: five barriers), and the counters show its waves waiting, not computing.  Here a wave
// owns groups of four image rows (group q of the window belongs to wave (q + q/8) mod 8, so that
// the rows it handles together differ mod 8 and its LDS reads do not collide): it zeroes THEM,
// asks for the halves of the occupied tiles that lie on THEM (a quarter wave per row, 16 bytes per
// lane), stores them, runs the row pass over those of them that hold emission (a lane = one row x
// one block of 8 output columns) and writes the results IN PLACE over the rows it has just read --
// nothing any other wave touches, so none of this needs a barrier.  One barrier before the column
// pass (which reads every wave's rows) and one after the sums are left.  Same arithmetic, same
// order of operations per output as k_conv_chi2's fixed-pitch instance: chi2 is bitwise the same.
template <typename T, bool REGRID, int TY = 64, int KP = 32, bool OBS = false>
__global__ void __launch_bounds__(CV_NT, KP > 32 ? (sizeof(T) == 4 ? 4 : 2) : sizeof(T) == 4 ? CV_MINW_F32 : 4)   // the 80-tap tile fills a CU's LDS once: 2 waves per SIMD; the float 32-tap one fits three times
k_conv_rows(FitK g, const T* __restrict__ iunif,
            const int* __restrict__ cellx, const double* __restrict__ fracx,
            const int* __restrict__ celly, const double* __restrict__ fracy,
            const WinDesc* __restrict__ list, int* __restrict__ count, int wnx, int wny,
            double* __restrict__ partial,
            const T* __restrict__ gxp, const T* __restrict__ gyp,
            const T* __restrict__ gxw, const T* __restrict__ gyw, int tail4_arg, int rev_total,
            const unsigned char* __restrict__ rowflag = nullptr) {
  // OBS (the 'mom0*' scalings' batch path): `iunif` is the regridded, moment-scaled cube on the
  // observed grid itself (g.nneed = nx in the FitK copy this launch gets, g.ny its rows), windows tile it
  // without overlap, plain beam taps, and a window leaves sum (data - model)^2 over its pixels
  // (NaN terms dropped, :601) in partial[e].  k_regrid_mom0 stored only the 64-pixel row
  // segments it marked in `rowflag`: what a staged tile holds outside them is stale and is
  // not written to the LDS tile.
  static_assert(!OBS || !REGRID, "the observed grid needs no regrid");
  // REGRID = false: every window of `list` belongs to a walker whose regrid is folded into the
  // beam taps (k_compact's first list): no conv tile, no bilinear phase, none of its state.
  // REGRID = true: the general form, run on k_compact's second list (the last *count entries
  // of the rev_total-entry array).
  extern __shared__ unsigned char lds_raw[];
  T* const lds = reinterpret_cast<T*>(lds_raw);
  using T2 = typename Real<T>::vec2;
  // TY x KP: 64 x 32 (K + 1 <= 32 taps, pitch 97, two workgroups per CU) or 48 x 80 (K <= 77:
  // BASELINE config 4's beam; pitch 145, one workgroup per CU)
  constexpr int TX = CONV_TX, R = CONV_R, PC = TX + 1, P = (TX + KP) | 1;
  constexpr int RO = TY / CV_NW;                            // output rows per wave in the column pass
  constexpr int ROWS = TY + KP - 1;                         // 95 / 127 window rows incl. halo and tap padding
  constexpr int NRQ = (TY - 1 + CV_NW - 1) / CV_NW;         // regrid rows per wave
  constexpr int NG = 4;                                     // row groups a wave can own (25 / 32 groups, 8 waves)
  constexpr int NFX = (TX + KP - 3 + PT_X - 1) / PT_X + 1;  // tile columns a window's rows can span (4 / 6)
  constexpr bool OCC2 = ((ROWS + PT_Y - 1) / PT_Y + 1) * NFX > 64;   // occupancy bitmap needs a second word
  static_assert(P == SEP_PITCH || KP != 32, "the pitch fz_tile_bytes gives");
  static_assert(P > TX + KP - 1 + 1, "room for the column shift");
  static_assert(RO * CV_NW == TY && RO <= R, "column pass: every wave RO rows");
  static_assert(NG * NFX <= 32 && ROWS <= 128, "masks");
  // rows with image data (r < TY + K - 1) must lie in the 32 owned groups for any alignment a <= 3
  // (slamhip_create checks K); rows beyond them up to ROWS are zeroed by the last wave
  __shared__ unsigned char s_rowlist[CV_NW][16];
  __shared__ double sg[CV_NW], sf[CV_NW];
  const bool tail4 = tail4_arg != 0;
  const int K = g.kbeam, ng = g.ngauss, n = g.nneed;
  const int nrow = OBS ? g.ny : n;                          // image rows (n = its columns and row pitch)
  const int vrows = TY + K - 1, vcols = TX + K - 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nlive = *count;
  if (REGRID) list += rev_total - nlive;
  constexpr int DW = sizeof(WinDesc) / 4;
  __shared__ __attribute__((aligned(16))) unsigned s_desc[2][DW];
  __shared__ int s_ticket;
  // XCD-LOCAL TICKETS.  Workgroups go to the eight XCDs round robin (blockIdx.x & 7), each XCD
  // has its own L2, and neighbouring windows of the list (same plane, same walker) share halo
  // tiles: XCD x walks the x-th eighth of the list with a ticket counter of its own (count[64 +
  // 32 x]), so that a plane's windows meet in one L2; a workgroup whose XCD's eighth is used up
  // goes on with the next XCD's.  Tickets come TICKET_B at a time (the counter is one address).
  __shared__ int s_tend;
  constexpr int TICKET_B = 4;
  const int nx8 = (nlive + 7) >> 3;
  const int myx = blockIdx.x & 7, lb = blockIdx.x >> 3;
  int vx = 0;                                               // thread 0: eighths given up so far
  auto draw = [&](int& end) -> int {                        // thread 0: next batch [return, end)
    while (vx < 8) {
      const int x = (myx + vx) & 7;
      const int lo = min(x * nx8, nlive), hi = min((x + 1) * nx8, nlive);
      const int first = min(((int)gridDim.x + 7 - x) >> 3, hi - lo);   // taken by position
      if (lo + first < hi) {
        const int idx = lo + first + atomicAdd(count + 64 + 32 * x, TICKET_B);
        if (idx < hi) { end = min(idx + TICKET_B, hi); return idx; }
      }
      ++vx;
    }
    end = nlive;
    return nlive;
  };
  {
    // first window by position in the own eighth, else by ticket; and the first batch
    const int lo = min(myx * nx8, nlive), hi = min((myx + 1) * nx8, nlive);
    if (tid == 0) {
      int e0 = 0, e1 = 0;
      int first = lo + lb < hi ? lo + lb : -1;
      int t0 = draw(e0);
      if (first < 0) {                                       // no window by position: the batch's first
        first = t0;
        t0 = t0 + 1 < e0 ? t0 + 1 : draw(e1);
        if (!(first + 1 < e0)) e0 = e1;
      }
      s_desc[1][0] = (unsigned)first;
      s_ticket = t0;
      s_tend = e0;
    }
  }
  __syncthreads();
  int it0 = (int)__builtin_amdgcn_readfirstlane(s_desc[1][0]);
  if (it0 >= nlive) return;                                 // whole workgroup: nothing left anywhere
  int nxt = __builtin_amdgcn_readfirstlane(s_ticket);
  int nend = __builtin_amdgcn_readfirstlane(s_tend);
  __syncthreads();
  if (tid < DW) s_desc[0][tid] = reinterpret_cast<const unsigned*>(&list[it0])[tid];
  __syncthreads();
  int par = 0;
  // (diagnostic build: cycles of wave 0 per phase, as in k_conv_chi2; the first list's launch only)
#ifdef SLAMHIP_PHASE_STAMPS
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
#define STAMP_NOW(t_) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP(p_) do { unsigned long long t_; STAMP_NOW(t_); tacc[p_] += t_ - tprev; tprev = t_; } while (0)
  STAMP_NOW(tprev);
#else
#define STAMP(p_) do { } while (0)
#endif
  // this lane in the staging (a quarter wave per row of a group, 16 bytes per lane) and in the
  // row pass (half waves take column blocks 0-3 / 4-7, eight row slots x four blocks each)
  const int srow = lane >> 4, scol = (lane & 15) * 2;
  for (int it = it0; it < nlive; par ^= 1) {
    unsigned nextw = 0;
    if (tid < DW && nxt < nlive)
      nextw = reinterpret_cast<const unsigned*>(&list[nxt])[tid];
    int fut = nlive, fend = nlive;
    if (!(nxt + 1 < nend) && tid == 0 && nxt < nlive) fut = draw(fend);
    const unsigned hd = s_desc[par][lane & 15];
    const unsigned ocv = s_desc[par][44 + (lane & 3)];      // occupancy bitmap (<= 52 tiles at 64 x 32, <= 102 at 48 x 80)
    static_assert(offsetof(WinDesc, occ) == 44 * 4, "descriptor layout");
    auto head = [&](int i) { return (int)__builtin_amdgcn_readlane(hd, i); };
    const int e = head(0);
    const int z = head(6);
    const int tb = head(7);
    const int tby = (tb >> 8) & 0xff, tbx = tb & 0xff;
    const int wi = (int)((unsigned)tb >> 16), k = z - wi * g.nv;
    const T* src = iunif + (long long)z * nrow * n;
    const int x0 = tbx * (OBS ? TX : TX - 1), y0 = tby * (OBS ? TY : TY - 1);
    const int wx0 = x0 - ng, wy0 = y0 - ng;
    const int ftx0 = max(wx0, 0) / PT_X, fty0 = max(wy0, 0) / PT_Y;
    const int fnx = min(wx0 + vcols - 1, n - 1) / PT_X - ftx0 + 1;
    const unsigned long long occ = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ocv, 1) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane(ocv, 0);
    unsigned long long occ_hi = 0ull;
    if constexpr (OCC2)
      occ_hi = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ocv, 3) << 32) |
               (unsigned)__builtin_amdgcn_readlane(ocv, 2);
    const int fxw = head(2), fyw = head(4);
    const bool fold_w = !REGRID || fxw < 0;                 // block-uniform
    const int fja = (fxw >> 8) & 255, fjb = (fxw >> 16) & 255;
    const int pixx = OBS ? (x0 + lane < g.nx ? x0 + lane : -1)
                         : (fold_w && lane >= fja && lane < fjb) ? head(1) + ((fxw & 1) ? lane : -lane) : -1;
    const int woff = wx0 & 1;
    T* tile = lds + woff;
    auto head64 = [&](int i) {
      return ((unsigned long long)(unsigned)head(i + 1) << 32) | (unsigned)head(i);
    };
    const unsigned long long m0 = head64(8), m1 = head64(10);

    // -- 0. this wave's row groups: group q holds window rows 4q - a .. 4q - a + 3, image rows
    //       4 (mq + q) ..; its tiles are the half `Q & 1` of tile row Q >> 1
    const int a = wy0 & 3, mq = wy0 >> 2;                   // floor semantics for negative wy0 too
    // (the tests are split into what depends on the tile column -- per lane, once per window --,
    // on the group -- wave-uniform -- and on the lane's row: the 16 / 24 slots then cost a scalar
    // bit test each; written slot by slot the set-up was a fifth of the window's time)
    unsigned colm = 0u;                                     // bit fx: this lane's pair of tile column fx lies in the window
#pragma unroll
    for (int fx = 0; fx < NFX; ++fx) {
      const int icol = (ftx0 + fx) * PT_X + scol, cc = icol - wx0;
      if (fx < fnx && icol < n && cc >= -woff && cc < vcols) colm |= 1u << fx;
    }
    T2 sv[NG][NFX];
    unsigned sok = 0u;                                      // bit NFX gi + fx: this lane holds a pair there
    unsigned uocc = 0u;                                     // wave-uniform, same bits: the tile half holds emission
    unsigned m16 = 0u;                                      // bit 4 gi + i: that owned row holds emission
    unsigned segm = 0u;                                     // OBS: bit 4 gi + j: segment seg0 + j of this lane's row was stored
    const int nseg = (g.nx + 63) >> 6, seg0 = max(wx0, 0) >> 6;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int q = 8 * gi + ((wv - gi) & 7);
      const int rb = 4 * q - a;                             // first window row of the group
      const int Q = mq + q;
      const int fy = (Q >> 1) - fty0;
      // four mask bits from window row rb on (rows < 0 are not lit)
      {
        const int sh = rb < 0 ? 0 : rb;
        const unsigned long long lo = sh < 64 ? m0 : m1, hi = sh < 64 ? m1 : 0ull;
        const int s6 = sh & 63;
        unsigned long long bits = (lo >> s6) | (s6 ? (hi << (64 - s6)) : 0ull);
        if (rb < 0) bits <<= -rb;
        if (rb < ROWS) m16 |= (unsigned)(bits & 15ull) << (4 * gi);
      }
      unsigned rowocc = 0u;                                 // wave-uniform: the group's tile halves with emission
      if (rb < vrows && rb + 4 > 0 && Q >= 0 && fy >= 0) {
        const int sh = fy * fnx;
        unsigned long long w;
        if constexpr (OCC2) w = sh < 64 ? (occ >> sh) | (sh ? occ_hi << (64 - sh) : 0ull) : occ_hi >> (sh - 64);
        else w = occ >> sh;
        rowocc = (unsigned)w & ((1u << fnx) - 1u);
      }
      uocc |= rowocc << (NFX * gi);
      const int r = rb + srow, irow = 4 * Q + srow;
      const unsigned lanem = (r >= 0 && r < vrows && irow < nrow) ? (colm & rowocc) : 0u;
      sok |= lanem << (NFX * gi);
      const T* rp = src + ((long long)irow * n + (ftx0 * PT_X + scol));
      if constexpr (OBS) {
        if (rowocc && lanem) {
          const unsigned char* rf = rowflag + ((long long)z * nrow + irow) * nseg + seg0;
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (seg0 + j < nseg && rf[j]) segm |= 1u << (4 * gi + j);
        }
      }
#pragma unroll
      for (int fx = 0; fx < NFX; ++fx)
        sv[gi][fx] = Real<T>::make2((T)0, (T)0);         // (left uninitialised the compiler's code is 35 % slower)
      if (rowocc) {                                         // wave-uniform: groups without emission cost one branch
#pragma unroll
        for (int fx = 0; fx < NFX; ++fx)
          if ((lanem >> fx) & 1u) sv[gi][fx] = *reinterpret_cast<const T2*>(rp + fx * PT_X);
      }
    }
    const double* dat = g.data + (long long)k * g.ny * g.nx;
    STAMP(0);

    // -- 1. zero the owned rows (all 95: the column pass reads the padded ones against zero taps;
    //       rows without emission only where the column pass reads), list those with emission
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int rb = 4 * (8 * gi + ((wv - gi) & 7)) - a;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = rb + i;
        if (r < 0 || r >= ROWS) continue;                   // wave-uniform
        T* rowp = lds + r * P;
        if ((m16 >> (4 * gi + i)) & 1u) {                   // wave-uniform: the row pass reads all of it
#pragma unroll
          for (int c0 = 0; c0 < P; c0 += 64)
            if (c0 + 64 <= P || lane < P - c0) rowp[c0 + lane] = 0;
        } else {
          rowp[woff + lane] = 0;                            // only the column pass reads it: 64 columns
        }
      }
    }
    if constexpr (4 * 8 * NG - 3 < ROWS) {                  // rows behind the last group (never lit)
      if (wv == CV_NW - 1)
        for (int r = 4 * 8 * NG - a; r < ROWS; ++r) lds[r * P + woff + lane] = 0;
    }
    const int nown = __popc(m16);
    if (lane < 16 && ((m16 >> lane) & 1u)) {
      const int gi = lane >> 2;
      s_rowlist[wv][__popc(m16 & ((1u << lane) - 1u))] =
          (unsigned char)(4 * (8 * gi + ((wv - gi) & 7)) - a + (lane & 3));
    }

    STAMP(1);
    // -- 2. the tile halves on the owned rows
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int rb = 4 * (8 * gi + ((wv - gi) & 7)) - a;
      T* ob = tile + (rb + srow) * P + (ftx0 * PT_X + scol - wx0);
      if ((uocc >> (NFX * gi)) & ((1u << NFX) - 1u)) {      // wave-uniform
#pragma unroll
        for (int fx = 0; fx < NFX; ++fx) {
          bool wr = (sok >> (NFX * gi + fx)) & 1u;
          if constexpr (OBS) wr = wr && ((segm >> (4 * gi + ((ftx0 + fx) >> 1) - seg0)) & 1u);
          if (wr) {
            ob[fx * PT_X] = sv[gi][fx].x;
            ob[fx * PT_X + 1] = sv[gi][fx].y;
          }
        }
      }
    }
    double fdv[RO];
    unsigned fvalid = 0u;
#pragma unroll
    for (int r = 0; r < RO; ++r) fdv[r] = 0.0;
    if (fold_w) {
      const int ya = OBS ? 0 : (fyw >> 8) & 255, yb = OBS ? min(TY, g.ny - y0) : (fyw >> 16) & 255;
      const int rlo = max(ya - wv * RO, 0), rhi = min(yb - wv * RO, RO);
      if (rhi > rlo) fvalid = ((1u << rhi) - 1u) & ~((1u << rlo) - 1u);
      const long long step = (OBS || (fyw & 1)) ? (long long)g.nx : -(long long)g.nx;
      const double* rowp = dat + (long long)(OBS ? y0 + wv * RO : head(3) + ((fyw & 1) ? wv * RO : -(wv * RO))) * g.nx;
      if (pixx >= 0) {
#pragma unroll
        for (int r = 0; r < RO; ++r) {
          if ((fvalid >> r) & 1u) fdv[r] = rowp[pixx];      // wave-uniform test
          rowp += step;
        }
      }
    }

    STAMP(2);
    // -- 3. row pass over the owned rows with emission, results in place (this wave's rows only)
    const T* __restrict__ wxr = (fold_w && !OBS) ? gxw + (long long)wi * KP : gxp;
    const T* __restrict__ wyr = (fold_w && !OBS) ? gyw + (long long)wi * KP : gyp;
    // A round takes up to 8 listed rows (8 lanes per row, R outputs per lane) and writes its results
    // back in place at once: its lanes have all read their rows by then, and the next round reads
    // other rows.  A remainder of 3-4 rows runs 16 lanes per row with R / 2 outputs each, one of 1-2
    // rows 32 lanes per row: half / a quarter of a round's multiply-adds instead of idle lanes (the
    // waves meet at the barrier below: the one with 9-12 rows sets the pace).
    for (int done = 0; done < nown;) {                      // wave-uniform
      const int rem = nown - done;
      if (rem > R / 2) { row_round<R>(tile, s_rowlist[wv] + done, min(rem, 8), lane, wxr, KP, tail4, P); done += 8; }
      else if (rem > R / 4) { row_round<R / 2>(tile, s_rowlist[wv] + done, rem, lane, wxr, KP, tail4, P); done += 4; }
      else { row_round<R / 4>(tile, s_rowlist[wv] + done, rem, lane, wxr, KP, tail4, P); done += 2; }
    }
    STAMP(3);
    lds_barrier();                                          // every wave's rows hold the intermediate
    STAMP(4);

    // -- 4. column pass: out[y][x] = sum_u gyf[u] tmp[y + u][x]  (lanes <-> x, waves <-> RO rows,
    //       tap blocks of R)
    T acc[RO];
#pragma unroll
    for (int r = 0; r < RO; ++r) acc[r] = 0;
    {
      const T* col = tile + (wv * RO) * P + lane;
      // the row masks from this wave's first row on; 64 bits reach every tap block of the 32-tap
      // tile, the 80-tap one slides a 128-bit mask down one tap block per trip
      unsigned long long mcol = wv ? (m0 >> (wv * RO)) | (m1 << (64 - wv * RO)) : m0;
      unsigned long long mcol_hi = m1 >> (wv * RO);
      for (int u0 = 0; u0 < KP; u0 += R) {
        unsigned long long bits;
        if constexpr (KP + RO - 1 <= 64) {
          bits = mcol >> u0;
        } else {
          bits = mcol;
          mcol = (mcol >> R) | (mcol_hi << (64 - R));
          mcol_hi >>= R;
        }
        if (tail4 && u0 == KP - R) {
          if (!(bits & ((1ull << (RO + R / 2 - 1)) - 1ull))) continue;
          T win[RO + R / 2 - 1];
#pragma unroll
          for (int r = 0; r < RO + R / 2 - 1; ++r) win[r] = lds_rd(col + (u0 + r) * P);
#pragma unroll
          for (int qq = 0; qq < R / 2; ++qq) {
            const T wgt = wyr[u0 + qq];
#pragma unroll
            for (int r = 0; r < RO; ++r) acc[r] = fma(wgt, win[r + qq], acc[r]);
          }
          continue;
        }
        if (!(bits & ((1ull << (RO + R - 1)) - 1ull))) continue;
        T win[RO + R - 1];
#pragma unroll
        for (int r = 0; r < RO + R - 1; ++r) win[r] = lds_rd(col + (u0 + r) * P);
#pragma unroll
        for (int qq = 0; qq < R; ++qq) {
          const T wgt = wyr[u0 + qq];
#pragma unroll
          for (int r = 0; r < RO; ++r) acc[r] = fma(wgt, win[r + qq], acc[r]);
        }
      }
    }
    STAMP(5);
    double gf = 0.0, ff = 0.0;
    if (fold_w) {
#pragma unroll
      for (int r = 0; r < RO; ++r) {
        if constexpr (OBS) {
          if (pixx >= 0 && ((fvalid >> r) & 1u)) {
            const double dd = fdv[r] - (double)acc[r], qq = dd * dd;
            if (qq == qq) gf += qq;                         // nansum (:601)
          }
        } else {
          const double av = (pixx >= 0 && ((fvalid >> r) & 1u)) ? (double)acc[r] : 0.0;
          gf = fma(av, fdv[r], gf);
          ff = fma(av, av, ff);
        }
      }
    } else if constexpr (REGRID) {
      // conv tile + bilinear regrid, as k_conv_chi2
      const int ixa = head(1), ixb = head(2);
      const int jya = head(3), jyb = head(4);
      const int ix = ixa + lane;
      int i1 = -1;
      double t1 = 0.0;
      double dv[NRQ], t0q[NRQ];
      int i0q[NRQ];
      if (ix < ixb) {
        i1 = cellx[(long long)wi * g.nx + ix] - x0;
        t1 = fracx[(long long)wi * g.nx + ix];
      }
#pragma unroll
      for (int q = 0; q < NRQ; ++q) {
        const int jy = jya + wv + CV_NW * q;
        const bool rowin = jy < jyb;
        i0q[q] = rowin ? celly[(long long)wi * g.ny + jy] - y0 : -1;
        t0q[q] = rowin ? fracy[(long long)wi * g.ny + jy] : 0.0;
        dv[q] = (ix < ixb && rowin) ? dat[(long long)jy * g.nx + ix] : 0.0;
      }
      const bool xown = i1 >= 0 && i1 < TX - 1;
      lds_barrier();                                        // column pass done reading LDS
#pragma unroll
      for (int r = 0; r < RO; ++r) lds[(wv * RO + r) * PC + lane] = acc[r];
      lds_barrier();
#pragma unroll
      for (int q = 0; q < NRQ; ++q) {
        const int jy = jya + wv + CV_NW * q;
        if (jy < jyb) {
          const int i0 = i0q[q];
          if (xown && i0 >= 0 && i0 < TY - 1) {
            const double t0 = t0q[q];
            const T* c = lds + i0 * PC + i1;
            const double v00 = c[0], v01 = c[1], v10 = c[PC], v11 = c[PC + 1];
            const double val = v00 * (1 - t0) * (1 - t1) + v01 * (1 - t0) * t1 +
                               v10 * t0 * (1 - t1) + v11 * t0 * t1;
            gf = fma(val, dv[q], gf);
            ff = fma(val, val, ff);
          }
        }
      }
      for (int ix2 = ix; ix2 < ixb; ix2 += 64) {
        const int i1b = cellx[(long long)wi * g.nx + ix2] - x0;
        if (i1b < 0 || i1b >= TX - 1) continue;
        const double t1b = fracx[(long long)wi * g.nx + ix2];
        for (int jy = jya + wv + (ix2 == ix ? CV_NW * NRQ : 0); jy < jyb; jy += CV_NW) {
          const int i0 = celly[(long long)wi * g.ny + jy] - y0;
          if (i0 < 0 || i0 >= TY - 1) continue;
          const double t0 = fracy[(long long)wi * g.ny + jy];
          const T* c = lds + i0 * PC + i1b;
          const double val = (double)c[0] * (1 - t0) * (1 - t1b) + (double)c[1] * (1 - t0) * t1b +
                             (double)c[PC] * t0 * (1 - t1b) + (double)c[PC + 1] * t0 * t1b;
          gf = fma(val, dat[(long long)jy * g.nx + ix2], gf);
          ff = fma(val, val, ff);
        }
      }
    }
    gf = wave_sum_dpp(gf);
    ff = wave_sum_dpp(ff);
    if (lane == 0) { sg[wv] = gf; sf[wv] = ff; }
    if (tid < DW) s_desc[par ^ 1][tid] = nextw;
    if (tid == 0) { s_ticket = fut; s_tend = fend; }
    STAMP(6);
    lds_barrier();                                          // also: every wave is done with this window's rows
    STAMP(7);
    if (tid == 0) {
      double av = 0.0, b2 = 0.0;
#pragma unroll
      for (int q = 0; q < CV_NW; ++q) { av += sg[q]; b2 += sf[q]; }
      if constexpr (OBS) {
        partial[e] = av;
      } else {
        partial[3 * (long long)e] = av;
        partial[3 * (long long)e + 1] = b2;
        partial[3 * (long long)e + 2] = 1.0;
      }
    }
    it = nxt;
    if (nxt + 1 < nend) nxt = nxt + 1;                      // still inside the batch
    else { nxt = __builtin_amdgcn_readfirstlane(s_ticket); nend = __builtin_amdgcn_readfirstlane(s_tend); }
  }
#ifdef SLAMHIP_PHASE_STAMPS
  if (!REGRID && tid == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(count + 32);
    for (int p_ = 0; p_ < 8; ++p_) atomicAdd(&dbg[p_], tacc[p_]);
  }
#endif
#undef STAMP
}

// k_conv_2d: general (rotated) beam, direct K x K taps on an LDS tile.
constexpr int C2_TX = 64, C2_TY = 16, C2_RY = 4;
__global__ void __launch_bounds__(256)
k_conv_2d(FitK g, const double* __restrict__ in, double* __restrict__ out) {
  extern __shared__ double lds[];
  const int K = g.kbeam, ng = g.ngauss, n = g.nneed;
  const int rows = C2_TY + K - 1, cols = C2_TX + K - 1;
  const int P = cols | 1;
  const int tid = threadIdx.x;
  const long long plane = (long long)n * n;
  const double* src = in + (long long)blockIdx.z * plane;
  double* dst = out + (long long)blockIdx.z * plane;
  const int x0 = blockIdx.x * C2_TX, y0 = blockIdx.y * C2_TY;
  for (int e = tid; e < rows * P; e += 256) {
    const int rr = e / P, cc = e - rr * P;
    const int gy = y0 - ng + rr, gx = x0 - ng + cc;
    double v = 0.0;
    if (cc < cols && gy >= 0 && gy < n && gx >= 0 && gx < n)
      v = src[(long long)gy * n + gx];
    lds[e] = v;
  }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;          // 4 row groups
  double acc[C2_RY];
#pragma unroll
  for (int r = 0; r < C2_RY; ++r) acc[r] = 0.0;
  // out[y][x] = sum_{dy,dx} beam[dy][dx] * in[y+ng-dy][x+ng-dx]
  //           = sum_{a,b} beam[K-1-a][K-1-b] * tile[y+a][x+b]
  for (int a = 0; a < K; ++a) {
    const double* brow = g.beam + (long long)(K - 1 - a) * K;
    for (int b = 0; b < K; ++b) {
      const double wgt = brow[K - 1 - b];
#pragma unroll
      for (int r = 0; r < C2_RY; ++r)
        acc[r] = fma(wgt, lds[(ty + 4 * r + a) * P + tx + b], acc[r]);
    }
  }
  const int gx = x0 + tx;
#pragma unroll
  for (int r = 0; r < C2_RY; ++r) {
    const int gy = y0 + ty + 4 * r;
    if (gx < n && gy < n) dst[(long long)gy * n + gx] = acc[r];
  }
}

// peaktounity (:492-497): per-channel max, then scale by 1/max (0 if max == 0).
__global__ void __launch_bounds__(256)
k_peak_unity(FitK g, const double* __restrict__ in, double* __restrict__ out,
             long long plane) {
  __shared__ double red[256];
  const double* src = in + (long long)blockIdx.x * plane;
  double* dst = out + (long long)blockIdx.x * plane;
  double m = -std::numeric_limits<double>::infinity();
  for (long long e = threadIdx.x; e < plane; e += 256) m = fmax(m, src[e]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  const double peak = red[0];
  const double sc = (peak == 0.0) ? 0.0 : 1.0 / peak;
  for (long long e = threadIdx.x; e < plane; e += 256) dst[e] = src[e] * sc;
}

// ---------------------------------------------------------------------------
// k_regrid: bilinear regrid onto the observed pixels (:475-483, scipy
// RegularGridInterpolator linear, fill 0 outside) fused with the two sums of
// get_scale (:485-490).  Wave shuffle reduction, then a block reduction; the
// per-block partials are summed in fixed order by k_finalize.
// ---------------------------------------------------------------------------
constexpr int RG_ROWS = 16;   // observed rows per block

template <bool WRITE>
__global__ void __launch_bounds__(RED_THREADS)
k_regrid(FitK g, const int* __restrict__ cellx, const double* __restrict__ fracx,
         const int* __restrict__ celly, const double* __restrict__ fracy,
         const double* __restrict__ iconv, double* __restrict__ partial,
         double* __restrict__ cube_out) {
  const int wi = blockIdx.z / g.nv, k = blockIdx.z - wi * g.nv;
  const int n = g.nneed;
  const int ix = blockIdx.x * RED_THREADS + threadIdx.x;
  const bool live = ix < g.nx;
  const int i1 = live ? cellx[(long long)wi * g.nx + ix] : -1;
  const double t1 = live ? fracx[(long long)wi * g.nx + ix] : 0.0;
  const double* src = iconv + ((long long)wi * g.nv + k) * n * n;
  const long long obase = (long long)k * g.ny * g.nx;
  const int j0 = blockIdx.y * RG_ROWS, j1 = min(g.ny, j0 + RG_ROWS);
  double gf = 0.0, ff = 0.0;
  for (int jy = j0; jy < j1; ++jy) {
    const int i0 = celly[(long long)wi * g.ny + jy];      // wave-uniform
    const double t0 = fracy[(long long)wi * g.ny + jy];
    if (!live) continue;
    double val = 0.0;                                     // fill_value=0 outside
    if (i0 == -2 || i1 == -2) {
      val = __builtin_nan("");
    } else if (i0 >= 0 && i1 >= 0) {
      const double* c = src + (long long)i0 * n + i1;
      const double v00 = c[0], v01 = c[1], v10 = c[n], v11 = c[n + 1];
      val = v00 * (1 - t0) * (1 - t1) + v01 * (1 - t0) * t1 +
            v10 * t0 * (1 - t1) + v11 * t0 * t1;
    }
    const long long e = obase + (long long)jy * g.nx + ix;
    if (WRITE) cube_out[(long long)wi * g.nv * g.ny * g.nx + e] = val;
    gf = fma(val, g.data[e], gf);
    ff = fma(val, val, ff);
  }
  __shared__ double sg[RED_THREADS / 64], sf[RED_THREADS / 64];
  gf = wave_sum(gf);
  ff = wave_sum(ff);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { sg[wv] = gf; sf[wv] = ff; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < RED_THREADS / 64; ++q) { a += sg[q]; b += sf[q]; }
    const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[3 * blk] = a;
    partial[3 * blk + 1] = b;
    partial[3 * blk + 2] = 1.0;
  }
}

// k_finalize: one block per walker; fixed-order sum of the partials, then
//   scale = gf/ff, 0 if ff == 0 or scale < 0                        (:488-489)
//   chi2  = sum((d - scale*I)^2)/sigma^2/pixperbeam                 (:601-602)
//         = (sum d^2 - 2 s gf + s^2 ff)/sigma^2/pixperbeam
// stats[w] = {gf, ff, scale, chi2, non-empty blocks, blocks}
__global__ void __launch_bounds__(256)
k_finalize(FitK g, const double* __restrict__ partial, int nblk,
           const unsigned char* __restrict__ winflag,
           double* __restrict__ stats, double* __restrict__ chi2_out) {
  __shared__ double sa[4], sb[4], sl[4];
  const int wi = blockIdx.x;
  const double* p = partial + (long long)wi * nblk * 3;
  const unsigned char* wf = winflag ? winflag + (long long)wi * nblk : nullptr;
  double a = 0.0, b = 0.0, live = 0.0;
  for (int q = threadIdx.x; q < nblk; q += 256) {
    if (wf && !wf[q]) continue;          // window never convolved: its slot is stale
    a += p[3 * q]; b += p[3 * q + 1]; live += p[3 * q + 2];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  live = wave_sum(live);
  if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; sl[threadIdx.x >> 6] = live; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    b = (sb[0] + sb[1]) + (sb[2] + sb[3]);
    live = (sl[0] + sl[1]) + (sl[2] + sl[3]);
    double s = a / b;
    if (b == 0.0 || s < 0.0) s = 0.0;
    double c2;
    if (b == 0.0) {
      // ff == 0 is tested first (:489): scale 0 whatever gf is, the model is 0 and
      // chi2 = nansum(data^2) (ssq_data skips NaNs like nansum, :601)
      c2 = g.ssq_data / (g.sigma * g.sigma) / g.pixperbeam;
    } else if (g.data_has_nan || a != a || b != b) {
      // np.sum propagates the NaN into the scale, the whole model becomes NaN
      // and nansum() of an all-NaN residual is 0 (:486-489, :601).
      c2 = 0.0;
    } else {
      c2 = (g.ssq_data - 2.0 * s * a + s * s * b) / (g.sigma * g.sigma) / g.pixperbeam;
    }
    if (stats) {
      stats[6 * wi + 0] = a; stats[6 * wi + 1] = b;
      stats[6 * wi + 2] = s; stats[6 * wi + 3] = c2;
      stats[6 * wi + 4] = live; stats[6 * wi + 5] = (double)nblk;
    }
    if (chi2_out) chi2_out[wi] = c2;
  }
}


// ---------------------------------------------------------------------------
// Accounting kernels for bench.py's roofline block (no counterpart in the
// reference, never on the hot path): what the last launch group actually
// touched and computed, counted from the scratch it left behind.
//   cnt[0] live conv windows              cnt[1] occupied Iunif tiles under them (halo included)
//   cnt[2] window rows with emission      cnt[3] column-pass tap blocks executed
//   cnt[4] lit (tile, channel) planes     cnt[5] stored geometry groups (64 samples x surface)
//   cnt[6] k_profile hits: (lane, sample rank, surface, channel block) that passed the velocity cull
//   cnt[7] the same per wave (64 samples): (wave, surface, channel block) with at least one such lane
// ---------------------------------------------------------------------------
__device__ inline void count_add(unsigned long long* dst, unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

__global__ void __launch_bounds__(256)
k_count_windows(const WinDesc* __restrict__ list, const int* __restrict__ count, int ty, int kpad,
                int rb, unsigned long long* __restrict__ cnt, int rev_total) {
  // rb: output rows per tap block -- CONV_R for the column pass of the factorised form; the
  // direct-tap form works on blocks of 2*CONV_R (64-row windows) or CONV_R output rows and
  // skips a block of CONV_R tap rows when the rb + CONV_R - 1 window rows under it are empty
  const int nlive = count[0];
  if (rev_total) list += rev_total - nlive;                 // k_compact's second list
  unsigned long long nocc = 0, rows = 0, blocks = 0, live = 0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < nlive; e += gridDim.x * 256) {
    const WinDesc& d = list[e];
    live += 1;
    nocc += d.nocc;
    rows += __popcll(d.rowmask[0]) + __popcll(d.rowmask[1]) + __popcll(d.rowmask[2]) + __popcll(d.rowmask[3]);
    for (int wv = 0; wv < ty / rb; ++wv)
      for (int u0 = 0; u0 < kpad; u0 += CONV_R) {             // as the column pass / direct taps of k_conv_chi2
        const int ra = wv * rb + u0, w0 = ra >> 6, sh = ra & 63;
        const unsigned long long lo = d.rowmask[w0], hi = w0 < 3 ? d.rowmask[w0 + 1] : 0ull;
        const unsigned long long bits = (lo >> sh) | (sh ? (hi << (64 - sh)) : 0ull);
        if (bits & ((1ull << (rb + CONV_R - 1)) - 1ull)) blocks += 1;
      }
  }
  count_add(&cnt[0], live);
  count_add(&cnt[1], nocc);
  count_add(&cnt[2], rows);
  count_add(&cnt[3], blocks);
}

__global__ void __launch_bounds__(256)
k_count_flags(const unsigned char* __restrict__ flags, long long nflag,
              const unsigned* __restrict__ gflag, long long ngflag,
              unsigned long long* __restrict__ cnt) {
  unsigned long long lit = 0, groups = 0;
  const long long stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nflag; e += stride) lit += flags[e] != 0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < ngflag; e += stride) groups += __popc(gflag[e] & 15u);
  count_add(&cnt[4], lit);
  count_add(&cnt[5], groups);
}

template <typename T2>
__global__ void __launch_bounds__(256)
k_count_hits(FitK g, const WalkerK* __restrict__ wk, const T2* __restrict__ geom,
             const unsigned* __restrict__ gflag, int nw, unsigned long long* __restrict__ cnt) {
  const long long slot = (long long)blockIdx.x * 256 + threadIdx.x;   // nslot is a multiple of 256
  unsigned long long hits = 0, wavehits = 0;
  const int nchunk = g.nvpad / CH;
  for (int wi = 0; wi < nw; ++wi) {
    const unsigned fm = gflag[(long long)wi * (g.nslot >> 6) + (slot >> 6)];
    const double hw = wk[wi].halfwidth, voff = wk[wi].voff;
    for (int s = 0; s < 4; ++s) {
      if (!((fm >> s) & 1u)) continue;                      // wave-uniform
      const double o = (double)geom[((long long)wi * 4 + s) * g.nslot + slot].x + voff;
      for (int cix = 0; cix < nchunk; ++cix) {
        const double cmid = 0.5 * (g.ch_vmin[cix] + g.ch_vmax[cix]);
        const double hwc = hw + 0.5 * (g.ch_vmax[cix] - g.ch_vmin[cix]);
        const bool hit = !(fabs(o - cmid) > hwc);
        hits += hit;
        const unsigned long long any = __ballot(hit);          // every lane of the wave takes part
        wavehits += ((threadIdx.x & 63) == 0 && any != 0ull) ? 1 : 0;
      }
    }
  }
  count_add(&cnt[6], hits);
  count_add(&cnt[7], wavehits);
}

// ---------------------------------------------------------------------------
// 'mom0*' scalings (:520-532): the UNCONVOLVED cube is regridded first, every
// pixel's spectrum is rescaled so that its moment 0 equals the deconvolved
// observed one, and the beam is applied on the observed grid.
// ---------------------------------------------------------------------------
// k_tile_krange: per (walker, Iunif tile) the channel range [lo, hi) in which
// k_profile lit the tile (lo = nv, hi = 0 for a tile that stayed dark), so that
// k_regrid_mom0 visits only channels that can hold emission under a pixel.
// kmask[walker][tile][(nv + 31) / 32]: the same occupancy as one bit per channel (k_regrid_mom0
// asks for a word per 32 channels and corner instead of a byte per channel and corner).
__global__ void __launch_bounds__(256)
k_tile_krange(FitK g, const unsigned char* __restrict__ flags, int ntile,
              int2* __restrict__ krange, unsigned* __restrict__ kmask) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntile) return;
  const int wi = blockIdx.y;
  const unsigned char* f = flags + (long long)wi * g.nv * ntile + t;
  const int nword = (g.nv + 31) >> 5;
  unsigned* km = kmask + ((long long)wi * ntile + t) * nword;
  int lo = g.nv, hi = 0;
  for (int k0 = 0; k0 < g.nv; k0 += 32) {
    unsigned word = 0u;
    for (int k = k0; k < min(k0 + 32, g.nv); ++k)
      if (f[(long long)k * ntile]) { lo = min(lo, k); hi = k + 1; word |= 1u << (k - k0); }
    km[k0 >> 5] = word;
  }
  krange[(long long)wi * ntile + t] = make_int2(lo, hi);
}

// k_mom_model: the model's moment on its own grid, M[j][i] = sum_k Iunif[k][j][i] over the lit
// channels of the pixel's tile (one workgroup per 32x8 tile and walker; every pixel is written,
// dark tiles with 0).  The regrid is linear, so the moment of the regridded cube (:521) is the
// regridded M: k_regrid_mom0 then reads Iunif once instead of twice.
__global__ void __launch_bounds__(256)
k_mom_model(FitK g, const double* __restrict__ iunif, const unsigned char* __restrict__ flags,
            const int2* __restrict__ krange, int ftx, int fty, double* __restrict__ mom) {
  const int tile = blockIdx.x, wi = blockIdx.y;
  const int ntile = ftx * fty, n = g.nneed;
  const int tby = tile / ftx, tbx = tile - tby * ftx;
  const int x = tbx * PT_X + (threadIdx.x & (PT_X - 1)), y = tby * PT_Y + threadIdx.x / PT_X;
  if (x >= n || y >= n) return;
  const int2 kr = krange[(long long)wi * ntile + tile];     // block-uniform
  const unsigned char* f = flags + (long long)wi * g.nv * ntile + tile;
  const double* src = iunif + ((long long)wi * g.nv * n + y) * n + x;
  const long long plane = (long long)n * n;
  double a = 0.0;
  int k = kr.x;
  for (; k + 4 <= kr.y; k += 4) {                           // four channels in flight
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = f[(long long)(k + u) * ntile] ? src[(k + u) * plane] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) a += v[u];
  }
  for (; k < kr.y; ++k)
    if (f[(long long)k * ntile]) a += src[k * plane];
  mom[(long long)wi * plane + (long long)y * n + x] = a;
}

// k_regrid_mom0: regrid (:475-483) and per-pixel moment-0 scale (:521-525) in
// one pass.  A wave is 64 neighbouring pixels of one observed row, a thread one
// pixel: the bilinear value of every channel is formed twice (once for the
// moment, once for the scaled output) instead of storing the unscaled cube;
// Iunif tiles that k_profile left unlit are never used (their occupancy byte is
// 0), so Iunif need not be cleared.  Raises the occupancy byte of the pixel's
// k_conv_obs_sep tile.  With `rowflag` ([walker][channel][row][64-pixel segment]
// bytes, cleared by the caller) a wave stores a channel only if one of its
// pixels is non-zero there and marks the segment; k_conv_obs_sep reads nothing
// from unmarked segments, so dark sky is neither written nor read.  Without it
// every pixel of every channel is stored.
__global__ void __launch_bounds__(256)
k_regrid_mom0(FitK g, const int* __restrict__ cellx, const double* __restrict__ fracx,
              const int* __restrict__ celly, const double* __restrict__ fracy,
              const double* __restrict__ iunif, const unsigned char* __restrict__ flags,
              const int2* __restrict__ krange, int ftx, int fty, double* __restrict__ cube,
              unsigned char* __restrict__ tileflag, int tile_ty, int ntx, int nty,
              unsigned char* __restrict__ rowflag,
              unsigned char* __restrict__ oflags, int oftx, int ofty,
              unsigned char* __restrict__ owin, int ownx, int owny,
              const double* __restrict__ mom, const unsigned* __restrict__ kmask) {
  // oflags / owin (with rowflag; may be null): the occupancy bytes of the observed cube in
  // k_profile's form -- [walker][channel][32x8 pixel tile] -- and the 64 x 64 beam windows whose
  // halo holds a lit pixel, for k_compact + k_conv_rows<OBS> (the batch path of one-term beams)
  const long long npix = (long long)g.ny * g.nx;
  const int lane = threadIdx.x & 63;
  const int jy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (jy >= g.ny) return;                          // whole waves
  const int ixr = blockIdx.x * 64 + lane;
  const bool valid = ixr < g.nx;
  const int ix = valid ? ixr : g.nx - 1;
  const long long p = (long long)jy * g.nx + ix;
  const int wi = blockIdx.z;
  const int i1 = cellx[(long long)wi * g.nx + ix], i0 = celly[(long long)wi * g.ny + jy];
  const double t1 = fracx[(long long)wi * g.nx + ix], t0 = fracy[(long long)wi * g.ny + jy];
  const int n = g.nneed;
  const bool nanq = i0 == -2 || i1 == -2;         // NaN query point
  const bool inside = i0 >= 0 && i1 >= 0;         // fill_value = 0 outside
  const long long ntile = (long long)ftx * fty;
  int ta = 0, tb = 0, tc = 0, td = 0;
  // channels outside [klo, khi) are dark in all four tiles under the pixel: their
  // value is exactly 0, which neither moves the moment nor needs its loads
  int klo = g.nv, khi = 0;
  const bool query = inside && !nanq && valid;
  if (query) {
    ta = (i0 / PT_Y) * ftx + i1 / PT_X;       tb = (i0 / PT_Y) * ftx + (i1 + 1) / PT_X;
    tc = ((i0 + 1) / PT_Y) * ftx + i1 / PT_X; td = ((i0 + 1) / PT_Y) * ftx + (i1 + 1) / PT_X;
    const int2* kr = krange + (long long)wi * ntile;
    const int2 ra = kr[ta], rb = kr[tb], rc = kr[tc], rd = kr[td];
    klo = min(min(ra.x, rb.x), min(rc.x, rd.x));
    khi = max(max(ra.y, rb.y), max(rc.y, rd.y));
  }
  const int c0 = query ? i0 : 0, c1 = query ? i1 : 0;   // a cell every lane may read
  // A NaN query point has a NaN moment and so a zero spectrum (:521-525), a point
  // outside the grid a zero one: both leave [klo, khi) empty.
  // Channels k0 .. k0+RG_U-1 at once (indices past the cube repeat its last
  // channel; the caller drops them): all flag bytes, then all four corners of
  // every channel whatever its flags say, so that one memory round trip serves
  // RG_U channels; corners in unlit tiles hold old data and are dropped here.
  constexpr int RG_U = 4;                          // (2 and 8 measured: 50.2 / 49.2 instead of 46.4 us/eval)
  // the four corner tiles' occupancy words of the 32-channel block in hand (k0 is wave-uniform
  // and a multiple of RG_U, so a group of RG_U channels lies in one word)
  const int nword = (g.nv + 31) >> 5;
  int curw = -1;
  unsigned wa = 0u, wb = 0u, wc = 0u, wd = 0u;
  auto values = [&](int k0, double* out) {
    bool fa[RG_U], fb[RG_U], fc[RG_U], fd[RG_U];
    double l00[RG_U], l01[RG_U], l10[RG_U], l11[RG_U];
    if ((k0 >> 5) != curw) {
      curw = k0 >> 5;
      const unsigned* km = kmask + (long long)wi * ntile * nword + curw;
      wa = km[(long long)ta * nword]; wb = km[(long long)tb * nword];
      wc = km[(long long)tc * nword]; wd = km[(long long)td * nword];
    }
#pragma unroll
    for (int u = 0; u < RG_U; ++u) {
      const int sh = (k0 + u) & 31;                // (k0 + u >= nv: the bit is 0)
      fa[u] = (wa >> sh) & 1u; fb[u] = (wb >> sh) & 1u; fc[u] = (wc >> sh) & 1u; fd[u] = (wd >> sh) & 1u;
    }
#pragma unroll
    for (int u = 0; u < RG_U; ++u) {
      const int kk = min(k0 + u, g.nv - 1);
      const double* c = iunif + (((long long)wi * g.nv + kk) * n + c0) * n + c1;
      l00[u] = c[0]; l01[u] = c[1]; l10[u] = c[n]; l11[u] = c[n + 1];
    }
#pragma unroll
    for (int u = 0; u < RG_U; ++u) {
      const double v00 = fa[u] ? l00[u] : 0.0, v01 = fb[u] ? l01[u] : 0.0;
      const double v10 = fc[u] ? l10[u] : 0.0, v11 = fd[u] ? l11[u] : 0.0;
      out[u] = v00 * (1 - t0) * (1 - t1) + v01 * (1 - t0) * t1 +
               v10 * t0 * (1 - t1) + v11 * t0 * t1;   // scipy _rgi.py:478-482, as k_regrid
    }
  };
  // the pixel's moment: k_mom_model's map regridded like a channel (dark tiles hold 0 there)
  double m0 = 0.0;
  if (query) {
    const double* mm = mom + ((long long)wi * n + c0) * n + c1;
    m0 = mm[0] * (1 - t0) * (1 - t1) + mm[1] * (1 - t0) * t1 + mm[n] * t0 * (1 - t1) + mm[n + 1] * t0 * t1;
  }
  m0 = m0 * g.dv;
  if (m0 < 0.0) m0 = __builtin_nan("");
  // m0 == 0: every channel is 0 (0*dec/0 -> NaN -> 0): the pixel is dark
  if (m0 == 0.0) { klo = g.nv; khi = 0; }
  // the wave walks the union of its pixels' channel ranges
  int wlo = klo, whi = khi;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    wlo = min(wlo, __shfl_xor(wlo, o));
    whi = max(whi, __shfl_xor(whi, o));
  }
  double* c = cube + (long long)wi * g.nv * npix + p;
  unsigned char* fl = tileflag
      ? tileflag + (long long)wi * g.nv * ntx * nty + (jy / tile_ty) * ntx + ix / C2_TX : nullptr;
  unsigned char* rf = rowflag
      ? rowflag + ((long long)wi * g.nv * g.ny + jy) * gridDim.x + blockIdx.x : nullptr;
  const double dec = g.mom0decon[p];
  const double big = std::numeric_limits<double>::max();
  if (!rf && valid) {
    for (int k = 0; k < min(wlo, g.nv); ++k) c[(long long)k * npix] = 0.0;
    for (int k = max(whi, wlo); k < g.nv; ++k) c[(long long)k * npix] = 0.0;
  }
  for (int k0 = wlo & ~(RG_U - 1); k0 < whi; k0 += RG_U) {   // (channels below wlo come out 0 and are skipped)
    double vv[RG_U];
    values(k0, vv);
#pragma unroll
    for (int u = 0; u < RG_U; ++u) {
      const int k = k0 + u;
      if (k >= whi) break;                         // wave-uniform
      double v = 0.0;
      if (k >= klo && k < khi) {
        v = vv[u] * dec / m0;
        if (v != v) v = 0.0;
        else if (v > big) v = big;
        else if (v < -big) v = -big;
      }
      const bool lit = v != 0.0;
      if (rf) {
        const unsigned long long lb = __ballot(lit);
        if (lb == 0ull) continue;                  // nothing of this wave in channel k
        if (lane == 0) rf[(long long)k * g.ny * gridDim.x] = 1;
        if (oflags && (lane & 31) == 0 && ((lb >> lane) & 0xffffffffull)) {
          // this half wave's 32 pixels are one tile row of tile (jy / 8, px0 / 32)
          const int px0 = blockIdx.x * 64 + lane, ngb = g.ngauss;
          const long long zk = (long long)wi * g.nv + k;
          oflags[zk * oftx * ofty + (jy / PT_Y) * oftx + px0 / PT_X] = 1;
          const int bx0 = max(px0 - ngb, 0) / CONV_TX, bx1 = min(ownx - 1, (px0 + PT_X - 1 + ngb) / CONV_TX);
          const int by0 = max(jy - ngb, 0) / 64, by1 = min(owny - 1, (jy + ngb) / 64);
          for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) owin[(zk * owny + by) * ownx + bx] = 1;
        }
      }
      if (valid) c[(long long)k * npix] = v;
      if (fl && lit) fl[(long long)k * ntx * nty] = 1;
    }
  }
}

// k_conv_obs: zero-padded 'same' convolution with the K x K beam on the observed
// (ny, nx) grid (:530), direct taps on an LDS tile, fused with the residual sum
// of lnprob (:601): partial[block] = sum over its pixels of (data - model)^2,
// NaN terms skipped like nansum.  WRITE also stores the model cube.
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_conv_obs(FitK g, const double* __restrict__ in, double* __restrict__ out,
           double* __restrict__ partial) {
  extern __shared__ double lds[];
  const int K = g.kbeam, ng = g.ngauss, nx = g.nx, ny = g.ny;
  const int rows = C2_TY + K - 1, cols = C2_TX + K - 1;
  const int P = cols | 1;
  const int tid = threadIdx.x;
  const long long plane = (long long)ny * nx;
  const int k = blockIdx.z % g.nv;
  const double* src = in + (long long)blockIdx.z * plane;
  const int x0 = blockIdx.x * C2_TX, y0 = blockIdx.y * C2_TY;
  for (int e = tid; e < rows * P; e += 256) {
    const int rr = e / P, cc = e - rr * P;
    const int gy = y0 - ng + rr, gx = x0 - ng + cc;
    double v = 0.0;
    if (cc < cols && gy >= 0 && gy < ny && gx >= 0 && gx < nx)
      v = src[(long long)gy * nx + gx];
    lds[e] = v;
  }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;
  double acc[C2_RY];
#pragma unroll
  for (int r = 0; r < C2_RY; ++r) acc[r] = 0.0;
  for (int a = 0; a < K; ++a) {
    const double* brow = g.beam + (long long)(K - 1 - a) * K;
    for (int b = 0; b < K; ++b) {
      const double wgt = brow[K - 1 - b];
#pragma unroll
      for (int r = 0; r < C2_RY; ++r)
        acc[r] = fma(wgt, lds[(ty + 4 * r + a) * P + tx + b], acc[r]);
    }
  }
  const int gx = x0 + tx;
  double ss = 0.0;
#pragma unroll
  for (int r = 0; r < C2_RY; ++r) {
    const int gy = y0 + ty + 4 * r;
    if (gx < nx && gy < ny) {
      const long long e = (long long)gy * nx + gx;
      if (WRITE) out[(long long)blockIdx.z * plane + e] = acc[r];
      const double d = g.data[(long long)k * plane + e] - acc[r];
      const double q = d * d;
      if (q == q) ss += q;
    }
  }
  __shared__ double sred[4];
  ss = wave_sum(ss);
  if ((tid & 63) == 0) sred[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) {
    const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[blk] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  }
}


// k_conv_obs_sep: the same convolution + residual sum with the beam written as a
// sum of `nrank` outer products (exactly one for an axis-aligned Gaussian; the
// leading singular triplets, tail < 1e-13, for a rotated one): per term a row
// pass into an LDS intermediate and a column pass, 2K instead of K^2 taps.
template <int TY, int NIT, bool WRITE>
__device__ __forceinline__ void
conv_obs_sep_window(const FitK& g, const double* __restrict__ in, double* __restrict__ out,
                    double* __restrict__ partial, int kpad, int nrank,
                    const unsigned char* __restrict__ tileflag, const double* __restrict__ empty_ss,
                    const double* __restrict__ gxp, const double* __restrict__ gyp,
                    const unsigned char* __restrict__ rowflag,
                    const int bx, const int by, const int bz, const int ntx, const int nty) {
  extern __shared__ double lds[];
  constexpr int TX = C2_TX;
  const int K = g.kbeam, ng = g.ngauss, nx = g.nx, ny = g.ny;
  if (tileflag) {
    // no emission in any tile under this window's halo: the model is 0 here and
    // the residual sum is the tile's sum of data^2, computed once per fit by this
    // same kernel on an all-zero cube (`empty_ss`)
    const int fx0 = max(bx * TX - ng, 0) / TX;
    const int fx1 = min(bx * TX + TX - 1 + ng, nx - 1) / TX;
    const int fy0 = max(by * TY - ng, 0) / TY;
    const int fy1 = min(by * TY + TY - 1 + ng, ny - 1) / TY;
    const unsigned char* fl = tileflag + (long long)bz * ntx * nty;
    bool some = false;
    const int nf = (fx1 - fx0 + 1) * (fy1 - fy0 + 1);
    for (int e = threadIdx.x; e < nf; e += 256) {
      const int fy = fy0 + e / (fx1 - fx0 + 1), fx = fx0 + e % (fx1 - fx0 + 1);
      some |= fl[fy * ntx + fx] != 0;
    }
    if (!__syncthreads_or(some)) {
      const long long blk1 = ((long long)(bz % g.nv) * nty + by) * ntx + bx;
      const long long blk = ((long long)bz * nty + by) * ntx + bx;
      if (threadIdx.x == 0) partial[blk] = empty_ss[blk1];
      if (WRITE) {
        const long long plane = (long long)ny * nx;
        for (int e = threadIdx.x; e < TX * TY; e += 256) {
          const int gy = by * TY + e / TX, gx = bx * TX + (e & (TX - 1));
          if (gx < nx && gy < ny) out[(long long)bz * plane + (long long)gy * nx + gx] = 0.0;
        }
      }
      return;
    }
  }
  constexpr int R = 8;                  // outputs per thread along the tap axis
  static_assert(TY % R == 0 && TY / R <= 4, "one 8-row block per wave");
  // tile and intermediate are padded to the tap count rounded up to R (kpad):
  // the sliding windows below read up to kpad+R-2 past their origin, against
  // zero weights there
  const int rows = TY + kpad - 1, cols = TX + kpad - 1;
  const int P = cols | 1, P2 = TX + 1;
  double* tile = lds;                   // [rows][P]
  double* tmp = nrank > 1 ? lds + rows * P : lds;   // [rows][P2]; overlays the tile for one term
  const int tid = threadIdx.x;
  const long long plane = (long long)ny * nx;
  const int k = bz % g.nv;
  const double* src = in + (long long)bz * plane;
  const int x0 = bx * TX, y0 = by * TY;
  const int vrows = TY + K - 1, vcols = TX + K - 1;
  // k_regrid_mom0 stored only the 64-pixel row segments it marked
  const int nseg = (nx + 63) >> 6;
  const unsigned char* rfl = rowflag ? rowflag + (long long)bz * ny * nseg : nullptr;
  bool any = false;
  for (int e = tid; e < rows * P; e += 256) {
    const int rr = e / P, cc = e - rr * P;
    const int gy = y0 - ng + rr, gx = x0 - ng + cc;
    double v = 0.0;
    if (rr < vrows && cc < vcols && gy >= 0 && gy < ny && gx >= 0 && gx < nx &&
        (!rfl || rfl[gy * nseg + (gx >> 6)]))
      v = src[(long long)gy * nx + gx];
    any |= v != 0.0;
    tile[e] = v;
  }
  const bool live = __syncthreads_or(any) != 0;   // empty sky convolves to zero
  const int lane = tid & 63, wv = tid >> 6;
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0;
  if (live) {
    const int nitem = rows * (TX / R);
    for (int rk = 0; rk < nrank; ++rk) {
      // taps as kernel arguments: block-uniform addresses -> scalar loads
      const double* __restrict__ wx = gxp + (long long)rk * kpad;
      const double* __restrict__ wy = gyp + (long long)rk * kpad;
      if (rk) __syncthreads();
      // row pass: tmp[row][x] = sum_u wx[u] * tile[row][x+u]; an item is R
      // neighbouring x of one row, lanes <-> rows (odd pitch: no bank conflicts).
      // Results stay in registers across the barrier so that, for a one-term
      // beam, the intermediate can overlay the tile.
      double res[NIT][R];
#pragma unroll
      for (int q = 0; q < NIT; ++q) {
        const int item = q * 256 + tid;
#pragma unroll
        for (int r = 0; r < R; ++r) res[q][r] = 0.0;
        if (item < nitem) {
          const int xb = item / rows, rr = item - xb * rows;
          const double* row = tile + rr * P + xb * R;
          double win[R];
#pragma unroll
          for (int r = 0; r < R - 1; ++r) win[r] = row[r];
          for (int u0 = 0; u0 < kpad; u0 += R) {
#pragma unroll
            for (int qq = 0; qq < R; ++qq) {
              win[(qq + R - 1) % R] = row[u0 + qq + R - 1];
              const double w = wx[u0 + qq];
#pragma unroll
              for (int r = 0; r < R; ++r) res[q][r] = fma(w, win[(r + qq) % R], res[q][r]);
            }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NIT; ++q) {
        const int item = q * 256 + tid;
        if (item < nitem) {
          const int xb = item / rows, rr = item - xb * rows;
#pragma unroll
          for (int r = 0; r < R; ++r) tmp[rr * P2 + xb * R + r] = res[q][r];
        }
      }
      __syncthreads();
      // column pass: wave wv owns output rows wv*R .. wv*R+R-1, lanes <-> x
      if (wv < TY / R) {
        const double* col = tmp + (wv * R) * P2 + lane;
        double win[R];
#pragma unroll
        for (int r = 0; r < R - 1; ++r) win[r] = col[r * P2];
        for (int u0 = 0; u0 < kpad; u0 += R) {
#pragma unroll
          for (int qq = 0; qq < R; ++qq) {
            win[(qq + R - 1) % R] = col[(u0 + qq + R - 1) * P2];
            const double w = wy[u0 + qq];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fma(w, win[(r + qq) % R], acc[r]);
          }
        }
      }
    }
  }
  const int gx = x0 + lane;
  double ss = 0.0;
  if (wv < TY / R) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gy = y0 + wv * R + r;
      if (gx < nx && gy < ny) {
        const long long e = (long long)gy * nx + gx;
        if (WRITE) out[(long long)bz * plane + e] = acc[r];
        const double d = g.data[(long long)k * plane + e] - acc[r];
        const double q = d * d;
        if (q == q) ss += q;
      }
    }
  }
  __shared__ double sred[4];
  ss = wave_sum(ss);
  if ((tid & 63) == 0) sred[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) {
    const long long blk = ((long long)bz * nty + by) * ntx + bx;
    partial[blk] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  }
}

template <int TY, int NIT, bool WRITE>
__global__ void __launch_bounds__(256)
k_conv_obs_sep(FitK g, const double* __restrict__ in, double* __restrict__ out,
               double* __restrict__ partial, int kpad, int nrank,
               const unsigned char* __restrict__ tileflag, const double* __restrict__ empty_ss,
               const double* __restrict__ gxp, const double* __restrict__ gyp,
               const unsigned char* __restrict__ rowflag) {
  conv_obs_sep_window<TY, NIT, WRITE>(g, in, out, partial, kpad, nrank, tileflag, empty_ss, gxp, gyp,
                                      rowflag, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// k_obs_list: thread per (walker, channel, beam window).  A window with no
// emission in any tile under its halo gets the tile's precomputed residual sum
// (`empty_ss`, see k_conv_obs_sep); the others are appended to `list`
// (count[0] = their number; order is arbitrary, results do not depend on it).
__global__ void __launch_bounds__(256)
k_obs_list(FitK g, const unsigned char* __restrict__ tileflag, const double* __restrict__ empty_ss,
           double* __restrict__ partial, int* __restrict__ list, int* __restrict__ count,
           int ntx, int nty, int TY, long long nwin) {
  const long long blk = (long long)blockIdx.x * 256 + threadIdx.x;
  bool live = false;
  if (blk < nwin) {
    const int bx = (int)(blk % ntx), by = (int)((blk / ntx) % nty);
    const long long bz = blk / ((long long)ntx * nty);
    const int ng = g.ngauss;
    const int fx0 = max(bx * C2_TX - ng, 0) / C2_TX;
    const int fx1 = min(bx * C2_TX + C2_TX - 1 + ng, g.nx - 1) / C2_TX;
    const int fy0 = max(by * TY - ng, 0) / TY;
    const int fy1 = min(by * TY + TY - 1 + ng, g.ny - 1) / TY;
    const unsigned char* fl = tileflag + bz * ntx * nty;
    for (int fy = fy0; fy <= fy1; ++fy)
      for (int fx = fx0; fx <= fx1; ++fx) live |= fl[fy * ntx + fx] != 0;
    if (!live) partial[blk] = empty_ss[((long long)(bz % g.nv) * nty + by) * ntx + bx];
  }
  // one slot reservation per workgroup (a single counter serves its returning atomics one
  // after the other at memory, ~12 ns each): the waves' counts meet in LDS
  const unsigned long long m = __ballot(live);
  const int lane = threadIdx.x & 63;
  __shared__ int s_cnt[4], s_base;
  if (lane == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int all = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    s_base = all ? atomicAdd(&count[0], all) : 0;
  }
  __syncthreads();
  if (live) {
    int base = s_base + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_cnt[w];
    list[base] = (int)blk;
  }
}

// k_conv_obs_live: persistent workgroups take the listed windows by ticket
// (count[1]) and run k_conv_obs_sep's window on each.
template <int TY, int NIT>
__global__ void __launch_bounds__(256)
k_conv_obs_live(FitK g, const double* __restrict__ in, double* __restrict__ partial, int kpad,
                int nrank, const double* __restrict__ gxp, const double* __restrict__ gyp,
                const unsigned char* __restrict__ rowflag, const int* __restrict__ list,
                int* __restrict__ count, int ntx, int nty) {
  __shared__ int s_ticket;
  const int nlive = count[0];
  // Tickets come eight at a time: the draw is a returning atomic on one address, waited
  // for before the window can start, and one per window left that round trip exposed.
  constexpr int TB = 8;
  for (;;) {
    if (threadIdx.x == 0) s_ticket = atomicAdd(&count[1], TB);
    __syncthreads();
    const int t0 = s_ticket;
    if (t0 >= nlive) return;                       // every wave leaves here
    const int t1 = min(t0 + TB, nlive);
    for (int t = t0; t < t1; ++t) {
      const int blk = list[t];
      const int bx = blk % ntx, by = (blk / ntx) % nty, bz = blk / (ntx * nty);
      conv_obs_sep_window<TY, NIT, false>(g, in, nullptr, partial, kpad, nrank, nullptr, nullptr, gxp, gyp,
                                          rowflag, bx, by, bz, ntx, nty);
      __syncthreads();
    }
  }
}

// k_obs_empty: sum of data^2 (NaN terms dropped) over each 64 x 64 window of each channel: the
// residual sum of a window whose model is zero.  One workgroup per (window, channel), once per fit.
__global__ void __launch_bounds__(256)
k_obs_empty(FitK g, int wnx, int wny, double* __restrict__ empty) {
  __shared__ double sa[4];
  const int bx = blockIdx.x % wnx, by = blockIdx.x / wnx, k = blockIdx.y;
  const double* d = g.data + (long long)k * g.ny * g.nx;
  double a = 0.0;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int gx = bx * 64 + (e & 63), gy = by * 64 + (e >> 6);
    if (gx < g.nx && gy < g.ny) {
      const double v = d[(long long)gy * g.nx + gx], q = v * v;
      if (q == q) a += q;
    }
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) empty[(long long)k * wnx * wny + blockIdx.x] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
}

// k_sum_chi2: one block per walker, fixed-order sum of its residual partials,
// chi2 = sum / sigma^2 / pixperbeam (:601-602).  stats = {nan, nan, nan, chi2, 0, nblk}
// (the mom0 scalings have no global amplitude).
__global__ void __launch_bounds__(256)
k_sum_chi2(FitK g, const double* __restrict__ partial, int nblk,
           double* __restrict__ stats, double* __restrict__ chi2_out,
           const unsigned char* __restrict__ winflag = nullptr, const double* __restrict__ empty = nullptr) {
  // winflag / empty (k_conv_rows<OBS>): windows without emission were never run; their residual
  // sum is the fit's sum of data^2 there (`empty`, [channel][window], once per fit)
  __shared__ double sa[4];
  const int wi = blockIdx.x;
  const double* p = partial + (long long)wi * nblk;
  const unsigned char* wf = winflag ? winflag + (long long)wi * nblk : nullptr;
  double a = 0.0;
  for (int q = threadIdx.x; q < nblk; q += 256) a += (!wf || wf[q]) ? p[q] : empty[q];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    const double c2 = a / (g.sigma * g.sigma) / g.pixperbeam;
    if (stats) {
      const double nn = __builtin_nan("");
      stats[6 * wi + 0] = nn; stats[6 * wi + 1] = nn; stats[6 * wi + 2] = nn;
      stats[6 * wi + 3] = c2; stats[6 * wi + 4] = 0.0; stats[6 * wi + 5] = (double)nblk;
    }
    if (chi2_out) chi2_out[wi] = c2;
  }
}

// float Iunif -> double, for the parity taps and cubemodel at precision 1
__global__ void k_widen(const float* __restrict__ in, double* __restrict__ out, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x)
    out[e] = (double)in[e];
}

// Moment 0 and 1 of a model cube over its channels (reference makemom01,
// _channelfit.py:40-50, as plotmom :724-725 asks for it): per pixel the channels
// are added in order, products rounded before they are added, as numpy's
// np.sum(d, axis=0) and np.sum(d * v, axis=0) do, so the maps are the
// reference's bit for bit; mom1 is NaN where mom0 < clip (= 3 sigma_mom0).
__global__ void __launch_bounds__(256)
k_moments(const double* __restrict__ cube, const double* __restrict__ v, int nv, long long npix,
          double dv, double clip, double* __restrict__ mom0, double* __restrict__ mom1) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  double s = 0.0, sv = 0.0;
  for (int c = 0; c < nv; ++c) {
    const double d = cube[(size_t)c * npix + i];
    s = __dadd_rn(s, d);
    sv = __dadd_rn(sv, __dmul_rn(d, v[c]));
  }
  const double m0 = __dmul_rn(s, dv);
  mom0[i] = m0;
  mom1[i] = m0 < clip ? __builtin_nan("") : __ddiv_rn(sv, s);
}

__global__ void k_scale(double* __restrict__ cube, long long n,
                        const double* __restrict__ stats) {
  const double s = stats[2];
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x)
    cube[e] = cube[e] * s;
}

}  // namespace

// ===========================================================================
// Host side
// ===========================================================================
// Scratch of the allocation-free entry point slamhip_chi2_batch_static: every GPU
// the library runs on holds its own copy of this array, at its own address,
// until the process ends.  In host code the array's name is only a handle for
// hipGetSymbolAddress: slamhip_create asks each device once for the address of
// its copy (static_scratch below) and everything else -- kernel arguments,
// carving -- goes through that pointer.
constexpr size_t STATIC_SCRATCH_BYTES = SLAMHIP_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

// Device scratch of one launch group: everything k_prep .. k_finalize write for
// up to `wgroup` walkers.  A context owns one set, carved from a single device
// allocation; the allocation-free entry point (slamhip_chi2_batch_static) carves
// a second one out of the library's static device array.
struct Scratch {
  int wgroup = 0;              // walkers per launch group
  int wgroup_now = 0;          // walkers of the launch group in flight (sizes k_compact's list)
  WalkerK* wk = nullptr;
  double* params = nullptr;
  void* geom = nullptr;        // [walker][surface][sample] {vlos*sqrt(M), weight} as double2 or float2
  unsigned* gflag = nullptr;   // [walker][sample / 64] bit s: surface s of these 64 samples was stored
  float2* grange = nullptr;    // [walker][sample / 64] range of the stored vlos*sqrt(M) of these 64 samples
  uint4* items = nullptr;      // k_plan's list: (walker * tiles + tile, channel block) that can hold emission
  void* iunif = nullptr;       // [walker][nv][nneed][nneed], double or float
  double* partial = nullptr;
  double* stats = nullptr;
  double* tk = nullptr;        // [walker][nvpad] scaled channel table (FitK::tk of a launch)
  double* chi2 = nullptr;
  int* cellx = nullptr;
  int* celly = nullptr;
  double* fracx = nullptr;
  double* fracy = nullptr;
  unsigned char* flags = nullptr;     // [walker][nv][fty][ftx] occupancy of Iunif tiles
  unsigned char* winflag = nullptr;   // [walker][nv][wny][wnx] conv windows with emission
  WinDesc* livelist = nullptr;
  int* livecount = nullptr;
  int* tile_ix = nullptr;
  int* tile_jy = nullptr;
  double* gxw = nullptr;       // [walker][kpad] folded row / column taps
  double* gyw = nullptr;
  float* gxw32 = nullptr;      // the same in single precision (f32 only)
  float* gyw32 = nullptr;
  // Calls that use this set are ordered: `done` is recorded behind the last
  // launch that touched it, and a call on another stream waits for it first.
  hipEvent_t done = nullptr;
  hipStream_t last = nullptr;
  bool used = false;
};

struct slamhip_ctx {
  int device = 0;
  FitK k{};
  std::string err;
  std::vector<void*> allocs;
  int wgroup = 0;          // walkers per launch group of `own`
  Scratch own;             // per-group scratch, one allocation
  Scratch fixed;           // the same layout inside the static device array (wgroup 0: the fit does not fit)
  size_t fixed_need1 = 0;  // bytes one walker of this fit needs there
  size_t max_blocks = 0;   // partial-sum slots per walker
  double* d_stage_p = nullptr;   // slamhip_chi2_batch: parameter / chi2 staging, grown on demand
  double* d_stage_c = nullptr;
  size_t stage_cap = 0;
  double* d_cube = nullptr;      // slamhip_cubemodel: one observed-grid cube, allocated at its first call
  double* d_iconv = nullptr;   // dense convolved cubes: one walker at create, grown on demand (dense mode only)
  int iconv_walkers = 0;
  double* d_obs = nullptr;     // 'mom0*' scalings: [walker][nv][ny][nx] regridded cube
  int2* d_krange = nullptr;    // 'mom0*' scalings: [walker][Iunif tile] lit channel range
  double* d_mom = nullptr;     // 'mom0*' scalings: [walker][n][n] moment of the model on its own grid
  unsigned* d_kmask = nullptr; // 'mom0*' scalings: [walker][Iunif tile][nv / 32] lit channels, one bit each
  unsigned char* d_rowflag = nullptr;   // 'mom0*' scalings: [walker][nv][ny][nx/64] stored row segments
  int* d_oblist = nullptr;              // 'mom0*' scalings: beam windows with emission, count[2]
  int* d_obcount = nullptr;
  int ob_persist = 0;                   // workgroups of k_conv_obs_live
  dim3 ob_grid{1, 1, 1};       // k_conv_obs grid for ONE walker (z = nv)
  int ob_blocks = 0;
  int ob_ty = 0;               // rows per k_conv_obs_sep tile (32 / 16), 0 = direct taps
  int ob_nit = 0;              // row-pass items per thread of that variant
  size_t ob_lds = 0;
  unsigned char* d_obflag = nullptr;  // [walker][nv][tiles] emission under k_conv_obs_sep tiles
  double* d_obempty = nullptr;        // [nv][tiles] residual sum of an empty tile (= sum data^2)
  // the batch path of one-term beams (k_compact + k_conv_rows<OBS> on the observed grid)
  bool ob_rows = false;
  int ob_wnx = 0, ob_wny = 0, ob_ftx = 0, ob_fty = 0;
  unsigned char* d_oflags = nullptr;  // [walker][nv][32x8 tiles of the observed grid] occupancy
  unsigned char* d_owin = nullptr;    // [walker][nv][64x64 windows] emission under the window's halo
  WinDesc* d_odesc = nullptr;         // descriptors of the windows with emission
  int* d_ocount = nullptr;            // LIVECOUNT_WORDS counters / tickets
  double* d_oempty = nullptr;         // [nv][window] sum of data^2 over the window's pixels
  int ob_nty = 0;
  int ftx = 1, fty = 1;
  int persist_blocks = 1;
  int fused = 0;           // hot path uses k_conv_chi2
  dim3 fz_grid{1, 1, 1};   // k_conv_chi2 grid for ONE walker (z = nv)
  int fz_sx = 1, fz_sy = 1;
  int red_blocks = 0;      // k_regrid partials per walker
  dim3 rg_grid{1, 1, 1};   // k_regrid grid for ONE walker (z = nv)
  int separable = 0;
  int conv_mode = 0;
  int kpad = 0;
  int sep_variant = 0;     // dense conv tile: 0: <64,3>  1: <64,5>  2: <32,4>  -1: none fits
  int fz_variant = 0;      // fused kernel's tile (same ladder, sized for the arithmetic type)
  int f32 = 0;             // profile accumulation, Iunif and beam passes in single precision (precision 1, 2)
  int g32 = 0;             // geometry in single precision too (precision 2)
  size_t item = 8;         // bytes of an Iunif / tile element
  size_t gitem = 8;        // bytes of a geometry value
  const float* gx32 = nullptr;   // single-precision copies of the beam taps (f32 only)
  const float* gy32 = nullptr;
  const float* ft32 = nullptr;
  double* d_iunif64 = nullptr;   // f32: one walker's Iunif as double for the parity taps / cubemodel
  int nrank = 1;           // outer-product terms of the factorised beam (0: not factorised)
  int fz_rank = 1;         // what the fused kernel uses in conv mode 0: nrank, or 0 = direct taps
  size_t fz_lds_direct = 0;
  bool direct_fixed = false;   // direct taps run the fixed-pitch instance
  bool sep_fixed = false;      // one-term factorised beam with kpad == 32 runs its fixed-pitch instance
  int fold_ok = 0;             // the fit may fold the regrid into the beam (conv mode 3 switches it off)
  // k_plan's list length of the last launch groups, copied to pinned host memory after each
  // (never waited for): the share of all possible triples that recent groups listed sizes
  // k_profile's one-item-per-workgroup launch for the next group; k_profile_rest covers a
  // list that turns out longer.  -1: no value has landed yet.
  static constexpr int HINT_SLOTS = 64;
  int* h_nit = nullptr;
  size_t h_ub[HINT_SLOTS] = {};
  unsigned h_next = 0;
  int persist_direct = 1;
  size_t sep_lds = 0, c2d_lds = 0, fz_lds = 0;
  hipStream_t stream = nullptr;
  // optional per-kernel HIP-event timing (slamhip_profile_begin/_end)
  bool profiling = false;
  struct Span { int kid; hipEvent_t a, b; };
  std::vector<Span> spans;
};

enum KernelId { KID_PREP = 0, KID_SURFACE, KID_PROFILE, KID_COMPACT, KID_CONV, KID_REGRID, KID_FINALIZE, KID_COUNT };
static const char* const kKernelNames[KID_COUNT] = {
    "k_prep", "k_surface", "k_profile", "k_compact", "k_conv", "k_regrid", "k_finalize"};

// Brackets one launch with events on the launch stream when profiling is on.
struct Timed {
  slamhip_ctx* c; hipStream_t st; hipEvent_t b = nullptr; bool on;
  Timed(slamhip_ctx* c_, int kid, hipStream_t st_) : c(c_), st(st_), on(c_->profiling) {
    if (!on) return;
    hipEvent_t a = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(a, st);
    c->spans.push_back({kid, a, b});
  }
  ~Timed() { if (on) (void)hipEventRecord(b, st); }
};

namespace {

int fail(slamhip_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define HIP_TRY(c, expr)                                                       \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess)                                                      \
      return fail((c), e_ == hipErrorOutOfMemory ? SLAMHIP_ERR_NOMEM           \
                                                 : SLAMHIP_ERR_HIP,            \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                  __FILE__, __LINE__);                                         \
  } while (0)

template <typename T>
int dev_alloc(slamhip_ctx* c, T** p, size_t count) {
  void* q = nullptr;
  HIP_TRY(c, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  c->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return SLAMHIP_OK;
}

template <typename T>
int dev_upload(slamhip_ctx* c, const T** p, const T* host, size_t count) {
  T* q = nullptr;
  int rc = dev_alloc(c, &q, count);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(q, host, count * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return SLAMHIP_OK;
}

// Address of the current device's copy of `scratch` (asked for once per device).
constexpr int MAX_DEVICES = 64;
std::mutex g_static_mu;
unsigned char* g_static_ptr[MAX_DEVICES];

int static_scratch(slamhip_ctx* c, unsigned char** out) {
  if (c->device < 0 || c->device >= MAX_DEVICES)
    return fail(c, SLAMHIP_ERR_ARG, "device index %d beyond %d", c->device, MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_static_mu);
  if (!g_static_ptr[c->device]) {
    void* p = nullptr;
    HIP_TRY(c, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p || (reinterpret_cast<uintptr_t>(p) & 255u))
      return fail(c, SLAMHIP_ERR_HIP, "static scratch: bad device address %p", p);
    g_static_ptr[c->device] = static_cast<unsigned char*>(p);
  }
  *out = g_static_ptr[c->device];
  return SLAMHIP_OK;
}

// Lays a scratch set for W walkers out behind `base` in 256-byte aligned pieces
// and returns the bytes it takes.  base == nullptr: size only; s may be null.
size_t carve(const slamhip_ctx* c, size_t W, unsigned char* base, Scratch* s) {
  const FitK& k = c->k;
  const size_t n = k.nneed;
  const size_t nwin = W * k.nv * c->fz_grid.x * c->fz_grid.y;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    unsigned char* p = base ? base + off : nullptr;
    off += (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
    return p;
  };
  Scratch t;
  t.wk = reinterpret_cast<WalkerK*>(take(W * sizeof(WalkerK)));
  t.params = reinterpret_cast<double*>(take(W * NP * sizeof(double)));
  t.geom = take(W * 4 * (size_t)k.nslot * 2 * c->gitem);
  t.gflag = reinterpret_cast<unsigned*>(take(W * (size_t)(k.nslot / 64) * sizeof(unsigned)));
  t.grange = reinterpret_cast<float2*>(take(W * (size_t)(k.nslot / 64) * sizeof(float2)));
  t.items = reinterpret_cast<uint4*>(take(W * (size_t)c->ftx * c->fty * (k.nvpad / CH) * sizeof(uint4)));
  t.iunif = take(W * k.nv * n * n * c->item);
  t.partial = reinterpret_cast<double*>(take(W * c->max_blocks * 3 * sizeof(double)));
  t.stats = reinterpret_cast<double*>(take(W * 6 * sizeof(double)));
  t.tk = reinterpret_cast<double*>(take(W * (size_t)k.nvpad * sizeof(double)));
  t.chi2 = reinterpret_cast<double*>(take(W * sizeof(double)));
  t.cellx = reinterpret_cast<int*>(take(W * k.nx * sizeof(int)));
  t.fracx = reinterpret_cast<double*>(take(W * k.nx * sizeof(double)));
  t.celly = reinterpret_cast<int*>(take(W * k.ny * sizeof(int)));
  t.fracy = reinterpret_cast<double*>(take(W * k.ny * sizeof(double)));
  t.flags = take(W * k.nv * c->ftx * c->fty + 8);   // +8: k_compact reads rows 8 bytes at a time
  t.winflag = take(nwin);
  t.livelist = reinterpret_cast<WinDesc*>(take(nwin * sizeof(WinDesc)));
  t.livecount = reinterpret_cast<int*>(take(LIVECOUNT_WORDS * sizeof(int)));
  t.tile_ix = reinterpret_cast<int*>(take(W * c->fz_grid.x * 2 * sizeof(int)));
  t.tile_jy = reinterpret_cast<int*>(take(W * c->fz_grid.y * 2 * sizeof(int)));
  t.gxw = reinterpret_cast<double*>(take(W * (size_t)c->kpad * sizeof(double)));
  t.gyw = reinterpret_cast<double*>(take(W * (size_t)c->kpad * sizeof(double)));
  t.gxw32 = reinterpret_cast<float*>(take(W * (size_t)c->kpad * sizeof(float)));
  t.gyw32 = reinterpret_cast<float*>(take(W * (size_t)c->kpad * sizeof(float)));
  if (s && base) {
    t.wgroup = s->wgroup; t.done = s->done; t.last = s->last; t.used = s->used;
    *s = t;
  }
  return off;
}

// Calls that share a scratch set are ordered on the device: a call on another
// stream than the set's last user first waits for that user's last launch.
int scratch_acquire(slamhip_ctx* c, Scratch& S, hipStream_t st) {
  if (S.done && S.used && S.last != st) HIP_TRY(c, hipStreamWaitEvent(st, S.done, 0));
  return SLAMHIP_OK;
}
int scratch_release(slamhip_ctx* c, Scratch& S, hipStream_t st) {
  if (S.done) {
    HIP_TRY(c, hipEventRecord(S.done, st));
    S.last = st;
    S.used = true;
  }
  return SLAMHIP_OK;
}

// One-sided Jacobi SVD of the K x K beam: B = sum_r sig[r] * U[:,r] (x) V[:,r],
// singular values in decreasing order.  K <= ~100, run once per fit.
void beam_svd(const double* B, int K, std::vector<double>& U, std::vector<double>& V,
              std::vector<double>& sig) {
  std::vector<double> A(B, B + (size_t)K * K);      // columns get orthogonalised
  V.assign((size_t)K * K, 0.0);
  for (int i = 0; i < K; ++i) V[(size_t)i * K + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < K - 1; ++p)
      for (int q = p + 1; q < K; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < K; ++i) {
          const double ap = A[(size_t)i * K + p], aq = A[(size_t)i * K + q];
          alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
        }
        if (gamma == 0.0 || alpha * beta == 0.0) continue;
        off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < K; ++i) {
          const double ap = A[(size_t)i * K + p], aq = A[(size_t)i * K + q];
          A[(size_t)i * K + p] = cs * ap - sn * aq;
          A[(size_t)i * K + q] = sn * ap + cs * aq;
          const double vp = V[(size_t)i * K + p], vq = V[(size_t)i * K + q];
          V[(size_t)i * K + p] = cs * vp - sn * vq;
          V[(size_t)i * K + q] = sn * vp + cs * vq;
        }
      }
    if (off < 1e-15) break;
  }
  std::vector<int> order(K);
  std::vector<double> nrm(K);
  for (int j = 0; j < K; ++j) {
    double s2 = 0;
    for (int i = 0; i < K; ++i) s2 += A[(size_t)i * K + j] * A[(size_t)i * K + j];
    nrm[j] = std::sqrt(s2);
    order[j] = j;
  }
  std::sort(order.begin(), order.end(), [&](int a, int b) { return nrm[a] > nrm[b]; });
  U.assign((size_t)K * K, 0.0);
  std::vector<double> Vs((size_t)K * K);
  sig.assign(K, 0.0);
  for (int r = 0; r < K; ++r) {
    const int j = order[r];
    sig[r] = nrm[j];
    for (int i = 0; i < K; ++i) {
      U[(size_t)i * K + r] = nrm[j] > 0 ? A[(size_t)i * K + j] / nrm[j] : 0.0;
      Vs[(size_t)i * K + r] = V[(size_t)i * K + j];
    }
  }
  V.swap(Vs);
}

// Samples that survive the layer merge (:469-472), listed per output pixel.
// A pixel of layer l inside the inner quadrant [nq1,nq3)^2 is replaced by the
// mean of its 2x2 children in layer l+1 (recursively), in avefour's order.
void collect_leaves(const slamhip_desc& d, int l, int j, int i,
                    std::vector<unsigned>& out) {
  const bool inner = j >= d.nq1 && j < d.nq3 && i >= d.nq1 && i < d.nq3;
  if (l + 1 < d.nlayer && inner) {
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        collect_leaves(d, l + 1, 2 * (j - d.nq1) + a, 2 * (i - d.nq1) + b, out);
  } else {
    out.push_back(((unsigned)l << 28) | ((unsigned)j << 14) | (unsigned)i);
  }
}

int check_desc(const slamhip_desc* d) {
  if (!d) return 1;
  if (d->nv < 1 || d->ny < 1 || d->nx < 1) return 2;
  if (d->nlayer < 1 || d->nlayer > 8) return 3;
  if (d->npix < 4 || d->npix > 16383) return 4;
  if (d->nq1 < 0 || d->nq3 > d->npix || d->nq3 - d->nq1 != d->npix / 2) return 5;
  if (d->ineed0 < 0 || d->ineed1 > d->npix || d->ineed1 - d->ineed0 < 2) return 6;
  if (d->kbeam < 1 || d->kbeam % 2 == 0) return 7;
  if (!d->Xnest0 || !d->Ynest0 || !d->xneed || !d->yneed || !d->xobs ||
      !d->yobs || !d->v_valid || !d->data_valid || !d->gaussbeam)
    return 8;
  if (!(d->dv != 0.0) || !(d->sigma > 0.0) || !(d->pixperbeam > 0.0)) return 9;
  // every child index 2*(j-nq1)+a must stay inside the next layer
  if (2 * (d->nq3 - 1 - d->nq1) + 1 >= d->npix) return 10;
  return 0;
}

template <int TY, int MAXR>
void launch_conv_sep(slamhip_ctx* c, const double* in, double* out, int planes,
                     hipStream_t st) {
  const int n = c->k.nneed;
  dim3 grid((n + CONV_TX - 1) / CONV_TX, (n + TY - 1) / TY, planes);
  hipLaunchKernelGGL((k_conv_sep<TY, MAXR>), grid, dim3(256), c->sep_lds, st,
                     c->k, in, out, c->kpad);
}

size_t sep_lds_bytes(int TY, int kpad) {
  const int rows = TY + kpad - 1;
  const int P = (CONV_TX + kpad - 1) | 1;
  return (size_t)rows * std::max(P, CONV_TX + 1) * sizeof(double);
}

// LDS tile of the fused kernel: pitch cols + 2 (odd; room for the staging's column
// shift and a pad column) and one leading element.
size_t fz_tile_bytes(int TY, int kpad) {
  const size_t rows = TY + kpad - 1, P = ((size_t)CONV_TX + kpad) | 1;
  return (rows * P + 2) * sizeof(double);
}

int run_conv(slamhip_ctx* c, const double* in, double* out, int planes,
             hipStream_t st) {
  const int n = c->k.nneed;
  const bool sep = c->separable && c->conv_mode != 1 && c->sep_variant >= 0;
  // grid.z is limited to 65535 planes per launch
  for (int p0 = 0; p0 < planes; p0 += 32768) {
    const int np_ = std::min(32768, planes - p0);
    const double* src = in + (long long)p0 * n * n;
    double* dst = out + (long long)p0 * n * n;
    Timed t_(c, KID_CONV, st);
    if (sep) {
      switch (c->sep_variant) {
        case 0: launch_conv_sep<64, 3>(c, src, dst, np_, st); break;
        case 1: launch_conv_sep<64, 5>(c, src, dst, np_, st); break;
        default: launch_conv_sep<32, 4>(c, src, dst, np_, st); break;
      }
    } else {
      if (c->c2d_lds > 160 * 1024)
        return fail(c, SLAMHIP_ERR_ARG, "beam kernel %d too large for the 2-D LDS tile",
                    c->k.kbeam);
      dim3 grid((n + C2_TX - 1) / C2_TX, (n + C2_TY - 1) / C2_TY, np_);
      hipLaunchKernelGGL(k_conv_2d, grid, dim3(256), c->c2d_lds, st, c->k, src, dst);
    }
  }
  HIP_TRY(c, hipGetLastError());
  return SLAMHIP_OK;
}

// The dense (non-fused) path keeps whole convolved cubes; the hot path never
// does, so that buffer only grows when the dense path is asked for a batch.
int ensure_iconv(slamhip_ctx* c, int nw) {
  if (nw <= c->iconv_walkers) return SLAMHIP_OK;
  HIP_TRY(c, hipDeviceSynchronize());            // the old buffer may still be read
  if (c->d_iconv) {
    (void)hipFree(c->d_iconv);
    c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), (void*)c->d_iconv), c->allocs.end());
    c->d_iconv = nullptr;
    c->iconv_walkers = 0;
  }
  const size_t per = (size_t)c->k.nv * c->k.nneed * c->k.nneed;
  int rc = dev_alloc(c, &c->d_iconv, per * nw);
  if (rc) return rc;
  c->iconv_walkers = nw;
  return SLAMHIP_OK;
}

// Stages shared by every entry point: params (device) -> iunif for `nw` walkers
// `dense`: the consumer reads Iunif without the occupancy flags, so tiles that
// k_profile skips must already hold zeros.
int run_model(slamhip_ctx* c, Scratch& S, const double* d_params, int nw, hipStream_t st, bool dense) {
  FitK k = c->k;
  k.tk = S.tk;
  if (dense)
    HIP_TRY(c, hipMemsetAsync(S.iunif, 0, c->item * (size_t)nw * k.nv * k.nneed * k.nneed, st));
  {
    Timed t_(c, KID_PREP, st);
    hipLaunchKernelGGL(k_prep, dim3(nw), dim3(256), 0, st, k, d_params, nw, S.wk,
                       S.cellx, S.fracx, S.celly, S.fracy, S.tile_ix,
                       S.tile_jy, (int)c->fz_grid.x, (int)c->fz_grid.y, c->fz_sx, c->fz_sy,
                       S.winflag, (int)(k.nv * c->fz_grid.x * c->fz_grid.y), S.livecount,
                       S.gxw, S.gyw, c->f32 ? S.gxw32 : (float*)nullptr, c->f32 ? S.gyw32 : (float*)nullptr);
  }
  {
    Timed t_(c, KID_SURFACE, st);
    dim3 g1((unsigned)((k.nslot + 255) / 256), 1, (nw + SURF_WPT - 1) / SURF_WPT);
    if (c->g32)
      hipLaunchKernelGGL(k_surface<float>, g1, dim3(256), 0, st, k, S.wk, (float2*)S.geom, S.gflag, S.grange, nw);
    else
      hipLaunchKernelGGL(k_surface<double>, g1, dim3(256), 0, st, k, S.wk, (double2*)S.geom, S.gflag, S.grange, nw);
  }
  {
    Timed t_(c, KID_PROFILE, st);
    const int ntile = c->ftx * c->fty;
    // occupancy bytes start at zero; k_profile raises those of the blocks it lights
    HIP_TRY(c, hipMemsetAsync(S.flags, 0, (size_t)nw * k.nv * ntile, st));
    int* nitems = S.livecount + 2;               // zeroed by k_prep with the other counters
    hipLaunchKernelGGL(k_plan, dim3((unsigned)((nw * ntile + 255) / 256)), dim3(256), 0, st, k, S.wk,
                       S.grange, nw, ntile, S.items, nitems);
    // One workgroup per item for as large a share of all possible triples as recent groups
    // listed (x 1.5, + 2 %; everything while no count has come back), in eighths for the
    // XCD mapping; a fixed grid for whatever a longer list leaves.
    const size_t every = (size_t)nw * ntile * (k.nvpad / CH);
    double share = -1.0;
    for (int i = 0; i < slamhip_ctx::HINT_SLOTS; ++i) {
      const int got = reinterpret_cast<volatile int*>(c->h_nit)[i];
      if (got >= 0 && c->h_ub[i]) share = std::max(share, (double)got / (double)c->h_ub[i]);
    }
    share = share < 0.0 ? 1.0 : std::min(1.0, share * 1.5 + 0.02);
    const int done8 = (int)std::max<size_t>(((size_t)((double)every * share) + 7) / 8, 1);
    dim3 g2((unsigned)done8 * 8u, 1, 1), g3(2048, 1, 1);
    {
      const unsigned slot = c->h_next++ % slamhip_ctx::HINT_SLOTS;
      c->h_ub[slot] = every;
      HIP_TRY(c, hipMemcpyAsync(&c->h_nit[slot], nitems, sizeof(int), hipMemcpyDeviceToHost, st));
    }
#define SLAM_PROFILE(FAST_, T_, TG_, T2_)                                                         \
    do {                                                                                          \
    hipLaunchKernelGGL((k_profile<FAST_, T_, TG_>), g2, dim3(256), 0, st, k, S.wk, (const T2_*)S.geom, \
                       S.gflag, (T_*)S.iunif, S.flags, c->ftx, c->fty, S.winflag, (int)c->fz_grid.x,  \
                       (int)c->fz_grid.y, c->fz_sx, c->fz_sy, S.items, nitems);                      \
    hipLaunchKernelGGL((k_profile_rest<FAST_, T_, TG_>), g3, dim3(256), 0, st, k, S.wk, (const T2_*)S.geom, \
                       S.gflag, (T_*)S.iunif, S.flags, c->ftx, c->fty, S.winflag, (int)c->fz_grid.x,  \
                       (int)c->fz_grid.y, c->fz_sx, c->fz_sy, S.items, nitems, done8);               \
    } while (0)
    if (c->g32)      { if (k.fast_idx) SLAM_PROFILE(true, float, float, float2); else SLAM_PROFILE(false, float, float, float2); }
    else if (c->f32) { if (k.fast_idx) SLAM_PROFILE(true, float, double, double2); else SLAM_PROFILE(false, float, double, double2); }
    else             { if (k.fast_idx) SLAM_PROFILE(true, double, double, double2); else SLAM_PROFILE(false, double, double, double2); }
#undef SLAM_PROFILE
  }
  HIP_TRY(c, hipGetLastError());
  return SLAMHIP_OK;
}

// k_compact's grid: a thread per window of the launch group's walkers, the walker count
// rounded up to whole blocks of CV_WB (its list order)
static unsigned compact_blocks(int nwin, int nw) {
  long long threads = nwin;
  if (CV_WB > 0) threads = (long long)(nwin / nw) * ((nw + CV_WB - 1) / CV_WB) * CV_WB;
  return (unsigned)((threads + 255) / 256);
}

// 'mom0*' scalings: Iunif -> observed grid -> per-pixel moment-0 scale -> beam on
// the observed grid -> residual sum.  `d_model` (may be null) receives the
// model cubes [nw][nv][ny][nx].
int run_group_mom0(slamhip_ctx* c, Scratch& S, const double* d_params, int nw, double* d_chi2,
                   double* d_model, bool convolving, hipStream_t st) {
  const FitK& k = c->k;
  int rc = run_model(c, S, d_params, nw, st, false);
  if (rc) return rc;
  {
    Timed t_(c, KID_REGRID, st);
    const long long npix = (long long)k.ny * k.nx;
    const bool flagged = c->ob_ty && c->conv_mode != 1 && convolving;
    const bool rows_path = flagged && c->ob_rows && !d_model && c->conv_mode != 4;
    if (flagged && !rows_path)
      HIP_TRY(c, hipMemsetAsync(c->d_obflag, 0, (size_t)nw * k.nv * c->ob_grid.x * c->ob_nty, st));
    if (rows_path) {
      HIP_TRY(c, hipMemsetAsync(c->d_oflags, 0, (size_t)nw * k.nv * c->ob_ftx * c->ob_fty, st));
      HIP_TRY(c, hipMemsetAsync(c->d_owin, 0, (size_t)nw * k.nv * c->ob_wnx * c->ob_wny, st));
    }
    const int ntile = c->ftx * c->fty;
    hipLaunchKernelGGL(k_tile_krange, dim3((ntile + 255) / 256, nw), dim3(256), 0, st, k, S.flags,
                       ntile, c->d_krange, c->d_kmask);
    hipLaunchKernelGGL(k_mom_model, dim3(ntile, nw), dim3(256), 0, st, k, (const double*)S.iunif, S.flags,
                       c->d_krange, c->ftx, c->fty, c->d_mom);
    const int nseg = (k.nx + 63) / 64;
    if (flagged)
      HIP_TRY(c, hipMemsetAsync(c->d_rowflag, 0, (size_t)nw * k.nv * k.ny * nseg, st));
    hipLaunchKernelGGL(k_regrid_mom0, dim3(nseg, (k.ny + 3) / 4, nw), dim3(256), 0, st, k,
                       S.cellx, S.fracx, S.celly, S.fracy, (const double*)S.iunif, S.flags,
                       c->d_krange, c->ftx, c->fty, c->d_obs, (flagged && !rows_path) ? c->d_obflag : (unsigned char*)nullptr,
                       c->ob_ty ? c->ob_ty : 1, (int)c->ob_grid.x, c->ob_nty,
                       flagged ? c->d_rowflag : (unsigned char*)nullptr,
                       rows_path ? c->d_oflags : (unsigned char*)nullptr, c->ob_ftx, c->ob_fty,
                       rows_path ? c->d_owin : (unsigned char*)nullptr, c->ob_wnx, c->ob_wny,
                       (const double*)c->d_mom, (const unsigned*)c->d_kmask);
  }
  if (!convolving) {       // peaktounity on the observed grid (:531-532), single-vector product
    hipLaunchKernelGGL(k_peak_unity, dim3(k.nv * nw), dim3(256), 0, st, k, c->d_obs, d_model,
                       (long long)k.ny * k.nx);
    HIP_TRY(c, hipGetLastError());
    return SLAMHIP_OK;
  }
  if (c->ob_ty && c->ob_rows && !d_model && c->conv_mode != 1 && c->conv_mode != 4) {
    // batch path, one-term beam of 25..32 padded taps, observed rows of even length: the windows with
    // emission listed with their tile bitmaps (k_compact, as for the 'uniform' scaling), then
    // the row-owning beam kernel on them; conv mode 4 keeps the older tile kernel below
    const int nwin = nw * k.nv * c->ob_wnx * c->ob_wny;
    FitK ko = k;
    ko.nneed = k.nx;                                        // the "model grid" of these kernels is the observed one
    {
      Timed t_(c, KID_COMPACT, st);
      HIP_TRY(c, hipMemsetAsync(c->d_ocount, 0, LIVECOUNT_WORDS * sizeof(int), st));
      hipLaunchKernelGGL(k_compact, dim3(compact_blocks(nwin, nw)), dim3(256), 0, st, ko, c->d_owin, nwin,
                         c->d_oflags, c->ob_ftx, c->ob_fty, (const int*)nullptr, (const int*)nullptr,
                         c->ob_wnx, c->ob_wny, CONV_TX, 64, 64, c->d_odesc, c->d_ocount, S.wk, 0, nw, k.ny);
    }
    {
      Timed t_(c, KID_CONV, st);
      const int tail4 = k.kbeam <= c->kpad - CONV_R / 2 ? 1 : 0;
      hipLaunchKernelGGL((k_conv_rows<double, false, 64, 32, true>), dim3(c->persist_blocks), dim3(CV_NT), c->fz_lds,
                         st, ko, (const double*)c->d_obs, S.cellx, S.fracx, S.celly, S.fracy, c->d_odesc,
                         c->d_ocount, c->ob_wnx, c->ob_wny, S.partial, k.gx_pad, k.gy_pad, k.gx_pad, k.gy_pad,
                         tail4, 0, c->d_rowflag);
    }
    Timed t2_(c, KID_FINALIZE, st);
    hipLaunchKernelGGL(k_sum_chi2, dim3(nw), dim3(256), 0, st, k, S.partial, k.nv * c->ob_wnx * c->ob_wny,
                       S.stats, d_chi2, c->d_owin, c->d_oempty);
    HIP_TRY(c, hipGetLastError());
    return SLAMHIP_OK;
  }
  if (c->ob_ty && c->conv_mode != 1) {
    Timed t_(c, KID_CONV, st);
    dim3 g(c->ob_grid.x, (k.ny + c->ob_ty - 1) / c->ob_ty, c->ob_grid.z * nw);
    if (!d_model && c->d_oblist) {
      // batch path: list the windows with emission, then persistent workgroups
      const long long nwin = (long long)g.x * g.y * g.z;
      HIP_TRY(c, hipMemsetAsync(c->d_obcount, 0, 2 * sizeof(int), st));
      hipLaunchKernelGGL(k_obs_list, dim3((unsigned)((nwin + 255) / 256)), dim3(256), 0, st, k,
                         c->d_obflag, c->d_obempty, S.partial, c->d_oblist, c->d_obcount,
                         (int)g.x, (int)g.y, c->ob_ty, nwin);
#define SLAM_OBL(TY_, NIT_)                                                                   \
      hipLaunchKernelGGL((k_conv_obs_live<TY_, NIT_>), dim3(c->ob_persist), dim3(256), c->ob_lds, st, k, \
                         c->d_obs, S.partial, c->kpad, c->nrank, k.gx_pad, k.gy_pad, c->d_rowflag, \
                         c->d_oblist, c->d_obcount, (int)g.x, (int)g.y)
      if (c->ob_ty == 32 && c->ob_nit == 2) SLAM_OBL(32, 2);
      else if (c->ob_ty == 32) SLAM_OBL(32, 4);
      else SLAM_OBL(16, 4);
#undef SLAM_OBL
    } else {
#define SLAM_OBS(TY_, NIT_, WR_)                                                            \
    hipLaunchKernelGGL((k_conv_obs_sep<TY_, NIT_, WR_>), g, dim3(256), c->ob_lds, st, k, c->d_obs, \
                       d_model, S.partial, c->kpad, c->nrank, c->d_obflag, c->d_obempty, k.gx_pad, k.gy_pad, \
                       c->d_rowflag)
    if (c->ob_ty == 32 && c->ob_nit == 2) { if (d_model) SLAM_OBS(32, 2, true); else SLAM_OBS(32, 2, false); }
    else if (c->ob_ty == 32)              { if (d_model) SLAM_OBS(32, 4, true); else SLAM_OBS(32, 4, false); }
    else                                  { if (d_model) SLAM_OBS(16, 4, true); else SLAM_OBS(16, 4, false); }
#undef SLAM_OBS
    }
    const int nblk = (int)(g.x * g.y) * k.nv;
    Timed t2_(c, KID_FINALIZE, st);
    hipLaunchKernelGGL(k_sum_chi2, dim3(nw), dim3(256), 0, st, k, S.partial, nblk,
                       S.stats, d_chi2);
    HIP_TRY(c, hipGetLastError());
    return SLAMHIP_OK;
  }
  if (c->c2d_lds > 160 * 1024)
    return fail(c, SLAMHIP_ERR_ARG, "beam kernel %d too large for the 2-D LDS tile", k.kbeam);
  {
    Timed t_(c, KID_CONV, st);
    dim3 g(c->ob_grid.x, c->ob_grid.y, c->ob_grid.z * nw);
    if (d_model)
      hipLaunchKernelGGL((k_conv_obs<true>), g, dim3(256), c->c2d_lds, st, k, c->d_obs, d_model,
                         S.partial);
    else
      hipLaunchKernelGGL((k_conv_obs<false>), g, dim3(256), c->c2d_lds, st, k, c->d_obs,
                         (double*)nullptr, S.partial);
  }
  {
    Timed t_(c, KID_FINALIZE, st);
    hipLaunchKernelGGL(k_sum_chi2, dim3(nw), dim3(256), 0, st, k, S.partial, c->ob_blocks,
                       S.stats, d_chi2);
  }
  HIP_TRY(c, hipGetLastError());
  return SLAMHIP_OK;
}

// The fused kernel's instance for the context's tile variant, tap form and
// arithmetic type T (its address for hipFuncSetAttribute, or a launch).
template <typename T>
const void* fused_instance(int variant, bool direct, bool fixed, bool multi = true) {
  if (!direct && !multi && fixed) return (const void*)k_conv_chi2<64, 2, false, SEP_PITCH, T, false>;
  if (!direct && !multi)
    return variant == 0 ? (const void*)k_conv_chi2<64, 2, false, 0, T, false>
         : variant == 1 ? (const void*)k_conv_chi2<64, 3, false, 0, T, false>
         : variant == 3 ? (const void*)k_conv_chi2<48, 2, false, 0, T, false>
                        : (const void*)k_conv_chi2<32, 2, false, 0, T, false>;
  if (!direct)
    return variant == 0 ? (const void*)k_conv_chi2<64, 2, false, 0, T>
         : variant == 1 ? (const void*)k_conv_chi2<64, 3, false, 0, T>
                        : (const void*)k_conv_chi2<32, 2, false, 0, T>;
  if (fixed) return (const void*)k_conv_chi2<64, 2, true, DIRECT_PITCH, T>;
  return variant == 0 ? (const void*)k_conv_chi2<64, 2, true, 0, T>
       : variant == 1 ? (const void*)k_conv_chi2<64, 3, true, 0, T>
                      : (const void*)k_conv_chi2<32, 2, true, 0, T>;
}

// the row-owning kernel's cases, both one-term beams in double: 25..32 padded taps on 64-row
// windows, or 80 padded taps of a K <= 77 beam on 48-row windows (its 32 row groups hold every row
// with image data)
static bool rows_kernel(const slamhip_ctx* c, int fz_rank) {
  if (fz_rank != 1 || c->conv_mode == 4) return false;
  return (c->fz_variant == 0 && c->sep_fixed) ||
         (c->fz_variant == 3 && c->kpad == 80 && c->k.kbeam <= 77);
}

template <typename T>
void launch_fused(slamhip_ctx* c, Scratch& S, int fz_rank, hipStream_t st, const T* gx, const T* gy,
                  const T* ft, const T* gxw, const T* gyw) {
  dim3 gz(fz_rank ? c->persist_blocks : c->persist_direct, 1, 1);
  const size_t fz_lds = fz_rank ? c->fz_lds : c->fz_lds_direct;
  // the taps any window of this launch uses: K, or K + 1 where the regrid is folded in
  const int ntap = c->k.kbeam + ((fz_rank == 1 && c->k.fold) ? 1 : 0);
  const int tail4 = ntap <= c->kpad - CONV_R / 2 ? 1 : 0;
#define SLAM_FUSED_M(TY_, MR_, DIR_, PITCH_, MULTI_)                                               \
  hipLaunchKernelGGL((k_conv_chi2<TY_, MR_, DIR_, PITCH_, T, MULTI_>), gz, dim3(CV_NT), fz_lds, st, \
                     c->k, (const T*)S.iunif, S.cellx, S.fracx, S.celly, S.fracy, S.livelist,       \
                     S.livecount, (int)c->fz_grid.x, (int)c->fz_grid.y, S.partial, c->kpad,         \
                     fz_rank, gx, gy, ft, gxw, gyw, tail4)
#define SLAM_FUSED(TY_, MR_, DIR_, PITCH_) SLAM_FUSED_M(TY_, MR_, DIR_, PITCH_, true)
#define SLAM_FUSED_SEP(TY_, MR_) \
  do { if (fz_rank == 1) SLAM_FUSED_M(TY_, MR_, false, 0, false); else SLAM_FUSED(TY_, MR_, false, 0); } while (0)
  switch (c->fz_variant) {
    case 0:
      if (rows_kernel(c, fz_rank)) {
        // folded walkers' windows, then (second list, usually empty: the workgroups leave at once)
        // those that keep the conv tile + bilinear regrid
        const int nwin = (int)(c->fz_grid.x * c->fz_grid.y * c->fz_grid.z) * S.wgroup_now;
        hipLaunchKernelGGL((k_conv_rows<T, false>), gz, dim3(CV_NT), fz_lds, st, c->k, (const T*)S.iunif, S.cellx,
                           S.fracx, S.celly, S.fracy, S.livelist, S.livecount, (int)c->fz_grid.x,
                           (int)c->fz_grid.y, S.partial, gx, gy, gxw, gyw, tail4, 0);
        hipLaunchKernelGGL((k_conv_rows<T, true>), gz, dim3(CV_NT), fz_lds, st, c->k, (const T*)S.iunif, S.cellx,
                           S.fracx, S.celly, S.fracy, S.livelist, S.livecount + 3, (int)c->fz_grid.x,
                           (int)c->fz_grid.y, S.partial, gx, gy, gxw, gyw, tail4, nwin);
      } else if (fz_rank == 1 && c->sep_fixed) SLAM_FUSED_M(64, 2, false, SEP_PITCH, false);
      else if (fz_rank) SLAM_FUSED_SEP(64, 2);
      else if (c->direct_fixed) SLAM_FUSED(64, 2, true, DIRECT_PITCH);
      else SLAM_FUSED(64, 2, true, 0);
      break;
    case 1: if (fz_rank) SLAM_FUSED_SEP(64, 3); else SLAM_FUSED(64, 3, true, 0); break;
    case 3:                                                   // one-term beams only (slamhip_create)
      if (rows_kernel(c, fz_rank)) {
        {
          const int nwin = (int)(c->fz_grid.x * c->fz_grid.y * c->fz_grid.z) * S.wgroup_now;
          hipLaunchKernelGGL((k_conv_rows<T, false, 48, 80>), gz, dim3(CV_NT), fz_lds, st, c->k, (const T*)S.iunif,
                             S.cellx, S.fracx, S.celly, S.fracy, S.livelist, S.livecount, (int)c->fz_grid.x,
                             (int)c->fz_grid.y, S.partial, gx, gy, gxw, gyw, tail4, 0);
          hipLaunchKernelGGL((k_conv_rows<T, true, 48, 80>), gz, dim3(CV_NT), fz_lds, st, c->k, (const T*)S.iunif,
                             S.cellx, S.fracx, S.celly, S.fracy, S.livelist, S.livecount + 3, (int)c->fz_grid.x,
                             (int)c->fz_grid.y, S.partial, gx, gy, gxw, gyw, tail4, nwin);
        }
      } else SLAM_FUSED_M(48, 2, false, 0, false);
      break;
    default: if (fz_rank) SLAM_FUSED_SEP(32, 2); else SLAM_FUSED(32, 2, true, 0); break;
  }
#undef SLAM_FUSED_SEP
#undef SLAM_FUSED
#undef SLAM_FUSED_M
}

int run_group(slamhip_ctx* c, Scratch& S, const double* d_params, int nw, double* d_chi2,
              hipStream_t st) {
  if (c->k.mom0decon) return run_group_mom0(c, S, d_params, nw, d_chi2, nullptr, true, st);
  const bool fused = c->fused && c->conv_mode != 1;
  // conv mode 2 keeps the factorised beam where mode 0 would take direct taps
  const int fz_rank = (c->conv_mode == 2 && c->nrank >= 1) ? c->nrank : c->fz_rank;
  int rc = run_model(c, S, d_params, nw, st, !fused);
  if (rc) return rc;
  if (fused) {
    const int nwin = (int)(c->fz_grid.x * c->fz_grid.y * c->fz_grid.z) * nw;
    {
      Timed t_(c, KID_COMPACT, st);
      hipLaunchKernelGGL(k_compact, dim3(compact_blocks(nwin, nw)), dim3(256), 0, st, c->k, S.winflag, nwin,
                         S.flags, c->ftx, c->fty, S.tile_ix, S.tile_jy,
                         (int)c->fz_grid.x, (int)c->fz_grid.y, c->fz_sx, c->fz_sy, c->fz_sy + 1,
                         S.livelist, S.livecount, S.wk, rows_kernel(c, fz_rank) ? 1 : 0, nw, 0);
      S.wgroup_now = nw;
    }
    {
      Timed t_(c, KID_CONV, st);
      if (c->f32)
        launch_fused<float>(c, S, fz_rank, st, c->gx32, c->gy32, c->ft32, S.gxw32, S.gyw32);
      else
        launch_fused<double>(c, S, fz_rank, st, c->k.gx_pad, c->k.gy_pad, c->k.beam_ft, S.gxw, S.gyw);
    }
    {
      Timed t_(c, KID_FINALIZE, st);
      hipLaunchKernelGGL(k_finalize, dim3(nw), dim3(256), 0, st, c->k, S.partial,
                         (int)(c->fz_grid.x * c->fz_grid.y * c->fz_grid.z), S.winflag,
                         S.stats, d_chi2);
    }
    HIP_TRY(c, hipGetLastError());
    return SLAMHIP_OK;
  }
  if (nw > c->iconv_walkers)                  // group_cap() keeps dense groups within the buffer
    return fail(c, SLAMHIP_ERR_STATE, "dense conv buffer holds %d walkers, group has %d", c->iconv_walkers, nw);
  rc = run_conv(c, (const double*)S.iunif, c->d_iconv, nw * c->k.nv, st);
  if (rc) return rc;
  {
    Timed t_(c, KID_REGRID, st);
    dim3 g4(c->rg_grid.x, c->rg_grid.y, c->rg_grid.z * nw);
    hipLaunchKernelGGL((k_regrid<false>), g4, dim3(RED_THREADS), 0, st, c->k, S.cellx,
                       S.fracx, S.celly, S.fracy, c->d_iconv, S.partial,
                       (double*)nullptr);
  }
  {
    Timed t_(c, KID_FINALIZE, st);
    hipLaunchKernelGGL(k_finalize, dim3(nw), dim3(256), 0, st, c->k, S.partial,
                       c->red_blocks, (const unsigned char*)nullptr, S.stats, d_chi2);
  }
  HIP_TRY(c, hipGetLastError());
  return SLAMHIP_OK;
}

}  // namespace

extern "C" {

int slamhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// The build (slam_amd/build.py) passes the hash of this file + include/slamhip.h as
// SLAM_SRC_SHA; the string also says whether this is the diagnostic phase-stamp build
// (its own file, libslamhip_stamps.so: never the product).
#ifndef SLAM_SRC_SHA
#define SLAM_SRC_SHA "unknown"
#endif
#ifdef SLAMHIP_PHASE_STAMPS
#define SLAMHIP_STAMPS_STATE "1"
#else
#define SLAMHIP_STAMPS_STATE "0"
#endif
#ifndef SLAM_VARIANT
#define SLAM_VARIANT "product"
#endif
const char* slamhip_version(void) {
  return "slamhip 0.3 (gfx950, fp64 / fp32) src=" SLAM_SRC_SHA " stamps=" SLAMHIP_STAMPS_STATE
         " variant=" SLAM_VARIANT;
}

const char* slamhip_last_error(slamhip_ctx* ctx) {
  return ctx ? ctx->err.c_str() : "null context";
}

int slamhip_create(const slamhip_desc* d, slamhip_ctx** out) {
  if (!out) return SLAMHIP_ERR_ARG;
  *out = nullptr;
  slamhip_ctx* c = new slamhip_ctx();
  *out = c;   // returned even on failure so the caller can read the error
  if (int bad = check_desc(d))
    return fail(c, SLAMHIP_ERR_ARG, "invalid descriptor (check %d)", bad);
  HIP_TRY(c, hipGetDevice(&c->device));
  if (!(d->beam_tol >= 0.0 && d->beam_tol <= 1e-3))
    return fail(c, SLAMHIP_ERR_ARG, "beam_tol %g outside [0, 1e-3]", d->beam_tol);
  if (d->precision < 0 || d->precision > 2)
    return fail(c, SLAMHIP_ERR_ARG, "precision %d: 0 (double), 1 (float) or 2 (float, geometry too)", d->precision);
  if (d->precision != 0 && d->mom0decon)
    return fail(c, SLAMHIP_ERR_ARG, "precision 1 / 2 (float) cover scaling 'uniform' only");
  c->f32 = d->precision >= 1;
  c->g32 = d->precision == 2;
  c->item = c->f32 ? sizeof(float) : sizeof(double);
  c->gitem = c->g32 ? sizeof(float) : sizeof(double);
  FitK& k = c->k;
  k.nv = d->nv; k.ny = d->ny; k.nx = d->nx; k.nlayer = d->nlayer; k.npix = d->npix;
  k.nq1 = d->nq1; k.nq3 = d->nq3; k.ineed0 = d->ineed0; k.ineed1 = d->ineed1;
  k.nneed = d->ineed1 - d->ineed0; k.kbeam = d->kbeam; k.ngauss = (d->kbeam - 1) / 2;
  k.disk = d->disk != 0; k.envelope = d->envelope != 0;
  k.dv = d->dv; k.sigma = d->sigma; k.pixperbeam = d->pixperbeam; k.incl0 = d->incl0;
  k.signmajor = d->signmajor; k.signminor = d->signminor; k.vunit = d->vunit;
  k.nvpad = (k.nv + CH - 1) / CH * CH;
  const int n = k.nneed, K = k.kbeam;
  const size_t ngrid = (size_t)k.nlayer * k.npix * k.npix;
  const size_t ncube = (size_t)k.nv * k.ny * k.nx;

  // --- sample list ---------------------------------------------------------
  // Samples that survive the layer merge, grouped per 32x8 pixel tile as
  // [sample rank][pixel in tile] and padded to the tile's largest count, so
  // that the lanes of k_profile (= pixels) read consecutive slots for a rank.
  const int ftx = (n + PT_X - 1) / PT_X, fty = (n + PT_Y - 1) / PT_Y;
  std::vector<unsigned> leaves;
  std::vector<int> tile_goff((size_t)ftx * fty), tile_nl((size_t)ftx * fty);
  long long nleaf = 0;
  {
    std::vector<unsigned> tmp;
    std::vector<std::vector<unsigned>> per(PT_X * PT_Y);
    for (int ty = 0; ty < fty; ++ty)
      for (int tx = 0; tx < ftx; ++tx) {
        int nl = 0;
        for (int p = 0; p < PT_X * PT_Y; ++p) {
          per[p].clear();
          const int i = tx * PT_X + (p % PT_X), j = ty * PT_Y + p / PT_X;
          if (i < n && j < n) collect_leaves(*d, 0, j + k.ineed0, i + k.ineed0, per[p]);
          nl = std::max(nl, (int)per[p].size());
          nleaf += (long long)per[p].size();
        }
        tile_goff[(size_t)ty * ftx + tx] = (int)leaves.size();
        tile_nl[(size_t)ty * ftx + tx] = nl;
        for (int r = 0; r < nl; ++r)
          for (int p = 0; p < PT_X * PT_Y; ++p)
            leaves.push_back(r < (int)per[p].size() ? per[p][r] : LEAF_PAD);
      }
  }
  k.nleaf = nleaf;
  k.nslot = (long long)leaves.size();

  // --- data statistics -------------------------------------------------------
  {
    double s = 0.0, comp = 0.0;   // Kahan
    int has_nan = 0;
    for (size_t e = 0; e < ncube; ++e) {
      const double v = d->data_valid[e];
      if (v != v) { has_nan = 1; continue; }
      const double y = v * v - comp, t = s + y;
      comp = (t - s) - y;
      s = t;
    }
    k.ssq_data = s;
    k.data_has_nan = has_nan;
  }

  // --- channel table padded to CH, with per-block velocity range -----------
  std::vector<double> vpad(k.nvpad, 1.0e300), cmin(k.nvpad / CH), cmax(k.nvpad / CH);
  for (int q = 0; q < k.nv; ++q) vpad[q] = d->v_valid[q];
  for (int b = 0; b < k.nvpad / CH; ++b) {
    double lo = std::numeric_limits<double>::infinity(), hi = -lo;
    for (int q = b * CH; q < std::min(k.nv, (b + 1) * CH); ++q) {
      lo = std::min(lo, vpad[q]);
      hi = std::max(hi, vpad[q]);
    }
    cmin[b] = lo; cmax[b] = hi;
  }
  // |t| <= 45.1 + 11 * (velocity span of a block in channel widths) for every
  // channel of a block that holds emission (rinv <= 11/|dv|); the fixed-point
  // index needs |t| < 2^11
  k.fast_idx = 1;
  for (int b = 0; b < k.nvpad / CH; ++b)
    if (!(std::fabs(cmax[b] - cmin[b]) <= 150.0 * std::fabs(k.dv))) k.fast_idx = 0;

  // --- beam: separable when it is an outer product of its central row/col --
  const int cc = k.ngauss;
  const double peak = d->gaussbeam[(size_t)cc * K + cc];
  double worst = 0.0;
  for (int y = 0; y < K && peak != 0.0; ++y)
    for (int x = 0; x < K; ++x) {
      const double sepv = d->gaussbeam[(size_t)y * K + cc] * d->gaussbeam[(size_t)cc * K + x] / peak;
      worst = std::max(worst, std::fabs(sepv - d->gaussbeam[(size_t)y * K + x]));
    }
  c->separable = (peak != 0.0 && worst <= 1e-14 * std::fabs(peak)) ? 1 : 0;
  c->kpad = (K + CONV_R - 1) / CONV_R * CONV_R;
  k.kpad = c->kpad;
  std::vector<double> gx(c->kpad, 0.0), gy(c->kpad, 0.0);
  c->nrank = 1;
  if (c->separable) {
    for (int u = 0; u < K; ++u) {   // flipped: correlation taps
      gx[u] = d->gaussbeam[(size_t)cc * K + (K - 1 - u)];
      gy[u] = d->gaussbeam[(size_t)(K - 1 - u) * K + cc] / peak;
    }
  } else if (K > 1) {
    // rotated beam: leading singular triplets until the dropped tail is below
    // 1e-13 of the largest (a Gaussian's singular values fall ~x7 per term)
    std::vector<double> U, V, sig;
    beam_svd(d->gaussbeam, K, U, V, sig);
    int R = 0;
    double tail = 0.0;
    const double tol = d->beam_tol > 0.0 ? d->beam_tol : 1e-13;   // opt-in truncation (slamhip_desc::beam_tol)
    for (int r = K - 1; r >= 0; --r) {
      if (tail + sig[r] > tol * sig[0]) { R = r + 1; break; }
      tail += sig[r];
    }
    if (R >= 1 && R <= MAX_RANK) {
      c->nrank = R;
      gx.assign((size_t)R * c->kpad, 0.0);
      gy.assign((size_t)R * c->kpad, 0.0);
      for (int r = 0; r < R; ++r)
        for (int u = 0; u < K; ++u) {
          gx[(size_t)r * c->kpad + u] = V[(size_t)(K - 1 - u) * K + r];
          gy[(size_t)r * c->kpad + u] = sig[r] * U[(size_t)(K - 1 - u) * K + r];
        }
    } else {
      c->nrank = 0;                  // not factorisable cheaply: direct 2-D taps only
    }
  }
  // LDS tile ladder: 64-row windows with two workgroups per CU, 64-row windows with one,
  // 32-row windows; the fused kernel's tile is sized in its arithmetic type, the dense
  // (parity tap) kernels always work in double.
  auto pick_variant = [&](size_t item) {
    auto tile = [&](int ty) { return fz_tile_bytes(ty, c->kpad) / sizeof(double) * item; };
    if (tile(64) <= 78 * 1024 && (64 + c->kpad - 1) * (CONV_TX / CONV_R) <= 3 * 256) return 0;
    if (tile(64) + 2 * c->kpad * item <= 156 * 1024 && (64 + c->kpad - 1) * (CONV_TX / CONV_R) <= 5 * 256) return 1;
    if (tile(32) + 2 * c->kpad * item <= 156 * 1024 && (32 + c->kpad - 1) * (CONV_TX / CONV_R) <= 4 * 256) return 2;
    return -1;
  };
  c->sep_variant = pick_variant(sizeof(double));
  c->fz_variant = pick_variant(c->item);
  // Large one-term beams (K = 77 in double: BASELINE config 4) fit neither 64-row tile; between
  // them and the 32-row one a 48-row window (6 of the 8 waves take the column pass) has 2.6 halo +
  // output rows in its row pass per output row instead of 3.4.
  if (c->fz_variant == 2 && c->nrank == 1 &&
      fz_tile_bytes(48, c->kpad) / sizeof(double) * c->item + 2 * c->kpad * c->item <= 156 * 1024 &&
      48 + c->kpad - 1 <= 128)
    c->fz_variant = 3;
  // in float the 64-row tile of such a beam fits a CU once, the 48-row one twice, and the
  // row-owning kernel has an instance for it
  if (c->f32 && c->fz_variant == 1 && c->nrank == 1 && c->kpad == 80 && K <= 77) c->fz_variant = 3;
  if (c->sep_variant >= 0) c->sep_lds = sep_lds_bytes(c->sep_variant == 2 ? 32 : 64, c->kpad);
  c->c2d_lds = (size_t)(C2_TY + K - 1) * ((C2_TX + K - 1) | 1) * sizeof(double);
  if (c->sep_lds > 48 * 1024) {   // opt in to large dynamic LDS for the chosen variant only
    const void* fn = c->sep_variant == 0 ? (const void*)k_conv_sep<64, 3>
                   : c->sep_variant == 1 ? (const void*)k_conv_sep<64, 5>
                                         : (const void*)k_conv_sep<32, 4>;
    HIP_TRY(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->sep_lds));
  }
  const size_t fz_ty = c->fz_variant == 2 ? 32 : c->fz_variant == 3 ? 48 : 64;
  const size_t fz_tile = c->fz_variant < 0 ? 0 : fz_tile_bytes((int)fz_ty, c->kpad) / sizeof(double) * c->item;
  c->fz_lds = fz_tile + 2 * (size_t)std::max(c->nrank, 1) * c->kpad * c->item;
  if (c->nrank > 1 && c->fz_variant >= 0) {
    // several beam terms: the row-pass intermediate cannot overlay the tile
    c->fz_lds += (fz_ty + c->kpad - 1) * (CONV_TX + 1) * c->item;
    if (c->fz_lds > 156 * 1024) c->nrank = 0;
  }
  // The fused kernel either applies the factorised beam or, when that takes
  // many terms (a rotated Gaussian: ~14 for a 1e-13 tail, i.e. more
  // multiply-adds than the K x K taps themselves, plus an intermediate),
  // direct taps on the staged tile.
  {
    // the common case (64-row windows, K <= 32) runs with the tile pitch fixed at DIRECT_PITCH
    c->direct_fixed = c->fz_variant == 0 && c->kpad <= 32;
    const size_t tile = c->direct_fixed ? ((fz_ty + c->kpad - 1) * DIRECT_PITCH + 2) * c->item : fz_tile;
    c->fz_lds_direct = std::max(tile, 2 * fz_ty * (CONV_TX + 1) * c->item);
  }
  const bool direct_ok = c->fz_variant >= 0 && c->fz_lds_direct <= 156 * 1024;   // taps + tile fit the LDS
  c->fz_rank = (c->nrank >= 1 && (c->nrank < DIRECT_MIN_RANK || !direct_ok)) ? c->nrank : 0;
  // a caller who accepted a truncated beam (beam_tol > 0) gets the factorised form while it is the
  // faster one: measured at C3, K = 27: 5 terms 161 us/eval, direct taps 175,
  // 7 terms 205 (each term pays its own row pass, two barriers and column pass at one workgroup
  // per CU, since the intermediate cannot overlay a tile that the next term still needs)
  if (d->beam_tol > 0.0 && c->nrank >= 1 && 5 * c->nrank <= K - 2) c->fz_rank = c->nrank;
  if (c->fz_variant >= 0) {
    const void* fz = c->f32 ? fused_instance<float>(c->fz_variant, false, false)
                            : fused_instance<double>(c->fz_variant, false, false);
    const void* fd = c->f32 ? fused_instance<float>(c->fz_variant, true, c->direct_fixed)
                            : fused_instance<double>(c->fz_variant, true, c->direct_fixed);
    c->sep_fixed = c->fz_variant == 0 && c->kpad == 32;
    if (c->nrank >= 1 && c->sep_fixed) {
      const void* f2 = c->f32 ? fused_instance<float>(0, false, true, false) : fused_instance<double>(0, false, true, false);
      HIP_TRY(c, hipFuncSetAttribute(f2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
      if (!c->f32) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<double, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<double, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
      } else {
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<float, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<float, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
      }
    }
    if (c->nrank == 1 && c->fz_variant == 3 && !c->f32) {
      HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<double, false, 48, 80>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
      HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<double, true, 48, 80>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
    }
    if (c->nrank == 1 && c->fz_variant == 3 && c->f32) {
      HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<float, false, 48, 80>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
      HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<float, true, 48, 80>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
    }
    if (c->nrank >= 1) {
      HIP_TRY(c, hipFuncSetAttribute(fz, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
      const void* f1 = c->f32 ? fused_instance<float>(c->fz_variant, false, false, false)
                              : fused_instance<double>(c->fz_variant, false, false, false);
      HIP_TRY(c, hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
    }
    if (direct_ok)
      HIP_TRY(c, hipFuncSetAttribute(fd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds_direct));
  }
  c->fused = c->fz_variant >= 0 && (c->fz_rank >= 1 || direct_ok);
  {
    c->fz_sx = CONV_TX - 1;
    c->fz_sy = (int)fz_ty - 1;
    c->fz_grid = dim3((n - 1 + c->fz_sx - 1) / c->fz_sx, (n - 1 + c->fz_sy - 1) / c->fz_sy, k.nv);
    // the window descriptor packs column, row and walker into one word (8, 8 and 16 bits)
    if (c->fz_grid.x > 255 || c->fz_grid.y > 255) c->fused = false;
  }
  if (c->c2d_lds > 48 * 1024 && c->c2d_lds <= 160 * 1024) {
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_2d,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->c2d_lds));
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_obs<true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->c2d_lds));
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_obs<false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->c2d_lds));
  }

  // --- uploads -------------------------------------------------------------
  int rc;
  if ((rc = dev_upload(c, &k.Xnest0, d->Xnest0, ngrid))) return rc;
  if ((rc = dev_upload(c, &k.Ynest0, d->Ynest0, ngrid))) return rc;
  if ((rc = dev_upload(c, &k.xneed, d->xneed, (size_t)n))) return rc;
  if ((rc = dev_upload(c, &k.yneed, d->yneed, (size_t)n))) return rc;
  if ((rc = dev_upload(c, &k.xobs, d->xobs, (size_t)k.nx))) return rc;
  if ((rc = dev_upload(c, &k.yobs, d->yobs, (size_t)k.ny))) return rc;
  if ((rc = dev_upload(c, &k.vpad, vpad.data(), vpad.size()))) return rc;
  if ((rc = dev_upload(c, &k.ch_vmin, cmin.data(), cmin.size()))) return rc;
  if ((rc = dev_upload(c, &k.ch_vmax, cmax.data(), cmax.size()))) return rc;
  if ((rc = dev_upload(c, &k.data, d->data_valid, ncube))) return rc;
  if ((rc = dev_upload(c, &k.beam, d->gaussbeam, (size_t)K * K))) return rc;
  k.mom0decon = nullptr;
  if (d->mom0decon && (rc = dev_upload(c, &k.mom0decon, d->mom0decon, (size_t)k.ny * k.nx))) return rc;
  if ((rc = dev_upload(c, &k.gx_pad, gx.data(), gx.size()))) return rc;
  if ((rc = dev_upload(c, &k.gy_pad, gy.data(), gy.size()))) return rc;
  {
    std::vector<double> ft((size_t)K * c->kpad, 0.0);
    for (int b = 0; b < K; ++b)
      for (int a = 0; a < K; ++a)
        ft[(size_t)b * c->kpad + a] = d->gaussbeam[(size_t)(K - 1 - a) * K + (K - 1 - b)];
    if ((rc = dev_upload(c, &k.beam_ft, ft.data(), ft.size()))) return rc;
    if (c->f32) {
      std::vector<float> f32v(ft.begin(), ft.end()), gxf(gx.begin(), gx.end()), gyf(gy.begin(), gy.end());
      if ((rc = dev_upload(c, &c->ft32, f32v.data(), f32v.size()))) return rc;
      if ((rc = dev_upload(c, &c->gx32, gxf.data(), gxf.size()))) return rc;
      if ((rc = dev_upload(c, &c->gy32, gyf.data(), gyf.size()))) return rc;
    }
  }
  if ((rc = dev_upload(c, &k.leaf, leaves.data(), leaves.size()))) return rc;
  if ((rc = dev_upload(c, &k.tile_goff, tile_goff.data(), tile_goff.size()))) return rc;
  if ((rc = dev_upload(c, &k.tile_nl, tile_nl.data(), tile_nl.size()))) return rc;

  // --- per-group scratch -----------------------------------------------------
  c->rg_grid = dim3((k.nx + RED_THREADS - 1) / RED_THREADS, (k.ny + RG_ROWS - 1) / RG_ROWS, k.nv);
  c->red_blocks = (int)(c->rg_grid.x * c->rg_grid.y * c->rg_grid.z);
  if (d->mom0decon && c->nrank >= 1) {
    for (int ty : {32, 16}) {
      const size_t rows_ = (size_t)ty + c->kpad - 1;
      const size_t need = (rows_ * ((C2_TX + c->kpad - 1) | 1) +
                           (c->nrank > 1 ? rows_ * (C2_TX + 1) : 0)) * sizeof(double);
      const size_t nitem = rows_ * (C2_TX / 8);
      if (need <= 150 * 1024 && nitem <= 1024) {
        c->ob_ty = ty; c->ob_lds = need; c->ob_nit = (ty == 32 && nitem <= 512) ? 2 : 4;
        break;
      }
    }
    if (c->ob_ty) {
      const void* fw = c->ob_ty == 32 ? (c->ob_nit == 2 ? (const void*)k_conv_obs_sep<32, 2, true>
                                                        : (const void*)k_conv_obs_sep<32, 4, true>)
                                      : (const void*)k_conv_obs_sep<16, 4, true>;
      const void* fn = c->ob_ty == 32 ? (c->ob_nit == 2 ? (const void*)k_conv_obs_sep<32, 2, false>
                                                        : (const void*)k_conv_obs_sep<32, 4, false>)
                                      : (const void*)k_conv_obs_sep<16, 4, false>;
      HIP_TRY(c, hipFuncSetAttribute(fw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->ob_lds));
      HIP_TRY(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->ob_lds));
      const void* fl = c->ob_ty == 32 ? (c->ob_nit == 2 ? (const void*)k_conv_obs_live<32, 2>
                                                        : (const void*)k_conv_obs_live<32, 4>)
                                      : (const void*)k_conv_obs_live<16, 4>;
      HIP_TRY(c, hipFuncSetAttribute(fl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->ob_lds));
    }
  }
  c->ob_grid = dim3((k.nx + C2_TX - 1) / C2_TX, (k.ny + C2_TY - 1) / C2_TY, k.nv);
  c->ob_blocks = (int)(c->ob_grid.x * c->ob_grid.y * c->ob_grid.z);
  c->max_blocks = std::max<size_t>(std::max<size_t>(c->red_blocks, c->ob_blocks),
                                   (size_t)c->fz_grid.x * c->fz_grid.y * c->fz_grid.z);
  c->ftx = (n + PT_X - 1) / PT_X;
  c->fty = (n + PT_Y - 1) / PT_Y;
  const size_t nseg = (size_t)(k.nx + 63) / 64;
  // what one more walker costs: its share of the scratch set plus, for the 'mom0*'
  // scalings, its regridded cube and row-segment marks
  // (+ the model's moment map, and the observed cube's tile / window flags and window list)
  const size_t mom0_walker = d->mom0decon
      ? ncube * sizeof(double) + (size_t)k.nv * k.ny * nseg + (size_t)n * n * sizeof(double) +
        (size_t)k.nv * (((k.nx + PT_X - 1) / PT_X) * ((k.ny + PT_Y - 1) / PT_Y) +
                        ((k.nx + 63) / 64) * ((k.ny + 63) / 64) * (1 + sizeof(WinDesc)))
      : 0;
  const size_t per_walker = carve(c, 1, nullptr, nullptr) + mom0_walker;
  size_t freeb = 0, totalb = 0;
  HIP_TRY(c, hipMemGetInfo(&freeb, &totalb));
  size_t budget = d->max_scratch_bytes > 0 ? (size_t)d->max_scratch_bytes : ((size_t)64 << 30);
  budget = std::min(budget, freeb / 10 * 7);
  c->wgroup = (int)std::max<size_t>(1, std::min<size_t>(128, budget / per_walker));
  // keep conv's grid.z within limits
  while (c->wgroup > 1 && (long long)c->wgroup * k.nv > 32768) --c->wgroup;
  while (c->wgroup > 1 && carve(c, c->wgroup, nullptr, nullptr) + c->wgroup * mom0_walker > budget) --c->wgroup;
  const size_t W = c->wgroup;
  Scratch& S = c->own;
  {
    unsigned char* base = nullptr;
    if ((rc = dev_alloc(c, &base, carve(c, W, nullptr, nullptr)))) return rc;
    carve(c, W, base, &S);
    S.wgroup = c->wgroup;
    k.tk = S.tk;
    HIP_TRY(c, hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
  }
  if ((rc = ensure_iconv(c, 1))) return rc;
  if (d->mom0decon && (rc = dev_alloc(c, &c->d_obs, W * ncube))) return rc;
  if (d->mom0decon && (rc = dev_alloc(c, &c->d_krange, W * (size_t)c->ftx * c->fty))) return rc;
  if (d->mom0decon && (rc = dev_alloc(c, &c->d_mom, W * (size_t)n * n))) return rc;
  if (d->mom0decon && (rc = dev_alloc(c, &c->d_kmask, W * (size_t)c->ftx * c->fty * ((k.nv + 31) / 32)))) return rc;
  if (d->mom0decon && (rc = dev_alloc(c, &c->d_rowflag, W * (size_t)k.nv * k.ny * nseg))) return rc;
  {
    // a conv window must not span more occupancy bytes than its LDS table holds
    const int ty = c->fz_variant == 2 ? 32 : c->fz_variant == 3 ? 48 : 64;
    const int spanx = (CONV_TX + K - 2) / PT_X + 2, spany = (ty + K - 2) / PT_Y + 2;
    if (c->fused && (spanx * spany > FLAG_MAX || spanx > 255 || spany > 255)) c->fused = 0;   // descriptor ids pack (fy, fx) in 8 bits each
    if ((int)(c->fz_grid.x + c->fz_grid.y) > PREP_TILES) c->fused = 0;
  }
  {
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
    const int per_cu = std::max(1, (int)((160 * 1024) / std::max<size_t>(c->fz_lds + 2048, 1)));
    c->persist_blocks = prop.multiProcessorCount * std::min(per_cu, 8);
    const int per_cu_d = std::max(1, (int)((160 * 1024) / std::max<size_t>(c->fz_lds_direct + 2048, 1)));
    c->persist_direct = prop.multiProcessorCount * std::min(per_cu_d, 8);
  }
  // --- the allocation-free entry point's scratch set: the same layout inside this
  //     GPU's copy of the static device array, for as many walkers as fit
  {
    unsigned char* sp = nullptr;
    if ((rc = static_scratch(c, &sp))) return rc;
    c->fixed_need1 = carve(c, 1, nullptr, nullptr);
    int wf = 0;
    if (c->fused && !d->mom0decon) {
      wf = (int)std::min<size_t>(W, STATIC_SCRATCH_BYTES / c->fixed_need1);
      while (wf > 0 && carve(c, wf, nullptr, nullptr) > STATIC_SCRATCH_BYTES) --wf;
    }
    if (wf > 0) carve(c, wf, sp, &c->fixed);
    c->fixed.wgroup = wf;
  }
  k.fold = (c->fused && c->fz_rank == 1 && !d->mom0decon) ? 1 : 0;   // one-term beam on the fused path
  c->fold_ok = k.fold;
  HIP_TRY(c, hipHostMalloc((void**)&c->h_nit, slamhip_ctx::HINT_SLOTS * sizeof(int), hipHostMallocDefault));
  for (int i = 0; i < slamhip_ctx::HINT_SLOTS; ++i) c->h_nit[i] = -1;
  HIP_TRY(c, hipStreamCreate(&c->stream));
  if (d->mom0decon && c->ob_ty) {
    // residual sum of every (channel, tile) when the model is zero there: this
    // fit's data^2 summed by the beam kernel itself on an all-zero cube
    c->ob_nty = (k.ny + c->ob_ty - 1) / c->ob_ty;
    const size_t ntile = (size_t)c->ob_grid.x * c->ob_nty;
    if ((rc = dev_alloc(c, &c->d_obflag, W * k.nv * ntile))) return rc;
    if ((rc = dev_alloc(c, &c->d_obempty, (size_t)k.nv * ntile))) return rc;
    if (W * k.nv * ntile < (size_t)1 << 31) {     // window ids are ints
      if ((rc = dev_alloc(c, &c->d_oblist, W * k.nv * ntile))) return rc;
      if ((rc = dev_alloc(c, &c->d_obcount, 64))) return rc;
      hipDeviceProp_t pr;
      HIP_TRY(c, hipGetDeviceProperties(&pr, c->device));
      const int per = std::max(1, std::min(8, (int)((160 * 1024) / (c->ob_lds + 1024))));
      c->ob_persist = pr.multiProcessorCount * per;
    }
    // the row-owning batch path: one-term beam, 32 padded taps (the 64-row tile twice per CU),
    // observed rows of even length (the kernel stages aligned pixel pairs)
    if (c->d_oblist && c->nrank == 1 && c->fz_variant == 0 && c->kpad == 32 && !c->f32 &&
        k.nx % 2 == 0 && k.nx >= 64) {
      c->ob_wnx = (k.nx + 63) / 64;
      c->ob_wny = (k.ny + 63) / 64;
      c->ob_ftx = (k.nx + PT_X - 1) / PT_X;
      c->ob_fty = (k.ny + PT_Y - 1) / PT_Y;
      const size_t nwn = (size_t)c->ob_wnx * c->ob_wny;
      if (W * k.nv * nwn < (size_t)1 << 30 && c->ob_wnx <= 255 && c->ob_wny <= 255 &&
          nwn * k.nv <= c->max_blocks * 3) {
        if ((rc = dev_alloc(c, &c->d_oflags, W * k.nv * c->ob_ftx * c->ob_fty + 64))) return rc;
        if ((rc = dev_alloc(c, &c->d_owin, W * k.nv * nwn))) return rc;
        if ((rc = dev_alloc(c, &c->d_odesc, W * k.nv * nwn))) return rc;
        if ((rc = dev_alloc(c, &c->d_ocount, LIVECOUNT_WORDS))) return rc;
        if ((rc = dev_alloc(c, &c->d_oempty, (size_t)k.nv * nwn))) return rc;
        hipLaunchKernelGGL(k_obs_empty, dim3((unsigned)nwn, k.nv), dim3(256), 0, c->stream, k, c->ob_wnx,
                           c->ob_wny, c->d_oempty);
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_conv_rows<double, false, 64, 32, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fz_lds));
        c->ob_rows = true;
      }
    }
    HIP_TRY(c, hipMemsetAsync(c->d_obs, 0, ncube * sizeof(double), c->stream));
    dim3 g(c->ob_grid.x, c->ob_nty, k.nv);
#define SLAM_OBS0(TY_, NIT_)                                                                 \
    hipLaunchKernelGGL((k_conv_obs_sep<TY_, NIT_, false>), g, dim3(256), c->ob_lds, c->stream, k, \
                       c->d_obs, (double*)nullptr, c->d_obempty, c->kpad, c->nrank,          \
                       (const unsigned char*)nullptr, (const double*)nullptr, k.gx_pad, k.gy_pad,  \
                       (const unsigned char*)nullptr)
    if (c->ob_ty == 32 && c->ob_nit == 2) SLAM_OBS0(32, 2);
    else if (c->ob_ty == 32) SLAM_OBS0(32, 4);
    else SLAM_OBS0(16, 4);
#undef SLAM_OBS0
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, hipDeviceSynchronize());
  return SLAMHIP_OK;
}

void slamhip_destroy(slamhip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->d_stage_p) (void)hipFree(c->d_stage_p);
  if (c->d_stage_c) (void)hipFree(c->d_stage_c);
  if (c->own.done) (void)hipEventDestroy(c->own.done);
  if (c->h_nit) (void)hipHostFree(c->h_nit);
  for (auto& sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int slamhip_info(slamhip_ctx* c, int64_t* nleaf, int32_t* wgroup, int32_t* sep) {
  if (!c) return SLAMHIP_ERR_ARG;
  if (nleaf) *nleaf = c->k.nleaf;
  if (wgroup) *wgroup = c->wgroup;
  if (sep) *sep = c->separable && c->sep_variant >= 0;
  return SLAMHIP_OK;
}

int slamhip_window_shape(slamhip_ctx* c, int32_t* rows, int32_t* cols, int32_t* kpad, int32_t* fold) {
  if (!c) return SLAMHIP_ERR_ARG;
  const bool mom0 = c->k.mom0decon != nullptr;
  if (rows) *rows = mom0 ? ((c->ob_rows && c->conv_mode != 4) ? 64 : c->ob_ty) : (c->fused ? c->fz_sy + 1 : 0);
  if (cols) *cols = mom0 ? C2_TX : CONV_TX;
  if (kpad) *kpad = c->kpad;
  if (fold) *fold = c->k.fold;
  return SLAMHIP_OK;
}

int slamhip_kernel_count(void) { return KID_COUNT; }

const char* slamhip_kernel_name(int i) {
  return (i >= 0 && i < KID_COUNT) ? kKernelNames[i] : "";
}

int slamhip_profile_begin(slamhip_ctx* c) {
  if (!c) return SLAMHIP_ERR_ARG;
  for (auto& sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  c->spans.clear();
  c->profiling = true;
  return SLAMHIP_OK;
}

int slamhip_profile_end(slamhip_ctx* c, double* total_ms, int64_t* launches) {
  if (!c || !total_ms || !launches) return SLAMHIP_ERR_ARG;
  c->profiling = false;
  for (int i = 0; i < KID_COUNT; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
  int rc = SLAMHIP_OK;
  for (auto& sp : c->spans) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(sp.b);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, sp.a, sp.b);
    if (e == hipSuccess) { total_ms[sp.kid] += ms; launches[sp.kid] += 1; }
    else rc = fail(c, SLAMHIP_ERR_HIP, "profile_end: %s", hipGetErrorString(e));
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  c->spans.clear();
  return rc;
}

int slamhip_beam_rank(slamhip_ctx* c) {
  if (!c) return SLAMHIP_ERR_ARG;
  return c->fused ? c->fz_rank : 0;
}

int slamhip_set_conv_mode(slamhip_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 4) return SLAMHIP_ERR_ARG;
  if (mode == 1 && c->f32)
    return fail(c, SLAMHIP_ERR_ARG, "conv mode 1 (dense, double) is not available at precision 1 (float)");
  if (mode == 1 && !c->k.mom0decon) {
    // the dense path keeps whole convolved cubes: room for a few walkers per
    // launch group is made here, so that the batch entry points never allocate
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_iconv(c, std::min(c->wgroup, 8));
    if (rc) return rc;
  }
  c->conv_mode = mode;
  c->k.fold = mode == 3 ? 0 : c->fold_ok;       // 3: conv tile + bilinear regrid for every walker
  return SLAMHIP_OK;
}

// Walkers one launch group may hold: the set's capacity, and in the dense
// (non-fused) conv mode no more than there are convolved cubes for (the dense
// buffer is grown by slamhip_set_conv_mode, never on the data path).
static int group_cap(const slamhip_ctx* c, const Scratch& S) {
  const bool fused = c->k.mom0decon || (c->fused && c->conv_mode != 1);
  return fused ? S.wgroup : std::max(1, std::min(S.wgroup, c->iconv_walkers));
}

static int batch_on(slamhip_ctx* c, Scratch& S, const double* params_dev, int nw,
                    double* chi2_dev, hipStream_t st) {
  int rc = scratch_acquire(c, S, st);
  if (rc) return rc;
  const int cap = group_cap(c, S);
  for (int w0 = 0; w0 < nw; w0 += cap) {
    const int m = std::min(cap, nw - w0);
    if ((rc = run_group(c, S, params_dev + (size_t)w0 * NP, m, chi2_dev + w0, st))) return rc;
  }
  return scratch_release(c, S, st);
}

int slamhip_chi2_batch_dev(slamhip_ctx* c, const double* params_dev, int nw,
                           double* chi2_dev, void* stream) {
  if (!c || c->wgroup < 1) return SLAMHIP_ERR_STATE;
  if (nw < 0 || (nw > 0 && (!params_dev || !chi2_dev)))
    return fail(c, SLAMHIP_ERR_ARG, "chi2_batch_dev: bad arguments");
  if (nw == 0) return SLAMHIP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  return batch_on(c, c->own, params_dev, nw, chi2_dev, static_cast<hipStream_t>(stream));
}

int slamhip_chi2_batch_static(slamhip_ctx* c, const double* params_dev, int nw,
                              double* chi2_dev, void* stream) {
  if (!c || c->wgroup < 1) return SLAMHIP_ERR_STATE;
  if (nw < 0 || (nw > 0 && (!params_dev || !chi2_dev)))
    return fail(c, SLAMHIP_ERR_ARG, "chi2_batch_static: bad arguments");
  if (c->k.mom0decon || !c->fused || c->conv_mode == 1)
    return fail(c, SLAMHIP_ERR_ARG, "chi2_batch_static: only the fused scaling='uniform' path "
                                    "(conv modes 0, 2 and 3) runs from the static scratch");
  if (c->fixed.wgroup < 1)
    return fail(c, SLAMHIP_ERR_NOMEM,
                "chi2_batch_static: one walker of this fit needs %zu bytes of scratch, the "
                "static scratch holds %zu (SLAMHIP_STATIC_SCRATCH_BYTES)",
                c->fixed_need1, (size_t)STATIC_SCRATCH_BYTES);
  if (nw == 0) return SLAMHIP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  // no event bookkeeping here (the set has no event): calls that overlap on one
  // GPU race, as the header says
  return batch_on(c, c->fixed, params_dev, nw, chi2_dev, static_cast<hipStream_t>(stream));
}

int slamhip_static_info(slamhip_ctx* c, int64_t* scratch_bytes, int64_t* bytes_per_walker,
                        int32_t* walkers_per_group) {
  if (!c) return SLAMHIP_ERR_ARG;
  if (scratch_bytes) *scratch_bytes = (int64_t)STATIC_SCRATCH_BYTES;
  if (bytes_per_walker) *bytes_per_walker = (int64_t)c->fixed_need1;
  if (walkers_per_group) *walkers_per_group = c->fixed.wgroup;
  return SLAMHIP_OK;
}

int slamhip_chi2_batch(slamhip_ctx* c, const double* params, int nw, double* chi2_out) {
  if (!c || c->wgroup < 1) return SLAMHIP_ERR_STATE;
  if (nw < 0 || (nw > 0 && (!params || !chi2_out)))
    return fail(c, SLAMHIP_ERR_ARG, "chi2_batch: bad arguments");
  if (nw == 0) return SLAMHIP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if ((size_t)nw > c->stage_cap) {
    // staging grows to the largest batch seen (a sampler sends the same size every step)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_stage_p) (void)hipFree(c->d_stage_p);
    if (c->d_stage_c) (void)hipFree(c->d_stage_c);
    c->d_stage_p = c->d_stage_c = nullptr;
    c->stage_cap = 0;
    const size_t cap = std::max<size_t>((size_t)nw, 256);
    HIP_TRY(c, hipMalloc((void**)&c->d_stage_p, cap * NP * sizeof(double)));
    HIP_TRY(c, hipMalloc((void**)&c->d_stage_c, cap * sizeof(double)));
    c->stage_cap = cap;
  }
  HIP_TRY(c, hipMemcpyAsync(c->d_stage_p, params, (size_t)nw * NP * sizeof(double),
                            hipMemcpyHostToDevice, c->stream));
  int rc = slamhip_chi2_batch_dev(c, c->d_stage_p, nw, c->d_stage_c, c->stream);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(chi2_out, c->d_stage_c, (size_t)nw * sizeof(double),
                            hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SLAMHIP_OK;
}

int slamhip_chi2_stats(slamhip_ctx* c, const double* params, int nw, double* stats_out) {
  if (!c || c->wgroup < 1) return SLAMHIP_ERR_STATE;
  if (nw < 0 || (nw > 0 && (!params || !stats_out)))
    return fail(c, SLAMHIP_ERR_ARG, "chi2_stats: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  Scratch& S = c->own;
  int rc = scratch_acquire(c, S, c->stream);
  if (rc) return rc;
  const int cap = group_cap(c, S);
  for (int w0 = 0; w0 < nw; w0 += cap) {
    const int m = std::min(cap, nw - w0);
    HIP_TRY(c, hipMemcpyAsync(S.params, params + (size_t)w0 * NP, (size_t)m * NP * sizeof(double),
                              hipMemcpyHostToDevice, c->stream));
    if ((rc = run_group(c, S, S.params, m, S.chi2, c->stream))) return rc;
    HIP_TRY(c, hipMemcpyAsync(stats_out + (size_t)w0 * 6, S.stats, (size_t)m * 6 * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return scratch_release(c, S, c->stream);
}

// Single-vector entry points: parameters into the context's scratch set on its
// own stream, ordered behind whatever used the set last.
static int single_prepare(slamhip_ctx* c, const double* params) {
  if (!c || c->wgroup < 1) return SLAMHIP_ERR_STATE;
  if (!params) return fail(c, SLAMHIP_ERR_ARG, "null params");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = scratch_acquire(c, c->own, c->stream);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->own.params, params, NP * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return SLAMHIP_OK;
}

// ... and their common end: wait for the stream, leave the set marked as used on it.
static int single_finish(slamhip_ctx* c, hipError_t e, const char* what) {
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)scratch_release(c, c->own, c->stream);
  if (e != hipSuccess) return fail(c, SLAMHIP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return SLAMHIP_OK;
}

// One walker's Iunif as double on the context's stream: the scratch itself, or
// (precision 1) its widened copy in a buffer kept from the first such call.
static int iunif_as_double(slamhip_ctx* c, const double** out) {
  Scratch& S = c->own;
  if (!c->f32) { *out = static_cast<const double*>(S.iunif); return SLAMHIP_OK; }
  const size_t n1 = (size_t)c->k.nv * c->k.nneed * c->k.nneed;
  int rc;
  if (!c->d_iunif64 && (rc = dev_alloc(c, &c->d_iunif64, n1))) return rc;
  hipLaunchKernelGGL(k_widen, dim3(1024), dim3(256), 0, c->stream, static_cast<const float*>(S.iunif),
                     c->d_iunif64, (long long)n1);
  HIP_TRY(c, hipGetLastError());
  *out = c->d_iunif64;
  return SLAMHIP_OK;
}

int slamhip_iunif(slamhip_ctx* c, const double* params, double* out) {
  int rc = single_prepare(c, params);
  if (rc) return rc;
  Scratch& S = c->own;
  if ((rc = run_model(c, S, S.params, 1, c->stream, true))) return rc;
  const size_t nb = (size_t)c->k.nv * c->k.nneed * c->k.nneed * sizeof(double);
  const double* iu = nullptr;
  if ((rc = iunif_as_double(c, &iu))) return rc;
  return single_finish(c, hipMemcpyAsync(out, iu, nb, hipMemcpyDeviceToHost, c->stream), "iunif");
}

int slamhip_iconv(slamhip_ctx* c, const double* params, double* out) {
  int rc = single_prepare(c, params);
  if (rc) return rc;
  Scratch& S = c->own;
  if ((rc = run_model(c, S, S.params, 1, c->stream, true))) return rc;
  const double* iu = nullptr;
  if ((rc = iunif_as_double(c, &iu))) return rc;
  if ((rc = run_conv(c, iu, c->d_iconv, c->k.nv, c->stream))) return rc;
  const size_t nb = (size_t)c->k.nv * c->k.nneed * c->k.nneed * sizeof(double);
  return single_finish(c, hipMemcpyAsync(out, c->d_iconv, nb, hipMemcpyDeviceToHost, c->stream), "iconv");
}

// One walker's model cube (the parameters already in the context's scratch set)
// left in c->d_cube on the context's stream; *has_stats tells whether S.stats
// holds (gf, ff, scale, chi2) for it.  The buffer is kept for later calls and
// carries two [ny][nx] maps behind the cube for slamhip_moments.
static int cube_on_device(slamhip_ctx* c, int convolving, bool* has_stats) {
  const FitK& k = c->k;
  Scratch& S = c->own;
  const size_t ncube = (size_t)k.nv * k.ny * k.nx;
  int rc;
  if (!c->d_cube && (rc = dev_alloc(c, &c->d_cube, ncube + 2 * (size_t)k.ny * k.nx))) return rc;
  double* d_cube = c->d_cube;
  *has_stats = true;
  if (k.mom0decon) {
    *has_stats = convolving != 0;
    return run_group_mom0(c, S, S.params, 1, S.chi2, d_cube, convolving != 0, c->stream);
  }
  if ((rc = run_model(c, S, S.params, 1, c->stream, true))) return rc;
  const double* iu = nullptr;
  if ((rc = iunif_as_double(c, &iu))) return rc;
  if (convolving) {
    if ((rc = run_conv(c, iu, c->d_iconv, k.nv, c->stream))) return rc;
  } else {
    hipLaunchKernelGGL(k_peak_unity, dim3(k.nv), dim3(256), 0, c->stream, k, iu, c->d_iconv,
                       (long long)k.nneed * k.nneed);
  }
  hipLaunchKernelGGL((k_regrid<true>), c->rg_grid, dim3(RED_THREADS), 0, c->stream, k,
                     S.cellx, S.fracx, S.celly, S.fracy, c->d_iconv, S.partial,
                     d_cube);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, c->stream, k, S.partial, c->red_blocks,
                     (const unsigned char*)nullptr, S.stats, (double*)nullptr);
  hipLaunchKernelGGL(k_scale, dim3(1024), dim3(256), 0, c->stream, d_cube, (long long)ncube, S.stats);
  HIP_TRY(c, hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_cubemodel(slamhip_ctx* c, const double* params, int convolving,
                      double* out, double* stats_out) {
  int rc = single_prepare(c, params);
  if (rc) return rc;
  if (!out) return fail(c, SLAMHIP_ERR_ARG, "null output");
  bool has_stats = false;
  if ((rc = cube_on_device(c, convolving, &has_stats))) return rc;
  const size_t ncube = (size_t)c->k.nv * c->k.ny * c->k.nx;
  hipError_t e = hipMemcpyAsync(out, c->d_cube, ncube * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess && stats_out) {
    if (has_stats) {
      e = hipMemcpyAsync(stats_out, c->own.stats, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    } else {
      for (int i = 0; i < 4; ++i) stats_out[i] = std::numeric_limits<double>::quiet_NaN();
    }
  }
  return single_finish(c, e, "cubemodel");
}

int slamhip_moments(slamhip_ctx* c, const double* params, double dv, double clip,
                    double* mom0_out, double* mom1_out) {
  int rc = single_prepare(c, params);
  if (rc) return rc;
  if (!mom0_out || !mom1_out) return fail(c, SLAMHIP_ERR_ARG, "null output");
  bool has_stats = false;
  if ((rc = cube_on_device(c, 1, &has_stats))) return rc;
  const FitK& k = c->k;
  const long long npix = (long long)k.ny * k.nx;
  double* maps = c->d_cube + (size_t)k.nv * npix;
  hipLaunchKernelGGL(k_moments, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->stream,
                     c->d_cube, k.vpad, k.nv, npix, dv, clip, maps, maps + npix);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(mom0_out, maps, npix * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(mom1_out, maps + npix, npix * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  return single_finish(c, e, "moments");
}

// k_prep alone for one parameter vector, its WalkerK copied back.
static int prep_only(slamhip_ctx* c, WalkerK* hw) {
  Scratch& S = c->own;
  FitK k = c->k;
  k.tk = S.tk;
  hipLaunchKernelGGL(k_prep, dim3(1), dim3(256), 0, c->stream, k, S.params, 1, S.wk,
                     S.cellx, S.fracx, S.celly, S.fracy, S.tile_ix,
                     S.tile_jy, (int)c->fz_grid.x, (int)c->fz_grid.y, c->fz_sx, c->fz_sy,
                     (unsigned char*)nullptr, 0, (int*)nullptr, (double*)nullptr, (double*)nullptr,
                     (float*)nullptr, (float*)nullptr);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hw, S.wk, sizeof *hw, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SLAMHIP_OK;
}

int slamhip_surface(slamhip_ctx* c, const double* params, int slot, double* x_out,
                    double* v_out) {
  int rc = single_prepare(c, params);
  if (rc) return rc;
  if (slot < 0 || slot > 3 || !x_out || !v_out) return fail(c, SLAMHIP_ERR_ARG, "surface: bad arguments");
  const FitK& k = c->k;
  WalkerK hw;
  if ((rc = prep_only(c, &hw))) return rc;
  const int s = hw.slot_surf[slot];
  if (s < 0) { (void)scratch_release(c, c->own, c->stream); return 1; }
  const size_t ngrid = (size_t)k.nlayer * k.npix * k.npix;
  double* d_buf = nullptr;               // parity tap, not a data path: its buffer lives for the call
  HIP_TRY(c, hipMalloc((void**)&d_buf, 2 * ngrid * sizeof(double)));
  hipLaunchKernelGGL(k_surface_tap, dim3((unsigned)((ngrid + 255) / 256)), dim3(256), 0, c->stream,
                     k, c->own.wk, s, d_buf, d_buf + ngrid);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(x_out, d_buf, ngrid * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(v_out, d_buf + ngrid, ngrid * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  rc = single_finish(c, e, "surface");
  (void)hipFree(d_buf);
  return rc;
}

int slamhip_prof(slamhip_ctx* c, double cs, double* prof_out, double* prof_d_out) {
  if (!c || c->wgroup < 1) return SLAMHIP_ERR_STATE;
  double p[NP] = {1, 1, cs, 0, -1, 0, 0, 0, 0, 0, 0, 0, 0};
  int rc = single_prepare(c, p);
  if (rc) return rc;
  WalkerK hw;
  if ((rc = prep_only(c, &hw))) return rc;
  (void)scratch_release(c, c->own, c->stream);
  if (prof_out) std::memcpy(prof_out, hw.prof, sizeof hw.prof);
  if (prof_d_out) *prof_d_out = hw.prof_d;
  return SLAMHIP_OK;
}

int slamhip_group_counts(slamhip_ctx* c, int nw, int64_t* out) {
  if (!c || c->wgroup < 1) return SLAMHIP_ERR_STATE;
  if (!out || nw < 1 || nw > c->own.wgroup) return fail(c, SLAMHIP_ERR_ARG, "group_counts: bad arguments");
  const bool mom0 = c->k.mom0decon != nullptr;
  if ((!mom0 && !c->fused) || c->conv_mode == 1 || (mom0 && !(c->ob_ty && c->d_oblist)))
    return fail(c, SLAMHIP_ERR_ARG, "group_counts: fused / listed-window paths only (not the dense conv mode)");
  HIP_TRY(c, hipSetDevice(c->device));
  Scratch& S = c->own;
  int rc = scratch_acquire(c, S, c->stream);
  if (rc) return rc;
  const FitK& k = c->k;
  // the counters live in the (re-initialised every group) tail of the ticket words
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(S.livecount + 16);
  HIP_TRY(c, hipMemsetAsync(cnt, 0, 8 * sizeof(unsigned long long), c->stream));
  if (!mom0) {
    const int ty = c->fz_sy + 1;
    const int fz_rank = (c->conv_mode == 2 && c->nrank >= 1) ? c->nrank : c->fz_rank;
    const int rb = fz_rank ? CONV_R : (ty == 64 ? 2 * CONV_R : CONV_R);
    hipLaunchKernelGGL(k_count_windows, dim3(512), dim3(256), 0, c->stream, S.livelist, S.livecount,
                       ty, c->kpad, rb, cnt, 0);
    if (rows_kernel(c, fz_rank))
      hipLaunchKernelGGL(k_count_windows, dim3(512), dim3(256), 0, c->stream, S.livelist, S.livecount + 3,
                         ty, c->kpad, rb, cnt, (int)(c->fz_grid.x * c->fz_grid.y * c->fz_grid.z) * nw);
  }
  if (mom0 && c->ob_rows && c->conv_mode != 4)             // the row-owning kernel's windows on the observed grid
    hipLaunchKernelGGL(k_count_windows, dim3(512), dim3(256), 0, c->stream, c->d_odesc, c->d_ocount,
                       64, c->kpad, CONV_R, cnt, 0);
  hipLaunchKernelGGL(k_count_flags, dim3(1024), dim3(256), 0, c->stream, S.flags,
                     (long long)nw * k.nv * c->ftx * c->fty, S.gflag, (long long)nw * (k.nslot >> 6), cnt);
  if (c->g32)
    hipLaunchKernelGGL(k_count_hits<float2>, dim3((unsigned)(k.nslot / 256)), dim3(256), 0, c->stream, k, S.wk,
                       (const float2*)S.geom, S.gflag, nw, cnt);
  else
    hipLaunchKernelGGL(k_count_hits<double2>, dim3((unsigned)(k.nslot / 256)), dim3(256), 0, c->stream, k, S.wk,
                       (const double2*)S.geom, S.gflag, nw, cnt);
  HIP_TRY(c, hipGetLastError());
  unsigned long long h[8];
  hipError_t e = hipMemcpyAsync(h, cnt, sizeof h, hipMemcpyDeviceToHost, c->stream);
  rc = single_finish(c, e, "group_counts");
  if (rc) return rc;
  for (int i = 0; i < 8; ++i) out[i] = (int64_t)h[i];
  if (mom0 && !(c->ob_rows && c->conv_mode != 4)) {     // 'mom0*' scalings: the beam windows with emission that k_obs_list listed for the last group
    int live = 0;
    HIP_TRY(c, hipMemcpy(&live, c->d_obcount, sizeof live, hipMemcpyDeviceToHost));
    out[0] = live;
  }
#ifdef SLAMHIP_PHASE_STAMPS
  {
    unsigned long long h2[8];
    (void)hipMemcpy(h2, S.livecount + 32, sizeof h2, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) fprintf(stderr, "stamp %d %llu\n", i, h2[i]);
    (void)hipMemcpyFromSymbol(h2, HIP_SYMBOL(g_pstamp), sizeof h2);
    for (int i = 0; i < 8; ++i) fprintf(stderr, "pstamp %d %llu\n", i, h2[i]);
  }
#endif
  return SLAMHIP_OK;
}

}  // extern "C"
