// slampv.hip -- MI355X (gfx950) implementation of SLAM pvfitting's mock-PV
// likelihood, batched over walkers.  C ABI in include/slampv.h.
//
// Pipeline per walker group (all fp64, wave64, no MFMA: gather + scan work):
//   k_pv_prep    per-walker constants                       mockpvd.py:204, _pvfitting.py:189-190
//   k_pv_tau     table gather, line-of-sight velocity, channel binning and
//                column sum along z for every level/axis    precalculation.py:136-144,181-192
//   k_pv_nest    dz scaling + nested add with the diagonal average, max(tau)
//                                                           mockpvd.py:279-304
//   k_pv_spec    optional spectral smoothing                mockpvd.py:306-314
//   k_pv_cut     1-exp(-tau), beam convolution of the centre row, sub-grid mean,
//                sign flip, flux-scale sums                 mockpvd.py:316-335,145; _pvfitting.py:149-150
//   k_pv_final   flux scale, chi2 with log term, lnprob     _pvfitting.py:149-151,189-195
//
// What is NOT computed, on purpose:
//   * the joint division by the global density maximum (mockpvd.py:136-137):
//     tau enters only as tau / nanmax(tau), so the factor cancels;
//   * `collapse` of the fine levels onto the coarse ones (grid.py:186-251): the
//     refined box of every coarse level is zeroed before its column sum
//     (mockpvd.py:298), so the collapsed values are never read.
// Sums are formed in a fixed order (no floating-point atomics): results are
// bitwise reproducible and independent of the batch composition.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "slampv.h"

namespace {

constexpr int ZCHUNK = 288;     // longest z run one lane walks
constexpr int ZU = 8;           // cells in flight per lane
constexpr int TAU_W = 8;        // walkers per k_pv_tau wave (neighbours in Rc)
constexpr int TAU_C = 64 / TAU_W;   // columns per wave
constexpr int TAU_WAVES = 1;    // waves per k_pv_tau workgroup (2, 4: same speed; 8: slower)
constexpr int TASK_COLS = TAU_C * TAU_WAVES;   // columns per task
constexpr int MAX_LEVELS = 12;
constexpr int NT = 256;         // threads of the small kernels
constexpr int CUT_R = 5;        // outputs per k_pv_cut work item
constexpr int CUT_PRE = 8;      // pixels per thread k_pv_cut fetches one channel ahead

struct LevelK {
  int nx, ny, nz, ncol;
  int nused;                    // columns whose sums can reach level 0 (see `used` below)
  const int* used;              // [nused] their pixel indices y*nx+x, ascending
  int bx0, bx1, by0, by1, nsub;
  int nchunk, zlen;             // z split of this level's columns
  double dz;
  const int* it[2];
  const double* jo[2];
  const double* er[2];
  const double* et[2];
  const double* ep[2];
  long long own_off[2];         // offset of [chunk][nv][nused] inside one walker's own buffer
};

struct Task { int axis, level, col0, chunk; };

struct PvK {
  int nlevels, nv, nx_out, nsubgrid, ntheta, nr, signs[2];
  int nx0, ny0, npix0, maxpix;
  int has_beam, has_spec, has_obs;
  double dlnr, g_si, m_sun, au_m, sigma, rarea;
  const double* table;   // [ntheta][nr][8]: 5 used, padded to one 64-B line
  const double* vedge;
  const double* gauss_v;
  const double* gauss_xy;
  const double* obs[2];
  long long own_stride;  // doubles per walker in the own buffer
  LevelK lv[MAX_LEVELS];
};

struct WalkerPV {
  double shift;   // ln(Rc)/dlnr
  double vunit, alpha, frho, taumax, s2;
};

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------
__global__ void k_pv_prep(PvK k, const double* __restrict__ params, int nw,
                          WalkerPV* __restrict__ wk, unsigned long long* __restrict__ tmaxbits,
                          int* __restrict__ order) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nw) return;
  const double* p = params + (size_t)w * SLAMPV_NPARAM;
  // position of this walker in the batch sorted by Rc (ties by index): walkers
  // with neighbouring Rc read the same table lines, k_pv_tau runs them together
  {
    const double rc = p[1];
    int rank = 0;
    for (int i = 0; i < nw; ++i) {
      const double ri = params[(size_t)i * SLAMPV_NPARAM + 1];
      rank += (ri < rc) || (ri == rc && i < w) || (ri == ri && rc != rc);
    }
    if (rc != rc) {           // NaN keys go last, by index
      rank = 0;
      for (int i = 0; i < nw; ++i) {
        const double ri = params[(size_t)i * SLAMPV_NPARAM + 1];
        rank += (ri == ri) || i < w;
      }
    }
    order[rank] = w;
  }
  WalkerPV o;
  o.shift = log(p[1]) / k.dlnr;                                   // precalculation.py:184
  o.vunit = sqrt(k.g_si * p[0] * k.m_sun / p[1] / k.au_m) * 1e-3;  // mockpvd.py:204
  o.alpha = p[2];
  o.taumax = p[3];
  o.frho = p[4];
  o.s2 = (1. + p[5] * p[5]) * (k.sigma * k.sigma);                // _pvfitting.py:189
  wk[w] = o;
  tmaxbits[2 * w] = 0ull;
  tmaxbits[2 * w + 1] = 0ull;
}

// ---------------------------------------------------------------------------
// One wave = TAU_C neighbouring columns of one (axis, level, z chunk) x TAU_W
// walkers.  A lane walks its column along z for its walker, keeps the running
// sum of the channel it is in and adds it to its own column of the chunk's
// image when the channel changes.
constexpr int STG_D = 4 * 64 + 32;   // doubles per staging buffer: jo, er, et, ep [64] + it [64] (int)

__global__ __launch_bounds__(64 * TAU_WAVES) void k_pv_tau(PvK k, const Task* __restrict__ tasks,
                                               const WalkerPV* __restrict__ wk,
                                               const int* __restrict__ order, int nw,
                                               int ntask, double* __restrict__ own) {
  static_assert(ZU * TAU_C == 64, "one staged (cell, column) pair per lane");
  extern __shared__ double sh[];
  const int nv = k.nv;
  double* s_edge = sh;                       // [nv+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* stg = sh + ((nv + 2) & ~1) + wave * (2 * STG_D);   // this wave's two staging buffers
  for (int i = threadIdx.x; i <= nv; i += 64 * TAU_WAVES) s_edge[i] = k.vedge[i];
  __syncthreads();
  // A wave is TAU_C neighbouring columns x TAU_W walkers that are neighbours
  // in the Rc-sorted batch (adjacent lanes = adjacent walkers): for one cell
  // their table gathers fall on the same or adjacent 64-B lines.  The cell
  // constants do not depend on the walker: each trip the 64 lanes fetch them
  // for ZU cells x TAU_C columns, one (cell, column) pair per lane, park them in
  // LDS and read them back per (column, walker) lane; the next trip's fetch is
  // in flight during this trip's gathers.  The cell constants of a task come
  // from HBM once for all walkers (L2).  Workgroups go round-robin over the 8
  // XCDs, each with its own L2: XCD x takes the x-th eighth of the sorted batch.
  const int ti = blockIdx.z * gridDim.y + blockIdx.y;   // grid.y alone is limited to 65535
  const int ngroup = (nw + TAU_W - 1) / TAU_W;
  const int per = (ngroup + 7) >> 3;
  const int b = blockIdx.x;
  const int grp = ngroup < 16 ? b : (b & 7) * per + (b >> 3);
  if (grp >= ngroup || ti >= ntask) return;   // whole workgroup; every barrier below is reached by all of the others
  const Task t = tasks[ti];
  const int slot = grp * TAU_W + (lane & (TAU_W - 1));
  const int w = order[min(slot, nw - 1)];
  const LevelK& L = k.lv[t.level];
  const WalkerPV q = wk[w];
  const int mc = lane / TAU_W;                // this lane's column within the wave
  const int ci = t.col0 + wave * TAU_C + mc;  // index into the level's used-column list
  const bool active = slot < nw && ci < L.nused;
  const int z0 = t.chunk * L.zlen;
  const int z1 = min(L.nz, z0 + L.zlen);
  const double e0 = s_edge[0];
  const double inv_dv = (double)nv / (s_edge[nv] - e0);
  const int* __restrict__ it = L.it[t.axis];
  const double* __restrict__ jo = L.jo[t.axis];
  const double* __restrict__ er = L.er[t.axis];
  const double* __restrict__ et = L.et[t.axis];
  const double* __restrict__ ep = L.ep[t.axis];
  const double jmax = (double)(k.nr - 1);
  const double nvp1 = (double)nv + 1.0;
  // fetch role: cell lu of the trip, column lc of the wave
  const int lu = lane / TAU_C, lc = lane % TAU_C;
  const int lci = t.col0 + wave * TAU_C + lc;
  const int lcol = lci < L.nused ? L.used[lci] : -1;
  int fi;
  double fj, fr, ft, fp;
  auto fetch = [&](int z) {
    const int zz = z + lu;
    const bool ok = lcol >= 0 && zz < z1;
    const size_t c = ok ? (size_t)zz * L.ncol + lcol : 0;
    fi = ok ? it[c] : -1;
    fj = ok ? jo[c] : 0.0;
    fr = ok ? er[c] : 0.0;
    ft = ok ? et[c] : 0.0;
    fp = ok ? ep[c] : 0.0;
  };
  // the lane's own column of this chunk's image: cleared, then updated in
  // place whenever the running channel changes (same lane, program order)
  double* __restrict__ o = own + (size_t)w * k.own_stride + L.own_off[t.axis]
      + (size_t)t.chunk * nv * L.nused + ci;
  fetch(z0);
  if (active)
    for (int v = 0; v < nv; ++v) o[(size_t)v * L.nused] = 0.0;
  int cur = -1;
  double acc = 0.0;
  int buf = 0;
  for (int z = z0; z < z1; z += ZU, buf ^= 1) {
    double* B = stg + buf * STG_D;
    int* Bi = reinterpret_cast<int*>(B + 4 * 64);
    B[lane] = fj; B[64 + lane] = fr; B[128 + lane] = ft; B[192 + lane] = fp;
    Bi[lane] = fi;
    __syncthreads();            // staging is per wave; this orders its LDS writes and reads
    if (z + ZU < z1) fetch(z + ZU);
    // ZU cells per trip: their table lines are requested together so that the
    // dependent gathers overlap
    int ci_[ZU];
    double2 ta[ZU], tb[ZU];
    double tv[ZU];
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int se = u * TAU_C + mc;
      ci_[u] = active ? Bi[se] : -1;
      double j = B[se] - q.shift;
      j = fmin(fmax(j, 0.0), jmax);           // NaN -> 0; the prior lets no such Rc through
      const double* e = k.table + ((size_t)max(ci_[u], 0) * k.nr + (int)j) * 8;
      ta[u] = *reinterpret_cast<const double2*>(e);       // rho_env, rho_disk
      tb[u] = *reinterpret_cast<const double2*>(e + 2);   // vr_env, vt_env
      tv[u] = e[4];                                       // vp_env + vp_disk
    }
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      if (ci_[u] < 0) continue;
      const double rho = ta[u].x + ta[u].y * q.frho;
      if (rho == 0.0) continue;               // adds nothing to any channel
      const double vr = tb[u].x * q.alpha;
      const int e = u * TAU_C + mc;           // line-of-sight projections, from the staging buffer
      double vl = vr * B[64 + e] + tb[u].y * B[128 + e];
      vl = vl + tv[u] * B[192 + e];
      vl = vl * q.vunit;
      // channel: vedge[b] <= vl < vedge[b+1], decided on the stored edges
      const double fb = (vl - e0) * inv_dv;
      if (!(fb >= -1.0 && fb <= nvp1)) continue;          // also NaN
      int bn = (int)floor(fb);
      bn = min(max(bn, 0), nv - 1);
      while (bn > 0 && vl < s_edge[bn]) --bn;
      while (bn < nv - 1 && vl >= s_edge[bn + 1]) ++bn;
      if (!(s_edge[bn] <= vl && vl < s_edge[bn + 1])) continue;   // outside every channel
      if (bn != cur) {
        if (cur >= 0) o[(size_t)cur * L.nused] += acc;
        cur = bn;
        acc = rho;
      } else {
        acc += rho;
      }
    }
  }
  if (cur >= 0) o[(size_t)cur * L.nused] += acc;
}

// ---------------------------------------------------------------------------
// One block per (channel, axis, walker): walks the levels from the innermost
// to level 0, keeping the previous level's channel image in LDS.
__global__ __launch_bounds__(NT) void k_pv_nest(PvK k, const double* __restrict__ own,
                                                double* __restrict__ tau0,
                                                unsigned long long* __restrict__ tmaxbits,
                                                int do_max) {
  extern __shared__ double sh[];
  double* buf[2] = {sh, sh + k.maxpix};
  __shared__ double s_red[NT / 64];
  const int v = blockIdx.x, ax = blockIdx.y, w = blockIdx.z;
  const int tid = threadIdx.x;
  int cur = 0;
  for (int l = k.nlevels - 1; l >= 0; --l) {
    const LevelK& L = k.lv[l];
    const double* o = own + (size_t)w * k.own_stride + L.own_off[ax] + (size_t)v * L.nused;
    const size_t cstride = (size_t)k.nv * L.nused;
    double* dst = buf[cur];
    const double* prev = buf[cur ^ 1];
    const bool has_child = l < k.nlevels - 1;
    const int n = L.nsub;
    const int nxc = has_child ? k.lv[l + 1].nx : 0;
    for (int ci = tid; ci < L.nused; ci += NT) {
      const int p = L.used[ci];
      double s = 0.0;
      for (int ch = 0; ch < L.nchunk; ++ch) s += o[ch * cstride + ci];
      s *= L.dz;
      if (has_child) {
        const int y = p / L.nx, x = p - y * L.nx;
        if (x >= L.bx0 && x <= L.bx1 && y >= L.by0 && y <= L.by1) {
          const int Y = (y - L.by0) * n, X = (x - L.bx0) * n;
          // mockpvd.py:272-277: only sub-cells (i, i), each entered n times
          double m = 0.0;
          int cnt = 0;
          for (int jrep = 0; jrep < n; ++jrep)
            for (int i = 0; i < n; ++i) {
              const double t = prev[(size_t)(Y + i) * nxc + X + i];
              if (t == t) { m += t; ++cnt; }
            }
          s += m / (double)cnt;
        }
      }
      dst[p] = s;
    }
    __syncthreads();
    cur ^= 1;
  }
  const double* fin = buf[cur ^ 1];
  double* out = tau0 + (((size_t)w * 2 + ax) * k.nv + v) * k.npix0;
  double mx = 0.0;
  for (int p = tid; p < k.npix0; p += NT) {
    const double t = fin[p];
    out[p] = t;
    mx = fmax(mx, t);
  }
  if (do_max) {
    mx = wave_max(mx);
    if ((tid & 63) == 0) s_red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < NT / 64; ++i) mx = fmax(mx, s_red[i]);
      // tau >= 0: the IEEE bit pattern orders like the value; max is order-free
      atomicMax(&tmaxbits[2 * w + ax], (unsigned long long)__double_as_longlong(mx));
    }
  }
}

// ---------------------------------------------------------------------------
// Spectral smoothing, scipy.signal.convolve(tau, g[:,None,None], 'same')
// (mockpvd.py:307-314): out[k] = sum_k' tau[k'] g[k-k'+c], c=(nv-1)/2.
__global__ __launch_bounds__(NT) void k_pv_spec(PvK k, const double* __restrict__ tin,
                                                double* __restrict__ tout,
                                                unsigned long long* __restrict__ tmaxbits) {
  __shared__ double s_red[NT / 64];
  const int ax = blockIdx.y, w = blockIdx.z;
  const int p = blockIdx.x * NT + threadIdx.x;
  const int nv = k.nv, c = (nv - 1) / 2;
  const size_t base = ((size_t)w * 2 + ax) * nv * k.npix0;
  double mx = 0.0;
  if (p < k.npix0) {
    for (int kk = 0; kk < nv; ++kk) {
      double s = 0.0;
      for (int kp = 0; kp < nv; ++kp) {
        const int gi = kk - kp + c;
        if (gi >= 0 && gi < nv) s += tin[base + (size_t)kp * k.npix0 + p] * k.gauss_v[gi];
      }
      tout[base + (size_t)kk * k.npix0 + p] = s;
      mx = fmax(mx, s);
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < NT / 64; ++i) mx = fmax(mx, s_red[i]);
    atomicMax(&tmaxbits[2 * w + ax], (unsigned long long)__double_as_longlong(mx));
  }
}

// ---------------------------------------------------------------------------
// One block per (channel subset, axis, walker): intensity, centre row of the
// beam convolution, sub-grid mean, flip, and the three sums the flux scale
// needs.  The block keeps the padded beam in LDS and walks its channels
// (blockIdx.x, + gridDim.x, ...), the next channel's optical depths being
// fetched while the current one is convolved.
__global__ __launch_bounds__(NT) void k_pv_cut(PvK k, const WalkerPV* __restrict__ wk,
                                               const double* __restrict__ tau0,
                                               const unsigned long long* __restrict__ tmaxbits,
                                               double* __restrict__ pv,
                                               double* __restrict__ part) {
  extern __shared__ double sh[];
  constexpr int R = CUT_R;
  const int ngx = (k.nx0 + R - 1) / R;     // groups of R neighbouring outputs
  const int GP = ngx * R + k.nx0 + R;      // padded beam row: tap gi lives at gi + padl, zeros around
  double* s_I = sh;                        // [npix0]
  double* s_g = s_I + k.npix0;             // [ny0][GP] (beam) or unused
  double* s_part = s_g + (k.has_beam ? (size_t)k.ny0 * GP : 0);   // [ny0][ngx*R]
  double* s_row = s_part + (k.has_beam ? (size_t)k.ny0 * ngx * R : 0);     // [nx0] fine PV row
  __shared__ double s_red[3][NT / 64];
  const int ax = blockIdx.y, w = blockIdx.z;
  const int tid = threadIdx.x;
  const int nx0 = k.nx0, ny0 = k.ny0;
  const double mx = __longlong_as_double((long long)tmaxbits[2 * w + ax]);
  const double tm = wk[w].taumax;
  const double* tw = tau0 + ((size_t)w * 2 + ax) * k.nv * k.npix0;
  double pre[CUT_PRE];                     // this thread's pixels of the channel to come
  auto prefetch = [&](int v) {
    const double* t = tw + (size_t)v * k.npix0;
#pragma unroll
    for (int i = 0; i < CUT_PRE; ++i) {
      const int p = tid + i * NT;
      pre[i] = p < k.npix0 ? t[p] : 0.0;
    }
  };
  prefetch(blockIdx.x);
  const int cy = (ny0 - 1) / 2, cx = (nx0 - 1) / 2;
  const int padl = nx0 - 1 - cx + R;
  if (k.has_beam) {
    for (int e = tid; e < ny0 * GP; e += NT) {
      const int gy = e / GP, gi = e - gy * GP - padl;
      s_g[e] = (gi >= 0 && gi < nx0) ? k.gauss_xy[(size_t)gy * nx0 + gi] : 0.0;
    }
  }
  const int yc = ny0 / 2;
  for (int v = blockIdx.x; v < k.nv; v += gridDim.x) {
  const double* t = tw + (size_t)v * k.npix0;
#pragma unroll
  for (int i = 0; i < CUT_PRE; ++i) {
    const int p = tid + i * NT;
    if (p < k.npix0) s_I[p] = 1. - exp(-pre[i] / mx * tm);        // mockpvd.py:316
  }
  for (int p = tid + CUT_PRE * NT; p < k.npix0; p += NT)
    s_I[p] = 1. - exp(-t[p] / mx * tm);
  __syncthreads();              // also: the beam is in place (first trip)
  if (v + (int)gridDim.x < k.nv) prefetch(v + gridDim.x);
  if (k.has_beam) {
    // scipy 'same', centre row only: out[yc][x] = sum I[y'][x'] g[yc+cy-y'][x+cx-x'].
    // An item is one input row y' for R neighbouring outputs x: I[y'][x'] is read
    // once per x' for all R, the beam row slides through a register window
    // (1 + 1 LDS reads per R multiply-adds); the rows are then summed in order.
    for (int item = tid; item < ngx * ny0; item += NT) {
      const int yp = item / ngx, xg = item - yp * ngx;
      const int gy = yc + cy - yp;
      double acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.0;
      if (gy >= 0 && gy < ny0) {
        const double* Ir = s_I + (size_t)yp * nx0;
        const double* G = s_g + (size_t)gy * GP + xg * R + cx + padl;   // G[r - x'] = g[gy][x + cx - x'], x = xg*R + r
        double win[R];
#pragma unroll
        for (int r = 1; r < R; ++r) win[r] = G[r];
        for (int xp0 = 0; xp0 < nx0; xp0 += R) {
#pragma unroll
          for (int q = 0; q < R; ++q) {
            const int xp = xp0 + q;
            const double Iv = xp < nx0 ? Ir[xp] : 0.0;
            win[(R - q) % R] = G[-xp];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fma(Iv, win[(r + R - q) % R], acc[r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) s_part[(size_t)yp * ngx * R + xg * R + r] = acc[r];
    }
    __syncthreads();
    for (int x = tid; x < nx0; x += NT) {
      double sum = 0.0;
      for (int yp = 0; yp < ny0; ++yp) sum += s_part[(size_t)yp * ngx * R + x];
      s_row[x] = sum;
    }
  } else {
    for (int x = tid; x < nx0; x += NT) s_row[x] = s_I[(size_t)yc * nx0 + x];
  }
  __syncthreads();
  double A = 0.0, B = 0.0, C = 0.0;
  const int n = k.nsubgrid;
  for (int X = tid; X < k.nx_out; X += NT) {
    double m;
    if (n > 1) {                                                   // mockpvd.py:332-334
      double s = 0.0;
      int cnt = 0;
      for (int i = 0; i < n; ++i) {
        const double r = s_row[i + X * n];
        if (r == r) { s += r; ++cnt; }
      }
      m = s / (double)cnt;
    } else {
      m = s_row[X];
    }
    const int Xo = k.signs[ax] > 0 ? X : k.nx_out - 1 - X;         // mockpvd.py:145
    const size_t oi = (((size_t)w * 2 + ax) * k.nv + v) * k.nx_out + Xo;
    pv[oi] = m;
    if (k.has_obs && m == m) {
      B += m * m;
      const double d = k.obs[ax][(size_t)v * k.nx_out + Xo];
      if (d == d) { A += d * m; C += 1.0; }
    }
  }
  A = wave_sum(A); B = wave_sum(B); C = wave_sum(C);
  if ((tid & 63) == 0) { s_red[0][tid >> 6] = A; s_red[1][tid >> 6] = B; s_red[2][tid >> 6] = C; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < NT / 64; ++i) { A += s_red[0][i]; B += s_red[1][i]; C += s_red[2][i]; }
    double* o = part + ((((size_t)w * 2 + ax) * k.nv) + v) * 3;
    o[0] = A; o[1] = B; o[2] = C;
  }
  __syncthreads();              // s_I, s_row, s_red are rewritten by the next channel
  }
}

// ---------------------------------------------------------------------------
__device__ double block_sum(double v, double* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = s_red[0];
  for (int i = 1; i < NT / 64; ++i) s += s_red[i];
  return s;
}

// One block per walker: flux scale (_pvfitting.py:149-150), then the residual
// sums on the scaled model directly (:193-195, :230-232).
__global__ __launch_bounds__(NT) void k_pv_final(PvK k, const WalkerPV* __restrict__ wk,
                                                 const double* __restrict__ pv,
                                                 const double* __restrict__ part,
                                                 const unsigned long long* __restrict__ tmaxbits,
                                                 double* __restrict__ lnp, double* __restrict__ stats) {
  __shared__ double s_red[NT / 64];
  const int w = blockIdx.x, tid = threadIdx.x;
  const int nrow = 2 * k.nv;
  double A = 0.0, B = 0.0;
  for (int r = tid; r < nrow; r += NT) {
    const double* o = part + ((size_t)w * nrow + r) * 3;
    A += o[0];
    B += o[1];
  }
  A = block_sum(A, s_red);
  B = block_sum(B, s_red);
  const double f = A / B;
  const double s2 = wk[w].s2;
  const double ls2 = log(s2);
  double c = 0.0, cnt = 0.0;
  const size_t n = (size_t)k.nv * k.nx_out;
  if (k.has_obs) {
    for (int ax = 0; ax < 2; ++ax)
      for (size_t i = tid; i < n; i += NT) {
        const double d = k.obs[ax][i];
        const double r = d - f * pv[((size_t)w * 2 + ax) * n + i];
        const double e = r * r / s2;
        if (e == e) { c += e; cnt += 1.0; }
      }
  }
  c = block_sum(c, s_red);
  cnt = block_sum(cnt, s_red);
  if (tid == 0) {
    const double sr = sqrt(k.rarea);
    const double full = c + cnt * ls2;
    if (lnp) lnp[w] = -0.5 * full / sr;
    if (stats) {
      double* o = stats + (size_t)w * 6;
      o[0] = f;
      o[1] = c / sr;
      o[2] = -0.5 * full / sr;
      o[3] = __longlong_as_double((long long)tmaxbits[2 * w]);
      o[4] = __longlong_as_double((long long)tmaxbits[2 * w + 1]);
      o[5] = cnt;
    }
  }
}

}  // namespace

// ===========================================================================
// Host side
// ===========================================================================
enum PvKernelId { PK_PREP = 0, PK_TAU, PK_NEST, PK_SPEC, PK_CUT, PK_FINAL, PK_COUNT };
static const char* const kPvNames[PK_COUNT] = {"k_pv_prep", "k_pv_tau", "k_pv_nest",
                                               "k_pv_spec", "k_pv_cut", "k_pv_final"};

struct slampv_ctx {
  int device = 0;
  PvK k{};
  std::string err;
  std::vector<void*> allocs;
  int ntask = 0;
  Task* d_tasks = nullptr;
  int wgroup = 0;
  long long live_cells = 0;
  WalkerPV* d_wk = nullptr;
  double* d_params = nullptr;
  double* d_own = nullptr;
  double* d_tauA = nullptr;
  double* d_tauB = nullptr;
  double* d_pv = nullptr;
  double* d_part = nullptr;
  double* d_lnp = nullptr;
  double* d_stats = nullptr;
  unsigned long long* d_tmax = nullptr;
  int* d_order = nullptr;
  size_t lds_tau = 0, lds_nest = 0, lds_cut = 0;
  hipStream_t stream = nullptr;
  // Every launch group runs in the context's one scratch set: a group on another stream
  // than the previous one first waits for that one's last kernel.
  hipEvent_t done = nullptr;
  hipStream_t last = nullptr;
  bool used = false;
  bool profiling = false;
  struct Span { int kid; hipEvent_t a, b; };
  std::vector<Span> spans;
};

namespace {

struct Timed {
  slampv_ctx* c; hipStream_t st; hipEvent_t b = nullptr; bool on;
  Timed(slampv_ctx* c_, int kid, hipStream_t st_) : c(c_), st(st_), on(c_->profiling) {
    if (!on) return;
    hipEvent_t a = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(a, st);
    c->spans.push_back({kid, a, b});
  }
  ~Timed() { if (on) (void)hipEventRecord(b, st); }
};

int fail(slampv_ctx* c, int code, const char* fmt, ...) {
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
      return fail((c), e_ == hipErrorOutOfMemory ? SLAMPV_ERR_NOMEM            \
                                                 : SLAMPV_ERR_HIP,             \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                  __FILE__, __LINE__);                                         \
  } while (0)

template <typename T>
int dev_alloc(slampv_ctx* c, T** p, size_t count) {
  void* q = nullptr;
  HIP_TRY(c, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  c->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return SLAMPV_OK;
}

template <typename T>
int dev_upload(slampv_ctx* c, const T** p, const T* host, size_t count) {
  T* q = nullptr;
  int rc = dev_alloc(c, &q, count);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(q, host, count * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return SLAMPV_OK;
}

int check_desc(slampv_ctx* c, const slampv_desc* d) {
  if (d->nlevels < 1 || d->nlevels > MAX_LEVELS)
    return fail(c, SLAMPV_ERR_ARG, "nlevels=%d outside [1,%d]", d->nlevels, MAX_LEVELS);
  if (d->nv < 1 || d->nx_out < 1 || d->nsubgrid < 1 || d->ntheta < 1 || d->nr < 1)
    return fail(c, SLAMPV_ERR_ARG, "nv, nx_out, nsubgrid, ntheta, nr must be positive");
  if (!d->table || !d->vedge || !d->levels)
    return fail(c, SLAMPV_ERR_ARG, "table, vedge and levels must be given");
  if ((d->obs_major == nullptr) != (d->obs_minor == nullptr))
    return fail(c, SLAMPV_ERR_ARG, "obs_major and obs_minor must both be given or both be NULL");
  if (std::abs(d->signmajor) != 1 || std::abs(d->signminor) != 1)
    return fail(c, SLAMPV_ERR_ARG, "signmajor/signminor must be +-1");
  if (!(d->dlnr > 0)) return fail(c, SLAMPV_ERR_ARG, "dlnr must be positive");
  for (int i = 0; i < d->nv; ++i)
    if (!(d->vedge[i] < d->vedge[i + 1]))
      return fail(c, SLAMPV_ERR_ARG, "vedge must increase (channel %d)", i);
  for (int l = 0; l < d->nlevels; ++l) {
    const slampv_level& L = d->levels[l];
    if (L.nx < 1 || L.ny < 1 || L.nz < 1)
      return fail(c, SLAMPV_ERR_ARG, "level %d has an empty axis", l);
    for (int a = 0; a < 2; ++a)
      if (!L.idx_t[a] || !L.j_org[a] || !L.elos_r[a] || !L.elos_t[a] || !L.elos_p[a])
        return fail(c, SLAMPV_ERR_ARG, "level %d axis %d: missing cell array", l, a);
    if (l < d->nlevels - 1) {
      const slampv_level& Cn = d->levels[l + 1];
      if (L.nsub < 1 || L.bx0 < 0 || L.by0 < 0 || L.bx1 >= L.nx || L.by1 >= L.ny ||
          L.bx1 < L.bx0 || L.by1 < L.by0)
        return fail(c, SLAMPV_ERR_ARG, "level %d: bad refinement box", l);
      if ((L.bx1 - L.bx0 + 1) * L.nsub != Cn.nx || (L.by1 - L.by0 + 1) * L.nsub != Cn.ny)
        return fail(c, SLAMPV_ERR_ARG, "level %d: box x nsub does not match level %d's pixels", l, l + 1);
    }
    const size_t n = (size_t)L.nx * L.ny * L.nz;
    for (int a = 0; a < 2; ++a)
      for (size_t i = 0; i < n; ++i)
        if (L.idx_t[a][i] >= d->ntheta)
          return fail(c, SLAMPV_ERR_ARG, "level %d axis %d: idx_t out of the table", l, a);
  }
  if (d->levels[0].nx != d->nx_out * d->nsubgrid)
    return fail(c, SLAMPV_ERR_ARG, "level-0 nx=%d != nx_out*nsubgrid=%d", d->levels[0].nx,
                d->nx_out * d->nsubgrid);
  return SLAMPV_OK;
}

int run_group(slampv_ctx* c, const double* d_params, int nw, double* d_lnp, double* d_stats,
              hipStream_t st) {
  const PvK& k = c->k;
  if (!c->done) HIP_TRY(c, hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  if (c->used && c->last != st) HIP_TRY(c, hipStreamWaitEvent(st, c->done, 0));
  {
    Timed t(c, PK_PREP, st);
    hipLaunchKernelGGL(k_pv_prep, dim3((nw + 63) / 64), dim3(64), 0, st, k, d_params, nw, c->d_wk,
                       c->d_tmax, c->d_order);
  }
  {
    Timed t(c, PK_TAU, st);
    const int ngroup = (nw + TAU_W - 1) / TAU_W;
    const int gx = ngroup < 16 ? ngroup : ((ngroup + 7) >> 3) << 3;
    const int gz = (c->ntask + 65534) / 65535, gy = (c->ntask + gz - 1) / gz;
    hipLaunchKernelGGL(k_pv_tau, dim3(gx, gy, gz), dim3(64 * TAU_WAVES), c->lds_tau, st, k, c->d_tasks,
                       c->d_wk, c->d_order, nw, c->ntask, c->d_own);
  }
  {
    Timed t(c, PK_NEST, st);
    hipLaunchKernelGGL(k_pv_nest, dim3(k.nv, 2, nw), dim3(NT), c->lds_nest, st, k, c->d_own,
                       c->d_tauA, c->d_tmax, k.has_spec ? 0 : 1);
  }
  const double* tau = c->d_tauA;
  if (k.has_spec) {
    Timed t(c, PK_SPEC, st);
    hipLaunchKernelGGL(k_pv_spec, dim3((k.npix0 + NT - 1) / NT, 2, nw), dim3(NT), 0, st, k,
                       c->d_tauA, c->d_tauB, c->d_tmax);
    tau = c->d_tauB;
  }
  {
    Timed t(c, PK_CUT, st);
    // enough workgroups to fill the GPU; beyond that, more channels per workgroup
    const int split = std::min(k.nv, std::max(1, 2048 / (2 * nw)));
    hipLaunchKernelGGL(k_pv_cut, dim3(split, 2, nw), dim3(NT), c->lds_cut, st, k, c->d_wk, tau,
                       c->d_tmax, c->d_pv, c->d_part);
  }
  {
    Timed t(c, PK_FINAL, st);
    hipLaunchKernelGGL(k_pv_final, dim3(nw), dim3(NT), 0, st, k, c->d_wk, c->d_pv, c->d_part,
                       c->d_tmax, d_lnp, d_stats);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(c->done, st));
  c->last = st;
  c->used = true;
  return SLAMPV_OK;
}

int usable(slampv_ctx* c) {
  if (!c) return SLAMPV_ERR_ARG;
  if (c->wgroup <= 0) return fail(c, SLAMPV_ERR_STATE, "context not initialised");
  HIP_TRY(c, hipSetDevice(c->device));
  return SLAMPV_OK;
}

}  // namespace

extern "C" {

int slampv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// SLAM_SRC_SHA: hash of this file + include/slampv.h, passed by slam_amd/build.py
#ifndef SLAM_SRC_SHA
#define SLAM_SRC_SHA "unknown"
#endif
const char* slampv_version(void) { return "slampv 0.2 (gfx950) src=" SLAM_SRC_SHA; }
int slampv_kernel_count(void) { return PK_COUNT; }
const char* slampv_kernel_name(int i) { return (i >= 0 && i < PK_COUNT) ? kPvNames[i] : ""; }

const char* slampv_last_error(slampv_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

void slampv_destroy(slampv_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& s : ctx->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  if (ctx->done) (void)hipEventDestroy(ctx->done);
  for (void* p : ctx->allocs) (void)hipFree(p);
  delete ctx;
}

int slampv_create(const slampv_desc* d, slampv_ctx** out) {
  if (!d || !out) return SLAMPV_ERR_ARG;
  slampv_ctx* c = new slampv_ctx();
  *out = c;
  int rc = check_desc(c, d);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(c, SLAMPV_ERR_HIP, "no HIP device visible");
  HIP_TRY(c, hipGetDevice(&c->device));
  PvK& k = c->k;
  k.nlevels = d->nlevels; k.nv = d->nv; k.nx_out = d->nx_out; k.nsubgrid = d->nsubgrid;
  k.ntheta = d->ntheta; k.nr = d->nr; k.signs[0] = d->signmajor; k.signs[1] = d->signminor;
  k.dlnr = d->dlnr; k.g_si = d->g_si; k.m_sun = d->m_sun; k.au_m = d->au_m;
  k.sigma = d->sigma; k.rarea = d->rarea;
  k.has_beam = d->gauss_xy != nullptr;
  k.has_spec = d->gauss_v != nullptr;
  k.has_obs = d->obs_major != nullptr;
  k.nx0 = d->levels[0].nx; k.ny0 = d->levels[0].ny; k.npix0 = k.nx0 * k.ny0;
  // tables: one 64-byte line per (theta, ln r) entry
  {
    const size_t n = (size_t)d->ntheta * d->nr;
    std::vector<double> t(n * 8, 0.0);
    for (size_t i = 0; i < n; ++i)
      for (int q = 0; q < SLAMPV_NTAB; ++q) t[i * 8 + q] = d->table[i * SLAMPV_NTAB + q];
    if ((rc = dev_upload(c, &k.table, t.data(), t.size()))) return rc;
  }
  if ((rc = dev_upload(c, &k.vedge, d->vedge, (size_t)d->nv + 1))) return rc;
  if (k.has_spec && (rc = dev_upload(c, &k.gauss_v, d->gauss_v, (size_t)d->nv))) return rc;
  if (k.has_beam && (rc = dev_upload(c, &k.gauss_xy, d->gauss_xy, (size_t)k.npix0))) return rc;
  if (k.has_obs) {
    const size_t n = (size_t)d->nv * d->nx_out;
    if ((rc = dev_upload(c, &k.obs[0], d->obs_major, n))) return rc;
    if ((rc = dev_upload(c, &k.obs[1], d->obs_minor, n))) return rc;
  }
  // Columns whose sums can reach level 0.  The nested add keeps only the
  // DIAGONAL sub-cells of a refined pixel (mockpvd.py:272-277), so of the next
  // level's pixels only (Y*n+i, X*n+i), i < n, of a pixel (Y, X) that is itself
  // kept are ever read: the others are neither computed nor stored.
  std::vector<std::vector<int>> used(d->nlevels);
  {
    std::vector<char> mark, next;
    for (int l = 0; l < d->nlevels; ++l) {
      const slampv_level& S = d->levels[l];
      if (l == 0) mark.assign((size_t)S.nx * S.ny, 1); else mark.swap(next);
      for (int p = 0; p < S.nx * S.ny; ++p) if (mark[p]) used[l].push_back(p);
      if (l + 1 < d->nlevels) {
        const slampv_level& Cn = d->levels[l + 1];
        next.assign((size_t)Cn.nx * Cn.ny, 0);
        for (int y = S.by0; y <= S.by1; ++y)
          for (int x = S.bx0; x <= S.bx1; ++x) {
            if (!mark[(size_t)y * S.nx + x]) continue;
            for (int i = 0; i < S.nsub; ++i)
              next[(size_t)((y - S.by0) * S.nsub + i) * Cn.nx + (x - S.bx0) * S.nsub + i] = 1;
          }
      }
    }
  }
  // levels, column tasks
  std::vector<Task> tasks;
  long long off = 0;
  k.maxpix = 0;
  for (int l = 0; l < d->nlevels; ++l) {
    const slampv_level& S = d->levels[l];
    LevelK& L = k.lv[l];
    L.nx = S.nx; L.ny = S.ny; L.nz = S.nz; L.ncol = S.nx * S.ny;
    L.nused = (int)used[l].size();
    if ((rc = dev_upload(c, &L.used, used[l].data(), used[l].size()))) return rc;
    L.bx0 = S.bx0; L.bx1 = S.bx1; L.by0 = S.by0; L.by1 = S.by1; L.nsub = S.nsub;
    L.dz = S.dz;
    L.nchunk = (S.nz + ZCHUNK - 1) / ZCHUNK;
    L.zlen = (S.nz + L.nchunk - 1) / L.nchunk;
    k.maxpix = std::max(k.maxpix, L.ncol);
    const size_t n = (size_t)L.ncol * L.nz;
    for (int a = 0; a < 2; ++a) {
      if ((rc = dev_upload(c, &L.it[a], S.idx_t[a], n))) return rc;
      if ((rc = dev_upload(c, &L.jo[a], S.j_org[a], n))) return rc;
      if ((rc = dev_upload(c, &L.er[a], S.elos_r[a], n))) return rc;
      if ((rc = dev_upload(c, &L.et[a], S.elos_t[a], n))) return rc;
      if ((rc = dev_upload(c, &L.ep[a], S.elos_p[a], n))) return rc;
      for (int p : used[l])
        for (int z = 0; z < S.nz; ++z) c->live_cells += S.idx_t[a][(size_t)z * L.ncol + p] >= 0;
      L.own_off[a] = off;
      off += (long long)L.nchunk * d->nv * L.nused;
      for (int ch = 0; ch < L.nchunk; ++ch)
        for (int c0 = 0; c0 < L.nused; c0 += TASK_COLS) tasks.push_back({a, l, c0, ch});
    }
  }
  k.own_stride = off;
  c->ntask = (int)tasks.size();
  {
    const Task* dt = nullptr;
    if ((rc = dev_upload(c, &dt, tasks.data(), tasks.size()))) return rc;
    c->d_tasks = const_cast<Task*>(dt);
  }
  // LDS budgets
  c->lds_tau = (size_t)(((d->nv + 2) & ~1) + TAU_WAVES * 2 * STG_D) * sizeof(double);
  c->lds_nest = (size_t)2 * k.maxpix * sizeof(double);
  {
    const size_t ngx = (k.nx0 + CUT_R - 1) / CUT_R, gp = ngx * CUT_R + k.nx0 + CUT_R;
    c->lds_cut = ((size_t)k.npix0 + (k.has_beam ? (size_t)k.ny0 * (gp + ngx * CUT_R) : 0) + k.nx0) * sizeof(double);
  }
  const size_t lds_cap = 150 * 1024;
  if (c->lds_tau > lds_cap)
    return fail(c, SLAMPV_ERR_ARG, "nv=%d needs %zu B of LDS per wave (> %zu)", d->nv, c->lds_tau, lds_cap);
  if (c->lds_nest > lds_cap || c->lds_cut > lds_cap)
    return fail(c, SLAMPV_ERR_ARG, "level images of %d pixels do not fit the LDS", k.maxpix);
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pv_tau),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_tau));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pv_nest),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_nest));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pv_cut),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_cut));
  // per-walker scratch
  const size_t per_walker = ((size_t)k.own_stride + (size_t)2 * 2 * d->nv * k.npix0 +
                             (size_t)2 * d->nv * d->nx_out + (size_t)2 * d->nv * 3 + 32) * sizeof(double);
  size_t budget = d->max_scratch_bytes > 0 ? (size_t)d->max_scratch_bytes : (size_t)8 << 30;
  size_t freeb = 0, totalb = 0;
  HIP_TRY(c, hipMemGetInfo(&freeb, &totalb));
  budget = std::min(budget, freeb / 2);
  int wg = (int)std::min<size_t>(std::max<size_t>(budget / per_walker, 1), 4096);
  c->wgroup = 0;
  if ((rc = dev_alloc(c, &c->d_wk, (size_t)wg))) return rc;
  if ((rc = dev_alloc(c, &c->d_params, (size_t)wg * SLAMPV_NPARAM))) return rc;
  if ((rc = dev_alloc(c, &c->d_own, (size_t)wg * k.own_stride))) return rc;
  if ((rc = dev_alloc(c, &c->d_tauA, (size_t)wg * 2 * d->nv * k.npix0))) return rc;
  if (k.has_spec && (rc = dev_alloc(c, &c->d_tauB, (size_t)wg * 2 * d->nv * k.npix0))) return rc;
  if ((rc = dev_alloc(c, &c->d_pv, (size_t)wg * 2 * d->nv * d->nx_out))) return rc;
  if ((rc = dev_alloc(c, &c->d_part, (size_t)wg * 2 * d->nv * 3))) return rc;
  if ((rc = dev_alloc(c, &c->d_lnp, (size_t)wg))) return rc;
  if ((rc = dev_alloc(c, &c->d_stats, (size_t)wg * 6))) return rc;
  if ((rc = dev_alloc(c, &c->d_tmax, (size_t)wg * 2))) return rc;
  if ((rc = dev_alloc(c, &c->d_order, (size_t)wg))) return rc;
  c->wgroup = wg;
  return SLAMPV_OK;
}

int slampv_info(slampv_ctx* c, int64_t* live_cells, int32_t* walkers_per_group,
                int32_t* tasks_per_walker) {
  if (!c) return SLAMPV_ERR_ARG;
  if (live_cells) *live_cells = c->live_cells;
  if (walkers_per_group) *walkers_per_group = c->wgroup;
  if (tasks_per_walker) *tasks_per_walker = c->ntask;
  return SLAMPV_OK;
}

int slampv_lnprob_batch_dev(slampv_ctx* c, const double* params_dev, int nw, double* lnp_dev,
                            void* stream) {
  int rc = usable(c);
  if (rc) return rc;
  if (nw < 0 || (nw > 0 && (!params_dev || !lnp_dev)))
    return fail(c, SLAMPV_ERR_ARG, "bad batch arguments");
  if (!c->k.has_obs) return fail(c, SLAMPV_ERR_STATE, "context was created without observations");
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int w0 = 0; w0 < nw; w0 += c->wgroup) {
    const int n = std::min(c->wgroup, nw - w0);
    if ((rc = run_group(c, params_dev + (size_t)w0 * SLAMPV_NPARAM, n, lnp_dev + w0, nullptr, st)))
      return rc;
  }
  return SLAMPV_OK;
}

static int host_batch(slampv_ctx* c, const double* params, int nw, double* lnp_out,
                      double* stats_out) {
  int rc = usable(c);
  if (rc) return rc;
  if (nw < 0 || (nw > 0 && !params)) return fail(c, SLAMPV_ERR_ARG, "bad batch arguments");
  for (int w0 = 0; w0 < nw; w0 += c->wgroup) {
    const int n = std::min(c->wgroup, nw - w0);
    HIP_TRY(c, hipMemcpyAsync(c->d_params, params + (size_t)w0 * SLAMPV_NPARAM,
                              (size_t)n * SLAMPV_NPARAM * sizeof(double), hipMemcpyHostToDevice,
                              c->stream));
    if ((rc = run_group(c, c->d_params, n, c->d_lnp, c->d_stats, c->stream))) return rc;
    if (lnp_out)
      HIP_TRY(c, hipMemcpyAsync(lnp_out + w0, c->d_lnp, (size_t)n * sizeof(double),
                                hipMemcpyDeviceToHost, c->stream));
    if (stats_out)
      HIP_TRY(c, hipMemcpyAsync(stats_out + (size_t)w0 * 6, c->d_stats,
                                (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return SLAMPV_OK;
}

int slampv_lnprob_batch(slampv_ctx* c, const double* params, int nw, double* lnp_out) {
  if (c && !c->k.has_obs) return fail(c, SLAMPV_ERR_STATE, "context was created without observations");
  if (c && nw > 0 && !lnp_out) return fail(c, SLAMPV_ERR_ARG, "lnp_out is NULL");
  return host_batch(c, params, nw, lnp_out, nullptr);
}

int slampv_stats(slampv_ctx* c, const double* params, int nw, double* stats_out) {
  if (c && nw > 0 && !stats_out) return fail(c, SLAMPV_ERR_ARG, "stats_out is NULL");
  return host_batch(c, params, nw, nullptr, stats_out);
}

int slampv_mockpvd(slampv_ctx* c, const double* params, double* major_out, double* minor_out) {
  if (c && (!major_out || !minor_out)) return fail(c, SLAMPV_ERR_ARG, "output is NULL");
  int rc = host_batch(c, params, 1, nullptr, nullptr);
  if (rc) return rc;
  const size_t n = (size_t)c->k.nv * c->k.nx_out;
  HIP_TRY(c, hipMemcpy(major_out, c->d_pv, n * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(minor_out, c->d_pv + n, n * sizeof(double), hipMemcpyDeviceToHost));
  return SLAMPV_OK;
}

int slampv_own_tau(slampv_ctx* c, const double* params, int axis, int level, double* out) {
  if (c && (axis < 0 || axis > 1 || level < 0 || level >= c->k.nlevels || !out))
    return fail(c, SLAMPV_ERR_ARG, "bad axis/level/output");
  int rc = host_batch(c, params, 1, nullptr, nullptr);
  if (rc) return rc;
  const LevelK& L = c->k.lv[level];
  const size_t nu = (size_t)c->k.nv * L.nused;
  std::vector<double> tmp(nu * L.nchunk);
  std::vector<int> used(L.nused);
  HIP_TRY(c, hipMemcpy(tmp.data(), c->d_own + L.own_off[axis], tmp.size() * sizeof(double),
                       hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(used.data(), L.used, used.size() * sizeof(int), hipMemcpyDeviceToHost));
  const double nan = std::numeric_limits<double>::quiet_NaN();
  for (size_t i = 0; i < (size_t)c->k.nv * L.ncol; ++i) out[i] = nan;
  for (int v = 0; v < c->k.nv; ++v)
    for (int ci = 0; ci < L.nused; ++ci) {
      double s = 0.0;
      for (int ch = 0; ch < L.nchunk; ++ch) s += tmp[ch * nu + (size_t)v * L.nused + ci];
      out[(size_t)v * L.ncol + used[ci]] = s;
    }
  return SLAMPV_OK;
}

int slampv_tau(slampv_ctx* c, const double* params, int axis, double* out) {
  if (c && (axis < 0 || axis > 1 || !out)) return fail(c, SLAMPV_ERR_ARG, "bad axis/output");
  int rc = host_batch(c, params, 1, nullptr, nullptr);
  if (rc) return rc;
  const size_t n = (size_t)c->k.nv * c->k.npix0;
  const double* src = c->k.has_spec ? c->d_tauB : c->d_tauA;
  HIP_TRY(c, hipMemcpy(out, src + (size_t)axis * n, n * sizeof(double), hipMemcpyDeviceToHost));
  return SLAMPV_OK;
}

int slampv_profile_begin(slampv_ctx* c) {
  int rc = usable(c);
  if (rc) return rc;
  for (auto& s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  c->spans.clear();
  c->profiling = true;
  return SLAMPV_OK;
}

int slampv_profile_end(slampv_ctx* c, double* total_ms, int64_t* launches) {
  int rc = usable(c);
  if (rc) return rc;
  c->profiling = false;
  HIP_TRY(c, hipDeviceSynchronize());
  for (int i = 0; i < PK_COUNT; ++i) {
    if (total_ms) total_ms[i] = 0.0;
    if (launches) launches[i] = 0;
  }
  for (auto& s : c->spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      if (total_ms) total_ms[s.kid] += ms;
      if (launches) launches[s.kid] += 1;
    }
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  c->spans.clear();
  return SLAMPV_OK;
}

}  // extern "C"
