/*
 * slamhip.h -- C ABI of libslamhip.so: the MI355X (gfx950) implementation of
 * SLAM channelfit's inner likelihood loop, batched over emcee walkers.
 *
 * The reference (jinshisai/SLAM) is pure Python and has no FFI of its own; the
 * de-facto boundary this library sits behind is
 *
 *   - the per-walker closure `lnprob(p)`      channelfit/_channelfit.py:594-603
 *   - the public method `ChannelFit.cubemodel` channelfit/_channelfit.py:499-537
 *
 * Each entry point below names the reference code it replaces.  Everything is
 * plain C: opaque handle, caller-owned row-major float64 arrays, `int` status
 * (0 = ok, <0 = error; text via slamhip_last_error).  No exceptions cross the
 * boundary.  One context drives ONE GPU (the current HIP device at create
 * time); multi-GPU use is one process per GPU, each with its own context and
 * its own shard of the walkers (walkers are independent -- no collective on the
 * data path).
 *
 * Parameter vectors are always 13 float64 in the reference's positional order
 * (_channelfit.py:219-221), natural units (NOT log10):
 *   Mstar, Rc, cs, h1, h2, pI, Rin, Ienv, xoff, yoff, voff, incloff, paoff
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMHIP_NPARAM 13
#define SLAMHIP_PROF_LEN 45 /* _channelfit.py:419: 2*int(2*11)+1 */

enum {
  SLAMHIP_OK = 0,
  SLAMHIP_ERR_ARG = -1,    /* bad descriptor / argument                  */
  SLAMHIP_ERR_HIP = -2,    /* a HIP runtime call failed                  */
  SLAMHIP_ERR_NOMEM = -3,  /* device allocation failed                   */
  SLAMHIP_ERR_STATE = -4   /* context unusable                           */
};

/* Everything `ChannelFit.makegrid` derives once per fit
 * (_channelfit.py:227-341).  All pointers are HOST pointers, copied to the GPU
 * by slamhip_create; the caller may free them afterwards. */
typedef struct slamhip_desc {
  int32_t nv;        /* len(v_valid)                                :270 */
  int32_t ny, nx;    /* observed cube, data_valid[nv][ny][nx]       :272-274 */
  int32_t nlayer;    /* nested-grid layers                          :298 */
  int32_t npix;      /* nested-grid side, multiple of 4             :294-295 */
  int32_t nq1, nq3;  /* inner quadrant [nq1,nq3) of each layer      :296-297 */
  int32_t ineed0, ineed1; /* layer-0 crop [ineed0,ineed1)           :337-339 */
  int32_t kbeam;     /* beam kernel side K = 2*ngauss+1             :329-333 */
  int32_t disk;      /* class flag `disk`                           :222 */
  int32_t envelope;  /* class flag `envelope`                       :223 */
  double dv;         /* channel width (may be < 0)           utils.py:262 */
  double sigma;      /* noise rms                                   :602 */
  double pixperbeam; /* sum(gaussbeam)                              :334 */
  double incl0;      /* inclination [deg]                           :258 */
  double signmajor, signminor; /* +-1                               :282-289 */
  double vunit;      /* sqrt(G Msun / au) [km/s]                    :29-32 */
  const double* Xnest0; /* [nlayer][npix][npix]                     :318-320 */
  const double* Ynest0; /* [nlayer][npix][npix]                            */
  const double* xneed;  /* [ineed1-ineed0] ascending                :340 */
  const double* yneed;  /* [ineed1-ineed0] ascending                :341 */
  const double* xobs;   /* [nx] observed x (usually descending)     :263 */
  const double* yobs;   /* [ny] observed y                          :263 */
  const double* v_valid;    /* [nv]                                 :270 */
  const double* data_valid; /* [nv][ny][nx]                         :274 */
  const double* gaussbeam;  /* [kbeam][kbeam], not normalised       :333 */
  int64_t max_scratch_bytes; /* cap on per-batch device scratch; 0 = default
                              * (64 GB or 70 % of the free memory, at most
                              * 128 walkers per launch group) */
  const double* mom0decon;  /* [ny][nx] deconvolved moment 0 for the 'mom0*'
                             * scalings (:343-364), NULL = scaling 'uniform'.
                             * With it, `gaussbeam` is the x-flipped kernel of
                             * :345, the model is regrid -> per-pixel moment-0
                             * scale -> beam on the observed grid (:520-530)
                             * and has no global amplitude (:533-536).        */
  int32_t precision;        /* arithmetic of the hot path (slamhip_chi2_batch*):
                             * 0 = double (the reference's; parity 1e-9);
                             * 1 = float: line-profile accumulation, Iunif and beam
                             *     passes in single precision (parity 1e-3 on cubes
                             *     and chi2).  Geometry (:382-452), the truncating
                             *     table index (:462-463) and the sums of :486-487 /
                             *     :601 stay in double, so every sample selects the
                             *     reference's table entry;
                             * 2 = as 1 with the geometry in single precision too:
                             *     about one sample in 1e5 then lands in the
                             *     neighbouring table entry (isolated voxels differ
                             *     by a few percent of their value; chi2 still 1e-3).
                             * 1 and 2: scaling 'uniform' only; slamhip_cubemodel /
                             * _iconv apply the (double) dense beam to the
                             * single-precision Iunif. */
  double beam_tol;          /* 0 (default) = the beam exactly as given (:329-335).
                             * > 0: OPT-IN approximation for beams that are not an
                             * outer product (bpa not a multiple of 90 deg): the
                             * K x K kernel is replaced by its leading singular
                             * triplets, as few as leave the dropped singular values
                             * summing to <= beam_tol x the largest (a rotated
                             * Gaussian's fall ~7x per term: 1e-7 keeps ~7 of 27),
                             * and the fused kernel runs them as row + column passes
                             * instead of K x K direct taps.  Model cube and chi2
                             * then differ from the exact ones by ~beam_tol relative
                             * (north_star allows 1e-5).  Used wherever the
                             * factorised beam is (the batch entry points; the
                             * 'mom0*' beam pass); slamhip_cubemodel / _iconv with
                             * scaling 'uniform' keep the exact dense taps.  No
                             * effect on separable beams.  Range [0, 1e-3]. */
} slamhip_desc;

typedef struct slamhip_ctx slamhip_ctx;

/* Replaces the state `makegrid` leaves on `self` (:227-341): uploads the
 * constants to the current GPU and builds the nested-grid sample list. */
int slamhip_create(const slamhip_desc* desc, slamhip_ctx** out);

/* Replaces `lnprob`'s body (:597-602) for a whole batch: for each of the `nw`
 * parameter rows computes cubemodel (:499-537, scaling='uniform') and
 *   chi2 = nansum((data_valid - model)^2) / sigma^2 / pixperbeam.
 * The caller applies -0.5 and the box prior (utils.py:32-36).
 * `params` [nw][13] and `chi2_out` [nw] are host pointers. Synchronous. */
int slamhip_chi2_batch(slamhip_ctx* ctx, const double* params, int nw,
                       double* chi2_out);

/* Same, with `params_dev` [nw][13] and `chi2_dev` [nw] already resident in
 * device memory on the context's GPU.  Work is enqueued on `stream` (a
 * hipStream_t passed as void*, NULL = default stream) and NOT waited for; the
 * call allocates nothing.
 * Ordering: every entry point of one context works in that context's single
 * scratch set.  The library orders them on the device -- a call on another
 * stream than the previous one first waits (hipStreamWaitEvent) for the
 * previous call's last kernel -- but the HOST calls on one context must not
 * overlap: one thread at a time per context.  The one exception is
 * slamhip_chi2_batch_static below: its scratch is shared by every context on
 * the GPU and carries no event, so the library does NOT order its calls. */
int slamhip_chi2_batch_dev(slamhip_ctx* ctx, const double* params_dev, int nw,
                           double* chi2_dev, void* stream);

/* Allocation-free variant of slamhip_chi2_batch_dev.  It has NO counterpart in
 * the reference: it is for callers that cannot call hipMalloc -- code under
 * HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same arguments, same results (bitwise), same kernels; the per-group scratch
 * (geometry, sparse Iunif tiles, window list, partial sums) lives in a static
 * device array of SLAMHIP_STATIC_SCRATCH_BYTES that each GPU the library runs
 * on holds until the process ends, instead of the context's allocation.  The
 * call issues no hipMalloc/hipFree, creates no events and takes as many walkers
 * per launch group as fit the array (slamhip_static_info).  Like every batch
 * entry point it enqueues, per launch group, kernels, one memset and one 4-byte
 * copy into pinned host memory that slamhip_create allocated (a launch-sizing
 * hint for later groups; nothing waits for it).
 * Limits: scaling 'uniform' on the fused path only (mom0decon == NULL, conv
 * modes 0, 2 and 3 -- every mode but the dense one); a fit that needs more than the array for ONE walker
 * (roughly 8*nv*nneed^2 + 64*2.5*npix^2 bytes: cubes up to about 256x256x64)
 * is rejected with SLAMHIP_ERR_NOMEM and a message naming both sizes.
 * All contexts on one GPU share the array and it keeps the last call's
 * contents (every launch group re-initialises what it reads): calls that overlap
 * on one GPU -- from any context, on any stream -- race; order them yourself. */
#define SLAMHIP_STATIC_SCRATCH_BYTES (64ull << 20)
int slamhip_chi2_batch_static(slamhip_ctx* ctx, const double* params_dev, int nw,
                              double* chi2_dev, void* stream);
/* Size of the static array, bytes one walker of this fit needs in it, walkers
 * per launch group of slamhip_chi2_batch_static (0: the fit does not fit). */
int slamhip_static_info(slamhip_ctx* ctx, int64_t* scratch_bytes,
                        int64_t* bytes_per_walker, int32_t* walkers_per_group);

/* Hot path with its intermediate sums exposed (parity tap for get_scale,
 * :485-490): per walker {gf, ff, scale, chi2, non-empty tiles, tiles}.
 * `params` [nw][13], `stats_out` [nw][6] are host pointers. Synchronous. */
int slamhip_chi2_stats(slamhip_ctx* ctx, const double* params, int nw,
                       double* stats_out);

/* Replaces `ChannelFit.cubemodel(*p, convolving)` (:499-537): writes the
 * scaled model cube [nv][ny][nx] to the host buffer `out`.
 * `stats_out` (may be NULL) receives {gf, ff, scale, chi2} (:485-490, :601). */
int slamhip_cubemodel(slamhip_ctx* ctx, const double* params, int convolving,
                      double* out, double* stats_out);

/* Replaces the data half of `ChannelFit.plotmom` (:720-737): the model cube of
 * `params` as slamhip_cubemodel(convolving = 1) builds it, reduced over its
 * channels on the device as `makemom01` (:40-50) does: mom0 = sum_k(d) * dv,
 * mom1 = sum_k(d * v) / sum_k(d), NaN where mom0 < clip (the caller passes
 * 3 * sigma_mom0).  Channels are added in order and products rounded first, so
 * both maps equal numpy's on the same cube bit for bit.  `mom0_out`, `mom1_out`
 * are host buffers [ny][nx].  Synchronous. */
int slamhip_moments(slamhip_ctx* ctx, const double* params, double dv, double clip,
                    double* mom0_out, double* mom1_out);

/* Parity taps.  `get_Iunif` output (:454-473) [nv][nneed][nneed], and the
 * beam-convolved cube (:530) on the same grid. */
int slamhip_iunif(slamhip_ctx* ctx, const double* params, double* out);
int slamhip_iconv(slamhip_ctx* ctx, const double* params, double* out);

/* Per-surface geometry tap (:382-452): for surface slot s (0..3 = x1(h1),
 * x2(h1), x1(h2), x2(h2)) writes xdisk and vlos (Mstar = 1) on the full nested
 * grid [nlayer][npix][npix]; returns 1 if the slot is empty (None in the
 * reference), 0 if written. */
int slamhip_surface(slamhip_ctx* ctx, const double* params, int slot,
                    double* xdisk_out, double* vlos_out);

/* Line-profile table the kernels use for `cs` (:412-427): prof[45], prof_d. */
int slamhip_prof(slamhip_ctx* ctx, double cs, double* prof_out,
                 double* prof_d_out);

/* Introspection: number of nested-grid samples that survive the layer merge,
 * walkers processed per launch group, 1 if the beam was factorised exactly. */
int slamhip_info(slamhip_ctx* ctx, int64_t* nleaf, int32_t* walkers_per_group,
                 int32_t* separable_beam);

/* Introspection for the accounting in bench.py (no counterpart in the
 * reference): output rows and columns of one beam-convolution window of the
 * batch path (the fused kernel's window on the model grid; with a 'mom0*'
 * scaling the beam window on the observed grid; rows = 0 if the path has no
 * windowed kernel), the beam's tap count padded to the kernels' unroll, and 1 if
 * walkers whose regrid is a constant-fraction shift get it folded into the beam
 * taps (K+1 taps per axis). */
int slamhip_window_shape(slamhip_ctx* ctx, int32_t* rows, int32_t* cols, int32_t* kpad,
                         int32_t* fold);

/* Number of outer-product terms the fused convolution uses for the beam: 1 for
 * an axis-aligned Gaussian; 0 = direct K x K taps, which is what a rotated
 * Gaussian gets (its factorisation, singular triplets down to a 1e-13 tail,
 * takes ~14 terms: more arithmetic than the taps themselves). */
int slamhip_beam_rank(slamhip_ctx* ctx);

/* Selects the convolution kernel: 0 = auto (fused kernel; factorised beam or
 * direct taps, see slamhip_beam_rank), 1 = force the dense, non-fused direct
 * 2-D path, 2 = fused kernel with the factorised beam even where 0 would take
 * direct taps, 3 = fused kernel without folding the regrid into the beam taps
 * (every walker keeps the convolved tile + bilinear regrid), 4 = as 0 but the
 * phase-by-phase window kernel (k_conv_chi2) where 0 runs the row-owning one
 * (k_conv_rows: one-term beams of 25..32 padded taps, or of 80 with K <= 77, in
 * double and float; chi2 bitwise the same on 64-row windows, to 1e-12 on 48-row
 * ones, whose waves split the output rows differently), and with a 'mom0*'
 * scaling the older tile kernel of the beam pass where 0 lists 64 x 64 windows
 * for the row-owning one (cross-checks). */
int slamhip_set_conv_mode(slamhip_ctx* ctx, int mode);

/* Measurement aid (no counterpart in the reference): between _begin and _end
 * every kernel launch of this context is bracketed by HIP events recorded on
 * the stream it is launched on.  _end waits for them and returns, per kernel
 * id 0..slamhip_kernel_count()-1, the summed duration [ms] and launch count. */
int slamhip_kernel_count(void);
const char* slamhip_kernel_name(int i);
int slamhip_profile_begin(slamhip_ctx* ctx);
int slamhip_profile_end(slamhip_ctx* ctx, double* total_ms, int64_t* launches);

/* Measurement aid (no counterpart in the reference): what the context's LAST
 * launch group of `nw` walkers touched and computed, counted from the scratch
 * it left: out[0] live conv windows, [1] occupied 32x8 Iunif tiles under them
 * (halo included), [2] window rows with emission, [3] tap blocks executed (the
 * column pass's 8-row x 8-tap blocks of the factorised beam; for direct taps
 * the (16- or 8-row output block, 8-tap-row block) pairs that are not skipped,
 * each run once per tap column), [4] lit (tile, channel) planes
 * of Iunif, [5] stored geometry groups (64 samples x surface), [6] k_profile
 * hits (lane, sample, surface, 16-channel block), [7] the same per 64-sample
 * wave (at least one lane hit).  With a 'mom0*' scaling out[0]-[3] describe the
 * 64 x 64 beam windows on the observed grid that the row-owning kernel ran
 * (tiles, rows and tap blocks of the regridded cube); where the older tile
 * kernel runs (rotated beam, other tap counts, conv mode 4) out[0] is the
 * number of its windows that held emission and [1]-[3] are 0.
 * bench.py prices the kernels' compulsory bytes and useful FP64 operations
 * from these.  Not available in the dense conv mode (1). */
int slamhip_group_counts(slamhip_ctx* ctx, int nw, int64_t* out);

const char* slamhip_last_error(slamhip_ctx* ctx);
void slamhip_destroy(slamhip_ctx* ctx);

/* Library-level: number of visible GPUs (0 if none), version string.
 * The version string ends in " src=<16 hex digits> stamps=<0|1> variant=<name>":
 * the hash of the sources the binary was built from (this header + slamhip.hip,
 * see slam_amd/build.py), whether it is the diagnostic phase-stamp build and
 * which build it is ("product", or the name of a diagnostic / experiment build,
 * which lives in its own file libslamhip_<name>.so and is never the product).
 * The Python binding refuses a library whose hash differs from the sources next
 * to it or whose variant is not the one asked for. */
int slamhip_device_count(void);
const char* slamhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLAMHIP_H */
