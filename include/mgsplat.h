/*
 * mgsplat.h -- C ABI of libmgsplat.so, the MI355X (gfx950) Gaussian-splatting hot path.
 *
 * This is the drop-in boundary.  Each entry point names the reference interface it replaces.
 *   RAST = third_party/gaussian-splatting/submodules/diff-gaussian-rasterization  (reference tree)
 *   MG   = agents/manigaussian_bc                                                 (reference tree)
 *
 * Reference FFI today (pybind11, torch types): RAST/ext.cpp:15-19 exports
 *   rasterize_gaussians           -> RasterizeGaussiansCUDA          (RAST/rasterize_points.cu:35-128)
 *   rasterize_gaussians_backward  -> RasterizeGaussiansBackwardCUDA  (RAST/rasterize_points.cu:130-225)
 *   mark_visible                  -> markVisible                     (RAST/rasterize_points.cu:227-246)
 * which wrap CudaRasterizer::Rasterizer::{forward,backward,markVisible}
 * (RAST/cuda_rasterizer/rasterizer.h:20-92).
 *
 * Differences that the C ABI forces, all documented in INTEGRATION.md:
 *   - no torch types: plain device pointers + sizes; the CALLER owns every buffer (outputs and the
 *     three opaque workspaces geom/binning/img that the reference grows through a std::function
 *     callback, RAST/rasterize_points.cu:27-33).  Sizes come from mgs_*_bytes().
 *   - the reference's forward is split in two calls because the binning workspace is sized by
 *     num_rendered, which the reference reads back mid-call (RAST/cuda_rasterizer/rasterizer_impl.cu:284).
 *   - every call takes an explicit hipStream_t (the reference uses the legacy default stream).
 *   - F (feature channels) is a run-time value; the reference fixes it at compile time
 *     (RAST/cuda_rasterizer/config.h:16).
 * All tensors are contiguous row-major float32 unless noted, on the device the stream belongs to.
 * All functions return 0 on success, <0 on error (message via mgs_last_error(), thread-local).
 */
#ifndef MGSPLAT_H_
#define MGSPLAT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGS_ABI_VERSION 8

/* error codes */
#define MGS_OK 0
#define MGS_ERR_INVALID_ARG (-1)   /* bad shape / null pointer / unsupported F */
#define MGS_ERR_HIP (-2)           /* a HIP runtime call or (debug=1) a kernel failed */
#define MGS_ERR_WORKSPACE (-3)     /* a caller-provided workspace is too small */
#define MGS_ERR_NON_RGB (-4)       /* reference: "For non-RGB, provide precomputed Gaussian colors!" */
#define MGS_NEED_CAPACITY 1        /* the binning workspace holds fewer instances / chunk records than this scene needs */
#define MGS_PENDING 2              /* mgs_forward_result: the device has not reported yet */
#define MGS_RETRY_TABLE_INIT 3     /* mgs_forward_result (ABI v8): the preprocess launch's table hand-shake gave up (a workgroup
                                      made no progress for ~1 s): nothing was binned, the images are background only.  Run the
                                      forward again on the same workspaces with opt.table_init = 1 (a blocking forward does that
                                      by itself; an asynchronous one reports it here instead of a generic MGS_ERR_HIP) */

#define MGS_MAX_FEATURE_CHANNELS 64

typedef void* mgs_stream_t; /* hipStream_t */

/* Per-call tuning.  Every switch selects another implementation of the same result contract; NONE of them changes the
 * layout of a workspace, so a forward and its backward may even run under different values.  There is no process-wide
 * option state: a zero-initialised MgsRasterArgs carries `set == 0`, which means "all defaults" (mgs_options_default). */
typedef struct MgsOptions {
  int32_t set;          /* 0: ignore the fields below and use the defaults                                        */
  int32_t tight_bins;   /* 1*: drop (Gaussian, tile) instances whose alpha >= 1/255 footprint misses the tile      */
  int32_t fast_exp;     /* 0*: the reference's exp, bit for bit (ocml expf); 1: v_exp_f32 (rel. error ~2e-7 |x|, -2.4 % time) */
  int32_t exact_cull;   /* 1*: exact ellipse-vs-block test on top of the bounding-box test in the render forward   */
  int32_t bin_mode;     /* 2*: per-tile tables in LDS (up to 4096 tiles; more: as 0) and the lists ordered by ONE bucket-rank
                           launch (one workgroup per tile, keys in registers / LDS: ABI v8); 1: tables in LDS, segment sort +
                           rank merge (rounds 2-5); 0: tables in memory, one atomic per instance, segment sort + rank merge --
                           any tile count.  Every mode yields the reference's per-tile order (depth bits, then index)     */
  int32_t seg;          /* 2048*: keys per LDS-sorted segment (512, 1024, 2048, 4096: for lists of >> 8192 per tile)  */
  int32_t gm_waves;     /* 12*: render backward (one workgroup per CU): 12 waves x 168 registers, two pixels per step;
                           16 / 8: the one-pixel-per-step forms of rounds 2-4 (16 x 128 / 8 x 256 registers)            */
  int32_t dbg;          /* 0*: diagnostics (256: phase timeline of the render forward, mgs_debug_read_trace; 512: test
                           hook -- the table-zeroing workgroup of the forward preprocess sleeps ~0.3 ms first; 1024: test
                           hook -- it never publishes the tables, the hand-shake gives up after ~1 s)                     */
  int32_t table_init;   /* 0*: the forward preprocess launch zeroes its own tile tables (workgroup 0 + a bounded hand-shake:
                           one launch fewer); 1: a zero-fill launch ahead of it -- no workgroup ever waits for another.
                           debug = 1 implies 1; a blocking forward whose hand-shake gave up re-runs itself with 1           */
} MgsOptions;
void mgs_options_default(MgsOptions* o);  /* fills every field, set = 1 */

/* Per-view configuration + inputs shared by forward and backward.
 * Mirrors the argument lists of Rasterizer::forward / ::backward (RAST/cuda_rasterizer/rasterizer.h:35-91)
 * and the fields of GaussianRasterizationSettings (RAST/diff_gaussian_rasterization/__init__.py:166-179). */
typedef struct MgsRasterArgs {
  int32_t P;               /* number of Gaussians (means3D.size(0))                                   */
  int32_t D;               /* active SH degree (settings.sh_degree)                                   */
  int32_t M;               /* SH coefficients per Gaussian (sh.size(1)), 0 when colors_precomp given  */
  int32_t F;               /* feature channels (language_feature.size(1)); ignored if !include_feature */
  int32_t W, H;            /* image size                                                              */
  float tanfovx, tanfovy;  /* may be NEGATIVE (PyRep focal convention)           */
  float scale_modifier;
  int32_t prefiltered;     /* reference traps the device if a culled point shows up; here: error flag */
  int32_t debug;           /* 1: synchronise + check after every stage (RAST auxiliary.h:166-173)     */
  int32_t include_feature;
  const float* background;      /* [3]                                                                */
  const float* means3D;         /* [P,3]                                                              */
  const float* shs;             /* [P,M,3] or NULL                                                    */
  const float* colors_precomp;  /* [P,3]  or NULL (exactly one of shs / colors_precomp)               */
  const float* language_feature;/* [P,F]  or NULL when !include_feature                               */
  const float* opacities;       /* [P,1]  (forward only; backward reads it from the geom workspace)   */
  const float* scales;          /* [P,3]  or NULL                                                     */
  const float* rotations;       /* [P,4]  or NULL (r,x,y,z), used un-normalised like the reference    */
  const float* cov3D_precomp;   /* [P,6]  or NULL (exactly one of scales+rotations / cov3D_precomp)   */
  const float* viewmatrix;      /* [16] transposed world->view, m[col*4+row]                          */
  const float* projmatrix;      /* [16] transposed full projection                                    */
  const float* campos;          /* [3]                                                                */
  /* opaque workspaces, caller-allocated device memory (uint8 tensors in the reference) */
  void* geom;    size_t geom_bytes;     /* >= mgs_geom_bytes(P, M, W, H)      */
  void* binning; size_t binning_bytes;  /* >= mgs_binning_bytes2(binning_capacity, chunk_pool, W, H, F)                */
  void* img;     size_t img_bytes;      /* >= mgs_img_bytes(W, H)             */
  /* Optional: the accumulator block of a LATER backward (its scratch followed by dL_dcolors and dL_dfeature,
   * contiguous, a multiple of 16 bytes).  Forward: if non-NULL the preprocess kernel zeroes it on the side, so the
   * backward needs no fill.  Backward: accum_prezeroed != 0 promises exactly that (and that nothing touched it since). */
  void* bwd_accum; size_t bwd_accum_bytes;
  int32_t accum_prezeroed;
  /* How the binning workspace is carved; forward and backward must pass the SAME pair (it is not an option: it describes
   * the buffer).  binning_capacity = instances it holds (0: the largest count that fits binning_bytes with a worst-case
   * chunk pool, i.e. a buffer sized by mgs_binning_bytes); chunk_pool = chunk records of the render state (0: the worst
   * case for that capacity, mgs_chunk_pool_max; a caller that remembers the `chunks_used` of earlier forwards of the same
   * scene can pass a fraction of it: ~6x less memory at BASELINE configs[2]). */
  int32_t binning_capacity;
  int32_t chunk_pool;
  uint32_t status_tag;     /* low 16 bits are echoed in the status words of an asynchronous forward                    */
  int32_t async_forward;   /* mgs_rasterize_forward[_views]: 1 = enqueue and return, see below                         */
  MgsOptions opt;
} MgsRasterArgs;

int mgs_abi_version(void);
const char* mgs_last_error(void);
/* 16 hex digits: SHA-256 prefix over the sources this binary was compiled from (csrc/Makefile).  Measurements kept under
 * profiles/ carry it, so that evidence can be matched to the binary being timed -- not to a working tree. */
const char* mgs_build_id(void);

/* Process-wide DIAGNOSTICS only -- never results, kernels or layouts (those are MgsOptions, per call).  The one key is
 * "profile": 0 off, 1 hipEvents around the render backward, 2 around every stage (mgs_profile_read). */
int mgs_set_option(const char* key, int value);
int mgs_get_option(const char* key);

/* Workspace sizes.  Replace required<GeometryState/ImageState/BinningState>()
 * (RAST/cuda_rasterizer/rasterizer_impl.h:65-72, rasterizer_impl.cu:155-194). */
size_t mgs_geom_bytes(int P, int M, int W, int H);
size_t mgs_img_bytes(int W, int H);
size_t mgs_binning_bytes(int R, int W, int H, int F);  /* F = feature channels rendered (0 if none); worst-case chunk pool */
size_t mgs_binning_bytes2(int R, int chunk_pool, int W, int H, int F);  /* explicit pool (0: worst case) */
/* Optional: bytes to ADD to the binning workspace (behind mgs_binning_bytes2 / mgs_views_binning_bytes2) so that the forward
 * preprocess writes the tile keys itself and the bin scatter launch disappears (P Gaussians per view, V views, V = 1 for the
 * single-view calls): tiles x P keys of 8 bytes, 51 MB at 100 000 Gaussians on 128 x 128.  0: not offered for this shape (more
 * than 4 096 tiles, or more than 512 MB).  A workspace without them works as before; a capacity of at least tiles x P / 2
 * instances (every worst-case workspace) gets the same path without them. */
size_t mgs_binning_direct_extra(int P, int V, int W, int H);
int mgs_chunk_pool_max(int R, int W, int H);           /* chunk records of the worst case: every chunk of every 8x8 block */
size_t mgs_backward_scratch_bytes(int P, int M, int F);

/* Forward, stage 1: preprocess + tile-count scan (K2, K3 of SURVEY.md 2b).
 * Replaces the first half of Rasterizer::forward (RAST/cuda_rasterizer/rasterizer_impl.cu:198-284).
 * Writes radii[P] (int32) and the geom workspace; *num_rendered (HOST int) receives the number of
 * (Gaussian, tile) instances of the reference's 3-sigma tile rects -- THE REFERENCE'S INTEGER (rasterizer_impl.cu:280-284),
 * whatever MgsOptions.tight_bins says (the instances actually binned are fewer under tight_bins = 1; the count is a safe
 * size for stage 2's workspace) -- this call synchronises the stream once, exactly where the reference does its blocking
 * cudaMemcpy (rasterizer_impl.cu:284). */
int mgs_rasterize_forward_preprocess(const MgsRasterArgs* a, int32_t* radii, int32_t* num_rendered,
                                     mgs_stream_t stream);

/* Forward, stage 2: per-tile depth-ordered instance lists (what duplicate-with-keys + radix sort + tile ranges produce in
 * the reference; here: key scatter, segment sort, rank merge) and the alpha-composite render (K4-K7).
 * Replaces rasterizer_impl.cu:286-355.  radii: the [P] int32 array stage 1 wrote.  out_color [3,H,W];
 * out_feature [F,H,W] (untouched if !include_feature).  Both are fully written (no pre-zeroing needed). */
int mgs_rasterize_forward_render(const MgsRasterArgs* a, int32_t num_rendered, const int32_t* radii,
                                 float* out_color, float* out_feature, mgs_stream_t stream);

/* Fused forward: stage 1 + stage 2 in one call with NO mid-call stream synchronisation (the reference blocks on a
 * cudaMemcpy at rasterizer_impl.cu:284; on MI355X that bubble costs more than the binning).  The binning workspace is
 * sized by the CALLER's guess (a->binning_capacity / a->chunk_pool, e.g. the high-water marks of earlier calls).
 *   host_status: 24 bytes (three 64-bit words) of PINNED, device-mapped host memory (hipHostMalloc / torch pin_memory),
 *     8-byte aligned, owned by this call until its result has been read; the device reports {tag, flags, instances binned}
 *     through word 0 and {tag, the reference's num_rendered} through word 2 as soon as the preprocess has run (word 2 is
 *     stored first: whoever sees word 0 sees word 2), and {tag, overflow, chunk records used} through word 1 when the render
 *     has finished.  The call sets the words to "pending" before enqueueing.  NULL: the call reads the counts back with a
 *     blocking copy instead (same results, slower, no chunk-pool report: a->chunk_pool must then be 0).
 *   a->async_forward == 0: returns once words 0 and 2 have arrived, i.e. it waits for the PREPROCESS only -- binning and
 *     render are enqueued and may still be running (ABI v7; v6 synchronised the stream here to read the reference's count):
 *     MGS_OK (images enqueued, *num_rendered set -- the reference's integer, see mgs_rasterize_forward_preprocess;
 *     mgs_forward_result reports the instances actually binned, which is what sizes a workspace) or MGS_NEED_CAPACITY
 *     (*num_rendered set, geom + radii valid, images NOT rendered: call mgs_rasterize_forward_render with a binning
 *     workspace of at least mgs_binning_bytes(*num_rendered, W, H, F)).
 *     A chunk-pool overflow (only possible with a->chunk_pool != 0) is reported by mgs_forward_result.
 *   a->async_forward == 1: enqueues everything and returns MGS_OK at once with *num_rendered = -1: no host-device
 *     synchronisation at all (the call can be captured into a HIP graph together with its backward).  If the scene outgrew
 *     the workspace the kernels render nothing valid and say so in the status words: the caller MUST look at
 *     mgs_forward_result before trusting the images -- at the latest when it next synchronises with the stream.  The
 *     backward may be enqueued (num_rendered = -1) before the result is known; it then computes on whatever the forward
 *     left and is equally invalid after an overflow. */
int mgs_rasterize_forward(const MgsRasterArgs* a, int32_t* radii, float* out_color, float* out_feature,
                          int32_t* num_rendered, uint64_t* host_status, mgs_stream_t stream);

/* Host-side decode of the status words of a forward (no HIP call, never blocks).  `a`: the arguments of that forward
 * (status_tag, binning_capacity, chunk_pool, binning_bytes and the shape are read).  Returns MGS_PENDING until both words
 * carry this call's tag; then MGS_OK, MGS_NEED_CAPACITY (instances > capacity, or the chunk pool overflowed: the images
 * and any backward of that forward are invalid), MGS_RETRY_TABLE_INIT (see the code) or MGS_ERR_INVALID_ARG (prefiltered
 * violation).  *num_rendered and
 * *chunks_used (each optional) receive the counts as soon as their word has arrived (-1 before); *ref_rendered (optional)
 * the reference's num_rendered (the 3-sigma-rect instances, RAST/cuda_rasterizer/rasterizer_impl.cu:280-284): the same
 * integer the blocking entry points return, so every path hands the caller one number. */
int mgs_forward_result(const MgsRasterArgs* a, const uint64_t* host_status, int32_t* num_rendered, int32_t* chunks_used,
                       int32_t* ref_rendered);

/* Backward (K8-K10).  Replaces Rasterizer::backward (rasterizer_impl.cu:359-463) and the output
 * allocation of RasterizeGaussiansBackwardCUDA (rasterize_points.cu:167-184).  Every non-NULL output
 * is fully written (zero where the reference leaves its zero-initialised value).
 *   dL_dout_color [3,H,W], dL_dout_feature [F,H,W] (NULL if !include_feature)
 *   dL_dmeans2D [P,3] (NDC units, z = 0), dL_dopacity [P,1], dL_dcolors [P,3], dL_dfeature [P,F],
 *   dL_dmeans3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscales [P,3], dL_drotations [P,4],
 *   dL_dconic [P,4] (optional, may be NULL; reference keeps it internal).
 * scratch: >= mgs_backward_scratch_bytes(P, M, F) device bytes, contents undefined on entry.
 * a->language_feature must be 16-byte aligned (feature rows are read as float4; MGS_ERR_INVALID_ARG otherwise). */
/* num_rendered: the forward's count, or -1 if the caller has not looked yet (asynchronous forward). */
int mgs_rasterize_backward(const MgsRasterArgs* a, int32_t num_rendered, const int32_t* radii,
                           const float* dL_dout_color, const float* dL_dout_feature, float* dL_dmeans2D,
                           float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dfeature,
                           float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                           float* dL_drotations, void* scratch, size_t scratch_bytes, mgs_stream_t stream);

/* ---- multi-view batches: V views of one Gaussian set in one call (the reference renders one view per
 * call, MG/neural_rendering.py:386 `assert bs == 1`).  MgsRasterArgs carries everything shared (its viewmatrix / projmatrix /
 * campos / tanfov fields are ignored); images are [V,3,H,W] / [V,F,H,W]; radii, dL_dmeans2D, dL_dconic are [V,P,.];
 * dL_dcolors is [V,P,3] with SH colours (colours differ per view) and [P,3] with colors_precomp; every other gradient is
 * per Gaussian, summed over the views on the device.  Workspaces are sized by the mgs_views_*_bytes functions.
 * host_status is required (see mgs_rasterize_forward; async_forward works the same).  Needs V <= 16 (any tile count; V * tiles <= 4096 keeps the
 * binning tables in LDS). */
typedef struct MgsView {
  float tanfovx, tanfovy;
  const float* viewmatrix;  /* [16] */
  const float* projmatrix;  /* [16] */
  const float* campos;      /* [3]  */
} MgsView;
size_t mgs_views_geom_bytes(int P, int M, int W, int H, int V);
size_t mgs_views_img_bytes(int W, int H, int V);
size_t mgs_views_binning_bytes(int R, int W, int H, int F, int V);
size_t mgs_views_binning_bytes2(int R, int chunk_pool, int W, int H, int F, int V);
int mgs_views_chunk_pool_max(int R, int W, int H, int V);
size_t mgs_views_backward_scratch_bytes(int P, int M, int F, int V);
int mgs_rasterize_forward_views(const MgsRasterArgs* a, int32_t V, const MgsView* views, int32_t* radii, float* out_color,
                                float* out_feature, int32_t* num_rendered, uint64_t* host_status, mgs_stream_t stream);
int mgs_forward_result_views(const MgsRasterArgs* a, int32_t V, const uint64_t* host_status, int32_t* num_rendered,
                             int32_t* chunks_used, int32_t* ref_rendered);  /* mgs_forward_result for a batch of V views */
int mgs_rasterize_backward_views(const MgsRasterArgs* a, int32_t V, const MgsView* views, int32_t num_rendered,
                                 const int32_t* radii, const float* dL_dout_color, const float* dL_dout_feature,
                                 float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                                 float* dL_dfeature, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                                 float* dL_drotations, void* scratch, size_t scratch_bytes, mgs_stream_t stream);

/* Replaces Rasterizer::markVisible (rasterizer_impl.cu:141-153).  present: uint8 [P] (torch.bool). */
int mgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, mgs_stream_t stream);

/* ---- deformation field: fused input assembly and apply epilogue (MG/models_embed.py:255-304) ----
 * dyna_input[n, :] = cat(point_latent[n,0:DL], xyz[n,3], f_dc[n,3], f_rest[n,9], rot[n,4], scale[n,3],
 *                        opacity[n,1], (feature[n,3] iff feature != NULL), PE(canon_xyz[n])[39] or z_feature,
 *                        action[0, 0:DA] broadcast)                         (models_embed.py:258-287)
 * z_feature is passed precomputed ([N, DZ], models_embed.py:208-213); sh is [N,4,3] = (f_dc, f_rest).
 * Row stride of out = DL + 23 + (feature?3:0) + DZ + DA. */
int mgs_deform_assemble_forward(int N, int DL, int DZ, int DA, const float* point_latent,
                                const float* xyz, const float* sh, const float* rot, const float* scale,
                                const float* opacity, const float* feature, const float* z_feature,
                                const float* action, float* out, mgs_stream_t stream);
/* Gradient of the assembly: only point_latent and z_feature receive gradient (everything else is
 * .detach()ed in the reference).  g_out [N, stride] -> g_point_latent [N,DL], g_z_feature [N,DZ]. */
int mgs_deform_assemble_backward(int N, int DL, int DZ, int DA, int has_feature, const float* g_out,
                                 float* g_point_latent, float* g_z_feature, mgs_stream_t stream);
/* next.xyz = xyz + delta[:,0:3]; next.rot = normalize(rot + delta[:,3:7])   (models_embed.py:295-299) */
int mgs_deform_apply_forward(int N, const float* xyz, const float* rot, const float* delta /*[N,7]*/,
                             float* xyz_out, float* rot_out, mgs_stream_t stream);
/* g_delta [N,7] from g_xyz_out [N,3], g_rot_out [N,4] (xyz, rot are detached: no other gradient). */
int mgs_deform_apply_backward(int N, const float* rot, const float* delta, const float* g_xyz_out,
                              const float* g_rot_out, float* g_delta, mgs_stream_t stream);

/* ---- deformation field MLP: the elementwise passes between the GEMMs of ResnetFC, fused -----------------------------------
 * (MG/.../resnetfc.py:10-62: ResnetBlockFC  x + fc_1(relu(fc_0(relu(x)))),  :65-177 ResnetFC).  The GEMMs stay with the
 * caller's BLAS; x, act, g_* are row-major [M, N] fp32, N a multiple of 4 with N/4 dividing 256.
 * forward:  relu_out = max(x, 0);  xb_out = x + bias[N]   (either output, and bias, may be NULL) */
int mgs_mlp_relu_bias(int M, int N, const float* x, const float* bias, float* relu_out, float* xb_out,
                      mgs_stream_t stream);
/* backward: g_out = g_pre * (act > 0) [+ g_res];  colsum[n] += sum_m g_out[m][n]  (the bias gradient; NULL: not wanted;
 * the caller zeroes it).  g_out may alias g_pre or g_res. */
int mgs_mlp_relu_backward(int M, int N, const float* g_pre, const float* act, const float* g_res, float* g_out,
                          float* colsum, mgs_stream_t stream);

/* ---- Gaussian-regressor epilogue (MG/models_embed.py:233-253, MG/gaussian_renderer/__init__.py:66-68) ----
 * raw [N,26] = xyz 3 | opacity 1 | scale 3 | rot 4 | f_dc 3 | feature 3 | f_rest 9 (the split of models_embed.py:121,139-141)
 *   xyz = xyz_in + raw.xyz;  opacity = sigmoid;  scale = min(exp, 0.05);  rot = normalize (eps 1e-12);
 *   sh [N,4,3] = (f_dc, f_rest);  feature = raw.feature;  feature_n = feature / (|feature| + 1e-12)
 * backward: any g_* may be NULL (treated as zero); g_raw [N,26] is fully written. */
int mgs_regress_epilogue_forward(int N, const float* raw, const float* xyz_in, float* xyz, float* opacity, float* scale,
                                 float* rot, float* sh, float* feature, float* feature_n, mgs_stream_t stream);
int mgs_regress_epilogue_backward(int N, const float* raw, const float* g_xyz, const float* g_opacity, const float* g_scale,
                                  const float* g_rot, const float* g_sh, const float* g_feature, const float* g_feature_n,
                                  float* g_raw, mgs_stream_t stream);

/* ---- per-point latent: voxel-feature trilinear gather + positional encoding (MG/models_embed.py:147-215, MG/utils.py:133-169) ----
 * out [N, C + 3 + 6K] = [ grid_sample(voxel [C,D,H,W], canon; align_corners, zero padding) | canon | sin/cos(pi 2^k canon) ],
 * canon = (xyz - bounds[0:3]) / (bounds[3:6] - bounds[0:3]); bounds is a HOST array of 6 floats.
 * backward: g_voxel [C,D,H,W] += d out[:, 0:C] / d voxel (atomic; zero it first); g_out rows are row_stride floats apart. */
int mgs_voxel_sample_pe_forward(int N, int C, int D, int H, int W, int K, float freq_factor, const float* bounds_host,
                                const float* voxel, const float* xyz, float* out, mgs_stream_t stream);
int mgs_voxel_sample_backward(int N, int C, int D, int H, int W, const float* bounds_host, const float* xyz,
                              const float* g_out, int row_stride, float* g_voxel, mgs_stream_t stream);

/* ---- camera calibration: replaces NeuralRenderer.get_novel_calib (MG/neural_rendering.py:205-248)
 * with getWorld2View2 / getProjectionMatrix / focal2fov (MG/graphics_utils.py:17-53) folded in.
 * c2w [V,16] = the saved cam2world extrinsics, K [V,9] = intrinsics, both row-major float32.  Outputs (any may be NULL):
 * world_view_transform [V,16], full_proj_transform [V,16] (the transposed matrices the rasterizer takes),
 * camera_center [V,3], fov [V,2] = (FovX, FovY) (negative for negative focal lengths, kept), tanfov [V,2] = tan(Fov/2).
 * mgs_novel_calib: device pointers, one launch on `stream`, no host synchronisation; *singular (device int32, optional,
 * zeroed by the caller) is set to 1 if some cam2world matrix is not invertible.
 * mgs_novel_calib_host: the same routine on host arrays (what a data-loader cache calls once per camera file). */
int mgs_novel_calib(int V, const float* c2w, const float* K, int W, int H, float znear, float zfar, float trans_x,
                    float trans_y, float trans_z, float scale, float* world_view_transform, float* full_proj_transform,
                    float* camera_center, float* fov, float* tanfov, int32_t* singular, mgs_stream_t stream);
int mgs_novel_calib_host(int V, const float* c2w, const float* K, int W, int H, float znear, float zfar, float trans_x,
                         float trans_y, float trans_z, float scale, float* world_view_transform,
                         float* full_proj_transform, float* camera_center, float* fov, float* tanfov);

/* Per-stage device timing (hipEvents on the caller's stream), enabled with
 * mgs_set_option("profile", 1) (render backward only) or 2 (every stage).  mgs_profile_read waits for the
 * recorded events, writes the summed milliseconds and launch counts per stage ([mgs_profile_num_stages()]),
 * and recycles the events when reset != 0. */
int mgs_profile_num_stages(void);
const char* mgs_profile_stage_name(int stage);
int mgs_profile_read(double* total_ms, int32_t* counts, int reset);

/* Diagnostic, BLOCKING (two small device->host copies + a stream synchronise): what the forward that last ran on a's
 * workspaces left for its backward, in the units that state is kept in -- *incidences = (8x8 pixel block, Gaussian) pairs of
 * the chunks some pixel of the block visited, *chunks = those 64-survivor chunks, *pixel_chunks = (pixel, chunk) pairs visited
 * (what the per-chunk state costs).  V = 0: a single-view forward, V > 0: a batch of V views.  bench.py prices the render
 * kernels' HBM traffic with them (the reference's dataflow has no such state: RAST/cuda_rasterizer/backward.cu:399-593
 * walks the tile lists again). */
int mgs_forward_stats(const MgsRasterArgs* a, int32_t V, int64_t* incidences, int64_t* chunks, int64_t* pixel_chunks,
                      mgs_stream_t stream);

/* Diagnostic: byte offsets, inside a geom workspace of mgs_geom_bytes(P, M, W, H) bytes, of what the forward preprocess wrote
 * per Gaussian: depths f32[P]; rec f32x4[2P] = {x, y (pixels), conic.x, conic.y}{conic.z, opacity, hx, hy}; rgb f32[3P];
 * cov3D f32[6P] (the reference's GeometryState: RAST/cuda_rasterizer/rasterizer_impl.h:30-46).  The parity tests compare
 * them bit for bit with the reference kernels' values. */
int mgs_debug_geom_layout(int P, int M, int W, int H, size_t* depths, size_t* rec, size_t* rgb, size_t* cov3D);

/* Diagnostic: where the binning of a forward left its per-tile lists.  `a` = that forward's arguments (its binning_bytes /
 * binning_capacity / chunk_pool describe the carving), V = 0 for a single view or the number of views of a batch.  Byte offsets
 * inside the binning workspace of keys_unsorted u64[capacity] (depth bits << 32 | id, tile-major, unordered inside a tile) and
 * point_list u32[capacity] (sorted ids), and inside the img workspace of ranges uint2[tiles] ({first, end} of each tile's slice).
 * The parity tests check that every tile's list is its slice of keys in (depth bits, id) order -- the reference's order
 * (RAST/cuda_rasterizer/rasterizer_impl.cu:306-320). */
int mgs_debug_binning_layout(const MgsRasterArgs* a, int32_t V, size_t* keys_unsorted, size_t* point_list, size_t* img_ranges,
                             int32_t* capacity);
/* ... and, when the forward preprocess wrote the keys itself ("direct" binning: the bucket rank, bin_mode 2, on a workspace
 * that has room for tiles x P keys -- see mgs_binning_direct_extra), where: *stride = P and tile t's unordered slice lies at
 * u64[*keys / 8 + t * P ...] of the binning workspace (its length is the tile's range); *stride = 0: the bin scatter kernel wrote
 * the compact keys_unsorted above. */
int mgs_debug_direct_keys(const MgsRasterArgs* a, int32_t V, size_t* keys, int32_t* stride);

/* Diagnostic: with MgsOptions.dbg = 256 the render forward stamps s_memtime per (workgroup < 512, wave, phase);
 * this copies the 512 * 16 * 24 uint64 stamps of the last forward to `host` (scripts/trace_fwd.py prints the timeline). */
int mgs_debug_read_trace(unsigned long long* host, size_t count);
/* ... and of the render backward: 512 * 16 * 16 stamps (first chunk of every wave; scripts/trace_bwd.py). */
int mgs_debug_read_trace_bwd(unsigned long long* host, size_t count);

/* Device self-test of the wave64 cross-lane primitives used by the render kernels (DPP rotations,
 * v_permlane16/32_swap butterflies).  Returns 0 if every primitive matches its definition. */
int mgs_selftest(mgs_stream_t stream);

/* Counter calibration (scripts/sq_counters.sh): one workgroup-per-CU launch of a kernel with a KNOWN instruction mix per
 * wave -- iters x (64 v_fma_f32 + 8 v_mfma_f32_32x32x2_f32 + 4 ds_read_b32) -- so that a rocprofv3 --pmc pass can be
 * validated against exact counts before its numbers for the render kernels are trusted.  sink: >= 256*1024 floats. */
int mgs_calibration_kernel(int iters, float* sink, mgs_stream_t stream);

/* ... and of the bucket-rank binning (bin_mode 2): 1024 workgroups x 16 stamps (scripts/trace_bin.py). */
int mgs_debug_read_trace_bin(unsigned long long* host, size_t count);

#ifdef __cplusplus
}
#endif
#endif /* MGSPLAT_H_ */
