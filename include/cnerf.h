/* cnerf.h — C ABI of libcnerf_hip.so: the MI355X (gfx950) NeRF render/train hot path.
 *
 * The reference (skhu101/ConsistentNeRF, nerf-pytorch-master/) has no FFI boundary: its hot path is
 * in-process Python composing ATen ops.  Every entry point below names the reference function
 * (file:line, H = run_nerf_helpers.py, R = run_nerf.py, V = run_nerf_view.py) whose composition of
 * ATen ops it replaces.  The Python drop-in surface (consistentnerf_amd/run_nerf.py) binds these
 * through ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit sizes, a `void* stream` (hipStream_t; NULL = default).
 *   - all tensors fp32 row-major contiguous unless a stride argument says otherwise; indices int64.
 *   - return 0 on success, <0 for an argument error (CNERF_E_*), >0 = hipError_t of a failed launch.
 *   - no entry point allocates, frees, or synchronises; workspaces are caller-provided and sized by
 *     the matching *_floats / *_bytes query.  Re-entrant across streams; no global mutable state.
 */
#ifndef CNERF_H
#define CNERF_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNERF_ABI_VERSION 6   /* 6: + the in-loop consistency step as ONE render whose row count lives on the device: cnerf_ss_batch (the combined
                                * batch + its live-row count), cnerf_mlp_fwd_live / cnerf_mlp_bwd_live / cnerf_mlp_bwd_pair_live (launches of a fixed
                                * capacity that stop at a device-side count), cnerf_closs_finish_ss2 (the two-segment loss tail); cnerf_closs gains
                                * a trailing `seg_row` (0 = the v5 behaviour), cnerf_ss_ref_rays' meta grows to 8 ints; otherwise v5 unchanged.
                                * 5: + cnerf_ss_ref_rays (the in-loop consistency block's warp / compaction / reference rays / occlusion mask as one
                                * launch), cnerf_closs_finish_ss (its primary terms folded into compositing); every v4 entry point unchanged.   4: + cnerf_sample_pixels (one-launch training batch of an image), the ConsistentNeRF losses folded into
                                * compositing (cnerf_closs, *_closs, cnerf_closs_finish); cnerf_masked_loss uses its workspace for
                                * batches > 16384 rays; every v3 entry point unchanged.   3: + in-kernel uniform streams (cnerf_rng, *_rng) and compositing with img2mse folded in (*_mse);
                                * every v2 entry point unchanged.   2: cnerf_adam_step takes its hyper-parameters as double; the *_pair, *_cam, *_bf entry points */

#define CNERF_OK 0
#define CNERF_E_ARG (-1)        /* null pointer / negative size / inconsistent shapes            */
#define CNERF_E_UNSUPPORTED (-2) /* architecture outside the compiled envelope (see cnerf_net)    */
#define CNERF_E_NODEVICE (-3)   /* no gfx950 device visible                                       */

/* Architecture of one NeRF MLP (H:67-130; created by create_nerf R:181-204).
 * Compiled envelope: W in {64,128,256}; 1 <= D <= 16 and D != skip+1; 3+6*multires <= 64;
 * 3+6*multires_views <= 32; use_viewdirs in {0,1}; output_ch in 4..8 (used only without viewdirs: rgb, sigma, spare channels). */
typedef struct cnerf_net {
  int32_t D;               /* trunk depth  (netdepth)                                        */
  int32_t W;               /* trunk width  (netwidth)                                        */
  int32_t multires;        /* L of the point encoding, -1 = identity (i_embed=-1)           */
  int32_t multires_views;  /* L of the direction encoding                                    */
  int32_t use_viewdirs;    /* view-dependent colour branch (H:116-126)                       */
  int32_t output_ch;       /* width of output_linear when use_viewdirs==0 (R:190)            */
  int32_t skip;            /* trunk layer after which gamma(x) is re-concatenated (4), or -1  */
} cnerf_net;

/* Number / order of the parameter tensors of a net, as in the reference state_dict minus the three
 * ConsistentNeRF scalars (H:79-84):  pts_linears.{0..D-1}.{weight,bias}, views_linears.0.{weight,bias},
 * then feature_linear, alpha_linear, rgb_linear (viewdirs) or output_linear.  Weights are [out,in]. */
#define CNERF_MAX_TENSORS 48
int cnerf_num_tensors(const cnerf_net* net);
/* shape of tensor i: rows (out), cols (in; 1 for a bias -> rows-long vector). */
int cnerf_tensor_shape(const cnerf_net* net, int i, int64_t* rows, int64_t* cols);

typedef struct cnerf_ptrs { float* p[CNERF_MAX_TENSORS]; } cnerf_ptrs;

const char* cnerf_strerror(int code);
int cnerf_abi_version(void);
/* 0 if device `dev` is a gfx950 part; fills name (<=63 chars), #CUs and LDS/CU bytes. */
int cnerf_device_info(int dev, char* name64, int* num_cus, int* lds_bytes);

/* ---- weights: nn.Parameter tensors <-> kernel layout ------------------------------------------ */
/* Floats in the packed buffer (forward K-grouped panels, transposed panels for dgrad, heads, biases). */
int64_t cnerf_packed_floats(const cnerf_net* net);
/* Re-pack the parameter tensors (device pointers, state-dict order above) into `packed`.  Called once
 * per optimiser step (replaces nothing in the reference; it is the price of the MFMA operand layout). */
int cnerf_pack_weights(const cnerf_net* net, const cnerf_ptrs* params, float* packed, void* stream);
/* The same for the two networks of one render_rays call (coarse R:362 + fine R:402) in ONE launch: after every optimiser step both
 * are stale together. */
int cnerf_pack_weights_pair(const cnerf_net* net0, const cnerf_ptrs* params0, float* packed0, const cnerf_net* net1,
                            const cnerf_ptrs* params1, float* packed1, void* stream);

/* ---- a3: sample placement  (render_rays R:355-382) --------------------------------------------- */
/* z[B,Nc] from near/far in rays[:,6:8]; t_vals[Nc] = linspace(0,1,Nc) supplied by the host so both
 * sides use identical constants; t_rand[B,Nc] (U[0,1), or NULL when perturb==0) drives the stratified
 * jitter; lindisp selects inverse-depth spacing. */
int cnerf_coarse_z(const float* rays, int ray_stride, int64_t B, int Nc, const float* t_vals,
                   const float* t_rand, int lindisp, float* z, void* stream);

/* The two torch.rand draws of render_rays — t_rand (R:376) and u (H:227) — generated INSIDE the kernels that consume them: a
 * counter-based stream (Philox4x32-10, one block per element; csrc/rng.hpp, restated in oracle/philox.py) indexed by the GLOBAL element
 * (row0 + ray) * cols + col, so that a rank holding rows [row0, row0 + B) of a sharded batch sees the rows the unsharded call sees.
 * Values lie on the 24-bit grid of [0, 1) (ATen's CPU torch.rand for float32).  state_dev != NULL: {seed, base offset} are read from
 * device memory (uint64[2]) and `offset` is added to the base — the form a captured hipGraph replays with fresh numbers. */
typedef struct cnerf_rng {
  uint64_t seed;              /* ignored when state_dev != NULL */
  uint64_t offset;            /* one value = one independent [rows, cols] stream */
  const uint64_t* state_dev;  /* NULL or device uint64[2] = {seed, base offset} */
  int64_t row0;               /* global row of this call's first ray */
} cnerf_rng;
/* out[rows, cols] = the stream itself (what the two entry points below consume; for tests and for callers that want the tensor). */
int cnerf_uniform_rng(const cnerf_rng* rng, int64_t rows, int cols, float* out, void* stream);
/* cnerf_coarse_z with t_rand[b, i] = stream element (row0 + b, i) of an [*, Nc] stream. */
int cnerf_coarse_z_rng(const float* rays, int ray_stride, int64_t B, int Nc, const float* t_vals, const cnerf_rng* rng,
                       int lindisp, float* z, void* stream);

/* ---- a5: stand-alone positional encoding (Embedder.embed H:15-45): x[M,3] -> out[M, 3+6L]. The render
 *      path does not use it (encodings are generated inside cnerf_mlp_fwd and never reach HBM). */
int cnerf_embed(const float* x, int64_t M, int L, float* out, void* stream);

/* ---- a4+a5+a6: positional encoding + MLP  (run_network R:37-52, Embedder H:15-63, NeRF.forward
 *      H:107-130), fused; fp32 MFMA (v_mfma_f32_32x32x2_f32), activations never leave the CU. ------ */
/* Points are either explicit (pts[M,3], M=B*S) or implicit o + d*z (pts==NULL, rays/z given, R:384).
 * View directions are rays[:, ray_stride-3 : ray_stride] (one per ray, broadcast over its S samples)
 * or `dirs[B,3]` when non-NULL.  raw[M, C], C = 4 with viewdirs else output_ch.
 * `stash` == NULL: inference.  Otherwise (training) the kernel also writes the activations the backward
 * needs; size = cnerf_mlp_stash_floats(net, M). */
int64_t cnerf_mlp_stash_floats(const cnerf_net* net, int64_t M);
int cnerf_mlp_fwd(const cnerf_net* net, const float* packed, const float* pts, const float* rays,
                  int ray_stride, const float* dirs, const float* z, int64_t B, int S, float* raw,
                  float* stash, void* stream);
/* NeRF.forward(x) itself (H:107-130): x_embedded[M, in_ch + in_ch_views] already holds gamma(x) | gamma(d) per
 * point (what callers of `model(embedded)` pass).  Same kernel, the encoding stage copies instead of computing;
 * the stash (and therefore cnerf_mlp_bwd with B=M, S=1) works unchanged. */
int cnerf_mlp_fwd_embedded(const cnerf_net* net, const float* packed, const float* x_embedded, int64_t M,
                           float* raw, float* stash, void* stream);
/* cnerf_mlp_fwd (rays form) over a batch padded to a fixed CAPACITY of B rays whose live row count sits in device memory
 * (live_rays[0] <= B, e.g. cnerf_ss_batch's `live`): the launch is sized for B — the host never reads the count, the step stays
 * free of host synchronisation and capturable as a hipGraph — and the 32-point tiles at or beyond live_rays[0] * S retire at once,
 * leaving ZERO raw outputs (sigma = 0: compositing gives such a ray weight 0) and no stash.  Training form only (stash required);
 * S must be a multiple of 32. */
int cnerf_mlp_fwd_live(const cnerf_net* net, const float* packed, const float* rays, int ray_stride, const float* z, int64_t B, int S,
                       float* raw, float* stash, const int32_t* live_rays, void* stream);
/* ---- OPT-IN reduced-precision inference forward (never the default path, never used for training) -------------------
 * The same fused encoding + MLP as cnerf_mlp_fwd (run_nerf.py:37-52, run_nerf_helpers.py:107-130), its GEMMs on the bf16
 * matrix cores with every operand split into `planes` bf16 terms and fp32 accumulation:
 *   planes = 1 plain bf16 (error per product ~2^-9), 2 "bf16x2": w0x0 + w0x1 + w1x0 (~2^-16), 3 "bf16x3": the 6 cross terms
 *   with i + j < 3 (~2^-23, fp32-like).  Encodings, biases, the sigma / rgb heads and accumulators stay fp32.
 * Networks with view directions, W in {128, 256}.  packed_bf: cnerf_packed_bf_bytes(net, planes) bytes, filled by
 * cnerf_pack_weights_bf from the same 2D+8 parameter tensors as cnerf_pack_weights. */
int64_t cnerf_packed_bf_bytes(const cnerf_net* net, int planes);
int cnerf_pack_weights_bf(const cnerf_net* net, const cnerf_ptrs* params, int planes, void* packed_bf, void* stream);
int cnerf_mlp_fwd_bf(const cnerf_net* net, const void* packed_bf, int planes, const float* pts, const float* rays,
                     int ray_stride, const float* dirs, const float* z, int64_t B, int S, float* raw, void* stream);
/* OPT-IN "bf16x3" TRAINING forward (second bench line only; the default training path is exact fp32): cnerf_mlp_fwd with its
 * GEMMs on the bf16 matrix cores at three planes per operand (the 6 cross terms, error per product ~2^-23: fp32-equivalent)
 * and the SAME training stash (fp32 activations + ReLU sign bits, cnerf_mlp_stash_floats floats) — cnerf_mlp_dgrad /
 * cnerf_mlp_wgrad (and their _pair forms) consume it unchanged.  packed_bf = cnerf_pack_weights_bf(net, params, 3, ...). */
int cnerf_mlp_fwd_bf_train(const cnerf_net* net, const void* packed_bf, const float* pts, const float* rays, int ray_stride,
                           const float* dirs, const float* z, int64_t B, int S, float* raw, float* stash, void* stream);
/* OPT-IN "bf16x3" dgrad: cnerf_mlp_dgrad (/ _pair) in the same three-plane arithmetic; reads the stash's sign bits, writes the
 * same gradient workspace (cnerf_mlp_bwd_ws_floats floats) that cnerf_mlp_wgrad (/ _pair) consumes.  packed_bf as above (its
 * three-plane form also carries the transposed panels). */
int cnerf_mlp_dgrad_bf(const cnerf_net* net, const void* packed_bf, const float* d_raw, int64_t B, int S, const float* stash,
                       float* workspace, void* stream);
/* OPT-IN "bf16x3" weight gradients: cnerf_mlp_wgrad (/ _pair) with the wide GEMMs (>= 8 32x32 tiles per wave) in the three-plane
 * arithmetic, the narrow ones (heads, encoding columns) exact fp32 in the same grid; same operands, workspace and reduction. */
int cnerf_mlp_wgrad_bf(const cnerf_net* net, int64_t B, int S, const float* stash, float* workspace, const cnerf_ptrs* grads,
                       int accumulate, void* stream);
int cnerf_mlp_wgrad_bf_pair(const cnerf_net* net0, int64_t B0, int S0, const float* stash0, float* workspace0,
                            const cnerf_ptrs* grads0, const cnerf_net* net1, int64_t B1, int S1, const float* stash1,
                            float* workspace1, const cnerf_ptrs* grads1, int accumulate, void* stream);
int cnerf_mlp_dgrad_bf_pair(const cnerf_net* net0, const void* packed_bf0, const float* d_raw0, int64_t B0, int S0,
                            const float* stash0, float* workspace0, const cnerf_net* net1, const void* packed_bf1,
                            const float* d_raw1, int64_t B1, int S1, const float* stash1, float* workspace1, void* stream);

/* Backward of the above (autograd of R:37-52 / H:107-130): d_raw[M,C] -> gradients of every parameter
 * tensor.  `grads` holds device pointers laid out like `params`; accumulate!=0 adds into them.
 * workspace size = cnerf_mlp_bwd_ws_floats(net, M). */
int64_t cnerf_mlp_bwd_ws_floats(const cnerf_net* net, int64_t M);
int cnerf_mlp_bwd(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S,
                  const float* stash, float* workspace, const cnerf_ptrs* grads, int accumulate,
                  void* stream);
/* Backward of TWO independent networks at once — the coarse and the fine network of a render_rays training step
 * (run_nerf.py:311-421; the fine level's sample depths are detached, run_nerf.py:397, so the two backward passes share
 * nothing once the forward is done): ONE activation-gradient grid (when both networks have the same architecture; two
 * otherwise), ONE weight-gradient grid and ONE reduction for both, so the smaller level does not pay its own ramp and
 * tail.  Results are identical to two cnerf_mlp_bwd calls.  The two gradient sets must be distinct tensors
 * (CNERF_E_ARG otherwise: one network serving both levels, run_nerf.py:402, takes two cnerf_mlp_bwd calls).
 * workspaceN = cnerf_mlp_bwd_ws_floats(netN, BN*SN) floats each. */
int cnerf_mlp_bwd_pair(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0,
                       const float* stash0, float* workspace0, const cnerf_ptrs* grads0,
                       const cnerf_net* net1, const float* packed1, const float* d_raw1, int64_t B1, int S1,
                       const float* stash1, float* workspace1, const cnerf_ptrs* grads1, int accumulate, void* stream);
/* cnerf_mlp_bwd / cnerf_mlp_bwd_pair of a forward that ran through cnerf_mlp_fwd_live: activation-gradient tiles stop at
 * live_rays[0] * S of their level, and the weight-gradient point ranges are cut over the live points only (both levels of the pair
 * belong to ONE ray batch: B0 == B1).  first_rayN (pair forms; 0 = all): level N's first `first_rayN` rays carry zero seeds and are
 * left out of its backward altogether — the primary rays of the one-render `--ss_loss` step on the coarse level when neither coarse
 * coin selects it (run_nerf_view_test.py:959,966) — pass the level's arrays UNSHIFTED.  Exact fp32. */
int cnerf_mlp_bwd_live(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S, const float* stash,
                       float* workspace, const cnerf_ptrs* grads, int accumulate, const int32_t* live_rays, void* stream);
int cnerf_mlp_bwd_pair_live(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0,
                            const float* stash0, float* workspace0, const cnerf_ptrs* grads0,
                            const cnerf_net* net1, const float* packed1, const float* d_raw1, int64_t B1, int S1,
                            const float* stash1, float* workspace1, const cnerf_ptrs* grads1, int accumulate,
                            const int32_t* live_rays, int64_t first_ray0, int64_t first_ray1, void* stream);
int cnerf_mlp_dgrad_pair_live(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0,
                              const float* stash0, float* workspace0, const cnerf_net* net1, const float* packed1,
                              const float* d_raw1, int64_t B1, int S1, const float* stash1, float* workspace1,
                              const int32_t* live_rays, int64_t first_ray0, int64_t first_ray1,
                              void* stream);   /* the two halves of cnerf_mlp_bwd_pair_live */
int cnerf_mlp_wgrad_pair_live(const cnerf_net* net0, int64_t B0, int S0, const float* stash0, float* workspace0,
                              const cnerf_ptrs* grads0, const cnerf_net* net1, int64_t B1, int S1, const float* stash1,
                              float* workspace1, const cnerf_ptrs* grads1, int accumulate, const int32_t* live_rays,
                              int64_t first_ray0, int64_t first_ray1, void* stream);
/* Its two halves (cf. cnerf_mlp_dgrad / cnerf_mlp_wgrad). */
int cnerf_mlp_dgrad_pair(const cnerf_net* net0, const float* packed0, const float* d_raw0, int64_t B0, int S0,
                         const float* stash0, float* workspace0, const cnerf_net* net1, const float* packed1,
                         const float* d_raw1, int64_t B1, int S1, const float* stash1, float* workspace1, void* stream);
int cnerf_mlp_wgrad_pair(const cnerf_net* net0, int64_t B0, int S0, const float* stash0, float* workspace0,
                         const cnerf_ptrs* grads0, const cnerf_net* net1, int64_t B1, int S1, const float* stash1,
                         float* workspace1, const cnerf_ptrs* grads1, int accumulate, void* stream);

/* The two halves of cnerf_mlp_bwd, separately launchable (same workspace): activation gradients
 * (fused dgrad chain, one wave per 32 points) and weight gradients (point-contracted GEMMs + reduction). */
int cnerf_mlp_dgrad(const cnerf_net* net, const float* packed, const float* d_raw, int64_t B, int S,
                    const float* stash, float* workspace, void* stream);
int cnerf_mlp_wgrad(const cnerf_net* net, int64_t B, int S, const float* stash, float* workspace,
                    const cnerf_ptrs* grads, int accumulate, void* stream);

/* ---- a7: alpha compositing  (raw2outputs R:265-308 / V:392-438) -------------------------------- */
/* One wave64 per ray.  raw[B,S,C] (C>=4; channels 0..2 colour logits, 3 density), z[B,S], rays for
 * |rays_d|, noise[B,S] already scaled by raw_noise_std or NULL.  Outputs: rgb[B,3] disp[B] acc[B]
 * depth[B] weights[B,S] (any may be NULL).  Transmittance product scan carried in fp64 like the CPU
 * reference's cumprod. */
int cnerf_composite_fwd(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                        const float* noise, int64_t B, int S, int white_bkgd, float* rgb, float* disp,
                        float* acc, float* depth, float* weights, void* stream);
/* d_raw[B,S,C] (channels >=4 zeroed) from upstream grads (any may be NULL = 0). */
int cnerf_composite_bwd(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                        const float* noise, int64_t B, int S, int white_bkgd, const float* g_rgb,
                        const float* g_disp, const float* g_acc, const float* g_depth, float* d_raw,
                        void* stream);

/* a7 with `img2mse(rgb_map, target)` (H:9; R:769-775) folded in — the training step's two loss launches, their sum, and the two
 * `d_x * g` of their backward disappear.  Forward: as cnerf_composite_fwd (rgb required) + loss[0] = mean((rgb - target)^2)
 * (+ loss_add[0] when given: the other level's term, added in fp32 like `img_loss + img_loss0`).  Per-workgroup fp64 partial sums in
 * `workspace` (cnerf_composite_mse_ws_floats(B) floats, 8-byte aligned) are summed in index order by the workgroup that finishes
 * last (two-level tickets); `counter` = cnerf_composite_mse_counter_words() device uint32 words that are zero on entry and left zero
 * (one block per stream in flight); B <= cnerf_composite_mse_max_rays() (130 560), else CNERF_E_UNSUPPORTED.  Backward: d_raw from the seed
 * (2 / (3 B)) (rgb - target) * g_loss[0] (g_loss NULL = 1) formed in registers; bit-identical to cnerf_mse + cnerf_composite_bwd. */
int64_t cnerf_composite_mse_ws_floats(int64_t B);
int64_t cnerf_composite_mse_counter_words(void);
int64_t cnerf_composite_mse_max_rays(void);
int cnerf_composite_fwd_mse(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                            const float* noise, int64_t B, int S, int white_bkgd, const float* target, const float* loss_add,
                            float* rgb, float* disp, float* acc, float* depth, float* weights, float* loss, float* workspace,
                            unsigned* counter, void* stream);
int cnerf_composite_bwd_mse(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride,
                            const float* noise, int64_t B, int S, int white_bkgd, const float* rgb, const float* target,
                            const float* g_loss, float* d_raw, void* stream);

/* a7 with ConsistentNeRF's masked losses folded in (run_nerf_view.py:1645-1648 img_loss, :1737 depth_loss, :1786-1788 / :1865 the
 * coarse level's, :1678-1726 the monocular patch term): what `run_nerf_view.render_loss` launches per step instead of 2 x
 * cnerf_masked_loss + 2 x cnerf_patch_depth_loss + the ATen glue between them.
 *   forward, per level:  cnerf_composite_fwd_closs = cnerf_composite_fwd + five fp64 partial sums per workgroup of 8 rays in
 *       `workspace` (cnerf_closs_ws_floats(B) floats, 8-byte aligned): squared colour error over mask == 1 and over mask == 0,
 *       squared (depth - prior) / far over mask == 1, the two counts.  mask NULL = every ray in the first set; prior NULL = no depth
 *       term.  No tickets, no atomics.
 *   cnerf_closs_finish (ONE workgroup, after the last level's forward): sums the partials of both levels in index order, forms
 *       img_loss = s1 / (3 n1) + coef s0 / (3 n0) [second term iff n0 > 0] and depth_loss = sd / n1 with the local counts or the
 *       caller's GLOBAL `counts[2]` (a batch sharded over ranks), evaluates the patch term of both levels (cnerf_patch_depth_loss's
 *       arithmetic; P <= 8 patches of n rays = the first P n rays of the batch) and accumulates in the reference's order:
 *       loss = rgb_w img_loss + patch_w patch + depth_w depth_loss (+ the same three of the coarse level).
 *       terms[8] = loss, img_loss, depth_loss, patch_loss, img_loss0, depth_loss0, patch_loss0, 0;  stats[8] = per level
 *       (2 / (3 n1), coef 2 / (3 n0), (2 / n1) / far, 0): the seed weights;  patch_d[levels][P n] = d patch_loss / d depth (NULL: not
 *       wanted).
 *   backward, per level:  cnerf_composite_bwd_closs = cnerf_composite_bwd with the seeds formed in registers:
 *       g_rgb = (w_m (rgb - target)) (rgb_w g), g_depth = (w_d (depth / far - prior / far)) (depth_w g) [mask == 1] + patch_d (patch_w g)
 *       [first n_patch_rays rays], g = g_loss[0] (NULL = 1) — the operations of cnerf_masked_loss / cnerf_patch_depth_loss followed by
 *       autograd's `d * g`: bit-identical d_raw.  `stats` = the level's 4 floats of cnerf_closs_finish. */
typedef struct cnerf_closs {
  const float* target;   /* [B,3] */
  const float* mask;     /* [B] floats 0 / 1, or NULL */
  const float* prior;    /* [B] depth prior, or NULL (no depth term) */
  float far;             /* depth terms compare depth / far (V:1737) */
  int64_t seg_row;       /* 0, or (cnerf_closs_finish_ss2) the first ray of the batch's SECOND segment: cnerf_composite_bwd_closs then
                          * takes `stats` as [2][4] and seeds rays >= seg_row with the second set */
} cnerf_closs;
typedef struct cnerf_closs_sum {
  const float* ws_last;      /* workspace of the last (fine) level's forward */
  const float* ws_coarse;    /* the coarse level's, or NULL (one level) */
  int64_t B;
  const float* counts;       /* device (n1, n0) or NULL */
  float coef, far, rgb_w, depth_w, patch_w;
  int32_t has_depth;
  const float* depth_last;   /* depth maps (patch term; NULL with P == 0) */
  const float* depth_coarse;
  const float* mono;         /* [P n] monocular prior at the patch rays */
  int32_t P, n;
} cnerf_closs_sum;
int64_t cnerf_closs_ws_floats(int64_t B);
int cnerf_composite_fwd_closs(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride, const float* noise,
                              int64_t B, int S, int white_bkgd, const cnerf_closs* L, float* rgb, float* disp, float* acc,
                              float* depth, float* weights, float* workspace, void* stream);
int cnerf_closs_finish(const cnerf_closs_sum* t, float* terms, float* stats, float* patch_d, void* stream);
/* The same tail for the PRIMARY render of the in-loop consistency step (run_nerf_view_test.py:941-969): the levels' compositing
 * launches ran with mask = cnerf_ss_ref_rays' `sel` (the rays `x[mask_bound][mask]` selects), prior = the batch's depth priors and
 * far = 1; coins4 = the four random.randint(0, 1) draws in the reference's order (rgb, depth, rgb0, depth0): a term is the mean over
 * the selected rays when its coin is 1, else img2mse(rgb, target_s) for the two colour terms (ALSO for the coarse one: the
 * reference's line :959 falls back to the FINE rgb; its gradient goes to the fine level's seed weights) and absent for the depth
 * terms; loss = img_loss + depth + img_loss0 + depth0 in that order.  terms / stats as cnerf_closs_finish (no patch term, no
 * global counts: P must be 0, counts NULL). */
int cnerf_closs_finish_ss(const cnerf_closs_sum* t, const int32_t* coins4, float* terms, float* stats, void* stream);
/* The WHOLE `--ss_loss` step (run_nerf_view_test.py:899-969) rendered as ONE batch of t->B rays = [the N primary rays | the M <= N
 * warped rays of the second render + N - M padding rays] (cnerf_ss_batch), cut at seg_row = N (a multiple of the 8 rays of a
 * compositing workgroup): mask / target / prior of the compositing launches = cnerf_ss_batch's mask2 / target2 / prior2, far = 1.
 *   second segment  (:930-938)  img2mse(rgb_ref, rgb_target_ref) [+ img2mse(depth_pred_ref, rays_depth_ref)] of the fine, then of the
 *                               coarse level: means over the M live rows (mask 1; the padding rows carry mask 0 and weight 0)
 *   first segment   (:941-969)  the four coin-gated terms of cnerf_closs_finish_ss
 * accumulated in the reference's order (second render's terms first).  M is the device-side count of mask-1 rows of the second
 * segment — the host never needs it; counts3 (nullable, device) = the GLOBAL (selected primary rays, primary rays, warped rays) of a
 * batch sharded over ranks.  M == 0 (nothing projects into the reference view: the reference never leaves its threshold loop): the
 * second segment's terms and seed weights are 0, the primary terms take their un-masked branch.  terms12 = the 8 of cnerf_closs_finish_ss (terms12[7] = M as a float) + img_ref, depth_ref, img0_ref,
 * depth0_ref;  stats16 = per level [2 segments][4]: (w1, w0, wd, -) — level l passes stats16 + 8 l to cnerf_composite_bwd_closs. */
int cnerf_closs_finish_ss2(const cnerf_closs_sum* t, const int32_t* coins4, int64_t seg_row, const float* counts3, float* terms12,
                           float* stats16, void* stream);
int cnerf_composite_bwd_closs(const float* raw, int raw_ch, const float* z, const float* rays, int ray_stride, const float* noise,
                              int64_t B, int S, int white_bkgd, const cnerf_closs* L, const float* rgb, const float* depth,
                              const float* stats, const float* g_loss, float rgb_w, float depth_w, float patch_w,
                              const float* patch_d, int64_t n_patch_rays, float* d_raw, void* stream);

/* ---- a8: inverse-CDF sampling  (sample_pdf H:206-250) ------------------------------------------ */
/* bins[B,Nb], weights[B,Nb-1], u[B,Nf] (u_row_stride 0 broadcasts one row) -> samples[B,Nf];
 * inds[B,Nf] int64 (searchsorted right=True result, the bit-exact parity target) optional.
 * The pdf normaliser sum(weights + 1e-5) (H:213) is associated exactly as ATen's CPU `sum` kernel does it (8-float vectors, four
 * interleaved accumulators, tail, lanes in order) and the CDF scan is carried in fp64 like ATen's `cumsum`: the indices equal the
 * reference's CPU run bit for bit, CDF ties (the u = 1.0 sample of the deterministic test-time stream) included.
 * PLATFORM OF THAT GUARANTEE: the reference run on the CPU with ATen's AVX2 / AVX-512 sum kernels (x86-64; what torch 2.x
 * dispatches to on the hosts this was checked on).  A reference run on a GPU, or on a CPU whose ATen associates `sum`
 * differently (NEON, ATEN_CPU_CAPABILITY=default), differs from it — and from this library — by one ulp of the normaliser,
 * which moves the index of samples sitting exactly on a CDF tie; away from ties the indices are association-independent. */
int cnerf_sample_pdf(const float* bins, const float* weights, const float* u, int64_t u_row_stride,
                     int64_t B, int Nb, int Nf, float* samples, int64_t* inds, void* stream);
/* a8+a9 fused for render_rays (R:395-399,415): z_mid, sample_pdf(z_mid, weights[:,1:-1]), sort of the
 * Nc+Nf depths, population std of the new samples.  z_fine[B,Nc+Nf], z_std[B]; samples/inds optional. */
int cnerf_resample(const float* z, const float* weights, const float* u, int64_t u_row_stride, int64_t B,
                   int Nc, int Nf, float* z_fine, float* z_std, float* samples, int64_t* inds,
                   void* stream);

/* cnerf_resample with u[b, k] = stream element (row0 + b, k) of an [*, Nf] stream. */
int cnerf_resample_rng(const float* z, const float* weights, const cnerf_rng* rng, int64_t B, int Nc, int Nf, float* z_fine,
                       float* z_std, float* samples, int64_t* inds, void* stream);

/* ---- a3 as one call: render_rays (R:311-421 / V:441-551) and its autograd ---------------------------------------- */
/* The launch sequence of one ray batch — coarse_z -> [encoding + MLP](coarse) -> composite -> (Nf > 0:) resample ->
 * [encoding + MLP](fine) -> composite — and its backward (composite_bwd -> dgrad -> wgrad per level; no gradient
 * through the resampling, R:397), composed from the entry points above inside ONE caller-provided workspace of
 * cnerf_render_ws_floats() floats that also carries everything the backward needs (the "saved-for-backward blob").
 * Randomness stays the caller's: t_rand[B,Nc] (NULL: no jitter), u[B,Nf] (row stride 0 broadcasts one row),
 * noise0[B,Nc] / noise1[B,Nc+Nf] already scaled by raw_noise_std (NULL: none); t_vals[Nc] = linspace(0,1,Nc).
 * fine == NULL with Nf > 0 evaluates the second level with the coarse network (R:402) and its gradients add up. */
/* A pinhole camera whose rays are generated inside the kernels that consume them (get_rays run_nerf_helpers.py:164-173,
 * viewdirs = d/|d| of the pre-NDC direction run_nerf.py:103-110, ndc_rays run_nerf_helpers.py:186-202 with near plane 1):
 * ray i of a call is pixel (first + i) of the H x W image, row-major.  What render(c2w=...) (run_nerf.py:97-101) and
 * render_path (run_nerf.py:140-178) feed the renderer, without the [H*W, 11] ray tensor in HBM. */
typedef struct cnerf_raygen {
  int32_t H, W;
  float fx, fy, cx, cy;        /* K[0][0], K[1][1], K[0][2], K[1][2]                                   */
  float c2w[12];               /* camera-to-world [3,4], row-major                                      */
  float near, far;
  int32_t use_viewdirs, ndc;
  float ndc_ax, ndc_ay;        /* -1/(W/(2 focal)), -1/(H/(2 focal)) (run_nerf_helpers.py:193-199)      */
  int64_t first;
} cnerf_raygen;

typedef struct cnerf_render_cfg {
  int32_t Nc, Nf;        /* N_samples, N_importance                                      */
  int32_t lindisp;       /* sample linearly in inverse depth (R:362-364)                  */
  int32_t white_bkgd;    /* R:305-306                                                     */
  int32_t ray_stride;    /* 8 or 11 floats per ray: o, d, near, far, (viewdirs)           */
  int32_t train;         /* keep the stashes: cnerf_render_bwd may follow                 */
} cnerf_render_cfg;
typedef struct cnerf_render_out {            /* device pointers; every one may be NULL                      */
  float *rgb_map, *disp_map, *acc_map, *depth_map;   /* [B,3] [B] [B] [B] of the last level                 */
  float *rgb0, *disp0, *acc0, *depth0;               /* coarse level (Nf > 0 only)                           */
  float *z_std;                                      /* [B] (Nf > 0 only, R:415)                             */
  float *raw, *z_vals, *weights;                     /* last level: [B,S,C] [B,S] [B,S], S = Nc + Nf         */
} cnerf_render_out;
typedef struct cnerf_render_grads {          /* upstream gradients of the maps above (NULL = zero)          */
  const float *g_rgb_map, *g_disp_map, *g_acc_map, *g_depth_map, *g_rgb0, *g_disp0, *g_acc0, *g_depth0;
} cnerf_render_grads;
int64_t cnerf_render_ws_floats(const cnerf_net* coarse, const cnerf_net* fine, const cnerf_render_cfg* cfg, int64_t B);
int cnerf_render_fwd(const cnerf_net* coarse, const float* packed_coarse, const cnerf_net* fine,
                     const float* packed_fine, const float* rays, int64_t B, const cnerf_render_cfg* cfg,
                     const float* t_vals, const float* t_rand, const float* u, int64_t u_row_stride,
                     const float* noise0, const float* noise1, const cnerf_render_out* out, float* workspace,
                     void* stream);
/* Same nets / rays / cfg / noise / workspace as the forward call it follows (cfg->train != 0).  Parameter gradients
 * go to grads_coarse / grads_fine (layout of cnerf_ptrs as for cnerf_mlp_bwd); accumulate != 0 adds into them. */
int cnerf_render_bwd(const cnerf_net* coarse, const float* packed_coarse, const cnerf_net* fine,
                     const float* packed_fine, const float* rays, int64_t B, const cnerf_render_cfg* cfg,
                     const float* noise0, const float* noise1, const cnerf_render_grads* g, float* workspace,
                     const cnerf_ptrs* grads_coarse, const cnerf_ptrs* grads_fine, int accumulate, void* stream);

/* cnerf_render_fwd for the rays of a camera generated in-kernel: the chunk [cam->first, cam->first + B) of the image.
 * Inference (cfg->train must be 0; cfg->ray_stride is ignored: viewdirs per cam->use_viewdirs).  Same workspace size
 * (cnerf_render_ws_floats with ray_stride 11) and bit-identical outputs to cnerf_gen_rays + cnerf_render_fwd. */
int cnerf_render_fwd_cam(const cnerf_net* coarse, const float* packed_coarse, const cnerf_net* fine,
                         const float* packed_fine, const cnerf_raygen* cam, int64_t B, const cnerf_render_cfg* cfg,
                         const float* t_vals, const float* t_rand, const float* u, int64_t u_row_stride,
                         const float* noise0, const float* noise1, const cnerf_render_out* out, float* workspace,
                         void* stream);

/* ---- a1: ray generation  (get_rays H:164-173, ndc_rays H:186-202, render R:100-125) ------------- */
/* Builds rays[H*W, 8|11] = o, d, near, far, (viewdirs) for a full image from c2w[3,4] (12 floats,
 * HOST pointer) and K = fx, fy, cx, cy; viewdirs are normalised pre-NDC directions (R:103-110);
 * ndc!=0 applies ndc_rays with near-plane 1 (R:116) using the two host-computed coefficients
 * ndc_ax = -1/(W/(2 focal)), ndc_ay = -1/(H/(2 focal)) (H:193-199; computed on the host with the
 * caller's own scalar types so they round exactly as in the reference expression). */
int cnerf_gen_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host,
                   float near, float far, int use_viewdirs, int ndc, float ndc_ax, float ndc_ay,
                   float* rays, void* stream);
/* Same packing for a caller-provided batch rays_o[B,3], rays_d[B,3] (render(rays=...)). */
int cnerf_pack_rays(const float* rays_o, const float* rays_d, int64_t B, float near, float far,
                    int use_viewdirs, int ndc, float ndc_ax, float ndc_ay, float* rays, void* stream);

/* ---- f-2: the training batch of ONE image in one launch  (run_nerf_view.py:1452-1517, run_nerf.py:730-757) --------------- */
/* Rows [0, n_patches ps^2): the pixels of the patches (corner patch_start[q] = (row, col); inside a patch the ROW index runs
 * fastest, V:1490-1494); rows after them: n_rand DISTINCT pixels of the grid [crop_r0, crop_r0 + crop_h) x [crop_c0, crop_c0 +
 * crop_w) (the whole image, or the centre crop of R:741-753) — `select_inds[n_rand]` (device int64 indices into the row-major
 * grid: the caller's own `np.random.choice(..., replace=False)`, V:1503; values must be < crop_h crop_w) or, when NULL, pi(0..n_rand-1)
 * for a pseudo-random permutation pi of the grid keyed by `rng` (8-round alternating Feistel network over Philox4x32-10, cycle-walked;
 * oracle/philox.py::permutation).  Per row b, any output may be NULL: rays[B, 8|11] exactly as cnerf_gen_rays writes that pixel's ray,
 * rays_od[2, B, 3] = the raw (rays_o, rays_d) of get_rays (H:164-173; the reference's `batch_rays`), target[B, 3] = the first three
 * channels of image[H, W, image_ch], extras_out[n_extras, B] = extras[e][H, W] at the pixel (extras: HOST array of n_extras <= 4
 * device pointers), coords[B, 2] = (row, col). */
typedef struct cnerf_pixel_batch {
  int32_t H, W;
  float fx, fy, cx, cy;
  float c2w[12];
  float near, far;
  int32_t use_viewdirs, ndc;
  float ndc_ax, ndc_ay;
  int32_t crop_r0, crop_c0, crop_h, crop_w;
  int32_t n_patches, patch_size;
  int32_t patch_start[16][2];
  int64_t n_rand;
  int32_t image_ch, n_extras;
} cnerf_pixel_batch;
int cnerf_sample_pixels(const cnerf_pixel_batch* cfg, const int64_t* select_inds, const cnerf_rng* rng, const float* image,
                        const float* const* extras, float* rays, float* rays_od, float* target, float* extras_out,
                        int64_t* coords, void* stream);

/* ---- a12/a13: cross-view depth warp and hard masks  (get_ref_rays V:576-627, get_test_label
 *      V:630-669, mask precompute V:994-1046) ---------------------------------------------------- */
/* World points P[N,3] into the reference camera w2c[3,4] (HOST, 12 floats) with K (fx,fy,cx,cy):
 * Xc[N,3] (axis-flipped to OpenCV when flip!=0, V:596), px[N], py[N] (rounded half-to-even, as
 * floats), inb[N] (uint8, strict bounds V:611-613).  Any output may be NULL. */
int cnerf_warp_points(const float* P, int64_t N, const float* w2c_host, float fx, float fy, float cx,
                      float cy, int H, int W, int flip, float* Xc, float* px, float* py, uint8_t* inb,
                      void* stream);
/* One (target, reference) pair of the hard-mask precompute: for every pixel of the target view
 * (rays from c2w_tgt, depth prior depth_tgt[H*W]) warp into the reference view and test
 * |Xc_z - depth_ref[y,x]| < thr, thr = thr0 * 2^k with the smallest k>=0 that lets >=1 pixel of the
 * pixel's 5120-chunk pass (V:1023-1029).  mask[H*W] (uint8) is OR-ed in place (V:1041);
 * thr_out[ceil(H*W/chunk)] optional (NaN for chunks with no in-bounds pixel). */
int cnerf_hard_mask_pair(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_tgt_host,
                         const float* w2c_ref_host, const float* depth_tgt, const float* depth_ref,
                         float thr0, int chunk, uint8_t* mask, float* thr_out, void* stream);

/* ---- a15: the in-loop consistency block's ray construction as ONE launch (run_nerf_view_test.py:905-925 with get_ref_rays
 *      :451-501 and get_rays_ref run_nerf_view.py:553-574) -------------------------------------------------------------------------
 * The batch's depth-prior points rays_o + depth * rays_d [N] (:905) are projected into the reference camera (cfg->w2c = the
 * inverse of cfg->ref.c2w, computed by the caller as :910 does; flip = 0 for this variant, 1 = the OpenGL->OpenCV flip of
 * run_nerf_view.py:596); the M in-bounds ones, COMPACTED in batch order (the reference's `x[mask]`, a host synchronisation each),
 * give:
 *   rays_od[2][N][3] (first M rows of each half)  the rays of the reference camera through the snapped pixels: origin = camera centre,
 *                    direction = ((px - cx) / fx, (py - cy) / fy, 1) @ c2w[:3,:3]^T                                  (:491-493)
 *   rows[M][8|11]    the rows render() assembles from those rays for cfg->ref.near / far / use_viewdirs / ndc (run_nerf.py:100-125)
 *   target[M][3], depth_tgt[M]   colour (image[H][W][image_ch]) and depth prior (depth_ref[H][W]) of the reference view there  (:495-497)
 *   depth_diff[M]    |z in the reference camera - depth_tgt|                                                          (:923)
 *   mask[M]          depth_diff < thr, thr = thr0 * 2^k with the smallest k >= 0 that lets one point pass (:921-925: doubled until
 *                    mask.sum() > 0, a host synchronisation per doubling in the reference; k <= 63)
 *   inb[N]           mask_bound;   rank[N] = row of ray i among the M (or -1);   sel[N] = 1.0 where inb AND mask[rank] — the rays
 *                    `x[mask_bound.squeeze()][mask.squeeze()]` selects in the primary render's terms (:941-969) — else 0.0
 *   meta[8]          M, k, the bits of thr (float), 1 if some depth_diff is NaN (a NaN passes no threshold, as in the reference's
 *                    loop: the doubling runs on the minimum of the others; k = 63 and an empty mask only if NO |diff| is finite),
 *                    the bits of the minimum |diff| of THIS call's points (float; +inf without in-bounds points), the number of
 *                    primary rays `sel` selects, 2 unused
 * depth_diff, rank and meta are required (workspaces of the second sweep); rows, rays_od, target, depth_tgt, inb, mask, sel may be
 * NULL.  One workgroup, deterministic.  M = 0 is reported, not an error (the reference loops forever there). */
typedef struct cnerf_ss_warp {
  cnerf_raygen ref;      /* the reference camera and the bounds / flags of the render() call the rays are built for */
  float w2c[12];         /* world-to-camera [3,4] of the reference view, row-major (HOST) */
  int32_t flip;
  int32_t image_ch;      /* floats per pixel of `image` (>= 3) */
  float thr0;            /* args.occlusion_threshold */
} cnerf_ss_warp;
int cnerf_ss_ref_rays(const cnerf_ss_warp* cfg, const float* rays_o, const float* rays_d, const float* depth, int64_t N,
                      const float* image, const float* depth_ref, float* rows, float* rays_od, float* target, float* depth_tgt,
                      float* depth_diff, uint8_t* inb, uint8_t* mask, float* sel, int32_t* rank, int32_t* meta, void* stream);
/* The same launch assembling the ONE batch the whole `--ss_loss` step renders (run_nerf_view.ss_step_loss; :899-969), capacity 2 N rows:
 *   rows2[2N][8|11]  [0, N) the batch's own rays packed as render() packs them (run_nerf.py:100-125), [N, N + M) the reference rays
 *                    (cnerf_ss_ref_rays' rows), [N + M, 2N) padding: the reference camera's optical axis, a valid ray
 *   target2[2N][3]   target_s | the reference view's colours at the snapped pixels | 0
 *   prior2[2N]       depth (the batch's prior, :894) | the reference view's depth prior there | 0
 *   mask2[2N]        sel | 1 | 0      — the loss mask of the compositing launches (cnerf_closs, cnerf_closs_finish_ss2)
 *   live[1]          N + M            — for cnerf_mlp_fwd_live / cnerf_mlp_bwd_pair_live: nothing on the host waits for M
 * amin_global (nullable, device float): the minimum of |z - D_ref| over ALL ranks' in-bounds points of a batch sharded across ranks
 * (all-reduce MIN of meta[4] of a first call): the doubling rule then yields the global batch's threshold on every rank.
 * rays_od, inb, mask, sel may be NULL; depth_diff, rank, meta[8] as for cnerf_ss_ref_rays. */
int cnerf_ss_batch(const cnerf_ss_warp* cfg, const float* rays_o, const float* rays_d, const float* depth, const float* target_s,
                   int64_t N, const float* image, const float* depth_ref, const float* amin_global, float* rows2, float* target2,
                   float* prior2, float* mask2, int32_t* live, float* rays_od, float* depth_diff, uint8_t* inb, uint8_t* mask,
                   float* sel, int32_t* rank, int32_t* meta, void* stream);

/* img2mse (run_nerf_helpers.py:9): loss[0] = mean((x - y)^2) over n elements; d_x (nullable) = 2 (x - y) / n, the
 * gradient of the loss w.r.t. x.  One launch, fixed summation order. */
int cnerf_mse(const float* x, const float* y, int64_t n, float* loss, float* d_x, void* stream);
/* The same for whole images (H:9 on a rendered frame, R:836-845): `workspace` of cnerf_mse_ws_floats(n) floats (8-byte aligned) holds
 * one fp64 partial per 16384 elements, summed in index order by a second stage; n <= 16384 or workspace == NULL: cnerf_mse. */
int64_t cnerf_mse_ws_floats(int64_t n);
/* img2mse_softLpmask (run_nerf_view.py:58; the `--softLpmask` loss branch, :1663-1664, :1760-1761): loss[0] = sum(w d^2) / sum(w),
 * d = x - y, w = |d|^coef + 1 with the denominator detached; d_x (nullable) = the gradient w.r.t. x.  One launch, fixed order. */
int cnerf_soft_lp_loss(const float* x, const float* y, int64_t n, float coef, float* loss, float* d_x, void* stream);
int cnerf_mse_ws(const float* x, const float* y, int64_t n, float* loss, float* d_x, float* workspace, void* stream);

/* ---- a14: masked photometric / depth losses  (V:1645-1648, V:1737, V:1786-1788, V:1865) --------- */
/* loss[0] = mean_{m==1}(rgb-t)^2 + coef*mean_{m==0}(rgb-t)^2 (second term only if some m==0);
 * loss[1] = mean_{m==1}((depth-prior)/far)^2 (0 if depth==NULL).  Gradients d_rgb[B,3], d_depth[B]
 * (scaled by g_scale) are written if non-NULL.  mask==NULL = plain MSE over all rays (R:769).
 * counts[2] (n1, n0) may be supplied (e.g. all-reduced across ranks) or NULL to count locally.
 * `workspace` of cnerf_loss_ws_floats() floats (8-byte aligned) or NULL: with it, batches of more than 16384 rays (up to 16.7 M)
 * run as one workgroup per 16384 rays + a fixed-order second stage; otherwise one workgroup. */
int64_t cnerf_loss_ws_floats(void);
int cnerf_masked_loss(const float* rgb, const float* target, const float* depth, const float* prior,
                      const float* mask, int64_t B, float far, float coef, const float* counts,
                      float g_scale, float* loss, float* d_rgb, float* d_depth, float* workspace,
                      void* stream);

/* ---- f-5: monocular-depth patch term (V:1678-1720) --------------------------------------------- */
/* depth_pred[P*n] = rendered depth of the P patches' rays (n = 16*16 in the reference, P = 4 <= 16), mono[P*n] = the
 * monocular (MiDaS) inverse-depth prior at the same pixels.  loss[0] = sum_p mean_i((gtn_i - prn_i + alpha_p)^2) / P / 2
 * with gtn / prn the min-max normalised prior / clipped inverse depth over the prior's valid set (mono > 0) and
 * alpha_p = mean(prn - gtn).  d_depth[P*n] (optional) = g_scale * dloss/d depth_pred, autograd's tie rules. */
int cnerf_patch_depth_loss(const float* depth_pred, const float* mono, int P, int n, float g_scale,
                           float* loss, float* d_depth, void* stream);

/* ---- f-1: optimiser tail  (clip_grad_value_ V:1983, Adam R:210/780, lr decay R:784-788) --------- */
/* In-place Adam over n contiguous floats; clip<=0 disables the value clip; step is 1-based.  The hyper-parameters are
 * doubles, as torch.optim.Adam holds them: 1 - beta2 formed from a float-rounded 0.999 is off by 1.3e-5 (relative), which
 * would show in exp_avg_sq. */
int cnerf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, double lr,
                    double beta1, double beta2, double eps, float clip, float grad_scale, void* stream);

/* The same step with every scalar in DEVICE memory (hyp8_dev: 8 floats as cnerf_adam_hyper writes them on the host): nothing
 * of the step is baked into the launch, so a captured hipGraph of the whole training step can be replayed while the host
 * rewrites the pinned source of hyp8 between replays (graph.py). */
int cnerf_adam_hyper(int step, double lr, double beta1, double beta2, double eps, float clip, float grad_scale,
                     float* hyp8_host);
int cnerf_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyp8_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNERF_H */
