/* dynam3d_hip.h -- C ABI of libdynam3d_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the Dynam3D per-step 3D-token path.
 * The reference is pure Python; its native operators are un-vendored pip packages.  Each entry
 * point below names the reference call site (file:line under /root/reference) it replaces.
 *   VLN-FF  = Dynam3D_VLN/vlnce_baselines/models/feature_fields.py
 *   VLN-POL = Dynam3D_VLN/vlnce_baselines/models/Policy_Dynam3D_VLN.py
 *   PRE-FF  = Dynam3D_Pretrain/src_3dff/models/feature_fields.py
 *
 * Conventions
 *   - extern "C"; every function returns int32 (0 = ok, <0 = D3D_E*); d3d_last_error() returns a
 *     thread-local message.
 *   - all `*_d` / device buffers are caller-allocated DEVICE pointers (torch owns the memory; the
 *     library neither frees nor retains them).  Host pointers are named `*_h`.
 *   - sizes are int64/int32 as written; last argument is the hipStream_t (as void*) to launch on.
 *   - launches are asynchronous; no implicit synchronisation.
 *   - geometry kernels are compiled with -ffp-contract=off and are bit-exact against
 *     oracle/geometry.py (one rounding per operation, fixed order).
 */
#ifndef DYNAM3D_HIP_H
#define DYNAM3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3D_OK 0
#define D3D_EINVAL (-1)
#define D3D_EHIP (-2)
#define D3D_ECAP (-3)
#define D3D_ESTATE (-4)

#define D3D_TOMBSTONE (-10000.0f) /* VLN-FF:357 */
#define D3D_FTS_DIM 768

const char* d3d_last_error(void);
int32_t d3d_version(void);
/* number of CUs / wave size seen by the library (sanity: 256 / 64 on MI355X) */
int32_t d3d_device_info(int32_t* n_cu, int32_t* wave_size, int32_t* lds_bytes);

/* ------------------------------------------------------------------------------------------------
 * Per-view camera pose, prepared on the host exactly like the reference does in Python floats
 * (math.cos/sin in double, then rounded to float32 at first use).
 * ---------------------------------------------------------------------------------------------- */
typedef struct d3d_pose {
    float cam[3];   /* (x, -z, y) of the habitat position: VLN-FF:336, 523, 830 */
    float cos_h;    /* (float)cos(heading)   -- unprojection  VLN-FF:290-291 */
    float sin_h;
    float cos_nh;   /* (float)cos(-heading)  -- frustum / agent frame  VLN-FF:95-99, 831-836 */
    float sin_nh;
    float heading;  /* (float)heading, added to the per-patch direction  VLN-FF:289 */
} d3d_pose;

/* a1  Dynam3D_VLN.preprocess_depth (VLN-POL:171-186): zero pixels <- column max, then metres.
 * depth (B,H,W) f32 in [0,1] -> out (B,H,W) f32.  lo/hi = depth_scale. */
int32_t d3d_preprocess_depth(const float* depth_d, float* out_d, int32_t B, int32_t H, int32_t W,
                             float lo, float hi, void* stream);

/* a2+a1 fused for the 24x24 copy: cv2.resize(INTER_NEAREST) (VLN-POL:339) then preprocess_depth
 * on the resized image (VLN-POL:341).  src (B,H,W) -> out (B,h,w). */
int32_t d3d_resize_nearest_preprocess(const float* depth_d, float* out_d, int32_t B, int32_t H, int32_t W,
                                      int32_t h, int32_t w, float lo, float hi, void* stream);

/* a6  get_patch_segm after the segmenter network (VLN-FF:407-420): 'last mask wins' label image -> nearest (h,w) resize
 * (F.interpolate, ATen float32 index rule) -> labels replaced by their rank in torch.unique order.  masks_d (total,H,W) u8
 * {0,1}; image i owns masks [mask_off_d[i], mask_off_d[i+1]) (none -> all zeros, the reference's except branch,
 * VLN-FF:424-426); segm_d (n_img, h*w) i32, n_seg_d (n_img) = number of dense labels.  max_masks = host-side bound. */
int32_t d3d_patch_segm_from_masks(const uint8_t* masks_d, const int32_t* mask_off_d, int32_t n_img, int32_t max_masks,
                                  int32_t H, int32_t W, int32_t h, int32_t w, int32_t* segm_d, int32_t* n_seg_d, void* stream);

/* f-3  An on-device class-agnostic mask generator for get_patch_segm (VLN-FF:400-430) while FastSAM's weights are unavailable: grid-seeded
 * colour + position k-means (SLIC-style) over all pixels, one workgroup per image, deterministic (integer cluster sums; restated bit for
 * bit by oracle/segment_ref.py).  rgb_d (n_img,H,W,3) u8; gx * gy <= 64 seeds; `iters` Lloyd updates + the final assignment;
 * compactness m: distance = |d rgb|^2 + (m / S)^2 |d xy|^2 with S the seed spacing.  masks_d (n_img, gx*gy, H, W) u8 {0,1} -- the input of
 * d3d_patch_segm_from_masks (clusters without pixels give all-zero masks, which the relabel drops); labels_d (n_img,H,W) i32 or NULL. */
int32_t d3d_segment_slic(const uint8_t* rgb_d, int32_t n_img, int32_t H, int32_t W, int32_t gx, int32_t gy, int32_t iters, float compactness,
                         uint8_t* masks_d, int32_t* labels_d, void* stream);

/* camera tables for a5/a13 are uploaded once per camera setting by the host wrapper:
 * tan_xy[P], tan_z[P], dir0[P] (see oracle/geometry.py::camera_tables; VLN-FF:283-287). */

/* a5  project_depth_to_3d_habitat + world offset (VLN-FF:276-293, 550-554).
 * For env e (0..n_env-1): depth24_d[e*P .. ] metres, pose_d[e], writes P rows starting at
 * row_base_d[e] of the env's slot slot_d[e] in the row pools (pool strides in rows = n_cap):
 *   rows_pos (slots, n_cap, 3) f32, rows_dir / rows_scale (slots, n_cap) f32. */
int32_t d3d_unproject_append(const float* depth24_d, const d3d_pose* pose_d, const int32_t* slot_d,
                             const int32_t* row_base_d, int32_t n_env, int32_t P, int32_t W,
                             const float* tan_xy_d, const float* tan_z_d, const float* dir0_d, float tan_half_hfov,
                             float* rows_pos_d, float* rows_dir_d, float* rows_scale_d, int64_t n_cap, void* stream);

/* append the CLIP grid features of the frame as float16 rows (VLN-FF:500, 567-570):
 * grid (n_env, P, 768) f32|f16 -> rows_fts (slots, n_cap, 768) f16 at row_base. */
int32_t d3d_append_fts(const void* grid_d, int32_t grid_is_f16, const int32_t* slot_d, const int32_t* row_base_d,
                       int32_t n_env, int32_t P, uint16_t* rows_fts_d, int64_t n_cap, void* stream);

/* a13 get_patch_3d_info (VLN-FF:296-326): depth24 (N,P) -> 5 x (N,P) f32. */
int32_t d3d_patch_3d_info(const float* depth24_d, int32_t N, int32_t P, int32_t W, const float* tan_xy_d,
                          const float* tan_z_d, const float* dir0_d, float tan_half_hfov, float* rel_x_d,
                          float* rel_y_d, float* rel_z_d, float* dir_d, float* scale_d, void* stream);

/* a4  get_frustum_mask_habitat + depth test + tomb-stoning (VLN-FF:88-115, 347-360).
 * For env e: tests rows [0, n_rows_d[e]) of slot slot_d[e] against depth image e (Hd,Wd) metres.
 * Hits are tomb-stoned in place (pos=-10000, fts/dir/scale=0) and their row indices are appended
 * (unordered) to hits_d[e*hit_cap ..]; n_hits_d[e] must be zeroed by the caller. mask_d (optional,
 * may be NULL) receives a byte mask (n_env, n_cap). */
int32_t d3d_frustum_cull(float* rows_pos_d, uint16_t* rows_fts_d, float* rows_dir_d, float* rows_scale_d,
                         int64_t n_cap, const int32_t* slot_d, const int32_t* n_rows_d, int32_t n_env,
                         int32_t max_rows, const float* depth_d, int32_t Hd, int32_t Wd, const d3d_pose* pose_d,
                         float fx, float fy, float cx, float cy, float near_, float far_, float slack,
                         int32_t* hits_d, int32_t* n_hits_d, int32_t hit_cap, uint8_t* mask_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Intrinsics / extrinsics path (posed RGB-D datasets with a pinhole camera).
 * ---------------------------------------------------------------------------------------------- */
typedef struct d3d_pinhole_view {
    float view[12]; /* rows 0..2 of the 4x4 world->camera matrix `batch_extrinsic[b][ix]` (PRE-FF:693) */
    float K[9];     /* `batch_camera_intrinsic[b][ix][:3,:3]` row-major (PRE-FF:101) */
} d3d_pinhole_view;

/* get_frustum_mask + depth test + tomb-stoning (PRE-FF:98-118, 693-712): as d3d_frustum_cull with the camera given by a
 * view matrix and an intrinsics matrix per env.  einsum accumulation order = ATen's CPU matmul (acc = m0*a, then fmaf per
 * term), bit-exact against tests/golden/g2b_frustum_pinhole.npz. */
int32_t d3d_frustum_cull_pinhole(float* rows_pos_d, uint16_t* rows_fts_d, float* rows_dir_d, float* rows_scale_d,
                                 int64_t n_cap, const int32_t* slot_d, const int32_t* n_rows_d, int32_t n_env,
                                 int32_t max_rows, const float* depth_d, int32_t Hd, int32_t Wd,
                                 const d3d_pinhole_view* views_d, float near_, float far_, float slack, int32_t* hits_d,
                                 int32_t* n_hits_d, int32_t hit_cap, uint8_t* mask_d, void* stream);
int32_t d3d_frustum_mask_pinhole(const float* points_d, int64_t n, const float* depth_d, int32_t Hd, int32_t Wd,
                                 const d3d_pinhole_view* view_host, float near_, float far_, float slack, uint8_t* mask_d,
                                 void* stream);

typedef struct d3d_pinhole_unproject {
    double fx, fy, cx, cy; /* batch_camera_intrinsic[b][ix] (PRE-FF:84) */
    double R[9], T[3];     /* batch_rot[b][ix], batch_trans[b][ix]: camera -> world (PRE-FF:910-912) */
    float scale_tan;       /* |tan(rel_direction[0][-1])| of the view-sized rays (PRE-FF:849-856, 909) */
    float depth_scale;     /* Open3D depth_scale (raw units per metre) and depth_trunc (metres) */
    float depth_trunc;
    float pad_;
} d3d_pinhole_unproject;

/* project_depth_to_3d + world transform + get_heading_angle + append (PRE-FF:81-94, 905-916; the Open3D call restated
 * from its published algorithm -- PARITY UNPINNED, see oracle/geometry.py::project_depth_to_3d).  depth_d (n_env,Hd,Wd)
 * f32 RAW sensor units (zero pixels take the image maximum; values are cast to uint16 like the reference does); writes
 * h*w rows at row_base_d[e] of slot slot_d[e].  If any pixel of image e is invalid, all its points are zero (PRE-FF:90). */
int32_t d3d_unproject_pinhole_append(const float* depth_d, int32_t Hd, int32_t Wd, const d3d_pinhole_unproject* cams_d,
                                     const int32_t* slot_d, const int32_t* row_base_d, int32_t n_env, int32_t h,
                                     int32_t w, int32_t input_width, float* rows_pos_d, float* rows_dir_d,
                                     float* rows_scale_d, int64_t n_cap, void* stream);

/* stand-alone mask (no mutation) for one point set -- used by tests and by torch-free callers. */
int32_t d3d_frustum_mask(const float* points_d, int64_t n, const float* depth_d, int32_t Hd, int32_t Wd,
                         const d3d_pose* pose_h, float fx, float fy, float cx, float cy, float near_, float far_,
                         float slack, uint8_t* mask_d, void* stream);

/* a8  torch_kdtree build_kd_tree / .query replacement (VLN-FF:246, 606-610; PRE-FF:364, 540, 584).
 * Brute force, ascending (dist^2, index), ties -> lowest index; d2 = (dx*dx + dy*dy) + dz*dz.
 * Batched: batch b uses points_d + b*point_stride (n_points_d[b] valid rows of 3 floats),
 * queries_d + b*query_stride (n_queries_d[b] rows), k_d[b] <= k_max <= 8 neighbours;
 * writes d2_d / idx_d at (b*max_queries + q)*k_max + j.  Strides are in floats. */
int32_t d3d_knn(const float* points_d, int64_t point_stride, const int32_t* n_points_d, const float* queries_d,
                int64_t query_stride, const int32_t* n_queries_d, const int32_t* k_d, int32_t n_batch,
                int32_t max_queries, int32_t k_max, float* d2_d, int32_t* idx_d, void* stream);
/* d3d_knn for LARGE point sets (the Pretrain GT instance cloud, PRE-FF:977-983): the point range is cut into n_chunks pieces that run as
 * separate workgroups (d3d_knn gives one workgroup ALL points of its environment: 24 workgroups for 4 608 queries x 2e5 points), partial
 * top-k lists in the caller's workspaces ws_d2_d / ws_idx_d (n_batch * max_queries * n_chunks * k_max elements each), merged in ascending
 * chunk order: the result is bit-identical to d3d_knn (same d^2, ties to the lower index). */
int32_t d3d_knn_chunked(const float* points_d, int64_t point_stride, const int32_t* n_points_d, const float* queries_d, int64_t query_stride,
                        const int32_t* n_queries_d, const int32_t* k_d, int32_t n_batch, int32_t max_queries, int32_t k_max, int32_t n_chunks,
                        float* ws_d2_d, int32_t* ws_idx_d, float* d2_d, int32_t* idx_d, void* stream);
/* a21  the renderer's query (PRE-FF:540-566: `patch_tree.query(sample_points, 4)` followed by "distance >= 1 m -> index -1, distance 1"):
 * same arguments and layout as d3d_knn, but only neighbours INSIDE `radius` are guaranteed -- every slot whose d^2 < radius^2 holds
 * exactly what d3d_knn reports there (same d^2 bits, same index, same order); a slot beyond the radius holds a farther point or
 * (inf, -1).  A workgroup boxes its 256 consecutive queries and scores only the points inside the box grown by the radius. */
int32_t d3d_knn_radius(const float* points_d, int64_t point_stride, const int32_t* n_points_d, const float* queries_d,
                       int64_t query_stride, const int32_t* n_queries_d, const int32_t* k_d, int32_t n_batch,
                       int32_t max_queries, int32_t k_max, float radius, float* d2_d, int32_t* idx_d, void* stream);

/* a7/a10 geometry: per-group centroid (float64 sequential mean, rounded once) + 7-vector
 * [pos-centroid, |pos|, sin dir, cos dir, scale] for every member token (VLN-FF:582-591, 662-673).
 * Groups are described CSR-style per token: tok_slot/tok_row (T) locate the member row, grp_off
 * (G+1) delimits groups.  Outputs: centroid (G,3), cell (G,3) = floor(centroid/cell_len) as int32,
 * geom (T,7).  If inst_pos_d != NULL and grp_inst_d[g] >= 0 the centroid is also stored to
 * inst_pos[(grp_slot[g]*m_cap + grp_inst[g])*3] (merged-instance position update, VLN-FF:663). */
int32_t d3d_group_stats7(const float* rows_pos_d, const float* rows_dir_d, const float* rows_scale_d, int64_t n_cap,
                         const int32_t* tok_slot_d, const int32_t* tok_row_d, const int32_t* grp_off_d, int32_t G,
                         int32_t T, float cell_x, float cell_y, float cell_z, float* centroid_d, int32_t* cell_d,
                         float* geom_d, float* inst_pos_d, const int32_t* grp_slot_d, const int32_t* grp_inst_d,
                         int64_t m_cap, void* stream);

/* a11 geometry: zone centre + 4-vector [p - centre, |p|] (VLN-FF:714-723, 739-749).
 * mode[g] = 0: members' true positions (new zone); 1: members' CELL CENTRES (updated zone, quirk).
 * Writes centre to zone_pos[(slot*z_cap + zone_row[g])*3]; an empty group yields NaN (mean of empty). */
int32_t d3d_group_stats4(const float* inst_pos_d, int64_t m_cap, const int32_t* tok_slot_d, const int32_t* tok_inst_d,
                         const int32_t* grp_off_d, const int32_t* grp_mode_d, const int32_t* grp_slot_d,
                         const int32_t* grp_zone_row_d, int32_t G, int32_t T, float cell_x, float cell_y, float cell_z,
                         float* geom_d, float* zone_pos_d, int64_t z_cap, void* stream);

/* gather float16 pool rows to float32 tokens: out[t,:] = f32(rows_fts[slot[t], row[t], :]). */
int32_t d3d_gather_fts(const uint16_t* rows_fts_d, int64_t n_cap, const int32_t* tok_slot_d, const int32_t* tok_row_d,
                       int32_t T, float* out_d, void* stream);

/* generic float32 row gather / scatter over (slots, cap, D) pools: used for instance/zone features. */
int32_t d3d_gather_rows_f32(const float* pool_d, int64_t cap, int32_t D, const int32_t* slot_d, const int32_t* row_d,
                            int32_t T, float* out_d, void* stream);
int32_t d3d_scatter_rows_f32(float* pool_d, int64_t cap, int32_t D, const int32_t* slot_d, const int32_t* row_d,
                             int32_t T, const float* src_d, const int32_t* src_row_d, void* stream);
/* rows <- constant (tomb-stoning of instances / zones: pos=-10000, fts=0; VLN-FF:378-379, 392-393) */
int32_t d3d_fill_rows_f32(float* pool_d, int64_t cap, int32_t D, const int32_t* slot_d, const int32_t* row_d,
                          int32_t T, float value, void* stream);

/* a9 input: [ft_3d(768), ft_2d(768), new_pos - pos_3d] for every (query, proposal) (VLN-FF:613-617).
 * pair_* (R) index the proposal instance (slot, inst) and the 2D instance (row of new_fts/new_pos). */
int32_t d3d_merge_input(const float* inst_fts_d, const float* inst_pos_d, int64_t m_cap, const float* new_fts_d,
                        const float* new_pos_d, const int32_t* pair_slot_d, const int32_t* pair_inst_d,
                        const int32_t* pair_new_d, int32_t R, float* out_d, void* stream);

/* a12 get_environment_features (VLN-FF:818-862): for env e, the ordered id list ids_d[e*max_ids ..]
 * (n_ids_d[e] entries, dict order) is transformed into the agent frame and filtered by radius;
 * survivors are compacted IN ORDER: rel (n_env,max_ids,3), fts (n_env,max_ids,768), count (n_env). */
int32_t d3d_agent_frame_compact(const float* pool_pos_d, const float* pool_fts_d, int64_t cap, const int32_t* slot_d,
                                const int32_t* ids_d, const int32_t* n_ids_d, int32_t n_env, int32_t max_ids,
                                const d3d_pose* pose_d, float radius, float* rel_d, float* fts_d, int32_t* kept_ids_d,
                                int32_t* count_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense kernels for the ViT / Phi-3 towers (a3, a15, a17).  dtype: 0 = bf16, 1 = fp16 (16-bit storage,
 * fp32 accumulation / statistics).
 * ---------------------------------------------------------------------------------------------- */
/* ROUNDING POINTS.  The reference evaluates these towers module by module in fp16 (OpenAI CLIP after `convert_weights`,
 * clip/model.py:373-395) / bf16 (llava, `torch_dtype=torch.bfloat16`, VLN-POL:125): every nn.Linear, activation, residual add, norm
 * and rotary product STORES a 16-bit tensor.  The fused kernels below reproduce exactly those stores on their fp32 registers
 * (e.g. epilogue 5 = round16(round16(acc + bias) + residual), not round16(acc + bias + residual)), so a fused launch returns what
 * the module sequence returns; inside a GEMM / attention the accumulation is float32 like the reference's library kernels.
 *
 * C = epilogue(A[M,K] W[N,K]^T): nn.Linear layout.  epilogue: 0 none, 1 +bias, 2 +bias QuickGELU (clip/model.py:162),
 * 3 +bias GELU, 4 +residual, 5 +bias +residual, 6 SwiGLU over per-16 interleaved gate/up rows of W (writes N/2 cols).
 * Needs N % 128 == 0, K % 64 == 0, lda/ldw % 8 == 0.  M <= 16 (KV-cache decode rows) with epilogue 0/1/4/6 streams the weights
 * once through a no-LDS kernel (N % 32 == 0, K % 32 == 0 suffice there). */
int32_t d3d_gemm_nt(const void* A_d, const void* W_d, void* C_d, const void* bias_d, const void* residual_d, int32_t M,
                    int32_t N, int32_t K, int64_t lda, int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue,
                    void* stream);
/* same with an explicit tile: 128 = 128x128x64 (4 waves, 2 workgroups/CU; 130 / 132: LDS ring depth forced), 164 = 128x64x64 (3 workgroups/CU,
 * the finer grid for small GEMMs), 256 / 257 = 256x256x64 staggered wave groups
 * stepping K-halves / whole K tiles, 258 = 257 with the partial last round of tiles split along K: the K-slices write fp32
 * partials to a library-owned per-stream workspace and a second launch on the same stream sums them in slice order
 * (deterministic) and runs the epilogue -- no workgroup waits for another one, any number of streams / processes may share
 * the GPU.  260 / 264 = 257 / 258 with the INTERLEAVED K loop (both waves of a SIMD run one software-pipelined stream: the
 * ds_read_b128 of the next K-half one behind every ~3rd MFMA, one barrier per K tile) -- what d3d_gemm_nt uses (D3D_GEMM_LOOP=0:
 * 257 / 258).  All 256-tile variants store through an LDS transposition (16-byte row-contiguous stores): they need ldc % 8 == 0 and
 * 16-byte aligned C / residual.  259 = experiment (LDS-DMA of W issued between the MFMAs); 261-263 = timing experiments with WRONG
 * results (cache-hot operands / no loads in the K loop / no epilogue); 301-303 = 257 / 260 / 262 with in-kernel cycle stamps printed
 * on stderr (bf16, epilogue 0 only).  d3d_gemm_nt picks a tile (and splits the M remainder) itself.  epilogue 7 = LeakyReLU(0.01) for the tcnn CutlassMLP replacement; 8 = its backward factor (see d3d_lrelu_bwd below). */
int32_t d3d_gemm_nt_tile(const void* A_d, const void* W_d, void* C_d, const void* bias_d, const void* residual_d, int32_t M,
                         int32_t N, int32_t K, int64_t lda, int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue,
                         int32_t tile, void* stream);
/* Allocates the split-K workspace of `stream` (64 MiB) now instead of inside the first d3d_gemm_nt that needs it: required before
 * the stream is captured into a hipGraph (no allocation may happen during capture). */
int32_t d3d_gemm_reserve_workspace(void* stream);
/* tinycudann `Network(otype="CutlassMLP")` forward in one call (PRE-FF:221-243, 484, 488): n_hidden + 1
 * bias-free layers y = act(x W^T), fp16 storage / fp32 accumulation, one d3d_gemm_nt launch per layer with the activation in
 * the epilogue.  x (n_rows, n_in) f16; weights: HOST array of n_hidden + 1 device pointers, layer l row-major (out_l, in_l)
 * with the last layer's rows zero-padded to n_out_padded (% 128; e.g. 769 -> 896); act / out_act: 0 none, 1 LeakyReLU(0.01);
 * scratch_a/b (n_rows, n_neurons) f16 each; y (n_rows, n_out_padded) f16. */
/* The fused small-MLP kernel behind d3d_mlp768_forward and the tcnn.Network autograd function (csrc/mlp_kernels.hip): a chain of up to
 * 4 bias-free layers y_l = f_l(y_{l-1} W_l^T) evaluated in ONE launch, a 64-row slab of activations resident in LDS across the layers,
 * weights streamed from L2.  widths[0..n_layers]: input width then every layer's (padded) output width (% 16, <= 896; contraction widths
 * % 32); weights[l]: (widths[l+1], widths[l]) row-major; modes[l]: 0 none, 1 LeakyReLU(0.01), 2 = multiply by LeakyReLU'(aux[l]) -- the
 * data-gradient chain of the backward pass (weights = the transposed matrices, aux[l] = the saved forward activation below);
 * outs[l]: optional HBM copy of layer l's output (saved activations / per-layer gradients), required for the last layer.  All arrays
 * are HOST arrays of n_layers entries read during the call.  dtype 0 bf16 / 1 fp16.  Rounding points = the unfused d3d_gemm_nt path. */
int32_t d3d_mlp_fused(const void* x_d, int64_t ldx, int64_t n_rows, int32_t n_layers, const int32_t* widths, const void* const* weights,
                      const int32_t* modes, const void* const* aux, const int64_t* ld_aux, void* const* outs, const int64_t* ld_outs,
                      int32_t dtype, void* stream);
int32_t d3d_mlp768_forward(const void* x_d, int64_t n_rows, int32_t n_in, const void* const* weights, int32_t n_hidden,
                           int32_t n_neurons, int32_t n_out_padded, int32_t act, int32_t out_act, void* scratch_a_d,
                           void* scratch_b_d, void* y_d, void* stream);
/* ---- float32 dense kernels of the 3D-token builder (a7, a9, a11, a14: VLN-FF:134-161, VLN-POL:83-111) ----------------------------
 * The set encoders, the merge discriminator and the prefix MLPs are float32 modules whose decisions are pinned bit for bit by golden
 * trajectories: they run in float32 on v_mfma_f32_16x16x4_f32 (exact f32 multiply-add chain).
 * C[M,N] = epilogue(A[M,K] W[N,K]^T): epilogue 0 none, 1 +bias, 2 +bias GELU(erf), 3 +bias +residual (residual (M,N), row stride ldc),
 * 4 +bias QuickGELU (clip/model.py:162-164), 5 +residual without bias (Phi-3's bias-free projections) -- 4 / 5 serve the float32 verification towers.
 * K % 16 == 0 (zero-pad operands), N % 4 == 0, strides % 4 == 0. */
int32_t d3d_gemm_nt_f32(const float* A_d, const float* W_d, float* C_d, const float* bias_d, const float* residual_d, int32_t M, int32_t N,
                        int32_t K, int64_t lda, int64_t ldw, int64_t ldc, int32_t epilogue, void* stream);
/* Same contract, on the 16-bit matrix cores at float32 accuracy: each float32 element is split into fp16 hi + lo while its tile is
 * staged and the product taken as three fp16 MFMAs (hi hi + hi lo + lo hi) with float32 accumulation -- ~1e-6 relative to a float32
 * GEMM, 5.3x its matrix peak (csrc/f32x3_kernels.hip).  K % 32 == 0.  The inference token builder's default.
 * a_exp_d (M) / w_exp_d (N), nullable: per-row exponents from d3d_row_exponents -- every operand row is scaled by 2^-exp (exact) before
 * the split and the result scaled back, so no finite row overflows or underflows fp16 (without them: |operand| < 65504 and a row of
 * uniformly tiny values loses bits).  status_d (nullable): bit 0 is OR-ed in when an output element is not finite; the library never
 * reads or clears it. */
int32_t d3d_gemm_nt_f32x3(const float* A_d, const float* W_d, float* C_d, const float* bias_d, const float* residual_d, int32_t M, int32_t N,
                          int32_t K, int64_t lda, int64_t ldw, int64_t ldc, int32_t epilogue, const int32_t* a_exp_d, const int32_t* w_exp_d,
                          int32_t* status_d, void* stream);
/* out_d[m] = floor(log2(max |X[m, :]|)) (0 for a zero row, clamped to +-126); a non-finite element ORs bit 1 into status_d (nullable) and
 * is left out of the maximum.  X (M, K) float32, K % 4 == 0, ld % 4 == 0. */
int32_t d3d_row_exponents(const float* X_d, int32_t M, int32_t K, int64_t ld, int32_t* out_d, int32_t* status_d, void* stream);
/* y = [gelu](x W^T + b) for 1 <= K <= 8 (geometry inputs of the position-embedding MLPs; W (N,K) contiguous) */
int32_t d3d_linear_smallk_f32(const float* x_d, const float* W_d, const float* bias_d, float* y_d, int32_t M, int32_t N, int32_t K, int64_t ldx,
                              int64_t ldy, int32_t gelu, void* stream);
/* y = x W^T + b for 1 <= N <= 8 (the merge discriminator's two logits, VLN-FF:157-161; W (N,K) contiguous, K % 4 == 0) */
int32_t d3d_linear_smalln_f32(const float* x_d, const float* W_d, const float* bias_d, float* y_d, int32_t M, int32_t N, int32_t K, int64_t ldx,
                              int64_t ldy, void* stream);
/* y = [gelu](LayerNorm(x [+ residual])) over rows of D <= 3072 floats: post-LN transformer glue and Linear-LN-GELU in one pass */
int32_t d3d_layer_norm_f32(const float* x_d, const float* residual_d /* optional */, const float* w_d, const float* b_d, float* y_d, int32_t rows,
                           int32_t D, int64_t ldx, int64_t ldr, int64_t ldy, float eps, int32_t gelu, void* stream);

/* ---- FLOAT32 VERIFICATION MODE of the dense towers (a3, a15, a17; csrc/verify_f32_kernels.hip) --------------------------------------
 * `PolicyConfig(clip_dtype=float32, llava_dtype=float32)` runs the towers' host wiring on float32 kernels under strict dispatch, so that
 * north_star's "logits within 1e-3 of the reference" is ASSERTED against the float32 oracle on the HIP GEMM / attention / RoPE / norm
 * wiring (the 16-bit product path sits a ~1.7e-2 noise band from float32 arithmetic, like the reference's own bf16 run).  GEMMs =
 * d3d_gemm_nt_f32, LayerNorms = d3d_layer_norm_f32; the entries below are the float32 twins of the 16-bit row / attention kernels.
 * d3d_attention_f32: the SDPA inside llava.generate (VLN-POL:463) and the ViT blocks (clip/model.py:178-180); the buffer contract of
 * d3d_flash_attention_v3 (fused [q | k | v] head blocks, dense or cu_seqlens-packed, causal, sliding window) with float32 elements. */
int32_t d3d_attention_f32(const float* qkv_d, float* out_d, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                          int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens_d, int32_t window,
                          void* stream);
/* HF Phi3RMSNorm in float32: y = x * rsqrt(mean(x^2) + eps) * w over rows of D floats (D, strides % 4 == 0) */
int32_t d3d_rms_norm_f32(const float* x_d, const float* w_d, float* y_d, int32_t rows, int32_t D, int64_t ldx, int64_t ldy, float eps, void* stream);
/* HF apply_rotary_pos_emb (half split) in float32, in place on the first n_rot_heads heads of every row; tables (positions, head_dim / 2) */
int32_t d3d_rope_inplace_f32(float* qkv_d, const float* cos_d, const float* sin_d, int32_t rows, int32_t S, int32_t n_rot_heads, int32_t head_dim,
                             int64_t ld, const int32_t* pos_of_row_d /* optional */, void* stream);
/* Phi3MLP: out (rows, I) = up * silu(gate) of gate_up (rows, 2 I) = [gate | up] */
int32_t d3d_swiglu_f32(const float* gate_up_d, float* out_d, int64_t rows, int32_t I, void* stream);
/* float32 twins of d3d_patchify / d3d_vit_embed_ln / d3d_assemble_prompt (clip/model.py:222-228; VLN-POL:448-456) */
int32_t d3d_patchify_f32(const float* pixels_d, float* out_d, int32_t B, int32_t S, int32_t patch, int32_t Kp, void* stream);
int32_t d3d_vit_embed_ln_f32(const float* patch_rows_d, const float* cls_d, const float* pos_d, const float* ln_w_d, const float* ln_b_d, float* y_d,
                             int32_t B, int32_t L, int32_t D, float eps, void* stream);
int32_t d3d_assemble_prompt_f32(const uint32_t* desc_d, const float* embed_d, const float* patch_feat_d, const float* patch_pos_d, const float* inst_d,
                                const float* zone_d, float* out_d, int32_t rows, int32_t D, void* stream);
/* ---- backward pass of the tcnn CutlassMLP replacement (the Pretrain path trains these networks, PRE-FF:221-243) ----
 * All three GEMMs of a layer are d3d_gemm_nt launches on 16-bit operands:
 *   forward        h_l  = act(h_{l-1} W_l^T)                         epilogue 0 / 7
 *   data gradient  dz_{l-1} = (dz_l W_l) * act'(h_{l-1})             A = dz_l (M,N_l), W = W_l^T stored (K_l,N_l), epilogue 8 with
 *                                                                    residual_d = h_{l-1} (the output whose sign gates the slope)
 *   weight gradient dW_l = dz_l^T h_{l-1}                            A = dz_l^T (N_l,Mp), W = h_{l-1}^T (K_l,Mp): d3d_transpose_pad16 */
int32_t d3d_transpose_pad16(const void* in_d /* (R,C), row stride ld_in */, void* out_d /* (C,Rp), columns [R,Rp) zero */, int32_t R, int32_t C,
                            int64_t ld_in, int32_t Rp /* % 64, >= R */, void* stream);
/* dz = dy * (y > 0 ? 1 : 0.01) over n 16-bit elements (n % 8 == 0): LeakyReLU(0.01) backward from the layer's OUTPUT y */
int32_t d3d_lrelu_bwd(const void* dy_d, const void* y_d, void* dz_d, int64_t n, int32_t dtype, void* stream);
/* ---- backward of the float32 token-builder modules and of the compositing (the pre-training step's gradients through
 * a7 / a11 / a22 / a23 on the device; replaces PyTorch autograd expressions, PRE-FF:134-161, 446-474) ----
 * d3d_layer_norm_bwd_f32: y = [gelu](LayerNorm(x) * w + b) (nn.LayerNorm [+ nn.GELU], the forward of d3d_layer_norm_f32 without residual).
 *   dx (rows, D); dw_part / db_part (ceil(rows / d3d_layer_norm_bwd_rows_per_block()), D): per-workgroup column sums of dz * xhat / dz,
 *   written in full (no zero-initialisation needed), summed over the first axis by the caller -- a fixed order, hence deterministic. */
int32_t d3d_layer_norm_bwd_f32(const float* x_d, const float* w_d, const float* b_d, const float* dy_d, float* dx_d, float* dw_part_d,
                               float* db_part_d, int32_t rows, int32_t D, int64_t ldx, int64_t lddy, int64_t lddx, float eps, int32_t gelu,
                               void* stream);
int32_t d3d_layer_norm_bwd_rows_per_block(void);
/* exact (erf) GELU on float32: dy_d == NULL: out = gelu(z); else out = dy * gelu'(z) (the backward of the same op). */
int32_t d3d_gelu_f32(const float* z_d, const float* dy_d, float* out_d, int64_t n, void* stream);
/* d3d_set_attention_bwd: gradient of d3d_set_attention (same qkv / set_off / q_rows).  out = the forward's result, dout = dL/dout (rows
 * that were not queried are ignored); dqkv (T, 3*H*64): the k / v thirds are written for every row, the q third for the QUERIED rows
 * only (zero-fill dqkv when q_rows > 0); lse_scratch / d_scratch (T, H) float32 each. */
int32_t d3d_set_attention_bwd(const float* qkv_d, const float* out_d, const float* dout_d, const int32_t* set_off_d, int32_t n_sets, int32_t n_heads,
                              int32_t max_len, int32_t q_rows, float* dqkv_d, float* lse_scratch_d, float* d_scratch_d, void* stream);
/* d3d_composite_bwd: gradient of d3d_composite's feature map w.r.t. the 16-bit sample features / densities it read (same arguments);
 * gout (n_rays, 768) = dL/d(feature map); dfeat (n_rays * n_imp, 768) and ddens (n_rays * n_imp) float32.  (The expected depth is not
 * differentiated: the reference's losses do not use it, PRE-TR:1056-1075.) */
int32_t d3d_composite_bwd(const void* feat16_d, int64_t ldf, const void* dens16_d, int64_t ldd, const float* rel_dist_d, const int32_t* topk_d,
                          const float* gout_d, int32_t n_rays, int32_t N, int32_t n_imp, float* dfeat_d, float* ddens_d, void* stream);

/* One KV-cache decode token through the whole Phi-3 stack (HF Phi3DecoderLayer x n_layers + final norm + lm_head under
 * `llava.generate`, VLN-POL:463), every launch issued from C++.  All pointers are device pointers except the per-layer pointer
 * ARRAYS, which are host arrays of device pointers.  gate_up weights in the per-16 interleaved row order of epilogue 6;
 * rows <= 16; x is updated in place; logits (rows, vocab) in the activations' dtype. */
typedef struct d3d_phi3_decode_args {
    int32_t n_layers, rows, hidden, heads, head_dim, mlp, vocab, dtype;
    float rms_eps;
    void* x;                          /* (rows, hidden) residual stream: embedding of the token to process -> final hidden state */
    void *h, *qkv, *attn, *act;       /* scratch: (rows, hidden), (rows, 3*hidden), (rows, hidden), (rows, mlp) */
    void* logits;                     /* (rows, vocab) */
    const void* const* qkv_w;         /* [n_layers] (3*hidden, hidden) */
    const void* const* o_w;           /* [n_layers] (hidden, hidden) */
    const void* const* gate_up_w;     /* [n_layers] (2*mlp, hidden), interleaved */
    const void* const* down_w;        /* [n_layers] (hidden, mlp) */
    const float* const* n1;           /* [n_layers] input_layernorm weight (float32) */
    const float* const* n2;           /* [n_layers] post_attention_layernorm weight */
    const float* norm_w;              /* final norm */
    const void* lm_head_w;            /* (vocab, hidden) */
    const float *cos_t, *sin_t;       /* RoPE tables (positions, head_dim/2) */
    const int32_t* pos;               /* (rows) position of this token in its sequence */
    const void* const* prompt_qkv;    /* [n_layers] the layer's prefill QKV buffer (packed rows, post-RoPE) */
    const int32_t* cu_seqlens;        /* (rows + 1) */
    void *knew, *vnew;                /* layer 0's (rows, t_max, heads, head_dim) side caches; layer l at + l * cache_layer_stride_bytes */
    int64_t cache_layer_stride_bytes;
    int32_t t_new, t_max, max_prompt_len;
    void* stream;
} d3d_phi3_decode_args;
int32_t d3d_phi3_decode_token(const d3d_phi3_decode_args* args);
/* C = epilogue(RMSNorm(A) W^T) for <= 16 rows (decode): HF Phi3RMSNorm with the float32 gain `norm_w` applied to the raw residual stream
 * inside the weight-streaming GEMM (every workgroup recomputes the row statistics under its first weight loads; bit-identical to
 * d3d_norm followed by d3d_gemm_nt).  epilogue 0 (none) or 6 (SwiGLU over interleaved gate/up rows).  K % 512 == 0, N % 32 == 0. */
int32_t d3d_gemm_nt_rmsnorm(const void* A_d, const float* norm_w_d, float eps, const void* W_d, void* C_d, int32_t M, int32_t N, int32_t K,
                            int64_t lda, int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue, void* stream);
/* llava-phi-3-mini's decoder (hidden 3072, mlp 8192, head_dim 96) at <= 8 rows runs as ONE cooperative launch per token (persistent
 * workgroups, grid barriers between the phases: csrc/decode_kernels.hip; D3D_DECODE_PERSISTENT=0 keeps the launch-per-op path).  A
 * grid barrier that times out raises a sticky per-stream flag instead of hanging the device; d3d_phi3_decode_status synchronises the
 * stream and reports it (D3D_EHIP) -- call it once per generation, after the last token. */
int32_t d3d_phi3_decode_status(void* stream);

/* One decode step of causal self-attention with a KV cache -- the `use_cache` branch of the HF Phi-3 attention under
 * `llava.generate(max_new_tokens=20, do_sample=False)` (VLN-POL:463).  qkv_new (B, 3H, hd): this step's fused projection (before
 * RoPE when cos/sin/pos are given: the kernel rotates q and k at position pos[b]; after RoPE when they are null); prompt_qkv: the layer's prefill buffer (packed rows, post-RoPE) with cu_seqlens (B+1); knew / vnew (B, Tmax, H, hd): the
 * generated tokens' keys / values -- this call appends token t_new and attends over prompt + tokens 0..t_new.  out (B, H, hd). */
int32_t d3d_decode_attention(const void* qkv_new_d, const void* prompt_qkv_d, const int32_t* cu_seqlens_d, void* knew_d, void* vnew_d,
                             void* out_d, int32_t B, int32_t H, int32_t head_dim, int32_t t_new, int32_t Tmax, int32_t max_prompt_len,
                             const float* cos_d, const float* sin_d, const int32_t* pos_d /* all three or none: fused RoPE of q, k */,
                             int32_t dtype, void* stream);
/* LayerNorm (rms = 0; clip/model.py:153-159: float32 statistics, one 16-bit store) or RMSNorm (rms = 1; HF Phi3RMSNorm:
 * `weight * x_normalised.to(input_dtype)` -- the normalised row is stored 16-bit BEFORE the gain is applied) over rows of D <= 4096 */
int32_t d3d_norm(const void* x_d, const float* w_d, const float* b_d, void* y_d, int32_t rows, int32_t D, int64_t ldx,
                 int64_t ldy, float eps, int32_t rms, int32_t dtype, void* stream);
/* in-place half-split rotary embedding on the first n_rot_heads heads of each fused-QKV row; position = row % S.  HF
 * `apply_rotary_pos_emb` on 16-bit tensors: round16(round16(x*cos) + round16(rotate_half(x)*sin)), cos/sin tables holding
 * 16-bit-representable values (the caller rounds them: HF casts cos/sin to the activations' dtype) */
int32_t d3d_rope_inplace(void* qkv_d, const float* cos_d, const float* sin_d, int32_t rows, int32_t S, int32_t n_rot_heads,
                         int32_t head_dim, int64_t ld, const int32_t* pos_of_row_d /* optional: explicit position per row */,
                         int32_t dtype, void* stream);
/* out[m,i] = up * silu(gate) for a plain [gate(I) | up(I)] projection output */
int32_t d3d_swiglu(const void* gate_up_d, void* out_d, int64_t rows, int32_t I, int32_t dtype, void* stream);
/* Round 6: the software-pipelined attention kernel (csrc/attn4_kernels.hip) -- the same contract as d3d_flash_attention_v3_rope_q without a
 * window (dense or cu_seqlens-packed, causal or not, head_dim 64 / 96, optional fused query RoPE): inside a wave the MFMAs of key blocks
 * b-1 / b+1 are interleaved with the online softmax of block b (32-key blocks).  An experiment: D3D_ATTN_V4=1 routes the window-free,
 * table-free launches of d3d_flash_attention_v3* here (2-3 % faster on the Phi-3 shape, level on the ViT shape; off by default).  Replaces the SDPA inside llava.generate (VLN-POL:463)
 * and the ViT blocks' attention (clip/model.py:178-180). */
int32_t d3d_flash_attention_v4(const void* qkv_d, void* out_d, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride,
                               int64_t batch_stride, int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len,
                               const int32_t* cu_seqlens_d, const float* rope_cos_d, const float* rope_sin_d, int32_t dtype, void* stream);
/* fused (flash) self-attention forward over a fused QKV projection buffer (clip/model.py:171-183 MultiheadAttention,
 * HF CLIP / Phi-3 attention; csrc/attn3_kernels.hip): qkv (B,S,Htot,hd) 16-bit, q/k/v heads start at q_off/k_off/v_off; strides in
 * elements; out (B,S,H,hd); hd in {64,96}; keys >= seq_len are masked; causal = lower-triangular.  cu_seqlens_d (optional, B+1 ints):
 * PACKED variable-length batch -- sequence b occupies rows [cu[b], cu[b+1]) of qkv / out, S is then the longest sequence and
 * batch_stride / seq_len are ignored.  `window` > 0 (causal only): a query attends to its last `window` keys, itself included -- HF's
 * sliding-window mask (Phi-3-mini-4k: 2047; transformers 4.46 modeling_phi3.py `_prepare_4d_causal_attention_mask_with_cache_position`:
 * key <= query - window is masked); 0 = no window.  32x32x16 MFMA tiles, one query column per lane pair, K/V tiles staged by LDS-DMA and
 * all fragment reads of a tile issued a phase ahead, XCD-aware grid.  (Rounds 1-3 exported two earlier kernels, d3d_flash_attention and
 * d3d_flash_attention_v2, as A/B baselines; retired in round 4, measurements in DESIGN.md section 4b.) */
int32_t d3d_flash_attention_v3(const void* qkv_d, void* out_d, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride,
                               int64_t batch_stride, int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len,
                               const int32_t* cu_seqlens_d, int32_t window, int32_t dtype, void* stream);
/* The same with the QUERIES' rotary embedding fused: the buffer holds un-rotated q and rotated k (d3d_rope_inplace over the k heads only);
 * rope_cos_d / rope_sin_d are d3d_rope_inplace's (positions, head_dim / 2) float32 tables, a query's position = its row inside its sequence.
 * Same bits as rotating q in place first.  Both null = d3d_flash_attention_v3. */
int32_t d3d_flash_attention_v3_rope_q(const void* qkv_d, void* out_d, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride,
                                      int64_t batch_stride, int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len,
                                      const int32_t* cu_seqlens_d, int32_t window, const float* rope_cos_d, const float* rope_sin_d, int32_t dtype,
                                      void* stream);
/* The same with a caller-built SCHEDULE (packed causal prefill, ragged prompts): wg_table_d[i] = (sequence << 20) | (head << 8) | query block
 * (128 queries per block) -- one query block per workgroup, launched in table order; n_wg entries.  The caller lists the heaviest blocks
 * (highest query block of each sequence) first and puts entry i of a (sequence, head) on i % 8 == (sequence * H + head) % 8 so that one
 * XCD's L2 serves that head's keys (dynam3d_amd/hip_dense.py attention_schedule).  Results are bit-identical to the unscheduled launch
 * (which pairs query blocks i and n-1-i per workgroup).  wg_table_d null = d3d_flash_attention_v3_rope_q. */
int32_t d3d_flash_attention_v3_sched(const void* qkv_d, void* out_d, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride,
                                     int64_t batch_stride, int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len,
                                     const int32_t* cu_seqlens_d, int32_t window, const float* rope_cos_d, const float* rope_sin_d,
                                     const int32_t* wg_table_d, int32_t n_wg, int32_t dtype, void* stream);
/* self-attention inside packed variable-length token sets (set encoders VLN-FF:134-155): float32, head_dim 64.
 * qkv (T, 3*H*64) = [q|k|v]; set g = tokens [set_off[g], set_off[g+1]); q_rows > 0 restricts the queries to the
 * first q_rows rows of every set (1 = CLS only).  out (T, H*64); rows that are not queried are left untouched. */
int32_t d3d_set_attention(const float* qkv_d, const int32_t* set_off_d, int32_t n_sets, int32_t n_heads, int32_t max_len,
                          int32_t q_rows, float* out_d, void* stream);
/* a3 front-end (resnet_encoders.py:267-271): uint8 HWC -> bicubic SxS (rounded back to uint8) -> /255 -> normalise, f32 CHW */
int32_t d3d_resize_normalize(const uint8_t* rgb_d, float* out_d, int32_t B, int32_t H, int32_t W, int32_t S,
                             const float* mean3_h, const float* std3_h, void* stream);

/* a3 / a15 patch embedding, A operand (clip/model.py:206, 222 `conv1`, stride = kernel = patch, no bias == a GEMM over unfolded
 * patches): normalised float32 CHW pixels (B,3,S,S) -> (B*(S/patch)^2, Kp) rows in the tower's dtype, k = c*patch^2 + i*patch + j,
 * columns [3*patch^2, Kp) zero (Kp % 64 == 0 for d3d_gemm_nt: 588 -> 640). */
int32_t d3d_patchify(const float* pixels_d, void* out_d, int32_t B, int32_t S, int32_t patch, int32_t Kp, int32_t dtype, void* stream);
/* a3 / a15 embeddings (clip/model.py:224-228): y[b,0] = LN(cls + pos[0]), y[b,1+p] = LN(patch_rows[b*(L-1)+p] + pos[1+p]); the sum is
 * rounded to the 16-bit dtype before ln_pre, as the reference's 16-bit add is.  cls (D), pos (L,D), patch_rows (B*(L-1),D), y (B*L,D). */
int32_t d3d_vit_embed_ln(const void* patch_rows_d, const void* cls_d, const void* pos_d, const float* ln_w_d, const float* ln_b_d, void* y_d,
                         int32_t B, int32_t L, int32_t D, float eps, int32_t dtype, void* stream);
/* The packed Phi-3 prompt of all environments in one pass (VLN-POL:448-456).  Output row t (of `rows`, a multiple of 256; row stride D)
 * is selected by desc_d[t] = (source << 28) | source_row: 0 embedding-table row, 1 patch token = patch_feat[row] + patch_pos[row] added in
 * float32 and rounded once (VLN-POL:448-453), 2 instance token, 3 zone token, 7 zero row.  All tensors (.., D) contiguous in `dtype`
 * (0 bf16 / 1 fp16), D % 8 == 0. */
int32_t d3d_assemble_prompt(const uint32_t* desc_d, const void* embed_d, const void* patch_feat_d, const void* patch_pos_d, const void* inst_d,
                            const void* zone_d, void* out_d, int32_t rows, int32_t D, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pretrain novel-view renderer.  The 768-wide tcnn MLPs (PRE-FF:221-243, 484, 488) are
 * d3d_gemm_nt launches with epilogue 7 (LeakyReLU 0.01, no bias), fp16.
 * ---------------------------------------------------------------------------------------------- */
/* a20 get_rays_habitat + world transform (PRE-FF:408-422, 524-530): float64 arithmetic rounded once.
 * rel_y (N) f64 = linspace(near,far,N); tan_xy/tan_z (R) f32; pose64 (n_env,5) = cx,cy,cz,cos h,sin h -> ray (n_env,R,N,3) */
int32_t d3d_rays_habitat(const double* rel_y_d, const float* tan_xy_d, const float* tan_z_d, const double* pose64_d,
                         int32_t n_env, int32_t R, int32_t N, float* ray_xyz_d, void* stream);
/* a20 get_rays + world transform, intrinsics mode (PRE-FF:390-405, 534): z (N) f64 = float32(near + spacing*(i+1)) widened;
 * cam16 (n_env,16) f64 = fx, fy, cx, cy, R[9] row-major, T[3] -> ray (n_env, view_h*view_w, N, 3) f32 */
int32_t d3d_rays_pinhole(const double* z_d, const double* cam16_d, int32_t n_env, int32_t view_h, int32_t view_w, int32_t N,
                         float* ray_xyz_d, void* stream);
/* a21 importance sampling (PRE-FF:543-556): from the k-NN table (n_rays*N, k): dist = sqrt(d2), >= radius -> none;
 * importance 1/sum(dist); top n_imp per ray by (importance desc, sample index asc) -> topk (n_rays,n_imp), neighbour
 * ids of the chosen samples sidx (n_rays,n_imp,k) (-1 = none), n_ranked (n_rays) samples with a neighbour in range */
int32_t d3d_ray_topk(const float* d2_d, const int32_t* idx_d, int32_t n_rays, int32_t N, int32_t k, float radius,
                     int32_t n_imp, int32_t* topk_d, int32_t* sidx_d, int32_t* n_ranked_d, void* stream);
/* a21/a22 front half (PRE-FF:586-616, 479-483): neighbour gather, 6-d relative geometry, Linear(6,768)+LayerNorm,
 * fp16 add with the neighbour's fp16 feature -> s16 (n_rays*n_imp, k*768) fp16.  pose3 (n_env,3) = cos(-h),sin(-h),h */
int32_t d3d_render_embed(const float* rows_pos_d, const float* rows_dir_d, const float* rows_scale_d, const uint16_t* rows_fts_d,
                         int64_t n_cap, const int32_t* ray_slot_d, const int32_t* ray_env_d, const float* ray_xyz_d,
                         const int32_t* topk_d, const int32_t* sidx_d, const float* pose3_d, const float* rel_direction_d,
                         int32_t n_rays, int32_t R, int32_t N, int32_t n_imp, int32_t k, float far_, const float* w6_d,
                         const float* b6_d, const float* ln_w_d, const float* ln_b_d, float eps, uint16_t* s16_d,
                         float* geom6_d, float* sample_xyz_d, void* stream);
/* a23 raw2feature (PRE-FF:446-474): softplus density, alpha compositing over the chosen samples, L2-normalised
 * feature (n_rays,768) f32 and expected depth (n_rays).  feat/dens fp16 with row strides ldf/ldd (elements). */
int32_t d3d_composite(const void* feat16_d, int64_t ldf, const void* dens16_d, int64_t ldd, const float* rel_dist_d,
                      const int32_t* topk_d, int32_t n_rays, int32_t N, int32_t n_imp, float* feature_map_d, float* depth_d,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Host-side bookkeeping (ids / dict semantics of VLN-FF:357-393, 433-475, 623-691, 694-756).
 * Integer-only control plane; all float work stays in the kernels above.  Not thread-safe per
 * handle.  compat: 0 = 'reference' (quirks F11/Z1 reproduced), 1 = 'fixed' (id == row).
 * ---------------------------------------------------------------------------------------------- */
typedef struct d3d_ff d3d_ff;

d3d_ff* d3d_ff_create(int32_t compat_fixed, int32_t patches_per_view, int32_t num_proposals);
void d3d_ff_destroy(d3d_ff* ff);
int32_t d3d_ff_reset(d3d_ff* ff, int32_t batch_size);          /* VLN-FF:186-206 */
/* cell index of a tomb-stoned instance position: floor(-10000 / cell_len) per axis */
int32_t d3d_ff_set_tomb_cell(d3d_ff* ff, int32_t cx, int32_t cy, int32_t cz);
int32_t d3d_ff_pop(d3d_ff* ff, int32_t env);                    /* VLN-FF:210-229 */
int32_t d3d_ff_batch_size(const d3d_ff* ff);
/* counters: which = 0 rows, 1 instance slots, 2 live instances, 3 zone rows, 4 live zones, 5 live patch ids */
int64_t d3d_ff_count(const d3d_ff* ff, int32_t env, int32_t which);

/* deletion cascade for one env (VLN-FF:362-393).  hits: row indices tomb-stoned by d3d_frustum_cull.
 * inst cells must be current (see d3d_ff_set_inst_cells).  Outputs the instance slots and zone rows
 * that must be tomb-stoned on the device. */
int32_t d3d_ff_apply_hits(d3d_ff* ff, int32_t env, const int32_t* hits_h, int32_t n_hits, int32_t* dead_inst_h,
                          int32_t* n_dead_inst, int32_t* dead_zone_h, int32_t* n_dead_zone, int32_t cap);

/* start of a view update: appends P rows, returns row base and k0 = min(#live instances, K) */
int32_t d3d_ff_begin_view(d3d_ff* ff, int32_t env, int32_t* row_base, int32_t* k0, int32_t* has_tree);

/* merge planning (VLN-FF:604-691).  segm (P) dense labels, n_seg groups; d2/idx/logits laid out
 * (n_seg, k_max[,2]) with k0 valid columns; new_cells (n_seg,3).  Outputs:
 *   seg_slot (n_seg): instance slot that receives segment s' centroid/feature if NEW, else -1
 *   dirty_inst (<= n_seg): merged instances needing centroid + re-encode, with CSR member rows
 *   k_eff: proposals actually used after the tomb-stone shrink (VLN-FF:607-610) */
int32_t d3d_ff_plan_merge(d3d_ff* ff, int32_t env, const int32_t* segm_h, int32_t n_seg, int32_t k0, int32_t k_max,
                          const float* d2_h, const int32_t* idx_h, const float* logits_h, const int32_t* new_cells_h,
                          int32_t* k_eff, int32_t* seg_slot_h, int32_t* dirty_inst_h, int32_t* n_dirty,
                          int32_t* dirty_off_h, int32_t* dirty_rows_h, int32_t rows_cap);

/* zone planning (VLN-FF:694-756).  dirty_cells (n_dirty,3) are the cells of the merged centroids.
 * Outputs per touched zone: row to write, mode (0 new / 1 update), member instance CSR. */
int32_t d3d_ff_plan_zones(d3d_ff* ff, int32_t env, const int32_t* dirty_cells_h, int32_t* n_touched,
                          int32_t* zone_row_h, int32_t* zone_mode_h, int32_t* zone_off_h, int32_t* zone_members_h,
                          int32_t zcap, int32_t mcap);

/* end of view: snapshot for the next KNN (tree rebuild, VLN-FF:815); returns #slots in the tree */
int32_t d3d_ff_end_view(d3d_ff* ff, int32_t env, int32_t* tree_slots);
int32_t d3d_ff_rebuild_tree(d3d_ff* ff, int32_t env, int32_t* tree_slots);   /* VLN-FF:396 */

/* dict-ordered live ids (VLN-FF:825, 844) */
int32_t d3d_ff_live_ids(const d3d_ff* ff, int32_t env, int32_t* inst_ids_h, int32_t* n_inst, int32_t* zone_ids_h,
                        int32_t* n_zone, int32_t cap);

/* debug / test export of the dictionaries (sizes via d3d_ff_count; members as CSR in dict order) */
int32_t d3d_ff_export_owner(const d3d_ff* ff, int32_t env, int32_t* owner_h, int64_t n);
int32_t d3d_ff_export_members(const d3d_ff* ff, int32_t env, int32_t which /*0 inst,1 zone*/, int32_t* ids_h,
                              int32_t* off_h, int32_t* flat_h, int64_t flat_cap);
int32_t d3d_ff_export_zone_keys(const d3d_ff* ff, int32_t env, int32_t* cells_h /*(n,3)*/, int32_t* ids_h, int32_t cap);

/* ------------------------------------------------------------------------------------------------
 * Device-resident planner of the memory update (csrc/ff_plan.h + ff_plan_kernels.hip): the same dict / id semantics as d3d_ff_*
 * above (VLN-FF:362-393 deletion cascade, 433-475 lowest unused ids, 623-691 new / merge bookkeeping, 694-756 zones, 825/844 dict
 * order), decided ON THE GPU from the kernels' own outputs -- the frustum hit list, the KNN table, the merge logits, the cells --, so
 * that an update reads back ONE small report per view instead of synchronising before every decision.  All pointers are device
 * pointers; the state is a set of int32 arrays owned by the caller, one row per storage slot:
 *   hdr [S][16] | rows [S][3][R]: owner, stamp_of_pid, pid_of_stamp | inst [S][6][M]: live, stamp, members, cell x/y/z |
 *   zone [S][8][Z]: live, stamp, key stamp, snapshot size, visit mark, key x/y/z | edges [S][2][2][E]: zone-member snapshots
 *   (double-buffered) | scratch [S][W], W >= max(8 * P + 16, M + Z).   A fresh slot is all zero except owner / pid_of_stamp = -1.
 * `slot[e]` maps environment e of the batch to its storage slot; one workgroup per environment, asynchronous on `stream`.
 * ---------------------------------------------------------------------------------------------- */
typedef struct d3d_ffdev_state {
    int32_t *hdr, *rows, *inst, *zone, *edges, *scratch;
    int32_t R, M, Z, E, W;
    int32_t compat_fixed, P, K;      /* compat as d3d_ff_create; P patches per view; K = num_proposal_instances */
    int32_t tomb[3];                 /* cell of a tomb-stoned position: floor(-10000 / cell_len) */
} d3d_ffdev_state;
#define D3D_FFDEV_HDR_WORDS 16
#define D3D_FFDEV_REPORT_WORDS 16

/* k0[e] = min(#live instances, K) if the env has a tree else 0 (VLN-FF:532); tree_slots[e] = points in its tree */
int32_t d3d_ffdev_begin_view(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, int32_t* k0, int32_t* tree_slots, void* stream);
/* the cascade behind d3d_frustum_cull's hit list (hits [B][hits_stride], n_hits [B]): VLN-FF:362-393; ends with the tree rebuild mark */
int32_t d3d_ffdev_apply_hits(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* hits, int64_t hits_stride,
                             const int32_t* n_hits, float* inst_pos, float* inst_fts, int64_t m_cap, float* zone_pos, float* zone_fts,
                             int64_t z_cap, int32_t fts_dim, void* stream);
/* VLN-FF:604-691 for one view.  order / tok_seg [B][P]: the patches sorted by (segment, patch) and their segments; seg_off
 * [B][n_max+1]; d2 / idx [B][n_max][k_max]; logits [B][n_max][k_max][2]; new_cells [B*n_max][3].  Out: seg_slot [B][n_max] (slot of
 * the NEW instance of a segment, -1 = merged), dirty_inst [B][n_max], dirty_off [B][n_max+1], dirty_rows [B][rows_stride] (member
 * rows of the merged instances, push order), report [B][16]. */
int32_t d3d_ffdev_plan_merge(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* order, const int32_t* tok_seg,
                             const int32_t* seg_off, const int32_t* n_seg, int32_t n_max, int32_t k_max, const int32_t* k0,
                             const float* d2, const int32_t* idx, const float* logits, const int32_t* new_cells, int32_t* seg_slot,
                             int32_t* dirty_inst, int32_t* dirty_off, int32_t* dirty_rows, int64_t rows_stride, int32_t* report,
                             void* stream);
/* per-environment merge plans -> the flat CSR tables of d3d_group_stats7 / d3d_gather_fts; groups past the real ones are empty and
 * point nowhere (grp_inst -1); totals [2 + 2B] = groups, tokens, token base per env, group base per env */
int32_t d3d_ffdev_flatten_merge(int32_t B, int32_t n_max, const int32_t* slot, const int32_t* dirty_inst, const int32_t* dirty_off,
                                const int32_t* dirty_rows, int64_t rows_stride, const int32_t* report, int32_t* tok_slot,
                                int32_t* tok_row, int64_t tok_cap, int32_t* grp_off, int32_t* grp_slot, int32_t* grp_inst,
                                int32_t* totals, void* stream);
/* VLN-FF:694-756 for one view.  merged_cells [groups][3]: d3d_group_stats7's cells over the flattened merge groups. */
int32_t d3d_ffdev_plan_zones(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* dirty_inst,
                             const int32_t* merged_cells, const int32_t* new_cells, const int32_t* n_seg, int32_t n_max,
                             int32_t* zone_row, int32_t* zone_mode, int32_t* zone_off, int32_t* zone_mem, int64_t mem_stride,
                             int32_t* report, void* stream);
int32_t d3d_ffdev_flatten_zones(int32_t B, int32_t n_max, const int32_t* slot, const int32_t* zone_row, const int32_t* zone_mode,
                                const int32_t* zone_off, const int32_t* zone_mem, int64_t mem_stride, const int32_t* report,
                                int32_t* tok_slot, int32_t* tok_inst, int64_t tok_cap, int32_t* grp_off, int32_t* grp_mode,
                                int32_t* grp_slot, int32_t* grp_row, int32_t* totals, void* stream);
/* dict-ordered live ids (VLN-FF:825, 844): inst_ids / zone_ids [B][max_ids], counts [B] */
int32_t d3d_ffdev_live_ids(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, int32_t* inst_ids, int32_t* n_inst,
                           int32_t* zone_ids, int32_t* n_zone, int32_t max_ids, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNAM3D_HIP_H */
