/*
 * scn_mi355x.h -- C ABI of libscn_mi355x.so: the MI355X-native (gfx950) replacement for the
 * native boundary of `sparseconvnet` (SCN), the un-vendored third-party package in which the
 * reference's sparse-convolution hot path runs.
 *
 * What it replaces.  The reference (LeonhardFeiner/sparse_rcnn) reaches native code only through
 * `import sparseconvnet as scn` (ndsis/modules/module_factory.py:5, model.py:6,
 * custom_operations.py:4, roi_select_sparse.py:3).  Upstream, every scn Python layer forwards to
 * a pybind11 module `sparseconvnet.SCN` (`Metadata_3`, `<Op>_updateOutput`, `<Op>_backward`); that
 * module is NOT under /root/reference and cannot be cited by line, so each entry point below cites
 * the REFERENCE call site whose work it performs.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no C++/torch types.  Every pointer is a DEVICE pointer
 *     unless its name ends in `_host`.  `stream` is a hipStream_t passed as void*.
 *   - Ownership: the caller owns every buffer (feature slabs, hash tables, rule tables, scratch).
 *     The library allocates nothing and keeps no state besides the last-error string, so the
 *     caller's stream-ordered allocator (PyTorch's caching allocator in the Python host layer)
 *     governs lifetime.  Data-dependent sizes use two phases: a *_scan/_build call that returns
 *     the size through a `_host` out-parameter (it synchronises `stream` once; documented per
 *     function), then the caller allocates and calls the matching *_fill.
 *   - Every function returns 0 on success or an SCN_E* code; scn_last_error_string() describes
 *     the last failure on the calling thread.  Nothing aborts.
 *   - Threading: functions are re-entrant; work is enqueued on `stream` and is asynchronous
 *     except where a `_host` out-parameter is documented to synchronise.
 *   - Feature slabs are row-major fp32 [rows][channels], channels contiguous.
 *   - Coordinates on device are int32 [n][4] = (x, y, z, batch); hash keys pack 16 bits per field.
 *   - Rule tables are int32 [n_off][n_out]: table[o][r] = input row feeding output row r through
 *     kernel offset o, or -1.  Compacted rules are two int32 arrays (in_rows, out_rows), offset-major,
 *     output row ascending inside an offset (the canonical order), with an int64
 *     prefix[n_off+1].
 */
#ifndef SCN_MI355X_H
#define SCN_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCN_OK 0
#define SCN_EINVAL 1      /* bad argument (null pointer, negative size, unsupported shape) */
#define SCN_ESIZE 2       /* size-constraint violation, e.g. (out-1)*stride+filter != in */
#define SCN_EHASH 3       /* hash table too small / coordinate outside the 16-bit key range */
#define SCN_EHIP 4        /* a HIP runtime call failed; see scn_last_error_string() */

/* History of the ABI (scn_abi_version() returns the value the library was BUILT with: a caller compares it with the
 * SCN_ABI_VERSION of the header it was compiled against):
 *   1  rounds 1-2: 78 entry points
 *   2  round 3: + 23 entry points (step executor: scn_exec_op / scn_exec_level structs, scn_exec_run*; deferred
 *      weight-gradient sums; bf16 elementwise forms; segment pooling; scn_tiles_build_x) -- shipped with the value still 1
 *   3  round 4: SCN_PYRAMID_FUSED (scn_pyramid_build_ex flag, larger scn_pyramid_workspace_bytes), hash slot function
 *      changed (tables built by version <= 2 libraries are not probe-compatible; no table outlives a Metadata, so only a
 *      caller that kept raw tables across a library upgrade is affected); + scn_exec_timing_enable / _collect, scn_pad_params_many, scn_conv_tiles_split_count, scn_dedup_launch_div, scn_child_table_div (107 entry points);
 *      scn_pool_fwd / _bwd (+ _bf16): `average` carries the pool volume above bit 8 (0 = the 2^3 of every configuration) */
/*   4  round 5: + scn_debug_set / scn_debug_get (developer switches no longer follow the ambient environment per launch);
 *      scn_exec_timing_collect forgets only the records it returned; + scn_nms_bits / scn_nms_scratch_bytes, scn_dilate_gather_fwd / _bwd,
 *      scn_parent_lookup_div, scn_topk_boxes / scn_topk_scratch_bytes, scn_cell_map (117 entry points) */
/*   5  round 6: + scn_conv_tiles_chain / scn_conv_tiles_chain_counts, scn_wgrad_tiles32 / _scratch_bytes (121 entry points); scn_tiles_build_x: bits 8-10 of `with_x`
 *      = log2 of the row bins in the sort key of a 27-offset table (0: as before); scn_pyramid_build_ex with
 *      SCN_PYRAMID_XCD_ORDER sorts level 0 by (row bin, mask); switches SCN_TS_PROG, SCN_TS_NO_CHAIN, SCN_TB_NO_BINS */
#define SCN_ABI_VERSION 5

/* flags for the gather-GEMM entry points */
#define SCN_F_RELU_IN 1      /* use max(X,0) as the input slab (fuses scn.ReLU before a conv, module_factory.py:88,173-176) */
#define SCN_F_W_TRANSPOSED 2 /* use W[o]^T: W is [n_off][cout][cin] seen from this call (backward-data) */
#define SCN_F_OFF_REVERSE 4  /* weight index n_off-1-o for table row o (SubM backward-data: R_o^T = R_{k^3-1-o}) */
#define SCN_F_RESIDUAL_LAST 8 /* scn_conv_tiles: add `residual` AFTER the ReLU-backward mask (gradient of a residual block:
                                dX = mask(conv^T dY1) + dY), default is before */
#define SCN_F_SPLIT_SUM 16    /* scn_conv_tiles: launch the tile kernel only; the caller runs scn_conv_tiles_finish next
                               * (lets a profiler bracket the two kernels separately; results are identical) */
#define SCN_F_TILE_ORDER_X 64 /* scn_conv_tiles_bf16: `tile_order` was built by scn_tiles_build_x(with_x): hand the tiles out by
                               * spatial bin, bin x to the workgroups of XCD x (same results, L2-local row gathers) */
#define SCN_F_GEMM_V1 32      /* scn_gemm_table / scn_gemm_rules: run the register-only kernels (operands straight from
                               * global memory) where the LDS-tiled ones would be taken; same bits -- the tests' cross-check */

typedef void* scn_stream_t;

int scn_abi_version(void);
const char* scn_last_error_string(void);

/* Developer switches (round 5; no reference counterpart -- A/B and cross-check knobs of this library's own kernel variants,
 * e.g. SCN_TS_NO_TAIL, SCN_TB_STREAM, SCN_PYRAMID_V1: the list is scn_debug.hip's table).  The environment is read ONCE, at
 * the first use of any switch; afterwards a switch changes only through scn_debug_set (value NULL: unset).  The product path
 * never scans the environment per launch (rounds 1-4: ~300 getenv calls per step).  Unknown name: SCN_EINVAL. */
int scn_debug_set(const char* name, const char* value);
int scn_debug_get(const char* name, int* is_set, int64_t* value);

/* ------------------------------------------------------------------------------------------
 * Index path: scn.Metadata / InputLayer rules / rulebooks
 * ---------------------------------------------------------------------------------------- */

/* Power-of-two slot count the hash tables for n keys must have. */
int64_t scn_hash_capacity(int64_t n);

/* int64 [n][4] coords (x,y,z,batch) as the reference hands them over (ndsis/data/data.py:95-98, consumed at
 * custom_operations.py:72-80) -> int32 [n][4].  *bad_host receives the number of rows with a field outside
 * [0,65535] (synchronises stream); such input is rejected with SCN_EHASH.  bad_host == NULL: asynchronous form --
 * nothing is copied back and the stream is not synchronised; *scratch1 (device) holds the count once the stream gets
 * there and the caller checks it behind its own event. */
int scn_coords_to_i32(const int64_t* coords, int64_t n, int32_t* out, int32_t* scratch1, int64_t* bad_host,
                      scn_stream_t stream);

/* Scratch bytes scn_dedup_build needs for n items. */
int64_t scn_dedup_scratch_bytes(int64_t n);

/* Row numbering by first occurrence (InputLayer rules: custom_operations.py:72-80 with mode 0-4;
 * strided-conv output sites: module_factory.py:232-234).  Key of item i = (batch, x>>shift, y>>shift, z>>shift).
 *   table_keys[cap], table_rows[cap]  hash of the distinct keys -> row id (kept by the caller as the grid of
 *                                     this spatial size; used later by scn_subm_table)
 *   item_row[n]      row id of every item        (InputLayer: point -> voxel row;  Convolution: fine row -> coarse row)
 *   row_count[n]     multiplicity of each row, first *n_rows_host entries valid (may be NULL)
 *   row_first[n]     smallest item index of each row (may be NULL)
 *   row_coords[n][4] coordinates (shifted) of each row
 *   n_rows_host      number of distinct rows (synchronises stream once) */
int scn_dedup_build(const int32_t* coords, int64_t n, int shift, uint64_t* table_keys, int32_t* table_rows,
                    int64_t cap, int32_t* item_row, int32_t* row_count, int32_t* row_first, int32_t* row_coords,
                    void* scratch, int64_t* n_rows_host, scn_stream_t stream);

/* scn_dedup_build without the host synchronisation: the row count goes to n_rows_dev (device int64[1]) in stream
 * order; the caller copies it back behind its own event and may queue work that does not need it meanwhile. */
int scn_dedup_launch(const int32_t* coords, int64_t n, int shift, uint64_t* table_keys, int32_t* table_rows,
                    int64_t cap, int32_t* item_row, int32_t* row_count, int32_t* row_first, int32_t* row_coords,
                    void* scratch, int64_t* n_rows_dev, scn_stream_t stream);

/* Submanifold neighbour table for filter k (odd; reference uses 1 and 3: module_factory.py:383-385,404-406):
 * table[o][r] = row of coords[r] + delta_o or -1, o = ((dx+h)k + (dy+h))k + (dz+h). */
int scn_subm_table(const int32_t* coords, int64_t n, const uint64_t* table_keys, const int32_t* table_rows,
                   int64_t cap, int k, int32_t* table, scn_stream_t stream);

/* Children table of a size=stride=2 Convolution (module_factory.py:232-234): child[o][c] = fine row with parent c
 * and offset o = ((x&1)*2 + (y&1))*2 + (z&1), or -1.  Also writes fine_off[n_fine] = that offset. */
int scn_child_table(const int32_t* fine_coords, const int32_t* parent, int64_t n_fine, int64_t n_coarse,
                    int32_t* child, int32_t* fine_off, scn_stream_t stream);
/* The same two steps for ANY filter_size = filter_stride = (sx, sy, sz) -- `get_downsampler(stride=...)` /
 * `get_upsampler` hand an int or one entry per axis to scn.Convolution / scn.Deconvolution (module_factory.py:221-258; every
 * shipped configuration uses 2): coarse site = (x / sx, y / sy, z / sz), numbered by first occurrence like scn_dedup_launch;
 * child table [sx sy sz][n_coarse] with offset o = ((x % sx) sy + y % sy) sz + z % sz, which is the 2^3 numbering for 2. */
int scn_dedup_launch_div(const int32_t* coords, int64_t n, int sx, int sy, int sz, uint64_t* table_keys, int32_t* table_rows,
                         int64_t cap, int32_t* item_row, int32_t* row_count, int32_t* row_first, int32_t* row_coords,
                         void* scratch, int64_t* n_rows_dev, scn_stream_t stream);
int scn_child_table_div(const int32_t* fine_coords, const int32_t* parent, int64_t n_fine, int64_t n_coarse, int sx, int sy,
                        int sz, int32_t* child, int32_t* fine_off, scn_stream_t stream);
/* The first of the two steps when the coarse grid ALREADY EXISTS on the Metadata (SparseConvNet keys its grids by spatial size:
 * 64 -> 16 by one stride-4 layer after 64 -> 32 -> 16 by two stride-2 layers lands in the same grid): parent[i] = the existing
 * grid's row of (x / sx, y / sy, z / sz), -1 where that site is not in the grid; *n_missing_dev = how many were not (the
 * caller refuses those: upstream would grow the grid under the tensors that live on it).  scn_child_table_div then builds
 * the child table against that numbering (n_coarse may exceed n_fine there). */
int scn_parent_lookup_div(const int32_t* fine_coords, int64_t n_fine, int sx, int sy, int sz, const uint64_t* table_keys,
                          const int32_t* table_rows, int64_t cap, int32_t* parent, int64_t* n_missing_dev, scn_stream_t stream);

/* Compaction of a rule table into (in,out) pairs -- wave ballot + prefix sum.
 * Phase 1: counts; block_sums must hold scn_rules_blocks(n_off, n_out) int32; writes prefix (device, int64[n_off+1])
 * and copies it to prefix_host (synchronises stream once).  prefix_host == NULL: asynchronous form -- no copy, no
 * synchronisation; the caller reads `prefix` back when it needs the sizes (the forward pass never does: only
 * scn_wgrad_rules / scn_gemm_rules take compacted rules). */
int64_t scn_rules_blocks(int n_off, int64_t n_out);
int scn_rules_scan(const int32_t* table, int n_off, int64_t n_out, int32_t* block_sums, int64_t* prefix,
                   int64_t* prefix_host, scn_stream_t stream);
/* Phase 2: in_rows/out_rows (and seg_of = offset index of each pair, may be NULL) of prefix_host[n_off] entries. */
int scn_rules_fill(const int32_t* table, int n_off, int64_t n_out, const int32_t* block_sums, int32_t* in_rows,
                   int32_t* out_rows, int32_t* seg_of, scn_stream_t stream);

/* All index structures of one forward pass of an n_levels U-Net in one call (what the reference's scn.Metadata builds
 * lazily on the host while InputLayer / SubmanifoldConvolution / Convolution run: custom_operations.py:67-86,
 * module_factory.py:232-234,404-406): InputLayer rules, and per level the SubM-k^3 table + rule scan + tiles and the
 * size=stride=2 coarse sites + child table + rule scan + tiles.  Same kernels, same order and bit-identical results as
 * the step-by-step entry points above; buffers are carved out of `workspace` (256-byte aligned, at least
 * scn_pyramid_workspace_bytes(n_points, n_levels, k) bytes), whose layout comes back in `desc` (HOST int64
 * [SCN_PYRAMID_DESC_LEN], offsets in bytes):
 *   desc[0] n_levels  [1] n_points  [2] bytes used  [3] out-of-range coordinate count
 *   desc[4] item_row int32[n_points]  [5] row_count int32 (first n_0 valid)  [6] row_first int32  [7] int32 coords [n_points][4]
 *   level l at L = desc + 8 + l*SCN_PYRAMID_LEVEL_STRIDE:
 *     L[0] rows n_l  [1] hash capacity  [2] coords int32[n][4]  [3] hash keys u64[cap]  [4] hash rows int32[cap]
 *     L[5] table int32[k^3][n]  [6] scan block sums int32[L[7]]  [8] prefix int64[k^3+1] (device)
 *     L[9] perm  [10] tstab  [11] tile_mask  [12] tile_order  [13] number of tiles
 *     strided rulebook l -> l+1:  L[14] parent int32[n_l]  [15] fine_off int32[n_l]  [16] child int32[8][n_{l+1}]
 *     L[17] block sums int32[L[18]]  [19] prefix int64[9] (device)  L[20..23] perm, tstab, tile_mask, tile_order  [24] tiles
 *     L[25..25+k^3] SubM rule prefix (host copy)   L[53..61] strided rule prefix (host copy)
 *     L[64] / L[65] compacted SubM rules in_rows / out_rows int32[prefix[k^3]]   L[66] / L[67] the strided ones
 * The call synchronises `stream` (row counts of the levels, rule-list sizes) and holds no interpreter state: a helper
 * thread may run it for the next batch while the caller queues the current one. */
#define SCN_PYRAMID_MAX_LEVELS 8
#define SCN_PYRAMID_LEVEL_STRIDE 72
#define SCN_PYRAMID_DESC_LEN (8 + SCN_PYRAMID_MAX_LEVELS * SCN_PYRAMID_LEVEL_STRIDE)
int64_t scn_pyramid_workspace_bytes(int64_t n_points, int n_levels, int k);
int scn_pyramid_build(const int64_t* coords, int64_t n_points, int n_levels, int k, void* workspace,
                      int64_t workspace_bytes, int64_t* desc, scn_stream_t stream);
/* The same build with options.  SCN_PYRAMID_TWO_QUEUES: the SubM work of every level (neighbour table, rule scan, mask sort,
 * tiles) goes to a library-owned side stream beside the chain that numbers the levels on `stream`; `stream` is ordered behind
 * the side stream before the call returns.  Shorter when the build has the GPU to itself (0.95 vs 1.18 ms at 150 k points,
 * four levels), slightly in the way when it runs next to another batch's matrix kernels -- so the inline callers (a forward
 * that builds its own index structures, the ROI batch) ask for it and the pipelined prefetch does not.  Same structures. */
#define SCN_PYRAMID_TWO_QUEUES 1
#define SCN_PYRAMID_XCD_ORDER 2   /* the SubM tiles of every level also get the XCD-local hand-out order (scn_tiles_build_x) */
/* SCN_PYRAMID_FUSED (round 4, k = 3): the same structures, bit for bit, from a build that never waits for the device before
 * its end: level sizes stay in device memory (every index kernel reads its row count from the word the numbering kernel
 * wrote; grids and buffer offsets by the upper bound n_l <= n_points), the levels run side by side inside each launch,
 * flag / scan / fill of the numbering is one look-back pass, and ONE device -> host copy brings back every size
 * (14 + n_levels - 1 launches and one host wait; the builder above: ~136 launches, one wait per level + one).  Only the
 * hash tables differ (every level gets the capacity of n_points).  Overrides SCN_PYRAMID_TWO_QUEUES. */
#define SCN_PYRAMID_FUSED 4
int scn_pyramid_build_ex(const int64_t* coords, int64_t n_points, int n_levels, int k, void* workspace,
                         int64_t workspace_bytes, int64_t* desc, int flags, scn_stream_t stream);

/* Sparse ROI crop (roi_select_sparse.py:157-180 get_inside_indicator / select_features / select_coords / roi_cut) as
 * count -> scan -> scatter; no [boxes][points] object is ever built.  boxes int32 [bb][8] =
 * (start x,y,z,sample ; stop x,y,z,sample+1), half-open.  The result is the reference's selection order: box-major,
 * ascending point row.
 *   scn_roi_units(n)        number of 256-point units of n points
 *   scn_roi_count           unit_offsets int32 [bb * scn_roi_units(n) + bb]: after the call, for every (box, unit) the
 *                           start of its run relative to the box's first output row (+ bb box totals behind them);
 *                           prefix (device int64 [bb+1]): first output row of every box, prefix[bb] = M.
 *                           prefix_host != NULL: copied back (synchronises stream once); NULL: asynchronous.
 *   scn_roi_fill            src_row[M] (point row), box_of[M], out_coords int64 [M][4] = (x,y,z of the point, box) --
 *                           select_coords' extended coordinates (may be NULL).
 *   scn_roi_inside          the dense bool matrix roi_cut also returns (:180), rebuilt from the list ON REQUEST:
 *                           inside_u8 [bb][n] zero-filled, then 1 at (box_of[m], src_row[m]). */
int64_t scn_roi_units(int64_t n);
int scn_roi_count(const int32_t* coords, int64_t n, const int32_t* boxes, int bb, int32_t* unit_offsets, int64_t* prefix,
                  int64_t* prefix_host, scn_stream_t stream);
int scn_roi_fill(const int32_t* coords, int64_t n, const int32_t* boxes, int bb, const int32_t* unit_offsets,
                 const int64_t* prefix, int32_t* src_row, int32_t* box_of, int64_t* out_coords, scn_stream_t stream);
int scn_roi_inside(const int32_t* src_row, const int32_t* box_of, int64_t m, int64_t n, int bb, uint8_t* inside_u8,
                   scn_stream_t stream);
/* out_coords[m] = (x,y,z of coords[src_row[m]], box_of[m]) as int64 rows (select_coords, roi_select_sparse.py:136-149) */
int scn_roi_coords(const int32_t* coords, const int32_t* src_row, const int32_t* box_of, int64_t m, int64_t* out_coords,
                   scn_stream_t stream);
/* BBoxTransformerSlice (roi_select_bbox_transform.py:56-70,87-97): boxes fp32 [bb][2][3] -> optional Divider
 * (box / resize, fp32, per axis; :15-21) -> floor(start), ceil(stop) (ndsis/utils/bbox.py:87-106) -> optional clip of
 * start to [0,S-1] / stop to [1,S] (bbox.py:62-84) -> sample interval appended -> int32 [bb][8] as scn_roi_count expects.
 * resize3_or_null: device float[3]. */
int scn_roi_boxes(const float* boxes, const int32_t* box_sample, int bb, const int32_t* spatial_size3_or_null,
                  const float* resize3_or_null, int32_t* out, scn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Feature path: gather-GEMM-scatter on fp32 MFMA
 * ---------------------------------------------------------------------------------------- */

/* Output-stationary gather GEMM:   Y[r] = residual[r] + bias + sum_o  in(X[table[o][r]]) . W[o']
 * (SubmanifoldConvolution fwd, module_factory.py:404-406; Convolution fwd via the child table, :232-234;
 *  NetworkInNetwork fwd with table == NULL (identity, n_off = 1), :366-367; and the backward-data of
 *  SubM / Deconvolution / NiN with SCN_F_W_TRANSPOSED).
 * W is [n_off][cin][cout] row-major as stored by the layer; with SCN_F_W_TRANSPOSED the call sees it as
 * [n_off][cout_of_call][cin_of_call]^T, i.e. pass the layer's weight unchanged and swap cin/cout.
 * bias, residual may be NULL.  relu_mask (may be NULL, [n_out][cout]): elements of Y are zeroed where relu_mask <= 0
 * (fuses the ReLU backward that follows a backward-data GEMM). */
int scn_gemm_table(const float* X, int64_t n_in, int cin, const int32_t* table, int n_off, int64_t n_out,
                   const float* W, const float* bias, const float* residual, const float* relu_mask, float* Y,
                   int cout, int flags, scn_stream_t stream);

/* Mask-sorted row tiles of a rule table (input of scn_conv_tiles).  Rows are regrouped by their offset mask
 * (bit o set <=> table[o][r] >= 0) with a stable radix sort; nt = ceil(n/16) tiles of 16 sorted rows.
 *   perm      int32 [nt*16]          sorted position -> original row (-1 padding)
 *   tstab     int32 [nt][n_off][16]  tstab[t][o][i] = table[o][perm[16t+i]]
 *   tile_mask uint32 [nt]            OR of the row masks of the tile
 *   tile_order int32 [nt]            tile ids by offset count descending (hand-out order of scn_conv_tiles) */
int64_t scn_tiles_scratch_bytes(int n_off, int64_t n);
int scn_tiles_build(const int32_t* table, int n_off, int64_t n, int32_t* perm, int32_t* tstab, uint32_t* tile_mask,
                    int32_t* tile_order, void* scratch, scn_stream_t stream);
/* The same build with a second hand-out order behind the first (with_x != 0): tile_order then holds scn_tiles_order_ints(n, 1)
 * = 2 NT + 16 ints -- [NT] tile ids by offset count descending (as above), [NT] tile ids by (spatial bin, offset count
 * descending) where the bin of a tile is (its first row * 8 / n) (rows are numbered in the order the points arrive, so a row
 * range is a region of the scene), [9] bin starts in that second list (bin_start[8] = NT).  scn_conv_tiles_bf16 with
 * SCN_F_TILE_ORDER_X hands the tiles of bin x to the workgroups of XCD x: their row gathers then meet in one L2 (the bf16 tile
 * kernel waits for L2 misses at the fine levels: 87 -> 71 us per launch at 600 k voxels, 24.5 -> 21.5 at 150 k; the fp32
 * kernel is bound by its matrix pipe and loses to the shorter per-bin lists, so it keeps the first order).
 * Round 6 (ABI 5): `with_x` bit 0 = the second order; bits 8-10 = lb, log2 of the ROW BINS in the sort key of a 27-offset table
 * (0 = the plain mask sort; lb <= 5): rows are sorted by (row * 2^lb / n, offset mask), so a tile's 16 rows come from one
 * region of the scene -- what the bf16 builds use at level 0 with lb = 3 (600 k voxels: 71 -> 53 us per launch; results are
 * independent of the row order inside the tile tables).  scn_tiles_order_ints takes bit 0 only. */
int64_t scn_tiles_order_ints(int64_t n, int with_x);
int scn_tiles_build_x(const int32_t* table, int n_off, int64_t n, int32_t* perm, int32_t* tstab, uint32_t* tile_mask,
                      int32_t* tile_order, int with_x, void* scratch, scn_stream_t stream);

/* Stable LSD radix sort of (uint32 key, int32 value) pairs on the low `bits` key bits -- the primitive behind
 * scn_tiles_build (rows by offset mask, tiles by offset count); exported so that it can be checked on its own.  It takes
 * the place of the hash-map iteration order upstream leaves undefined (canonical orders are sorted orders).
 * vals == NULL: values are the input positions.  Outputs must not alias the inputs.  ceil(bits/9) passes of <= 512 bins;
 * n <= 4096 runs as one launch out of LDS. */
int64_t scn_sort_pairs_scratch_bytes(int64_t n);
int scn_sort_pairs(const uint32_t* keys, const int32_t* vals, int64_t n, int bits, uint32_t* keys_out,
                   int32_t* vals_out, void* scratch, scn_stream_t stream);

/* THE HOT KERNEL.  Output-stationary convolution over mask-sorted tiles, accumulators in registers, weights in LDS:
 *     Y[r] = residual[r] + bias + sum_o in(X[table[o][r]]) . W[o']          (same result as scn_gemm_table)
 * Serves SubmanifoldConvolution fwd / backward-data (module_factory.py:404-406), Convolution fwd (:232-234) and
 * Deconvolution backward-data (:256-258).  n_off <= 27.  Per-row accumulation order is fixed (offsets ascending). */
int64_t scn_conv_tiles_scratch_bytes(int cin, int64_t n_out, int cout);   /* K-chunk slabs */
/* Layers with more than 32 input channels split K over workgroups.  `arrival` (int32, at least
 * scn_conv_tiles_arrival_counters(cin, n_out, cout) entries; may be shared by every call on one stream) lets the kernel add
 * the K-chunk partial sums itself: the wave that arrives last at a (tile, column chunk) adds them in ascending K-chunk
 * order.  CONTRACT: all entries are ZERO when the call is made; the kernel leaves them zero.  arrival == NULL (or
 * SCN_F_SPLIT_SUM): the partial sums are added by a second launch instead -- same association, same bits. */
int64_t scn_conv_tiles_arrival_counters(int cin, int64_t n_out, int cout);
int scn_conv_tiles(const float* X, int64_t n_in, int cin, const int32_t* tstab, const uint32_t* tile_mask, const int32_t* perm,
                   const int32_t* tile_order, int n_off, int64_t n_out, const float* W, const float* bias, const float* residual,
                   const float* relu_mask, float* Y, int cout, int flags, void* scratch, int32_t* arrival,
                   scn_stream_t stream);
/* CHAINED launch (round 6): 2-4 DEPENDENT convolutions over the same tiles -- the SubM 3^3 launches of a level's residual
 * units (module_factory.py:127-183 builds x + SubM3(ReLU(SubM3(ReLU(x)))); two units per level, :513-530), forward or
 * backward-data -- as ONE launch.  Role r reads what role r-1 wrote (X, residual or relu_mask of role r may be Y of an
 * earlier role); the workgroups of role r are placed on the CUs role r-1 leaves, stage their weights there and wait for
 * role r-1 to complete before their first row gather (scn_conv_ts.hip, "CHAIN").  Same arithmetic per role as n_roles
 * scn_conv_tiles calls: same bits.  Shapes the chained kernel does not cover (channel counts that are no multiple of 32,
 * levels on the four-waves-per-tile loop, SCN_TS_NO_CHAIN=1) run as n_roles plain calls.  All roles share n_in, cin, the
 * tile structures, n_out, cout, SCN_F_W_TRANSPOSED and SCN_F_OFF_REVERSE; `flags` may differ in the other bits. */
typedef struct scn_conv_role {
    const void* X; const void* W; const void* bias; const void* residual; const void* relu_mask; void* Y;
    int32_t flags; int32_t reserved;
} scn_conv_role;
int scn_conv_tiles_chain(int n_roles, const scn_conv_role* roles, int64_t n_in, int cin, const int32_t* tstab,
                         const uint32_t* tile_mask, const int32_t* perm, const int32_t* tile_order, int n_off, int64_t n_out,
                         int cout, void* scratch, int32_t* arrival, scn_stream_t stream);
/* out[0] = chained launches, out[1] = roles they carried, since the last reset (fp32 and bf16 together). */
void scn_conv_tiles_chain_counts(int64_t out[2], int reset);
/* Which kernel variant the scn_conv_tiles calls of this process took since the last reset: out[0] = launches on the fast path
 * (raw-buffer gathers: 16-byte rows, < 2^23 rows, < 4 GB slabs), out[1] = launches that fell back to the general kernel
 * (same results, 64-bit addressing), out[2] = launches with the in-launch K reduction, out[3] = launches that left the K
 * reduction to a second launch.  reset != 0 zeroes the counters after reading.  (bench.py reports it as `fast_path`: a
 * workload that outgrows the fast path's limits changes kernels, and the JSON line says so.) */
void scn_conv_tiles_path_counts(int64_t out[4], int reset);
/* Launches (a subset of out[0]) that took the four-waves-per-tile loop of latency-bound levels: fewer (tile, slice) pairs
 * than half the chip's waves -- the coarse levels of the reference's six-level plan (scannet_config/run.py:539-549: 2 983 /
 * 734 / 180 rows), deep levels of an ROI batch.  Same arithmetic per product, an element's sum associated by wave share. */
int64_t scn_conv_tiles_split_count(int reset);
/* Second half of a scn_conv_tiles call made with SCN_F_SPLIT_SUM: adds the K-chunk slabs (no-op when cin <= 32: the
 * tile kernel has written Y).  Same arguments as that call. */
int scn_conv_tiles_finish(int cin, int64_t n_out, const float* bias, const float* residual, const float* relu_mask,
                          float* Y, int cout, int flags, void* scratch, scn_stream_t stream);

/* The same convolution for bf16 STORAGE (BASELINE configs 3-5): X, residual, relu_mask and Y are bf16
 * (uint16 bit patterns, row-major, cin % 8 == 0 and cout % 8 == 0: 16-byte row pieces); products accumulate in fp32 on
 * v_mfma_f32_16x16x32_bf16 and the result is rounded to bf16 once.  Same tiles, flags and per-row summation order as
 * scn_conv_tiles.  The layer's master weights stay fp32: scn_conv_tiles_bf16_pack rounds them (round-to-nearest-even)
 * into `image` (scn_conv_tiles_bf16_image_bytes bytes, 16-byte aligned), laid out as the kernel's workgroups stage it;
 * SCN_F_W_TRANSPOSED / SCN_F_OFF_REVERSE are properties of the IMAGE (pass them to the pack; the convolution ignores
 * them), so one layer has a forward image and a backward-data image.  An image is valid until the weights change.
 * `arrival`: zeroed counters for the in-launch K reduction, contract as in scn_conv_tiles
 * (scn_conv_tiles_bf16_arrival_counters entries); NULL: second launch.
 * Serves SubmanifoldConvolution / Convolution forward and SubM / Deconvolution backward-data
 * (module_factory.py:232-234,256-258,404-406). */
int64_t scn_conv_tiles_bf16_image_bytes(int cin, int cout, int n_off);
int scn_conv_tiles_bf16_pack(const float* W, int cin, int cout, int n_off, int flags, uint16_t* image, scn_stream_t stream);
/* The same for n images in one launch (all arrays on the HOST; W / image entries are device pointers): a network packs the
 * forward and backward-data images of all its convolutions once per step. */
int scn_conv_tiles_bf16_pack_many(int n, const float* const* W_host, const int32_t* cin_host, const int32_t* cout_host,
                                  const int32_t* n_off_host, const int32_t* flags_host, uint16_t* const* image_host,
                                  scn_stream_t stream);
int64_t scn_conv_tiles_bf16_scratch_bytes(int cin, int64_t n_out, int cout);
int64_t scn_conv_tiles_bf16_arrival_counters(int cin, int64_t n_out, int cout);
int scn_conv_tiles_bf16(const uint16_t* X, int64_t n_in, int cin, const int32_t* tstab, const uint32_t* tile_mask,
                        const int32_t* perm, const int32_t* tile_order, int n_off, int64_t n_out, const uint16_t* image,
                        const float* bias, const uint16_t* residual, const uint16_t* relu_mask, uint16_t* Y, int cout,
                        int flags, void* scratch, int32_t* arrival, scn_stream_t stream);

/* Rule-list gather GEMM with row scatter:  Y[out_rows[p]] = bias + in(X[in_rows[p]]) . W[o(p)]
 * where every output row occurs in exactly one pair (Deconvolution fwd, module_factory.py:256-258; backward-data
 * of Convolution).  prefix_host = int64[n_off+1] on the HOST. */
int scn_gemm_rules(const float* X, int cin, const int32_t* in_rows, const int32_t* out_rows,
                   const int64_t* prefix_host, int n_off, const float* W, const float* bias, const float* relu_mask,
                   float* Y, int cout, int flags, scn_stream_t stream);

/* Weight gradient:  dW[o] = sum_{p in R_o} in(X[in_rows[p]])^T . dY[out_rows[p]]     (dW fully overwritten)
 * in_rows == out_rows == NULL means the identity rule list of length prefix_host[1] (NetworkInNetwork).
 * One kernel streams the rules (MFMA operands straight from the gathered rows, no LDS staging) into per-unit partial
 * blocks, a second adds them in a fixed order: bitwise reproducible.  scratch: scn_wgrad_scratch_bytes(...) bytes. */
int64_t scn_wgrad_scratch_bytes(int cin, int cout, const int64_t* prefix_host, int n_off);
int scn_wgrad_rules(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                    const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, void* scratch,
                    int flags, scn_stream_t stream);

/* Weight AND bias gradient in one pass: as scn_wgrad_rules, plus db[c] = sum over the rules of the offsets named in
 * the bit mask db_offsets of dY[out_p][c].  The caller names offsets whose rule lists together contain every output row
 * exactly once (centre offset of a submanifold conv: 1 << (k^3/2); all offsets of a Deconvolution; the identity list),
 * so db equals the column sum of dY.  Any channel count (rows that are not 16-byte aligned take element-wise loads). */
/* Row GEMM over the parts of a JoinTable, A tile staged through LDS (scn_gemm_lt.hip):
 *     [Y0 | Y1][r] = residual[r] + bias + in([X0[r] | X1[r]]) . W        for r < n
 * Two sources (cx1 > 0): NetworkInNetwork(2C -> C) applied to JoinTable([up, skip]) without the concatenated slab
 * (module_factory.py:298-301, 365-367); W is the layer's [cx0 + cx1][cy0] weight.  Two destinations (cy1 > 0) with
 * SCN_F_W_TRANSPOSED: its backward-data, dUp = dY . W[:c0]^T and dSkip = dY . W[c0:]^T from one read of dY; W is the layer's
 * weight unchanged ([cy0 + cy1][cx0] seen from this call).  cx1 == cy1 == 0: the identity-table scn_gemm_table.
 * Channel counts of the sources are multiples of 8, slabs 16-byte aligned; residual / relu_mask only with one destination.
 * bf16_storage != 0: features, residual, relu_mask and outputs are uint16 bf16 bit patterns (W, bias fp32; exact widening,
 * fp32 arithmetic, one rounding of the result). */
int scn_gemm_rows2(const void* X0, int cx0, const void* X1, int cx1, int64_t n, const float* W, const float* bias,
                   const void* residual, const void* relu_mask, void* Y0, int cy0, void* Y1, int cy1, int flags,
                   int bf16_storage, scn_stream_t stream);

/* bf16-storage forms of scn_gemm_table / scn_gemm_rules (features, residual, relu_mask and Y are uint16 bf16 bit
 * patterns; W and bias fp32; rows are widened exactly and the arithmetic is the fp32 kernels'; one rounding of Y). */
int scn_gemm_table_bf16(const uint16_t* X, int64_t n_in, int cin, const int32_t* table, int n_off, int64_t n_out,
                        const float* W, const float* bias, const uint16_t* residual, const uint16_t* relu_mask,
                        uint16_t* Y, int cout, int flags, scn_stream_t stream);
int scn_gemm_rules_bf16(const uint16_t* X, int cin, const int32_t* in_rows, const int32_t* out_rows,
                        const int64_t* prefix_host, int n_off, const float* W, const float* bias,
                        const uint16_t* relu_mask, uint16_t* Y, int cout, int flags, scn_stream_t stream);

/* The same weight gradient for bf16-stored operands (X, dY: uint16 bit patterns; dW fp32): rows are gathered packed
 * and widened to fp32 in registers (exact), arithmetic and summation order are those of scn_wgrad_rules.  Scratch:
 * scn_wgrad_scratch_bytes. */
int scn_wgrad_rules_bf16(const uint16_t* X, int cin, const uint16_t* dY, int cout, const int32_t* in_rows,
                         const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, void* scratch,
                         int flags, scn_stream_t stream);
/* scn_wgrad_bias_rules for bf16-stored operands (dW, db fp32). */
int scn_wgrad_bias_rules_bf16(const uint16_t* X, int cin, const uint16_t* dY, int cout, const int32_t* in_rows,
                              const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, float* db,
                              uint32_t db_offsets, void* scratch, int flags, scn_stream_t stream);
int scn_wgrad_bias_rules(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                         const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, float* db,
                         uint32_t db_offsets, void* scratch, int flags, scn_stream_t stream);

/* The two weight (and bias) gradients of a pre-activation residual unit (module_factory.py:127-183: x + SubM3(ReLU(SubM3(
 * ReLU(x))))) in ONE launch and one sum: both convolutions walk the same rule list with the same channel counts;
 * (X0, dY0) and (X1, dY1) are their operand pairs (fp32).  dW = [2][n_off][cin][cout]; db = [2][cout], or NULL with
 * db_offsets == 0.  Per problem the units, the per-unit arithmetic and the fixed-order sum are those of
 * scn_wgrad_bias_rules under this call's plan; the launch pays its tail and its fixed costs once for twice the work.
 * Scratch: scn_wgrad_scratch_bytes2.  prefix_host[0] must be 0; n_off <= 32. */
/* Deferred unit sums (round 3d).  Between scn_wgrad_defer_begin() and scn_wgrad_defer_flush(stream) ON THE CALLING THREAD the
 * scn_wgrad_* calls launch their unit kernels only and record their sum; the flush adds the recorded sums in batched launches
 * (up to six launches' sums per launch; per layer the arithmetic and its order are unchanged -- same bits).  Every call made
 * in between needs a scratch region of its own that stays untouched until the flush.  Used by scn_exec_run for the weight
 * gradients of a network level (27 sum launches of ~7 us per backbone step become 8). */
int scn_wgrad_defer_begin(void);
int scn_wgrad_defer_flush(scn_stream_t stream);
int64_t scn_wgrad_scratch_bytes2(int cin, int cout, const int64_t* prefix_host, int n_off);
int scn_wgrad_bias_rules2(const float* X0, const float* dY0, const float* X1, const float* dY1, int cin, int cout,
                          const int32_t* in_rows, const int32_t* out_rows, const int64_t* prefix_host, int n_off,
                          float* dW, float* db, uint32_t db_offsets, void* scratch, int flags, scn_stream_t stream);

/* The general form: n_prob = 2 ... 4 operand pairs (host arrays of device pointers) on one rule list -- the four
 * convolutions of two stacked residual units (module_factory.py:513-530 `num_units=2`) in one launch.
 * dW = [n_prob][n_off][cin][cout], db = [n_prob][cout] or NULL with db_offsets == 0. */
int64_t scn_wgrad_scratch_bytes_n(int cin, int cout, const int64_t* prefix_host, int n_off, int n_prob);
int scn_wgrad_bias_rules_n(const float* const* Xs, const float* const* dYs, int n_prob, int cin, int cout,
                          const int32_t* in_rows, const int32_t* out_rows, const int64_t* prefix_host, int n_off,
                          float* dW, float* db, uint32_t db_offsets, void* scratch, int flags, scn_stream_t stream);
int scn_wgrad_bias_rules_n_bf16(const uint16_t* const* Xs, const uint16_t* const* dYs, int n_prob, int cin, int cout,
                               const int32_t* in_rows, const int32_t* out_rows, const int64_t* prefix_host, int n_off,
                               float* dW, float* db, uint32_t db_offsets, void* scratch, int flags, scn_stream_t stream);
/* ... for bf16-stored operand pairs (uint16 bit patterns; dW, db fp32; the bf16-MFMA kernel where it applies). */
int scn_wgrad_bias_rules2_bf16(const uint16_t* X0, const uint16_t* dY0, const uint16_t* X1, const uint16_t* dY1, int cin,
                               int cout, const int32_t* in_rows, const int32_t* out_rows, const int64_t* prefix_host,
                               int n_off, float* dW, float* db, uint32_t db_offsets, void* scratch, int flags,
                               scn_stream_t stream);

/* db[c] = sum_r dY[r][c]   (bias gradient of every conv-type layer).  scratch: SCN_COLSUM_BLOCKS*c floats. */
#define SCN_COLSUM_BLOCKS 512
int scn_colsum(const float* dY, int64_t n, int c, float* db, void* scratch, scn_stream_t stream);
int scn_colsum_bf16(const uint16_t* dY, int64_t n, int c, float* db, void* scratch, scn_stream_t stream);   /* bf16-stored dY */

/* ------------------------------------------------------------------------------------------
 * Elementwise / normalisation / IO
 * ---------------------------------------------------------------------------------------- */

/* scn.ReLU (module_factory.py:88) and scn.AddTable (:52-54,308) */
int scn_relu_fwd(const float* X, int64_t count, float* Y, scn_stream_t stream);
int scn_relu_bwd(const float* X, const float* dY, int64_t count, float* dX, scn_stream_t stream);
int scn_add(const float* A, const float* B, int64_t count, float* Y, scn_stream_t stream);

/* scn.BatchNormReLU / BatchNormLeakyReLU (module_factory.py:92-102): training statistics over all rows. */
int64_t scn_bn_scratch_bytes(int c);
int scn_bn_stats(const float* X, int64_t n, int c, float* mean, float* var_biased,
                 void* scratch /* scn_bn_scratch_bytes(c) */, scn_stream_t stream);
int scn_bn_fwd(const float* X, int64_t n, int c, const float* mean, const float* var, float eps, const float* gamma,
               const float* beta, float leak, float* Y, scn_stream_t stream);
/* dX, dgamma, dbeta for y = lrelu((x-mean)*invstd*gamma+beta); training=1 propagates through the batch statistics */
int scn_bn_bwd(const float* X, const float* dY, int64_t n, int c, const float* mean, const float* var, float eps,
               const float* gamma, const float* beta, float leak, int training, float* dX, float* dgamma, float* dbeta,
               void* scratch /* scn_bn_scratch_bytes(c) */, scn_stream_t stream);

/* The same layer with batch statistics over ALL ranks of a data-parallel step (the reference's BatchNorm sees its whole
 * batch as one feature matrix, module_factory.py:92-102; one scene per GPU splits that matrix): the reductions are
 * exposed as float64 [2][c] sums so that the caller can all-reduce them between the two halves.
 *   forward : scn_bn_sums -> (sum x, sum x^2) -> all-reduce with the row counts -> mean, var -> scn_bn_fwd
 *   backward: scn_bn_bwd_reduce -> local dgamma, dbeta and (sum g, sum g x^) -> all-reduce -> scn_bn_bwd_apply with
 *             n_stat = rows of all ranks. */
int scn_bn_sums(const float* X, int64_t n, int c, double* sums, void* scratch, scn_stream_t stream);
int scn_bn_bwd_reduce(const float* X, const float* dY, int64_t n, int c, const float* mean, const float* var, float eps,
                      const float* gamma, const float* beta, float leak, float* dgamma, float* dbeta, double* sums,
                      void* scratch, scn_stream_t stream);
int scn_bn_bwd_apply(const float* X, const float* dY, int64_t n, int c, const float* mean, const float* var, float eps,
                     const float* gamma, const float* beta, float leak, const double* sums, int64_t n_stat, float* dX,
                     scn_stream_t stream);

/* InputLayerFunction (custom_operations.py:72-80): mode 0 copy, 1 last, 2 first, 3 sum, 4 mean.
 * acc64: scratch of n_rows*c doubles (modes 3,4).  row_last (mode 1): scratch of n_rows int32. */
int scn_input_fwd(const float* feats, const int32_t* item_row, const int32_t* row_count, const int32_t* row_first,
                  int64_t n_items, int64_t n_rows, int c, int mode, float* Y, double* acc64, int32_t* row_last,
                  scn_stream_t stream);
int scn_input_bwd(const float* dY, const int32_t* item_row, const int32_t* row_count, const int32_t* row_first,
                  const int32_t* row_last, int64_t n_items, int c, int mode, float* dfeats, scn_stream_t stream);
/* OutputLayerFunction (custom_operations.py:7-10): Y[i] = X[item_row[i]]; backward = segment sum. */
int scn_gather_rows(const float* X, const int32_t* rows, int64_t m, int c, float* Y, scn_stream_t stream);
int scn_segment_sum(const float* dY, const int32_t* item_row, int64_t n_items, int64_t n_rows, int c, float* dX,
                    double* acc64, scn_stream_t stream);

/* Per-sample pooling of a slab -- the reference's SparseGlobalPool / split_batch (ndsis/modules/custom_operations.py:24-59;
 * sparse class network model.py:507-512, module_factory.py:655-657).  The sample of row r is coords[r][3] (the grid's int32
 * [n][4] rows, x y z b).  op: 0 mean (torch.mean, the reference's default), 1 sum, 2 max (torch.amax).
 *   fwd: Y[b][:] over the rows of sample b; a sample without rows pools to zeros (custom_operations.py:53-54).
 *        cnt[b] = rows of sample b; *unsorted != 0 iff some row has a smaller sample index than the row before it.
 *   bwd: mean dX[r] = dY[b] / cnt[b]; sum dX[r] = dY[b]; max: dY[b][c] split evenly among the rows that hold the maximum.
 * scratch: scn_segment_pool_scratch_bytes(n_samples, c) (fp64 accumulators / tie counts).  No host synchronisation.
 * scn_sample_counts: cnt and *unsorted alone (split_batch: rows grouped by sample are returned as row ranges). */
int64_t scn_segment_pool_scratch_bytes(int n_samples, int c);
int scn_sample_counts(const int32_t* coords, int64_t n, int n_samples, int32_t* cnt, int32_t* unsorted, scn_stream_t stream);
int scn_segment_pool_fwd(const float* X, const int32_t* coords, int64_t n, int c, int n_samples, int op, float* Y,
                         int32_t* cnt, int32_t* unsorted, void* scratch, scn_stream_t stream);
int scn_segment_pool_bwd(const float* X, const float* Y, const float* dY, const int32_t* coords, int64_t n, int c,
                         int n_samples, int op, const int32_t* cnt, float* dX, void* scratch, scn_stream_t stream);

/* scn.SparseToDense (module_factory.py:429-435): out [B][C][X][Y][Z] pre-zeroed by the caller. */
int scn_sparse_to_dense_fwd(const float* X, const int32_t* coords, int64_t n, int c, const int64_t* size3_host,
                            float* out, scn_stream_t stream);
int scn_sparse_to_dense_bwd(const float* dOut, const int32_t* coords, int64_t n, int c, const int64_t* size3_host,
                            float* dX, scn_stream_t stream);

/* scn.MaxPooling / scn.AveragePooling with pool_size = pool_stride = 2 (module_factory.py:315-354), on the child /
 * parent tables of the strided rulebook.  max: Y[c] = max(0, existing children); avg: Y[c] = sum(existing children) / 8. */
int scn_pool_fwd(const float* X, const int32_t* child, int64_t n_coarse, int c, int average, float* Y,
                 scn_stream_t stream);
int scn_pool_bwd(const float* X, const float* Y, const float* dY, const int32_t* parent, int64_t n_fine, int c,
                 int average, float* dX, scn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Mask-head epilogue on the device: consumers of the ROI selection in CSR form
 * (rows of the crop are box-major; box_of[r] = box, src_point[r] = point row in the batch).
 * ---------------------------------------------------------------------------------------- */

/* SparseMaskPredictor.forward (model.py:859-882): out (pre-zeroed, the per-sample dense [boxes][points] masks back to
 * back) gets sigmoid(scores[r][class_of_box[box]]) at row_base[box] + src_point[r]; classes < 0, >= num_valid
 * (num_valid > 0) or >= k leave zeros.  row_base[box] = offset of the box's row in out - first point row of its sample. */
int scn_mask_scatter(const float* scores, int64_t m, int k, const int32_t* src_point, const int32_t* box_of,
                     const int64_t* class_of_box, int num_valid, const int64_t* row_base, float* out,
                     scn_stream_t stream);

/* SparseMaskLossSelector gathers (model.py:1157-1227): pred[r] = scores[r][label_of_box[box]],
 * gt[r] = gt_flat[gt_base[box] + src_point[r]] (gt_base[box] = offset of the associated ground-truth mask row in
 * gt_flat - first point row of the sample; < 0 or label < 0: box not kept, rows get 0 and keep_row[r] = 0).
 * keep_row may be NULL.  _bwd: dscores[r][c] = dpred[r] at c = label of a kept box, 0 elsewhere. */
int scn_mask_gather(const float* scores, int64_t m, int k, const int32_t* src_point, const int32_t* box_of,
                    const int64_t* label_of_box, const int64_t* gt_base, const float* gt_flat, float* pred, float* gt,
                    uint8_t* keep_row, scn_stream_t stream);
int scn_mask_gather_bwd(const float* dpred, int64_t m, int k, const int32_t* box_of, const int64_t* label_of_box,
                        const int64_t* gt_base, float* dscores, scn_stream_t stream);

/* Greedy non-maximum suppression of score-sorted 3-D boxes (ndsis/utils/bbox.py:713-759 non_maximum_supression as
 * called by ProposalSelector, proposal_selector.py:60-89): boxes fp32 [batch][n][2][3] = (start xyz, stop xyz), sorted
 * by descending score inside a scene; keep[batch][n] = 1 for boxes that survive.  A box is suppressed by an earlier
 * surviving box whose IoU with it is > overlap_threshold (fp32, the reference's operation order: bit-exact decisions).
 * n <= 8192.  One workgroup per scene, one launch instead of n. */
int scn_nms(const float* boxes, int batch, int n, float overlap_threshold, uint8_t* keep, scn_stream_t stream);
/* The same selection, round 5: the upper triangle of the n x n suppression relation as a bit matrix (one wave per 64 x 64 block,
 * spread over the chip) and ONE serial walk over its rows out of LDS -- keep[] equals scn_nms's bit for bit; 1024 boxes:
 * 12 + 42 us (rocprofv3, one scene) instead of 600 (scn_nms walks the boxes with two workgroup barriers each).  n <= 4096.
 * scratch: scn_nms_scratch_bytes(batch, n) = batch * n * ceil(n / 64) * 8 bytes, need not be initialised. */
int64_t scn_nms_scratch_bytes(int batch, int n);
int scn_nms_bits(const float* boxes, int batch, int n, float overlap_threshold, uint8_t* keep, void* scratch,
                 scn_stream_t stream);

/* Exact top-k of a score field with the boxes gathered along -- what ProposalSelector runs before its NMS
 * (ndsis/modules/proposal_selector.py:60-75: torch.topk(rpn_score, num_keep_pre_nms, sorted) and rpn_bbox[batch, indices]).
 * scores fp32 [batch][n], boxes fp32 [batch][n][6] or NULL; out_scores [batch][k] descending, out_index int64 [batch][k],
 * out_boxes [batch][k][6].  Equal scores come out by ascending index (torch.topk leaves their order open); NaN counts as
 * the largest value, as in torch.  Radix select (two 11-bit histogram passes over an order-preserving key, one compaction,
 * one LDS sort of the candidates): 4 launches instead of torch.topk's 19 on a [1, 524 288] field.  k <= 2048, k <= n.
 * scratch: scn_topk_scratch_bytes(batch) bytes that are ZERO before the first call; every call leaves them zero again
 * (after a failed call: zero them).  One stream at a time per scratch. */
int64_t scn_topk_scratch_bytes(int batch);
int scn_topk_boxes(const float* scores, const float* boxes, int batch, int64_t n, int k, float* out_scores,
                   int64_t* out_index, float* out_boxes, void* scratch, scn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Voxelisation front-end: the deterministic core of augment_coords
 * (ndsis/data/sparse_augmentation.py:81-126) and the batch column of collate_fn (data.py:95-98).
 * The reference's random draws (distortion matrix, sub-pixel offset, cut-out start) are inputs.
 * ---------------------------------------------------------------------------------------- */

/* aug[n][3] = points[n][3] @ rot_and_scale (3x3 row-major, HOST), each element as fma(z, R2j, fma(y, R1j, x*R0j)) -- the
 * association of torch's CPU matmul for K = 3, so the truncation below sees the reference's fp32 values;
 * shift_max[0..2] = -min(aug) + offset (complete_shift, sparse_augmentation.py:95-99), shift_max[3..5] = max(aug).
 * No host synchronisation.  scratch: scn_vox_scratch_bytes(n). */
int64_t scn_vox_scratch_bytes(int64_t n);
int scn_vox_project(const float* points, int64_t n, const float* rot_and_scale_host, const float* offset_host,
                    float* aug, float* shift_max, void* scratch, scn_stream_t stream);

/* discrete[n][3] = trunc(aug + shift)  (`.long()`, :100);  table[i] = i if 0 <= discrete - test_start < size on every axis
 * else -1 (both HOST int32[3]; NULL: no cut-out, every row is kept).  fix_cut_out (:42-47) tests the UNMOVED coordinates
 * (test_start = 0); random_cut_out (:50-78) tests against its start positions.  Feed `table` (one segment) to
 * scn_rules_scan/_fill to obtain the kept rows in ascending order. */
int scn_vox_discretize(const float* aug, int64_t n, const float* shift, const int32_t* test_start_host,
                       const int32_t* size_host, int32_t* discrete, int32_t* table, scn_stream_t stream);

/* out int64 [m][4] = (discrete[rows[j]] - start, batch_index): the rows of one sample of collate_fn's coords_batch. */
int scn_vox_gather(const int32_t* discrete, const int32_t* rows, int64_t m, const int32_t* start_host,
                   int64_t batch_index, int64_t* out, scn_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Channel padding of PARAMETERS, many tensors per launch (scn_elem.hip).  No reference counterpart: the reference's mask
 * network has a 23-channel level (model.py:573-596: 16 U-Net channels ++ 7 raw ones, scannet_config/run.py:741-810); this
 * package runs it on slabs padded to 24 columns and hands its layers zero-padded copies of their logical weights
 * (module_factory.py:383-406 SubmanifoldConvolution, :256-258 Deconvolution, :366-367 NetworkInNetwork keep their logical
 * parameter shapes).  Tensor j is [fv][rows][cols]; desc_host[11 j ..] = fv, src_rows, src_cols, dst_rows, dst_cols, then two
 * row segments (src row, count, dst row): source rows [s, s + count) land on destination rows [d, d + count), every other
 * destination element is zero.  backward = 0: out_host[j] (padded shape) <- in_host[j] (logical shape);
 * backward = 1: out_host[j] (LOGICAL shape) <- the matching elements of in_host[j] (PADDED shape; NULL: zeros).
 * ---------------------------------------------------------------------------------------- */
int scn_pad_params_many(int n, const float* const* in_host, float* const* out_host, const int32_t* desc_host, int backward,
                        scn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * bf16-STORAGE forms of the HBM-bound feature kernels (BASELINE configs 3-5; scn_elem_bf16.hip): slabs are uint16 bf16 bit
 * patterns, values are widened exactly, the arithmetic is that of the fp32 forms above, one round-to-nearest-even per
 * result.  scn.OutputLayer / the ROI feature gather (custom_operations.py:7-10; roi_select_sparse.py:125-133),
 * scn.MaxPooling / AveragePooling (module_factory.py:315-354), scn.SparseToDense (:429-435), scn.AddTable (:308), and
 * the storage casts on either side of a bf16-stored stretch of a network.
 * ---------------------------------------------------------------------------------------- */
int scn_cast_f32_to_bf16(const float* X, int64_t count, uint16_t* Y, scn_stream_t stream);
int scn_cast_bf16_to_f32(const uint16_t* X, int64_t count, float* Y, scn_stream_t stream);
int scn_add_bf16(const uint16_t* A, const uint16_t* B, int64_t count, uint16_t* Y, scn_stream_t stream);
int scn_gather_rows_bf16(const uint16_t* X, const int32_t* rows, int64_t m, int c, uint16_t* Y, scn_stream_t stream);
int scn_segment_sum_bf16(const uint16_t* dY, const int32_t* item_row, int64_t n_items, int64_t n_rows, int c, uint16_t* dX,
                         double* acc64, scn_stream_t stream);
int scn_pool_fwd_bf16(const uint16_t* X, const int32_t* child, int64_t n_coarse, int c, int average, uint16_t* Y,
                      scn_stream_t stream);
int scn_pool_bwd_bf16(const uint16_t* X, const uint16_t* Y, const uint16_t* dY, const int32_t* parent, int64_t n_fine, int c,
                      int average, uint16_t* dX, scn_stream_t stream);
int scn_sparse_to_dense_fwd_bf16(const uint16_t* X, const int32_t* coords, int64_t n, int c, const int64_t* size3_host,
                                 uint16_t* out, scn_stream_t stream);
int scn_sparse_to_dense_bwd_bf16(const uint16_t* dOut, const int32_t* coords, int64_t n, int c, const int64_t* size3_host,
                                 uint16_t* dX, scn_stream_t stream);

/* Dilation gather (round 5; the first dense same-convolution behind scn.SparseToDense, module_factory.py:581-611 + :396-414: a
 * 3^3 conv3d with padding 1 on a volume whose only non-zero cells are the n active sites of a sparse level).  With
 * P[r][o][:] = X[r] . W[o] from ONE row GEMM over the active rows ([n][27][c]; o = (a*3+b)*3+c' over the kernel taps),
 *   fwd: out[cell][:] = bias + sum_o P[map[cell + (a-1, b-1, c'-1)]][o][:]   (ascending o; map: cell -> active row or -1, [B X Y Z])
 *   bwd: dP[r][o][:]  = dOut[cell(r) - (a-1, b-1, c'-1)][:] or 0 outside the volume   (cell_of_row: int64 [n])
 * out / dOut are the channels-last volume [B X Y Z][c]; bf16 != 0: P, out, dOut, dP are bf16 (fp32 accumulation).  bias may be
 * NULL.  The bias gradient is the column sum of dOut (scn_colsum*). */
int scn_dilate_gather_fwd(const void* P, const int32_t* map, int batch, const int64_t* size3_host, int c, int bf16,
                          const float* bias, void* out, scn_stream_t stream);
int scn_dilate_gather_bwd(const void* dOut, const int64_t* cell_of_row, int64_t n, const int64_t* size3_host, int c, int bf16,
                          void* dP, scn_stream_t stream);
/* The two index vectors of the above from a level's int32 coordinates (x, y, z, sample) [n][4]: cell_of_row[r] =
 * ((sample X + x) Y + y) Z + z and map[cell] = r, -1 on empty cells (map: int32 [batch X Y Z], need not be initialised).
 * *n_outside_dev = rows outside the volume (the caller checks it when it next waits for the device; such rows get cell 0
 * and no map entry). */
int scn_cell_map(const int32_t* coords, int64_t n, int batch, const int64_t* size3_host, int64_t* cell_of_row, int32_t* map,
                 int32_t* n_outside_dev, scn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Step executor (scn_exec.hip): ONE call walks the launch plan of a whole network pass.
 *
 * The reference drives the scn surface layer by layer from Python (module_factory.py builds nested scn.Sequential trees;
 * model.py:414-431 runs them); at ~330 launches per backbone step (~760 with the mask branch) the interpreter, not the
 * GPU, bounds the bf16 steps.  A plan is a flat list of ops over three tables the caller fills per step -- feature slabs
 * (`bufs`), parameters / weight images (`params`) and parameter gradients (`grads`) -- and a table of per-level index
 * structures.  Each op is exactly one of the entry points above with exactly the arguments the layer-by-layer path would
 * pass (same kernels, same bits); the executor adds no arithmetic of its own.
 * ---------------------------------------------------------------------------------------- */
enum {
    SCN_OP_GEMM_IDENT = 1,   /* scn_gemm_table(_bf16), identity table: SubM 1^3 / NetworkInNetwork / Linear; fwd or bwd-data  */
    SCN_OP_CONV_SUBM = 2,    /* scn_conv_tiles(_bf16) over the level's 3^3 tiles (fwd; bwd-data with SCN_F_W_TRANSPOSED|OFF_REVERSE) */
    SCN_OP_CONV_CHILD = 3,   /* scn_conv_tiles(_bf16) over the level's child tiles: Convolution fwd / Deconvolution bwd-data */
    SCN_OP_RULES_CHILD = 4,  /* scn_gemm_rules(_bf16) coarse -> fine: Deconvolution fwd / Convolution bwd-data              */
    SCN_OP_ROWS2 = 5,        /* scn_gemm_rows2: NiN over two joined parts (fwd), or its two-destination bwd-data             */
    SCN_OP_WGRAD_SUBM = 6,   /* scn_wgrad_bias_rules(_bf16) on the level's 3^3 rules                                          */
    SCN_OP_WGRAD2_SUBM = 7,  /* scn_wgrad_bias_rules2(_bf16): both convolutions of a residual unit                             */
    SCN_OP_WGRAD_DOWN = 8,   /* Convolution: scn_wgrad_rules(_bf16) on the child rules (X fine, dY coarse)                     */
    SCN_OP_WGRAD_UP = 9,     /* Deconvolution: scn_wgrad_bias_rules(_bf16), roles swapped (X coarse, dY fine), db over all 8  */
    SCN_OP_WGRAD_IDENT = 10, /* identity list: NiN / SubM 1^3 / Linear; aux = element offset into grads[w] (joined parts)     */
    SCN_OP_COLSUM = 11,      /* scn_colsum(_bf16): bias gradient alone                                                        */
    SCN_OP_ADD = 12,         /* y = x + x1 (scn_add / scn_add_bf16): the two gradient paths of a skip connection             */
    SCN_OP_CAST = 13         /* storage cast: SCN_XF_BF16 set = fp32 -> bf16, else bf16 -> fp32                               */
};
#define SCN_XF_BF16 (1 << 16)        /* op flag: the op's feature slabs are bf16-stored                                      */
#define SCN_XF_COARSE_ROWS (1 << 17) /* op flag (GEMM_IDENT / WGRAD_IDENT / COLSUM / ADD / CAST): rows = n_coarse of `level` */

typedef struct scn_exec_level {
    int64_t n;                       /* rows of this level */
    const int32_t* tstab;            /* SubM 3^3 tiles of this level (scn_tiles_build) */
    const uint32_t* tile_mask;
    const int32_t* perm;
    const int32_t* tile_order;
    const int32_t* in_rows;          /* SubM 3^3 compacted rules */
    const int32_t* out_rows;
    const int64_t* prefix_host;      /* int64[28] on the host */
    int64_t n_coarse;                /* rows of the next (coarser) level; 0: none */
    const int32_t* c_tstab;          /* child tiles [nt_coarse][8][16] */
    const uint32_t* c_tile_mask;
    const int32_t* c_perm;
    const int32_t* c_tile_order;
    const int32_t* c_in_rows;        /* strided rules: in = fine rows, out = coarse rows */
    const int32_t* c_out_rows;
    const int64_t* c_prefix_host;    /* int64[9] on the host */
    int64_t flags;                   /* SCN_XL_* */
} scn_exec_level;
#define SCN_XL_TILE_ORDER_X 1        /* tile_order continues with the XCD-local order (scn_tiles_build_x): bf16 SubM convolutions use it */

typedef struct scn_exec_op {
    int32_t op;                      /* SCN_OP_* */
    int32_t flags;                   /* SCN_F_* of the underlying call | SCN_XF_* */
    int32_t level;
    int32_t cin, cout;               /* of the CALL (backward-data calls swap the layer's) */
    int32_t x, y;                    /* buffer ids: source slab, destination slab (wgrad ops: X operand, dY operand) */
    int32_t r, m;                    /* residual operand, relu-mask operand; -1: none */
    int32_t x1, y1;                  /* ROWS2: second source / second destination; WGRAD2: second operand pair; ADD: x1 */
    int32_t c1;                      /* ROWS2: channels of the second source (fwd) / second destination (bwd) */
    int32_t w, b;                    /* params[] ids (weight or bf16 image, bias) -- wgrad / colsum ops: grads[] ids; -1: none */
    int32_t aux;                     /* WGRAD_IDENT: element offset into grads[w]; WGRAD_*: unused */
    int32_t reserved;
} scn_exec_op;

/* sizeof(scn_exec_op) (which = 0) / sizeof(scn_exec_level) (which = 1): lets a binding check its struct layout. */
int64_t scn_exec_struct_bytes(int which);
/* Scratch bytes and zeroed arrival counters (the scn_conv_tiles contract) the plan needs for these level sizes. */
int scn_exec_requirements(const scn_exec_op* ops, int n_ops, const scn_exec_level* levels, int n_levels,
                          int64_t* scratch_bytes, int64_t* arrival_counters);
/* Walk the plan on `stream`.  Stops at the first failing op (its status is returned, scn_last_error_string names the op). */
int scn_exec_run(const scn_exec_op* ops, int n_ops, const scn_exec_level* levels, int n_levels, void* const* bufs,
                 const void* const* params, void* const* grads, void* scratch, int64_t scratch_bytes, int32_t* arrival,
                 scn_stream_t stream);
/* The same with the parameter-gradient ops (SCN_OP_WGRAD_* / SCN_OP_COLSUM: leaves of a backward pass) on `side_stream`, each
 * behind an event of `stream`; `stream` is ordered behind the last of them before the call returns.  side_scratch: a
 * second scratch buffer of at least scratch_bytes.  side_stream == NULL: scn_exec_run. */
int scn_exec_run_streams(const scn_exec_op* ops, int n_ops, const scn_exec_level* levels, int n_levels, void* const* bufs,
                         const void* const* params, void* const* grads, void* scratch, int64_t scratch_bytes,
                         int32_t* arrival, scn_stream_t stream, scn_stream_t side_stream, void* side_scratch,
                         int64_t side_scratch_bytes);
/* Launch timing inside a pass (round 4; what bench.py's `roofline` samples): while enabled, every scn_exec_run* call brackets
 * its SCN_OP_CONV_SUBM / SCN_OP_CONV_CHILD ops -- one launch of the dominant tile kernel each -- with HIP timing events on
 * `stream`.  Process-wide (a node's backward pass runs on the autograd thread).  scn_exec_timing_collect waits for the
 * recorded events, writes per record the elapsed milliseconds and info[7] = (op, bf16 storage, cin, cout, rows in, rows out,
 * rules of the table), returns the number of records written (at most cap) and forgets THOSE (round 5: a caller whose buffer
 * is smaller than the backlog calls again for the rest; rounds 3-4 dropped what did not fit).
 * on = 2: EVERY op of a pass is bracketed (weight gradients, row GEMMs, casts: tools/exec_launch_table.py); the deferred unit
 * sums of a pass run behind its last op and belong to no record. */
int scn_exec_timing_enable(int on);
int64_t scn_exec_timing_collect(float* ms, int64_t* info, int64_t cap);

/* Round 6 (experiment, opt-in; nothing in the package calls it): the weight gradient of a SubM 3^3 layer with 32 input and 32
 * output channels in TILE-major form -- one row gather per rule: the dY rows of a mask-sorted tile are loaded once into LDS and
 * serve every offset of the tile, the offsets are dealt to the waves of a workgroup (profiles/r6_wgrad_one_gather.txt).  Serves
 * the same call sites as scn_wgrad_rules (the weight gradient of scn.SubmanifoldConvolution: module_factory.py:396-414).
 * X [n_in][32], dY [n_out][32] fp32; tstab / tile_mask / perm from scn_tiles_build over the layer's 27-offset table;
 * prefix_host[28] = the rule prefix (weights of the offset -> wave deal); dW [27][32][32]; scratch >= scn_wgrad_tiles32_scratch_bytes(). */
int64_t scn_wgrad_tiles32_scratch_bytes(void);
int scn_wgrad_tiles32(const float* X, int64_t n_in, const float* dY, int64_t n_out, const int32_t* tstab,
                      const uint32_t* tile_mask, const int32_t* perm, const int64_t* prefix_host, float* dW, void* scratch,
                      int relu_in, scn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SCN_MI355X_H */
