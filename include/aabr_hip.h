/*
 * aabr_hip.h -- C ABI of libaabr_hip.so, the MI355X (gfx950) implementation of the
 * sparse-3D detection hot path of xuyongzhi/Automatic-As-built-Reconstruction:
 * voxel scatter -> hash grid -> rule tables -> sparse conv (MFMA) -> BN -> rotated IoU/NMS.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless its name ends in `_host`;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *    synchronises unless stated;
 *  - every entry point returns 0 on success or a negative AABR_E* code;
 *    aabr_last_error() returns a static message for the calling thread;
 *  - row-major, contiguous arrays; features are float32 [rows, planes];
 *    site coordinates are int32 [rows,4] = (x, y, z, batch); rule tables are
 *    int32 [filter_volume][rows] "gather tables" (entry = partner row, or -1).
 *
 * The reference binds this path through two pybind11 extension modules
 * (SparseConvNet/sparseconvnet/SCN/pybind.cpp:11-235 `sparseconvnet.SCN`,
 *  maskrcnn_benchmark/csrc/vision.cpp:9-20 `_C`) whose arguments are at::Tensor /
 * Metadata<3>&.  Each function below cites the reference interface it replaces;
 * INTEGRATION.md shows the ctypes binding that reproduces the reference names.
 */
#ifndef AABR_HIP_H
#define AABR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AABR_OK 0
#define AABR_EINVAL (-1)  /* bad argument (null pointer, negative size, unsupported shape) */
#define AABR_ELAUNCH (-2) /* HIP launch / runtime error */
#define AABR_ERANGE (-3)  /* coordinate outside the supported [0, 65534] range */

const char *aabr_last_error(void);
/* ABI version: bumped whenever a signature, a record layout or AABR_META_WORDS changes; a binding written for one value
 * must refuse a library that reports another (`_hip.load()` does).  500 = round 5 (16-word meta blocks, brick grids);
 * 600 = round 6 (regression targets out of the label kernel, list encode / decode, fused small-map records). */
#define AABR_ABI_VERSION 600
int aabr_version(void);
/* Tuning knobs for experiments and tests (no counterpart in the reference; the defaults are what ships): CONV_WIDE,
 * CONV_WIDE_BF16 (0 = never / 1 = whenever supported), WIDE_ROWS, WIDE_NBUF, CONV_WLDS,
 * CONV_SMALL, CONV_NBW, CONV_WPB, VOXEL_MEAN, GEOM_JOBS (0: aabr_geom_run issues its stream builders book by book).  A knob takes its value from the environment variable AABR_<NAME>,
 * read ONCE at its first use in the process; aabr_set_knob overrides it (unset != 0: back to "no value").  No entry
 * point reads the environment on its launch path.                                                              */
int aabr_set_knob(const char *name, int value, int unset);
/* bit 0: a `make DEV=1` build -- it additionally carries the phase-clock variants of k_conv_cs that tools/tools_cs_phases.py
 * reads (flags >> 8 debug bits); a release build answers those flags with an error.  The A/B kernels of rounds 3-4 that
 * were measured slower (row-stationary bf16, three-term fp32 split, four-workgroup ring, deferred accumulate, eight-wave
 * workgroups) were removed in round 5, the register-accumulator kernel over the gather table (k_conv_rb, 0.73-0.91x) in
 * round 6; their tables under profiles/ are the record.                                  */
int aabr_build_flags(void);
/* number of int32 words of the `meta` block written by the geometry builders */
#define AABR_META_WORDS 16
/* meta[0] = number of active sites, meta[1] = max points per site (input layer only; brick levels: number of bricks),
 * meta[2] = error flag (non-zero: coordinate out of range), meta[3] = total of 2nd scan lane,
 * meta[4], meta[5] = internal (chunk ticket, redo flag of the packed input-layer forms),
 * meta[8..11] (input layer) = largest x, y, z and batch index over the valid points, -1 when there is none: the extent
 * a brick grid's directory is sized by */

/* ---- input pipeline ---------------------------------------------------------------------------
 * Device-side form of the dataset's host quantisation (data3d/suncg_utils/suncg_dataset.py:126-188,
 * test-time path: no augmentation): a = xyz*scale - min(xyz)*scale; keep 0 <= a < full_scale;
 * locs = trunc(a) (int64 [n,4], 4th column = batch_index); feats_xyz (optional, float32 rows of
 * stride feat_stride) receives a/scale.  xyz / xyz_min are float32 or float64 device arrays
 * ([n,3] / [3]); full_scale_dev int32[3] on the device.  Points outside the range are not
 * compacted away: they get the coordinate sentinel (-1,-1,-1), which aabr_input_layer_sites skips
 * (the reference drops them on the host before the input layer, suncg_dataset.py:183-185).      */
int aabr_quantize_points(const void *xyz, int is_double, int64_t n, double scale, const void *xyz_min,
                         const int32_t *full_scale_dev, int64_t batch_index, int64_t *locs,
                         float *feats_xyz, int feat_stride, void *stream);

/* ---- hash grid ---------------------------------------------------------------------------
 * A grid is an open-addressing table of cap 16-byte entries {uint64 key (packed b,x,y,z), uint32 first, int32 val
 * (row of the site)}, handed over through the `keys` parameters (16-byte aligned).  cap must be a power of two
 * >= 2 * (number of inserted keys).  Probing is linear inside the key's home block of 4096 slots (csrc/common.h
 * grid_next): a probe chain stays inside one 64 KiB window and the LDS-binned voxel scatter can build a block per
 * workgroup.
 * Replaces SparseGrid / SparseGridMap (SCN/Metadata/Metadata.h:24-33).                       */

/* Voxel scatter, geometry half -- replaces Metadata<3>::inputLayer -> inputLayerRules
 * (SCN/Metadata/Metadata.cpp:405-417, IOLayersRules.h:18-125), modes 1..4.  One fill + two kernels
 * (csrc/voxel_scatter.hip): hash insert + first-point atomicMin; then ONE pass over the points that numbers the
 * voxels in first-seen order (chunk scan + decoupled look-back) and links every further point of a voxel into
 * the site's chain.
 *   coords       int64 [n, ncols] (ncols 3 or 4; 4th column = batch index)   (API layout)
 *   keys         the hash grid: cap entries of 16 bytes {uint64 key, uint32 first, int32 val}, 16-byte aligned,
 *                contents overwritten (interleaved so that the insert's CAS, its first-seen minimum and every
 *                later probe of a slot touch one 64-byte sector); followed directly by `meta` the two are cleared
 *                with a single fill
 *   slot         int32 [n]   scratch: hash slot of every point
 *   point_site   int32 [n]   out: output row of every input row (-1: dropped)
 *   site_coords  int32 [n,4] out: first V rows valid, first-seen order
 *   first_pt     int32 [n]   out: first (lowest-index) point of each site
 *   cnt_extra    int32 [n]   out: points per site minus one
 *   head, nxt    int32 [n]   out: chain of the site's further points (head[site] -> nxt[point] -> ... -> -1)
 *   status       int32 [aabr_input_layer_status_words(n)] scratch (8-byte aligned)
 *   meta         int32 [AABR_META_WORDS] out (device): V, (maxActive: written by aabr_input_layer_forward), error */
int64_t aabr_input_layer_status_words(int64_t n);
int aabr_input_layer_sites(const int64_t *coords, int64_t n, int ncols, uint64_t *keys, int64_t cap,
                           int32_t *slot, int32_t *point_site, int32_t *site_coords, int32_t *first_pt,
                           int32_t *cnt_extra, int32_t *head, int32_t *nxt, int32_t *status, int32_t *meta,
                           void *stream);
/* The same result (site list, first-seen numbering, chains, finished hash grid) by the two forms BASELINE.json's
 * north_star asks about, for inputs whose (batch, x, y, z, point index) fit ONE 64-bit word -- field widths from the
 * layer's spatial size (IOLayersRules.h:18-125 receives it as `spatialSize`) and n; aabr_input_layer_pack_bits returns
 * the bits left for the batch index (0: does not fit, use aabr_input_layer_sites):
 *   variant 1  one device atomic per point (CAS of the word into an 8-byte side table, atomicMin only for the
 *              smaller of two points of one voxel) instead of two;
 *   variant 2  LDS-staged hash binning with coalesced HBM writes and no device atomics on the table: the words are
 *              partitioned by hash block (4096 slots), every block's table is built in LDS by one workgroup and its
 *              4096 finished 16-byte entries are written with coalesced stores (no fill of the grid).
 * words: cap 8-byte words of scratch (side table / record regions); cursor: cap/4096 int32 (variant 2; 4096 <= cap
 * <= 2^24).  When a point does not fit the word (coordinate >= spatial size, batch index beyond the bits left) or a
 * block overflows its record region, meta[5] comes back 0 (else -1) and the caller must run aabr_input_layer_sites
 * on the same buffers instead; the other outputs are then unspecified.  n > 0.                                */
int aabr_input_layer_pack_bits(int64_t n, const int32_t *spatial_host);
int aabr_input_layer_sites_packed(const int64_t *coords, int64_t n, int ncols, const int32_t *spatial_host,
                                  int variant, uint64_t *keys, int64_t cap, uint64_t *words, int32_t *cursor,
                                  int32_t *slot, int32_t *point_site, int32_t *site_coords, int32_t *first_pt,
                                  int32_t *cnt_extra, int32_t *head, int32_t *nxt, int32_t *status, int32_t *meta,
                                  void *stream);

/* Voxel scatter, feature half -- replaces InputLayer_ForwardPass / InputLayer_fp_
 * (SCN/CPU/IOLayers.cpp:11-29, SCN/CUDA/IOLayers.cu:14-41).  mode: 1 keep-first-listed,
 * 2 keep-last-listed, 3 sum, 4 mean (exactly the reference's mode table, ioLayers.py:33-38).
 * Also writes last_pt [V] (highest-index point of each site; may be NULL) and meta[1] = maxActive. */
int aabr_input_layer_forward(const float *in_feats, float *out_feats, int64_t V, int planes,
                             const int32_t *first_pt, const int32_t *cnt_extra, const int32_t *head,
                             const int32_t *nxt, int32_t *last_pt, int mode, int32_t *meta, void *stream);
/* replaces InputLayer_BackwardPass / InputLayer_bp_ (CPU/IOLayers.cpp:30-47, IOLayers.cu:43-70) */
int aabr_input_layer_backward(float *d_in_feats, const float *d_out_feats, int64_t n, int planes,
                              const int32_t *point_site, const int32_t *first_pt, const int32_t *last_pt,
                              const int32_t *cnt_extra, int mode, void *stream);

/* Reference-format input rule table rules[1] (IOLayersRules.h:112-124): int32 [V, 1+maxActive]. */
int aabr_input_layer_rule_table(const int32_t *first_pt, const int32_t *last_pt, const int32_t *cnt_extra,
                                const int32_t *head, const int32_t *nxt, int64_t V, int max_active, int mode,
                                int32_t *rules, void *stream);

/* Submanifold rule table -- replaces Metadata<3>::getSubmanifoldRuleBook ->
 * SubmanifoldConvolution_SgToRules (Metadata.cpp:429-443, SubmanifoldConvolutionRules.h:11-45).
 * table[k*V + v] = row of the site at coords[v] + offset_k (offset enumeration of
 * RectangularRegion::offset, RectangularRegions.h:30-38: z fastest), or -1.
 * counts (int32 [vol * ceil(V/256)], optional) receives per-offset, per-256-row-block rule
 * counts (sum over the second index = rules at that offset).                                 */
int aabr_submanifold_table(const int32_t *site_coords, int64_t V, const uint64_t *keys,
                           int64_t cap, const int32_t *filter_size_host,
                           int32_t *table, int32_t *counts, void *stream);

/* Per-sample row offsets of a batch-contiguous site list -- replaces SparseGrid::ctr (Metadata.h:24-33; read by
 * Metadata::getSpatialLocations, Metadata.cpp:147-168, and by the anchor generator's per-example index scopes,
 * anchor_generator_sparse3d.py:137-147).  out int32 [max_samples + 2]: out[0] = V (read from meta[0] on the device,
 * clipped to V_max), out[1 + b] = first row with batch index >= b for b = 0 .. max_samples.                    */
int aabr_sample_offsets(const int32_t *site_coords, const int32_t *meta, int64_t V_max, int max_samples,
                        int32_t *out, void *stream);

/* ---- brick grids (extension; csrc/brick.hip, csrc/geom.h) -------------------------------------------------------------
 * A level of the scene stored by WHERE its sites are instead of in a hash table -- what replaces the reference's
 * per-sample google::dense_hash_map (Metadata.h:24-34) when Metadata_3 is created with site_order="brick":
 *   dir    [nb * sbx * sby * sbz] 16-byte entries {uint64 word, uint32 prefix, pad}: one entry per super-brick (16^3 voxels)
 *          of the level's extent, one bit per brick (4^3 voxels), prefix = occupied bricks in front of the word;
 *   bricks [NB] 16-byte entries {uint64 cell mask, int32 base, pad} in directory order, base = sites in front of the brick;
 *   row of the site at (b, x, y, z) = base + popcount(mask below cell (x&3)<<4 | (y&3)<<2 | (z&3)).
 * dims_host = {sbx, sby, sbz, nb}.  Sites are numbered brick by brick, cell by cell: spatial neighbours are neighbours
 * in memory, a lookup is two dependent 16-byte loads (no hashing), and the row order inside a sample is a permutation
 * of the reference's first-seen order (parity modulo a per-sample permutation).
 *
 * aabr_brick_build: the level that holds, for every input site u < min(vin_bound, *vin_count_dev) (vin_count_dev may
 * be NULL) and every cell of its output region under (size, stride, out_spatial) (OutputRegionCalculator,
 * RectangularRegions.h:109-119; size = stride = 1: the level of the input sites themselves; size == stride up to 65536:
 * the composition of non-overlapping levels as in aabr_convolution_sites), one site -- what
 * Convolution_InputSgToRulesAndOutputSg creates (ConvolutionRules.h:11-34).  Writes dir, bricks, bcoord [NB] int32x4
 * brick coordinates, out_coords [V] int32x4 (x, y, z, batch) in row order and meta (meta[0] = V, meta[1] = NB, meta[2] != 0:
 * a site outside the extent `dims` covers, or more than nb_cap bricks / v_cap sites).  Nothing is read back: a chain of
 * levels is enqueued back to back with device-side counts.  scratch: aabr_brick_scratch_words(dir words, nb_cap) int32.
 * flags bit 0: the caller has zeroed dir, bricks, meta and scratch (a pyramid's levels share ONE fill).  A one-workgroup
 * form for the coarse levels (every phase in a single launch) was built and measured slower than the eight small launches it
 * replaced (returning atomics in a serial loop: 930 vs 615 us for the 12 strided grids of the bench batch) and removed.  */
int64_t aabr_brick_scratch_words(int64_t dir_words, int64_t nb_cap);
int aabr_brick_build(const int32_t *in_coords, int64_t vin_bound, const int32_t *vin_count_dev, const int32_t *size_host,
                     const int32_t *stride_host, const int32_t *out_spatial_host, const int32_t *dims_host, void *dir,
                     void *bricks, int64_t nb_cap, int32_t *bcoord, int32_t *out_coords, int64_t v_cap, int32_t *meta,
                     int32_t *scratch, int flags, void *stream);
/* Input level: the voxel scatter numbers its sites in first-seen order (IOLayersRules.h:86-91); old_coords [V] are those
 * sites, (dir, bricks) the brick level built from them.  new_of_old[r] / old_of_new[i] = the permutation; first_pt /
 * cnt_extra / head re-indexed by the new rows; point_site2[p] = new row of point p's site.                           */
int aabr_brick_renumber(const int32_t *old_coords, int64_t V, const int32_t *dims_host, const void *dir, const void *bricks,
                        int32_t *new_of_old, int32_t *old_of_new, const int32_t *first_pt, const int32_t *cnt_extra,
                        const int32_t *head, int32_t *first_pt2, int32_t *cnt_extra2, int32_t *head2,
                        const int32_t *point_site, int64_t n, int32_t *point_site2, int32_t *meta, void *stream);
/* Voxel scatter straight into a brick grid -- InputLayer (Metadata::inputLayer -> inputLayerRules, Metadata.cpp:405-417,
 * IOLayersRules.h:18-125) without a hash table and without first-seen numbers: the POINTS are the items of the input
 * level's aabr_brick_build (size = stride = 1), a voxel's row follows from where it lies.
 *   aabr_points_prepare: coords int64 [n, ncols] -> pc int32 [n,4] (x, y, z, batch; x = -1: a point the layer skips),
 *     meta (AABR_META_WORDS, starts at all ones): meta[2] == 0: a coordinate outside [0, 65534]; meta[8..11] = largest
 *     x, y, z, batch index of the valid points (-1: none) -- the extent the directory is sized by.  first_pt / cnt_extra /
 *     head (each n words; all three or none, may be NULL): set to their starting values (-1, 0, -1) in the same pass, so
 *     aabr_points_sites (flags bit 0) need not fill them.
 *   aabr_points_sites (after aabr_brick_build(pc, n, ..)): point_site[i] = row of point i's voxel (-1: skipped),
 *     first_pt[row] = lowest point index of the voxel, cnt_extra[row] = its further points, head / nxt = their chain --
 *     the arrays aabr_input_layer_forward / _backward / _rule_table read; sized n (rows < V are meaningful).
 *     flags bit 0: first_pt / cnt_extra / head already hold their starting values (aabr_points_prepare wrote them).   */
int aabr_points_prepare(const int64_t *coords, int64_t n, int ncols, int32_t *pc, int32_t *meta, int32_t *first_pt,
                        int32_t *cnt_extra, int32_t *head, void *stream);
int aabr_points_sites(const int32_t *pc, int64_t n, const int32_t *dims_host, const void *dir, const void *bricks,
                      int32_t *point_site, int32_t *first_pt, int32_t *cnt_extra, int32_t *head, int32_t *nxt,
                      int32_t *meta, int flags, void *stream);
/* aabr_submanifold_table / aabr_convolution_tables2 over brick levels: same tables, same block counts
 * (Metadata.cpp:429-443,484-510; SubmanifoldConvolutionRules.h:26-45; ConvolutionRules.h:11-34).                       */
int aabr_brick_submanifold_table(const int32_t *site_coords, int64_t V, const int32_t *dims_host, const void *dir,
                                 const void *bricks, const int32_t *filter_size_host, int32_t *table, int32_t *counts,
                                 void *stream);
int aabr_brick_convolution_tables(const int32_t *in_coords, int64_t V_in, const int32_t *in_dims_host, const void *in_dir,
                                  const void *in_bricks, const int32_t *out_coords, int64_t V_out,
                                  const int32_t *out_dims_host, const void *out_dir, const void *out_bricks,
                                  const int32_t *size_host, const int32_t *stride_host, const int32_t *out_spatial_host,
                                  int32_t *table_out, int32_t *table_in, int32_t *counts, int32_t *counts_in, void *stream);

/* Strided convolution geometry -- replaces Metadata<3>::getRuleBook ->
 * Convolution_InputSgToRulesAndOutputSg (Metadata.cpp:484-510, ConvolutionRules.h:11-34,
 * RectangularRegions.h:95-119).  Creates the output grid (sites numbered in first-seen order
 * over input rows ascending, then output-region order) and reports V_out in meta[0].
 *   out_keys: out_cap 16-byte grid entries as in aabr_input_layer_sites, out_cap a power of two >= 1.5 E;
 *   scratch int32 [E + 4*ceil(E/256) + 16], E = V_in * max_out_per_in,
 *   max_out_per_in = prod(ceil(size/stride)); out_site_coords int32 [E,4].
 * size, stride <= 64 per axis; for size == stride (one output site per input site) up to 65536, so that a chain of
 * non-overlapping levels can be built from its FIRST grid in one step each (size = stride = product of the chain's
 * strides): the site set and its first-seen order are those of the level-by-level construction.          */
int aabr_convolution_sites(const int32_t *in_coords, int64_t V_in, const int32_t *size_host,
                           const int32_t *stride_host, const int32_t *out_spatial_host,
                           uint64_t *out_keys, int64_t out_cap,
                           int32_t *scratch, int32_t *out_site_coords, int32_t *meta,
                           void *stream);
/* counts (optional): int32 [vol * ceil(V_out/256)] per-block rule counts, as above.
 * table_out[k*V_out + o] = input row at offset k of output o's window (or -1);
 * table_in [k*V_in  + u] = output row whose window holds input u at offset k (or -1).       */
int aabr_convolution_tables(const int32_t *in_coords, int64_t V_in, const uint64_t *in_keys,
                            int64_t in_cap,
                            const int32_t *out_coords, int64_t V_out, const uint64_t *out_keys,
                            int64_t out_cap, const int32_t *size_host,
                            const int32_t *stride_host, const int32_t *out_spatial_host,
                            int32_t *table_out, int32_t *table_in, int32_t *counts,
                            void *stream);
/* the same with the per-block rule counts of table_in as well (counts_in int32 [vol * ceil(V_in/256)], may be
 * NULL): what the weight-gradient pass of a transposed convolution sizes its chunks with.               */
int aabr_convolution_tables2(const int32_t *in_coords, int64_t V_in, const uint64_t *in_keys,
                            int64_t in_cap,
                            const int32_t *out_coords, int64_t V_out, const uint64_t *out_keys,
                            int64_t out_cap, const int32_t *size_host,
                            const int32_t *stride_host, const int32_t *out_spatial_host,
                            int32_t *table_out, int32_t *table_in, int32_t *counts,
                             int32_t *counts_in, void *stream);

/* Reference-format rule book from a gather table: for offset k the (in,out) pairs in
 * ascending `out` order -- the layout of RuleBook = vector<vector<Int>> (Metadata.h:34).
 * rules int32 [vol][V][2] (first counts[k] pairs of each slab valid); in_col selects which
 * column receives the table entry (0: table entry = input row; 1: swapped / deconvolution).  */
int aabr_table_to_rulebook(const int32_t *table, int64_t V, int vol, int32_t *rules,
                           int32_t *counts, void *stream);

/* Metadata<3>::getSpatialLocations (Metadata.cpp:147-168): int32 [V,4] -> int64 [V,4]. */
int aabr_spatial_locations(const int32_t *site_coords, int64_t V, int64_t *locations,
                           void *stream);

/* ---- compiled rule books -------------------------------------------------------------------
 * A gather table is compiled once per rule book (and reused by every layer at that scale, forward
 * and backward) into the two streaming layouts the MFMA kernels read:
 *  (1) tile blocks: for each tile of 64 output rows, blocks of 16 (partner row, local row) pairs
 *      that share one filter offset -- input of aabr_conv_forward;
 *  (2) offset pairs: for each offset the (partner row, row) pairs in ascending row order -- the
 *      reference's RuleBook layout (Metadata.h:34) -- input of aabr_conv_backward_weight.
 * block_counts is the `counts` output of the table builder.  Limits: V < 2^25.                   */
int64_t aabr_tile_blocks_words(int64_t V, int vol);   /* int32 words of `blocks` */
int aabr_build_tile_blocks(const int32_t *table, int64_t V, int vol, int32_t *blocks, void *stream);
int64_t aabr_offset_pairs_words(int64_t V, int vol);  /* int32 words of `pairs` */
int aabr_build_offset_pairs(const int32_t *table, const int32_t *block_counts, int64_t V, int vol,
                            int32_t *pairs, void *stream);

/* Wide-layer forms of the same contraction (csrc/conv_wide.hip): 128-row output tiles whose blocks
 * share one set of weights per filter offset (registers / LDS) instead of streaming 32 KiB of packed weights per
 * 16-pair block.  Replaces the same reference loops as aabr_conv_forward (SCN/CPU/Convolution.cpp:45-185,
 * SCN/CPU/Deconvolution.cpp:7-77; the CUDA twin it stands in for is dConvolution_KMxKN_forwardA/B,
 * SCN/CUDA/Convolution.cu:57-203).
 *   aabr_conv_wide_tile_rows: 0 = use aabr_conv_forward; 128 = rows per tile of the block stream that
 *     aabr_conv_forward_wide wants for this shape (supported AND expected to beat the 64-row-tile kernels);
 *   aabr_wide_blocks_words / aabr_build_wide_blocks: gather table [vol][V] -> per-tile per-offset blocks of 16
 *     (partner row, local row) pairs (vol <= 63);
 *   aabr_conv_pack_weights: W [vol][nIn][nOut] (transpose: W[k]^T) -> packed MFMA A-operand layout;
 *   aabr_conv_forward_wide: n_in % 32 == 0, n_out % 64 == 0; `wpack` must already hold the packed weights of
 *     this orientation; flags bit1: mirrored offsets (submanifold input-gradient through the forward table). */
int aabr_conv_wide_tile_rows(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol);
int64_t aabr_wide_blocks_words(int64_t V, int vol, int tile_rows);
int aabr_build_wide_blocks(const int32_t *table, int64_t V, int vol, int tile_rows, int32_t *blocks, void *stream);
int aabr_conv_pack_weights(const float *W, int vol, int n_in, int n_out, int transpose, float *wpack, void *stream);
int aabr_conv_forward_wide(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                           int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                           int flags, const float *wpack, void *stream);
/* the same with out = conv + residual ([V_out, n_out] fp32, may be NULL): the residual / lateral add that follows
 * the convolution in the reference's graph (AddTable, tables.py:27-41; add_feature_planes, utils.py:38-44) folded
 * into the write-out; a + b is commutative bit for bit, so the result equals the separate add's.           */
int aabr_conv_forward_wide_res(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                               int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                               int flags, const float *wpack, const float *residual, void *stream);
/* the same, and the write-out also forms the TRAINING STATISTICS of the BatchNormalization that reads this output
 * (replaces the statistics loop of SCN/CPU/BatchNormalization.cpp:20-32 / the first reduction of
 * SCN/CUDA/BatchNormalization.cu:14-72 -- one read pass over the feature matrix less): per output tile one
 * [2][n_out] pair of fp64 column sums (sum x, sum x^2) of exactly the values stored (after bias / residual, after the
 * bf16 rounding), at stats[tile * 2 * n_out ...]; aabr_conv_wide_stats_doubles() doubles, tile_rows >= 64.  Hand
 * them to aabr_bn_forward_parts with nparts = ceil(V_out / tile_rows): it combines them in tile order, so the
 * result is reproducible bit for bit.  stats == NULL: plain aabr_conv_forward_wide_res.                        */
int64_t aabr_conv_wide_stats_doubles(int64_t V_out, int tile_rows, int n_out);
int aabr_conv_forward_wide_stats(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                                 int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                 int flags, const float *wpack, const float *residual, double *stats, void *stream);
int aabr_conv_forward_wide_bf16_stats(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                      int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                      const float *bias, int flags, const uint16_t *wpack, double *stats, void *stream);
/* Input-gradient form of the same launch (flags = transposed | mirrored, as aabr_conv_forward_wide): its output is the
 * d_out of the BatchNormalization(+leaky ReLU) whose result the convolution consumed; the write-out forms THAT
 * BatchNorm's backward statistics (replaces the first loop of BatchNormalization_BackwardPass,
 * SCN/CPU/BatchNormalization.cpp:66-84): per tile [2][n_out] fp64 sums of d and (x - mean) * d, d = d_out masked by the
 * sign of the forward activation recomputed from the BatchNorm input `bn_in` (leakiness >= 0).  fp32 storage.      */
int aabr_conv_forward_wide_bwd_stats(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                                     int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                     int flags, const float *wpack, const float *residual, double *stats,
                                     const float *bn_in, const float *save_mean, const float *save_invstd,
                                     const float *bn_weight, const float *bn_bias, float leakiness, void *stream);
/* ... and the BatchNorm backward that takes them (nparts = ceil(rows / tile_rows)); everything else as
 * aabr_bn_backward_add (d_in_add may be NULL).                                                                   */
int aabr_bn_backward_parts(const float *in, float *d_in, const float *out, const float *d_out, int64_t rows, int planes,
                           const float *save_mean, const float *save_invstd, const float *weight, const float *bias,
                           float *d_weight, float *d_bias, float leakiness, const double *parts, int nparts,
                           float *scratch, const float *d_in_add, void *stream);
/* bf16 storage of the two: the sign comes from the BatchNorm's STORED output `bn_out` (as aabr_bn_backward_bf16 reads
 * it), x and d are the stored (rounded) values, sums in fp64 -- the same terms k_bn_partials forms.               */
int aabr_conv_forward_wide_bf16_bwd_stats(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                          int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                          const float *bias, int flags, const uint16_t *wpack, double *stats,
                                          const uint16_t *bn_in, const uint16_t *bn_out, const float *save_mean,
                                          float leakiness, void *stream);
/* The general bf16-storage launch of the compiled pass: aabr_conv_forward_wide_bf16 with an optional `residual` (bf16
 * [V_out, n_out]: out = bf16(bf16(conv + bias) + residual) -- the residual / lateral add, or the gradient sum of a tensor
 * with a second consumer, folded into the write-out, bit for bit what the separate bf16 add stores), optional `stats`
 * of the stored values (as aabr_conv_forward_wide_bf16_stats; bn_in NULL) or, with bn_in / bn_out / save_mean /
 * leakiness, the backward statistics of aabr_conv_forward_wide_bf16_bwd_stats.                                        */
int aabr_conv_forward_wide_bf16_res(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats, int n_out,
                                    int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                    int flags, const uint16_t *wpack, const uint16_t *residual, double *stats,
                                    const uint16_t *bn_in, const uint16_t *bn_out, const float *save_mean, float leakiness,
                                    void *stream);
int aabr_bn_backward_parts_bf16(const uint16_t *in, uint16_t *d_in, const uint16_t *out, const uint16_t *d_out,
                                int64_t rows, int planes, const float *save_mean, const float *save_invstd,
                                const float *weight, const float *bias, float *d_weight, float *d_bias, float leakiness,
                                const double *parts, int nparts, float *scratch, void *stream);
/* BatchNormalization forward in training mode with the statistics' partial sums given (layout above; any producer
 * that writes [nparts][2][planes] fp64 sums of x and x^2 will do): everything else as aabr_bn_forward[_bf16].   */
int aabr_bn_forward_parts(const float *in, float *out, int64_t rows, int planes, float *save_mean,
                          float *save_invstd, float *running_mean, float *running_var, const float *weight,
                          const float *bias, float eps, float momentum, float leakiness, const double *parts,
                          int nparts, float *scratch, void *stream);
int aabr_bn_forward_parts_bf16(const uint16_t *in, uint16_t *out, int64_t rows, int planes, float *save_mean,
                               float *save_invstd, float *running_mean, float *running_var, const float *weight,
                               const float *bias, float eps, float momentum, float leakiness, const double *parts,
                               int nparts, float *scratch, void *stream);

/* Name of the kernel instance (template arguments included) the last aabr_conv_forward[_bf16] /
 * aabr_conv_backward_weight[_bf16] call on this thread dispatched -- measurement provenance only.   */
const char *aabr_conv_last_variant(void);

/* ---- sparse convolution (fp32 features, fp32 MFMA) ----------------------------------------
 * out[o] = bias + sum_k in[table[k][o]] @ W[wk(k)]      (rows with table == -1 contribute 0)
 * Replaces {Submanifold,}Convolution_updateOutput / Deconvolution_updateOutput
 * (SCN/sparseconvnet_cuda.cpp:281-310; CPU/Convolution.cpp:45-79,117-149;
 *  CPU/Deconvolution.cpp:7-41; CUDA/Convolution.cu:57-233,444-521,618-642).
 *   rows_in  number of rows of `in_feats` (bounds the gather; V_out for submanifold layers)
 *   blocks   tile blocks compiled from the gather table whose entries index `in_feats`
 *   W        float32 [vol, nIn, nOut]  (the reference's [vol, groups=1, nIn, nOut])
 *   flags    bit0: use W[k]^T (input-gradient pass: `in` has nOut planes, `out` nIn planes,
 *            CPU/Convolution.cpp:108-112); bit1: weight index vol-1-k (submanifold
 *            input-gradient through the forward table); bit2: `wpack` already holds the packed
 *            weights for this (W, bit0) pair (skips the repack launch); bits 8-9: timing
 *            experiments only (skip MFMAs / gathers).
 *   wpack    float32 scratch, aabr_conv_wpack_floats(vol,nIn,nOut) elements                   */
int64_t aabr_conv_wpack_floats(int vol, int n_in, int n_out);
int aabr_conv_forward(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                      int64_t V_out, const int32_t *blocks, int vol, const float *W,
                      const float *bias, int flags, float *wpack, void *stream);
/* Training steps need W packed in both orientations once per weight version: this fills the forward
 * layout (what aabr_conv_forward builds for flags bit0 = 0) and the input-gradient layout (bit0 = 1)
 * in one launch; the conv calls then pass flags bit2.  Each pack holds
 * aabr_conv_wpack_floats(vol,nIn,nOut) elements (float32 / bf16 bit patterns).                   */
int aabr_conv_pack_weights2(const float *W, int vol, int n_in, int n_out, float *wpack_fwd,
                            float *wpack_t, void *stream);
int aabr_conv_pack_weights2_bf16(const float *W, int vol, int n_in, int n_out, uint16_t *wpack_fwd,
                                 uint16_t *wpack_t, void *stream);
/* All convolutions of a network in one launch.  jobs_dev: device array of n_jobs records of 48 bytes
 *   { const float *W; void *wpack_fwd; void *wpack_t; int32 vol, n_in, n_out, bf16; int64 first_block; }
 * first_block[j] = sum of aabr_conv_pack_job_blocks(...) of the jobs before j; total_blocks = that sum over
 * all jobs.  Same layouts as aabr_conv_pack_weights2[_bf16] (bf16 != 0: bf16 bit patterns).  The reference
 * has no counterpart: its GEMMs read W[k] in place (CPU/Convolution.cpp:60-66).                       */
int64_t aabr_conv_pack_job_blocks(int vol, int n_in, int n_out);
int aabr_conv_pack_weights_jobs(const void *jobs_dev, int n_jobs, int64_t total_blocks, void *stream);
/* dW[k] = sum over offset k's pairs (t, o) of in[t]^T (x) d_out[o]; d_bias (optional) = column
 * sums of d_out.  max_chunks bounds the number of chunks of c = aabr_conv_dw_chunk_pairs(V, vol, nIn, nOut) pairs:
 * sum_k ceil(R_k/c) when the rule counts are known on the host, else ceil(vol*V/c) + vol; scratch float32
 * [aabr_conv_dw_scratch_floats(max_chunks, nIn, nOut)].  Deterministic (no atomics).
 * Replaces the dW half of *_backward (CPU/Convolution.cpp:81-115,151-185;
 * CUDA/Convolution.cu:249-441,526-667).                                                     */
int64_t aabr_conv_dw_scratch_floats(int64_t max_chunks, int n_in, int n_out);
int aabr_conv_dw_chunk_pairs(int64_t V_out, int vol, int n_in, int n_out); /* 256 or 1024 */
int aabr_conv_backward_weight(const float *in_feats, int n_in, const float *d_out, int n_out,
                              int64_t V_out, const int32_t *pairs, int vol, int64_t max_chunks,
                              float *dW, float *d_bias, float *scratch, void *stream);

/* ---- batch normalisation + leaky ReLU ------------------------------------------------------
 * Replaces BatchNormalization_updateOutput / _backward (SCN/CPU/BatchNormalization.cpp:12-157,
 * SCN/CUDA/BatchNormalization.cu:14-238).  weight/bias may be NULL.  scratch: float32
 * [aabr_bn_scratch_floats(planes)].                                                         */
int64_t aabr_bn_scratch_floats(int planes);
int aabr_bn_forward(const float *in, float *out, int64_t rows, int planes, float *save_mean,
                    float *save_invstd, float *running_mean, float *running_var,
                    const float *weight, const float *bias, float eps, float momentum,
                    int train, float leakiness, float *scratch, void *stream);
/* backward: `bias` is the forward pass's bias (as the reference's BatchNormalization_backward receives it,
 * pybind.cpp:219-221).  fp32 storage with leakiness >= 0 derives the activation mask from in, save_mean,
 * save_invstd, weight and bias (bit-identical to reading it from `out`, which may then be NULL).       */
int aabr_bn_backward(const float *in, float *d_in, const float *out, const float *d_out,
                     int64_t rows, int planes, const float *save_mean, const float *save_invstd,
                     const float *weight, const float *bias, float *d_weight, float *d_bias, float leakiness,
                     float *scratch, void *stream);

/* the same with d_in = BatchNorm gradient + d_in_add ([rows, planes] fp32, may be NULL): the gradient sum autograd
 * forms for a tensor with a second consumer (identity branch of a residual block, lateral connection) folded into
 * the apply pass.                                                                                       */
int aabr_bn_backward_add(const float *in, float *d_in, const float *out, const float *d_out, int64_t rows,
                         int planes, const float *save_mean, const float *save_invstd, const float *weight,
                         const float *bias, float *d_weight, float *d_bias, float leakiness, float *scratch,
                         const float *d_in_add, void *stream);

/* ---- bf16 feature storage (extension; BASELINE.json configs 3-5) ------------------------------
 * The reference instantiates its operators for float only (SCN/sparseconvnet_cuda.cpp:281-310).
 * These variants keep the SAME rule-book formats, fp32 parameters (W, bias, batch-norm affine and
 * statistics) and fp32/fp64 accumulation; only the [rows, planes] feature matrices (and their
 * gradients) are bfloat16, passed as uint16_t bit patterns.  Convolution: nIn % 32 == 0 and
 * nOut % 32 == 0 (v_mfma_f32_16x16x32_bf16 consumes one 32-plane chunk per instruction), every
 * buffer < 2 GiB; flags bits 0-2 as aabr_conv_forward; wpack = uint16
 * [aabr_conv_wpack_bf16_elems(vol,nIn,nOut)].  dW / d_bias stay fp32.                           */
int64_t aabr_conv_wpack_bf16_elems(int vol, int n_in, int n_out);
int aabr_conv_forward_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                           int n_out, int64_t V_out, const int32_t *blocks, int vol, const float *W,
                           const float *bias, int flags, uint16_t *wpack, void *stream);
/* bf16 storage through the wide-layer kernel (128-row tiles, columns split over the waves, gathered rows shared
 * through LDS): n_in % 64 == 0 (above 256: % 256), n_out % 64 == 0; the block stream is the fp32 wide kernel's
 * (aabr_build_wide_blocks), the weight pack aabr_conv_pack_weights2_bf16's.                              */
int aabr_conv_wide_tile_rows_bf16(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol);
int aabr_conv_forward_wide_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats, int n_out,
                                int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                int flags, const uint16_t *wpack, void *stream);
int aabr_conv_backward_weight_bf16(const uint16_t *in_feats, int n_in, const uint16_t *d_out,
                                   int n_out, int64_t V_out, const int32_t *pairs, int vol,
                                   int64_t max_chunks, float *dW, float *d_bias, float *scratch,
                                   void *stream);
int aabr_bn_forward_bf16(const uint16_t *in, uint16_t *out, int64_t rows, int planes,
                         float *save_mean, float *save_invstd, float *running_mean,
                         float *running_var, const float *weight, const float *bias, float eps,
                         float momentum, int train, float leakiness, float *scratch, void *stream);
int aabr_bn_backward_bf16(const uint16_t *in, uint16_t *d_in, const uint16_t *out,
                          const uint16_t *d_out, int64_t rows, int planes, const float *save_mean,
                          const float *save_invstd, const float *weight, const float *bias, float *d_weight,
                          float *d_bias, float leakiness, float *scratch, void *stream);
/* aabr_bn_backward_add for bf16 storage: d_in = bf16(bf16(BatchNorm gradient) + d_in_add); parts / nparts as
 * aabr_bn_backward_parts_bf16, or NULL / 0.                                                                        */
int aabr_bn_backward_add_bf16(const uint16_t *in, uint16_t *d_in, const uint16_t *out, const uint16_t *d_out,
                              int64_t rows, int planes, const float *save_mean, const float *save_invstd,
                              const float *weight, const float *bias, float *d_weight, float *d_bias, float leakiness,
                              const double *parts, int nparts, float *scratch, const uint16_t *d_in_add, void *stream);

/* ---- the proposal stage of a whole batch (extension) --------------------------------------------
 * The reference's RPNPostProcessor loops over the examples in Python (rpn/inference_3d.py:95-149).
 * aabr_rpn_gather_logits: out[b][j] = j-th logit of example b's cross-scale anchor list (maps in order; tables
 *   [nb][n_maps+1] / [nb][n_maps] on the host as for aabr_rpn_decode_maps), -inf from its end to lmax -- one
 *   top-k over dim 1 then selects for every example.
 * aabr_rpn_proposals_batch: per example aabr_rpn_decode_maps on selected[b] followed by aabr_rotate_nms_sorted of
 *   the decoded list; boxes / nms_boxes [nb,k,7], scores [nb,k], mask [nb][k*ceil(k/64)], keep [nb,k],
 *   meta [nb][AABR_META_WORDS] (meta[b][0] = number kept, left on the device).  At most 8 maps, 16 examples.  */
/* aabr_rpn_topk_maps (round 5): the per-example `objectness.topk(pre_nms_top_n, sorted=True)` of RPNPostProcessor
 *   (rpn/inference_3d.py:107-112) for ALL examples of a step in four launches, on the logits (sigmoid is monotone), with
 *   nothing concatenated: selected[b][0..k_host[b]) = indices into example b's cross-scale anchor list (tables as for
 *   aabr_rpn_gather_logits) in descending logit order, equal logits by ascending index.  info[2b] = candidates gathered,
 *   info[2b+1] != 0: more than 4096 logits share the 24-bit prefix at the cut (exact ties en masse) -- selected[b] is then
 *   not valid and the caller falls back to a full sort.  k <= 2048.  scratch: aabr_rpn_topk_scratch_words(nb) int32.  */
int64_t aabr_rpn_topk_scratch_words(int nb);
int aabr_rpn_topk_maps(int n_maps, const void *const *logit_ptrs, int nb, const int32_t *seg_begin_host,
                       const int32_t *site_begin_host, int num_anchors, const int32_t *k_host, int64_t *selected,
                       int64_t sel_stride, int32_t *info, int32_t *scratch, void *stream);
int aabr_rpn_gather_logits(int n_maps, const void *const *logit_ptrs, int nb, const int32_t *seg_begin_host,
                           const int32_t *site_begin_host, int num_anchors, int64_t lmax, float *out, void *stream);
int aabr_rpn_proposals_batch(int n_maps, const void *const *coords_ptrs, const void *const *logit_ptrs,
                             const void *const *regression_ptrs, int nb, const int32_t *seg_begin_host,
                             const int32_t *site_begin_host, const float *strides_host, const float *base_anchors,
                             int num_anchors, float voxel_scale, const float *weights_host, float clip,
                             float nms_min_yx, float nms_min_z, const int64_t *selected, int64_t k, float *boxes,
                             float *nms_boxes, float *scores, float nms_thresh, int only_xy, int64_t post_max,
                             uint64_t *mask, int64_t *keep, int32_t *meta, void *stream);

/* Offset split of the wide kernel for coarse maps (extension; same contraction, same results up to the summation
 * order over filter offsets, which is fixed: part order): when a layer has too few (tile, 64-column slab) items to
 * fill the chip -- the coarse FPN scales -- every item is cut into `parts` workgroups, each sweeping vol / parts filter
 * offsets into its own fp32 partial tile in `scratch` (parts x V_out x n_out floats, 16-byte aligned); a second launch
 * adds the parts in part order, then bias and residual.  aabr_conv_wide_split returns (parts << 16) | tile_rows when
 * a launch should take this path (asked after aabr_conv_wide_tile_rows returned 0), else 0; blocks =
 * aabr_build_wide_blocks(.., tile_rows).  fp32 storage.                                                        */
int aabr_conv_wide_split(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol);
int64_t aabr_conv_wide_split_scratch_floats(int64_t V_out, int n_out, int parts);
int aabr_conv_forward_wide_split(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                                 int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                 int flags, const float *wpack, const float *residual, int parts, float *scratch,
                                 void *stream);

/* the same for bf16 feature storage (in / out / wpack bf16, parts fp32, one rounding in the second stage; no residual) */
int aabr_conv_wide_split_bf16(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol);
int aabr_conv_forward_wide_split_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats, int n_out,
                                      int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                      int flags, const uint16_t *wpack, int parts, float *scratch, void *stream);
/* ... with `residual` (bf16 [V_out, n_out], may be NULL): out = bf16(bf16(sum of the parts + bias) + residual), what the
 * separate bf16 add of the consumer would have stored.                                                                */
int aabr_conv_forward_wide_split_bf16_res(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                          int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                          const float *bias, int flags, const uint16_t *wpack, int parts, float *scratch,
                                          const uint16_t *residual, void *stream);

/* 32 -> 32 plane layers (the finest scales; csrc/conv_narrow.hip): the same sum as aabr_conv_forward (Convolution.cpp:
 * 117-185), read from the GATHER TABLE `table` [vol][V_out] (input row of output row o at offset k, or -1 -- what
 * aabr_submanifold_table / aabr_convolution_tables leave behind) instead of a block stream, with the fp32 master weights
 * W [vol][32][32] staged in LDS by the launch itself (no pack call).  flags: bit 0 transposed weights, bit 1 mirrored
 * offsets (the submanifold input-gradient form).  aabr_conv_narrow_ok: 1 when the dispatch hands a launch to it (bf16
 * storage from 400,000 output rows on: where the tile kernels' per-block weight stream stops fitting the L2s).       */
int aabr_conv_narrow_ok(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol, int bf16);
int aabr_conv_forward_narrow(const float *in_feats, int64_t rows_in, float *out_feats, int64_t V_out, const int32_t *table,
                             int vol, const float *W, const float *bias, int flags, void *stream);
int aabr_conv_forward_narrow_bf16(const uint16_t *in_feats, int64_t rows_in, uint16_t *out_feats, int64_t V_out,
                                  const int32_t *table, int vol, const float *W, const float *bias, int flags, void *stream);
/* ... with the BatchNorm statistics of the write-out, one [2][32] fp64 pair per workgroup of the launch
 * (aabr_conv_narrow_parts(V_out) of them): forward sums of the stored values (consumer aabr_bn_forward_parts_bf16) /
 * backward sums of the BatchNorm whose d_out the launch writes (consumer aabr_bn_backward_parts_bf16), as the wide kernel's
 * aabr_conv_forward_wide_bf16_stats / _bwd_stats form them per tile. */
int aabr_conv_narrow_parts(int64_t V_out);
int aabr_conv_forward_narrow_bf16_stats(const uint16_t *in_feats, int64_t rows_in, uint16_t *out_feats, int64_t V_out,
                                        const int32_t *table, int vol, const float *W, const float *bias, int flags,
                                        double *stats, void *stream);
int aabr_conv_forward_narrow_bf16_bwd_stats(const uint16_t *in_feats, int64_t rows_in, uint16_t *out_feats, int64_t V_out,
                                            const int32_t *table, int vol, const float *W, const float *bias, int flags,
                                            double *stats, const uint16_t *bn_in, const uint16_t *bn_out,
                                            const float *save_mean, float leakiness, void *stream);

/* ---- compiled launch plans (extension) --------------------------------------------------------
 * The reference enters its library once per layer and direction from Python (SCN/pybind.cpp:134-221 behind
 * sparseconvnet/ layer modules).  A host that has compiled the static part of a network into a list of launches hands
 * the whole list over with ONE call: every record stands for one of the entry points above, called with the
 * record's fields in the order given here -- nothing is computed differently.
 *   kind AABR_PLAN_CONV       aabr_conv_forward[_bf16](p0 in, i32[0] n_in, i64[0] rows_in, p1 out, i32[1] n_out,
 *                             i64[1] V_out, p2 blocks, i32[2] vol, p3 W, p4 bias, i32[3] flags, p5 wpack)
 *        AABR_PLAN_CONV_WIDE  aabr_conv_forward_wide_stats(p0, i32[0], i64[0], p1, i32[1], i64[1], p2 blocks,
 *                             i32[4] tile_rows, i32[2] vol, p4 bias, i32[3] flags, p5 wpack, p3 residual, p6 stats);
 *                             i32[5] == 1: aabr_conv_forward_wide_bwd_stats(..., p6 stats, p7 bn_in, p8 save_mean,
 *                             p9 save_invstd, p10 bn_weight, p11 bn_bias, f32[0] leakiness); bf16 storage:
 *                             aabr_conv_forward_wide_bf16_bwd_stats(..., p6 stats, p7 bn_in, p9 bn_out, p8 save_mean,
 *                             f32[0] leakiness)
 *        AABR_PLAN_CONV_DW    aabr_conv_backward_weight[_bf16](p0 in, i32[0] n_in, p1 d_out, i32[1] n_out,
 *                             i64[0] V_out, p2 pairs, i32[2] vol, i64[1] max_chunks, p3 dW, p4 d_bias, p5 scratch)
 *        AABR_PLAN_BN_FWD     aabr_bn_forward[_bf16](p0 in, p1 out, i64[0] rows, i32[0] planes, p2 save_mean,
 *                             p3 save_invstd, p4 running_mean, p5 running_var, p6 weight, p7 bias, f32[0] eps,
 *                             f32[1] momentum, i32[1] train, f32[2] leakiness, p8 scratch); p9 != NULL:
 *                             aabr_bn_forward_parts[_bf16](..., p9 parts, i32[2] nparts, p8 scratch)
 *        AABR_PLAN_BN_BWD     aabr_bn_backward[_bf16](p0 in, p1 d_in, p2 out, p3 d_out, i64[0] rows, i32[0] planes,
 *                             p4 save_mean, p5 save_invstd, p6 weight, p10 bias, p7 d_weight, p8 d_bias,
 *                             f32[2] leakiness, p9 scratch); fp32 with p11 != NULL: aabr_bn_backward_add(..., p11);
 *                             i64[1] != 0: aabr_bn_backward_parts[_bf16](..., parts = (double *)i64[1], i32[1] nparts,
 *                             p9[, p11 in fp32 storage])
 *        AABR_PLAN_ADD        aabr_add(p0 a, p1 b, p2 out, i64[0] n)
 *        AABR_PLAN_CAST       aabr_cast_storage(p0 in, p1 out, i64[0] n, flags & AABR_PLAN_TO_BF16)
 *   flags & AABR_PLAN_BF16 selects the bf16-storage entry point.  Stops at the first failing record and returns
 *   its code (aabr_last_error() holds that entry point's message).                                    */
#define AABR_PLAN_CONV 1
#define AABR_PLAN_CONV_WIDE 2
#define AABR_PLAN_CONV_DW 3
#define AABR_PLAN_BN_FWD 4
#define AABR_PLAN_BN_BWD 5
#define AABR_PLAN_ADD 6
#define AABR_PLAN_CAST 7
#define AABR_PLAN_CONV_WIDE_SPLIT 9 /* aabr_conv_forward_wide_split[_bf16](p0, i32[0], i64[0], p1, i32[1], i64[1], p2 blocks,
                                       i32[4] tile_rows, i32[2] vol, p4 bias, i32[3] flags, p5 wpack, p3 residual (fp32
                                       storage only), i32[5] parts, p6 scratch) */
#define AABR_PLAN_CONV_NARROW 10 /* aabr_conv_forward_narrow[_bf16](p0 in, i64[0] rows_in, p1 out, i64[1] V_out, p2 table,
                                   i32[2] vol, p3 W, p4 bias, i32[3] flags); bf16 storage with p6 != NULL: .._bf16_stats(.., p6
                                   stats); i32[5] == 1: .._bf16_bwd_stats(.., p6 stats, p7 bn_in, p9 bn_out, p8 save_mean,
                                   f32[0] leakiness) */
#define AABR_PLAN_BF16 1
#define AABR_PLAN_TO_BF16 2
#define AABR_PLAN_JOIN 8 /* the caller's stream waits for the second stream in front of this record */
#define AABR_PLAN_SIDE 4 /* run this record on the library's second stream: it starts after everything recorded
                            before it, and the caller's stream waits for it before aabr_plan_run returns it */
typedef struct AabrPlanOp {
  int32_t kind, flags;
  int32_t i32[6];
  float f32[4];
  int64_t i64[4];
  void *p[12];
} AabrPlanOp; /* 176 bytes, no padding */
int aabr_plan_run(const AabrPlanOp *ops, int n_ops, void *stream);
/* Pipelined form (extension): a pass's list handed over in PARTS.  aabr_plan_submit copies the records, queues them for the
 * library's launcher thread and returns at once -- the caller fills the next part while this one's launches go out
 * (issuing ~450 launches costs the host 1.7 ms per training step, filling their records about as much).  Parts are
 * issued strictly in submission order with the semantics of aabr_plan_run; with hold_side != 0 the part leaves the
 * second stream unjoined for the part that follows (the last part of a pass passes 0).  aabr_plan_drain blocks until
 * every submitted part has been ISSUED and returns the first failing part's code (aabr_last_error() holds its
 * message): call it before enqueuing anything else on the streams used.  The failure is STICKY: from the failing part
 * on every submitted part -- of this pass or of a later one -- is dropped until aabr_plan_drain has returned the code.
 * aabr_plan_run refuses to run while submitted parts are queued or being issued (AABR_EINVAL); after the drain
 * aabr_conv_last_variant() on the calling thread names what the drained parts dispatched last. */
int aabr_plan_submit(const AabrPlanOp *ops, int n_ops, void *stream, int hold_side);
int aabr_plan_drain(void);
/* (tools) the launcher's counters since process start: time spent issuing parts, parts issued, times it found its
 * queue empty */
void aabr_plan_launcher_stats(int64_t *busy_ns, int64_t *parts, int64_t *sleeps);
/* Geometry plan (extension): the rule-book builders of a pass handed over as ONE list, each record = one of the
 * entry points above called with the record's fields -- nothing is computed differently:
 *   AABR_GEOM_SUBM_TABLE    aabr_submanifold_table(p0 coords, i64[0] V, p1 grid, i64[1] cap, i32[0..2] filter,
 *                           p2 table, p3 counts)
 *   AABR_GEOM_CONV_TABLES   aabr_convolution_tables2(p0 in_coords, i64[0] V_in, p1 in_grid, i64[1] in_cap,
 *                           p2 out_coords, i64[2] V_out, p3 out_grid, i64[3] out_cap, i32[0..2] size,
 *                           i32[3..5] stride, i32[6..8] out_spatial, p4 table_out, p5 table_in, p6 counts, p7 counts_in)
 *   AABR_GEOM_TILE_BLOCKS   aabr_build_tile_blocks(p0 table, i64[0] V, i32[0] vol, p1 blocks)
 *   AABR_GEOM_WIDE_BLOCKS   aabr_build_wide_blocks(p0 table, i64[0] V, i32[0] vol, i32[1] tile_rows, p1 blocks)
 *   AABR_GEOM_OFFSET_PAIRS  aabr_build_offset_pairs(p0 table, p1 block_counts, i64[0] V, i32[0] vol, p2 pairs)
 *   AABR_GEOM_CONV_SITES    aabr_convolution_sites(p0 in_coords, i64[0] V_in, i32[0..2] size, i32[3..5] stride,
 *                           i32[6..8] out_spatial, p1 out_grid, i64[1] out_cap, p2 scratch, p3 out_coords, p4 meta)
 *   AABR_GEOM_SAMPLE_OFFSETS aabr_sample_offsets(p0 coords, p1 meta, i64[0] V_max, i32[0] max_samples, p2 out)
 * Stops at the first failing record and returns its code.                                                      */
#define AABR_GEOM_SUBM_TABLE 1
#define AABR_GEOM_CONV_TABLES 2
#define AABR_GEOM_TILE_BLOCKS 3
#define AABR_GEOM_WIDE_BLOCKS 4
#define AABR_GEOM_OFFSET_PAIRS 5
#define AABR_GEOM_CONV_SITES 7
#define AABR_GEOM_SAMPLE_OFFSETS 8
/* brick grids (one allocation per level: the directory, then the bricks; dims packed sbx | sby << 16 | sbz << 32 | nb << 48):
 *   AABR_GEOM_BRICK_BUILD   aabr_brick_build(p0 in_coords, i64[0] vin_bound, p1 vin_count_dev, i32[0..2] size, i32[3..5]
 *                           stride, i32[6..8] out_spatial, i64[1] dims, p2 level, i64[2] nb_cap, p3 bcoord, p4 out_coords,
 *                           i64[3] v_cap, p5 meta, p6 scratch, i32[9] flags)
 *   AABR_GEOM_BRICK_SUBM    aabr_brick_submanifold_table(p0 coords, i64[0] V, i64[1] dims, p1 level, i32[0..2] filter,
 *                           p2 table, p3 block counts)
 *   AABR_GEOM_BRICK_TABLES  aabr_brick_convolution_tables(p0 in_coords, i64[0] V_in, i64[2] in dims, p1 in level,
 *                           p2 out_coords, i64[1] V_out, i64[3] out dims, p3 out level, i32[0..2] size, i32[3..5] stride,
 *                           i32[6..8] out_spatial, p4 table_out, p5 table_in, p6 counts, p7 counts_in)                  */
#define AABR_GEOM_BRICK_BUILD 9
#define AABR_GEOM_BRICK_SUBM 10
#define AABR_GEOM_BRICK_TABLES 11
typedef struct AabrGeomOp {
  int32_t kind, pad;
  int32_t i32[10];
  int64_t i64[4];
  void *p[8];
} AabrGeomOp; /* 144 bytes, no padding */
int aabr_geom_run(const AabrGeomOp *ops, int n_ops, void *stream);

/* Mailbox (extension): a small result (counts the host needs to size the next launches) handed over WITHOUT a stream or
 * event wait on the receiving side.  aabr_mailbox_create: coherent pinned host memory, [0] = sequence word (starts 0),
 * [1] reserved, payload from byte 8.  aabr_mailbox_post (one small kernel on `stream`): copies `bytes` (% 4) from
 * device memory into the payload, then stores `seq` (!= 0) into the sequence word with a system-scope release; the
 * host spins on that word and then reads the payload.  Why: hipStreamSynchronize / hipEventSynchronize / polling
 * hipEventQuery on the producing stream were measured to return only when the process's OTHER streams had drained
 * (csrc/plan.hip, profiles/r03_step_timeline.txt).  One post in flight per mailbox.                             */
int aabr_mailbox_create(int64_t payload_bytes, void **box);
int aabr_mailbox_destroy(void *box);
int aabr_mailbox_post(const void *src, int64_t bytes, void *box, uint32_t seq, void *stream);

/* out = a + b elementwise over n elements (fp32, or bf16 storage with the sum formed in fp32 and rounded to
 * nearest even); fp32 <-> bf16 storage cast.  What the layer API gets from torch (`a + b`, `.to(dtype)`; the
 * reference: AddTable, tables.py:27-41) as plan records.                                           */
int aabr_add(const void *a, const void *b, void *out, int64_t n, int bf16, void *stream);
int aabr_cast_storage(const void *in, void *out, int64_t n, int to_bf16, void *stream);
/* out_host[j][0] = sum of the n_host[j] int32 values at counts_host[j], for n_jobs rule books in one launch (the
 * host arrays hold DEVICE pointers).  The reference returns the rule total of a layer from host memory
 * (`forward_pass_multiplyAdd_count`, submanifoldConvolution.py:85-94); here the per-block rule counts live on the
 * device and the totals are kept there until somebody reads the counter.                             */
int aabr_sum_counts(const int32_t *const *counts_host, const int64_t *n_host, double *const *out_host, int n_jobs,
                    void *stream);

/* ---- rotated IoU / NMS ---------------------------------------------------------------------
 * iou[n,k] = devRotateIoUEval(query k, box n, criterion), then forced to 1 where the five
 * parameters differ by < 1e-6 -- rotate_iou_gpu_eval + check_same_boxes
 * (second/core/non_max_suppression/nms_gpu.py:552-717).  boxes [N,5], query [K,5]
 * = (xc, yc, size_a, size_b, yaw).                                                           */
int aabr_rotate_iou_eval(const float *boxes, int64_t N, const float *query, int64_t K,
                         int criterion, float *iou, void *stream);
/* boxes_iou_3d (utils3d/rotate_nms_3d_torch.py:23-90) on [.,7] yx_zb boxes;
 * aug_host[4] = {target_Y, target_Z, anchor_Y, anchor_Z}.                                    */
int aabr_boxes_iou_3d(const float *targets, int64_t M, const float *anchors, int64_t K,
                      const float *aug_host, int criterion, int only_xy, float *iou,
                      void *stream);
/* RPN label generation, fused and batched over the examples of a step.  Per example b and per
 * anchor of its concatenated list [map][site][yaw] (maps / segment tables as in aabr_rpn_decode_maps, one row of
 * seg_begin_host [nb][n_maps+1] and site_begin_host [nb][n_maps] per example): the IoU with every ground-truth box of
 * the example -- boxlist_iou_3d(target, anchor, aug_thickness, criterion, flag='rpn_label_generation'),
 * modeling/rpn/loss_3d.py:91-96, i.e. aabr_boxes_iou_3d on the anchors of anchor_generator_sparse3d.py:88-104 --
 * and Matcher.__call__ as make_rpn_loss_evaluator builds and calls it (modeling/rpn/loss_3d.py:96-100,338-344;
 * modeling/matcher.py:50-196): the entries whose |yaw difference| (utils3d/geometric_torch.py:4-21, target - anchor
 * wrapped to [-pi/2, pi/2)) is not below yaw_threshold are zeroed (no mask when yaw_threshold > 1.58, matcher.py:51);
 * matched_val = best masked value over the ground truths, matched_idx = its index (first maximum), or -1 below
 * bg_iou / -2 below fg_iou; with allow_low_quality_matches != 0 (the RPN's setting) set_low_quality_matches_
 * follows: an anchor that ties with the row maximum of any ground truth gets its best index back, and an anchor
 * still at -1 with an entry above max(0.02, row maximum - 0.05) of any ground truth becomes -2.  All -1 for an
 * example without ground truth.  Outputs are concatenated over the examples in order (sum_b N_b entries); iou_out
 * (optional, may be NULL) receives the UNMASKED [G_b, N_b] IoU matrices back to back.  target_ptrs[b] = device
 * [G_b, 7] yx_zb boxes; aug_host[4] = {target_Y, target_Z, anchor_Y, anchor_Z}; row_max_scratch = device words,
 * sum_b G_b of them (needed with allow_low_quality_matches; the call clears them itself).                      */
int aabr_rpn_label_generation(int n_maps, const void *const *coords_ptrs, int nb, const int32_t *seg_begin_host,
                              const int32_t *site_begin_host, const float *strides_host, const float *base_anchors,
                              int num_anchors, float voxel_scale, const void *const *target_ptrs,
                              const int32_t *n_targets_host, const float *aug_host, int criterion, int only_xy,
                              float fg_iou, float bg_iou, float yaw_threshold, int allow_low_quality_matches,
                              int64_t *matched_idx, float *matched_val, float *iou_out, uint32_t *row_max_scratch,
                              void *stream);
/* The same call with the other half of RPNLossComputation.prepare_targets (modeling/rpn/loss_3d.py:186-196):
 * regression_targets [sum_b N_b, 7] (optional, may be NULL) = box_coder.encode(target[matched_idx.clamp(min=0)], anchor) for
 * EVERY anchor, in the pass that has the anchor and its match in registers -- BoxCoder3D.encode_centroid_box
 * (modeling/box_coder_3d.py:46-51): second_box_encode(., ., smooth_dim=True) (second/pytorch/core/box_torch_ops.py:82-116),
 * yaw difference through limit_period(., 0.5, pi), times weights_host[7]; the un-thickened anchor and ground-truth box.
 * An example without ground truth encodes every anchor against itself (loss_3d.py:91-94: matched_targets = anchor).   */
int aabr_rpn_label_generation_targets(int n_maps, const void *const *coords_ptrs, int nb, const int32_t *seg_begin_host,
                                      const int32_t *site_begin_host, const float *strides_host,
                                      const float *base_anchors, int num_anchors, float voxel_scale,
                                      const void *const *target_ptrs, const int32_t *n_targets_host,
                                      const float *aug_host, int criterion, int only_xy, float fg_iou, float bg_iou,
                                      float yaw_threshold, int allow_low_quality_matches, int64_t *matched_idx,
                                      float *matched_val, float *iou_out, uint32_t *row_max_scratch,
                                      const float *weights_host, float *regression_targets, void *stream);
/* BoxCoder3D.encode (modeling/box_coder_3d.py:34-51, centroid form) on two [n,7] lists: out [n,7].                  */
int aabr_box_encode(const float *targets, const float *anchors, int64_t n, const float *weights_host, float *out,
                    void *stream);
/* BoxCoder3D.decode (modeling/box_coder_3d.py:40-44,53-80, centroid form): encodings [n, 7*num_classes] / weights, sizes
 * clamped at `clip` (bbox_xform_clip), second_box_decode with smooth_dim (box_torch_ops.py:118-154) against anchors [n,7]
 * (one anchor per row, shared by its classes), yaw through limit_period(., 0.5, pi): out [n, 7*num_classes].         */
int aabr_box_decode(const float *encodings, const float *anchors, int64_t n, int num_classes,
                    const float *weights_host, float clip, float *out, void *stream);
/* RPN glue: anchors of the selected flat indices t = site*A + yaw, generated from
 * the sparse site coordinates (modeling/rpn/anchor_generator_sparse3d.py:88-104:
 * centroid = loc / voxel_scale * stride, + base anchor), fused with BoxCoder3D.decode_centroid_box
 * (modeling/box_coder_3d.py:53-80; second_box_decode with smooth_dim, box_torch_ops.py:118-154;
 * limit_period to [-pi/2, pi/2]).  site_coords int32 [V,4]; selected int64 [k] indices relative
 * to (site_begin, reg_begin) of one example; regression float32 [*,7]; base_anchors [A,7];
 * boxes float32 [k,7] out (yx_zb).                                                             */
int aabr_rpn_decode(const int32_t *site_coords, int64_t site_begin, const int64_t *selected, int64_t k,
                    const float *regression, int64_t reg_begin, const float *base_anchors,
                    int num_anchors, float voxel_scale, const float *stride_host,
                    const float *weights_host, float clip, float *boxes, void *stream);

/* Cross-scale RPN proposals front end -- what RPNPostProcessor.forward_for_single_feature_map
 * (modeling/rpn/inference_3d.py:82-163) does per example AFTER cat_scales_obj_reg (rpn_sparse3d.py:19-77)
 * regrouped the scales example-major: for the k selected (top-k by objectness) entries of the example's
 * concatenated anchor list [map][site][yaw] -- anchor from the site coordinates
 * (anchor_generator_sparse3d.py:88-104) -> BoxCoder3D.decode_centroid_box (box_coder_3d.py:53-80) ->
 * sigmoid of the logit (:113) -> the boxlist_nms_3d thickness clamps (structures/boxlist_ops_3d.py:42-44)
 * into an NMS-only copy.  Host tables (n_maps <= 8): device pointers of each map's site list [V_m,4] int32,
 * logits [V_m*A] and regression [V_m*A,7]; seg_begin_host[n_maps+1] = first local anchor index of each map in
 * this example's list; site_begin_host[n_maps] = this example's first site row in each map;
 * strides_host[n_maps*3]; base_anchors device [n_maps*A,7].  selected int64 [k] (descending score).
 * Outputs: boxes [k,7], nms_boxes [k,7] (may be NULL), scores [k] (may be NULL).                 */
int aabr_rpn_decode_maps(int n_maps, const void *const *coords_ptrs, const void *const *logit_ptrs,
                         const void *const *regression_ptrs, const int32_t *seg_begin_host,
                         const int32_t *site_begin_host, const float *strides_host,
                         const float *base_anchors, int num_anchors, float voxel_scale,
                         const float *weights_host, float clip, float nms_min_yx, float nms_min_z,
                         const int64_t *selected, int64_t k, float *boxes, float *nms_boxes, float *scores,
                         void *stream);

/* Greedy rotated NMS over boxes already sorted by descending score: rotate_nms_3d_cc
 * (second/core/non_max_suppression/nms_cpu.py:32-44) with the suppression rule of
 * spconv-1.x rotate_non_max_suppression_cpu (IoU(i,j) > 0 and >= thresh).
 *   boxes7 [n,7]; mask scratch uint64 [n * ceil(n/64)]; keep int64 [n] out (indices into the
 *   sorted list, ascending = descending score); meta[0] = number kept (<= post_max).         */
int aabr_rotate_nms_sorted(const float *boxes7, int64_t n, float thresh, int only_xy,
                           int64_t post_max, uint64_t *mask, int64_t *keep, int32_t *meta,
                           void *stream);
/* maskrcnn_benchmark `_C.nms` (csrc/nms.h:10-28, csrc/cpu/nms_cpu.cpp:5-75,
 * csrc/cuda/nms.cu:23-131): axis-aligned, +1 pixel convention; dets [n,4] sorted by
 * descending score by the caller; keep = indices into the sorted list.                       */
int aabr_nms_sorted(const float *dets4, int64_t n, float thresh, uint64_t *mask, int64_t *keep,
                    int32_t *meta, void *stream);

/* ---- SparseToDense + rotated 3-D ROI align ----------------------------------
 * SparseToDense_updateOutput / _updateGradInput (SCN/CPU/SparseToDense.cpp:7-87, pybind.cpp:124-133):
 * out float32 [batch, planes, X, Y, Z] (zero-filled here), out[b][p][(x*Y+y)*Z+z] = in[row][p].   */
int aabr_sparse_to_dense_forward(const int32_t *site_coords, int64_t V, const float *in_feats, int planes,
                                 const int32_t *spatial_host, int64_t batch_size, float *out,
                                 void *stream);
int aabr_sparse_to_dense_backward(const int32_t *site_coords, int64_t V, float *d_in_feats, int planes,
                                  const int32_t *spatial_host, const float *d_out, void *stream);
/* `_C.roi_align_rotated_3d_forward / _backward` (maskrcnn_benchmark/csrc/vision.cpp:19-20,
 * csrc/cuda/ROIAlignRotated3D_cuda.cu:89-346): input [B,C,H,W,Z], rois [n,8] = (batch, center_w,
 * center_h, center_z, width, height, zsize, theta in degrees), output [n,C,ph,pw,pz].  The backward
 * zero-fills grad_input and accumulates with fp32 atomics, as the reference does.                   */
int aabr_roi_align_rotated_3d_forward(const float *input, const float *rois, int64_t num_rois,
                                      float spatial_scale, int channels, int height, int width, int zsize,
                                      int pooled_h, int pooled_w, int pooled_z, int sampling_ratio,
                                      float *output, void *stream);
int aabr_roi_align_rotated_3d_backward(const float *grad_output, const float *rois, int64_t num_rois,
                                       float spatial_scale, int pooled_h, int pooled_w, int pooled_z,
                                       int batch_size, int channels, int height, int width, int zsize,
                                       int sampling_ratio, float *grad_input, void *stream);

/* ---- fused sparse ROI-align (never densify) -----------------------------------------
 * Same contract as sparse_3d_to_dense_2d + _C.roi_align_rotated_3d_* (tools_3d_2d.py:7-48,
 * ROIAlignRotated3D_cuda.cu:16-346) without the dense [B,C,X,Y,Z] tensor: `cellmap` int32
 * [B, height, width, zsize] over the occupied extent (max coordinate + 1 per axis) holds the site row or -1
 * (aabr_roi_cellmap fills it); forward gathers feature rows [V, C] directly (bit-identical to the dense
 * form), backward adds into d_feats [V, C] (zeroed here; fp32 atomics on active cells only).          */
int aabr_roi_cellmap(const int32_t *site_coords, int64_t V, const int32_t *extent_host, int batch_size,
                     int32_t *cellmap, void *stream);
int aabr_roi_align_rotated_3d_sparse_forward(const float *feats, int channels, const int32_t *cellmap,
                                             int batch_size, int height, int width, int zsize,
                                             const float *rois, int64_t num_rois, float spatial_scale,
                                             int pooled_h, int pooled_w, int pooled_z, int sampling_ratio,
                                             float *output, void *stream);
int aabr_roi_align_rotated_3d_sparse_backward(const float *grad_output, int channels, const int32_t *cellmap,
                                              int batch_size, int height, int width, int zsize,
                                              const float *rois, int64_t num_rois, float spatial_scale,
                                              int pooled_h, int pooled_w, int pooled_z, int sampling_ratio,
                                              int64_t V, float *d_feats, void *stream);

#ifdef __cplusplus
}
#endif
#endif
