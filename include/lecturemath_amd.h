/*
 * lecturemath_amd.h -- C ABI of the MI355X-native LectureMath hot path (liblecturemath_hip.so, also
 * installed under the reference's own name ./accessmath_lib.so).
 *
 * Plain C, plain pointers and sizes, no torch types.  "d_" parameters are DEVICE pointers (HBM,
 * e.g. torch.Tensor.data_ptr()), "h_" parameters are host pointers.  `stream` is a hipStream_t passed
 * as void* (NULL = default stream; from PyTorch: torch.cuda.current_stream().cuda_stream).
 * Every int-returning function returns 0 (LM_OK) on success and an LM_ERR_* code otherwise;
 * lm_last_error() then describes the failure.  Nothing throws across the ABI.
 *
 * Reference interfaces replaced (paths relative to /root/reference/ACCESS2021_release):
 *   the five exports of accessmath_lib.c, bound with ctypes.CDLL('./accessmath_lib.so') at
 *   AccessMath/preprocessing/content/labeler.py:24, binarizer.py:22, tools/adaptive_equalizer.py:25
 *   + the third-party arithmetic the hot path calls around them (scipy.ndimage.label labeler.py:126,
 *   numpy crops labeler.py:183, numpy AND/count_nonzero connected_component.py:228, torch sigmoid +
 *   numpy threshold FCN_lecturenet.py:452-467, torch conv stack FCN_lecturenet.py:260-403).
 */
#ifndef LECTUREMATH_AMD_H
#define LECTUREMATH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LM_OK 0
#define LM_ERR_ARG 1
#define LM_ERR_HIP 2
#define LM_ERR_CAPACITY 3
#define LM_ERR_STATE 4

typedef struct LmCtx LmCtx;
typedef struct LmStream LmStream;

/* ---- library ---- */
int lm_abi_version(void);
const char* lm_last_error(void);
/* 1 when the library was built by hipcc for gfx950, 0 for the test-only CPU emulation build */
int lm_is_device_build(void);

/* ====================================================================================================
 * 1. Drop-in exports: the reference's accessmath_lib.c symbols, same names / signatures / return values.
 * ==================================================================================================== */

/* accessmath_lib.c:357-359 (argtypes/restype set at labeler.py:159-165).  HOST pointers, caller-allocated
 * outputs of length count_labels.  labels int32 H x W C-contiguous, ages fp32 H x W.  Returns 0. */
int CC_AgeBoundaries(int* labels, float* ages, int width, int height, int count_labels,
                     int* out_mins_y, int* out_maxs_y, int* out_mins_x, int* out_maxs_x,
                     int* out_counts, float* output_age);

/* ====================================================================================================
 * 2. Per-frame CC path on device-resident batches.
 * ==================================================================================================== */

/* Workspace for batches of up to max_batch frames of width x height (device memory, current device). */
LmCtx* lm_ctx_create(int width, int height, int max_batch);
void lm_ctx_destroy(LmCtx* ctx);

/* FCN_LectureNet.binarize post-processing + worker inversion (FCN_lecturenet.py:452,461-467;
 * FCN_lecturenet_binarizer.py:54): out = 255 - (trunc(sigmoid(logit)*255) >= thr ? 255 : 0), n pixels. */
int lm_threshold_invert(const float* d_logits, uint8_t* d_out, int64_t n, int thr, void* stream);
/* same without (invert = 0) or with (invert != 0) the worker's inversion: FCN_LectureNet.binarize itself returns the
 * non-inverted image (:461-467) */
int lm_threshold(const float* d_logits, uint8_t* d_out, int64_t n, int thr, int invert, void* stream);

/* scipy.ndimage.label (labeler.py:126) for n_frames frames (uint8, non-zero = foreground, contiguous
 * [n_frames][height][width]).  d_labels (int32, same shape) may be NULL when only the CC records are
 * wanted.  Leaves the run structures of the batch in the workspace for the calls below. */
int lm_label_batch(LmCtx* ctx, const uint8_t* d_binary, int n_frames, int32_t* d_labels, void* stream);

/* lm_threshold + lm_label_batch in ONE pass over fp32 logits (FCN_lecturenet.py:452-467, FCN_lecturenet_binarizer.py:54,
 * labeler.py:126): the frames are never materialised as bytes unless d_binary (may be NULL) asks for them.  Rows must be a
 * multiple of 4 pixels and 16-byte aligned for the fused kernel; otherwise the two calls run (d_binary required then).
 * lm_label_was_fused tells which of the two the last call on the context took (1 = fused). */
int lm_label_batch_logits(LmCtx* ctx, const float* d_logits, int n_frames, int thr, int invert, uint8_t* d_binary, int32_t* d_labels,
                          void* stream);
int lm_label_was_fused(LmCtx* ctx);

/* Number of labels of every frame of the last batch -> h_counts[n_frames] (synchronises the stream). */
int lm_label_counts(LmCtx* ctx, int32_t* h_counts, void* stream);

/* CC_AgeBoundaries (accessmath_lib.c:357-413) for the last labelled batch, computed from the runs.
 * Results stay on the device; lm_cc_stats_read copies frame `frame`'s arrays (length = its label count)
 * to host arrays in the reference's order mins_y, maxs_y, mins_x, maxs_x, counts. */
int lm_cc_stats_batch(LmCtx* ctx, void* stream);
int lm_cc_stats_read(LmCtx* ctx, int frame, int count_labels, int32_t* h_mins_y, int32_t* h_maxs_y,
                     int32_t* h_mins_x, int32_t* h_maxs_x, int32_t* h_counts, void* stream);

/* Live timing of the labelling launch sequence (pack .. write_labels) with HIP events recorded on the
 * caller's stream around every lm_label_batch call (also the ones lm_stream_push makes).
 * lm_ctx_profile_read synchronises the recorded events, returns the summed milliseconds, the number of
 * calls and the number of frames they covered since the last read, and resets the accumulators. */
int lm_ctx_set_profiling(LmCtx* ctx, int enable);
int lm_ctx_profile_read(LmCtx* ctx, double* h_total_ms, int64_t* h_calls, int64_t* h_frames);

/* Host-pointer convenience: one frame in, scipy-ordered labels out; returns the label count (>= 0)
 * or -LM_ERR_*.  Uses an internal lazily-created workspace on the current device. */
int lm_label_host(const uint8_t* h_img, int width, int height, int32_t* h_labels);

/* ====================================================================================================
 * 3. Frame stream: CC records + temporal matching (CCStabilityEstimator.add_frame,
 *    content/cc_stability_estimator.py:41-155), state resident in HBM.
 * ==================================================================================================== */

LmStream* lm_stream_create(LmCtx* ctx, int max_frames, int64_t max_ccs, int64_t max_crop_words, int max_uniques,
                           double min_recall, double min_precision, int max_gap, int min_pixels);
void lm_stream_destroy(LmStream* s);
int lm_stream_reset(LmStream* s, void* stream);
/* MIN_CC_PIXELS filter (labeler.py:22,177) applied to frames pushed from now on (filter_small=False callers use 1) */
int lm_stream_set_min_pixels(LmStream* s, int min_pixels);

/* Push the next n_frames binary frames (device, uint8): label -> stats -> kept CCs + crops -> match.
 * d_labels may be NULL.  Asynchronous on `stream`. */
int lm_stream_push(LmStream* s, const uint8_t* d_binary, int n_frames, int32_t* d_labels, void* stream);

/* The two halves of lm_stream_push, for frame-range sharding across GPUs: lm_stream_push_records only labels the frames and
 * appends their CC records + crops (per-frame work, any rank); lm_stream_match runs the sequential temporal matching over
 * the next n_frames unmatched frames (the rank that owns the whole stream, after lm_stream_import of the gathered records). */
int lm_stream_push_records(LmStream* s, const uint8_t* d_binary, int n_frames, int32_t* d_labels, void* stream);
int lm_stream_match(LmStream* s, int n_frames, void* stream);
/* lm_stream_push_records in two calls, for callers that schedule the bandwidth-bound labelling apart from the rest:
 * lm_label_batch(ctx of the stream, ...) labels n_frames frames, lm_stream_push_labelled appends their records + crops. */
int lm_stream_push_labelled(LmStream* s, int n_frames, void* stream);

/* Steps 01-02 of a whole stream of fp32 logits resident on the device in ONE call (FCN_lecturenet.py:452-467 thresholding,
 * FCN_lecturenet_binarizer.py:54 inversion, labeler.py:116-191, cc_stability_estimator.py:41-155): per batch of `batch` frames
 * threshold+invert -> label -> records on stream_wide and the temporal matching on stream_match (behind an event per batch;
 * pass the same stream twice for a single queue, do_match = 0 for the records only).  schedule 0: every batch's matching starts
 * as soon as its records are in; schedule 1: the labelling launches (bandwidth bound) are kept apart from the wide kernels of the
 * matching -- matching of batch k-1 starts when batch k has been labelled, batch k+1 is labelled when that matching has reached
 * its single-workgroup replay.  d_binary (may be NULL: the fused logits -> runs kernel needs no byte frames) is scratch for
 * `batch` frames, d_labels receives the label image of every batch in turn (or NULL).  Asynchronous. */
int lm_stream_run_logits(LmStream* s, const float* d_logits, int n_frames, int batch, uint8_t* d_binary, int32_t* d_labels, int thr,
                         int do_match, int schedule, void* stream_wide, void* stream_match);

/* Frame-range sharding across the GPUs of a node: the CC records + crops of frames
 * [first_frame, first_frame + n_frames) of a stream as ONE flat DEVICE buffer (32-byte aligned), so that the gather to the rank
 * that replays the temporal matching is one RCCL send/recv per rank -- no pickling, no host copy.  Replaces the reference's
 * per-lecture pickles as the hand-off between the per-frame half (labeler.py:116-191) and add_frame's matching
 * (cc_stability_estimator.py:71-145).  lm_stream_pack_size synchronises and returns the bytes lm_stream_pack will write;
 * lm_stream_append_packed appends a packed block behind the last frame of `s` (frame numbers and crop offsets are rebased on
 * the device); the appended frames are unmatched until lm_stream_match. */
int lm_stream_pack_size(LmStream* s, int first_frame, int n_frames, int64_t* h_bytes, void* stream);
int lm_stream_pack(LmStream* s, int first_frame, int n_frames, void* d_buf, int64_t bytes, void* stream);
int lm_stream_append_packed(LmStream* s, const void* d_buf, int64_t bytes, void* stream);

/* Hand-off of a MATCHED stream (step 03 on a rank of its own): the unique index of every kept CC (cc_idx_per_frame,
 * cc_stability_estimator.py:102,117) + {n_cc, n_unique, tempo_count, n_frames} as one flat DEVICE buffer of
 * lm_stream_assign_bytes() bytes (32-byte aligned).  The receiver appends the packed block(s) of the same frames first
 * (lm_stream_append_packed), then imports the assignment: the stream is then what the sender's was after matching, as far as
 * lm_group_run reads it (the active list stays behind: step 03 does not use it). */
int lm_stream_assign_bytes(LmStream* s, int64_t* bytes, void* stream);
int lm_stream_export_assign(LmStream* s, void* d_out, int64_t bytes, void* stream);
int lm_stream_import_assign(LmStream* s, const void* d_in, int64_t bytes, void* stream);

/* Diagnostic: sizes of the last batch handed to the batched matcher (lm_match_batch.hip):
 * out5 = {in-batch sources, CC tiles, pairs against earlier uniques, pairs against in-batch sources, frames}. Synchronises. */
int lm_stream_match_stats(LmStream* s, int64_t* out5, void* stream);

/* Synchronise and read the stream's counters: out[0]=n_frames, [1]=n_cc, [2]=n_crop_words, [3]=n_unique,
 * [4]=n_active, [5]=tempo_count, [6]=device error code. */
int lm_stream_counters(LmStream* s, int64_t* h_out7, void* stream);

/* Rebuild a stream on the device from host arrays in lm_stream_read's format (step 02 -> step 03 through the pickled
 * hand-off).  h_active / h_active_cc / h_active_last (n_active each: unique index, global index of its first-seen CC,
 * last frame it was matched; ascending unique index) restore the matching state so that lm_stream_push may continue;
 * pass NULL / 0 when only lm_group_run follows.  n_matched = leading frames whose records already carry their unique
 * index (n_frames for a finished stream, 0 for gathered raw records). */
int lm_stream_import(LmStream* s, const int32_t* h_rec, const int64_t* h_frame_off, const int64_t* h_crop_off,
                     const uint32_t* h_crop, int n_frames, int64_t n_cc, int64_t n_crop_words, int n_unique,
                     int64_t tempo_count, const int32_t* h_active, const int32_t* h_active_cc, const int32_t* h_active_last,
                     int n_active, int n_matched, void* stream);

/* Copy results to the host (call lm_stream_counters first to size the buffers).
 * h_rec: n_cc x 8 int32 = cc_id, min_x, max_x, min_y, max_y, size, frame, assigned unique index.
 * h_frame_off: n_frames + 1 int64.  h_crop_off: n_cc int64 word offsets.  h_crop: n_crop_words uint32.
 * h_active: n_active int32.  Any pointer may be NULL to skip that part. */
int lm_stream_read(LmStream* s, int32_t* h_rec, int64_t* h_frame_off, int64_t* h_crop_off, uint32_t* h_crop,
                   int32_t* h_active, void* stream);

/* ====================================================================================================
 * 4. Step 03, CC grouping in space-time (pre_ST3D_v3.0_03_cc_grouping.py:22-118 and the CCStabilityEstimator
 *    methods it calls, content/cc_stability_estimator.py:181-681) over a finished stream.
 * ==================================================================================================== */
typedef struct LmGroups LmGroups;

/* Runs split_stable_cc_by_gaps(max_gap, min_times), get_stable_cc_idxs(min_times),
 * compute_overlapping_stable_cc(t_window), compute_groups(min_recall), compute_groups_temporal_information,
 * compute_conflicting_groups, compute_group_images(img_threshold) and prepares frames_from_groups when
 * reconstruct_tables != 0.  Box join, pixel overlaps, group images and frame rendering run on the device;
 * the order-dependent list bookkeeping runs on the host.  Returns NULL on failure (lm_last_error). */
LmGroups* lm_group_run(LmStream* s, int max_gap, int min_times, int t_window, double min_recall, double img_threshold,
                       int reconstruct_tables, void* stream);
void lm_group_destroy(LmGroups* g);

/* frames_from_groups (:638-681, the encoded channel 0): frames [first, first + n) -> d_out [n][height][width] uint8 */
int lm_group_render(LmGroups* g, int first, int n, uint8_t* d_out, void* stream);

/* Step 04 helper: exact per-frame sums of n_frames uint8 frames resident on the device (d_sums[f] = sum of the frame's
 * bytes).  Replaces the integer part of VideoSegmenter.compute_binary_sums (AccessMath/preprocessing/content/
 * video_segmenter.py:22-28, `binary.sum() / 255`); the float64 division by 255 stays with the caller. */
int lm_frame_sums(const uint8_t* d_frames, int n_frames, int64_t pixels_per_frame, uint64_t* d_sums, void* stream);

/* Step 05 helper: which pairs (i < j) of n images placed in the frame share an ink pixel?  Replaces the all-pairs
 * ConnectedComponent.getOverlapFMeasure loop of CCStabilityEstimator.compute_overlapping_CC_groups
 * (AccessMath/preprocessing/content/cc_stability_estimator.py:696-714) and the incompatibility tests of
 * KeyframeExtractor.GenerateFromST3DForIntervals (keyframe_extractor.py:85-91); "recall > 0 or precision > 0" of
 * connected_component.py:202-250 is "one common pixel".  Host pointers: h_boxes [n][4] = min_x, max_x, min_y, max_y
 * (inclusive), h_img_off [n + 1] byte offsets into h_images (image k is h x w uint8, non-zero = ink).  Up to cap pairs,
 * sorted by (i, j), are written to h_pairs [cap][2]; *n_pairs = pairs found (LM_ERR_CAPACITY when that exceeds cap). */
int lm_image_pairs_overlap(const int32_t* h_boxes, const uint8_t* h_images, const int64_t* h_img_off, int n, int32_t* h_pairs,
                           int64_t cap, int64_t* n_pairs, void* stream);

/* ---- the remaining exports of the reference's accessmath_lib.c (classical, pre-FCN binarizers) --------
 * Same symbols and C signatures as the reference, host pointers (ctypes), results bit-identical to the C library:
 *   accessmath_lib.c:7-111    speaker_detection_handle_frame (bound by AccessMath/preprocessing/video_worker/ speaker detection)
 *   accessmath_lib.c:113-173  regionCumulativeDistribution   (adaptive_equalizer.py:274-291)
 *   accessmath_lib.c:175-329  adapthisteq                    (adaptive_equalizer.py, binarizer.py:139-246)
 *   accessmath_lib.c:331-355  combine_results                (binarizer.py:382-402) */
int speaker_detection_handle_frame(unsigned char* frame, unsigned char* last_frame, int width, int height, int channels, int threshold,
                                   int jump_cells, double* change_boundaries, double* change_avg, double* change_deviation);
void regionCumulativeDistribution(unsigned char* grayscale, int width, int height, int min_x, int max_x, int min_y, int max_y,
                                  double slope_max, double* output);
int adapthisteq(unsigned char* grayscale, int width, int height, double slope, int grid_x, int grid_y, unsigned char* output);
int combine_results(unsigned char* only_board, unsigned char* equalized, int width, int height, unsigned char threshold,
                    unsigned char* final_content);

/* Host arrays owned by g (valid until lm_group_destroy): *ptr, *count (elements).  Array ids and element types:
 *  0 UNIQ_CC i32[n_uniq]  first-seen CC record of every unique (after the split)
 *  1 ULIST_OFF i64[n_uniq+1], 2 ULIST_CC i32   CSR of unique_cc_frames (entries are global CC indices)
 *  3 ASSIGN i32[n_cc]  unique index of every kept CC after the split (cc_idx_per_frame)
 *  4 STABLE i32   5 PAIR_A i32, 6 PAIR_B i32, 7 PAIR_MATCH i32  (sorted bbox-overlapping stable pairs + pixel matches)
 *  8 TOV_OFF i64[n_uniq+1], 9 TOV_OTHER i32, 10 TOV_RECALL f64, 11 TOV_PRECISION f64       time_overlapping_cc
 * 12 AOV_OFF i64[n_uniq+1], 13 AOV_OTHER i32, 14 AOV_MATCHED i32, 15 AOV_SIZE_OTHER i32, 16 AOV_SIZE_SELF i32   all_overlapping_cc
 * 17 GRP_OFF i64[n_groups+1], 18 GRP_MEMBERS i32 (cc_groups), 19 GID i32[n_uniq] (group_idx_per_cc, -1 = none)
 * 20 AGES_OFF i64[n_groups+1], 21 AGES i32 (group_ages), 22 GPF_OFF i64[n_frames+1], 23 GPF i32 (groups_per_frame)
 * 24 CONF_G1 i32, 25 CONF_G2 i32, 26 CONF_MATCHED i64, 27 CONF_UNMATCHED i64, 28 CONF_UNION i64, 29 CONF_INTER f64
 * 30 BOUNDS i32[n_groups][4] (min_x, max_x, min_y, max_y)
 * 31 GIMG_OFF i64[n_items+1] byte offsets, 32 GIMG_ITEM_OFF i64[n_groups+1] first item of a group, 33 GIMG u8 (copied on demand)
 * 34 SCALARS i64[6]: n_split, total_intersections, n_groups, n_uniq, n_frames, group image bytes */
int lm_group_array(LmGroups* g, int which, const void** ptr, int64_t* count);

/* ====================================================================================================
 * 5. FCN-LectureNet inference (lecturenet_v1/FCN_lecturenet.py: encode_decode :260-323, forward :364-403).
 * ==================================================================================================== */
typedef struct LmFcn LmFcn;

/* widths18 = d1..d5, mid, u5,c5, u4,c4, u3,c3, u2,c2, u1,c1, pixel_features_1, pixel_features_2 (CreateFromConfig
 * :621-646; every width a multiple of 8).  Activation buffers are allocated for frames up to max_h x max_w. */
LmFcn* lm_fcn_create(const int32_t* widths18, int pixel_kernel, int kernel, int max_h, int max_w);
void lm_fcn_destroy(LmFcn* f);

/* One layer's weights, BatchNorm already folded in and packed by the host (lecturemath_amd/fcn.py documents the
 * layout; layer ids: 0-4 conv_down_block_1..5, 5 mid_block, 6-10 transposed_conv_5..1, 11-15 conv_up_block_5..1,
 * 16 conv_text_mask_out, 17 conv_reconstruct, 18 conv_pixels_1, 19 conv_pixels_2, 20 conv_out).  HOST pointers. */
int lm_fcn_set_layer(LmFcn* f, int layer, const float* h_w, int64_t w_count, const float* h_bias, int bias_count, int cin,
                     int cout, int k, int ck);

/* Operand format of one layer whose weights were packed for the fp16-split kernels (ck <= 0 above): MFMA products per
 * operand pair, 3 = hi.hi + hi.lo + lo.hi (~22 bits per operand), 2 = activations split / weights rounded to f16,
 * 1 = both operands rounded to f16.  The torch reference computes these contractions in fp32 (FCN_lecturenet.py:260-403);
 * lecturemath_amd/fcn.py holds the shipped per-layer assignment and profiles/ the error attribution behind it. */
int lm_fcn_set_layer_terms(LmFcn* f, int layer, int terms);

/* forward() for one RGB frame (device, uint8 [h][w][3]).  Device fp32 outputs (each may be NULL): d_out [h*w]
 * binarization logit (no sigmoid), d_text [h*w] text-mask logit, d_rec [3][h*w] tanh reconstruction. */
int lm_fcn_forward(LmFcn* f, const uint8_t* d_rgb, int h, int w, float* d_out, float* d_text, float* d_rec, void* stream);

/* Frame pre / post-processing of binarize()'s > 2.5 MP branch on the device (csrc/lm_resize.hip):
 * PIL.Image.resize(..., LANCZOS) (FCN_lecturenet.py:434-437; Pillow's two-pass 8-bit resampling with 22-bit fixed-point taps --
 * the taps and their bounds per output column / row are DEVICE tables computed by lecturemath_amd/resize.py) and
 * cv2.resize(..., INTER_NEAREST) back to the frame's size (:481-486).  channels: 1 or 3 (interleaved).  d_tmp: in_h * out_w *
 * channels bytes of scratch. */
int lm_resample_rgb8(const uint8_t* d_in, int in_h, int in_w, int channels, uint8_t* d_tmp, uint8_t* d_out, int out_h, int out_w,
                     const int32_t* d_bounds_h, const int32_t* d_kk_h, int ksize_h, const int32_t* d_bounds_v, const int32_t* d_kk_v,
                     int ksize_v, void* stream);
int lm_upsample_nearest_u8(const uint8_t* d_in, int in_h, int in_w, int channels, uint8_t* d_out, int out_h, int out_w, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Second FCN engine (csrc/lm_fcn2.hip): the same forward pass for the shipped topology (3x3 encoder / decoder, 7x7 pixel
 * branch, every width a multiple of 16) on planar f16 activations and one gather-GEMM kernel, operand format per layer.
 * lecturemath_amd/fcn2.py builds the per-layer recipes; lecturemath_amd/fcn.py picks this engine when the network fits it.
 * lo25[t] = 1 keeps the lo parts of activation tensor t (a consumer runs the split format). */
typedef struct LmFcn2 LmFcn2;
LmFcn2* lm_fcn2_create(const int32_t* widths18, const int32_t* lo25, int max_h, int max_w);
void lm_fcn2_destroy(LmFcn2* f);
/* desc: kh, kw, operand format (1 = f16, 2 = activations hi + lo, 3 = both split, 4 = weights hi + lo), mt, epilogue, nchunks, planes
 * per chunk, ngroups, nslices, flags (bits 0-3: column tiles of 16 pixels per wave row group, 1 or 2; bit 8: loader wave; bits 16-23: LDS
 * target in KB), double-buffered planes, LDS bytes of a weight buffer, outputs; then planes [nchunks * npc][tensor, octet], weight groups
 * [ngroups][first slice, slices, chunk], slice table [nslices][4] (LDS byte offsets of the four k-groups' (plane, tap) pairs inside the
 * patch buffer).  HOST pointers.  Fails for a variant the library holds no kernel instance of (lecturemath_amd/fcn2.py: have_instance). */
int lm_fcn2_set_layer(LmFcn2* f, int layer, const int32_t* desc, int ndesc, const void* h_w, int64_t wbytes, int wblocks,
                      const float* h_bias, int nbias);
/* same contract as lm_fcn_forward */
int lm_fcn2_forward(LmFcn2* f, const uint8_t* d_rgb, int h, int w, float* d_out, float* d_text, float* d_rec, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LECTUREMATH_AMD_H */
