// lm_stream.h -- device-resident state of one frame stream (steps 02/03 of the pipeline).
//
// Everything the temporal stages need stays in HBM for the whole stream (sized for 288 GB):
//   cc[]        one 32-byte record per kept CC of every frame, in (frame, label) order
//   crop[]      bit-row crops of those CCs (absolute 32-px column alignment, see lm_cc_kernels.hip K8)
//   assign[]    unique-CC index every kept CC was matched to (cc_stability_estimator.py:102,117)
//   active*[]   ascending list of uniques still matchable (cc_stability_estimator.py:126-145): unique index,
//               first-seen CC, compact box for the candidate scan, last frame matched
#pragma once
#include "lm_common.h"

struct __attribute__((aligned(32))) LmCcRec {
    int32_t cc_id;      // label-1 in the frame's UNFILTERED labelling (labeler.py:177-185)
    int32_t size;
    int16_t min_x, max_x, min_y, max_y;   // inclusive box (frames up to 32767 px per side)
    unsigned long long crop_off;          // word offset into crop[]
    int32_t frame;
    int32_t pad;
};

struct LmCounters {
    long long n_cc;
    unsigned long long n_words;
    unsigned long long tempo_count;   // bbox-overlapping (cur, unique) pairs tested, cc_stability_estimator.py:85
    int n_frames;        // frames whose CCs have been emitted
    int n_matched;       // frames that went through matching
    int n_uniq;
    int n_active;
    int error;           // sticky LM_ERR_* raised on device (capacity)
    int pad;
};

struct LmStream {
    LmCtx* ctx;
    long long cap_cc;
    unsigned long long cap_words;
    int cap_frames, cap_uniq;
    double min_recall, min_precision;
    int max_gap, min_pixels;
    LmCcRec* cc;
    int32_t* assign;
    long long* frame_cc_off;    // [cap_frames + 1]
    uint32_t* crop;
    uint32_t* chash;            // [cap_cc] sum over the crop words of mix32(word + 0x9e3779b9 * index), left by lm_k_emit (twin detection key)
    // active list, parallel arrays in ascending unique index
    int32_t* active;            // unique index
    int32_t* active_cc;         // global cc index of the unique's first-seen CC
    unsigned long long* active_box;   // min_x | max_x<<16 | min_y<<32 | max_y<<48
    int32_t* active_last;       // last frame the unique was matched
    // per-frame decisions (reused every frame; never re-initialised thanks to the frame tag)
    unsigned long long* best;   // [ctx cap] frame tag << 32 | smallest accepted active-list position of the frame's i-th CC
    LmCounters* counters;       // device
    long long* batch_cc_base;   // [max_batch] staging for emit
    unsigned long long* batch_word_base;
    // reusable device scratch (hipMalloc / hipFree are device-wide sync points: none in steady state)
    void* rd_scratch;           // record packing for lm_stream_read
    size_t rd_scratch_bytes;
    void* garena;               // cached bump arena handed to lm_group_run
    size_t garena_bytes;
    int garena_busy;
    void* gpin;                 // cached pinned staging handed to lm_group_run (hipHostMalloc / hipHostFree wait for the whole device)
    size_t gpin_bytes;
    int gpin_busy;
    struct LmMatchBatch* mb;    // tables of the batched matcher (lm_match_batch.hip); host struct holding device pointers
    int last_match_frames;      // frames in the last batch handed to the batched matcher (lm_stream_match_stats)
    int match_per_frame;        // 1: one lm_k_match + lm_k_update per frame (LM_MATCH_PER_FRAME=1), 0: batched matcher
    hipEvent_t* run_events;     // lm_stream_run_logits: one event per batch (records -> matching), grown on demand
    int n_run_events;
    void* twin_stream;          // lm_stream_run_logits: queue of the twin-detection kernels (they need the batch's records only, not the matching state)
    int tempo_f0, tempo_B;      // lm_stream_run_logits (gated schedule): tempo_count kernel of the previous batch still to be launched (B > 0)
    int frames_pushed;          // host-side mirrors (frames are pushed and matched in order)
    int frames_matched;
};
