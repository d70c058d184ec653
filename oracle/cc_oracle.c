/*
 * oracle/cc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's per-frame connected-component path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (lecturemath_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function here is checked against outputs of the reference itself
 * run in the build container (tests/golden/make_golden.py -> tests/golden/ fixtures, and directly in
 * tests/test_oracle_vs_reference.py when /root/reference is present).  The reference ships no
 * tests or golden vectors of its own for this path.
 *
 * What each function follows (paths relative to /root/reference/ACCESS2021_release):
 *   orc_label4            scipy.ndimage.label as called at AccessMath/preprocessing/content/labeler.py:126
 *                         (third-party: scipy, un-pinned by the reference; behaviour probed on scipy 1.15.3:
 *                         default cross structure = 4-connectivity, any non-zero pixel is foreground,
 *                         int32 labels numbered 1..n in raster order of each component's first pixel)
 *   orc_age_boundaries    CC_AgeBoundaries, accessmath_lib.c:357-413
 *   orc_extract           the per-label crop loop, labeler.py:171-189 (MIN_CC_PIXELS filter, labeler.py:22)
 *   orc_overlap           ConnectedComponent.getOverlapFMeasure(other, False, False),
 *                         AM_CommonTools/data/connected_component.py:202-250 (the match count only;
 *                         callers divide)
 *   orc_threshold_invert  FCN_LectureNet.binarize post-processing, lecturenet_v1/FCN_lecturenet.py:452,461-467
 *                         followed by the worker's inversion, video_worker/FCN_lecturenet_binarizer.py:54
 *   orc_stab_*            CCStabilityEstimator.__init__/add_frame, content/cc_stability_estimator.py:11-155
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ labelling */

static int32_t uf_find(int32_t *parent, int32_t a)
{
    int32_t r = a;
    while (parent[r] != r) r = parent[r];
    while (parent[a] != r) { int32_t n = parent[a]; parent[a] = r; a = n; }
    return r;
}

/* 4-connected labelling, labels 1..n in raster order of first pixel. Returns n (or -1 on OOM). */
int orc_label4(const uint8_t *img, int w, int h, int32_t *labels)
{
    size_t npx = (size_t)w * (size_t)h;
    int32_t *parent = (int32_t *)malloc((npx / 2 + 2) * sizeof(int32_t));
    if (!parent) return -1;
    int32_t next = 1; /* provisional ids start at 1; 0 = background */
    for (int y = 0; y < h; y++) {
        const uint8_t *row = img + (size_t)y * w;
        int32_t *lrow = labels + (size_t)y * w;
        const int32_t *urow = lrow - w;
        for (int x = 0; x < w; x++) {
            if (!row[x]) { lrow[x] = 0; continue; }
            int32_t left = (x > 0) ? lrow[x - 1] : 0;
            int32_t up = (y > 0) ? urow[x] : 0;
            if (!left && !up) {
                parent[next] = next;
                lrow[x] = next++;
            } else if (left && up) {
                int32_t a = uf_find(parent, left), b = uf_find(parent, up);
                int32_t m = a < b ? a : b;
                parent[a] = m;
                parent[b] = m;
                lrow[x] = m;
            } else {
                lrow[x] = left ? left : up;
            }
        }
    }
    /* second pass: number roots in order of first appearance */
    int32_t *final_id = (int32_t *)calloc((size_t)next + 1, sizeof(int32_t));
    if (!final_id) { free(parent); return -1; }
    int32_t n = 0;
    for (size_t i = 0; i < npx; i++) {
        int32_t l = labels[i];
        if (!l) continue;
        int32_t r = uf_find(parent, l);
        if (!final_id[r]) final_id[r] = ++n;
        labels[i] = final_id[r];
    }
    free(final_id);
    free(parent);
    return n;
}

/* ------------------------------------------------------------------ per-label statistics */

int orc_age_boundaries(const int32_t *labels, const float *ages, int width, int height, int count_labels,
                       int32_t *mins_y, int32_t *maxs_y, int32_t *mins_x, int32_t *maxs_x,
                       int32_t *counts, float *out_age)
{
    for (int i = 0; i < count_labels; i++) {
        mins_y[i] = height; maxs_y[i] = 0;
        mins_x[i] = width;  maxs_x[i] = 0;
        counts[i] = 0;      out_age[i] = -1.0f;
    }
    size_t p = 0;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++, p++) {
            int32_t l = labels[p];
            if (l <= 0) continue;
            int k = l - 1;
            if (y < mins_y[k]) mins_y[k] = y;
            if (y > maxs_y[k]) maxs_y[k] = y;
            if (x < mins_x[k]) mins_x[k] = x;
            if (x > maxs_x[k]) maxs_x[k] = x;
            counts[k]++;
            float a = ages ? ages[p] : 0.0f;
            if (out_age[k] < 0.0f || a < out_age[k]) out_age[k] = a;
        }
    return 0;
}

/* ------------------------------------------------------------------ crop extraction
 * Two calls: with crops == NULL it only counts/returns sizes; otherwise it fills.
 * rec layout per kept CC (6 x int32): cc_id, min_x, max_x, min_y, max_y, size.
 * crop_off[k] = byte offset of CC k's (h x w) uint8 0/255 crop inside crops.
 * Returns the number of kept CCs; *crop_bytes receives the total arena size. */
int orc_extract(const int32_t *labels, int width, int height, int count_labels, int min_pixels,
                int32_t *rec, int64_t *crop_off, uint8_t *crops, int64_t *crop_bytes)
{
    int32_t *st = (int32_t *)malloc((size_t)(count_labels > 0 ? count_labels : 1) * 5 * sizeof(int32_t));
    float *ag = (float *)malloc((size_t)(count_labels > 0 ? count_labels : 1) * sizeof(float));
    int32_t *mny = st, *mxy = st + count_labels, *mnx = st + 2 * count_labels, *mxx = st + 3 * count_labels,
            *cnt = st + 4 * count_labels;
    orc_age_boundaries(labels, NULL, width, height, count_labels, mny, mxy, mnx, mxx, cnt, ag);
    int kept = 0;
    int64_t off = 0;
    for (int k = 0; k < count_labels; k++) {
        if (cnt[k] < min_pixels) continue;
        int cw = mxx[k] - mnx[k] + 1, ch = mxy[k] - mny[k] + 1;
        if (rec) {
            int32_t *r = rec + (size_t)kept * 6;
            r[0] = k; r[1] = mnx[k]; r[2] = mxx[k]; r[3] = mny[k]; r[4] = mxy[k]; r[5] = cnt[k];
        }
        if (crop_off) crop_off[kept] = off;
        if (crops) {
            uint8_t *dst = crops + off;
            for (int yy = 0; yy < ch; yy++) {
                const int32_t *src = labels + (size_t)(mny[k] + yy) * width + mnx[k];
                for (int xx = 0; xx < cw; xx++) dst[(size_t)yy * cw + xx] = (src[xx] == k + 1) ? 255 : 0;
            }
        }
        off += (int64_t)cw * ch;
        kept++;
    }
    if (crop_bytes) *crop_bytes = off;
    free(st);
    free(ag);
    return kept;
}

/* ------------------------------------------------------------------ pixel overlap
 * boxes are inclusive (min_x, max_x, min_y, max_y); crops are row-major (h x w) uint8, non-zero = ink.
 * Returns the number of pixels set in both, 0 when the boxes are disjoint. */
int64_t orc_overlap(const int32_t *box_a, const uint8_t *crop_a, const int32_t *box_b, const uint8_t *crop_b)
{
    int ax0 = box_a[0], ax1 = box_a[1], ay0 = box_a[2], ay1 = box_a[3];
    int bx0 = box_b[0], bx1 = box_b[1], by0 = box_b[2], by1 = box_b[3];
    if (!(ay1 >= by0 && by1 >= ay0 && ax1 >= bx0 && bx1 >= ax0)) return 0;
    int x0 = ax0 > bx0 ? ax0 : bx0, x1 = ax1 < bx1 ? ax1 : bx1;
    int y0 = ay0 > by0 ? ay0 : by0, y1 = ay1 < by1 ? ay1 : by1;
    int aw = ax1 - ax0 + 1, bw = bx1 - bx0 + 1;
    int64_t m = 0;
    for (int y = y0; y <= y1; y++) {
        const uint8_t *ra = crop_a + (size_t)(y - ay0) * aw + (x0 - ax0);
        const uint8_t *rb = crop_b + (size_t)(y - by0) * bw + (x0 - bx0);
        for (int x = 0; x <= x1 - x0; x++) m += (ra[x] & rb[x]) != 0;
    }
    return m;
}

/* ------------------------------------------------------------------ threshold + inversion
 * sigmoid in fp32 (torch.sigmoid on an fp32 tensor), *255 in fp32 (numpy float32 * int),
 * astype(uint8) truncation, >= thr -> 255 else 0, then 255 - binary.
 * NOTE: expf here is the host libm's; the reference uses torch's vectorised sigmoid, so pixels whose
 * fp32 sigmoid*255 lands within 1 ulp of thr can differ between the two -- see DESIGN.md (threshold edge). */
void orc_threshold_invert(const float *logits, int64_t n, int thr, uint8_t *out)
{
    for (int64_t i = 0; i < n; i++) {
        float s = 1.0f / (1.0f + expf(-logits[i]));
        float v = s * 255.0f;
        uint8_t u = (uint8_t)v; /* truncation; v is in [0,255] */
        uint8_t b = (u >= thr) ? 255 : 0;
        out[i] = (uint8_t)(255 - b);
    }
}

/* ------------------------------------------------------------------ temporal stability (step 02)
 * State mirrors what the reference keeps: unique CCs (first-seen box/size/crop), their frame lists,
 * last-seen frame, ordered active list, running count of bbox-overlapping pairs. */
typedef struct {
    int32_t box[4]; /* min_x, max_x, min_y, max_y */
    int32_t size;
    uint8_t *crop;  /* owned copy of the first-seen crop */
    int32_t *frames; /* pairs (frame, raw_label) */
    int32_t n_frames, cap_frames;
    int32_t last_frame;
} orc_unique;

typedef struct {
    int width, height;
    double min_recall, min_precision;
    int max_gap;
    int img_idx;
    int64_t tempo_count;
    orc_unique *uniq;
    int32_t n_uniq, cap_uniq;
    int32_t *active;
    int32_t n_active, cap_active;
    /* per-frame assignment log: for every frame, (unique_idx, cc_id) per kept CC, in CC order */
    int32_t *log;       /* pairs */
    int64_t n_log, cap_log;
    int64_t *frame_off; /* offset (in pairs) of each frame's first entry; n = img_idx + 1 */
    int64_t cap_frame_off;
} orc_stab;

orc_stab *orc_stab_new(int width, int height, double min_recall, double min_precision, int max_gap)
{
    orc_stab *s = (orc_stab *)calloc(1, sizeof(orc_stab));
    s->width = width; s->height = height;
    s->min_recall = min_recall; s->min_precision = min_precision; s->max_gap = max_gap;
    s->cap_frame_off = 1024;
    s->frame_off = (int64_t *)malloc(s->cap_frame_off * sizeof(int64_t));
    s->frame_off[0] = 0;
    return s;
}

void orc_stab_free(orc_stab *s)
{
    if (!s) return;
    for (int32_t i = 0; i < s->n_uniq; i++) { free(s->uniq[i].crop); free(s->uniq[i].frames); }
    free(s->uniq); free(s->active); free(s->log); free(s->frame_off); free(s);
}

static void uniq_push_frame(orc_unique *u, int32_t frame, int32_t raw_label)
{
    if (u->n_frames == u->cap_frames) {
        u->cap_frames = u->cap_frames ? u->cap_frames * 2 : 4;
        u->frames = (int32_t *)realloc(u->frames, (size_t)u->cap_frames * 2 * sizeof(int32_t));
    }
    u->frames[2 * u->n_frames] = frame;
    u->frames[2 * u->n_frames + 1] = raw_label;
    u->n_frames++;
}

static int32_t stab_new_unique(orc_stab *s, const int32_t *rec, const uint8_t *crop)
{
    if (s->n_uniq == s->cap_uniq) {
        s->cap_uniq = s->cap_uniq ? s->cap_uniq * 2 : 1024;
        s->uniq = (orc_unique *)realloc(s->uniq, (size_t)s->cap_uniq * sizeof(orc_unique));
    }
    orc_unique *u = &s->uniq[s->n_uniq];
    memset(u, 0, sizeof(*u));
    u->box[0] = rec[1]; u->box[1] = rec[2]; u->box[2] = rec[3]; u->box[3] = rec[4];
    u->size = rec[5];
    size_t nb = (size_t)(rec[2] - rec[1] + 1) * (size_t)(rec[4] - rec[3] + 1);
    u->crop = (uint8_t *)malloc(nb);
    memcpy(u->crop, crop, nb);
    u->last_frame = s->img_idx;
    uniq_push_frame(u, s->img_idx, rec[0] + 1);
    if (s->n_active == s->cap_active) {
        s->cap_active = s->cap_active ? s->cap_active * 2 : 1024;
        s->active = (int32_t *)realloc(s->active, (size_t)s->cap_active * sizeof(int32_t));
    }
    s->active[s->n_active++] = s->n_uniq;
    return s->n_uniq++;
}

static void stab_log(orc_stab *s, int32_t uidx, int32_t cc_id)
{
    if (s->n_log == s->cap_log) {
        s->cap_log = s->cap_log ? s->cap_log * 2 : 4096;
        s->log = (int32_t *)realloc(s->log, (size_t)s->cap_log * 2 * sizeof(int32_t));
    }
    s->log[2 * s->n_log] = uidx;
    s->log[2 * s->n_log + 1] = cc_id;
    s->n_log++;
}

/* One frame given as an already binary image (input_binary=True, the only mode the v3.0 scripts use). */
int orc_stab_add_frame(orc_stab *s, const uint8_t *binary, int min_pixels)
{
    int w = s->width, h = s->height;
    int32_t *labels = (int32_t *)malloc((size_t)w * h * sizeof(int32_t));
    int n = orc_label4(binary, w, h, labels);
    int kept = 0;
    int32_t *rec = NULL; int64_t *coff = NULL; uint8_t *crops = NULL; int64_t cbytes = 0;
    if (n > 0) {
        kept = orc_extract(labels, w, h, n, min_pixels, NULL, NULL, NULL, &cbytes);
        rec = (int32_t *)malloc((size_t)(kept ? kept : 1) * 6 * sizeof(int32_t));
        coff = (int64_t *)malloc((size_t)(kept ? kept : 1) * sizeof(int64_t));
        crops = (uint8_t *)malloc((size_t)(cbytes ? cbytes : 1));
        orc_extract(labels, w, h, n, min_pixels, rec, coff, crops, &cbytes);
    }
    free(labels);

    if (s->img_idx == 0) {
        for (int k = 0; k < kept; k++) {
            int32_t u = stab_new_unique(s, rec + 6 * k, crops + coff[k]);
            stab_log(s, u, rec[6 * k]);
        }
    } else {
        int32_t n_active_at_start = s->n_active; /* uniques born in this frame are not candidates */
        for (int k = 0; k < kept; k++) {
            const int32_t *r = rec + 6 * k;
            int32_t found = -1;
            for (int32_t a = 0; a < n_active_at_start; a++) {
                orc_unique *u = &s->uniq[s->active[a]];
                /* inclusive bbox overlap == half-open interval overlap on both axes */
                if (!(r[1] <= u->box[1] && u->box[0] <= r[2] && r[3] <= u->box[3] && u->box[2] <= r[4])) continue;
                s->tempo_count++;
                if (found >= 0) continue;
                int32_t cb[4] = { r[1], r[2], r[3], r[4] };
                int64_t match = orc_overlap(cb, crops + coff[k], u->box, u->crop);
                double recall = (double)match / (double)r[5];
                double precision = (double)match / (double)u->size;
                if (recall >= s->min_recall && precision >= s->min_precision) found = s->active[a];
            }
            if (found >= 0) {
                uniq_push_frame(&s->uniq[found], s->img_idx, r[0] + 1);
                s->uniq[found].last_frame = s->img_idx;
                stab_log(s, found, r[0]);
            } else {
                int32_t u = stab_new_unique(s, r, crops + coff[k]);
                stab_log(s, u, r[0]);
            }
        }
        /* retire */
        int32_t o = 0;
        for (int32_t a = 0; a < s->n_active; a++) {
            int32_t ui = s->active[a];
            if (s->img_idx - s->uniq[ui].last_frame >= s->max_gap) continue;
            s->active[o++] = ui;
        }
        s->n_active = o;
    }
    s->img_idx++;
    if (s->img_idx + 1 >= s->cap_frame_off) {
        s->cap_frame_off *= 2;
        s->frame_off = (int64_t *)realloc(s->frame_off, (size_t)s->cap_frame_off * sizeof(int64_t));
    }
    s->frame_off[s->img_idx] = s->n_log;
    free(rec); free(coff); free(crops);
    return kept;
}

/* accessors (plain copies so Python can rebuild the reference's list structures) */
int32_t orc_stab_n_unique(const orc_stab *s) { return s->n_uniq; }
int32_t orc_stab_n_frames(const orc_stab *s) { return s->img_idx; }
int64_t orc_stab_tempo_count(const orc_stab *s) { return s->tempo_count; }
int64_t orc_stab_n_log(const orc_stab *s) { return s->n_log; }
int32_t orc_stab_n_active(const orc_stab *s) { return s->n_active; }
void orc_stab_get_active(const orc_stab *s, int32_t *out) { memcpy(out, s->active, (size_t)s->n_active * 4); }
void orc_stab_get_log(const orc_stab *s, int32_t *pairs, int64_t *frame_off)
{
    memcpy(pairs, s->log, (size_t)s->n_log * 2 * sizeof(int32_t));
    memcpy(frame_off, s->frame_off, (size_t)(s->img_idx + 1) * sizeof(int64_t));
}
/* rec: 6 x int32 per unique: min_x, max_x, min_y, max_y, size, n_frames */
void orc_stab_get_unique_recs(const orc_stab *s, int32_t *rec)
{
    for (int32_t i = 0; i < s->n_uniq; i++) {
        const orc_unique *u = &s->uniq[i];
        int32_t *r = rec + 6 * (size_t)i;
        r[0] = u->box[0]; r[1] = u->box[1]; r[2] = u->box[2]; r[3] = u->box[3]; r[4] = u->size; r[5] = u->n_frames;
    }
}
void orc_stab_get_unique_frames(const orc_stab *s, int32_t idx, int32_t *pairs)
{
    memcpy(pairs, s->uniq[idx].frames, (size_t)s->uniq[idx].n_frames * 2 * sizeof(int32_t));
}
void orc_stab_get_unique_crop(const orc_stab *s, int32_t idx, uint8_t *out)
{
    const orc_unique *u = &s->uniq[idx];
    memcpy(out, u->crop, (size_t)(u->box[1] - u->box[0] + 1) * (size_t)(u->box[3] - u->box[2] + 1));
}
