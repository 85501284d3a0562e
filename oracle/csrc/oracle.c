/*
 * oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code) of the integer / byte /
 * geometry stages of the reference hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path never does.
 *
 * Stages restated here, with the reference call site each one follows:
 *   orc_pil_resize_bilinear   torchvision Resize((640,640)) on a PIL image
 *                             (app/ml/models/text_detector.py:99-104,124) == Pillow 10.1
 *                             ImagingResample, 8-bit, BILINEAR with antialias
 *                             [pinned: bit-exact against the PIL installed in this container,
 *                              tests/test_oracle_preprocess.py]
 *   orc_cv_resize_linear_u8   cv2.resize(img,(128,32)) (app/ml/models/text_recognizer.py:118)
 *                             == OpenCV 4.8.1 resize INTER_LINEAR, 8-bit fixed point
 *                             [PARITY UNPINNED: cv2 is not installed here and the reference holds
 *                              no golden for it; restated from the published algorithm]
 *   orc_postprocess           TextDetector._post_process (text_detector.py:143-178):
 *                             threshold -> cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
 *                             -> contourArea -> minAreaRect -> boxPoints -> np.int0 -> bbox maths
 *                             [PARITY UNPINNED for the OpenCV 4.8.1 pieces: Suzuki-Abe border
 *                              following, convexHull, rotatingCalipers and RotatedRect::points are
 *                              restated from the published algorithms; bbox arithmetic and the
 *                              filters follow the reference lines literally]
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/build.py).  Float
 * arithmetic order matters for the calipers; contraction is off on purpose.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------ Pillow resample
 * Restates the 8-bit path of Pillow's src/libImaging/Resample.c (precompute_coeffs / normalize_coeffs_8bpc /
 * ImagingResampleHorizontal_8bpc / ...Vertical_8bpc; Pillow is HPND-licensed, (c) Secret Labs AB / Fredrik Lundh / Alex Clark
 * and contributors): same coefficient formula, same 22-bit fixed point, same rounding, so that it can be pinned bit-exact
 * against the PIL installed here (tests/test_oracle_stages.py, tests/test_oracle_trocr.py). */

#define PIL_PRECISION_BITS (32 - 8 - 2)

static double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

/* coefficient table for one axis; returns ksize, fills bounds[2*out] and kk[out*ksize] (int) */
static int pil_coeffs(int in_size, int out_size, int **bounds_out, int **kk_out) {
    double scale = (double)in_size / (double)out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    double support = 1.0 * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    int *bounds = (int *)malloc(sizeof(int) * 2 * out_size);
    int *kk = (int *)calloc((size_t)out_size * ksize, sizeof(int));
    double *pre = (double *)malloc(sizeof(double) * ksize);
    for (int xx = 0; xx < out_size; xx++) {
        double center = (xx + 0.5) * scale;
        double ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; x++) {
            double w = bilinear_filter((x + xmin - center + 0.5) * ss);
            pre[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; x++) {
            double v = (ww != 0.0) ? pre[x] / ww : pre[x];
            kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PIL_PRECISION_BITS))
                                               : (int)(0.5 + v * (1 << PIL_PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(pre);
    *bounds_out = bounds;
    *kk_out = kk;
    return ksize;
}

static inline uint8_t clip8(int v) {
    v >>= PIL_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* src: h x w x 3 (row stride = src_stride bytes); dst: oh x ow x 3 contiguous.  Horizontal pass
 * first into a uint8 intermediate, then vertical (Pillow's order). */
int orc_pil_resize_bilinear(const uint8_t *src, int h, int w, long src_stride, uint8_t *dst, int oh, int ow) {
    int *bx, *kx, *by, *ky;
    int ksx = pil_coeffs(w, ow, &bx, &kx);
    int ksy = pil_coeffs(h, oh, &by, &ky);
    uint8_t *tmp = (uint8_t *)malloc((size_t)h * ow * 3);
    for (int y = 0; y < h; y++) {
        const uint8_t *row = src + (size_t)y * src_stride;
        for (int xx = 0; xx < ow; xx++) {
            int xmin = bx[2 * xx], n = bx[2 * xx + 1];
            const int *k = kx + (size_t)xx * ksx;
            for (int c = 0; c < 3; c++) {
                int ss = 1 << (PIL_PRECISION_BITS - 1);
                for (int x = 0; x < n; x++) ss += row[(x + xmin) * 3 + c] * k[x];
                tmp[((size_t)y * ow + xx) * 3 + c] = clip8(ss);
            }
        }
    }
    for (int yy = 0; yy < oh; yy++) {
        int ymin = by[2 * yy], n = by[2 * yy + 1];
        const int *k = ky + (size_t)yy * ksy;
        for (int xx = 0; xx < ow * 3; xx++) {
            int ss = 1 << (PIL_PRECISION_BITS - 1);
            for (int y = 0; y < n; y++) ss += tmp[(size_t)(y + ymin) * ow * 3 + xx] * k[y];
            dst[(size_t)yy * ow * 3 + xx] = clip8(ss);
        }
    }
    free(tmp); free(bx); free(kx); free(by); free(ky);
    return 0;
}

/* ------------------------------------------------------------------ cv2.resize INTER_LINEAR, 8U */

static inline short sat_short_round(float v) {
    long r = lrintf(v); /* round-half-even like cvRound */
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

static void cv_linear_axis(int ssize, int dsize, int *ofs, short *coef /* 2 per dst */) {
    double inv_scale = (double)dsize / ssize;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        coef[2 * d] = sat_short_round((1.f - f) * 2048.f);
        coef[2 * d + 1] = sat_short_round(f * 2048.f);
    }
}

/* src: sh x sw x 3 with byte row stride; dst: dh x dw x 3 contiguous */
int orc_cv_resize_linear_u8(const uint8_t *src, int sh, int sw, long src_stride, uint8_t *dst, int dh, int dw) {
    if (sh <= 0 || sw <= 0) return -1;
    if (sw == 2 * dw && sh == 2 * dh) { /* exact 2x2 decimation is routed to the area kernel */
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < 3; c++) {
                    const uint8_t *r0 = src + (size_t)(2 * y) * src_stride + (2 * x) * 3 + c;
                    const uint8_t *r1 = r0 + src_stride;
                    dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)((r0[0] + r0[3] + r1[0] + r1[3] + 2) >> 2);
                }
        return 0;
    }
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    short *alpha = (short *)malloc(sizeof(short) * 2 * dw), *beta = (short *)malloc(sizeof(short) * 2 * dh);
    cv_linear_axis(sw, dw, xofs, alpha);
    cv_linear_axis(sh, dh, yofs, beta);
    for (int y = 0; y < dh; y++) {
        int sy0 = yofs[y], sy1 = sy0 + 1 < sh ? sy0 + 1 : sh - 1;
        const uint8_t *r0 = src + (size_t)sy0 * src_stride, *r1 = src + (size_t)sy1 * src_stride;
        int b0 = beta[2 * y], b1 = beta[2 * y + 1];
        for (int x = 0; x < dw; x++) {
            int sx0 = xofs[x], sx1 = sx0 + 1 < sw ? sx0 + 1 : sw - 1;
            int a0 = alpha[2 * x], a1 = alpha[2 * x + 1];
            for (int c = 0; c < 3; c++) {
                int h0 = r0[sx0 * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
                int h1 = r1[sx0 * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
                int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
    free(xofs); free(yofs); free(alpha); free(beta);
    return 0;
}

/* ------------------------------------------------------------------ post-process */

typedef struct { int x, y; } ipt;
typedef struct { float x, y; } fpt;

typedef struct {
    int bbox[4];      /* x1,y1,x2,y2 in frame pixels */
    int poly[8];      /* 4 corners (x,y), map space, truncated */
    float conf;       /* mean of prob over the re-projected box */
    float area;       /* contourArea of the traced border (diagnostic) */
    int first_x, first_y; /* raster-first pixel of the component (diagnostic / ordering) */
} orc_det;

typedef struct { ipt *p; int n, cap; } ptvec;
static void pv_push(ptvec *v, int x, int y) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->p = (ipt *)realloc(v->p, sizeof(ipt) * v->cap); }
    v->p[v->n].x = x; v->p[v->n].y = y; v->n++;
}

static const int CODE_DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int CODE_DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

/* Suzuki-Abe outer-border following from pixel (x,y) of the padded image `img` (row stride `step`),
 * marking visited border pixels 2, or -126 where the border's right-hand neighbour is background.
 * CHAIN_APPROX_SIMPLE: a point is emitted only where the chain direction changes. */
static void trace_outer(int8_t *img, int step, int x, int y, ptvec *out) {
    int deltas[16];
    for (int i = 0; i < 8; i++) deltas[i] = CODE_DY[i] * step + CODE_DX[i];
    memcpy(deltas + 8, deltas, 8 * sizeof(int));
    int8_t *i0 = img + (size_t)y * step + x, *i1, *i3, *i4 = 0;
    int s = 4, s_end = 4, prev_s;
    ipt pt = {x, y};
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);
    if (s == s_end) { /* isolated pixel */
        *i0 = (int8_t)-126;
        pv_push(out, pt.x, pt.y);
        return;
    }
    i3 = i0;
    prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (int8_t)-126;
        else if (*i3 == 1) *i3 = 2;
        if (s != prev_s) { pv_push(out, pt.x, pt.y); prev_s = s; }
        pt.x += CODE_DX[s];
        pt.y += CODE_DY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

static int cmp_ipt(const void *a, const void *b) {
    const ipt *p = (const ipt *)a, *q = (const ipt *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}
static long long cross3(ipt o, ipt a, ipt b) {
    return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
}

/* Strict convex hull (no collinear vertices).  Output order follows what cv::convexHull(clockwise=false)
 * hands to rotatingCalipers for a Suzuki-traced outer contour: on screen (y down) the vertices run
 * clockwise, the sequence ENDS at the contour's first point (raster-first pixel: min y, then min x). */
static int convex_hull(const ipt *pts, int n, ipt *hull /* cap >= n+1 */) {
    ipt *s = (ipt *)malloc(sizeof(ipt) * n);
    memcpy(s, pts, sizeof(ipt) * n);
    qsort(s, n, sizeof(ipt), cmp_ipt);
    int m = 0;
    for (int i = 0; i < n; i++) if (i == 0 || cmp_ipt(&s[i], &s[m - 1]) != 0) s[m++] = s[i];
    if (m < 3) { memcpy(hull, s, sizeof(ipt) * m); free(s); return m; }
    ipt *h = (ipt *)malloc(sizeof(ipt) * (2 * m + 2));
    int k = 0;
    for (int i = 0; i < m; i++) { /* lower chain in (x,y): with y down this is the screen-top chain */
        while (k >= 2 && cross3(h[k - 2], h[k - 1], s[i]) <= 0) k--;
        h[k++] = s[i];
    }
    for (int i = m - 2, t = k + 1; i >= 0; i--) {
        while (k >= t && cross3(h[k - 2], h[k - 1], s[i]) <= 0) k--;
        h[k++] = s[i];
    }
    k--; /* last == first */
    /* Andrew's chain with cross>0 kept is counter-clockwise in a y-up frame == clockwise on screen */
    int start = 0;
    for (int i = 1; i < k; i++)
        if (h[i].y < h[start].y || (h[i].y == h[start].y && h[i].x < h[start].x)) start = i;
    for (int i = 0; i < k; i++) hull[i] = h[(start + 1 + i) % k];
    free(h); free(s);
    return k;
}

/* rotating calipers, minimum-area rectangle; float32 arithmetic in the published order */
static void min_area_rect(const fpt *points, int n, float out[6]) {
    float minarea = FLT_MAX;
    int seq[4] = {-1, -1, -1, -1};
    int left = 0, bottom = 0, right = 0, top = 0;
    float *inv_len = (float *)malloc(sizeof(float) * n);
    fpt *vect = (fpt *)malloc(sizeof(fpt) * n);
    float orientation = 0, base_a, base_b = 0;
    float left_x, right_x, top_y, bottom_y;
    fpt pt0 = points[0];
    left_x = right_x = pt0.x;
    top_y = bottom_y = pt0.y;
    for (int i = 0; i < n; i++) {
        if (pt0.x < left_x) left_x = pt0.x, left = i;
        if (pt0.x > right_x) right_x = pt0.x, right = i;
        if (pt0.y > top_y) top_y = pt0.y, top = i;
        if (pt0.y < bottom_y) bottom_y = pt0.y, bottom = i;
        fpt pt = points[(i + 1 < n) ? i + 1 : 0];
        double dx = (double)pt.x - (double)pt0.x, dy = (double)pt.y - (double)pt0.y;
        vect[i].x = (float)dx;
        vect[i].y = (float)dy;
        inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
        pt0 = pt;
    }
    {
        double ax = vect[n - 1].x, ay = vect[n - 1].y;
        for (int i = 0; i < n; i++) {
            double bx = vect[i].x, by = vect[i].y;
            double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : -1.f; break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
    int best_left = 0, best_bottom = 0;
    float best_a = 0, best_b = 0, best_w = 0, best_h = 0;
    for (int k = 0; k < n; k++) {
        float dp[4] = {
            +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y,
            -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y,
            -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y,
            +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y,
        };
        float maxcos = dp[0] * inv_len[seq[0]];
        int main_element = 0;
        for (int i = 1; i < 4; i++) {
            float cosalpha = dp[i] * inv_len[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            int pindex = seq[main_element];
            float lead_x = vect[pindex].x * inv_len[pindex];
            float lead_y = vect[pindex].y * inv_len[pindex];
            switch (main_element) {
            case 0: base_a = lead_x; base_b = lead_y; break;
            case 1: base_a = lead_y; base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        seq[main_element] += 1;
        seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
        {
            float dx = points[seq[1]].x - points[seq[3]].x;
            float dy = points[seq[1]].y - points[seq[3]].y;
            float width = dx * base_a + dy * base_b;
            dx = points[seq[2]].x - points[seq[0]].x;
            dy = points[seq[2]].y - points[seq[0]].y;
            float height = -dx * base_b + dy * base_a;
            float area = width * height;
            if (area <= minarea) {
                minarea = area;
                best_left = seq[3]; best_a = base_a; best_w = width;
                best_b = base_b; best_h = height; best_bottom = seq[0];
            }
        }
    }
    {
        float A1 = best_a, B1 = best_b, A2 = -best_b, B2 = best_a;
        float C1 = A1 * points[best_left].x + points[best_left].y * B1;
        float C2 = A2 * points[best_bottom].x + points[best_bottom].y * B2;
        float idet = 1.f / (A1 * B2 - A2 * B1);
        float px = (C1 * B2 - C2 * B1) * idet;
        float py = (A1 * C2 - A2 * C1) * idet;
        out[0] = px; out[1] = py;
        out[2] = A1 * best_w; out[3] = B1 * best_w;
        out[4] = A2 * best_h; out[5] = B2 * best_h;
    }
    free(inv_len); free(vect);
}

#define ORC_PI 3.1415926535897932384626433832795

/* minAreaRect + boxPoints + np.int0 on an int contour; box[8] = 4 x (x,y).
 *
 * Why an axis-aligned component whose contour spans x = 0..159 comes out as [0,158] and not [0,159] (tests/test_gpu_e2e_detector.py
 * feeds the reference's own 160x160 all-foreground map): nothing here is exact.  rotatingCalipers normalises every hull edge with
 * a float32 reciprocal length: for the edge (159, 0) that is inv_len = float32(1/159) = 0.0062893084 and the unit vector's x becomes
 * 159 * inv_len = 0.99999994 (one ulp below 1).  The rectangle's width is the projection of the opposite corner on that vector,
 * 159 * 0.99999994 = 158.99998 in float32, its centre 79.49999, and RotatedRect::points() rebuilds the far corner as
 * centre + 0.5 * width = 158.99998.  np.int0 truncates toward zero -> 158.  The near corner is rebuilt as 0.0 (or -4.9e-15) -> 0.
 * OpenCV 4.8.1's C++ performs the same float32 operations in the same order (modules/imgproc/src/rotcalipers.cpp,
 * types.cpp RotatedRect::points), so the same one-pixel shrink is expected from it, but with cv2 not importable here that stays
 * PARITY UNPINNED: the HIP post-process is bit-exact against THIS restatement (csrc/postprocess.hip replays it with
 * -ffp-contract=off), not against a run of OpenCV. */
static void min_area_box(const ipt *contour, int npts, int box[8]) {
    ipt *hull = (ipt *)malloc(sizeof(ipt) * (npts + 1));
    int n = convex_hull(contour, npts, hull);
    float cx = 0, cy = 0, w = 0, hgt = 0, angle = 0;
    if (n > 2) {
        fpt *hp = (fpt *)malloc(sizeof(fpt) * n);
        for (int i = 0; i < n; i++) { hp[i].x = (float)hull[i].x; hp[i].y = (float)hull[i].y; }
        float o[6];
        min_area_rect(hp, n, o);
        cx = o[0] + (o[2] + o[4]) * 0.5f;
        cy = o[1] + (o[3] + o[5]) * 0.5f;
        w = (float)sqrt((double)o[2] * o[2] + (double)o[3] * o[3]);
        hgt = (float)sqrt((double)o[4] * o[4] + (double)o[5] * o[5]);
        angle = (float)atan2((double)o[3], (double)o[2]);
        free(hp);
    } else if (n == 2) {
        cx = ((float)hull[0].x + (float)hull[1].x) * 0.5f;
        cy = ((float)hull[0].y + (float)hull[1].y) * 0.5f;
        double dx = (double)hull[1].x - hull[0].x, dy = (double)hull[1].y - hull[0].y;
        w = (float)sqrt(dx * dx + dy * dy);
        hgt = 0;
        angle = (float)atan2(dy, dx);
    } else if (n == 1) {
        cx = (float)hull[0].x; cy = (float)hull[0].y;
    }
    angle = (float)((double)(angle * 180.f) / ORC_PI);
    double rad = (double)angle * ORC_PI / 180.;
    float b = (float)cos(rad) * 0.5f;
    float a = (float)sin(rad) * 0.5f;
    float p[8];
    p[0] = cx - a * hgt - b * w;
    p[1] = cy + b * hgt - a * w;
    p[2] = cx + a * hgt - b * w;
    p[3] = cy - b * hgt - a * w;
    p[4] = 2 * cx - p[0];
    p[5] = 2 * cy - p[1];
    p[6] = 2 * cx - p[2];
    p[7] = 2 * cy - p[3];
    for (int i = 0; i < 8; i++) box[i] = (int)p[i]; /* np.int0: truncate toward zero */
    free(hull);
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* prob: h x w float32.  Returns number of detections written (<= max_out), in the order
 * cv2.findContours returns contours in 4.8.1 (reverse raster discovery order).  If more than max_out
 * detections exist the surplus (earliest discovered) ones are dropped and the return value is the
 * total count. */
int orc_postprocess(const float *prob, int h, int w, int orig_w, int orig_h, float thr, orc_det *out, int max_out) {
    int step = w + 2;
    int8_t *img = (int8_t *)calloc((size_t)(h + 2) * step, 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * step + x + 1] = prob[(size_t)y * w + x] > thr ? 1 : 0;
    orc_det *found = NULL;
    int nfound = 0, cap = 0;
    ptvec contour = {0, 0, 0};
    for (int y = 1; y <= h; y++) {
        int8_t *row = img + (size_t)y * step;
        int prev = 0, lnbd_x = 0;
        for (int x = 1; x <= w + 1; x++) {
            int p = row[x];
            if (p == prev) continue;
            if (prev == 0 && p == 1 && !(row[lnbd_x] > 0)) {
                contour.n = 0;
                trace_outer(img, step, x, y, &contour);
                /* contourArea: exact in double for lattice points */
                double a00 = 0;
                ipt pv = contour.p[contour.n - 1];
                for (int i = 0; i < contour.n; i++) {
                    ipt q = contour.p[i];
                    a00 += (double)pv.x * q.y - (double)pv.y * q.x;
                    pv = q;
                }
                double area = fabs(a00 * 0.5);
                if (!(area < 100)) {
                    for (int i = 0; i < contour.n; i++) { contour.p[i].x -= 1; contour.p[i].y -= 1; } /* unpad */
                    int box[8];
                    min_area_box(contour.p, contour.n, box);
                    int xmin = box[0], xmax = box[0], ymin = box[1], ymax = box[1];
                    for (int i = 1; i < 4; i++) {
                        xmin = imin(xmin, box[2 * i]); xmax = imax(xmax, box[2 * i]);
                        ymin = imin(ymin, box[2 * i + 1]); ymax = imax(ymax, box[2 * i + 1]);
                    }
                    /* text_detector.py:160-166: clamp to the literal 640, scale with int(v*W/640) */
                    long x1 = imax(0, xmin), y1 = imax(0, ymin), x2 = imin(640, xmax), y2 = imin(640, ymax);
                    x1 = (long)((double)(x1 * orig_w) / 640.0);
                    y1 = (long)((double)(y1 * orig_h) / 640.0);
                    x2 = (long)((double)(x2 * orig_w) / 640.0);
                    y2 = (long)((double)(y2 * orig_h) / 640.0);
                    if (x2 - x1 > 10 && y2 - y1 > 10) {
                        /* text_detector.py:169-170: numpy slice semantics (bounds clamp, empty -> nan) */
                        long sy0 = y1 * 640 / orig_h, sy1 = y2 * 640 / orig_h;
                        long sx0 = x1 * 640 / orig_w, sx1 = x2 * 640 / orig_w;
                        if (sy0 > h) sy0 = h; if (sy1 > h) sy1 = h;
                        if (sx0 > w) sx0 = w; if (sx1 > w) sx1 = w;
                        double acc = 0; long cnt = 0;
                        for (long yy = sy0; yy < sy1; yy++)
                            for (long xx = sx0; xx < sx1; xx++) { acc += prob[(size_t)yy * w + xx]; cnt++; }
                        if (nfound == cap) { cap = cap ? cap * 2 : 64; found = (orc_det *)realloc(found, sizeof(orc_det) * cap); }
                        orc_det *d = &found[nfound++];
                        d->bbox[0] = (int)x1; d->bbox[1] = (int)y1; d->bbox[2] = (int)x2; d->bbox[3] = (int)y2;
                        memcpy(d->poly, box, sizeof(box));
                        d->conf = cnt ? (float)(acc / (double)cnt) : NAN;
                        d->area = (float)area;
                        d->first_x = x - 1; d->first_y = y - 1;
                    }
                }
                p = row[x];
            }
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
    int nout = nfound < max_out ? nfound : max_out;
    for (int i = 0; i < nout; i++) out[i] = found[nfound - 1 - i];
    free(found); free(contour.p); free(img);
    return nfound;
}

/* Exposed pieces for unit tests */
int orc_min_area_box(const int *pts_xy, int npts, int *box8) {
    min_area_box((const ipt *)pts_xy, npts, box8);
    return 0;
}
