/*
 * oracle/cvref.c -- TEST INFRASTRUCTURE ONLY (never linked into, loaded by, or
 * called from the product package respmon_amd/).
 *
 * CPU restatement, in plain C, of the third-party OpenCV 3.x arithmetic that the
 * reference hot path reaches through `cv2` (the reference holds no native code of
 * its own; the algorithm lives in the un-vendored, un-pinned dependency
 * "opencv3" from conda channel menpo, reference README.md:12).  Call sites
 * restated here:
 *
 *   cv2.pyrDown              pyramid.py:14
 *   cv2.pyrUp(dstsize=)      pyramid.py:25-26, pyramid.py:55
 *   cv2.threshold            base.py:566
 *   cv2.findContours         base.py:568   (RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
 *   cv2.contourArea          base.py:571-572
 *   cv2.boundingRect         base.py:575
 *   cv2.goodFeaturesToTrack  base.py:365-366
 *   cv2.calcOpticalFlowPyrLK base.py:371-372
 *   cv2.cvtColor(BGR2GRAY)   base.py:230
 *
 * PARITY UNPINNED for everything in this file: no OpenCV source, binary, version
 * pin, test or golden vector exists under /root/reference and cv2 is not
 * installable in the build image.  The semantics follow OpenCV 3.4's published
 * imgproc/video algorithms and are anchored by the
 * analytic known-answer tests in tests/test_oracle_cv.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no FMA so every
 * double-precision operation rounds exactly once, like OpenCV's scalar path).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* OpenCV borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* ------------------------------------------------------------------------- *
 * cv2.pyrDown on CV_64F, default dstsize ((w+1)/2,(h+1)/2), BORDER_DEFAULT.
 * pyramid.py:14.  Horizontal pass first (per source row), vertical second,
 * operation order as in OpenCV's scalar PyrDown path.
 * ------------------------------------------------------------------------- */
static void pyr_down_hrow(const double *s, int w, double *r, int dw)
{
    for (int x = 0; x < dw; ++x) {
        int c = 2 * x;
        if (c - 2 >= 0 && c + 2 < w) {
            r[x] = s[c] * 6 + (s[c - 1] + s[c + 1]) * 4 + s[c - 2] + s[c + 2];
        } else {
            double m2 = s[reflect101(c - 2, w)], m1 = s[reflect101(c - 1, w)];
            double p1 = s[reflect101(c + 1, w)], p2 = s[reflect101(c + 2, w)];
            r[x] = s[reflect101(c, w)] * 6 + (m1 + p1) * 4 + m2 + p2;
        }
    }
}

int rmo_pyr_down(const double *src, int h, int w, double *dst)
{
    int dh = (h + 1) / 2, dw = (w + 1) / 2;
    double *rows = (double *)malloc(sizeof(double) * (size_t)dw * 5);
    if (!rows) return -1;
    for (int y = 0; y < dh; ++y) {
        double *rr[5];
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y - 2 + k, h);
            rr[k] = rows + (size_t)k * dw;
            pyr_down_hrow(src + (size_t)sy * w, w, rr[k], dw);
        }
        double *d = dst + (size_t)y * dw;
        for (int x = 0; x < dw; ++x)
            d[x] = (rr[2][x] * 6 + (rr[1][x] + rr[3][x]) * 4 + rr[0][x] + rr[4][x]) * (1.0 / 256);
    }
    free(rows);
    return 0;
}

/* ------------------------------------------------------------------------- *
 * cv2.pyrUp on CV_64F with explicit dstsize (pyramid.py:25-26, 55).
 * Requires |dw - 2*sw| == dw % 2 with dw <= 2*sw (the only case the pyramid
 * produces: sw == (dw+1)/2).  SURVEY App. B2.
 * ------------------------------------------------------------------------- */
static void pyr_up_hrow(const double *s, int sw, double *r /* 2*sw wide */)
{
    if (sw == 1) { r[0] = r[1] = s[0] * 8; return; }
    r[0] = s[0] * 6 + s[1] * 2;
    r[1] = (s[0] + s[1]) * 4;
    for (int x = 1; x < sw - 1; ++x) {
        r[2 * x] = s[x - 1] + s[x] * 6 + s[x + 1];
        r[2 * x + 1] = (s[x] + s[x + 1]) * 4;
    }
    r[2 * (sw - 1)] = s[sw - 2] + s[sw - 1] * 7;
    r[2 * (sw - 1) + 1] = s[sw - 1] * 8;
}

int rmo_pyr_up(const double *src, int sh, int sw, double *dst, int dh, int dw)
{
    if (!((dw == 2 * sw || dw == 2 * sw - 1) && (dh == 2 * sh || dh == 2 * sh - 1))) return -2;
    int rw = 2 * sw;
    double *rows = (double *)malloc(sizeof(double) * (size_t)rw * 3);
    if (!rows) return -1;
    for (int y = 0; y < sh; ++y) {
        /* source rows y-1 (reflect-101 at the top), y, y+1 (replicate at the bottom) */
        int y0 = (y == 0) ? (sh > 1 ? 1 : 0) : y - 1;
        int y2 = (y == sh - 1) ? sh - 1 : y + 1;
        double *r0 = rows, *r1 = rows + rw, *r2 = rows + 2 * rw;
        pyr_up_hrow(src + (size_t)y0 * sw, sw, r0);
        pyr_up_hrow(src + (size_t)y * sw, sw, r1);
        pyr_up_hrow(src + (size_t)y2 * sw, sw, r2);
        double *d0 = dst + (size_t)(2 * y) * dw;
        for (int x = 0; x < dw; ++x)
            d0[x] = (r0[x] + r1[x] * 6 + r2[x]) * (1.0 / 64);
        if (2 * y + 1 < dh) {
            double *d1 = dst + (size_t)(2 * y + 1) * dw;
            for (int x = 0; x < dw; ++x)
                d1[x] = ((r1[x] + r2[x]) * 4) * (1.0 / 64);
        }
    }
    free(rows);
    return 0;
}

/* ------------------------------------------------------------------------- *
 * cv2.threshold(img_u8, thresh, maxval, THRESH_BINARY)   base.py:566
 * ------------------------------------------------------------------------- */
void rmo_threshold_binary(const uint8_t *src, int n, int thresh, int maxval, uint8_t *dst)
{
    for (int i = 0; i < n; ++i) dst[i] = (src[i] > thresh) ? (uint8_t)maxval : 0;
}

/* ------------------------------------------------------------------------- *
 * cv2.findContours(img, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)   base.py:568
 * Suzuki-Abe border following as in OpenCV's cvFindNextContour/icvFetchContour,
 * on a zero-padded copy (OpenCV >= 3.2 rule: image-frame pixels count).
 * 8-connected foreground.  Working image is signed char: 0 background,
 * 1 unvisited foreground, 2 visited border pixel, 2|-128 visited border pixel
 * whose right neighbour was examined as background ("right exit").
 *
 * Output: contours in DISCOVERY (raster) order.  cv2 returns them in reverse
 * discovery order; the Python side of the oracle reverses the list.
 *   pts     : int32 pairs (x, y), concatenated
 *   offsets : n_contours+1 prefix offsets (in points)
 * Returns number of contours, or -1 if a capacity is exceeded / alloc fails.
 * ------------------------------------------------------------------------- */
static const int DX8[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };
static const int DY8[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };

int rmo_find_contours_external(const uint8_t *img, int h, int w, int approx_simple,
                               int32_t *pts, int pts_cap, int32_t *offsets, int off_cap)
{
    int step = w + 2, ph = h + 2;
    signed char *im = (signed char *)calloc((size_t)step * ph, 1);
    if (!im) return -1;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            im[(size_t)(y + 1) * step + x + 1] = img[(size_t)y * w + x] ? 1 : 0;
    int deltas[16];
    for (int i = 0; i < 8; ++i) deltas[i] = deltas[i + 8] = DY8[i] * step + DX8[i];

    int ncont = 0, npts = 0;
    const int nbd = 2;
    if (off_cap < 1) { free(im); return -1; }
    offsets[0] = 0;
    for (int y = 1; y < ph - 1; ++y) {
        signed char *row = im + (size_t)y * step;
        int prev = 0;
        int lnbd_x = 0; /* last border pixel met in this row; column 0 is the zero frame */
        for (int x = 1; x < step; ++x) {
            int p = row[x];
            if (p == prev) continue;
            int is_hole = 0;
            int take = 1;
            if (!(prev == 0 && p == 1)) {
                /* not the start of an outer border: maybe a hole border */
                if (p != 0 || prev < 1) take = 0;
                else is_hole = 1;
            }
            if (take && (is_hole || row[lnbd_x] > 0)) take = 0; /* RETR_EXTERNAL */
            if (take) {
                /* ---- icvFetchContour: trace the outer border starting at (x, y) ---- */
                signed char *i0 = row + x, *i1, *i3, *i4 = 0;
                int s, s_end, prev_s = -1;
                int cx = x - 1, cy = y - 1; /* unpadded coordinates */
                if (ncont + 1 >= off_cap) { free(im); return -1; }
                s_end = s = 4; /* outer border: start looking from the left neighbour */
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + deltas[s];
                } while (*i1 == 0 && s != s_end);
                if (s == s_end) {
                    /* isolated pixel */
                    *i0 = (signed char)(nbd | -128);
                    if (npts + 1 > pts_cap) { free(im); return -1; }
                    pts[2 * npts] = cx; pts[2 * npts + 1] = cy; ++npts;
                } else {
                    i3 = i0;
                    prev_s = s ^ 4;
                    for (;;) {
                        s_end = s;
                        for (;;) {
                            i4 = i3 + deltas[++s];
                            if (*i4 != 0) break;
                        }
                        s &= 7;
                        /* mark the pixel */
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(nbd | -128);
                        else if (*i3 == 1) *i3 = (signed char)nbd;
                        if (!approx_simple || s != prev_s) {
                            if (npts + 1 > pts_cap) { free(im); return -1; }
                            pts[2 * npts] = cx; pts[2 * npts + 1] = cy; ++npts;
                            prev_s = s;
                        }
                        cx += DX8[s]; cy += DY8[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                ++ncont;
                offsets[ncont] = npts;
                p = row[x];
            }
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
    free(im);
    return ncont;
}

/* cv2.contourArea(contour) (oriented=False): shoelace over the closed polygon,
 * accumulated in double as OpenCV does (a00 += xi_1*yi - xi*yi_1), base.py:571. */
double rmo_contour_area(const int32_t *pts, int n)
{
    if (n == 0) return 0.0;
    double a00 = 0;
    double px = (double)pts[2 * (n - 1)], py = (double)pts[2 * (n - 1) + 1];
    for (int i = 0; i < n; ++i) {
        double x = (double)pts[2 * i], y = (double)pts[2 * i + 1];
        a00 += px * y - x * py;
        px = x; py = y;
    }
    a00 *= 0.5;
    return fabs(a00);
}

/* cv2.boundingRect(int points): (minx, miny, maxx-minx+1, maxy-miny+1), base.py:575 */
void rmo_bounding_rect(const int32_t *pts, int n, int32_t *xywh)
{
    int minx = pts[0], maxx = pts[0], miny = pts[1], maxy = pts[1];
    for (int i = 1; i < n; ++i) {
        int x = pts[2 * i], y = pts[2 * i + 1];
        if (x < minx) minx = x;
        if (x > maxx) maxx = x;
        if (y < miny) miny = y;
        if (y > maxy) maxy = y;
    }
    xywh[0] = minx; xywh[1] = miny; xywh[2] = maxx - minx + 1; xywh[3] = maxy - miny + 1;
}

/* cv2.cvtColor(BGR2GRAY) on uint8: Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14  (base.py:230) */
void rmo_bgr2gray(const uint8_t *bgr, int n, uint8_t *gray)
{
    for (int i = 0; i < n; ++i)
        gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + 8192) >> 14);
}
