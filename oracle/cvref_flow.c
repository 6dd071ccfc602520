/*
 * oracle/cvref_flow.c -- TEST INFRASTRUCTURE ONLY (see header of cvref.c).
 *
 * CPU restatement of the two OpenCV calls behind the reference's optical-flow
 * motion extraction:
 *
 *   cv2.goodFeaturesToTrack(img_u8, mask=None, maxCorners, qualityLevel,
 *                           minDistance, blockSize)            base.py:365-366
 *   cv2.calcOpticalFlowPyrLK(prev_u8, next_u8, pts, None, winSize, maxLevel,
 *                            criteria=(EPS|COUNT, n, eps))      base.py:371-372
 *
 * PARITY UNPINNED (no OpenCV under /root/reference; un-pinned dependency
 * "opencv3", README.md:12).  Semantics follow OpenCV 3.4 imgproc/featureselect
 * + video/lkpyramid (generic, non-SIMD code paths) as listed in SURVEY.md
 * Appendix B4/B5; anchored by the analytic tests in tests/test_oracle_flow.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static int refl101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* ------------------------------------------------------------------------- *
 * cornerMinEigenVal(src_u8, blockSize, ksize=3, BORDER_DEFAULT) -> float32
 * ------------------------------------------------------------------------- */
int rmo_corner_min_eigen_val(const uint8_t *img, int h, int w, int block_size, float *eig)
{
    size_t n = (size_t)h * w;
    float *dx = (float *)malloc(n * sizeof(float));
    float *dy = (float *)malloc(n * sizeof(float));
    float *cov = (float *)malloc(n * 3 * sizeof(float));
    double *rs = (double *)malloc(n * 3 * sizeof(double));
    if (!dx || !dy || !cov || !rs) { free(dx); free(dy); free(cov); free(rs); return -1; }

    double scale = (double)(1 << 2) * block_size * 255.0;
    scale = 1.0 / scale;
    /* Sobel(): the SMOOTHING kernel [1 2 1] carries the scale, as float32 */
    float k1 = (float)(1.0 * scale), k2 = (float)(2.0 * scale);

    /* Dx: row filter [-1 0 1] (u8 -> f32, accumulate left to right), then column
       filter scale*[1 2 1]: (S0 + S2)*f1 + S1*f0.                                  */
    /* Dy: row filter scale*[1 2 1] (generic row filter: k0*a; += k1*b; += k2*c),
       then column filter [-1 0 1]: S2 - S0.                                         */
    float *rowdx = (float *)malloc(n * sizeof(float));
    float *rowdy = (float *)malloc(n * sizeof(float));
    if (!rowdx || !rowdy) { free(dx); free(dy); free(cov); free(rs); free(rowdx); free(rowdy); return -1; }
    for (int y = 0; y < h; ++y) {
        const uint8_t *s = img + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            float a = (float)s[refl101(x - 1, w)], b = (float)s[x], c = (float)s[refl101(x + 1, w)];
            float t = -1.0f * a;
            t += 0.0f * b;
            t += 1.0f * c;
            rowdx[(size_t)y * w + x] = t;
            float u = k1 * a;
            u += k2 * b;
            u += k1 * c;
            rowdy[(size_t)y * w + x] = u;
        }
    }
    for (int y = 0; y < h; ++y) {
        int y0 = refl101(y - 1, h), y2 = refl101(y + 1, h);
        for (int x = 0; x < w; ++x) {
            float s0 = rowdx[(size_t)y0 * w + x], s1 = rowdx[(size_t)y * w + x], s2 = rowdx[(size_t)y2 * w + x];
            dx[(size_t)y * w + x] = (s0 + s2) * k1 + s1 * k2;
            dy[(size_t)y * w + x] = rowdy[(size_t)y2 * w + x] - rowdy[(size_t)y0 * w + x];
        }
    }
    free(rowdx); free(rowdy);
    for (size_t i = 0; i < n; ++i) {
        float a = dx[i], b = dy[i];
        cov[3 * i] = a * a; cov[3 * i + 1] = a * b; cov[3 * i + 2] = b * b;
    }
    /* boxFilter(normalize=false, ksize=block x block, anchor centre, BORDER_DEFAULT):
       row sums then column sums in double, cast to float */
    int r = block_size / 2; /* anchor = ksize/2 */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < block_size; ++k)
                    s += (double)cov[3 * ((size_t)y * w + refl101(x - r + k, w)) + c];
                rs[3 * ((size_t)y * w + x) + c] = s;
            }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float v[3];
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < block_size; ++k)
                    s += rs[3 * ((size_t)refl101(y - r + k, h) * w + x) + c];
                v[c] = (float)s;
            }
            float a = v[0] * 0.5f, b = v[1], c = v[2] * 0.5f;
            eig[(size_t)y * w + x] = (float)((a + c) - sqrtf((a - c) * (a - c) + b * b));
        }
    free(dx); free(dy); free(cov); free(rs);
    return 0;
}

typedef struct { float v; int idx; } cand_t;
static int cand_cmp(const void *pa, const void *pb)
{
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    /* OpenCV >= 3.4 greaterThanPtr: ties -> higher address (raster index) first */
    return (a->idx > b->idx) ? -1 : (a->idx < b->idx) ? 1 : 0;
}

/* Returns number of corners written to out_xy (float32 x,y pairs); 0 == cv2 returns None */
int rmo_good_features_to_track(const uint8_t *img, int h, int w, int max_corners,
                               double quality_level, double min_distance, int block_size,
                               float *out_xy, int out_cap)
{
    size_t n = (size_t)h * w;
    float *eig = (float *)malloc(n * sizeof(float));
    float *dil = (float *)malloc(n * sizeof(float));
    cand_t *cand = (cand_t *)malloc(n * sizeof(cand_t));
    if (!eig || !dil || !cand) { free(eig); free(dil); free(cand); return -1; }
    if (rmo_corner_min_eigen_val(img, h, w, block_size, eig) != 0) { free(eig); free(dil); free(cand); return -1; }
    double max_val = 0; /* minMaxLoc */
    {
        float m = eig[0];
        for (size_t i = 1; i < n; ++i) if (eig[i] > m) m = eig[i];
        max_val = (double)m;
    }
    float thr = (float)(max_val * quality_level);
    for (size_t i = 0; i < n; ++i) eig[i] = (eig[i] > thr) ? eig[i] : 0.0f; /* THRESH_TOZERO */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float m = -FLT_MAX;
            for (int yy = y - 1; yy <= y + 1; ++yy)
                for (int xx = x - 1; xx <= x + 1; ++xx)
                    if (yy >= 0 && yy < h && xx >= 0 && xx < w && eig[(size_t)yy * w + xx] > m)
                        m = eig[(size_t)yy * w + xx];
            dil[(size_t)y * w + x] = m;
        }
    int nc = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            float v = eig[(size_t)y * w + x];
            if (v != 0 && v == dil[(size_t)y * w + x]) { cand[nc].v = v; cand[nc].idx = y * w + x; ++nc; }
        }
    qsort(cand, (size_t)nc, sizeof(cand_t), cand_cmp);
    int ncorners = 0;
    if (min_distance >= 1) {
        float md2 = (float)(min_distance * min_distance);
        for (int i = 0; i < nc; ++i) {
            int y = cand[i].idx / w, x = cand[i].idx % w;
            int good = 1;
            for (int j = 0; j < ncorners; ++j) {
                float ddx = (float)x - out_xy[2 * j], ddy = (float)y - out_xy[2 * j + 1];
                if (ddx * ddx + ddy * ddy < md2) { good = 0; break; }
            }
            if (good) {
                if (ncorners >= out_cap) break;
                out_xy[2 * ncorners] = (float)x; out_xy[2 * ncorners + 1] = (float)y; ++ncorners;
                if (max_corners > 0 && ncorners == max_corners) break;
            }
        }
    } else {
        for (int i = 0; i < nc; ++i) {
            if (ncorners >= out_cap) break;
            out_xy[2 * ncorners] = (float)(cand[i].idx % w); out_xy[2 * ncorners + 1] = (float)(cand[i].idx / w); ++ncorners;
            if (max_corners > 0 && ncorners == max_corners) break;
        }
    }
    free(eig); free(dil); free(cand);
    return ncorners;
}

/* ------------------------------------------------------------------------- *
 * calcOpticalFlowPyrLK
 * ------------------------------------------------------------------------- */
/* uint8 pyrDown: integer 5-tap, (sum + 128) >> 8, BORDER_REFLECT_101 */
void rmo_pyr_down_u8(const uint8_t *src, int h, int w, uint8_t *dst)
{
    int dh = (h + 1) / 2, dw = (w + 1) / 2;
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            const uint8_t *s = src + (size_t)refl101(2 * y - 2 + k, h) * w;
            int *r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; ++x)
                r[x] = s[refl101(2 * x, w)] * 6 + (s[refl101(2 * x - 1, w)] + s[refl101(2 * x + 1, w)]) * 4
                     + s[refl101(2 * x - 2, w)] + s[refl101(2 * x + 2, w)];
        }
        for (int x = 0; x < dw; ++x) {
            int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 + rows[x] + rows[4 * dw + x];
            dst[(size_t)y * dw + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

/* calcSharrDeriv: int16 (Ix, Iy) interleaved, reflect-101 inside the image */
void rmo_scharr_deriv(const uint8_t *src, int h, int w, int16_t *d)
{
    int *t0 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
    int *t1 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * w;
        const uint8_t *r1 = src + (size_t)y * w;
        const uint8_t *r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * w;
        for (int x = 0; x < w; ++x) {
            t0[x + 1] = (int16_t)((r0[x] + r2[x]) * 3 + r1[x] * 10);
            t1[x + 1] = (int16_t)(r2[x] - r0[x]);
        }
        int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
        t0[0] = t0[x0 + 1]; t0[w + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[w + 1] = t1[x1 + 1];
        for (int x = 0; x < w; ++x) {
            d[2 * ((size_t)y * w + x)] = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * ((size_t)y * w + x) + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0); free(t1);
}

#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

typedef struct { int h, w; uint8_t *img; int16_t *deriv; } lvl_t;

static inline int px(const lvl_t *L, int y, int x) /* image + REFLECT_101 pad */
{
    return L->img[(size_t)refl101(y, L->h) * L->w + refl101(x, L->w)];
}
static inline int dv(const lvl_t *L, int y, int x, int c) /* derivative + zero pad */
{
    if (y < 0 || y >= L->h || x < 0 || x >= L->w) return 0;
    return L->deriv[2 * ((size_t)y * L->w + x) + c];
}

/* Number of pyramid levels OpenCV's buildOpticalFlowPyramid keeps (returns maxLevel used) */
int rmo_lk_max_level(int h, int w, int win_w, int win_h, int max_level)
{
    int sh = h, sw = w;
    for (int level = 0; level <= max_level; ++level) {
        sw = (sw + 1) / 2; sh = (sh + 1) / 2;
        if (sw <= win_w || sh <= win_h) return level;
    }
    return max_level;
}

/*
 * pts_in / pts_out: float32 (x, y) pairs; status: uint8 (1 tracked).
 * criteria: max_count, epsilon as given to cv2 (type EPS|COUNT).
 */
int rmo_calc_optical_flow_pyr_lk(const uint8_t *prev, const uint8_t *next, int h, int w,
                                 const float *pts_in, int npts, int win_w, int win_h,
                                 int max_level, int max_count, double epsilon,
                                 float *pts_out, uint8_t *status)
{
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (epsilon < 0) epsilon = 0;
    if (epsilon > 10) epsilon = 10;
    epsilon *= epsilon;
    const float min_eig_threshold = (float)1e-4;

    max_level = rmo_lk_max_level(h, w, win_w, win_h, max_level);
    lvl_t P[16], N[16];
    if (max_level > 15) return -1;
    {
        int sh = h, sw = w;
        for (int l = 0; l <= max_level; ++l) {
            P[l].h = N[l].h = sh; P[l].w = N[l].w = sw;
            P[l].img = (uint8_t *)malloc((size_t)sh * sw);
            N[l].img = (uint8_t *)malloc((size_t)sh * sw);
            P[l].deriv = (int16_t *)malloc((size_t)sh * sw * 2 * sizeof(int16_t));
            N[l].deriv = 0;
            if (l == 0) { memcpy(P[l].img, prev, (size_t)sh * sw); memcpy(N[l].img, next, (size_t)sh * sw); }
            else {
                rmo_pyr_down_u8(P[l - 1].img, P[l - 1].h, P[l - 1].w, P[l].img);
                rmo_pyr_down_u8(N[l - 1].img, N[l - 1].h, N[l - 1].w, N[l].img);
            }
            rmo_scharr_deriv(P[l].img, sh, sw, P[l].deriv);
            sh = (sh + 1) / 2; sw = (sw + 1) / 2;
        }
    }
    for (int i = 0; i < npts; ++i) status[i] = 1;

    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int W_BITS = 14;
    int16_t *Iwin = (int16_t *)malloc(sizeof(int16_t) * (size_t)win_w * win_h);
    int16_t *dIwin = (int16_t *)malloc(sizeof(int16_t) * (size_t)win_w * win_h * 2);

    for (int level = max_level; level >= 0; --level) {
        const lvl_t *I = &P[level], *J = &N[level];
        for (int p = 0; p < npts; ++p) {
            float sc = (float)(1. / (1 << level));
            float prev_x = pts_in[2 * p] * sc, prev_y = pts_in[2 * p + 1] * sc;
            float next_x, next_y;
            if (level == max_level) { next_x = prev_x; next_y = prev_y; }
            else { next_x = pts_out[2 * p] * 2.f; next_y = pts_out[2 * p + 1] * 2.f; }
            pts_out[2 * p] = next_x; pts_out[2 * p + 1] = next_y;

            prev_x -= half_x; prev_y -= half_y;
            int ipx = (int)floorf(prev_x), ipy = (int)floorf(prev_y);
            if (ipx < -win_w || ipx >= I->w || ipy < -win_h || ipy >= I->h) {
                if (level == 0) status[p] = 0;
                continue;
            }
            float a = prev_x - ipx, b = prev_y - ipy;
            int iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
            int iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < win_h; ++y)
                for (int x = 0; x < win_w; ++x) {
                    int yy = ipy + y, xx = ipx + x;
                    int ival = DESCALE(px(I, yy, xx) * iw00 + px(I, yy, xx + 1) * iw01 +
                                       px(I, yy + 1, xx) * iw10 + px(I, yy + 1, xx + 1) * iw11, W_BITS - 5);
                    int ixval = DESCALE(dv(I, yy, xx, 0) * iw00 + dv(I, yy, xx + 1, 0) * iw01 +
                                        dv(I, yy + 1, xx, 0) * iw10 + dv(I, yy + 1, xx + 1, 0) * iw11, W_BITS);
                    int iyval = DESCALE(dv(I, yy, xx, 1) * iw00 + dv(I, yy, xx + 1, 1) * iw01 +
                                        dv(I, yy + 1, xx, 1) * iw10 + dv(I, yy + 1, xx + 1, 1) * iw11, W_BITS);
                    Iwin[y * win_w + x] = (int16_t)ival;
                    dIwin[2 * (y * win_w + x)] = (int16_t)ixval;
                    dIwin[2 * (y * win_w + x) + 1] = (int16_t)iyval;
                    iA11 += (float)(ixval * ixval);
                    iA12 += (float)(ixval * iyval);
                    iA22 += (float)(iyval * iyval);
                }
            float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win_w * win_h);
            if (min_eig < min_eig_threshold || D < FLT_EPSILON) {
                if (level == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;
            next_x -= half_x; next_y -= half_y;
            float pdx = 0, pdy = 0;
            for (int j = 0; j < max_count; ++j) {
                int inx = (int)floorf(next_x), iny = (int)floorf(next_y);
                if (inx < -win_w || inx >= J->w || iny < -win_h || iny >= J->h) {
                    if (level == 0) status[p] = 0;
                    break;
                }
                a = next_x - inx; b = next_y - iny;
                iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
                iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float ib1 = 0, ib2 = 0;
                for (int y = 0; y < win_h; ++y)
                    for (int x = 0; x < win_w; ++x) {
                        int yy = iny + y, xx = inx + x;
                        int diff = DESCALE(px(J, yy, xx) * iw00 + px(J, yy, xx + 1) * iw01 +
                                           px(J, yy + 1, xx) * iw10 + px(J, yy + 1, xx + 1) * iw11, W_BITS - 5)
                                   - Iwin[y * win_w + x];
                        ib1 += (float)(diff * dIwin[2 * (y * win_w + x)]);
                        ib2 += (float)(diff * dIwin[2 * (y * win_w + x) + 1]);
                    }
                float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
                float dx = (float)((A12 * b2 - A22 * b1) * D);
                float dy = (float)((A12 * b1 - A11 * b2) * D);
                next_x += dx; next_y += dy;
                pts_out[2 * p] = next_x + half_x; pts_out[2 * p + 1] = next_y + half_y;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabs(dx + pdx) < 0.01 && fabs(dy + pdy) < 0.01) {
                    pts_out[2 * p] -= dx * 0.5f; pts_out[2 * p + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
        }
    }
    free(Iwin); free(dIwin);
    for (int l = 0; l <= max_level; ++l) { free(P[l].img); free(N[l].img); free(P[l].deriv); }
    return max_level;
}
