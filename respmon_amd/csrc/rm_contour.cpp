// respmon_amd/csrc/rm_contour.cpp -- host stage of locate(): base.py:568-575
//   cv2.findContours(thresh, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) -> max(contours, key=cv2.contourArea)
//   -> cv2.boundingRect
//
// The thresholded image is a few MB and the border following is inherently serial, so this
// stage runs on the host over the binary image the GPU wrote to pinned memory.  Only what
// locate() consumes is produced: for every external border the shoelace area and the bounding
// box are accumulated while the border is followed (no point lists; CHAIN_APPROX_SIMPLE only
// removes collinear points, which changes neither quantity: all terms are exact integers).
//
// Border following is Suzuki-Abe as used by OpenCV (8-connected foreground, outer borders
// only, components nested inside a hole are skipped), on a zero-framed working copy so that
// image-frame pixels count (OpenCV >= 3.2 rule).  Selection rule: cv2 lists
// contours in reverse discovery order and Python's max() keeps the first maximum, i.e. among
// equal areas the LAST discovered border wins.
#include "rm_contour.h"

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

namespace rm {

namespace {
enum : signed char { BG = 0, FG = 1, SEEN = 2, SEEN_RIGHT_EXIT = (signed char)(2 | -128) };

struct Tracer {
    std::vector<signed char> buf;
    int step = 0;
    int nbr[16];

    int rows_dirty_h = 0;            // geometry of the previous image (the buffer is kept between calls)
    std::vector<char> dirty;         // rows of the working copy that are not all-background

    void prepare(const uint8_t *bin, int H, int W, const uint32_t *row_any)
    {
        const int new_step = W + 2;
        if (new_step != step || rows_dirty_h != H) {
            step = new_step; rows_dirty_h = H;
            buf.assign((size_t)step * (H + 2), BG);
            dirty.assign(H, 0);
        }
        for (int y = 0; y < H; ++y) {
            signed char *d = buf.data() + (size_t)(y + 1) * step + 1;
            const bool any = row_any ? row_any[y] != 0 : true;
            if (!any) {
                if (dirty[y]) { std::memset(d, BG, (size_t)W); dirty[y] = 0; }  // only rows the last image touched
                continue;
            }
            dirty[y] = 1;
            if ((int)clr_x0.size() == H) { clr_x0[y] = 0; clr_x1[y] = W - 1; }
            const uint8_t *s = bin + (size_t)y * W;
            int x = 0;
            // foreground = non-zero byte, 8 pixels per step: bit 7 of ((b & 0x7f) + 0x7f) | b is set iff b != 0
            for (; x + 8 <= W; x += 8) {
                unsigned long long v;
                std::memcpy(&v, s + x, 8);
                v = ((((v & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | v) & 0x8080808080808080ull) >> 7;
                std::memcpy(d + x, &v, 8);
            }
            for (; x < W; ++x) d[x] = s[x] ? FG : BG;
        }
        win_y0 = 0; win_y1 = H - 1;   // (this entry keeps no row window: a later prepare_bits() must visit every row)
        const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
        const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
        for (int i = 0; i < 8; ++i) nbr[i] = nbr[i + 8] = dy[i] * step + dx[i];
    }

    std::vector<uint32_t> own_row_any;
    std::vector<int> row_x0, row_x1;   // first / last foreground column of each dirty row (image coordinates)
    std::vector<int> clr_x0, clr_x1;   // column range of a dirty row that is not background
    std::vector<int> grp_a, grp_b;     // first / last 64-pixel group of a row that holds foreground

    // 64 pixels starting at bit position `pos` of the packed image (bits past `end` read as background)
    static inline uint64_t get64(const uint64_t *bits, size_t pos, size_t end)
    {
        const size_t wi = pos >> 6, sh = pos & 63, last = (end - 1) >> 6;
        uint64_t v = bits[wi] >> sh;
        if (sh && wi + 1 <= last) v |= bits[wi + 1] << (64 - sh);
        const size_t left = end - pos;
        if (left < 64) v &= (~0ull) >> (64 - left);
        return v;
    }

    int win_y0 = 0, win_y1 = -1;     // rows the previous prepare_bits() found foreground in (they hold marks to clear)

    // rows outside [ya, yb] are known to hold no foreground in `bits`; rows the previous image dirtied are visited as well
    void prepare_bits(const uint64_t *bits, int H, int W, int ya = 0, int yb = -2)
    {
        if (yb == -2) yb = H - 1;
        static uint64_t lut[256];
        static bool lut_ok = false;
        if (!lut_ok) {  // byte of 8 pixel bits -> 8 bytes FG / BG
            for (int b = 0; b < 256; ++b) {
                uint64_t v = 0;
                for (int k = 0; k < 8; ++k) v |= (uint64_t)((b >> k) & 1) << (8 * k);
                lut[b] = v;
            }
            lut_ok = true;
        }
        const int new_step = W + 2;
        if (new_step != step || rows_dirty_h != H) {
            step = new_step; rows_dirty_h = H;
            buf.assign((size_t)step * (H + 2), BG);
            dirty.assign(H, 0);
            win_y0 = 0; win_y1 = -1;
        }
        if ((int)clr_x0.size() != H) { clr_x0.assign(H, 0); clr_x1.assign(H, W - 1); }
        own_row_any.assign(H, 0);
        row_x0.assign(H, 0); row_x1.assign(H, -1);
        grp_a.assign(H, 0); grp_b.assign(H, -1);
        if (ya < 0) ya = 0;
        if (yb > H - 1) yb = H - 1;
        if (win_y1 >= win_y0) {   // marks of the previous image
            if (yb < ya) { ya = win_y0; yb = win_y1; }
            else { ya = ya < win_y0 ? ya : win_y0; yb = yb > win_y1 ? yb : win_y1; }
        }
        win_y0 = H; win_y1 = -1;
        for (int y = ya; y <= yb; ++y) {
            signed char *d = buf.data() + (size_t)(y + 1) * step + 1;
            const size_t r0 = (size_t)y * W;
            // the part of the working row the previous image (and its border marks) touched is cleared; only the
            // 64-pixel groups that hold foreground are written afterwards
            if (dirty[y]) { std::memset(d + clr_x0[y], BG, (size_t)(clr_x1[y] - clr_x0[y] + 1)); dirty[y] = 0; }
            // most rows are empty: test the row's words in place (edge words masked) before any unpacking
            const size_t b0 = r0, b1 = r0 + W - 1, w0 = b0 >> 6, w1 = b1 >> 6;
            const uint64_t m0 = ~0ull << (b0 & 63), m1 = ~0ull >> (63 - (b1 & 63));
            size_t wf = w1 + 1, wl = w0;   // first / last word of the row that holds foreground
            if (w0 == w1) { if (bits[w0] & m0 & m1) { wf = w0; wl = w0; } }
            else {
                if (bits[w0] & m0) { wf = w0; wl = w0; }
                for (size_t wq = w0 + 1; wq < w1; ++wq)
                    if (bits[wq]) { if (wf > w1) wf = wq; wl = wq; }
                if (bits[w1] & m1) { if (wf > w1) wf = w1; wl = w1; }
            }
            if (wf > w1) continue;
            // 64-pixel groups of this row (group k = columns 64k..64k+63) that can hold those words
            const int xa = (int)(((wf << 6) > b0 ? (wf << 6) - b0 : 0) / 64) * 64;
            const size_t last_bit = ((wl << 6) + 63 < b1 ? (wl << 6) + 63 : b1) - b0;
            const int xb = (int)(last_bit / 64) * 64;
            int x0 = -1, x1 = -1;
            for (int x = xa; x <= xb; x += 64) {
                const uint64_t v = get64(bits, r0 + x, r0 + W);
                if (!v) continue;
                if (x0 < 0) x0 = x + __builtin_ctzll(v);
                x1 = x + 63 - __builtin_clzll(v);
                const int n = (W - x < 64) ? W - x : 64;
                if (n == 64) {
                    for (int k = 0; k < 8; ++k) { const uint64_t e = lut[(v >> (8 * k)) & 0xff]; std::memcpy(d + x + 8 * k, &e, 8); }
                } else {
                    for (int k = 0; k < n; ++k) d[x + k] = (signed char)((v >> k) & 1);
                }
            }
            if (x0 < 0) continue;
            dirty[y] = 1; own_row_any[y] = 1; row_x0[y] = x0; row_x1[y] = x1; grp_a[y] = xa; grp_b[y] = xb;
            if (y < win_y0) win_y0 = y;
            win_y1 = y;
            clr_x0[y] = x0 & ~63; clr_x1[y] = ((x1 | 63) < W - 1) ? (x1 | 63) : W - 1;   // whole groups were written
        }
        const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
        const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
        for (int i = 0; i < 8; ++i) nbr[i] = nbr[i + 8] = dy[i] * step + dx[i];
    }

    long long steps = 0;   // border steps walked since the entry point reset it
    // follow the outer border that starts at `start` (padded coords px,py); returns twice the signed area
    long long follow(signed char *start, int px, int py, int &minx, int &miny, int &maxx, int &maxy)
    {
        static const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
        static const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
        minx = maxx = px; miny = maxy = py;
        int dir = 4, stop = 4;
        signed char *first_nb;
        do {  // clockwise search for the first foreground neighbour, starting after "left"
            dir = (dir - 1) & 7;
            first_nb = start + nbr[dir];
        } while (*first_nb == BG && dir != stop);
        if (dir == stop) { *start = SEEN_RIGHT_EXIT; return 0; }
        long long twice_area = 0;
        signed char *cur = start;
        int cx = px, cy = py;
        for (;;) {
            int from = dir;
            signed char *nxt;
            for (;;) {  // counter-clockwise search from the direction we came from
                nxt = cur + nbr[++dir];
                if (*nxt != BG) break;
            }
            dir &= 7;
            if ((unsigned)(dir - 1) < (unsigned)from) *cur = SEEN_RIGHT_EXIT;
            else if (*cur == FG) *cur = SEEN;
            int nx = cx + dx[dir], ny = cy + dy[dir];
            twice_area += (long long)cx * ny - (long long)nx * cy;
            ++steps;
            cx = nx; cy = ny;
            if (cx < minx) minx = cx;
            if (cx > maxx) maxx = cx;
            if (cy < miny) miny = cy;
            if (cy > maxy) maxy = cy;
            if (nxt == start && cur == first_nb) break;
            cur = nxt;
            dir = (dir + 4) & 7;
        }
        return twice_area;
    }
};
}  // namespace

static thread_local Tracer g_tracer;   // working copy reused across calls (one context drives one thread at a time)

static int scan_prepared(Tracer &tr, int H, const uint32_t *row_any, RoiResult *out, bool have_ranges);

int largest_external_contour(const uint8_t *bin, int H, int W, const uint32_t *row_any, RoiResult *out)
{
    out->found = 0; out->n_contours = 0; out->area = 0.0; out->steps = 0;
    out->x = out->y = out->w = out->h = 0;
    if (H <= 0 || W <= 0) return 0;
    g_tracer.prepare(bin, H, W, row_any);
    g_tracer.steps = 0;
    const int rc = scan_prepared(g_tracer, H, row_any, out, false);
    out->steps = g_tracer.steps;
    return rc;
}

// Raster scan for border starts driven by the PACKED image: the foreground runs of a row come from bit tricks on
// the 64-pixel groups, so background pixels are never touched, and the marks the border follower leaves (only ever
// on foreground pixels) are read back per run.  Same decisions as scan_prepared():
//   a run starts a new outer border iff its first pixel is still unmarked and the last marked pixel to its left in
//   this row is not a SEEN (entering) border, i.e. we are not inside an already traced component (RETR_EXTERNAL);
//   the "last marked pixel" then moves to the last marked pixel of this run, if it has one.
static int scan_runs(Tracer &tr, const uint64_t *bits, int H, int W, RoiResult *out, int ya = 0, int yb = -2)
{
    const int step = tr.step;
    double best = -1.0;
    if (yb == -2 || yb > H - 1) yb = H - 1;
    if (ya < 0) ya = 0;
    for (int y = ya; y <= yb; ++y) {
        if (!tr.own_row_any[y]) continue;
        signed char *row = tr.buf.data() + (size_t)(y + 1) * step + 1;   // row[c] = pixel c
        const size_t r0 = (size_t)y * W;
        int last_mark = BG;      // value of the last marked pixel seen in this row (BG: none, the zero frame)
        int run_start = -1;
        uint64_t prev_bit = 0;
        auto handle_run = [&](int a, int b) {   // pixels [a, b) are foreground in the image
            if (row[a] == FG && !(last_mark > 0)) {
                int minx, miny, maxx, maxy;
                long long a2 = tr.follow(row + a, a + 1, y + 1, minx, miny, maxx, maxy);
                double area = 0.5 * (double)(a2 < 0 ? -a2 : a2);
                ++out->n_contours;
                if (area >= best) {
                    best = area;
                    out->found = 1;
                    out->x = minx - 1; out->y = miny - 1;
                    out->w = maxx - minx + 1; out->h = maxy - miny + 1;
                    out->area = area;
                }
            }
            for (int x = b - 1; x >= a; --x)
                if (row[x] & -2) { last_mark = row[x]; break; }
        };
        for (int x = tr.grp_a[y]; x <= tr.grp_b[y]; x += 64) {
            const uint64_t v = Tracer::get64(bits, r0 + x, r0 + W);
            uint64_t t = v ^ ((v << 1) | prev_bit);   // bit k set: pixel x+k differs from pixel x+k-1
            prev_bit = v >> 63;
            while (t) {
                const int pos = x + __builtin_ctzll(t);
                t &= t - 1;
                if (run_start < 0) run_start = pos;
                else { handle_run(run_start, pos); run_start = -1; }
            }
        }
        if (run_start >= 0) {   // the run reaches the end of the last group that holds foreground
            const int end = (tr.grp_b[y] + 64 < W) ? tr.grp_b[y] + 64 : W;
            handle_run(run_start, end);
        }
    }
    return 0;
}

int largest_external_contour_bits(const uint64_t *bits, int H, int W, RoiResult *out)
{
    out->found = 0; out->n_contours = 0; out->area = 0.0; out->steps = 0;
    out->x = out->y = out->w = out->h = 0;
    if (H <= 0 || W <= 0) return 0;
    g_tracer.prepare_bits(bits, H, W);
    g_tracer.steps = 0;
    const int rc = scan_runs(g_tracer, bits, H, W, out);
    out->steps = g_tracer.steps;
    return rc;
}

// One hole-free blob?  If every row of [y0, y1] holds exactly ONE run of foreground and the runs of neighbouring rows touch
// (8-connectivity: they overlap or meet diagonally), the foreground is a single 8-connected component without holes: cv2.findContours
// (RETR_EXTERNAL) lists ONE contour and its boundingRect is the bounding box of the runs (base.py:568-575) -- no border needs following,
// no working copy unpacking.  Rows outside [y0, y1] hold no foreground (the caller's row flags).  Returns 1 and fills `out` for
// such an image, 0 for anything else (the caller then follows the borders).  Word arithmetic on the packed rows: ~3 us for the
// 235 rows of the synthetic 1080p stream against ~15 us of unpacking and border following.
int simple_shape_bits_rows(const uint64_t *bits, int H, int W, int y0, int y1, RoiResult *out)
{
    if (H <= 0 || W <= 0 || y1 < y0 || y0 < 0 || y1 >= H) return 0;
    int pf = 0, pl = -1, xmin = W, xmax = -1, ya = -1, yb = -1;
    int state = 0;   // 0: above the blob (the window's rows are a superset: empty rows may lead and trail), 1: inside, 2: below
    for (int y = y0; y <= y1; ++y) {
        const size_t p0 = (size_t)y * W, end = p0 + W;
        int first = -1, last = -1, runs = 0;
        uint64_t prev = 0;
        for (size_t p = p0; p < end;) {
            const size_t off = p & 63;
            const size_t n = std::min<size_t>(64 - off, end - p);
            uint64_t m = bits[p >> 6] >> off;
            if (n < 64) m &= (~0ull) >> (64 - n);
            if (m) {
                if (first < 0) first = (int)(p - p0) + __builtin_ctzll(m);
                last = (int)(p - p0) + 63 - __builtin_clzll(m);
                runs += __builtin_popcountll(m & ~((m << 1) | prev));
                if (runs > 1) return 0;
            }
            prev = (m >> (n - 1)) & 1ull;
            p += n;
        }
        if (runs == 0) { if (state == 1) state = 2; continue; }
        if (state == 2) return 0;                                                    // foreground below an empty row: two components
        if (state == 1 && !(first <= pl + 1 && last >= pf - 1)) return 0;            // the runs of the two rows do not touch
        if (state == 0) { state = 1; ya = y; }
        yb = y; pf = first; pl = last;
        xmin = first < xmin ? first : xmin; xmax = last > xmax ? last : xmax;
    }
    if (ya < 0) return 0;   // no foreground at all: the caller's path reports "no contour"
    out->found = 1; out->n_contours = 1; out->steps = 0;
    out->x = xmin; out->y = ya; out->w = xmax - xmin + 1; out->h = yb - ya + 1;
    out->area = -1.0;   // (not computed: the only contour is the largest one whatever its area)
    return 1;
}

// The same rule from the per-row records of k_heat_rows_u8 (rm_kernels.h): rec[y] = first | last << 16 | runs << 32 | 1 << 48 for a row
// that holds foreground, 0 for one that does not.  *y0 / *y1: first / last row with foreground (y1 < y0: none) -- what the border
// following needs when the rule does not settle the image.  Returns 1 and fills `out` for ONE hole-free blob, 0 otherwise.
int simple_shape_row_records(const uint64_t *rec, int H, int W, int *y0, int *y1, RoiResult *out)
{
    int pf = 0, pl = -1, xmin = W, xmax = -1, ya = H, yb = -1;
    bool ok = true;
    for (int y = 0; y < H; ++y) {
        const uint64_t r = rec[y];
        if (!r) continue;
        const int first = (int)(r & 0xffffu), last = (int)((r >> 16) & 0xffffu), runs = (int)((r >> 32) & 0xffffu);
        if (yb >= 0) {
            if (y != yb + 1) ok = false;                                    // foreground below an empty row: two components
            else if (!(first <= pl + 1 && last >= pf - 1)) ok = false;      // the runs of the two rows do not touch
        }
        if (runs != 1) ok = false;
        if (y < ya) ya = y;
        yb = y; pf = first; pl = last;
        xmin = first < xmin ? first : xmin; xmax = last > xmax ? last : xmax;
    }
    *y0 = ya; *y1 = yb;
    if (yb < 0 || !ok) return 0;
    out->found = 1; out->n_contours = 1; out->steps = 0;
    out->x = xmin; out->y = ya; out->w = xmax - xmin + 1; out->h = yb - ya + 1;
    out->area = -1.0;   // (not computed: the only contour is the largest one whatever its area)
    return 1;
}

int largest_external_contour_bits_rows(const uint64_t *bits, int H, int W, int y0, int y1, RoiResult *out)
{
    out->found = 0; out->n_contours = 0; out->area = 0.0; out->steps = 0;
    out->x = out->y = out->w = out->h = 0;
    if (H <= 0 || W <= 0) return 0;
    g_tracer.prepare_bits(bits, H, W, y0, y1 < y0 ? y0 - 1 : y1);
    if (y1 < y0) return 0;
    g_tracer.steps = 0;
    const int rc = scan_runs(g_tracer, bits, H, W, out, y0, y1);
    out->steps = g_tracer.steps;
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// Labelled variant.  The outer border of a component starts at its smallest pixel index (top row, leftmost pixel), its
// bounding box is the component's, and twice its area is at most 2 (w-1)(h-1); components nested in a hole have a strictly
// smaller area than the enclosing border, so it does not matter that RETR_EXTERNAL would not list them.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct BitImage {
    const uint64_t *bits; int H, W;
    inline bool fg(int x, int y) const
    {
        if ((unsigned)x >= (unsigned)W || (unsigned)y >= (unsigned)H) return false;
        const size_t p = (size_t)y * W + x;
        return (bits[p >> 6] >> (p & 63)) & 1ull;
    }
};

// The packed image sits in pinned memory the device has just written: no host cache holds its lines, and a long border visits them
// in an order no hardware prefetcher follows (~150 us for the frame-spanning blob of a noise image, nearly all of it cache misses).
// Requesting the rows of the component's bounding box up front turns that into one streaming read that overlaps the walk.
static void prefetch_rows(const BitImage &im, int y0, int y1)
{
    if (y1 - y0 < 8) return;
    const char *p = (const char *)(im.bits + (((size_t)y0 * im.W) >> 6));
    const char *e = (const char *)(im.bits + ((((size_t)(y1 + 1) * im.W) + 63) >> 6));
    for (; p < e; p += 64) __builtin_prefetch(p, 0, 3);
}

static thread_local long long g_bits_steps = 0;   // border steps follow_bits() walked since the entry point reset it
// twice the signed shoelace area of the outer border that starts at (sx, sy): Tracer::follow without marks
long long follow_bits(const BitImage &im, int sx, int sy)
{
    static const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    static const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int dir = 4, fx, fy;
    do {
        dir = (dir - 1) & 7;
        fx = sx + dx[dir]; fy = sy + dy[dir];
    } while (!im.fg(fx, fy) && dir != 4);
    if (dir == 4) return 0;   // isolated pixel
    long long twice_area = 0;
    int cx = sx, cy = sy;
    // away from the image frame a neighbour is one add on the pixel index; on the frame every probe is bounds-checked
    const int W = im.W, H = im.H;
    long long off[8];
    for (int d = 0; d < 8; ++d) off[d] = (long long)dy[d] * W + dx[d];
    const uint64_t *bits = im.bits;
    for (;;) {
        int nx, ny;
        if (cx > 0 && cy > 0 && cx < W - 1 && cy < H - 1) {
            const long long p = (long long)cy * W + cx;
            for (;;) {
                ++dir;
                const long long q = p + off[dir & 7];
                if ((bits[q >> 6] >> (q & 63)) & 1ull) break;
            }
            dir &= 7;
            nx = cx + dx[dir]; ny = cy + dy[dir];
        } else {
            for (;;) {
                ++dir;
                nx = cx + dx[dir & 7]; ny = cy + dy[dir & 7];
                if (im.fg(nx, ny)) break;
            }
            dir &= 7;
        }
        twice_area += (long long)cx * ny - (long long)nx * cy;
        ++g_bits_steps;
        if (nx == sx && ny == sy && cx == fx && cy == fy) break;
        cx = nx; cy = ny;
        dir = (dir + 4) & 7;
    }
    return twice_area;
}
}  // namespace

int largest_external_contour_labelled(const uint64_t *bits, int H, int W, const LabelComp *comps, size_t n, RoiResult *out)
{
    out->found = 0; out->n_contours = (int)n; out->area = 0.0; out->steps = 0;
    out->x = out->y = out->w = out->h = 0;
    if (H <= 0 || W <= 0 || n == 0) return 0;
    g_bits_steps = 0;
    const BitImage im{bits, H, W};
    auto bound2 = [](const LabelComp &c) { return 2ll * (long long)c.w1 * (long long)c.h1; };   // twice the largest area inside the box
    long long best2 = -1; size_t best_i = 0;
    auto consider = [&](size_t i) {
        const LabelComp &c = comps[i];
        long long a2 = 0;
        if (c.w1 > 0 && c.h1 > 0) { prefetch_rows(im, c.root / W, c.root / W + c.h1); a2 = follow_bits(im, c.root % W, c.root / W); if (a2 < 0) a2 = -a2; }
        // max() over cv2's reversed list keeps, among equal areas, the border discovered last = the largest start index
        if (a2 > best2 || (a2 == best2 && c.root > comps[best_i].root)) { best2 = a2; best_i = i; }
    };
    // the component with the largest bound sets the bar (on a blob in noise it is the winner) ...
    long long top = -1; size_t top_i = 0;
    for (size_t i = 0; i < n; ++i) {
        const long long b = bound2(comps[i]);
        if (b > top) { top = b; top_i = i; }
    }
    consider(top_i);
    // ... then everything that can still reach (or tie with) the best area so far; the bar rises as borders are followed
    for (size_t i = 0; i < n; ++i)
        if (i != top_i && bound2(comps[i]) >= best2) consider(i);
    const LabelComp &c = comps[best_i];
    out->found = 1;
    out->x = c.minx; out->y = c.root / W;
    out->w = c.w1 + 1; out->h = c.h1 + 1;
    out->area = 0.5 * (double)best2;
    out->steps = g_bits_steps;
    return 0;
}

// The same from the per-workgroup summaries of k_ccl_publish (rm_ccl.h): tops[2 b] = the record with the largest bound among the
// components workgroup b published (root -1: none), tops[2 b + 1] = {low, high word of the second largest bound there}.  Follows the
// best top's border, then every other top that can still reach its area; when that area also beats every second bound no unread
// component can win or tie, and the full list (`comps`, read only then) is not touched.  Same result as the function above.
// the summary records alone: true (and the ROI in *out) when twice the area of the top record's contour -- at least 2 N - P - 2,
// rm_ccl.h ccl_piece_2n_minus_p -- already beats the box bound of every other component, i.e. the other tops and every workgroup's
// second bound: the winner is known without following its border (a blob that spans a noisy frame: tens of thousands of steps)
bool labelled_tops_settled(const LabelComp *tops, int nblocks, int W, size_t n, RoiResult *out)
{
    auto bound2 = [](const LabelComp &c) { return 2ll * (long long)c.w1 * (long long)c.h1; };
    long long top = -1; int top_b = -1;
    for (int b = 0; b < nblocks; ++b)
        if (tops[2 * b].root >= 0 && bound2(tops[2 * b]) > top) { top = bound2(tops[2 * b]); top_b = b; }
    if (top_b < 0 || W <= 0) return false;
    const long long low2 = (long long)tops[2 * top_b + 1].w1 - 2;
    if (low2 <= 0) return false;
    for (int b = 0; b < nblocks; ++b) {
        if (b != top_b && tops[2 * b].root >= 0 && bound2(tops[2 * b]) >= low2) return false;
        const LabelComp &s = tops[2 * b + 1];
        if ((long long)(((unsigned long long)(unsigned int)s.minx << 32) | (unsigned int)s.root) >= low2) return false;
    }
    const LabelComp &c = tops[2 * top_b];
    out->found = 1; out->n_contours = (int)n; out->steps = 0;
    out->x = c.minx; out->y = c.root / W;
    out->w = c.w1 + 1; out->h = c.h1 + 1;
    out->area = -1.0;   // (not computed: no other contour can reach this one's lower bound)
    return true;
}

int largest_external_contour_labelled_tops(const uint64_t *bits, int H, int W, const LabelComp *tops, int nblocks, const LabelComp *comps, size_t n,
                                           RoiResult *out, bool area_bound_shortcut)
{
    out->found = 0; out->n_contours = (int)n; out->area = 0.0; out->steps = 0;
    out->x = out->y = out->w = out->h = 0;
    if (H <= 0 || W <= 0 || n == 0) return 0;
    g_bits_steps = 0;
    const BitImage im{bits, H, W};
    auto bound2 = [](const LabelComp &c) { return 2ll * (long long)c.w1 * (long long)c.h1; };
    long long best2 = -1; int best_b = -1;
    auto consider = [&](int b) {
        const LabelComp &c = tops[2 * b];
        long long a2 = 0;
        if (c.w1 > 0 && c.h1 > 0) { prefetch_rows(im, c.root / W, c.root / W + c.h1); a2 = follow_bits(im, c.root % W, c.root / W); if (a2 < 0) a2 = -a2; }
        if (a2 > best2 || (a2 == best2 && c.root > tops[2 * best_b].root)) { best2 = a2; best_b = b; }
    };
    long long top = -1; int top_b = -1;
    for (int b = 0; b < nblocks; ++b)
        if (tops[2 * b].root >= 0 && bound2(tops[2 * b]) > top) { top = bound2(tops[2 * b]); top_b = b; }
    if (top_b < 0) return largest_external_contour_labelled(bits, H, W, comps, n, out);
    if (area_bound_shortcut && labelled_tops_settled(tops, nblocks, W, n, out)) return 0;
    consider(top_b);
    for (int b = 0; b < nblocks; ++b)
        if (b != top_b && tops[2 * b].root >= 0 && bound2(tops[2 * b]) >= best2) consider(b);
    for (int b = 0; b < nblocks; ++b) {
        const LabelComp &s = tops[2 * b + 1];
        const long long second = (long long)(((unsigned long long)(unsigned int)s.minx << 32) | (unsigned int)s.root);
        if (second >= best2) {   // an unread component may win or tie
            const long long walked = g_bits_steps;
            const int rc = largest_external_contour_labelled(bits, H, W, comps, n, out);
            out->steps += walked;
            return rc;
        }
    }
    const LabelComp &c = tops[2 * best_b];
    out->found = 1;
    out->x = c.minx; out->y = c.root / W;
    out->w = c.w1 + 1; out->h = c.h1 + 1;
    out->area = 0.5 * (double)best2;
    out->steps = g_bits_steps;
    return 0;
}

// `have_ranges`: pixels left of row_x0 / right of row_x1 are background without marks (borders only visit
// foreground pixels), so the raster scan of a row may start at its first and stop after its last foreground pixel
static int scan_prepared(Tracer &tr, int H, const uint32_t *row_any, RoiResult *out, bool have_ranges)
{
    const int full_step = tr.step;
    double best = -1.0;
    for (int y = 0; y < H; ++y) {
        if (row_any && !row_any[y]) continue;
        signed char *row = tr.buf.data() + (size_t)(y + 1) * full_step;
        int prev = BG;
        int last_border_x = 0;  // column 0 is the zero frame: "outside"
        int x = have_ranges ? tr.row_x0[y] + 1 : 1;                       // padded coordinates: pixel c sits at c + 1
        const int step = have_ranges ? tr.row_x1[y] + 3 : full_step;      // one background pixel past the last foreground
        while (x < step) {
            // skip to the next transition, 8 pixels at a time
            {
                unsigned long long pat = 0x0101010101010101ull * (unsigned char)prev;
                while (x + 8 <= step) {
                    unsigned long long wv;
                    std::memcpy(&wv, row + x, 8);
                    if (wv != pat) break;
                    x += 8;
                }
                while (x < step && row[x] == prev) ++x;
                if (x >= step) break;
            }
            int p = row[x];
            bool outer_start = (prev == BG && p == FG);
            // RETR_EXTERNAL: ignore hole borders and outer borders lying inside a component
            if (outer_start && !(row[last_border_x] > 0)) {
                int minx, miny, maxx, maxy;
                long long a2 = tr.follow(row + x, x, y + 1, minx, miny, maxx, maxy);
                double area = 0.5 * (double)(a2 < 0 ? -a2 : a2);
                ++out->n_contours;
                if (area >= best) {
                    best = area;
                    out->found = 1;
                    out->x = minx - 1; out->y = miny - 1;
                    out->w = maxx - minx + 1; out->h = maxy - miny + 1;
                    out->area = area;
                }
                p = row[x];
            }
            prev = p;
            if (prev & -2) last_border_x = x;
            ++x;
        }
    }
    return 0;
}

}  // namespace rm
