// respmon_amd/csrc/rm_ccl.h -- device half of the ROI stage for NOISY thresholded images (base.py:568-575).
//
//   cv2.findContours(thresh, RETR_EXTERNAL, ...) -> max(contours, key=cv2.contourArea) -> cv2.boundingRect
//
// locate() consumes ONE contour.  On a noisy heatmap (BASELINE configs 2 / 5: level 2 of a 4-level pyramid is noise
// dominated) the thresholded image holds thousands of specks, and following every one of their borders on the host was half
// of the step (7 466 borders, 0.94 ms at 720p; 4.5 ms at 4K).  None of them can win:
//   * an external contour is the outer border of one 8-connected component, its boundingRect is the component's bounding
//     box, and its shoelace area (a polygon through pixel centres inside that box) is at most (w-1)*(h-1);
//   * a component nested in a hole of another one (which RETR_EXTERNAL does not list) lies strictly inside the enclosing
//     outer border, so its area is strictly smaller and it can never be the maximum either way.
// So the device labels the components (union-find over the bit-packed image, root = smallest pixel index = the pixel
// Suzuki-Abe starts the outer border at), reduces a bounding box per root and hands the host one 16-byte record per
// component; the host follows only borders whose bound can reach the best area found so far (rm_contour.cpp,
// largest_external_contour_labelled) -- typically one.  Exact: same contour, same ties (last discovered = largest root).
//
// Two launches over the H*W bits (and a small publishing kernel), one thread per pixel (84 % of the threads of a 16 %-foreground
// image leave at once); their start state comes from k_heat_to_u8, whose ballot is the pixel's word:
//   (k_heat_to_u8) label = first pixel of the pixel's run inside its 64-bit word (rows break runs); that first pixel holds the box of its piece
//   k_ccl_union   joins with the row above / the word to the left, lock-free (atomicMin on the larger root)
//   k_ccl_bbox    the first pixel of every piece folds the piece's box into its root's
//   k_ccl_publish the roots k_ccl_bbox listed -> {root, minx, width-1, height-1} records + their count in pinned host memory
// Rows of whole 64-pixel words (720p, 1080p, 4K) take k_ccl_tile / k_ccl_seam / k_ccl_fold (at the end of this file) in place of the first
// two: the components of every 64 x 32 tile on LDS labels first, then the seams.  Same roots, boxes and counts for k_ccl_publish.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rm {

struct CclComp { int root, minx, w1, h1; };   // box: x = minx, y = root / W, width w1 + 1, height h1 + 1
// (CclBox, the per-root bounding box, is declared next to k_heat_to_u8 in rm_kernels.h)

__device__ inline bool ccl_bit(const unsigned long long *bits, size_t p) { return (bits[p >> 6] >> (p & 63)) & 1ull; }

__device__ inline int ccl_find(const int *label, int a)
{
    int l = label[a];
    while (l != a) { a = l; l = label[a]; }
    return a;
}

// find with path halving: every node passed is re-hung under its grandparent.  Labels only ever move to a smaller index of the same
// tree, so the plain stores race benignly with each other and with the atomicMin of a union (a store that overwrites a fresh hook
// can only hit a node that was no root when the hook's atomicMin ran -- that union has seen old != b and carries on from `old`).
// Without it a blob of h rows leaves chains h links deep (each row's run hooked under the run above before that one was hooked
// itself), and every later find pays an L2 round trip per link: k_ccl_bbox 89 us on a 1080p frame of noise blobs.
__device__ inline int ccl_find_halving(int *label, int a)
{
    int l = label[a];
    if (l != a) {
        int prev = a, next;
        while (l > (next = label[l])) { label[prev] = next; prev = l; l = next; }
    }
    return l;
}

__device__ inline void ccl_union(int *label, int a, int b)
{
    for (;;) {
        a = ccl_find_halving(label, a);
        b = ccl_find_halving(label, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }   // a < b: b's tree hangs under a
        const int old = atomicMin(&label[b], a);
        if (old == b) return;
        b = old;                                         // somebody re-rooted b meanwhile: join with what it points to now
    }
}

// the unions of ONE foreground pixel p = (x, y): with the word to the left (a run that continues from the previous word of its row)
// and with the row above
__device__ inline void ccl_pixel_unions(const unsigned long long *bits, int *label, size_t p, int y, int x, int W)
{
    const bool w_fg = x > 0 && ccl_bit(bits, p - 1);
    if (w_fg && (p & 63) == 0) ccl_union(label, (int)p, (int)p - 1);
    if (y == 0) return;
    const size_t up = p - (size_t)W;
    const bool n_fg = ccl_bit(bits, up);
    const bool nw_fg = x > 0 && ccl_bit(bits, up - 1);
    if (n_fg) {
        // west neighbour joined with its own north (= our north-west), which touches our north: already one component
        if (!(w_fg && nw_fg)) ccl_union(label, (int)p, (int)up);
        return;
    }
    if (nw_fg && !w_fg) ccl_union(label, (int)p, (int)(up - 1));   // otherwise west joins with it (it is west's north)
    if (x + 1 < W && ccl_bit(bits, up + 1)) {
        const bool e_fg = ccl_bit(bits, p + 1);
        if (!e_fg) ccl_union(label, (int)p, (int)(up + 1));        // otherwise east joins with it (it is east's north)
    }
}

RM_KERNEL __launch_bounds__(256) void k_ccl_union(const unsigned long long *bits, size_t npix, int H, int W, int *label)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix || !ccl_bit(bits, p)) return;
    const int y = (int)(p / (size_t)W), x = (int)(p - (size_t)y * W);
    ccl_pixel_unions(bits, label, p, y, x, W);
}

// A workgroup = a tile of 64 columns x CCL_BOX_ROWS rows, one thread per pixel.  The first pixel of a piece of a run (a run cut at
// word boundaries: k_heat_to_u8 left the piece's box with it) finds its root and folds the piece's box into a small LDS table keyed
// by root; the table's few entries then go to the roots' boxes.  Atomics on ONE address cost ~10 ns each whatever the grid does: a
// frame of noise blobs has a root half the image hangs under, and one atomicMax per piece of every new row of it made this kernel
// 89-107 us at 1080p (a thread per pixel in row-shaped workgroups); a tile sends at most three per root it touches.
constexpr int CCL_BOX_ROWS = 16;
constexpr int CCL_BOX_SLOTS = 32;

__device__ inline void ccl_box_fold(CclBox *box, int r, int minx, int maxx, int maxy)
{
    int *b = (int *)&box[r];
    // the box only ever grows: a (possibly stale) value that already covers this one makes the atomic unnecessary
    const CclBox cur = box[r];
    if (minx < cur.minx) atomicMin(b + 0, minx);
    if (maxx > cur.maxx) atomicMax(b + 1, maxx);
    if (maxy > cur.maxy) atomicMax(b + 2, maxy);
}

// 64 pixels starting at pixel q as one word (bit k = pixel q + k; zeros past the image)
__device__ inline unsigned long long ccl_window(const unsigned long long *bits, size_t q, size_t nwords)
{
    const size_t i = q >> 6;
    const unsigned int s = (unsigned int)(q & 63);
    unsigned long long v = bits[i] >> s;
    if (s && i + 1 < nwords) v |= bits[i + 1] << (64 - s);
    return v;
}

// A LOWER bound of a component's contour area from counts alone, so that the host need not follow the border of a blob that spans
// the frame (full-frame noise at 1080p: a 30 000-step border, 165 us of a 1.4 ms step, while every rival's box is smaller than it).
// The outer border is a closed lattice walk of B steps around N_filled >= N pixels, and Pick's theorem (it stays true for the walks
// border following produces: spurs walked twice, pinch pixels visited twice) gives  area = N_filled - B / 2 - 1.  Every step leaves
// its pixel through a crack (a side shared with a background pixel or the frame) no other step uses, so B <= P, the component's
// crack count:                      2 * area >= 2 N - P - 2.
// This is the piece's share of 2 N - P: a piece is a run of L ones inside one word and one row, starting at pixel p = (x, y).
__device__ inline int ccl_piece_2n_minus_p(const unsigned long long *bits, size_t npix, int H, int W, size_t p, int x, int y)
{
    const size_t nwords = (npix + 63) >> 6;
    const unsigned int sh = (unsigned int)(p & 63);
    const unsigned long long inv = ~(bits[p >> 6] >> sh);
    int L = inv ? __builtin_ctzll(inv) : 64;
    if (L > 64 - (int)sh) L = 64 - (int)sh;
    if (L > W - x) L = W - x;
    const unsigned long long mask = L == 64 ? ~0ull : ((1ull << L) - 1ull);
    int cracks = (x == 0 || !ccl_bit(bits, p - 1)) ? 1 : 0;
    cracks += (x + L == W || !ccl_bit(bits, p + L)) ? 1 : 0;
    cracks += y == 0 ? L : __builtin_popcountll(mask & ~ccl_window(bits, p - (size_t)W, nwords));
    cracks += y == H - 1 ? L : __builtin_popcountll(mask & ~ccl_window(bits, p + (size_t)W, nwords));
    return 2 * L - cracks;
}

// The same for rows of whole words (W % 64 == 0: 720p, 1080p, 4K), where a wave of k_ccl_bbox covers exactly one word: the word, its
// neighbours in the row and the words above / below are wave-uniform loads shared by every piece of the word, instead of five dependent
// loads per piece (k_ccl_bbox<false> at 4K x 512: 147 -> see DESIGN 4.4).  `lane` = the piece's first bit.
struct CclWords { unsigned long long cur, up, down; bool left_fg, right_fg; };   // left_fg / right_fg: the pixel before bit 0 / after bit 63 is foreground
__device__ inline CclWords ccl_words(const unsigned long long *bits, int H, int W, int y, int x0)
{
    const size_t i = ((size_t)y * W + x0) >> 6, wpr = (size_t)(W >> 6);
    CclWords w;
    w.cur = bits[i];
    w.up = y > 0 ? bits[i - wpr] : 0ull;
    w.down = y < H - 1 ? bits[i + wpr] : 0ull;
    w.left_fg = x0 > 0 && (bits[i - 1] >> 63);
    w.right_fg = x0 + 64 < W && (bits[i + 1] & 1ull);
    return w;
}
__device__ inline int ccl_piece_2n_minus_p_words(const CclWords &w, int lane)
{
    const unsigned long long inv = ~(w.cur >> lane);
    int L = inv ? __builtin_ctzll(inv) : 64;
    if (L > 64 - lane) L = 64 - lane;
    const unsigned long long mask = (L == 64 ? ~0ull : ((1ull << L) - 1ull)) << lane;
    int cracks = (lane > 0 || !w.left_fg) ? 1 : 0;
    cracks += (lane + L < 64 || !w.right_fg) ? 1 : 0;
    cracks += __builtin_popcountll(mask & ~w.up) + __builtin_popcountll(mask & ~w.down);
    return 2 * L - cracks;
}

// The roots met on the way (a root is the first pixel of its piece) are listed for k_ccl_publish: counted in LDS, ONE reservation
// per tile on counters[0] (every atomic on the one counter takes ~10 ns of the L2's atomic unit), in no particular order -- the
// host's choice does not depend on it.  roots[] holds `cap` entries; counters[0] keeps counting past it (the host then follows
// every border itself).
// TABLE = false: no LDS table, every piece folds its box straight into its root's -- for images of thousands of specks (a component
// rarely leaves its tile, the table overflows and its probes are pure overhead: 4K x 512 skip 2, 130 us with the table).
template <bool TABLE>
__global__ __launch_bounds__(64 * CCL_BOX_ROWS) void k_ccl_bbox(const unsigned long long *bits, size_t npix, int H, int W, int *label,
                                                                 CclBox *box, unsigned int *counters, int *roots, unsigned int cap)
{
    __shared__ int s_key[CCL_BOX_SLOTS], s_minx[CCL_BOX_SLOTS], s_maxx[CCL_BOX_SLOTS], s_maxy[CCL_BOX_SLOTS], s_cnt[CCL_BOX_SLOTS];
    __shared__ unsigned int s_nroots, s_base;
    __shared__ int s_hot, s_hot_cnt;   // TABLE = false: the counts of ONE root per tile (the first one met) meet in LDS -- see below
    const int tid = threadIdx.x;
    if (tid == 0) { s_hot = -1; s_hot_cnt = 0; }
    if (TABLE && tid < CCL_BOX_SLOTS) { s_key[tid] = -1; s_minx[tid] = 0x7fffffff; s_maxx[tid] = -1; s_maxy[tid] = -1; s_cnt[tid] = 0; }
    if (tid == 0) s_nroots = 0;
    __syncthreads();
    const int x = blockIdx.x * 64 + (tid & 63), y = blockIdx.y * CCL_BOX_ROWS + (tid >> 6);
    int my_root = -1; unsigned int my_slot = 0;
    const bool whole_words = (W & 63) == 0;     // a wave = one word of one row
    CclWords ww = {0ull, 0ull, 0ull, false, false};
    if (whole_words && y < H) ww = ccl_words(bits, H, W, y, blockIdx.x * 64);   // (wave-uniform)
    if (x < W && y < H) {
        const size_t p = (size_t)y * W + x;
        if (whole_words ? (((ww.cur >> (tid & 63)) & 1ull) && ((tid & 63) == 0 || !((ww.cur >> ((tid & 63) - 1)) & 1ull)))
                        : (ccl_bit(bits, p) && ((p & 63) == 0 || x == 0 || !ccl_bit(bits, p - 1)))) {
            const int r = ccl_find(label, (int)p);
            // -> box[root].cnt (k_heat_to_u8 left 0 there)
            const int cnt = whole_words ? ccl_piece_2n_minus_p_words(ww, tid & 63) : ccl_piece_2n_minus_p(bits, npix, H, W, p, x, y);
            if (r == (int)p) { my_root = r; my_slot = atomicAdd(&s_nroots, 1u); atomicAdd(&box[p].cnt, cnt); }
            if (r != (int)p) {                       // (the root's own piece is in its box already)
                const CclBox mine = box[p];
                unsigned int h = ((unsigned int)r * 2654435761u) >> 27;
                int k = TABLE ? 0 : CCL_BOX_SLOTS;
                for (; k < CCL_BOX_SLOTS; ++k, h = (h + 1) & (CCL_BOX_SLOTS - 1)) {
                    const int seen = atomicCAS(&s_key[h], -1, r);
                    if (seen == -1 || seen == r) break;
                }
                if (k < CCL_BOX_SLOTS) {
                    atomicMin(&s_minx[h], mine.minx); atomicMax(&s_maxx[h], mine.maxx); atomicMax(&s_maxy[h], mine.maxy); atomicAdd(&s_cnt[h], cnt);
                } else {   // more roots in the tile than the table holds / no table
                    ccl_box_fold(box, r, mine.minx, mine.maxx, mine.maxy);
                    // the count cannot skip its atomic the way the box does, and a blob among the specks sends thousands of pieces to ONE
                    // address (~10 ns each: 4K x 512, k_ccl_bbox 116 -> 259 us).  The pieces of the first root a tile meets add up in LDS
                    // and leave as one atomic per tile; the specks' pieces go to addresses of their own.
                    const int seen = atomicCAS(&s_hot, -1, r);
                    if (seen == -1 || seen == r) atomicAdd(&s_hot_cnt, cnt); else atomicAdd(&box[r].cnt, cnt);
                }
            }
        }
    }
    __syncthreads();
    if (TABLE && tid < CCL_BOX_SLOTS && s_key[tid] >= 0) {
        ccl_box_fold(box, s_key[tid], s_minx[tid], s_maxx[tid], s_maxy[tid]);
        atomicAdd(&box[s_key[tid]].cnt, s_cnt[tid]);
    }
    if (tid == 0 && s_hot >= 0) atomicAdd(&box[s_hot].cnt, s_hot_cnt);
    if (s_nroots == 0) return;                       // (uniform)
    if (tid == 0) s_base = atomicAdd(&counters[0], s_nroots);
    __syncthreads();
    if (my_root >= 0 && s_base + my_slot < cap) roots[s_base + my_slot] = my_root;
}

// roots -> {root, minx, width-1, height-1} records in pinned host memory (the boxes are final behind the kernel boundary);
// host[0].root = the number of components (> cap: the list overflowed and the host follows every border itself).
// host[1 .. 2 CCL_PUB_BLOCKS]: per workgroup b of this launch, host[1 + 2 b] = the record with the largest bound 2 (w-1)(h-1) among the
// components the workgroup published (root -1: none) and host[2 + 2 b] = {low, high word of the SECOND largest bound there,
// 2 N - P of the top record's component (a lower bound of its area: ccl_piece_2n_minus_p), -}.
// The host follows the border of the best of these 64 records and is done when its area beats every second bound and every other
// top (a blob among thousands of specks: one border, 1.5 KB read) -- reading the whole list the device has just written
// (120 KB of lines no host cache holds at 7 466 components) was most of the 62 us the GPU idled per 720p step.
// The full list follows from host[1 + 2 CCL_PUB_BLOCKS] on, 16-byte records, consecutive lanes -> consecutive records.
constexpr int CCL_PUB_BLOCKS = 64;
// `list`: where the full list goes -- behind the summaries in the pinned memory, or a device buffer the host copies only when the summaries do
// not settle the winner (rm_roi.hip: the "lazy" labelled stage).
RM_KERNEL __launch_bounds__(256) void k_ccl_publish(const int *roots, const CclBox *box, int W, const unsigned int *counters, unsigned int cap,
                                                     CclComp *host, CclComp *list)
{
    __shared__ long long s_b1[256], s_b2[256];
    __shared__ int s_i1[256];
    const unsigned int total = counters[0];
    const unsigned int n = total < cap ? total : cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) { CclComp h; h.root = (int)total; h.minx = 0; h.w1 = 0; h.h1 = 0; host[0] = h; }
    long long b1 = -1, b2 = -1;     // largest / second largest bound this thread met
    int i1 = -1;                    // list index of the largest
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int p = roots[i];
        const CclBox bb = box[p];
        CclComp c; c.root = p; c.minx = bb.minx; c.w1 = bb.maxx - bb.minx; c.h1 = bb.maxy - p / W;
        list[i] = c;
        const long long b = 2ll * (long long)c.w1 * (long long)c.h1;
        if (b > b1) { b2 = b1; b1 = b; i1 = (int)i; } else if (b > b2) b2 = b;
    }
    const int tid = threadIdx.x;
    s_b1[tid] = b1; s_b2[tid] = b2; s_i1[tid] = i1;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (tid < d) {
            const long long o1 = s_b1[tid + d], o2 = s_b2[tid + d];
            long long m1 = s_b1[tid], m2 = s_b2[tid];
            int mi = s_i1[tid];
            if (o1 > m1) { m2 = m1 > o2 ? m1 : o2; m1 = o1; mi = s_i1[tid + d]; } else { m2 = o1 > m2 ? o1 : m2; }
            s_b1[tid] = m1; s_b2[tid] = m2; s_i1[tid] = mi;
        }
        __syncthreads();
    }
    if (tid == 0 && blockIdx.x < CCL_PUB_BLOCKS) {
        CclComp t; t.root = -1; t.minx = 0; t.w1 = 0; t.h1 = 0;
        if (s_i1[0] >= 0) {
            const int p = roots[s_i1[0]];
            const CclBox bb = box[p];
            t.root = p; t.minx = bb.minx; t.w1 = bb.maxx - bb.minx; t.h1 = bb.maxy - p / W;
        }
        CclComp sec; sec.root = (int)(unsigned int)(s_b2[0] & 0xffffffffll); sec.minx = (int)(s_b2[0] >> 32); sec.h1 = 0;
        sec.w1 = s_i1[0] >= 0 ? box[roots[s_i1[0]]].cnt : 0;   // 2 N - P of the top record's component (ccl_piece_2n_minus_p)
        host[1 + 2 * blockIdx.x] = t;
        host[2 + 2 * blockIdx.x] = sec;
    }
}

// ---- rows of whole words (W % 64 == 0: 720p, 1080p, 4K): the components of a TILE first, in LDS (round 6) -------------------------------
// k_ccl_union / k_ccl_bbox above take every union and every find through global memory: an L2 round trip per link, a thread per pixel,
// one box fold per piece.  Most of that work is local.  Here a workgroup owns a tile of one word x CCL_TILE_ROWS rows:
//   k_ccl_tile  the words into LDS; labels (local pixel index, root = smallest) start at each pixel's run start inside its word; the
//               unions with the row above INSIDE the tile on the LDS labels (the per-pixel rule of ccl_pixel_unions; what it reads
//               beyond the word's first / last bit is left to the seams); every piece's box and 2 N - P folded into its local
//               root's; out go: label[] of every foreground pixel = its tile component's first pixel (the "tile root"), that
//               pixel's box[] = the tile component's, and the tile's list of tile roots (a fixed segment per tile: no counter shared
//               between workgroups)
//   k_ccl_seam  the unions the tiles could not see, through global memory as before but 1 / 32 of the rows and 2 / 64 of the columns:
//               the complete rule for the pixels of the first row of every tile; for the other rows the first pixel of every word
//               (run continues from the previous word; north-west neighbour) and the last one (north-east neighbour)
//   k_ccl_fold  a workgroup per CCL_FOLD_TILES tiles walks their tile roots: a tile root that is its own root is a component (listed
//               for k_ccl_publish, one reservation per workgroup); the others fold their box and count into the root's through the LDS
//               table of k_ccl_bbox<true> -- one set of atomics per root and workgroup
// The result is the one the other path leaves: root = first pixel of the component, box[root] and box[root].cnt complete, the roots
// listed in no particular order.
constexpr int CCL_TILE_ROWS = 32;
constexpr int CCL_TILE_CAP = 16 * CCL_TILE_ROWS;   // tile roots a tile can hold: 32 pieces in a word, on every other row
constexpr int CCL_FOLD_TILES = 8;

__device__ inline int ccl_lds_find(int *lab, int a)
{
    int l = lab[a];
    if (l != a) {
        int prev = a, next;
        while (l > (next = lab[l])) { lab[prev] = next; prev = l; l = next; }
    }
    return l;
}
__device__ inline void ccl_lds_union(int *lab, int a, int b)
{
    for (;;) {
        a = ccl_lds_find(lab, a);
        b = ccl_lds_find(lab, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&lab[b], a);
        if (old == b) return;
        b = old;
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_ccl_tile(const unsigned long long *bits, int H, int W, int *label, CclBox *box, int *troots, int *tile_n)
{
    constexpr int TR = CCL_TILE_ROWS, NP = TR * 64;
    __shared__ unsigned long long s_w[TR + 2], s_side[2 * TR];   // rows y0 - 1 .. y0 + TR of the tile's word; the words left / right of rows y0 .. y0 + TR - 1
    // boxes and counts live at the run starts only, and two run starts of a row are at least two columns apart: slot (row, column / 2)
    constexpr int NB = TR * 32;
    __shared__ int s_lab[NP], s_minx[NB], s_maxx[NB], s_maxy[NB], s_cnt[NB];
    __shared__ int s_n;
    auto bi = [](int q) __attribute__((always_inline)) { return ((q >> 6) << 5) | ((q & 63) >> 1); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wpr = W >> 6, wx = blockIdx.x, y0 = blockIdx.y * TR, x0 = wx * 64;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    // every word the tile looks at, in ONE round of loads: thread k < TR + 2 its own column's row y0 - 1 + k, the next 2 TR the neighbours'
    if (tid < TR + 2) {
        const int y = y0 - 1 + tid;
        s_w[tid] = (y >= 0 && y < H) ? bits[(size_t)y * wpr + wx] : 0ull;
    } else if (tid < 3 * TR + 2) {
        const int k = tid - (TR + 2), r = k >> 1, y = y0 + r, wn = (k & 1) ? wx + 1 : wx - 1;
        s_side[k] = (y < H && wn >= 0 && wn < wpr) ? bits[(size_t)y * wpr + wn] : 0ull;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    const unsigned long long *sw = s_w + 1;   // sw[r]: row y0 + r
    if (__ballot(lane < TR && sw[lane < TR ? lane : 0] != 0ull) == 0ull) {   // (the same answer in every wave) nothing in the tile
        if (tid == 0) tile_n[tile] = 0;
        return;
    }
    // ---- start state: label = the pixel's run start inside its word; a run start holds its piece's box (columns local to the tile) and 2 N - P
    for (int r = wave; r < TR; r += NW) {
        const unsigned long long m = sw[r];
        if (!((m >> lane) & 1ull)) continue;
        CclWords ww;
        ww.cur = m; ww.up = sw[r - 1]; ww.down = sw[r + 1];
        ww.left_fg = (s_side[2 * r] >> 63) != 0ull; ww.right_fg = (s_side[2 * r + 1] & 1ull) != 0ull;
        const unsigned long long zeros_below = ~m & ((1ull << lane) - 1ull);
        const int run0 = zeros_below ? 64 - __builtin_clzll(zeros_below) : 0;
        const int q = r * 64 + lane;
        s_lab[q] = r * 64 + run0;
        if (run0 == lane) {
            const unsigned long long zeros_above = ~(m >> lane);
            int len = zeros_above ? __builtin_ctzll(zeros_above) : 64;
            if (len > 64 - lane) len = 64 - lane;
            const int b = bi(q);
            s_minx[b] = lane; s_maxx[b] = lane + len - 1; s_maxy[b] = r;
            s_cnt[b] = ccl_piece_2n_minus_p_words(ww, lane);
        }
    }
    __syncthreads();
    // ---- the unions with the row above inside the tile, by levels: first the odd rows with the even row above them (pairs of rows), then
    // the first row of every second pair with the pair above, ... -- the trees two blocks of rows bring to a level are as deep as the
    // level at most.  (All rows at once: a blob that fills the tile hooks row r under row r - 1 before that one is hooked itself, 32 links
    // deep, and every find walks them through LDS one round trip at a time: 31 us at 1080p.)
    for (int lvl = 0; (1 << lvl) < TR; ++lvl) {
        for (int r = (1 << lvl) + wave * (2 << lvl); r < TR; r += NW * (2 << lvl)) {
            const unsigned long long m = sw[r], u = sw[r - 1];
            if (m == 0ull || u == 0ull) continue;                     // (uniform)
            if (!((m >> lane) & 1ull)) continue;
            const int q = r * 64 + lane, qu = q - 64;
            const bool n_fg = (u >> lane) & 1ull;
            const bool w_fg = lane > 0 && ((m >> (lane - 1)) & 1ull), nw_fg = lane > 0 && ((u >> (lane - 1)) & 1ull);
            if (n_fg) {
                if (!(w_fg && nw_fg)) ccl_lds_union(s_lab, q, qu);
                continue;
            }
            if (nw_fg && !w_fg) ccl_lds_union(s_lab, q, qu - 1);
            if (lane < 63 && ((u >> (lane + 1)) & 1ull) && !((m >> (lane + 1)) & 1ull)) ccl_lds_union(s_lab, q, qu + 1);
        }
        __syncthreads();
    }
    // ---- every piece's box and count into its local root's (a non-root's entries are only its own piece's: nobody folds into it)
    for (int r = wave; r < TR; r += NW) {
        const unsigned long long m = sw[r];
        if (!((m >> lane) & 1ull) || (lane > 0 && ((m >> (lane - 1)) & 1ull))) continue;
        const int q = r * 64 + lane;
        const int root = ccl_lds_find(s_lab, q);
        if (root == q) continue;
        s_lab[q] = root;
        const int b = bi(q), br = bi(root);
        atomicMin(&s_minx[br], s_minx[b]); atomicMax(&s_maxx[br], s_maxx[b]); atomicMax(&s_maxy[br], s_maxy[b]);
        atomicAdd(&s_cnt[br], s_cnt[b]);
    }
    __syncthreads();
    // ---- out: the labels of the foreground pixels, the tile roots with their boxes
    for (int r = wave; r < TR; r += NW) {
        const unsigned long long m = sw[r];
        if (!((m >> lane) & 1ull)) continue;
        const int q = r * 64 + lane;
        int root = s_lab[q];                                     // (pixel -> its run start -> the root, as a rule)
        while (s_lab[root] != root) root = s_lab[root];
        const size_t g = (size_t)(y0 + r) * W + x0 + lane;
        const int groot = (y0 + (root >> 6)) * W + x0 + (root & 63);
        label[g] = groot;
        if (root == q) {
            const int b = bi(q);
            CclBox e; e.minx = x0 + s_minx[b]; e.maxx = x0 + s_maxx[b]; e.maxy = y0 + s_maxy[b]; e.cnt = s_cnt[b];
            box[g] = e;
            troots[(size_t)tile * CCL_TILE_CAP + atomicAdd(&s_n, 1)] = groot;
        }
    }
    __syncthreads();
    if (tid == 0) tile_n[tile] = s_n;
}

RM_KERNEL __launch_bounds__(256) void k_ccl_seam(const unsigned long long *bits, int H, int W, int *label)
{
    constexpr int TR = CCL_TILE_ROWS;
    const int wpr = W >> 6;
    const size_t na = (size_t)((H - 1) / TR) * W;      // a thread per pixel of the rows TR, 2 TR, ... (the first row of a tile below another)
    const size_t nb = (size_t)H * wpr;                 // a thread per word of the other rows: its first and its last pixel
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < na) {
        const int y = ((int)(i / (size_t)W) + 1) * TR, x = (int)(i % (size_t)W);
        const size_t p = (size_t)y * W + x;
        if (ccl_bit(bits, p)) ccl_pixel_unions(bits, label, p, y, x, W);
        return;
    }
    if (i - na >= nb) return;
    const size_t j = i - na;
    const int y = (int)(j / (size_t)wpr), wj = (int)(j % (size_t)wpr);
    if (y > 0 && y % TR == 0) return;                  // (that row's pixels all ran above)
    const unsigned long long m = bits[j];
    if (!(m & 1ull) && !(m >> 63)) return;
    const size_t p = j << 6;
    const bool w_fg = wj > 0 && (bits[j - 1] >> 63);
    if ((m & 1ull) && w_fg) ccl_union(label, (int)p, (int)p - 1);
    if (y == 0) return;
    const unsigned long long u = bits[j - wpr];
    if ((m & 1ull) && !(u & 1ull) && !w_fg && wj > 0 && (bits[j - wpr - 1] >> 63)) ccl_union(label, (int)p, (int)(p - W - 1));
    if ((m >> 63) && !(u >> 63) && wj + 1 < wpr && (bits[j - wpr + 1] & 1ull) && !(bits[j + 1] & 1ull))
        ccl_union(label, (int)p + 63, (int)(p + 63 - W + 1));
}

RM_KERNEL __launch_bounds__(256) void k_ccl_fold(int W, const int *label, CclBox *box, const int *troots, const int *tile_n, int ntiles,
                                                  unsigned int *counters, int *roots, unsigned int cap)
{
    __shared__ int s_key[CCL_BOX_SLOTS], s_minx[CCL_BOX_SLOTS], s_maxx[CCL_BOX_SLOTS], s_maxy[CCL_BOX_SLOTS], s_cnt[CCL_BOX_SLOTS];
    __shared__ int s_off[CCL_FOLD_TILES + 1];
    __shared__ int s_roots[CCL_FOLD_TILES * CCL_TILE_CAP];
    __shared__ unsigned int s_nroots, s_base;
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * CCL_FOLD_TILES;
    if (tid == 0) {
        int o = 0;
        for (int k = 0; k < CCL_FOLD_TILES; ++k) { s_off[k] = o; o += t0 + k < ntiles ? tile_n[t0 + k] : 0; }
        s_off[CCL_FOLD_TILES] = o;
        s_nroots = 0;
    }
    if (tid < CCL_BOX_SLOTS) { s_key[tid] = -1; s_minx[tid] = 0x7fffffff; s_maxx[tid] = -1; s_maxy[tid] = -1; s_cnt[tid] = 0; }
    __syncthreads();
    const int total = s_off[CCL_FOLD_TILES];
    if (total == 0) return;                            // (uniform)
    for (int i = tid; i < total; i += 256) {
        int k = 0;
        while (i >= s_off[k + 1]) ++k;
        const int g = troots[(size_t)(t0 + k) * CCL_TILE_CAP + (i - s_off[k])];
        const int r = ccl_find(label, g);
        if (r == g) { s_roots[atomicAdd(&s_nroots, 1u)] = g; continue; }
        const CclBox mine = box[g];
        unsigned int h = ((unsigned int)r * 2654435761u) >> 27;
        int q = 0;
        for (; q < CCL_BOX_SLOTS; ++q, h = (h + 1) & (CCL_BOX_SLOTS - 1)) {
            const int seen = atomicCAS(&s_key[h], -1, r);
            if (seen == -1 || seen == r) break;
        }
        if (q < CCL_BOX_SLOTS) {
            atomicMin(&s_minx[h], mine.minx); atomicMax(&s_maxx[h], mine.maxx); atomicMax(&s_maxy[h], mine.maxy); atomicAdd(&s_cnt[h], mine.cnt);
        } else {
            ccl_box_fold(box, r, mine.minx, mine.maxx, mine.maxy);
            atomicAdd(&box[r].cnt, mine.cnt);
        }
    }
    __syncthreads();
    if (tid < CCL_BOX_SLOTS && s_key[tid] >= 0) {
        ccl_box_fold(box, s_key[tid], s_minx[tid], s_maxx[tid], s_maxy[tid]);
        atomicAdd(&box[s_key[tid]].cnt, s_cnt[tid]);
    }
    if (s_nroots == 0) return;                         // (uniform)
    if (tid == 0) s_base = atomicAdd(&counters[0], s_nroots);
    __syncthreads();
    for (unsigned int i = tid; i < s_nroots; i += 256)
        if (s_base + i < cap) roots[s_base + i] = s_roots[i];
}

}  // namespace rm
