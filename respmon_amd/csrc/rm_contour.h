// host contour stage of locate() -- see rm_contour.cpp
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace rm {
struct RoiResult {
    int found;       // 0 = no contour (reference returns None, base.py:569-570)
    int x, y, w, h;  // cv2.boundingRect of the selected contour
    double area;     // cv2.contourArea of the selected contour
    int n_contours;
    long long steps; // border steps this call walked on the host (what its time is proportional to: rm_roi.hip decides from it, not from a clock)
};
// bin: H*W bytes (non-zero = foreground); row_any (nullable): per-row "has foreground" flags
int largest_external_contour(const uint8_t *bin, int H, int W, const uint32_t *row_any, RoiResult *out);
// the same on a bit-packed image: bit (p & 63) of word (p >> 6) = pixel p = y*W + x; words beyond H*W bits are not read
int largest_external_contour_bits(const uint64_t *bits, int H, int W, RoiResult *out);
// ... when the caller knows that only rows [y0, y1] can hold foreground now or held any in the previous image given to
// this thread's tracer (y1 < y0: no such row): the other rows are neither read nor cleaned
int largest_external_contour_bits_rows(const uint64_t *bits, int H, int W, int y0, int y1, RoiResult *out);
// the shortcut in front of it: 1 and the ROI when rows [y0, y1] (all the foreground there is) form ONE hole-free blob -- one run per row,
// neighbouring runs touching --, 0 otherwise; nothing is followed, nothing unpacked
int simple_shape_bits_rows(const uint64_t *bits, int H, int W, int y0, int y1, RoiResult *out);
// ... from the per-row records of k_heat_rows_u8 (first | last << 16 | runs << 32 | 1 << 48; 0 = no foreground in the row); also returns the
// rows [*y0, *y1] that hold foreground (y1 < y0: none)
int simple_shape_row_records(const uint64_t *rec, int H, int W, int *y0, int *y1, RoiResult *out);
// ... when the device has labelled the 8-connected components (rm_ccl.h): one record per component, root = its smallest
// pixel index (where the outer border starts), bounding box x = minx, y = root / W, width w1 + 1, height h1 + 1.  Only borders whose bound
// (w-1)*(h-1) can reach the best area found so far are followed, straight on the packed image.  Same result as the calls above.
struct LabelComp { int root, minx, w1, h1; };
int largest_external_contour_labelled(const uint64_t *bits, int H, int W, const LabelComp *comps, size_t n, RoiResult *out);
// ... reading the 2 x nblocks summary records of k_ccl_publish first, the full list only when they do not settle the winner
bool labelled_tops_settled(const LabelComp *tops, int nblocks, int W, size_t n, RoiResult *out);
// (area_bound_shortcut: the top record's workgroup also published 2 N - P of its component in tops[2 b + 1].w1 -- a lower bound of the
//  contour's area that can settle the winner without following any border)
int largest_external_contour_labelled_tops(const uint64_t *bits, int H, int W, const LabelComp *tops, int nblocks, const LabelComp *comps, size_t n,
                                           RoiResult *out, bool area_bound_shortcut = true);
}  // namespace rm
