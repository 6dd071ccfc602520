// host contour stage of locate() -- see rm_contour.cpp
#pragma once
#include <stdint.h>

namespace rm {
struct RoiResult {
    int found;       // 0 = no contour (reference returns None, base.py:569-570)
    int x, y, w, h;  // cv2.boundingRect of the selected contour
    double area;     // cv2.contourArea of the selected contour
    int n_contours;
};
// bin: H*W bytes (non-zero = foreground); row_any (nullable): per-row "has foreground" flags
int largest_external_contour(const uint8_t *bin, int H, int W, const uint32_t *row_any, RoiResult *out);
// the same on a bit-packed image: bit (p & 63) of word (p >> 6) = pixel p = y*W + x; words beyond H*W bits are not read
int largest_external_contour_bits(const uint64_t *bits, int H, int W, RoiResult *out);
}  // namespace rm
