"""Host helpers with the reference's names (reference tools.py:48-82)."""
import time

import numpy as np


def reduce_bounding_box(x, y, w, h, maximum_area):
    """Shrink (x, y, w, h) about its centre until w*h <= maximum_area (reference tools.py:48-57);
    identity at the default maximum_bounding_box_area = inf (base.py:80)."""
    area = w * h
    if area <= maximum_area:
        return x, y, w, h
    scale = np.sqrt(float(maximum_area) / float(area))
    nw, nh = w * scale, h * scale
    return (int(np.round(x + (w - nw) / 2.)), int(np.round(y + (h - nh) / 2.)),
            int(np.round(nw)), int(np.round(nh)))


class Benchmarker:
    """Named wall-clock timers (reference tools.py:60-82): add_tag / tick_start / tick_end / get_report / has_tag."""

    def __init__(self):
        self.starts = {}
        self.ticks = {}

    def add_tag(self, tag):
        self.ticks[tag] = []

    def tick_start(self, tag):
        self.starts[tag] = time.time()

    def tick_end(self, tag):
        self.ticks[tag].append(time.time() - self.starts[tag])

    def has_tag(self, tag):
        return tag in self.ticks

    def get_report(self):
        rows = ["Tag, Average Time (seconds), Iterations"]
        for tag, vals in self.ticks.items():
            rows.append("{0}, {1}, {2}".format(tag, np.mean(vals) if len(vals) else float("nan"), len(vals)))
        return "\r\n".join(rows)
