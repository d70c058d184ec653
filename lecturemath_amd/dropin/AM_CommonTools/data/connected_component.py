"""ConnectedComponent record for the MI355X-native path.

Same attribute names (they are the pickle schema) and the same hot-path methods as the reference's
AM_CommonTools/data/connected_component.py (ctor :25-41, getOverlapFMeasure :202-250, getBoxArea :46-47,
getOverlapArea :54-67, getWidth/getHeight :283-287).  Instances are produced from device records by the Labeler /
CCStabilityEstimator of this package; the methods below are host conveniences for single objects -- batched overlap
arithmetic runs in liblecturemath_hip.so.
"""
import numpy as np


class ConnectedComponent:
    NormalizedSize = 128
    MinScalingSize = 10

    def __init__(self, cc_id, min_x, max_x, min_y, max_y, size, img):
        self.cc_id = cc_id
        self.min_x, self.min_y, self.max_x, self.max_y = min_x, min_y, max_x, max_y
        self.size = size
        self.img = img            # uint8 0/255 mask of the box
        self.normalized = None
        self.start_time = None
        self.end_time = None
        self.next_cc = None
        self.prev_cc = None

    def getBoundingBox(self):
        return (self.min_x, self.max_x), (self.min_y, self.max_y)

    def getWidth(self):
        return self.max_x - self.min_x + 1

    def getHeight(self):
        return self.max_y - self.min_y + 1

    def getBoxArea(self):
        return self.getWidth() * self.getHeight()

    def getCenter(self):
        return (self.min_x + self.max_x) / 2.0, (self.min_y + self.max_y) / 2.0

    def _boxes_touch(self, other):
        return (self.min_x <= other.max_x and other.min_x <= self.max_x and
                self.min_y <= other.max_y and other.min_y <= self.max_y)

    def getOverlapArea(self, other):
        if not self._boxes_touch(other):
            return 0.0
        return ((min(self.max_x, other.max_x) - max(self.min_x, other.min_x) + 1) *
                (min(self.max_y, other.max_y) - max(self.min_y, other.min_y) + 1))

    def getOverlapFMeasure(self, other, verbose=False, single_score=True):
        match = 0
        if self._boxes_touch(other):
            x0, x1 = max(self.min_x, other.min_x), min(self.max_x, other.max_x)
            y0, y1 = max(self.min_y, other.min_y), min(self.max_y, other.max_y)
            mine = self.img[y0 - self.min_y:y1 - self.min_y + 1, x0 - self.min_x:x1 - self.min_x + 1]
            theirs = other.img[y0 - other.min_y:y1 - other.min_y + 1, x0 - other.min_x:x1 - other.min_x + 1]
            match = int(np.count_nonzero(np.bitwise_and(mine, theirs)))
        elif single_score:
            return 0.0
        else:
            return 0.0, 0.0
        if single_score:
            return (2.0 * match) / float(self.size + other.size)
        return match / float(self.size), match / float(other.size)

    def __str__(self):
        return "ConnectedComponent -> Id = %s\n -> X : [%s, %s] \n -> Y : [%s, %s]" % (
            self.cc_id, self.min_x, self.max_x, self.min_y, self.max_y)
