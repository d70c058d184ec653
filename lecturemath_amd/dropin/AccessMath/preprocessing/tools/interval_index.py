"""IntervalIndex with the reference's attribute layout (tools/interval_index.py:15-99): intervals[start][end] -> list of
payloads.  The MI355X path does its box joins on the device (lm_k_match_scan / lm_k_selfjoin); this class only exists so
that a pickled CCStabilityEstimator carries the same object graph, and for callers that use it directly."""


class Interval:
    def __init__(self, start, end, data):
        self.start, self.end, self.data = start, end, data

    def __eq__(self, other):
        return (self.start, self.end, self.data) == (other.start, other.end, other.data)


class IntervalIndex:
    def __init__(self, only_data=False):
        self.intervals = {}
        self.only_data = only_data

    def _payload(self, start, end, data):
        return data if self.only_data else Interval(start, end, data)

    def add(self, start, end, data):
        for pos in range(len(self.intervals), end + 1):
            self.intervals[pos] = {}
        self.intervals[start].setdefault(end, []).append(self._payload(start, end, data))

    def remove(self, start, end, data):
        self.intervals[start][end].remove(self._payload(start, end, data))

    def find_matches(self, other):
        """All (own payload, other payload) pairs whose half-open intervals overlap (sweep over start positions)."""
        out = []
        mine, theirs = [], []          # currently open: (end, payload)
        for pos in range(min(len(self.intervals), len(other.intervals))):
            mine = [(e, p) for e, p in mine if e != pos]
            theirs = [(e, p) for e, p in theirs if e != pos]
            new_mine = [(e, p) for e, lst in self.intervals[pos].items() for p in lst]
            new_theirs = [(e, p) for e, lst in other.intervals[pos].items() for p in lst]
            out += [(a, b) for _, a in mine for _, b in new_theirs]
            out += [(a, b) for _, a in new_mine for _, b in theirs]
            out += [(a, b) for _, a in new_mine for _, b in new_theirs]
            mine += new_mine
            theirs += new_theirs
        return out
