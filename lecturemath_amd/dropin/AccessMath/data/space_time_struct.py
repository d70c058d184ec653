"""SpaceTimeStruct: the pickled ST3D output of step 03 (same attributes as AccessMath/data/space_time_struct.py:5-16,
consumed by steps 04/05)."""


class SpaceTimeStruct:
    def __init__(self, frame_times, frame_indices, frame_height, frame_width, group_ages, group_images, group_boundaries):
        self.frame_times = frame_times
        self.frame_indices = frame_indices
        self.width = frame_width
        self.height = frame_height
        self.cc_group_ages = group_ages
        self.cc_group_images = group_images
        self.cc_group_boundaries = group_boundaries

    def groups_in_frame_range(self, frame_start, frame_end, group_list=None):
        groups = list(self.cc_group_ages.keys()) if group_list is None else group_list
        return [g for g in groups
                if self.frame_indices[self.cc_group_ages[g][0]] <= frame_end and
                frame_start <= self.frame_indices[self.cc_group_ages[g][-1]]]

    def groups_in_space_region(self, r_min_x, r_max_x, r_min_y, r_max_y, group_list=None):
        groups = list(self.cc_group_ages.keys()) if group_list is None else group_list
        out = []
        for g in groups:
            x0, x1, y0, y1 = self.cc_group_boundaries[g]
            if x0 <= r_max_x and r_min_x <= x1 and y0 <= r_max_y and r_min_y <= y1:
                out.append(g)
        return out
